// CavityForceComputeHIP.cc -- see the header.  Never built against HOOMD-blue in this repository's image (it has no HOOMD-blue
// headers); compiled and run there only against the STAND-IN declarations of tests/stubs/hoomd_cpp (tests/test_hoomd_cpp_shim.py:
// handle scopes, energy cache, N = 0, workspace growth), which says nothing about a real HOOMD-blue.
//
// What the reference's GPU class does per step and this one does not (src/CavityForceComputeGPU.cc:129-226):
// four hipMemsets, one H2D and two blocking D2H copies, hipDeviceSynchronize, a host ArrayHandle on the force AND
// position arrays (a full device->host migration under GlobalArray) and an O(N) host scan.  Here: acquire four
// device handles, enqueue one kernel (two above ~2.4e6 particles, one single-block kernel up to 1024) on the null stream
// (HOOMD-blue's stream), release.
// Energies are fetched lazily, once per EVALUATION, when a getter is called (EnergyTracker polls them every step).
#include "CavityForceComputeHIP.h"

#include <stdexcept>
#include <string>

namespace hoomd
    {
namespace cavitymd
    {
static_assert(sizeof(Scalar) == 8, "libcavmd implements the double-precision (HOOMD_LONGREAL_SIZE=64) layouts");
static_assert(sizeof(Scalar4) == sizeof(cavmd_double4), "Scalar4 layout");
static_assert(sizeof(int3) == sizeof(cavmd_int3), "int3 layout");

static void check(int status, const char* where)
    {
    if (status == CAVMD_ERR_SYNC_TIMEOUT)
        {
        // Reported by the call AFTER the evaluation it concerns: a single-launch evaluation whose workgroups were not resident
        // together AND that its last workgroup could not complete (the rare outcome; a starved evaluation is normally completed,
        // late but bit-identical, and reports nothing).  libcavmd itself recovers -- the workspace is on two launches from now
        // on and the next call works -- but the forces of THAT step were NaN and HOOMD-blue has integrated them by now, so the
        // run cannot be continued; re-enqueueing here would only hide where it went wrong.
        throw std::runtime_error(std::string("cavitymd (HIP): ") + where
                                 + ": the previous step's cavity-force evaluation was starved of compute units and could not be "
                                   "completed (its forces were NaN and have been integrated); restart from the last checkpoint with "
                                   "CAVMD_PERSISTENT=0 in the environment if the GPU is shared with other processes");
        }
    if (status != CAVMD_OK)
        throw std::runtime_error(std::string("cavitymd (HIP): ") + where + ": " + cavmd_error_string(status));
    }

CavityForceComputeHIP::CavityForceComputeHIP(std::shared_ptr<SystemDefinition> sysdef,
                                             Scalar omegac,
                                             Scalar couplstr,
                                             Scalar phmass)
    : ForceCompute(sysdef), m_params(cavmd_make_params(omegac, couplstr, phmass))
    {
    if (!m_exec_conf->isCUDAEnabled())
        throw std::runtime_error("cavitymd (HIP): a GPU execution configuration is required; there is no CPU path");
    ensureWorkspace(m_pdata->getMaxN());
    }

CavityForceComputeHIP::~CavityForceComputeHIP()
    {
    cavmd_destroy(m_ws);
    }

void CavityForceComputeHIP::ensureWorkspace(size_t n)
    {
    if (m_ws && n <= m_capacity)
        return;
    cavmd_destroy(m_ws);
    m_ws = nullptr;
    m_capacity = n > 0 ? n : 1;
    check(cavmd_create(-1, m_capacity, &m_ws), "cavmd_create");
    }

void CavityForceComputeHIP::setParams(Scalar omegac, Scalar couplstr, Scalar phmass)
    {
    m_params = cavmd_make_params(omegac, couplstr, phmass);
    }

pybind11::dict CavityForceComputeHIP::getParams()
    {
    pybind11::dict v;
    v["omegac"] = m_params.omegac;
    v["couplstr"] = m_params.couplstr;
    v["K"] = m_params.K;
    v["phmass"] = m_params.phmass;
    return v;
    }

void CavityForceComputeHIP::computeForces(uint64_t timestep)
    {
    const unsigned int N = m_pdata->getN();
    m_eval_seq += 1; // the energies cached below belong to the PREVIOUS evaluation from here on
    if (N == 0)
        {
        // nothing to compute: the reference zeroes its energies and returns (src/CavityForceCompute.cc:148-156)
        m_energy[0] = m_energy[1] = m_energy[2] = 0.0;
        m_energy_seq = m_eval_seq;
        return;
        }
    ensureWorkspace(N);

    // the CPU reference lets getTypeByName throw when no type is named 'L'; its GPU class zeroes the energies
    // instead (src/CavityForceComputeGPU.cc:114-123).  -1 matches no particle: forces and energies become zero.
    int L_typeid = -1;
    try
        {
        L_typeid = (int)m_pdata->getTypeByName("L");
        }
    catch (...)
        {
        }

    const BoxDim box = m_pdata->getGlobalBox();
    const Scalar3 L = box.getL();

        {
        ArrayHandle<Scalar4> d_pos(m_pdata->getPositions(), access_location::device, access_mode::read);
        ArrayHandle<Scalar> d_charge(m_pdata->getCharges(), access_location::device, access_mode::read);
        ArrayHandle<int3> d_image(m_pdata->getImages(), access_location::device, access_mode::read);
        ArrayHandle<Scalar4> d_force(m_force, access_location::device, access_mode::overwrite);
        // m_virial and m_torque are not touched: the reference leaves them zero as well.
        check(cavmd_compute_hoomd(m_ws,
                                  nullptr, // HOOMD-blue works on the null stream
                                  N,
                                  reinterpret_cast<const cavmd_double4*>(d_pos.data),
                                  d_charge.data,
                                  reinterpret_cast<const cavmd_int3*>(d_image.data),
                                  L.x,
                                  L.y,
                                  L.z,
                                  L_typeid,
                                  &m_params,
                                  reinterpret_cast<cavmd_double4*>(d_force.data)),
              "cavmd_compute_hoomd");
        }
    (void)timestep;
    }

// Keyed on the evaluation counter, not on the timestep: setParams(...) followed by sim.run(0) recomputes at the SAME
// timestep and must not be answered from the cache.
void CavityForceComputeHIP::fetchEnergies()
    {
    if (m_energy_seq == m_eval_seq)
        return;
    check(cavmd_energies(m_ws, m_energy), "cavmd_energies");
    m_energy_seq = m_eval_seq;
    }

Scalar CavityForceComputeHIP::getHarmonicEnergy()
    {
    fetchEnergies();
    return m_energy[0];
    }

Scalar CavityForceComputeHIP::getCouplingEnergy()
    {
    fetchEnergies();
    return m_energy[1];
    }

Scalar CavityForceComputeHIP::getDipoleSelfEnergy()
    {
    fetchEnergies();
    return m_energy[2];
    }

namespace detail
    {
void export_CavityForceComputeHIP(pybind11::module& m)
    {
    pybind11::class_<CavityForceComputeHIP, ForceCompute, std::shared_ptr<CavityForceComputeHIP>>(
        m,
        "CavityForceComputeHIP")
        .def(pybind11::init<std::shared_ptr<SystemDefinition>, Scalar, Scalar, Scalar>(),
             pybind11::arg("sysdef"),
             pybind11::arg("omegac"),
             pybind11::arg("couplstr"),
             pybind11::arg("phmass") = 1.0)
        .def("setParams", &CavityForceComputeHIP::setParams)
        .def("getParams", &CavityForceComputeHIP::getParams)
        .def("getHarmonicEnergy", &CavityForceComputeHIP::getHarmonicEnergy)
        .def("getCouplingEnergy", &CavityForceComputeHIP::getCouplingEnergy)
        .def("getDipoleSelfEnergy", &CavityForceComputeHIP::getDipoleSelfEnergy);
    }
    } // namespace detail
    } // namespace cavitymd
    } // namespace hoomd
