// _cavitymd -- pybind11 flavour of the shim over the C ABI (include/cavmd.h), stand-alone (no HOOMD headers).
//
// Mirrors the Python-visible surface of the reference's pybind11 module (src/cavitymd/module.cc:27-33 and the exports
// at src/CavityForceCompute.cc:212-224, src/CavityForceComputeGPU.cc:257-264): a class with setParams / getParams /
// getHarmonicEnergy / getCouplingEnergy / getDipoleSelfEnergy whose computeForces runs the HIP kernels.  Because there is
// no SystemDefinition without HOOMD, the constructor takes a capacity and computeForces takes raw DEVICE pointers
// (integers, e.g. tensor.data_ptr()).  The HOOMD-aware twin is csrc/hoomd_shim/ (compile-gated).
// Nothing is computed on the host here: every method forwards to libcavmd and turns a non-zero status into
// std::runtime_error (-> Python RuntimeError), as the reference's classes do (src/CavityForceComputeGPU.cc:106-109).
#include <pybind11/pybind11.h>

#include <cstdint>
#include <stdexcept>
#include <string>

#include "cavmd.h"

namespace py = pybind11;

namespace
{
void check(int status, const char* where)
{
    if (status != CAVMD_OK)
        throw std::runtime_error(std::string("libcavmd ") + where + ": [" + std::to_string(status) + "] "
                                 + cavmd_error_string(status));
}

class CavityForceComputeHIP
{
public:
    CavityForceComputeHIP(size_t max_N, double omegac, double couplstr, double phmass, int device)
        : m_params(cavmd_make_params(omegac, couplstr, phmass))
    {
        check(cavmd_create(device, max_N, &m_ws), "cavmd_create");
    }
    ~CavityForceComputeHIP() { cavmd_destroy(m_ws); }
    CavityForceComputeHIP(const CavityForceComputeHIP&) = delete;
    CavityForceComputeHIP& operator=(const CavityForceComputeHIP&) = delete;

    void setParams(double omegac, double couplstr, double phmass) { m_params = cavmd_make_params(omegac, couplstr, phmass); }

    py::dict getParams() const
    {
        py::dict v;
        v["omegac"] = m_params.omegac;
        v["couplstr"] = m_params.couplstr;
        v["K"] = m_params.K;
        v["phmass"] = m_params.phmass;
        return v;
    }

    // HOOMD-native AoS device arrays: pos/force Scalar4, charge Scalar, image int3
    void computeForces(std::uintptr_t pos, std::uintptr_t charge, std::uintptr_t image, size_t N, double Lx, double Ly,
                       double Lz, int L_typeid, std::uintptr_t force, std::uintptr_t stream)
    {
        check(cavmd_compute_hoomd(m_ws, reinterpret_cast<void*>(stream), N, reinterpret_cast<const cavmd_double4*>(pos),
                                  reinterpret_cast<const double*>(charge), reinterpret_cast<const cavmd_int3*>(image), Lx,
                                  Ly, Lz, L_typeid, &m_params, reinterpret_cast<cavmd_double4*>(force)),
              "cavmd_compute_hoomd");
    }

    double energy(int k)
    {
        double e[3];
        check(cavmd_energies(m_ws, e), "cavmd_energies");
        return e[k];
    }
    double getHarmonicEnergy() { return energy(0); }
    double getCouplingEnergy() { return energy(1); }
    double getDipoleSelfEnergy() { return energy(2); }

    py::tuple getEnergies()
    {
        double e[3];
        check(cavmd_energies(m_ws, e), "cavmd_energies");
        return py::make_tuple(e[0], e[1], e[2]);
    }

    py::dict getResult()
    {
        cavmd_result r;
        check(cavmd_result_read(m_ws, &r), "cavmd_result_read");
        py::dict v;
        v["dipole"] = py::make_tuple(r.dipole[0], r.dipole[1], r.dipole[2]);
        v["total_dipole"] = py::make_tuple(r.total_dipole[0], r.total_dipole[1], r.total_dipole[2]);
        v["q"] = py::make_tuple(r.q[0], r.q[1], r.q[2]);
        v["Dq"] = py::make_tuple(r.Dq[0], r.Dq[1]);
        v["energy"] = py::make_tuple(r.energy[0], r.energy[1], r.energy[2]);
        v["photon_force"] = py::make_tuple(r.photon_force[0], r.photon_force[1], r.photon_force[2]);
        v["photon_idx"] = r.photon_idx;
        v["n_photon_typed"] = r.n_photon_typed;
        v["n_particles"] = r.n_particles;
        v["sequence"] = r.sequence;
        return v;
    }

private:
    cavmd_params m_params;
    cavmd_workspace* m_ws = nullptr;
};

// Low-overhead call on a workspace that someone else owns (cavitymd._capi.Workspace creates it through ctypes):
// one pybind11 call costs ~1.5 us less than the same call through ctypes, which matters at the reference's N = 501,
// where the whole evaluation takes 5 us on the GPU.
void compute_hoomd(std::uintptr_t ws, std::uintptr_t stream, size_t N, std::uintptr_t pos, std::uintptr_t charge,
                   std::uintptr_t image, double Lx, double Ly, double Lz, int L_typeid, double omegac, double couplstr,
                   double K, double phmass, std::uintptr_t force)
{
    cavmd_params p;
    p.omegac = omegac;
    p.couplstr = couplstr;
    p.K = K;
    p.phmass = phmass;
    check(cavmd_compute_hoomd(reinterpret_cast<cavmd_workspace*>(ws), reinterpret_cast<void*>(stream), N,
                              reinterpret_cast<const cavmd_double4*>(pos), reinterpret_cast<const double*>(charge),
                              reinterpret_cast<const cavmd_int3*>(image), Lx, Ly, Lz, L_typeid, &p,
                              reinterpret_cast<cavmd_double4*>(force)),
          "cavmd_compute_hoomd");
}
} // namespace

PYBIND11_MODULE(_cavitymd, m)
{
    m.doc() = "pybind11 shim over libcavmd (HIP cavity force for MI355X); see include/cavmd.h";
    m.def("version", &cavmd_version);
    m.def("compute_hoomd", &compute_hoomd, py::arg("ws"), py::arg("stream"), py::arg("N"), py::arg("pos"), py::arg("charge"),
          py::arg("image"), py::arg("Lx"), py::arg("Ly"), py::arg("Lz"), py::arg("L_typeid"), py::arg("omegac"),
          py::arg("couplstr"), py::arg("K"), py::arg("phmass"), py::arg("force"));
    py::class_<CavityForceComputeHIP>(m, "CavityForceComputeHIP")
        .def(py::init<size_t, double, double, double, int>(), py::arg("max_N"), py::arg("omegac"), py::arg("couplstr"),
             py::arg("phmass") = 1.0, py::arg("device") = -1)
        .def("setParams", &CavityForceComputeHIP::setParams, py::arg("omegac"), py::arg("couplstr"), py::arg("phmass") = 1.0)
        .def("getParams", &CavityForceComputeHIP::getParams)
        .def("computeForces", &CavityForceComputeHIP::computeForces, py::arg("pos"), py::arg("charge"), py::arg("image"),
             py::arg("N"), py::arg("Lx"), py::arg("Ly"), py::arg("Lz"), py::arg("L_typeid"), py::arg("force"),
             py::arg("stream") = 0)
        .def("getHarmonicEnergy", &CavityForceComputeHIP::getHarmonicEnergy)
        .def("getCouplingEnergy", &CavityForceComputeHIP::getCouplingEnergy)
        .def("getDipoleSelfEnergy", &CavityForceComputeHIP::getDipoleSelfEnergy)
        .def("getEnergies", &CavityForceComputeHIP::getEnergies)
        .def("getResult", &CavityForceComputeHIP::getResult);
}
