// CavityForceComputeHIP.h -- HOOMD-blue 4.x ForceCompute whose computeForces() enqueues libcavmd's HIP kernels.
//
// COMPILE-GATED: built only when CMake finds HOOMD-blue (find_package(HOOMD)); the build/test image of this
// repository has no HOOMD headers, so this file has never met HOOMD-blue.  There it is compiled and executed against the
// stand-in declarations of tests/stubs/hoomd_cpp only (tests/test_hoomd_cpp_shim.py), which keeps it from rotting and
// exercises its own logic -- and proves nothing about the real API.  It is the C++ half of the drop-in:
// it exports the same Python-visible names as the reference's _cavitymd.CavityForceComputeGPU
// (reference: src/CavityForceComputeGPU.h:30-56, src/CavityForceComputeGPU.cc:257-264) and therefore slots into
// the attach ladder of hoomd.cavitymd.CavityForce (reference: src/cavitymd/forces.py:97-173) as the first rung.
#ifndef CAVITY_FORCE_COMPUTE_HIP_H_
#define CAVITY_FORCE_COMPUTE_HIP_H_

#include "hoomd/ForceCompute.h"
#include "hoomd/HOOMDMath.h"

#include <memory>
#include <pybind11/pybind11.h>

#include "cavmd.h"

namespace hoomd
    {
namespace cavitymd
    {
class PYBIND11_EXPORT CavityForceComputeHIP : public ForceCompute
    {
    public:
    CavityForceComputeHIP(std::shared_ptr<SystemDefinition> sysdef,
                          Scalar omegac,
                          Scalar couplstr,
                          Scalar phmass = Scalar(1.0));
    virtual ~CavityForceComputeHIP();

    void setParams(Scalar omegac, Scalar couplstr, Scalar phmass = Scalar(1.0));
    pybind11::dict getParams();
    Scalar getHarmonicEnergy();
    Scalar getCouplingEnergy();
    Scalar getDipoleSelfEnergy();

    protected:
    void computeForces(uint64_t timestep) override;

    private:
    void fetchEnergies();
    void ensureWorkspace(size_t n);

    cavmd_params m_params;
    cavmd_workspace* m_ws = nullptr;
    size_t m_capacity = 0;
    uint64_t m_eval_seq = 0;   //!< evaluations enqueued so far (computeForces calls)
    uint64_t m_energy_seq = 0; //!< evaluation the cached energies belong to (0 = none: the constructor's zeros)
    double m_energy[3] = {0.0, 0.0, 0.0};
    };

namespace detail
    {
void export_CavityForceComputeHIP(pybind11::module& m);
    }
    } // namespace cavitymd
    } // namespace hoomd
#endif
