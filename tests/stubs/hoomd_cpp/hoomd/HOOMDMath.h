// STAND-IN for hoomd/HOOMDMath.h -- NOT HOOMD-blue, and no evidence of compatibility with it.
//
// The build/test image has no HOOMD-blue.  tests/stubs/hoomd_cpp/ declares the handful of HOOMD-blue names that
// cav-hoomd_amd/csrc/hoomd_shim/CavityForceComputeHIP.{h,cc} touches, with the semantics that file relies on (double
// precision Scalar types; GlobalArray / ArrayHandle scopes; ParticleData getters; ForceCompute::compute -> computeForces),
// so that the shim is COMPILED by every test run and EXECUTED on the GPU box (tests/test_hoomd_cpp_shim.py): typos, signature
// drift, handle scopes, the energy cache and the N = 0 path are exercised.  What this proves about a real HOOMD-blue: nothing.
// Row f1 of SURVEY.md section 8 stays "blocked: no HOOMD in the image".
#ifndef STANDIN_HOOMD_MATH_H_
#define STANDIN_HOOMD_MATH_H_

#include <hip/hip_runtime.h> // double4, double3, int3 -- what HOOMD-blue's GPU builds get from the HIP headers as well

namespace hoomd
    {
typedef double Scalar; // HOOMD_LONGREAL_SIZE = 64, HOOMD-blue's default build
typedef double3 Scalar3;
typedef double4 Scalar4;
    } // namespace hoomd
#endif
