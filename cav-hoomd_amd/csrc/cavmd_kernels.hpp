// cavmd_kernels.hpp -- CDNA4 (gfx950) device code of the cavity-force path (umbrella header).
//
// An evaluation is ONE launch for 1024 < N <~ 2.4e6 (cavmd_persistent_kernel.hpp: both phases below in one grid of
// co-resident blocks, charges kept in LDS between them), one single-block launch up to 1024 particles, and otherwise
// TWO launches, all bandwidth- or latency-bound (no MFMA: ~60 VALU operations per 92 bytes):
//
//   dipole_partials_kernel       streams pos (32 B) + charge (8 B) + image (12 B) per particle, unwraps, forms the addends
//                                c_i * r_i exactly as the reference does (one rounding per operation, no FMA), accumulates
//                                them per lane in double-double (TwoSum), reduces lane -> wave (DPP) -> block (LDS) in a
//                                fixed order and writes ONE partial per block.  Also finds the photon (minimum index whose
//                                type tag is L).  reference: findPhotonParticle + computeUnwrappedPositions +
//                                computeDipoleMoment, src/CavityForceCompute.cc:73-129.
//   force_map_aos_fused_kernel   prologue: every block folds the <= 256 partials in the same fixed order, unwraps the photon
//                                and evaluates energies, Dq and the photon force with the reference's operator association
//                                (src/CavityForceCompute.cc:169-183, 203-207); block 0 publishes cavmd_result.  Body: streams
//                                charge (8 B) and writes force (32 B) per particle as dense 16-byte chunks: even chunk =
//                                (Fx, Fy), odd chunk = (Fz, w) = (0, 0); the photon's two chunks carry F_L.  Every entry of
//                                the force array is written, so the reference's memset pass (:145) is not needed (:183-207).
//   (finalize_kernel + force_map_aos_kernel: the same work as three launches, kept for A/B measurements;
//    StridedInput / force_map_strided_kernel: the snapshot layouts of the hoomd.md.force.Custom surface.)
//
// No atomics on floating-point data, no dependence on dispatch order: results are bit-reproducible for
// a given (N, launch geometry).  The kernel boundary is the only inter-workgroup hand-off.
//
// Files: cavmd_reduce.hpp (double-double arithmetic, DPP, block trees), cavmd_force_kernels.hpp (the force path),
//        cavmd_persistent_kernel.hpp (the single-launch evaluation), cavmd_observable_kernels.hpp (rows f2-f4).
#pragma once

#include "cavmd_reduce.hpp"
#include "cavmd_force_kernels.hpp"
#include "cavmd_persistent_kernel.hpp"
#include "cavmd_observable_kernels.hpp"
