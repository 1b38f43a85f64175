// cavmd_observable_kernels.hpp -- kernels of the observables next to the force path:
// density field rho(k), cavity-mode energies, sum |F_i| / m_i.  Overview: cavmd_kernels.hpp.
#pragma once

#include "cavmd_reduce.hpp"

#pragma clang fp contract(off)

// =====================================================================================================================
// Observable next to the force path: density field rho(k) = sum_j exp(i k.r_j) over the WRAPPED
// positions of all particles, for a set of wavevectors (reference: compute_density_field, src/cavitymd/analysis.py:34-47,
// a Python loop over 50 wavevectors of numpy cos/sin over all particles).
//
// Mapping: LANE = WAVEVECTOR.  A wave loads 64 particles' positions with one coalesced round (lane = particle), then
// walks them one by one: the particle's coordinates are broadcast with v_readlane (SGPR operands), every lane forms
// k_lane . r = (x kx + y ky) + z kz and adds cos / sin to its own two accumulators.  The particle loop therefore has no
// cross-lane traffic and no per-lane register pressure (2 accumulators), whatever the number of wavevectors; more than
// 64 wavevectors are handled in chunks of 64 (blockIdx.y).  The kernel is bound by fp64 transcendental throughput
// (~N * n_k sincos), not by memory: positions are 24 N bytes per chunk.
// =====================================================================================================================
namespace cavmd
{
__device__ __forceinline__ double readlane_f64(double v, int src_lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}

// sin and cos of one argument for |x| < 1e8: n = rint(x * 2/pi), r = x - n * pi/2 by a two-constant Cody-Waite step
// with explicit FMAs (pi/2 = P1 + P2 to 107 bits; error ~1 ulp of r for |n| < 2^27), then the fdlibm kernel
// polynomials on |r| <= pi/4 (S1..S6, C1..C6 of __kernel_sin / __kernel_cos, < 1 ulp each) and the quadrant swap.
// ~27 fp64 operations against ~100 for the device library's sincos, whose Payne-Hanek path is kept for huge arguments.
// The constants arrive as a KERNEL ARGUMENT, i.e. in SGPRs: with literal constants hipcc materialises every
// coefficient in a VGPR and emits v_mov_b64 + v_fmac_f64 pairs (18 extra moves per call); an SGPR addend can only be
// the third operand of a three-address v_fma_f64.
struct SinCosCoef
{
    double inv_pio2, p1, p2;
    double s1, s2, s3, s4, s5, s6;
    double c1, c2, c3, c4, c5, c6;
};
inline SinCosCoef make_sincos_coef()
{
    SinCosCoef k;
    k.inv_pio2 = 6.36619772367581382433e-01;
    k.p1 = 1.57079632679489655800e+00; // double(pi/2)
    k.p2 = 6.12323399573676603587e-17; // pi/2 - p1
    k.s1 = -1.66666666666666324348e-01; k.s2 = 8.33333333332248946124e-03; k.s3 = -1.98412698298579493134e-04;
    k.s4 = 2.75573137070700676789e-06;  k.s5 = -2.50507602534068634195e-08; k.s6 = 1.58969099521155010221e-10;
    k.c1 = 4.16666666666666019037e-02;  k.c2 = -1.38888888888741095749e-03; k.c3 = 2.48015872894767294178e-05;
    k.c4 = -2.75573143513906633035e-07; k.c5 = 2.08757232129817482790e-09;  k.c6 = -1.13596475577881948265e-11;
    return k;
}
__device__ __forceinline__ void sincos_reduced(const SinCosCoef& k, double x, double& s, double& c)
{
    const double n = __builtin_rint(x * k.inv_pio2);
    double r = __builtin_fma(-n, k.p1, x);
    r = __builtin_fma(-n, k.p2, r);
    const double z = r * r;
    // sin(r) = r + r^3 (S1 + z (S2 + z (S3 + z (S4 + z (S5 + z S6)))))
    double ps = __builtin_fma(z, k.s6, k.s5);
    ps = __builtin_fma(z, ps, k.s4);
    ps = __builtin_fma(z, ps, k.s3);
    ps = __builtin_fma(z, ps, k.s2);
    ps = __builtin_fma(z, ps, k.s1);
    const double sr = __builtin_fma(z * r, ps, r);
    // cos(r) = 1 - (z/2 - z^2 (C1 + z (C2 + z (C3 + z (C4 + z (C5 + z C6))))))
    double pc = __builtin_fma(z, k.c6, k.c5);
    pc = __builtin_fma(z, pc, k.c4);
    pc = __builtin_fma(z, pc, k.c3);
    pc = __builtin_fma(z, pc, k.c2);
    pc = __builtin_fma(z, pc, k.c1);
    const double cr = 1.0 - __builtin_fma(-z * z, pc, 0.5 * z);
    const int q = (int)n;
    // quadrant: odd q swaps sin and cos; the signs go straight into the sign bit of the high word
    const bool swap = q & 1;
    const double ss = swap ? cr : sr;
    const double cc = swap ? sr : cr;
    s = __hiloint2double(__double2hiint(ss) ^ ((q & 2) << 30), __double2loint(ss));
    c = __hiloint2double(__double2hiint(cc) ^ (((q + 1) & 2) << 30), __double2loint(cc));
}

// part layout: [chunk][block][2][64] doubles
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void density_partials_kernel(const char* __restrict__ pos, size_t pos_stride, unsigned N,
                                                                 const double* __restrict__ kvec, unsigned n_k,
                                                                 SinCosCoef coef, double* __restrict__ part)
{
    constexpr int NW = BLOCK / kWave;
    __shared__ double s_acc[NW][2][kWave];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    const unsigned chunk = blockIdx.y;
    const unsigned k = chunk * kWave + lane;
    const bool active = k < n_k;
    const double kx = active ? kvec[3 * k + 0] : 0.0;
    const double ky = active ? kvec[3 * k + 1] : 0.0;
    const double kz = active ? kvec[3 * k + 2] : 0.0;

    double re = 0.0, im = 0.0;
    const unsigned ntiles = (N + kWave - 1) / kWave;
    // wave index made provably uniform so that the particle addresses below are SGPR values (scalar loads)
    const unsigned wave_u = (unsigned)__builtin_amdgcn_readfirstlane(wave);
    const unsigned gw = blockIdx.x * NW + wave_u, GW = gridDim.x * NW;
    const double ksum = (fabs(kx) + fabs(ky)) + fabs(kz);
    for (unsigned tile = gw; tile < ntiles; tile += GW)
    {
        // lane = particle: one coalesced round, used for the tile's coordinate bound and for the rare slow path
        const size_t i = (size_t)tile * kWave + lane;
        double px = 0.0, py = 0.0, pz = 0.0;
        if (i < N)
        {
            const double* p = reinterpret_cast<const double*>(pos + i * pos_stride);
            px = p[0];
            py = p[1];
            pz = p[2];
        }
        const unsigned left = N - tile * kWave;
        const int cnt = left < (unsigned)kWave ? (int)left : kWave; // wave-uniform
        // |k . r| <= (|kx| + |ky| + |kz|) * max|coordinate|: ONE range check per tile instead of one per particle
        double m = fmax(fmax(fabs(px), fabs(py)), fabs(pz));
        m = fmax(m, dpp_f64<0xB1, 0xF>(m, 0.0));
        m = fmax(m, dpp_f64<0x4E, 0xF>(m, 0.0));
        m = fmax(m, dpp_f64<0x124, 0xF>(m, 0.0));
        m = fmax(m, dpp_f64<0x128, 0xF>(m, 0.0));
        m = fmax(m, __shfl_xor(m, 16, kWave));
        m = fmax(m, __shfl_xor(m, 32, kWave));
        const bool all_finite = !__any(!(fabs(px) < 1.0e300) || !(fabs(py) < 1.0e300) || !(fabs(pz) < 1.0e300));
        if (all_finite && !__any(!(ksum * m < 1.0e8)))
        {
            // fast path: every |k . r| of this tile is below 1e8.  The coordinates come through the scalar cache
            // (s_load on a uniform address): broadcasting them costs no VALU issue slots.
            const char* base = pos + (size_t)tile * kWave * pos_stride;
            constexpr int PU = 4; // particles per iteration: amortises the loop and lets independent sincos chains interleave
            int j = 0;
            for (; j + PU <= cnt; j += PU)
            {
                double x[PU], y[PU], z[PU];
#pragma unroll
                for (int u = 0; u < PU; ++u)
                {
                    const double* p = reinterpret_cast<const double*>(base + (size_t)(j + u) * pos_stride);
                    x[u] = p[0];
                    y[u] = p[1];
                    z[u] = p[2];
                }
#pragma unroll
                for (int u = 0; u < PU; ++u)
                {
                    double sn, cs;
                    sincos_reduced(coef, (x[u] * kx + y[u] * ky) + z[u] * kz, sn, cs);
                    re += cs;
                    im += sn;
                }
            }
            for (; j < cnt; ++j)
            {
                const double* p0 = reinterpret_cast<const double*>(base + (size_t)j * pos_stride);
                double s0, c0;
                sincos_reduced(coef, (p0[0] * kx + p0[1] * ky) + p0[2] * kz, s0, c0);
                re += c0;
                im += s0;
            }
        }
        else
        {
            for (int j = 0; j < cnt; ++j)
            {
                const double x = readlane_f64(px, j), y = readlane_f64(py, j), z = readlane_f64(pz, j);
                const double kr = (x * kx + y * ky) + z * kz;
                double sn, cs;
                if (__any(!(fabs(kr) < 1.0e8))) // wave-uniform; also catches NaN/Inf
                    sincos(kr, &sn, &cs);
                else
                    sincos_reduced(coef, kr, sn, cs);
                re += cs;
                im += sn;
            }
        }
    }
    s_acc[wave][0][lane] = re;
    s_acc[wave][1][lane] = im;
    __syncthreads();
    if (wave == 0)
    {
#pragma unroll
        for (int w = 1; w < NW; ++w)
        {
            re += s_acc[w][0][lane];
            im += s_acc[w][1][lane];
        }
        double* out = part + ((size_t)chunk * gridDim.x + blockIdx.x) * 2 * kWave;
        out[lane] = re;
        out[kWave + lane] = im;
    }
}

// ---- second mapping: LANE = PARTICLE ------------------------------------------------------------------------------------
// With lane = wavevector a wave runs at n_k / 64 of its rate when n_k is not a multiple of 64 (the reference's default of
// 50 wavevectors leaves 14 lanes idle: 22 % of a kernel that is bound by fp64 instruction issue).  Here a lane owns one
// particle of a 64-particle tile and walks a chunk of KC wavevectors, which arrive through the scalar path (uniform
// addresses -> s_load, SGPR operands); each lane keeps 2 KC running sums in registers (KC = 25: 100 VGPRs), all 64 lanes
// are busy, and the KC independent sincos chains of a tile interleave by themselves.  The sums over lanes, waves and blocks
// come afterwards: wave tree, LDS, then density_fold_kernel on the same [chunk-of-64][block][2][64] partial layout as the
// first mapping.  Same term arithmetic (k.r association, sincos) as density_partials_kernel.
__device__ __forceinline__ double wave_sum_f64(double v)
{
    v += dpp_f64<0xB1, 0xF>(v, 0.0);
    v += dpp_f64<0x4E, 0xF>(v, 0.0);
    v += dpp_f64<0x124, 0xF>(v, 0.0);
    v += dpp_f64<0x128, 0xF>(v, 0.0);
    v += __shfl_xor(v, 16, kWave);
    v += __shfl_xor(v, 32, kWave);
    return v;
}

template <int BLOCK, int KC>
__global__ __launch_bounds__(BLOCK) void density_partials_lp_kernel(const char* __restrict__ pos, size_t pos_stride, unsigned N,
                                                                    const double* __restrict__ kvec, unsigned n_k,
                                                                    SinCosCoef coef, double* __restrict__ part)
{
    constexpr int NW = BLOCK / kWave;
    __shared__ double s_acc[NW][2 * KC];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    const unsigned k0 = blockIdx.y * KC;
    const unsigned kcount = n_k - k0 < (unsigned)KC ? n_k - k0 : (unsigned)KC; // wave-uniform
    // the largest |kx| + |ky| + |kz| of the chunk, for the one range check per tile
    double ksum = 0.0;
    for (unsigned kk = 0; kk < kcount; ++kk)
    {
        const double* kv = kvec + 3 * (size_t)(k0 + kk);
        ksum = fmax(ksum, (fabs(kv[0]) + fabs(kv[1])) + fabs(kv[2]));
    }
    // the chunk's wavevectors, loaded once: uniform values, so they live in SGPRs (a padded kk of the last, partial chunk
    // repeats the chunk's first wavevector; it is never stored).  6 KC SGPRs next to the 30 of the sincos coefficients do not
    // all fit the ~100 a wave has: hipcc parks 9 / 41 / 141 of them in VGPR lanes at KC = 5 / 10 / 25 (v_readlane in the tile
    // loop, no scratch memory).  Round 3 measured the alternative -- fetching the wavevectors again for every tile through the
    // scalar cache (pointer laundered so the loads stay in the loop: 0 / 3 / 18 spills) -- and it is SLOWER: 57.5 vs 53 us at
    // n_k = 17 and 113.6 vs 95 us at n_k = 50 (KC = 5, N = 1e6; profiles/r03/ab_density_reload_wavevectors.txt): a
    // v_readlane costs one issue slot, a scalar load a round trip the short tile (KC sincos) cannot hide.  The spills stay.
    double kx[KC], ky[KC], kz[KC];
#pragma unroll
    for (int kk = 0; kk < KC; ++kk)
    {
        const double* kv = kvec + 3 * (size_t)(k0 + ((unsigned)kk < kcount ? kk : 0));
        kx[kk] = kv[0];
        ky[kk] = kv[1];
        kz[kk] = kv[2];
    }
    double re[KC], im[KC];
#pragma unroll
    for (int kk = 0; kk < KC; ++kk)
        re[kk] = im[kk] = 0.0;
    const unsigned ntiles = (N + kWave - 1) / kWave;
    const unsigned wave_u = (unsigned)__builtin_amdgcn_readfirstlane(wave);
    const unsigned gw = blockIdx.x * NW + wave_u, GW = gridDim.x * NW;
    // the position of the NEXT tile is fetched while the current one is being worked on (a tile is only KC sincos per lane:
    // without the prefetch every tile starts with an exposed global-load latency)
    auto fetch = [&](unsigned tile, double& x, double& y, double& z) {
        size_t i = (size_t)tile * kWave + lane;
        i = i < N ? i : (size_t)N - 1;
        const double* p = reinterpret_cast<const double*>(pos + i * pos_stride);
        x = p[0];
        y = p[1];
        z = p[2];
    };
    double nx = 0.0, ny = 0.0, nz = 0.0;
    if (gw < ntiles)
        fetch(gw, nx, ny, nz);
    for (unsigned tile = gw; tile < ntiles; tile += GW)
    {
        const bool valid = (size_t)tile * kWave + lane < N; // false only in the lanes beyond N of the last tile
        const double px = nx, py = ny, pz = nz;
        if (tile + GW < ntiles)
            fetch(tile + GW, nx, ny, nz);
        const double m = fmax(fmax(fabs(px), fabs(py)), fabs(pz));
        // fast path (wave-uniform): every |k . r| of this tile and chunk is below 1e8 (and finite), no padding lanes
        if (!__any(!(ksum * m < 1.0e8) || !valid))
        {
#pragma unroll
            for (int kk = 0; kk < KC; ++kk)
            {
                double sn, cs;
                sincos_reduced(coef, (px * kx[kk] + py * ky[kk]) + pz * kz[kk], sn, cs);
                re[kk] += cs;
                im[kk] += sn;
            }
        }
        else
        {
#pragma unroll
            for (int kk = 0; kk < KC; ++kk)
            {
                const double kr = (px * kx[kk] + py * ky[kk]) + pz * kz[kk];
                double sn, cs;
                if (__any(!(fabs(kr) < 1.0e8))) // wave-uniform; also catches NaN/Inf
                    sincos(kr, &sn, &cs);
                else
                    sincos_reduced(coef, kr, sn, cs);
                re[kk] += valid ? cs : 0.0;
                im[kk] += valid ? sn : 0.0;
            }
        }
    }
#pragma unroll
    for (int kk = 0; kk < KC; ++kk)
    {
        const double r = wave_sum_f64(re[kk]), q = wave_sum_f64(im[kk]);
        if (lane == 0)
        {
            s_acc[wave][2 * kk] = r;
            s_acc[wave][2 * kk + 1] = q;
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 * kcount)
    {
        double v = s_acc[0][threadIdx.x];
#pragma unroll
        for (int wv = 1; wv < NW; ++wv)
            v += s_acc[wv][threadIdx.x];
        const unsigned k = k0 + threadIdx.x / 2, c = threadIdx.x & 1;
        // the [chunk-of-64][block][2][64] layout density_fold_kernel reads
        part[(((size_t)(k / kWave) * gridDim.x + blockIdx.x) * 2 + c) * kWave + (k % kWave)] = v;
    }
}

// one block per chunk of 64 wavevectors: thread (w, lane) folds blocks w, w+NW, ... of wavevector `lane` with TwoSum,
// the NW waves meet in LDS.  out: interleaved (re, im) per wavevector.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void density_fold_kernel(const double* __restrict__ part, unsigned nblocks, unsigned n_k,
                                                             double* __restrict__ out)
{
    constexpr int NW = BLOCK / kWave;
    __shared__ double s_acc[NW][4][kWave];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    const unsigned chunk = blockIdx.x;
    double rh = 0.0, rl = 0.0, ih = 0.0, il = 0.0;
    constexpr int BATCH = 8; // loads in flight per lane: without it every TwoSum waits for its own load
    for (unsigned b0 = wave; b0 < nblocks; b0 += NW * BATCH)
    {
        double vr[BATCH], vi[BATCH];
#pragma unroll
        for (int j = 0; j < BATCH; ++j)
        {
            const unsigned b = b0 + j * NW;
            const double* p = part + ((size_t)chunk * nblocks + (b < nblocks ? b : 0)) * 2 * kWave;
            vr[j] = b < nblocks ? p[lane] : 0.0;
            vi[j] = b < nblocks ? p[kWave + lane] : 0.0;
        }
#pragma unroll
        for (int j = 0; j < BATCH; ++j)
        {
            dd_acc(rh, rl, vr[j]);
            dd_acc(ih, il, vi[j]);
        }
    }
    s_acc[wave][0][lane] = rh; s_acc[wave][1][lane] = rl;
    s_acc[wave][2][lane] = ih; s_acc[wave][3][lane] = il;
    __syncthreads();
    if (wave == 0)
    {
#pragma unroll
        for (int w = 1; w < NW; ++w)
        {
            dd_merge(rh, rl, s_acc[w][0][lane], s_acc[w][1][lane]);
            dd_merge(ih, il, s_acc[w][2][lane], s_acc[w][3][lane]);
        }
        const unsigned k = chunk * kWave + lane;
        if (k < n_k)
        {
            out[2 * k] = rh + rl;
            out[2 * k + 1] = ih + il;
        }
    }
}

// Cavity-mode kinetic energy (reference: CavityModeTracker.compute_cavity_properties, src/cavitymd/analysis.py:1324-1368):
// KE = 1/2 m v.v of the photon found by the last force evaluation; HOOMD keeps the mass in vel.w.
// out[0..3] = KE, harmonic PE (from the result block), KE + PE, temperature = (2/3) KE / k_B.
// The four numbers reach the host through mapped pinned memory: values first, then the call's sequence number
// (system-scope release) that cavmd_cavity_mode spins on -- no copy, no stream synchronisation.
struct HostMode
{
    double v[4];
    uint64_t ready;
};
__global__ void cavity_mode_kernel(const cavmd_result* __restrict__ res, const cavmd_double4* __restrict__ vel, double kB,
                                   double* __restrict__ out, HostMode* __restrict__ host, uint64_t sequence)
{
    const int p = res->photon_idx;
    double ke = 0.0, pe = 0.0, tot = 0.0, temp = 0.0;
    if (p >= 0)
    {
        const cavmd_double4 v = vel[p];
        ke = 0.5 * v.w * ((v.x * v.x + v.y * v.y) + v.z * v.z);
        pe = res->energy[0];
        tot = ke + pe;
        temp = (2.0 / 3.0) * ke / kB;
    }
    out[0] = ke;
    out[1] = pe;
    out[2] = tot;
    out[3] = temp;
    host->v[0] = ke;
    host->v[1] = pe;
    host->v[2] = tot;
    host->v[3] = temp;
    __hip_atomic_store(&host->ready, sequence, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
} // namespace cavmd

// =====================================================================================================================
// Row f4 (data-parallel part): S = sum_i |F_i| / m_i over the net force, the quantity AdaptiveTimestepUpdater turns into
// dt = sqrt(tol / S) (reference: src/cavitymd/simulation.py:66-92, via a host snapshot and a Python list comprehension).
// One streaming pass over the Scalar4 net-force array and the Scalar4 velocity array (HOOMD keeps the mass in vel.w):
// 64 B of lines per particle, 40 B algorithmic.  Same fixed-order compensated tree as the dipole.
// =====================================================================================================================
namespace cavmd
{
// One scalar handed to the host without a copy or a stream synchronisation: mapped, coherent pinned memory; the value is
// stored first, then the call's sequence number is release-stored at system scope (the same hand-off as cavmd_result).
struct HostScalar
{
    double value;
    uint64_t ready;
};
__device__ __forceinline__ void publish_scalar(HostScalar* __restrict__ host, double v, uint64_t sequence)
{
    host->value = v;
    __hip_atomic_store(&host->ready, sequence, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// six-step DPP wave tree + LDS fold for ONE double-double value; total in thread 0
template <int BLOCK>
__device__ __forceinline__ DD block_reduce_dd1(DD v)
{
    constexpr int NW = BLOCK / kWave;
    __shared__ double s_v[NW][2];
    v = dd_sum(v, dd_dpp<0xB1>(v));
    v = dd_sum(v, dd_dpp<0x4E>(v));
    v = dd_sum(v, dd_dpp<0x124>(v));
    v = dd_sum(v, dd_dpp<0x128>(v));
    v = dd_sum(v, dd_shfl_xor(v, 16));
    v = dd_sum(v, dd_shfl_xor(v, 32));
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    if (lane == 0)
    {
        s_v[wave][0] = v.hi;
        s_v[wave][1] = v.lo;
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
#pragma unroll
        for (int w = 1; w < NW; ++w)
            dd_merge(v.hi, v.lo, s_v[w][0], s_v[w][1]);
    }
    return v;
}

// One-launch tail of a scalar reduction: every block stores its double-double partial (8-byte agent-scope write-through
// stores), drains them and takes a ticket; the block that draws the LAST ticket folds all partials in index order (thread t:
// partials t, t + BLOCK, ...; then the block tree) -- a fixed order whichever block it is, so the result is bit-reproducible
// -- and resets the ticket counter for the next call.  Replaces a second launch (host call + kernel boundary + ramp:
// 23 -> 15 us per kinetic-energy call at N = 1e6).  Returns true in the folding block; its total is valid in thread 0.
// Ordering: this is NOT a release/acquire pair of the memory model but the hand-off form MI355X_MICROARCH.md lists as measured
// valid on gfx950 (inter-workgroup visibility, hand-off table, row 1): every byte handed off is stored `sc1` (write-through:
// the 8-byte agent-scope stores below), the one storing wave runs `s_waitcnt vmcnt(0)` after them, the same lane then signals
// with an agent-scope atomic add, the consumer is "the workgroup whose add came last, told by the value its add returned", it
// loads only after that add has returned (the other waves behind a workgroup barrier it then joins), and every load of the
// handed-off bytes is `sc1`.  A release fence before the add and an acquire fence in the folding block would write back and
// invalidate a whole L2 / L1 per block (~1.7 us each, on the critical path of a 9 us kernel that has written 16 bytes).
// A launch that dies half-way leaves the counter part-way: the host puts it back when it reports the failure (wait_scalar).
template <int BLOCK>
__device__ __forceinline__ bool fold_by_last_block(DD& acc, double* __restrict__ part, unsigned* __restrict__ ticket)
{
    __shared__ int s_last;
    const unsigned G = gridDim.x;
    if (threadIdx.x == 0)
    {
        __hip_atomic_store(part + blockIdx.x, acc.hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(part + G + blockIdx.x, acc.lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the partial has left this XCD's L2 before the ticket is drawn
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == G - 1);
    }
    __syncthreads();
    if (!s_last)
        return false;
    DD tot {0.0, 0.0};
    for (unsigned p = threadIdx.x; p < G; p += BLOCK)
        dd_merge(tot.hi, tot.lo, __hip_atomic_load(part + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                 __hip_atomic_load(part + G + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    __syncthreads(); // block_reduce_dd1's LDS array was used by the caller's reduction
    acc = block_reduce_dd1<BLOCK>(tot);
    if (threadIdx.x == 0)
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}

template <int BLOCK, int UNROLL>
__global__ __launch_bounds__(BLOCK) void force_mass_fused_kernel(const v2d* __restrict__ force2, const v2d* __restrict__ vel2,
                                                                 unsigned N, double* __restrict__ part,
                                                                 unsigned* __restrict__ ticket, double* __restrict__ out,
                                                                 HostScalar* __restrict__ host, uint64_t sequence)
{
    constexpr unsigned TILE = BLOCK * UNROLL;
    DD acc {0.0, 0.0};
    const unsigned tiles = (N + TILE - 1) / TILE;
    for (unsigned t = blockIdx.x; t < tiles; t += gridDim.x)
    {
        const size_t base = (size_t)t * TILE + threadIdx.x;
        v2d fxy[UNROLL], fzw[UNROLL], vzw[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
        {
            const size_t i = base + (size_t)u * BLOCK;
            const bool ok = i < N;
            const v2d zero = {0.0, 0.0}, one = {0.0, 1.0};
            fxy[u] = ok ? __builtin_nontemporal_load(force2 + 2 * i) : zero;
            fzw[u] = ok ? __builtin_nontemporal_load(force2 + 2 * i + 1) : zero;
            // the mass sits in vel.w: only the (vz, m) half of each velocity entry is asked for.  (Asking for the (vx, vy) half
            // as well, so that the wave's loads cover whole lines, measured slower: 19.1 vs 17.5 us at N = 1e6, round 3.)
            vzw[u] = ok ? __builtin_nontemporal_load(vel2 + 2 * i + 1) : one;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
        {
            const double n2 = (fxy[u].x * fxy[u].x + fxy[u].y * fxy[u].y) + fzw[u].x * fzw[u].x;
            dd_acc(acc.hi, acc.lo, sqrt(n2) / vzw[u].y); // |F_i| / m_i; padding lanes add 0 / 1
        }
    }
    acc = block_reduce_dd1<BLOCK>(acc);
    if (fold_by_last_block<BLOCK>(acc, part, ticket) && threadIdx.x == 0)
    {
        out[0] = acc.hi + acc.lo;
        publish_scalar(host, acc.hi + acc.lo, sequence);
    }
}

} // namespace cavmd

// =====================================================================================================================
// Row f4, thermostat side: the translational kinetic energy the Bussi reservoir thermostat consumes
// (src/BussiReservoirThermostat.h:49-54: m_thermo->getTranslationalKineticEnergy() over the thermostat's group) and the
// velocity rescaling its factor feeds (HOOMD's integration method multiplies the group's velocities by alpha).
// KE = 1/2 sum_j m_j (vx^2 + vy^2 + vz^2) over the group's members, HOOMD's Scalar4 velocity array (mass in .w); the terms
// are formed with one rounding per operation and summed in the same fixed-order compensated tree as the dipole.
// members == nullptr: all particles 0 .. n-1 (streaming, 32 B per particle); else a device array of particle indices
// (HOOMD's ParticleGroup index list), gathered 32 B per member.
// =====================================================================================================================
namespace cavmd
{
// sum_j m_j (vx^2 + vy^2 + vz^2) over this block's tiles (tile t -> block t % gridDim.x), one double-double per lane
template <int BLOCK, int UNROLL>
__device__ __forceinline__ DD kinetic_partial(const v2d* __restrict__ vel2, const unsigned* __restrict__ members, unsigned n)
{
    constexpr unsigned TILE = BLOCK * UNROLL;
    DD acc {0.0, 0.0};
    const unsigned tiles = (n + TILE - 1) / TILE;
    for (unsigned t = blockIdx.x; t < tiles; t += gridDim.x)
    {
        const size_t base = (size_t)t * TILE + threadIdx.x;
        v2d vxy[UNROLL], vzw[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
        {
            const size_t k = base + (size_t)u * BLOCK;
            const bool ok = k < n;
            const v2d zero = {0.0, 0.0};
            const size_t j = ok ? (members ? (size_t)members[k] : k) : 0;
            vxy[u] = ok ? vel2[2 * j] : zero;
            vzw[u] = ok ? vel2[2 * j + 1] : zero; // padding lanes: mass 0 -> term 0
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
            dd_acc(acc.hi, acc.lo, vzw[u].y * ((vxy[u].x * vxy[u].x + vxy[u].y * vxy[u].y) + vzw[u].x * vzw[u].x));
    }
    return acc;
}

template <int BLOCK, int UNROLL>
__global__ __launch_bounds__(BLOCK) void kinetic_fused_kernel(const v2d* __restrict__ vel2, const unsigned* __restrict__ members,
                                                              unsigned n, double* __restrict__ part,
                                                              unsigned* __restrict__ ticket, double* __restrict__ out,
                                                              HostScalar* __restrict__ host, uint64_t sequence)
{
    DD acc = kinetic_partial<BLOCK, UNROLL>(vel2, members, n);
    acc = block_reduce_dd1<BLOCK>(acc);
    if (fold_by_last_block<BLOCK>(acc, part, ticket) && threadIdx.x == 0)
    {
        out[0] = 0.5 * (acc.hi + acc.lo);
        publish_scalar(host, 0.5 * (acc.hi + acc.lo), sequence);
    }
}

// ---- the whole translational thermostat step on the device (round 3) --------------------------------------------------------
// The rule of compute_rescale_factor (src/BussiReservoirThermostat.h:177-225) with c = exp(-dt / tau) supplied by the caller
// (it depends on host parameters only; everything else is +, -, *, /, sqrt, correctly rounded on host and device alike, and
// this file is built with -ffp-contract=off): the host entry point cavmd_bussi_rescale_factor and the kernel below run this
// very function, so they give the same bits for the same kinetic energy.
__host__ __device__ inline double bussi_alpha_from_c(double K, double degrees_of_freedom, double c, double set_T, double R,
                                                     double gamma_variate)
{
    if (degrees_of_freedom == 0) // :183-184
        return 1.0;
    const double r_gamma = (degrees_of_freedom > 1.0) ? 2.0 * gamma_variate : 0.0; // :192-199
    const double v = set_T / 2.0 / K;                                              // :201-203
    const double term1 = v * (1.0 - c) * (r_gamma + R * R);
    const double term2 = 2.0 * R * sqrt(v * (1.0 - c) * c);
    const double magnitude = sqrt(c + term1 + term2);                              // :206-207
    const double K_bar = set_T * degrees_of_freedom / 2.0;                         // :211-213, Bussi et al. 2009 eq. (A8)
    const double sign_term = R + sqrt(c * degrees_of_freedom * K / ((1.0 - c) * K_bar));
    return (sign_term >= 0.0) ? magnitude : -magnitude;
}

// State of the on-device thermostat (device memory; a copy goes to mapped host memory after every step).
struct BussiDevice
{
    double reservoir;     // cumulative energy handed to the bath, sum of KE (1 - alpha^2)   (src/BussiReservoirThermostat.h:86-95)
    double instantaneous; // the last step's share
    double alpha;         // the last step's factor
    double kinetic;       // the kinetic energy the last step saw
    uint64_t steps;       // steps applied
    uint64_t errors;      // steps refused: degrees of freedom without kinetic energy ("requires non-zero initial momenta", :57-61)
};
struct HostBussi
{
    BussiDevice state;
    uint64_t ready; // sequence number of the step `state` belongs to
};
struct BussiStepArgs
{
    double dof, c, set_T, normal_variate, gamma_variate;
};

// Launch 1 of the on-device step: one double-double partial per block, nothing else (plain stores; the kernel boundary makes
// them visible to launch 2).  No ticket, no last-block fold: that latency-bound tail (~3.5 us of the 9 us kinetic_fused_kernel
// takes at N = 1e6) is only needed when ONE kernel has to hand the total to the host.
template <int BLOCK, int UNROLL>
__global__ __launch_bounds__(BLOCK) void kinetic_partials_kernel(const v2d* __restrict__ vel2, const unsigned* __restrict__ members,
                                                                 unsigned n, double* __restrict__ part)
{
    DD acc = kinetic_partial<BLOCK, UNROLL>(vel2, members, n);
    acc = block_reduce_dd1<BLOCK>(acc);
    if (threadIdx.x == 0)
    {
        part[blockIdx.x] = acc.hi;
        part[gridDim.x + blockIdx.x] = acc.lo;
    }
}

// v_j.xyz *= alpha for the members of the group (mass in .w untouched).  A block sweeps tiles of BLOCK * UNROLL members: all
// 2 * UNROLL 16-byte loads of a lane are issued before the first store (the grid-stride loop this replaces, one particle per
// lane and round trip, ran at 5.2 TB/s of read + write at N = 1e7; see profiles/r03/observables_kernel_stats_1e7.csv).
template <int BLOCK, int UNROLL>
struct ScaleTile
{
    v2d xy[UNROLL], zw[UNROLL];
    size_t j[UNROLL]; // particle index per slot, (size_t)-1 = padding
};
template <int BLOCK, int UNROLL>
__device__ __forceinline__ void scale_tile_load(const v2d* __restrict__ vel2, const unsigned* __restrict__ members, unsigned n,
                                                unsigned t, ScaleTile<BLOCK, UNROLL>& r)
{
    const size_t base = (size_t)t * (BLOCK * UNROLL) + threadIdx.x;
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
    {
        const size_t k = base + (size_t)u * BLOCK;
        r.j[u] = k < n ? (members ? (size_t)members[k] : k) : (size_t)-1;
        if (r.j[u] != (size_t)-1)
        {
            r.xy[u] = vel2[2 * r.j[u]];
            r.zw[u] = vel2[2 * r.j[u] + 1];
        }
    }
}
template <int BLOCK, int UNROLL>
__device__ __forceinline__ void scale_tile_store(v2d* __restrict__ vel2, double alpha, ScaleTile<BLOCK, UNROLL>& r)
{
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
    {
        if (r.j[u] == (size_t)-1)
            continue;
        r.xy[u].x *= alpha;
        r.xy[u].y *= alpha;
        r.zw[u].x *= alpha;
        vel2[2 * r.j[u]] = r.xy[u];
        vel2[2 * r.j[u] + 1] = r.zw[u];
    }
}
// tiles first, first + gridDim.x, ...; `pre`: the block's first tile has been loaded by the caller already (before a prologue
// whose latency those loads then hide)
template <int BLOCK, int UNROLL>
__device__ __forceinline__ void scale_velocities_body(v2d* __restrict__ vel2, const unsigned* __restrict__ members, unsigned n,
                                                      double alpha, ScaleTile<BLOCK, UNROLL>* pre = nullptr)
{
    constexpr unsigned TILE = BLOCK * UNROLL;
    const unsigned tiles = (n + TILE - 1) / TILE;
    unsigned t = blockIdx.x;
    if (pre && t < tiles)
    {
        scale_tile_store<BLOCK, UNROLL>(vel2, alpha, *pre);
        t += gridDim.x;
    }
    for (; t < tiles; t += gridDim.x)
    {
        ScaleTile<BLOCK, UNROLL> r;
        scale_tile_load<BLOCK, UNROLL>(vel2, members, n, t, r);
        __builtin_amdgcn_sched_barrier(0);
        scale_tile_store<BLOCK, UNROLL>(vel2, alpha, r);
    }
}

template <int BLOCK, int UNROLL>
__global__ __launch_bounds__(BLOCK) void scale_velocities_kernel(v2d* __restrict__ vel2, const unsigned* __restrict__ members,
                                                                 unsigned n, double alpha)
{
    scale_velocities_body<BLOCK, UNROLL>(vel2, members, n, alpha);
}

// Launch 2 of the on-device step: EVERY block folds launch 1's partials itself, in the order fold_by_last_block uses (thread t:
// partials t, t + BLOCK, ...; then the block tree), so every block holds the kinetic energy cavmd_kinetic_energy would return,
// bit for bit, evaluates the rule (a pure function of it and of the arguments: the same alpha in every block) and rescales its
// share of the group -- the structure of the two-launch force path (force_map_aos_fused_kernel).  Block 0 also keeps the books:
// reservoir counters in device memory and their copy, with the step's sequence number, in mapped host memory.  alpha == 1
// (a refused step, or 0 degrees of freedom) leaves the array untouched -- multiplying by 1.0 would give the same bits.
template <int BLOCK, int UNROLL>
__global__ __launch_bounds__(BLOCK) void bussi_rescale_fused_kernel(v2d* __restrict__ vel2, const unsigned* __restrict__ members,
                                                                    unsigned n, const double* __restrict__ part, unsigned G1,
                                                                    BussiStepArgs a, BussiDevice* __restrict__ state,
                                                                    HostBussi* __restrict__ host, uint64_t sequence)
{
    __shared__ double s_alpha;
    // this block's first tile is asked for before the prologue, whose latency (partials, block tree, the rule) it then hides
    ScaleTile<BLOCK, UNROLL> first;
    const unsigned tiles = (n + BLOCK * UNROLL - 1) / (BLOCK * UNROLL);
    if (blockIdx.x < tiles)
        scale_tile_load<BLOCK, UNROLL>(vel2, members, n, blockIdx.x, first);
    DD tot {0.0, 0.0};
    for (unsigned p = threadIdx.x; p < G1; p += BLOCK)
        dd_merge(tot.hi, tot.lo, part[p], part[G1 + p]);
    tot = block_reduce_dd1<BLOCK>(tot);
    if (threadIdx.x == 0)
    {
        const double K = 0.5 * (tot.hi + tot.lo);
        const bool refused = (a.dof != 0 && K == 0); // "Bussi thermostat requires non-zero initial momenta." (:57-61)
        const double alpha = refused ? 1.0 : bussi_alpha_from_c(K, a.dof, a.c, a.set_T, a.normal_variate, a.gamma_variate);
        s_alpha = alpha;
        if (blockIdx.x == 0)
        {
            BussiDevice s = *state;
            s.kinetic = K;
            s.alpha = alpha;
            if (refused)
            {
                s.instantaneous = 0.0; // nothing is rescaled; counted, and reported by the next cavmd_bussi_device_read
                s.errors += 1;
            }
            else
            {
                const double delta = K * (1.0 - alpha * alpha); // src/BussiReservoirThermostat.h:86-95
                s.reservoir += delta;
                s.instantaneous = delta;
                s.steps += 1;
            }
            *state = s;
            host->state = s;
            __hip_atomic_store(&host->ready, sequence, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    __syncthreads();
    const double alpha = s_alpha;
    if (alpha == 1.0)
        return;
    scale_velocities_body<BLOCK, UNROLL>(vel2, members, n, alpha, &first);
}
} // namespace cavmd
