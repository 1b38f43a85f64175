// cavmd_kernels.hpp -- CDNA4 (gfx950) device code of the cavity-force path.
//
// Three kernels per evaluation, all bandwidth- or latency-bound (no MFMA: ~30 flop per 92 B):
//
//   dipole_partials   streams pos (32 B) + charge (8 B) + image (12 B) per particle, unwraps, forms the
//                     addends c_i * r_i exactly as the reference does (one rounding per operation, no
//                     FMA), accumulates them per lane in double-double (TwoSum), reduces lane -> wave
//                     (shuffles) -> block (LDS) in a fixed order and writes ONE partial per block.
//                     Also finds the photon (minimum index whose type tag is L).
//                     reference: findPhotonParticle + computeUnwrappedPositions + computeDipoleMoment,
//                     src/CavityForceCompute.cc:73-129.
//   finalize          one block: reduces the <= 4096 partials in index order, unwraps the photon, evaluates
//                     the three energies, Dq and the photon force with the reference's operator
//                     association (src/CavityForceCompute.cc:169-183, 203-207) and writes cavmd_result.
//   force_map         streams charge (8 B) and writes force (32 B) per particle as dense 16-byte chunks:
//                     even chunk = (Fx, Fy), odd chunk = (Fz, w) = (0, 0); the photon's two chunks carry F_L.
//                     Every entry of the force array is written, so the reference's memset pass
//                     (src/CavityForceCompute.cc:145) is not needed.  reference: :183-207.
//
// No atomics on floating-point data, no dependence on dispatch order: results are bit-reproducible for
// a given (N, launch geometry).  The kernel boundary is the only inter-workgroup hand-off.
#pragma once

#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>

#include "cavmd.h"

// The reference is built without FMA contraction (flag-less x86-64); keep every a + b*c as two
// roundings on the device too, and keep the TwoSum error terms from being "simplified".
#pragma clang fp contract(off)

namespace cavmd
{

typedef double v2d __attribute__((ext_vector_type(2)));
typedef int v3i __attribute__((ext_vector_type(3)));

constexpr int kWave = 64;         // gfx950 wavefront
constexpr int kNumPartDoubles = 9; // main sum hi/lo x3 + L-typed sum x3
constexpr int kNumPartInts = 2;    // min L index, count of L-typed particles

// ---- double-double helpers ---------------------------------------------------------------------
// (hi, lo) += t, error-free (Knuth TwoSum; 6 flops + 1)
__device__ __forceinline__ void dd_acc(double& hi, double& lo, double t)
{
    const double s = hi + t;
    const double bb = s - hi;
    const double e = (hi - (s - bb)) + (t - bb);
    hi = s;
    lo += e;
}
// (hi, lo) += (ohi, olo)
__device__ __forceinline__ void dd_merge(double& hi, double& lo, double ohi, double olo)
{
    const double s = hi + ohi;
    const double bb = s - hi;
    const double e = (hi - (s - bb)) + (ohi - bb);
    hi = s;
    lo = (lo + olo) + e;
}
// renormalise so that hi = fl(hi + lo)
__device__ __forceinline__ void dd_norm(double& hi, double& lo)
{
    const double s = hi + lo;
    const double bb = s - hi;
    const double e = (hi - (s - bb)) + (lo - bb);
    hi = s;
    lo = e;
}

__device__ __forceinline__ double shfl_down_f64(double v, int delta)
{
    return __shfl_down(v, delta, kWave);
}

// Per-lane running state of the reduction.
struct Accum
{
    double hx = 0.0, lx = 0.0, hy = 0.0, ly = 0.0, hz = 0.0, lz = 0.0; // molecules: double-double
    double sx = 0.0, sy = 0.0, sz = 0.0;                               // L-typed particles (normally one)
    int lmin = INT_MAX;                                                // smallest index of type L
    int lcnt = 0;                                                      // how many of type L

    // one particle: positions already unwrapped by the caller
    __device__ __forceinline__ void add(unsigned idx, double rx, double ry, double rz, double c, int tag, int L_typeid)
    {
        const double tx = c * rx;
        const double ty = c * ry;
        const double tz = c * rz;
        const bool isL = (tag == L_typeid);
        dd_acc(hx, lx, isL ? 0.0 : tx);
        dd_acc(hy, ly, isL ? 0.0 : ty);
        dd_acc(hz, lz, isL ? 0.0 : tz);
        sx += isL ? tx : 0.0;
        sy += isL ? ty : 0.0;
        sz += isL ? tz : 0.0;
        lmin = isL ? min(lmin, (int)idx) : lmin;
        lcnt += isL ? 1 : 0;
    }
    __device__ __forceinline__ void merge(const Accum& o)
    {
        dd_merge(hx, lx, o.hx, o.lx);
        dd_merge(hy, ly, o.hy, o.ly);
        dd_merge(hz, lz, o.hz, o.lz);
        sx += o.sx;
        sy += o.sy;
        sz += o.sz;
        lmin = min(lmin, o.lmin);
        lcnt += o.lcnt;
    }
    __device__ __forceinline__ Accum shfl_down(int delta) const
    {
        Accum o;
        o.hx = shfl_down_f64(hx, delta);
        o.lx = shfl_down_f64(lx, delta);
        o.hy = shfl_down_f64(hy, delta);
        o.ly = shfl_down_f64(ly, delta);
        o.hz = shfl_down_f64(hz, delta);
        o.lz = shfl_down_f64(lz, delta);
        o.sx = shfl_down_f64(sx, delta);
        o.sy = shfl_down_f64(sy, delta);
        o.sz = shfl_down_f64(sz, delta);
        o.lmin = __shfl_down(lmin, delta, kWave);
        o.lcnt = __shfl_down(lcnt, delta, kWave);
        return o;
    }
};

// Fixed-shape block reduction: lanes -> wave (5+1 shuffle steps) -> LDS -> thread 0 folds the waves in
// wave order.  Returns the block total in thread 0 (other threads hold partial garbage).
template <int BLOCK>
__device__ __forceinline__ Accum block_reduce(Accum a)
{
    constexpr int NW = BLOCK / kWave;
    __shared__ double s_d[NW][kNumPartDoubles];
    __shared__ int s_i[NW][kNumPartInts];
#pragma unroll
    for (int delta = kWave / 2; delta > 0; delta >>= 1)
    {
        const Accum o = a.shfl_down(delta);
        a.merge(o);
    }
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    if (lane == 0)
    {
        s_d[wave][0] = a.hx; s_d[wave][1] = a.lx; s_d[wave][2] = a.hy; s_d[wave][3] = a.ly;
        s_d[wave][4] = a.hz; s_d[wave][5] = a.lz; s_d[wave][6] = a.sx; s_d[wave][7] = a.sy;
        s_d[wave][8] = a.sz;
        s_i[wave][0] = a.lmin;
        s_i[wave][1] = a.lcnt;
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
#pragma unroll
        for (int w = 1; w < NW; ++w)
        {
            Accum o;
            o.hx = s_d[w][0]; o.lx = s_d[w][1]; o.hy = s_d[w][2]; o.ly = s_d[w][3];
            o.hz = s_d[w][4]; o.lz = s_d[w][5]; o.sx = s_d[w][6]; o.sy = s_d[w][7];
            o.sz = s_d[w][8];
            o.lmin = s_i[w][0];
            o.lcnt = s_i[w][1];
            a.merge(o);
        }
    }
    return a;
}

// ---- input layouts -------------------------------------------------------------------------------
// HOOMD-native AoS: Scalar4 pos (type tag in the low 32 bits of .w), Scalar charge, int3 image.
struct AosInput
{
    const v2d* __restrict__ pos2;      // 2 x 16 B per particle
    const double* __restrict__ charge;
    const int* __restrict__ image;     // 3 ints per particle, 12-byte stride

    struct Raw
    {
        v2d xy, zw;
        double c;
        int ix, iy, iz;
    };
    __device__ __forceinline__ Raw load(size_t i) const
    {
        Raw r;
        r.xy = pos2[2 * i];
        r.zw = pos2[2 * i + 1];
        r.c = charge[i];
        const int* im = image + 3 * i;
        r.ix = im[0];
        r.iy = im[1];
        r.iz = im[2];
        return r;
    }
    static __device__ __forceinline__ double x(const Raw& r) { return r.xy.x; }
    static __device__ __forceinline__ double y(const Raw& r) { return r.xy.y; }
    static __device__ __forceinline__ double z(const Raw& r) { return r.zw.x; }
    static __device__ __forceinline__ int tag(const Raw& r) { return __double2loint(r.zw.y); }
};

// Snapshot layout with byte strides: position (N,3) f64, typeid (N,) i32, image (N,3) i32, charge (N,) f64.
struct StridedInput
{
    const char* __restrict__ pos;
    const char* __restrict__ tid;
    const char* __restrict__ img;
    const char* __restrict__ chg;
    size_t pos_stride, tid_stride, img_stride, chg_stride;

    struct Raw
    {
        double px, py, pz, c;
        int ix, iy, iz, t;
    };
    __device__ __forceinline__ Raw load(size_t i) const
    {
        Raw r;
        const double* p = reinterpret_cast<const double*>(pos + i * pos_stride);
        r.px = p[0];
        r.py = p[1];
        r.pz = p[2];
        r.c = *reinterpret_cast<const double*>(chg + i * chg_stride);
        const int* im = reinterpret_cast<const int*>(img + i * img_stride);
        r.ix = im[0];
        r.iy = im[1];
        r.iz = im[2];
        r.t = *reinterpret_cast<const int*>(tid + i * tid_stride);
        return r;
    }
    static __device__ __forceinline__ double x(const Raw& r) { return r.px; }
    static __device__ __forceinline__ double y(const Raw& r) { return r.py; }
    static __device__ __forceinline__ double z(const Raw& r) { return r.pz; }
    static __device__ __forceinline__ int tag(const Raw& r) { return r.t; }
};

// Where the per-block partials live (SoA so the finalize kernel reads them coalesced).
struct Partials
{
    double* d;      // kNumPartDoubles arrays of `stride` doubles
    int* i;         // kNumPartInts arrays of `stride` ints
    unsigned stride;
};

// ---- kernel 1: per-block partial dipole sums + photon search --------------------------------------
template <class Input, int BLOCK, int UNROLL>
__global__ __launch_bounds__(BLOCK) void dipole_partials_kernel(Input in, unsigned N, double Lx, double Ly, double Lz,
                                                                int L_typeid, Partials part)
{
    constexpr unsigned TILE = BLOCK * UNROLL;
    Accum acc;
    const unsigned full_tiles = N / TILE;
    for (unsigned t = blockIdx.x; t < full_tiles; t += gridDim.x)
    {
        const size_t base = (size_t)t * TILE + threadIdx.x;
        typename Input::Raw raw[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
            raw[u] = in.load(base + (size_t)u * BLOCK);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
        {
            const double rx = Input::x(raw[u]) + (double)raw[u].ix * Lx;
            const double ry = Input::y(raw[u]) + (double)raw[u].iy * Ly;
            const double rz = Input::z(raw[u]) + (double)raw[u].iz * Lz;
            acc.add((unsigned)(base + (size_t)u * BLOCK), rx, ry, rz, raw[u].c, Input::tag(raw[u]), L_typeid);
        }
    }
    // ragged tail: one block takes it, bounds-checked
    if (blockIdx.x == full_tiles % gridDim.x)
    {
        const size_t base = (size_t)full_tiles * TILE + threadIdx.x;
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
        {
            const size_t i = base + (size_t)u * BLOCK;
            if (i < N)
            {
                const typename Input::Raw r = in.load(i);
                const double rx = Input::x(r) + (double)r.ix * Lx;
                const double ry = Input::y(r) + (double)r.iy * Ly;
                const double rz = Input::z(r) + (double)r.iz * Lz;
                acc.add((unsigned)i, rx, ry, rz, r.c, Input::tag(r), L_typeid);
            }
        }
    }
    acc = block_reduce<BLOCK>(acc);
    if (threadIdx.x == 0)
    {
        const unsigned b = blockIdx.x, s = part.stride;
        part.d[0 * s + b] = acc.hx; part.d[1 * s + b] = acc.lx;
        part.d[2 * s + b] = acc.hy; part.d[3 * s + b] = acc.ly;
        part.d[4 * s + b] = acc.hz; part.d[5 * s + b] = acc.lz;
        part.d[6 * s + b] = acc.sx; part.d[7 * s + b] = acc.sy; part.d[8 * s + b] = acc.sz;
        part.i[0 * s + b] = acc.lmin;
        part.i[1 * s + b] = acc.lcnt;
    }
}

// ---- kernel 2: final reduction + scalars ------------------------------------------------------------
template <class Input, int BLOCK>
__global__ __launch_bounds__(BLOCK) void finalize_kernel(Input in, unsigned N, unsigned nparts, double Lx, double Ly,
                                                         double Lz, cavmd_params prm, Partials part, uint64_t sequence,
                                                         cavmd_result* __restrict__ res)
{
    Accum acc;
    const unsigned s = part.stride;
    // thread t folds partials t, t+BLOCK, ... in index order: a fixed tree for a fixed nparts
    for (unsigned p = threadIdx.x; p < nparts; p += BLOCK)
    {
        Accum o;
        o.hx = part.d[0 * s + p]; o.lx = part.d[1 * s + p];
        o.hy = part.d[2 * s + p]; o.ly = part.d[3 * s + p];
        o.hz = part.d[4 * s + p]; o.lz = part.d[5 * s + p];
        o.sx = part.d[6 * s + p]; o.sy = part.d[7 * s + p]; o.sz = part.d[8 * s + p];
        o.lmin = part.i[0 * s + p];
        o.lcnt = part.i[1 * s + p];
        acc.merge(o);
    }
    acc = block_reduce<BLOCK>(acc);
    if (threadIdx.x != 0)
        return;

    dd_norm(acc.hx, acc.lx);
    dd_norm(acc.hy, acc.ly);
    dd_norm(acc.hz, acc.lz);
    double dx = acc.hx, dy = acc.hy, dz = acc.hz;
    const int photon = (acc.lmin == INT_MAX) ? -1 : acc.lmin;

    const double g = prm.couplstr, K = prm.K;
    double qx = 0.0, qy = 0.0, qz = 0.0;
    double eh = 0.0, ec = 0.0, ed = 0.0;
    double Dqx = 0.0, Dqy = 0.0;
    double fx = 0.0, fy = 0.0, fz = 0.0;
    if (photon >= 0)
    {
        const typename Input::Raw r = in.load((size_t)photon);
        qx = Input::x(r) + (double)r.ix * Lx;
        qy = Input::y(r) + (double)r.iy * Ly;
        qz = Input::z(r) + (double)r.iz * Lz;
        if (acc.lcnt > 1)
        {
            // Degenerate input (the driver enforces exactly one 'L', examples/05_advanced_run.py:548-550):
            // the reference skips only the FIRST L-typed particle in the dipole (src/CavityForceCompute.cc:122),
            // so the later ones are added back here.
            dx += acc.sx - r.c * qx;
            dy += acc.sy - r.c * qy;
            dz += acc.sz - r.c * qz;
        }
        // src/CavityForceCompute.cc:174-176, dot() = a.x*b.x + a.y*b.y + a.z*b.z
        eh = 0.5 * K * (qx * qx + qy * qy + qz * qz);
        ec = g * (dx * qx + dy * qy + 0.0 * 0.0);
        ed = 0.5 * (g * g / K) * (dx * dx + dy * dy + 0.0 * 0.0);
        // :183
        const double gK = g / K;
        Dqx = qx + gK * dx;
        Dqy = qy + gK * dy;
        // :203-207
        fx = -K * qx - g * dx;
        fy = -K * qy - g * dy;
        fz = -K * qz - g * 0.0;
    }
    else
    {
        // no photon: the reference zeroes energies and returns before it computes a dipole (:148-156)
        dx = dy = dz = 0.0;
        acc.lx = acc.ly = acc.lz = 0.0;
    }
    res->dipole[0] = dx; res->dipole[1] = dy; res->dipole[2] = dz;
    res->q[0] = qx; res->q[1] = qy; res->q[2] = qz;
    res->Dq[0] = Dqx; res->Dq[1] = Dqy;
    res->energy[0] = eh; res->energy[1] = ec; res->energy[2] = ed;
    res->photon_force[0] = fx; res->photon_force[1] = fy; res->photon_force[2] = fz;
    res->dipole_lo[0] = acc.lx; res->dipole_lo[1] = acc.ly; res->dipole_lo[2] = acc.lz;
    res->photon_idx = photon;
    res->n_photon_typed = acc.lcnt;
    res->n_particles = N;
    res->n_partials = nparts;
    res->sequence = sequence;
    res->reserved[0] = res->reserved[1] = res->reserved[2] = res->reserved[3] = 0.0;
}

// ---- kernel 3: force map, HOOMD AoS force array written as dense 16-byte chunks ---------------------
template <bool NT>
__device__ __forceinline__ void store_chunk(v2d* p, v2d v)
{
    if (NT)
        __builtin_nontemporal_store(v, p);
    else
        *p = v;
}

template <int BLOCK, int UNROLL, bool NT>
__global__ __launch_bounds__(BLOCK) void force_map_aos_kernel(const double* __restrict__ charge,
                                                              const v2d* __restrict__ pos2, // only read if several L-typed
                                                              unsigned N, double g, int L_typeid,
                                                              const cavmd_result* __restrict__ res, v2d* __restrict__ force2)
{
    constexpr unsigned TILE = BLOCK * UNROLL; // in 16-byte chunks; chunk k: particle k>>1, half k&1
    const double Dqx = res->Dq[0], Dqy = res->Dq[1];
    const int photon = res->photon_idx;
    const int nL = res->n_photon_typed;
    const double Fx = res->photon_force[0], Fy = res->photon_force[1], Fz = res->photon_force[2];
    const double ng = -g;
    const size_t nchunks = 2 * (size_t)N;
    const size_t pchunk = photon >= 0 ? 2 * (size_t)photon : ~(size_t)0; // photon's first chunk
    const unsigned full_tiles = (unsigned)(nchunks / TILE);
    const bool odd = threadIdx.x & 1; // BLOCK and TILE are even, so the half is fixed per thread
    const v2d zero = {0.0, 0.0};

    if (photon < 0)
    {
        // no photon: all forces are zero (src/CavityForceCompute.cc:145-156)
        for (size_t k = (size_t)blockIdx.x * BLOCK + threadIdx.x; k < nchunks; k += (size_t)gridDim.x * BLOCK)
            store_chunk<NT>(force2 + k, zero);
        return;
    }

    if (nL <= 1)
    {
        for (unsigned t = blockIdx.x; t < full_tiles; t += gridDim.x)
        {
            const size_t base = (size_t)t * TILE + threadIdx.x;
            double c[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
                c[u] = charge[(base + (size_t)u * BLOCK) >> 1];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
            {
                const size_t k = base + (size_t)u * BLOCK;
                const double s = ng * c[u]; // ((-g) * charge) * Dq, src/CavityForceCompute.cc:194
                v2d v = {s * Dqx, s * Dqy};
                v = odd ? zero : v;
                if ((k | 1) == (pchunk | 1))
                    v = odd ? (v2d) {Fz, 0.0} : (v2d) {Fx, Fy};
                store_chunk<NT>(force2 + k, v);
            }
        }
        if (blockIdx.x == full_tiles % gridDim.x)
        {
            const size_t base = (size_t)full_tiles * TILE + threadIdx.x;
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
            {
                const size_t k = base + (size_t)u * BLOCK;
                if (k < nchunks)
                {
                    const double s = ng * charge[k >> 1];
                    v2d v = {s * Dqx, s * Dqy};
                    v = odd ? zero : v;
                    if ((k | 1) == (pchunk | 1))
                        v = odd ? (v2d) {Fz, 0.0} : (v2d) {Fx, Fy};
                    store_chunk<NT>(force2 + k, v);
                }
            }
        }
        return;
    }

    // Several L-typed particles (degenerate): the reference gives a molecular force only to particles
    // whose type is not L (src/CavityForceCompute.cc:190-191), so the type tag has to be read.
    for (size_t k = (size_t)blockIdx.x * BLOCK + threadIdx.x; k < nchunks; k += (size_t)gridDim.x * BLOCK)
    {
        const size_t p = k >> 1;
        const int tag = __double2loint(pos2[2 * p + 1].y);
        const double s = ng * charge[p];
        v2d v = {s * Dqx, s * Dqy};
        v = (odd || tag == L_typeid) ? zero : v;
        if ((k | 1) == (pchunk | 1))
            v = odd ? (v2d) {Fz, 0.0} : (v2d) {Fx, Fy};
        store_chunk<NT>(force2 + k, v);
    }
}

// ---- kernel 3': force map for the snapshot layout (strided (N,3) force + optional potential energy) ---
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void force_map_strided_kernel(StridedInput in, unsigned N, double g, int L_typeid,
                                                                  const cavmd_result* __restrict__ res,
                                                                  char* __restrict__ force, size_t force_stride,
                                                                  char* __restrict__ pe, size_t pe_stride)
{
    const double Dqx = res->Dq[0], Dqy = res->Dq[1];
    const int photon = res->photon_idx;
    const int nL = res->n_photon_typed;
    const double Fx = res->photon_force[0], Fy = res->photon_force[1], Fz = res->photon_force[2];
    const double ng = -g;
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < N; i += (size_t)gridDim.x * BLOCK)
    {
        double fx = 0.0, fy = 0.0, fz = 0.0;
        if (photon >= 0)
        {
            const double c = *reinterpret_cast<const double*>(in.chg + i * in.chg_stride);
            const double s = ng * c;
            fx = s * Dqx;
            fy = s * Dqy;
            if (nL > 1)
            {
                const int t = *reinterpret_cast<const int*>(in.tid + i * in.tid_stride);
                if (t == L_typeid)
                {
                    fx = 0.0;
                    fy = 0.0;
                }
            }
            if ((int)i == photon)
            {
                fx = Fx;
                fy = Fy;
                fz = Fz;
            }
        }
        double* f = reinterpret_cast<double*>(force + i * force_stride);
        f[0] = fx;
        f[1] = fy;
        f[2] = fz;
        if (pe)
            *reinterpret_cast<double*>(pe + i * pe_stride) = 0.0;
    }
}

} // namespace cavmd
