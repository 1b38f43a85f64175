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

// Diagnostic hook: a developer build (csrc/microbench.hip) defines CAVMD_STAMP(k) to record s_memtime at
// numbered points of the finalize chain; in the product it expands to nothing.
#ifndef CAVMD_STAMP
#define CAVMD_STAMP(k)
#endif

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

// ---- cross-lane movement: DPP (data-parallel primitives), no LDS crossbar round trip -----------------------------
// dpp_ctrl encodings (GFX9 family, which gfx950 belongs to): quad_perm[a,b,c,d] = a|b<<2|c<<4|d<<6,
// row_half_mirror 0x141, row_mirror 0x140, row_bcast:15 0x142, row_bcast:31 0x143.  A "row" is 16 lanes.
// Lanes that the control/row mask does not write receive `ident`.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v, double ident)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(ident), __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(ident), __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i32(int v, int ident)
{
    return __builtin_amdgcn_update_dpp(ident, v, CTRL, ROW_MASK, 0xf, false);
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
    // one step of the wave tree: combine with the lane the DPP control selects
    template <int CTRL, int ROW_MASK>
    __device__ __forceinline__ void dpp_step(bool withL)
    {
        dd_merge(hx, lx, dpp_f64<CTRL, ROW_MASK>(hx, 0.0), dpp_f64<CTRL, ROW_MASK>(lx, 0.0));
        dd_merge(hy, ly, dpp_f64<CTRL, ROW_MASK>(hy, 0.0), dpp_f64<CTRL, ROW_MASK>(ly, 0.0));
        dd_merge(hz, lz, dpp_f64<CTRL, ROW_MASK>(hz, 0.0), dpp_f64<CTRL, ROW_MASK>(lz, 0.0));
        if (withL)
        {
            sx += dpp_f64<CTRL, ROW_MASK>(sx, 0.0);
            sy += dpp_f64<CTRL, ROW_MASK>(sy, 0.0);
            sz += dpp_f64<CTRL, ROW_MASK>(sz, 0.0);
            lmin = min(lmin, dpp_i32<CTRL, ROW_MASK>(lmin, INT_MAX));
            lcnt += dpp_i32<CTRL, ROW_MASK>(lcnt, 0);
        }
    }
};

// ---- block reduction ---------------------------------------------------------------------------------------------
struct DD
{
    double hi, lo;
};
__device__ __forceinline__ DD dd_sel(bool c, DD a, DD b)
{
    DD r;
    r.hi = c ? a.hi : b.hi;
    r.lo = c ? a.lo : b.lo;
    return r;
}
__device__ __forceinline__ DD dd_sum(DD a, DD b)
{
    dd_merge(a.hi, a.lo, b.hi, b.lo);
    return a;
}
template <int CTRL>
__device__ __forceinline__ DD dd_dpp(DD v)
{
    DD r;
    r.hi = dpp_f64<CTRL, 0xF>(v.hi, 0.0);
    r.lo = dpp_f64<CTRL, 0xF>(v.lo, 0.0);
    return r;
}
__device__ __forceinline__ DD dd_shfl_xor(DD v, int mask)
{
    DD r;
    r.hi = __shfl_xor(v.hi, mask, kWave);
    r.lo = __shfl_xor(v.lo, mask, kWave);
    return r;
}

// Generic version (any BLOCK): wave tree in six DPP steps (xor 1, xor 2 inside quads; mirror inside 8 and inside 16
// lanes; row 0->1 and 2->3; rows 0-1 -> rows 2-3) that leave the wave total in lane 63, then one LDS hop and thread 0
// folds the waves in wave order.  18 double-double merges per wave.
template <int BLOCK>
__device__ __forceinline__ Accum block_reduce_generic(Accum a)
{
    constexpr int NW = BLOCK / kWave;
    __shared__ double s_d[NW][kNumPartDoubles];
    __shared__ int s_i[NW][kNumPartInts];
    const bool anyL = __any(a.lcnt != 0);
    a.dpp_step<0xB1, 0xF>(anyL);  // quad_perm [1,0,3,2]
    a.dpp_step<0x4E, 0xF>(anyL);  // quad_perm [2,3,0,1]
    a.dpp_step<0x141, 0xF>(anyL); // row_half_mirror
    a.dpp_step<0x140, 0xF>(anyL); // row_mirror: every lane of a row now holds the row total
    a.dpp_step<0x142, 0xA>(anyL); // row_bcast:15 into rows 1 and 3
    a.dpp_step<0x143, 0xC>(anyL); // row_bcast:31 into rows 2 and 3: lane 63 holds the wave total
    CAVMD_STAMP(2);
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    if (lane == kWave - 1)
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
        a.hx = s_d[0][0]; a.lx = s_d[0][1]; a.hy = s_d[0][2]; a.ly = s_d[0][3];
        a.hz = s_d[0][4]; a.lz = s_d[0][5]; a.sx = s_d[0][6]; a.sy = s_d[0][7];
        a.sz = s_d[0][8];
        a.lmin = s_i[0][0];
        a.lcnt = s_i[0][1];
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
    CAVMD_STAMP(3);
    return a;
}

// 256-thread version: recursive halving.  The three double-double components are treated as four slots {x, y, z, 0}.
//   xor 1 (quad_perm):  even lanes keep {x, y} and receive the partner's, odd lanes keep {z, 0}      2 merges
//   xor 2 (quad_perm):  lanes 0/2 of a quad split {x, y}, lanes 1/3 split {z, 0}                      1 merge
//                       -> lane&3 = 0: x, 1: z, 2: y, 3: nothing; from here on a lane carries ONE component
//   row_ror:4, row_ror:8 (stay on the same lane&3): every lane holds its component's 16-lane total     2 merges
//   xor 16, xor 32 (ds_bpermute, 4 dwords each): its 64-lane total                                    2 merges
// 7 merges instead of 18.  The four waves meet through LDS; lanes 0-15 of wave 0 (lane = 4*wave + slot) finish with
// the same two row rotations, and thread 0 picks y and z up from lanes 2 and 1 with quad_perm broadcasts.
// L-typed particles (normally one per system) go through a cheap six-step tree only in waves that saw one.
// Every step is a fixed permutation, so the result is bit-reproducible.  Returns the block total in thread 0.
__device__ __forceinline__ Accum block_reduce_256(Accum a)
{
    constexpr int NW = 4;
    __shared__ double s_main[NW][4][2];
    __shared__ double s_L[NW][3];
    __shared__ int s_Li[NW][2];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;

    // L-typed part: only in waves that hold one (wave-uniform branch); result in lane 63
    if (__any(a.lcnt != 0))
    {
#define CAVMD_L_STEP(CTRL, MASK)                                     \
    a.sx += dpp_f64<CTRL, MASK>(a.sx, 0.0);                          \
    a.sy += dpp_f64<CTRL, MASK>(a.sy, 0.0);                          \
    a.sz += dpp_f64<CTRL, MASK>(a.sz, 0.0);                          \
    a.lmin = min(a.lmin, dpp_i32<CTRL, MASK>(a.lmin, INT_MAX));      \
    a.lcnt += dpp_i32<CTRL, MASK>(a.lcnt, 0);
        CAVMD_L_STEP(0xB1, 0xF)
        CAVMD_L_STEP(0x4E, 0xF)
        CAVMD_L_STEP(0x141, 0xF)
        CAVMD_L_STEP(0x140, 0xF)
        CAVMD_L_STEP(0x142, 0xA)
        CAVMD_L_STEP(0x143, 0xC)
#undef CAVMD_L_STEP
    }

    const bool b0 = lane & 1, b1 = lane & 2;
    const DD X {a.hx, a.lx}, Y {a.hy, a.ly}, Z {a.hz, a.lz}, W {0.0, 0.0};
    // xor 1
    const DD P = dd_sum(dd_sel(b0, Z, X), dd_dpp<0xB1>(dd_sel(b0, X, Z)));
    const DD Q = dd_sum(dd_sel(b0, W, Y), dd_dpp<0xB1>(dd_sel(b0, Y, W)));
    // xor 2
    DD R = dd_sum(dd_sel(b1, Q, P), dd_dpp<0x4E>(dd_sel(b1, P, Q)));
    // inside the 16-lane row, staying on lane&3
    R = dd_sum(R, dd_dpp<0x124>(R)); // row_ror:4
    R = dd_sum(R, dd_dpp<0x128>(R)); // row_ror:8
    // across rows
    R = dd_sum(R, dd_shfl_xor(R, 16));
    R = dd_sum(R, dd_shfl_xor(R, 32));
    CAVMD_STAMP(2);

    if (lane < 4)
    {
        s_main[wave][lane][0] = R.hi;
        s_main[wave][lane][1] = R.lo;
    }
    if (lane == kWave - 1)
    {
        s_L[wave][0] = a.sx;
        s_L[wave][1] = a.sy;
        s_L[wave][2] = a.sz;
        s_Li[wave][0] = a.lmin;
        s_Li[wave][1] = a.lcnt;
    }
    __syncthreads();
    if (wave == 0)
    {
        DD T {0.0, 0.0};
        if (lane < 4 * NW)
        {
            T.hi = s_main[lane >> 2][lane & 3][0];
            T.lo = s_main[lane >> 2][lane & 3][1];
        }
        T = dd_sum(T, dd_dpp<0x124>(T));
        T = dd_sum(T, dd_dpp<0x128>(T)); // lanes 0..3: block totals of x, z, y, (nothing)
        const DD Tz = dd_dpp<0x55>(T);   // quad_perm [1,1,1,1]
        const DD Ty = dd_dpp<0xAA>(T);   // quad_perm [2,2,2,2]
        a.hx = T.hi; a.lx = T.lo;
        a.hy = Ty.hi; a.ly = Ty.lo;
        a.hz = Tz.hi; a.lz = Tz.lo;
        if (lane == 0)
        {
            a.sx = s_L[0][0]; a.sy = s_L[0][1]; a.sz = s_L[0][2];
            a.lmin = s_Li[0][0];
            a.lcnt = s_Li[0][1];
#pragma unroll
            for (int w = 1; w < NW; ++w)
            {
                a.sx += s_L[w][0]; a.sy += s_L[w][1]; a.sz += s_L[w][2];
                a.lmin = min(a.lmin, s_Li[w][0]);
                a.lcnt += s_Li[w][1];
            }
        }
    }
    CAVMD_STAMP(3);
    return a;
}

template <int BLOCK>
__device__ __forceinline__ Accum block_reduce(Accum a)
{
    if constexpr (BLOCK == 256)
        return block_reduce_256(a);
    else
        return block_reduce_generic<BLOCK>(a);
}

// ---- input layouts -------------------------------------------------------------------------------
// HOOMD-native AoS: Scalar4 pos (type tag in the low 32 bits of .w), Scalar charge, int3 image.
// NT: 0 = plain loads, 1 = pos and image non-temporal (read once per evaluation) but charge temporal (the force
// map reads it again), 2 = all three non-temporal.
template <int NT>
struct AosInputT
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
        const int* im = image + 3 * i;
        if (NT)
        {
            r.xy = __builtin_nontemporal_load(pos2 + 2 * i);
            r.zw = __builtin_nontemporal_load(pos2 + 2 * i + 1);
            r.c = (NT == 2) ? __builtin_nontemporal_load(charge + i) : charge[i];
            r.ix = __builtin_nontemporal_load(im + 0);
            r.iy = __builtin_nontemporal_load(im + 1);
            r.iz = __builtin_nontemporal_load(im + 2);
        }
        else
        {
            r.xy = pos2[2 * i];
            r.zw = pos2[2 * i + 1];
            r.c = charge[i];
            r.ix = im[0];
            r.iy = im[1];
            r.iz = im[2];
        }
        return r;
    }
    static __device__ __forceinline__ double x(const Raw& r) { return r.xy.x; }
    static __device__ __forceinline__ double y(const Raw& r) { return r.xy.y; }
    static __device__ __forceinline__ double z(const Raw& r) { return r.zw.x; }
    static __device__ __forceinline__ int tag(const Raw& r) { return __double2loint(r.zw.y); }
};
typedef AosInputT<0> AosInput;

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

// cavmd_params plus the two quotients the formulas need, divided once on the host (IEEE division is correctly rounded
// on host and device alike, so this changes no bit; it removes two ~150-cycle fp64 divisions from the prologue).
struct DeviceParams
{
    double g;    // couplstr
    double K;    // phmass * omegac^2
    double gK;   // g / K            (src/CavityForceCompute.cc:183)
    double g2K;  // g * g / K        (src/CavityForceCompute.cc:176)
};

// Where the per-block partials live (SoA so the finalize kernel reads them coalesced).
struct Partials
{
    double* d;      // kNumPartDoubles arrays of `stride` doubles
    int* i;         // kNumPartInts arrays of `stride` ints
    unsigned stride;
};

// ---- kernel 1: per-block partial dipole sums + photon search --------------------------------------
// One tile = BLOCK * UNROLL particles; block b takes tiles b, b + grid, ...  All UNROLL particles' loads of a lane are
// issued together (16 loads in flight per lane at UNROLL = 4; a scheduling barrier keeps hipcc from sinking them
// behind each other's waits).  PIPE = 1 additionally double-buffers tiles: the next tile's loads are issued before
// the current tile's arithmetic, so that the ~57 VALU operations per particle overlap with memory even at one wave
// per SIMD.
template <class Input, int UNROLL>
struct TileRegs
{
    typename Input::Raw raw[UNROLL];
};

template <class Input, int BLOCK, int UNROLL>
__device__ __forceinline__ void tile_load(const Input& in, size_t base, TileRegs<Input, UNROLL>& r)
{
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
        r.raw[u] = in.load(base + (size_t)u * BLOCK);
}

template <class Input, int BLOCK, int UNROLL>
__device__ __forceinline__ void tile_accumulate(const TileRegs<Input, UNROLL>& r, size_t base, double Lx, double Ly,
                                                double Lz, int L_typeid, Accum& acc)
{
    bool isL[UNROLL];
    bool any = false;
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
    {
        isL[u] = (Input::tag(r.raw[u]) == L_typeid);
        any = any || isL[u];
    }
    if (!__any(any))
    {
        // fast path (wave-uniform): no lane of this wave holds an L-typed particle in this tile
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
        {
            const double c = r.raw[u].c;
            dd_acc(acc.hx, acc.lx, c * (Input::x(r.raw[u]) + (double)r.raw[u].ix * Lx));
            dd_acc(acc.hy, acc.ly, c * (Input::y(r.raw[u]) + (double)r.raw[u].iy * Ly));
            dd_acc(acc.hz, acc.lz, c * (Input::z(r.raw[u]) + (double)r.raw[u].iz * Lz));
        }
    }
    else
    {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
        {
            const double rx = Input::x(r.raw[u]) + (double)r.raw[u].ix * Lx;
            const double ry = Input::y(r.raw[u]) + (double)r.raw[u].iy * Ly;
            const double rz = Input::z(r.raw[u]) + (double)r.raw[u].iz * Lz;
            acc.add((unsigned)(base + (size_t)u * BLOCK), rx, ry, rz, r.raw[u].c, Input::tag(r.raw[u]), L_typeid);
        }
    }
}

template <class Input, int BLOCK, int UNROLL, bool PIPE>
__global__ __launch_bounds__(BLOCK) void dipole_partials_kernel(Input in, unsigned N, double Lx, double Ly, double Lz,
                                                                int L_typeid, Partials part)
{
    constexpr unsigned TILE = BLOCK * UNROLL;
    Accum acc;
    const unsigned full_tiles = N / TILE;
    const unsigned G = gridDim.x;
    if (!PIPE)
    {
        for (unsigned t = blockIdx.x; t < full_tiles; t += G)
        {
            const size_t base = (size_t)t * TILE + threadIdx.x;
            TileRegs<Input, UNROLL> A;
            tile_load<Input, BLOCK, UNROLL>(in, base, A);
            __builtin_amdgcn_sched_barrier(0);
            tile_accumulate<Input, BLOCK, UNROLL>(A, base, Lx, Ly, Lz, L_typeid, acc);
        }
    }
    else
    {
        // ping-pong A/B so that no register copies are needed
        TileRegs<Input, UNROLL> A, B;
        unsigned t = blockIdx.x;
        if (t < full_tiles)
            tile_load<Input, BLOCK, UNROLL>(in, (size_t)t * TILE + threadIdx.x, A);
        while (t < full_tiles)
        {
            if (t + G < full_tiles)
                tile_load<Input, BLOCK, UNROLL>(in, (size_t)(t + G) * TILE + threadIdx.x, B);
            __builtin_amdgcn_sched_barrier(0);
            tile_accumulate<Input, BLOCK, UNROLL>(A, (size_t)t * TILE + threadIdx.x, Lx, Ly, Lz, L_typeid, acc);
            t += G;
            if (t >= full_tiles)
                break;
            if (t + G < full_tiles)
                tile_load<Input, BLOCK, UNROLL>(in, (size_t)(t + G) * TILE + threadIdx.x, A);
            __builtin_amdgcn_sched_barrier(0);
            tile_accumulate<Input, BLOCK, UNROLL>(B, (size_t)t * TILE + threadIdx.x, Lx, Ly, Lz, L_typeid, acc);
            t += G;
        }
    }
    // ragged tail: one block takes it, bounds-checked
    if (blockIdx.x == full_tiles % G)
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

// ---- final reduction + scalars (shared by the stand-alone finalize kernel and the fused force map) ------------
// Everything an evaluation produces besides the per-particle forces, as held by thread 0 of a block.
struct Scalars
{
    double d[3], dlo[3], q[3], Dq[2], e[3], f[3], dtot[3];
    int photon, nL;
};

// Folds the `nparts` per-block partials in a FIXED order (thread t takes partials t, t+BLOCK, ... in index order,
// then the fixed-shape block tree), unwraps the photon and evaluates energies, Dq and the photon force with the
// reference's operator association (src/CavityForceCompute.cc:169-183, 203-207).  All threads of the block must
// call it; the result is valid in thread 0 only.  Any block that calls it with the same arguments gets the same
// bits, which is what lets every block of the fused force map redo it instead of waiting on a separate launch.
template <class Input, int BLOCK>
__device__ __forceinline__ Scalars reduce_partials_and_finalize(const Input& in, unsigned N, unsigned nparts, double Lx,
                                                                double Ly, double Lz, const DeviceParams& prm,
                                                                const Partials& part, bool want_energies)
{
    // Speculative fetch of the last particle: the driver appends the photon last (examples/05_advanced_run.py:
    // 497-505), so this usually removes a dependent memory round trip after the reduction.
    CAVMD_STAMP(0);
    const typename Input::Raw guess = in.load((size_t)(N - 1));

    Accum acc;
    const unsigned s = part.stride;
    constexpr int BATCH = 4; // partial sets in flight per thread
    for (unsigned base = 0; base < nparts; base += BATCH * BLOCK)
    {
        Accum o[BATCH];
#pragma unroll
        for (int j = 0; j < BATCH; ++j)
        {
            const unsigned p = base + j * BLOCK + threadIdx.x;
            if (p < nparts)
            {
                o[j].hx = part.d[0 * s + p]; o[j].lx = part.d[1 * s + p];
                o[j].hy = part.d[2 * s + p]; o[j].ly = part.d[3 * s + p];
                o[j].hz = part.d[4 * s + p]; o[j].lz = part.d[5 * s + p];
                o[j].sx = part.d[6 * s + p]; o[j].sy = part.d[7 * s + p]; o[j].sz = part.d[8 * s + p];
                o[j].lmin = part.i[0 * s + p];
                o[j].lcnt = part.i[1 * s + p];
            }
        }
#pragma unroll
        for (int j = 0; j < BATCH; ++j)
            if (base + j * BLOCK < nparts) // block-uniform: skip batches nobody loaded
                acc.merge(o[j]);           // (a default-constructed Accum is the identity for the ragged last one)
    }
    CAVMD_STAMP(1);
    acc = block_reduce<BLOCK>(acc);

    Scalars sc;
    dd_norm(acc.hx, acc.lx);
    dd_norm(acc.hy, acc.ly);
    dd_norm(acc.hz, acc.lz);
    double dx = acc.hx, dy = acc.hy, dz = acc.hz;
    // all particles, L-typed included (the photon normally has charge 0, so this usually equals d)
    sc.dtot[0] = dx + acc.sx;
    sc.dtot[1] = dy + acc.sy;
    sc.dtot[2] = dz + acc.sz;
    const int photon = (acc.lmin == INT_MAX) ? -1 : acc.lmin;
    const double g = prm.g, K = prm.K;
    double qx = 0.0, qy = 0.0, qz = 0.0, eh = 0.0, ec = 0.0, ed = 0.0, Dqx = 0.0, Dqy = 0.0, fx = 0.0, fy = 0.0, fz = 0.0;
    if (threadIdx.x == 0)
    {
        if (photon >= 0)
        {
            typename Input::Raw r = guess;
            if ((unsigned)photon != N - 1)
                r = in.load((size_t)photon);
            qx = Input::x(r) + (double)r.ix * Lx;
            qy = Input::y(r) + (double)r.iy * Ly;
            qz = Input::z(r) + (double)r.iz * Lz;
            if (acc.lcnt > 1)
            {
                // Degenerate input (the driver enforces exactly one 'L', examples/05_advanced_run.py:548-550): the
                // reference skips only the FIRST L-typed particle in the dipole (src/CavityForceCompute.cc:122), so
                // the later ones are added back here.
                dx += acc.sx - r.c * qx;
                dy += acc.sy - r.c * qy;
                dz += acc.sz - r.c * qz;
            }
            if (want_energies)
            {
                // src/CavityForceCompute.cc:174-176, dot() = a.x*b.x + a.y*b.y + a.z*b.z
                eh = 0.5 * K * (qx * qx + qy * qy + qz * qz);
                ec = g * (dx * qx + dy * qy + 0.0 * 0.0);
                ed = 0.5 * prm.g2K * (dx * dx + dy * dy + 0.0 * 0.0);
            }
            // :183
            Dqx = qx + prm.gK * dx;
            Dqy = qy + prm.gK * dy;
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
    }
    CAVMD_STAMP(4);
    sc.d[0] = dx; sc.d[1] = dy; sc.d[2] = dz;
    sc.dlo[0] = acc.lx; sc.dlo[1] = acc.ly; sc.dlo[2] = acc.lz;
    sc.q[0] = qx; sc.q[1] = qy; sc.q[2] = qz;
    sc.Dq[0] = Dqx; sc.Dq[1] = Dqy;
    sc.e[0] = eh; sc.e[1] = ec; sc.e[2] = ed;
    sc.f[0] = fx; sc.f[1] = fy; sc.f[2] = fz;
    sc.photon = photon;
    sc.nL = acc.lcnt;
    return sc;
}

__device__ __forceinline__ void write_result(cavmd_result* __restrict__ res, const Scalars& sc, unsigned N,
                                             unsigned nparts, uint64_t sequence)
{
#pragma unroll
    for (int k = 0; k < 3; ++k)
    {
        res->dipole[k] = sc.d[k];
        res->q[k] = sc.q[k];
        res->energy[k] = sc.e[k];
        res->photon_force[k] = sc.f[k];
        res->dipole_lo[k] = sc.dlo[k];
        res->total_dipole[k] = sc.dtot[k];
    }
    res->Dq[0] = sc.Dq[0];
    res->Dq[1] = sc.Dq[1];
    res->photon_idx = sc.photon;
    res->n_photon_typed = sc.nL;
    res->n_particles = N;
    res->n_partials = nparts;
    res->sequence = sequence;
    res->reserved = 0.0;
}

// ---- kernel 2 (three-launch path): one block publishes the result block ---------------------------------------
template <class Input, int BLOCK>
__global__ __launch_bounds__(BLOCK) void finalize_kernel(Input in, unsigned N, unsigned nparts, double Lx, double Ly,
                                                         double Lz, DeviceParams prm, Partials part, uint64_t sequence,
                                                         cavmd_result* __restrict__ res)
{
    const Scalars sc = reduce_partials_and_finalize<Input, BLOCK>(in, N, nparts, Lx, Ly, Lz, prm, part, true);
    if (threadIdx.x == 0)
        write_result(res, sc, N, nparts, sequence);
}

// ---- force map, HOOMD AoS force array written as dense 16-byte chunks ------------------------------------------
template <bool NT>
__device__ __forceinline__ void store_chunk(v2d* p, v2d v)
{
    if (NT)
        __builtin_nontemporal_store(v, p);
    else
        *p = v;
}

// What every thread of the force map needs to know about the evaluation.
struct MapScalars
{
    double Dqx, Dqy, Fx, Fy, Fz;
    int photon, nL;
};

// Body shared by the three-launch and the fused force map.  Chunk k is 16 bytes: particle k>>1, half k&1.
// Even chunk = (Fx, Fy) = ((-g c) Dq_x, (-g c) Dq_y), odd chunk = (Fz, w) = (0, 0); the photon's chunks carry F_L.
// PRE: the caller has already loaded the charges of this block's first full tile into c_first (issued before its
// prologue so that their latency is hidden behind it).
template <int BLOCK, int UNROLL, bool NT, bool PRE>
__device__ __forceinline__ void force_map_body(const MapScalars m, const double* __restrict__ charge,
                                               const v2d* __restrict__ pos2, unsigned N, double g, int L_typeid,
                                               v2d* __restrict__ force2, const double (&c_first)[UNROLL])
{
    constexpr unsigned TILE = BLOCK * UNROLL;
    const double ng = -g;
    const size_t nchunks = 2 * (size_t)N;
    const size_t pchunk = m.photon >= 0 ? 2 * (size_t)m.photon : ~(size_t)0; // photon's first chunk
    const unsigned full_tiles = (unsigned)(nchunks / TILE);
    const bool odd = threadIdx.x & 1; // BLOCK and TILE are even, so the half is fixed per thread
    const v2d zero = {0.0, 0.0};

    if (m.photon < 0)
    {
        // no photon: all forces are zero (src/CavityForceCompute.cc:145-156)
        for (size_t k = (size_t)blockIdx.x * BLOCK + threadIdx.x; k < nchunks; k += (size_t)gridDim.x * BLOCK)
            store_chunk<NT>(force2 + k, zero);
        return;
    }

    if (m.nL <= 1)
    {
        for (unsigned t = blockIdx.x; t < full_tiles; t += gridDim.x)
        {
            const size_t base = (size_t)t * TILE + threadIdx.x;
            double c[UNROLL];
            if (PRE && t == blockIdx.x)
            {
#pragma unroll
                for (int u = 0; u < UNROLL; ++u)
                    c[u] = c_first[u];
            }
            else
            {
#pragma unroll
                for (int u = 0; u < UNROLL; ++u)
                    c[u] = charge[(base + (size_t)u * BLOCK) >> 1];
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
            {
                const size_t k = base + (size_t)u * BLOCK;
                const double s = ng * c[u]; // ((-g) * charge) * Dq, src/CavityForceCompute.cc:194
                v2d v = {s * m.Dqx, s * m.Dqy};
                v = odd ? zero : v;
                if ((k | 1) == (pchunk | 1))
                    v = odd ? (v2d) {m.Fz, 0.0} : (v2d) {m.Fx, m.Fy};
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
                    v2d v = {s * m.Dqx, s * m.Dqy};
                    v = odd ? zero : v;
                    if ((k | 1) == (pchunk | 1))
                        v = odd ? (v2d) {m.Fz, 0.0} : (v2d) {m.Fx, m.Fy};
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
        v2d v = {s * m.Dqx, s * m.Dqy};
        v = (odd || tag == L_typeid) ? zero : v;
        if ((k | 1) == (pchunk | 1))
            v = odd ? (v2d) {m.Fz, 0.0} : (v2d) {m.Fx, m.Fy};
        store_chunk<NT>(force2 + k, v);
    }
}

// three-launch path: scalars come from the result block the finalize kernel wrote
template <int BLOCK, int UNROLL, bool NT>
__global__ __launch_bounds__(BLOCK) void force_map_aos_kernel(const double* __restrict__ charge,
                                                              const v2d* __restrict__ pos2, // only read if several L-typed
                                                              unsigned N, double g, int L_typeid,
                                                              const cavmd_result* __restrict__ res, v2d* __restrict__ force2)
{
    MapScalars m;
    m.Dqx = res->Dq[0]; m.Dqy = res->Dq[1];
    m.Fx = res->photon_force[0]; m.Fy = res->photon_force[1]; m.Fz = res->photon_force[2];
    m.photon = res->photon_idx;
    m.nL = res->n_photon_typed;
    const double none[UNROLL] = {};
    force_map_body<BLOCK, UNROLL, NT, false>(m, charge, pos2, N, g, L_typeid, force2, none);
}

// two-launch path: every block folds the partials itself (same fixed order -> same bits in every block), block 0
// publishes the result block; no separate finalize launch and no inter-workgroup hand-off inside the launch.
template <int BLOCK, int UNROLL, bool NT>
__global__ __launch_bounds__(BLOCK) void force_map_aos_fused_kernel(AosInput in, unsigned N, unsigned nparts, double Lx,
                                                                    double Ly, double Lz, DeviceParams prm, int L_typeid,
                                                                    Partials part, uint64_t sequence,
                                                                    cavmd_result* __restrict__ res,
                                                                    v2d* __restrict__ force2)
{
    __shared__ double s_m[5];
    __shared__ int s_mi[2];
    // charges of the first tile: independent of the prologue, so issue them first
    double c_first[UNROLL] = {};
    {
        constexpr unsigned TILE = BLOCK * UNROLL;
        const unsigned full_tiles = (unsigned)((2 * (size_t)N) / TILE);
        if (blockIdx.x < full_tiles)
        {
            const size_t base = (size_t)blockIdx.x * TILE + threadIdx.x;
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
                c_first[u] = in.charge[(base + (size_t)u * BLOCK) >> 1];
        }
    }
    const Scalars sc = reduce_partials_and_finalize<AosInput, BLOCK>(in, N, nparts, Lx, Ly, Lz, prm, part, blockIdx.x == 0);
    if (threadIdx.x == 0)
    {
        s_m[0] = sc.Dq[0]; s_m[1] = sc.Dq[1]; s_m[2] = sc.f[0]; s_m[3] = sc.f[1]; s_m[4] = sc.f[2];
        s_mi[0] = sc.photon;
        s_mi[1] = sc.nL;
        if (blockIdx.x == 0)
            write_result(res, sc, N, nparts, sequence);
    }
    __syncthreads();
    MapScalars m;
    m.Dqx = s_m[0]; m.Dqy = s_m[1]; m.Fx = s_m[2]; m.Fy = s_m[3]; m.Fz = s_m[4];
    m.photon = s_mi[0];
    m.nL = s_mi[1];
    force_map_body<BLOCK, UNROLL, NT, true>(m, in.charge, in.pos2, N, prm.g, L_typeid, force2, c_first);
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
    const unsigned gw = blockIdx.x * NW + wave, GW = gridDim.x * NW;
    for (unsigned tile = gw; tile < ntiles; tile += GW)
    {
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
        for (int j = 0; j < cnt; ++j)
        {
            const double x = readlane_f64(px, j), y = readlane_f64(py, j), z = readlane_f64(pz, j);
            const double kr = (x * kx + y * ky) + z * kz;
            double s, c;
            if (__any(!(fabs(kr) < 1.0e8))) // wave-uniform; also catches NaN/Inf
                sincos(kr, &s, &c);
            else
                sincos_reduced(coef, kr, s, c);
            re += c;
            im += s;
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
__global__ void cavity_mode_kernel(const cavmd_result* __restrict__ res, const cavmd_double4* __restrict__ vel, double kB,
                                   double* __restrict__ out)
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

template <int BLOCK, int UNROLL>
__global__ __launch_bounds__(BLOCK) void force_mass_partials_kernel(const v2d* __restrict__ force2,
                                                                    const v2d* __restrict__ vel2, unsigned N,
                                                                    double* __restrict__ part /* [2][gridDim] */)
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
    if (threadIdx.x == 0)
    {
        part[blockIdx.x] = acc.hi;
        part[gridDim.x + blockIdx.x] = acc.lo;
    }
}

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void force_mass_fold_kernel(const double* __restrict__ part, unsigned nparts,
                                                                double* __restrict__ out)
{
    DD acc {0.0, 0.0};
    for (unsigned p = threadIdx.x; p < nparts; p += BLOCK)
        dd_merge(acc.hi, acc.lo, part[p], part[nparts + p]);
    acc = block_reduce_dd1<BLOCK>(acc);
    if (threadIdx.x == 0)
        out[0] = acc.hi + acc.lo;
}
} // namespace cavmd
