// cavmd_reduce.hpp -- fixed-order compensated reductions on CDNA4: double-double (TwoSum) arithmetic, DPP cross-lane
// movement, the per-lane accumulator of the dipole reduction and the block trees built from them.
// Part of the device code of libcavmd (see cavmd_kernels.hpp for the overview).
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
#if defined(CAVMD_DIAG_ACC_COST) // micro-benchmark diagnostics only (WRONG or padded arithmetic): what the accumulation costs
#if CAVMD_DIAG_ACC_COST == 0     // plain double sum: 1 instruction instead of 7
    hi += t;
    return;
#else                            // the compensated sum TWICE (the second into a dead-looking but live copy): 14 instead of 7
    {
        const double s2 = lo + t;
        const double b2 = s2 - lo;
        const double e2 = (lo - (s2 - b2)) + (t - b2);
        asm volatile("" ::"v"(s2), "v"(e2));
    }
#endif
#endif
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

// Fold of the 16 lanes of one DPP row (lanes 16 r .. 16 r + 15 of a wave) in a fixed order: the recursive halving of
// block_reduce_256 stopped at the row (5 double-double merges).  The total is valid in every lane with (lane & 3) == 0,
// in particular in the row's first lane.  All 64 lanes of the wave must be active (DPP reads neighbours).
// Used for the two-level fold of the per-block partials: 16 consecutive partials -> group total -> total of <= 16 groups,
// by the fused force map (all in one block) and by the single-launch kernel (group leaders, then every block).
__device__ __forceinline__ Accum row_fold16(Accum a)
{
    if (__any(a.lcnt != 0))
    {
#define CAVMD_L_STEP(CTRL)                                          \
    a.sx += dpp_f64<CTRL, 0xF>(a.sx, 0.0);                          \
    a.sy += dpp_f64<CTRL, 0xF>(a.sy, 0.0);                          \
    a.sz += dpp_f64<CTRL, 0xF>(a.sz, 0.0);                          \
    a.lmin = min(a.lmin, dpp_i32<CTRL, 0xF>(a.lmin, INT_MAX));      \
    a.lcnt += dpp_i32<CTRL, 0xF>(a.lcnt, 0);
        CAVMD_L_STEP(0xB1)  // quad_perm [1,0,3,2]
        CAVMD_L_STEP(0x4E)  // quad_perm [2,3,0,1]
        CAVMD_L_STEP(0x141) // row_half_mirror
        CAVMD_L_STEP(0x140) // row_mirror: every lane of the row holds the row's L totals
#undef CAVMD_L_STEP
    }
    const int lane = threadIdx.x & (kWave - 1);
    const bool b0 = lane & 1, b1 = lane & 2;
    const DD X {a.hx, a.lx}, Y {a.hy, a.ly}, Z {a.hz, a.lz}, W {0.0, 0.0};
    const DD P = dd_sum(dd_sel(b0, Z, X), dd_dpp<0xB1>(dd_sel(b0, X, Z)));
    const DD Q = dd_sum(dd_sel(b0, W, Y), dd_dpp<0xB1>(dd_sel(b0, Y, W)));
    DD R = dd_sum(dd_sel(b1, Q, P), dd_dpp<0x4E>(dd_sel(b1, P, Q)));
    R = dd_sum(R, dd_dpp<0x124>(R)); // row_ror:4
    R = dd_sum(R, dd_dpp<0x128>(R)); // row_ror:8 -> lane&3 = 0: x, 1: z, 2: y totals of the row
    const DD Tz = dd_dpp<0x55>(R);   // quad_perm [1,1,1,1]
    const DD Ty = dd_dpp<0xAA>(R);   // quad_perm [2,2,2,2]
    a.hx = R.hi; a.lx = R.lo;
    a.hy = Ty.hi; a.ly = Ty.lo;
    a.hz = Tz.hi; a.lz = Tz.lo;
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

} // namespace cavmd
