// cavmd_persistent_kernel.hpp -- the whole cavity-force evaluation in ONE launch (2048 < N <~ 5e6).
//
//   phase 1   every block streams its tiles (pos 32 + charge 8 + image 12 B per particle) exactly as
//             dipole_partials_kernel does -- same tile assignment, same double-double accumulation, same block tree,
//             hence the same per-block partial, bit for bit -- and parks the charges it read in LDS.
//   hand-off  thread 0 of every block publishes the block's partial as 8-byte {tag, 32-bit value} granules; every block
//             then waits until the granules of ALL blocks carry this evaluation's tag and folds them in the same fixed
//             order as reduce_partials_and_finalize (thread t: records t, t + BLOCK, ...; then the block tree), so every
//             block obtains the same bits and no float atomics are involved.  Block 0 publishes cavmd_result.
//   phase 2   every block writes the forces of its own tiles from the charges in LDS (dense 16-byte chunks): the
//             charge array is not read a second time (84 instead of 92 bytes per particle cross the memory bus) and the
//             second launch with its ramp, drain and re-fold prologue disappears.
//
// Inter-workgroup protocol (cdna_hip_programming.md, Guideline 16, form R2 "the data is the flag"): each granule is ONE
// naturally aligned 8-byte relaxed agent-scope atomic (global_store/load_dwordx2 sc0 sc1), carries its own tag and is
// validated individually by the reader, so no release/acquire ordering between granules, no separate flag, no fence and
// no dependence on dispatch order, timing or XCD placement are needed -- only that all blocks of the grid are resident
// together (grid <= CUs x blocks per CU, checked on the host against the occupancy query).  The tag is the epoch word of
// the workspace, read from DEVICE memory at kernel start and advanced by block 0 once its wait has succeeded (every
// block has published by then, so every block has read it): a captured launch replays correctly, nothing needs zeroing
// between launches.  Every spin is bounded: on a time-out the block raises the sync_error word of the host-visible result
// block and fills its share of the force array with NaN, then exits like the others.
#pragma once

#include "cavmd_force_kernels.hpp"

#pragma clang fp contract(off)

namespace cavmd
{

constexpr int kGranulesPerRecord = 2 * kNumPartDoubles + kNumPartInts; // 9 doubles as 18 halves + lmin + lcnt
constexpr unsigned kSpinLimit = 4000000;                               // ~ seconds; a healthy wait is microseconds

struct SyncState
{
    unsigned long long* granules; // [kGranulesPerRecord][stride] {tag << 32 | value}
    unsigned* epoch;              // tag of the next evaluation (never 0)
    unsigned stride;
};

__device__ __forceinline__ void granule_store(unsigned long long* g, unsigned tag, unsigned value)
{
    __hip_atomic_store(g, ((unsigned long long)tag << 32) | value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long granule_load(const unsigned long long* g)
{
    return __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// thread 0 of a block: its block total -> granules of record `b`
__device__ __forceinline__ void publish_record(const SyncState& st, unsigned b, unsigned tag, const Accum& a)
{
    const double d[kNumPartDoubles] = {a.hx, a.lx, a.hy, a.ly, a.hz, a.lz, a.sx, a.sy, a.sz};
    unsigned long long* g = st.granules + b;
#pragma unroll
    for (int i = 0; i < kNumPartDoubles; ++i)
    {
        granule_store(g + (size_t)(2 * i) * st.stride, tag, (unsigned)__double2loint(d[i]));
        granule_store(g + (size_t)(2 * i + 1) * st.stride, tag, (unsigned)__double2hiint(d[i]));
    }
    granule_store(g + (size_t)(2 * kNumPartDoubles) * st.stride, tag, (unsigned)a.lmin);
    granule_store(g + (size_t)(2 * kNumPartDoubles + 1) * st.stride, tag, (unsigned)a.lcnt);
}

// Wait until record `r` carries `tag` in every granule, then return it.  POLL_ONE: spin on the granule that was stored
// last and fetch the other 19 only once it shows the tag (all 20 are validated either way).  Returns false on time-out.
template <bool POLL_ONE>
__device__ __forceinline__ bool wait_record(const SyncState& st, unsigned r, unsigned tag, Accum& o)
{
    const unsigned long long* g = st.granules + r;
    unsigned v[kGranulesPerRecord];
    unsigned spins = 0;
    for (;;)
    {
        bool ok = true;
        if (POLL_ONE)
        {
            const unsigned long long x = granule_load(g + (size_t)(kGranulesPerRecord - 1) * st.stride);
            ok = (unsigned)(x >> 32) == tag;
        }
        if (ok)
        {
#pragma unroll
            for (int k = 0; k < kGranulesPerRecord; ++k)
            {
                const unsigned long long x = granule_load(g + (size_t)k * st.stride);
                v[k] = (unsigned)x;
                ok = ok && ((unsigned)(x >> 32) == tag);
            }
        }
        if (ok)
            break;
        if (++spins > kSpinLimit)
            return false;
        __builtin_amdgcn_s_sleep(4);
    }
    o.hx = __hiloint2double((int)v[1], (int)v[0]);   o.lx = __hiloint2double((int)v[3], (int)v[2]);
    o.hy = __hiloint2double((int)v[5], (int)v[4]);   o.ly = __hiloint2double((int)v[7], (int)v[6]);
    o.hz = __hiloint2double((int)v[9], (int)v[8]);   o.lz = __hiloint2double((int)v[11], (int)v[10]);
    o.sx = __hiloint2double((int)v[13], (int)v[12]); o.sy = __hiloint2double((int)v[15], (int)v[14]);
    o.sz = __hiloint2double((int)v[17], (int)v[16]);
    o.lmin = (int)v[18];
    o.lcnt = (int)v[19];
    return true;
}

extern __shared__ __attribute__((aligned(16))) double s_dyn_charge[];

template <int BLOCK, int UNROLL, bool NT_STORE, bool POLL_ONE>
__global__ __launch_bounds__(BLOCK) void cavity_persistent_kernel(AosInputT<2> in, unsigned N, double Lx, double Ly, double Lz,
                                                                  DeviceParams prm, int L_typeid, SyncState st,
                                                                  uint64_t sequence, cavmd_result* __restrict__ res,
                                                                  HostResult* __restrict__ res_host,
                                                                  v2d* __restrict__ force2)
{
    constexpr unsigned TILE = BLOCK * UNROLL; // particles per tile; the same tile is 2 * TILE force chunks
    constexpr int MU = 2 * UNROLL;            // 16-byte chunk stores per thread and tile
    __shared__ double s_m[5];
    __shared__ int s_mi[3];
    double* s_charge = s_dyn_charge;
    const unsigned G = gridDim.x, b = blockIdx.x, tid = threadIdx.x;

    // this evaluation's tag, and the speculative photon row (the driver appends the photon last)
    const unsigned tag = __hip_atomic_load(st.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const PhotonRow guess = photon_row(in, (size_t)(N - 1));

    // ---- phase 1: partial dipole of this block's tiles, charges parked in LDS ------------------------------------------
    Accum acc;
    const unsigned full_tiles = N / TILE;
    unsigned slot = 0; // tiles of this block so far
    for (unsigned t = b; t < full_tiles; t += G, ++slot)
    {
        const size_t base = (size_t)t * TILE + tid;
        TileRegs<AosInputT<2>, UNROLL> A;
        tile_load<AosInputT<2>, BLOCK, UNROLL>(in, base, A);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
            s_charge[slot * TILE + u * BLOCK + tid] = A.raw[u].c;
        tile_accumulate<AosInputT<2>, BLOCK, UNROLL>(A, base, Lx, Ly, Lz, L_typeid, acc);
    }
    const bool has_tail = (b == full_tiles % G) && (full_tiles * TILE < N);
    if (has_tail)
    {
        const size_t base = (size_t)full_tiles * TILE + tid;
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
        {
            const size_t i = base + (size_t)u * BLOCK;
            double c = 0.0;
            if (i < N)
            {
                const typename AosInputT<2>::Raw r = in.load(i);
                const double rx = AosInputT<2>::x(r) + (double)r.ix * Lx;
                const double ry = AosInputT<2>::y(r) + (double)r.iy * Ly;
                const double rz = AosInputT<2>::z(r) + (double)r.iz * Lz;
                acc.add((unsigned)i, rx, ry, rz, r.c, AosInputT<2>::tag(r), L_typeid);
                c = r.c;
            }
            s_charge[slot * TILE + u * BLOCK + tid] = c;
        }
    }
    acc = block_reduce<BLOCK>(acc);

    // ---- hand-off: publish this block's partial, gather everybody's, fold in the fixed order -----------------------------
    if (tid == 0)
        publish_record(st, b, tag, acc);
    Accum tot;
    bool failed = false;
    for (unsigned r = tid; r < G; r += BLOCK)
    {
        Accum o;
        if (!wait_record<POLL_ONE>(st, r, tag, o))
        {
            failed = true;
            break;
        }
        tot.merge(o);
    }
    // (the barrier also separates the two uses of the block tree's LDS arrays)
    const bool any_failed = __syncthreads_or(failed);
    tot = block_reduce<BLOCK>(tot);
    const Scalars sc = scalars_from_total<AosInputT<2>>(tot, guess, in, N, Lx, Ly, Lz, prm, b == 0);
    if (tid == 0)
    {
        s_m[0] = sc.Dq[0]; s_m[1] = sc.Dq[1]; s_m[2] = sc.f[0]; s_m[3] = sc.f[1]; s_m[4] = sc.f[2];
        s_mi[0] = sc.photon;
        s_mi[1] = sc.nL;
        s_mi[2] = any_failed;
        if (any_failed)
            __hip_atomic_store(&res_host->sync_error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (b == 0)
        {
            // every block has published, hence read the epoch: advance it for the next launch (0 is never a tag)
            __hip_atomic_store(st.epoch, tag + 1u ? tag + 1u : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!any_failed)
            {
                write_result(res, sc, N, G, sequence);
                publish_to_host(res_host, sc, N, G, sequence);
            }
        }
    }
    __syncthreads();
    MapScalars m;
    m.Dqx = s_m[0]; m.Dqy = s_m[1]; m.Fx = s_m[2]; m.Fy = s_m[3]; m.Fz = s_m[4];
    m.photon = s_mi[0];
    m.nL = s_mi[1];
    const bool bad = s_mi[2];

    // ---- phase 2: forces of this block's own tiles, charges from LDS ------------------------------------------------------
    const size_t nchunks = 2 * (size_t)N;
    const v2d zero = {0.0, 0.0};
    if (bad || m.photon < 0 || m.nL > 1)
    {
        // time-out (NaN: loud), no photon (zeros, src/CavityForceCompute.cc:145-156), or several L-typed particles (the
        // type tag has to be read: only non-L particles get a molecular force, :190-191) -- all rare, all from global memory
        const double nan = __builtin_nan("");
        const double ng = -prm.g;
        const bool odd = tid & 1;
        for (unsigned t = b; (size_t)t * TILE < N; t += G)
        {
#pragma unroll
            for (int u = 0; u < MU; ++u)
            {
                const size_t k = (size_t)t * 2 * TILE + (size_t)u * BLOCK + tid;
                if (k >= nchunks)
                    continue;
                v2d v = zero;
                if (bad)
                    v = (v2d) {nan, nan};
                else if (m.photon >= 0)
                {
                    const size_t p = k >> 1;
                    const int ptag = __double2loint(in.pos2[2 * p + 1].y);
                    const double s = ng * in.charge[p];
                    v = (v2d) {s * m.Dqx, s * m.Dqy};
                    v = (odd || ptag == L_typeid) ? zero : v;
                    if (p == (size_t)m.photon)
                        v = odd ? (v2d) {m.Fz, 0.0} : (v2d) {m.Fx, m.Fy};
                }
                store_chunk<NT_STORE>(force2 + k, v);
            }
        }
        return;
    }

    const double ng = -prm.g;
    const size_t pchunk = 2 * (size_t)m.photon; // photon's first chunk
    const bool odd = tid & 1;                   // BLOCK is even, so the half is fixed per thread
    slot = 0;
    for (unsigned t = b; t < full_tiles; t += G, ++slot)
    {
        const size_t base = (size_t)t * 2 * TILE + tid;
        double c[MU];
#pragma unroll
        for (int u = 0; u < MU; ++u)
            c[u] = s_charge[slot * TILE + ((u * BLOCK + tid) >> 1)];
#pragma unroll
        for (int u = 0; u < MU; ++u)
        {
            const size_t k = base + (size_t)u * BLOCK;
            const double s = ng * c[u]; // ((-g) * charge) * Dq, src/CavityForceCompute.cc:194
            v2d v = {s * m.Dqx, s * m.Dqy};
            v = odd ? zero : v;
            if ((k | 1) == (pchunk | 1))
                v = odd ? (v2d) {m.Fz, 0.0} : (v2d) {m.Fx, m.Fy};
            store_chunk<NT_STORE>(force2 + k, v);
        }
    }
    if (has_tail)
    {
        const size_t base = (size_t)full_tiles * 2 * TILE + tid;
#pragma unroll
        for (int u = 0; u < MU; ++u)
        {
            const size_t k = base + (size_t)u * BLOCK;
            if (k < nchunks)
            {
                const double s = ng * s_charge[slot * TILE + ((u * BLOCK + tid) >> 1)];
                v2d v = {s * m.Dqx, s * m.Dqy};
                v = odd ? zero : v;
                if ((k | 1) == (pchunk | 1))
                    v = odd ? (v2d) {m.Fz, 0.0} : (v2d) {m.Fx, m.Fy};
                store_chunk<NT_STORE>(force2 + k, v);
            }
        }
    }
}

} // namespace cavmd
