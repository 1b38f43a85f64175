// cavmd_force_kernels.hpp -- the cavity-force kernels: input layouts, dipole_partials_kernel, the fixed-order fold +
// scalars (reduce_partials_and_finalize), finalize_kernel and the force-map kernels.  Overview: cavmd_kernels.hpp.
#pragma once

#include "cavmd_reduce.hpp"

#pragma clang fp contract(off)

namespace cavmd
{

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
        r.px = __builtin_nontemporal_load(p + 0);
        r.py = __builtin_nontemporal_load(p + 1);
        r.pz = __builtin_nontemporal_load(p + 2);
        r.c = *reinterpret_cast<const double*>(chg + i * chg_stride); // temporal: the force map reads it again
        const int* im = reinterpret_cast<const int*>(img + i * img_stride);
        r.ix = __builtin_nontemporal_load(im + 0);
        r.iy = __builtin_nontemporal_load(im + 1);
        r.iz = __builtin_nontemporal_load(im + 2);
        r.t = __builtin_nontemporal_load(reinterpret_cast<const int*>(tid + i * tid_stride));
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

// The speculatively fetched last particle (the driver appends the photon last), as plain scalars: handing the
// layout's whole Raw aggregate through the prologue made hipcc keep it in a stack object (52 B of scratch stores per
// thread of every block of the fused force map: 1.19x its algorithmic HBM traffic at N = 1e6).
struct PhotonRow
{
    double x, y, z;
    int ix, iy, iz;
};
template <class Input>
__device__ __forceinline__ PhotonRow photon_row(const Input& in, size_t i)
{
    const typename Input::Raw r = in.load(i);
    PhotonRow p;
    p.x = Input::x(r); p.y = Input::y(r); p.z = Input::z(r);
    p.ix = r.ix; p.iy = r.iy; p.iz = r.iz;
    return p;
}

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

// From the block total (valid in thread 0) to everything else: renormalise, unwrap the photon and evaluate energies,
// Dq and the photon force with the reference's operator association (src/CavityForceCompute.cc:169-183, 203-207).
// `guess` is the speculatively fetched last particle (the driver appends the photon last).
template <class Input>
__device__ __forceinline__ Scalars scalars_from_total(Accum acc, const PhotonRow guess, const Input& in,
                                                      unsigned N, double Lx, double Ly, double Lz,
                                                      const DeviceParams& prm, bool want_energies)
{
    Scalars sc;
    // `want_energies` marks the one block that publishes cavmd_result: only it needs d_z, the all-particle dipole and the
    // energies; every other block is on its way to the force map and needs Dq and F_L only.
    dd_norm(acc.hx, acc.lx);
    dd_norm(acc.hy, acc.ly);
    if (want_energies)
        dd_norm(acc.hz, acc.lz);
    double dx = acc.hx, dy = acc.hy, dz = acc.hz;
    sc.dtot[0] = sc.dtot[1] = sc.dtot[2] = 0.0;
    if (want_energies)
    {
        // all particles, L-typed included (the photon normally has charge 0, so this usually equals d): the L-typed sum
        // joins the double-double BEFORE the final rounding, so the low word is not thrown away when the two terms cancel
        double th = acc.hx, tl = acc.lx;
        dd_acc(th, tl, acc.sx);
        sc.dtot[0] = th + tl;
        th = acc.hy; tl = acc.ly;
        dd_acc(th, tl, acc.sy);
        sc.dtot[1] = th + tl;
        th = acc.hz; tl = acc.lz;
        dd_acc(th, tl, acc.sz);
        sc.dtot[2] = th + tl;
    }
    const int photon = (acc.lmin == INT_MAX) ? -1 : acc.lmin;
    const double g = prm.g, K = prm.K;
    double qx = 0.0, qy = 0.0, qz = 0.0, eh = 0.0, ec = 0.0, ed = 0.0, Dqx = 0.0, Dqy = 0.0, fx = 0.0, fy = 0.0, fz = 0.0;
    if (threadIdx.x == 0)
    {
        if (photon >= 0)
        {
            // field-by-field select between the speculative row and the fallback load
            double px = guess.x, py = guess.y, pz = guess.z;
            int pix = guess.ix, piy = guess.iy, piz = guess.iz;
            if ((unsigned)photon != N - 1)
            {
                const PhotonRow r = photon_row(in, (size_t)photon);
                px = r.x; py = r.y; pz = r.z;
                pix = r.ix; piy = r.iy; piz = r.iz;
            }
            qx = px + (double)pix * Lx;
            qy = py + (double)piy * Ly;
            qz = pz + (double)piz * Lz;
            if (acc.lcnt > 1)
            {
                // Degenerate input (the driver enforces exactly one 'L', examples/05_advanced_run.py:548-550): the
                // reference skips only the FIRST L-typed particle in the dipole (src/CavityForceCompute.cc:122), so
                // the later ones are added back here.  (Rare: the photon's charge is fetched here, not speculatively.)
                const double pc = in.load((size_t)photon).c;
                dx += acc.sx - pc * qx;
                dy += acc.sy - pc * qy;
                dz += acc.sz - pc * qz;
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
    const PhotonRow guess = photon_row(in, (size_t)(N - 1));

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
    if (BLOCK == 256 && nparts <= 256)
    {
        // Two-level fold, the same tree the single-launch kernel walks across workgroups (cavmd_persistent_kernel.hpp):
        // 16 consecutive partials -> group total (one DPP row each), then the <= 16 group totals.
        __shared__ double s_gd[16][kNumPartDoubles];
        __shared__ int s_gi[16][kNumPartInts];
        acc = row_fold16(acc);
        CAVMD_STAMP(2);
        if ((threadIdx.x & 15) == 0)
        {
            const int gidx = threadIdx.x >> 4;
            s_gd[gidx][0] = acc.hx; s_gd[gidx][1] = acc.lx; s_gd[gidx][2] = acc.hy; s_gd[gidx][3] = acc.ly;
            s_gd[gidx][4] = acc.hz; s_gd[gidx][5] = acc.lz; s_gd[gidx][6] = acc.sx; s_gd[gidx][7] = acc.sy;
            s_gd[gidx][8] = acc.sz;
            s_gi[gidx][0] = acc.lmin;
            s_gi[gidx][1] = acc.lcnt;
        }
        __syncthreads();
        if (threadIdx.x < kWave)
        {
            Accum o;
            if (threadIdx.x < (nparts + 15) / 16)
            {
                const int gidx = threadIdx.x;
                o.hx = s_gd[gidx][0]; o.lx = s_gd[gidx][1]; o.hy = s_gd[gidx][2]; o.ly = s_gd[gidx][3];
                o.hz = s_gd[gidx][4]; o.lz = s_gd[gidx][5]; o.sx = s_gd[gidx][6]; o.sy = s_gd[gidx][7];
                o.sz = s_gd[gidx][8];
                o.lmin = s_gi[gidx][0];
                o.lcnt = s_gi[gidx][1];
            }
            Accum t;
            t.merge(o);
            acc = row_fold16(t);
        }
        CAVMD_STAMP(3);
    }
    else
        acc = block_reduce<BLOCK>(acc);
    return scalars_from_total<Input>(acc, guess, in, N, Lx, Ly, Lz, prm, want_energies);
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

// Host hand-off of the result block: `host` points at mapped pinned host memory.  The block is written first, then the
// evaluation's sequence number is release-stored at SYSTEM scope into host->ready, so a host thread that acquires
// ready == sequence sees the whole block (cavmd_result_read spins on it instead of paying a stream synchronisation).
struct HostResult
{
    cavmd_result result;
    uint64_t ready;
    unsigned sync_error; // single-launch kernel, starved evaluation: kSyncFailed / kSyncRepaired (cavmd_persistent_kernel.hpp)
    unsigned pad;
};
__device__ __forceinline__ void publish_to_host(HostResult* __restrict__ host, const Scalars& sc, unsigned N,
                                                unsigned nparts, uint64_t sequence)
{
    write_result(&host->result, sc, N, nparts, sequence);
    __hip_atomic_store(&host->ready, sequence, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- kernel 2 (three-launch path): one block publishes the result block ---------------------------------------
template <class Input, int BLOCK>
__global__ __launch_bounds__(BLOCK) void finalize_kernel(Input in, unsigned N, unsigned nparts, double Lx, double Ly,
                                                         double Lz, DeviceParams prm, Partials part, uint64_t sequence,
                                                         cavmd_result* __restrict__ res, HostResult* __restrict__ res_host)
{
    const Scalars sc = reduce_partials_and_finalize<Input, BLOCK>(in, N, nparts, Lx, Ly, Lz, prm, part, true);
    if (threadIdx.x == 0)
    {
        write_result(res, sc, N, nparts, sequence);
        publish_to_host(res_host, sc, N, nparts, sequence);
    }
}

// ---- force map, HOOMD AoS force array written as dense 16-byte chunks ------------------------------------------
// Store policy of the force array: 0 plain (write-back), 1 non-temporal, 2 write-through (sc1: the bytes leave the XCD's L2
// while the kernel runs instead of at its end-of-kernel release; after 320 MB of non-temporal stores at N = 1e7 the next
// kernel in the stream started 8.7 us late).
template <int NT>
__device__ __forceinline__ void store_chunk(v2d* p, v2d v)
{
    if (NT == 2)
        asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
    else if (NT == 1)
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
template <int BLOCK, int UNROLL, int NT, bool PRE>
__device__ __forceinline__ void force_map_body(const MapScalars m, const double* __restrict__ charge,
                                               const v2d* __restrict__ pos2, unsigned N, double g, int L_typeid,
                                               v2d* __restrict__ force2, const double (&c_first)[UNROLL],
                                               bool reverse = false)
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
        // Tiles blockIdx.x, blockIdx.x + grid, ...; `reverse` walks the same tiles from the last to the first, so that
        // the charge lines the reduction touched most recently (same XCD, see cavmd_capi.hip) are asked for first.
        const unsigned count = blockIdx.x < full_tiles ? (full_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
        for (unsigned j = 0; j < count; ++j)
        {
            const unsigned t = blockIdx.x + (reverse ? count - 1 - j : j) * gridDim.x;
            const size_t base = (size_t)t * TILE + threadIdx.x;
            double c[UNROLL];
            if (PRE && j == 0)
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
template <int BLOCK, int UNROLL, int NT>
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
template <int BLOCK, int UNROLL, int NT>
__global__ __launch_bounds__(BLOCK) void force_map_aos_fused_kernel(AosInput in, unsigned N, unsigned nparts, double Lx,
                                                                    double Ly, double Lz, DeviceParams prm, int L_typeid,
                                                                    Partials part, uint64_t sequence,
                                                                    cavmd_result* __restrict__ res,
                                                                    HostResult* __restrict__ res_host,
                                                                    v2d* __restrict__ force2, bool reverse)
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
            const unsigned count = (full_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
            const unsigned t0 = blockIdx.x + (reverse ? count - 1 : 0) * gridDim.x;
            const size_t base = (size_t)t0 * TILE + threadIdx.x;
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
        {
            write_result(res, sc, N, nparts, sequence);
            publish_to_host(res_host, sc, N, nparts, sequence);
        }
    }
    __syncthreads();
    MapScalars m;
    m.Dqx = s_m[0]; m.Dqy = s_m[1]; m.Fx = s_m[2]; m.Fy = s_m[3]; m.Fz = s_m[4];
    m.photon = s_mi[0];
    m.nL = s_mi[1];
    force_map_body<BLOCK, UNROLL, NT, true>(m, in.charge, in.pos2, N, prm.g, L_typeid, force2, c_first, reverse);
}

constexpr int kSmallSystemLdsCharges = 4096; // charges the single-block kernel keeps in LDS (32 KB); beyond: re-read

// ---- small systems: ONE block, ONE launch ----------------------------------------------------------------------
// The reference's production system is N = 501 (examples/init-0.gsd, 500 SLURM replicas of it).  At that size two
// launches are pure latency (~4 us each); a single block that reduces, finalises and maps in one go halves it.  Used
// below kSmallSystemMaxN particles, where one CU's bandwidth is not yet the limit.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void cavity_small_system_kernel(AosInput in, unsigned N, double Lx, double Ly, double Lz,
                                                                    DeviceParams prm, int L_typeid, uint64_t sequence,
                                                                    cavmd_result* __restrict__ res,
                                                                    HostResult* __restrict__ res_host,
                                                                    v2d* __restrict__ force2)
{
    __shared__ double s_m[5];
    __shared__ int s_mi[2];
    __shared__ double s_c[kSmallSystemLdsCharges]; // the charges, for the force phase (no second trip to global memory)
    const PhotonRow guess = photon_row(in, (size_t)(N - 1));
    Accum acc;
    constexpr int BATCH = 4; // particles in flight per lane
    for (unsigned base = 0; base < N; base += BATCH * BLOCK)
    {
        AosInput::Raw r[BATCH];
#pragma unroll
        for (int j = 0; j < BATCH; ++j)
        {
            const unsigned i = base + j * BLOCK + threadIdx.x;
            r[j] = in.load(i < N ? i : N - 1); // clamped, the duplicate is masked out below
        }
#pragma unroll
        for (int j = 0; j < BATCH; ++j)
        {
            const unsigned i = base + j * BLOCK + threadIdx.x;
            const double rx = AosInput::x(r[j]) + (double)r[j].ix * Lx;
            const double ry = AosInput::y(r[j]) + (double)r[j].iy * Ly;
            const double rz = AosInput::z(r[j]) + (double)r[j].iz * Lz;
            if (i < N)
            {
                acc.add(i, rx, ry, rz, r[j].c, AosInput::tag(r[j]), L_typeid);
                if (i < (unsigned)kSmallSystemLdsCharges)
                    s_c[i] = r[j].c;
            }
        }
    }
    acc = block_reduce<BLOCK>(acc);
    const Scalars sc = scalars_from_total<AosInput>(acc, guess, in, N, Lx, Ly, Lz, prm, true);
    if (threadIdx.x == 0)
    {
        s_m[0] = sc.Dq[0]; s_m[1] = sc.Dq[1]; s_m[2] = sc.f[0]; s_m[3] = sc.f[1]; s_m[4] = sc.f[2];
        s_mi[0] = sc.photon;
        s_mi[1] = sc.nL;
    }
    __syncthreads();
    const double Dqx = s_m[0], Dqy = s_m[1], Fx = s_m[2], Fy = s_m[3], Fz = s_m[4];
    const int photon = s_mi[0], nL = s_mi[1];
    const double ng = -prm.g;
    const unsigned nchunks = 2 * N;
    const bool odd = threadIdx.x & 1;
    const v2d zero = {0.0, 0.0};
    for (unsigned k = threadIdx.x; k < nchunks; k += BLOCK)
    {
        const unsigned p = k >> 1;
        v2d v = zero;
        if (photon >= 0)
        {
            const double c = p < (unsigned)kSmallSystemLdsCharges ? s_c[p] : in.charge[p];
            const double sgc = ng * c; // ((-g) * charge) * Dq, src/CavityForceCompute.cc:194
            v = (v2d) {sgc * Dqx, sgc * Dqy};
            const bool typed_L = (nL > 1) && (__double2loint(in.pos2[2 * p + 1].y) == L_typeid);
            v = (odd || typed_L) ? zero : v;
            if ((int)p == photon)
                v = odd ? (v2d) {Fz, 0.0} : (v2d) {Fx, Fy};
        }
        force2[k] = v;
    }
    // The result goes to the host AFTER the force stores have been issued: its system-scope release (~0.6 us) then overlaps
    // their drain instead of standing in front of them.
    if (threadIdx.x == 0)
    {
        write_result(res, sc, N, 1u, sequence);
        publish_to_host(res_host, sc, N, 1u, sequence);
    }
}

// ---- force map for the snapshot layout (strided (N,3) force + optional potential energy) ------------------------
// One particle per lane; UNROLL charges are loaded before the stores.  When the force rows are packed (stride 24) a
// wave's 64 rows are one contiguous 1.5 KB run, so the three 8-byte stores per lane still fill whole lines.
template <int BLOCK, int UNROLL>
__device__ __forceinline__ void force_map_strided_body(const MapScalars m, const StridedInput& in, unsigned N, double g,
                                                       int L_typeid, char* __restrict__ force, size_t force_stride,
                                                       char* __restrict__ pe, size_t pe_stride)
{
    const double ng = -g;
    constexpr size_t TILE = (size_t)BLOCK * UNROLL;
    const size_t tiles = ((size_t)N + TILE - 1) / TILE;
    for (size_t t = blockIdx.x; t < tiles; t += gridDim.x)
    {
        const size_t base = t * TILE + threadIdx.x;
        double c[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
        {
            const size_t i = base + (size_t)u * BLOCK;
            c[u] = (m.photon >= 0 && i < N) ? *reinterpret_cast<const double*>(in.chg + i * in.chg_stride) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
        {
            const size_t i = base + (size_t)u * BLOCK;
            if (i >= N)
                continue;
            double fx = 0.0, fy = 0.0, fz = 0.0;
            if (m.photon >= 0)
            {
                const double s = ng * c[u]; // ((-g) * charge) * Dq, src/CavityForceCompute.cc:194
                fx = s * m.Dqx;
                fy = s * m.Dqy;
                if (m.nL > 1 && *reinterpret_cast<const int*>(in.tid + i * in.tid_stride) == L_typeid)
                {
                    fx = 0.0;
                    fy = 0.0;
                }
                if ((int)i == m.photon)
                {
                    fx = m.Fx;
                    fy = m.Fy;
                    fz = m.Fz;
                }
            }
            double* f = reinterpret_cast<double*>(force + i * force_stride);
            __builtin_nontemporal_store(fx, f + 0);
            __builtin_nontemporal_store(fy, f + 1);
            __builtin_nontemporal_store(fz, f + 2);
            if (pe)
                __builtin_nontemporal_store(0.0, reinterpret_cast<double*>(pe + i * pe_stride));
        }
    }
}

// three-launch variant (scalars from the result block)
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void force_map_strided_kernel(StridedInput in, unsigned N, double g, int L_typeid,
                                                                  const cavmd_result* __restrict__ res,
                                                                  char* __restrict__ force, size_t force_stride,
                                                                  char* __restrict__ pe, size_t pe_stride)
{
    MapScalars m;
    m.Dqx = res->Dq[0]; m.Dqy = res->Dq[1];
    m.Fx = res->photon_force[0]; m.Fy = res->photon_force[1]; m.Fz = res->photon_force[2];
    m.photon = res->photon_idx;
    m.nL = res->n_photon_typed;
    force_map_strided_body<BLOCK, 4>(m, in, N, g, L_typeid, force, force_stride, pe, pe_stride);
}

// two-launch variant: the same prologue as force_map_aos_fused_kernel
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void force_map_strided_fused_kernel(StridedInput in, unsigned N, unsigned nparts,
                                                                        double Lx, double Ly, double Lz, DeviceParams prm,
                                                                        int L_typeid, Partials part, uint64_t sequence,
                                                                        cavmd_result* __restrict__ res,
                                                                        HostResult* __restrict__ res_host,
                                                                        char* __restrict__ force, size_t force_stride,
                                                                        char* __restrict__ pe, size_t pe_stride)
{
    __shared__ double s_m[5];
    __shared__ int s_mi[2];
    const Scalars sc = reduce_partials_and_finalize<StridedInput, BLOCK>(in, N, nparts, Lx, Ly, Lz, prm, part,
                                                                         blockIdx.x == 0);
    if (threadIdx.x == 0)
    {
        s_m[0] = sc.Dq[0]; s_m[1] = sc.Dq[1]; s_m[2] = sc.f[0]; s_m[3] = sc.f[1]; s_m[4] = sc.f[2];
        s_mi[0] = sc.photon;
        s_mi[1] = sc.nL;
        if (blockIdx.x == 0)
        {
            write_result(res, sc, N, nparts, sequence);
            publish_to_host(res_host, sc, N, nparts, sequence);
        }
    }
    __syncthreads();
    MapScalars m;
    m.Dqx = s_m[0]; m.Dqy = s_m[1]; m.Fx = s_m[2]; m.Fy = s_m[3]; m.Fz = s_m[4];
    m.photon = s_mi[0];
    m.nL = s_mi[1];
    force_map_strided_body<BLOCK, 4>(m, in, N, prm.g, L_typeid, force, force_stride, pe, pe_stride);
}

} // namespace cavmd
