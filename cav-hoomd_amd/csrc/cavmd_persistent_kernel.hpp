// cavmd_persistent_kernel.hpp -- the whole cavity-force evaluation in ONE launch (1024 < N <~ 2.4e6 by default).
//
//   phase 1   every block streams its tiles (pos 32 + charge 8 + image 12 B per particle) exactly as
//             dipole_partials_kernel does -- same tile assignment, same double-double accumulation, same block tree,
//             hence the same per-block partial, bit for bit -- and parks the charges it read in LDS.
//   hand-off  a two-level all-reduce across the (<= 256) workgroups, walking the same fixed tree as the two-launch
//             path's fold (reduce_partials_and_finalize): thread 0 of every block publishes the block's partial as a
//             record of 8-byte {tag, 32-bit value} granules; the first block of every group of 16 consecutive blocks
//             gathers its group's records, folds them (row_fold16) and publishes the group total; every block then
//             gathers the <= 16 group totals and folds them.  Every block obtains the same bits; no float atomics.
//             The last block publishes cavmd_result.
//   phase 2   every block writes the forces of its own tiles from the charges in LDS (dense 16-byte chunks): up to ~5e6
//             particles the charge array is not read a second time (84 instead of 92 bytes per particle cross the memory
//             bus; beyond that the tiles that do not fit in LDS are re-read) and the second launch with its ramp, drain and
//             re-fold prologue disappears.
//
// Inter-workgroup protocol (cdna_hip_programming.md, Guideline 16, form R2 "the data is the flag"): each granule is ONE
// naturally aligned 8-byte relaxed agent-scope atomic (global_store/load_dwordx2 sc1), carries its own tag and is
// validated individually by the reader, so no release/acquire ordering between granules, no separate flag, no fence and
// no dependence on dispatch order, timing or XCD placement are needed -- only that all blocks of the grid are resident
// together (grid <= CUs, one block per CU fits by construction).  Why two levels: a flat all-gather (every block reads
// every record) makes up to 1024 waves spin on the same 320 lines of the memory side -- measured 4.6 us per hand-off at
// N = 1e6 whichever way the waiting was organised (profiles/r02/microbench_persistent_v1_flat_*.txt); here a waiting block
// polls 20 lines, from ONE wave.  The tag is the epoch word of the workspace, read from DEVICE memory at kernel start and
// advanced by the publishing block once it holds the total (every block has published by then, so every block has read it): a
// captured launch replays correctly, nothing needs zeroing between launches.  Grids of at most 16 blocks skip the first
// level (every block gathers the block records itself).
//
// Starvation: when other grids hold CUs, part of this grid cannot start while the rest waits for it.  Every spin is
// bounded; a block that gives up poisons its tiles with NaN, is counted and LEAVES, which lets the missing blocks start; they
// give up in turn (the count rides along with their polls), and the last one, which knows from the count that every record
// is in place, folds them in the same order and writes every tile itself -- late, but the same bits (bail path below).
// An evaluation whose blocks were not ALL counted on the bail path (some had completed before the first give-up) leaves
// epoch[2] (the poison word) set to its launch nonce: every block of every LATER launch reads it with the tag at kernel
// entry and fails at once and as a whole (NaN forces, kSyncFailed), without touching the records or the give-up count of
// the failed launch -- so launches already queued behind a failure, or replays of a captured graph, end
// loudly whatever the slabs and counters were left like; only the host clears the word (cavmd_capi.hip, sync_state_dirty).
#pragma once

#include "cavmd_force_kernels.hpp"

#pragma clang fp contract(off)

// Diagnostic hook (csrc/microbench.hip): per-block time stamps at numbered points of the kernel; nothing in the product.
#ifndef CAVMD_PSTAMP
#define CAVMD_PSTAMP(k)
#endif

namespace cavmd
{

constexpr int kGranulesPerRecord = 2 * kNumPartDoubles + kNumPartInts; // 9 doubles as 18 halves + lmin + lcnt
constexpr unsigned kSpinLimit = 1000000; // poll rounds of a bounded wait: ~0.2-0.4 s; a healthy wait is microseconds
constexpr unsigned kRepairRounds = 4096; // the last block of a starved evaluation finds every record in place: no real wait

constexpr int kGroup = 16;           // blocks per first-level group = lanes of a DPP row
constexpr unsigned kMaxPersistGrid = kGroup * kGroup;
#ifndef CAVMD_GROUP_COPIES // (micro-benchmark sweeps only)
#define CAVMD_GROUP_COPIES 8
#endif
#ifndef CAVMD_POLL_SLEEP
#define CAVMD_POLL_SLEEP 1
#endif
constexpr unsigned kGroupCopies = CAVMD_GROUP_COPIES; // the group totals are published in 8 copies; block b polls copy b % 8, so a line of
                                     // group records has 32 pollers instead of 256 (a one-to-255 broadcast through one
                                     // line costs ~1 us more than through lines with 32 pollers: scripts/dev/pingpong.hip)

struct SyncState
{
    unsigned long long* granules; // kMaxPersistGrid block records, then kGroupCopies x kGroup group records; a record is
                                  // kGranulesPerRecord granules {tag << 32 | value} = 160 contiguous bytes
    unsigned* epoch;              // epoch[0]: tag of the next evaluation (never 0); epoch[1]: blocks of the running evaluation
                                  // that gave up waiting (0 outside a starved evaluation); epoch[2]: poison -- the nonce of
                                  // the launch some block of which gave up, from its first give-up until its last block has
                                  // put the state back in order (or for good, until the host wipes it, if nobody could)
    unsigned spin_limit;          // poll rounds of a bounded wait (kSpinLimit; tests shorten it)
    int late_block;               // fault injection (FAULT instantiation only): this block starts late_ticks of the 100 MHz
    unsigned late_ticks;          // wall clock late, as if its CU had been held by another grid; -1: none
    int silent_block;             // fault injection (FAULT instantiation only): this block never publishes its record, as if
                                  // it were not resident -> the evaluation cannot be completed; -1: none
};
// HostResult::sync_error
constexpr unsigned kSyncFailed = 1u;   // some block gave up and nobody could complete the evaluation: its forces hold NaN
constexpr unsigned kSyncRepaired = 2u; // every block gave up; the last one completed the whole evaluation alone: results valid

__device__ __forceinline__ unsigned long long granule_load(const unsigned long long* g)
{
    return __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// lane 0's Accum -> the 20 words of a record in LDS (word 2 i = low half, 2 i + 1 = high half of double i)
// (written as 32-bit words, the type they are read back as: no aliasing between double and unsigned accesses to the same LDS)
__device__ __forceinline__ void record_to_lds(unsigned* s_rec, const Accum& a)
{
    const double d[kNumPartDoubles] = {a.hx, a.lx, a.hy, a.ly, a.hz, a.lz, a.sx, a.sy, a.sz};
#pragma unroll
    for (int i = 0; i < kNumPartDoubles; ++i)
    {
        s_rec[2 * i] = (unsigned)__double2loint(d[i]);
        s_rec[2 * i + 1] = (unsigned)__double2hiint(d[i]);
    }
    s_rec[2 * kNumPartDoubles] = (unsigned)a.lmin;
    s_rec[2 * kNumPartDoubles + 1] = (unsigned)a.lcnt;
}

__device__ __forceinline__ void record_from_lds(const unsigned* w, Accum& o)
{
    o.hx = __hiloint2double((int)w[1], (int)w[0]);   o.lx = __hiloint2double((int)w[3], (int)w[2]);
    o.hy = __hiloint2double((int)w[5], (int)w[4]);   o.ly = __hiloint2double((int)w[7], (int)w[6]);
    o.hz = __hiloint2double((int)w[9], (int)w[8]);   o.lz = __hiloint2double((int)w[11], (int)w[10]);
    o.sx = __hiloint2double((int)w[13], (int)w[12]); o.sy = __hiloint2double((int)w[15], (int)w[14]);
    o.sz = __hiloint2double((int)w[17], (int)w[16]);
    o.lmin = (int)w[18];
    o.lcnt = (int)w[19];
}

// Wave 0 (all 64 lanes active): the Accum held by lane 0 -> record `rec` of `copies` consecutive copies of a slab of
// `slab_records` records.  Lane 0 spreads the 20 words over the lanes through `s_pub` (>= 20 words of LDS that only this
// wave touches; a readfirstlane + select chain for the same job compiled to ~100 instructions, 0.3 us); one wave
// instruction then stores three copies (60 lanes, 160 contiguous bytes per copy).  Twenty separate 8-byte write-through
// stores from a single lane would leave the CU one after the other -- the guide's Pitfall 7 in miniature.
__device__ __forceinline__ void publish_record(unsigned long long* slab, unsigned slab_records, unsigned copies, unsigned rec,
                                               unsigned tag, const Accum& a, unsigned* s_pub)
{
    const unsigned lane = threadIdx.x;
    if (lane == 0)
        record_to_lds(s_pub, a);
    // same wave: LDS operations complete in order, no barrier needed
    const unsigned field = lane % kGranulesPerRecord, copy0 = lane / kGranulesPerRecord; // copy0 = 0, 1, 2 (3: idle lanes)
    const unsigned long long granule = ((unsigned long long)tag << 32) | s_pub[field];
    for (unsigned c = copy0; c < copies && copy0 < 3; c += 3)
        __hip_atomic_store(slab + ((size_t)c * slab_records + rec) * kGranulesPerRecord + field, granule, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
}

// Wave 0 (all 64 lanes active): wait until records first .. first + count - 1 (count <= 16, contiguous in the slab) carry
// `tag` in every granule, then hand record first + l to lane l (identity Accum for l >= count).  A round is 5 coalesced
// wave loads (count * 160 bytes <= 20 lines); every granule is validated by its own tag; the words reach their lanes
// through a 1280-byte LDS transpose that only this wave touches.  own_slot >= 0: record first + own_slot is this block's
// own -- it is taken from `own` (lane 0's registers) instead of from memory, where the block's store may not have landed.
// Gives up after `max_rounds` rounds (1 = a single look; kSpinLimit = the bounded wait).  Returns whether the records
// were complete (wave-uniform).
__device__ __forceinline__ bool gather_records(const unsigned long long* slab, unsigned first, unsigned count, unsigned tag,
                                               unsigned* s_words, Accum& o, unsigned max_rounds, int own_slot,
                                               const Accum& own, const unsigned* giveups = nullptr)
{
    constexpr int ROUNDS = (kGroup * kGranulesPerRecord + kWave - 1) / kWave; // 5
    const unsigned lane = threadIdx.x;
    const unsigned total = count * kGranulesPerRecord;
    const unsigned long long* g = slab + (size_t)first * kGranulesPerRecord;
    unsigned long long x[ROUNDS];
    for (unsigned spins = 1;; ++spins)
    {
        // all five loads in flight together (clamped index instead of a branch per load: hipcc otherwise waits for each
        // load before it issues the next, five round trips per round); with them, where asked for, the count of blocks that
        // have given up on this evaluation: once one has, nobody may complete on its own any more (see the bail path)
        const unsigned gone = giveups ? __hip_atomic_load(giveups, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
#pragma unroll
        for (int j = 0; j < ROUNDS; ++j)
        {
            const unsigned i = j * kWave + lane;
            x[j] = granule_load(g + (i < total ? i : total - 1));
        }
        bool ok = true;
#pragma unroll
        for (int j = 0; j < ROUNDS; ++j)
        {
            const unsigned i = j * kWave + lane;
            const bool mine = own_slot >= 0 && (i < total ? i : total - 1) / kGranulesPerRecord == (unsigned)own_slot;
            ok = ok && (mine || (unsigned)(x[j] >> 32) == tag);
        }
        if (gone != 0)
            return false;
        if (__all(ok))
            break;
        if (spins >= max_rounds)
            return false;
        if (CAVMD_POLL_SLEEP > 0)
            __builtin_amdgcn_s_sleep(CAVMD_POLL_SLEEP);
    }
#pragma unroll
    for (int j = 0; j < ROUNDS; ++j)
        if ((unsigned)(j * kWave) + lane < total)
            s_words[j * kWave + lane] = (unsigned)x[j];
    // same wave: LDS operations complete in order, no barrier needed
    if (own_slot >= 0 && lane == 0)
        record_to_lds(s_words + own_slot * kGranulesPerRecord, own);
    if (lane < count)
        record_from_lds(s_words + lane * kGranulesPerRecord, o);
    return true;
}

// The particles of one block: `nfull` full tiles at first, first + step, ... and one ragged tile of `tail_count` particles
// at `tail_base`.  Two partitions of the N particles over the G blocks:
//   strided   tile t belongs to block t % G (exactly dipole_partials_kernel's assignment: the single-launch kernel then
//             produces the two-launch path's partials bit for bit -- what the hand-off's parity test leans on);
//   balanced  block b takes the contiguous 64-particle units [b U / G, (b + 1) U / G), U = ceil(N / 64): every block gets
//             the same number of particles to within 64, instead of differing by a whole 512-particle tile (one tile is
//             13 % of a block's share at N = 1e6).
struct BlockRange
{
    size_t first, step, tail_base;
    unsigned nfull, tail_count;
};
//   hybrid    (partition 2; MEASURED AND REJECTED in round 3, micro-benchmark only) the full rounds strided -- every block the
//             same floor(full_tiles / G) tiles -- and what is left (fewer than G tiles' worth of particles) cut into G
//             contiguous shares of whole 64-particle units, one ragged tile per block: equal work to within 64 particles AND
//             the strided sweep for all but the last round.  20.5 vs 19.8 us at N = 1e6: the slowest block is not the one
//             with a tile more (profiles/r03/microbench_persistent_hybrid_partition_*.txt).
template <unsigned TILE>
__device__ __forceinline__ BlockRange block_range(unsigned N, unsigned G, unsigned b, int balanced)
{
    BlockRange r;
    if (balanced == 2)
    {
        const unsigned rounds = (N / TILE) / G;
        const size_t done = (size_t)rounds * G * TILE;
        const unsigned long long U = ((unsigned long long)(N - done) + kWave - 1) / kWave;
        const size_t s = done + (size_t)(U * b / G) * kWave;
        size_t e = done + (size_t)(U * (b + 1) / G) * kWave;
        e = e < N ? e : N;
        r.first = (size_t)b * TILE;
        r.step = (size_t)G * TILE;
        r.nfull = rounds;
        r.tail_base = s;
        r.tail_count = (unsigned)(e > s ? e - s : 0);
    }
    else if (balanced)
    {
        const unsigned long long U = ((unsigned long long)N + kWave - 1) / kWave;
        const size_t s = (size_t)(U * b / G) * kWave;
        size_t e = (size_t)(U * (b + 1) / G) * kWave;
        e = e < N ? e : N;
        const size_t n = e > s ? e - s : 0;
        r.first = s;
        r.step = TILE;
        r.nfull = (unsigned)(n / TILE);
        r.tail_base = s + (size_t)r.nfull * TILE;
        r.tail_count = (unsigned)(n - (size_t)r.nfull * TILE);
    }
    else
    {
        const unsigned full_tiles = N / TILE;
        r.first = (size_t)b * TILE;
        r.step = (size_t)G * TILE;
        r.nfull = full_tiles > b ? (full_tiles - b + G - 1) / G : 0;
        r.tail_base = (size_t)full_tiles * TILE;
        r.tail_count = (b == full_tiles % G) ? (unsigned)(N - r.tail_base) : 0;
    }
    return r;
}

extern __shared__ __attribute__((aligned(16))) double s_dyn_charge[];

// EARLYZ (MEASURED AND REJECTED in round 3; instantiated by the micro-benchmark only, the library uses EARLYZ = 0): half of
// every force entry -- the odd 16-byte chunk (F_z, w) = (0, 0) of every particle that is not the photon -- does not depend on
// the total.  With EARLYZ waves 1..3 of a block write those chunks of the block's own tiles WHILE wave 0 runs the hand-off,
// i.e. while the memory system would otherwise sit idle (~3-4 us per evaluation at N = 1e6), and phase 2 writes the even
// chunks only (plus the photon's odd chunk, over the zero).  Result: 26.5 instead of 19.8 us at N = 1e6 (34.4 with
// non-temporal early stores), slower at every size from 1e5 to 2e6 -- both passes store 16 bytes out of every 32, and the
// memory side pays per line touched (profiles/r03/microbench_persistent_early_zero_chunks_*.txt).  Kept so that the table can
// be reproduced (`CAVMD_TRY_EARLYZ=1 ./microbench_persistent`).
template <int BLOCK, int UNROLL, int NT_STORE, bool FAULT = false, int EARLYZ = 0>
__global__ __launch_bounds__(BLOCK) void cavity_persistent_kernel(AosInputT<2> in, unsigned N, double Lx, double Ly, double Lz,
                                                                  DeviceParams prm, int L_typeid, SyncState st,
                                                                  uint64_t sequence, cavmd_result* __restrict__ res,
                                                                  HostResult* __restrict__ res_host,
                                                                  v2d* __restrict__ force2, unsigned lds_slots, int balanced)
{
    constexpr unsigned TILE = BLOCK * UNROLL; // particles per tile; the same tile is 2 * TILE force chunks
    constexpr int MU = 2 * UNROLL;            // 16-byte chunk stores per thread and tile
    __shared__ double s_m[5];
    __shared__ int s_mi[3];
    double* s_charge = s_dyn_charge;
    const unsigned G = gridDim.x, b = blockIdx.x, tid = threadIdx.x;

    CAVMD_PSTAMP(0);
    if constexpr (FAULT) // tests only: this block starts late, as if its CU had been held by another grid
    {
        if ((int)b == st.late_block)
        {
            const unsigned long long t0 = wall_clock64();
            while (wall_clock64() - t0 < (unsigned long long)st.late_ticks)
                __builtin_amdgcn_s_sleep(64);
        }
    }
    // this evaluation's tag, and the speculative photon row (the driver appends the photon last)
    const unsigned tag = __hip_atomic_load(st.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // The poison word holds the NONCE of the launch whose blocks raised it.  A nonce of another launch = an earlier evaluation
    // that nobody could complete: this launch fails as a whole (see the end of the kernel).  This launch's own nonce = blocks of
    // this very grid have given up while this one could not start: the bail path below, not a failure yet.
    // The nonce: address of the launch's AQL dispatch packet (grid-uniform, an SGPR pair; consecutive dispatches of a queue --
    // replays of a captured graph included -- occupy different slots of its ring: 128 or 256 bytes apart on gfx950 / ROCm 7.2,
    // scripts/dev/dispatch_id_probe.hip) mixed with the host's sequence number; never 0.
    const unsigned nonce = ((unsigned)((uintptr_t)__builtin_amdgcn_dispatch_ptr() >> 6) * 2654435761u
                            ^ (unsigned)sequence * 40503u) | 1u;
    const unsigned poison = __hip_atomic_load(st.epoch + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool poisoned = poison != 0u && poison != nonce;
    const PhotonRow guess = photon_row(in, (size_t)(N - 1));

    // ---- phase 1: partial dipole of this block's tiles; the charges of its first lds_slots tiles are parked in LDS ---------
    Accum acc;
    const BlockRange rg = block_range<TILE>(N, G, b, balanced);
    for (unsigned slot = 0; slot < rg.nfull; ++slot)
    {
        const size_t base = rg.first + (size_t)slot * rg.step + tid;
        TileRegs<AosInputT<2>, UNROLL> A;
        tile_load<AosInputT<2>, BLOCK, UNROLL>(in, base, A);
        __builtin_amdgcn_sched_barrier(0);
        if (slot < lds_slots)
        {
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
                s_charge[slot * TILE + u * BLOCK + tid] = A.raw[u].c;
        }
        tile_accumulate<AosInputT<2>, BLOCK, UNROLL>(A, base, Lx, Ly, Lz, L_typeid, acc);
    }
    if (rg.tail_count)
    {
        const size_t base = rg.tail_base + tid;
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
        {
            const unsigned o = u * BLOCK + tid;
            double c = 0.0;
            if (o < rg.tail_count)
            {
                const size_t i = base + (size_t)u * BLOCK;
                const typename AosInputT<2>::Raw r = in.load(i);
                const double rx = AosInputT<2>::x(r) + (double)r.ix * Lx;
                const double ry = AosInputT<2>::y(r) + (double)r.iy * Ly;
                const double rz = AosInputT<2>::z(r) + (double)r.iz * Lz;
                acc.add((unsigned)i, rx, ry, rz, r.c, AosInputT<2>::tag(r), L_typeid);
                c = r.c;
            }
            if (rg.nfull < lds_slots)
                s_charge[rg.nfull * TILE + o] = c;
        }
    }
    CAVMD_PSTAMP(1);
    acc = block_reduce<BLOCK>(acc);
    CAVMD_PSTAMP(2);

    // ---- hand-off: two-level all-reduce across the workgroups (wave 0; the other waves wait at the barrier below) ---------
    __shared__ __attribute__((aligned(8))) unsigned s_words[kGroup * kGranulesPerRecord];
    unsigned long long* const block_slab = st.granules;
    unsigned long long* const group_slab = st.granules + (size_t)kMaxPersistGrid * kGranulesPerRecord;
    if constexpr (EARLYZ != 0)
    {
        // waves 1 .. BLOCK/64 - 1: the (F_z, w) chunks of this block's tiles, tile by tile round-robin over those waves; a wave
        // instruction stores 64 chunks 32 bytes apart (the odd halves of 2 KiB of force entries)
        constexpr unsigned ZW = BLOCK / kWave - 1;
        if (tid >= kWave)
        {
            const unsigned w = tid / kWave - 1, lane = tid % kWave;
            const v2d z = {0.0, 0.0};
            for (unsigned slot = w; slot < rg.nfull; slot += ZW)
            {
                v2d* const f = force2 + 2 * (rg.first + (size_t)slot * rg.step) + 1;
#pragma unroll
                for (unsigned j = 0; j < TILE / kWave; ++j)
                    store_chunk<(EARLYZ == 2 ? 1 : 0)>(f + 2 * (size_t)(j * kWave + lane), z);
            }
            if (rg.tail_count && w == rg.nfull % ZW)
            {
                v2d* const f = force2 + 2 * rg.tail_base + 1;
                for (unsigned o = lane; o < rg.tail_count; o += kWave)
                    store_chunk<(EARLYZ == 2 ? 1 : 0)>(f + 2 * (size_t)o, z);
            }
        }
    }
    if (tid < kWave)
    {
        bool ok = !poisoned;
        bool silent = false;
#ifdef CAVMD_FAULT_SILENT_BLOCK // microbench only: this block never publishes, as if it were not resident
        silent = (b == CAVMD_FAULT_SILENT_BLOCK);
#endif
        if constexpr (FAULT)
            silent = silent || ((int)b == st.silent_block);
        if (!silent && !poisoned)
            publish_record(block_slab, kMaxPersistGrid, 1, b, tag, acc, s_words);
        Accum o, t;
        if (poisoned)
        {
            // nothing is published, gathered or counted
        }
        else if (G <= (unsigned)kGroup)
        {
            // A grid of at most 16 blocks (N up to ~4000) is ONE group: every block gathers the block records itself, its own
            // from registers -- one hop instead of two, and the same fold (the second level would only add zeros).
            ok = gather_records(block_slab, 0, G, tag, s_words, o, st.spin_limit, (int)b, acc, st.epoch + 1);
            CAVMD_PSTAMP(7);
        }
        else
        {
            if ((b & (kGroup - 1)) == 0)
            {
                // Level 1: the first block of group b / 16 gathers the group's records (its own from registers), folds them
                // and publishes the group total.  (Letting EVERY block take one look at its group on arrival, so that the
                // last arriver folds without waiting to be seen, measured slower: 9.30 vs 9.15 us at N = 1e5, 19.9 vs 19.75 at
                // 1e6 -- 240 extra gathers in flight when the last records land.)
                const unsigned count = min(G - b, (unsigned)kGroup);
                Accum o1, t1;
                ok = gather_records(block_slab, b, count, tag, s_words, o1, st.spin_limit, 0, acc);
                // (a group whose gather timed out publishes nothing: every block then times out on the group totals and
                // the whole evaluation fails loudly, instead of a wrong total spreading with a valid tag)
                if (ok)
                {
                    t1.merge(o1);
                    t1 = row_fold16(t1);
                    publish_record(group_slab, kGroup, kGroupCopies, b / kGroup, tag, t1, s_words);
                }
            }
            CAVMD_PSTAMP(7);
            ok = gather_records(group_slab + (size_t)(b % kGroupCopies) * kGroup * kGranulesPerRecord, 0,
                                (G + kGroup - 1) / kGroup, tag, s_words, o, st.spin_limit, -1, acc, st.epoch + 1)
                 && ok;
        }
        t.merge(o);
        const Accum tot = row_fold16(t);
        CAVMD_PSTAMP(3);
        // The block that publishes cavmd_result (a system-scope release: ~0.6 us before its force stores can start) is the
        // LAST block: it holds the fewest tiles (the ragged tail, or one tile less than the first blocks), so the detour is
        // taken from its slack instead of from the kernel's critical path.
        const bool publisher = (b == G - 1);
        const Scalars sc = scalars_from_total<AosInputT<2>>(tot, guess, in, N, Lx, Ly, Lz, prm, publisher);
        if (tid == 0)
        {
            s_m[0] = sc.Dq[0]; s_m[1] = sc.Dq[1]; s_m[2] = sc.f[0]; s_m[3] = sc.f[1]; s_m[4] = sc.f[2];
            s_mi[0] = sc.photon;
            s_mi[1] = sc.nL;
            s_mi[2] = !ok;
            if (publisher && ok)
            {
                // every block has published, hence read the epoch: advance it for the next launch (0 is never a tag)
                __hip_atomic_store(st.epoch, tag + 1u ? tag + 1u : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                write_result(res, sc, N, G, sequence);
                publish_to_host(res_host, sc, N, G, sequence);
            }
        }
    }
    CAVMD_PSTAMP(4);
    __syncthreads();
    CAVMD_PSTAMP(5);
    MapScalars m;
    m.Dqx = s_m[0]; m.Dqy = s_m[1]; m.Fx = s_m[2]; m.Fy = s_m[3]; m.Fz = s_m[4];
    m.photon = s_mi[0];
    m.nL = s_mi[1];
    const bool bad = s_mi[2];

    // ---- phase 2: forces of this block's own tiles, charges from LDS ------------------------------------------------------
    const size_t nchunks = 2 * (size_t)N;
    const v2d zero = {0.0, 0.0};
    // The slow map, all from global memory: tiles t0, t0 + step, ...; NaN everywhere (poison), or the general rule -- no photon:
    // zeros (src/CavityForceCompute.cc:145-156); several L-typed particles: the type tag has to be read, only non-L particles
    // get a molecular force (:190-191).
    auto slow_map = [&](unsigned t0, unsigned step, bool poison, const MapScalars& ms) {
        const double nan = __builtin_nan("");
        const double ng = -prm.g;
        const bool odd = tid & 1;
        for (unsigned t = t0; (size_t)t * TILE < N; t += step)
        {
#pragma unroll
            for (int u = 0; u < MU; ++u)
            {
                const size_t k = (size_t)t * 2 * TILE + (size_t)u * BLOCK + tid;
                if (k >= nchunks)
                    continue;
                v2d v = zero;
                if (poison)
                    v = (v2d) {nan, nan};
                else if (ms.photon >= 0)
                {
                    const size_t p = k >> 1;
                    const int ptag = __double2loint(in.pos2[2 * p + 1].y);
                    const double s = ng * in.charge[p];
                    v = (v2d) {s * ms.Dqx, s * ms.Dqy};
                    v = (odd || ptag == L_typeid) ? zero : v;
                    if (p == (size_t)ms.photon)
                        v = odd ? (v2d) {ms.Fz, 0.0} : (v2d) {ms.Fx, ms.Fy};
                }
                store_chunk<NT_STORE>(force2 + k, v);
            }
        }
    };
    if (bad)
    {
        // ---- this block gave up waiting: some block of the grid was not resident (other grids hold the CUs) -----------------
        // Loud by default: NaN over the block's own tiles and the failure flag.  Then the block LEAVES, which frees its CU for
        // the blocks that have not started.  Those find no group totals (the group leaders have left), give up in turn -- and
        // the last one to do so, which knows from the count that every block has published its record, completes the whole
        // evaluation alone: same records, same fold, same bits; a millisecond instead of microseconds, once, after which the
        // host keeps this workspace on two launches.  If some blocks did get the total while others gave up (a race of
        // microseconds after a wait of a third of a second), the count never completes and the evaluation stays failed.
        __shared__ unsigned s_last;
        slow_map(b, G, true, m);
        __threadfence();
        __syncthreads();
        if (tid == 0)
        {
            __hip_atomic_store(&res_host->sync_error, kSyncFailed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (poisoned)
                s_last = 0u; // a launch behind an evaluation nobody could complete: fail as a whole, touch nothing
            else
            {
                // poison first (the counting atomic below orders it): it stays unless ALL blocks end up counted here, i.e.
                // unless the last of them can put records, count and epoch back in order
                __hip_atomic_store(st.epoch + 2, nonce, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_last = __hip_atomic_fetch_add(st.epoch + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == G - 1;
            }
        }
        __syncthreads();
        if (!s_last)
            return;
        __shared__ __attribute__((aligned(8))) unsigned s_groups[kGroup * kGranulesPerRecord];
        if (tid < kWave)
        {
            bool ok = true;
            Accum o, t;
            if (G <= (unsigned)kGroup)
                ok = gather_records(block_slab, 0, G, tag, s_words, o, kRepairRounds, -1, acc);
            else
            {
                const unsigned ngroups = (G + kGroup - 1) / kGroup;
                for (unsigned g = 0; g < ngroups; ++g)
                {
                    Accum o1, t1;
                    ok = gather_records(block_slab, g * kGroup, min(G - g * kGroup, (unsigned)kGroup), tag, s_words, o1,
                                        kRepairRounds, -1, acc)
                         && ok;
                    t1.merge(o1);
                    t1 = row_fold16(t1);
                    if (tid == 0)
                        record_to_lds(s_groups + g * kGranulesPerRecord, t1);
                }
                // same wave: LDS operations complete in order, no barrier needed
                if (tid < ngroups)
                    record_from_lds(s_groups + tid * kGranulesPerRecord, o);
            }
            t.merge(o);
            const Accum tot = row_fold16(t);
            const Scalars sc = scalars_from_total<AosInputT<2>>(tot, guess, in, N, Lx, Ly, Lz, prm, true);
            if (tid == 0)
            {
                s_m[0] = sc.Dq[0]; s_m[1] = sc.Dq[1]; s_m[2] = sc.f[0]; s_m[3] = sc.f[1]; s_m[4] = sc.f[2];
                s_mi[0] = sc.photon;
                s_mi[1] = sc.nL;
                s_mi[2] = !ok;
                if (ok)
                {
                    // the verdict first, the result's ready flag (a release) after it: a host that sees the result sees "repaired"
                    __hip_atomic_store(&res_host->sync_error, kSyncRepaired, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    write_result(res, sc, N, G, sequence);
                    publish_to_host(res_host, sc, N, G, sequence);
                }
            }
        }
        __syncthreads();
        MapScalars mr;
        mr.Dqx = s_m[0]; mr.Dqy = s_m[1]; mr.Fx = s_m[2]; mr.Fy = s_m[3]; mr.Fz = s_m[4];
        mr.photon = s_mi[0];
        mr.nL = s_mi[1];
        if (!s_mi[2])
            slow_map(0, 1, false, mr); // over the other blocks' NaN: their stores were fenced before they were counted
        if (tid == 0)
        {
            // every block has read the epoch and been counted, none is left in the hand-off: count, epoch and poison are put
            // back in order for the next evaluation (whether or not THIS one could be completed: that is sync_error's to say)
            __hip_atomic_store(st.epoch + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(st.epoch, tag + 1u ? tag + 1u : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(st.epoch + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    if (m.photon < 0 || m.nL > 1)
    {
        slow_map(b, G, false, m); // rare, from global memory
        return;
    }

    const double ng = -prm.g;
    const size_t pchunk = 2 * (size_t)m.photon; // photon's first chunk
    if constexpr (EARLYZ != 0)
    {
        // the even chunks only: lane = particle, charges straight from LDS (or re-read beyond the LDS budget); the photon's
        // lane also writes its odd chunk (F_z, 0) over the zero of the early pass (same workgroup, after the barrier above)
        auto put = [&](size_t p, double c) {
            const double s = ng * c;
            v2d v = {s * m.Dqx, s * m.Dqy};
            if (p == (size_t)m.photon)
            {
                v = (v2d) {m.Fx, m.Fy};
                store_chunk<NT_STORE>(force2 + 2 * p + 1, (v2d) {m.Fz, 0.0});
            }
            store_chunk<NT_STORE>(force2 + 2 * p, v);
        };
        for (unsigned slot = 0; slot < rg.nfull; ++slot)
        {
            const size_t p0 = rg.first + (size_t)slot * rg.step + tid;
            double c[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
                c[u] = slot < lds_slots ? s_charge[slot * TILE + u * BLOCK + tid]
                                        : __builtin_nontemporal_load(in.charge + p0 + (size_t)u * BLOCK);
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
                put(p0 + (size_t)u * BLOCK, c[u]);
        }
        if (rg.tail_count)
        {
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
            {
                const unsigned o = u * BLOCK + tid;
                if (o < rg.tail_count)
                    put(rg.tail_base + o, rg.nfull < lds_slots ? s_charge[rg.nfull * TILE + o] : in.charge[rg.tail_base + o]);
            }
        }
        CAVMD_PSTAMP(6);
        return;
    }
    const bool odd = tid & 1;                   // BLOCK is even, so the half is fixed per thread
    auto store_tile = [&](unsigned slot, const double (&c)[MU]) {
        const size_t base = 2 * (rg.first + (size_t)slot * rg.step) + tid;
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
    };
    const unsigned resident = rg.nfull < lds_slots ? rg.nfull : lds_slots;
    for (unsigned slot = 0; slot < resident; ++slot)
    {
        double c[MU];
#pragma unroll
        for (int u = 0; u < MU; ++u)
            c[u] = s_charge[slot * TILE + ((u * BLOCK + tid) >> 1)];
        store_tile(slot, c);
    }
    if (resident < rg.nfull)
    {
        // Tiles beyond the LDS budget: their charges are read a second time (as the two-launch path reads all of them), D
        // tiles per batch and two batches in flight (ping-pong register sets): the loads of the next batch are issued before
        // the stores of the current one, so the wave only ever waits for loads that have had a batch of stores to land.
        constexpr int D = 8;
        auto load_batch = [&](unsigned s0, double (&c)[D][MU]) {
#pragma unroll
            for (int d = 0; d < D; ++d)
            {
                const unsigned slot = s0 + d < rg.nfull ? s0 + d : rg.nfull - 1; // clamped: no branch around the loads
                const size_t p0 = rg.first + (size_t)slot * rg.step;
#pragma unroll
                for (int u = 0; u < MU; ++u)
                    c[d][u] = __builtin_nontemporal_load(in.charge + p0 + ((u * BLOCK + tid) >> 1));
            }
        };
        auto store_batch = [&](unsigned s0, const double (&c)[D][MU]) {
#pragma unroll
            for (int d = 0; d < D; ++d)
                if (s0 + d < rg.nfull)
                    store_tile(s0 + d, c[d]);
        };
        double A[D][MU], B[D][MU];
        load_batch(resident, A);
        for (unsigned s0 = resident; s0 < rg.nfull; s0 += 2 * D)
        {
            if (s0 + D < rg.nfull)
                load_batch(s0 + D, B);
            store_batch(s0, A);
            if (s0 + 2 * D < rg.nfull)
                load_batch(s0 + 2 * D, A);
            if (s0 + D < rg.nfull)
                store_batch(s0 + D, B);
        }
    }
    if (rg.tail_count)
    {
        const size_t base = 2 * rg.tail_base + tid;
#pragma unroll
        for (int u = 0; u < MU; ++u)
        {
            const unsigned o = (u * BLOCK + tid) >> 1; // particle within the ragged tile
            if (o < rg.tail_count)
            {
                const size_t k = base + (size_t)u * BLOCK;
                const double cc = rg.nfull < lds_slots ? s_charge[rg.nfull * TILE + o] : in.charge[rg.tail_base + o];
                const double s = ng * cc;
                v2d v = {s * m.Dqx, s * m.Dqy};
                v = odd ? zero : v;
                if ((k | 1) == (pchunk | 1))
                    v = odd ? (v2d) {m.Fz, 0.0} : (v2d) {m.Fx, m.Fy};
                store_chunk<NT_STORE>(force2 + k, v);
            }
        }
    }
    CAVMD_PSTAMP(6);
}

} // namespace cavmd
