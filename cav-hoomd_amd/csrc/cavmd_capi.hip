// cavmd_capi.hip -- implementation of include/cavmd.h on top of the kernels in cavmd_kernels.hpp.
//
// Host side of the replaced reference code: CavityForceComputeGPU::computeForces
// (src/CavityForceComputeGPU.cc:102-253) and kernel::gpu_compute_cavity_force
// (src/CavityForceComputeGPU.cu:507-617).  Where the reference does 4 memsets, 1 H2D and 2 blocking
// D2H copies, a device synchronise and a host scan of the position array per step, this enqueues
// one kernel (two above ~2.4e6 particles) on the caller's stream and returns; the energies reach the host through a
// block of mapped pinned memory that cavmd_energies polls (no copy, no stream synchronisation).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <vector>

#include "cavmd.h"
#include "cavmd_kernels.hpp"

using namespace cavmd;

namespace
{
constexpr int kReduceBlock = 256;
constexpr int kReduceUnroll = 2; // particles per lane and tile; 2 beats 4 by 6 % at 1e6 and 11 % at 3e5, ties at 1e7
constexpr int kFinalizeBlock = 256;
constexpr int kMapBlock = 256;
constexpr int kMapUnroll = 4; // 1024 chunks = 512 particles per tile: the SAME particle range as a reduction tile, so with
                              // grids that are multiples of 8 tile t is reduced and mapped on the same XCD (t mod 8)
constexpr int kMaxBlocksPerCU = 16;
constexpr int kEventsPerSlot = 6;
constexpr int kProfileSlots = 512; // evaluations buffered between profile reads
constexpr size_t kMaxSamples = 4096;
constexpr int kSmallBlock = 256;
constexpr int kSmallSystemMaxN = 1024;         // single-block path (one batch of 4 x 256 particles) wins up to ~1000 particles against the
                                               // single-launch kernel: 5.0 vs 6.0 us at N = 501, 6.0 vs 6.0 at 1001, 7.0 vs 6.1 at 1101,
                                               // 7.9 vs 6.2 at 1401 (profiles/r02/ab_small_system.txt)
constexpr size_t kNtStoreMinN = 200000;       // force stores: neutral at 1e5, -3.6 % at 3e5, -4.3 % at 1e6, -5.8 % at 1e7
constexpr size_t kChargeTemporalMaxN = 25000000; // charges stay temporal while the 8 N bytes fit in the 256 MiB Infinity Cache
                                                 // next to the streams: re-measured in round 2 (the round-1 crossover at
                                                 // 5e6 dated from before the scratch-traffic fix): -8 % per evaluation at
                                                 // 6e6 and 1e7, -5 % at 2e7, tie at 5e7 (profiles/r02/ab_two_launch_knobs.txt)
constexpr int kPersistBlock = 256;
constexpr size_t kPersistMaxLds = 156 * 1024; // dynamic LDS of the single-launch kernel (charges of a block's tiles); 160 KiB per CU
constexpr uint64_t kSuspendFirst = 1ull << 16, kSuspendMax = 1ull << 31, kSuspendForever = ~0ull;
constexpr size_t kPersistSharedLds = 76 * 1024; // default ceiling: two such blocks (+ 1.7 KiB static each) fit on one CU, so two concurrent grids stay resident

static_assert(sizeof(cavmd_double4) == 32, "Scalar4 layout");
static_assert(sizeof(cavmd_int3) == 12, "int3 layout");
static_assert(sizeof(cavmd_params) == 32, "params layout");
static_assert(sizeof(cavmd_result) == 192, "result layout");
} // namespace

struct cavmd_workspace
{
    int device = -1;
    int num_cu = 0;
    char arch[64] = {0};
    size_t max_N = 0;
    unsigned max_parts = 0;
    double* d_part = nullptr;
    int* d_ipart = nullptr;
    cavmd_result* d_result = nullptr;
    HostResult* h_result = nullptr;     // pinned + mapped: the publishing block writes the result + ready flag here
    HostResult* h_result_dev = nullptr; // device-side address of h_result
    hipStream_t last_stream = nullptr;
    bool computed = false;
    uint64_t sequence = 0;
    // tunables
    // Defaults from interleaved A/B runs on MI355X (csrc/microbench.hip; profiles/r01/microbench_*.txt):
    int reduce_blocks_per_cu = 1; // <= 256 partials: the fused force map folds them with one load per thread
    int map_blocks_per_cu = 2;    // every fused block re-folds the partials, so few, long-lived blocks
    int map_nt_store = -1;        // -1 auto (non-temporal from kNtStoreMinN particles up), 0 plain, 1 non-temporal, 2 write-through
    int reduce_nt_load = -1;      // -1 auto, 0 plain, 1 pos+image non-temporal, 2 all non-temporal
    int fused_finalize = 1;       // 1: two launches (finalize folded into the force map), 0: three launches
    int map_reverse = -1;         // -1 auto, 1: the force map walks its tiles last-to-first, 0: first-to-last
    int small_system_max_n = kSmallSystemMaxN; // at or below this N: one single-block launch does everything; 0 disables
    int reduce_unroll = -1;       // particles per lane and tile of the reduction: -1 auto, 1 or 2
    int persistent = -1;          // -1 auto, 0 never, 1 whenever the grid is <= 256 blocks: ONE launch per evaluation
    int rho_lane_particle = -1;   // density field mapping: 0 lane = wavevector, 1 / 2 / 3 lane = particle with 25 / 10 / 5
                                  // wavevectors per chunk, -1 auto (lane = particle with 5 where n_k fills the 64-lane
                                  // chunks of the first mapping to less than 3/4)
    int persistent_lds_kb = 0;    // LDS budget per block of the single-launch kernel in KiB (0 = default: all of a CU's usable LDS
                                  // when forced on, half of it when chosen automatically); tiles beyond it are read twice
    int persistent_balanced = -1; // partition of the particles over the blocks of the single-launch kernel: -1 auto, 0 tiles
                                  // dealt round-robin (the two-launch path's partition), 1 contiguous, equal shares
    // single-launch evaluation: granule slab + epoch word (device), see cavmd_persistent_kernel.hpp
    unsigned long long* d_granules = nullptr;
    unsigned* d_epoch = nullptr;
    int debug_spin_limit = 0;     // tests: poll rounds of the single-launch kernel's bounded waits (0 = kSpinLimit)
    int debug_late_block = -1;    // tests: this block of the single-launch grid starts debug_late_ticks late (-1 = none)
    int debug_late_ticks = 0;     //        (100 MHz wall clock)
    int debug_silent_block = -1;  // tests: this block never publishes its record: the evaluation cannot be completed (-1 = none)
    int debug_skip_publish = 0;   // tests: the kernels publish into a scratch block instead of the one the host reads -- what a
                                  //        launch that failed on the device looks like from the host
    HostResult* h_scratch = nullptr; // (hooks build only) that scratch block, pinned + mapped
    HostResult* h_scratch_dev = nullptr;
    // after a starved evaluation the single launch is suspended: until sequence reaches suspend_until, then one probe; every
    // further starvation multiplies the pause by 8 (2^16 evaluations at first, 2^31 at most); a FAILED one suspends for good
    uint64_t suspend_until = 0;
    uint64_t suspend_backoff = kSuspendFirst;
    bool sync_state_dirty = false;  // a starved evaluation may have left records or counts behind: wipe before the next single launch
    bool sync_timeout_seen = false; // an inter-workgroup wait of the single-launch kernel gave up once: two launches from then on
    bool captured = false; // some evaluation was enqueued into a stream capture: the host-side flag protocol is off
    // profiling
    bool profiling = false;
    std::vector<hipEvent_t> events; // kEventsPerSlot per slot: start/stop of each of the three kernels
    int pending = 0;
    std::vector<unsigned> slot_mask; // which of the three kernels a slot's evaluation launched
    double acc_ms[3] = {0, 0, 0};
    uint64_t acc_launches = 0;
    std::vector<float> samples; // 3 per evaluation, capped at kMaxSamples evaluations
    // observables (rows f2 / f3)
    size_t n_k = 0;
    unsigned n_chunks = 0;
    unsigned rho_blocks = 0;
    double* d_kvec = nullptr;
    double* d_rho_part = nullptr;
    double* d_rho = nullptr;
    double* h_rho = nullptr; // pinned
    hipStream_t rho_stream = nullptr;
    bool rho_computed = false;
    double* d_mode = nullptr;
    HostMode* h_mode = nullptr;     // pinned, mapped, coherent: cavity_mode_kernel publishes here
    HostMode* h_mode_dev = nullptr;
    uint64_t mode_sequence = 0;
    double* d_fm_part = nullptr; // [2][max_parts] + 1 result
    HostScalar* h_fm = nullptr;     // pinned, mapped, coherent: the scalar reductions publish here, the host spins on `ready`
    HostScalar* h_fm_dev = nullptr; // device-side address of h_fm
    uint64_t fm_sequence = 0;
    unsigned* d_fm_ticket = nullptr; // ticket counter of the one-launch scalar reductions (reset by the folding block)
    // on-device Bussi thermostat (cavmd_bussi_step_device)
    BussiDevice* d_bussi = nullptr;
    HostBussi* h_bussi = nullptr;     // pinned, mapped, coherent
    HostBussi* h_bussi_dev = nullptr;
    uint64_t bussi_sequence = 0;
    uint64_t bussi_refused_seen = 0;  // refusals already reported to the caller
    hipStream_t bussi_stream = nullptr; // stream of the last enqueued step: the one whose idleness ends a wait for its flag
};

namespace
{
struct DeviceGuard
{
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev)
        {
            switched = (hipSetDevice(dev) == hipSuccess);
        }
    }
    ~DeviceGuard()
    {
        if (switched)
            (void)hipSetDevice(prev);
    }
};

inline int hip_status(hipError_t e)
{
    return e == hipSuccess ? CAVMD_OK : (int)e;
}

#define CAVMD_HIP_TRY(expr)              \
    do                                   \
    {                                    \
        hipError_t _e = (expr);          \
        if (_e != hipSuccess)            \
            return (int)_e;              \
    } while (0)

DeviceParams derive(const cavmd_params* p)
{
    DeviceParams d;
    d.g = p->couplstr;
    d.K = p->K;
    d.gK = p->couplstr / p->K;                   // as `m_params.couplstr / m_params.K`, src/CavityForceCompute.cc:183
    d.g2K = p->couplstr * p->couplstr / p->K;    // as `couplstr * couplstr / K`, :176 (host code is built -ffp-contract=off)
    return d;
}

bool params_ok(const cavmd_params* p)
{
    return p && isfinite(p->omegac) && isfinite(p->couplstr) && isfinite(p->K) && isfinite(p->phmass) && p->K != 0.0;
}

int drain_profile(cavmd_workspace* ws)
{
    for (int s = 0; s < ws->pending; ++s)
    {
        hipEvent_t* ev = &ws->events[kEventsPerSlot * s];
        const unsigned used = ws->slot_mask[s];
        float sample[3] = {0.f, 0.f, 0.f};
        for (int k = 0; k < 3; ++k)
        {
            if (!(used & (1u << k)))
                continue;
            CAVMD_HIP_TRY(hipEventSynchronize(ev[2 * k + 1]));
            float ms = 0.f;
            CAVMD_HIP_TRY(hipEventElapsedTime(&ms, ev[2 * k], ev[2 * k + 1]));
            ws->acc_ms[k] += (double)ms;
            sample[k] = ms;
        }
        ws->acc_launches += 1;
        if (ws->samples.size() >= 3 * kMaxSamples)
            ws->samples.erase(ws->samples.begin(), ws->samples.begin() + 3);
        ws->samples.insert(ws->samples.end(), sample, sample + 3);
    }
    ws->pending = 0;
    return CAVMD_OK;
}

// A captured evaluation carries a frozen `sequence` argument: from the second replay on the host-visible ready flag
// already holds that value, so the flag protocol of cavmd_result_read cannot tell a finished replay from a running one.
// Once a workspace has been captured its results are read behind a device synchronisation instead (include/cavmd.h).
void note_capture(cavmd_workspace* ws, hipStream_t stream)
{
    // (the null stream cannot be captured: HOOMD-blue's and torch's default path pays nothing for the query)
    if (ws->captured || stream == nullptr)
        return;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
        ws->captured = true;
}

// Where the single-launch evaluation is the default (measured on MI355X, profiles/r02/microbench_persistent_*.txt).
bool persistent_auto(size_t N)
{
    (void)N;
    return true;
}
// Contiguous equal shares measured SLOWER than tiles dealt round-robin wherever the streaming matters (21.5 vs 20.3 us at
// N = 1e6, 63.8 vs 57.8 at 4e6: 256 sequential streams a fixed distance apart load the HBM channels less evenly than one
// 4 MB window that all blocks sweep together); only 3e5 gained (12.0 vs 12.2).  Kept as a tunable, off.
bool persistent_balanced_auto(size_t N)
{
    (void)N;
    return false;
}

unsigned grid_for(size_t work_items, unsigned tile, int num_cu, int blocks_per_cu)
{
    size_t tiles = (work_items + tile - 1) / tile;
    size_t cap = (size_t)num_cu * (size_t)blocks_per_cu;
    size_t g = tiles < cap ? tiles : cap;
    return (unsigned)(g ? g : 1);
}

constexpr int kScaleBlocksPerCu = 4; // velocity rescale: 256-thread blocks per CU (4 particles per lane and tile)

// Scratch of the scalar reductions (sum |F| / m, kinetic energy): partials + the host-visible scalar.
int ensure_scalar_scratch(cavmd_workspace* ws)
{
    if (ws->d_fm_part)
        return CAVMD_OK;
    CAVMD_HIP_TRY(hipMalloc((void**)&ws->d_fm_part, sizeof(double) * (2 * (size_t)ws->max_parts + 1)));
    CAVMD_HIP_TRY(hipHostMalloc((void**)&ws->h_fm, sizeof(HostScalar), hipHostMallocMapped | hipHostMallocCoherent));
    memset(ws->h_fm, 0, sizeof(HostScalar));
    CAVMD_HIP_TRY(hipHostGetDevicePointer((void**)&ws->h_fm_dev, ws->h_fm, 0));
    CAVMD_HIP_TRY(hipMalloc((void**)&ws->d_fm_ticket, 128));
    CAVMD_HIP_TRY(hipMemset(ws->d_fm_ticket, 0, 128));
    return CAVMD_OK;
}

// Wait for the scalar the fold kernel publishes (flag in mapped host memory; the stream going idle ends the wait too, e.g.
// after a failed launch) -- about a PCIe write after the kernel has it, instead of a copy plus a stream synchronisation.
int wait_scalar(cavmd_workspace* ws, hipStream_t stream, double* out)
{
    const uint64_t want = ws->fm_sequence;
    for (;;)
    {
        if (__atomic_load_n(&ws->h_fm->ready, __ATOMIC_ACQUIRE) == want)
            break;
        const hipError_t q = hipStreamQuery(stream);
        if (q == hipSuccess)
        {
            if (__atomic_load_n(&ws->h_fm->ready, __ATOMIC_ACQUIRE) != want)
            {
                // the kernel never published (failed or aborted launch): its blocks may have left the ticket counter
                // part-way, after which no block would ever be "last" again -- put it back before reporting
                (void)hipMemsetAsync(ws->d_fm_ticket, 0, 128, stream);
                return (int)hipErrorLaunchFailure;
            }
            break;
        }
        if (q != hipErrorNotReady)
            return (int)q;
    }
    *out = ws->h_fm->value;
    return CAVMD_OK;
}

// The single-launch kernel keeps the charges of a block's tiles in dynamic LDS (up to kPersistMaxLds); HIP wants the
// ceiling raised per kernel before a launch may ask for more than 64 KiB.
hipError_t allow_large_lds()
{
    // (per device: called from cavmd_create under its device guard)
    hipError_t once = [] {
        hipError_t e = hipSuccess;
#define CAVMD_ALLOW(UNR, NTS)                                                                                    \
    if (e == hipSuccess)                                                                                        \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&cavity_persistent_kernel<kPersistBlock, UNR, NTS>), \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPersistMaxLds);
        CAVMD_ALLOW(1, 0)
        CAVMD_ALLOW(1, 1)
        CAVMD_ALLOW(1, 2)
        CAVMD_ALLOW(2, 0)
        CAVMD_ALLOW(2, 1)
        CAVMD_ALLOW(2, 2)
#undef CAVMD_ALLOW
#ifdef CAVMD_TEST_HOOKS
        // the fault-injection instantiations (tests: "debug_late_block", "debug_silent_block"; libcavmd_hooks.so only)
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&cavity_persistent_kernel<kPersistBlock, 1, 0, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPersistMaxLds);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&cavity_persistent_kernel<kPersistBlock, 2, 0, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPersistMaxLds);
#endif
        return e;
    }();
    return once;
}

// Per-kernel timing for bench.py's roofline leg.  When profiling is on, every kernel is launched through
// hipExtLaunchKernelGGL with its own start/stop events: those take the dispatch packet's begin/end timestamps (what
// rocprofv3 --kernel-trace reports), unlike hipEventRecord markers between kernels, which add ~2.5 us each.
struct LaunchScope
{
    cavmd_workspace* ws;
    hipStream_t stream;
    hipEvent_t* ev = nullptr;
    unsigned used = 0;
    int status = CAVMD_OK;
    LaunchScope(cavmd_workspace* w, hipStream_t s) : ws(w), stream(s)
    {
        if (!ws->profiling)
            return;
        if (ws->pending == kProfileSlots)
            status = drain_profile(ws);
        if (status == CAVMD_OK)
            ev = &ws->events[kEventsPerSlot * ws->pending];
    }
    template <class K, class... Args>
    int launch_lds(int slot, size_t lds_bytes, K kernel, unsigned grid, unsigned block, Args... args)
    {
        if (ev)
        {
            hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(block), lds_bytes, stream, ev[2 * slot], ev[2 * slot + 1], 0,
                                  args...);
            used |= 1u << slot;
        }
        else
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), lds_bytes, stream, args...);
        return hip_status(hipGetLastError());
    }
    template <class K, class... Args>
    int launch(int slot, K kernel, unsigned grid, unsigned block, Args... args)
    {
        return launch_lds(slot, 0, kernel, grid, block, args...);
    }
    void commit()
    {
        if (ev)
        {
            ws->slot_mask[ws->pending] = used;
            ws->pending += 1;
        }
    }
};
} // namespace

extern "C"
{

cavmd_params cavmd_make_params(double omegac, double couplstr, double phmass)
{
    cavmd_params p;
    p.omegac = omegac;
    p.couplstr = couplstr;
    p.phmass = phmass;
    p.K = phmass * omegac * omegac; // src/CavityForceCompute.h:41, evaluated left to right
    return p;
}

int cavmd_create(int device, size_t max_N, cavmd_workspace** out_ws)
{
    if (!out_ws)
        return CAVMD_ERR_INVALID_VALUE;
    *out_ws = nullptr;
    if (max_N > (size_t)INT_MAX)
        return CAVMD_ERR_CAPACITY; // indices travel as int32 (photon_idx), as in the reference
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return CAVMD_ERR_NO_DEVICE;
    if (device < 0)
    {
        if (hipGetDevice(&device) != hipSuccess)
            return CAVMD_ERR_NO_DEVICE;
    }
    if (device >= count)
        return CAVMD_ERR_INVALID_VALUE;

    cavmd_workspace* ws = new (std::nothrow) cavmd_workspace();
    if (!ws)
        return (int)hipErrorOutOfMemory;
    ws->device = device;
    ws->max_N = max_N;
    // deployment-level switch for GPUs shared by several processes (see "persistent" in cavmd.h): CAVMD_PERSISTENT=0|1
    if (const char* env = getenv("CAVMD_PERSISTENT"))
    {
        if (!strcmp(env, "0"))
            ws->persistent = 0;
        else if (!strcmp(env, "1"))
            ws->persistent = 1;
    }
    DeviceGuard guard(device);

    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess)
    {
        delete ws;
        return (int)e;
    }
    ws->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    strncpy(ws->arch, prop.gcnArchName, sizeof(ws->arch) - 1);
    ws->max_parts = (unsigned)(ws->num_cu * kMaxBlocksPerCU);

    e = hipMalloc((void**)&ws->d_part, sizeof(double) * kNumPartDoubles * ws->max_parts);
    if (e == hipSuccess)
        e = hipMalloc((void**)&ws->d_ipart, sizeof(int) * kNumPartInts * ws->max_parts);
    if (e == hipSuccess)
        e = hipMalloc((void**)&ws->d_result, sizeof(cavmd_result));
    if (e == hipSuccess)
        e = hipMemset(ws->d_result, 0, sizeof(cavmd_result));
    if (e == hipSuccess)
        e = hipMalloc((void**)&ws->d_granules, sizeof(unsigned long long) * 2 * kGranulesPerRecord * kMaxPersistGrid);
    if (e == hipSuccess) // tag 0 = never valid
        e = hipMemset(ws->d_granules, 0, sizeof(unsigned long long) * 2 * kGranulesPerRecord * kMaxPersistGrid);
    if (e == hipSuccess)
        e = hipMalloc((void**)&ws->d_epoch, 4 * sizeof(unsigned));
    if (e == hipSuccess)
    {
        const unsigned init[4] = {1u, 0u, 0u, 0u}; // first tag; no block has given up; not poisoned
        e = hipMemcpy(ws->d_epoch, init, sizeof(init), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess)
        e = allow_large_lds();
    if (e == hipSuccess) // coherent: the polled flag must not depend on HIP_HOST_COHERENT
        e = hipHostMalloc((void**)&ws->h_result, sizeof(HostResult), hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess)
        e = hipHostGetDevicePointer((void**)&ws->h_result_dev, ws->h_result, 0);
    if (e != hipSuccess)
    {
        cavmd_destroy(ws);
        return (int)e;
    }
    memset(ws->h_result, 0, sizeof(HostResult));
    *out_ws = ws;
    return CAVMD_OK;
}

int cavmd_destroy(cavmd_workspace* ws)
{
    if (!ws)
        return CAVMD_OK;
    DeviceGuard guard(ws->device);
    for (hipEvent_t ev : ws->events)
        (void)hipEventDestroy(ev);
    if (ws->d_part)
        (void)hipFree(ws->d_part);
    if (ws->d_ipart)
        (void)hipFree(ws->d_ipart);
    if (ws->d_result)
        (void)hipFree(ws->d_result);
    if (ws->d_granules)
        (void)hipFree(ws->d_granules);
    if (ws->d_epoch)
        (void)hipFree(ws->d_epoch);
    if (ws->h_result)
        (void)hipHostFree(ws->h_result);
    if (ws->d_kvec)
        (void)hipFree(ws->d_kvec);
    if (ws->d_rho_part)
        (void)hipFree(ws->d_rho_part);
    if (ws->d_rho)
        (void)hipFree(ws->d_rho);
    if (ws->h_rho)
        (void)hipHostFree(ws->h_rho);
    if (ws->d_mode)
        (void)hipFree(ws->d_mode);
    if (ws->h_mode)
        (void)hipHostFree(ws->h_mode);
    if (ws->d_fm_part)
        (void)hipFree(ws->d_fm_part);
    if (ws->h_fm)
        (void)hipHostFree(ws->h_fm);
    if (ws->d_fm_ticket)
        (void)hipFree(ws->d_fm_ticket);
    if (ws->h_scratch)
        (void)hipHostFree(ws->h_scratch);
    if (ws->d_bussi)
        (void)hipFree(ws->d_bussi);
    if (ws->h_bussi)
        (void)hipHostFree(ws->h_bussi);
    delete ws;
    return CAVMD_OK;
}

namespace
{
// Where the kernels publish the result block for the host: the workspace's mapped block -- or, in the test-hooks build with
// "debug_skip_publish" set, a scratch block the host never looks at (what a launch that died on the device looks like).
inline HostResult* host_block(cavmd_workspace* ws)
{
#ifdef CAVMD_TEST_HOOKS
    if (ws->debug_skip_publish && ws->h_scratch_dev)
        return ws->h_scratch_dev;
#endif
    return ws->h_result_dev;
}

// A single-launch evaluation whose blocks were not resident together (other grids held the CUs) either got completed by its
// last block alone (kSyncRepaired: results valid, it just took a second) or failed (kSyncFailed: NaN forces) -- see the
// bail path of cavity_persistent_kernel.  Whoever notices first -- the next enqueue or the result read -- suspends the
// single-launch path for this workspace: what starved the grid is a property of how the GPU is shared at the moment, not of
// one step, and the two-launch path does not depend on residency.  Returns 0 (nothing happened), kSyncRepaired or kSyncFailed.
unsigned consume_sync_timeout(cavmd_workspace* ws)
{
    if (!ws->h_result || !__atomic_load_n(&ws->h_result->sync_error, __ATOMIC_ACQUIRE))
        return 0;
    // kSyncFailed is provisional while the kernel runs (the first block that gives up raises it, the last one may still
    // complete the evaluation): the verdict is the flag once the stream has drained.  A stream that is being captured cannot
    // be waited for; the provisional value then counts.
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (ws->last_stream == nullptr || hipStreamIsCapturing(ws->last_stream, &cap) != hipSuccess
        || cap == hipStreamCaptureStatusNone)
        (void)hipStreamSynchronize(ws->last_stream);
    const unsigned verdict = __atomic_load_n(&ws->h_result->sync_error, __ATOMIC_ACQUIRE);
    __atomic_store_n(&ws->h_result->sync_error, 0u, __ATOMIC_RELEASE);
    ws->sync_timeout_seen = true;
    ws->sync_state_dirty = true;
    if (verdict == kSyncRepaired)
    {
        // two launches for a while, then one probe: whoever held the CUs may have gone.  A probe that starves again costs one
        // slow (valid) evaluation and an 8 times longer pause.
        if (ws->sequence - ws->suspend_until > ws->suspend_backoff)
            ws->suspend_backoff = kSuspendFirst; // the single launch had been healthy for longer than the last pause: start over
        ws->suspend_until = ws->sequence + ws->suspend_backoff;
        ws->suspend_backoff = ws->suspend_backoff * 8 < kSuspendMax ? ws->suspend_backoff * 8 : kSuspendMax;
        return kSyncRepaired;
    }
    ws->suspend_until = kSuspendForever; // not understood: stay off until the caller switches it on again
    ws->computed = false;                // the result block still holds the evaluation BEFORE the failed one
    return kSyncFailed;
}
} // namespace

int cavmd_compute_hoomd(cavmd_workspace* ws, void* stream_, size_t N, const cavmd_double4* d_pos, const double* d_charge,
                        const cavmd_int3* d_image, double Lx, double Ly, double Lz, int L_typeid,
                        const cavmd_params* params, cavmd_double4* d_force)
{
    // argument validation first, as kernel::gpu_compute_cavity_force does (src/CavityForceComputeGPU.cu:522-532)
    if (!ws || !d_pos || !d_charge || !d_image || !d_force || !params)
        return CAVMD_ERR_INVALID_VALUE;
    if (((uintptr_t)d_pos & 15) || ((uintptr_t)d_force & 15) || ((uintptr_t)d_charge & 7) || ((uintptr_t)d_image & 3))
        return CAVMD_ERR_INVALID_VALUE;
    if (N == 0)
        return CAVMD_OK;
    if (N > ws->max_N)
        return CAVMD_ERR_CAPACITY;
    if (!params_ok(params))
        return CAVMD_ERR_BAD_PARAMS;
    // an EARLIER evaluation was starved and nobody read the result since.  Failed (its forces are NaN): report it here,
    // nothing is enqueued by this call.  Repaired: nothing to report.  Either way two launches from now on.
    if (consume_sync_timeout(ws) == kSyncFailed)
        return CAVMD_ERR_SYNC_TIMEOUT;

    hipStream_t stream = (hipStream_t)stream_;
    DeviceGuard guard(ws->device);
    note_capture(ws, stream);
    LaunchScope ls(ws, stream);
    if (ls.status != CAVMD_OK)
        return ls.status;

    AosInput in;
    in.pos2 = reinterpret_cast<const v2d*>(d_pos);
    in.charge = d_charge;
    in.image = reinterpret_cast<const int*>(d_image);
    Partials part {ws->d_part, ws->d_ipart, ws->max_parts};
    const unsigned n = (unsigned)N;
    const DeviceParams dp = derive(params);
    int st;

    // ---- small systems (the reference's own N = 501): one block reduces, finalises and maps in ONE launch
    if (ws->small_system_max_n > 0 && N <= (size_t)ws->small_system_max_n)
    {
        ws->sequence += 1;
        st = ls.launch(0, cavity_small_system_kernel<kSmallBlock>, 1u, kSmallBlock, in, n, Lx, Ly, Lz, dp, L_typeid,
                       ws->sequence, ws->d_result, host_block(ws), reinterpret_cast<v2d*>(d_force));
        if (st != CAVMD_OK)
            return st;
        ls.commit();
        ws->last_stream = stream;
        ws->computed = true;
        return CAVMD_OK;
    }

    // ---- launch 1: per-block partial sums + photon search
    // Tile depth: 2 particles per lane (512 per block; 8 loads in flight per lane); 1 while that leaves at most about one
    // tile per CU (re-measured with the single-launch kernel: 1 wins by 9 % at N = 5e4 and 4 % at 1.3e5, ties at 1e5,
    // loses by 6 % from 2e5 up: profiles/r02/ab_reduce_unroll.txt).
    int unroll = kReduceUnroll;
    while (unroll > 1 && N / ((size_t)kReduceBlock * unroll) < (size_t)ws->num_cu * 5 / 4)
        unroll >>= 1;
    if (ws->reduce_unroll == 1 || ws->reduce_unroll == 2)
        unroll = ws->reduce_unroll;
    const unsigned g1 = grid_for(N, kReduceBlock * unroll, ws->num_cu, ws->reduce_blocks_per_cu);
    v2d* force2 = reinterpret_cast<v2d*>(d_force);
    // Force stores bypass the caches for all but small N: the array is consumed much later (by the integrator, after
    // every other force of the step), and not leaving 32 N dirty bytes behind shortens this kernel's drain and spares
    // the next reduction the evictions (measured on whole evaluations, profiles/r01/microbench_*.txt).
    const int nt_store = ws->map_nt_store < 0 ? (N >= kNtStoreMinN ? 1 : 0) : ws->map_nt_store;

    // ---- ONE launch (cavmd_persistent_kernel.hpp): the same grid as launch 1 below; with the strided partition also the
    // same tiles, hence the same partials and (same fold) the same bits as two launches.  Needs the whole grid resident at
    // once: g1 <= CUs x blocks per CU by construction, <= 256 blocks (the two-level all-reduce), LDS and registers admit
    // that many blocks per CU.  The charges of a block's first lds_slots tiles stay in LDS; tiles beyond (N >~ 5e6) are
    // read a second time.
    {
        const size_t tile = (size_t)kReduceBlock * unroll;
        const bool balanced = ws->persistent_balanced < 0 ? persistent_balanced_auto(N) : (ws->persistent_balanced != 0);
        size_t slots;
        if (balanced)
        {
            const size_t units = (N + kWave - 1) / kWave;
            slots = (((units + g1 - 1) / g1) * kWave + tile - 1) / tile; // the largest share, in tiles (ragged one included)
        }
        else
            slots = ((N + tile - 1) / tile + g1 - 1) / g1;
        size_t budget = kPersistMaxLds / (size_t)ws->reduce_blocks_per_cu - 1024;
        if (ws->persistent_lds_kb > 0 && (size_t)ws->persistent_lds_kb * 1024 < budget)
            budget = (size_t)ws->persistent_lds_kb * 1024;
        const size_t cap_slots = budget / (tile * sizeof(double));
        const size_t lds_slots = slots < cap_slots ? slots : cap_slots;
        const size_t lds = lds_slots * tile * sizeof(double);
        const bool resident = g1 <= kMaxPersistGrid && ws->reduce_blocks_per_cu <= 4;
        // auto: only while every tile of a block fits in LDS.  With overflow tiles re-read in the second phase (one block per
        // CU) the kernel loses to two launches: 142 against 137 us at N = 1e7 (profiles/r02/ab_overflow.txt).
        // Nor by default beyond half a CU's LDS per block (N >~ 2.4e6): two such grids from different streams or processes
        // could then not be resident side by side, and two half-resident grids would wait for each other until their
        // bounded spins give up (a loud CAVMD_ERR_SYNC_TIMEOUT, but a failure).  persistent = 1 lifts both limits.
        if (resident && ws->sequence >= ws->suspend_until
            && (ws->persistent > 0
                || (ws->persistent < 0 && slots <= cap_slots && lds <= kPersistSharedLds && persistent_auto(N))))
        {
            if (ws->sync_state_dirty)
            {
                // re-enabled after a starved evaluation: records or give-up counts of that launch must not meet this one
                CAVMD_HIP_TRY(hipMemsetAsync(ws->d_granules, 0, sizeof(unsigned long long) * 2 * kGranulesPerRecord * kMaxPersistGrid, stream));
                CAVMD_HIP_TRY(hipMemsetAsync(ws->d_epoch + 1, 0, 2 * sizeof(unsigned), stream)); // give-up count and poison
                ws->sync_state_dirty = false;
            }
            ws->sequence += 1;
            const AosInputT<2> inx {in.pos2, in.charge, in.image};
            const SyncState sync {ws->d_granules, ws->d_epoch, ws->debug_spin_limit > 0 ? (unsigned)ws->debug_spin_limit : kSpinLimit,
                                  ws->debug_late_block, (unsigned)ws->debug_late_ticks, ws->debug_silent_block};
#define CAVMD_LAUNCH_PERSIST(UNR, NTS)                                                                               \
    st = ls.launch_lds(0, lds, cavity_persistent_kernel<kPersistBlock, UNR, NTS>, g1, kPersistBlock, inx, n, Lx, Ly, Lz, \
                       dp, L_typeid, sync, ws->sequence, ws->d_result, host_block(ws), force2, (unsigned)lds_slots, balanced);
#ifdef CAVMD_TEST_HOOKS
            if (ws->debug_late_block >= 0 || ws->debug_silent_block >= 0)
            {
                // fault injection (tests): one block starts late -> the grid starves itself and has to be repaired; or one block
                // never publishes -> the evaluation cannot be completed
                if (unroll == 2)
                    st = ls.launch_lds(0, lds, cavity_persistent_kernel<kPersistBlock, 2, 0, true>, g1, kPersistBlock, inx, n, Lx, Ly,
                                       Lz, dp, L_typeid, sync, ws->sequence, ws->d_result, host_block(ws), force2,
                                       (unsigned)lds_slots, balanced);
                else
                    st = ls.launch_lds(0, lds, cavity_persistent_kernel<kPersistBlock, 1, 0, true>, g1, kPersistBlock, inx, n, Lx, Ly,
                                       Lz, dp, L_typeid, sync, ws->sequence, ws->d_result, host_block(ws), force2,
                                       (unsigned)lds_slots, balanced);
            }
            else
#endif
            if (unroll == 2)
            {
                if (nt_store == 2)
                    CAVMD_LAUNCH_PERSIST(2, 2)
                else if (nt_store == 1)
                    CAVMD_LAUNCH_PERSIST(2, 1)
                else
                    CAVMD_LAUNCH_PERSIST(2, 0)
            }
            else
            {
                if (nt_store == 2)
                    CAVMD_LAUNCH_PERSIST(1, 2)
                else if (nt_store == 1)
                    CAVMD_LAUNCH_PERSIST(1, 1)
                else
                    CAVMD_LAUNCH_PERSIST(1, 0)
            }
#undef CAVMD_LAUNCH_PERSIST
            if (st != CAVMD_OK)
                return st;
            ls.commit();
            ws->last_stream = stream;
            ws->computed = true;
            return CAVMD_OK;
        }
    }

    // Load policy of the reduction.  pos and image are read once per evaluation: non-temporal.  charge is read again
    // by the force map: keeping it temporal lets the map find it in the Infinity Cache while 8 N bytes fit there
    // (measured per evaluation: -8 % at N = 6e6 and 1e7, -5 % at 2e7, tie at 5e7); non-temporal above.
    int nt = ws->reduce_nt_load;
    if (nt < 0)
        nt = (N <= kChargeTemporalMaxN) ? 1 : 2;
#define CAVMD_LAUNCH_REDUCE(NTMODE, UNR)                                                                              \
    {                                                                                                                 \
        AosInputT<NTMODE> inx {in.pos2, in.charge, in.image};                                                         \
        st = ls.launch(0, dipole_partials_kernel<AosInputT<NTMODE>, kReduceBlock, UNR, false>, g1, kReduceBlock, inx, \
                       n, Lx, Ly, Lz, L_typeid, part);                                                                \
    }
#define CAVMD_LAUNCH_REDUCE_NT(UNR)                                                                                   \
    {                                                                                                                 \
        if (nt == 0)                                                                                                  \
            CAVMD_LAUNCH_REDUCE(0, UNR)                                                                               \
        else if (nt == 1)                                                                                             \
            CAVMD_LAUNCH_REDUCE(1, UNR)                                                                               \
        else                                                                                                          \
            CAVMD_LAUNCH_REDUCE(2, UNR)                                                                               \
    }
    if (unroll == 2)
        CAVMD_LAUNCH_REDUCE_NT(2)
    else
        CAVMD_LAUNCH_REDUCE_NT(1)
#undef CAVMD_LAUNCH_REDUCE_NT
#undef CAVMD_LAUNCH_REDUCE
    if (st != CAVMD_OK)
        return st;

    ws->sequence += 1;
    const unsigned g2 = grid_for(2 * N, kMapBlock * kMapUnroll, ws->num_cu, ws->map_blocks_per_cu);
    // Reverse tile order in the force map: the charge lines the reduction touched last are then asked for first.  It only
    // matters where the per-XCD share of the charges (N bytes) is about the size of an XCD's 4 MiB L2: -3.3 % per
    // evaluation at N = 4e6, neutral at 3e5 / 1e6 / 2e6 / 1e7 (scripts/ab_tunable.py map_reverse 0 1 ...).
    const bool map_reverse = ws->map_reverse < 0 ? (N > 2500000 && N <= 5000000) : (ws->map_reverse != 0);
    if (ws->fused_finalize)
    {
        // ---- launch 2 of 2: every force-map block folds the partials itself, block 0 publishes the result block
        if (nt_store == 2)
            st = ls.launch(2, force_map_aos_fused_kernel<kMapBlock, kMapUnroll, 2>, g2, kMapBlock, in, n, g1, Lx, Ly,
                           Lz, dp, L_typeid, part, ws->sequence, ws->d_result, host_block(ws), force2, map_reverse);
        else if (nt_store == 1)
            st = ls.launch(2, force_map_aos_fused_kernel<kMapBlock, kMapUnroll, 1>, g2, kMapBlock, in, n, g1, Lx, Ly,
                           Lz, dp, L_typeid, part, ws->sequence, ws->d_result, host_block(ws), force2, map_reverse);
        else
            st = ls.launch(2, force_map_aos_fused_kernel<kMapBlock, kMapUnroll, 0>, g2, kMapBlock, in, n, g1, Lx, Ly,
                           Lz, dp, L_typeid, part, ws->sequence, ws->d_result, host_block(ws), force2, map_reverse);
    }
    else
    {
        // ---- three-launch variant (kept for A/B): finalize, then a force map that reads the result block
        st = ls.launch(1, finalize_kernel<AosInput, kFinalizeBlock>, 1u, kFinalizeBlock, in, n, g1, Lx, Ly, Lz, dp,
                       part, ws->sequence, ws->d_result, host_block(ws));
        if (st != CAVMD_OK)
            return st;
        const cavmd_result* res = ws->d_result;
        const double g = params->couplstr;
        if (nt_store)
            st = ls.launch(2, force_map_aos_kernel<kMapBlock, kMapUnroll, true>, g2, kMapBlock, d_charge, in.pos2, n, g,
                           L_typeid, res, force2);
        else
            st = ls.launch(2, force_map_aos_kernel<kMapBlock, kMapUnroll, false>, g2, kMapBlock, d_charge, in.pos2, n, g,
                           L_typeid, res, force2);
    }
    if (st != CAVMD_OK)
        return st;
    ls.commit();

    ws->last_stream = stream;
    ws->computed = true;
    return CAVMD_OK;
}

int cavmd_compute_soa(cavmd_workspace* ws, void* stream_, size_t N, const double* d_position, size_t position_stride,
                      const int32_t* d_typeid, size_t typeid_stride, const int32_t* d_image, size_t image_stride,
                      const double* d_charge, size_t charge_stride, double Lx, double Ly, double Lz, int L_typeid,
                      const cavmd_params* params, double* d_force, size_t force_stride, double* d_potential_energy,
                      size_t potential_energy_stride)
{
    if (!ws || !d_position || !d_typeid || !d_image || !d_charge || !d_force || !params)
        return CAVMD_ERR_INVALID_VALUE;
    if (position_stride < 24 || typeid_stride < 4 || image_stride < 12 || charge_stride < 8 || force_stride < 24
        || (d_potential_energy && potential_energy_stride < 8))
        return CAVMD_ERR_INVALID_VALUE;
    if (((uintptr_t)d_position & 7) || (position_stride & 7) || ((uintptr_t)d_charge & 7) || (charge_stride & 7)
        || ((uintptr_t)d_force & 7) || (force_stride & 7) || ((uintptr_t)d_typeid & 3) || (typeid_stride & 3)
        || ((uintptr_t)d_image & 3) || (image_stride & 3)
        || (d_potential_energy && (((uintptr_t)d_potential_energy & 7) || (potential_energy_stride & 7))))
        return CAVMD_ERR_INVALID_VALUE;
    if (N == 0)
        return CAVMD_OK;
    if (N > ws->max_N)
        return CAVMD_ERR_CAPACITY;
    if (!params_ok(params))
        return CAVMD_ERR_BAD_PARAMS;

    // HOOMD's GPU local snapshot hands out strided VIEWS of its Scalar4 buffers (position = pos[:, :3], typeid = the
    // int in pos.w, force = force4[:, :3], potential_energy = force4[:, 3]).  That is exactly the AoS layout, so the
    // force.Custom route gets the tuned two-launch path.
    {
        const char* p0 = reinterpret_cast<const char*>(d_position);
        const char* f0 = reinterpret_cast<const char*>(d_force);
        if (position_stride == 32 && typeid_stride == 32 && reinterpret_cast<const char*>(d_typeid) == p0 + 24
            && image_stride == 12 && charge_stride == 8 && force_stride == 32 && d_potential_energy
            && potential_energy_stride == 32 && reinterpret_cast<const char*>(d_potential_energy) == f0 + 24
            && !((uintptr_t)p0 & 15) && !((uintptr_t)f0 & 15))
        {
            return cavmd_compute_hoomd(ws, stream_, N, reinterpret_cast<const cavmd_double4*>(d_position), d_charge,
                                       reinterpret_cast<const cavmd_int3*>(d_image), Lx, Ly, Lz, L_typeid, params,
                                       reinterpret_cast<cavmd_double4*>(d_force));
        }
    }

    hipStream_t stream = (hipStream_t)stream_;
    DeviceGuard guard(ws->device);
    note_capture(ws, stream);
    LaunchScope ls(ws, stream);
    if (ls.status != CAVMD_OK)
        return ls.status;

    StridedInput in;
    in.pos = reinterpret_cast<const char*>(d_position);
    in.tid = reinterpret_cast<const char*>(d_typeid);
    in.img = reinterpret_cast<const char*>(d_image);
    in.chg = reinterpret_cast<const char*>(d_charge);
    in.pos_stride = position_stride;
    in.tid_stride = typeid_stride;
    in.img_stride = image_stride;
    in.chg_stride = charge_stride;
    Partials part {ws->d_part, ws->d_ipart, ws->max_parts};
    const unsigned n = (unsigned)N;
    const DeviceParams dp = derive(params);

    // same tile-depth rule as cavmd_compute_hoomd, so that both layouts share one summation tree (and give equal bits)
    int unroll = kReduceUnroll;
    while (unroll > 1 && N / ((size_t)kReduceBlock * unroll) < (size_t)ws->num_cu * 5 / 4)
        unroll >>= 1;
    if (ws->reduce_unroll == 1 || ws->reduce_unroll == 2)
        unroll = ws->reduce_unroll;
    const unsigned g1 = grid_for(N, kReduceBlock * unroll, ws->num_cu, ws->reduce_blocks_per_cu);
    int st;
    if (unroll == 2)
        st = ls.launch(0, dipole_partials_kernel<StridedInput, kReduceBlock, 2, false>, g1, kReduceBlock, in, n, Lx, Ly, Lz,
                       L_typeid, part);
    else
        st = ls.launch(0, dipole_partials_kernel<StridedInput, kReduceBlock, 1, false>, g1, kReduceBlock, in, n, Lx, Ly, Lz,
                       L_typeid, part);
    if (st != CAVMD_OK)
        return st;
    ws->sequence += 1;
    char* f_out = reinterpret_cast<char*>(d_force);
    char* pe_out = reinterpret_cast<char*>(d_potential_energy);
    const unsigned g2 = grid_for(N, kMapBlock * 4, ws->num_cu, ws->map_blocks_per_cu);
    if (ws->fused_finalize)
    {
        st = ls.launch(2, force_map_strided_fused_kernel<kMapBlock>, g2, kMapBlock, in, n, g1, Lx, Ly, Lz, dp, L_typeid, part,
                       ws->sequence, ws->d_result, host_block(ws), f_out, force_stride, pe_out, potential_energy_stride);
    }
    else
    {
        st = ls.launch(1, finalize_kernel<StridedInput, kFinalizeBlock>, 1u, kFinalizeBlock, in, n, g1, Lx, Ly, Lz, dp,
                       part, ws->sequence, ws->d_result, host_block(ws));
        if (st != CAVMD_OK)
            return st;
        const cavmd_result* res = ws->d_result;
        st = ls.launch(2, force_map_strided_kernel<kMapBlock>, g2, kMapBlock, in, n, params->couplstr, L_typeid, res,
                       f_out, force_stride, pe_out, potential_energy_stride);
    }
    if (st != CAVMD_OK)
        return st;
    ls.commit();

    ws->last_stream = stream;
    ws->computed = true;
    return CAVMD_OK;
}

int cavmd_result_read(cavmd_workspace* ws, cavmd_result* out)
{
    if (!ws || !out)
        return CAVMD_ERR_INVALID_VALUE;
    if (!ws->computed)
        return CAVMD_ERR_NOT_COMPUTED;
    DeviceGuard guard(ws->device);
    bool published = true;
    if (ws->captured)
    {
        // graph replays: the flag cannot be trusted (frozen sequence) and the replay stream is unknown -> wait for the device
        CAVMD_HIP_TRY(hipDeviceSynchronize());
    }
    else
    {
        // The publishing block stores the result block and then a sequence flag (system-scope release) into mapped pinned
        // host memory.  Spin on that flag: the energies arrive as soon as the block that computes the scalars has them,
        // about a PCIe write after, instead of a stream synchronisation (~15 us).  The stream going idle ends the wait as
        // well (a failed launch, or a timed-out single-launch kernel, never sets the flag).
        const uint64_t want = ws->sequence;
        for (;;)
        {
            if (__atomic_load_n(&ws->h_result->ready, __ATOMIC_ACQUIRE) == want)
                break;
            const hipError_t q = hipStreamQuery(ws->last_stream);
            if (q == hipSuccess)
            {
                published = __atomic_load_n(&ws->h_result->ready, __ATOMIC_ACQUIRE) == want;
                break;
            }
            if (q != hipErrorNotReady)
                return (int)q;
        }
    }
    // A starved evaluation that failed (NaN forces): the result block still holds the PREVIOUS evaluation, which is
    // invalidated so that a second read reports "nothing computed" instead of handing that out as if it were current.
    // One that its last block completed has published its result like any other.
    if (consume_sync_timeout(ws) == kSyncFailed)
        return CAVMD_ERR_SYNC_TIMEOUT;
    if (!published)
    {
        // the stream is idle and the flag does not carry this evaluation's sequence: its launch failed on the device.  The
        // block in host memory belongs to an EARLIER evaluation: not handed out, and invalidated (as wait_scalar does for the
        // scalar reductions)
        ws->computed = false;
        return (int)hipErrorLaunchFailure;
    }
    memcpy(out, &ws->h_result->result, sizeof(cavmd_result));
    return CAVMD_OK;
}

int cavmd_energies(cavmd_workspace* ws, double out[3])
{
    if (!ws || !out)
        return CAVMD_ERR_INVALID_VALUE;
    if (!ws->computed)
    {
        // the reference's getters return the 0.0 its constructor stored (src/CavityForceCompute.cc:33-36)
        out[0] = out[1] = out[2] = 0.0;
        return CAVMD_OK;
    }
    cavmd_result r;
    int st = cavmd_result_read(ws, &r);
    if (st != CAVMD_OK)
        return st;
    out[0] = r.energy[0];
    out[1] = r.energy[1];
    out[2] = r.energy[2];
    return CAVMD_OK;
}

int cavmd_result_device_ptr(cavmd_workspace* ws, const cavmd_result** out)
{
    if (!ws || !out)
        return CAVMD_ERR_INVALID_VALUE;
    *out = ws->d_result;
    return CAVMD_OK;
}

int cavmd_set_wavevectors(cavmd_workspace* ws, size_t n_k, const double* h_wavevectors)
{
    if (!ws || !h_wavevectors || n_k == 0 || n_k > (size_t)1 << 20)
        return CAVMD_ERR_INVALID_VALUE;
    DeviceGuard guard(ws->device);
    if (ws->d_kvec)
        (void)hipFree(ws->d_kvec);
    if (ws->d_rho_part)
        (void)hipFree(ws->d_rho_part);
    if (ws->d_rho)
        (void)hipFree(ws->d_rho);
    if (ws->h_rho)
        (void)hipHostFree(ws->h_rho);
    ws->d_kvec = ws->d_rho_part = ws->d_rho = ws->h_rho = nullptr;
    ws->rho_computed = false;
    ws->n_k = n_k;
    ws->n_chunks = (unsigned)((n_k + kWave - 1) / kWave);
    ws->rho_blocks = (unsigned)ws->num_cu * 4; // capacity of the partial buffer: three 256-thread blocks per CU (lane = particle
                                               // mapping) or one 1024-thread block per CU (lane = wavevector)
    CAVMD_HIP_TRY(hipMalloc((void**)&ws->d_kvec, sizeof(double) * 3 * n_k));
    CAVMD_HIP_TRY(hipMalloc((void**)&ws->d_rho_part, sizeof(double) * 2 * kWave * (size_t)ws->n_chunks * ws->rho_blocks));
    CAVMD_HIP_TRY(hipMalloc((void**)&ws->d_rho, sizeof(double) * 2 * n_k));
    CAVMD_HIP_TRY(hipHostMalloc((void**)&ws->h_rho, sizeof(double) * 2 * n_k, hipHostMallocDefault));
    CAVMD_HIP_TRY(hipMemcpy(ws->d_kvec, h_wavevectors, sizeof(double) * 3 * n_k, hipMemcpyHostToDevice));
    return CAVMD_OK;
}

int cavmd_density_field(cavmd_workspace* ws, void* stream_, size_t N, const double* d_position, size_t position_stride)
{
    if (!ws || !d_position || position_stride < 24 || (position_stride & 7) || ((uintptr_t)d_position & 7))
        return CAVMD_ERR_INVALID_VALUE;
    if (ws->n_k == 0)
        return CAVMD_ERR_NOT_COMPUTED; // no wavevectors stored yet
    if (N > (size_t)INT_MAX)
        return CAVMD_ERR_CAPACITY;
    hipStream_t stream = (hipStream_t)stream_;
    DeviceGuard guard(ws->device);
    constexpr int kBlock = 1024;
    const size_t tiles = (N + kWave - 1) / kWave;
    unsigned gb;
    // lane = wavevector costs ceil(n_k / 64) * 64 lane-slots per particle, lane = particle n_k slots that measured 1.33x
    // as expensive each (N = 1e6: n_k = 17: 53 vs 87 us, 50: 99 vs 95, 64: 118 vs 88, 100: 170 vs 165)
    int lp = ws->rho_lane_particle;
    if (lp < 0)
        lp = (ws->n_k * 4 < (size_t)ws->n_chunks * kWave * 3) ? 3 : 0;
    if (lp)
    {
        // lane = particle: 256-thread blocks, KC wavevectors per chunk (2 KC running sums per lane in registers)
        constexpr int kLpBlock = 256;
        size_t g = (tiles + (kLpBlock / kWave) - 1) / (kLpBlock / kWave);
        if (g > ws->rho_blocks)
            g = ws->rho_blocks;
        gb = (unsigned)(g ? g : 1);
#define CAVMD_LAUNCH_LP(KCV)                                                                                              \
    hipLaunchKernelGGL((density_partials_lp_kernel<kLpBlock, KCV>), dim3(gb, (unsigned)((ws->n_k + KCV - 1) / KCV)),        \
                       dim3(kLpBlock), 0, stream, reinterpret_cast<const char*>(d_position), position_stride, (unsigned)N, \
                       ws->d_kvec, (unsigned)ws->n_k, make_sincos_coef(), ws->d_rho_part);
        if (lp == 1)
            CAVMD_LAUNCH_LP(25)
        else if (lp == 2)
            CAVMD_LAUNCH_LP(10)
        else
            CAVMD_LAUNCH_LP(5)
#undef CAVMD_LAUNCH_LP
    }
    else
    {
        size_t g = (tiles + (kBlock / kWave) - 1) / (kBlock / kWave);
        if (g > (size_t)ws->num_cu)
            g = (size_t)ws->num_cu;
        gb = (unsigned)(g ? g : 1);
        hipLaunchKernelGGL((density_partials_kernel<kBlock>), dim3(gb, ws->n_chunks), dim3(kBlock), 0, stream,
                           reinterpret_cast<const char*>(d_position), position_stride, (unsigned)N, ws->d_kvec,
                           (unsigned)ws->n_k, make_sincos_coef(), ws->d_rho_part);
    }
    CAVMD_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL((density_fold_kernel<kBlock>), dim3(ws->n_chunks), dim3(kBlock), 0, stream, ws->d_rho_part,
                       gb, (unsigned)ws->n_k, ws->d_rho);
    CAVMD_HIP_TRY(hipGetLastError());
    ws->rho_stream = stream;
    ws->rho_computed = true;
    return CAVMD_OK;
}

int cavmd_density_field_read(cavmd_workspace* ws, double* h_out)
{
    if (!ws || !h_out)
        return CAVMD_ERR_INVALID_VALUE;
    if (!ws->rho_computed)
        return CAVMD_ERR_NOT_COMPUTED;
    DeviceGuard guard(ws->device);
    CAVMD_HIP_TRY(hipMemcpyAsync(ws->h_rho, ws->d_rho, sizeof(double) * 2 * ws->n_k, hipMemcpyDeviceToHost, ws->rho_stream));
    CAVMD_HIP_TRY(hipStreamSynchronize(ws->rho_stream));
    memcpy(h_out, ws->h_rho, sizeof(double) * 2 * ws->n_k);
    return CAVMD_OK;
}

int cavmd_cavity_mode(cavmd_workspace* ws, void* stream_, const cavmd_double4* d_vel, double kB, double out[4])
{
    if (!ws || !d_vel || !out || !(kB > 0.0) || ((uintptr_t)d_vel & 15))
        return CAVMD_ERR_INVALID_VALUE;
    if (!ws->computed)
        return CAVMD_ERR_NOT_COMPUTED;
    // the photon index and E_h are read from the last evaluation's device-side result: if that evaluation was starved and
    // could not be completed, the block on the device still belongs to the evaluation BEFORE it -> say so instead
    if (consume_sync_timeout(ws) == kSyncFailed)
        return CAVMD_ERR_SYNC_TIMEOUT;
    hipStream_t stream = (hipStream_t)stream_;
    DeviceGuard guard(ws->device);
    if (!ws->d_mode)
    {
        CAVMD_HIP_TRY(hipMalloc((void**)&ws->d_mode, sizeof(double) * 4));
        CAVMD_HIP_TRY(hipHostMalloc((void**)&ws->h_mode, sizeof(HostMode), hipHostMallocMapped | hipHostMallocCoherent));
        memset(ws->h_mode, 0, sizeof(HostMode));
        CAVMD_HIP_TRY(hipHostGetDevicePointer((void**)&ws->h_mode_dev, ws->h_mode, 0));
    }
    const cavmd_result* res = ws->d_result;
    ws->mode_sequence += 1;
    hipLaunchKernelGGL(cavity_mode_kernel, dim3(1), dim3(1), 0, stream, res, d_vel, kB, ws->d_mode, ws->h_mode_dev,
                       ws->mode_sequence);
    CAVMD_HIP_TRY(hipGetLastError());
    // spin on the flag (the stream going idle ends the wait too, e.g. after a failed launch)
    for (;;)
    {
        if (__atomic_load_n(&ws->h_mode->ready, __ATOMIC_ACQUIRE) == ws->mode_sequence)
            break;
        const hipError_t q = hipStreamQuery(stream);
        if (q == hipSuccess)
        {
            if (__atomic_load_n(&ws->h_mode->ready, __ATOMIC_ACQUIRE) != ws->mode_sequence)
                return (int)hipErrorLaunchFailure;
            break;
        }
        if (q != hipErrorNotReady)
            return (int)q;
    }
    for (int k = 0; k < 4; ++k)
        out[k] = ws->h_mode->v[k];
    return CAVMD_OK;
}

int cavmd_force_mass_sum(cavmd_workspace* ws, void* stream_, size_t N, const cavmd_double4* d_net_force,
                         const cavmd_double4* d_vel, double* out)
{
    if (!ws || !d_net_force || !d_vel || !out || ((uintptr_t)d_net_force & 15) || ((uintptr_t)d_vel & 15))
        return CAVMD_ERR_INVALID_VALUE;
    if (N > (size_t)INT_MAX)
        return CAVMD_ERR_CAPACITY;
    if (N == 0)
    {
        *out = 0.0;
        return CAVMD_OK;
    }
    hipStream_t stream = (hipStream_t)stream_;
    DeviceGuard guard(ws->device);
    {
        const int st0 = ensure_scalar_scratch(ws);
        if (st0 != CAVMD_OK)
            return st0;
    }
    constexpr int kBlock = 256, kUnroll = 4;
    // one block per CU: every block draws a ticket from ONE counter (~12 ns each, serialised at the memory side); with four
    // blocks per CU the 1024 tickets alone took 12 us
    const unsigned g = grid_for(N, kBlock * kUnroll, ws->num_cu, 1);
    double* d_out = ws->d_fm_part + 2 * (size_t)ws->max_parts;
    ws->fm_sequence += 1;
    hipLaunchKernelGGL((force_mass_fused_kernel<kBlock, kUnroll>), dim3(g), dim3(kBlock), 0, stream,
                       reinterpret_cast<const v2d*>(d_net_force), reinterpret_cast<const v2d*>(d_vel), (unsigned)N,
                       ws->d_fm_part, ws->d_fm_ticket, d_out, ws->h_fm_dev, ws->fm_sequence);
    CAVMD_HIP_TRY(hipGetLastError());
    return wait_scalar(ws, stream, out);
}

int cavmd_kinetic_energy(cavmd_workspace* ws, void* stream_, const cavmd_double4* d_vel, const uint32_t* d_members,
                         size_t n_members, double* out)
{
    if (!ws || !d_vel || !out || ((uintptr_t)d_vel & 15) || ((uintptr_t)d_members & 3))
        return CAVMD_ERR_INVALID_VALUE;
    if (n_members > (size_t)INT_MAX)
        return CAVMD_ERR_CAPACITY;
    if (n_members == 0)
    {
        *out = 0.0;
        return CAVMD_OK;
    }
    hipStream_t stream = (hipStream_t)stream_;
    DeviceGuard guard(ws->device);
    {
        const int st0 = ensure_scalar_scratch(ws);
        if (st0 != CAVMD_OK)
            return st0;
    }
    constexpr int kBlock = 256, kUnroll = 4;
    const unsigned g = grid_for(n_members, kBlock * kUnroll, ws->num_cu, 1); // one ticket per CU, see cavmd_force_mass_sum
    double* d_out = ws->d_fm_part + 2 * (size_t)ws->max_parts;
    ws->fm_sequence += 1;
    hipLaunchKernelGGL((kinetic_fused_kernel<kBlock, kUnroll>), dim3(g), dim3(kBlock), 0, stream,
                       reinterpret_cast<const v2d*>(d_vel), d_members, (unsigned)n_members, ws->d_fm_part, ws->d_fm_ticket,
                       d_out, ws->h_fm_dev, ws->fm_sequence);
    CAVMD_HIP_TRY(hipGetLastError());
    return wait_scalar(ws, stream, out);
}

int cavmd_scale_velocities(cavmd_workspace* ws, void* stream_, cavmd_double4* d_vel, const uint32_t* d_members,
                           size_t n_members, double alpha)
{
    if (!ws || !d_vel || ((uintptr_t)d_vel & 15) || ((uintptr_t)d_members & 3))
        return CAVMD_ERR_INVALID_VALUE;
    if (n_members > (size_t)INT_MAX)
        return CAVMD_ERR_CAPACITY;
    if (n_members == 0)
        return CAVMD_OK;
    hipStream_t stream = (hipStream_t)stream_;
    DeviceGuard guard(ws->device);
    constexpr int kBlock = 256, kUnroll = 4;
    const unsigned g = grid_for(n_members, kBlock * kUnroll, ws->num_cu, kScaleBlocksPerCu);
    hipLaunchKernelGGL((scale_velocities_kernel<kBlock, kUnroll>), dim3(g), dim3(kBlock), 0, stream, reinterpret_cast<v2d*>(d_vel),
                       d_members, (unsigned)n_members, alpha);
    return hip_status(hipGetLastError());
}

// ---- Bussi reservoir thermostat: the scalar rule (host arithmetic; the file is built with -ffp-contract=off) ---------
int cavmd_bussi_rescale_factor(double K, double degrees_of_freedom, double deltaT, double set_T, double tau,
                               double normal_variate, double gamma_variate, double* alpha)
{
    if (!alpha)
        return CAVMD_ERR_INVALID_VALUE;
    // src/BussiReservoirThermostat.h:186-190  c = exp(-dt / tau), 0 for tau == 0 (instantaneous thermalisation); the rest of
    // the rule (:183-213) is bussi_alpha_from_c, the function the on-device step runs too
    const double c = (tau != 0.0) ? exp(-deltaT / tau) : 0.0;
    *alpha = bussi_alpha_from_c(K, degrees_of_freedom, c, set_T, normal_variate, gamma_variate);
    return CAVMD_OK;
}

int cavmd_bussi_step(cavmd_bussi_reservoir* state, double K_translational, double dof_translational, double K_rotational,
                     double dof_rotational, double deltaT, double set_T, double tau, const double variates[4],
                     double factors[2])
{
    if (!state || !variates || !factors)
        return CAVMD_ERR_INVALID_VALUE;
    // src/BussiReservoirThermostat.h:45-48
    if (deltaT == 0.0)
    {
        factors[0] = factors[1] = 1.0;
        return CAVMD_OK;
    }
    // :57-61 "Bussi thermostat requires non-zero initial momenta."
    if ((dof_translational != 0 && K_translational == 0) || (dof_rotational != 0 && K_rotational == 0))
        return CAVMD_ERR_BAD_PARAMS;
    double at = 1.0, ar = 1.0;
    (void)cavmd_bussi_rescale_factor(K_translational, dof_translational, deltaT, set_T, tau, variates[0], variates[1], &at);
    (void)cavmd_bussi_rescale_factor(K_rotational, dof_rotational, deltaT, set_T, tau, variates[2], variates[3], &ar);
    // :86-95  energy handed to the reservoir = KE_old - KE_new = KE_old (1 - alpha^2)
    const double delta_t = K_translational * (1.0 - at * at);
    const double delta_r = K_rotational * (1.0 - ar * ar);
    state->reservoir_translational += delta_t;
    state->reservoir_rotational += delta_r;
    state->instantaneous_translational = delta_t;
    state->instantaneous_rotational = delta_r;
    factors[0] = at;
    factors[1] = ar;
    return CAVMD_OK;
}

namespace
{
int ensure_bussi_state(cavmd_workspace* ws)
{
    if (ws->d_bussi)
        return CAVMD_OK;
    CAVMD_HIP_TRY(hipMalloc((void**)&ws->d_bussi, sizeof(BussiDevice)));
    CAVMD_HIP_TRY(hipMemset(ws->d_bussi, 0, sizeof(BussiDevice)));
    CAVMD_HIP_TRY(hipHostMalloc((void**)&ws->h_bussi, sizeof(HostBussi), hipHostMallocMapped | hipHostMallocCoherent));
    memset(ws->h_bussi, 0, sizeof(HostBussi));
    CAVMD_HIP_TRY(hipHostGetDevicePointer((void**)&ws->h_bussi_dev, ws->h_bussi, 0));
    return CAVMD_OK;
}
} // namespace

int cavmd_bussi_step_device(cavmd_workspace* ws, void* stream_, cavmd_double4* d_vel, const uint32_t* d_members,
                            size_t n_members, double dof_translational, double deltaT, double set_T, double tau,
                            double normal_variate, double gamma_variate)
{
    if (!ws || !d_vel || ((uintptr_t)d_vel & 15) || ((uintptr_t)d_members & 3))
        return CAVMD_ERR_INVALID_VALUE;
    if (n_members > (size_t)INT_MAX)
        return CAVMD_ERR_CAPACITY;
    if (deltaT == 0.0 || n_members == 0) // src/BussiReservoirThermostat.h:45-48: factors {1, 1}, counters untouched
        return CAVMD_OK;
    hipStream_t stream = (hipStream_t)stream_;
    DeviceGuard guard(ws->device);
    {
        int st0 = ensure_scalar_scratch(ws);
        if (st0 == CAVMD_OK)
            st0 = ensure_bussi_state(ws);
        if (st0 != CAVMD_OK)
            return st0;
    }
    constexpr int kBlock = 256, kUnroll = 4;
    BussiStepArgs a;
    a.dof = dof_translational;
    a.c = (tau != 0.0) ? exp(-deltaT / tau) : 0.0; // :186-190
    a.set_T = set_T;
    a.normal_variate = normal_variate;
    a.gamma_variate = gamma_variate;
    const unsigned g = grid_for(n_members, kBlock * kUnroll, ws->num_cu, 1);
    ws->bussi_sequence += 1;
    ws->bussi_stream = stream;
    hipLaunchKernelGGL((kinetic_partials_kernel<kBlock, kUnroll>), dim3(g), dim3(kBlock), 0, stream,
                       reinterpret_cast<const v2d*>(d_vel), d_members, (unsigned)n_members, ws->d_fm_part);
    CAVMD_HIP_TRY(hipGetLastError());
    const unsigned g2 = grid_for(n_members, kBlock * kUnroll, ws->num_cu, kScaleBlocksPerCu);
    hipLaunchKernelGGL((bussi_rescale_fused_kernel<kBlock, kUnroll>), dim3(g2), dim3(kBlock), 0, stream,
                       reinterpret_cast<v2d*>(d_vel), d_members, (unsigned)n_members, ws->d_fm_part, g, a, ws->d_bussi,
                       ws->h_bussi_dev, ws->bussi_sequence);
    return hip_status(hipGetLastError());
}

int cavmd_bussi_device_read(cavmd_workspace* ws, cavmd_bussi_device_state* out)
{
    if (!ws || !out)
        return CAVMD_ERR_INVALID_VALUE;
    memset(out, 0, sizeof(*out));
    if (!ws->d_bussi || ws->bussi_sequence == 0)
        return CAVMD_OK;
    hipStream_t stream = ws->bussi_stream; // the stream the last step went to, whatever stream the caller is on now
    DeviceGuard guard(ws->device);
    const uint64_t want = ws->bussi_sequence;
    for (;;)
    {
        if (__atomic_load_n(&ws->h_bussi->ready, __ATOMIC_ACQUIRE) == want)
            break;
        const hipError_t q = hipStreamQuery(stream);
        if (q == hipSuccess)
        {
            if (__atomic_load_n(&ws->h_bussi->ready, __ATOMIC_ACQUIRE) != want)
                return (int)hipErrorLaunchFailure; // a launch that never published
            break;
        }
        if (q != hipErrorNotReady)
            return (int)q;
    }
    const BussiDevice s = ws->h_bussi->state;
    out->reservoir_translational = s.reservoir;
    out->instantaneous_translational = s.instantaneous;
    out->last_alpha = s.alpha;
    out->last_kinetic_energy = s.kinetic;
    out->steps = s.steps;
    out->refused = s.errors;
    if (s.errors != ws->bussi_refused_seen)
    {
        ws->bussi_refused_seen = s.errors;
        return CAVMD_ERR_BAD_PARAMS; // "Bussi thermostat requires non-zero initial momenta."
    }
    return CAVMD_OK;
}

int cavmd_bussi_device_reset(cavmd_workspace* ws, void* stream_)
{
    if (!ws)
        return CAVMD_ERR_INVALID_VALUE;
    if (!ws->d_bussi)
        return CAVMD_OK;
    hipStream_t stream = (hipStream_t)stream_;
    DeviceGuard guard(ws->device);
    // wait for the last step's publication first so that the host copy can be reset consistently
    cavmd_bussi_device_state unused;
    const int st = cavmd_bussi_device_read(ws, &unused);
    if (st != CAVMD_OK && st != CAVMD_ERR_BAD_PARAMS)
        return st;
    CAVMD_HIP_TRY(hipMemsetAsync(ws->d_bussi, 0, sizeof(BussiDevice), stream));
    memset(&ws->h_bussi->state, 0, sizeof(BussiDevice));
    ws->bussi_refused_seen = 0;
    return CAVMD_OK;
}

int cavmd_profile_enable(cavmd_workspace* ws, int on)
{
    if (!ws)
        return CAVMD_ERR_INVALID_VALUE;
    DeviceGuard guard(ws->device);
    if (on && ws->events.empty())
    {
        ws->events.resize(kEventsPerSlot * kProfileSlots);
        ws->slot_mask.assign(kProfileSlots, 0);
        for (size_t i = 0; i < ws->events.size(); ++i)
        {
            hipError_t e = hipEventCreate(&ws->events[i]);
            if (e != hipSuccess)
            {
                for (size_t j = 0; j < i; ++j)
                    (void)hipEventDestroy(ws->events[j]);
                ws->events.clear();
                return (int)e;
            }
        }
    }
    if (!on && ws->pending)
    {
        int st = drain_profile(ws);
        if (st != CAVMD_OK)
            return st;
    }
    ws->profiling = on != 0;
    return CAVMD_OK;
}

int cavmd_profile_read(cavmd_workspace* ws, double ms[3], uint64_t* launches)
{
    if (!ws || !ms || !launches)
        return CAVMD_ERR_INVALID_VALUE;
    DeviceGuard guard(ws->device);
    int st = drain_profile(ws);
    if (st != CAVMD_OK)
        return st;
    for (int k = 0; k < 3; ++k)
    {
        ms[k] = ws->acc_ms[k];
        ws->acc_ms[k] = 0.0;
    }
    *launches = ws->acc_launches;
    ws->acc_launches = 0;
    ws->samples.clear();
    return CAVMD_OK;
}

int cavmd_profile_samples(cavmd_workspace* ws, double* out, size_t cap, size_t* n)
{
    if (!ws || !out || !n)
        return CAVMD_ERR_INVALID_VALUE;
    DeviceGuard guard(ws->device);
    int st = drain_profile(ws);
    if (st != CAVMD_OK)
        return st;
    const size_t have = ws->samples.size() / 3;
    const size_t take = have < cap ? have : cap;
    const size_t first = have - take;
    for (size_t i = 0; i < 3 * take; ++i)
        out[i] = (double)ws->samples[3 * first + i];
    *n = take;
    return CAVMD_OK;
}

int cavmd_set_tunable(cavmd_workspace* ws, const char* name, int value)
{
    if (!ws || !name)
        return CAVMD_ERR_INVALID_VALUE;
    if (!strcmp(name, "reduce_blocks_per_cu"))
    {
        if (value < 1 || value > kMaxBlocksPerCU)
            return CAVMD_ERR_INVALID_VALUE;
        ws->reduce_blocks_per_cu = value;
        return CAVMD_OK;
    }
    if (!strcmp(name, "map_blocks_per_cu"))
    {
        if (value < 1 || value > kMaxBlocksPerCU)
            return CAVMD_ERR_INVALID_VALUE;
        ws->map_blocks_per_cu = value;
        return CAVMD_OK;
    }
    if (!strcmp(name, "map_nt_store"))
    {
        if (value < -1 || value > 2)
            return CAVMD_ERR_INVALID_VALUE;
        ws->map_nt_store = value;
        return CAVMD_OK;
    }
    if (!strcmp(name, "reduce_nt_load"))
    {
        if (value < -1 || value > 2)
            return CAVMD_ERR_INVALID_VALUE;
        ws->reduce_nt_load = value;
        return CAVMD_OK;
    }
    if (!strcmp(name, "fused_finalize"))
    {
        if (value != 0 && value != 1)
            return CAVMD_ERR_INVALID_VALUE;
        ws->fused_finalize = value;
        return CAVMD_OK;
    }
    if (!strcmp(name, "map_reverse"))
    {
        if (value < -1 || value > 1)
            return CAVMD_ERR_INVALID_VALUE;
        ws->map_reverse = value;
        return CAVMD_OK;
    }
    if (!strcmp(name, "small_system_max_n"))
    {
        if (value < 0 || value > (1 << 20))
            return CAVMD_ERR_INVALID_VALUE;
        ws->small_system_max_n = value;
        return CAVMD_OK;
    }
    if (!strcmp(name, "persistent"))
    {
        if (value < -1 || value > 1)
            return CAVMD_ERR_INVALID_VALUE;
        ws->persistent = value;
        ws->suspend_until = 0; // the caller's word ends a suspension (the hand-off slabs are wiped before the next single launch)
        ws->suspend_backoff = kSuspendFirst;
        return CAVMD_OK;
    }
#ifdef CAVMD_TEST_HOOKS
    if (!strcmp(name, "debug_suspend_first"))
    {
        if (value < 1)
            return CAVMD_ERR_INVALID_VALUE;
        ws->suspend_backoff = (uint64_t)value;
        return CAVMD_OK;
    }
    if (!strcmp(name, "debug_spin_limit"))
    {
        if (value < 0)
            return CAVMD_ERR_INVALID_VALUE;
        ws->debug_spin_limit = value;
        return CAVMD_OK;
    }
    if (!strcmp(name, "debug_late_block"))
    {
        if (value < -1 || value >= (int)kMaxPersistGrid)
            return CAVMD_ERR_INVALID_VALUE;
        ws->debug_late_block = value;
        return CAVMD_OK;
    }
    if (!strcmp(name, "debug_late_ticks"))
    {
        if (value < 0 || value > 100000000) // at most one second
            return CAVMD_ERR_INVALID_VALUE;
        ws->debug_late_ticks = value;
        return CAVMD_OK;
    }
    if (!strcmp(name, "debug_silent_block"))
    {
        if (value < -1 || value >= (int)kMaxPersistGrid)
            return CAVMD_ERR_INVALID_VALUE;
        ws->debug_silent_block = value;
        return CAVMD_OK;
    }
    if (!strcmp(name, "debug_skip_publish"))
    {
        if (value < 0 || value > 1)
            return CAVMD_ERR_INVALID_VALUE;
        if (value && !ws->h_scratch)
        {
            DeviceGuard guard(ws->device);
            CAVMD_HIP_TRY(hipHostMalloc((void**)&ws->h_scratch, sizeof(HostResult), hipHostMallocMapped | hipHostMallocCoherent));
            memset(ws->h_scratch, 0, sizeof(HostResult));
            CAVMD_HIP_TRY(hipHostGetDevicePointer((void**)&ws->h_scratch_dev, ws->h_scratch, 0));
        }
        ws->debug_skip_publish = value;
        return CAVMD_OK;
    }
#endif
    if (!strcmp(name, "sync_timeout_seen"))
    {
        // 0 forgets a time-out seen earlier.  In the test-hooks build also a fault-injection hook: raises the flag of the
        // host-visible block as a starved single-launch kernel would -- 1: the evaluation failed, 2: its last block completed it
#ifdef CAVMD_TEST_HOOKS
        if (value < 0 || value > 2)
            return CAVMD_ERR_INVALID_VALUE;
        if (value)
            __atomic_store_n(&ws->h_result->sync_error, value == 2 ? kSyncRepaired : kSyncFailed, __ATOMIC_RELEASE);
        else
            ws->sync_timeout_seen = false;
#else
        if (value != 0) // raising the flag is a fault-injection hook: libcavmd_hooks.so only
            return CAVMD_ERR_INVALID_VALUE;
        ws->sync_timeout_seen = false;
#endif
        return CAVMD_OK;
    }
    if (!strcmp(name, "reduce_unroll"))
    {
        if (value != -1 && value != 1 && value != 2)
            return CAVMD_ERR_INVALID_VALUE;
        ws->reduce_unroll = value;
        return CAVMD_OK;
    }
    if (!strcmp(name, "rho_lane_particle"))
    {
        if (value < -1 || value > 3)
            return CAVMD_ERR_INVALID_VALUE;
        ws->rho_lane_particle = value;
        return CAVMD_OK;
    }
    if (!strcmp(name, "persistent_lds_kb"))
    {
        if (value < 0 || value > 156)
            return CAVMD_ERR_INVALID_VALUE;
        ws->persistent_lds_kb = value;
        return CAVMD_OK;
    }
    if (!strcmp(name, "persistent_balanced"))
    {
        if (value < -1 || value > 1)
            return CAVMD_ERR_INVALID_VALUE;
        ws->persistent_balanced = value;
        return CAVMD_OK;
    }
    return CAVMD_ERR_INVALID_VALUE;
}

int cavmd_get_tunable(cavmd_workspace* ws, const char* name, int* value)
{
    if (!ws || !name || !value)
        return CAVMD_ERR_INVALID_VALUE;
    if (!strcmp(name, "reduce_blocks_per_cu"))
        *value = ws->reduce_blocks_per_cu;
    else if (!strcmp(name, "map_blocks_per_cu"))
        *value = ws->map_blocks_per_cu;
    else if (!strcmp(name, "map_nt_store"))
        *value = ws->map_nt_store;
    else if (!strcmp(name, "reduce_nt_load"))
        *value = ws->reduce_nt_load;
    else if (!strcmp(name, "fused_finalize"))
        *value = ws->fused_finalize;
    else if (!strcmp(name, "map_reverse"))
        *value = ws->map_reverse;
    else if (!strcmp(name, "small_system_max_n"))
        *value = ws->small_system_max_n;
    else if (!strcmp(name, "persistent"))
        *value = ws->persistent;
    else if (!strcmp(name, "sync_timeout_seen"))
        *value = ws->sync_timeout_seen ? 1 : 0;
    else if (!strcmp(name, "persistent_suspended"))
        *value = ws->suspend_until == kSuspendForever ? 2 : (ws->sequence < ws->suspend_until ? 1 : 0);
#ifdef CAVMD_TEST_HOOKS
    else if (!strcmp(name, "debug_spin_limit"))
        *value = ws->debug_spin_limit;
    else if (!strcmp(name, "debug_late_block"))
        *value = ws->debug_late_block;
    else if (!strcmp(name, "debug_late_ticks"))
        *value = ws->debug_late_ticks;
    else if (!strcmp(name, "debug_silent_block"))
        *value = ws->debug_silent_block;
    else if (!strcmp(name, "debug_skip_publish"))
        *value = ws->debug_skip_publish;
    else if (!strcmp(name, "test_hooks"))
        *value = 1;
#else
    else if (!strcmp(name, "test_hooks"))
        *value = 0;
#endif
    else if (!strcmp(name, "reduce_unroll"))
        *value = ws->reduce_unroll;
    else if (!strcmp(name, "rho_lane_particle"))
        *value = ws->rho_lane_particle;
    else if (!strcmp(name, "persistent_lds_kb"))
        *value = ws->persistent_lds_kb;
    else if (!strcmp(name, "persistent_balanced"))
        *value = ws->persistent_balanced;
    else
        return CAVMD_ERR_INVALID_VALUE;
    return CAVMD_OK;
}

int cavmd_device_info(cavmd_workspace* ws, int* device, int* compute_units, char* arch_name, size_t arch_name_len)
{
    if (!ws)
        return CAVMD_ERR_INVALID_VALUE;
    if (device)
        *device = ws->device;
    if (compute_units)
        *compute_units = ws->num_cu;
    if (arch_name && arch_name_len)
    {
        strncpy(arch_name, ws->arch, arch_name_len - 1);
        arch_name[arch_name_len - 1] = 0;
    }
    return CAVMD_OK;
}

const char* cavmd_error_string(int status)
{
    switch (status)
    {
    case CAVMD_OK:
        return "success";
    case CAVMD_ERR_INVALID_VALUE:
        return "invalid value (null or misaligned pointer, bad stride or size)";
    case CAVMD_ERR_NO_DEVICE:
        return "no HIP device available (this library has no CPU fallback)";
    case CAVMD_ERR_CAPACITY:
        return "N exceeds the workspace capacity";
    case CAVMD_ERR_BAD_PARAMS:
        return "bad cavity parameters (K == 0 or non-finite)";
    case CAVMD_ERR_NOT_COMPUTED:
        return "no evaluation has been enqueued on this workspace yet";
    case CAVMD_ERR_SYNC_TIMEOUT:
        return "a single-launch evaluation was starved (its workgroups were not resident together) and could not be completed; "
               "forces of that evaluation are NaN";
    default:
        break;
    }
    if (status > 0)
        return hipGetErrorString((hipError_t)status);
    return "unknown cavmd status";
}

int cavmd_version(void)
{
    return CAVMD_VERSION_MAJOR * 1000 + CAVMD_VERSION_MINOR;
}

} // extern "C"
