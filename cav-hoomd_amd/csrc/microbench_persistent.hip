// microbench_persistent.hip -- developer tool (not part of the product or of bench.py): the single-launch evaluation
// (cavmd_persistent_kernel.hpp) against the two-launch path, interleaved round by round on HBM-cold frames, plus a
// per-block time line of the single-launch kernel (wall_clock64 stamps, 10 ns resolution).
//
//   ./microbench_persistent [N=1000001] [frames=7] [rounds=9] [launches_per_round=20]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <random>
#include <string>
#include <vector>

#include "cavmd.h"

constexpr int kStampSlots = 8;
__device__ unsigned long long g_pstamps[4096 * kStampSlots];
__device__ unsigned g_xcc[4096];
#define CAVMD_PSTAMP(k)                                                \
    do                                                                 \
    {                                                                  \
        if (threadIdx.x == 0)                                          \
        {                                                              \
            g_pstamps[blockIdx.x * kStampSlots + (k)] = wall_clock64(); \
            if ((k) == 0)                                              \
                g_xcc[blockIdx.x] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 0xF; /* HW_REG_XCC_ID[3:0] */ \
        }                                                              \
    } while (0)
#include "cavmd_kernels.hpp"

using namespace cavmd;

#define CHECK(x)                                                                          \
    do                                                                                    \
    {                                                                                     \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess)                                                             \
        {                                                                                 \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

struct Variant
{
    std::string name;
    std::function<void(int frame)> launch;
    std::vector<double> us;
};

int main(int argc, char** argv)
{
    const size_t N = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1000001;
    const int frames = argc > 2 ? atoi(argv[2]) : 7;
    const int rounds = argc > 3 ? atoi(argv[3]) : 9;
    const int per_round = argc > 4 ? atoi(argv[4]) : 20;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int CU = prop.multiProcessorCount;
    printf("device %s, %d CUs; N=%zu frames=%d rounds=%d launches/round=%d\n", prop.gcnArchName, CU, N, frames, rounds,
           per_round);

    std::mt19937_64 rng(1234);
    std::uniform_real_distribution<double> U(-100.0, 100.0), C(-1.0, 1.0);
    std::vector<cavmd_double4> h_pos(N);
    std::vector<double> h_chg(N);
    std::vector<cavmd_int3> h_img(N);
    for (size_t i = 0; i < N; ++i)
    {
        h_pos[i].x = U(rng); h_pos[i].y = U(rng); h_pos[i].z = U(rng);
        uint64_t tag = (i == N - 1) ? 2 : (i & 1);
        memcpy(&h_pos[i].w, &tag, 8);
        h_chg[i] = (i == N - 1) ? 0.0 : C(rng);
        h_img[i].x = (int)(rng() % 5) - 2; h_img[i].y = (int)(rng() % 5) - 2; h_img[i].z = (int)(rng() % 5) - 2;
    }
    std::vector<cavmd_double4*> d_pos(frames), d_frc(frames);
    std::vector<double*> d_chg(frames);
    std::vector<cavmd_int3*> d_img(frames);
    // CAVMD_SLAB=1: all frames carved from ONE allocation (2 MiB-aligned pieces) instead of four hipMallocs per frame -- a
    // diagnostic for the placement dependence of the phase-1 spread
    char* slab = nullptr;
    size_t slab_off = 0;
    auto piece = [&](size_t bytes) { char* p = slab + slab_off; slab_off += (bytes + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1); return p; };
    if (getenv("CAVMD_SLAB"))
        CHECK(hipMalloc((void**)&slab, (size_t)frames * (N * 84 + (8u << 20)) + (2u << 20)));
    for (int f = 0; f < frames; ++f)
    {
        if (slab)
        {
            d_pos[f] = (cavmd_double4*)piece(N * 32); d_frc[f] = (cavmd_double4*)piece(N * 32);
            d_chg[f] = (double*)piece(N * 8);          d_img[f] = (cavmd_int3*)piece(N * 12);
        }
        else
        {
        CHECK(hipMalloc((void**)&d_pos[f], N * 32)); CHECK(hipMalloc((void**)&d_frc[f], N * 32));
        CHECK(hipMalloc((void**)&d_chg[f], N * 8));  CHECK(hipMalloc((void**)&d_img[f], N * 12));
        }
        CHECK(hipMemcpy(d_pos[f], h_pos.data(), N * 32, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(d_chg[f], h_chg.data(), N * 8, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(d_img[f], h_img.data(), N * 12, hipMemcpyHostToDevice));
    }
    for (int f = 0; f < frames; ++f)
        printf("frame %d: pos %p  force %p  charge %p  image %p\n", f, (void*)d_pos[f], (void*)d_frc[f], (void*)d_chg[f], (void*)d_img[f]);
    const unsigned max_parts = CU * 16;
    double* d_part; int* d_ipart; cavmd_result* d_res; HostResult* d_hres;
    unsigned long long* d_gran; unsigned* d_epoch;
    CHECK(hipMalloc((void**)&d_part, sizeof(double) * kNumPartDoubles * max_parts));
    CHECK(hipMalloc((void**)&d_ipart, sizeof(int) * kNumPartInts * max_parts));
    CHECK(hipMalloc((void**)&d_res, sizeof(cavmd_result)));
    CHECK(hipMalloc((void**)&d_hres, sizeof(HostResult)));
    CHECK(hipMalloc((void**)&d_gran, sizeof(unsigned long long) * 2 * kGranulesPerRecord * kMaxPersistGrid));
    CHECK(hipMemset(d_gran, 0, sizeof(unsigned long long) * 2 * kGranulesPerRecord * kMaxPersistGrid));
    CHECK(hipMalloc((void**)&d_epoch, 16));
    const unsigned epoch_init[4] = {1u, 0u, 0u, 0u};
    CHECK(hipMemcpy(d_epoch, epoch_init, 16, hipMemcpyHostToDevice));
    Partials part {d_part, d_ipart, max_parts};
    SyncState sync {d_gran, d_epoch, kSpinLimit, -1, 0u, -1};
    const double L = 215.4;
    const unsigned n = (unsigned)N;
    hipStream_t st = 0;
    DeviceParams P; P.g = 1e-3; P.K = 0.0091 * 0.0091; P.gK = P.g / P.K; P.g2K = P.g * P.g / P.K;

    auto in0 = [&](int f) { AosInput in; in.pos2 = (const v2d*)d_pos[f]; in.charge = d_chg[f]; in.image = (const int*)d_img[f]; return in; };
    auto in1 = [&](int f) { AosInputT<1> in; in.pos2 = (const v2d*)d_pos[f]; in.charge = d_chg[f]; in.image = (const int*)d_img[f]; return in; };
    auto in2 = [&](int f) { AosInputT<2> in; in.pos2 = (const v2d*)d_pos[f]; in.charge = d_chg[f]; in.image = (const int*)d_img[f]; return in; };
    auto grid = [&](size_t items, unsigned tile, int bpc) {
        size_t tiles = (items + tile - 1) / tile; size_t cap = (size_t)CU * bpc; return (unsigned)std::max<size_t>(1, std::min(tiles, cap)); };

    int unroll = 2;
    while (unroll > 1 && N / ((size_t)256 * unroll) < (size_t)CU * 5 / 4)
        unroll >>= 1;
    const bool nts = N >= 200000;
#define ALLOW1(UNR, NTS, EZ) CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&cavity_persistent_kernel<256, UNR, NTS, false, EZ>), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024))
#define ALLOW(UNR, NTS) do { ALLOW1(UNR, NTS, 0); ALLOW1(UNR, NTS, 1); ALLOW1(UNR, NTS, 2); } while (0)
    ALLOW(1, false); ALLOW(1, true); ALLOW(2, false); ALLOW(2, true);
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&cavity_persistent_kernel<512, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&cavity_persistent_kernel<512, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));

#if defined(CAVMD_FAULT_SILENT_BLOCK) || defined(CAVMD_FAULT_LATE_BLOCK)
    {
        // Fault injection, one launch.
        //   silent block: a block never publishes its record.  Expected: every block gives up after its bounded wait, the last
        //     one cannot complete the evaluation either (a record is missing): the kernel ends, sync_error = 1 (failed), no
        //     result published, every force entry NaN.
        //   late block: a block starts after all the others have given up (as if its CU had been held by another grid).
        //     Expected: it gives up in turn, is the last to do so and completes the whole evaluation alone: sync_error = 2
        //     (repaired), result published, forces and dipole bit for bit those of the two-launch path.
        std::vector<double> h_f(4 * N, 0.0), h_want(4 * N, 0.0);
        const unsigned g1 = grid(N, 256 * unroll, 1);
        const size_t tile = 256 * unroll;
        const size_t slots = ((N + tile - 1) / tile + g1 - 1) / g1;
        // the two-launch path on the same frame: what a repaired evaluation has to reproduce
        if (unroll == 2)
            hipLaunchKernelGGL((dipole_partials_kernel<AosInputT<1>, 256, 2, false>), dim3(g1), dim3(256), 0, st, in1(0), n, L, L, L, 2, part);
        else
            hipLaunchKernelGGL((dipole_partials_kernel<AosInputT<1>, 256, 1, false>), dim3(g1), dim3(256), 0, st, in1(0), n, L, L, L, 2, part);
        hipLaunchKernelGGL((force_map_aos_fused_kernel<256, 4, false>), dim3(grid(2 * N, 1024, 2)), dim3(256), 0, st, in0(0), n, g1, L, L, L, P, 2, part, 1ull, d_res, d_hres, (v2d*)d_frc[0], false);
        CHECK(hipDeviceSynchronize());
        cavmd_result want_res; CHECK(hipMemcpy(&want_res, d_res, sizeof(want_res), hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(h_want.data(), d_frc[0], 32 * N, hipMemcpyDeviceToHost));
        CHECK(hipMemset(d_frc[0], 0x5A, 32 * N));
        CHECK(hipMemset(d_hres, 0, sizeof(HostResult)));
        CHECK(hipMemset(d_res, 0, sizeof(cavmd_result)));
        hipEvent_t a, b2;
        CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b2));
        CHECK(hipEventRecord(a, st));
#ifdef CAVMD_FAULT_LATE_BLOCK
        SyncState fsync = sync;
        fsync.spin_limit = 20000;                    // ~8 ms instead of ~0.3 s
        fsync.late_block = CAVMD_FAULT_LATE_BLOCK;
        fsync.late_ticks = 6000000;                  // 60 ms of the 100 MHz clock
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&cavity_persistent_kernel<256, 2, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&cavity_persistent_kernel<256, 1, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
        if (unroll == 2)
            hipLaunchKernelGGL((cavity_persistent_kernel<256, 2, true, true>), dim3(g1), dim3(256), slots * tile * 8, st, in2(0), n, L, L, L, P, 2, fsync, 7ull, d_res, d_hres, (v2d*)d_frc[0], (unsigned)slots, false);
        else
            hipLaunchKernelGGL((cavity_persistent_kernel<256, 1, true, true>), dim3(g1), dim3(256), slots * tile * 8, st, in2(0), n, L, L, L, P, 2, fsync, 7ull, d_res, d_hres, (v2d*)d_frc[0], (unsigned)slots, false);
#else
        if (unroll == 2)
            hipLaunchKernelGGL((cavity_persistent_kernel<256, 2, true>), dim3(g1), dim3(256), slots * tile * 8, st, in2(0), n, L, L, L, P, 2, sync, 7ull, d_res, d_hres, (v2d*)d_frc[0], (unsigned)slots, false);
        else
            hipLaunchKernelGGL((cavity_persistent_kernel<256, 1, true>), dim3(g1), dim3(256), slots * tile * 8, st, in2(0), n, L, L, L, P, 2, sync, 7ull, d_res, d_hres, (v2d*)d_frc[0], (unsigned)slots, false);
#endif
        CHECK(hipEventRecord(b2, st));
        CHECK(hipEventSynchronize(b2));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b2));
        HostResult hr; CHECK(hipMemcpy(&hr, d_hres, sizeof(hr), hipMemcpyDeviceToHost));
        cavmd_result got_res; CHECK(hipMemcpy(&got_res, d_res, sizeof(got_res), hipMemcpyDeviceToHost));
        unsigned after[3]; CHECK(hipMemcpy(after, d_epoch, 12, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(h_f.data(), d_frc[0], 32 * N, hipMemcpyDeviceToHost));
        size_t nan = 0, differ = 0;
        for (size_t i = 0; i < h_f.size(); ++i) { nan += (h_f[i] != h_f[i]); differ += memcmp(&h_f[i], &h_want[i], 8) != 0; }
        const bool same_dipole = memcmp(got_res.dipole, want_res.dipole, sizeof(got_res.dipole)) == 0
                                 && memcmp(hr.result.dipole, want_res.dipole, sizeof(got_res.dipole)) == 0;
#ifdef CAVMD_FAULT_SILENT_BLOCK
        (void)differ; (void)same_dipole;
        printf("fault injection (block %d silent, grid %u): kernel returned after %.1f ms, sync_error=%u, ready=%llu, NaN force entries %zu of %zu, epoch %u, give-up count left %u, poison word %s\n",
               CAVMD_FAULT_SILENT_BLOCK, g1, ms, hr.sync_error, (unsigned long long)hr.ready, nan, h_f.size(), after[0], after[1], after[2] ? "set" : "clear");
        // (with at most 16 blocks every block gathers the block records itself: the silent block, which has its own record in
        // registers, completes its own tile and leaves; the others can never be complete -- a stuck give-up count and an epoch
        // that was not advanced are what the host wipes before it would use the single launch again)
        return (hr.sync_error == 1 && nan >= h_f.size() - 4 * (size_t)tile && hr.ready == 0) ? 0 : 1;
#else
        printf("fault injection (block %d late, grid %u): kernel returned after %.1f ms, sync_error=%u, ready=%llu, NaN force entries %zu, force entries differing from the two-launch path %zu of %zu, dipole identical %d, epoch %u, give-up count left %u, poison word %s\n",
               CAVMD_FAULT_LATE_BLOCK, g1, ms, hr.sync_error, (unsigned long long)hr.ready, nan, differ, h_f.size(), (int)same_dipole, after[0], after[1], after[2] ? "set" : "clear");
        return (hr.sync_error == 2 && nan == 0 && differ == 0 && hr.ready == 7 && same_dipole && after[0] == 2 && after[1] == 0 && after[2] == 0) ? 0 : 1;
#endif
    }
#endif
    std::vector<Variant> V;
    // the product's two-launch sequence
    V.push_back({"two launches (K1 nt1 bpc1 | fused map u4 bpc2)", [&](int f) {
                     const unsigned g1 = grid(N, 256 * unroll, 1);
                     if (unroll == 2)
                         hipLaunchKernelGGL((dipole_partials_kernel<AosInputT<1>, 256, 2, false>), dim3(g1), dim3(256), 0, st, in1(f), n, L, L, L, 2, part);
                     else
                         hipLaunchKernelGGL((dipole_partials_kernel<AosInputT<1>, 256, 1, false>), dim3(g1), dim3(256), 0, st, in1(f), n, L, L, L, 2, part);
                     const unsigned g2 = grid(2 * N, 1024, 2);
                     if (nts)
                         hipLaunchKernelGGL((force_map_aos_fused_kernel<256, 4, true>), dim3(g2), dim3(256), 0, st, in0(f), n, g1, L, L, L, P, 2, part, 1ull, d_res, d_hres, (v2d*)d_frc[f], false);
                     else
                         hipLaunchKernelGGL((force_map_aos_fused_kernel<256, 4, false>), dim3(g2), dim3(256), 0, st, in0(f), n, g1, L, L, L, P, 2, part, 1ull, d_res, d_hres, (v2d*)d_frc[f], false);
                 }, {}});
    auto persist = [&](int f, int bpc, int balanced = 0, int earlyz = 0) {
        const unsigned g1 = grid(N, 256 * unroll, bpc);
        const size_t tile = 256 * unroll;
        size_t slots = ((N + tile - 1) / tile + g1 - 1) / g1;
        if (balanced == 1) { const size_t units = (N + 63) / 64; slots = (((units + g1 - 1) / g1) * 64 + tile - 1) / tile; }
        if (balanced == 2) slots = (N / tile) / g1 + 1;
        const size_t cap = (156 * 1024 - 1024) / (tile * 8);
        const unsigned lds_slots = (unsigned)std::min(slots, cap);
        const size_t lds = (size_t)lds_slots * tile * 8;
#define PL(UNR, NTS, EZ) hipLaunchKernelGGL((cavity_persistent_kernel<256, UNR, NTS, false, EZ>), dim3(g1), dim3(256), lds, st, in2(f), n, L, L, L, P, 2, sync, 1ull, d_res, d_hres, (v2d*)d_frc[f], lds_slots, balanced)
#define PLZ(UNR, NTS) do { if (earlyz == 0) PL(UNR, NTS, 0); else if (earlyz == 1) PL(UNR, NTS, 1); else PL(UNR, NTS, 2); } while (0)
        if (unroll == 2) { if (nts) PLZ(2, true); else PLZ(2, false); }
        else { if (nts) PLZ(1, true); else PLZ(1, false); }
    };
    V.push_back({"single launch, tiles dealt round-robin", [&](int f) { persist(f, 1, 0); }, {}});
    V.push_back({"single launch, balanced contiguous shares", [&](int f) { persist(f, 1, 1); }, {}});
    V.push_back({"single launch, full rounds strided + last round split evenly", [&](int f) { persist(f, 1, 2); }, {}});
    if (getenv("CAVMD_TRY_EARLYZ")) {
        V.push_back({"single launch, zero chunks early (plain stores)", [&](int f) { persist(f, 1, 0, 1); }, {}});
        V.push_back({"single launch, zero chunks early (nt stores)", [&](int f) { persist(f, 1, 0, 2); }, {}});
    }
    auto persist512 = [&](int f, int unr) {
        const size_t tile = 512 * unr;
        const unsigned g1 = grid(N, tile, 1);
        size_t slots = ((N + tile - 1) / tile + g1 - 1) / g1;
        const size_t cap = (156 * 1024 - 1024) / (tile * 8);
        const unsigned lds_slots = (unsigned)std::min(slots, cap);
        const size_t lds = (size_t)lds_slots * tile * 8;
        if (unr == 1) hipLaunchKernelGGL((cavity_persistent_kernel<512, 1, true>), dim3(g1), dim3(512), lds, st, in2(f), n, L, L, L, P, 2, sync, 1ull, d_res, d_hres, (v2d*)d_frc[f], lds_slots, false);
        else hipLaunchKernelGGL((cavity_persistent_kernel<512, 2, true>), dim3(g1), dim3(512), lds, st, in2(f), n, L, L, L, P, 2, sync, 1ull, d_res, d_hres, (v2d*)d_frc[f], lds_slots, false);
    };
    if (getenv("CAVMD_TRY_512")) {
        V.push_back({"single launch, 512 threads x 1", [&](int f) { persist512(f, 1); }, {}});
        V.push_back({"single launch, 512 threads x 2", [&](int f) { persist512(f, 2); }, {}});
    }

    {
        // every variant must give the bits of the first one (the product's two-launch path) in every force entry -- except the
        // variants that partition the particles differently (other summation grouping: last bits of the dipole may differ)
        std::vector<double> want(4 * N), got(4 * N);
        for (size_t vi = 0; vi < V.size(); ++vi)
        {
            CHECK(hipMemset(d_frc[0], 0x5A, 32 * N));
            V[vi].launch(0);
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy((vi == 0 ? want : got).data(), d_frc[0], 32 * N, hipMemcpyDeviceToHost));
            if (vi > 0)
            {
                size_t differ = 0; double worst = 0;
                for (size_t i = 0; i < got.size(); ++i)
                {
                    differ += memcmp(&got[i], &want[i], 8) != 0;
                    const double d = got[i] - want[i], sc = want[i] < 0 ? -want[i] : want[i];
                    if (sc > 0 && (d < 0 ? -d : d) / sc > worst) worst = (d < 0 ? -d : d) / sc;
                }
                printf("check %-60s entries differing from two launches: %zu of %zu (worst relative %.1e)\n", V[vi].name.c_str(), differ, got.size(), worst);
            }
        }
    }
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    int frame = 0;
    for (auto& v : V) for (int f = 0; f < frames; ++f) v.launch(f);
    CHECK(hipDeviceSynchronize());
    for (int r = 0; r < rounds; ++r)
    {
        for (auto& v : V)
        {
            CHECK(hipEventRecord(e0, st));
            for (int k = 0; k < per_round; ++k) { v.launch(frame); frame = (frame + 1) % frames; }
            CHECK(hipEventRecord(e1, st));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            v.us.push_back(1e3 * ms / per_round);
        }
    }
    CHECK(hipGetLastError());
    printf("%-52s %10s %10s %12s\n", "variant", "med us", "min us", "evals/s(med)");
    for (auto& v : V)
    {
        std::sort(v.us.begin(), v.us.end());
        const double med = v.us[v.us.size() / 2], mn = v.us.front();
        printf("%-52s %10.2f %10.2f %12.0f\n", v.name.c_str(), med, mn, 1e6 / med);
    }

    // ---- time line of the single-launch kernel: per-block stamps, relative to the first block's start ----------------------
    const int timeline_earlyz = getenv("CAVMD_TIMELINE_EARLYZ") ? atoi(getenv("CAVMD_TIMELINE_EARLYZ")) : 0;
    const int timeline_partition = getenv("CAVMD_TIMELINE_PARTITION") ? atoi(getenv("CAVMD_TIMELINE_PARTITION")) : 0;
    printf("time line of the variant with partition %d, EARLYZ %d\n", timeline_partition, timeline_earlyz);
    for (int mode = 0; mode < 1; ++mode)
    {
        const unsigned g1 = grid(N, 256 * unroll, 1);
        const char* names[8] = {"start", "phase1 done", "block tree done", "total in wave 0", "scalars in LDS",
                                "barrier passed", "stores issued", "group level done"};
        std::vector<std::vector<double>> med(8), mx(8), mnv(8);
        std::vector<std::vector<double>> xcc_p1(30, std::vector<double>(8, 0.0));
        std::vector<int> xcc_frame(30, 0);
        for (int rep = 0; rep < 30; ++rep)
        {
            for (int k = 0; k < 12; ++k) persist((rep + k) % frames, 1, timeline_partition, timeline_earlyz); // steady state: the stamps are those of the last launch
            CHECK(hipDeviceSynchronize());
            std::vector<unsigned long long> h(4096 * kStampSlots);
            CHECK(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_pstamps), sizeof(unsigned long long) * h.size()));
            unsigned long long t0 = ~0ull;
            for (unsigned b = 0; b < g1; ++b) t0 = std::min(t0, h[b * kStampSlots]);
            {
                std::vector<unsigned> hx(4096);
                CHECK(hipMemcpyFromSymbol(hx.data(), HIP_SYMBOL(g_xcc), sizeof(unsigned) * hx.size()));
                std::vector<double> sum(8, 0.0); std::vector<int> cnt(8, 0);
                for (unsigned b = 0; b < g1; ++b) { const unsigned x = hx[b] & 7; sum[x] += (double)(h[b * kStampSlots + 1] - t0) * 0.01; cnt[x]++; }
                for (int x = 0; x < 8; ++x) xcc_p1[rep][x] = cnt[x] ? sum[x] / cnt[x] : 0.0;
                xcc_frame[rep] = (rep + 11) % frames;
                if (rep == 29) { printf("  blocks per XCC_ID in the last run:"); for (int x = 0; x < 8; ++x) printf(" %d", cnt[x]); printf("\n"); }
            }
            if (rep == 29 && getenv("CAVMD_DUMP_BLOCKS"))
            {
                printf("  per block: b tiles start phase1_done tree_done\n");
                const unsigned full_tiles = (unsigned)(N / (256 * unroll));
                for (unsigned b = 0; b < g1; ++b)
                {
                    const unsigned nt = full_tiles > b ? (full_tiles - b + g1 - 1) / g1 : 0;
                    printf("  blk %3u %u %.2f %.2f %.2f\n", b, nt, (double)(h[b * kStampSlots] - t0) * 0.01,
                           (double)(h[b * kStampSlots + 1] - t0) * 0.01, (double)(h[b * kStampSlots + 2] - t0) * 0.01);
                }
            }
            if (rep == 29)
            {
                printf("  per blockIdx%%8 group (start / tree done / fold done, us): ");
                for (unsigned x = 0; x < 8; ++x)
                {
                    double a = 1e9, c = 0, d = 0; unsigned cnt = 0;
                    for (unsigned b = x; b < g1; b += 8) { a = std::min(a, (double)(h[b * kStampSlots] - t0) * 0.01); c = std::max(c, (double)(h[b * kStampSlots + 2] - t0) * 0.01); d += (double)(h[b * kStampSlots + 4] - t0) * 0.01; ++cnt; }
                    printf("[%u: %.2f %.2f %.2f] ", x, a, c, d / cnt);
                }
                printf("\n  blocks 0..15 start: ");
                for (unsigned b = 0; b < 16 && b < g1; ++b) printf("%.2f ", (double)(h[b * kStampSlots] - t0) * 0.01);
                printf("\n  blocks 0..15 fold done: ");
                for (unsigned b = 0; b < 16 && b < g1; ++b) printf("%.2f ", (double)(h[b * kStampSlots + 4] - t0) * 0.01);
                printf("\n");
            }
            for (int k = 0; k < 8; ++k)
            {
                std::vector<double> v(g1);
                for (unsigned b = 0; b < g1; ++b) v[b] = (double)(h[b * kStampSlots + k] - t0) * 0.01; // 100 MHz -> us
                std::sort(v.begin(), v.end());
                med[k].push_back(v[g1 / 2]); mx[k].push_back(v.back()); mnv[k].push_back(v.front());
            }
        }
        printf("  mean phase-1-done per XCC_ID (us), one row per run (12 back-to-back launches each, stamps of the last):\n");
        for (int rep = 0; rep < 30; ++rep) { printf("   run %2d (frame %d):", rep, xcc_frame[rep]); for (int x = 0; x < 8; ++x) printf(" %6.2f", xcc_p1[rep][x]); printf("\n"); }
        printf("time line, single launch, grid %u (us after the first block's start; median over 30 runs of the per-run min / median / max over blocks)\n",
               g1);
        for (int k : {0, 1, 2, 7, 3, 4, 5, 6})
        {
            std::sort(med[k].begin(), med[k].end()); std::sort(mx[k].begin(), mx[k].end()); std::sort(mnv[k].begin(), mnv[k].end());
            printf("  %-22s min %6.2f  median %6.2f  max %6.2f\n", names[k], mnv[k][15], med[k][15], mx[k][15]);
        }
    }
    return 0;
}
