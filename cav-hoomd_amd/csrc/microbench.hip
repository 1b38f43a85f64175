// microbench.hip -- developer tool (not part of the product or of bench.py): times variants of the cavity-force
// kernels against each other in ONE process, interleaved round by round, on HBM-cold data.
//
//   ./microbench [N=10000000] [frames=2] [rounds=9] [launches_per_round=5]
//
// Prints, per variant: median and min microseconds per launch and the achieved GB/s on the algorithmic byte
// count.  Includes pure read / pure write ceilings with the same access shapes.
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

// s_memtime stamps at numbered points of the finalize chain (block 0, thread 0 only); see cavmd_kernels.hpp
__device__ unsigned long long g_stamps[16];
#define CAVMD_STAMP(k)                                                                           \
    do                                                                                           \
    {                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        unsigned long long t_;                                                                   \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");              \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        if (threadIdx.x == 0 && blockIdx.x == 0)                                                 \
            g_stamps[k] = t_;                                                                    \
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

// ---- ceilings ------------------------------------------------------------------------------------------------
// reads the three input streams with the production access shapes, minimal arithmetic
template <int BLOCK, int UNROLL>
__global__ __launch_bounds__(BLOCK) void read_ceiling_kernel(AosInput in, unsigned N, double* out)
{
    constexpr unsigned TILE = BLOCK * UNROLL;
    double acc = 0.0;
    int iacc = 0;
    const unsigned full_tiles = N / TILE;
    for (unsigned t = blockIdx.x; t < full_tiles; t += gridDim.x)
    {
        const size_t base = (size_t)t * TILE + threadIdx.x;
        AosInput::Raw raw[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
            raw[u] = in.load(base + (size_t)u * BLOCK);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
        {
            acc += raw[u].xy.x + raw[u].xy.y + raw[u].zw.x + raw[u].zw.y + raw[u].c;
            iacc += raw[u].ix + raw[u].iy + raw[u].iz;
        }
    }
    if (acc == 1.2345e300 && iacc == 77)
        out[0] = acc;
}

// one flat 16-B/lane stream over `bytes` (float4-copy style read)
template <int BLOCK, int UNROLL>
__global__ __launch_bounds__(BLOCK) void flat_read_kernel(const v2d* __restrict__ p, size_t nvec, double* out)
{
    constexpr size_t TILE = (size_t)BLOCK * UNROLL;
    v2d acc = {0.0, 0.0};
    const size_t full = nvec / TILE;
    for (size_t t = blockIdx.x; t < full; t += gridDim.x)
    {
        v2d v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
            v[u] = p[t * TILE + (size_t)u * BLOCK + threadIdx.x];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
            acc += v[u];
    }
    if (acc.x == 1.2345e300)
        out[0] = acc.y;
}

template <int BLOCK, int UNROLL, bool NT>
__global__ __launch_bounds__(BLOCK) void flat_write_kernel(v2d* __restrict__ p, size_t nvec)
{
    constexpr size_t TILE = (size_t)BLOCK * UNROLL;
    const v2d z = {1.0, 2.0};
    const size_t full = nvec / TILE;
    for (size_t t = blockIdx.x; t < full; t += gridDim.x)
    {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
            store_chunk<NT>(p + t * TILE + (size_t)u * BLOCK + threadIdx.x, z);
    }
}

// the "natural" force map: one particle per lane, two 16-B stores at 32-B lane stride
template <int BLOCK, int UNROLL, bool NT>
__global__ __launch_bounds__(BLOCK) void force_map_natural_kernel(const double* __restrict__ charge, unsigned N, double g,
                                                                  const cavmd_result* __restrict__ res,
                                                                  v2d* __restrict__ force2)
{
    constexpr unsigned TILE = BLOCK * UNROLL;
    const double Dqx = res->Dq[0], Dqy = res->Dq[1];
    const double ng = -g;
    const v2d zero = {0.0, 0.0};
    const unsigned full_tiles = N / TILE;
    for (unsigned t = blockIdx.x; t < full_tiles; t += gridDim.x)
    {
        const size_t base = (size_t)t * TILE + threadIdx.x;
        double c[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
            c[u] = charge[base + (size_t)u * BLOCK];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
        {
            const size_t i = base + (size_t)u * BLOCK;
            const double s = ng * c[u];
            const v2d v = {s * Dqx, s * Dqy};
            store_chunk<NT>(force2 + 2 * i, v);
            store_chunk<NT>(force2 + 2 * i + 1, zero);
        }
    }
}

// ---- harness -----------------------------------------------------------------------------------------------------
struct Variant
{
    std::string name;
    double bytes; // algorithmic bytes per launch
    std::function<void(int frame)> launch;
    std::vector<double> us;
};

int main(int argc, char** argv)
{
    const size_t N = argc > 1 ? strtoull(argv[1], nullptr, 10) : 10000000;
    const int frames = argc > 2 ? atoi(argv[2]) : 2;
    const int rounds = argc > 3 ? atoi(argv[3]) : 9;
    const int per_round = argc > 4 ? atoi(argv[4]) : 5;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int CU = prop.multiProcessorCount;
    printf("device %s, %d CUs; N=%zu frames=%d rounds=%d launches/round=%d\n", prop.gcnArchName, CU, N, frames, rounds,
           per_round);

    // host data
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
    for (int f = 0; f < frames; ++f)
    {
        CHECK(hipMalloc((void**)&d_pos[f], N * 32)); CHECK(hipMalloc((void**)&d_frc[f], N * 32));
        CHECK(hipMalloc((void**)&d_chg[f], N * 8));  CHECK(hipMalloc((void**)&d_img[f], N * 12));
        CHECK(hipMemcpy(d_pos[f], h_pos.data(), N * 32, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(d_chg[f], h_chg.data(), N * 8, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(d_img[f], h_img.data(), N * 12, hipMemcpyHostToDevice));
    }
    const unsigned max_parts = CU * 16;
    double* d_part; int* d_ipart; cavmd_result* d_res; double* d_sink; HostResult* d_hres;
    CHECK(hipMalloc((void**)&d_part, sizeof(double) * kNumPartDoubles * max_parts));
    CHECK(hipMalloc((void**)&d_ipart, sizeof(int) * kNumPartInts * max_parts));
    CHECK(hipMalloc((void**)&d_res, sizeof(cavmd_result)));
    CHECK(hipMalloc((void**)&d_sink, 64));
    CHECK(hipMalloc((void**)&d_hres, sizeof(HostResult)));
    CHECK(hipMemset(d_part, 0, sizeof(double) * kNumPartDoubles * max_parts));
    CHECK(hipMemset(d_ipart, 0, sizeof(int) * kNumPartInts * max_parts));
    Partials part {d_part, d_ipart, max_parts};
    const double L = 215.4;
    const unsigned n = (unsigned)N;
    hipStream_t st = 0;

    auto in_of = [&](int f) { AosInput in; in.pos2 = (const v2d*)d_pos[f]; in.charge = d_chg[f]; in.image = (const int*)d_img[f]; return in; };
    auto in_nt2_of = [&](int f) { AosInputT<2> in; in.pos2 = (const v2d*)d_pos[f]; in.charge = d_chg[f]; in.image = (const int*)d_img[f]; return in; };
    auto in_nt1_of = [&](int f) { AosInputT<1> in; in.pos2 = (const v2d*)d_pos[f]; in.charge = d_chg[f]; in.image = (const int*)d_img[f]; return in; };
    auto grid = [&](size_t items, unsigned tile, int bpc) {
        size_t tiles = (items + tile - 1) / tile; size_t cap = (size_t)CU * bpc; return (unsigned)std::max<size_t>(1, std::min(tiles, cap)); };

    std::vector<Variant> V;
    const double B1 = 52.0 * N, B2 = 40.0 * N;
#define K1(NAME, BLOCK, UNROLL, PIPE, NTL, BPC)                                                                         \
    V.push_back({NAME, B1, [&](int f) {                                                                                 \
                     const unsigned g1 = grid(N, BLOCK * UNROLL, BPC);                                                  \
                     if (NTL == 2)                                                                                      \
                         hipLaunchKernelGGL((dipole_partials_kernel<AosInputT<2>, BLOCK, UNROLL, PIPE>), dim3(g1), dim3(BLOCK), 0, st, in_nt2_of(f), n, L, L, L, 2, part); \
                     else if (NTL == 1)                                                                                 \
                         hipLaunchKernelGGL((dipole_partials_kernel<AosInputT<1>, BLOCK, UNROLL, PIPE>), dim3(g1), dim3(BLOCK), 0, st, in_nt1_of(f), n, L, L, L, 2, part); \
                     else                                                                                               \
                         hipLaunchKernelGGL((dipole_partials_kernel<AosInput, BLOCK, UNROLL, PIPE>), dim3(g1), dim3(BLOCK), 0, st, in_of(f), n, L, L, L, 2, part); \
                 }, {}})
    K1("K1 simple nt2 bpc1", 256, 4, false, 2, 1);
    K1("K1 simple nt2 bpc2", 256, 4, false, 2, 2);
    K1("K1 simple nt2 bpc4", 256, 4, false, 2, 4);
    K1("K1 simple nt2 bpc8", 256, 4, false, 2, 8);
    K1("K1 dbuf   nt2 bpc1", 256, 4, true, 2, 1);
    K1("K1 dbuf   nt2 bpc2", 256, 4, true, 2, 2);
    K1("K1 dbuf   nt1 bpc1", 256, 4, true, 1, 1);
    K1("K1 dbuf   nt1 bpc2", 256, 4, true, 1, 2);
    K1("K1 dbuf u2 nt2 bpc2", 256, 2, true, 2, 2);
    K1("K1 dbuf u2 nt2 bpc4", 256, 2, true, 2, 4);
    K1("K1 dbuf b512 u2 nt2 bpc1", 512, 2, true, 2, 1);
    V.push_back({"READ ceiling (3 streams, same shapes) b256 u4 bpc8", B1, [&](int f) {
                     hipLaunchKernelGGL((read_ceiling_kernel<256, 4>), dim3(grid(N, 1024, 8)), dim3(256), 0, st, in_of(f), n, d_sink); }, {}});
    V.push_back({"READ flat 16B pos-array only b256 u4 bpc8", 32.0 * N, [&](int f) {
                     hipLaunchKernelGGL((flat_read_kernel<256, 4>), dim3(grid(2 * N, 1024, 8)), dim3(256), 0, st, (const v2d*)d_pos[f], 2 * N, d_sink); }, {}});
    V.push_back({"READ flat 16B pos-array only b256 u8 bpc8", 32.0 * N, [&](int f) {
                     hipLaunchKernelGGL((flat_read_kernel<256, 8>), dim3(grid(2 * N, 2048, 8)), dim3(256), 0, st, (const v2d*)d_pos[f], 2 * N, d_sink); }, {}});

#define K2(NAME, BLOCK, UNROLL, NTS, BPC)                                                                               \
    V.push_back({NAME, B2, [&](int f) {                                                                                 \
                     hipLaunchKernelGGL((force_map_aos_kernel<BLOCK, UNROLL, NTS>), dim3(grid(2 * N, BLOCK * UNROLL, BPC)), \
                                        dim3(BLOCK), 0, st, d_chg[f], (const v2d*)d_pos[f], n, 1e-3, 2, d_res, (v2d*)d_frc[f]); \
                 }, {}})
    K2("K2 chunk b256 u4 bpc8 (v1)", 256, 4, false, 8);
    K2("K2 chunk b256 u4 bpc8 nt", 256, 4, true, 8);
    K2("K2 chunk b256 u4 bpc16", 256, 4, false, 16);
    K2("K2 chunk b256 u4 bpc4", 256, 4, false, 4);
    K2("K2 chunk b256 u8 bpc8", 256, 8, false, 8);
    K2("K2 chunk b256 u8 bpc8 nt", 256, 8, true, 8);
    K2("K2 chunk b256 u2 bpc16", 256, 2, false, 16);
    K2("K2 chunk b512 u4 bpc4", 512, 4, false, 4);
    V.push_back({"K2 natural b256 u4 bpc8", B2, [&](int f) {
                     hipLaunchKernelGGL((force_map_natural_kernel<256, 4, false>), dim3(grid(N, 1024, 8)), dim3(256), 0, st, d_chg[f], n, 1e-3, d_res, (v2d*)d_frc[f]); }, {}});
    V.push_back({"K2 natural b256 u4 bpc8 nt", B2, [&](int f) {
                     hipLaunchKernelGGL((force_map_natural_kernel<256, 4, true>), dim3(grid(N, 1024, 8)), dim3(256), 0, st, d_chg[f], n, 1e-3, d_res, (v2d*)d_frc[f]); }, {}});
    V.push_back({"WRITE flat 16B force-array only b256 u4 bpc8", 32.0 * N, [&](int f) {
                     hipLaunchKernelGGL((flat_write_kernel<256, 4, false>), dim3(grid(2 * N, 1024, 8)), dim3(256), 0, st, (v2d*)d_frc[f], 2 * N); }, {}});
    V.push_back({"WRITE flat 16B force-array only b256 u4 bpc8 nt", 32.0 * N, [&](int f) {
                     hipLaunchKernelGGL((flat_write_kernel<256, 4, true>), dim3(grid(2 * N, 1024, 8)), dim3(256), 0, st, (v2d*)d_frc[f], 2 * N); }, {}});
    V.push_back({"finalize (2048 partials)", 0.0, [&](int f) {
                     DeviceParams p; p.g = 1e-3; p.K = 0.0091 * 0.0091; p.gK = p.g / p.K; p.g2K = p.g * p.g / p.K;
                     hipLaunchKernelGGL((finalize_kernel<AosInput, 256>), dim3(1), dim3(256), 0, st, in_of(f), n, 2048u, L, L, L, p, part, 1ull, d_res, d_hres); }, {}});
    V.push_back({"finalize (512 partials)", 0.0, [&](int f) {
                     DeviceParams p; p.g = 1e-3; p.K = 0.0091 * 0.0091; p.gK = p.g / p.K; p.g2K = p.g * p.g / p.K;
                     hipLaunchKernelGGL((finalize_kernel<AosInput, 256>), dim3(1), dim3(256), 0, st, in_of(f), n, 512u, L, L, L, p, part, 1ull, d_res, d_hres); }, {}});


    // ---- whole evaluations: reduce -> finalize -> map on the same frame, as cavmd_compute_hoomd enqueues them ----
    const double B = 92.0 * N;
    DeviceParams P; P.g = 1e-3; P.K = 0.0091 * 0.0091; P.gK = P.g / P.K; P.g2K = P.g * P.g / P.K;
#define SEQ3(NAME, KB, KU, KP, KNT, KBPC, MB, MU, MNT, MBPC)                                                                \
    V.push_back({NAME, B, [&](int f) {                                                                                  \
                     const unsigned g1 = grid(N, KB * KU, KBPC);                                                        \
                     if (KNT == 2)                                                                                      \
                         hipLaunchKernelGGL((dipole_partials_kernel<AosInputT<2>, KB, KU, KP>), dim3(g1), dim3(KB), 0, st, in_nt2_of(f), n, L, L, L, 2, part); \
                     else if (KNT == 1)                                                                                 \
                         hipLaunchKernelGGL((dipole_partials_kernel<AosInputT<1>, KB, KU, KP>), dim3(g1), dim3(KB), 0, st, in_nt1_of(f), n, L, L, L, 2, part); \
                     else                                                                                               \
                         hipLaunchKernelGGL((dipole_partials_kernel<AosInput, KB, KU, KP>), dim3(g1), dim3(KB), 0, st, in_of(f), n, L, L, L, 2, part); \
                     hipLaunchKernelGGL((finalize_kernel<AosInput, 256>), dim3(1), dim3(256), 0, st, in_of(f), n, g1, L, L, L, P, part, 1ull, d_res, d_hres); \
                     hipLaunchKernelGGL((force_map_aos_kernel<MB, MU, MNT>), dim3(grid(2 * N, MB * MU, MBPC)), dim3(MB), 0, st, d_chg[f], (const v2d*)d_pos[f], n, 1e-3, 2, d_res, (v2d*)d_frc[f]); \
                 }, {}})
#define SEQ2(NAME, KB, KU, KP, KNT, KBPC, MB, MU, MNT, MBPC)                                                                \
    V.push_back({NAME, B, [&](int f) {                                                                                  \
                     const unsigned g1 = grid(N, KB * KU, KBPC);                                                        \
                     if (KNT == 2)                                                                                      \
                         hipLaunchKernelGGL((dipole_partials_kernel<AosInputT<2>, KB, KU, KP>), dim3(g1), dim3(KB), 0, st, in_nt2_of(f), n, L, L, L, 2, part); \
                     else if (KNT == 1)                                                                                 \
                         hipLaunchKernelGGL((dipole_partials_kernel<AosInputT<1>, KB, KU, KP>), dim3(g1), dim3(KB), 0, st, in_nt1_of(f), n, L, L, L, 2, part); \
                     else                                                                                               \
                         hipLaunchKernelGGL((dipole_partials_kernel<AosInput, KB, KU, KP>), dim3(g1), dim3(KB), 0, st, in_of(f), n, L, L, L, 2, part); \
                     hipLaunchKernelGGL((force_map_aos_fused_kernel<MB, MU, MNT>), dim3(grid(2 * N, MB * MU, MBPC)), dim3(MB), 0, st, in_of(f), n, g1, L, L, L, P, 2, part, 1ull, d_res, d_hres, (v2d*)d_frc[f], false); \
                 }, {}})
    // NT = charge-load policy of the reduction (1 below 5e6 particles, 2 above), stores non-temporal
#define SWEEP(NT)                                                                                                   \
    SEQ2("SEQ2 prev: K1 u4 bpc1 | fused u4 bpc2 nts", 256, 4, false, NT, 1, 256, 4, true, 2);                       \
    SEQ2("SEQ2 K1 u2 bpc1 | fused u4 bpc2 nts", 256, 2, false, NT, 1, 256, 4, true, 2);                             \
    SEQ2("SEQ2 K1 u2 bpc1 | fused u8 bpc2 nts", 256, 2, false, NT, 1, 256, 8, true, 2);                             \
    SEQ2("SEQ2 K1 u2 bpc1 | fused u8 bpc3 nts", 256, 2, false, NT, 1, 256, 8, true, 3);                             \
    SEQ2("SEQ2 K1 u2 bpc1 | fused u8 bpc1 nts", 256, 2, false, NT, 1, 256, 8, true, 1);                             \
    SEQ2("SEQ2 K1 u2 bpc1 | fused u4 bpc3 nts", 256, 2, false, NT, 1, 256, 4, true, 3);                             \
    SEQ2("SEQ2 K1 u1 bpc1 | fused u4 bpc2 nts", 256, 1, false, NT, 1, 256, 4, true, 2);                             \
    SEQ2("SEQ2 K1 u1 bpc1 | fused u8 bpc2 nts", 256, 1, false, NT, 1, 256, 8, true, 2);                             \
    SEQ2("SEQ2 K1 u1 bpc2 | fused u8 bpc2 nts", 256, 1, false, NT, 2, 256, 8, true, 2);                             \
    SEQ2("SEQ2 K1 b512 u2 bpc1 | fused u8 bpc2 nts", 512, 2, false, NT, 1, 256, 8, true, 2);                        \
    SEQ2("SEQ2 K1 b512 u1 bpc1 | fused u8 bpc2 nts", 512, 1, false, NT, 1, 256, 8, true, 2);                        \
    SEQ2("SEQ2 K1 u2 bpc1 | fused b512 u4 bpc1 nts", 256, 2, false, NT, 1, 512, 4, true, 1);                        \
    SEQ2("SEQ2 K1 u2 bpc1 | fused b512 u8 bpc1 nts", 256, 2, false, NT, 1, 512, 8, true, 1);
    if (N <= 5000000) { SWEEP(1) } else { SWEEP(2) }

    // a valid result block for K2 (and initialised partials for finalize)
    V[1].launch(0);
    for (auto& v : V) if (v.name == "finalize (2048 partials)") v.launch(0);
    CHECK(hipDeviceSynchronize());

    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    int frame = 0;
    for (auto& v : V) { v.launch(frame); frame = (frame + 1) % frames; } // warm-up / first-touch
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
    // finalize chain stamps (cycles of s_memtime between numbered points), run alone on an idle GPU
    for (unsigned np : {256u, 1024u})
    {
        unsigned long long best[5] = {~0ull, ~0ull, ~0ull, ~0ull, ~0ull};
        for (int rep = 0; rep < 20; ++rep)
        {
            hipLaunchKernelGGL((finalize_kernel<AosInput, 256>), dim3(1), dim3(256), 0, st, in_of(rep % frames), n, np, L, L, L, P, part, 1ull, d_res, d_hres);
            CHECK(hipDeviceSynchronize());
            unsigned long long h[16];
            CHECK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamps), sizeof(h)));
            for (int k = 0; k < 4; ++k) best[k] = std::min(best[k], h[k + 1] - h[k]);
            best[4] = std::min(best[4], h[4] - h[0]);
        }
        printf("finalize stamps nparts=%u (min over 20, s_memtime ticks): loads %llu | wave tree %llu | lds+serial %llu | scalar math %llu | total %llu\n",
               np, best[0], best[1], best[2], best[3], best[4]);
    }
    printf("%-52s %10s %10s %10s %10s\n", "variant", "med us", "min us", "med GB/s", "max GB/s");
    for (auto& v : V)
    {
        std::sort(v.us.begin(), v.us.end());
        const double med = v.us[v.us.size() / 2], mn = v.us.front();
        printf("%-52s %10.2f %10.2f %10.1f %10.1f\n", v.name.c_str(), med, mn, v.bytes / med * 1e-3, v.bytes / mn * 1e-3);
    }
    return 0;
}
