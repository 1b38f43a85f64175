// pingpong.hip -- developer probe: latency of an 8-byte sc1 (agent-scope relaxed atomic) hand-off between two workgroups.
// Block A stores {k}, block B spins until it reads k and answers with {k}; one iteration = two hops.
//   ./pingpong [partner_block=1] [iters=2000] [grid=16] [sleep=0]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void pingpong(unsigned long long* a, unsigned long long* b, int partner, int iters, int sleep, unsigned long long* out)
{
    if (threadIdx.x != 0) return;
    const int me = blockIdx.x;
    if (me != 0 && me != partner) return;
    const unsigned long long t0 = wall_clock64();
    for (int k = 1; k <= iters; ++k)
    {
        if (me == 0)
        {
            __hip_atomic_store(a, (unsigned long long)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned long long)k)
                if (sleep) __builtin_amdgcn_s_sleep(1);
        }
        else
        {
            while (__hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned long long)k)
                if (sleep) __builtin_amdgcn_s_sleep(1);
            __hip_atomic_store(b, (unsigned long long)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (me == 0) out[0] = wall_clock64() - t0;
}
// one-to-many: block 0 stores k; every other block (one lane) spins until it sees k and then adds 1 to a counter; block 0
// waits for the counter to reach (grid-1)*k.  Measures broadcast + fan-in.
__global__ void bcast(unsigned long long* a, unsigned* cnt, int iters, unsigned long long* out)
{
    if (threadIdx.x != 0) return;
    const unsigned G = gridDim.x;
    const unsigned long long t0 = wall_clock64();
    for (int k = 1; k <= iters; ++k)
    {
        if (blockIdx.x == 0)
        {
            __hip_atomic_store(a, (unsigned long long)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (G - 1) * (unsigned)k) {}
        }
        else
        {
            while (__hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned long long)k) {}
            __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (blockIdx.x == 0) out[0] = wall_clock64() - t0;
}
int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    unsigned long long *a, *b, *out; unsigned* cnt;
    hipMalloc((void**)&a, 4096); hipMalloc((void**)&b, 4096); hipMalloc((void**)&out, 64); hipMalloc((void**)&cnt, 4096);
    for (int partner : {1, 2, 7, 8, 16, 100})
        for (int sleep : {0, 1})
        {
            hipMemset(a, 0, 4096); hipMemset(b, 0, 4096);
            hipLaunchKernelGGL(pingpong, dim3(128), dim3(64), 0, 0, a, b + 64, partner, iters, sleep, out);
            unsigned long long t; hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost);
            printf("pingpong block 0 <-> %3d sleep %d: %.3f us per hop\n", partner, sleep, t * 0.01 / iters / 2);
        }
    for (int grid : {2, 9, 33, 65, 129, 256})
    {
        hipMemset(a, 0, 4096); hipMemset(cnt, 0, 4096);
        hipLaunchKernelGGL(bcast, dim3(grid), dim3(64), 0, 0, a, cnt, iters / 4, out);
        unsigned long long t; hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost);
        printf("broadcast to %3d blocks + counter fan-in: %.3f us per round\n", grid - 1, t * 0.01 / (iters / 4));
    }
    return 0;
}
