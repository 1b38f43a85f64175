// Does (unsigned long long)(uintptr_t)__builtin_amdgcn_dispatch_ptr() give a launch-unique, grid-uniform value -- also across replays of a captured graph?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned long long* out)
{
    if (threadIdx.x == 0)
        out[blockIdx.x] = (unsigned long long)(uintptr_t)__builtin_amdgcn_dispatch_ptr();
}
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main()
{
    unsigned long long *d, h[4];
    CHECK(hipMalloc((void**)&d, 32));
    hipStream_t s1, s2;
    CHECK(hipStreamCreate(&s1)); CHECK(hipStreamCreate(&s2));
    for (int k = 0; k < 3; ++k)
    {
        hipLaunchKernelGGL(probe, dim3(4), dim3(64), 0, s1, d);
        CHECK(hipStreamSynchronize(s1)); CHECK(hipMemcpy(h, d, 32, hipMemcpyDeviceToHost));
        printf("stream 1 launch %d: %llu %llu %llu %llu\n", k, h[0], h[1], h[2], h[3]);
    }
    for (int k = 0; k < 2; ++k)
    {
        hipLaunchKernelGGL(probe, dim3(4), dim3(64), 0, s2, d);
        CHECK(hipStreamSynchronize(s2)); CHECK(hipMemcpy(h, d, 32, hipMemcpyDeviceToHost));
        printf("stream 2 launch %d: %llu %llu %llu %llu\n", k, h[0], h[1], h[2], h[3]);
    }
    hipGraph_t g; hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(s1, hipStreamCaptureModeGlobal));
    hipLaunchKernelGGL(probe, dim3(4), dim3(64), 0, s1, d);
    CHECK(hipStreamEndCapture(s1, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int k = 0; k < 3; ++k)
    {
        CHECK(hipGraphLaunch(ge, s1));
        CHECK(hipStreamSynchronize(s1)); CHECK(hipMemcpy(h, d, 32, hipMemcpyDeviceToHost));
        printf("graph replay %d: %llu %llu %llu %llu\n", k, h[0], h[1], h[2], h[3]);
    }
    return 0;
}
