// grid_barrier.hip -- cost of a software grid barrier on MI355X (is a persistent
// whole-iteration kernel worth it compared with 5 kernel boundaries per iteration?)
// Build: hipcc --offload-arch=gfx950 -O3 grid_barrier.hip -o grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ void grid_sync(unsigned *counter, unsigned target)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        atomicAdd(counter, 1u);
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target)
            __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

__global__ void k_barriers(unsigned *counter, int rounds, float *sink)
{
    float x = threadIdx.x;
    for (int r = 0; r < rounds; ++r) {
        x = x * 1.0001f + 1.0f;
        grid_sync(counter, (unsigned)(r + 1) * gridDim.x);
    }
    if (x == 12345.f) sink[0] = x;
}

__global__ void k_empty(float *sink) { if (threadIdx.x == 9999) sink[0] = 1.f; }

int main(int argc, char **argv)
{
    const int rounds = 200;
    unsigned *c; float *sink;
    hipMalloc(&c, 4); hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {64, 256, 512, 1024, 2048}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(c, 0, 4);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_barriers, dim3(blocks), dim3(256), 0, 0, c, rounds, sink);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("blocks %4d: %.2f us per grid barrier\n", blocks, ms * 1e3 / rounds);
        }
    }
    // for comparison: back-to-back dependent kernel launches on one stream
    for (int blocks : {1, 1024}) {
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(256), 0, 0, sink);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("empty kernel, %4d blocks: %.2f us per launch (stream-ordered)\n", blocks, ms * 1e3 / rounds);
    }
    return 0;
}
