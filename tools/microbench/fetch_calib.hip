// fetch_calib.hip -- what does rocprofv3's FETCH_SIZE count on THIS path's access pattern, a 16-byte-per-lane GATHER
// (MI355X_MICROARCH.md "HBM": calibrated for wide coalesced streams only; Infinity-Cache hits appear to be counted)?
// Kernels, each 2^24 lanes, every lane one float4 load:
//   k_gather<footprint>  lane l reads row hash(l, pass) of a table of `rows` float4 -- rows chosen so that the table is
//                        64 MiB (fits the 256 MiB Infinity Cache with room: warmed by earlier launches) or 2 GiB (does not)
//   k_stream             lane l reads row l of the 2 GiB table + pass offset (the calibrated case: a coalesced stream)
// Known per launch: 2^24 lanes x 16 B = 256 MiB requested; a random gather touches ~2^24 distinct 64-B sectors
// (1 GiB of sectors) / 128-B lines (2 GiB of lines); the 64 MiB table is touched ~4 times over per launch.
// Run under rocprofv3 --pmc FETCH_SIZE (and TCC_EA0_RDREQ_sum, TCC_HIT_sum TCC_MISS_sum in their own passes) --kernel-trace:
//   tools/gpu_fetch_calib.sh
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ unsigned mix(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__global__ void k_gather_64MiB(const float4 *t, unsigned mask, unsigned pass, float *sink)
{
    const unsigned g = blockIdx.x * blockDim.x + threadIdx.x;
    const float4 v = t[mix(g * 2654435761u + pass) & mask];
    if (v.x == 12345.f) sink[0] = v.y;
}
__global__ void k_gather_2GiB(const float4 *t, unsigned mask, unsigned pass, float *sink)
{
    const unsigned g = blockIdx.x * blockDim.x + threadIdx.x;
    const float4 v = t[mix(g * 2654435761u + pass) & mask];
    if (v.x == 12345.f) sink[0] = v.y;
}
__global__ void k_stream_2GiB(const float4 *t, unsigned mask, unsigned pass, float *sink)
{
    const unsigned g = blockIdx.x * blockDim.x + threadIdx.x;
    const float4 v = t[(g + pass * (1u << 24)) & mask];
    if (v.x == 12345.f) sink[0] = v.y;
}
int main()
{
    const size_t big = 2ull << 30, small = 64ull << 20;
    float4 *t; float *sink;
    if (hipMalloc(&t, big) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(t, 0, big);
    const unsigned lanes = 1u << 24, tpb = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int kind = 0; kind < 3; ++kind) {
        const unsigned mask = (unsigned)((kind == 0 ? small : big) / 16 - 1);
        float ms_tot = 0;
        for (unsigned pass = 0; pass < 6; ++pass) {
            hipEventRecord(e0);
            if (kind == 0) hipLaunchKernelGGL(k_gather_64MiB, dim3(lanes / tpb), dim3(tpb), 0, 0, t, mask, pass, sink);
            else if (kind == 1) hipLaunchKernelGGL(k_gather_2GiB, dim3(lanes / tpb), dim3(tpb), 0, 0, t, mask, pass, sink);
            else hipLaunchKernelGGL(k_stream_2GiB, dim3(lanes / tpb), dim3(tpb), 0, 0, t, mask, pass, sink);
            hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1); if (pass) ms_tot += ms;
        }
        printf("%s: %.1f us per launch, requested 256 MiB per launch -> %.0f GB/s of requested bytes\n",
               kind == 0 ? "k_gather_64MiB" : (kind == 1 ? "k_gather_2GiB" : "k_stream_2GiB"), ms_tot / 5 * 1e3, 268.435456 / (ms_tot / 5));
    }
    return 0;
}
