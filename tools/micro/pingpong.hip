// Round trips of one tagged 8-byte word between two workgroups of one launch (the resident runs' exchange in the small): partner on the
// SAME XCD against a partner on ANOTHER XCD, agent-scope relaxed atomics as cvo_kernels.hip's run_exchange uses.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pingpong tools/micro/pingpong.hip && /tmp/pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf; }

struct Mail {
    unsigned long long ping[2], pong[2];   // [0]: same XCD pair, [1]: cross pair (separate cache lines below)
    unsigned long long pad[12];
};

__global__ void __launch_bounds__(64) k_pp(unsigned *xcc, unsigned *arrived, unsigned long long *words /* 4 x 32 */, long long *out, int rounds,
                                             int scope_sys)
{
    extern __shared__ char big[];   // (a block per compute unit)
    if (threadIdx.x == 0) {
        big[0] = 1;
        xcc[blockIdx.x] = xcc_id();
        __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(arrived, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) __builtin_amdgcn_s_sleep(4);
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    const unsigned mine = xcc[0];
    int same = -1, other = -1;
    for (unsigned b = 1; b < gridDim.x; ++b) {
        const unsigned x = __hip_atomic_load(&xcc[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (x == mine && same < 0) same = (int)b;
        if (x != mine && other < 0) other = (int)b;
    }
    for (int which = 0; which < 2; ++which) {
        const int partner = which == 0 ? same : other;
        unsigned long long *ping = words + (which * 2) * 32, *pong = words + (which * 2 + 1) * 32;
        if (partner < 0) continue;
        if (blockIdx.x == 0) {
            const long long t0 = (long long)wall_clock64();
            for (int r = 1; r <= rounds; ++r) {
                __hip_atomic_store(ping, (unsigned long long)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(pong, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned long long)r) {}
            }
            out[which] = (long long)wall_clock64() - t0;
            out[2 + which] = partner;
            out[4 + which] = xcc[partner];
        } else if ((int)blockIdx.x == partner) {
            for (int r = 1; r <= rounds; ++r) {
                while (__hip_atomic_load(ping, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned long long)r) {}
                __hip_atomic_store(pong, (unsigned long long)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

int main()
{
    unsigned *xcc, *arrived; unsigned long long *words; long long *out;
    const int blocks = 256, rounds = 2000;
    hipMalloc(&xcc, blocks * 4); hipMalloc(&arrived, 4); hipMalloc(&words, 4 * 32 * 8); hipMalloc(&out, 8 * 8);
    hipFuncSetAttribute((const void *)k_pp, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(arrived, 0, 4); hipMemset(words, 0, 4 * 32 * 8); hipMemset(out, 0, 64);
        hipLaunchKernelGGL(k_pp, dim3(blocks), dim3(64), 100 * 1024, 0, xcc, arrived, words, out, rounds, 0);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
        long long h[8]; hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
        std::vector<unsigned> hx(blocks); hipMemcpy(hx.data(), xcc, blocks * 4, hipMemcpyDeviceToHost);
        printf("block 0 on XCD %u; same-XCD partner block %lld (XCD %lld): %.3f us per round trip; other-XCD partner block %lld (XCD %lld): %.3f us per round trip\n",
               hx[0], h[2], h[4], h[0] * 0.01 / rounds, h[3], h[5], h[1] * 0.01 / rounds);
        if (rep == 0) { printf("XCD of blocks 0..31:"); for (int b = 0; b < 32; ++b) printf(" %u", hx[b]); printf("\n"); }
    }
    return 0;
}
