// xcd_barrier.hip -- go / no-go for an XCD-resident solver: what does a barrier among the
// 32 workgroups of ONE XCD cost on MI355X, with and without the exchange of partial sums a registration's
// iteration needs (9 doubles per block published, all blocks' sums read back by every block)?
//   mode 0  barrier only: per-group monotonic counter, relaxed agent-scope add + relaxed agent-scope poll
//   mode 1  + every block publishes 9 doubles write-through (agent-scope 8-byte stores) before it arrives and
//             reads all 32 x 9 back with agent-scope loads after the barrier (the reduction of an iteration)
//   mode 2  mode 1 with an agent-scope acquire fence after the barrier (what plain loads of other blocks' bulk
//             data -- a kept list, a transformed cloud -- would need)
// 8 groups run at once, group g = blocks with blockIdx.x % 8 == g (observed placement: block b runs on XCD b % 8;
// the kernel reads XCC_ID and reports how many blocks sit where the rule says).
// Build: hipcc --offload-arch=gfx950 -O3 xcd_barrier.hip -o xcd_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int GROUPS = 8;

__global__ void __launch_bounds__(256) k_xcd_barrier(unsigned *counters, double *slab, int per_group, int rounds, int mode,
                                                     double *sink, int *placement_ok)
{
    const int g = blockIdx.x % GROUPS, r = blockIdx.x / GROUPS;   // rank within the group
    const int xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf;   // XCC_ID
    if (threadIdx.x == 0 && xcc == g) atomicAdd(placement_ok, 1);
    unsigned *cnt = counters + g * 32;   // (a cache line of its own)
    double *mine = slab + ((size_t)g * per_group + r) * 16;
    double acc = 0.0;
    __shared__ double s_sum;
    for (int it = 0; it < rounds; ++it) {
        if (mode >= 1) {
            if (threadIdx.x < 9) __hip_atomic_store(&mine[threadIdx.x], (double)(it + r + threadIdx.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (threadIdx.x < 64) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(it + 1) * (unsigned)per_group;
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
            if (mode == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        if (mode >= 1) {
            // every block reduces all blocks' sums (the same order everywhere): thread t < per_group takes block t
            double v = 0.0;
            if ((int)threadIdx.x < per_group) {
                const double *o = slab + ((size_t)g * per_group + threadIdx.x) * 16;
                for (int k = 0; k < 9; ++k) v += __hip_atomic_load(&o[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            if (threadIdx.x == 0) s_sum = v;
            __syncthreads();
            acc += s_sum;
        }
    }
    if (acc == 12345.678 && threadIdx.x == 0) sink[0] = acc;
}

__global__ void k_empty(double *sink) { if (threadIdx.x == 9999) sink[0] = 1.0; }

int main()
{
    const int rounds = 400;
    unsigned *c; double *slab, *sink; int *ok;
    hipMalloc(&c, GROUPS * 32 * sizeof(unsigned)); hipMalloc(&slab, GROUPS * 64 * 16 * sizeof(double)); hipMalloc(&sink, 8); hipMalloc(&ok, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int per_group : {8, 16, 32, 64})
        for (int mode = 0; mode < 3; ++mode) {
            float best = 1e9f; int okh = 0;
            for (int rep = 0; rep < 3; ++rep) {
                hipMemset(c, 0, GROUPS * 32 * sizeof(unsigned)); hipMemset(ok, 0, 4);
                hipDeviceSynchronize();
                hipEventRecord(e0);
                hipLaunchKernelGGL(k_xcd_barrier, dim3(GROUPS * per_group), dim3(256), 0, 0, c, slab, per_group, rounds, mode, sink, ok);
                hipEventRecord(e1);
                hipDeviceSynchronize();
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
                hipMemcpy(&okh, ok, 4, hipMemcpyDeviceToHost);
            }
            printf("blocks per XCD group %2d (x8 groups), mode %d: %.2f us per barrier round; %d of %d blocks on XCD (block %% 8)\n",
                   per_group, mode, best * 1e3 / rounds, okh, GROUPS * per_group);
        }
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, 0, sink);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("empty kernel, 256 blocks: %.2f us per launch (stream-ordered)\n", ms * 1e3 / rounds);
    return 0;
}
