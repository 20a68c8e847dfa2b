// xcd_exchange.hip -- what does ONE exchange of partial sums among the 32 workgroups of one XCD cost when the
// barrier and the data travel TOGETHER (round 5, the resident solver of one registration)?
//   mode 0  "LL": every double goes out as two 8-byte words (tag32 | half32), agent-scope relaxed atomic stores; a
//           reader polls the words themselves (agent-scope relaxed atomic loads) until the tags match: no separate
//           flag, no fence, architecturally safe (8-byte single-copy atomicity).  Two generations of slots.
//   mode 1  16-byte units (double, tag64) with global_store/load_dwordx4 sc1 -- relies on 16-byte stores being seen whole
//   mode 2  reference: write-through stores + s_waitcnt + counter barrier + agent-scope loads (xcd_barrier.hip mode 1)
// Grid = 256 blocks of 512 threads, only blocks with blockIdx.x % 8 == 0 work (32 blocks, one XCD by the observed
// placement rule) -- the shape of kt_run.  Every round is checked: the sum of all blocks' 9 values must be what the
// round number says.
// Build: hipcc --offload-arch=gfx950 -O3 xcd_exchange.hip -o xcd_exchange
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int GMAX = 256, NV = 9, TPB = 512;
typedef unsigned long long u64;

__device__ __forceinline__ double value_of(int it, int r, int k) { return (double)(it * 7 + r * 3 + k) * 0.125; }

// fixed order whatever the arrival order: 8 partial chains per value (rows c, c + 8, ...), then a fixed tree
__device__ __forceinline__ void block_sum(const double *s_val, double *s_part, double *s_tot, int G, int tid)
{
    if (tid < 8 * NV) {
        const int k = tid % NV, c = tid / NV;
        double a = 0.0;
        for (int q = c; q < G; q += 8) a += s_val[q * NV + k];
        s_part[c * NV + k] = a;
    }
    __syncthreads();
    if (tid < NV)
        s_tot[tid] = ((s_part[tid] + s_part[NV + tid]) + (s_part[2 * NV + tid] + s_part[3 * NV + tid])) +
                     ((s_part[4 * NV + tid] + s_part[5 * NV + tid]) + (s_part[6 * NV + tid] + s_part[7 * NV + tid]));
}

__global__ void __launch_bounds__(TPB) k_exchange(u64 *mail, unsigned *counter, int rounds, int mode, int *errors, int *placement_ok, int G, int stride)
{
    const bool reducer = mode == 3 && blockIdx.x == gridDim.x - 1;
    if (!reducer && (blockIdx.x % stride) != 0u) return;
    const int r = blockIdx.x / stride;
    if (!reducer && r >= G) return;
    const int xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf;
    if (threadIdx.x == 0 && xcc == (int)(blockIdx.x & 7u)) atomicAdd(placement_ok, 1);
    __shared__ double s_val[GMAX * NV];
    __shared__ double s_tot[NV];
    __shared__ double s_part[8 * NV];
    const int tid = threadIdx.x;
    int bad = 0;
    for (int it = 0; it < rounds; ++it) {
        const unsigned seq = (unsigned)it + 1u;
        const int gen = (int)(seq & 1u);
        if (mode == 0) {
            u64 *slot = mail + ((size_t)gen * G) * (2 * NV);
            if (tid < 2 * NV) {
                const double v = value_of(it, r, tid >> 1);
                const u64 bits = (u64)__double_as_longlong(v);
                const unsigned half = (tid & 1) ? (unsigned)(bits >> 32) : (unsigned)bits;
                __hip_atomic_store(&slot[(size_t)r * (2 * NV) + tid], ((u64)seq << 32) | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // thread t polls words t, t + 512, ...: ALL of them requested before any is looked at (one after the other, every word is a
            // round trip of its own: 256 blocks = 9 words per thread took 11 us)
            {
                constexpr int KMAX = (GMAX * 2 * NV + TPB - 1) / TPB;
                const int nw = G * 2 * NV;
                bool all;
                do {
                    u64 w[KMAX];
#pragma unroll
                    for (int k = 0; k < KMAX; ++k) {
                        const int wi = tid + k * TPB;
                        w[k] = wi < nw ? __hip_atomic_load(&slot[wi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((u64)seq << 32);
                    }
                    all = true;
#pragma unroll
                    for (int k = 0; k < KMAX; ++k) all = all && (unsigned)(w[k] >> 32) == seq;
                    if (all) {
#pragma unroll
                        for (int k = 0; k < KMAX; ++k) {
                            const int wi = tid + k * TPB;
                            if (wi < nw) reinterpret_cast<unsigned *>(s_val)[wi] = (unsigned)w[k];
                        }
                    }
                } while (!all);
            }
        } else if (mode == 1) {
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            u32x4 *slot = reinterpret_cast<u32x4 *>(mail) + ((size_t)gen * G) * NV;
            if (tid < NV) {
                const u64 bits = (u64)__double_as_longlong(value_of(it, r, tid));
                u32x4 v; v.x = (unsigned)bits; v.y = (unsigned)(bits >> 32); v.z = seq; v.w = 0u;
                u32x4 *p = &slot[(size_t)r * NV + tid];
                asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
            }
            for (int ui = tid; ui < G * NV; ui += TPB) {
                u32x4 v;
                const u32x4 *p = &slot[ui];
                do {
                    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
                } while (v.z != seq);
                s_val[ui] = __longlong_as_double((long long)(((u64)v.y << 32) | v.x));
            }
        } else if (mode == 3) {
            // two hops: rows in (LL words), the reducer (the LAST block of the grid: it holds no row) polls them all, adds in the fixed order and
            // publishes NV totals as LL words; every block polls the totals
            u64 *slot = mail + ((size_t)gen * (GMAX + 1)) * (2 * NV);
            u64 *totw = slot + (size_t)GMAX * (2 * NV);
            if (!reducer && tid < 2 * NV) {
                const double v = value_of(it, r, tid >> 1);
                const u64 bits = (u64)__double_as_longlong(v);
                const unsigned half = (tid & 1) ? (unsigned)(bits >> 32) : (unsigned)bits;
                __hip_atomic_store(&slot[(size_t)r * (2 * NV) + tid], ((u64)seq << 32) | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (reducer) {
                {
                    constexpr int KMAX = (GMAX * 2 * NV + TPB - 1) / TPB;
                    const int nw = G * 2 * NV;
                    bool all;
                    do {
                        u64 w[KMAX];
#pragma unroll
                        for (int k = 0; k < KMAX; ++k) {
                            const int wi = tid + k * TPB;
                            w[k] = wi < nw ? __hip_atomic_load(&slot[wi], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((u64)seq << 32);
                        }
                        all = true;
#pragma unroll
                        for (int k = 0; k < KMAX; ++k) all = all && (unsigned)(w[k] >> 32) == seq;
                        if (all) {
#pragma unroll
                            for (int k = 0; k < KMAX; ++k) {
                                const int wi = tid + k * TPB;
                                if (wi < nw) reinterpret_cast<unsigned *>(s_val)[wi] = (unsigned)w[k];
                            }
                        }
                    } while (!all);
                }
                __syncthreads();
                block_sum(s_val, s_part, s_tot, G, tid);
                __syncthreads();
                if (tid < 2 * NV) {
                    const u64 bits = (u64)__double_as_longlong(s_tot[tid >> 1]);
                    const unsigned half = (tid & 1) ? (unsigned)(bits >> 32) : (unsigned)bits;
                    __hip_atomic_store(&totw[tid], ((u64)seq << 32) | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
                if (tid < 2 * NV) {
                    u64 w;
                    do { w = __hip_atomic_load(&totw[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((unsigned)(w >> 32) != seq);
                    reinterpret_cast<unsigned *>(s_tot)[tid] = (unsigned)w;
                }
            }
            __syncthreads();
            if (tid < NV) {
                const double want = 0.125 * ((double)G * (double)(it * 7 + tid) + 3.0 * 0.5 * (double)G * (double)(G - 1));
                if (s_tot[tid] != want) ++bad;
            }
            __syncthreads();
            continue;
        } else {
            double *slot = reinterpret_cast<double *>(mail) + ((size_t)gen * G) * 16;
            if (tid < NV) __hip_atomic_store(&slot[(size_t)r * 16 + tid], value_of(it, r, tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tid < 64) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < seq * (unsigned)G) __builtin_amdgcn_s_sleep(1);
            }
            __syncthreads();
            for (int ui = tid; ui < G * NV; ui += TPB) s_val[ui] = __hip_atomic_load(&slot[(size_t)(ui / NV) * 16 + (ui % NV)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        block_sum(s_val, s_part, s_tot, G, tid);
        __syncthreads();
        if (tid < NV) {
            const double want = 0.125 * ((double)G * (double)(it * 7 + tid) + 3.0 * 0.5 * (double)G * (double)(G - 1));
            if (s_tot[tid] != want) ++bad;   // (multiples of 1/8 below 2^40: every order gives the same double)
        }
    }
    if (bad) atomicAdd(errors, bad);
}

int main()
{
    const int rounds = 2000;
    u64 *mail; unsigned *cnt; int *err, *ok;
    const size_t mail_bytes = 2 * (GMAX + 1) * 18 * 8 + 2 * GMAX * 16 * 16;
    hipMalloc(&mail, mail_bytes); hipMalloc(&cnt, 256); hipMalloc(&err, 4); hipMalloc(&ok, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int shapes[][2] = {{32, 8}, {64, 4}, {128, 2}, {256, 1}, {32, 1}, {8, 32}, {16, 1}};   // (blocks, stride): one XCD; 2, 4, 8 XCDs; 32 blocks spread over 8 XCDs; 8 blocks one per XCD...
    for (auto &sh : shapes)
    for (int mode = 0; mode < 4; ++mode) {
        if (mode == 1) continue;
        const int G = sh[0], stride = sh[1];
        float best = 1e9f; int errh = 0, okh = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(mail, 0, mail_bytes); hipMemset(cnt, 0, 256); hipMemset(err, 0, 4); hipMemset(ok, 0, 4);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_exchange, dim3(G * stride + (mode == 3 ? 1 : 0)), dim3(TPB), 0, 0, mail, cnt, rounds, mode, err, ok, G, stride);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
            int e; hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost); errh += e;
            hipMemcpy(&okh, ok, 4, hipMemcpyDeviceToHost);
        }
        printf("%3d blocks, every %d-th of the grid, mode %d (%s): %.3f us per exchange round, %d wrong sums, %d of %d blocks on XCD (block %% 8)\n", G, stride, mode,
               mode == 0 ? "LL 8-byte words" : (mode == 1 ? "16-byte tagged units" : (mode == 2 ? "stores + counter barrier + loads" : "two hops: rows in, one block adds, totals out")),
               best * 1e3 / rounds, errh, okh, G);
    }
    return 0;
}
