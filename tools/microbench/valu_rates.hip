// valu_rates.hip -- per-instruction issue cost of the ops the sweep kernel is
// made of, measured on gfx950 with s_memtime inside the kernel.
// Build: hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void __launch_bounds__(256) bench(long long *cyc, float *sink, int iters)
{
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f;
    float r0 = a, r1 = a + 1, r2 = a + 2, r3 = a + 3, r4 = a + 4, r5 = a + 5, r6 = a + 6, r7 = a + 7;
    f32x2 p0 = {a, b}, p1 = {b, c}, p2 = {c, a}, p3 = {a, a}, q = {b, c};
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0};
    __shared__ float4 lds[256];
    lds[threadIdx.x] = make_float4(a, b, c, a);
    __syncthreads();
    float sb = __builtin_amdgcn_readfirstlane(__float_as_int(b)) * 1e-30f + 1.0f;   // uniform -> SGPR
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {   // v_fma_f32, 8 independent chains
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                              : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(b), "v"(c));)
        } else if (KIND == 1) {   // v_sub_f32
            REP8(asm volatile("v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %8\n v_sub_f32 %2, %2, %8\n v_sub_f32 %3, %3, %8\n"
                              "v_sub_f32 %4, %4, %8\n v_sub_f32 %5, %5, %8\n v_sub_f32 %6, %6, %8\n v_sub_f32 %7, %7, %8\n"
                              : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(b));)
        } else if (KIND == 2) {   // v_mul_f32
            REP8(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                              "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                              : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(b));)
        } else if (KIND == 3) {   // v_cmp_lt_f32 -> vcc
            REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n"
                              "v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8\n"
                              : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(b) : "vcc");)
        } else if (KIND == 4) {   // v_pk_fma_f32
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                              "v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));)
        } else if (KIND == 5) {   // v_pk_add_f32
            REP8(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                              "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));)
        } else if (KIND == 6) {   // v_pk_mul_f32
            REP8(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                              "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));)
        } else if (KIND == 7) {   // v_exp_f32
            REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                              "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                              : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7));)
        } else if (KIND == 8) {   // v_mfma_f32_16x16x4_f32, 4 accumulators
            REP8(acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
                 acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc1, 0, 0, 0);
                 acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc2, 0, 0, 0);
                 acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc3, 0, 0, 0);
                 acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
                 acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc1, 0, 0, 0);
                 acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc2, 0, 0, 0);
                 acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc3, 0, 0, 0);)
        } else if (KIND == 9) {   // the pair test: 3 sub (sgpr operand), mul, 2 fma, cmp
            REP8(asm volatile("v_sub_f32 %0, %8, %4\n v_sub_f32 %1, %8, %5\n v_sub_f32 %2, %8, %6\n v_mul_f32 %3, %0, %0\n"
                              "v_fma_f32 %3, %1, %1, %3\n v_fma_f32 %3, %2, %2, %3\n v_cmp_lt_f32 vcc, %3, %7\n"
                              : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(r4), "v"(r5), "v"(r6), "v"(r7), "s"(sb) : "vcc");)
        } else if (KIND == 10) {   // ds_read_b128 broadcast (uniform address)
            asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:16\n ds_read_b128 %2, %4 offset:32\n ds_read_b128 %3, %4 offset:48\n"
                         "ds_read_b128 %0, %4 offset:64\n ds_read_b128 %1, %4 offset:80\n ds_read_b128 %2, %4 offset:96\n ds_read_b128 %3, %4 offset:112\n s_waitcnt lgkmcnt(0)\n"
                         : "=&v"(acc0), "=&v"(acc1), "=&v"(acc2), "=&v"(acc3) : "v"(0) : "memory");
        } else if (KIND == 11) {   // v_fma_f64
            double d0 = r0, d1 = r1, d2 = r2, d3 = r3, db = 1.0000001, dc = 0.25;
            REP8(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                              "v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(db), "v"(dc));)
            r0 = (float)(d0 + d1 + d2 + d3);
        } else if (KIND == 12) {   // v_pk pair test: 2 rows per lane: 3 pk_add(neg), pk_mul, 2 pk_fma + 2 cmp
            REP8(asm volatile("v_pk_add_f32 %0, %4, %5 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %1, %4, %5 neg_lo:[0,1] neg_hi:[0,1]\n"
                              "v_pk_add_f32 %2, %4, %5 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_mul_f32 %3, %0, %0\n"
                              "v_pk_fma_f32 %3, %1, %1, %3\n v_pk_fma_f32 %3, %2, %2, %3\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q), "v"(q));)
        } else if (KIND == 13) {   // v_add_f64
            double d0 = r0, d1 = r1, d2 = r2, d3 = r3, db = 1.0000001;
            REP8(asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n"
                              "v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(db));)
            r0 = (float)(d0 + d1 + d2 + d3);
        } else if (KIND == 14) {   // v_mul_f64
            double d0 = r0, d1 = r1, d2 = r2, d3 = r3, db = 1.0000001;
            REP8(asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n"
                              "v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(db));)
            r0 = (float)(d0 + d1 + d2 + d3);
        } else if (KIND == 15) {   // v_cvt_f64_f32
            double d0 = 0, d1 = 0, d2 = 0, d3 = 0;
            REP8(asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7\n"
                              "v_cvt_f64_f32 %0, %5\n v_cvt_f64_f32 %1, %6\n v_cvt_f64_f32 %2, %7\n v_cvt_f64_f32 %3, %4\n"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(r4), "v"(r5), "v"(r6), "v"(r7));)
            r0 = (float)(d0 + d1 + d2 + d3);
        } else if (KIND == 16) {   // v_cvt_f32_f64
            double d0 = r0, d1 = r1, d2 = r2, d3 = r3;
            REP8(asm volatile("v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7\n"
                              "v_cvt_f32_f64 %0, %5\n v_cvt_f32_f64 %1, %6\n v_cvt_f32_f64 %2, %7\n v_cvt_f32_f64 %3, %4\n"
                              : "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(d0), "v"(d1), "v"(d2), "v"(d3));)
        } else if (KIND == 17) {   // 4 v_fma_f32 + 4 v_fma_f64 interleaved (do the two kinds share issue time?)
            double d0 = r0, d1 = r1, d2 = r2, d3 = r3, db = 1.0000001, dc = 0.25;
            REP8(asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f32 %4, %4, %10, %11\n v_fma_f64 %1, %1, %8, %9\n v_fma_f32 %5, %5, %10, %11\n"
                              "v_fma_f64 %2, %2, %8, %9\n v_fma_f32 %6, %6, %10, %11\n v_fma_f64 %3, %3, %8, %9\n v_fma_f32 %7, %7, %10, %11\n"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(db), "v"(dc), "v"(b), "v"(c));)
            r0 = (float)(d0 + d1 + d2 + d3);
        } else if (KIND == 18) {   // v_ldexp_f64 / v_rndne_f64 / v_cvt_i32_f64 mix of the exp (3 of each kind... 8 per group)
            double d0 = r0, d1 = r1, d2 = r2, d3 = r3; int e0 = 1, e1 = 2;
            REP8(asm volatile("v_rndne_f64 %0, %0\n v_rndne_f64 %1, %1\n v_ldexp_f64 %2, %2, %6\n v_ldexp_f64 %3, %3, %6\n"
                              "v_cvt_i32_f64 %4, %0\n v_cvt_i32_f64 %5, %1\n v_ldexp_f64 %2, %2, %6\n v_rndne_f64 %3, %3\n"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(e0), "+v"(e1) : "v"(0));)
            r0 = (float)(d0 + d1 + d2 + d3) + e0 + e1;
        } else if (KIND == 19) {   // v_mbcnt_lo + v_mbcnt_hi + v_lshlrev + v_and (integer ops of the list bookkeeping)
            int e0 = threadIdx.x, e1 = 2, e2 = 3, e3 = 4;
            REP8(asm volatile("v_mbcnt_lo_u32_b32 %0, %4, %0\n v_mbcnt_hi_u32_b32 %1, %4, %1\n v_lshlrev_b32 %2, 3, %2\n v_and_b32 %3, 63, %3\n"
                              "v_mbcnt_lo_u32_b32 %0, %4, %0\n v_mbcnt_hi_u32_b32 %1, %4, %1\n v_lshl_add_u32 %2, %2, 3, %3\n v_and_or_b32 %3, %3, %4, %0\n"
                              : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3) : "v"(0x55aa55aa));)
            r0 = (float)(e0 + e1 + e2 + e3);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    float s = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + p0.x + p1.y + p2.x + p3.y + acc0.x + acc1.y + acc2.z + acc3.w;
    if (s == 12345.678f) sink[0] = s;
}

template <int KIND>
void run(const char *name, int inst_per_iter, int blocks_per_cu, int threads)
{
    const int iters = 2000;
    const int nblk = 256 * blocks_per_cu;
    long long *cyc; float *sink;
    hipMalloc(&cyc, nblk * sizeof(long long));
    hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    bench<KIND><<<nblk, threads>>>(cyc, sink, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    bench<KIND><<<nblk, threads>>>(cyc, sink, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(nblk);
    hipMemcpy(h.data(), cyc, nblk * sizeof(long long), hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= nblk;
    const int waves_per_simd = blocks_per_cu * (threads / 64) / 4 > 0 ? blocks_per_cu * (threads / 64) / 4 : 1;
    const double wave_insts = (double)iters * inst_per_iter;
    // memtime ticks at 100 MHz on gfx9 (s_memrealtime) or shader clock (s_memtime): report both views
    const double total_insts = wave_insts * nblk * (threads / 64);
    const double simd_cycles_per_inst_wall = (ms * 1e-3 * 2.4e9) * 1024.0 / total_insts;
    printf("%-34s waves/SIMD %d  wall %.3f ms  ticks/wave %.0f  ticks/inst/wave %.3f  SIMD-cyc/inst@2.4GHz %.3f  Ginst/s %.1f\n",
           name, waves_per_simd, ms, avg, avg / wave_insts, simd_cycles_per_inst_wall, total_insts / ms / 1e6);
    hipFree(cyc); hipFree(sink);
}

int main()
{
    for (int bpc : {4, 8}) {
        printf("--- %d block(s) of 256 threads per CU ---\n", bpc);
        run<0>("v_fma_f32", 64, bpc, 256);
        run<1>("v_sub_f32", 64, bpc, 256);
        run<2>("v_mul_f32", 64, bpc, 256);
        run<3>("v_cmp_lt_f32 vcc", 64, bpc, 256);
        run<4>("v_pk_fma_f32", 64, bpc, 256);
        run<5>("v_pk_add_f32", 64, bpc, 256);
        run<6>("v_pk_mul_f32", 64, bpc, 256);
        run<7>("v_exp_f32", 64, bpc, 256);
        run<8>("v_mfma_f32_16x16x4_f32", 64, bpc, 256);
        run<9>("pair test (7 inst, sgpr y)", 56, bpc, 256);
        run<10>("ds_read_b128 broadcast x8 + wait", 8, bpc, 256);
        run<11>("v_fma_f64", 64, bpc, 256);
        run<12>("pk pair test (6 pk inst / 2 pairs)", 48, bpc, 256);
        run<13>("v_add_f64", 64, bpc, 256);
        run<14>("v_mul_f64", 64, bpc, 256);
        run<15>("v_cvt_f64_f32", 64, bpc, 256);
        run<16>("v_cvt_f32_f64", 64, bpc, 256);
        run<17>("4 v_fma_f64 + 4 v_fma_f32 interleaved", 64, bpc, 256);
        run<18>("rndne/ldexp/cvt_i32 f64 mix", 64, bpc, 256);
        run<19>("mbcnt/lshl/and integer mix", 64, bpc, 256);
    }
    return 0;
}
