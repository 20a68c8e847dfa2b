// cvo_kernels.hip -- gfx950 (MI355X, CDNA4) kernels for the CVO inner loop.
//
//   k_transform : transform_pcd            (ref src/cvo.cpp:310-315)
//   k_sweep     : se_kernel fused with the consumer of A:
//                   SWEEP_FLOW  compute_flow          (ref src/cvo.cpp:99-210)
//                   SWEEP_STEP  compute_step_size      (ref src/cvo.cpp:249-289)
//                   SWEEP_SELF  acvo Axx / Ayy terms   (ref src/adaptive_cvo.cpp:156-265)
//   k_taylor    : per-source-point Taylor vectors      (ref src/cvo.cpp:226-238)
//   k_finalize  : fixed-order float64 reduction of the per-block partials
//
// The Gram matrix A is never materialised: every sweep re-tests all
// target x source pairs (dense, wave64: one target row per lane and
// ROWS_PER_LANE rows per thread, the source chunk broadcast from LDS), and
// the rare survivors of the distance test are compacted through a per-wave
// LDS queue so that the expensive part (two float64 exponentials, the colour
// distance, the accumulations) always runs on full wavefronts.
//
// Arithmetic contract: compiled with -ffp-contract=off; every FMA
// below is an explicit __builtin_fmaf.  Per-pair terms are float32 in the
// reference's operation order, accumulated in float64.
#include "cvo_device.h"

namespace cvo_dev {

constexpr int ROWS_PER_LANE = 4;

int rows_per_tile() { return BLOCK * ROWS_PER_LANE; }

__device__ __forceinline__ float4 nan4()
{
    const float q = __builtin_nanf("");
    return make_float4(q, q, q, q);
}

// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(BLOCK) k_transform(const TransformArgs a)
{
    const int j = blockIdx.x * BLOCK + threadIdx.x;
    if (j >= a.n) return;
    const float4 p = a.src[j];
    float4 o;
    // Eigen: transform.linear()*p + translation, coefficient order, no FMA
    o.x = ((a.Rt[0] * p.x + a.Rt[1] * p.y) + a.Rt[2] * p.z) + a.t[0];
    o.y = ((a.Rt[3] * p.x + a.Rt[4] * p.y) + a.Rt[5] * p.z) + a.t[1];
    o.z = ((a.Rt[6] * p.x + a.Rt[7] * p.y) + a.Rt[8] * p.z) + a.t[2];
    o.w = 0.0f;
    a.dst[j] = o;
}

void launch_transform(const TransformArgs &a, hipStream_t s)
{
    if (a.n <= 0) return;
    hipLaunchKernelGGL(k_transform, dim3((a.n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, a);
}

// ---------------------------------------------------------------------------
// taylor row layout (16 floats):
//   [0..2] xiz  [3..5] xi2z  [6..8] xi3z  [9..11] xi4z
//   [12] normxiz2  [13] xiz_dot_xi2z  [14] epsil_const  [15] 0
__device__ __forceinline__ float mv_row(const float *m, float x, float y, float z)
{
    return (m[0] * x + m[1] * y) + m[2] * z;
}

__global__ void __launch_bounds__(BLOCK) k_taylor(const TaylorArgs a)
{
    const int j = blockIdx.x * BLOCK + threadIdx.x;
    if (j >= a.n) return;
    const float4 p = a.pos[j];
    float xiz[3], xi2z[3], xi3z[3], xi4z[3];
    // omega x y + v
    xiz[0] = (a.omega[1] * p.z - a.omega[2] * p.y) + a.v[0];
    xiz[1] = (a.omega[2] * p.x - a.omega[0] * p.z) + a.v[1];
    xiz[2] = (a.omega[0] * p.y - a.omega[1] * p.x) + a.v[2];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        xi2z[r] = mv_row(a.W2 + 3 * r, p.x, p.y, p.z) + a.u2[r];
        xi3z[r] = mv_row(a.W3 + 3 * r, p.x, p.y, p.z) + a.u3[r];
        xi4z[r] = mv_row(a.W4 + 3 * r, p.x, p.y, p.z) + a.u4[r];
    }
    const float normxiz2 = (xiz[0] * xiz[0] + xiz[1] * xiz[1]) + xiz[2] * xiz[2];
    const float xz12 = -((xiz[0] * xi2z[0] + xiz[1] * xi2z[1]) + xiz[2] * xi2z[2]);
    const float eps_c = ((xi2z[0] * xi2z[0] + xi2z[1] * xi2z[1]) + xi2z[2] * xi2z[2]) +
                        2 * ((xiz[0] * xi3z[0] + xiz[1] * xi3z[1]) + xiz[2] * xi3z[2]);
    float4 *out = reinterpret_cast<float4 *>(a.taylor + (size_t)j * TAYLOR_STRIDE);
    out[0] = make_float4(xiz[0], xiz[1], xiz[2], xi2z[0]);
    out[1] = make_float4(xi2z[1], xi2z[2], xi3z[0], xi3z[1]);
    out[2] = make_float4(xi3z[2], xi4z[0], xi4z[1], xi4z[2]);
    out[3] = make_float4(normxiz2, xz12, eps_c, 0.0f);
}

void launch_taylor(const TaylorArgs &a, hipStream_t s)
{
    if (a.n <= 0) return;
    hipLaunchKernelGGL(k_taylor, dim3((a.n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, a);
}

// ---------------------------------------------------------------------------
// pair weight for a pair that passed d2 < tau; 0 if dropped
__device__ __forceinline__ float d2_feat(const float4 fa0, const float fa4, const float4 fb0,
                                         const float fb4)
{
    const float e0 = fa0.x - fb0.x, e1 = fa0.y - fb0.y, e2 = fa0.z - fb0.z, e3 = fa0.w - fb0.w,
                e4 = fa4 - fb4;
    float r = e0 * e0;
    r = __builtin_fmaf(e1, e1, r);
    r = __builtin_fmaf(e2, e2, r);
    r = __builtin_fmaf(e3, e3, r);
    r = __builtin_fmaf(e4, e4, r);
    return r;
}

__device__ __forceinline__ float pair_weight(const KernConsts &kc, float d2, const float *feat_a,
                                             int i, const float *feat_b, int j)
{
    const float4 fa0 = *reinterpret_cast<const float4 *>(feat_a + (size_t)i * FEAT_STRIDE);
    const float fa4 = feat_a[(size_t)i * FEAT_STRIDE + 4];
    const float4 fb0 = *reinterpret_cast<const float4 *>(feat_b + (size_t)j * FEAT_STRIDE);
    const float fb4 = feat_b[(size_t)j * FEAT_STRIDE + 4];
    const float d2c = d2_feat(fa0, fa4, fb0, fb4);
    if (!(d2c < kc.tau_c)) return 0.0f;
    const float k = (float)(kc.s2_d * exp((double)d2 * kc.ninv_2l2));
    const float ck = (float)(kc.cs2_d * exp((double)d2c * kc.ninv_2cl2));
    const float a = ck * k;
    return a > kc.sp ? a : 0.0f;
}

template <int MODE> struct NAcc;
template <> struct NAcc<SWEEP_FLOW> { static constexpr int n = NACC_FLOW; };
template <> struct NAcc<SWEEP_STEP> { static constexpr int n = NACC_STEP; };
template <> struct NAcc<SWEEP_SELF> { static constexpr int n = NACC_SELF; };

// One compacted candidate: full evaluation of the pair and accumulation.
template <int MODE>
__device__ __forceinline__ void process_pair(const SweepArgs &a, int i, int j, const float4 xi,
                                             const float4 yj, double *acc)
{
    const float e0 = xi.x - yj.x, e1 = xi.y - yj.y, e2 = xi.z - yj.z;
    const float d2 = __builtin_fmaf(e2, e2, __builtin_fmaf(e1, e1, e0 * e0));
    if (!(d2 < a.kc.tau)) return;   // also rejects NaN padding
    const float w = pair_weight(a.kc, d2, a.feat_a, i, a.feat_b, j);
    if (!(w > 0.0f)) return;
    if (MODE == SWEEP_FLOW) {
        // cross(x_i, y_j), y_j - x_i ; (1/c * A_ij) * cross  (ref cvo.cpp:191-198)
        const float c0 = xi.y * yj.z - xi.z * yj.y;
        const float c1 = xi.z * yj.x - xi.x * yj.z;
        const float c2 = xi.x * yj.y - xi.y * yj.x;
        const float f0 = yj.x - xi.x, f1 = yj.y - xi.y, f2 = yj.z - xi.z;
        const float ac = a.kc.inv_c * w, ad = a.kc.inv_d * w;
        acc[0] += (double)(ac * c0);
        acc[1] += (double)(ac * c1);
        acc[2] += (double)(ac * c2);
        acc[3] += (double)(ad * f0);
        acc[4] += (double)(ad * f1);
        acc[5] += (double)(ad * f2);
        acc[6] += (double)w;
        acc[7] += (double)((a.kc.inv_l3 * w) * d2);
        acc[8] += 1.0;
    } else if (MODE == SWEEP_STEP) {
        const float4 *t = reinterpret_cast<const float4 *>(a.taylor + (size_t)j * TAYLOR_STRIDE);
        const float4 t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3];
        // diff_xy = x_i - y_j is e0,e1,e2
        const float cb = a.kc.cb, cg = a.kc.cg, cd = a.kc.cd;
        const float beta = ((cb * t0.x) * e0 + (cb * t0.y) * e1) + (cb * t0.z) * e2;
        const float g_dot = ((2.0f * t0.w) * e0 + (2.0f * t1.x) * e1) + (2.0f * t1.y) * e2;
        const float gamma = cg * (t3.x + g_dot);
        const float d_dot = ((-t1.z) * e0 + (-t1.w) * e1) + (-t2.x) * e2;
        const float delta = cd * (t3.y + d_dot);
        const float e_dot = ((2.0f * t2.y) * e0 + (2.0f * t2.z) * e1) + (2.0f * t2.w) * e2;
        const float epsil = cg * (t3.z + e_dot);
        const double A = (double)w;
        const double b = (double)beta, g = (double)gamma;
        acc[0] += (double)(w * beta);
        acc[1] += A * (g + (double)(beta * beta) / 2.0);
        acc[2] += A * ((double)(delta + beta * gamma) + (double)(beta * beta * beta) / 6.0);
        acc[3] += A * ((((double)(epsil + beta * delta) + 0.5 * b * b * g) + 0.5 * g * g) +
                       1 / 24.0 * b * b * b * b);
    } else {
        if (i >= a.first_counted) acc[0] += (double)((a.kc.inv_l3 * w) * d2);
        acc[1] += 1.0;
    }
}

template <int MODE>
__global__ void __launch_bounds__(BLOCK) k_sweep(const SweepArgs a)
{
    constexpr int NACC = NAcc<MODE>::n;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4 *ypos = reinterpret_cast<float4 *>(smem);                         // [jt]
    unsigned *queue = reinterpret_cast<unsigned *>(smem + (size_t)a.jt * 16); // [4][QCAP]
    double *red = reinterpret_cast<double *>(smem + (size_t)a.jt * 16 + 4 * QCAP * 4); // [4][NACC]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int row0 = a.row_lo + blockIdx.y * (BLOCK * ROWS_PER_LANE);
    const int j0 = blockIdx.x * a.jt;
    const int jn = min(a.jt, a.nb - j0);

    float x0[ROWS_PER_LANE], x1[ROWS_PER_LANE], x2[ROWS_PER_LANE];
#pragma unroll
    for (int r = 0; r < ROWS_PER_LANE; ++r) {
        const int i = row0 + r * BLOCK + tid;
        const float4 p = (i < a.row_hi) ? a.pos_a[i] : nan4();
        x0[r] = p.x; x1[r] = p.y; x2[r] = p.z;
    }
    for (int t = tid; t < jn; t += BLOCK) ypos[t] = a.pos_b[j0 + t];
    __syncthreads();

    double acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = 0.0;

    unsigned *q = queue + wid * QCAP;
    int qn = 0;   // wave-uniform
    const float tau = a.kc.tau;

    for (int jj = 0; jj < jn; ++jj) {
        const float4 y = ypos[jj];   // LDS broadcast
#pragma unroll
        for (int r = 0; r < ROWS_PER_LANE; ++r) {
            const float e0 = x0[r] - y.x, e1 = x1[r] - y.y, e2 = x2[r] - y.z;
            const float d2 = __builtin_fmaf(e2, e2, __builtin_fmaf(e1, e1, e0 * e0));
            const bool pass = d2 < tau;
            const unsigned long long m = __ballot(pass);
            if (m) {
                if (pass) {
                    const unsigned below = __builtin_amdgcn_mbcnt_hi(
                        (unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                    q[qn + below] = ((unsigned)(r * BLOCK + tid) << 16) | (unsigned)jj;
                }
                qn += __popcll(m);
                if (qn >= 64) {
                    qn -= 64;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    const unsigned e = q[qn + lane];
                    const int il = (int)(e >> 16), cj = (int)(e & 0xffffu);
                    process_pair<MODE>(a, row0 + il, j0 + cj, a.pos_a[row0 + il], ypos[cj], acc);
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
    }
    if (qn > 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < qn) {
            const unsigned e = q[lane];
            const int il = (int)(e >> 16), cj = (int)(e & 0xffffu);
            process_pair<MODE>(a, row0 + il, j0 + cj, a.pos_a[row0 + il], ypos[cj], acc);
        }
    }

    // wave reduction (xor butterfly: every lane ends with the same float64 sum,
    // order fixed by the lane ids => deterministic), then 4 waves in order
#pragma unroll
    for (int k = 0; k < NACC; ++k) {
        double s = acc[k];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) red[wid * NACC + k] = s;
    }
    __syncthreads();
    if (tid < NACC) {
        const double s = ((red[tid] + red[NACC + tid]) + red[2 * NACC + tid]) + red[3 * NACC + tid];
        a.partials[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * NACC + tid] = s;
    }
}

void launch_sweep(int mode, const SweepArgs &a, dim3 grid, hipStream_t s)
{
    const size_t smem = (size_t)a.jt * 16 + 4 * QCAP * 4 + 4 * NACC_MAX * sizeof(double);
    switch (mode) {
    case SWEEP_FLOW:
        hipLaunchKernelGGL(k_sweep<SWEEP_FLOW>, grid, dim3(BLOCK), smem, s, a);
        break;
    case SWEEP_STEP:
        hipLaunchKernelGGL(k_sweep<SWEEP_STEP>, grid, dim3(BLOCK), smem, s, a);
        break;
    default:
        hipLaunchKernelGGL(k_sweep<SWEEP_SELF>, grid, dim3(BLOCK), smem, s, a);
        break;
    }
}

// ---------------------------------------------------------------------------
// totals[k] = sum over blocks of partials[b][k], fixed order: thread t adds
// blocks t, t+256, ... then a binary tree over the 256 threads.
__global__ void __launch_bounds__(BLOCK) k_finalize(const double *partials, int nblocks, int nacc,
                                                    double *totals)
{
    __shared__ double sh[BLOCK];
    for (int k = 0; k < nacc; ++k) {
        double s = 0.0;
        for (int b = threadIdx.x; b < nblocks; b += BLOCK) s += partials[(size_t)b * nacc + k];
        sh[threadIdx.x] = s;
        __syncthreads();
        for (int off = BLOCK / 2; off >= 1; off >>= 1) {
            if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
            __syncthreads();
        }
        if (threadIdx.x == 0) totals[k] = sh[0];
        __syncthreads();
    }
}

void launch_finalize(const double *partials, int nblocks, int nacc, double *totals, hipStream_t s)
{
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(BLOCK), 0, s, partials, nblocks, nacc, totals);
}

}   // namespace cvo_dev
