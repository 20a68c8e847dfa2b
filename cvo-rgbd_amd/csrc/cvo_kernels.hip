// cvo_kernels.hip -- gfx950 (MI355X, CDNA4) kernels for the CVO inner loop.
//
//   k_sweep     : transform_pcd + se_kernel fused with the consumer of A
//                   SWEEP_FLOW  compute_flow          (ref src/cvo.cpp:99-210,310-315)
//                   SWEEP_STEP  compute_step_size      (ref src/cvo.cpp:213-289)
//                   SWEEP_SELF  acvo Axx / Ayy terms   (ref src/adaptive_cvo.cpp:156-265)
//   k_post_flow : fixed-order float64 reduction of the block partials, then the
//                 O(1) maths that follows compute_flow (twist, dl, Taylor consts)
//   k_post_step : same for compute_step_size: cubic, break tests, Exp_SEK3,
//                 R/T update, length-scale update (ref src/cvo.cpp:291-307,380-410)
//   k_prepare   : inverse transform + kernel constants from (R, T, ell)
//
// The whole align() loop is device-resident: the state lives in a DevState in
// HBM, every kernel starts by reading it (and returns at once when the
// registration has converged), so the host only enqueues launches and polls.
//
// The Gram matrix A is never materialised: every sweep re-tests all
// target x source pairs (dense, wave64: ROWS_PER_LANE target rows per lane, the
// source chunk broadcast from LDS), and the rare survivors of the distance
// test are compacted through a per-wave LDS queue so that the expensive part
// (colour distance, two float64 exponentials, float64 accumulation) always
// runs on full wavefronts.
//
// Arithmetic contract: compiled with -ffp-contract=off; every FMA
// below is an explicit __builtin_fmaf.  Per-pair terms are float32 in the
// reference's operation order, accumulated in float64.
#include "cvo_device.h"

namespace cvo_dev {

__device__ __forceinline__ float4 nan4()
{
    const float q = __builtin_nanf("");
    return make_float4(q, q, q, q);
}

// Eigen: transform.linear()*p + translation, coefficient order, no FMA
__device__ __forceinline__ float4 apply_tf(const float *Rt, const float *t, const float4 p)
{
    float4 o;
    o.x = ((Rt[0] * p.x + Rt[1] * p.y) + Rt[2] * p.z) + t[0];
    o.y = ((Rt[3] * p.x + Rt[4] * p.y) + Rt[5] * p.z) + t[1];
    o.z = ((Rt[6] * p.x + Rt[7] * p.y) + Rt[8] * p.z) + t[2];
    o.w = 0.0f;
    return o;
}

__device__ __forceinline__ float mv_row(const float *m, float x, float y, float z)
{
    return (m[0] * x + m[1] * y) + m[2] * z;
}

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

// pair weight for a pair that passed d2 < tau; 0 if dropped (ref cvo.cpp:143-153)
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
__device__ __forceinline__ void process_pair(const SweepArgs &a, const KernConsts &kc, int i,
                                             int j, const float4 xi, const float4 yj, double *acc)
{
    const float e0 = xi.x - yj.x, e1 = xi.y - yj.y, e2 = xi.z - yj.z;
    const float d2 = __builtin_fmaf(e2, e2, __builtin_fmaf(e1, e1, e0 * e0));
    if (!(d2 < kc.tau)) return;   // also rejects NaN padding
    const float w = pair_weight(kc, d2, a.feat_a, i, a.feat_b, j);
    if (!(w > 0.0f)) return;
    if (MODE == SWEEP_FLOW) {
        // cross(x_i, y_j), y_j - x_i ; (1/c * A_ij) * cross  (ref cvo.cpp:191-198)
        const float c0 = xi.y * yj.z - xi.z * yj.y;
        const float c1 = xi.z * yj.x - xi.x * yj.z;
        const float c2 = xi.x * yj.y - xi.y * yj.x;
        const float f0 = yj.x - xi.x, f1 = yj.y - xi.y, f2 = yj.z - xi.z;
        const float ac = kc.inv_c * w, ad = kc.inv_d * w;
        acc[0] += (double)(ac * c0);
        acc[1] += (double)(ac * c1);
        acc[2] += (double)(ac * c2);
        acc[3] += (double)(ad * f0);
        acc[4] += (double)(ad * f1);
        acc[5] += (double)(ad * f2);
        acc[6] += (double)w;
        acc[7] += (double)((kc.inv_l3 * w) * d2);
        acc[8] += 1.0;
    } else if (MODE == SWEEP_STEP) {
        // Taylor vectors of y_j (ref cvo.cpp:226-238), evaluated for survivors only
        const cvo_math::XiConsts &xc = a.st->xi;
        float xiz[3], xi2z[3], xi3z[3], xi4z[3];
        xiz[0] = (xc.omega[1] * yj.z - xc.omega[2] * yj.y) + xc.v[0];
        xiz[1] = (xc.omega[2] * yj.x - xc.omega[0] * yj.z) + xc.v[1];
        xiz[2] = (xc.omega[0] * yj.y - xc.omega[1] * yj.x) + xc.v[2];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            xi2z[r] = mv_row(xc.W2 + 3 * r, yj.x, yj.y, yj.z) + xc.u2[r];
            xi3z[r] = mv_row(xc.W3 + 3 * r, yj.x, yj.y, yj.z) + xc.u3[r];
            xi4z[r] = mv_row(xc.W4 + 3 * r, yj.x, yj.y, yj.z) + xc.u4[r];
        }
        const float normxiz2 = (xiz[0] * xiz[0] + xiz[1] * xiz[1]) + xiz[2] * xiz[2];
        const float xz12 = -((xiz[0] * xi2z[0] + xiz[1] * xi2z[1]) + xiz[2] * xi2z[2]);
        const float eps_c = ((xi2z[0] * xi2z[0] + xi2z[1] * xi2z[1]) + xi2z[2] * xi2z[2]) +
                            2 * ((xiz[0] * xi3z[0] + xiz[1] * xi3z[1]) + xiz[2] * xi3z[2]);
        // diff_xy = x_i - y_j is (e0,e1,e2); ref cvo.cpp:256-280
        const float cb = kc.cb, cg = kc.cg, cd = kc.cd;
        const float beta = ((cb * xiz[0]) * e0 + (cb * xiz[1]) * e1) + (cb * xiz[2]) * e2;
        const float g_dot = ((2.0f * xi2z[0]) * e0 + (2.0f * xi2z[1]) * e1) + (2.0f * xi2z[2]) * e2;
        const float gamma = cg * (normxiz2 + g_dot);
        const float d_dot = ((-xi3z[0]) * e0 + (-xi3z[1]) * e1) + (-xi3z[2]) * e2;
        const float delta = cd * (xz12 + d_dot);
        const float e_dot = ((2.0f * xi4z[0]) * e0 + (2.0f * xi4z[1]) * e1) + (2.0f * xi4z[2]) * e2;
        const float epsil = cg * (eps_c + e_dot);
        const double A = (double)w;
        const double b = (double)beta, g = (double)gamma;
        acc[0] += (double)(w * beta);
        acc[1] += A * (g + (double)(beta * beta) / 2.0);
        acc[2] += A * ((double)(delta + beta * gamma) + (double)(beta * beta * beta) / 6.0);
        acc[3] += A * ((((double)(epsil + beta * delta) + 0.5 * b * b * g) + 0.5 * g * g) +
                       1 / 24.0 * b * b * b * b);
    } else {
        if (i >= a.first_counted) acc[0] += (double)((kc.inv_l3 * w) * d2);
        acc[1] += 1.0;
    }
}

template <int MODE>
__global__ void __launch_bounds__(BLOCK) k_sweep(const SweepArgs a)
{
    constexpr int NACC = NAcc<MODE>::n;
    if (a.check_done && a.st->done != 0) return;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4 *ypos = reinterpret_cast<float4 *>(smem);                                     // [jt]
    unsigned *queue = reinterpret_cast<unsigned *>(smem + (size_t)a.jt * 16);            // [4][QCAP]
    double *red = reinterpret_cast<double *>(smem + (size_t)a.jt * 16 + 4 * QCAP * 4);  // [4][NACC]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int row0 = a.row_lo + blockIdx.y * ROWS_PER_TILE;
    const int j0 = blockIdx.x * a.jt;
    const int jn = min(a.jt, a.nb - j0);
    const KernConsts kc = a.st->kc;
    const float *Rt = a.st->Rt;
    const float *tt = a.st->t;

    float x0[ROWS_PER_LANE], x1[ROWS_PER_LANE], x2[ROWS_PER_LANE];
#pragma unroll
    for (int r = 0; r < ROWS_PER_LANE; ++r) {
        const int i = row0 + r * BLOCK + tid;
        float4 p = nan4();
        if (i < a.row_hi) {
            p = a.pos_a[i];
            if (a.tf_a) p = apply_tf(Rt, tt, p);
        }
        x0[r] = p.x; x1[r] = p.y; x2[r] = p.z;
    }
    for (int t = tid; t < jn; t += BLOCK) {
        float4 p = a.pos_b[j0 + t];
        if (a.tf_b) p = apply_tf(Rt, tt, p);
        ypos[t] = p;
    }
    __syncthreads();

    double acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = 0.0;

    unsigned *q = queue + wid * QCAP;
    int qn = 0;   // wave-uniform
    const float tau = kc.tau;

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
                    float4 xi = a.pos_a[row0 + il];
                    if (a.tf_a) xi = apply_tf(Rt, tt, xi);
                    process_pair<MODE>(a, kc, row0 + il, j0 + cj, xi, ypos[cj], acc);
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
            float4 xi = a.pos_a[row0 + il];
            if (a.tf_a) xi = apply_tf(Rt, tt, xi);
            process_pair<MODE>(a, kc, row0 + il, j0 + cj, xi, ypos[cj], acc);
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
// Fixed-order reduction of partials[nblocks][NACC] by one 256-thread block:
// thread t adds blocks t, t+256, ...; xor butterfly inside each wave; the four
// wave sums are added in wave order.  Result broadcast to all threads via LDS.
template <int NACC>
__device__ void block_reduce_partials(const double *part, int nblocks, double *sh /*[4*NACC_MAX]*/,
                                      double *out /*[NACC], thread 0 writes*/)
{
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    double s[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) s[k] = 0.0;
    for (int b = tid; b < nblocks; b += BLOCK) {
#pragma unroll
        for (int k = 0; k < NACC; ++k) s[k] += part[(size_t)b * NACC + k];
    }
#pragma unroll
    for (int k = 0; k < NACC; ++k) {
        double v = s[k];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) sh[wid * NACC_MAX + k] = v;
    }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < NACC; ++k)
            out[k] = ((sh[k] + sh[NACC_MAX + k]) + sh[2 * NACC_MAX + k]) + sh[3 * NACC_MAX + k];
    }
    __syncthreads();
}

__global__ void __launch_bounds__(BLOCK) k_post_flow(const PostFlowArgs a)
{
    __shared__ double sh[4 * NACC_MAX];
    DevState *st = a.st;
    if (a.check_done && st->done != 0) return;
    if (a.flags & POST_REDUCE) {
        block_reduce_partials<NACC_FLOW>(a.part_flow, a.nb_flow, sh, st->red + RED_FLOW);
        if (a.prm.mode == CVO_HIP_MODE_ACVO) {
            block_reduce_partials<NACC_SELF>(a.part_xx, a.nb_xx, sh, st->red + RED_XX);
            block_reduce_partials<NACC_SELF>(a.part_yy, a.nb_yy, sh, st->red + RED_YY);
        } else if (threadIdx.x == 0) {
            st->red[RED_XX] = st->red[RED_XX + 1] = st->red[RED_YY] = st->red[RED_YY + 1] = 0.0;
        }
    }
    if ((a.flags & POST_MATH) && threadIdx.x == 0) {
        const double *red = st->red;
        float omega[3], v[3];
        for (int q = 0; q < 3; ++q) {
            omega[q] = (float)red[q];        // omega = double_omega.cast<float>()
            v[q] = (float)red[3 + q];
            st->omega[q] = omega[q];
            st->v[q] = v[q];
        }
        st->xi = cvo_math::make_xi_consts(omega, v);
        double dl = 0.0;
        const long long nnz = (long long)red[8];
        long long nnz_xx = 0, nnz_yy = 0;
        if (a.prm.mode == CVO_HIP_MODE_ACVO) {   // ref src/adaptive_cvo.cpp:222-231,271
            nnz_xx = (long long)red[RED_XX + 1];
            nnz_yy = (long long)red[RED_YY + 1];
            const double num = (red[RED_YY] - 2.0 * red[7]) + red[RED_XX];
            dl = num / (double)(nnz_xx + nnz_yy - 2 * nnz);
        }
        st->dl = dl;
        if (a.trace && st->k < a.trace_cap) {
            cvo_hip_trace &tr = a.trace[st->k];
            tr.k = st->k;
            tr.exit_code = 0;
            tr.ell = st->ell;
            for (int q = 0; q < 3; ++q) {
                tr.omega[q] = omega[q]; tr.v[q] = v[q];
                tr.omega_d[q] = red[q]; tr.v_d[q] = red[3 + q];
            }
            tr.sum_a = red[6];
            tr.dl = dl;
            tr.nnz = nnz; tr.nnz_xx = nnz_xx; tr.nnz_yy = nnz_yy;
        }
    }
}

__global__ void __launch_bounds__(BLOCK) k_post_step(const PostStepArgs a)
{
    __shared__ double sh[4 * NACC_MAX];
    DevState *st = a.st;
    if (a.check_done && st->done != 0) return;
    if (a.flags & POST_REDUCE)
        block_reduce_partials<NACC_STEP>(a.part_step, a.nb_step, sh, st->red + RED_STEP);
    if (!((a.flags & POST_MATH) && threadIdx.x == 0)) return;

    const DevParams &p = a.prm;
    const bool acvo = p.mode == CVO_HIP_MODE_ACVO;
    const int k = st->k;
    double bcde[4];
    for (int q = 0; q < 4; ++q) bcde[q] = st->red[RED_STEP + q];
    const float step = cvo_math::pick_step(bcde, p.min_step);
    float omega[3], v[3];
    for (int q = 0; q < 3; ++q) { omega[q] = st->omega[q]; v[q] = st->v[q]; }
    cvo_hip_trace *tr = (a.trace && k < a.trace_cap) ? &a.trace[k] : nullptr;
    if (tr) {
        for (int q = 0; q < 4; ++q) tr->bcde[q] = bcde[q];
        tr->step = step;
        tr->dist = __builtin_nanf("");
    }
    st->n_exec = k + 1;
    for (int q = 0; q < 9; ++q) st->used_Rt[q] = st->Rt[q];
    for (int q = 0; q < 3; ++q) st->used_t[q] = st->t[q];

    // break A: both twist norms below eps (ref cvo.cpp:380 float norms,
    // adaptive_cvo.cpp:509 double norms of the float vectors)
    bool brk;
    if (acvo) {
        const double nw = sqrt((double)omega[0] * omega[0] +
                               ((double)omega[1] * omega[1] + (double)omega[2] * omega[2]));
        const double nv = sqrt((double)v[0] * v[0] + ((double)v[1] * v[1] + (double)v[2] * v[2]));
        brk = nw < (double)p.eps && nv < (double)p.eps;
    } else {
        brk = cvo_math::norm_fixed3(omega) < p.eps && cvo_math::norm_fixed3(v) < p.eps;
    }
    if (brk) {
        st->iter = k;
        st->done = 1;
        if (tr) tr->exit_code = 1;
        return;
    }
    // integrate: T = R*dT + T ; R = R*dR  (ref cvo.cpp:391-399)
    float dR[9], dT[3], RdT[3];
    cvo_math::exp_se3(omega, v, step, dR, dT);
    cvo_math::Mat3 R, dRm;
    for (int q = 0; q < 9; ++q) { R.m[q] = st->R[q]; dRm.m[q] = dR[q]; }
    cvo_math::mulv(R, dT, RdT);
    for (int q = 0; q < 3; ++q) st->T[q] = RdT[q] + st->T[q];
    const cvo_math::Mat3 Rn = cvo_math::mul(R, dRm);
    for (int q = 0; q < 9; ++q) st->R[q] = Rn.m[q];

    const float dist = cvo_math::dist_se3(omega, v, step);
    if (tr) tr->dist = dist;
    if (dist < p.eps_2) {   // break B
        st->iter = k;
        st->done = 2;
        if (tr) tr->exit_code = 2;
        return;
    }
    // length-scale update
    float ell = st->ell;
    if (acvo) {   // ref src/adaptive_cvo.cpp:538-545
        ell = (float)((double)ell + p.dl_step * st->dl);
        if (ell >= st->ell_max) {
            ell = (float)(st->ell_max * 0.7);
            st->ell_max = (float)(st->ell_max * 0.7);
        }
        ell = (ell < p.ell_min) ? p.ell_min : ell;
    } else {      // ref src/cvo.cpp:408-410
        ell = (k > 2) ? (float)0.10 : ell;
        ell = (k > 9) ? (float)0.06 : ell;
        ell = (k > 19) ? (float)0.03 : ell;
    }
    st->ell = ell;
    st->k = k + 1;
    if (k + 1 >= p.max_iter) {
        st->done = 3;   // MAX_ITER exhausted: `iter` keeps its stale value
        return;
    }
    prepare_iteration(st, p);
}

__global__ void k_prepare(DevState *st, const DevParams prm)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) prepare_iteration(st, prm);
}

void launch_prepare(DevState *st, const DevParams &prm, hipStream_t s)
{
    hipLaunchKernelGGL(k_prepare, dim3(1), dim3(64), 0, s, st, prm);
}

void launch_post_flow(const PostFlowArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(k_post_flow, dim3(1), dim3(BLOCK), 0, s, a);
}

void launch_post_step(const PostStepArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(k_post_step, dim3(1), dim3(BLOCK), 0, s, a);
}

}   // namespace cvo_dev
