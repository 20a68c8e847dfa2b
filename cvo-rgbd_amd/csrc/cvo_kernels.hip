// cvo_kernels.hip -- gfx950 (MI355X, CDNA4) kernels for the CVO inner loop.
//
//   k_filter    : transform_pcd + the neighbour search of se_kernel
//                 (ref src/cvo.cpp:310-315,110-125): ALL target x source pairs
//                 are tested on the matrix cores (f32 MFMA, K = 4) against a
//                 conservative squared-distance bound; 16x16 tiles with
//                 survivors are appended (with their 256-bit pair mask) to a
//                 tile list in HBM.
//   k_process   : exact evaluation of the tile list / of the kept list
//                   PROC_FLOW  rest of se_kernel + compute_flow (ref cvo.cpp:126-210)
//                   PROC_STEP  compute_step_size sums          (ref cvo.cpp:213-289)
//                   PROC_SELF  acvo Axx / Ayy terms   (ref adaptive_cvo.cpp:156-265)
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
// Why a list.  The reference builds a sparse Gram matrix A once per iteration
// and uses it twice (flow, step size).  Here the dense filter plays the kd-tree
// and the candidate list plays A: it is consumed by PROC_FLOW (which also
// records every kept weight) and again by PROC_STEP, so the all-pairs work is
// done once per iteration, and the expensive per-survivor arithmetic (colour
// distance, two float64 exponentials, float64 accumulation) runs perfectly
// load-balanced on full wavefronts no matter how the survivors cluster.
//
// Arithmetic contract: compiled with -ffp-contract=off; every FMA
// below is an explicit __builtin_fmaf.  Per-pair terms are float32 in the
// reference's operation order, accumulated in float64.  The MFMA filter only
// decides which pairs are LOOKED AT; it never decides membership in A.
#include "cvo_device.h"

namespace cvo_dev {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Eigen: transform.linear()*p + translation, coefficient order, no FMA
__device__ __forceinline__ float4 apply_tf(const float *Rt, const float *t, const float4 p)
{
    float4 o;
    o.x = ((Rt[0] * p.x + Rt[1] * p.y) + Rt[2] * p.z) + t[0];
    o.y = ((Rt[3] * p.x + Rt[4] * p.y) + Rt[5] * p.z) + t[1];
    o.z = ((Rt[6] * p.x + Rt[7] * p.y) + Rt[8] * p.z) + t[2];
    o.w = p.w;   // caller's index bits ride along
    return o;
}

__device__ __forceinline__ float mv_row(const float *m, float x, float y, float z)
{
    return (m[0] * x + m[1] * y) + m[2] * z;
}

// ---------------------------------------------------------------------------
// k_filter
// ---------------------------------------------------------------------------
// LDS carve of one filter block (all 16-byte aligned):
//   bop   [jt/16][64] float      : MFMA B operands of the column chunk, per group of
//                                  16 columns k-major: [-2y'0 x16][-2y'1 x16][-2y'2 x16][|y'|^2 x16]
//   xrow  [ROWS_PER_TILE] float4 : (x'0, x'1, x'2, |x'|^2) of the block's rows
//   stage [4][TILE_STAGE] TileEntry : per-wave staging of the tile entries
size_t filter_smem_bytes(int jt)
{
    return (size_t)jt * 16 + ROWS_PER_TILE * 16 + 4 * TILE_STAGE * sizeof(TileEntry);
}

// append the wave's `n` staged tile entries to sub-list `sub` (exact-size slice)
__device__ __forceinline__ void flush_tiles(const TileEntry *stage, int n, int lane, unsigned sub,
                                            const FilterArgs &a)
{
    unsigned base = 0;
    if (lane == 0) base = atomicAdd(&a.st->sub[a.list][sub], (unsigned)n);
    base = __builtin_amdgcn_readfirstlane(base);
    if (base + (unsigned)n <= a.subcap) {
        const uint4 *src = reinterpret_cast<const uint4 *>(stage);
        uint4 *dst = reinterpret_cast<uint4 *>(a.tiles + (size_t)sub * a.subcap + base);
        for (int w = lane; w < n * 3; w += 64) dst[w] = src[w];
    } else if (lane == 0) {
        atomicOr(&a.st->cnt[2 * a.list + 1], 1u);   // overflow: the host grows the list and resumes
    }
}

__global__ void __launch_bounds__(BLOCK) k_filter(const FilterArgs a)
{
    const long long t_start = a.dbg ? (long long)__builtin_readcyclecounter() : 0;
    const long long w_start = a.dbg ? (long long)wall_clock64() : 0;
    if (a.check_done && a.st->done != 0) return;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *bop = reinterpret_cast<float *>(smem);
    float4 *xrow = reinterpret_cast<float4 *>(smem + (size_t)a.jt * 16);
    TileEntry *stage_all =
        reinterpret_cast<TileEntry *>(smem + (size_t)a.jt * 16 + ROWS_PER_TILE * 16);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int row0 = a.row_lo + blockIdx.y * ROWS_PER_TILE;
    const int j0 = blockIdx.x * a.jt;
    const int jn = min(a.jt, a.nb - j0);
    const int ngroups = (jn + 15) >> 4;
    const float *Rt = a.st->Rt;
    const float *tt = a.st->t;
    const float cx = a.st->center[0], cy = a.st->center[1], cz = a.st->center[2];
    const float tauf = a.st->tauf[a.list];

    // ---- prologue.  Every global load is issued before anything waits on one
    // (clamped addresses instead of control flow), so the block pays ONE memory
    // round trip for its 256 rows and its column chunk, not one per load.
    const int i_row = row0 + tid;
    float4 prow = a.pos_a[min(i_row, a.row_hi - 1)];
    const int ncol = ngroups * 16;
    for (int t0 = 0; t0 < ncol; t0 += 2 * BLOCK) {
        const int ta = t0 + tid, tb = t0 + BLOCK + tid;
        float4 pa = a.pos_b[j0 + min(ta, jn - 1)];
        float4 pb = a.pos_b[j0 + min(tb, jn - 1)];
        if (a.tf_b) { pa = apply_tf(Rt, tt, pa); pb = apply_tf(Rt, tt, pb); }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int t = u ? tb : ta;
            const float4 p = u ? pb : pa;
            if (t < ncol) {
                // MFMA B operands of column t: [-2y'0, -2y'1, -2y'2, |y'|^2]
                const float ax = p.x - cx, ay = p.y - cy, az = p.z - cz;
                const bool real = t < jn;
                float *g = bop + (t >> 4) * 64 + (t & 15);
                g[0] = real ? -2.0f * ax : 0.0f;
                g[16] = real ? -2.0f * ay : 0.0f;
                g[32] = real ? -2.0f * az : 0.0f;
                g[48] = real ? __builtin_fmaf(az, az, __builtin_fmaf(ay, ay, ax * ax)) : PAD_BIG;
            }
        }
    }
    {   // the block's rows: (x'0, x'1, x'2, |x'|^2) into LDS, one row per thread
        if (a.tf_a) prow = apply_tf(Rt, tt, prow);
        const float ax = prow.x - cx, ay = prow.y - cy, az = prow.z - cz;
        const bool real = i_row < a.row_hi;
        xrow[tid] = real ? make_float4(ax, ay, az, __builtin_fmaf(az, az, __builtin_fmaf(ay, ay, ax * ax)))
                         : make_float4(0.0f, 0.0f, 0.0f, PAD_BIG);
    }
    __syncthreads();
    // MFMA A operands: lane l holds A[row = l&15][k = l>>4] of each 16-row tile,
    // k = 3 multiplies |y'|^2 by one.  C operands: lane l holds rows (l>>4)*4 + r:
    // |x'|^2 - tauf, so that D = C + A.B is the filter value itself.
    const int kk = lane >> 4;
    float areg[TILES_PER_WAVE];
    f32x4 creg[TILES_PER_WAVE];
#pragma unroll
    for (int t = 0; t < TILES_PER_WAVE; ++t) {
        const float *xr = reinterpret_cast<const float *>(xrow + wid * ROWS_PER_WAVE + t * 16);
        areg[t] = (kk == 3) ? 1.0f : xr[(lane & 15) * 4 + kk];
#pragma unroll
        for (int r = 0; r < 4; ++r) creg[t][r] = xr[(kk * 4 + r) * 4 + 3] - tauf;
    }

    TileEntry *stage = stage_all + wid * TILE_STAGE;
    int ne = 0;   // wave-uniform: staged tile entries
    // this wave's flushes walk round-robin over the sub-lists
    unsigned sub = ((blockIdx.y * gridDim.x + blockIdx.x) * 4u + (unsigned)wid) * 37u;
    const unsigned rbase = (unsigned)(row0 + wid * ROWS_PER_WAVE);

    const long long t_loop = a.dbg ? (long long)__builtin_readcyclecounter() : 0;
    float bnext = bop[lane];
    for (int g = 0; g < ngroups; ++g) {
        const float b = bnext;
        if (g + 1 < ngroups) bnext = bop[(g + 1) * 64 + lane];   // prefetch the next column tile
        f32x4 d[TILES_PER_WAVE];
#pragma unroll
        for (int t = 0; t < TILES_PER_WAVE; ++t)
            d[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[t], b, creg[t], 0, 0, 0);
        // any filter value negative?  OR of the sign bits
        int ort[TILES_PER_WAVE];
#pragma unroll
        for (int t = 0; t < TILES_PER_WAVE; ++t)
            ort[t] = (__float_as_int(d[t][0]) | __float_as_int(d[t][1])) |
                     (__float_as_int(d[t][2]) | __float_as_int(d[t][3]));
        if (__ballot(((ort[0] | ort[1]) | (ort[2] | ort[3])) < 0) == 0ull) continue;
        // a tile with survivors: its four ballots ARE the 256-bit pair mask
#pragma unroll
        for (int t = 0; t < TILES_PER_WAVE; ++t) {
            if (__ballot(ort[t] < 0) == 0ull) continue;
            const unsigned long long m0 = __ballot(d[t][0] < 0.0f), m1 = __ballot(d[t][1] < 0.0f),
                                     m2 = __ballot(d[t][2] < 0.0f), m3 = __ballot(d[t][3] < 0.0f);
            if (lane == 0) {
                TileEntry te;
                te.row = rbase + (unsigned)t * 16u;
                te.col = (unsigned)(j0 + g * 16);
                te.npairs = (unsigned)(__popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3));
                te.pad_ = 0u;
                te.m[0] = m0; te.m[1] = m1; te.m[2] = m2; te.m[3] = m3;
                stage[ne] = te;
            }
            if (++ne == TILE_STAGE) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                flush_tiles(stage, ne, lane, (sub++) & (NSUB - 1), a);
                __builtin_amdgcn_wave_barrier();
                ne = 0;
            }
        }
    }
    const long long t_tail = a.dbg ? (long long)__builtin_readcyclecounter() : 0;
    if (ne > 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        flush_tiles(stage, ne, lane, sub & (NSUB - 1), a);
    }
    if (a.dbg && lane == 0) {   // probe: start, prologue end, loop end, exit clocks of every wave
        long long *o = a.dbg + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wid) * 8;
        o[6] = w_start; o[7] = (long long)wall_clock64();   // 100 MHz constant clock
        o[0] = t_start; o[1] = t_loop; o[2] = t_tail; o[3] = (long long)__builtin_readcyclecounter();
        o[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_ID: wave slot, SIMD, CU, SE
        o[5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // XCC_ID
    }
}

void launch_filter(const FilterArgs &a, dim3 grid, hipStream_t s)
{
    hipLaunchKernelGGL(k_filter, grid, dim3(BLOCK), filter_smem_bytes(a.jt), s, a);
}

// ---------------------------------------------------------------------------
// k_process
// ---------------------------------------------------------------------------
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
                                             unsigned i, const float *feat_b, unsigned j)
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
template <> struct NAcc<PROC_FLOW> { static constexpr int n = NACC_FLOW; };
template <> struct NAcc<PROC_STEP> { static constexpr int n = NACC_STEP; };
template <> struct NAcc<PROC_SELF> { static constexpr int n = NACC_SELF; };

// One pair of the exact pass.  PROC_FLOW / PROC_SELF: membership test of
// se_kernel (ref cvo.cpp:125-152) and the flow / self sums; returns the weight
// (0 = not in A).  PROC_STEP: `w` is the recorded weight of a member of A.
template <int MODE>
__device__ __forceinline__ float eval_pair(const ProcessArgs &a, const KernConsts &kc, unsigned i,
                                           unsigned j, float w, double *acc)
{
    const float *Rt = a.st->Rt;
    const float *tt = a.st->t;
    float4 xi = a.pos_a[i];
    if (a.tf_a) xi = apply_tf(Rt, tt, xi);
    float4 yj = a.pos_b[j];
    if (a.tf_b) yj = apply_tf(Rt, tt, yj);
    const float e0 = xi.x - yj.x, e1 = xi.y - yj.y, e2 = xi.z - yj.z;
    float d2 = 0.0f;
    if (MODE != PROC_STEP) {
        d2 = __builtin_fmaf(e2, e2, __builtin_fmaf(e1, e1, e0 * e0));
        w = (d2 < kc.tau) ? pair_weight(kc, d2, a.feat_a, i, a.feat_b, j) : 0.0f;
    }
    if (!(w > 0.0f)) return 0.0f;
    if (MODE == PROC_FLOW) {
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
    } else if (MODE == PROC_STEP) {
        // Taylor vectors of y_j (ref cvo.cpp:226-238), for members of A only
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
        if (__float_as_int(xi.w) >= a.first_counted) acc[0] += (double)((kc.inv_l3 * w) * d2);
        acc[1] += 1.0;
    }
    return w;
}

// per-wave LDS of a PROC_FLOW / PROC_SELF block
struct __attribute__((aligned(16))) ProcWaveLds {
    uint2 pairq[PAIR_QUEUE];     // compaction queue of (row, column)
    uint2 kept_ij[KEPT_STAGE];   // staged members of A (PROC_FLOW)
    float kept_a[KEPT_STAGE];
};

// append the wave's `n` staged kept triplets to the kept list (exact-size slice)
__device__ __forceinline__ void flush_kept(const ProcWaveLds *L, int n, int lane, unsigned sub,
                                           const ProcessArgs &a)
{
    unsigned base = 0;
    if (lane == 0) base = atomicAdd(&a.st->sub[LIST_KEPT][sub], (unsigned)n);
    base = __builtin_amdgcn_readfirstlane(base);
    if (base + (unsigned)n <= a.kept_subcap) {
        const size_t o = (size_t)sub * a.kept_subcap + base;
        for (int t = lane; t < n; t += 64) {
            a.kept_ij[o + t] = L->kept_ij[t];
            a.kept_a[o + t] = L->kept_a[t];
        }
    } else if (lane == 0) {
        atomicOr(&a.st->cnt[2 * LIST_KEPT + 1], 1u);
    }
}

template <int MODE>
__global__ void __launch_bounds__(BLOCK) k_process(const ProcessArgs a)
{
    constexpr int NACC = NAcc<MODE>::n;
    if (a.check_done && a.st->done != 0) return;
    __shared__ double red[4 * NACC_MAX];
    __shared__ ProcWaveLds wl[(MODE == PROC_STEP) ? 1 : 4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const KernConsts kc = a.st->kc;
    // PROC_PARTS blocks share one sub-list
    const unsigned sub = blockIdx.x & (NSUB - 1), part = blockIdx.x / NSUB;

    double acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = 0.0;

    if (MODE == PROC_STEP) {
        // stream the members of A recorded by PROC_FLOW
        unsigned n = a.st->sub[LIST_KEPT][sub];
        if (n > a.kept_subcap) n = a.kept_subcap;
        const size_t sbase = (size_t)sub * a.kept_subcap;
        for (unsigned off = part * BLOCK + tid; off < n; off += PROC_PARTS * BLOCK) {
            const uint2 e = a.kept_ij[sbase + off];
            eval_pair<MODE>(a, kc, e.x, e.y, a.kept_a[sbase + off], acc);
        }
    } else {
        ProcWaveLds *L = &wl[wid];
        unsigned n = a.st->sub[a.list][sub];
        if (n > a.subcap) n = a.subcap;   // overflowed list: the iteration is redone anyway
        const TileEntry *tl = a.tiles + (size_t)sub * a.subcap;
        int qn = 0;   // wave-uniform: queued pairs
        int nk = 0;   // wave-uniform: staged kept triplets
        unsigned ksub = (blockIdx.x * 4u + (unsigned)wid) * 37u;
        // evaluate the queued pairs q[base .. base+cnt) (cnt <= 64, wave-uniform)
        auto run_batch = [&](int base, int cnt) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            float w = 0.0f;
            uint2 pr = make_uint2(0u, 0u);
            if (lane < cnt) {
                pr = L->pairq[base + lane];
                w = eval_pair<MODE>(a, kc, pr.x, pr.y, 0.0f, acc);
            }
            if (MODE == PROC_FLOW) {   // record the members of A
                const unsigned long long km = __ballot(w > 0.0f);
                if (w > 0.0f) {
                    const unsigned below = __builtin_amdgcn_mbcnt_hi(
                        (unsigned)(km >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)km, 0u));
                    L->kept_ij[nk + below] = pr;
                    L->kept_a[nk + below] = w;
                }
                nk += __popcll(km);
                if (nk > KEPT_STAGE - 64) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    flush_kept(L, nk, lane, (ksub++) & (NSUB - 1), a);
                    nk = 0;
                }
            }
            __builtin_amdgcn_wave_barrier();
        };
        // the 4 * PROC_PARTS waves of this sub-list take its tile entries in turn
        for (unsigned e = part * 4u + (unsigned)wid; e < n; e += 4u * PROC_PARTS) {
            // expand the tile's 256-bit mask into the queue: bit l of m[r] is
            // row (l>>4)*4 + r, column l&15 of the tile
            const TileEntry *te = tl + e;
            const unsigned trow = te->row, tcol = te->col;
#pragma unroll 1
            for (int r = 0; r < 4; ++r) {
                const unsigned long long m = te->m[r];
                if (m == 0ull) continue;
                if ((m >> lane) & 1ull) {
                    const unsigned below = __builtin_amdgcn_mbcnt_hi(
                        (unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                    L->pairq[qn + below] = make_uint2(trow + (unsigned)((lane >> 4) * 4 + r),
                                                      tcol + (unsigned)(lane & 15));
                }
                qn += __popcll(m);
                if (qn >= 64) {
                    qn -= 64;
                    run_batch(qn, 64);
                }
            }
        }
        if (qn > 0) run_batch(0, qn);
        if (MODE == PROC_FLOW && nk > 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            flush_kept(L, nk, lane, ksub & (NSUB - 1), a);
        }
    }

    // block reduction: xor butterfly inside each wave, then the 4 waves in order
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
        a.partials[(size_t)blockIdx.x * NACC + tid] = s;
    }
}

void launch_process(int mode, const ProcessArgs &a, hipStream_t s)
{
    switch (mode) {
    case PROC_FLOW:
        hipLaunchKernelGGL(k_process<PROC_FLOW>, dim3(PROC_BLOCKS), dim3(BLOCK), 0, s, a);
        break;
    case PROC_STEP:
        hipLaunchKernelGGL(k_process<PROC_STEP>, dim3(PROC_BLOCKS), dim3(BLOCK), 0, s, a);
        break;
    default:
        hipLaunchKernelGGL(k_process<PROC_SELF>, dim3(PROC_BLOCKS), dim3(BLOCK), 0, s, a);
        break;
    }
}

// ---------------------------------------------------------------------------
// Fixed-order reduction of partials[nblocks][NACC] by one 256-thread block:
// thread t adds blocks t, t+256, ...; xor butterfly inside each wave; the four
// wave sums are added in wave order.
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
    const bool acvo = a.prm.mode == CVO_HIP_MODE_ACVO;
    if (a.flags & POST_REDUCE) {
        block_reduce_partials<NACC_FLOW>(a.part_flow, PROC_BLOCKS, sh, st->red + RED_FLOW);
        if (acvo) {
            block_reduce_partials<NACC_SELF>(a.part_xx, PROC_BLOCKS, sh, st->red + RED_XX);
            block_reduce_partials<NACC_SELF>(a.part_yy, PROC_BLOCKS, sh, st->red + RED_YY);
        } else if (threadIdx.x == 0) {
            st->red[RED_XX] = st->red[RED_XX + 1] = st->red[RED_YY] = st->red[RED_YY + 1] = 0.0;
        }
        // a candidate list overflowed on this rank: poison nnz so that, after the
        // all-reduce, EVERY rank takes the same "grow the list and redo" exit
        if (threadIdx.x == 0 &&
            (st->cnt[2 * LIST_XY + 1] | st->cnt[2 * LIST_XX + 1] | st->cnt[2 * LIST_YY + 1] |
             st->cnt[2 * LIST_KEPT + 1]))
            st->red[8] = __builtin_nan("");
    }
    if ((a.flags & POST_MATH) && threadIdx.x == 0) {
        // nothing of an overflowed iteration is usable; the host enlarges the
        // list and resumes from the same (untouched) state
        if (st->red[8] != st->red[8]) {
            st->done = NEED_BIGGER_LIST;
            return;
        }
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
        if (acvo) {   // ref src/adaptive_cvo.cpp:222-231,271
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
        block_reduce_partials<NACC_STEP>(a.part_step, PROC_BLOCKS, sh, st->red + RED_STEP);
    if (!(a.flags & POST_MATH)) return;
    // the lists of this iteration have been consumed: empty them for the next one
    for (int q = threadIdx.x; q < LIST_N * NSUB; q += BLOCK) (&st->sub[0][0])[q] = 0u;
    if (threadIdx.x != 0) return;

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
        st->done = DONE_BREAK_A;
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
        st->done = DONE_BREAK_B;
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
        st->done = DONE_MAX_ITER;   // `iter` keeps its stale value
        return;
    }
    prepare_iteration(st, p);
}

__global__ void k_prepare(DevState *st, const DevParams prm)
{
    for (int q = threadIdx.x; q < LIST_N * NSUB; q += BLOCK) (&st->sub[0][0])[q] = 0u;
    if (threadIdx.x == 0) prepare_iteration(st, prm);
}

void launch_prepare(DevState *st, const DevParams &prm, hipStream_t s)
{
    hipLaunchKernelGGL(k_prepare, dim3(1), dim3(BLOCK), 0, s, st, prm);
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
