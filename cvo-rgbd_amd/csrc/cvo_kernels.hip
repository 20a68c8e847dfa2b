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
#include <algorithm>
#include <atomic>
#include <cstdlib>

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
    o.w = p.w;   // the point's 5th feature rides along
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
//   near  [FILTER_KMAX][4] u32   : per (item, wave): the column segments that can hold a pair
constexpr int FILTER_KMAX = 64;   // items culled per pass
size_t filter_smem_bytes(int jt)
{
    return (size_t)jt * 16 + ROWS_PER_TILE * 16 + 4 * TILE_STAGE * sizeof(TileEntry) +
           FILTER_KMAX * 4 * 4;
}

// Can a point of sphere a be within sqrt(tauf) of a point of sphere b?
// Conservative against float32 rounding (relative and absolute slack).
__device__ __forceinline__ bool spheres_near(const float4 sa, const float4 sb, float reach)
{
    const float ex = sa.x - sb.x, ey = sa.y - sb.y, ez = sa.z - sb.z;
    const float d2 = __builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex));
    const float rr = (sa.w + sb.w + reach) * 1.00001f + 1e-5f;
    return d2 * 0.99999f <= rr * rr;
}

// Append the wave's `n` staged tile entries (n <= TILE_STAGE = 128) to the 32 sub-lists of its row
// region, DEALT in runs of L = 1 << DEAL_SHIFT = 4 entries (64-byte stores; runs of 1, 2, 4 measured alike,
// whole flushes 3 % slower: profiles/r02_ab.txt): run c (entries cL .. cL+L-1) goes to
// sub-list ((c + rot) & 31) * 8 + region.  Whole flushes to one sub-list each -- the first
// version -- left the sub-lists of a 10k x 10k pair with 2.6 (ell = 0.15) to 4.9 (ell = 0.03)
// times the mean load on the fullest one (a few flushes of up to 128 entries per sub-list:
// Poisson), and the list kernels run as long as their fullest block.  Lane j < 32 reserves
// the room of sub-list j with one returning atomic (32 distinct addresses: one wave
// instruction, as before).
__device__ __forceinline__ void flush_tiles(const TileEntry *stage, int n, int lane, unsigned rot, unsigned region,
                                            const FilterArgs &a, int list, TileEntry *tiles, const int par)
{
    static_assert(TILE_STAGE <= 128, "two entries per lane");
    constexpr int sh = DEAL_SHIFT, L = 1 << sh;
    unsigned base = 0;
    if (lane < 32) {
        int cnt = 0;
        for (int c = (int)(((unsigned)lane - rot) & 31u); (c << sh) < n; c += 32) cnt += min(n - (c << sh), L);
        if (cnt) base = atomicAdd(&a.st->sub[list][((unsigned)lane << 3) | region], (unsigned)cnt);
    }
    bool over = false;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        // (the shuffle runs with every lane enabled: a disabled source lane would read as zero)
        const int k = lane + 64 * h;
        const unsigned c = (unsigned)k >> sh;
        const unsigned j = (c + rot) & 31u;
        const unsigned pos = (unsigned)__shfl((int)base, (int)j, 64) + ((c >> 5) << sh) + ((unsigned)k & (unsigned)(L - 1));
        if (k < n) {
            if (pos < a.subcap) tiles[(size_t)((j << 3) | region) * a.subcap + pos] = stage[k];
            else over = true;
        }
    }
    if (__ballot(over) != 0ull && lane == 0)
        atomicOr(&a.st->ovf[par][list], 1u);   // overflow: the host grows the list and resumes
}

// transform_pcd as a pass of its own (ref cvo.cpp:310-315), for the launches that many registrations
// share: the moving cloud under the iteration's [Rt|t], once per point instead of once per pair in
// the flow and step passes (18 VALU operations per 64 pairs in each).  Same arithmetic, same values.
__device__ __forceinline__ void pretransform_body(const FilterArgs &a, const unsigned bid, const unsigned nblocks)
{
    if (!a.pos_bt) return;
    if (a.check_done && a.st->done != 0) return;
    const float *Rt = a.st->Rt;
    const float *tt = a.st->t;
    // (nblocks: the blocks that take part -- at most one per BLOCK points, see kt_filter)
    for (unsigned j = bid * BLOCK + threadIdx.x; j < (unsigned)a.nb; j += nblocks * BLOCK) {
        const float4 y0 = a.pos_b[j];
        float4 y = apply_tf(Rt, tt, y0);
        y.w = y0.w;
        a.pos_bt[j] = y;
    }
}

// hd: the copy of the state's head this launch starts from (a.st unless head mode); par: the launch's parity
// (which row of DevState::ovf an overflow is flagged in; 0 unless head mode)
template <bool PIPE = true>   // (false: the merged launches of one registration on its own, at the scalar-register ceiling)
__device__ __forceinline__ void filter_body(const FilterArgs &a, const unsigned bid, const unsigned nblocks,
                                            const DevState *__restrict__ hd, const int par)
{
    // Persistent blocks: the (column chunk, row tile) items of this registration's
    // a.gx x a.gy work grid are dealt round-robin to the gridDim.x blocks of the
    // launch, so that a launch that has nothing to do (list re-used, loop finished)
    // costs a few hundred blocks, not thousands.
    // XCD-aware dealing: block b runs on XCD b % 8 (workgroups go round-robin to the
    // XCDs) and takes items of the row tiles of region b % 8 only (the 8 eighths of
    // the row-tile range; rows are in Morton order), the same regions whose tile
    // entries that XCD's list-kernel blocks consume: every XCD's L2 holds one region
    // of the clouds for the whole iteration instead of all of them.
    const int reg = (int)(bid & 7u), slot = (int)(bid >> 3), nslot = (int)(nblocks >> 3);
    const int by_lo = (reg * a.gy + 7) / 8, by_hi = ((reg + 1) * a.gy + 7) / 8;   // tiles of this region
    const int nitems = (by_hi - by_lo) * a.gx;
    if (slot >= nitems) return;
#ifdef CVO_FILTER_PROBE   // tools/microbench/filter_probe.hip: per-wave phase clocks (costs ~10 scalar registers)
    const long long t_start = a.dbg ? (long long)__builtin_readcyclecounter() : 0;
    const long long w_start = a.dbg ? (long long)wall_clock64() : 0;
#endif
    // first round trip: the loop-control word, the state constants and this
    // thread's bounding spheres are all fetched before anything waits
    // (inside align() a list that is still valid is consumed again: nothing to do)
    // (async xy: this launch builds the buffer the plan step scheduled, if any)
    const int kind = a.async_xy;   // 0: synchronous list; 1: xy, 2: xx, 3: yy built ahead into the idle buffer
    const int target = kind == 1 ? hd->xy_target : (kind >= 2 ? hd->sf_target[kind - 2] : 0);
    const int done_word = a.check_done ? (hd->done | (kind ? (target < 0) : hd->reuse[a.list])) : 0;
    const int out_list = kind == 1 ? (target == 1 ? (int)LIST_XYB : (int)LIST_XY)
                                   : (kind >= 2 ? self_list_id(kind - 2, target == 1 ? 1 : 0) : a.list);
    TileEntry *out_tiles = (kind && target == 1) ? a.tiles_b : a.tiles;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *bop = reinterpret_cast<float *>(smem);
    float4 *xrow = reinterpret_cast<float4 *>(smem + (size_t)a.jt * 16);
    TileEntry *stage_all =
        reinterpret_cast<TileEntry *>(smem + (size_t)a.jt * 16 + ROWS_PER_TILE * 16);
    unsigned *nearmask = reinterpret_cast<unsigned *>(smem + (size_t)a.jt * 16 + ROWS_PER_TILE * 16 +
                                                      4 * TILE_STAGE * sizeof(TileEntry));

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    // (an asynchronous xy build runs at the transform its plan step recorded for it: the state's own in
    // the classic launches -- nothing moved since the plan --, one slot older in head mode)
    const float *Rt = kind == 1 ? hd->xy_Rt[target == 1 ? 1 : 0] : hd->Rt;
    const float *tt = kind == 1 ? hd->xy_t[target == 1 ? 1 : 0] : hd->t;
    const float cx = hd->center[0], cy = hd->center[1], cz = hd->center[2];
    const float tauf = kind == 1 ? hd->tauf_build
                                 : (kind >= 2 ? hd->sf_tauf_build[kind - 2] : hd->tauf[a.list]);
    // ---- culling, for all the items of this block at once.  The clouds are in
    // Morton order, so the 64 rows of a wave and every run of 64 columns are
    // compact patches with precomputed bounding spheres (rigid motion moves a
    // sphere's centre, not its radius).  A (wave, column segment) pair whose
    // spheres are more than sqrt(tauf) apart cannot hold a pair with d2 < tau.
    // One thread per (item, wave, segment) test, all loads in flight together: the
    // items that hold nothing cost no round trip of their own.
    const int k_all = (nitems - slot + nslot - 1) / nslot;
    const int ncf = a.jt / SEG;   // column segments of a full item
    const float reach = sqrtf(tauf);
    for (int k0 = 0; k0 < k_all; k0 += FILTER_KMAX) {
    const int kn = min(FILTER_KMAX, k_all - k0);
    for (int q = tid; q < FILTER_KMAX * 4; q += BLOCK) nearmask[q] = 0u;
    __syncthreads();
    for (int t = tid; t < kn * 4 * ncf; t += BLOCK) {
        const int k = t / (4 * ncf), rem = t - k * 4 * ncf;
        const int w = rem / ncf, u = rem - w * ncf;
        const int item = slot + (k0 + k) * nslot;
        const int bx = item % a.gx, by = by_lo + item / a.gx;
        const int row0 = a.row_lo + by * ROWS_PER_TILE;
        const int j0 = bx * a.jt;
        const int ncseg = (min(a.jt, a.nb - j0) + SEG - 1) / SEG;
        // clamped so that every thread can load unconditionally (the rows of a
        // wave touch at most two segments)
        const int r_first = row0 + w * ROWS_PER_WAVE;
        const int r_last = min(r_first + ROWS_PER_WAVE, a.row_hi) - 1;
        const int last_seg = (a.row_hi - 1) >> 6;
        const int sg0 = min(r_first >> 6, last_seg), sg1 = min(max(r_last, r_first) >> 6, last_seg);
        float4 sb = a.seg_b[(j0 >> 6) + min(u, ncseg - 1)];
        float4 sa0 = a.seg_a[sg0], sa1 = a.seg_a[sg1];
        if (done_word != 0) return;   // first wait: everything above is in flight
        if (u < ncseg && r_first < a.row_hi) {
            if (a.tf_b) { const float r = sb.w; sb = apply_tf(Rt, tt, sb); sb.w = r; }
            if (a.tf_a) {
                float r = sa0.w; sa0 = apply_tf(Rt, tt, sa0); sa0.w = r;
                r = sa1.w; sa1 = apply_tf(Rt, tt, sa1); sa1.w = r;
            }
            if (spheres_near(sa0, sb, reach) || spheres_near(sa1, sb, reach))
                atomicOr(&nearmask[k * 4 + w], 1u << u);
        }
    }
    if (done_word != 0) return;
    __syncthreads();
    for (int k = 0; k < kn; ++k) {
    if ((nearmask[k * 4] | nearmask[k * 4 + 1] | nearmask[k * 4 + 2] | nearmask[k * 4 + 3]) == 0u)
        continue;   // nothing near in this item
    const int item = slot + (k0 + k) * nslot;
    const int bx = item % a.gx, by = by_lo + item / a.gx;
    const int row0 = a.row_lo + by * ROWS_PER_TILE;
    const int j0 = bx * a.jt;
    const int jn = min(a.jt, a.nb - j0);
    const int ngroups = (jn + 15) >> 4;
#ifdef CVO_FILTER_NO_PF   // (A/B builds)
    constexpr bool PF = false;
#else
    constexpr bool PF = PIPE;
#endif
    // (PF: wave-uniform, so that the loop over the column groups runs on scalar branches)
    const unsigned mynear = PF ? (unsigned)__builtin_amdgcn_readfirstlane((int)nearmask[k * 4 + wid]) : nearmask[k * 4 + wid];

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
    // XCD-aware sub-list choice: the 256 sub-lists are consumed by the list kernels'
    // blocks b = sub (mod 256), and workgroups go round-robin to the 8 XCDs, so
    // sub-list s is read on XCD s % 8.  Rows are in Morton order: the 8 eighths of
    // the row range are compact regions; a wave's entries go to the 32 sub-lists of
    // ITS region (round-robin inside), so that each XCD's L2 serves one region of
    // both clouds instead of all of them (matters from ~50k points on).
    const unsigned region =
        (unsigned)(((long long)(row0 + wid * ROWS_PER_WAVE - a.row_lo) * 8) / max(a.row_hi - a.row_lo, 1)) & 7u;
    unsigned sub = ((by * a.gx + bx) * 4u + (unsigned)wid) * 37u;
    const unsigned rbase = (unsigned)(row0 + wid * ROWS_PER_WAVE);

#ifdef CVO_FILTER_PROBE
    const long long t_loop = a.dbg ? (long long)__builtin_readcyclecounter() : 0;
#endif
    // The loop over the groups of 16 columns, software-pipelined by hand: the next LIVE group (culled segments of 64
    // columns are skipped on scalar registers) is known one step ahead and its B operands are requested before this
    // group's matrix instructions go out -- exactly one LDS read in flight per step, so the wait in front of the
    // matrix instructions covers the operand of THIS step only.  (An LDS round trip per group was what a wave with
    // one or two neighbours on its SIMD waited for most.)
    auto next_live = [&](int g) {   // first group >= g whose segment is not culled (ngroups: none)
        while (g < ngroups && (g & 3) == 0 && ((mynear >> (g >> 2)) & 1u) == 0u) g += 4;
        return g;
    };
    int g = PF ? next_live(0) : 0;
    float b_cur = PF ? bop[min(g, ngroups - 1) * 64 + lane] : 0.0f;
    while (g < ngroups) {
        float b;
        int g_this;
        if (PF) {
            const int gn = next_live(g + 1);
            b = b_cur;
            b_cur = bop[min(gn, ngroups - 1) * 64 + lane];
            g_this = g;
            g = gn;
        } else {
            if ((g & 3) == 0 && ((mynear >> (g >> 2)) & 1u) == 0u) { g += 4; continue; }   // culled segment
            b = bop[g * 64 + lane];
            g_this = g;
            ++g;
        }
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
        // Every result register with survivors becomes one entry: its ballot IS the
        // pair mask.  All 16 ballots are taken, lane k (< 16) adopts mask k, and the
        // lanes with a non-empty mask write their entries in one LDS store.
        unsigned mlo = 0u, mhi = 0u;
#pragma unroll
        for (int t = 0; t < TILES_PER_WAVE; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned long long m = __ballot(d[t][r] < 0.0f);
                if (lane == t * 4 + r) { mlo = (unsigned)m; mhi = (unsigned)(m >> 32); }
            }
        }
        const bool has = (mlo | mhi) != 0u;   // only lanes < 16 can be true
        const unsigned long long nz = __ballot(has);
        if (has) {
            const unsigned below = __builtin_amdgcn_mbcnt_lo((unsigned)nz, 0u);
            stage[ne + below] = make_uint4(rbase + (unsigned)(lane >> 2) * 16u,
                                           (unsigned)(j0 + g_this * 16) | ((unsigned)(lane & 3) << 30),
                                           mlo, mhi);
        }
        ne += __popcll(nz);
        if (ne > TILE_STAGE - 16) {   // no room for another full group: flush (rare)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            flush_tiles(stage, ne, lane, sub, region, a, out_list, out_tiles, par);
            sub += (unsigned)((ne + (1 << DEAL_SHIFT) - 1) >> DEAL_SHIFT);
            __builtin_amdgcn_wave_barrier();
            ne = 0;
        }
    }
#ifdef CVO_FILTER_PROBE
    const long long t_tail = a.dbg ? (long long)__builtin_readcyclecounter() : 0;
#endif
    if (ne > 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        flush_tiles(stage, ne, lane, sub, region, a, out_list, out_tiles, par);
    }
#ifdef CVO_FILTER_PROBE
    if (a.dbg && lane == 0) {   // probe: start, prologue end, loop end, exit clocks of every wave
        long long *o = a.dbg + ((size_t)(by * a.gx + bx) * 4 + wid) * 8;
        o[6] = w_start; o[7] = (long long)wall_clock64();   // 100 MHz constant clock
        o[0] = t_start; o[1] = t_loop; o[2] = t_tail; o[3] = (long long)__builtin_readcyclecounter();
        o[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_ID: wave slot, SIMD, CU, SE
        o[5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // XCC_ID
    }
#endif
    __syncthreads();   // the next item re-uses the LDS staging areas
    }   // live items
    __syncthreads();
    }   // chunks of FILTER_KMAX items
}

__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(6, 8)))
k_filter(const Grp<FilterArgs> grp)
{
    filter_body(grp.a[blockIdx.z], blockIdx.x, gridDim.x, grp.a[blockIdx.z].st, 0);
}

// blocks of one registration's filter launch (persistent, see k_filter)
static long long filter_blocks_max() { return 2048; }   // (1024 / 2048 / 4096 persistent blocks: within 1 %, profiles/r02_ab.txt 7)

// grid.x of one registration: a multiple of 8 (one eighth of the blocks per XCD /
// row region), enough for the largest region's items up to the cap
static unsigned filter_grid_x(long long nitems, long long cap)
{
    const long long per_region = (nitems + 7) / 8 + 1;   // regions differ by at most one row tile
    const long long want = std::min<long long>(per_region * 8, std::max<long long>(cap, 8));
    return (unsigned)((want + 7) / 8 * 8);
}

void launch_filter(const FilterArgs &a, dim3 grid, hipStream_t s, hipEvent_t ev_start,
                   hipEvent_t ev_stop)
{
    Grp<FilterArgs> g;
    g.a[0] = a;
    g.a[0].gx = (int)grid.x; g.a[0].gy = (int)grid.y;
    grid = dim3(filter_grid_x((long long)grid.x * grid.y, filter_blocks_max()), 1, 1);
    if (ev_start && ev_stop)   // the events take the dispatch packet's own begin / end timestamps
        hipExtLaunchKernelGGL(k_filter, grid, dim3(BLOCK), filter_smem_bytes(a.jt), s, ev_start,
                              ev_stop, 0, g);
    else
        hipLaunchKernelGGL(k_filter, grid, dim3(BLOCK), filter_smem_bytes(a.jt), s, g);
}

void launch_filter_group(const FilterArgs *a, int n, hipStream_t s)
{
    Grp<FilterArgs> g;
    dim3 grid(1, 1, (unsigned)n);
    int jt = 0;
    // the launch as a whole gets about as many blocks as a single registration would
    const long long cap = std::max<long long>(64, filter_blocks_max() / (2 * n));
    for (int i = 0; i < n; ++i) {
        g.a[i] = a[i];
        grid.x = std::max(grid.x, filter_grid_x((long long)a[i].gx * a[i].gy, cap));
        jt = std::max(jt, a[i].jt);
    }
    hipLaunchKernelGGL(k_filter, grid, dim3(BLOCK), filter_smem_bytes(jt), s, g);
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

// ---------------------------------------------------------------------------
// exp(x) for the kernel weights, x <= 0 (ref cvo.cpp:149-150 calls the double
// overload of exp).  2^(k/64) table + degree-5 polynomial on |r| <= ln2/128, the
// scheme of every libm: at most one ulp from glibc's exp (differs from it in the
// last bit for 25 % of the arguments) and -- what the arithmetic contract needs --
// the float32 value of sigma^2 exp(x) was the same for all of 2*10^8 random
// arguments in [-6, 0].  14 float64 operations instead of ~30 in the device libm;
// the two exp are the largest single item of the per-pair work.
__device__ const double c_exp2_64[64] = {
    0x1.0000000000000p+0, 0x1.02c9a3e778061p+0, 0x1.059b0d3158574p+0, 0x1.0874518759bc8p+0,
    0x1.0b5586cf9890fp+0, 0x1.0e3ec32d3d1a2p+0, 0x1.11301d0125b51p+0, 0x1.1429aaea92de0p+0,
    0x1.172b83c7d517bp+0, 0x1.1a35beb6fcb75p+0, 0x1.1d4873168b9aap+0, 0x1.2063b88628cd6p+0,
    0x1.2387a6e756238p+0, 0x1.26b4565e27cddp+0, 0x1.29e9df51fdee1p+0, 0x1.2d285a6e4030bp+0,
    0x1.306fe0a31b715p+0, 0x1.33c08b26416ffp+0, 0x1.371a7373aa9cbp+0, 0x1.3a7db34e59ff7p+0,
    0x1.3dea64c123422p+0, 0x1.4160a21f72e2ap+0, 0x1.44e086061892dp+0, 0x1.486a2b5c13cd0p+0,
    0x1.4bfdad5362a27p+0, 0x1.4f9b2769d2ca7p+0, 0x1.5342b569d4f82p+0, 0x1.56f4736b527dap+0,
    0x1.5ab07dd485429p+0, 0x1.5e76f15ad2148p+0, 0x1.6247eb03a5585p+0, 0x1.6623882552225p+0,
    0x1.6a09e667f3bcdp+0, 0x1.6dfb23c651a2fp+0, 0x1.71f75e8ec5f74p+0, 0x1.75feb564267c9p+0,
    0x1.7a11473eb0187p+0, 0x1.7e2f336cf4e62p+0, 0x1.82589994cce13p+0, 0x1.868d99b4492edp+0,
    0x1.8ace5422aa0dbp+0, 0x1.8f1ae99157736p+0, 0x1.93737b0cdc5e5p+0, 0x1.97d829fde4e50p+0,
    0x1.9c49182a3f090p+0, 0x1.a0c667b5de565p+0, 0x1.a5503b23e255dp+0, 0x1.a9e6b5579fdbfp+0,
    0x1.ae89f995ad3adp+0, 0x1.b33a2b84f15fbp+0, 0x1.b7f76f2fb5e47p+0, 0x1.bcc1e904bc1d2p+0,
    0x1.c199bdd85529cp+0, 0x1.c67f12e57d14bp+0, 0x1.cb720dcef9069p+0, 0x1.d072d4a07897cp+0,
    0x1.d5818dcfba487p+0, 0x1.da9e603db3285p+0, 0x1.dfc97337b9b5fp+0, 0x1.e502ee78b3ff6p+0,
    0x1.ea4afa2a490dap+0, 0x1.efa1bee615a27p+0, 0x1.f50765b6e4540p+0, 0x1.fa7c1819e90d8p+0,
};

__device__ __forceinline__ double exp_neg(double x, const double *tab /* LDS copy of c_exp2_64 */)
{
    const double kd = __builtin_rint(x * 0x1.71547652b82fep+6);          // x * 64/ln2
    const int k = (int)kd;
    double r = __builtin_fma(-kd, 0x1.62e42fee00000p-7, x);              // ln2/64, high part (exact product)
    r = __builtin_fma(-kd, 0x1.a39ef35793c76p-39, r);                    // low part
    double p = 1.0 / 120.0;
    p = __builtin_fma(p, r, 1.0 / 24.0);
    p = __builtin_fma(p, r, 1.0 / 6.0);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = p * r;
    const double s = tab[k & 63];
    return __builtin_ldexp(__builtin_fma(s, p, s), k >> 6);
}

// pair weight for a pair that passed d2 < tau; 0 if dropped.
// WEIGHT 0: the C++ objects' (ref cvo.cpp:143-153).  WEIGHT 1: the MATLAB object's (SURVEY 8
// a9, ref rkhs_se3_registration.m:40-73,125-127): linear colour inner product, threshold on
// K alone -- a separate instantiation (k_process<PROC_FLOW, 1>), so that the kernels of the
// main path carry nothing of it (a run-time branch here cost them 6 %).
template <int WEIGHT>
__device__ __forceinline__ float pair_weight(const KernConsts &kc, float d2, const float4 fa0,
                                             const float fa4, const float4 fb0, const float fb4,
                                             const double *etab)
{
    if (WEIGHT == 1) {
        const float km = (float)(kc.s2_d * exp_neg((double)d2 * kc.ninv_2l2, etab));
        if (!(km >= kc.sp)) return 0.0f;
        const float ci = kc.cscale * ((fa0.x * fb0.x + fa0.y * fb0.y) + fa0.z * fb0.z);
        return ci * km;
    }
    const float d2c = d2_feat(fa0, fa4, fb0, fb4);
    if (!(d2c < kc.tau_c)) return 0.0f;
    const float k = (float)(kc.s2_d * exp_neg((double)d2 * kc.ninv_2l2, etab));
    const float ck = (float)(kc.cs2_d * exp_neg((double)d2c * kc.ninv_2cl2, etab));
    const float a = ck * k;
    return a > kc.sp ? a : 0.0f;
}

// the two halves of pair_weight<0>, for the flow pass with a candidate list: ck is a function of the two
// points' features alone (0 = the colour cut fails; a passing pair has ck > 0)
__device__ __forceinline__ float colour_weight(const KernConsts &kc, const float4 fa0, const float fa4,
                                               const float4 fb0, const float fb4, const double *etab)
{
    const float d2c = d2_feat(fa0, fa4, fb0, fb4);
    if (!(d2c < kc.tau_c)) return 0.0f;
    return (float)(kc.cs2_d * exp_neg((double)d2c * kc.ninv_2cl2, etab));
}
__device__ __forceinline__ float weight_from_ck(const KernConsts &kc, float d2, float ck, const double *etab)
{
    const float k = (float)(kc.s2_d * exp_neg((double)d2 * kc.ninv_2l2, etab));
    const float a = ck * k;
    return a > kc.sp ? a : 0.0f;
}


// ---------------------------------------------------------------------------
// Wave-wide float64 sums of N per-lane values, written to dst[0..N).
//
// A butterfly per value costs 6 exchanges (12 ds_bpermute for a double), and the
// LDS pipe -- not the VALU -- was what the list kernels waited on.  This is a
// reduce-scatter instead: at the level with lane mask OFF the lower lane of each
// pair keeps the first half of the values and the upper lane the second half,
// each adding what its partner held of its own half, so the number of live
// values halves per level (9 -> 5 -> 3 -> 2 -> 1: 13 exchanges instead of 54).
// Levels 32 and 16 use gfx950's v_permlane32_swap / v_permlane16_swap, levels 8,
// 4, 2, 1 DPP moves (row_mirror, row_half_mirror, quad_perm: partners l^15, l^7,
// l^2, l^1 -- the masks 32,16,15,7,2,1 are independent, so every level joins two
// disjoint halves): nothing goes through the LDS crossbar.  The order of the
// additions is fixed: results are reproducible.
template <int CTRL> __device__ __forceinline__ double dpp_mov_f64(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

template <int OFF> __device__ __forceinline__ double wave_xchg(double x)
{
    if constexpr (OFF == 8) return dpp_mov_f64<0x140>(x);        // row_mirror
    else if constexpr (OFF == 4) return dpp_mov_f64<0x141>(x);   // row_half_mirror
    else if constexpr (OFF == 2) return dpp_mov_f64<0x4E>(x);    // quad_perm [2,3,0,1]
    else if constexpr (OFF == 1) return dpp_mov_f64<0xB1>(x);    // quad_perm [1,0,3,2]
    else return __shfl_xor(x, OFF, 64);
}

template <int N, int OFF> struct WaveRS {
    static constexpr int H = (N + 1) / 2;
    static __device__ __forceinline__ void run(double *v, int lane)
    {
        if constexpr (N > 1) {
            const bool upper = (lane & OFF) != 0;
#pragma unroll
            for (int q = 0; q < H; ++q) {
                const double hi = (q + H < N) ? v[q + H] : 0.0;
                if constexpr (OFF >= 16) {
                    // gfx950 v_permlane{32,16}_swap: the upper half (odd rows) of the
                    // first operand trades places with the lower half (even rows) of
                    // the second, so {kept, received} come out without any select
                    const unsigned xl = (unsigned)__double2loint(v[q]), xh = (unsigned)__double2hiint(v[q]);
                    const unsigned yl = (unsigned)__double2loint(hi), yh = (unsigned)__double2hiint(hi);
                    double k0, k1;
                    if constexpr (OFF == 32) {
                        const auto rl = __builtin_amdgcn_permlane32_swap(xl, yl, false, false);
                        const auto rh = __builtin_amdgcn_permlane32_swap(xh, yh, false, false);
                        k0 = __hiloint2double((int)rh[0], (int)rl[0]);
                        k1 = __hiloint2double((int)rh[1], (int)rl[1]);
                    } else {
                        const auto rl = __builtin_amdgcn_permlane16_swap(xl, yl, false, false);
                        const auto rh = __builtin_amdgcn_permlane16_swap(xh, yh, false, false);
                        k0 = __hiloint2double((int)rh[0], (int)rl[0]);
                        k1 = __hiloint2double((int)rh[1], (int)rl[1]);
                    }
                    v[q] = k0 + k1;
                } else {
                    const double send = upper ? v[q] : hi;
                    const double keep = upper ? hi : v[q];
                    v[q] = keep + wave_xchg<OFF>(send);
                }
            }
        } else {
            v[0] += wave_xchg<OFF>(v[0]);
        }
        if constexpr (OFF > 1) WaveRS<(N > 1 ? H : 1), OFF / 2>::run(v, lane);
    }
    // which of the N sums this lane ends up holding in v[0] (-1: none / a duplicate)
    static __device__ __forceinline__ int slot(int lane)
    {
        int p = 0;
        if constexpr (OFF > 1) p = WaveRS<(N > 1 ? H : 1), OFF / 2>::slot(lane);
        if (p < 0) return -1;
        if constexpr (N > 1) {
            p += (lane & OFF) ? H : 0;
            return p < N ? p : -1;
        } else {
            return (lane & OFF) ? -1 : p;   // plain butterfly level: one writer per pair
        }
    }
};

template <int N>
__device__ __forceinline__ void wave_sums(double (&v)[N], int lane, double *dst)
{
    WaveRS<N, 32>::run(v, lane);
    const int p = WaveRS<N, 32>::slot(lane);
    if (p >= 0) dst[p] = v[0];
}

template <int MODE> struct NAcc;
template <> struct NAcc<PROC_FLOW> { static constexpr int n = NACC_FLOW; };
template <> struct NAcc<PROC_STEP> { static constexpr int n = NACC_STEP; };
template <> struct NAcc<PROC_SELF> { static constexpr int n = NACC_SELF; };

// x / 6.0, correctly rounded, in three operations instead of the division sequence (~12): with
// c = RN(1/6), q0 = x c, r = x - 6 q0 (exact in one fma), the quotient is q0 + r / 6 exactly, and
// RN(q0 + r c) differs from it by less than 2^-53 ulp -- while x / 6 = (x / 2) / 3 is never closer
// than 1/6 ulp to a rounding boundary (thirds), so both round to the same double.  (Finite x.)
__device__ __forceinline__ double div6(double x)
{
    const double c = 0x1.5555555555555p-3;
    const double q0 = x * c;
    const double r = __builtin_fma(-6.0, q0, x);
    return __builtin_fma(r, c, q0);
}

// the streamed lists (candidate record, kept list) are read once per pass: CVO_NT_LISTS (A/B builds) marks those loads
// non-temporal, so that they leave the L2 to the gathered clouds
__device__ __forceinline__ uint2 list_load(const uint2 *p)
{
#ifdef CVO_NT_LISTS
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(p));
    return make_uint2(v.x, v.y);
#else
    return *p;
#endif
}

// a 16-byte gather at a 32-bit byte offset from a wave-uniform base (clouds of < 2^27 points): the
// address is scalar base + vector offset, one shift per gather instead of 64-bit address arithmetic
__device__ __forceinline__ const float4 *gather16(const void *base, unsigned byte_off)
{
    return reinterpret_cast<const float4 *>(static_cast<const char *>(base) + byte_off);
}

// What a list pass takes from the state's head.  The classic launches point it at the state in global
// memory (scalar loads where a value is needed); in head mode the block has just computed the head
// itself (LDS) and keeps the hot part in scalar registers (proc_head_lds).
struct ProcHead {
    const float *Rt, *tt;          // inverse transform of the slot
    const cvo_math::XiConsts *xi;  // PROC_STEP
    KernConsts kc;
    int done_word;                 // done | stall
    int n_fixed;
    int second;                    // the pass reads the second buffer of its (asynchronous) tile list
    unsigned list_bad;             // the list the pass consumes overflowed while it was written
    int ck_nblk;                   // DevState::ck_nblk of the pass's tile list (head mode: xy_ck of the buffer in use)
    int par;                       // row of DevState::ovf this launch flags overflows in
    uint2 *cand;                   // the candidate record of the list (buffer) the pass reads, its per-wave counts
    uint32_t *cand_cnt;
    int need_d2;                   // ProcessArgs::need_d2 -- or 0 where the kernel is built for loops that never read that sum
};

template <int MODE>
__device__ __forceinline__ ProcHead proc_head_global(const ProcessArgs &a, const DevState *st, const int par)
{
    ProcHead h;
    h.Rt = st->Rt; h.tt = st->t; h.xi = &st->xi;
    h.kc = st->kc;
    // (async xy: a stall slot only builds; PROC_FLOW reads the buffer in use)
    h.done_word = a.check_done ? (st->done | (a.async_xy ? st->stall : 0)) : 0;
    // (acvo Ayy rule, SURVEY 8a quirk 5: the caller's count of fixed points lives in the state, so
    // that the kernel arguments -- and with them a captured graph -- do not depend on it)
    h.n_fixed = (MODE == PROC_SELF && a.first_counted) ? st->n_fixed : 0;
    h.second = ((MODE == PROC_FLOW && a.async_xy && st->xy_active == 1) ||
                (MODE == PROC_SELF && a.async_self && st->sf_active[a.async_self - 1] == 1)) ? 1 : 0;
    // A list that overflowed while it was built holds counters past what was written (an
    // append that does not fit is dropped, its count stays): entries from memory nobody
    // initialised would be taken for row / column numbers.  The iteration is redone with a
    // larger list anyway: consume nothing of it.  (Same for the kept list of PROC_STEP.  A list that
    // was built ahead is only ever switched to after its flag was seen clear: plan_xy_async.)
    const bool ahead = (MODE == PROC_FLOW && a.async_xy) || (MODE == PROC_SELF && a.async_self);
    h.list_bad = ahead ? 0u : a.st->ovf[par][MODE == PROC_STEP ? (int)LIST_KEPT : a.list];
    h.ck_nblk = st->ck_nblk[a.list];
    h.par = par;
    h.cand = a.cand; h.cand_cnt = a.cand_cnt;
    h.need_d2 = a.need_d2;
    return h;
}

// Kept-list entries (ProcessArgs::kept_packed).  `raw_w`: the entry's word of the weight array (mode 0 only).
__device__ __forceinline__ uint2 kept_pack(const int mode, const unsigned ebase, const unsigned i, const unsigned j, const float w)
{
    if (mode == 1) return make_uint2(i | (j << 16), __float_as_uint(w));
    const unsigned wb = __float_as_uint(w);
    return make_uint2(i | (j << 18), (j >> 14) | (((wb >> 23) - ebase) << 4) | ((wb & 0x7fffffu) << 8));
}
__device__ __forceinline__ void kept_unpack(const int mode, const unsigned ebase, const uint2 e, const float raw_w, unsigned &i,
                                            unsigned &j, float &w)
{
    if (mode == 0) { i = e.x; j = e.y; w = raw_w; }
    else if (mode == 1) { i = e.x & 0xffffu; j = e.x >> 16; w = __uint_as_float(e.y); }
    else {
        i = e.x & 0x3ffffu;
        j = (e.x >> 18) | ((e.y & 0xfu) << 14);
        w = __uint_as_float(((((e.y >> 4) & 0xfu) + ebase) << 23) | ((e.y >> 8) & 0x7fffffu));
    }
}

// One pair of the exact pass.  PROC_FLOW / PROC_SELF: membership test of
// se_kernel (ref cvo.cpp:125-152) and the flow / self sums; returns the weight
// (0 = not in A).  PROC_STEP: `w` is the recorded weight of a member of A.
// CK (PROC_FLOW, WEIGHT 0): 0 as the reference writes it; 1 also hands the pair's colour weight out
// through *ck_io (computed for every pair, inside tau or not); 2 takes it from *ck_io, no features read.
// What eval_pair reads of a launch's arguments.  The list passes hand it a copy whose members are pinned in scalar
// registers (pin_pair_src): read through the argument table, the compiler re-issues the scalar loads of the two cloud
// addresses and the transform switches in EVERY round of a streaming loop -- cheap instructions, but each a round trip
// through the scalar cache that the round's gathers wait for.
// (A pointer that went through the pin is an opaque value to the compiler: without the explicit global address
// space its loads would become flat ones.)
#define CVO_GLOBAL __attribute__((address_space(1)))
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
struct PairSrc {
    const CVO_GLOBAL char *pos_a; const float *feat_a;
    const CVO_GLOBAL char *pos_b; const float *feat_b;
    int tf_a, tf_b;
};
#ifdef CVO_NO_PIN   // (A/B builds)
constexpr bool kPinBuild = false;
#else
constexpr bool kPinBuild = true;
#endif
#ifdef CVO_NO_PF
constexpr bool kPrefetchBuild = false;
#else
constexpr bool kPrefetchBuild = true;
#endif
// PIN false: the kernels that play several roles per launch (the merged and head-mode launches of one registration on
// its own) sit at the scalar-register ceiling already -- pinned values would come back as v_readlane traffic -- and
// are bound by their launch chain, not by these loops: they keep the plain form.
template <bool PIN, class T> __device__ __forceinline__ const CVO_GLOBAL char *pin_global(const T *p)
{
    unsigned long long v = (unsigned long long)p;
    if (PIN && kPinBuild) asm volatile("" : "+s"(v));
    return (const CVO_GLOBAL char *)v;
}
template <bool PIN> __device__ __forceinline__ unsigned pin_u32(unsigned v)
{
    if (PIN && kPinBuild) asm volatile("" : "+s"(v));
    return v;
}
template <bool PIN> __device__ __forceinline__ PairSrc pin_pair_src(const ProcessArgs &a)
{
    PairSrc p;
    p.pos_a = pin_global<PIN>(a.pos_a); p.pos_b = pin_global<PIN>(a.pos_b);
    p.feat_a = a.feat_a; p.feat_b = a.feat_b;
    p.tf_a = (int)pin_u32<PIN>((unsigned)a.tf_a); p.tf_b = (int)pin_u32<PIN>((unsigned)a.tf_b);
    return p;
}
template <bool W>
__device__ __forceinline__ float4 load_pos(const void *base, unsigned byte_off)
{
    return *reinterpret_cast<const float4 *>(static_cast<const char *>(base) + byte_off);   // (.w unused: the compiler narrows the load itself)
}
template <bool W>
__device__ __forceinline__ float4 load_pos(const CVO_GLOBAL char *base, unsigned byte_off)
{
    if (W) {
        const f32x4_t v = *reinterpret_cast<const CVO_GLOBAL f32x4_t *>(base + byte_off);
        return make_float4(v.x, v.y, v.z, v.w);
    }
    typedef float f32x3_t __attribute__((ext_vector_type(3)));
    const f32x3_t v = *reinterpret_cast<const CVO_GLOBAL f32x3_t *>(base + byte_off);   // (12 of the row's 16 bytes)
    return make_float4(v.x, v.y, v.z, 0.0f);
}
__device__ __forceinline__ uint2 load8(const CVO_GLOBAL char *base, unsigned idx)
{
#ifdef CVO_NT_LISTS
    const u32x2_t v = __builtin_nontemporal_load(reinterpret_cast<const CVO_GLOBAL u32x2_t *>(base) + idx);
#else
    const u32x2_t v = reinterpret_cast<const CVO_GLOBAL u32x2_t *>(base)[idx];
#endif
    return make_uint2(v.x, v.y);
}
__device__ __forceinline__ void store8(const CVO_GLOBAL char *base, unsigned idx, unsigned x, unsigned y)
{
    u32x2_t v; v.x = x; v.y = y;
    CVO_GLOBAL u32x2_t *q = reinterpret_cast<CVO_GLOBAL u32x2_t *>(const_cast<CVO_GLOBAL char *>(base)) + idx;
#ifdef CVO_NT_KEPT_ST
    __builtin_nontemporal_store(v, q);
#else
    *q = v;
#endif
}

// the Taylor constants of an iteration are the same for every lane: a copy in scalar registers (read once per wave; read
// through the state's pointer inside a streaming loop they come back as loads of every round)
__device__ __forceinline__ cvo_math::XiConsts xi_uniform(const cvo_math::XiConsts &g)
{
    cvo_math::XiConsts xc;
    auto uni = [](float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); };
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        xc.omega[q] = uni(g.omega[q]); xc.v[q] = uni(g.v[q]);
        xc.u2[q] = uni(g.u2[q]); xc.u3[q] = uni(g.u3[q]); xc.u4[q] = uni(g.u4[q]);
    }
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        xc.W2[q] = uni(g.W2[q]); xc.W3[q] = uni(g.W3[q]); xc.W4[q] = uni(g.W4[q]);
    }
    return xc;
}

// The sums of one member of A, shared by the list passes (eval_pair) and the resident runs (kt_run).
// compute_flow (ref src/cvo.cpp:191-204): xi = x_i, yj = the transformed y_j, w = A_ij > 0, d2 = |x_i - y_j|^2
__device__ __forceinline__ void pair_flow_sums(const KernConsts &kc, const float4 xi, const float4 yj, const float w, const float d2,
                                           const int need_d2, double *acc)
{
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
    // (the sum of the weights goes into trace records and cvo_hip_flow's answer, the sum of a d2 is acvo's dl term:
    // neither is read inside a cvo loop that keeps no trace -- ProcessArgs::need_d2)
#ifdef CVO_SUM_A_ALWAYS   // (A/B builds: profiles/r04_ab.txt 12)
    acc[6] += (double)w;
    if (need_d2) acc[7] += (double)((kc.inv_l3 * w) * d2);
#else
    if (need_d2) {
        acc[6] += (double)w;
        acc[7] += (double)((kc.inv_l3 * w) * d2);
    }
#endif
    // (acc[8], the number of members: counted per wave by the caller, not per pair here)
}

// the float64 part of a member's step terms (ref src/cvo.cpp:275-280), shared by the one-member and the two-member forms below
__device__ __forceinline__ void step_tail(const float w, const float beta, const float gamma, const float delta, const float epsil, double *acc)
{
    const double A = (double)w;
    const double b = (double)beta, g = (double)gamma;
    acc[0] += (double)(w * beta);
#ifdef CVO_STEP_TAIL_LITERAL
    // the source line's operations one by one (ref cvo.cpp:275-280 under C's promotion rules): 33 float64-rate
    // instructions per member; kept for A/B builds (profiles/r04_ab.txt 11)
    acc[1] += A * (g + (double)(beta * beta) / 2.0);
    acc[2] += A * ((double)(delta + beta * gamma) + div6((double)(beta * beta * beta)));
    acc[3] += A * ((((double)(epsil + beta * delta) + 0.5 * b * b * g) + 0.5 * g * g) +
                   1 / 24.0 * b * b * b * b);
#else
    // The float64 part of the terms with fused operations: every float32 product of the source line is formed
    // and promoted as written (beta*beta, beta*beta*beta, delta + beta*gamma, epsil + beta*delta); what is
    // fused are float64 operations whose separate roundings the reference's own sum order already outweighs
    // (a term moves by <= 3 ulp(float64), the sum of ~10^5..10^6 of them is order-dependent at 10^-13).
    // 20 float64-rate instructions per member instead of 33: the step pass is issue-bound while the kernel is
    // wide (profiles/r04_ab.txt 8, 11).
    acc[1] = __builtin_fma(A, __builtin_fma((double)(beta * beta), 0.5, g), acc[1]);
    acc[2] = __builtin_fma(A, __builtin_fma((double)(beta * beta * beta), 0x1.5555555555555p-3,
                                            (double)(delta + beta * gamma)), acc[2]);
    const double b2 = b * b, hg = 0.5 * g;
    double t = __builtin_fma(b2, hg, (double)(epsil + beta * delta));
    t = __builtin_fma(hg, g, t);
    t = __builtin_fma(b2 * (1 / 24.0), b2, t);
    acc[3] = __builtin_fma(A, t, acc[3]);
#endif
}

// compute_step_size (ref src/cvo.cpp:226-238,256-280): (e0, e1, e2) = x_i - y_j
__device__ __forceinline__ void pair_step_sums(const KernConsts &kc, const cvo_math::XiConsts &xc, const float4 yj, const float e0,
                                           const float e1, const float e2, const float w, double *acc)
{
    // Taylor vectors of y_j (ref cvo.cpp:226-238), for members of A only
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
    step_tail(w, beta, gamma, delta, epsil, acc);
}

// The same for TWO members at once (round 6; A/B builds with -DCVO_STEP_TWO, measured and not adopted: profiles/r06_ab.txt 22): the float32 part --
// ~130 of a member's ~180 instructions -- on pairs of floats, which gfx950 executes as packed instructions (v_pk_mul_f32 / v_pk_add_f32: two IEEE
// operations per lane and issue slot, each rounded as the single one is; the constants come from scalar registers, broadcast by the
// instruction).  Every expression is pair_step_sums' own, operation for operation; the float64 tails follow one after the other, member 0
// first: a lane that takes its members two at a time adds exactly what it added one at a time, in the same order.  A member with w = 0 (a
// lane's odd last one) adds exact zeros.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void pair_step_sums2(const KernConsts &kc, const cvo_math::XiConsts &xc, const f32x2 yx, const f32x2 yy, const f32x2 yz,
                                                const f32x2 e0, const f32x2 e1, const f32x2 e2, const f32x2 w, double *acc)
{
    f32x2 xiz[3], xi2z[3], xi3z[3], xi4z[3];
    xiz[0] = (xc.omega[1] * yz - xc.omega[2] * yy) + xc.v[0];
    xiz[1] = (xc.omega[2] * yx - xc.omega[0] * yz) + xc.v[1];
    xiz[2] = (xc.omega[0] * yy - xc.omega[1] * yx) + xc.v[2];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        xi2z[r] = ((xc.W2[3 * r] * yx + xc.W2[3 * r + 1] * yy) + xc.W2[3 * r + 2] * yz) + xc.u2[r];
        xi3z[r] = ((xc.W3[3 * r] * yx + xc.W3[3 * r + 1] * yy) + xc.W3[3 * r + 2] * yz) + xc.u3[r];
        xi4z[r] = ((xc.W4[3 * r] * yx + xc.W4[3 * r + 1] * yy) + xc.W4[3 * r + 2] * yz) + xc.u4[r];
    }
    const f32x2 normxiz2 = (xiz[0] * xiz[0] + xiz[1] * xiz[1]) + xiz[2] * xiz[2];
    const f32x2 xz12 = -((xiz[0] * xi2z[0] + xiz[1] * xi2z[1]) + xiz[2] * xi2z[2]);
    const f32x2 eps_c = ((xi2z[0] * xi2z[0] + xi2z[1] * xi2z[1]) + xi2z[2] * xi2z[2]) +
                        2.0f * ((xiz[0] * xi3z[0] + xiz[1] * xi3z[1]) + xiz[2] * xi3z[2]);
    const float cb = kc.cb, cg = kc.cg, cd = kc.cd;
    const f32x2 beta = ((cb * xiz[0]) * e0 + (cb * xiz[1]) * e1) + (cb * xiz[2]) * e2;
    const f32x2 g_dot = ((2.0f * xi2z[0]) * e0 + (2.0f * xi2z[1]) * e1) + (2.0f * xi2z[2]) * e2;
    const f32x2 gamma = cg * (normxiz2 + g_dot);
    const f32x2 d_dot = ((-xi3z[0]) * e0 + (-xi3z[1]) * e1) + (-xi3z[2]) * e2;
    const f32x2 delta = cd * (xz12 + d_dot);
    const f32x2 e_dot = ((2.0f * xi4z[0]) * e0 + (2.0f * xi4z[1]) * e1) + (2.0f * xi4z[2]) * e2;
    const f32x2 epsil = cg * (eps_c + e_dot);
    step_tail(w.x, beta.x, gamma.x, delta.x, epsil.x, acc);
    step_tail(w.y, beta.y, gamma.y, delta.y, epsil.y, acc);
}

template <int MODE, int WEIGHT = 0, int CK = 0, class ARGS = ProcessArgs>
__device__ __forceinline__ float eval_pair(const ARGS &a, const ProcHead &hd, const KernConsts &kc, unsigned i,
                                           unsigned j, float w, double *acc,
                                           const cvo_math::XiConsts &xc, const double *etab = nullptr,
                                           const int first_counted = 0, float *ck_io = nullptr,
                                           uint2 *pf_out = nullptr, const CVO_GLOBAL char *pf_base = nullptr, unsigned pf_idx = 0)
{
    const float *Rt = hd.Rt;
    const float *tt = hd.tt;
    // (both gathers -- and the features' -- are requested before anything waits or branches)
    constexpr bool need_w = MODE != PROC_STEP && CK != 2;
    float4 xi = load_pos<need_w>(a.pos_a, i * 16u);
    float4 yj = load_pos<need_w>(a.pos_b, j * 16u);
    // (a streaming loop's next record: requested BEHIND the gathers -- loads return in order -- so that the wait for the
    // gathers leaves it in flight while the pair is evaluated)
    if (pf_out) *pf_out = load8(pf_base, pf_idx);
    // the features are fetched together with the positions (one memory round
    // trip per pair instead of two); ~97 % of the filtered pairs need them
    float4 fa0 = make_float4(0.f, 0.f, 0.f, 0.f), fb0 = fa0;
    float fa4 = 0.f, fb4 = 0.f;
    int row_index = 0;
    if (MODE != PROC_STEP && CK != 2) {
        fa0 = *gather16(a.feat_a, i * (unsigned)(FEAT_STRIDE * 4));
        fb0 = *gather16(a.feat_b, j * (unsigned)(FEAT_STRIDE * 4));
        fa4 = xi.w;   // the 5th feature travels in pos.w
        fb4 = yj.w;
        if (MODE == PROC_SELF)   // the caller's index of the row (acvo Ayy rule)
            row_index = __float_as_int(a.feat_a[(size_t)i * FEAT_STRIDE + FEAT_INDEX_SLOT]);
    }
    if (a.tf_a) xi = apply_tf(Rt, tt, xi);
    if (a.tf_b) yj = apply_tf(Rt, tt, yj);
    const float e0 = xi.x - yj.x, e1 = xi.y - yj.y, e2 = xi.z - yj.z;
    float d2 = 0.0f;
    if (MODE != PROC_STEP) {
        d2 = __builtin_fmaf(e2, e2, __builtin_fmaf(e1, e1, e0 * e0));
        if (CK == 0) {
            w = (d2 < kc.tau) ? pair_weight<WEIGHT>(kc, d2, fa0, fa4, fb0, fb4, etab) : 0.0f;
        } else {
            // (PROC_SELF: the sign of the recorded weight says whether the row counts -- acvo Ayy rule)
            const float ck = CK == 2 ? __builtin_fabsf(*ck_io) : colour_weight(kc, fa0, fa4, fb0, fb4, etab);
            if (CK == 2 && MODE == PROC_SELF) row_index = (*ck_io < 0.0f) ? -1 : 0;
            if (CK == 1) *ck_io = (MODE == PROC_SELF && row_index < first_counted) ? -ck : ck;
            w = (d2 < kc.tau && ck > 0.0f) ? weight_from_ck(kc, d2, ck, etab) : 0.0f;
        }
    }
    if (!(w > 0.0f)) return 0.0f;
    if (MODE == PROC_FLOW) {
        pair_flow_sums(kc, xi, yj, w, d2, hd.need_d2, acc);
    } else if (MODE == PROC_STEP) {
#ifdef CVO_PROBE_STEP_EXP   // (probe builds, profiles/r05_ab.txt 9: the least a step pass without a stored weight must do -- d2 and one exp per member)
        {
            const float d2p = __builtin_fmaf(e2, e2, __builtin_fmaf(e1, e1, e0 * e0));
            const float kp = (float)(kc.s2_d * exp_neg((double)d2p * kc.ninv_2l2, etab ? etab : c_exp2_64));   // (kt_step_twist passes no table)
            asm volatile("" ::"v"(kp));
        }
#endif
        pair_step_sums(kc, xc, yj, e0, e1, e2, w, acc);
    } else {
        if (CK == 2 ? row_index >= 0 : row_index >= first_counted) acc[0] += (double)((kc.inv_l3 * w) * d2);
        // (acc[1], the number of members: counted per wave by the caller)
    }
    return w;
}

// The step pass over one wave's slice of the kept list, two members per lane and trip (A/B builds, see process_body).  e / w: the lane's first
// entry (and, unpacked lists, its weight), already requested by the caller.
template <bool PF>
__device__ __forceinline__ void step_members2(const PairSrc &src, const ProcHead &hd, const KernConsts &kc, const cvo_math::XiConsts &xc,
                                              const CVO_GLOBAL char *kept_w, const float *kept_a, const int packed, const unsigned ebase,
                                              const unsigned wcap, const unsigned n, const int lane, uint2 e, float w, double *acc)
{
    const float *Rt = hd.Rt, *tt = hd.tt;
    uint2 eb = load8(kept_w, min((unsigned)lane + 64u, wcap - 1u));
    for (unsigned off = lane; off < n; off += 128u) {
        const bool two = off + 64u < n;
        if (off >= 128u && !PF) { e = load8(kept_w, off); eb = load8(kept_w, min(off + 64u, wcap - 1u)); }
        float wa = w, wb = 0.0f;
        if (!packed) { wa = kept_a[off]; wb = kept_a[min(off + 64u, wcap - 1u)]; }
        if (!two) { eb = e; wb = wa; }
        unsigned ia, ja, ib, jb;
        float ma, mb;
        kept_unpack(packed, ebase, e, wa, ia, ja, ma);
        kept_unpack(packed, ebase, eb, wb, ib, jb, mb);
        if (!two) mb = 0.0f;
        // (the four gathers, then the next trip's entries behind them: loads return in order)
        const float4 xa = load_pos<false>(src.pos_a, ia * 16u), ya = load_pos<false>(src.pos_b, ja * 16u);
        const float4 xb = load_pos<false>(src.pos_a, ib * 16u), yb = load_pos<false>(src.pos_b, jb * 16u);
        if (PF) { e = load8(kept_w, min(off + 128u, wcap - 1u)); eb = load8(kept_w, min(off + 192u, wcap - 1u)); }
        f32x2 px = {xa.x, xb.x}, py = {xa.y, xb.y}, pz = {xa.z, xb.z};
        f32x2 qx = {ya.x, yb.x}, qy = {ya.y, yb.y}, qz = {ya.z, yb.z};
        if (src.tf_a) {   // (apply_tf's expressions)
            const f32x2 ox = ((Rt[0] * px + Rt[1] * py) + Rt[2] * pz) + tt[0];
            const f32x2 oy = ((Rt[3] * px + Rt[4] * py) + Rt[5] * pz) + tt[1];
            const f32x2 oz = ((Rt[6] * px + Rt[7] * py) + Rt[8] * pz) + tt[2];
            px = ox; py = oy; pz = oz;
        }
        if (src.tf_b) {
            const f32x2 ox = ((Rt[0] * qx + Rt[1] * qy) + Rt[2] * qz) + tt[0];
            const f32x2 oy = ((Rt[3] * qx + Rt[4] * qy) + Rt[5] * qz) + tt[1];
            const f32x2 oz = ((Rt[6] * qx + Rt[7] * qy) + Rt[8] * qz) + tt[2];
            qx = ox; qy = oy; qz = oz;
        }
        const f32x2 mw = {ma, mb};
        pair_step_sums2(kc, xc, qx, qy, qz, px - qx, py - qy, pz - qz, mw, acc);
    }
}

// LDS of a list-kernel block: handed in, so that launches whose blocks play different
// roles (flow pass, self passes, filter) overlay one allocation instead of adding them up
constexpr int PROC_SMEM = 4 * NACC_MAX * 8 + 4 * PAIR_QUEUE * 8 + 4 * 64 * 8;

// The tile list of one registration, expanded and evaluated by the block's four waves (PROC_FLOW,
// PROC_SELF).  REC (PROC_FLOW): every candidate is recorded for the passes that follow (ProcessArgs::cand).
// Returns false when the loop has stopped (nothing to reduce).
template <int MODE, int WEIGHT, int REC, bool PIPE>
__device__ __forceinline__ bool expand_lists(const ProcessArgs &a, const ProcHead &hd, const KernConsts &kc, const unsigned bid, const int wid,
                                             const int lane, const unsigned wave, const int done_word,
                                             const unsigned list_bad, const int in_list, const TileEntry *in_tiles,
                                             const int first_counted, const double *s_etab, uint2 *pairq_all,
                                             double (&acc)[NAcc<MODE>::n])
{
        // nblk >= NSUB: nblk / NSUB blocks share one sub-list of the tile list;
        // nblk < NSUB (many registrations per launch): a block takes NSUB / nblk
        // sub-lists, one after the other (sub-list s stays on XCD s % 8 either way)
        const bool shared = a.nblk >= NSUB;
        const unsigned nsl = shared ? 1u : (unsigned)(NSUB / a.nblk);
        const unsigned part = shared ? bid / NSUB : 0u;
        unsigned sub = bid & (NSUB - 1);
        unsigned n = list_bad ? 0u : a.st->sub[in_list][sub];
        const TileEntry *tl = in_tiles + (size_t)sub * a.subcap;
        // The waves of a sub-list take its entries in turn, 64 at a time: lane l
        // fetches the wave's l-th entry of the round (one memory round trip per 64
        // entries, the first one together with the count).
        const unsigned stride = shared ? 4u * (unsigned)(a.nblk / NSUB) : 4u;
        const unsigned e0 = part * 4u + (unsigned)wid;
        TileEntry mine = tl[min(e0 + (unsigned)lane * stride, a.subcap - 1)];
        if (done_word != 0) return false;
        uint2 *pairq = pairq_all + wid * PAIR_QUEUE;
        const PairSrc src = pin_pair_src<PIPE>(a);   // (the clouds' addresses and the transform switches in scalar registers)
        int qn = 0;   // wave-uniform: queued pairs
        unsigned nk = 0;   // wave-uniform: members of A recorded so far
        unsigned co = 0;   // wave-uniform (REC): candidates recorded so far
        const size_t kbase = (size_t)wave * a.kept_wcap;
        // evaluate the queued pairs q[base .. base+cnt) (cnt <= 64, wave-uniform)
        auto run_batch = [&](int base, int cnt) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            float w = 0.0f;
            uint2 pr = make_uint2(0u, 0u);
            float ck = 0.0f;
            if (lane < cnt) {
                pr = pairq[base + lane];
                w = eval_pair<MODE, WEIGHT, REC>(src, hd, kc, pr.x, pr.y, 0.0f, acc, *hd.xi, s_etab, first_counted, &ck);   // (xi: PROC_STEP only)
            }
            if (REC) {   // every candidate goes on record, at the place the wave met it (its colour weight with it)
                if (co + (unsigned)cnt <= a.kept_wcap) {
                    if (lane < cnt) hd.cand[kbase + co + lane] = make_uint2(pr.x | (pr.y << 16), __float_as_uint(ck));
                } else if (lane == 0) {
                    atomicOr(&a.st->ovf[hd.par][LIST_KEPT], 1u);   // slice full: grow and redo
                }
                co += (unsigned)cnt;
            }
            if (MODE == PROC_SELF) nk += (unsigned)__popcll(__ballot(w > 0.0f));   // (scalar: the member count)
            if (MODE == PROC_FLOW) {   // record the members of A in this wave's slice
                const unsigned long long km = __ballot(w > 0.0f);
                const unsigned add = (unsigned)__popcll(km);
                if (nk + add <= a.kept_wcap) {
#ifndef CVO_PROBE_NO_KEPT_STORE   // (probe builds: the expansion pass without its kept-list store, profiles/r06_ab.txt 6)
                    if (w > 0.0f) {
#else
                    if (false) {
#endif
                        const unsigned below = __builtin_amdgcn_mbcnt_hi(
                            (unsigned)(km >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)km, 0u));
                        if (a.kept_packed) {
                            a.kept_ij[kbase + nk + below] = kept_pack(a.kept_packed, a.kept_ebase, pr.x, pr.y, w);
                        } else {
                            a.kept_ij[kbase + nk + below] = pr;
                            a.kept_a[kbase + nk + below] = w;
                        }
                    }
                } else if (lane == 0) {
                    atomicOr(&a.st->ovf[hd.par][LIST_KEPT], 1u);   // slice full: grow and redo
                }
                nk += add;
            }
            __builtin_amdgcn_wave_barrier();
        };
        for (unsigned sl = 0; sl < nsl; ++sl) {
        // the next sub-list's size and first entries are requested before this one is worked on
        unsigned n_next = 0;
        TileEntry mine_next = mine;
        const TileEntry *tl_next = tl;
        if (sl + 1 < nsl) {
            const unsigned sub_next = sub + (unsigned)a.nblk;
            n_next = list_bad ? 0u : a.st->sub[in_list][sub_next];
            tl_next = in_tiles + (size_t)sub_next * a.subcap;
            mine_next = tl_next[min(e0 + (unsigned)lane * stride, a.subcap - 1)];
        }
        if (n > a.subcap) n = a.subcap;   // overflowed list: the iteration is redone anyway
#ifdef CVO_NO_ENTRY_PF   // (A/B builds)
        constexpr bool EPF = false;
#else
        constexpr bool EPF = PIPE;   // the next 64 entries are requested while these 64 are expanded and evaluated
#endif
        TileEntry ahead = mine;
        for (unsigned eb = e0; eb < n; eb += 64u * stride) {
            if (eb != e0) mine = EPF ? ahead : tl[min(eb + (unsigned)lane * stride, a.subcap - 1)];
            if (EPF && eb + 64u * stride < n) ahead = tl[min(eb + (64u + (unsigned)lane) * stride, a.subcap - 1)];
            const unsigned left = (n - eb + stride - 1) / stride;   // entries of this round
            const int cnt = (int)(left < 64u ? left : 64u);
            for (int k = 0; k < cnt; ++k) {
                // broadcast lane k's entry and expand its mask into the queue:
                // bit l is row (l>>4)*4 + r, column l&15 of the tile
                const unsigned tx = (unsigned)__builtin_amdgcn_readlane((int)mine.x, k);
                const unsigned ty = (unsigned)__builtin_amdgcn_readlane((int)mine.y, k);
                const unsigned long long m =
                    (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)mine.z, k) |
                    ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)mine.w, k) << 32);
                const unsigned r = ty >> 30, tcol = ty & 0x3fffffffu;
                if (__builtin_amdgcn_inverse_ballot_w64(m)) {   // the mask IS the set of lanes that act: no per-lane test
                    const unsigned below = __builtin_amdgcn_mbcnt_hi(
                        (unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                    pairq[qn + below] =
                        make_uint2(tx + (unsigned)(lane >> 4) * 4u + r, tcol + (unsigned)(lane & 15));
                }
                qn += __popcll(m);
                if (qn >= 64) {
                    qn -= 64;
                    run_batch(qn, 64);
                }
            }
        }
        sub += (unsigned)a.nblk; n = n_next; tl = tl_next; mine = mine_next;
        }   // sub-lists of this block
        if (qn > 0) run_batch(0, qn);
        if (MODE == PROC_FLOW && lane == 0) a.kept_cnt[wave] = nk;
        if (REC && lane == 0) hd.cand_cnt[wave] = co;
        // the member count of the wave joins the sums (one lane holds it; a slice that overflowed still counts)
        if (lane == 0) acc[MODE == PROC_FLOW ? 8 : 1] = (double)nk;
        return true;
}

// The candidate list of one registration, streamed by the wave that recorded it: lane l takes the
// wave's l-th candidate of the round -- nothing to expand, full rounds but the last, the colour weight
// read back with the pair.  PROC_FLOW: the members of A of THIS iteration go to the kept list as always.
template <int MODE, bool PIPE>
__device__ __forceinline__ bool stream_candidates(const ProcessArgs &a, const ProcHead &hd, const KernConsts &kc, const int lane,
                                                  const unsigned wave, const int done_word, const double *s_etab,
                                                  double (&acc)[NAcc<MODE>::n])
{
    unsigned wcap = a.kept_wcap;
    const size_t base = (size_t)wave * wcap;
    unsigned n = hd.cand_cnt[wave];
    uint2 e = list_load(&hd.cand[base + lane]);
    if (done_word != 0) return false;
    if (n > wcap) n = wcap;
    unsigned nk = 0;
    // (the loop's addresses and switches in scalar registers: PairSrc)
    const PairSrc src = pin_pair_src<PIPE>(a);
    const CVO_GLOBAL char *cand_w = pin_global<PIPE>(hd.cand + base);
    const CVO_GLOBAL char *kept_w = pin_global<PIPE>(a.kept_ij + base);
    wcap = pin_u32<PIPE>(wcap);
    constexpr bool PF = PIPE && kPrefetchBuild;   // the next round's records are requested behind this round's gathers
    for (unsigned b0 = 0; b0 < n; b0 += 64u) {
        if (!PF && b0 != 0u) e = load8(cand_w, min(b0 + (unsigned)lane, wcap - 1u));
        uint2 e_next = e;
        uint2 *const pf = PF ? &e_next : nullptr;
        const unsigned ci = e.x & 0xffffu, cj = e.x >> 16;
        float w = 0.0f;
        if (b0 + (unsigned)lane < n) {
            float ck = __uint_as_float(e.y);
            w = eval_pair<MODE, 0, 2>(src, hd, kc, ci, cj, 0.0f, acc, *hd.xi, s_etab, 0, &ck, pf, cand_w,
                                      min(b0 + 64u + (unsigned)lane, wcap - 1u));
        }
        const unsigned long long km = __ballot(w > 0.0f);
        if (MODE == PROC_FLOW && w > 0.0f) {   // (members <= candidates <= the slice: it cannot overflow here)
            const unsigned below = __builtin_amdgcn_mbcnt_hi((unsigned)(km >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)km, 0u));
            // (candidate records exist for clouds of up to 65536 rows only, whose kept entries are packed the same
            // way: the record's first word IS the entry's)
#ifndef CVO_PROBE_NO_KEPT_STORE   // (probe builds, profiles/r05_ab.txt 9: what the flow pass would gain if it kept no list at all)
            store8(kept_w, nk + below, e.x, __float_as_uint(w));
#endif
        }
        nk += (unsigned)__popcll(km);
        if (PF) e = e_next;
    }
    if (lane == 0) {
#if defined(CVO_PROBE_NO_KEPT_STORE) || defined(CVO_PROBE_KEPT_CNT0)
        if (MODE == PROC_FLOW) a.kept_cnt[wave] = 0u;   // (the step pass of the probe reads nothing)
#else
        if (MODE == PROC_FLOW) a.kept_cnt[wave] = nk;
#endif
        acc[MODE == PROC_FLOW ? 8 : 1] = (double)nk;
    }
    return true;
}

// CAND false: the launch never keeps a candidate list (the merged launches of one registration on its
// own, whose xy list is built beside the pass): that code is left out of the kernel
template <int MODE, int WEIGHT = 0, bool CAND = true, bool PIPE = true>
__device__ __forceinline__ void process_body(const ProcessArgs &a, const unsigned bid, char *scratch, const ProcHead &hd)
{
    if ((int)bid >= a.nblk) return;
    constexpr int NACC = NAcc<MODE>::n;
    double *red = reinterpret_cast<double *>(scratch);
    uint2 *pairq_all = reinterpret_cast<uint2 *>(scratch + 4 * NACC_MAX * 8);
    // every wave keeps its own copy of the exp table (no block barrier needed)
    double *s_etab_all = reinterpret_cast<double *>(scratch + 4 * NACC_MAX * 8 + 4 * PAIR_QUEUE * 8);
#ifdef CVO_PROBE_STEP_EXP
    s_etab_all[threadIdx.x] = c_exp2_64[threadIdx.x & 63];
#else
    if (MODE != PROC_STEP) s_etab_all[threadIdx.x] = c_exp2_64[threadIdx.x & 63];
#endif
    const double *s_etab = s_etab_all + ((MODE == PROC_STEP) ? 0 : (threadIdx.x >> 6) * 64);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned wave = bid * 4u + (unsigned)wid;   // 0 .. PROC_WAVES-1
    // first round trip: loop-control word, kernel constants, list sizes
    const int done_word = hd.done_word;
    const KernConsts kc = hd.kc;
    const int first_counted = hd.n_fixed;
    const bool second = hd.second != 0;
    const int in_list = !second ? a.list : (MODE == PROC_FLOW ? (int)LIST_XYB : self_list_id(a.async_self - 1, 1));
    const TileEntry *in_tiles = second ? a.tiles_b : a.tiles;
    const unsigned list_bad = hd.list_bad;

    double acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = 0.0;

    if (MODE == PROC_STEP) {
        // stream the members of A that wave `wave` of PROC_FLOW recorded
        const size_t base = (size_t)wave * a.kept_wcap;
        unsigned n = a.kept_cnt[wave];
        // the first entries are fetched together with the count
        const int packed = a.kept_packed;
        uint2 e = list_load(&a.kept_ij[base + lane]);
        float w = packed ? 0.0f : a.kept_a[base + lane];
        if (done_word != 0) return;
        if (n > a.kept_wcap) n = a.kept_wcap;
        if (list_bad) n = 0;
        const PairSrc src = pin_pair_src<PIPE>(a);
        const cvo_math::XiConsts xc = xi_uniform(*hd.xi);
        const CVO_GLOBAL char *kept_w = pin_global<PIPE>(a.kept_ij + base);
        const unsigned ebase = pin_u32<PIPE>(a.kept_ebase), wcap = pin_u32<PIPE>(a.kept_wcap);
        constexpr bool PF = PIPE && kPrefetchBuild;
#if defined(CVO_STEP_TWO) && !defined(CVO_PROBE_STEP_EXP)
        // (A/B builds: TWO members per lane and trip -- entries off and off + 64, the ones the lane takes in two trips of one --, the float32
        // part of their terms through packed instructions, their float64 tails in the old order: the same partial sums bit for bit)
        step_members2<PF>(src, hd, kc, xc, kept_w, a.kept_a + base, packed, ebase, wcap, n, lane, e, w, acc);
        if (false)
#endif
        for (unsigned off = lane; off < n; off += 64) {
            if (off >= 64) { if (!PF) e = load8(kept_w, off); if (!packed) w = a.kept_a[base + off]; }
            uint2 e_next = e;
            uint2 *const pf = PF ? &e_next : nullptr;
            unsigned mi, mj;
            float mw;
            kept_unpack(packed, ebase, e, w, mi, mj, mw);
#ifdef CVO_PROBE_STEP_EXP
            eval_pair<MODE>(src, hd, kc, mi, mj, mw, acc, xc, s_etab, 0, nullptr, pf, kept_w, min(off + 64u, wcap - 1u));
#else
            eval_pair<MODE>(src, hd, kc, mi, mj, mw, acc, xc, nullptr, 0, nullptr, pf, kept_w, min(off + 64u, wcap - 1u));
#endif
            if (PF) e = e_next;
        }
    } else {
        bool alive;
        // (lists built ahead: only the xy list of a head-mode plan keeps records, one per buffer -- cand_b)
        if (CAND && WEIGHT == 0 && hd.cand && (MODE == PROC_FLOW ? (!a.async_xy || a.cand_b != nullptr)
                                                                 : (!a.async_self || a.cand_b != nullptr))) {
            if (hd.ck_nblk == a.nblk) alive = stream_candidates<MODE, PIPE>(a, hd, kc, lane, wave, done_word, s_etab, acc);
            else alive = expand_lists<MODE, WEIGHT, WEIGHT == 0 ? 1 : 0, PIPE>(a, hd, kc, bid, wid, lane, wave, done_word, list_bad, in_list, in_tiles, first_counted, s_etab, pairq_all, acc);
        } else {
            alive = expand_lists<MODE, WEIGHT, 0, PIPE>(a, hd, kc, bid, wid, lane, wave, done_word, list_bad, in_list, in_tiles, first_counted, s_etab, pairq_all, acc);
        }
        if (!alive) return;
    }

    // block reduction: reduce-scatter inside each wave, then the 4 waves in order
    wave_sums<NACC>(acc, lane, red + wid * NACC);
    __syncthreads();
    if (tid < NACC) {
        const double s = ((red[tid] + red[NACC + tid]) + red[2 * NACC + tid]) + red[3 * NACC + tid];
        a.partials[(size_t)tid * a.nblk + bid] = s;   // [value][block]: coalesced for the readers
    }
}

template <int MODE, int WEIGHT = 0>
__global__ void __launch_bounds__(BLOCK) k_process(const Grp<ProcessArgs> grp)
{
    __shared__ __attribute__((aligned(16))) char scratch[PROC_SMEM];
    const ProcessArgs &a = grp.a[blockIdx.z];
    process_body<MODE, WEIGHT>(a, blockIdx.x, scratch, proc_head_global<MODE>(a, a.st, 0));
}

// ---------------------------------------------------------------------------
// k_step_twist = the tail of compute_flow (what k_post_flow does) + PROC_STEP in one
// launch: every block reduces the PROC_FLOW partial sums itself -- 1024 threads, one
// partial row each, the same fixed order in every block, so all blocks hold the same
// twist -- and goes on to stream its slices of the kept list.  One launch, one kernel
// boundary and one single-block bubble less per iteration.  Block 0 also leaves the
// twist, dl and the trace record in the state for k_post_step.  With ranks to sum over whose
// exchange runs through the mailboxes (ProcessArgs::comm) the flow-side sums are exchanged inside the
// launch, by every block (round 5); the stream-level all-reduces keep k_post_flow and PROC_STEP apart.
constexpr int STEP_BLOCK = 1024;
constexpr int STEP_WAVES = STEP_BLOCK / 64;

// cvo_math::make_xi_consts (se3_math.hpp; ref src/cvo.cpp:226-238) by the first twelve lanes of a wave: every element of W^2, W^3, W^4
// and of W v, W^2 v, W^3 v with the expression the serial routine has for it (mul / mulv: (a0 b0 + a1 b1) + a2 b2, no contraction), three
// dependent levels through LDS instead of ~200 instructions of one lane in a row.  xi and wm[9] are in LDS; all 64 lanes call it.
__device__ __forceinline__ void xi_consts_wave(cvo_math::XiConsts *xi, float *wm, const float omega[3], const float v[3], const int lane)
{
    if (lane == 0) {
        const cvo_math::Mat3 W = cvo_math::skew(omega);
#pragma unroll
        for (int k = 0; k < 9; ++k) wm[k] = W.m[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) { xi->omega[k] = omega[k]; xi->v[k] = v[k]; }
    }
    const int r = lane < 9 ? lane / 3 : lane - 9, c = lane < 9 ? lane - 3 * (lane / 3) : 0;
#pragma unroll
    for (int level = 1; level <= 3; ++level) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < 12) {
            const float *A = level == 1 ? wm : (level == 2 ? xi->W2 : xi->W3);
            const float a0 = A[3 * r], a1 = A[3 * r + 1], a2 = A[3 * r + 2];
            const float b0 = lane < 9 ? wm[c] : xi->v[0], b1 = lane < 9 ? wm[3 + c] : xi->v[1], b2 = lane < 9 ? wm[6 + c] : xi->v[2];
            const float o = (a0 * b0 + a1 * b1) + a2 * b2;
            float *dst = level == 1 ? (lane < 9 ? xi->W2 + lane : xi->u2 + r)
                                    : (level == 2 ? (lane < 9 ? xi->W3 + lane : xi->u3 + r) : (lane < 9 ? xi->W4 + lane : xi->u4 + r));
            *dst = o;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ bool mailbox_allreduce_every_block(const CommTable &ct, DevState *gst, const DevHead *hd, double *vals, int count, double *sh, int *sh_fail);

// returns true if this block delivered a row of step partial sums (false: the registration has
// stopped, the slot is a stall, a list overflowed, or the block is surplus)
// hd: the copy of the state's head the slot runs on (a.st unless head mode); par: the slot's parity
__device__ __forceinline__ bool step_twist_body(const ProcessArgs &a, DevState *hd, const int par, const bool head_mode)
{
    const int nfat = a.nblk / (STEP_BLOCK / BLOCK);   // blocks of this registration
    if ((int)blockIdx.x >= nfat) return false;
    constexpr int NACC = NACC_STEP;
    __shared__ double sh[STEP_WAVES * NACC_MAX];
    __shared__ double tot[NACC_MAX + 4];
    __shared__ cvo_math::XiConsts s_xi;
    __shared__ float s_wm[12];
    __shared__ int s_overflow;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // this wave streams the kept slice of one PROC_FLOW wave; the 16 of a block are
    // taken from PROC_FLOW blocks that ran on the same XCD (block id mod 8), whose
    // rows and columns belong to one region of the clouds (see k_filter)
    const unsigned fb = (blockIdx.x & 7u) + 8u * ((blockIdx.x >> 3) * 4u + ((unsigned)wid >> 2));
    const unsigned wave = fb * 4u + ((unsigned)wid & 3u);
    // first round trip: loop control, constants, this thread's PROC_FLOW partial row,
    // the wave's first kept entries
    const int done_word = a.check_done ? (hd->done | (a.async_xy ? hd->stall : 0)) : 0;
    const KernConsts kc = hd->kc;
    // (async xy: the buffer being built beside this launch is the next plan step's business)
    const unsigned ovf = (a.async_xy ? 0u : a.st->ovf[par][LIST_XY]) | a.st->ovf[par][LIST_XX] |
                         a.st->ovf[par][LIST_YY] | a.st->ovf[par][LIST_KEPT];
    // head mode: the flags of the other parity -- raised by the flow launch of the previous slot, read
    // by that slot's step launch and by the head of this slot's flow launch -- are cleared here, before
    // the next flow launch raises them again (not once the loop has stopped: the host wants to see them)
    if (head_mode && blockIdx.x == 0 && tid < 8 && hd->done == 0) a.st->ovf[par ^ 1][tid] = 0u;
    double pf[NACC_FLOW];
#pragma unroll
    for (int k = 0; k < NACC_FLOW; ++k)
        pf[k] = (tid < a.nblk) ? a.flow_part[(size_t)k * a.nblk + tid] : 0.0;
    const size_t base = (size_t)wave * a.kept_wcap;
    unsigned n = a.kept_cnt[wave];
    const int packed = a.kept_packed;
    uint2 e = a.kept_ij[base + lane];
    float w = packed ? 0.0f : a.kept_a[base + lane];
    if (done_word != 0) return false;

    // ---- the twist (ref cvo.cpp:201-209) from the partial sums
    wave_sums<NACC_FLOW>(pf, lane, sh + wid * NACC_MAX);
    __syncthreads();
    if (tid < NACC_FLOW) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < STEP_WAVES; ++q) t += sh[q * NACC_MAX + tid];
        tot[tid] = t;
    } else if (tid < NACC_FLOW + 4) {
        tot[tid] = 0.0;
    }
    __syncthreads();
    // acvo: block 0 also needs the Axx / Ayy sums for dl (ref adaptive_cvo.cpp:222-231,271)
    if (blockIdx.x == 0 && a.acvo && wid < 2) {
        const double *part = wid == 0 ? a.xx_part : a.yy_part;
        double ps[NACC_SELF] = {0.0, 0.0};
        for (int b = lane; b < a.nblk; b += 64) {
            ps[0] += part[b];
            ps[1] += part[(size_t)a.nblk + b];
        }
        wave_sums<NACC_SELF>(ps, lane, tot + NACC_FLOW + 2 * wid);   // tot[9..10] xx, tot[11..12] yy
    }
    if (a.comm) {
        // With ranks to sum over (the target cloud is split, ref src/cvo.cpp:201-204 across ranks): this rank's 13 flow-side sums
        // travel through the mailboxes INSIDE the launch -- block 0 sends, every block reads its rank's own mailbox and adds in
        // rank order -- and the step pass goes on in the same launch: no post-flow launch, no single-block bubble.  A list that
        // overflowed on this rank poisons nnz before the sums travel, so that every block of every rank takes the same exit.
        __shared__ double sh_mail[MAX_WORLD * MAIL_VALS];
        __shared__ int sh_fail;
        __syncthreads();
        if (blockIdx.x == 0 && tid == 0 && ovf != 0u) tot[8] = __builtin_nan("");
        const bool comm_ok = mailbox_allreduce_every_block(*a.comm, a.st, hd, tot, RED_STEP - RED_FLOW, sh_mail, &sh_fail);
        if (!comm_ok) {
            // (ANY block: the limit is per block, with a clock of its own -- a block other than 0 that gives up while block 0 still
            // gets its answer writes no step row, and the next head must not reduce a stale one without knowing; every writer stores
            // the same value)
            if (tid == 0) {
                __hip_atomic_store(&hd->done, (int32_t)DONE_COMM_ERROR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (a.done_mirror) *a.done_mirror = DONE_COMM_ERROR;
            }
            return false;
        }
    }
    if (tid < 64) {   // (the Taylor constants by the first wave, element by element: xi_consts_wave)
        float omega[3], v[3];
        for (int q = 0; q < 3; ++q) { omega[q] = (float)tot[q]; v[q] = (float)tot[3 + q]; }
        xi_consts_wave(&s_xi, s_wm, omega, v, lane);
        if (tid == 0) s_overflow = a.comm ? (tot[8] != tot[8]) : (ovf != 0u);
    }
    __syncthreads();
    const bool overflow = s_overflow != 0;
    if (blockIdx.x == 0 && tid == 0) {
        DevState *st = hd;
        // nothing of an overflowed iteration is usable; the host enlarges the list
        // and resumes from the same (untouched) state
        if (overflow) {
            st->done = NEED_BIGGER_LIST;
            if (a.done_mirror) *a.done_mirror = NEED_BIGGER_LIST;
        } else {
            for (int q = 0; q < NACC_FLOW; ++q) st->red[RED_FLOW + q] = tot[q];
            for (int q = 0; q < 4; ++q) st->red[RED_XX + q] = a.acvo ? tot[NACC_FLOW + q] : 0.0;
            for (int q = 0; q < 3; ++q) { st->omega[q] = s_xi.omega[q]; st->v[q] = s_xi.v[q]; }
            st->xi = s_xi;
            double dl = 0.0;
            const long long nnz = (long long)tot[8];
            long long nnz_xx = 0, nnz_yy = 0;
            if (a.acvo) {
                nnz_xx = (long long)tot[NACC_FLOW + 1];
                nnz_yy = (long long)tot[NACC_FLOW + 3];
                const double num = (tot[NACC_FLOW + 2] - 2.0 * tot[7]) + tot[NACC_FLOW];
                dl = num / (double)(nnz_xx + nnz_yy - 2 * nnz);
            }
            st->dl = dl;
            if (a.trace && st->k < a.trace_cap) {
                cvo_hip_trace &tr = a.trace[st->k];
                tr.k = st->k;
                tr.exit_code = 0;
                tr.ell = st->ell;
                for (int q = 0; q < 3; ++q) {
                    tr.omega[q] = s_xi.omega[q]; tr.v[q] = s_xi.v[q];
                    tr.omega_d[q] = tot[q]; tr.v_d[q] = tot[3 + q];
                }
                tr.sum_a = tot[6];
                tr.dl = dl;
                tr.nnz = nnz; tr.nnz_xx = nnz_xx; tr.nnz_yy = nnz_yy;
            }
        }
    }
    if (overflow) return false;
    // the constants are the same for every lane: keep them in scalar registers
    cvo_math::XiConsts xc;
    {
        auto uni = [](float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); };
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            xc.omega[q] = uni(s_xi.omega[q]); xc.v[q] = uni(s_xi.v[q]);
            xc.u2[q] = uni(s_xi.u2[q]); xc.u3[q] = uni(s_xi.u3[q]); xc.u4[q] = uni(s_xi.u4[q]);
        }
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            xc.W2[q] = uni(s_xi.W2[q]); xc.W3[q] = uni(s_xi.W3[q]); xc.W4[q] = uni(s_xi.W4[q]);
        }
    }

    // ---- compute_step_size sums over this wave's slice of the kept list
    ProcHead phd;
    phd.need_d2 = 0;
    phd.Rt = hd->Rt; phd.tt = hd->t;   // (eval_pair<PROC_STEP> reads nothing else of it)
    double acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = 0.0;
    unsigned wcap = a.kept_wcap, ebase = a.kept_ebase;
    if (n > wcap) n = wcap;
    // (addresses and switches of the loop in scalar registers, the next entries requested behind the gathers: PairSrc)
    const PairSrc src = pin_pair_src<true>(a);
    const CVO_GLOBAL char *kept_w = pin_global<true>(a.kept_ij + base);
    wcap = pin_u32<true>(wcap); ebase = pin_u32<true>(ebase);
    constexpr bool PF = kPrefetchBuild;
    for (unsigned off = lane; off < n; off += 64) {
        if (off >= 64) { if (!PF) e = load8(kept_w, off); if (!packed) w = a.kept_a[base + off]; }
        uint2 e_next = e;
        uint2 *const pf = PF ? &e_next : nullptr;
        unsigned mi, mj;
        float mw;
        kept_unpack(packed, ebase, e, w, mi, mj, mw);
        eval_pair<PROC_STEP>(src, phd, kc, mi, mj, mw, acc, xc, nullptr, 0, nullptr, pf, kept_w, min(off + 64u, wcap - 1u));
        if (PF) e = e_next;
    }
    __syncthreads();   // sh is re-used
    wave_sums<NACC>(acc, lane, sh + wid * NACC);
    __syncthreads();
    if (tid < NACC) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < STEP_WAVES; ++q) t += sh[q * NACC + tid];
        a.partials[(size_t)tid * nfat + blockIdx.x] = t;
    }
    return true;
}

__global__ void __launch_bounds__(STEP_BLOCK) k_step_twist(const Grp<ProcessArgs> grp)
{
    step_twist_body(grp.a[blockIdx.z], grp.a[blockIdx.z].st, 0, false);
}

void launch_step_twist_group(const ProcessArgs *a, int n, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop)
{
    Grp<ProcessArgs> g;
    int nblk = 32;
    for (int i = 0; i < n; ++i) { g.a[i] = a[i]; nblk = std::max(nblk, a[i].nblk); }
    const dim3 grid((unsigned)(nblk / (STEP_BLOCK / BLOCK)), 1, (unsigned)n);
    if (ev_start && ev_stop) hipExtLaunchKernelGGL(k_step_twist, grid, dim3(STEP_BLOCK), 0, s, ev_start, ev_stop, 0, g);
    else hipLaunchKernelGGL(k_step_twist, grid, dim3(STEP_BLOCK), 0, s, g);
}

void launch_process_group(int mode, const ProcessArgs *a, int n, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop)
{
    Grp<ProcessArgs> g;
    for (int i = 0; i < n; ++i) g.a[i] = a[i];
    int nblk = 1;
    for (int i = 0; i < n; ++i) nblk = std::max(nblk, a[i].nblk);
    const dim3 grid((unsigned)nblk, 1, (unsigned)n);
    if (ev_start && ev_stop) {   // profiling: the events take the dispatch's own begin / end timestamps
        if (mode == PROC_FLOW && a[0].weight == 1)
            hipExtLaunchKernelGGL((k_process<PROC_FLOW, 1>), grid, dim3(BLOCK), 0, s, ev_start, ev_stop, 0, g);
        else if (mode == PROC_FLOW)
            hipExtLaunchKernelGGL(k_process<PROC_FLOW>, grid, dim3(BLOCK), 0, s, ev_start, ev_stop, 0, g);
        else if (mode == PROC_STEP)
            hipExtLaunchKernelGGL(k_process<PROC_STEP>, grid, dim3(BLOCK), 0, s, ev_start, ev_stop, 0, g);
        else
            hipExtLaunchKernelGGL(k_process<PROC_SELF>, grid, dim3(BLOCK), 0, s, ev_start, ev_stop, 0, g);
        return;
    }
    switch (mode) {
    case PROC_FLOW:
        if (a[0].weight == 1) hipLaunchKernelGGL((k_process<PROC_FLOW, 1>), grid, dim3(BLOCK), 0, s, g);
        else hipLaunchKernelGGL(k_process<PROC_FLOW>, grid, dim3(BLOCK), 0, s, g);
        break;
    case PROC_STEP:
        hipLaunchKernelGGL(k_process<PROC_STEP>, grid, dim3(BLOCK), 0, s, g);
        break;
    default:
        hipLaunchKernelGGL(k_process<PROC_SELF>, grid, dim3(BLOCK), 0, s, g);
        break;
    }
}

void launch_process(int mode, const ProcessArgs &a, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop)
{
    launch_process_group(mode, &a, 1, s, ev_start, ev_stop);
}

// ---------------------------------------------------------------------------
// Fixed-order reduction of partials[NACC][nblocks] by one 256-thread block:
// thread t adds blocks t, t+256, ...; xor butterfly inside each wave; the four
// wave sums are added in wave order.
// (two halves, so that a caller can have the loads in flight while it waits for
// something else: thread_load_partials only issues them and adds in a fixed order)
template <int NACC, int NPART = PROC_BLOCKS>
__device__ __forceinline__ void thread_load_partials(const double *part, int nblocks, double (&s)[NACC])
{
    const int tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < NACC; ++k) s[k] = 0.0;
    // nblocks == PROC_BLOCKS: a fixed trip count, fully unrolled, so that all the
    // loads are in flight together (one memory round trip, not one per step)
    static_assert(NPART % BLOCK == 0, "partials per thread must be whole");
#pragma unroll
    for (int u = 0; u < NPART / BLOCK; ++u) {
        const int b = tid + u * BLOCK;
#pragma unroll
        for (int k = 0; k < NACC; ++k) s[k] += (b < nblocks) ? part[(size_t)k * nblocks + b] : 0.0;
    }
}

template <int NACC>
__device__ __forceinline__ void block_finish_partials(double (&s)[NACC], double *sh /*[4*NACC_MAX]*/,
                                                      double *out /*[NACC], thread 0 writes*/)
{
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    wave_sums<NACC>(s, lane, sh + wid * NACC_MAX);
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < NACC; ++k)
            out[k] = ((sh[k] + sh[NACC_MAX + k]) + sh[2 * NACC_MAX + k]) + sh[3 * NACC_MAX + k];
    }
    __syncthreads();
}

template <int NACC, int NPART = PROC_BLOCKS>
__device__ void block_reduce_partials(const double *part, int nblocks, double *sh /*[4*NACC_MAX]*/,
                                      double *out /*[NACC], thread 0 writes*/)
{
    double s[NACC];
    thread_load_partials<NACC, NPART>(part, nblocks, s);
    block_finish_partials<NACC>(s, sh, out);
}

// ---------------------------------------------------------------------------
// The post-step part of an iteration and the plan of the next one (ref src/cvo.cpp:291-307,380-410,
// src/adaptive_cvo.cpp:509-545, src/LieGroup.cpp:159-186) -- "the head".
//
// One block: the state's head moves HBM -> LDS cooperatively (one round trip, together with the step
// partials and the overflow flags), wave 0 takes a PRIVATE copy of it -- registers: the whole O(1)
// chain then runs without a memory access; with the state in LDS every field of it was a dependent
// ds_read / ds_write round trip on one lane -- runs the maths with all 64 lanes (the cubic uses them;
// the rest is the same value in every lane), lane 0 puts the head back into LDS, and the block writes it
// out.  Three forms:
//   HM_CLASSIC  k_post_step / kt_post_step, a launch of its own, in place (fused groups, sharded and
//               large registrations, the low-level entry points);
//   HM_HEAD     "head mode" (one registration with its launches to itself): there is no post-step
//               launch.  EVERY flow / self block of the flow launch of slot s + 1 starts with this body:
//               it reduces the step partials of slot s in the same fixed order, runs the same maths on the
//               same inputs -- all blocks hold the same new head, bit for bit -- and goes on with the
//               flow pass from its own LDS copy; block 0 alone publishes the head, the trace and the
//               host's mirrors.  Nothing is fenced and nobody waits for anybody (the ticket tail of round 2
//               paid an L2 write-back per block): two dependent launches per iteration instead of three.
//               The launch reads one copy of the head and block 0 writes the OTHER one (a block that
//               starts late must still find the old head): FB_s reads copy s & 1, writes copy ~s & 1;
//               the step launch of slot s works on the copy FB_s wrote.  Batches have an even number of
//               slots, so the head is back in copy 0 (DevState itself) when a batch ends.
// The filter blocks that ride in a head-mode flow launch do not run the head: they build what the
// PREVIOUS plan named, at the transform it recorded (cvo_device.h plan_xy_async).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void state_head_to_lds(const DevHead *g, DevHead *l)
{
    const uint4 *src = reinterpret_cast<const uint4 *>(g);
    uint4 *dst = reinterpret_cast<uint4 *>(l);
    static_assert(DEVSTATE_HEAD_BYTES / 16 <= BLOCK, "one 16-byte piece per thread");
    if (threadIdx.x < (int)(DEVSTATE_HEAD_BYTES / 16)) dst[threadIdx.x] = src[threadIdx.x];
    __syncthreads();
}

__device__ __forceinline__ void state_head_from_lds(const DevHead *l, DevHead *g)
{
    const uint4 *src = reinterpret_cast<const uint4 *>(l);
    uint4 *dst = reinterpret_cast<uint4 *>(g);
    if (threadIdx.x < (int)(DEVSTATE_HEAD_BYTES / 16)) dst[threadIdx.x] = src[threadIdx.x];
}

// cvo_math::section_root with one lane per interior point (all 64 lanes of a
// wave must call it with the same bracket): same arithmetic per point, same
// selection rule (lowest lane whose point is not left of the root).
__device__ __forceinline__ double readlane_f64(double x, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), l), hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double section_root_wave(const cvo_math::CubicBracket &B, int lane)
{
    double lo = B.lo, hi = B.hi;
    for (int round = 0; round < 64; ++round) {
        if ((float)lo == (float)hi) break;
        if (round == 1) {   // (wave-uniform: every lane holds the same bracket)
            double r;
            if (cvo_math::section_shortcut(B, lo, hi, &r)) return r;
        }
        const double w = hi - lo;
        const double x = cvo_math::section_point(lo, w, lane);
        const bool inside = x > lo && x < hi;
        const double f = cvo_math::cubic_eval(B.a, B.b, B.c, x);
        const bool go_right = inside && (B.increasing ? (f < 0.0) : (f > 0.0));
        const unsigned long long gm = __ballot(go_right), im = __ballot(inside);
        const int first = (~gm == 0ull) ? 64 : (int)__builtin_ctzll(~gm);   // serial loop's break index
        double nlo = lo, nhi = hi;
        // (`first` is wave-uniform: v_readlane instead of a trip through the LDS crossbar)
        if (first > 0) nlo = readlane_f64(x, first - 1);
        if (first < 64 && ((im >> first) & 1ull)) nhi = readlane_f64(x, first);
        if (nlo == lo && nhi == hi) break;
        lo = nlo;
        hi = nhi;
    }
    return hi;
}

// ---------------------------------------------------------------------------
// Mailbox all-reduce over the ranks (cvo_device.h: Mailbox, CommTable), called by
// a whole block between its reduction and its O(1) maths: vals[0..count) (LDS or global,
// written before the call) are replaced by their sums over all ranks, added in rank order.
// Lane r of wave 0 serves rank r: it stores this rank's values and then the sequence number
// into rank r's mailbox (system-scope stores: peer memory over xGMI), polls this rank's own
// slot of sender r and copies it out.  The poll is bounded (CommTable::timeout_ticks): a
// peer that never shows up ends the registration with DONE_COMM_ERROR instead of hanging
// the GPU.  Returns false on a time-out (block-uniform).
// `lds_head`: the block's copy of the state's head, which it writes back afterwards: the new sequence number is left there as well
// (DevHead::mail_snap: what a later k_step_twist launch numbers its own exchange from).
__device__ bool mailbox_allreduce(const CommTable &ct, DevState *gst, double *vals, int count,
                                  double *sh /*[MAX_WORLD * MAIL_VALS]*/, int *sh_fail, DevHead *lds_head)
{
    __syncthreads();
    const int lane = threadIdx.x;
    if (lane < 64) {
        const unsigned long long seq = gst->mail_seq + 1ull;
        const int gen = (int)(seq & 1ull);
        bool ok = true;
        if (lane < ct.world) {
            MailSlot *dst = &ct.peer[lane]->slot[gen][ct.rank];
            for (int i = 0; i < count; ++i)
                __hip_atomic_store(&dst->v[i], vals[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the flag must not overtake the write-back)
            __hip_atomic_store(&dst->seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const MailSlot *src = &ct.peer[ct.rank]->slot[gen][lane];
            const long long t0 = (long long)wall_clock64();
            while (__hip_atomic_load(&src->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
                if ((long long)wall_clock64() - t0 > ct.timeout_ticks) { ok = false; break; }
                __builtin_amdgcn_s_sleep(8);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
            for (int i = 0; i < count; ++i)
                sh[lane * MAIL_VALS + i] = __hip_atomic_load(&src->v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        const bool all_ok = __ballot(!ok) == 0ull;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < count && all_ok) {
            double t = 0.0;
            for (int r = 0; r < ct.world; ++r) t += sh[r * MAIL_VALS + lane];   // rank order, on every rank
            vals[lane] = t;
        }
        if (lane == 0) {
            gst->mail_seq = seq;
            lds_head->mail_snap = seq;
            *sh_fail = all_ok ? 0 : 1;
        }
    }
    __syncthreads();
    return *sh_fail == 0;
}

// The same exchange inside a launch whose EVERY block needs the sums (k_step_twist with ranks; the reductions of
// ref src/cvo.cpp:201-204 across ranks): block 0 alone sends this rank's values -- to every rank's mailbox, its own included --,
// every block polls its rank's OWN mailbox (local memory, system-scope loads) and adds the slots up in rank order: all blocks of all
// ranks hold the same bits.  The exchange's number comes from the head the launch runs on (DevHead::mail_snap + 1: constant while
// the launch runs; block 0 advances DevState::mail_seq, which nobody reads in this launch).  Two generations still do: a rank
// sends its next exchange from a later launch, after all its blocks have read this one, and the one after that only once every
// peer has answered the next one -- which a peer does from a launch behind the one whose blocks may still be reading.
// vals[0..count): LDS, this rank's values going in (block 0's are sent), the sums coming out.  All threads of the block call it.
__device__ bool mailbox_allreduce_every_block(const CommTable &ct, DevState *gst, const DevHead *hd, double *vals, int count,
                                              double *sh /*[MAX_WORLD * MAIL_VALS]*/, int *sh_fail)
{
    __syncthreads();
    const int lane = threadIdx.x;
    if (lane < 64) {
        const unsigned long long seq = hd->mail_snap + 1ull;
        const int gen = (int)(seq & 1ull);
        bool ok = true;
        if (lane < ct.world) {
            if (blockIdx.x == 0) {
                MailSlot *dst = &ct.peer[lane]->slot[gen][ct.rank];
                for (int i = 0; i < count; ++i)
                    __hip_atomic_store(&dst->v[i], vals[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the flag must not overtake the write-back)
                __hip_atomic_store(&dst->seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            const MailSlot *src = &ct.peer[ct.rank]->slot[gen][lane];
            const long long t0 = (long long)wall_clock64();
            while (__hip_atomic_load(&src->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
                if ((long long)wall_clock64() - t0 > ct.timeout_ticks) { ok = false; break; }
                __builtin_amdgcn_s_sleep(8);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
            for (int i = 0; i < count; ++i)
                sh[lane * MAIL_VALS + i] = __hip_atomic_load(&src->v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        const bool all_ok = __ballot(!ok) == 0ull;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < count && all_ok) {
            double t = 0.0;
            for (int r = 0; r < ct.world; ++r) t += sh[r * MAIL_VALS + lane];   // rank order, in every block of every rank
            vals[lane] = t;
        }
        if (lane == 0) {
            if (blockIdx.x == 0) gst->mail_seq = seq;
            *sh_fail = all_ok ? 0 : 1;
        }
    }
    __syncthreads();
    return *sh_fail == 0;
}

__device__ __forceinline__ void post_flow_body(const PostFlowArgs &a)
{
    __shared__ double sh[4 * NACC_MAX];
    __shared__ __attribute__((aligned(16))) DevHead s_st;
    // One round trip: every thread fetches a piece of the state's head into LDS;
    // the maths below runs on that copy and the head is written back at the end.
    state_head_to_lds(a.st, &s_st);
    DevHead *st = &s_st;
    if (a.check_done && (st->done != 0 || (a.prm.async_xy && st->stall))) return;
    const bool acvo = a.prm.mode == CVO_HIP_MODE_ACVO;
    if (a.flags & POST_REDUCE) {
        block_reduce_partials<NACC_FLOW>(a.part_flow, a.nblk, sh, st->red + RED_FLOW);
        if (acvo) {
            block_reduce_partials<NACC_SELF>(a.part_xx, a.nblk, sh, st->red + RED_XX);
            block_reduce_partials<NACC_SELF>(a.part_yy, a.nblk, sh, st->red + RED_YY);
        } else if (threadIdx.x == 0) {
            st->red[RED_XX] = st->red[RED_XX + 1] = st->red[RED_YY] = st->red[RED_YY + 1] = 0.0;
        }
        // a candidate list overflowed on this rank: poison nnz so that, after the
        // all-reduce, EVERY rank takes the same "grow the list and redo" exit
        if (threadIdx.x == 0 &&
            ((a.prm.async_xy ? 0u : a.st->ovf[0][LIST_XY]) | a.st->ovf[0][LIST_XX] |
             a.st->ovf[0][LIST_YY] | a.st->ovf[0][LIST_KEPT]))
            st->red[8] = __builtin_nan("");
    }
    bool comm_ok = true;
    if (a.comm) {
        __shared__ double sh_mail[MAX_WORLD * MAIL_VALS];
        __shared__ int sh_fail;
        comm_ok = mailbox_allreduce(*a.comm, a.st, st->red + RED_FLOW, RED_STEP - RED_FLOW, sh_mail, &sh_fail, st);
        if (!comm_ok && threadIdx.x == 0) st->done = DONE_COMM_ERROR;
    }
    if ((a.flags & POST_MATH) && comm_ok && threadIdx.x < 64) {   // (the Taylor constants by the first wave: xi_consts_wave)
        // (sh is free again -- the reductions above end with a block barrier --: the wave's scratch for W)
        const double *red = st->red;
        float omega[3], v[3];
        for (int q = 0; q < 3; ++q) { omega[q] = (float)red[q]; v[q] = (float)red[3 + q]; }
        if (!(red[8] != red[8])) xi_consts_wave(&st->xi, reinterpret_cast<float *>(sh), omega, v, (int)threadIdx.x);
    }
    if ((a.flags & POST_MATH) && comm_ok && threadIdx.x == 0) {
        // nothing of an overflowed iteration is usable; the host enlarges the
        // list and resumes from the same (untouched) state
        const bool overflow = st->red[8] != st->red[8];
        if (overflow) st->done = NEED_BIGGER_LIST;
        const double *red = st->red;
        float omega[3], v[3];
        for (int q = 0; q < 3; ++q) {
            omega[q] = (float)red[q];        // omega = double_omega.cast<float>()
            v[q] = (float)red[3 + q];
            st->omega[q] = omega[q];
            st->v[q] = v[q];
        }
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
        if (!overflow && a.trace && st->k < a.trace_cap) {
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
    __syncthreads();
    if (threadIdx.x == 0 && a.done_mirror && s_st.done != RUNNING) *a.done_mirror = s_st.done;
    state_head_from_lds(&s_st, a.st);
}

__global__ void __launch_bounds__(BLOCK) k_post_flow(const Grp<PostFlowArgs> grp)
{
    post_flow_body(grp.a[blockIdx.z]);
}

enum HeadMode { HM_CLASSIC = 0, HM_HEAD = 1 };

// The maths of the head on wave 0 (all 64 lanes, every value wave-uniform).  Two stages, so that the
// registers hold what the chain needs when it needs it: the post-step part runs on the few fields it
// touches (read from the LDS copy in one go); the plan then takes a private copy of the head's plan
// fields (a second batch of LDS reads, none of them on the chain before), runs in registers, and lane 0
// puts back what changed.  The large, rarely written fields (transform records, kernel constants) go
// straight to the LDS copy when they change (cvo_device.h `bulk`).
// `flag[l]`: overflow flag of list l in the row the finished builds were flagged in.
// `publisher`: this block writes the trace.
// head_post: the post-step part (ref src/cvo.cpp:291-307,380-410) of the slot that ended -- cubic, stop tests, Exp_SEK3, update,
// length scale -- on the head in LDS; leaves R, T, ell, the counters and `done` there.  `was_pending`: a slot is complete when the
// head that follows it has run (head mode counts it then).
template <int HM>
__device__ __forceinline__ void head_post(DevHead *lds, const PostStepArgs &a, const bool run_post, const bool publisher, const bool timed,
                                          long long (&clk)[4], const cvo_math::ExpPre *pre = nullptr /* the twist-only stage of Exp_SEK3, dist_se3 and
                                          the stop test where it was formed ahead (a resident run: behind its step exchange); null: formed here */)
{
    const DevParams &p = a.prm;
    const bool acvo = p.mode == CVO_HIP_MODE_ACVO;
    const bool lane0 = (threadIdx.x & 63) == 0;
    const bool was_pending = HM == HM_CLASSIC ? true : (lds->pending != 0);
    float R[9], T[3], Rt[9], t[3], omega[3], v[3];
#pragma unroll
    for (int q = 0; q < 9; ++q) { R[q] = lds->R[q]; Rt[q] = lds->Rt[q]; }
#pragma unroll
    for (int q = 0; q < 3; ++q) { T[q] = lds->T[q]; t[q] = lds->t[q]; omega[q] = lds->omega[q]; v[q] = lds->v[q]; }
    float ell = lds->ell, ell_max = lds->ell_max;
    const double dl = lds->dl;
    double bcde[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) bcde[q] = lds->red[RED_STEP + q];
    int k = lds->k, done = lds->done, iter = lds->iter, n_exec = lds->n_exec;
    const int n_slots = lds->n_slots + (was_pending ? 1 : 0);   // head mode: a slot is complete when the head that follows it has run
    if (run_post) {
        if (timed) clk[0] = (long long)__builtin_readcyclecounter();
        const cvo_math::CubicBracket cb = cvo_math::cubic_bracket(bcde);
        const float step = cvo_math::finish_step(
            cb.found, cb.found ? section_root_wave(cb, (int)(threadIdx.x & 63)) : 0.0, p.min_step);
        if (timed) clk[1] = (long long)__builtin_readcyclecounter();
        cvo_hip_trace *tr = (publisher && lane0 && a.trace && k < a.trace_cap) ? &a.trace[k] : nullptr;
        if (tr) {
            for (int q = 0; q < 4; ++q) tr->bcde[q] = bcde[q];
            tr->step = step;
            tr->dist = __builtin_nanf("");
        }
        n_exec = k + 1;
        if (lane0) {   // the transform this iteration used (what align() accumulates, ref cvo.cpp:413-415)
#pragma unroll
            for (int q = 0; q < 9; ++q) lds->used_Rt[q] = Rt[q];
#pragma unroll
            for (int q = 0; q < 3; ++q) lds->used_t[q] = t[q];
        }
        // break A: both twist norms below eps (ref cvo.cpp:380 float norms,
        // adaptive_cvo.cpp:509 double norms of the float vectors)
        const cvo_math::ExpPre P = pre ? *pre : cvo_math::exp_se3_pre(omega, v);
        bool brk;
        if (acvo) brk = P.nw_d < (double)p.eps && P.nv_d < (double)p.eps;
        else brk = P.nw < p.eps && P.nv < p.eps;
        if (brk) {
            iter = k;
            done = DONE_BREAK_A;
            if (tr) tr->exit_code = 1;
        } else {
            // integrate: T = R*dT + T ; R = R*dR  (ref cvo.cpp:391-399)
            float dR[9], dT[3], RdT[3];
            cvo_math::exp_se3_with(P, v, step, dR, dT);
            cvo_math::Mat3 Rm, dRm;
#pragma unroll
            for (int q = 0; q < 9; ++q) { Rm.m[q] = R[q]; dRm.m[q] = dR[q]; }
            cvo_math::mulv(Rm, dT, RdT);
#pragma unroll
            for (int q = 0; q < 3; ++q) T[q] = RdT[q] + T[q];
            const cvo_math::Mat3 Rn = cvo_math::mul(Rm, dRm);
#pragma unroll
            for (int q = 0; q < 9; ++q) R[q] = Rn.m[q];

            const float dist = cvo_math::dist_se3_with(P, step);
            if (tr) tr->dist = dist;
            if (dist < p.eps_2) {   // break B
                iter = k;
                done = DONE_BREAK_B;
                if (tr) tr->exit_code = 2;
            } else {
                // length-scale update
                if (acvo) {   // ref src/adaptive_cvo.cpp:538-545
                    ell = (float)((double)ell + p.dl_step * dl);
                    if (ell >= ell_max) {
                        ell = (float)(ell_max * 0.7);
                        ell_max = (float)(ell_max * 0.7);
                    }
                    ell = (ell < p.ell_min) ? p.ell_min : ell;
                } else {      // ref src/cvo.cpp:408-410
                    ell = (k > 2) ? (float)0.10 : ell;
                    ell = (k > 9) ? (float)0.06 : ell;
                    ell = (k > 19) ? (float)0.03 : ell;
                }
                if (k + 1 >= p.max_iter) done = DONE_MAX_ITER;   // `iter` keeps its stale value
                k = k + 1;
            }
        }
        if (timed) clk[2] = (long long)__builtin_readcyclecounter();
    }
    if (lane0) {
#pragma unroll
        for (int q = 0; q < 9; ++q) lds->R[q] = R[q];
#pragma unroll
        for (int q = 0; q < 3; ++q) lds->T[q] = T[q];
        lds->ell = ell; lds->ell_max = ell_max;
        lds->k = k; lds->iter = iter; lds->n_exec = n_exec; lds->n_slots = n_slots;
        lds->done = done;
        if (HM != HM_CLASSIC) lds->pending = 0;   // (head_plan raises it again if this launch starts the next slot)
    }
}

// head_plan: the plan of the slot that begins (cvo_device.h prepare_iteration) on a private copy of the head's plan fields
// (registers), lane 0 puts back what changed.  The large, rarely written fields (transform records, kernel constants) go
// straight to the LDS copy when they change (cvo_device.h `bulk`).  Follows head_post in the same wave (the loop must be running).
// `flag[l]`: overflow flag of list l in the row the finished builds were flagged in; `ran_post`: head_post executed a slot.
template <int HM>
__device__ __forceinline__ void head_plan(DevHead *lds, const PostStepArgs &a, const bool ran_post, const bool stalled,
                                          const unsigned (&flag)[LIST_N])
{
    const DevParams &p = a.prm;
    const bool async = p.async_xy != 0, aself = p.async_self != 0;
    const bool lane0 = (threadIdx.x & 63) == 0;
    // what the builds that have ended made of their lists (judged by the plan below; worked out here, so
    // that three verdicts travel down the chain instead of seven flags)
    PlanBuilds pb = HM == HM_CLASSIC ? plan_builds_classic(lds, false, false, false) : plan_builds_head(lds, false, false, false);
    pb.xy_failed = async && pb.xy_fresh >= 0 && (pb.xy_fresh ? flag[LIST_XYB] : flag[LIST_XY]) != 0u;
    // (classic, synchronous self lists: a stall slot runs no k_step_twist / k_post_flow, which is where
    // an overflow of the xx / yy lists is normally caught: lists built in a stall slot are checked here)
    pb.sf_failed[0] = aself ? (pb.sf_fresh[0] >= 0 && (pb.sf_fresh[0] ? flag[LIST_XXB] : flag[LIST_XX]) != 0u)
                            : (HM == HM_CLASSIC && stalled && flag[LIST_XX] != 0u);
    pb.sf_failed[1] = aself ? (pb.sf_fresh[1] >= 0 && (pb.sf_fresh[1] ? flag[LIST_YYB] : flag[LIST_YY]) != 0u)
                            : (HM == HM_CLASSIC && stalled && flag[LIST_YY] != 0u);
    // (this iteration's flow pass has recorded or streamed the candidate list: it matches the tile list --
    // until plan_lists schedules a rebuild)
    // head mode: the flow pass of the slot that ended has recorded (or streamed) the candidates of the xy buffer
    // it read -- a.ck_nblk[LIST_XY] blocks, unless its slice of the record overflowed
    const int xy_ck_done = (HM == HM_HEAD && ran_post && a.ck_nblk[LIST_XY] != 0 && flag[LIST_KEPT] == 0u) ? a.ck_nblk[LIST_XY] : 0;
    const int xy_ck_buf = lds->xy_active ? 1 : 0;
    const int sf_ck_done[2] = {(HM == HM_HEAD && ran_post && flag[LIST_KEPT] == 0u) ? a.ck_nblk[LIST_XX] : 0,
                               (HM == HM_HEAD && ran_post && flag[LIST_KEPT] == 0u) ? a.ck_nblk[LIST_YY] : 0};
    const int sf_ck_buf[2] = {lds->sf_active[0] ? 1 : 0, lds->sf_active[1] ? 1 : 0};
    bool ck_ok[3] = {false, false, false};
    if (HM == HM_CLASSIC && ran_post && flag[LIST_KEPT] == 0u) {
#pragma unroll
        for (int l = 0; l < 3; ++l) ck_ok[l] = a.ck_nblk[l] != 0 && flag[l] == 0u;
    }
    int done = lds->done;
    int pending = 0;
    {
        // ---- the plan of the slot that begins (a stall slot: the state did not move, plan only)
        DevHead L;
        __builtin_memcpy(&L, lds, sizeof(DevHead));
        if (HM == HM_CLASSIC) {
#pragma unroll
            for (int l = 0; l < 3; ++l)
                if (ck_ok[l]) L.ck_nblk[l] = a.ck_nblk[l];
        }
        if (xy_ck_done) { if (xy_ck_buf) L.xy_ck[1] = xy_ck_done; else L.xy_ck[0] = xy_ck_done; }
#pragma unroll
        for (int l = 0; l < 2; ++l)
            if (sf_ck_done[l]) { if (sf_ck_buf[l]) L.sf_ck[l][1] = sf_ck_done[l]; else L.sf_ck[l][0] = sf_ck_done[l]; }
        // (lane 0 alone stores through `bulk`; the other lanes compute the same values and drop them)
        prepare_iteration(&L, lds, lane0, p, pb);
        // the list built beside the slot that ended overflowed: the iteration itself was fine and is
        // kept; park so that the host enlarges the buffers (the flags stay up for it)
        if (pb.xy_failed || pb.sf_failed[0] || pb.sf_failed[1]) done = NEED_BIGGER_LIST;
        if (HM == HM_HEAD) pending = (done == RUNNING) ? 1 : 0;   // (this launch starts the next slot)
        if (lane0) {
#pragma unroll
            for (int q = 0; q < 9; ++q) lds->Rt[q] = L.Rt[q];
#pragma unroll
            for (int q = 0; q < 3; ++q) { lds->t[q] = L.t[q]; lds->tauf[q] = L.tauf[q]; lds->list_r[q] = L.list_r[q];
                                          lds->list_ok[q] = L.list_ok[q]; lds->reuse[q] = L.reuse[q]; lds->ck_nblk[q] = L.ck_nblk[q]; }
            lds->kc_ell = L.kc_ell; lds->run_hint = L.run_hint; lds->r_last = L.r_last;
            lds->xy_active = L.xy_active; lds->xy_target = L.xy_target; lds->stall = L.stall; lds->xy_fresh = L.xy_fresh;
            lds->tauf_build = L.tauf_build;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                lds->xy_ok[q] = L.xy_ok[q]; lds->xy_ck[q] = L.xy_ck[q]; lds->xy_r[q] = L.xy_r[q]; lds->rec_count[q] = L.rec_count[q];
                lds->sf_active[q] = L.sf_active[q]; lds->sf_target[q] = L.sf_target[q]; lds->sf_fresh[q] = L.sf_fresh[q];
                lds->sf_ok[q][0] = L.sf_ok[q][0]; lds->sf_ok[q][1] = L.sf_ok[q][1];
                lds->sf_ck[q][0] = L.sf_ck[q][0]; lds->sf_ck[q][1] = L.sf_ck[q][1];
                lds->sf_r[q][0] = L.sf_r[q][0]; lds->sf_r[q][1] = L.sf_r[q][1];
                lds->sf_tauf_build[q] = L.sf_tauf_build[q];
            }
            lds->done = done;
            if (HM != HM_CLASSIC) lds->pending = pending;
        }
    }
}

template <int HM>
__device__ __forceinline__ void head_math(DevHead *lds, const PostStepArgs &a, const bool run_post, const bool stalled,
                                          const unsigned (&flag)[LIST_N], const bool publisher, const bool timed,
                                          long long (&clk)[4])
{
    head_post<HM>(lds, a, run_post, publisher, timed, clk);
    // (one wave, LDS operations in order: the plan reads what the post-step part has just stored)
    if (lds->done == RUNNING) head_plan<HM>(lds, a, run_post, stalled, flag);
    if (timed) clk[3] = (long long)__builtin_readcyclecounter();
}

// What the publishing block does once the head's maths has run (all its threads): the tile lists the coming
// launches rebuild are emptied, the others are kept ...
template <int HM>
__device__ __forceinline__ void head_prepare_lists(const PostStepArgs &a, const DevHead *s_st)
{
    const int tid = threadIdx.x;
    const int nthr = blockDim.x;
    const bool async = a.prm.async_xy != 0, aself = a.prm.async_self != 0;
    // this slot's bits of the table's build masks (kt_filter): set where the coming filter launch has that list to build
    if (a.build_mask && tid < 3) {
        const bool sync_list = !((async && tid == LIST_XY) || (aself && tid != LIST_XY));
        const bool build = sync_list && s_st->done == RUNNING && s_st->reuse[tid] == 0;
        if (build) atomicOr(&a.build_mask[tid], a.slot_bit);
        else atomicAnd(&a.build_mask[tid], ~a.slot_bit);
    }
    if (!(s_st->done == RUNNING || s_st->done == NEED_BIGGER_LIST)) return;
#pragma unroll
    for (int l = 0; l < 3; ++l) {   // synchronous lists (classic plans)
        if (s_st->reuse[l] || (async && l == LIST_XY) || (aself && l != LIST_XY)) continue;
        for (int q = tid; q < NSUB; q += nthr) a.st->sub[l][q] = 0u;
        if (tid == 0) atomicOr(&a.st->built[l][(s_st->k >> 5) & 63], 1u << (s_st->k & 31));
    }
    if (async && s_st->xy_target >= 0) {   // the build the plan has just named
        const int l = s_st->xy_target ? (int)LIST_XYB : (int)LIST_XY;
        for (int q = tid; q < NSUB; q += nthr) a.st->sub[l][q] = 0u;
    }
    if (aself)
        for (int l = 0; l < 2; ++l)
            if (s_st->sf_target[l] >= 0) {
                const int id = self_list_id(l, s_st->sf_target[l]);
                for (int q = tid; q < NSUB; q += nthr) a.st->sub[id][q] = 0u;
            }
    // classic: every launch of the coming slot flags its overflows in row 0 again (a parked
    // loop keeps the flags: the host needs them to know what to grow)
    if (HM == HM_CLASSIC && tid < 8 && s_st->done == RUNNING) a.st->ovf[0][tid] = 0u;
}
// ... and the head goes out: the host's mirrors, then the state
__device__ __forceinline__ void head_publish(const PostStepArgs &a, const DevHead *s_st, DevHead *out, const bool math)
{
    // a loop that has stopped with a verdict: the final head into the host's pinned copy, all of it there before the `done` word
    const bool final_out = a.final_mirror != nullptr && (s_st->done == DONE_BREAK_A || s_st->done == DONE_BREAK_B || s_st->done == DONE_MAX_ITER);
    if (final_out) {
        // (with a check word over the copy: the host takes the mirror only when the word matches what it has read -- a piece that
        // lands after the `done` word is then a retry, not a wrong state; head_check_word)
        const uint4 *src = reinterpret_cast<const uint4 *>(s_st);
        uint4 *dst = reinterpret_cast<uint4 *>(a.final_mirror);
        constexpr int pieces = (int)(DEVSTATE_HEAD_BYTES / 16);
        static_assert(pieces <= 64 && offsetof(DevHead, head_check_) == DEVSTATE_HEAD_BYTES - 12, "one wave copies the head; the check word is the last piece's second word");
        if (threadIdx.x < 64) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if ((int)threadIdx.x < pieces) v = src[threadIdx.x];
            if ((int)threadIdx.x == pieces - 1) v.y = 0u;   // (the word itself counts as zero)
            unsigned sum = (int)threadIdx.x < pieces ? head_check_mix(v.x, v.y, v.z, v.w, threadIdx.x) : 0u;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) sum += (unsigned)__shfl_xor((int)sum, off, 64);
            if ((int)threadIdx.x == pieces - 1) v.y = sum;
            if ((int)threadIdx.x < pieces) dst[threadIdx.x] = v;
        }
        __threadfence_system();
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        // (members of A expected in the slot that begins: what the host picks the next batch's plan by, kt_run -- in front
        // of the slot count the host paces its batches on)
        // (not with a verdict: the host leaves on the `done` word and may have reset the other two for its next registration
        // before a write that was issued in front of that word has landed -- writes to host memory are seen to pass each other)
        if (math && a.hint_mirror && s_st->done == RUNNING) *a.hint_mirror = s_st->run_hint;
        if (math && a.progress_mirror && s_st->done == RUNNING) *a.progress_mirror = s_st->n_slots;
        if (a.done_mirror && s_st->done != RUNNING) *a.done_mirror = s_st->done;
    }
    state_head_from_lds(s_st, out);
}

// The whole head of one block.  in / out: the copies of the state's head the launch reads / writes (the
// same in the classic and flush forms); st: the state itself (the tail: sub-list counters, overflow
// flags).  Returns true if the slot that begins may run (head mode: the loop is running, no stall).
// `count_pa`: the flow pass of a plan with resident runs -- the publishing block counts the candidates of the record in use
// (DevHead::run_hint, exact where the record is complete: what the host picks a RUN batch by).
template <int HM>
__device__ __forceinline__ bool head_body(const PostStepArgs &a, const DevHead *in, DevHead *out, DevHead *s_st,
                                          double *sh /*[4 * NACC_MAX]*/, const int par, const bool publisher,
                                          const ProcessArgs *count_pa = nullptr)
{
    const int tid = threadIdx.x;
    __shared__ unsigned s_cand_total;
    const bool count = HM == HM_HEAD && publisher && count_pa != nullptr && a.run_mail != nullptr && count_pa->cand_cnt != nullptr &&
                       count_pa->cand_cnt_b != nullptr;
    if (count && tid == 0) s_cand_total = 0u;
    const bool reduce = HM != HM_CLASSIC || (a.flags & POST_REDUCE) != 0;
    const bool math = HM != HM_CLASSIC || (a.flags & POST_MATH) != 0;
    const bool timed = a.dbg != nullptr && publisher;   // (CVO_HIP_POST_DEBUG: phase clocks of the publishing block)
    const long long c0 = timed ? (long long)__builtin_readcyclecounter() : 0;
    // one round trip: the step partials, the overflow flags of the builds that have ended (classic: row
    // 0, where everything is flagged; head mode: the row of the previous flow launch), the state's head
    double sp[NACC_STEP];
    if (reduce) thread_load_partials<NACC_STEP>(a.part_step, a.nblk, sp);
    const unsigned my_flag = a.st->ovf[HM == HM_CLASSIC ? 0 : (par ^ 1)][tid & 7];
    state_head_to_lds(in, s_st);
    if (a.check_done && s_st->done != 0) {
        // the loop has stopped: head mode hands the head on, so that every later launch finds the verdict
        // whichever copy it reads
        if (HM == HM_HEAD && publisher) state_head_from_lds(s_st, out);
        return false;
    }
    const bool async = a.prm.async_xy != 0;
    // a stall slot executed no iteration (asynchronous builds: only the plan runs)
    const bool stalled = async && s_st->stall != 0;
    const bool pending = HM == HM_CLASSIC ? true : (s_st->pending != 0);
    const bool run_post = pending && !stalled;
    const long long c1 = timed ? (long long)__builtin_readcyclecounter() : 0;
    if (reduce && run_post) block_finish_partials<NACC_STEP>(sp, sh, s_st->red + RED_STEP);
    const long long c2 = timed ? (long long)__builtin_readcyclecounter() : 0;
    bool comm_ok = true;
    if (HM == HM_CLASSIC && a.comm && !stalled) {
        __shared__ double sh_mail[MAX_WORLD * MAIL_VALS];
        __shared__ int sh_fail;
        comm_ok = mailbox_allreduce(*a.comm, a.st, s_st->red + RED_STEP, RED_N - RED_STEP, sh_mail, &sh_fail, s_st);
        if (!comm_ok && tid == 0) s_st->done = DONE_COMM_ERROR;
    }
    if (math && comm_ok) {
        const int act0 = s_st->xy_active ? 1 : 0;   // (the buffer in use when the slot that ended began)
        if (tid < 64) {   // wave 0
            unsigned flag[LIST_N];
#pragma unroll
            for (int l = 0; l < LIST_N; ++l) flag[l] = (unsigned)__builtin_amdgcn_readlane((int)my_flag, l);
            long long clk[4] = {0, 0, 0, 0};
            head_math<HM>(s_st, a, run_post, stalled, flag, publisher, timed, clk);
            if (timed && tid == 0 && run_post) {
                a.dbg[0] += 1; a.dbg[1] += c1 - c0; a.dbg[2] += c2 - c1;
                a.dbg[3] += clk[1] - clk[0]; a.dbg[4] += clk[2] - clk[1]; a.dbg[5] += clk[3] - clk[2];
                a.dbg[6] = c0;   // (head mode: the flow pass that follows adds its own time, kt_hflow_build)
            }
        } else if (count) {
            // the other waves, beside the head's maths: the slices of that buffer's record (as kt_run numbers them)
            const uint32_t *c = act0 ? count_pa->cand_cnt_b : count_pa->cand_cnt;
            const unsigned wcap = count_pa->kept_wcap;
            const int nsl = 4 * count_pa->nblk;
            unsigned sum = 0u;
            for (int sl = tid - 64; sl < nsl; sl += BLOCK - 64) {
                const unsigned v = c[sl];
                sum += v < wcap ? v : wcap;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) sum += (unsigned)__shfl_xor((int)sum, off, 64);
            if ((tid & 63) == 0) atomicAdd(&s_cand_total, sum);
        }
        __syncthreads();
        if (count) {
            // exact where it can be: the loop runs on that buffer still, its record is complete (a flow pass has written it) and
            // no build is named or in flight (a run would decline: the estimate of prepare_iteration stands)
            if (tid == 0 && s_st->done == RUNNING && (s_st->xy_active ? 1 : 0) == act0 && (act0 ? s_st->xy_ck[1] : s_st->xy_ck[0]) == count_pa->nblk) {
                if (act0) s_st->rec_count[1] = (int32_t)s_cand_total; else s_st->rec_count[0] = (int32_t)s_cand_total;
                if (s_st->stall == 0 && s_st->xy_target < 0 && s_st->xy_fresh < 0) s_st->run_hint = (int32_t)s_cand_total;
            }
            __syncthreads();
        }
        if (publisher) head_prepare_lists<HM>(a, s_st);
    }
    if (publisher) head_publish(a, s_st, out, math);
    return s_st->done == RUNNING && !(async && s_st->stall != 0);
}

__device__ __forceinline__ void post_step_body(const PostStepArgs &a)
{
    __shared__ double sh[4 * NACC_MAX];
    __shared__ __attribute__((aligned(16))) DevHead s_st;
    head_body<HM_CLASSIC>(a, a.st, a.st, &s_st, sh, 0, true);
}

__global__ void __launch_bounds__(BLOCK) k_post_step(const Grp<PostStepArgs> grp)
{
    post_step_body(grp.a[blockIdx.z]);
}

// First iteration of an align() (or of its resumption after a list grew): no
// list is valid, everything is rebuilt.
__global__ void k_prepare(DevState *st, const DevParams prm, uint32_t *build_masks, const PrepareInit init)
{
    if (init.on) {   // a registration begins: zeros but for what the caller carries (everything in front of the mailbox sequence number)
        static_assert(DEVSTATE_INIT_BYTES % 4 == 0, "word stores");
        uint32_t *z = reinterpret_cast<uint32_t *>(st);
        for (int q = threadIdx.x; q < (int)(DEVSTATE_INIT_BYTES / 4); q += BLOCK) z[q] = 0u;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int q = 0; q < 9; ++q) st->R[q] = init.R[q];
            for (int q = 0; q < 3; ++q) { st->T[q] = init.T[q]; st->center[q] = init.center[q]; }
            st->ell = init.ell; st->ell_max = init.ell_max;
            st->iter = init.iter; st->n_fixed = init.n_fixed; st->done = init.done;
            st->xmax = init.xmax; st->y0max = init.y0max;
        }
        __syncthreads();
    }
    // (a registration begins in this table: every list is to be built, the build masks' bits of every slot go up -- kt_filter)
    if (build_masks && threadIdx.x < 4) build_masks[threadIdx.x] = 0xffffffffu;
    for (int q = threadIdx.x; q < LIST_N * NSUB; q += BLOCK) (&st->sub[0][0])[q] = 0u;
    if (threadIdx.x < 16) (&st->ovf[0][0])[threadIdx.x] = 0u;
    if (threadIdx.x == 0) {
        for (int l = 0; l < 3; ++l) {
            st->list_ok[l] = 0;
            atomicOr(&st->built[l][(st->k >> 5) & 63], 1u << (st->k & 31));
        }
        for (int l = 0; l < 2; ++l) {      // async self lists: nothing built yet
            st->sf_ok[l][0] = st->sf_ok[l][1] = 0;
            st->sf_ck[l][0] = st->sf_ck[l][1] = 0;
            st->sf_active[l] = 0; st->sf_target[l] = -1; st->sf_fresh[l] = -1;
        }
        st->xy_ok[0] = st->xy_ok[1] = 0;   // async xy: the first slot only builds
        st->xy_ck[0] = st->xy_ck[1] = 0;
        st->rec_count[0] = st->rec_count[1] = 0;
        st->xy_active = 0;
        st->xy_target = -1;
        st->xy_fresh = -1;
        st->stall = 0;
        st->pending = 0;
        st->mail_snap = st->mail_seq;
        prepare_iteration(st, st, true, prm, plan_builds_none());
    }
}

void launch_prepare(DevState *st, const DevParams &prm, hipStream_t s, uint32_t *build_masks, const PrepareInit *init)
{
    PrepareInit in{};
    if (init) in = *init;
    hipLaunchKernelGGL(k_prepare, dim3(1), dim3(BLOCK), 0, s, st, prm, build_masks, in);
}

void launch_post_flow_group(const PostFlowArgs *a, int n, hipStream_t s)
{
    Grp<PostFlowArgs> g;
    for (int i = 0; i < n; ++i) g.a[i] = a[i];
    hipLaunchKernelGGL(k_post_flow, dim3(1, 1, (unsigned)n), dim3(BLOCK), 0, s, g);
}

void launch_post_step_group(const PostStepArgs *a, int n, hipStream_t s)
{
    Grp<PostStepArgs> g;
    for (int i = 0; i < n; ++i) g.a[i] = a[i];
    hipLaunchKernelGGL(k_post_step, dim3(1, 1, (unsigned)n), dim3(BLOCK), 0, s, g);
}

// ---------------------------------------------------------------------------
// The same kernels reading their arguments from a table of Slots in device memory
// (cvo_device.h "Argument tables"): blockIdx.z = slot, op[q] = this launch's arguments.
// These are the kernels of the align() loop; the by-value forms above serve the low-level
// entry points and the profiling mode.
// The table is read through the CONSTANT address space: nothing writes it while a kernel runs
// (the host updates it with stream-ordered copies between launches), so its fields are scalar,
// invariant loads the compiler may issue late or repeat -- exactly what kernel arguments are.
// (Through a plain global pointer the same kernels need ~30 more vector registers.)
typedef const __attribute__((address_space(4))) Slot *CSlot;
#define CVO_SLOT(tab)                          \
    CSlot cs = (CSlot)(tab) + blockIdx.z;      \
    if (cs->active == 0) return
#define CVO_ARG(T, field) (*(const T *)(&cs->field))
// the filter argument block of role 0 (the xy list), 1, 2 (the xx / yy lists): op[q], op[q + 1], op[q + 2]
#define CVO_FILTER_ROLE(role) (*(const FilterArgs *)(&cs->op[q + (role)].f))

__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(6, 8)))
kt_filter(const Slot *__restrict__ tab, const int ql)
{
    const int q = ql & QP_MASK, list = (ql >> 8) & 3;   // (the list rides in the launch's argument: no load in front of the mask's)
    // The table's build mask of this list (in front of the slots, kTableHeaderBytes): bit z clear = slot z has certainly nothing to
    // build in this launch -- what 60 of 65 filter launches of a registration find.  One word for all blocks of all slots (a scalar
    // load that hits after the first), read BEFORE the slot's own arguments and state: a block that is not needed leaves after it,
    // which is what lets a crowded engine give a build 256 blocks per slot instead of 64 (profiles/r05_ab.txt 10).  The
    // first blocks of a slot stay for the transform pass that rides here (one per BLOCK points).
    typedef const __attribute__((address_space(4))) uint32_t *CMask;   // (a scalar load, like the table's own words)
    const CMask masks = (CMask)(reinterpret_cast<const char *>(tab) - kTableHeaderBytes);
    const bool may_build = ((masks[list] >> blockIdx.z) & 1u) != 0u;
    constexpr unsigned PT_BLOCKS = 64;   // (blocks that stay for the transform pass: 16 384 points in one sweep)
    if (!may_build && blockIdx.x >= PT_BLOCKS) return;
    CVO_SLOT(tab);
    if (may_build) filter_body(CVO_ARG(FilterArgs, op[q].f), blockIdx.x, gridDim.x, cs->op[q].f.st, 0);
    if (blockIdx.x < PT_BLOCKS)
        pretransform_body(CVO_ARG(FilterArgs, op[q].f), blockIdx.x, min(gridDim.x, PT_BLOCKS));   // (after: nothing of it is live across the build)
}

// acvo, synchronous lists, one registration: the three filters in one launch (blockIdx.y = list)
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(6, 8)))
kt_filter_group(const Slot *__restrict__ tab, const int q)
{
    // (the build masks: see kt_filter; blockIdx.y = list = LIST_XY, LIST_XX, LIST_YY)
    typedef const __attribute__((address_space(4))) uint32_t *CMask;
    const CMask masks = (CMask)(reinterpret_cast<const char *>(tab) - kTableHeaderBytes);
    const bool may_build = ((masks[blockIdx.y] >> blockIdx.z) & 1u) != 0u;
    constexpr unsigned PT_BLOCKS = 64;
    if (!may_build && (blockIdx.y != 0 || blockIdx.x >= PT_BLOCKS)) return;
    CVO_SLOT(tab);
    if (may_build) filter_body(CVO_FILTER_ROLE((int)blockIdx.y), blockIdx.x, gridDim.x, cs->op[q + (int)blockIdx.y].f.st, 0);
    if (blockIdx.y == 0 && blockIdx.x < PT_BLOCKS) pretransform_body(CVO_FILTER_ROLE(0), blockIdx.x, min(gridDim.x, PT_BLOCKS));
}

// kt_process<PROC_FLOW, 0> is built for loops that never read the sum of a d2 (the cvo loop: ProcessArgs::need_d2 == 0 in
// every slot; kt_flow_d2 below is the same pass with the sum) -- two vector registers and two instructions per member
// less, which is what lets the flow pass of a crowded engine run seven waves per SIMD instead of six (64 / 256
// distinct pairs per call 3 671 -> 3 775 / 4 166 -> 4 280 registrations/s; with the sum in, seven waves spill more
// and acvo loses 1.5 %: profiles/r03_ab.txt 32)
#ifndef CVO_FLOW_WAVES
#define CVO_FLOW_WAVES 5   // (7 / 6 / 5 measured alike on the round-4 loops, profiles/r04_ab.txt 18; at 6 the kernel spilled 4 vector registers)
#endif
template <int MODE, int WEIGHT = 0>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu((MODE == PROC_FLOW && WEIGHT == 0) ? CVO_FLOW_WAVES : 1, 8)))
kt_process(const Slot *__restrict__ tab, const int q)
{
    __shared__ __attribute__((aligned(16))) char scratch[PROC_SMEM];
    CVO_SLOT(tab);
    const ProcessArgs &a = CVO_ARG(ProcessArgs, op[q].p);
    ProcHead hd = proc_head_global<MODE>(a, a.st, 0);
    if (MODE == PROC_FLOW && WEIGHT == 0) hd.need_d2 = 0;
    process_body<MODE, WEIGHT>(a, blockIdx.x, scratch, hd);
}

__global__ void __launch_bounds__(BLOCK) kt_flow_d2(const Slot *__restrict__ tab, const int q)
{
    __shared__ __attribute__((aligned(16))) char scratch[PROC_SMEM];
    CVO_SLOT(tab);
    const ProcessArgs &a = CVO_ARG(ProcessArgs, op[q].p);
    process_body<PROC_FLOW, 0>(a, blockIdx.x, scratch, proc_head_global<PROC_FLOW>(a, a.st, 0));
}

// acvo, one registration: both self passes in one launch (blockIdx.y = xx / yy)
__global__ void __launch_bounds__(BLOCK) kt_self2(const Slot *__restrict__ tab, const int q)
{
    __shared__ __attribute__((aligned(16))) char scratch[PROC_SMEM];
    CVO_SLOT(tab);
    const ProcessArgs &a = CVO_ARG(ProcessArgs, op[q + blockIdx.y].p);
    process_body<PROC_SELF>(a, blockIdx.x, scratch, proc_head_global<PROC_SELF>(a, a.st, 0));
}

// (head mode, qp & QP_HEAD: the slot runs on the copy of the head its flow launch wrote)
__global__ void __launch_bounds__(STEP_BLOCK) kt_step_twist(const Slot *__restrict__ tab, const int qp)
{
    CSlot cs = (CSlot)(tab) + blockIdx.z;
    const int q = qp & QP_MASK, par = (qp & QP_PARITY) ? 1 : 0;
    const bool head_mode = (qp & QP_HEAD) != 0;
    const ProcessArgs &a = CVO_ARG(ProcessArgs, op[q].p);
    // The words of the table a block needs before it can request any data are asked for TOGETHER with the
    // slot's switch (the table is read through the scalar cache, and every dependent read of it is a round
    // trip of its own: switch -> block count -> state address -> state was four in a row; the empty asm
    // pins the requests in front of the first branch).
    {
        const int active = cs->active;
        const int nblk = a.nblk, check_done = a.check_done, async_xy = a.async_xy, packed = a.kept_packed;
        const unsigned wcap = a.kept_wcap;
        const DevState *st = a.st, *st2 = a.st2;
        const double *fp = a.flow_part;
        const uint32_t *kc = a.kept_cnt;
        const uint2 *kij = a.kept_ij;
        const float *ka = a.kept_a;
        asm volatile("" ::"s"(active), "s"(nblk), "s"(check_done), "s"(async_xy), "s"(packed), "s"(wcap), "s"(st), "s"(st2),
                     "s"(fp), "s"(kc), "s"(kij), "s"(ka));
        if (active == 0) return;
    }
    DevState *const st_a = a.st, *const st_b = a.st2;   // (both already here: a select of values, not of addresses to read)
    step_twist_body(a, head_mode ? (par ? st_a : st_b) : st_a, par, head_mode);
}

__global__ void __launch_bounds__(BLOCK) kt_post_flow(const Slot *__restrict__ tab, const int q)
{
    CVO_SLOT(tab);
    // (the post kernels are one long dependent chain on one lane: their arguments are fetched once,
    // up front, instead of where the chain first needs them)
    const PostFlowArgs a = CVO_ARG(PostFlowArgs, op[q].pf);
    post_flow_body(a);
}

__global__ void __launch_bounds__(BLOCK) kt_post_step(const Slot *__restrict__ tab, const int q)
{
    CVO_SLOT(tab);
    const PostStepArgs a = CVO_ARG(PostStepArgs, op[q].ps);
    post_step_body(a);
}

// The merged launches of a registration with its launches to itself (asynchronous list builds,
// cvo_device.h plan_xy_async / plan_self_async):
//   kt_flow_build  : blocks [0, np) run the flow pass of this slot on the xy buffer in use, the
//                    next n0 are k_filter blocks that build the idle buffer at this slot's
//                    transform when the plan step asked for it (and return after their first
//                    load otherwise): the filter is off the launch chain, three dependent
//                    launches per iteration, builds that nobody waits for;
//   kt_flow_build3 : acvo -- the same plus the xx / yy filters (synchronous lists, consumed by
//                    the kt_self2 launch that follows): no k_filter launch of its own is left;
//   kt_flow_build6 : acvo with the self lists built ahead as well -- flow pass, both self passes
//                    and all three builds in ONE launch.
// A block reads the argument block of the role it plays.  Built for 4 waves per SIMD: every vector
// register in registers, no scratch (held to 6 waves -- 80 registers -- 8 to 12 of them spill and
// a registration is 1.5 % slower, profiles/r02_ab.txt).
#define CVO_MERGED_KERNELS(SUFFIX, WAVES)                                                                  \
    __global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(WAVES, 8)))               \
    kt_flow_build##SUFFIX(const Slot *__restrict__ tab, const int q)                                       \
    {                                                                                                      \
        extern __shared__ __attribute__((aligned(16))) char smem[];                                        \
        CVO_SLOT(tab);                                                                                     \
        const int np = cs->op[q].np;                                                                       \
        if ((int)blockIdx.x < np) {                                                                        \
            const ProcessArgs &pa = CVO_ARG(ProcessArgs, op[q].p);                                         \
            process_body<PROC_FLOW, 0, false, false>(pa, blockIdx.x, smem, proc_head_global<PROC_FLOW>(pa, pa.st, 0)); \
            return;                                                                                        \
        }                                                                                                  \
        filter_body<false>(CVO_ARG(FilterArgs, op[q].f), blockIdx.x - (unsigned)np, (unsigned)cs->op[q].n0,       \
                    cs->op[q].f.st, 0);                                                                    \
    }                                                                                                      \
    __global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(WAVES, 8)))               \
    kt_flow_build3##SUFFIX(const Slot *__restrict__ tab, const int q)                                      \
    {                                                                                                      \
        extern __shared__ __attribute__((aligned(16))) char smem[];                                        \
        CVO_SLOT(tab);                                                                                     \
        int b = (int)blockIdx.x;                                                                           \
        const int np = cs->op[q].np, n0 = cs->op[q].n0, n1 = cs->op[q].n1, n2 = cs->op[q].n2;             \
        if (b < np) {                                                                                      \
            const ProcessArgs &pa = CVO_ARG(ProcessArgs, op[q].p);                                         \
            process_body<PROC_FLOW, 0, false, false>(pa, (unsigned)b, smem, proc_head_global<PROC_FLOW>(pa, pa.st, 0)); \
            return;                                                                                        \
        }                                                                                                  \
        b -= np;                                                                                           \
        const int role = b < n0 ? 0 : (b < n0 + n1 ? 1 : 2);                                               \
        filter_body<false>(CVO_FILTER_ROLE(role), (unsigned)(b - (role == 0 ? 0 : (role == 1 ? n0 : n0 + n1))),   \
                    (unsigned)(role == 0 ? n0 : (role == 1 ? n1 : n2)), cs->op[q + role].f.st, 0);         \
    }                                                                                                      \
    __global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(WAVES, 8)))               \
    kt_flow_build6##SUFFIX(const Slot *__restrict__ tab, const int q)                                      \
    {                                                                                                      \
        extern __shared__ __attribute__((aligned(16))) char smem[];                                        \
        CVO_SLOT(tab);                                                                                     \
        int b = (int)blockIdx.x;                                                                           \
        const int np = cs->op[q].np;                                                                       \
        if (b < np) {                                                                                      \
            const ProcessArgs &pa = CVO_ARG(ProcessArgs, op[q].p);                                         \
            process_body<PROC_FLOW, 0, false, false>(pa, (unsigned)b, smem, proc_head_global<PROC_FLOW>(pa, pa.st, 0)); \
            return;                                                                                        \
        }                                                                                                  \
        b -= np;                                                                                           \
        if (b < 2 * np) {                                                                                  \
            const int w = b >= np ? 1 : 0;                                                                 \
            const ProcessArgs &pa = CVO_ARG(ProcessArgs, op[q + 1 + w].p);                                 \
            /* (CAND false: the records of double-buffered lists belong to head mode, see plan_lone) */   \
            process_body<PROC_SELF, 0, false, false>(pa, (unsigned)(b - w * np), smem, proc_head_global<PROC_SELF>(pa, pa.st, 0)); \
            return;                                                                                        \
        }                                                                                                  \
        b -= 2 * np;                                                                                       \
        const int n0 = cs->op[q].n0, n1 = cs->op[q].n1;                                                    \
        const int role = b < n0 ? 0 : (b < n0 + n1 ? 1 : 2);                                               \
        filter_body<false>(CVO_FILTER_ROLE(role), (unsigned)(b - (role == 0 ? 0 : (role == 1 ? n0 : n0 + n1))),   \
                    (unsigned)(role == 0 ? n0 : (role == 1 ? n1 : cs->op[q].n2)), cs->op[q + role].f.st, 0); \
    }
CVO_MERGED_KERNELS(_w4, 4)

// ---------------------------------------------------------------------------
// Head mode (see "the head" above): the flow launch of a registration on its own, every flow / self
// block of which starts with the post-step part of the previous slot and the plan of this one.
// qp = op index | slot parity << 8 | QP_HEAD.
// ---------------------------------------------------------------------------
// What the list pass of a head-mode block takes from the head the block has just computed (LDS): the
// hot part -- kernel constants, inverse transform -- goes to scalar registers.
template <int MODE>
__device__ __forceinline__ ProcHead proc_head_lds(const ProcessArgs &a, const DevHead *h, const int par, float (&rt)[12])
{
    auto uni = [](float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); };
    auto unid = [](double x) {
        return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
    };
    ProcHead hd;
#pragma unroll
    for (int q = 0; q < 9; ++q) rt[q] = uni(h->Rt[q]);
#pragma unroll
    for (int q = 0; q < 3; ++q) rt[9 + q] = uni(h->t[q]);
    hd.Rt = rt; hd.tt = rt + 9; hd.xi = nullptr;
    const KernConsts &k = h->kc;
    hd.kc.tau = uni(k.tau); hd.kc.tau_c = uni(k.tau_c); hd.kc.sp = uni(k.sp);
    hd.kc.inv_c = uni(k.inv_c); hd.kc.inv_d = uni(k.inv_d); hd.kc.inv_l3 = uni(k.inv_l3);
    hd.kc.cb = uni(k.cb); hd.kc.cg = uni(k.cg); hd.kc.cd = uni(k.cd); hd.kc.cscale = uni(k.cscale);
    hd.kc.s2_d = unid(k.s2_d); hd.kc.cs2_d = unid(k.cs2_d);
    hd.kc.ninv_2l2 = unid(k.ninv_2l2); hd.kc.ninv_2cl2 = unid(k.ninv_2cl2);
    hd.done_word = 0;   // (head_body has looked)
    hd.n_fixed = (MODE == PROC_SELF && a.first_counted) ? __builtin_amdgcn_readfirstlane(h->n_fixed) : 0;
    hd.second = ((MODE == PROC_FLOW && __builtin_amdgcn_readfirstlane(h->xy_active) == 1) ||
                 (MODE == PROC_SELF && __builtin_amdgcn_readfirstlane(h->sf_active[a.async_self == 2 ? 1 : 0]) == 1)) ? 1 : 0;
    hd.list_bad = 0u;   // (lists built ahead are only switched to after their flag was seen clear)
    hd.ck_nblk = 0;
    hd.par = par;
    hd.cand = nullptr; hd.cand_cnt = nullptr;
    hd.need_d2 = a.need_d2;
    if (a.cand_b) {   // the record of the buffer in use (xy list; acvo: xx / yy)
        const int l = a.async_self == 2 ? 1 : 0;
        const int ck = MODE == PROC_FLOW ? (hd.second ? h->xy_ck[1] : h->xy_ck[0]) : (hd.second ? h->sf_ck[l][1] : h->sf_ck[l][0]);
        hd.ck_nblk = __builtin_amdgcn_readfirstlane(ck);
        hd.cand = hd.second ? a.cand_b : a.cand;
        hd.cand_cnt = hd.second ? a.cand_cnt_b : a.cand_cnt;
    }
    return hd;
}

// CVO_SLOT for the head-mode launches: the slot's switch, the role boundary and the words of the head's
// argument block that stand in front of its first data request, asked for together (see kt_step_twist)
#define CVO_SLOT_HEAD(tab)                                                                                    \
    CSlot cs = (CSlot)(tab) + blockIdx.z;                                                                     \
    const int q = qp & QP_MASK, par = (qp & QP_PARITY) ? 1 : 0;                                               \
    {                                                                                                         \
        const PostStepArgs &hps = CVO_ARG(PostStepArgs, op[q].ps);                                            \
        const int active = cs->active, np_ = cs->op[q].np, hn = hps.nblk, hcd = hps.check_done;               \
        const DevState *hst = hps.st, *hst2 = hps.st2;                                                        \
        const double *hpart = hps.part_step;                                                                  \
        const long long *hdbg = hps.dbg;                                                                      \
        asm volatile("" ::"s"(active), "s"(np_), "s"(hn), "s"(hcd), "s"(hst), "s"(hst2), "s"(hpart), "s"(hdbg)); \
        if (active == 0) return;                                                                              \
    }
#define CVO_HEAD_KERNELS(SUFFIX, WAVES)                                                                    \
    __global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(WAVES, 8)))               \
    kt_hflow_build##SUFFIX(const Slot *__restrict__ tab, const int qp)                                     \
    {                                                                                                      \
        extern __shared__ __attribute__((aligned(16))) char smem[];                                        \
        __shared__ double sh[4 * NACC_MAX];                                                                \
        __shared__ __attribute__((aligned(16))) DevHead s_st;                                              \
        CVO_SLOT_HEAD(tab);                                                                                \
        const int np = cs->op[q].np;                                                                       \
        if ((int)blockIdx.x >= np) {                                                                       \
            const FilterArgs &f = CVO_ARG(FilterArgs, op[q].f);                                            \
            filter_body<false>(f, blockIdx.x - (unsigned)np, (unsigned)cs->op[q].n0, par ? f.st2 : f.st, par);    \
            return;                                                                                        \
        }                                                                                                  \
        const PostStepArgs ps = CVO_ARG(PostStepArgs, op[q].ps);                                           \
        const ProcessArgs &pa = CVO_ARG(ProcessArgs, op[q].p);                                             \
        if (!head_body<HM_HEAD>(ps, par ? ps.st2 : ps.st, par ? ps.st : ps.st2, &s_st, sh, par,            \
                                blockIdx.x == 0, &pa)) return;                                             \
        float rt[12];                                                                                      \
        process_body<PROC_FLOW, 0, true, false>(pa, blockIdx.x, smem, proc_head_lds<PROC_FLOW>(pa, &s_st, par, rt)); \
        if (ps.dbg && blockIdx.x == 0 && threadIdx.x == 0 && ps.dbg[6] != 0) {   /* block 0: head + flow pass */ \
            ps.dbg[7] += (long long)__builtin_readcyclecounter() - ps.dbg[6];                              \
            ps.dbg[6] = 0;                                                                                 \
        }                                                                                                  \
    }                                                                                                      \
    __global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(WAVES, 8)))               \
    kt_hflow_build6##SUFFIX(const Slot *__restrict__ tab, const int qp)                                    \
    {                                                                                                      \
        extern __shared__ __attribute__((aligned(16))) char smem[];                                        \
        __shared__ double sh[4 * NACC_MAX];                                                                \
        __shared__ __attribute__((aligned(16))) DevHead s_st;                                              \
        CVO_SLOT_HEAD(tab);                                                                                \
        int b = (int)blockIdx.x;                                                                           \
        const int np = cs->op[q].np;                                                                       \
        if (b >= 3 * np) {                                                                                 \
            b -= 3 * np;                                                                                   \
            const int n0 = cs->op[q].n0, n1 = cs->op[q].n1;                                                \
            const int role = b < n0 ? 0 : (b < n0 + n1 ? 1 : 2);                                           \
            const FilterArgs &f = CVO_FILTER_ROLE(role);                                                   \
            filter_body<false>(f, (unsigned)(b - (role == 0 ? 0 : (role == 1 ? n0 : n0 + n1))),                   \
                        (unsigned)(role == 0 ? n0 : (role == 1 ? n1 : cs->op[q].n2)), par ? f.st2 : f.st, par); \
            return;                                                                                        \
        }                                                                                                  \
        const PostStepArgs ps = CVO_ARG(PostStepArgs, op[q].ps);                                           \
        const int role = b < np ? 0 : (b < 2 * np ? 1 : 2);   /* flow, xx, yy */                          \
        const unsigned rb = (unsigned)(b - role * np);                                                     \
        const ProcessArgs &pa = CVO_ARG(ProcessArgs, op[q + role].p);                                      \
        if (!head_body<HM_HEAD>(ps, par ? ps.st2 : ps.st, par ? ps.st : ps.st2, &s_st, sh, par,            \
                                blockIdx.x == 0)) return;                                                  \
        float rt[12];                                                                                      \
        if (role == 0) {                                                                                   \
            process_body<PROC_FLOW, 0, true, false>(pa, rb, smem, proc_head_lds<PROC_FLOW>(pa, &s_st, par, rt));  \
            return;                                                                                        \
        }                                                                                                  \
        process_body<PROC_SELF, 0, true, false>(pa, rb, smem, proc_head_lds<PROC_SELF>(pa, &s_st, par, rt));      \
    }
CVO_HEAD_KERNELS(_w4, 4)


// ---------------------------------------------------------------------------
// Resident runs (round 5): whole iterations of ONE registration in one launch.
//
// The narrow part of a registration on its own -- a few ten thousand candidates per iteration, the tile list and
// its candidate record valid for the next dozen iterations -- spent its time at the two launch boundaries of an
// iteration, in the prologues behind them and in the head's state load, not in its arithmetic (17 us per iteration
// at 10k x 10k of which ~4 us are the two passes, profiles/r05_ab.txt 0).  kt_run executes up to `run_iters` WHOLE
// iterations (ref src/cvo.cpp:366-410: transform, flow sums, twist, step sums, cubic, Exp, update, length scale,
// stop tests) in ONE launch of 1 + RUN_G blocks of 512 threads:
//   * blocks 1 .. g are SOLVERS (g = 8, 16 or 32, as few as hold the record: an exchange costs by the block).  THE
//     CANDIDATES LIVE IN REGISTERS for the life of the run.  The record of the tile list in use (ProcessArgs::cand,
//     written by the classic flow pass after the last build) does not change while the list is re-used, and neither
//     do x_i, y0_j and the colour weight of a candidate: at entry solver lane l takes candidates l, l + 512 g, ...
//     (a flat numbering over the record's slices: balanced whatever the slices hold), loads (i, j, ck), x_i and y0_j
//     ONCE, and an iteration's flow pass is transform + exact test + sums on registers -- no memory access at all;
//     the members' weights stay in registers for the step pass: no kept list either;
//   * between the passes the solvers exchange their partial sums through a RunMail (cvo_device.h): tagged 8-byte
//     words, polled by everybody; every block adds the g rows in one fixed order and holds the same totals, bit for bit;
//   * every block runs the twist constants and the POST-STEP part of the head (cubic, Exp_SEK3, update, length scale,
//     stop tests: head_post) itself, on its own copy of the state's head in LDS -- that chain is the critical path
//     of an iteration and its result is needed everywhere;
//   * block 0 is the HEAD BLOCK: no candidates, no row in the exchanges (it reads them).  It alone runs the PLAN of
//     the next slot (head_plan: filter bounds, travel of the cloud against the lists' radii, builds) while the
//     solvers are already in the slot's passes, and tells them what they need of it -- stall, build named -- in a
//     verdict word that travels with the slot's second exchange; it writes the trace records and the host's mirrors
//     and puts the head back when the run ends.
// A run is one launch of a head-mode plan's batch: [kt_run, HF, ST, HF, ST] (cvo_plan.cpp).  It starts where a
// head-mode flow launch of parity 0 would (head in copy 0, the previous slot's step sums in part_step, its
// overflow flags in row 1) and ends where a step launch of parity 1 leaves off (head in copy 0, pending, the last
// slot's step sums as row 0 of part_step), so that classic launches and runs can follow each other in any order.
// It declines -- returns with nothing written but the host's run counter -- unless the loop is running on a
// candidate record that fits into the registers, and it ends after the slot whose head has named a list build
// (the filter blocks of the next classic flow launch make it; two classic slots later the next run starts on the new
// record), when the loop stops, or after `run_iters` iterations.  All blocks decide the same from the same inputs.
// A poll that does not fill within its limit (PostStepArgs::run_timeout_ticks, else RUN_TIMEOUT_TICKS) ends the run with DONE_RUN_TIMEOUT
// instead of hanging; the host then registers the pair again without runs.
constexpr long long RUN_TIMEOUT_TICKS = 100000000LL;   // 1 s of the 100 MHz wall clock
enum { RUN_V_STALL = 1, RUN_V_BUILD = 2, RUN_V_RELOAD = 4 };   // the head block's verdict on the slot that is running (RELOAD, side builds: the plan
                                                               // has changed lists -- the slot stands, and before the next one every block loads its
                                                               // candidates from the other buffer's record)
enum { RUN_GO = 1, RUN_ABORT = 2 };                    // ... and on a large run's entry (RunMail::entry_go)
constexpr long long RUN_ENTRY_TICKS = 20000LL;         // 200 us: how long the head block of a large run waits for its solvers to start

// One exchange among the g solver blocks: this block's NV sums -- thread t < 2 NV holds sum t >> 1 in my_val and sends its half of it
// (round 6: the threads that add the waves' rows send what they have added; no copy through LDS, no barrier in front; `row` < 0:
// this block only reads) -> tot[0..NV) = the sum of the g rows, added in ONE fixed order (four chains, then a tree).  All
// threads call it.  seq: the exchange's number (every block counts the same).  verdict_out (block-uniform, may be null): the
// head block's verdict word of this exchange is waited for as well and handed out.  Returns false on a time-out (block-uniform).
// after_post(): called by every thread once the block's sums are on their way -- work that hides behind the exchange's latency.
struct RunNoWork { __device__ __forceinline__ void operator()() const {} };
// Polling K tagged words per thread: sweeps of all K words until every tag matches.  A sweep that comes back incomplete costs a whole round
// trip to the L2 / the memory side before the next look (0.4-0.55 us, tools/micro/pingpong.hip).  TWO sweeps in flight -- the second issued half
// a round trip behind the first, each re-issued as soon as it has been looked at, taking turns without a register move (a move would wait for
// the load it moves) -- look twice as often and were measured SLOWER (-DCVO_RUN_POLL_GAP=8 against 0: cvo 3k 1 298 against 1 325, 6k 1 116 / 1 138,
// 10k 1 092 / 1 105, acvo 3k 1 232 / 1 272): the polls of 249 blocks are themselves the traffic the rows travel in.  Naps between sweeps of 0 / 1 / 4 / 12
// units (-DCVO_RUN_POLL_SLEEP): alike up to 4, -2...-4 % at 12 (profiles/r06_ab.txt 17).
// src(k): address of this thread's k-th word, nullptr for none.  Returns false on a time-out (this thread's; *s_fail is set).
#ifndef CVO_RUN_POLL_GAP
#define CVO_RUN_POLL_GAP 0   // s_sleep units (64 clocks) between the first two sweeps; 0 = one sweep in flight
#endif
#ifndef CVO_RUN_POLL_SLEEP
#define CVO_RUN_POLL_SLEEP 1 // s_sleep units between two sweeps of one thread (one sweep in flight)
#endif
template <int K, class SRC>
__device__ __forceinline__ bool run_poll(const SRC &src, const unsigned tag, const long long timeout_ticks, unsigned long long (&w)[K], int *s_fail)
{
    const unsigned long long none = (unsigned long long)tag << 32;
    auto sweep = [&](unsigned long long (&v)[K]) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const unsigned long long *p = src(k);
            v[k] = p ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : none;
        }
    };
    auto in = [&](const unsigned long long (&v)[K]) {
        bool ok = true;
#pragma unroll
        for (int k = 0; k < K; ++k) ok = ok && (unsigned)(v[k] >> 32) == tag;
        return ok;
    };
    const long long t0 = (long long)wall_clock64();
    if (CVO_RUN_POLL_GAP == 0) {
        for (;;) {
            sweep(w);
            if (in(w)) return true;
            if ((long long)wall_clock64() - t0 > timeout_ticks) { *s_fail = 1; return false; }
            __builtin_amdgcn_s_sleep(CVO_RUN_POLL_SLEEP);
        }
    }
    unsigned long long b[K];
    sweep(w);
    __builtin_amdgcn_s_sleep(CVO_RUN_POLL_GAP);
    sweep(b);
    for (;;) {
        if (in(w)) return true;
        if ((long long)wall_clock64() - t0 > timeout_ticks) { *s_fail = 1; return false; }
        sweep(w);
        if (in(b)) {
#pragma unroll
            for (int k = 0; k < K; ++k) w[k] = b[k];
            return true;
        }
        sweep(b);
    }
}
template <int NV, int KMAX, class AFTER>
__device__ __forceinline__ bool run_exchange_k(RunMail *mail, const int row, const int g, const unsigned long long seq, const double my_val,
                                             double *all /* LDS [RUN_G * NV] */, double *part /* LDS [8 * NV] */, double *tot /* LDS [NV] */,
                                             int *s_fail, unsigned *verdict_out, unsigned *s_verdict, const long long timeout_ticks, const AFTER &after_post)
{
    const int tid = threadIdx.x;
    const unsigned tag = (unsigned)seq;
    unsigned long long *slot = &mail->w[seq & (unsigned long long)(RUN_GEN - 1)][0][0];   // (four generations: cvo_device.h RunMail)
    // (*s_fail: cleared once, before the run's first exchange -- a run does not go on after a time-out)
    if (row >= 0 && tid < 2 * NV) {
        const unsigned long long bits = (unsigned long long)__double_as_longlong(my_val);
        const unsigned half = (tid & 1) ? (unsigned)(bits >> 32) : (unsigned)bits;
        __hip_atomic_store(&slot[row * (2 * RUN_NV) + tid], ((unsigned long long)tag << 32) | half, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    after_post();
    // thread t polls words t, t + 512, ...: ALL of a sweep requested before any is looked at (KMAX per thread: 2 up to 32 solvers,
    // 9 at 248)
    const int nrow = g * 2 * NV, nwords = nrow + (verdict_out ? 1 : 0);
    if constexpr (KMAX <= 2) {   // (the runs of up to RUN_G_SMALL solvers: two sweeps in flight, run_poll)
        unsigned long long w[KMAX];
        auto src = [&](const int k) -> const unsigned long long * {
            const int wi = tid + k * RUN_BLOCK;
            if (wi >= nwords) return nullptr;
            const int r = wi / (2 * NV), c = wi - r * (2 * NV);
            return wi == nrow ? &mail->w[(seq >> 1) & 1ull][RUN_G][0] : &slot[r * (2 * RUN_NV) + c];
        };
        if (run_poll<KMAX>(src, tag, timeout_ticks, w, s_fail)) {
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                const int wi = tid + k * RUN_BLOCK;
                if (wi < nrow) reinterpret_cast<unsigned *>(all)[wi] = (unsigned)w[k];
                else if (wi == nrow && verdict_out) *s_verdict = (unsigned)w[k];
            }
        }
    } else
    {
        const long long t0 = (long long)wall_clock64();
        bool all_in;
        do {
            unsigned long long w[KMAX];
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                const int wi = tid + k * RUN_BLOCK;
                // (the verdict: word 0 of the row behind the solvers'; its generation goes by the slot -- every second exchange
                // carries one, their numbers have one parity)
                const int r = wi / (2 * NV), c = wi - r * (2 * NV);
                const unsigned long long *src = wi == nrow ? &mail->w[(seq >> 1) & 1ull][RUN_G][0] : &slot[r * (2 * RUN_NV) + c];
                w[k] = wi < nwords ? __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((unsigned long long)tag << 32);
            }
            all_in = true;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) all_in = all_in && (unsigned)(w[k] >> 32) == tag;
            if (all_in) {
#pragma unroll
                for (int k = 0; k < KMAX; ++k) {
                    const int wi = tid + k * RUN_BLOCK;
                    if (wi < nrow) reinterpret_cast<unsigned *>(all)[wi] = (unsigned)w[k];   // (little endian: word 2k is the low half of value k)
                    else if (wi == nrow && verdict_out) *s_verdict = (unsigned)w[k];
                }
            } else if ((long long)wall_clock64() - t0 > timeout_ticks) {
                *s_fail = 1;
                break;
            } else {
                __builtin_amdgcn_s_sleep(1);
            }
        } while (!all_in);
    }
    __syncthreads();
    // one fixed order whatever the arrival order: eight chains per value (rows c, c + 8, ...), then a tree
    if (tid < 8 * NV) {
        const int k = tid % NV, c = tid / NV;
        double a = 0.0;
        for (int q = c; q < g; q += 8) a += all[q * NV + k];
        part[c * NV + k] = a;
    }
    __syncthreads();
    if (tid < NV)
        tot[tid] = ((part[tid] + part[NV + tid]) + (part[2 * NV + tid] + part[3 * NV + tid])) +
                   ((part[4 * NV + tid] + part[5 * NV + tid]) + (part[6 * NV + tid] + part[7 * NV + tid]));
    __syncthreads();
    if (verdict_out) *verdict_out = *s_verdict;
    return *s_fail == 0;
}
// The same exchange in TWO LEVELS, for the runs of more than RUN_G_SMALL solvers (round 6).  With everybody polling everybody's row an
// exchange among g blocks moves g^2 rows through the memory system -- 249 blocks x 36-51 KB per exchange at 248 solvers, 3.3-3.8 us against 2.2
// among 32 (tools/microbench/xcd_exchange.hip: 3.3 / 5.5 / 9.7 us among 64 / 128 / 256 blocks).  The one-level exchange adds the rows in
// RUN_CHAINS chains (rows c, c + 8, ...) and then a tree over the chains; here solver c -- the LEADER of chain c -- alone polls the rows of its
// chain (workgroups are dealt to the eight XCDs in turn: a chain's blocks share an XCD and its L2), adds them in the same order and posts the
// chain's sums (RunMail::p), and every block polls the eight chain sums and does the tree: g + 8 x 249 rows instead of 249 g, the same
// additions in the same order -- the same totals bit for bit.  Two hops instead of one.
template <int NV, class AFTER>
__device__ __forceinline__ bool run_exchange_hier(RunMail *mail, const int row, const int g, const unsigned long long seq, const double my_val,
                                                  double *all /* LDS [RUN_G * NV] */, double *part /* LDS [8 * NV] */, double *tot /* LDS [NV] */,
                                                  int *s_fail, unsigned *verdict_out, unsigned *s_verdict, const long long timeout_ticks, const AFTER &after_post)
{
    static_assert(RUN_CHAINS == 8 && RUN_CHAINS * 2 * NV + 1 <= RUN_BLOCK, "one word per thread at the second level");
    constexpr int KL = ((RUN_G + RUN_CHAINS - 1) / RUN_CHAINS * 2 * NV + RUN_BLOCK - 1) / RUN_BLOCK;   // words per thread of a leader's sweep
    const int tid = threadIdx.x;
    const unsigned tag = (unsigned)seq;
    const unsigned long long gen = seq & (unsigned long long)(RUN_GEN - 1);
    unsigned long long *slot = &mail->w[gen][0][0];
    unsigned long long *pslot = &mail->p[gen][0][0];
    if (row >= 0 && tid < 2 * NV) {
        const unsigned long long bits = (unsigned long long)__double_as_longlong(my_val);
        const unsigned half = (tid & 1) ? (unsigned)(bits >> 32) : (unsigned)bits;
        __hip_atomic_store(&slot[row * (2 * RUN_NV) + tid], ((unsigned long long)tag << 32) | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    after_post();
    if (row >= 0 && row < RUN_CHAINS) {   // ---- first level: the leader of chain `row` (block-uniform)
        const int nr = (g - row + RUN_CHAINS - 1) / RUN_CHAINS, nwords = nr * 2 * NV;   // rows row, row + 8, ... of the chain, in the order they are added
        unsigned long long w[KL];
        auto src = [&](const int k) -> const unsigned long long * {
            const int wi = tid + k * RUN_BLOCK;
            if (wi >= nwords) return nullptr;
            const int r = wi / (2 * NV), c = wi - r * (2 * NV);
            return &slot[(row + r * RUN_CHAINS) * (2 * RUN_NV) + c];
        };
        if (run_poll<KL>(src, tag, timeout_ticks, w, s_fail)) {
#pragma unroll
            for (int k = 0; k < KL; ++k) {
                const int wi = tid + k * RUN_BLOCK;
                if (wi < nwords) reinterpret_cast<unsigned *>(all)[wi] = (unsigned)w[k];   // (all[r * NV + k]: row `row + 8 r` of the chain)
            }
        }
        __syncthreads();
        if (tid < 2 * NV && *s_fail == 0) {   // (two threads per value: each adds the chain and sends its half of the sum)
            const int k = tid >> 1;
            double a = 0.0;
            for (int r = 0; r < nr; ++r) a += all[r * NV + k];
            const unsigned long long bits = (unsigned long long)__double_as_longlong(a);
            const unsigned half = (tid & 1) ? (unsigned)(bits >> 32) : (unsigned)bits;
            __hip_atomic_store(&pslot[row * (2 * RUN_NV) + tid], ((unsigned long long)tag << 32) | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    {   // ---- second level: everybody, the eight chain sums (and the verdict word)
        const int nrow = RUN_CHAINS * 2 * NV, nwords = nrow + (verdict_out ? 1 : 0);
        unsigned long long w[1];
        auto src = [&](const int) -> const unsigned long long * {
            if (tid >= nwords) return nullptr;
            const int r = tid / (2 * NV), c = tid - r * (2 * NV);
            return tid == nrow ? &mail->w[(seq >> 1) & 1ull][RUN_G][0] : &pslot[r * (2 * RUN_NV) + c];
        };
        if (run_poll<1>(src, tag, timeout_ticks, w, s_fail)) {
            if (tid < nrow) reinterpret_cast<unsigned *>(part)[tid] = (unsigned)w[0];   // (part[c * NV + k], as the one-level exchange lays it out)
            else if (tid == nrow && verdict_out) *s_verdict = (unsigned)w[0];
        }
    }
    __syncthreads();
    if (tid < NV)
        tot[tid] = ((part[tid] + part[NV + tid]) + (part[2 * NV + tid] + part[3 * NV + tid])) +
                   ((part[4 * NV + tid] + part[5 * NV + tid]) + (part[6 * NV + tid] + part[7 * NV + tid]));
    __syncthreads();
    if (verdict_out) *verdict_out = *s_verdict;
    return *s_fail == 0;
}
// solvers from which an exchange runs in two levels: 65 for cvo, 33 for acvo (whose flow exchange carries 13 sums against 9).  One registration at a
// time, registrations/s at 3k / 6k / 10k / 14k points, never / from 33 / from 65 / from 129 solvers (profiles/r06_ab.txt 15):
//   cvo  1 281 1 071 1 036 627 / 1 309 1 115 1 082 668 / 1 325 1 122 1 098 670 / 1 304 1 116 1 091 653
//   acvo 1 249 1 054   799 774 / 1 268 1 076   849 856 / 1 258 1 090   842 845 / 1 268 1 089   808 851
#ifndef CVO_RUN_HIER_CVO
#define CVO_RUN_HIER_CVO 65     // (A/B builds: 9999 = never)
#endif
#ifndef CVO_RUN_HIER_ACVO
#define CVO_RUN_HIER_ACVO 33
#endif
template <int NV, int HIER_FROM, class AFTER = RunNoWork>
__device__ __forceinline__ bool run_exchange(RunMail *mail, const int row, const int g, const unsigned long long seq, const double my_val,
                                             double *all, double *part, double *tot, int *s_fail, unsigned *verdict_out, unsigned *s_verdict,
                                             const long long timeout_ticks, const AFTER &after_post = AFTER())
{
    static_assert(RUN_G_SMALL * 2 * NV + 1 <= 2 * RUN_BLOCK, "two words per thread up to RUN_G_SMALL solvers");
    constexpr int KBIG = (RUN_G * 2 * NV + 1 + RUN_BLOCK - 1) / RUN_BLOCK;
    if (g <= RUN_G_SMALL) return run_exchange_k<NV, 2>(mail, row, g, seq, my_val, all, part, tot, s_fail, verdict_out, s_verdict, timeout_ticks, after_post);
    if (g >= HIER_FROM) return run_exchange_hier<NV>(mail, row, g, seq, my_val, all, part, tot, s_fail, verdict_out, s_verdict, timeout_ticks, after_post);
    return run_exchange_k<NV, KBIG>(mail, row, g, seq, my_val, all, part, tot, s_fail, verdict_out, s_verdict, timeout_ticks, after_post);
}

// The passes of a run over NR candidates per lane, straight-line: the NR chains (transform, exact test, a float64 exp, the
// sums' terms) are independent and a lane has nothing else to hide their latency with -- two waves per SIMD --, so there is no
// branch between them for the scheduler to stop at.  A candidate that is no member has w = 0 and every one of its terms is an
// exact zero (0 x finite, summed into a float64: acc + 0 = acc), which is also what a lane without a candidate holds (ck = 0):
// nothing is skipped and nothing changes.  (With a uniform branch per candidate a round cost ~1 600 ticks of a wave's time,
// profiles/r05_ab.txt 6.)
template <int NR, int NA>
__device__ __forceinline__ unsigned run_flow_rounds(const float (&rt)[12], const KernConsts &kc, const float (&cx)[NA][3],
                                                     const float (&cy)[NA][3], const float (&cck)[NA], float (&cw)[NA],
                                                     const double *etab, const int need_d2, double (&acc)[NACC_FLOW])
{
    unsigned nk = 0;
    asm volatile("; run_flow_rounds begin %0" ::"n"(NR));
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const float4 xi = make_float4(cx[r][0], cx[r][1], cx[r][2], 0.0f);
        const float4 yj = apply_tf(rt, rt + 9, make_float4(cy[r][0], cy[r][1], cy[r][2], 0.0f));
        const float e0 = xi.x - yj.x, e1 = xi.y - yj.y, e2 = xi.z - yj.z;
        const float d2 = __builtin_fmaf(e2, e2, __builtin_fmaf(e1, e1, e0 * e0));
        const float ck = cck[r];
        // (weight_from_ck's own comparison, as a select: the exp of a pair outside tau is formed and dropped)
        const float a = ck * (float)(kc.s2_d * exp_neg((double)d2 * kc.ninv_2l2, etab));
        const float w = (d2 < kc.tau && ck > 0.0f && a > kc.sp) ? a : 0.0f;
        cw[r] = w;
        pair_flow_sums(kc, xi, yj, w, d2, need_d2, acc);
        nk += (unsigned)__popcll(__ballot(w > 0.0f));
    }
    asm volatile("; run_flow_rounds end %0" ::"n"(NR));
    return nk;
}
template <int NR, int NA>
__device__ __forceinline__ void run_step_rounds(const float (&rt)[12], const KernConsts &kc, const cvo_math::XiConsts &xc,
                                                const float (&cx)[NA][3], const float (&cy)[NA][3], const float (&cw)[NA],
                                                double (&sacc)[NACC_STEP])
{
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const float4 yj = apply_tf(rt, rt + 9, make_float4(cy[r][0], cy[r][1], cy[r][2], 0.0f));
        pair_step_sums(kc, xc, yj, cx[r][0] - yj.x, cx[r][1] - yj.y, cx[r][2] - yj.z, cw[r], sacc);
    }
}

// acvo (kt_run_acvo): the xx and yy candidates of a lane (ref src/adaptive_cvo.cpp:157-158,213-265 -- se_kernel on (x, x) and on
// the transformed (y, y), the dl sums of their rows), the arithmetic of eval_pair<PROC_SELF, 0, 2> pair for pair.
//   xx: x never moves -- a candidate is its d2 and its colour weight (the sign: does the row count, the Ayy rule's twin); per
//       iteration one exp, the cut, the sum.  (Its sums are a function of the length scale alone: the caller keeps them while
//       that stands still.)
//   yy: the reference forms |y_i - y_j|^2 from the TRANSFORMED cloud of the iteration, so its roundings -- and with them
//       membership at the cut -- move with the transform: both points stay in registers and are transformed every time.
template <int NR>
__device__ __forceinline__ unsigned run_xx_rounds(const KernConsts &kc, const float (&xd2)[RUN_A], const float (&xck)[RUN_A], const double *etab,
                                                   double &sum)
{
    unsigned nk = 0;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const float d2 = xd2[r], ck = __builtin_fabsf(xck[r]);
        const float a = ck * (float)(kc.s2_d * exp_neg((double)d2 * kc.ninv_2l2, etab));
        const float w = (d2 < kc.tau && ck > 0.0f && a > kc.sp) ? a : 0.0f;
        // (a row that does not count, a pair that is no member: an exact zero)
        sum += (double)((kc.inv_l3 * ((xck[r] < 0.0f) ? 0.0f : w)) * d2);
        nk += (unsigned)__popcll(__ballot(w > 0.0f));
    }
    return nk;
}
template <int NR>
__device__ __forceinline__ unsigned run_yy_rounds(const float (&rt)[12], const KernConsts &kc, const float (&ya)[RUN_A][3], const float (&yb)[RUN_A][3],
                                                   const float (&yck)[RUN_A], const double *etab, double &sum)
{
    unsigned nk = 0;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const float4 yi = apply_tf(rt, rt + 9, make_float4(ya[r][0], ya[r][1], ya[r][2], 0.0f));
        const float4 yj = apply_tf(rt, rt + 9, make_float4(yb[r][0], yb[r][1], yb[r][2], 0.0f));
        const float e0 = yi.x - yj.x, e1 = yi.y - yj.y, e2 = yi.z - yj.z;
        const float d2 = __builtin_fmaf(e2, e2, __builtin_fmaf(e1, e1, e0 * e0));
        const float ck = __builtin_fabsf(yck[r]);
        const float a = ck * (float)(kc.s2_d * exp_neg((double)d2 * kc.ninv_2l2, etab));
        const float w = (d2 < kc.tau && ck > 0.0f && a > kc.sp) ? a : 0.0f;
        sum += (double)((kc.inv_l3 * ((yck[r] < 0.0f) ? 0.0f : w)) * d2);
        nk += (unsigned)__popcll(__ballot(w > 0.0f));
    }
    return nk;
}

__global__ void kt_run(const Slot *__restrict__ tab, const int qs);
__global__ void kt_run_side(const Slot *__restrict__ tab, const int qs);
// ... and over the NL candidates per lane that live in LDS behind the registers' (the widest runs: RUN_L): lane t's candidate of
// round r as two 16-byte pieces, [x0 x1 x2 ck] at lc[(2 r) * RUN_BLOCK + t] and [y0 y1 y2 w] behind it -- every lane reads and
// writes its own pieces only (no barrier), a wave's accesses are consecutive (no bank conflict).  The same arithmetic per pair.
__device__ __forceinline__ unsigned run_flow_lds(const int nl, float4 *lc, const float (&rt)[12], const KernConsts &kc, const double *etab,
                                                 const int need_d2, double (&acc)[NACC_FLOW])
{
    // (two rounds per trip, independent chains side by side; nl is even: the loader fills an odd last round with lanes without a candidate)
    unsigned nk = 0;
#pragma unroll 1
    for (int r = 0; r < nl; r += 2) {
        float w[2], d2[2];
        float4 xi[2], yj[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const float4 a4 = lc[(2 * (r + u)) * RUN_BLOCK], b4 = lc[(2 * (r + u) + 1) * RUN_BLOCK];
            xi[u] = make_float4(a4.x, a4.y, a4.z, 0.0f);
            yj[u] = apply_tf(rt, rt + 9, make_float4(b4.x, b4.y, b4.z, 0.0f));
            const float e0 = xi[u].x - yj[u].x, e1 = xi[u].y - yj[u].y, e2 = xi[u].z - yj[u].z;
            d2[u] = __builtin_fmaf(e2, e2, __builtin_fmaf(e1, e1, e0 * e0));
            const float ck = a4.w;
            const float a = ck * (float)(kc.s2_d * exp_neg((double)d2[u] * kc.ninv_2l2, etab));
            w[u] = (d2[u] < kc.tau && ck > 0.0f && a > kc.sp) ? a : 0.0f;
            reinterpret_cast<float *>(&lc[(2 * (r + u) + 1) * RUN_BLOCK])[3] = w[u];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            pair_flow_sums(kc, xi[u], yj[u], w[u], d2[u], need_d2, acc);
            nk += (unsigned)__popcll(__ballot(w[u] > 0.0f));
        }
    }
    return nk;
}
__device__ __forceinline__ void run_step_lds(const int nl, const float4 *lc, const float (&rt)[12], const KernConsts &kc,
                                             const cvo_math::XiConsts &xc, double (&sacc)[NACC_STEP])
{
#pragma unroll 1
    for (int r = 0; r < nl; r += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const float4 a4 = lc[(2 * (r + u)) * RUN_BLOCK], b4 = lc[(2 * (r + u) + 1) * RUN_BLOCK];
            const float4 yj = apply_tf(rt, rt + 9, make_float4(b4.x, b4.y, b4.z, 0.0f));
            pair_step_sums(kc, xc, yj, a4.x - yj.x, a4.y - yj.y, a4.z - yj.z, b4.w, sacc);
        }
    }
}

unsigned run_grid() { return 1u + RUN_G; }
// (more than 64 KB of dynamic LDS per block must be asked for, once per device)
hipError_t run_allow_lds()
{
    static std::atomic<unsigned char> allowed[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (allowed[dev].load(std::memory_order_acquire)) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kt_run), hipFuncAttributeMaxDynamicSharedMemorySize, (int)RUN_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(kt_run_side), hipFuncAttributeMaxDynamicSharedMemorySize, (int)RUN_LDS_BYTES);
    if (e == hipSuccess) allowed[dev].store(1, std::memory_order_release);
    return e;
}

// ACVO (kt_run_acvo, round 6): the same for an adaptive-cvo registration (ref src/adaptive_cvo.cpp:490-555) -- three candidate sets on
// chip (xy as above; xx and yy: run_xx_rounds / run_yy_rounds), up to RUN_A per lane each and all in registers (acvo's length scale
// falls to its floor within a few iterations: its records are narrow), 13 sums in the first exchange (the nine of the flow side,
// sum and count of Axx, tail sum and count of Ayy), dl and the length-scale update in every block's head_post, kernel constants
// made again in every block whenever the length scale has moved, a build of ANY of the three lists ends the run.
// SIDE: the kernels with side builds (kt_run_side, kt_run_acvo_side: kt_run "side builds"; an option) are instantiations of their own -- with
// that code in, the compiler spilled 123 vector registers of the plain kernels instead of 19 and every run was 6 % slower.
template <bool ACVO, bool SIDE = false>
__device__ __forceinline__ void run_body(const Slot *__restrict__ tab, const int qs, float4 *const s_lc)
{
    __shared__ unsigned long long s_ticket;
    constexpr int NVF = ACVO ? 13 : NACC_FLOW;   // doubles of the flow-side exchange
    constexpr int HIER_FROM = ACVO ? CVO_RUN_HIER_ACVO : CVO_RUN_HIER_CVO;   // (run_exchange)
    const bool head_block = blockIdx.x == 0;
    const int srow = (int)blockIdx.x - 1;   // a solver's row in the exchanges (-1: the head block)
    CSlot cs = (CSlot)(tab);
    if (cs->active == 0) return;
    // (every block of every launch that gets this far draws a ticket, whatever it does next: RunMail::entry_ticket)
    {
        RunMail *const m0 = CVO_ARG(PostStepArgs, op[qs & 15].ps).run_mail;
        if (m0 == nullptr) return;
        // (a launch adds 1 + RUN_G in all whatever its grid -- block 0 of a launch of fewer blocks draws the difference as well --, so
        // that ticket / (1 + RUN_G) numbers the launches)
        const unsigned long long draw = blockIdx.x == 0 ? (unsigned long long)(1 + RUN_G) - (unsigned long long)(gridDim.x - 1u) : 1ull;
        if (threadIdx.x == 0) s_ticket = __hip_atomic_fetch_add(&m0->entry_ticket, draw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const int qf = qs & 15, qt = (qs >> 4) & 15;
    const ProcessArgs &pa = CVO_ARG(ProcessArgs, op[qf].p);     // the flow pass of the plan's classic launches
    // (its head's arguments, read from the table where they are needed: held by value -- ~60 scalar registers for the life of the kernel --
    // kt_run spilled 7 vector registers and kt_run_acvo 35, to 32 / 80 B of scratch; by reference neither spills any: acvo 3k 1 217 -> 1 247 /s,
    // 10k 761 -> 798, cvo +0.5 %; -DCVO_RUN_PS_VALUE brings the copy back, profiles/r06_ab.txt 13)
#ifdef CVO_RUN_PS_VALUE
    const PostStepArgs ps = CVO_ARG(PostStepArgs, op[qf].ps);
#else
    const PostStepArgs &ps = CVO_ARG(PostStepArgs, op[qf].ps);
#endif
    const ProcessArgs &ta = CVO_ARG(ProcessArgs, op[qt].p);     // its step launch (the trace)
    const ProcessArgs &xa = CVO_ARG(ProcessArgs, op[ACVO ? qf + 1 : qf].p);   // acvo: the self passes of the plan's flow launch (xx, yy)
    const ProcessArgs &ya = CVO_ARG(ProcessArgs, op[ACVO ? qf + 2 : qf].p);
    DevState *const gst = ps.st;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);

    __shared__ __attribute__((aligned(16))) DevHead s_st;
    __shared__ double s_red[RUN_WAVES * NACC_MAX];
    __shared__ double s_self[RUN_WAVES * 4];   // acvo: the waves' xx / yy sums (sum, count, tail sum, count)
    __shared__ double s_xx_keep[2];            // acvo: this block's Axx sums ...
    __shared__ float s_xx_ell;                 // ... and the length scale they were made at (they are a function of nothing else)
    // (the exchanges' rows and the record's prefix sums share their LDS: the prefix sums are dead once the candidates are loaded,
    // and every exchange begins with a barrier)
    constexpr size_t S_ALL_BYTES = sizeof(double) * RUN_G * NVF, S_PREF_BYTES = sizeof(unsigned) * (PROC_WAVES + 1);
    __shared__ __attribute__((aligned(16))) unsigned char s_union[S_ALL_BYTES > S_PREF_BYTES ? S_ALL_BYTES : S_PREF_BYTES];
    double *const s_all = reinterpret_cast<double *>(s_union);
    unsigned *const s_pref = reinterpret_cast<unsigned *>(s_union);
    // (acvo: the prefix sums of the self records beside it -- its blocks keep no candidates in LDS and have the room)
    __shared__ unsigned s_pref_self[ACVO ? 2 * (PROC_WAVES + 1) : 2];
    __shared__ double s_part[8 * NVF];
    __shared__ double s_tot[NVF + 4];
    // side builds (head block): the buffer whose build is in flight beside the run (-1: none), handed = it has ended and the plan has been
    // told (xy_fresh), the count of RunMail::side_done that says it has ended, verdict bits the coming slot gets on top of the plan's
    __shared__ int s_side_t, s_side_handed, s_side_fail, s_side_count;
    __shared__ unsigned s_side_total;
    __shared__ unsigned long long s_side_expect;
    __shared__ unsigned s_vbits;
    __shared__ cvo_math::ExpPre s_pre;   // the twist-only stage of the head's Exp_SEK3 / dist_se3 / stop test, formed behind the step exchange
    __shared__ double s_dl;   // acvo: dl of the slot (ref src/adaptive_cvo.cpp:271), every block's own
    __shared__ double s_etab[64];
    __shared__ cvo_math::XiConsts s_xi;
    __shared__ float s_wm[12];
    __shared__ int s_fail;
    __shared__ unsigned s_verdict;
    __shared__ double s_bak[NACC_FLOW];
    __shared__ float s_bakf[6];
    __shared__ unsigned s_wsum[RUN_WAVES];
    __shared__ unsigned long long s_seq;

#ifdef CVO_RUN_CLOCKS   // (A/B builds: where block 1's time goes, DevState::run_clk)
    long long clk_t = (long long)__builtin_readcyclecounter(), clk_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define RUN_CLK(i) do { const long long n_ = (long long)__builtin_readcyclecounter(); clk_acc[i] += n_ - clk_t; clk_t = n_; } while (0)
#else
#define RUN_CLK(i) do { } while (0)
#endif
    // ---- entry: one round trip for the head, the previous slot's step sums, its overflow flags and the slice
    // counts of both records
    double sp[NACC_STEP];
    thread_load_partials<NACC_STEP, PROC_BLOCKS / STEP_TWIST_ROWS_DIV>(ps.part_step, ps.nblk, sp);   // (threads >= BLOCK read rows beyond: guarded)
    const unsigned my_flag = gst->ovf[1][tid & 7];
    const int nsl = 4 * pa.nblk;   // slices of a record (waves of the pass that wrote it)
    constexpr int PER = PROC_WAVES / RUN_BLOCK;   // 8 slices per thread
    unsigned cnt_a[PER], cnt_b[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int sl = tid * PER + q;
        cnt_a[q] = (sl < nsl && pa.cand_cnt) ? pa.cand_cnt[sl] : 0u;
        cnt_b[q] = (sl < nsl && pa.cand_cnt_b) ? pa.cand_cnt_b[sl] : 0u;
    }
    // (acvo: the records of the self lists, both buffers of each -- the head says which is in use)
    unsigned cxx_a[ACVO ? PER : 1], cxx_b[ACVO ? PER : 1], cyy_a[ACVO ? PER : 1], cyy_b[ACVO ? PER : 1];
    const int nsl_x = ACVO ? 4 * xa.nblk : 0, nsl_y = ACVO ? 4 * ya.nblk : 0;
    if constexpr (ACVO) {
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int sl = tid * PER + q;
            cxx_a[q] = (sl < nsl_x && xa.cand_cnt) ? xa.cand_cnt[sl] : 0u;
            cxx_b[q] = (sl < nsl_x && xa.cand_cnt_b) ? xa.cand_cnt_b[sl] : 0u;
            cyy_a[q] = (sl < nsl_y && ya.cand_cnt) ? ya.cand_cnt[sl] : 0u;
            cyy_b[q] = (sl < nsl_y && ya.cand_cnt_b) ? ya.cand_cnt_b[sl] : 0u;
        }
    }
    if (tid == 0) s_seq = gst->run_seq;
    if (tid < 64) s_etab[tid] = c_exp2_64[tid];
    state_head_to_lds(gst, &s_st);   // (with its barrier)
    // the head block alone tells the host that this run is over (whatever way it ends)
    // (not once the loop has stopped: the host leaves on the `done` word, which went out before, and may have begun the next
    // registration -- and reset this mirror -- by now)
    // (the mirror's word: the count, and RUN_MIRROR_ENTERED if this run carried slots)
    bool entered = false, aborted = false;
    auto run_over = [&]() {
        if (head_block && tid == 0) {
            const int c = gst->run_count + 1;
            gst->run_count = c;
            // (a batch may bring two runs: the word of the second also says what the first did)
            if (entered) gst->run_last_entered = c;
            if (aborted) gst->run_last_aborted = c;
            const bool e2 = entered || (c > 1 && gst->run_last_entered == c - 1), a2 = aborted || (c > 1 && gst->run_last_aborted == c - 1);
            if (ps.run_mirror && s_st.done == RUNNING) *ps.run_mirror = c | (e2 ? RUN_MIRROR_ENTERED : 0) | (a2 ? RUN_MIRROR_ABORTED : 0);
        }
    };
    // what would make this launch a plain head-mode flow launch's business: a loop that has stopped, a stall slot,
    // a build this launch's filter blocks would have to make
    // (... or a record that is expected to hold far more than a run's registers: decided before the head's maths, cheaply)
    if (s_st.done != RUNNING || s_st.stall != 0 || s_st.xy_target >= 0 || pa.cand == nullptr || pa.cand_b == nullptr ||
        ps.run_mail == nullptr || s_st.run_hint > 2 * RUN_CAP ||
        // (a build has just ended: the plan is about to switch to a list that has no record yet; or the list in use has none)
        // (the pass of a pending slot has recorded the list it read: the entry head is about to say so, head_plan)
        s_st.xy_fresh >= 0 ||
        ((s_st.xy_active ? s_st.xy_ck[1] : s_st.xy_ck[0]) != pa.nblk && !(s_st.pending != 0 && ps.ck_nblk[LIST_XY] == pa.nblk)) ||
        (ACVO && (s_st.sf_fresh[0] >= 0 || s_st.sf_fresh[1] >= 0 || s_st.run_hint > (int)(1.25f * (float)(RUN_LANES * RUN_A)) ||
                  ((s_st.sf_active[0] ? s_st.sf_ck[0][1] : s_st.sf_ck[0][0]) != xa.nblk && !(s_st.pending != 0 && ps.ck_nblk[LIST_XX] == xa.nblk)) ||
                  ((s_st.sf_active[1] ? s_st.sf_ck[1][1] : s_st.sf_ck[1][0]) != ya.nblk && !(s_st.pending != 0 && ps.ck_nblk[LIST_YY] == ya.nblk)))) ||
        (ACVO && (s_st.sf_target[0] >= 0 || s_st.sf_target[1] >= 0 || xa.cand == nullptr || xa.cand_b == nullptr || ya.cand == nullptr ||
                  ya.cand_b == nullptr || 4 * xa.nblk > PROC_WAVES || 4 * ya.nblk > PROC_WAVES))) {
#ifdef CVO_RUN_WHY
        if (head_block && tid == 0) gst->run_clk[s_st.done != RUNNING ? 0 : (s_st.stall != 0 ? 1 : (s_st.xy_target >= 0 ? 2 : 3))] += 1;
#endif
        run_over(); return; }
    const unsigned long long seq0 = s_seq;
    unsigned nexch = 0;

    // the head of the first slot, whole, in every block: post-step part of the slot that ended (if one is pending), plan of this one
    {
        const bool pending = s_st.pending != 0;
        if (pending) {
            // (the partial rows: BLOCK threads hold them, reduce as head_body does -- same order, same sums)
            if (tid < BLOCK) {
                wave_sums<NACC_STEP>(sp, lane, s_red + wid * NACC_MAX);
            }
            __syncthreads();
            if (tid == 0) {
#pragma unroll
                for (int k = 0; k < NACC_STEP; ++k)
                    s_st.red[RED_STEP + k] = ((s_red[k] + s_red[NACC_MAX + k]) + s_red[2 * NACC_MAX + k]) + s_red[3 * NACC_MAX + k];
            }
            __syncthreads();
        }
        if (tid < 64) {
            unsigned flag[LIST_N];
#pragma unroll
            for (int l = 0; l < LIST_N; ++l) flag[l] = (unsigned)__builtin_amdgcn_readlane((int)my_flag, l);
            long long clk[4] = {0, 0, 0, 0};
            head_math<HM_HEAD>(&s_st, ps, pending, false, flag, head_block, false, clk);
        }
        __syncthreads();
    }
    // can the slot that begins run here?  (the loop is running on a buffer whose record is current and fits)
    const int act = s_st.xy_active ? 1 : 0;
    bool ok = s_st.done == RUNNING && s_st.stall == 0 && (act ? s_st.xy_ck[1] : s_st.xy_ck[0]) == pa.nblk;
    const int act_x = (ACVO && s_st.sf_active[0]) ? 1 : 0, act_y = (ACVO && s_st.sf_active[1]) ? 1 : 0;
    if (ACVO)
        ok = ok && (act_x ? s_st.sf_ck[0][1] : s_st.sf_ck[0][0]) == xa.nblk && (act_y ? s_st.sf_ck[1][1] : s_st.sf_ck[1][0]) == ya.nblk;
    // a record's slices, numbered flat: s_pref[s] = candidates in front of slice s (thread t holds the counts of slices
    // t PER .. t PER + PER - 1); returns the record's total.  Two barriers; s_pref is whole when it returns.
    auto prefix = [&](const unsigned (&cnt)[PER], const unsigned wcap, unsigned *const dst) -> unsigned {
        unsigned mine[PER], sum = 0;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const unsigned c = cnt[q];
            mine[q] = sum;
            sum += c < wcap ? c : wcap;
        }
        unsigned inc = sum;   // inclusive scan over the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned o = (unsigned)__shfl_up((int)inc, off, 64);
            if (lane >= off) inc += o;
        }
        __syncthreads();   // (whoever still reads s_pref / s_wsum of the scan before)
        if (lane == 63) s_wsum[wid] = inc;
        __syncthreads();
        unsigned base = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < RUN_WAVES; ++w) {
            const unsigned v = s_wsum[w];
            if (w < wid) base += v;
            tot += v;
        }
        base += inc - sum;
#pragma unroll
        for (int q = 0; q < PER; ++q) dst[tid * PER + q] = base + mine[q];
        if (tid == 0) dst[PROC_WAVES] = tot;
        __syncthreads();
        return tot;
    };
    // (acvo: the counts of the self records' buffers in use; their prefix sums have arrays of their own)
    unsigned sel_x[ACVO ? PER : 1], sel_y[ACVO ? PER : 1];
    unsigned total_x = 0, total_y = 0;
    if constexpr (ACVO) {
#pragma unroll
        for (int q = 0; q < PER; ++q) { sel_x[q] = act_x ? cxx_b[q] : cxx_a[q]; sel_y[q] = act_y ? cyy_b[q] : cyy_a[q]; }
        total_x = prefix(reinterpret_cast<const unsigned (&)[PER]>(sel_x), xa.kept_wcap, s_pref_self);
        total_y = prefix(reinterpret_cast<const unsigned (&)[PER]>(sel_y), ya.kept_wcap, s_pref_self + (PROC_WAVES + 1));
    }
    unsigned total = 0;
    {
        unsigned sel[PER];
#pragma unroll
        for (int q = 0; q < PER; ++q) sel[q] = act ? cnt_b[q] : cnt_a[q];
        total = prefix(sel, pa.kept_wcap, s_pref);
    }
    if (head_block && tid == 0) gst->run_candidates = (int32_t)total;
    if (ok && tid == 0) { if (act) s_st.rec_count[1] = (int32_t)total; else s_st.rec_count[0] = (int32_t)total; }   // (every block alike; the record is current)
    if (total > (unsigned)RUN_CAP || total == 0u) ok = false;
    // (acvo: the widest of the three records decides)
    const unsigned total_max = ACVO ? (total > total_x ? (total > total_y ? total : total_y) : (total_x > total_y ? total_x : total_y)) : total;
    if (ACVO && total_max > (unsigned)(RUN_LANES * RUN_A)) ok = false;
#ifdef CVO_RUN_WHY
    if (!ok && head_block && tid == 0)
        gst->run_clk[s_st.done != RUNNING ? 4 : (s_st.stall != 0 ? 5 : ((act ? s_st.xy_ck[1] : s_st.xy_ck[0]) != pa.nblk ? 6 : (total == 0u ? 8 : 7)))] += 1;
#endif
    if (!ok) { run_over(); return; }   // (nothing has been written: the classic launches behind this one do the same head again)
    // Solvers: as few as give every lane one candidate up to 32 (an exchange among 8 blocks costs less than among 32 in isolation, but a
    // second candidate per lane costs a pass more than that saves: profiles/r05_ab.txt 1, 6); above, the exchange grows with the blocks
    // (2.3 / 2.9 / 3.9 us among 64 / 128 / 256) and a candidate per lane costs ~0.45 us of the two passes: up to three per lane, then the
    // next size, the whole GPU for the widest records
    const unsigned per = (unsigned)RUN_BLOCK;
    const int gdev = ps.run_g_max >= 8 && ps.run_g_max <= RUN_G ? ps.run_g_max : RUN_G;   // (fewer compute units: smaller runs)
    const int gmax = gdev < (int)gridDim.x - 1 ? gdev : (int)gridDim.x - 1;                // (a launch for a narrow record brings fewer blocks)
    // (acvo: a lane's candidates come in threes and every iteration transforms five points for them: two per lane, then the next size)
#ifndef CVO_ACVO_PER_LANE
#define CVO_ACVO_PER_LANE 3u   // (2 / 3 / 4: 10k 740 / 756 / 755 registrations/s, 6k 981 / 982 / 953, 3k alike: profiles/r06_ab.txt 3)
#endif
#ifndef CVO_RUN_PER_LANE_HI
#define CVO_RUN_PER_LANE_HI_CVO 3u
#define CVO_RUN_PER_LANE_HI_ACVO 1u   // (above 32 solvers an exchange runs in two levels and hardly grows with the blocks: acvo 6k 1 083-1 086 -> 1 096-1 102 /s, seed 1001
                                      // 978-992 -> 1 003-1 010, the other sizes alike; cvo, whose 33-64 solvers exchange in one level, keeps 3: profiles/r06_ab.txt 19)
#else
#define CVO_RUN_PER_LANE_HI_CVO CVO_RUN_PER_LANE_HI
#define CVO_RUN_PER_LANE_HI_ACVO CVO_RUN_PER_LANE_HI
#endif
    const int gwant = ACVO ? (total_max <= 8u * per ? 8 : (total_max <= 16u * per ? 16 : (total_max <= CVO_ACVO_PER_LANE * 32u * per ? 32 :
                              (total_max <= CVO_RUN_PER_LANE_HI_ACVO * 64u * per ? 64 : (total_max <= CVO_RUN_PER_LANE_HI_ACVO * 128u * per ? 128 : RUN_G)))))
                           : (total <= 8u * per ? 8 : (total <= 16u * per ? 16 : (total <= 3u * 32u * per ? 32 :
                              (total <= CVO_RUN_PER_LANE_HI_CVO * 64u * per ? 64 : (total <= CVO_RUN_PER_LANE_HI_CVO * 128u * per ? 128 : RUN_G)))));
    const int g = gwant < gmax ? gwant : gmax;
    if (total_max > (unsigned)g * per * (unsigned)(ACVO ? RUN_A : RUN_R + RUN_L)) { run_over(); return; }   // (block-uniform; nothing has been written)
    if (srow >= g) return;
    {
        // ---- entry hand-shake (RunMail::entry_ticket): nothing is written before all of the run is known to be resident.  Every run,
        // the small ones too: a block takes a whole compute unit (256 registers per lane, 150 KB of LDS) and spins for its peers, so
        // runs of several registrations -- host threads, processes, the few registrations of a small cvo_hip_align_many call -- whose
        // blocks together outnumber the compute units would keep each other's missing blocks off the GPU for ever (seen: two
        // processes x eight 3k registrations, every exchange timing out after its second)
        const unsigned long long nb = 1ull + RUN_G, launch = s_ticket / nb, base = launch * nb;   // (nb: what every launch draws in all)
        if (head_block && tid == 0) {
            const long long t0 = (long long)wall_clock64();
            unsigned go = RUN_GO;
            // (ALL blocks of the launch, not the g + 1 that take part: blocks are dealt to the eight XCDs in turn and each XCD starts
            // its share when it has room -- with other spinning kernels about, g + 1 tickets were seen drawn while a solver of
            // the run had not started; a block that does not take part leaves at once and frees its unit)
            while (__hip_atomic_load(&ps.run_mail->entry_ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < base + nb) {
                if ((long long)wall_clock64() - t0 > RUN_ENTRY_TICKS) { go = RUN_ABORT; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            __hip_atomic_store(&ps.run_mail->entry_go, ((launch + 1ull) << 32) | go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (launch + 1: never the zero the mail starts with)
            s_verdict = go;
        } else if (tid == 0) {
            const long long t0 = (long long)wall_clock64();
            unsigned long long w;
            unsigned go = RUN_ABORT;
            for (;;) {
                w = __hip_atomic_load(&ps.run_mail->entry_go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((w >> 32) == ((launch + 1ull) & 0xffffffffull)) { go = (unsigned)w; break; }
                if ((long long)wall_clock64() - t0 > RUN_TIMEOUT_TICKS) break;   // (the head block started first: cannot happen)
                __builtin_amdgcn_s_sleep(2);
            }
            s_verdict = go;
        }
        __syncthreads();
        if (s_verdict != RUN_GO) { aborted = true; run_over(); return; }
        __syncthreads();
    }
    const unsigned lanes = (unsigned)g * RUN_BLOCK;
    entered = true;   // (from here on the run executes at least one slot)

    // ---- the candidates of this lane: c = l + r * lanes (load_xy: at entry; with side builds again whenever the plan has changed lists --
    // act_cur / tot_cur: the xy buffer in use and its record's candidates, s_pref: that record's prefix sums)
    int act_cur = act;
    unsigned tot_cur = total;
    int rmax = 0, nl = 0;
    constexpr int NRX = ACVO ? RUN_A : RUN_R;   // xy candidates of a lane in registers
    float cx[NRX][3], cy[NRX][3], cck[NRX], cw[NRX];
    float4 *const lc = s_lc + tid;
    auto load_xy = [&]() {
    const unsigned total = tot_cur;
    const int act = act_cur;
    const unsigned wave_first = head_block ? total : (unsigned)srow * RUN_BLOCK + (unsigned)wid * 64u;   // flat number of the wave's first candidate of round 0
    rmax = wave_first < total ? (int)((total - wave_first + lanes - 1) / lanes) : 0;   // wave-uniform: rounds with any candidate
    {
        const uint2 *rec = act ? pa.cand_b : pa.cand;
        const unsigned wcap = pa.kept_wcap;
        uint2 e[NRX];
#pragma unroll
        for (int r = 0; r < NRX; ++r) {
            e[r] = make_uint2(0u, 0u);
            if (r < rmax) {
                const unsigned c0 = wave_first + (unsigned)r * lanes;   // < total
                // slice of c0: the last s with s_pref[s] <= c0, by two 64-way steps (s_pref is non-decreasing; PROC_WAVES = 64 * 64)
                const unsigned coarse = s_pref[lane * 64];
                const int k1 = __popcll(__ballot(coarse <= c0)) - 1;
                const unsigned fine = s_pref[k1 * 64 + lane];
                int sl = k1 * 64 + __popcll(__ballot(fine <= c0)) - 1;
                const unsigned c = c0 + (unsigned)lane;
                if (c < total) {
                    while (c >= s_pref[sl + 1]) ++sl;   // (slices are a few dozen candidates long: a few steps)
                    e[r] = rec[(size_t)sl * wcap + (c - s_pref[sl])];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < NRX; ++r) {
            cck[r] = 0.0f; cw[r] = 0.0f;
            cx[r][0] = cx[r][1] = cx[r][2] = 0.0f;
            cy[r][0] = cy[r][1] = cy[r][2] = 0.0f;
            if (r < rmax) {
                const bool have = wave_first + (unsigned)r * lanes + (unsigned)lane < total;
                const float4 x = pa.pos_a[e[r].x & 0xffffu];
                const float4 y = pa.pos_b[e[r].x >> 16];
                cx[r][0] = x.x; cx[r][1] = x.y; cx[r][2] = x.z;
                cy[r][0] = y.x; cy[r][1] = y.y; cy[r][2] = y.z;
                cck[r] = have ? __uint_as_float(e[r].y) : 0.0f;   // (0: the pair is never a member)
            }
        }
    }
    // ... and those behind the registers' (rounds RUN_R .. rmax - 1: only a run of all RUN_G solvers has them) into LDS
    static_assert(RUN_L % 2 == 0, "run_flow_lds / run_step_lds take two rounds per trip");
    nl = (!ACVO && rmax > RUN_R) ? ((rmax - RUN_R + 1) & ~1) : 0;   // (wave-uniform; even: a last odd round is filled with lanes without a candidate)
    {
        const uint2 *rec = act ? pa.cand_b : pa.cand;
        const unsigned wcap = pa.kept_wcap;
#pragma unroll 1
        for (int r0 = 0; r0 < nl; r0 += 2) {   // (two rounds per trip: both records requested before either gather, all four gathers before a store)
            uint2 e[2];
            bool have[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const unsigned c0 = wave_first + (unsigned)(RUN_R + r0 + u) * lanes;   // (< total but for the filling round)
                const unsigned cs = c0 < total ? c0 : total - 1u;
                const unsigned coarse = s_pref[lane * 64];
                const int k1 = __popcll(__ballot(coarse <= cs)) - 1;
                const unsigned fine = s_pref[k1 * 64 + lane];
                int sl = k1 * 64 + __popcll(__ballot(fine <= cs)) - 1;
                const unsigned c = c0 + (unsigned)lane;
                e[u] = make_uint2(0u, 0u);
                have[u] = c < total;
                if (have[u]) {
                    while (c >= s_pref[sl + 1]) ++sl;
                    e[u] = rec[(size_t)sl * wcap + (c - s_pref[sl])];
                }
            }
            float4 x[2], y[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) { x[u] = pa.pos_a[e[u].x & 0xffffu]; y[u] = pa.pos_b[e[u].x >> 16]; }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                lc[(2 * (r0 + u)) * RUN_BLOCK] = make_float4(x[u].x, x[u].y, x[u].z, have[u] ? __uint_as_float(e[u].y) : 0.0f);
                lc[(2 * (r0 + u) + 1) * RUN_BLOCK] = make_float4(y[u].x, y[u].y, y[u].z, 0.0f);
            }
        }
    }
    };   // load_xy
    load_xy();
    // ---- acvo: the lane's xx and yy candidates (the same flat numbering over their records' slices)
    float xd2[ACVO ? RUN_A : 1], xck[ACVO ? RUN_A : 1];
    float yya[ACVO ? RUN_A : 1][3], yyb[ACVO ? RUN_A : 1][3], yck[ACVO ? RUN_A : 1];
    int rmax_x = 0, rmax_y = 0;
    if constexpr (ACVO) {
#pragma unroll
        for (int set = 0; set < 2; ++set) {
            const ProcessArgs &sa = set == 0 ? xa : ya;
            const unsigned tot_s = set == 0 ? total_x : total_y;
            const uint2 *rec = (set == 0 ? act_x : act_y) ? sa.cand_b : sa.cand;
            const unsigned wcap = sa.kept_wcap;
            const unsigned *const pref = s_pref_self + (set == 0 ? 0 : PROC_WAVES + 1);   // (this record's prefix sums)
            const unsigned first_s = head_block ? tot_s : (unsigned)srow * RUN_BLOCK + (unsigned)wid * 64u;
            const int rm = first_s < tot_s ? (int)((tot_s - first_s + lanes - 1) / lanes) : 0;
            if (set == 0) rmax_x = rm; else rmax_y = rm;
            uint2 e[RUN_A];
#pragma unroll
            for (int r = 0; r < RUN_A; ++r) {
                e[r] = make_uint2(0u, 0u);
                if (r < rm) {
                    const unsigned c0 = first_s + (unsigned)r * lanes;   // < tot_s
                    const unsigned coarse = pref[lane * 64];
                    const int k1 = __popcll(__ballot(coarse <= c0)) - 1;
                    const unsigned fine = pref[k1 * 64 + lane];
                    int sl = k1 * 64 + __popcll(__ballot(fine <= c0)) - 1;
                    const unsigned c = c0 + (unsigned)lane;
                    if (c < tot_s) {
                        while (c >= pref[sl + 1]) ++sl;
                        e[r] = rec[(size_t)sl * wcap + (c - pref[sl])];
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < RUN_A; ++r) {
                const bool have = r < rm && first_s + (unsigned)r * lanes + (unsigned)lane < tot_s;
                // (a lane without a candidate reads row 0 of both clouds and holds a colour weight of 0: never a member)
                const float4 a4 = sa.pos_a[e[r].x & 0xffffu];
                const float4 b4 = sa.pos_b[e[r].x >> 16];
                const float ckr = have ? __uint_as_float(e[r].y) : 0.0f;   // (the sign: the row counts or not, eval_pair)
                if (set == 0) {
                    const float e0 = a4.x - b4.x, e1 = a4.y - b4.y, e2 = a4.z - b4.z;   // (x never moves: d2 once)
                    xd2[r] = __builtin_fmaf(e2, e2, __builtin_fmaf(e1, e1, e0 * e0));
                    xck[r] = ckr;
                } else {
                    yya[r][0] = a4.x; yya[r][1] = a4.y; yya[r][2] = a4.z;
                    yyb[r][0] = b4.x; yyb[r][1] = b4.y; yyb[r][2] = b4.z;
                    yck[r] = ckr;
                }
            }
        }
        if (tid == 0) { s_xx_ell = -1.0f; s_xx_keep[0] = 0.0; s_xx_keep[1] = 0.0; }   // (no Axx sums yet)
    }
    // the row of flags the entry head has read is cleared as the slot's step launch would (nothing is flagged in a run);
    // the counters of a build the entry head has named are zeroed as its flow launch would
    if (head_block) {
        if (tid < 8) gst->ovf[1][tid] = 0u;
        head_prepare_lists<HM_HEAD>(ps, &s_st);
    }

    bool comm_ok = true;
    if (tid == 0) s_fail = 0;   // (the barriers of the first pass lie between this and the first poll)
    const int need_d2 = ACVO ? 1 : pa.need_d2;   // (acvo: the sum of a d2 is a term of dl)
    const int iters = ps.run_iters > 0 ? ps.run_iters : 1;
    const long long run_timeout = ps.run_timeout_ticks > 0 ? ps.run_timeout_ticks : RUN_TIMEOUT_TICKS;
    // ---- SIDE BUILDS (round 6).  A run used to end for every list build: two launch-per-pass slots (one builds beside its passes, the next
    // expands and records) and a new entry, 45-75 us for two iterations.  A run of up to RUN_G_SIDE solvers now has its next xy list built
    // BESIDE it: when its head names a build, the head block puts the head where the side kernels read it (the second copy of the state's
    // head, idle during a run), zeroes the list's counters and asks the host (PostStepArgs::side_mirror) for kt_side_filter + kt_side_record
    // on the context's side stream -- on the compute units the run does not hold --; the run iterates on (the old list holds every pair
    // until its room is used up: the plan checks that every slot, and a build is named early, PostStepArgs::run_build_at); the plan is
    // kept from judging the build (xy_target stays, xy_fresh does not appear) until RunMail::side_done says the record is written; the slot
    // whose plan then switches lists is the run's last (its passes stand if the old list still held every pair for it, else they are void,
    // as in a stall slot), and the next kt_run launch -- queued right behind -- enters on the new record.  A run never leaves with a side
    // build in flight (side_finish): the launches behind it would build the same list again.
    // (cvo only: with acvo's self lists planned beside it the soak found 7 of 300 registrations wrong -- the counters of a self list named while
    // an xy build was in flight --; the option does not pay, so acvo simply does not have it, profiles/r06_ab.txt 12)
    const bool side_can = SIDE && !ACVO && head_block && ps.side_mirror != nullptr && g <= RUN_G_SIDE && ps.st2 != nullptr;
#ifdef CVO_SIDE_DEBUG   // (probe builds: what the side builds did, DevState::run_clk through cvo_hip_get_run_clocks)
#define SIDE_DBG(i) do { if (head_block && tid == 0) gst->run_clk[i] += 1; } while (0)
#else
#define SIDE_DBG(i) do { } while (0)
#endif
    SIDE_DBG(side_can ? 0 : 1);
    if (head_block && tid == 0) { s_side_t = -1; s_side_handed = 0; s_side_fail = 0; s_vbits = 0u; s_side_count = 0; s_side_total = 0u; }
    bool reload_next = false;   // (block-uniform: the verdict of the slot that has just run said RUN_V_RELOAD)
    auto side_ended = [&]() -> bool {
        return __hip_atomic_load(&ps.run_mail->side_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= s_side_expect;
    };
    // (before the head goes out, whatever way the run ends; all threads of the head block; false: the side kernels never came)
    auto side_finish = [&]() -> bool {
        if (!SIDE || !side_can) return true;
        __syncthreads();
        if (tid == 0 && s_side_t >= 0 && !s_side_handed) {
            const long long t0 = (long long)wall_clock64();
            while (!side_ended()) {
                if ((long long)wall_clock64() - t0 > (ps.run_timeout_ticks > 0 ? ps.run_timeout_ticks : RUN_TIMEOUT_TICKS)) { s_side_fail = 1; SIDE_DBG(9); break; }
                __builtin_amdgcn_s_sleep(8);
            }
            if (!s_side_fail) {   // built and recorded: the next head judges it (its overflow flags, if any, are up in row 1: that head reads them)
                const int t = s_side_t;
                s_st.xy_fresh = t; s_st.xy_target = -1;
                if (t) s_st.xy_ck[1] = pa.nblk; else s_st.xy_ck[0] = pa.nblk;
                s_side_handed = 1;
            }
        }
        __syncthreads();
        return s_side_fail == 0;
    };
    // the head block's verdict on the slot that begins travels with the slot's second exchange (its number is known in advance)
    auto post_verdict = [&](const unsigned long long seq_b) {
        if (head_block && tid == 0) {
            // (a build that is being made beside the run does not end it)
            const bool build = (s_st.xy_target >= 0 && !(side_can && s_side_t == s_st.xy_target)) ||
                               (ACVO && (s_st.sf_target[0] >= 0 || s_st.sf_target[1] >= 0));
            const unsigned v = (s_st.stall != 0 ? (unsigned)RUN_V_STALL : 0u) | (build ? (unsigned)RUN_V_BUILD : 0u) | (side_can ? s_vbits : 0u);
            __hip_atomic_store(&ps.run_mail->w[(seq_b >> 1) & 1ull][RUN_G][0], ((seq_b & 0xffffffffull) << 32) | v, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    post_verdict(seq0 + 2);
    RUN_CLK(0);
    for (int it = 0;; ++it) {
        if (ps.run_fault > 0 && srow == 0 && it + 1 == ps.run_fault) return;   // (test switch: a solver that is lost to its peers)
        // ---- this slot's constants from the head in LDS: plain broadcast reads into vector registers (through scalar
        // registers -- v_readfirstlane of every word -- the kernel spilled 200 of them and an iteration's two passes spent
        // more time moving constants than on their pairs, profiles/r05_ab.txt 3)
        float rt[12];
#pragma unroll
        for (int q = 0; q < 9; ++q) rt[q] = s_st.Rt[q];
#pragma unroll
        for (int q = 0; q < 3; ++q) rt[9 + q] = s_st.t[q];
        const KernConsts kc = s_st.kc;
        RUN_CLK(1);
        // ---- flow pass (se_kernel's exact tests + compute_flow, ref src/cvo.cpp:125-152,164-210) on registers
        double acc[NACC_FLOW];
#pragma unroll
        for (int k = 0; k < NACC_FLOW; ++k) acc[k] = 0.0;
        unsigned nk = 0;
        static_assert(RUN_R == 8 && RUN_A >= 3 && RUN_A <= 8, "the cases below");
        if constexpr (ACVO) {
            switch (rmax) {   // (wave-uniform; at most RUN_A rounds)
            case 0: break;
            case 1: nk = run_flow_rounds<1>(rt, kc, cx, cy, cck, cw, s_etab, need_d2, acc); break;
            case 2: nk = run_flow_rounds<2>(rt, kc, cx, cy, cck, cw, s_etab, need_d2, acc); break;
            case 3: nk = run_flow_rounds<3>(rt, kc, cx, cy, cck, cw, s_etab, need_d2, acc); break;
            default: nk = run_flow_rounds<RUN_A>(rt, kc, cx, cy, cck, cw, s_etab, need_d2, acc); break;
            }
        } else {
            switch (rmax) {   // (wave-uniform; rounds beyond the wave's last candidate would be all zeros)
            case 0: break;
            case 1: nk = run_flow_rounds<1>(rt, kc, cx, cy, cck, cw, s_etab, need_d2, acc); break;
            case 2: nk = run_flow_rounds<2>(rt, kc, cx, cy, cck, cw, s_etab, need_d2, acc); break;
            case 3: nk = run_flow_rounds<3>(rt, kc, cx, cy, cck, cw, s_etab, need_d2, acc); break;
            case 4: nk = run_flow_rounds<4>(rt, kc, cx, cy, cck, cw, s_etab, need_d2, acc); break;
            case 5: case 6: nk = run_flow_rounds<6>(rt, kc, cx, cy, cck, cw, s_etab, need_d2, acc); break;
            default: nk = run_flow_rounds<8>(rt, kc, cx, cy, cck, cw, s_etab, need_d2, acc); break;
            }
            if (nl > 0) nk += run_flow_lds(nl, lc, rt, kc, s_etab, need_d2, acc);
        }
        RUN_CLK(2);
        // ---- acvo: Ayy of this transform, Axx when the length scale has moved (ref src/adaptive_cvo.cpp:157-158, 213-265)
        double self[4] = {0.0, 0.0, 0.0, 0.0};   // sum xx, count xx, tail sum yy, count yy
        bool xx_fresh = false;
        if constexpr (ACVO) {
            unsigned ny = 0;
            switch (rmax_y) {
            case 0: break;
            case 1: ny = run_yy_rounds<1>(rt, kc, yya, yyb, yck, s_etab, self[2]); break;
            case 2: ny = run_yy_rounds<2>(rt, kc, yya, yyb, yck, s_etab, self[2]); break;
            case 3: ny = run_yy_rounds<3>(rt, kc, yya, yyb, yck, s_etab, self[2]); break;
            default: ny = run_yy_rounds<RUN_A>(rt, kc, yya, yyb, yck, s_etab, self[2]); break;
            }
            if (lane == 0) self[3] = (double)ny;
            xx_fresh = !(s_xx_ell == s_st.kc_ell);   // (block-uniform; every block holds the same length scale)
            if (xx_fresh) {
                unsigned nx = 0;
                switch (rmax_x) {
                case 0: break;
                case 1: nx = run_xx_rounds<1>(kc, xd2, xck, s_etab, self[0]); break;
                case 2: nx = run_xx_rounds<2>(kc, xd2, xck, s_etab, self[0]); break;
                case 3: nx = run_xx_rounds<3>(kc, xd2, xck, s_etab, self[0]); break;
                default: nx = run_xx_rounds<RUN_A>(kc, xd2, xck, s_etab, self[0]); break;
                }
                if (lane == 0) self[1] = (double)nx;
            }
        }
        double my_val = 0.0;   // (thread t < 2 NVF: sum t >> 1 of this block, what the thread sends)
        if (!head_block) {
            if (lane == 0) acc[8] = (double)nk;
            wave_sums<NACC_FLOW>(acc, lane, s_red + wid * NACC_MAX);
            if constexpr (ACVO) wave_sums<4>(self, lane, s_self + wid * 4);
            RUN_CLK(3);
            __syncthreads();
            const int k = tid >> 1;
            if (k < NACC_FLOW) {
#pragma unroll
                for (int q = 0; q < RUN_WAVES; ++q) my_val += s_red[q * NACC_MAX + k];
            } else if (ACVO && k < NVF) {
                const int ks = k - NACC_FLOW;
#pragma unroll
                for (int q = 0; q < RUN_WAVES; ++q) my_val += s_self[q * 4 + ks];
                if (ks < 2) {   // Axx: made now, or the sums this block made when the length scale last moved (both threads of a sum store the same)
                    if (xx_fresh) s_xx_keep[ks] = my_val; else my_val = s_xx_keep[ks];
                }
            }
        }
        if (ACVO && xx_fresh && tid == 0) s_xx_ell = s_st.kc_ell;   // (read again behind the barriers of the exchange)
        RUN_CLK(4);
        ++nexch;
        if (!run_exchange<NVF, HIER_FROM>(ps.run_mail, srow, g, seq0 + nexch, my_val, s_all, s_part, s_tot, &s_fail, nullptr, &s_verdict, run_timeout)) { comm_ok = false; SIDE_DBG(10); break; }
        RUN_CLK(5);
        // ---- the tail of compute_flow (ref src/cvo.cpp:201-209): twist, Taylor constants
        if (tid < 64) {
            float omega[3], v[3];
            for (int q = 0; q < 3; ++q) { omega[q] = (float)s_tot[q]; v[q] = (float)s_tot[3 + q]; }
            xi_consts_wave(&s_xi, s_wm, omega, v, lane);
            if (tid == 0) {
                if (head_block) {   // (a stall verdict voids the slot: the head then goes out with the sums it came with)
                    for (int q = 0; q < NACC_FLOW; ++q) s_bak[q] = s_st.red[RED_FLOW + q];
                    for (int q = 0; q < 3; ++q) { s_bakf[q] = s_st.omega[q]; s_bakf[3 + q] = s_st.v[q]; }
                }
                for (int q = 0; q < NACC_FLOW; ++q) s_st.red[RED_FLOW + q] = s_tot[q];
                for (int q = 0; q < 3; ++q) { s_st.omega[q] = s_xi.omega[q]; s_st.v[q] = s_xi.v[q]; }
            }
        } else if (ACVO && tid == 64) {
            // dl (ref src/adaptive_cvo.cpp:222-231,271; the expression of step_twist_body), in every block: its head_post moves the
            // length scale by it.  Kept beside the head until the slot stands (a stall verdict voids it)
            const long long nnz = (long long)s_tot[8], nnz_xx = (long long)s_tot[NACC_FLOW + 1], nnz_yy = (long long)s_tot[NACC_FLOW + 3];
            const double num = (s_tot[NACC_FLOW + 2] - 2.0 * s_tot[7]) + s_tot[NACC_FLOW];
            s_dl = num / (double)(nnz_xx + nnz_yy - 2 * nnz);
            for (int q = 0; q < 4; ++q) s_self[q] = s_tot[NACC_FLOW + q];   // (s_tot: the step exchange writes it next)
        }
        __syncthreads();
        const cvo_math::XiConsts xc = s_xi;
        RUN_CLK(6);
        // ---- compute_step_size sums (ref src/cvo.cpp:213-289) over the members, whose weights are still in registers
        double sacc[NACC_STEP];
#pragma unroll
        for (int k = 0; k < NACC_STEP; ++k) sacc[k] = 0.0;
        if constexpr (ACVO) {
            switch (rmax) {
            case 0: break;
            case 1: run_step_rounds<1>(rt, kc, xc, cx, cy, cw, sacc); break;
            case 2: run_step_rounds<2>(rt, kc, xc, cx, cy, cw, sacc); break;
            case 3: run_step_rounds<3>(rt, kc, xc, cx, cy, cw, sacc); break;
            default: run_step_rounds<RUN_A>(rt, kc, xc, cx, cy, cw, sacc); break;
            }
        } else {
            switch (rmax) {
            case 0: break;
            case 1: run_step_rounds<1>(rt, kc, xc, cx, cy, cw, sacc); break;
            case 2: run_step_rounds<2>(rt, kc, xc, cx, cy, cw, sacc); break;
            case 3: run_step_rounds<3>(rt, kc, xc, cx, cy, cw, sacc); break;
            case 4: run_step_rounds<4>(rt, kc, xc, cx, cy, cw, sacc); break;
            case 5: case 6: run_step_rounds<6>(rt, kc, xc, cx, cy, cw, sacc); break;
            default: run_step_rounds<8>(rt, kc, xc, cx, cy, cw, sacc); break;
            }
            if (nl > 0) run_step_lds(nl, lc, rt, kc, xc, sacc);
        }
        RUN_CLK(7);
        double my_step = 0.0;
        if (!head_block) {
            wave_sums<NACC_STEP>(sacc, lane, s_red + wid * NACC_MAX);
            __syncthreads();
            if (tid < 2 * NACC_STEP) {
#pragma unroll
                for (int q = 0; q < RUN_WAVES; ++q) my_step += s_red[q * NACC_MAX + (tid >> 1)];
            }
        }
        RUN_CLK(8);
        ++nexch;
        unsigned verdict = 0u;
        // (the last wave forms the twist-only stage of the head's chain while the step sums travel: ~120 instructions the first wave
        // no longer runs in a row behind the exchange, cvo_math::exp_se3_pre)
        auto pre_work = [&]() {
            if (wid == RUN_WAVES - 1) {
                const cvo_math::ExpPre P = cvo_math::exp_se3_pre(s_xi.omega, s_xi.v);
                if (lane == 0) s_pre = P;
            }
        };
        if (!run_exchange<NACC_STEP, HIER_FROM>(ps.run_mail, srow, g, seq0 + nexch, my_step, s_all, s_part, s_tot, &s_fail, &verdict, &s_verdict, run_timeout, pre_work)) { comm_ok = false; SIDE_DBG(11); break; }
        RUN_CLK(9);
        if (verdict & RUN_V_STALL) {
            // no buffer holds every pair for this slot's transform (a jump): what the passes have summed is void.  The head goes
            // out as the head-mode flow launch of a stall slot would publish it (it runs no passes: the next launch's filter
            // blocks build what the plan has named); the head block did its list preparation when it planned
            if (head_block) {
                if (tid == 0) {
                    for (int q = 0; q < NACC_FLOW; ++q) s_st.red[RED_FLOW + q] = s_bak[q];
                    for (int q = 0; q < 3; ++q) { s_st.omega[q] = s_bakf[q]; s_st.v[q] = s_bakf[3 + q]; }
                    s_st.stall = 1;   // (a slot made void because the plan changed lists under the solvers: the next head must take it for the stall slot it is)
                }
                if (!side_finish()) { comm_ok = false; break; }
                __syncthreads();
                head_publish(ps, &s_st, gst, true);
            }
            break;
        }
        if constexpr (SIDE) reload_next = (verdict & RUN_V_RELOAD) != 0u;
        // ---- the slot stands.  The head block: what the slot's step launch leaves in the head, the trace record
        if (ACVO && !head_block && tid == 0) s_st.dl = s_dl;   // (what this block's head_post moves the length scale by)
        if (head_block && tid == 0) {
            for (int q = 0; q < 4; ++q) s_st.red[RED_XX + q] = ACVO ? s_self[q] : 0.0;
            s_st.xi = s_xi;
            s_st.dl = ACVO ? s_dl : 0.0;
            if (ta.trace && s_st.k < ta.trace_cap) {
                cvo_hip_trace &tr = ta.trace[s_st.k];
                tr.k = s_st.k;
                tr.exit_code = 0;
                tr.ell = s_st.ell;
                for (int q = 0; q < 3; ++q) {
                    tr.omega[q] = s_xi.omega[q]; tr.v[q] = s_xi.v[q];
                    tr.omega_d[q] = s_st.red[RED_FLOW + q]; tr.v_d[q] = s_st.red[RED_FLOW + 3 + q];
                }
                tr.sum_a = s_st.red[RED_FLOW + 6];
                tr.dl = s_st.dl;
                tr.nnz = (long long)s_st.red[RED_FLOW + 8];
                tr.nnz_xx = ACVO ? (long long)s_st.red[RED_XX + 1] : 0; tr.nnz_yy = ACVO ? (long long)s_st.red[RED_YY + 1] : 0;
            }
        }
        // ---- the slot is complete but for its post-step part.  Leave here -- as a step launch would leave it -- when the
        // head of this slot has named a build (the next classic flow launch's filter blocks make it) or the run is over
        if (it + 1 >= iters || (verdict & RUN_V_BUILD)) {
            if (head_block) {
                if (!side_finish()) { comm_ok = false; break; }
                __syncthreads();
                // (the step sums as row 0 of the rows the next head reduces; the other rows are zero: x + 0 = x in any order)
                for (int q = tid; q < NACC_STEP * ps.nblk; q += RUN_BLOCK) {
                    const int k = q / ps.nblk, b = q - k * ps.nblk;
                    const_cast<double *>(ps.part_step)[q] = b == 0 ? s_tot[k] : 0.0;
                }
                head_publish(ps, &s_st, gst, false);
            }
            break;
        }
        // ---- the post-step part of the head: cubic, break tests, Exp_SEK3, update, length scale (ref src/cvo.cpp:291-307,
        // 380-410), in every block; then, for the passes of the next slot, its transform and -- when the length scale moved --
        // its kernel constants (the two pieces of the plan that the passes read: the same functions, the same values)
        if (tid < NACC_STEP) s_st.red[RED_STEP + tid] = s_tot[tid];
        __syncthreads();
        RUN_CLK(10);
        if (tid < 64) {
            long long clk[4] = {0, 0, 0, 0};
#ifdef CVO_RUN_CLOCKS
            head_post<HM_HEAD>(&s_st, ps, true, head_block, true, clk, &s_pre);
            clk_acc[15] += clk[1] - clk[0];   // (of head_post: the cubic and its root)
#else
            head_post<HM_HEAD>(&s_st, ps, true, head_block, false, clk, &s_pre);
#endif
            RUN_CLK(11);
            if (s_st.done == RUNNING) {
                if (head_block) {
                    unsigned flag[LIST_N];
#pragma unroll
                    for (int l = 0; l < LIST_N; ++l) flag[l] = 0u;   // (nothing is built and no slice can overflow in a run)
                    if (SIDE && side_can && ps.run_build_at > 0.0f) {   // (builds beside the run are named early: they take a few iterations)
                        PostStepArgs psr = ps;
                        psr.prm.build_at = ps.run_build_at;
                        head_plan<HM_HEAD>(&s_st, psr, true, false, flag);
                    } else {
                        head_plan<HM_HEAD>(&s_st, ps, true, false, flag);
                    }
                } else {
                    float Rt[9], t[3];
                    cvo_math::inverse_tf(s_st.R, s_st.T, Rt, t);
                    if (!(s_st.kc_ell == s_st.ell)) {
                        const KernConsts k = make_kconsts(ps.prm, s_st.ell);
                        if (tid == 0) { s_st.kc = k; s_st.kc_ell = s_st.ell; }
                    }
                    if (tid == 0) {
#pragma unroll
                        for (int q = 0; q < 9; ++q) s_st.Rt[q] = Rt[q];
#pragma unroll
                        for (int q = 0; q < 3; ++q) s_st.t[q] = t[q];
                        s_st.pending = 1;
                    }
                }
            }
        }
        __syncthreads();
        RUN_CLK(12);
        if (s_st.done != RUNNING) {   // the loop has stopped: the final head goes out
            if (head_block) {
                if (!side_finish()) { comm_ok = false; break; }
                if (!(side_can && s_side_t >= 0)) head_prepare_lists<HM_HEAD>(ps, &s_st);
                head_publish(ps, &s_st, gst, true);
            }
            break;
        }
        if constexpr (SIDE) if (side_can) {   // (the head block; block-uniform)
            // the plan that has just run and the side build
            if (tid == 0) {
                unsigned vb = 0u;
                if (s_side_t >= 0) {
                    const int t = s_side_t;
                    if (!s_side_handed) {
                        // (the plan took the target for built -- in the launch-per-pass path the launch that carries the plan builds it --: it
                        // is, once the side kernels are through)
                        if (side_ended()) {
                            SIDE_DBG(3);
                            s_side_handed = 1;
                            s_side_count = 1;   // (all threads count the record's candidates below)
                            if (t) s_st.xy_ck[1] = pa.nblk; else s_st.xy_ck[0] = pa.nblk;   // (its record is written)
                            // a list or a record slice that overflowed: the flags are up in row 1, the head of the next launch reads them
                            // and parks the loop as always -- the run ends with the coming slot, before its own plan would judge the build
                            if ((__hip_atomic_load(&gst->ovf[1][LIST_KEPT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) |
                                 __hip_atomic_load(&gst->ovf[1][t ? LIST_XYB : LIST_XY], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0u)
                                vb |= (unsigned)RUN_V_BUILD;
                        } else {
                            s_st.xy_target = t; s_st.xy_fresh = -1;
                        }
                    } else {
                        s_side_t = -1; s_side_handed = 0;   // (the plan that has just run judged it)
                    }
                }
                if ((s_st.xy_active ? 1 : 0) != act_cur) {
                    // the plan has changed lists; the solvers hold the old one's candidates: this slot is the run's last.  Its passes stand if
                    // the old list still holds every pair for the slot's transform (plan_xy_async's own test), else they are void
                    const float r_now = sqrtf(s_st.kc.tau), slack = 1.0e-4f * (1.0f + s_st.xmax + s_st.y0max);
                    const float need = (r_now + (act_cur ? xy_travel<1>(&s_st, &s_st) : xy_travel<0>(&s_st, &s_st))) * 1.0001f + slack;
                    const bool held = (act_cur ? s_st.xy_ok[1] : s_st.xy_ok[0]) != 0 && need <= (act_cur ? s_st.xy_r[1] : s_st.xy_r[0]);
                    // (... and the run goes on where the new record fits the solvers it has: every block loads its candidates anew)
                    const bool fits = s_side_total > 0u && s_side_total <= (unsigned)g * (unsigned)RUN_BLOCK * (unsigned)(ACVO ? RUN_A : RUN_R + RUN_L);
                    vb |= held ? (fits ? (unsigned)RUN_V_RELOAD : (unsigned)RUN_V_BUILD) : (unsigned)RUN_V_STALL;
                    SIDE_DBG(held ? (fits ? 6 : 4) : 5);
                }
                s_vbits = vb;
            }
            __syncthreads();
            if (s_side_count != 0) {   // the side build has just ended: what its record holds (all threads; s_pref is free between two exchanges)
                const int t = s_side_t;
                unsigned sel[PER];
#pragma unroll
                for (int q = 0; q < PER; ++q) {
                    const int sl = tid * PER + q;
                    sel[q] = sl < nsl ? (t ? pa.cand_cnt_b[sl] : pa.cand_cnt[sl]) : 0u;
                }
                const unsigned tot = prefix(sel, pa.kept_wcap, s_pref);
                if (tid == 0) {
                    s_side_total = tot;
                    if (t) s_st.rec_count[1] = (int32_t)tot; else s_st.rec_count[0] = (int32_t)tot;
                    s_side_count = 0;
                }
                __syncthreads();
            }
            // a build the plan has just named: beside the run
            if (s_side_t < 0 && s_vbits == 0u && s_st.xy_target >= 0 && s_st.stall == 0 &&
                !(ACVO && (s_st.sf_target[0] >= 0 || s_st.sf_target[1] >= 0))) {
                head_prepare_lists<HM_HEAD>(ps, &s_st);                      // (its counters)
                state_head_from_lds(&s_st, static_cast<DevHead *>(ps.st2));   // (what the side kernels read: target, its transform and bound, the kernel constants)
                __threadfence_system();
                __syncthreads();
                if (tid == 0) {
                    s_side_expect = __hip_atomic_load(&ps.run_mail->side_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + (unsigned long long)cs->op[qf].np;
                    s_side_t = s_st.xy_target; s_side_handed = 0;
                    SIDE_DBG(2);
                    const unsigned long long req = ps.run_mail->side_req + 1ull;
                    ps.run_mail->side_req = req;
                    // (a system-scope store: a plain one to host memory may sit in this device's cache until the kernel ends -- and the host
                    // must see this one while the run goes on)
                    __hip_atomic_store(ps.side_mirror, (int32_t)(req & 0x7fffffffull), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                __syncthreads();
            }
        }
        if (head_block) {
            post_verdict(seq0 + nexch + 2);
            if (tid == 0) {   // the host's mirrors, once per slot
                if (ps.hint_mirror) *ps.hint_mirror = s_st.run_hint;
                if (ps.progress_mirror) *ps.progress_mirror = s_st.n_slots;
            }
            if (!(side_can && s_side_t >= 0)) head_prepare_lists<HM_HEAD>(ps, &s_st);   // (a build this head has named: its counters; not those of a list that is being filled)
        }
        if constexpr (SIDE) {
            if (reload_next) {
                // the plan has changed lists (a side build's): this block's candidates from the other buffer's record, as at entry
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // (another kernel wrote it while this one ran)
                act_cur ^= 1;
                unsigned sel[PER];
#pragma unroll
                for (int q = 0; q < PER; ++q) {
                    const int sl = tid * PER + q;
                    sel[q] = sl < nsl ? (act_cur ? pa.cand_cnt_b[sl] : pa.cand_cnt[sl]) : 0u;
                }
                tot_cur = prefix(sel, pa.kept_wcap, s_pref);
                load_xy();
                reload_next = false;
                SIDE_DBG(7);
            }
        }
        RUN_CLK(13);
    }
    if (!head_block) {
#ifdef CVO_RUN_CLOCKS
        if (srow == 0 && tid == 0) {
            RUN_CLK(14);
            for (int q = 0; q < 16; ++q) gst->run_clk[q] += clk_acc[q];
        }
#endif
        return;
    }
    if (!comm_ok) {
        // An exchange timed out: nothing this run has summed can be trusted -- and nothing of it is in the state.  WHAT A RUN WRITES
        // BEFORE ITS EXIT: row 1 of the overflow flags cleared (they were clear: the entry head saw them), the counters of a build its
        // heads named (the build's own launch zeroes them again), trace records (rewritten with the same values), the host's hint and
        // progress mirrors; the head, part_step and the run counters only on the way out, below and in head_publish.  The verdict is
        // the run's own -- no mailbox is involved, the context's exchanges with other ranks (if it has any) are intact --, and the host
        // answers it by registering the pair again without runs (cvo_job.cpp job_pump).
        if (tid == 0) {
            gst->done = DONE_RUN_TIMEOUT;
            if (ps.done_mirror) *ps.done_mirror = DONE_RUN_TIMEOUT;
        }
    }
    if (tid == 0) {
        gst->run_seq = seq0 + nexch;
        gst->run_entered += 1;
        gst->run_iterations += (int)(nexch / 2u);
    }
    __syncthreads();
    run_over();
}

__global__ void __launch_bounds__(RUN_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2)))
kt_run(const Slot *__restrict__ tab, const int qs)
{
    extern __shared__ __attribute__((aligned(16))) float4 s_lc[];   // [RUN_L][2][RUN_BLOCK]: the candidates behind the registers'
    run_body<false>(tab, qs, s_lc);
}
__global__ void __launch_bounds__(RUN_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2)))
kt_run_acvo(const Slot *__restrict__ tab, const int qs)
{
    run_body<true>(tab, qs, nullptr);   // (no candidate lives in LDS: a launch without dynamic LDS)
}
__global__ void __launch_bounds__(RUN_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2)))
kt_run_side(const Slot *__restrict__ tab, const int qs)
{
    extern __shared__ __attribute__((aligned(16))) float4 s_lc[];
    run_body<false, true>(tab, qs, s_lc);
}
__global__ void __launch_bounds__(RUN_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2)))
kt_run_acvo_side(const Slot *__restrict__ tab, const int qs)
{
    run_body<true, true>(tab, qs, nullptr);
}

// The side build of a run's next xy list (kt_run "side builds"): the filter over all pairs at the transform the run's plan recorded, then the
// expansion of the tile list with its colour weights into the buffer's candidate record -- the two things the launch-per-pass path does in
// the flow launches of two slots.  Both read the head the run's head block put into the second copy of the state's head.
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(4, 8)))
kt_side_filter(const Slot *__restrict__ tab, const int q)
{
    CSlot cs = (CSlot)(tab);
    if (cs->active == 0) return;
    const FilterArgs &f = CVO_ARG(FilterArgs, op[q].f);
    filter_body<false>(f, blockIdx.x, gridDim.x, f.st2, 1);   // (overflows are flagged in row 1: the row the head of the launch after the run reads)
}
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(4, 8)))
kt_side_record(const Slot *__restrict__ tab, const int q)
{
    __shared__ __attribute__((aligned(16))) char scratch[PROC_SMEM];
    CSlot cs = (CSlot)(tab);
    if (cs->active == 0) return;
    const ProcessArgs &pa = CVO_ARG(ProcessArgs, op[q].p);
    RunMail *const mail = CVO_ARG(PostStepArgs, op[q].ps).run_mail;
    const DevState *h = pa.st2;
    const int t = __builtin_amdgcn_readfirstlane(h->xy_target);
    if (t >= 0 && mail != nullptr) {
        // the expansion pass of the flow launch (expand_lists with its record), on the buffer that was built: what it sums and keeps besides
        // the record goes to buffers nobody reads before they are written again (the kept list, the flow partials)
        ProcHead hd;
        hd.Rt = h->Rt; hd.tt = h->t; hd.xi = &h->xi;
        hd.kc = h->kc;
        hd.done_word = 0; hd.n_fixed = 0;
        hd.second = t ? 1 : 0;
        hd.list_bad = 0u;
        hd.ck_nblk = 0;     // (no record yet: expand and record)
        hd.par = 1;
        hd.cand = t ? pa.cand_b : pa.cand;
        hd.cand_cnt = t ? pa.cand_cnt_b : pa.cand_cnt;
        hd.need_d2 = 0;
        process_body<PROC_FLOW, 0, true, false>(pa, blockIdx.x, scratch, hd);
    }
    // (the record is in memory before the count that says so)
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0 && mail != nullptr)
        __hip_atomic_fetch_add(&mail->side_done, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

unsigned filter_grid_cap(long long nitems, long long cap) { return filter_grid_x(nitems, cap); }
long long filter_blocks_cap() { return filter_blocks_max(); }

void launch_table(const Slot *tab, const TLaunch &l, hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop, int parity)
{
    const dim3 g(l.gx, 1, l.gz);
    const int qp = l.q | (parity ? QP_PARITY : 0) | QP_HEAD;   // head-mode launches
    if (ev_start && ev_stop && (l.kernel == TK_FLOW || l.kernel == TK_FLOW_D2)) {   // engine profiling: the dispatch's own begin / end
        // (the kernel the plan names: TK_FLOW is built without the sum of a d2, acvo's plans need kt_flow_d2)
        if (l.kernel == TK_FLOW_D2) hipExtLaunchKernelGGL(kt_flow_d2, g, dim3(BLOCK), 0, s, ev_start, ev_stop, 0, tab, l.q);
        else hipExtLaunchKernelGGL(kt_process<PROC_FLOW>, g, dim3(BLOCK), 0, s, ev_start, ev_stop, 0, tab, l.q);
        return;
    }
    switch (l.kernel) {
    case TK_FILTER: hipLaunchKernelGGL(kt_filter, g, dim3(BLOCK), l.smem, s, tab, l.q | (l.list << 8)); break;
    case TK_FILTER_GROUP: hipLaunchKernelGGL(kt_filter_group, dim3(l.gx, 3, l.gz), dim3(BLOCK), l.smem, s, tab, l.q); break;
    case TK_FLOW: hipLaunchKernelGGL(kt_process<PROC_FLOW>, g, dim3(BLOCK), 0, s, tab, l.q); break;
    case TK_FLOW_D2: hipLaunchKernelGGL(kt_flow_d2, g, dim3(BLOCK), 0, s, tab, l.q); break;
    case TK_FLOW_MATLAB: hipLaunchKernelGGL((kt_process<PROC_FLOW, 1>), g, dim3(BLOCK), 0, s, tab, l.q); break;
    case TK_STEP: hipLaunchKernelGGL(kt_process<PROC_STEP>, g, dim3(BLOCK), 0, s, tab, l.q); break;
    case TK_SELF: hipLaunchKernelGGL(kt_process<PROC_SELF>, g, dim3(BLOCK), 0, s, tab, l.q); break;
    case TK_SELF2: hipLaunchKernelGGL(kt_self2, dim3(l.gx, 2, l.gz), dim3(BLOCK), 0, s, tab, l.q); break;
    case TK_STEP_TWIST: hipLaunchKernelGGL(kt_step_twist, g, dim3(STEP_BLOCK), 0, s, tab, l.q); break;
    case TK_FLOW_BUILD: hipLaunchKernelGGL(kt_flow_build_w4, g, dim3(BLOCK), l.smem, s, tab, l.q); break;
    case TK_FLOW_BUILD3: hipLaunchKernelGGL(kt_flow_build3_w4, g, dim3(BLOCK), l.smem, s, tab, l.q); break;
    case TK_FLOW_BUILD6: hipLaunchKernelGGL(kt_flow_build6_w4, g, dim3(BLOCK), l.smem, s, tab, l.q); break;
    case TK_POST_FLOW: hipLaunchKernelGGL(kt_post_flow, g, dim3(BLOCK), 0, s, tab, l.q); break;
    case TK_POST_STEP: hipLaunchKernelGGL(kt_post_step, g, dim3(BLOCK), 0, s, tab, l.q); break;
    case TK_HFLOW_BUILD: hipLaunchKernelGGL(kt_hflow_build_w4, g, dim3(BLOCK), l.smem, s, tab, qp); break;
    case TK_HFLOW_BUILD6: hipLaunchKernelGGL(kt_hflow_build6_w4, g, dim3(BLOCK), l.smem, s, tab, qp); break;
    case TK_HSTEP_TWIST: hipLaunchKernelGGL(kt_step_twist, g, dim3(STEP_BLOCK), 0, s, tab, qp); break;
    // (l.list != 0: the plan has side builds -- the kernels that carry them)
    case TK_RUN:
        if (l.list) hipLaunchKernelGGL(kt_run_side, dim3(l.gx), dim3(RUN_BLOCK), RUN_LDS_BYTES, s, tab, l.q);
        else hipLaunchKernelGGL(kt_run, dim3(l.gx), dim3(RUN_BLOCK), RUN_LDS_BYTES, s, tab, l.q);   // (run_allow_lds first: plan_lone's caller)
        break;
    case TK_RUN_ACVO:
        if (l.list) hipLaunchKernelGGL(kt_run_acvo_side, dim3(l.gx), dim3(RUN_BLOCK), 0, s, tab, l.q);
        else hipLaunchKernelGGL(kt_run_acvo, dim3(l.gx), dim3(RUN_BLOCK), 0, s, tab, l.q);
        break;
    case TK_SIDE_FILTER: hipLaunchKernelGGL(kt_side_filter, g, dim3(BLOCK), l.smem, s, tab, l.q); break;
    case TK_SIDE_RECORD: hipLaunchKernelGGL(kt_side_record, g, dim3(BLOCK), 0, s, tab, l.q); break;
    default: break;
    }
}

void launch_post_flow(const PostFlowArgs &a, hipStream_t s) { launch_post_flow_group(&a, 1, s); }
void launch_post_step(const PostStepArgs &a, hipStream_t s) { launch_post_step_group(&a, 1, s); }

}   // namespace cvo_dev
