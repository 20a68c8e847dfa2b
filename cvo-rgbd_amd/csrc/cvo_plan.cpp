// cvo_plan.cpp -- the launches of one iteration of align() (ref src/cvo.cpp:366-377: transform_pcd, se_kernel,
// compute_flow, compute_step_size) recorded, merged into launch plans, written into argument tables and captured
// into batches; and the same launches issued eagerly for the low-level entry points.
#include "cvo_internal.h"

using namespace cvo_dev;
using namespace cvo_impl;

namespace cvo_impl {

// Filter grid: 256 rows per block; the column chunk is sized so that about
// four 256-thread blocks per CU are resident while every block still amortises
// its staging over many MFMA column tiles.
FilterPlan plan_filter(int nrows, int nb)
{
    FilterPlan p{};
    const int tiles = std::max(1, (nrows + ROWS_PER_TILE - 1) / ROWS_PER_TILE);
    // many small blocks: most are culled at once (bounding spheres), the others
    // should be short so that the few dense ones do not become a tail
    const int want_blocks = 4096;
    const int chunks_want = std::max(1, (want_blocks + tiles / 2) / tiles);
    int jt = (nb + chunks_want - 1) / chunks_want;
    jt = std::max(jt, 64);
    jt = std::min(jt, 2048);
    jt = (jt + SEG - 1) & ~(SEG - 1);   // whole bounding-sphere segments (4 MFMA column tiles)
    p.jt = jt;
    const int chunks = std::max(1, (nb + jt - 1) / jt);
    p.grid = dim3(chunks, tiles);
    return p;
}

int ensure_buf(cvo_hip_ctx *ctx, DevBuf &b, size_t bytes)
{
    if (bytes <= b.bytes) return CVO_HIP_OK;
    if (b.p) HIP_TRY(ctx, hipFree(b.p));
    b.p = nullptr;
    b.bytes = 0;
    if (hipMalloc(&b.p, bytes) != hipSuccess) {
        b.p = nullptr;
        return fail(ctx, CVO_HIP_ERR_NOMEM, "hipMalloc failed (candidate list / partials)");
    }
    b.bytes = bytes;
    return CVO_HIP_OK;
}

// List capacities (entries).  Tile lists: there are at most ceil(rows/16) *
// ceil(cols/16) tiles; room for all of them (x2, the sub-lists fill unevenly)
// when that is small, else a quarter.  Kept list: 4 % of all pairs (the widest
// length-scale keeps ~1.2 % on surface-like clouds), at least 1 Mi.  align()
// grows a list that overflows and redoes the iteration.
int ensure_list(cvo_hip_ctx *ctx, int list, int nrows, int nb, double at_least)
{
    List &L = ctx->lists[list];
    double want;
    uint32_t min_sub;
    if (list == LIST_KEPT) {
        const double all = (double)std::max(nrows, 0) * (double)std::max(nb, 0);
        want = std::max(all * 0.04, 1048576.0);
        want = std::min(want, std::max(all * 1.25, 1.0));
        min_sub = 64 * (PROC_WAVES / NSUB);   // every PROC_FLOW wave's slice holds >= 64 entries
    } else {
        const double all = std::ceil(std::max(nrows, 0) / 16.0 + 1.0) * std::ceil(std::max(nb, 0) / 16.0 + 1.0);
        // (a 16 x 16 tile yields up to four entries, one per MFMA result register: 4 x all
        // can never overflow; beyond 64 MB start from a quarter of the tiles and grow on demand)
        want = (all * 4.0 * sizeof(TileEntry) <= 64.0e6) ? all * 4.0 : std::max(all * 0.25, 64.0e6 / sizeof(TileEntry));
        min_sub = TILE_STAGE;
    }
    if (at_least <= 0.0) {   // test hook: start from a tiny list to exercise the grow-and-redo path
        if (ctx->opt.list_init > 0.0) want = ctx->opt.list_init;   // ("list_init")
    }
    want = std::max(want, at_least);
    want = std::min(want, 4.0e9);
    const uint32_t cap = std::max<uint32_t>((uint32_t)((want + NSUB - 1) / NSUB), min_sub) * NSUB;
    if (cap <= L.cap) return CVO_HIP_OK;
    int rc;
    if (list == LIST_KEPT) {
        rc = ensure_buf(ctx, L.a, (size_t)cap * sizeof(uint2));
        if (!rc) rc = ensure_buf(ctx, L.b, (size_t)cap * sizeof(float));
    } else {
        rc = ensure_buf(ctx, L.a, (size_t)cap * sizeof(TileEntry));
    }
    if (rc) return rc;
    L.cap = cap;
    return CVO_HIP_OK;
}

void shard_ranges(const cvo_hip_ctx *ctx, int &rlo, int &rhi, int &slo, int &shi)
{
    rlo = ctx->sharded ? ctx->row_lo : 0;
    rhi = ctx->sharded ? std::min(ctx->row_hi, ctx->fixed.n) : ctx->fixed.np;   // (padding rows are inert)
    slo = ctx->sharded ? ctx->srow_lo : 0;
    shi = ctx->sharded ? std::min(ctx->srow_hi, ctx->moving.n) : ctx->moving.np;
    rlo = std::min(rlo, rhi);
    slo = std::min(slo, shi);
}

// Geometry of the MFMA pre-filter: coordinates relative to the centre of the
// fixed cloud's bounding box; radii from the farthest bounding-box corners.
int fill_filter_geometry(cvo_hip_ctx *ctx, DevState *h)
{
    // (every compute entry point passes here before it queues anything: hand-overs still on their way end now)
    int rc_ready = cloud_ready(ctx, ctx->fixed);
    if (!rc_ready) rc_ready = cloud_ready(ctx, ctx->moving);
    if (rc_ready) return rc_ready;
    const Cloud &cf = ctx->fixed.n > 0 ? ctx->fixed : ctx->moving;
    h->n_fixed = ctx->fixed.n;
    for (int a = 0; a < 3; ++a) h->center[a] = 0.5f * (cf.lo[a] + cf.hi[a]);
    auto radius = [&](const Cloud &c) {
        if (c.n <= 0) return 0.0f;
        double r2 = 0.0;
        for (int a = 0; a < 3; ++a) {
            const double d = std::max(std::fabs((double)c.lo[a] - h->center[a]),
                                      std::fabs((double)c.hi[a] - h->center[a]));
            r2 += d * d;
        }
        return (float)(std::sqrt(r2) * 1.0001 + 1e-6);
    };
    h->xmax = radius(ctx->fixed);
    h->y0max = radius(ctx->moving);
    return CVO_HIP_OK;
}

// The entry points that exchange partial sums through the mailboxes refuse to start once an exchange has
// timed out: the ranks' sequence numbers no longer agree (job_finish), and another exchange would spin for
// its whole time-out or add up mismatched slots.
int mailboxes_usable(cvo_hip_ctx *ctx)
{
    if (ctx->comm_table && ctx->mail_broken)
        return fail(ctx, CVO_HIP_ERR_COMM, "the mailboxes of this context are unusable after a timed-out exchange: "
                                           "call cvo_hip_mailbox_create and cvo_hip_mailbox_connect again on every rank");
    return CVO_HIP_OK;
}


// Wherever the xy filter of an iteration is a recorded launch of its own -- members of a crowded
// engine, sharded and large registrations (no build riding in the flow launch) -- it also writes
// the transformed moving cloud, and the list passes of the iteration read that.
bool pre_transform(const cvo_hip_ctx *ctx)
{
    return ctx->plan_recording && ctx->in_loop && !ctx->use_async;   // (a table plan: kt_filter / kt_filter_group)
}

// The dense all-pairs filter of one list (with optional HIP-event bracket: this
// is the kernel the roofline is quoted on).
int enqueue_filter(cvo_hip_ctx *ctx, int list, const Cloud &ca, int row_lo, int row_hi, int tf_a,
                   const Cloud &cb, int tf_b, int check_done)
{
    const float4 *pos_a = ca.pos, *pos_b = cb.pos;
    const int nb = cb.np;
    const int nrows = row_hi - row_lo;
    if (nrows <= 0 || nb <= 0) return CVO_HIP_OK;
    int rc = ensure_list(ctx, list, nrows, nb, 0);
    if (rc) return rc;
    const FilterPlan pl = plan_filter(nrows, nb);
    FilterArgs a{};
    a.pos_a = pos_a; a.pos_b = pos_b;
    a.seg_a = ca.seg; a.seg_b = cb.seg;
    a.st = ctx->st;
    a.st2 = static_cast<DevState *>(ctx->st2);   // (only its head exists: head mode reads / writes nothing else of it)
    a.tiles = (TileEntry *)ctx->lists[list].a.p;
    a.subcap = ctx->lists[list].cap / NSUB;
    a.list = list;
    a.row_lo = row_lo; a.row_hi = row_hi;
    a.nb = nb; a.jt = pl.jt;
    a.tf_a = tf_a; a.tf_b = tf_b;
    a.check_done = check_done;
    a.gx = (int)pl.grid.x; a.gy = (int)pl.grid.y;
    if (pre_transform(ctx) && list == LIST_XY && tf_b && !tf_a && cb.pos == ctx->moving.pos) {
        rc = ensure_buf(ctx, ctx->pos_bt, (size_t)cb.np * sizeof(float4));
        if (rc) return rc;
        a.pos_bt = (float4 *)ctx->pos_bt.p;
    }
    const bool side = list == LIST_XY && ctx->in_loop && ctx->use_async;
    if (side) {   // build beside the flow pass, into the buffer the plan step named
        rc = ensure_list(ctx, LIST_XYB, 0, 0, (double)ctx->lists[LIST_XY].cap);
        if (rc) return rc;
        a.async_xy = 1;
        a.tiles_b = (TileEntry *)ctx->lists[LIST_XYB].a.p;
    }
    if (side) {   // no launch of its own: rides with the flow pass (enqueue_process)
        ctx->xy_build = a;
        ctx->have_xy_build = true;
        return CVO_HIP_OK;
    }
    const bool ahead = (list == LIST_XX || list == LIST_YY) && ctx->in_loop && ctx->use_async_self && ctx->rec;
    if (ahead) {   // built ahead into the idle buffer, by filter blocks of the flow launch
        const int other = list == LIST_XX ? LIST_XXB : LIST_YYB;
        rc = ensure_list(ctx, other, 0, 0, (double)ctx->lists[list].cap);
        if (rc) return rc;
        a.async_xy = list == LIST_XX ? 2 : 3;
        a.tiles_b = (TileEntry *)ctx->lists[other].a.p;
        RecOp op; op.kind = RecOp::FILTER; op.mode = kFilterAhead; op.f = a;
        ctx->rec->push_back(op);
        return CVO_HIP_OK;
    }
    if (ctx->rec) {
        RecOp op; op.kind = RecOp::FILTER; op.f = a;
        ctx->rec->push_back(op);
        return CVO_HIP_OK;
    }
    EventPair ev{};
    if (ctx->profiling) {
        HIP_TRY(ctx, hipEventCreate(&ev.a));
        HIP_TRY(ctx, hipEventCreate(&ev.b));
        ev.kind = list;
        ev.iter_tag = ctx->iter_tag;
        ev.pairs = (double)nrows * (double)nb;
    }
    // profiling: the two events are attached to the dispatch itself (kernel begin /
    // end timestamps, what rocprofv3's kernel trace reports), not recorded around it
    launch_filter(a, pl.grid, ctx->stream, ev.a, ev.b);
    if (ctx->profiling) ctx->events.push_back(ev);
    HIP_TRY(ctx, hipGetLastError());
    return CVO_HIP_OK;
}

int enqueue_process(cvo_hip_ctx *ctx, int mode, int list, DevBuf &part, const float4 *pos_a,
                    const float *feat_a, int tf_a, const float4 *pos_b, const float *feat_b,
                    int tf_b, int first_counted, int check_done)
{
    int rc = ensure_buf(ctx, part, (size_t)PROC_WAVES * NACC_MAX * sizeof(double));
    if (rc) return rc;
    rc = ensure_list(ctx, list, 0, 0, 0);   // an (empty) list object must exist
    if (rc) return rc;
    if (!ctx->kept_cnt.p) {
        rc = ensure_buf(ctx, ctx->kept_cnt, PROC_WAVES * sizeof(uint32_t));
        if (rc) return rc;
        HIP_TRY(ctx, hipMemsetAsync(ctx->kept_cnt.p, 0, PROC_WAVES * sizeof(uint32_t), ctx->stream));
    }
    if (mode == PROC_FLOW)   // the kept list is sized from the pair set this pass evaluates
        rc = ensure_list(ctx, LIST_KEPT, ctx->fixed.np, ctx->moving.np, 0);
    else
        rc = ensure_list(ctx, LIST_KEPT, 0, 0, 0);
    if (rc) return rc;
    ProcessArgs a{};
    a.pos_a = pos_a; a.feat_a = feat_a;
    a.pos_b = pos_b; a.feat_b = feat_b;
    a.tiles = (const TileEntry *)ctx->lists[list].a.p;
    a.kept_ij = (uint2 *)ctx->lists[LIST_KEPT].a.p;
    a.kept_a = (float *)ctx->lists[LIST_KEPT].b.p;
    a.kept_cnt = (uint32_t *)ctx->kept_cnt.p;
    a.partials = (double *)part.p;
    a.st = ctx->st;
    a.st2 = static_cast<DevState *>(ctx->st2);
    a.subcap = ctx->lists[list].cap / NSUB;
    a.nblk = ctx->proc_blocks;
    a.kept_wcap = ctx->lists[LIST_KEPT].cap / (uint32_t)(4 * ctx->proc_blocks);
    a.list = list;
    a.first_counted = first_counted;
    a.tf_a = tf_a; a.tf_b = tf_b;
    if (pre_transform(ctx) && ctx->pos_bt.p) {   // (written by this iteration's xy filter launch)
        if (tf_b && pos_b == ctx->moving.pos) { a.pos_b = (const float4 *)ctx->pos_bt.p; a.tf_b = 0; }
        if (tf_a && pos_a == ctx->moving.pos) { a.pos_a = (const float4 *)ctx->pos_bt.p; a.tf_a = 0; }   // acvo: the yy pass
    }
    a.check_done = check_done;
    a.need_d2 = (ctx->prm.mode == CVO_HIP_MODE_ACVO || !ctx->in_loop || ctx->cur_trace_cap > 0) ? 1 : 0;   // (a trace record holds the sum of the weights)
    a.weight = ctx->prm.color_scale > 0.0f ? 1 : 0;   // the MATLAB object's weight: its own instantiation
    const bool no_pack = ctx->opt.no_pack;   // (test switch "kept_pack" = 0: 8 + 4 byte kept entries)
    a.kept_packed = (!no_pack && ctx->fixed.np <= 65536 && ctx->moving.np <= 65536) ? 1 : 0;
    if (!a.kept_packed && !no_pack && a.weight == 0 && ctx->fixed.np <= 262144 && ctx->moving.np <= 262144) {
        // 8 bytes for larger clouds too (ProcessArgs::kept_packed == 2): a member's weight a = ck * k is a positive
        // float32 with sp < a <= fl(fl(c_sigma^2) fl(sigma^2)) -- the two exp are <= 1 (ref cvo.cpp:143-153 as
        // pair_weight computes it); if those two bounds lie within 16 binades, 4 bits of exponent do
        const float amax = (float)ctx->dprm.cs2_d * (float)ctx->dprm.s2_d;
        uint32_t blo, bhi;
        std::memcpy(&blo, &ctx->dprm.sp, sizeof(blo));
        std::memcpy(&bhi, &amax, sizeof(bhi));
        const uint32_t elo = blo >> 23, ehi = bhi >> 23;   // (both positive: the sign bit is clear)
        if (ctx->dprm.sp > 0.0f && amax > ctx->dprm.sp && elo >= 1 && ehi < 255 && ehi - elo <= 15) {
            a.kept_packed = 2;
            a.kept_ebase = elo;
        }
    }
    if ((mode == PROC_FLOW && list == LIST_XY) || (mode == PROC_SELF && (list == LIST_XX || list == LIST_YY))) {
        const bool no_cand = ctx->opt.no_cand;
        ctx->ck_nblk[list] = 0;
        // (clouds of up to 65536 rows: i and j share a word.  12-byte records for larger clouds were built, bit-identical,
        // and measured SLOWER -- one 200k x 200k registration 81.3 -> 91.0 ms, 100k x 100k 21.7 -> 23.8, acvo 18.9 -> 21.8: at
        // those sizes the list passes are bound by memory requests and the record is more bytes to stream; profiles/r03_ab.txt 8)
        if (!no_cand && pre_transform(ctx) && !(ctx->prm.color_scale > 0.0f) && a.kept_packed == 1) {   // (the same plans: synchronous lists)
            // (an optimisation: if its memory cannot be had, the pass expands the tile list every time)
            int rc_c = ensure_buf(ctx, ctx->cand[list], (size_t)ctx->lists[LIST_KEPT].cap * sizeof(uint2));
            if (!rc_c) rc_c = ensure_buf(ctx, ctx->cand_cnt[list], PROC_WAVES * sizeof(uint32_t));
            if (!rc_c) {
                a.cand = (uint2 *)ctx->cand[list].p;
                a.cand_cnt = (uint32_t *)ctx->cand_cnt[list].p;
                ctx->ck_nblk[list] = a.nblk;
            } else {
                (void)hipGetLastError();
                ctx->err = "";
            }
        }
    }
    if (ctx->in_loop && ctx->use_async) {
        a.async_xy = 1;
        a.tiles_b = (const TileEntry *)ctx->lists[LIST_XYB].a.p;
        // Head mode (one registration on its own, plan_lone): the flow pass keeps a candidate record per buffer of
        // the double-buffered xy list -- the pass after a buffer is switched to expands and records, the passes
        // over the same buffer stream (DevHead::xy_ck).  The kernels of every other plan ignore the fields.
        const bool no_cand = ctx->opt.no_cand;
        if (mode == PROC_FLOW && list == LIST_XY && ctx->plan_recording && ctx->lone && ctx->allow_head && !multi_rank(ctx) &&
            !no_cand && a.kept_packed == 1 && !(ctx->prm.color_scale > 0.0f)) {
            const size_t bytes = (size_t)ctx->lists[LIST_KEPT].cap * sizeof(uint2);
            int rc_c = ensure_buf(ctx, ctx->cand[LIST_XY], bytes);
            if (!rc_c) rc_c = ensure_buf(ctx, ctx->cand_xyb, bytes);
            if (!rc_c) rc_c = ensure_buf(ctx, ctx->cand_cnt[LIST_XY], PROC_WAVES * sizeof(uint32_t));
            if (!rc_c) rc_c = ensure_buf(ctx, ctx->cand_cnt_xyb, PROC_WAVES * sizeof(uint32_t));
            if (!rc_c) {
                a.cand = (uint2 *)ctx->cand[LIST_XY].p;
                a.cand_cnt = (uint32_t *)ctx->cand_cnt[LIST_XY].p;
                a.cand_b = (uint2 *)ctx->cand_xyb.p;
                a.cand_cnt_b = (uint32_t *)ctx->cand_cnt_xyb.p;
                ctx->ck_nblk[LIST_XY] = a.nblk;
            } else {   // (an optimisation: without its memory the pass expands the tile list every time)
                (void)hipGetLastError();
                ctx->err = "";
            }
        }
    }
    if (mode == PROC_SELF && ctx->in_loop && ctx->use_async_self) {
        a.async_self = list == LIST_XX ? 1 : 2;
        a.tiles_b = (const TileEntry *)ctx->lists[list == LIST_XX ? LIST_XXB : LIST_YYB].a.p;
        // (head mode: candidate records for both buffers of the self lists too, see the xy list above)
        const bool no_cand = ctx->opt.no_cand;
        if (ctx->plan_recording && ctx->lone && ctx->allow_head && !multi_rank(ctx) && !no_cand && a.kept_packed == 1 &&
            !(ctx->prm.color_scale > 0.0f)) {
            const int l = list == LIST_XX ? 0 : 1;
            const size_t bytes = (size_t)ctx->lists[LIST_KEPT].cap * sizeof(uint2);
            int rc_c = ensure_buf(ctx, ctx->cand[list], bytes);
            if (!rc_c) rc_c = ensure_buf(ctx, ctx->cand_sfb[l], bytes);
            if (!rc_c) rc_c = ensure_buf(ctx, ctx->cand_cnt[list], PROC_WAVES * sizeof(uint32_t));
            if (!rc_c) rc_c = ensure_buf(ctx, ctx->cand_cnt_sfb[l], PROC_WAVES * sizeof(uint32_t));
            if (!rc_c) {
                a.cand = (uint2 *)ctx->cand[list].p;
                a.cand_cnt = (uint32_t *)ctx->cand_cnt[list].p;
                a.cand_b = (uint2 *)ctx->cand_sfb[l].p;
                a.cand_cnt_b = (uint32_t *)ctx->cand_cnt_sfb[l].p;
                ctx->ck_nblk[list] = a.nblk;
            } else {
                (void)hipGetLastError();
                ctx->err = "";
            }
        }
    }
    const bool twist = mode == PROC_STEP && ctx->merge_twist;
    if (twist) {
        a.flow_part = (const double *)ctx->part_flow.p;
        a.xx_part = (const double *)ctx->part_xx.p;
        a.yy_part = (const double *)ctx->part_yy.p;
        a.trace = ctx->cur_trace; a.trace_cap = ctx->cur_trace_cap;
        a.acvo = ctx->prm.mode == CVO_HIP_MODE_ACVO;
        a.done_mirror = ctx->done_mirror;
        a.comm = ctx->comm_table;   // (ranks through the mailboxes: the exchange runs inside the launch)
    }
    const bool build = mode == PROC_FLOW && ctx->have_xy_build;
    ctx->have_xy_build = ctx->have_xy_build && mode != PROC_FLOW;
    if (ctx->rec) {
        RecOp op; op.kind = RecOp::PROCESS; op.mode = twist ? kProcStepTwist : (build ? kFlowBuild : mode);
        op.p = a;
        if (build) op.f = ctx->xy_build;
        ctx->rec->push_back(op);
        return CVO_HIP_OK;
    }
    EventPair ev{};
    const bool timed = ctx->profiling && !build && (mode == PROC_FLOW || mode == PROC_STEP);
    if (timed) {
        HIP_TRY(ctx, hipEventCreate(&ev.a));
        HIP_TRY(ctx, hipEventCreate(&ev.b));
        ev.kind = mode == PROC_FLOW ? kEvProcFlow : kEvProcStep;
        ev.iter_tag = ctx->iter_tag;
        ev.pairs = 0.0;
    }
    // (a build riding in the flow launch exists in the table path only: the asynchronous scheme is off
    // whenever launches are issued by value -- profiling, stream-level all-reduces)
    if (build) return fail(ctx, CVO_HIP_ERR_INVALID, "asynchronous build outside the table path");
    if (twist) launch_step_twist_group(&a, 1, ctx->stream, ev.a, ev.b);
    else launch_process(mode, a, ctx->stream, ev.a, ev.b);
    if (timed) ctx->events.push_back(ev);
    HIP_TRY(ctx, hipGetLastError());
    return CVO_HIP_OK;
}

void emit_post_flow(cvo_hip_ctx *ctx, const PostFlowArgs &pa)
{
    if (ctx->rec) {
        RecOp op; op.kind = RecOp::POST_FLOW; op.pf = pa;
        ctx->rec->push_back(op);
    } else {
        launch_post_flow(pa, ctx->stream);
    }
}

void emit_post_step(cvo_hip_ctx *ctx, const PostStepArgs &pa)
{
    if (ctx->rec) {
        RecOp op; op.kind = RecOp::POST_STEP; op.ps = pa;
        ctx->rec->push_back(op);
    } else {
        launch_post_step(pa, ctx->stream);
    }
}

// n_exec >= 0: launches tagged with an iteration >= n_exec were queued past
// convergence and returned at once; they are not sweeps and are not counted.
// Inside align() a launch whose list is re-used returns at once as well: `fin`
// (the final state) tells which iterations rebuilt which list.
int drain_events(cvo_hip_ctx *ctx, int n_exec, const DevState *fin)
{
    for (auto &ev : ctx->events) {
        float ms = 0.f;
        HIP_TRY(ctx, hipEventSynchronize(ev.b));
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ev.a, ev.b));
        bool live = !(n_exec >= 0 && ev.iter_tag >= n_exec);
        if (live && fin && ev.iter_tag >= 0 && ev.kind >= 0 && ev.kind < 3)
            live = (fin->built[ev.kind][(ev.iter_tag >> 5) & 63] >> (ev.iter_tag & 31)) & 1u;
        if (!live) {
            // skipped launch
        } else if (ev.kind == kEvProcFlow) {
            ctx->prof.proc_flow_ms += ms; ctx->prof.proc_flow_launches++;
        } else if (ev.kind == kEvProcStep) {
            ctx->prof.step_ms += ms; ctx->prof.step_launches++;
        } else if (ev.kind == LIST_XY) {
            ctx->prof.flow_ms += ms; ctx->prof.flow_launches++; ctx->prof.flow_pairs += ev.pairs;
        } else {
            ctx->prof.self_ms += ms; ctx->prof.self_launches++; ctx->prof.self_pairs += ev.pairs;
        }
        (void)hipEventDestroy(ev.a);
        (void)hipEventDestroy(ev.b);
    }
    ctx->events.clear();
    return CVO_HIP_OK;
}

// Sums over ranks: either between the kernels (RCCL / the caller's hook: a stream-level
// all-reduce, two extra launches per reduction and no graph capture) or inside the post
// kernels through the mailboxes (nothing for the host to do).
bool host_reduce(const cvo_hip_ctx *ctx) { return !ctx->comm_table && (ctx->comm || ctx->user_allreduce); }
bool multi_rank(const cvo_hip_ctx *ctx) { return ctx->comm_table || ctx->comm || ctx->user_allreduce; }

// the parameter block of the kernels of align(): the context's, plus the mode of this run
DevParams loop_params(const cvo_hip_ctx *ctx)
{
    DevParams dp = ctx->dprm;
    // Width of the lists: a wider list is rebuilt less often but costs every flow pass more
    // candidates ((1 + margin)^2).  Measured (profiles/r02_ab.txt): up to ~14k points a side, where a
    // build is a large part of an iteration, 25 % beats 15 % (32 distinct 10k x 10k pairs 2273 ->
    // 2398 registrations/s, one at a time 1.77 -> 1.71 ms); at 20k x 20k it loses (917 -> 792).
    dp.list_margin = ((double)ctx->fixed.n * (double)ctx->moving.n <= 2.0e8) ? 0.25f : 0.15f;
    // acvo with resident runs: its length scale sits at its floor for most of a registration -- lists of ~24 mm with 6 mm of room,
    // used up every few iterations while the cloud still moves, and every rebuild ends a run (two launch-per-pass slots and a new
    // entry); a candidate inside a run costs next to nothing: 50 % (3k x 3k 1 110 -> 1 187 registrations/s, 10k 712 -> 749; 80 % and
    // 120 % lose again: the exchanges grow with the solvers the wider records need -- profiles/r06_ab.txt 2)
    if (ctx->prm.mode == CVO_HIP_MODE_ACVO && ctx->lone && ctx->use_async_self && runs_allowed(ctx) && !ctx->opt.no_acvo_run && ctx->allow_head &&
        (double)ctx->fixed.n * (double)ctx->moving.n <= 2.0e8)
        dp.list_margin = 0.5f;
    if (ctx->opt.list_margin >= 0.0f) dp.list_margin = ctx->opt.list_margin;   // (test switch "list_margin"; 0 = rebuild every iteration)
    dp.async_xy = ctx->use_async ? 1 : 0;
    dp.async_self = ctx->use_async_self ? 1 : 0;
    // Head mode: a build is named a slot earlier than it is made and costs its launch 10 us; later is better
    // (0.7 / 0.85 / 0.9 / 0.95 of the margin gone: 10k x 10k 711 / 728 / 733 / 732 registrations/s, 14k 432 / 444 / 445 /
    // 444, 6k 694 / 694 / 706 / 705, 3k 788 / 794 / 792 / 792; profiles/r03_ab.txt 18)
    if (ctx->use_async && ctx->lone && ctx->allow_head && !multi_rank(ctx)) dp.build_at = 0.9f;
    // (plans with resident runs, the conditions of enqueue_step: what a run's registers hold -- the plan keeps a list that has become
    // wide while its record still fits, plan_xy_async)
    if (ctx->use_async && ctx->lone && ctx->allow_head && ctx->allow_run && !multi_rank(ctx) && ctx->prm.mode == CVO_HIP_MODE_CVO &&
        !(ctx->prm.color_scale > 0.0f) && !ctx->post_dbg && ctx->allow_merge && ctx->fixed.np <= 65536 && ctx->moving.np <= 65536 && !ctx->opt.no_cand)
        dp.run_cand_cap = (float)ctx->run_g_max * (float)(RUN_BLOCK * (RUN_R + RUN_L));
    return dp;
}

// all-reduce `count` doubles of st->red starting at `off` over the ranks
int reduce_over_ranks(cvo_hip_ctx *ctx, int off, int count)
{
    double *buf = reinterpret_cast<double *>(reinterpret_cast<char *>(ctx->st) +
                                             offsetof(DevState, red)) + off;
    if (ctx->comm_table) return CVO_HIP_OK;   // exchanged inside the post kernel already
    if (ctx->comm) {
        if (cvo_comm_allreduce(ctx->comm, buf, count, ctx->stream) != 0)
            return fail(ctx, CVO_HIP_ERR_COMM, cvo_comm_last_error(ctx->comm));
    } else if (ctx->user_allreduce) {
        if (ctx->user_allreduce(ctx->user_allreduce_arg, buf, count, (void *)ctx->stream) != 0)
            return fail(ctx, CVO_HIP_ERR_COMM, "user all-reduce failed");
    }
    return CVO_HIP_OK;
}

// flow side of one iteration: dense filter(s) -> candidate list(s) -> exact
// evaluation -> reduction (+ all-reduce) (+ the O(1) maths)
int enqueue_flow(cvo_hip_ctx *ctx, bool tf_moving, int check_done, bool do_math,
                 cvo_hip_trace *trace, int trace_cap)
{
    const bool acvo = ctx->prm.mode == CVO_HIP_MODE_ACVO;
    const int tfm = tf_moving ? 1 : 0;
    int rlo, rhi, slo, shi;
    shard_ranges(ctx, rlo, rhi, slo, shi);
    // acvo on its own stream: the three filters share one launch and so do the two
    // self passes (the argument blocks are recorded, then issued as groups)
    const bool group_lists = acvo && !ctx->rec && !ctx->profiling;
    std::vector<RecOp> local;
    if (group_lists) ctx->rec = &local;
    int rc = enqueue_filter(ctx, LIST_XY, ctx->fixed, rlo, rhi, 0, ctx->moving, tfm, check_done);
    if (!rc)
        rc = enqueue_process(ctx, PROC_FLOW, LIST_XY, ctx->part_flow, ctx->fixed.pos, ctx->fixed.feat,
                             0, ctx->moving.pos, ctx->moving.feat, tfm, 0, check_done);
    if (!rc && acvo) {
        // Axx rows of this shard vs all of x; Ayy rows of this shard vs all of y
        rc = enqueue_filter(ctx, LIST_XX, ctx->fixed, rlo, rhi, 0, ctx->fixed, 0, check_done);
        if (!rc)
            rc = enqueue_process(ctx, PROC_SELF, LIST_XX, ctx->part_xx, ctx->fixed.pos,
                                 ctx->fixed.feat, 0, ctx->fixed.pos, ctx->fixed.feat, 0, 0, check_done);
        if (!rc)
            rc = enqueue_filter(ctx, LIST_YY, ctx->moving, slo, shi, tfm, ctx->moving, tfm, check_done);
        if (!rc)
            rc = enqueue_process(ctx, PROC_SELF, LIST_YY, ctx->part_yy, ctx->moving.pos,
                                 ctx->moving.feat, tfm, ctx->moving.pos, ctx->moving.feat, tfm,
                                 1 /* rows below st->n_fixed do not count */, check_done);
    }
    if (group_lists) {
        ctx->rec = nullptr;
        if (!rc) {
            FilterArgs f[3], build{}, ahead[2];
            ProcessArgs flow{}, self[2];
            int nf = 0, ns = 0, na = 0;
            bool have_flow = false, have_build = false;
            for (const RecOp &op : local) {
                if (op.kind == RecOp::FILTER && op.mode == kFilterAhead && na < 2) ahead[na++] = op.f;
                else if (op.kind == RecOp::FILTER && nf < 3) f[nf++] = op.f;
                else if (op.kind == RecOp::PROCESS && (op.mode == PROC_FLOW || op.mode == kFlowBuild)) {
                    flow = op.p; have_flow = true;
                    if (op.mode == kFlowBuild) { build = op.f; have_build = true; }
                }
                else if (op.kind == RecOp::PROCESS && op.mode == PROC_SELF && ns < 2) self[ns++] = op.p;
            }
            // (eager by-value launches: synchronous lists only, see enqueue_process)
            if (nf) launch_filter_group(f, nf, ctx->stream);
            if (have_flow && !have_build) launch_process_group(PROC_FLOW, &flow, 1, ctx->stream);
            else if (have_flow) rc = fail(ctx, CVO_HIP_ERR_INVALID, "asynchronous build outside the table path");
            if (ns) launch_process_group(PROC_SELF, self, ns, ctx->stream);
            HIP_TRY(ctx, hipGetLastError());
        }
    }
    if (rc) return rc;
    if (ctx->merge_twist) return CVO_HIP_OK;   // k_step_twist does the rest of compute_flow
    PostFlowArgs pa{};
    pa.st = ctx->st;
    pa.prm = ctx->in_loop ? loop_params(ctx) : ctx->dprm;
    pa.trace = trace; pa.trace_cap = trace_cap;
    pa.check_done = check_done;
    pa.done_mirror = ctx->done_mirror;
    pa.nblk = ctx->proc_blocks;
    pa.part_flow = (const double *)ctx->part_flow.p;
    pa.part_xx = (const double *)ctx->part_xx.p;
    pa.part_yy = (const double *)ctx->part_yy.p;
    pa.comm = ctx->comm_table;
    if (host_reduce(ctx)) {
        pa.flags = POST_REDUCE;
        emit_post_flow(ctx, pa);
        rc = reduce_over_ranks(ctx, RED_FLOW, RED_STEP - RED_FLOW);
        if (rc) return rc;
        if (do_math) {
            pa.flags = POST_MATH;
            emit_post_flow(ctx, pa);
        }
    } else {
        pa.flags = POST_REDUCE | (do_math ? POST_MATH : 0);
        emit_post_flow(ctx, pa);
    }
    HIP_TRY(ctx, hipGetLastError());
    return CVO_HIP_OK;
}

// step-size side: streams the xy list again with the weights PROC_FLOW kept
int enqueue_step(cvo_hip_ctx *ctx, int check_done, bool do_math, cvo_hip_trace *trace,
                 int trace_cap)
{
    int rc = enqueue_process(ctx, PROC_STEP, LIST_XY, ctx->part_step, ctx->fixed.pos,
                             ctx->fixed.feat, 0, ctx->moving.pos, ctx->moving.feat, 1, 0,
                             check_done);
    if (rc) return rc;
    PostStepArgs pa{};
    pa.st = ctx->st;
    pa.st2 = static_cast<DevState *>(ctx->st2);
    pa.prm = ctx->in_loop ? loop_params(ctx) : ctx->dprm;
    pa.trace = trace; pa.trace_cap = trace_cap;
    pa.check_done = check_done;
    pa.done_mirror = ctx->done_mirror;
    pa.progress_mirror = ctx->progress_mirror;
    for (int l = 0; l < 3; ++l) pa.ck_nblk[l] = ctx->plan_recording ? ctx->ck_nblk[l] : 0;
    pa.nblk = ctx->merge_twist ? ctx->proc_blocks / STEP_TWIST_ROWS_DIV : ctx->proc_blocks;
    pa.part_step = (const double *)ctx->part_step.p;
    pa.dbg = ctx->post_dbg;
    pa.comm = ctx->comm_table;
    // Resident runs (cvo_kernels.hip kt_run): one cvo registration with its launches to itself, in head mode, on
    // candidate records (plan_lone decides whether the plan really is a head-mode plan)
    if (ctx->plan_recording && ctx->lone && ctx->allow_head && runs_allowed(ctx) && ctx->use_async && ctx->merge_twist &&
        !multi_rank(ctx) && (ctx->prm.mode == CVO_HIP_MODE_CVO || (ctx->use_async_self && !ctx->opt.no_acvo_run)) &&
        !(ctx->prm.color_scale > 0.0f) && !ctx->post_dbg) {
        const bool fresh = ctx->run_mail.p == nullptr;
        if (ensure_buf(ctx, ctx->run_mail, sizeof(RunMail)) == CVO_HIP_OK) {
            if (fresh) HIP_TRY(ctx, hipMemsetAsync(ctx->run_mail.p, 0, sizeof(RunMail), loop_stream(ctx)));
            pa.run_mail = (RunMail *)ctx->run_mail.p;
            pa.run_mirror = ctx->run_mirror;
            pa.run_iters = 64;
            pa.run_g_max = std::max(8, std::min(ctx->run_g_max, ctx->run_g_call));
            pa.run_timeout_ticks = (long long)(ctx->opt.run_timeout_ms * 1.0e5);   // (100 MHz; 0: the kernel's own second)
            pa.run_fault = ctx->opt.run_fault;
            // side builds (kt_run): a stream of the library's own beside the caller's, made once
            if (!ctx->opt.no_side_builds) {
                if (!ctx->side_stream && hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking) != hipSuccess) {
                    (void)hipGetLastError();
                    ctx->side_stream = nullptr;
                }
                if (ctx->side_stream) {
                    pa.side_mirror = ctx->side_mirror;
                    pa.run_build_at = ctx->opt.run_build_at;
                }
            }
        } else {   // (an optimisation: without its memory the plan has no runs)
            (void)hipGetLastError();
            ctx->err = "";
        }
    }
    if (ctx->plan_recording && ctx->lone) pa.hint_mirror = ctx->hint_mirror;
    // (one registration on its own, one rank: its final head goes to pinned memory with the verdict, job_pump)
    if (ctx->plan_recording && ctx->lone && !multi_rank(ctx) && !ctx->opt.no_final_mirror) pa.final_mirror = ctx->final_mirror;
    if (host_reduce(ctx)) {
        pa.flags = POST_REDUCE;
        emit_post_step(ctx, pa);
        rc = reduce_over_ranks(ctx, RED_STEP, RED_N - RED_STEP);
        if (rc) return rc;
        if (do_math) {
            pa.flags = POST_MATH;
            emit_post_step(ctx, pa);
        }
    } else {
        pa.flags = POST_REDUCE | (do_math ? POST_MATH : 0);
        emit_post_step(ctx, pa);
    }
    HIP_TRY(ctx, hipGetLastError());
    return CVO_HIP_OK;
}

// Low-level entry points run one list at a time and cannot resume: grow the
// lists until nothing overflows.  Returns 1 if the caller must redo its launches.
int check_overflow_and_grow(cvo_hip_ctx *ctx, bool *redo)
{
    DevState *h = &ctx->st_host[0];
    HIP_TRY(ctx, hipMemcpyAsync(h->sub, reinterpret_cast<char *>(ctx->st) + offsetof(DevState, sub),
                                sizeof(DevState) - offsetof(DevState, sub), hipMemcpyDeviceToHost,
                                ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *redo = false;
    for (int l = 0; l < LIST_N; ++l)
        if (h->ovf[0][l] | h->ovf[1][l]) {
            uint32_t worst = 0;
            for (int q = 0; q < NSUB; ++q) worst = std::max(worst, h->sub[l][q]);
            const double need = std::max((double)worst * NSUB, (double)ctx->lists[l].cap);
            const double grown = std::min(4.0e9, need * 1.25 + 1024.0);
            int rc = ensure_list(ctx, l, 0, 0, grown);
            if (rc) return rc;
            *redo = true;
        }
    return CVO_HIP_OK;
}

// Allocate (or grow) every device buffer the loop will touch for the clouds
// that are set, so that no allocation can happen inside a graph capture.
int prepare_buffers(cvo_hip_ctx *ctx)
{
    const bool acvo = ctx->prm.mode == CVO_HIP_MODE_ACVO;
    int rlo, rhi, slo, shi;
    shard_ranges(ctx, rlo, rhi, slo, shi);
    // (padded sizes, the ones enqueue_filter sees: the capacities must not move while a batch is captured)
    int rc = ensure_list(ctx, LIST_XY, rhi - rlo, ctx->moving.np, 0);
    if (!rc) rc = ensure_list(ctx, LIST_XYB, 0, 0, (double)ctx->lists[LIST_XY].cap);   // second xy buffer
    if (!rc) rc = ensure_list(ctx, LIST_KEPT, ctx->fixed.np, ctx->moving.np, 0);
    if (!rc && acvo) rc = ensure_list(ctx, LIST_XX, rhi - rlo, ctx->fixed.np, 0);
    if (!rc && acvo) rc = ensure_list(ctx, LIST_YY, shi - slo, ctx->moving.np, 0);
    if (!rc && acvo) rc = ensure_list(ctx, LIST_XXB, 0, 0, (double)ctx->lists[LIST_XX].cap);
    if (!rc && acvo) rc = ensure_list(ctx, LIST_YYB, 0, 0, (double)ctx->lists[LIST_YY].cap);
    for (DevBuf *b : {&ctx->part_flow, &ctx->part_xx, &ctx->part_yy, &ctx->part_step})
        if (!rc) rc = ensure_buf(ctx, *b, (size_t)PROC_WAVES * NACC_MAX * sizeof(double));
    if (!rc && !ctx->kept_cnt.p) {
        rc = ensure_buf(ctx, ctx->kept_cnt, PROC_WAVES * sizeof(uint32_t));
        if (!rc) HIP_TRY(ctx, hipMemsetAsync(ctx->kept_cnt.p, 0, PROC_WAVES * sizeof(uint32_t), loop_stream(ctx)));
    }
    if (!rc) ctx->warm = true;
    return rc;
}

int enqueue_iterations(cvo_hip_ctx *ctx, int count, int tag0, int trace_cap)
{
    int rc = CVO_HIP_OK;
    // (k_step_twist: one rank, or ranks whose sums travel through the mailboxes -- the exchange then runs inside the launch;
    // the stream-level all-reduces need their own launches in between; ranks that share ONE GPU -- tests, rehearsals -- keep the
    // single-block exchange: every block of a rank's launch spinning for a peer keeps that peer's kernels off the GPU.  The test
    // switch forces the in-launch exchange there, for clouds whose launches leave room)
    ctx->merge_twist = ctx->allow_merge && !host_reduce(ctx) && !(ctx->comm_table && ctx->mail_shared_device && !ctx->opt.twist_on_shared_gpu);
    ctx->in_loop = true;
    ctx->cur_trace = ctx->trace_dev;
    ctx->cur_trace_cap = trace_cap;
    for (int q = 0; q < count && !rc; ++q) {
        ctx->iter_tag = tag0 >= 0 ? tag0 + q : -1;
        rc = enqueue_flow(ctx, true, 1, true, ctx->trace_dev, trace_cap);
        if (!rc) rc = enqueue_step(ctx, 1, true, ctx->trace_dev, trace_cap);
    }
    ctx->merge_twist = false;
    ctx->in_loop = false;
    ctx->iter_tag = -1;
    return rc;
}

void drop_graphs(cvo_hip_ctx *ctx) { ctx->plans.drop(); }

// ---------------------------------------------------------------------------
// From the recorded launches of one iteration (RecOp) to a launch plan + the slot contents.
TLaunch mk_launch(int kernel, int q, unsigned gx, unsigned gz, unsigned smem)
{
    TLaunch l{};
    l.kernel = kernel; l.q = q; l.gx = gx; l.gz = gz; l.smem = smem;
    return l;
}

long long filter_items(const FilterArgs &f) { return (long long)f.gx * f.gy; }
constexpr long long kFusedFilterBlocks = 256;   // (64 / 256 / 512 per slot: profiles/r05_ab.txt 10)

// One registration with the launches to itself: the flow side of an iteration is merged into
// as few launches as its scheme allows (what enqueue_flow does for eager launches):
//   flow pass + xy build + both self passes + both self builds      -> kt_flow_build6
//   flow pass + xy build + xx / yy filters (self passes afterwards) -> kt_flow_build3, kt_self2
//   flow pass + xy build                                            -> kt_flow_build
//   synchronous lists                                               -> kt_filter(_group), kt_process, kt_self2
// Head mode (cvo_kernels.hip "the head"), where the scheme allows it -- asynchronous builds, step pass with the
// twist in front, one rank: the post-step launch is gone; its argument block rides in the flow launch's entry
// (op[q].ps), every flow / self block runs it as its head.
bool plan_lone(const std::vector<RecOp> &ops, Slot &slot, std::vector<TLaunch> &plan, const bool allow_head, bool *head_mode,
               std::vector<TLaunch> *pre, std::vector<TLaunch> *side)
{
    plan.clear();
    if (pre) pre->clear();
    if (side) side->clear();
    *head_mode = false;
    std::memset(&slot, 0, sizeof(slot));
    slot.active = 1;
    const long long fbmax = filter_blocks_cap();
    FilterArgs f[3], build{}, ahead[2];
    ProcessArgs flow{}, self[2];
    int nf = 0, ns = 0, na = 0;
    bool have_flow = false, have_build = false;
    size_t at = 0;
    for (; at < ops.size(); ++at) {   // the flow side: up to the first post / step launch
        const RecOp &op = ops[at];
        if (op.kind == RecOp::FILTER && op.mode == kFilterAhead && na < 2) ahead[na++] = op.f;
        else if (op.kind == RecOp::FILTER && nf < 3) f[nf++] = op.f;
        else if (op.kind == RecOp::PROCESS && (op.mode == PROC_FLOW || op.mode == kFlowBuild)) {
            flow = op.p; have_flow = true;
            if (op.mode == kFlowBuild) { build = op.f; have_build = true; }
        } else if (op.kind == RecOp::PROCESS && op.mode == PROC_SELF && ns < 2) self[ns++] = op.p;
        else break;
    }
    int q = 0;
    const int ns_all = ns;
    const bool self_async[2] = {ns > 0 && self[0].async_self != 0, ns > 1 && self[1].async_self != 0};
    auto smem_of = [](int jt) { return (unsigned)filter_smem_bytes(jt); };
    auto smem_head = [](int jt) { return (unsigned)filter_smem_bytes(jt); };
    // what follows the flow side must be exactly: step pass with the twist, post-step (reduce + maths, no exchange)
    const bool rest_fits = at + 2 == ops.size() && ops[at].kind == RecOp::PROCESS && ops[at].mode == kProcStepTwist &&
                           ops[at + 1].kind == RecOp::POST_STEP && ops[at + 1].ps.comm == nullptr &&
                           ops[at + 1].ps.flags == (POST_REDUCE | POST_MATH) && ops[at + 1].ps.st2 != nullptr;
    // (acvo: flow pass and both self passes, 3 x np blocks, all run the head; with the 1024 blocks per pass
    // of round 2 three heads per SIMD took turns at the vector ALU and an iteration was a third SLOWER,
    // 40.5 -> 55 us at 10k x 10k -- job_begin gives acvo's passes 256 / 128 blocks now, profiles/r03_ab.txt)
    const bool head = allow_head && rest_fits && have_flow && have_build &&
                      ((na == 2 && ns == 2 && nf == 0) || (na == 0 && ns == 0 && nf == 0));
    if (!head) {
        // The candidate records of double-buffered lists (ProcessArgs::cand_b, DevHead::xy_ck / sf_ck) belong to
        // head mode alone: enqueue_process fills them in before the plan is known.  A plan that falls back to the
        // classic merged launches (CVO_HIP_NO_MERGE, CVO_HIP_NO_HEAD) must not stream them -- its post-step
        // kernel would tie ONE record to both buffers (DevHead::ck_nblk) and a pass over the second buffer would
        // stream the first one's pairs.
        auto strip = [](ProcessArgs &p) {
            if (p.cand_b) { p.cand = nullptr; p.cand_b = nullptr; p.cand_cnt = nullptr; p.cand_cnt_b = nullptr; }
        };
        if (have_flow && flow.async_xy) strip(flow);
        for (int w = 0; w < ns; ++w)
            if (self[w].async_self) strip(self[w]);
    }
    if (have_flow && have_build && ((na == 2 && ns == 2) || nf == 2)) {
        // (op[q]: flow pass + xy build; op[q + 1], op[q + 2]: the xx / yy filters and, `six`, the self passes)
        const bool six = na == 2 && ns == 2;
        OpArgs &o = slot.op[q];
        o.p = flow; o.f = build;
        for (int w = 0; w < 2; ++w) {
            slot.op[q + 1 + w].f = six ? ahead[w] : f[w];
            if (six) slot.op[q + 1 + w].p = self[w];
        }
        const long long cap = std::max<long long>(64, fbmax / 2);
        o.np = std::max(8, flow.nblk);
        o.n0 = (int)filter_grid_cap(filter_items(o.f), cap);
        o.n1 = (int)filter_grid_cap(filter_items(slot.op[q + 1].f), cap);
        o.n2 = (int)filter_grid_cap(filter_items(slot.op[q + 2].f), cap);
        const int jt = std::max(o.f.jt, std::max(slot.op[q + 1].f.jt, slot.op[q + 2].f.jt));
        if (head) { o.ps = ops[at + 1].ps; }
        else { o.ps.run_mail = nullptr; }
        plan.push_back(mk_launch(head ? TK_HFLOW_BUILD6 : (six ? TK_FLOW_BUILD6 : TK_FLOW_BUILD3), q,
                                 (unsigned)((six ? 3 : 1) * o.np + o.n0 + o.n1 + o.n2), 1, head ? smem_head(jt) : smem_of(jt)));
        // acvo: a RUN batch begins with kt_run_acvo (head and flow pass from this entry, the self passes from the next two, the
        // trace from the step launch's, op[q + 3]): candidate records on both buffers of all three lists, the clouds read as they
        // came (x never transformed, y by the slot's transform)
        if (head && six && pre && o.ps.run_mail && flow.cand && flow.cand_b && flow.kept_packed == 1 && flow.tf_a == 0 && flow.tf_b == 1 &&
            flow.weight == 0 && 4 * flow.nblk <= PROC_WAVES && self[0].cand && self[0].cand_b && self[1].cand && self[1].cand_b &&
            self[0].tf_a == 0 && self[0].tf_b == 0 && self[1].tf_a == 1 && self[1].tf_b == 1 && 4 * self[0].nblk <= PROC_WAVES &&
            4 * self[1].nblk <= PROC_WAVES && q + 3 < 16) {
            pre->push_back(mk_launch(TK_RUN_ACVO, q | ((q + 3) << 4), 1u + (unsigned)std::max(8, std::min((int)RUN_G, o.ps.run_g_max)), 1));   // (no more blocks than the run may use: a block that is not needed still takes a compute unit until it knows)
            pre->push_back(mk_launch(TK_RUN_ACVO, q | ((q + 3) << 4), 1u + RUN_G_SMALL, 1));   // (for narrow records: launch_batch picks one)
            o.ps.side_mirror = nullptr;   // (side builds are cvo's alone: see kt_run's side_can)
        }
        q += 3;
        if (six) ns = 0;
        nf = 0;
    } else {
        if (nf == 3) {
            const long long cap = std::max<long long>(64, fbmax / (2 * 3));
            unsigned gx = 1; int jt = 0;
            for (int i = 0; i < 3; ++i) {
                slot.op[q + i].f = f[i];
                gx = std::max(gx, filter_grid_cap(filter_items(f[i]), cap));
                jt = std::max(jt, f[i].jt);
            }
            plan.push_back(mk_launch(TK_FILTER_GROUP, q, gx, 1, smem_of(jt)));
            q += 3;
        } else {
            for (int i = 0; i < nf; ++i) {
                slot.op[q].f = f[i];
                plan.push_back(mk_launch(TK_FILTER, q, filter_grid_cap(filter_items(f[i]), fbmax), 1, smem_of(f[i].jt)));
                plan.back().list = f[i].list;
                ++q;
            }
        }
        if (have_flow && have_build) {
            OpArgs &o = slot.op[q];
            o.p = flow; o.f = build;
            const long long cap = std::max<long long>(64, fbmax / 2);   // (blocks of a build riding in a flow launch: / 1 ... / 8 measured alike)
            o.np = std::max(8, flow.nblk);
            o.n0 = (int)std::max(8u, filter_grid_cap(filter_items(build), cap));
            if (head) { o.ps = ops[at + 1].ps; }
            else { o.ps.run_mail = nullptr; }
            plan.push_back(mk_launch(head ? TK_HFLOW_BUILD : TK_FLOW_BUILD, q, (unsigned)(o.np + o.n0), 1,
                                     head ? smem_head(build.jt) : smem_of(build.jt)));
            // a RUN batch begins with a resident run (kt_run reads the head's and the flow pass's arguments from this entry,
            // the trace from the step launch's, which is the next one): candidate records on both buffers, the moving
            // cloud read as it came
            if (head && pre && o.ps.run_mail && flow.cand && flow.cand_b && flow.kept_packed == 1 && flow.tf_a == 0 && flow.tf_b == 1 &&
                flow.weight == 0 && 4 * flow.nblk <= PROC_WAVES) {
                pre->push_back(mk_launch(TK_RUN, q | ((q + 1) << 4), 1u + (unsigned)std::max(8, std::min((int)RUN_G, o.ps.run_g_max)), 1));   // (no more blocks than the run may use: a block that is not needed still takes a compute unit until it knows)
                pre->push_back(mk_launch(TK_RUN, q | ((q + 1) << 4), 1u + RUN_G_SMALL, 1));   // (for a narrow record: launch_batch picks one)
                if (side && o.ps.side_mirror && o.f.st2) {
                    side->push_back(mk_launch(TK_SIDE_FILTER, q, (unsigned)o.n0, 1, smem_of(build.jt)));
                    side->push_back(mk_launch(TK_SIDE_RECORD, q, (unsigned)o.np, 1));
                }
            }
            ++q;
        } else if (have_flow) {
            slot.op[q].p = flow;
            plan.push_back(mk_launch(flow.weight == 1 ? TK_FLOW_MATLAB : (flow.need_d2 ? TK_FLOW_D2 : TK_FLOW), q, (unsigned)std::max(1, flow.nblk), 1));
            ++q;
        }
    }
    if (ns == 2) {
        slot.op[q].p = self[0]; slot.op[q + 1].p = self[1];
        plan.push_back(mk_launch(TK_SELF2, q, (unsigned)std::max(self[0].nblk, self[1].nblk), 1));
        q += 2;
    } else if (ns == 1) {
        slot.op[q].p = self[0];
        plan.push_back(mk_launch(TK_SELF, q, (unsigned)self[0].nblk, 1));
        ++q;
    }
    for (; at < ops.size(); ++at) {   // the rest, one launch each
        if (q >= MAX_OPS) return false;
        const RecOp &op = ops[at];
        OpArgs &o = slot.op[q];
        if (op.kind == RecOp::POST_FLOW) { o.pf = op.pf; plan.push_back(mk_launch(TK_POST_FLOW, q, 1, 1)); }
        else if (op.kind == RecOp::POST_STEP) {
            o.ps = op.ps;
            if (have_flow && flow.async_xy) o.ps.ck_nblk[LIST_XY] = 0;   // (no record without head mode, see above)
            for (int w = 0; w < 2; ++w)
                if (ns_all > w && self_async[w]) o.ps.ck_nblk[LIST_XX + w] = 0;
            plan.push_back(mk_launch(TK_POST_STEP, q, 1, 1));
        }
        else if (op.kind == RecOp::PROCESS && op.mode == kProcStepTwist) {
            o.p = op.p;
            plan.push_back(mk_launch(head ? TK_HSTEP_TWIST : TK_STEP_TWIST, q, (unsigned)std::max(8, std::max(32, op.p.nblk) / 4), 1));
            if (head) { ++q; break; }   // (the post-step launch that follows is the head of the next flow launch)
        } else if (op.kind == RecOp::PROCESS && op.mode == PROC_STEP) {
            o.p = op.p;
            plan.push_back(mk_launch(TK_STEP, q, (unsigned)std::max(1, op.p.nblk), 1));
        } else return false;
        ++q;
    }
    *head_mode = head;
    return q <= MAX_OPS;
}

// A fused group: one launch per recorded launch, blockIdx.z = slot.  `ops[i]` = member i's
// recorded iteration (all of the same shape), `slots[i]` its slot image; geometry = what
// serves every member (zdim slots share the launch).
bool plan_fused(const std::vector<const std::vector<RecOp> *> &ops, const std::vector<Slot *> &slots, int zdim,
                std::vector<TLaunch> &plan)
{
    plan.clear();
    if (ops.empty()) return true;
    const size_t nq = ops[0]->size();
    if (nq > (size_t)MAX_OPS) return false;
    for (const auto *o : ops)
        if (o->size() != nq) return false;
    const long long fbmax = filter_blocks_cap();
    // acvo, synchronous lists: filter xy, flow, filter xx, self, filter yy, self are recorded in the
    // reference's order; the three filters are independent of the passes, so the slots hold them
    // first -- three filters (one launch, blockIdx.y = list), flow, two self passes (one launch) --
    // 6 launches per iteration instead of 9
    std::vector<size_t> perm(nq);
    for (size_t q = 0; q < nq; ++q) perm[q] = q;
    {
        const std::vector<RecOp> &r = *ops[0];
        auto is_f = [&](size_t q) { return q < nq && r[q].kind == RecOp::FILTER && r[q].mode != kFilterAhead; };
        auto is_p = [&](size_t q, int mode) { return q < nq && r[q].kind == RecOp::PROCESS && r[q].mode == mode; };
        if (is_f(0) && is_p(1, PROC_FLOW) && is_f(2) && is_p(3, PROC_SELF) && is_f(4) && is_p(5, PROC_SELF)) {
            const size_t order[6] = {0, 2, 4, 1, 3, 5};
            for (size_t q = 0; q < 6; ++q) perm[q] = order[q];
        }
    }
    for (size_t qs = 0; qs < nq; ++qs) {
        const size_t q = perm[qs];   // recorded op q lives in slot entry qs
        const RecOp &first = (*ops[0])[q];
        for (const auto *o : ops)
            if ((*o)[q].kind != first.kind || (*o)[q].mode != first.mode) return false;
        unsigned gx = 1, smem = 0;
        int kernel = -1, np = 8;
        unsigned nfb = 8;
        for (size_t i = 0; i < ops.size(); ++i) {
            const RecOp &op = (*ops[i])[q];
            OpArgs &o = slots[i]->op[qs];
            switch (op.kind) {
            case RecOp::FILTER: {
                o.f = op.f;
                kernel = TK_FILTER;
                // (blocks per slot: the blocks of a slot that has nothing to build leave on the table's build mask, before any
                // load of their own -- kt_filter --, so a build can have 256 of them where 64 used to be the price of 60 idle launches)
                const long long cap = std::max<long long>(kFusedFilterBlocks, fbmax / (2 * zdim));
                gx = std::max(gx, filter_grid_cap(filter_items(op.f), cap));
                smem = std::max(smem, (unsigned)filter_smem_bytes(op.f.jt));
                break;
            }
            case RecOp::PROCESS:
                o.p = op.p;
                if (op.mode == kProcStepTwist) {
                    kernel = TK_STEP_TWIST;
                    gx = std::max(gx, (unsigned)(std::max(32, op.p.nblk) / 4));
                } else if (op.mode == kFlowBuild) {
                    kernel = TK_FLOW_BUILD;
                    o.f = op.f;
                    const long long cap = std::max<long long>(64, fbmax / (2 * zdim));
                    np = std::max(np, op.p.nblk);
                    nfb = std::max(nfb, filter_grid_cap(filter_items(op.f), cap));
                    smem = std::max(smem, (unsigned)filter_smem_bytes(op.f.jt));
                } else {
                    kernel = op.mode == PROC_FLOW ? (op.p.weight == 1 ? TK_FLOW_MATLAB : TK_FLOW)
                                                  : (op.mode == PROC_STEP ? TK_STEP : TK_SELF);
                    gx = std::max(gx, (unsigned)op.p.nblk);
                }
                break;
            case RecOp::POST_FLOW: o.pf = op.pf; kernel = TK_POST_FLOW; break;
            case RecOp::POST_STEP: o.ps = op.ps; kernel = TK_POST_STEP; break;
            }
        }
        if (kernel == TK_FLOW) {   // (TK_FLOW is built without the sum of a d2, which the cvo loop never reads)
            bool d2 = false;
            for (size_t i = 0; i < ops.size(); ++i) d2 = d2 || (*ops[i])[q].p.need_d2 != 0;
            if (d2) kernel = TK_FLOW_D2;
        }
        if (kernel == TK_FLOW_BUILD) {
            gx = (unsigned)np + nfb;
            for (Slot *sl : slots) { sl->op[qs].np = np; sl->op[qs].n0 = (int)nfb; }
        }
        plan.push_back(mk_launch(kernel, (int)qs, gx, (unsigned)zdim, smem));
        if (kernel == TK_FILTER) plan.back().list = first.f.list;
    }
    // three filters / two self passes in a row become one launch each
    std::vector<TLaunch> merged;
    for (size_t i = 0; i < plan.size(); ++i) {
        auto run_of = [&](int kernel, size_t n) {
            if (i + n > plan.size()) return false;
            for (size_t k = 0; k < n; ++k)
                if (plan[i + k].kernel != kernel || plan[i + k].q != plan[i].q + (int)k) return false;
            return true;
        };
        if (run_of(TK_FILTER, 3)) {
            TLaunch l = plan[i];
            l.kernel = TK_FILTER_GROUP;
            for (size_t k = 1; k < 3; ++k) { l.gx = std::max(l.gx, plan[i + k].gx); l.smem = std::max(l.smem, plan[i + k].smem); }
            merged.push_back(l);
            i += 2;
        } else if (run_of(TK_SELF, 2)) {
            TLaunch l = plan[i];
            l.kernel = TK_SELF2;
            l.gx = std::max(l.gx, plan[i + 1].gx);
            merged.push_back(l);
            i += 1;
        } else {
            merged.push_back(plan[i]);
        }
    }
    plan.swap(merged);
    return true;
}

// The post-step launches of a plan that has filter launches of its own keep their slot's bits of the table's build masks
// (TableBuf, kt_filter): where the masks live, which bit is this slot's.
void set_build_masks(Slot &slot, const std::vector<TLaunch> &plan, uint32_t *masks, int z)
{
    bool filters = false;
    for (const TLaunch &l : plan) filters = filters || l.kernel == TK_FILTER || l.kernel == TK_FILTER_GROUP;
    for (const TLaunch &l : plan)
        if (l.kernel == TK_POST_STEP) {
            slot.op[l.q].ps.build_mask = filters ? masks : nullptr;
            slot.op[l.q].ps.slot_bit = 1u << z;
        }
}

bool same_plan(const std::vector<TLaunch> &a, const std::vector<TLaunch> &b)
{
    return a.size() == b.size() && (a.empty() || std::memcmp(a.data(), b.data(), a.size() * sizeof(TLaunch)) == 0);
}

void launch_plan_eager(const Slot *tab, const std::vector<TLaunch> &plan, int iterations, hipStream_t s, const std::vector<TLaunch> *pre)
{
    if (pre)
        for (const TLaunch &l : *pre) launch_table(tab, l, s, nullptr, nullptr, 0);
    for (int k = 0; k < iterations; ++k)
        for (const TLaunch &l : plan) launch_table(tab, l, s, nullptr, nullptr, k & 1);
}

// kBatch iterations of `plan` on table `tab`: through a cached graph when allowed, else eagerly.
int run_plan(const Slot *tab, PlanCache &cache, const std::vector<TLaunch> &plan, hipStream_t s, bool use_graph,
             int iterations, const std::vector<TLaunch> *pre)
{
    static const std::vector<TLaunch> none;
    const std::vector<TLaunch> &front = pre ? *pre : none;
    if (!use_graph || cache.fails >= 64) {
        launch_plan_eager(tab, plan, iterations, s, pre);
        return hipGetLastError() == hipSuccess ? CVO_HIP_OK : CVO_HIP_ERR_HIP;
    }
    PlanGraph *hit = nullptr;
    for (auto &g : cache.graphs)
        if (g.iterations == iterations && same_plan(g.plan, plan) && same_plan(g.pre, front)) { hit = &g; break; }
    if (hit) ++cache.hits;
    if (!hit) {
        // The capture window needs the library's lock exclusively (cvo_lock.h).  Not getting it within its
        // millisecond -- other host threads are inside their own entry points -- is neither a capture nor a
        // failed one: this batch goes out eagerly, nothing is counted, the next batch tries again.
        cvo_lock::Capture alone;   // (held: no other thread of this library is inside the runtime)
        if (!alone.ok) {
            launch_plan_eager(tab, plan, iterations, s, pre);
            return hipGetLastError() == hipSuccess ? CVO_HIP_OK : CVO_HIP_ERR_HIP;
        }
        ++cache.captures;
        if (cache.graphs.size() >= 12) {   // evict the least recently used entry
            size_t lru = 0;
            for (size_t i = 1; i < cache.graphs.size(); ++i)
                if (cache.graphs[i].stamp < cache.graphs[lru].stamp) lru = i;
            if (cache.graphs[lru].exec) (void)hipGraphExecDestroy(cache.graphs[lru].exec);
            if (cache.graphs[lru].graph) (void)hipGraphDestroy(cache.graphs[lru].graph);
            cache.graphs.erase(cache.graphs.begin() + lru);
        }
        PlanGraph g;
        g.plan = plan;
        g.pre = front;
        g.iterations = iterations;
        // A capture can be spoilt from outside (another thread's HIP work: cvo_lock.h).  Nothing
        // has been launched then: the batch goes out eagerly and the next one tries again.
        hipError_t e = hipErrorUnknown;
        if (hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed) == hipSuccess) {
            launch_plan_eager(tab, plan, iterations, s, pre);
            e = hipStreamEndCapture(s, &g.graph);
        }
        if (e != hipSuccess || !g.graph || hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0) != hipSuccess) {
            if (g.graph) (void)hipGraphDestroy(g.graph);
            (void)hipGetLastError();
            ++cache.fails;
            launch_plan_eager(tab, plan, iterations, s, pre);
            return hipGetLastError() == hipSuccess ? CVO_HIP_OK : CVO_HIP_ERR_HIP;
        }
        cache.fails = 0;
        cache.graphs.push_back(g);
        hit = &cache.graphs.back();
    }
    hit->stamp = ++cache.clock;
    return hipGraphLaunch(hit->exec, s) == hipSuccess ? CVO_HIP_OK : CVO_HIP_ERR_HIP;
}

hipStream_t loop_stream(const cvo_hip_ctx *ctx) { return ctx->loop_stream ? ctx->loop_stream : ctx->stream; }

// Record the launches of ONE iteration of this context's align() (nothing is launched).
int record_iteration(cvo_hip_ctx *ctx, std::vector<RecOp> &ops, int trace_cap)
{
    ops.clear();
    ctx->rec = &ops;
    ctx->plan_recording = true;
    for (int l = 0; l < 3; ++l) ctx->ck_nblk[l] = 0;   // (set again by the passes of this plan that keep a candidate list)
    const int rc = enqueue_iterations(ctx, 1, -1, trace_cap);
    ctx->plan_recording = false;
    ctx->rec = nullptr;
    return rc;
}

// The registration on its own table: (re)make its plan and slot, send the slot if it changed.
// Called when an align() begins and when it resumes after a list grew (the arguments only
// change then: buffers, sizes, parameters, trace).
int prepare_lone_plan(cvo_hip_ctx *ctx, int trace_cap)
{
    if (ctx->table.init(1, loop_stream(ctx)) != 0) return fail(ctx, CVO_HIP_ERR_NOMEM, "argument table allocation failed");
    std::vector<RecOp> ops;
    int rc = record_iteration(ctx, ops, trace_cap);
    if (rc) return rc;
    Slot slot;
    if (!plan_lone(ops, slot, ctx->plan, ctx->allow_head, &ctx->head_mode, &ctx->plan_pre, &ctx->plan_side))
        return fail(ctx, CVO_HIP_ERR_INVALID, "launch plan does not fit the argument table");
    if (!ctx->plan_side.empty())
        for (TLaunch &l : ctx->plan_pre) l.list = 1;   // (launch_table: the run kernels that carry the side builds)
    if (!ctx->plan_pre.empty() && run_allow_lds() != hipSuccess) {
        // (resident runs are an optimisation: a device or partition that refuses kt_run its LDS gets the plain head-mode plan -- the
        // plan's classic launches are complete without the run in front of them)
        (void)hipGetLastError();
        ctx->plan_pre.clear();
        ctx->plan_side.clear();
        ctx->allow_run = false;
    }
    if (ctx->opt.comm_debug)
        fprintf(stderr, "[cvo_hip] lone plan: %zu launches, head mode %d, pre %zu, side %zu, side mirror %p, side stream %p\n", ctx->plan.size(), (int)ctx->head_mode,
                ctx->plan_pre.size(), ctx->plan_side.size(), (void *)ctx->side_mirror, (void *)ctx->side_stream);
    set_build_masks(slot, ctx->plan, ctx->table.masks(), 0);
    ctx->plan_has_final_mirror = false;
    for (const TLaunch &l : ctx->plan)   // (the launches whose heads publish: the post-step launch, or the head-mode flow launch)
        if ((l.kernel == TK_POST_STEP || l.kernel == TK_HFLOW_BUILD || l.kernel == TK_HFLOW_BUILD6) && slot.op[l.q].ps.final_mirror) ctx->plan_has_final_mirror = true;
    if (ctx->table.sync(&slot, loop_stream(ctx)) != 0) return fail(ctx, CVO_HIP_ERR_HIP, "argument table upload failed");
    return CVO_HIP_OK;
}

// Launch one batch of kBatch iterations: through the context's table (graph or eager table
// launches); profiling and the stream-level all-reduces (RCCL, caller's hook) keep the
// classic by-value launches -- they need their own launches / host calls in between.
// with_run: a RUN batch -- the plan's resident run, then kRunBatchSlots classic slots (job_pump asks for it when the plan has a
// run and the registration is narrow enough)
int launch_batch(cvo_hip_ctx *ctx, int tag0, int trace_cap, bool with_run, int slots, bool small_run, bool two_runs)
{
    if (ctx->profiling || host_reduce(ctx)) {
        const int rc = enqueue_iterations(ctx, kBatch, tag0, trace_cap);
        if (!rc) ctx->warm = true;
        return rc;
    }
    // A head-mode plan (one registration with its launches to itself: two launches per iteration, or most iterations inside
    // resident runs) launches eagerly: the host paces its batches on the slot mirror anyway, and every boundary between two
    // captured batches cost ~9 us of idle stream (cvo 10k x 10k: three of them, 978 -> 1 003 registrations/s, 3k x 3k 1 190 -> 1 232;
    // acvo 10k 615 -> 635, 3k 815 -> 849: profiles/r05_ab.txt 14, 16).  CVO_HIP_RUN_GRAPHS=1 (read when a context is created) brings the captured batches back.
    const bool graphs = ctx->use_graphs && (!ctx->head_mode || ctx->head_graphs);
    // (the run's launch: all blocks, or RUN_G_SMALL + 1 for a record that needs no more -- every block of the launch must have
    // started before the run begins, kt_run's entry hand-shake)
    std::vector<TLaunch> front;
    if (with_run) front.push_back(ctx->plan_pre[(small_run && ctx->plan_pre.size() > 1) ? 1 : 0]);
    // (with side builds a run ends when its next list stands ready: the second launch enters on it at once, no word from the host in between)
    if (with_run && two_runs && !ctx->plan_side.empty()) front.push_back(front[0]);
    const int rc = with_run ? run_plan(ctx->table.dev, ctx->plans, ctx->plan, loop_stream(ctx), graphs, kRunBatchSlots, &front)
                            : run_plan(ctx->table.dev, ctx->plans, ctx->plan, loop_stream(ctx), graphs, slots);
    if (rc) return fail(ctx, rc, "launching a batch of iterations failed");
    return CVO_HIP_OK;
}

int zero_counters(cvo_hip_ctx *ctx)
{
    HIP_TRY(ctx, hipMemsetAsync(reinterpret_cast<char *>(ctx->st) + offsetof(DevState, ovf), 0,
                                sizeof(uint32_t) * 16, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(reinterpret_cast<char *>(ctx->st) + offsetof(DevState, sub), 0,
                                sizeof(uint32_t) * LIST_N * NSUB, ctx->stream));
    return CVO_HIP_OK;
}

// host -> device copy of a few DevState fields through pinned staging slot 0
int push_state_fields(cvo_hip_ctx *ctx, size_t off, size_t bytes)
{
    HIP_TRY(ctx, hipMemcpyAsync(reinterpret_cast<char *>(ctx->st) + off,
                                reinterpret_cast<char *>(&ctx->st_host[kPollSlots]) + off, bytes,
                                hipMemcpyHostToDevice, ctx->stream));
    return CVO_HIP_OK;
}

int fetch_red(cvo_hip_ctx *ctx, int off, int count, double *out)
{
    DevState *h = &ctx->st_host[0];
    HIP_TRY(ctx, hipMemcpyAsync(h->red + off,
                                reinterpret_cast<char *>(ctx->st) + offsetof(DevState, red) +
                                    off * sizeof(double),
                                count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    std::memcpy(out, h->red + off, count * sizeof(double));
    return CVO_HIP_OK;
}


}   // namespace cvo_impl

