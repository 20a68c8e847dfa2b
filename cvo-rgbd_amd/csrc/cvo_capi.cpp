// cvo_capi.cpp -- implementation of include/cvo_hip.h: context, device memory,
// kernel orchestration and the align() loop of cvo::cvo / acvo::acvo
// (ref src/cvo.cpp:361-420, src/adaptive_cvo.cpp:490-555) on one MI355X.
//
// The loop is device-resident: registration state (R, T, ell, twist, partial
// sums, trace) lives in HBM, the O(1) maths between the sweeps runs in the
// k_post_* kernels, and the host only enqueues batches of iterations and
// polls a `done` word through pinned memory.  After convergence the remaining
// queued kernels return at once.
//
// There is no CPU fallback in this library: every entry point that computes
// needs a gfx950 device and fails with CVO_HIP_ERR_NODEVICE / _HIP otherwise.
#include "cvo_internal.h"

using namespace cvo_dev;
using namespace cvo_impl;

namespace cvo_impl {

int fail(cvo_hip_ctx *ctx, int code, const char *msg)
{
    if (ctx) ctx->err = msg;
    return code;
}

// ---- options (cvo_hip_set_option): one table of keys; the environment variables of earlier rounds are the DEFAULTS of the
// same options, read once per context at cvo_hip_create (env_defaults below holds the library's only getenv of this file).
static std::atomic<bool> g_engine_debug{false};
bool engine_debug_on() { return g_engine_debug.load(std::memory_order_relaxed); }

namespace {
struct OptDef {
    const char *key;
    const char *env;      // environment variable that sets its default (nullptr: none)
    int env_kind;         // 1: a flag -- present means env_value; 2: a number -- the variable's text
    double env_value;
};
const OptDef kOptions[] = {
    {"graph_capture", "CVO_HIP_GRAPH", 1, 1.0},           {"no_graph", "CVO_HIP_NO_GRAPH", 1, 1.0},
    {"head_graphs", "CVO_HIP_RUN_GRAPHS", 1, 1.0},        {"head_mode", "CVO_HIP_NO_HEAD", 1, 0.0},
    {"resident_runs", "CVO_HIP_NO_RUN", 1, 0.0},          {"run_solvers_max", "CVO_HIP_RUN_G_MAX", 2, 0.0},
    {"run_candidates_max", "CVO_HIP_RUN_CAND", 2, 0.0},   {"run_timeout_ms", nullptr, 0, 0.0},
    {"run_fault", nullptr, 0, 0.0},                       {"merged_launches", "CVO_HIP_NO_MERGE", 1, 0.0},
    {"async_builds", "CVO_HIP_NO_ASYNC", 1, 0.0},         {"list_pass_blocks", "CVO_HIP_PROC_BLOCKS", 2, 0.0},
    {"post_debug", "CVO_HIP_POST_DEBUG", 1, 1.0},         {"mailbox_timeout_s", "CVO_HIP_MAILBOX_TIMEOUT_S", 2, 0.0},
    {"candidate_records", "CVO_HIP_NO_CAND", 1, 0.0},     {"sync_upload", "CVO_HIP_SYNC_UPLOAD", 1, 1.0},
    {"engine_debug", "CVO_HIP_ENGINE_DEBUG", 1, 1.0},     {"one_launch_hand_over", "CVO_HIP_NO_CLOUD_ONE", 1, 0.0},
    {"small_calls_alone", "CVO_HIP_NO_ALONE", 1, 0.0},    {"fused_groups", "CVO_HIP_NO_FUSE", 1, 0.0},
    {"engines", "CVO_HIP_ENGINES_FORCE", 2, 0.0},         {"list_init", "CVO_HIP_LIST_INIT", 2, 0.0},
    {"kept_pack", "CVO_HIP_NO_PACK", 1, 0.0},             {"list_margin", "CVO_HIP_LIST_MARGIN", 2, 0.0},
    {"final_mirror", "CVO_HIP_NO_FINAL_MIRROR", 1, 0.0},  {"twist_on_shared_gpu", "CVO_HIP_TWIST_ON_SHARED_GPU", 1, 1.0},
    {"comm_debug", "CVO_HIP_COMM_DEBUG", 1, 1.0},         {"wait_policy", "CVO_HIP_WAIT_POLICY", 2, 0.0},
    {"acvo_runs", "CVO_HIP_NO_ACVO_RUN", 1, 0.0},         {"tail_alone", "CVO_HIP_TAIL_ALONE", 2, 0.0},         {"side_builds", "CVO_HIP_SIDE", 1, 1.0},
    {"run_build_at", "CVO_HIP_RUN_BUILD_AT", 2, 0.0},         {"alone_max", "CVO_HIP_ALONE_MAX", 2, 0.0},
    {"narrow_merge", "CVO_HIP_NARROW_MERGE", 1, 1.0},     {"narrow_blocks", "CVO_HIP_NARROW_BLOCKS", 2, 0.0},
    {"engine_crowd", "CVO_HIP_ENGINE_CROWD", 2, 0.0},     {"engine_merge_max", "CVO_HIP_ENGINE_MERGE_MAX", 2, 0.0},
};
void env_defaults(cvo_hip_ctx *ctx)
{
    for (const OptDef &d : kOptions) {
        if (!d.env) continue;
        const char *e = getenv(d.env);
        if (!e) continue;
        (void)apply_option(ctx, d.key, d.env_kind == 1 ? d.env_value : atof(e));
    }
}
}   // namespace

// (value: 0 / 1 for switches; returns CVO_HIP_ERR_INVALID for an unknown key or a value out of range)
int apply_option(cvo_hip_ctx *ctx, const char *key, double v)
{
    auto is = [&](const char *k) { return std::strcmp(key, k) == 0; };
    const bool on = v != 0.0;
    CtxOptions &o = ctx->opt;
    if (is("graph_capture")) { ctx->use_graphs = on && !o.no_graph; if (!ctx->use_graphs) drop_graphs(ctx); }
    else if (is("no_graph")) { o.no_graph = on; if (on) { ctx->use_graphs = false; drop_graphs(ctx); } }
    else if (is("head_graphs")) ctx->head_graphs = on;
    else if (is("head_mode")) ctx->allow_head = on;
    else if (is("resident_runs")) ctx->allow_run = on;
    else if (is("run_solvers_max")) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device) != hipSuccess || cus <= 0) cus = RUN_G + 8;
        ctx->run_g_max = v > 0.0 ? std::max(8, std::min((int)RUN_G, (int)v)) : std::max(8, std::min((int)RUN_G, cus - 8));
    }
    else if (is("run_candidates_max")) { if (v < 0.0) return CVO_HIP_ERR_INVALID; o.run_cand = (int)v; }
    else if (is("run_timeout_ms")) { if (v < 0.0 || v > 6.0e4) return CVO_HIP_ERR_INVALID; o.run_timeout_ms = v; }
    else if (is("run_fault")) { if (v < 0.0) return CVO_HIP_ERR_INVALID; o.run_fault = (int)v; }
    else if (is("merged_launches")) ctx->allow_merge = on;
    else if (is("async_builds")) ctx->allow_async = ctx->allow_async_self = on;
    else if (is("list_pass_blocks")) {   // list-kernel blocks of a lone registration (0: by the clouds)
        const int b = (int)v;
        if (b == 0) { ctx->proc_blocks_forced = false; ctx->proc_blocks = ctx->proc_blocks_default = PROC_BLOCKS; }
        else if (b == 64 || b == 128 || b == 256 || b == 512 || b == 1024) { ctx->proc_blocks = ctx->proc_blocks_default = b; ctx->proc_blocks_forced = true; }
        else return CVO_HIP_ERR_INVALID;
    }
    else if (is("post_debug")) o.post_debug = on;
    else if (is("mailbox_timeout_s")) { if (!(v > 0.0)) return CVO_HIP_ERR_INVALID; o.mailbox_timeout_s = v; }
    else if (is("candidate_records")) o.no_cand = !on;
    else if (is("sync_upload")) o.sync_upload = on;
    else if (is("engine_debug")) { o.engine_debug = on; if (on) g_engine_debug.store(true, std::memory_order_relaxed); }
    else if (is("one_launch_hand_over")) o.no_cloud_one = !on;
    else if (is("small_calls_alone")) o.no_alone = !on;
    else if (is("fused_groups")) o.no_fuse = !on;
    else if (is("engines")) { if (v < 0.0 || v > 8.0) return CVO_HIP_ERR_INVALID; o.engines_force = (int)v; }
    else if (is("list_init")) { if (v < 0.0) return CVO_HIP_ERR_INVALID; o.list_init = v; }
    else if (is("kept_pack")) o.no_pack = !on;
    else if (is("list_margin")) { if (v > 4.0) return CVO_HIP_ERR_INVALID; o.list_margin = v < 0.0 ? -1.0f : (float)v; }
    else if (is("final_mirror")) o.no_final_mirror = !on;
    else if (is("twist_on_shared_gpu")) o.twist_on_shared_gpu = on;
    else if (is("comm_debug")) o.comm_debug = on;
    else if (is("wait_policy")) { if (v < 0.0 || v > 2.0) return CVO_HIP_ERR_INVALID; o.wait_policy = (int)v; }
    else if (is("acvo_runs")) o.no_acvo_run = !on;
    else if (is("tail_alone")) { if (v < 0.0 || v > 32.0) return CVO_HIP_ERR_INVALID; o.tail_alone = (int)v; }
    else if (is("run_restart")) o.no_restart = !on;
    else if (is("side_builds")) o.no_side_builds = !on;
    else if (is("run_build_at")) { if (!(v > 0.0 && v <= 1.0)) return CVO_HIP_ERR_INVALID; o.run_build_at = (float)v; }
    else if (is("alone_max")) { if (v < 0.0 || v > 64.0) return CVO_HIP_ERR_INVALID; o.alone_max = (int)v; }
    else if (is("narrow_merge")) o.narrow_merge = on;
    else if (is("narrow_blocks")) { const int b = (int)v; if (b != 0 && b != 32 && b != 64) return CVO_HIP_ERR_INVALID; o.narrow_blocks = b; }
    else if (is("engine_crowd")) { if (v < 0.0) return CVO_HIP_ERR_INVALID; o.engine_crowd = (int)std::min(v, 1.0e6); }
    else if (is("engine_merge_max")) { if (v < 0.0 || v > 32.0) return CVO_HIP_ERR_INVALID; o.engine_merge_max = (int)v; }
    else return CVO_HIP_ERR_INVALID;
    return CVO_HIP_OK;
}

// Parameters the kernels can work with: a known mode, finite values, positive kernel scales
// and thresholds (log of a non-positive quotient would make NaN radii and NaN twists that
// only surface as "align loop ended without a verdict").  Returns nullptr if fine.
const char *params_problem(const cvo_hip_params &p)
{
    if (p.mode != CVO_HIP_MODE_CVO && p.mode != CVO_HIP_MODE_ACVO)
        return "params.mode must be CVO_HIP_MODE_CVO or CVO_HIP_MODE_ACVO (MATLAB: default_params() returns mode CVO)";
    if (p.max_iter < 0) return "params.max_iter < 0";
    const float pos[] = {p.ell_init, p.sigma, p.sp_thres, p.c, p.d, p.c_ell, p.c_sigma};
    const char *pos_name[] = {"ell_init", "sigma", "sp_thres", "c", "d", "c_ell", "c_sigma"};
    static thread_local char msg[96];
    for (int i = 0; i < 7; ++i)
        if (!(pos[i] > 0.0f) || !std::isfinite(pos[i])) {
            snprintf(msg, sizeof(msg), "params.%s must be positive and finite", pos_name[i]);
            return msg;
        }
    if (p.mode == CVO_HIP_MODE_ACVO && (!(p.c_sp_thres > 0.0f) || !std::isfinite(p.c_sp_thres)))
        return "params.c_sp_thres must be positive and finite";
    if (p.mode == CVO_HIP_MODE_ACVO && (!(p.ell_max_init > 0.0f) || !std::isfinite(p.ell_max_init) ||
                                        !(p.ell_min >= 0.0f) || !std::isfinite(p.dl_step)))
        return "params.ell_max_init / ell_min / dl_step out of range";
    const float fin[] = {p.min_step, p.eps, p.eps_2, p.color_scale, p.ell_min};
    for (float v : fin)
        if (!std::isfinite(v)) return "params: min_step, eps, eps_2, color_scale, ell_min must be finite";
    if (p.color_scale < 0.0f) return "params.color_scale < 0";
    return nullptr;
}

DevParams make_dev_params(const cvo_hip_params &p)
{
    DevParams d{};
    d.mode = p.mode;
    d.max_iter = p.max_iter;
    d.ell_init = p.ell_init;
    d.ell_min = p.ell_min;
    d.ell_max_init = p.ell_max_init;
    d.sp = p.sp_thres;
    d.c_sp = (p.mode == CVO_HIP_MODE_ACVO) ? p.c_sp_thres : p.sp_thres;
    d.c = p.c;
    d.d = p.d;
    d.c_ell = p.c_ell;
    d.min_step = p.min_step;
    d.eps = p.eps;
    d.eps_2 = p.eps_2;
    const float s2 = p.sigma * p.sigma;
    const float cs2 = p.c_sigma * p.c_sigma;
    // `log(sp_thres/s2)` is the float overload in the reference (ref cvo.cpp:102)
    d.log_sp_s2 = (float)std::log((double)(p.sp_thres / s2));
    d.tau_c = (float)(-2.0 * p.c_ell * p.c_ell *
                      (double)(float)std::log((double)(d.c_sp / p.c_sigma / p.c_sigma)));
    d.s2_d = (double)s2;
    d.cs2_d = (double)cs2;
    d.dl_step = p.dl_step;
    d.color_scale = p.color_scale;
    // tile-list re-use (cvo_device.h plan_lists); CVO_HIP_LIST_MARGIN=0 rebuilds every iteration
    d.build_at = 0.7f;   // (measured 0.3 / 0.5 / 0.7: 1.80 / 1.77 / 1.73 ms per 10k x 10k registration)
    d.list_margin = 0.15f;   // (loop_params picks the margin of an align() by size; CVO_HIP_LIST_MARGIN overrides it there)
    return d;
}


}   // namespace cvo_impl

extern "C" {


const char *cvo_hip_error_string(int status)
{
    switch (status) {
    case CVO_HIP_OK: return "ok";
    case CVO_HIP_ERR_INVALID: return "invalid argument or call order";
    case CVO_HIP_ERR_HIP: return "HIP runtime error";
    case CVO_HIP_ERR_NOMEM: return "out of memory";
    case CVO_HIP_ERR_COMM: return "RCCL / all-reduce error";
    case CVO_HIP_ERR_NODEVICE: return "no usable HIP device";
    case CVO_HIP_ERR_RUN: return "a resident run timed out and the registration could not be redone";
    default: return "unknown status";
    }
}

const char *cvo_hip_last_error(const cvo_hip_ctx *ctx) { return ctx ? ctx->err.c_str() : ""; }

int cvo_hip_device_count(int *count)
{
    if (!count) return CVO_HIP_ERR_INVALID;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *count = n;
    return CVO_HIP_OK;
}

int cvo_hip_default_params(int mode, cvo_hip_params *p)
{
    if (!p || (mode != CVO_HIP_MODE_CVO && mode != CVO_HIP_MODE_ACVO && mode != CVO_HIP_MODE_MATLAB))
        return CVO_HIP_ERR_INVALID;
    std::memset(p, 0, sizeof(*p));
    p->mode = mode == CVO_HIP_MODE_MATLAB ? CVO_HIP_MODE_CVO : mode;
    p->max_iter = 2000;
    p->sigma = 0.1f;
    p->c = 7.0f;
    p->d = 7.0f;
    p->c_sigma = 1.0f;
    p->min_step = (float)(2 * 1.0e-1);
    p->eps = (float)(5 * 1.0e-5);
    p->eps_2 = (float)1.0e-5;
    if (mode == CVO_HIP_MODE_ACVO) {
        p->ell_init = 0.1f;
        p->ell_min = 0.0391f;
        p->ell_max_init = 0.15f;
        p->sp_thres = 8.315e-3f;
        p->c_sp_thres = 8.315e-3f;
        p->c_ell = 0.5f;
        p->dl_step = 0.3;
    } else {
        p->ell_init = 0.15f;
        p->ell_min = 0.0f;
        p->ell_max_init = 0.15f;
        p->sp_thres = 8e-3f;
        p->c_sp_thres = 8e-3f;
        p->c_ell = 200.0f;
        p->dl_step = 0.0;
    }
    if (mode == CVO_HIP_MODE_MATLAB) {   // ref rkhs_se3_registration.m:10-28
        p->sp_thres = 1e-3f;
        p->c_sp_thres = 1e-3f;
        p->eps = 5e-4f;
        p->eps_2 = 1e-4f;
        p->color_scale = 1e-5f;
    }
    return CVO_HIP_OK;
}

int cvo_hip_init_state(const cvo_hip_params *p, cvo_hip_state *s)
{
    if (!p || !s) return CVO_HIP_ERR_INVALID;
    std::memset(s, 0, sizeof(*s));
    s->R[0] = s->R[4] = s->R[8] = 1.0f;
    s->ell = p->ell_init;
    s->ell_max = p->ell_max_init;
    for (float *m : {s->transform, s->prev_transform, s->accum_transform})
        m[0] = m[5] = m[10] = m[15] = 1.0f;
    return CVO_HIP_OK;
}

int cvo_hip_create(int device, void *stream, const cvo_hip_params *p, cvo_hip_ctx **out)
{
    cvo_lock::Api api_guard;
    if (!p || !out) return CVO_HIP_ERR_INVALID;
    *out = nullptr;
    if (params_problem(*p)) return CVO_HIP_ERR_INVALID;   // (no context to hold the text: see cvo_hip_set_params)
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n)
        return CVO_HIP_ERR_NODEVICE;
    cvo_hip_ctx *ctx = new (std::nothrow) cvo_hip_ctx();
    if (!ctx) return CVO_HIP_ERR_NOMEM;
    ctx->device = device;
    ctx->prm = *p;
    ctx->dprm = make_dev_params(*p);
    auto bail = [&](int code) {
        cvo_hip_destroy(ctx);
        return code;
    };
    if (hipSetDevice(device) != hipSuccess) return bail(CVO_HIP_ERR_HIP);
    if (stream) {
        ctx->stream = (hipStream_t)stream;
    } else {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess)
            return bail(CVO_HIP_ERR_HIP);
        ctx->own_stream = true;
    }
    if (hipMalloc((void **)&ctx->st, sizeof(DevState)) != hipSuccess) return bail(CVO_HIP_ERR_NOMEM);
    if (hipMalloc((void **)&ctx->st2, sizeof(DevHead)) != hipSuccess) return bail(CVO_HIP_ERR_NOMEM);
    if (hipHostMalloc((void **)&ctx->st_host, (kPollSlots + 2) * sizeof(DevState),
                      hipHostMallocDefault) != hipSuccess)
        return bail(CVO_HIP_ERR_NOMEM);
    std::memset(ctx->st_host, 0, (kPollSlots + 1) * sizeof(DevState));
    if (hipMemsetAsync(ctx->st, 0, sizeof(DevState), ctx->stream) != hipSuccess)
        return bail(CVO_HIP_ERR_HIP);
    for (int i = 0; i < kPollSlots; ++i)
        if (hipEventCreateWithFlags(&ctx->poll_ev[i], hipEventDisableTiming) != hipSuccess)
            return bail(CVO_HIP_ERR_HIP);
    ctx->done_mirror = reinterpret_cast<int32_t *>(&ctx->st_host[kPollSlots + 1]);
    ctx->progress_mirror = ctx->done_mirror + 16;   // (its own cache line)
    ctx->final_mirror = reinterpret_cast<DevHead *>(reinterpret_cast<char *>(ctx->done_mirror) + 2048);
    static_assert(sizeof(DevState) >= 2048 + sizeof(DevHead), "the mirrors share a pinned DevState");
    ctx->run_mirror = ctx->done_mirror + 32;
    ctx->hint_mirror = ctx->done_mirror + 48;
    ctx->side_mirror = ctx->done_mirror + 64;
    *ctx->side_mirror = 0;
    *ctx->done_mirror = 0;
    *ctx->progress_mirror = 0;
    *ctx->run_mirror = 0;
    *ctx->hint_mirror = -1;
    {   // (a run's blocks must all be resident at once, one per compute unit: a partition with fewer units gets smaller runs)
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) cus = RUN_G + 8;
        ctx->run_g_max = std::max(8, std::min((int)RUN_G, cus - 8));
    }
    // Stream capture is a process-wide affair in this runtime (cvo_lock.h): the library's own
    // entry points keep out of each other's captures, but HIP work of OTHER code in the process
    // (torch on another thread, say) cannot be kept out and would fail with "previous error
    // during capture".  So batches are captured into hipGraphs by default only on a stream the
    // library created itself; with a caller-supplied stream the caller opts in
    // (cvo_hip_set_graph_capture / cvo_hip_set_option "graph_capture") once it knows no other thread of the
    // process uses HIP while an align() is being set up.  "no_graph" forbids captures.
    ctx->use_graphs = ctx->own_stream;
    // every other switch: cvo_hip_set_option; the defaults the environment names, read here and nowhere else
    env_defaults(ctx);
    if (ctx->opt.post_debug) {
        if (hipMalloc((void **)&ctx->post_dbg, 8 * sizeof(long long)) != hipSuccess) return bail(CVO_HIP_ERR_NOMEM);
        (void)hipMemset(ctx->post_dbg, 0, 8 * sizeof(long long));
    }
    *out = ctx;
    return CVO_HIP_OK;
}

int cvo_hip_destroy(cvo_hip_ctx *ctx)
{
    cvo_lock::Api api_guard;
    if (!ctx) return CVO_HIP_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (auto &ev : ctx->events) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
    for (int i = 0; i < kPollSlots; ++i)
        if (ctx->poll_ev[i]) (void)hipEventDestroy(ctx->poll_ev[i]);
    if (ctx->comm) cvo_comm_destroy(ctx->comm);
    for (void *q : ctx->mail_opened)
        if (q) (void)hipIpcCloseMemHandle(q);
    if (ctx->comm_table) (void)hipFree(ctx->comm_table);
    if (ctx->mailbox) (void)hipFree(ctx->mailbox);
    drop_graphs(ctx);
    ctx->table.destroy();
    if (ctx->post_dbg) {
        long long h[8];
        if (hipMemcpy(h, ctx->post_dbg, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess && h[0] > 0)
            fprintf(stderr, "[cvo_hip] post-step part over %lld launches, avg ticks: state load %.0f, reduce %.0f, "
                    "cubic %.0f, exp+update %.0f, prepare %.0f; head mode: block 0 of the flow launch start to end %.0f\n",
                    h[0], (double)h[1] / h[0], (double)h[2] / h[0],
                    (double)h[3] / h[0], (double)h[4] / h[0], (double)h[5] / h[0], (double)h[7] / h[0]);
        (void)hipFree(ctx->post_dbg);
    }
    for (void *p : {(void *)ctx->fixed.pos, (void *)ctx->fixed.feat, (void *)ctx->moving.pos,
                    (void *)ctx->moving.feat, (void *)ctx->fixed.seg, (void *)ctx->moving.seg,
                    (void *)ctx->scratch_a.pos, (void *)ctx->scratch_a.feat, (void *)ctx->scratch_a.seg,
                    (void *)ctx->scratch_b.pos, (void *)ctx->scratch_b.feat, (void *)ctx->scratch_b.seg, (void *)ctx->st, (void *)ctx->st2, ctx->part_flow.p, ctx->part_xx.p,
                    ctx->part_yy.p, ctx->part_step.p, ctx->run_mail.p, (void *)ctx->trace_dev, ctx->kept_cnt.p, ctx->pos_bt.p, ctx->cand[0].p, ctx->cand[1].p, ctx->cand[2].p, ctx->cand_xyb.p, ctx->cand_cnt_xyb.p, ctx->cand_sfb[0].p, ctx->cand_sfb[1].p, ctx->cand_cnt_sfb[0].p, ctx->cand_cnt_sfb[1].p, ctx->cand_cnt[0].p,
                    ctx->cand_cnt[1].p, ctx->cand_cnt[2].p})
        if (p) (void)hipFree(p);
    for (int l = 0; l < LIST_N; ++l) {
        if (ctx->lists[l].a.p) (void)hipFree(ctx->lists[l].a.p);
        if (ctx->lists[l].b.p) (void)hipFree(ctx->lists[l].b.p);
    }
    if (ctx->st_host) (void)hipHostFree(ctx->st_host);
    for (Cloud *c : {&ctx->fixed, &ctx->moving, &ctx->scratch_a, &ctx->scratch_b}) {
        if (c->pending && (c->wait_ev || c->ready_ev)) (void)hipEventSynchronize(c->wait_ev ? c->wait_ev : c->ready_ev);
        if (c->stage) (void)hipHostFree(c->stage);
        if (c->bbox_pin) (void)hipHostFree(c->bbox_pin);
        if (c->ready_ev) (void)hipEventDestroy(c->ready_ev);
    }
    for (DevBuf *b : {&ctx->raw_xyz, &ctx->raw_feat, &ctx->sort_keys[0], &ctx->sort_keys[1], &ctx->sort_idx[0],
                      &ctx->sort_idx[1], &ctx->sort_tmp})
        if (b->p) (void)hipFree(b->p);
    if (ctx->bbox_dev) (void)hipFree(ctx->bbox_dev);
    if (ctx->bbox_host) (void)hipHostFree(ctx->bbox_host);
    if (ctx->side_stream) { (void)hipStreamSynchronize(ctx->side_stream); (void)hipStreamDestroy(ctx->side_stream); }
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return CVO_HIP_OK;
}

int cvo_hip_set_params(cvo_hip_ctx *ctx, const cvo_hip_params *p)
{
    cvo_lock::Api api_guard;
    if (!ctx || !p) return CVO_HIP_ERR_INVALID;
    if (const char *why = params_problem(*p)) return fail(ctx, CVO_HIP_ERR_INVALID, why);
    drop_graphs(ctx);   // (captured batches hold the parameter block by value)
    ctx->prm = *p;
    ctx->dprm = make_dev_params(*p);
    return CVO_HIP_OK;
}

int cvo_hip_shard_range(int n, int rank, int world, int *lo, int *hi)
{
    if (!lo || !hi || world <= 0 || rank < 0 || rank >= world || n < 0) return CVO_HIP_ERR_INVALID;
    *lo = (int)((int64_t)n * rank / world);
    *hi = (int)((int64_t)n * (rank + 1) / world);
    return CVO_HIP_OK;
}

int cvo_hip_set_shard(cvo_hip_ctx *ctx, int row_lo, int row_hi, int srow_lo, int srow_hi)
{
    cvo_lock::Api api_guard;
    if (!ctx || row_lo < 0 || row_hi < row_lo || srow_lo < 0 || srow_hi < srow_lo)
        return CVO_HIP_ERR_INVALID;
    ctx->row_lo = row_lo; ctx->row_hi = row_hi;
    ctx->srow_lo = srow_lo; ctx->srow_hi = srow_hi;
    ctx->sharded = true;
    return CVO_HIP_OK;
}

int cvo_hip_comm_unique_id(void *id_bytes_128)
{
    if (!id_bytes_128) return CVO_HIP_ERR_INVALID;
    return cvo_comm_unique_id(id_bytes_128) == 0 ? CVO_HIP_OK : CVO_HIP_ERR_COMM;
}

int cvo_hip_comm_init(cvo_hip_ctx *ctx, const void *id_bytes_128, int rank, int world)
{
    cvo_lock::Api api_guard;
    if (!ctx || !id_bytes_128 || world <= 0 || rank < 0 || rank >= world)
        return CVO_HIP_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->comm) { cvo_comm_destroy(ctx->comm); ctx->comm = nullptr; }
    ctx->comm = cvo_comm_create(id_bytes_128, rank, world);
    if (!ctx->comm) return fail(ctx, CVO_HIP_ERR_COMM, "ncclCommInitRank failed");
    return CVO_HIP_OK;
}

int cvo_hip_mailbox_create(cvo_hip_ctx *ctx, int rank, int world, void *ipc_handle_64, void **dev_ptr)
{
    cvo_lock::Api api_guard;
    if (!ctx || world < 1 || world > MAX_WORLD || rank < 0 || rank >= world) return CVO_HIP_ERR_INVALID;
    static_assert(sizeof(hipIpcMemHandle_t) <= CVO_HIP_MAILBOX_HANDLE_BYTES, "IPC handle size");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->comm_table) { HIP_TRY(ctx, hipFree(ctx->comm_table)); ctx->comm_table = nullptr; }
    for (void *&q : ctx->mail_opened)
        if (q) { (void)hipIpcCloseMemHandle(q); q = nullptr; }
    if (!ctx->mailbox) {
        // peers write into it while this rank's kernel polls it: memory that no cache of this
        // device holds back -- uncached where the runtime offers it, else fine-grained, else plain
        // (polls and payload reads are system-scope loads either way)
        void *p = nullptr;
        if (hipExtMallocWithFlags(&p, sizeof(Mailbox), hipDeviceMallocUncached) != hipSuccess) {
            (void)hipGetLastError();
            p = nullptr;
            if (hipExtMallocWithFlags(&p, sizeof(Mailbox), hipDeviceMallocFinegrained) != hipSuccess) {
                (void)hipGetLastError();
                p = nullptr;
                if (hipMalloc(&p, sizeof(Mailbox)) != hipSuccess)
                    return fail(ctx, CVO_HIP_ERR_NOMEM, "hipMalloc(mailbox) failed");
            }
        }
        ctx->mailbox = (Mailbox *)p;
    }
    // sequence numbers restart with a new set of peers: empty the slots and the counter
    HIP_TRY(ctx, hipMemset(ctx->mailbox, 0, sizeof(Mailbox)));
    HIP_TRY(ctx, hipMemset(reinterpret_cast<char *>(ctx->st) + offsetof(DevState, mail_seq), 0, sizeof(unsigned long long)));
    {   // which GPU this rank's kernels run on, for the peers to read (cvo_hip_mailbox_connect)
        char bus[64] = {0};
        unsigned long long id = 1469598103934665603ull;
        if (hipDeviceGetPCIBusId(bus, sizeof(bus) - 1, ctx->device) != hipSuccess) { (void)hipGetLastError(); std::snprintf(bus, sizeof(bus), "device %d", ctx->device); }
        for (const char *c = bus; *c; ++c) id = (id ^ (unsigned char)*c) * 1099511628211ull;
        ctx->mail_dev_id = id | 1ull;
        HIP_TRY(ctx, hipMemcpy(&ctx->mailbox->owner_dev, &ctx->mail_dev_id, sizeof(id), hipMemcpyHostToDevice));
    }
    ctx->mail_shared_device = false;
    HIP_TRY(ctx, hipDeviceSynchronize());   // (null-stream fills: the context's stream does not wait for them by itself)
    ctx->mail_rank = rank;
    ctx->mail_world = world;
    ctx->mail_broken = false;
    if (ipc_handle_64) {
        std::memset(ipc_handle_64, 0, CVO_HIP_MAILBOX_HANDLE_BYTES);
        hipIpcMemHandle_t h;
        if (hipIpcGetMemHandle(&h, ctx->mailbox) != hipSuccess) {
            (void)hipGetLastError();
            return fail(ctx, CVO_HIP_ERR_COMM, "hipIpcGetMemHandle(mailbox) failed (HSA_ENABLE_IPC_MODE_LEGACY=0 set?)");
        }
        std::memcpy(ipc_handle_64, &h, sizeof(h));
    }
    if (dev_ptr) *dev_ptr = ctx->mailbox;
    return CVO_HIP_OK;
}

int cvo_hip_mailbox_connect(cvo_hip_ctx *ctx, const void *ipc_handles, void *const *dev_ptrs)
{
    cvo_lock::Api api_guard;
    if (!ctx || !ctx->mailbox || ctx->mail_world < 1 || (!ipc_handles && !dev_ptrs && ctx->mail_world > 1))
        return CVO_HIP_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    CommTable t{};
    t.rank = ctx->mail_rank;
    t.world = ctx->mail_world;
    const double secs = ctx->opt.mailbox_timeout_s;   // (cvo_hip_set_option "mailbox_timeout_s")
    t.timeout_ticks = (long long)(secs * 1.0e8);   // wall_clock64(): 100 MHz
    for (int r = 0; r < t.world; ++r) {
        if (r == t.rank) { t.peer[r] = ctx->mailbox; continue; }
        void *p = nullptr;
        if (dev_ptrs) {
            p = dev_ptrs[r];   // same process: a pointer this device can reach (peer access enabled by the owner of the devices)
            hipPointerAttribute_t at{};
            if (p && hipPointerGetAttributes(&at, p) == hipSuccess && at.device != ctx->device) {
                const hipError_t e = hipDeviceEnablePeerAccess(at.device, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
                    (void)hipGetLastError();
                    return fail(ctx, CVO_HIP_ERR_COMM, "hipDeviceEnablePeerAccess failed");
                }
            }
            (void)hipGetLastError();
        } else {
            hipIpcMemHandle_t h;
            std::memcpy(&h, reinterpret_cast<const char *>(ipc_handles) + (size_t)r * CVO_HIP_MAILBOX_HANDLE_BYTES, sizeof(h));
            if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
                (void)hipGetLastError();
                return fail(ctx, CVO_HIP_ERR_COMM, "hipIpcOpenMemHandle(peer mailbox) failed");
            }
            ctx->mail_opened[r] = p;
        }
        if (!p) return fail(ctx, CVO_HIP_ERR_INVALID, "null peer mailbox");
        t.peer[r] = (Mailbox *)p;
        {   // does the peer share this rank's GPU?  (then the exchange stays in the single-block post kernels: plan side)
            unsigned long long peer_id = 0;
            if (hipMemcpy(&peer_id, &((Mailbox *)p)->owner_dev, sizeof(peer_id), hipMemcpyDeviceToHost) != hipSuccess) {
                (void)hipGetLastError();
                peer_id = ctx->mail_dev_id;   // (unknown: assume the worst)
            }
            if (peer_id == ctx->mail_dev_id || peer_id == 0ull) ctx->mail_shared_device = true;
        }
    }
    CommTable *d = nullptr;
    HIP_TRY(ctx, hipMalloc((void **)&d, sizeof(CommTable)));
    if (hipMemcpy(d, &t, sizeof(t), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(d);
        return fail(ctx, CVO_HIP_ERR_HIP, "hipMemcpy(comm table) failed");
    }
    ctx->comm_table = d;
    return CVO_HIP_OK;
}

int cvo_hip_set_allreduce(cvo_hip_ctx *ctx, cvo_hip_allreduce_fn fn, void *user)
{
    cvo_lock::Api api_guard;
    if (!ctx) return CVO_HIP_ERR_INVALID;
    ctx->user_allreduce = fn;
    ctx->user_allreduce_arg = user;
    return CVO_HIP_OK;
}

int cvo_hip_transform_pcd(cvo_hip_ctx *ctx, const float R[9], const float T[3])
{
    cvo_lock::Api api_guard;
    if (!ctx || !R || !T) return CVO_HIP_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // update_tf(): the sweeps apply [Rt|t] while staging the moving cloud
    DevState *h = &ctx->st_host[kPollSlots];
    std::memcpy(h->R, R, sizeof(h->R));
    std::memcpy(h->T, T, sizeof(h->T));
    cvo_math::inverse_tf(R, T, h->Rt, h->t);
    h->done = 0;
    int rc = fill_filter_geometry(ctx, h);
    if (rc) return rc;
    rc = push_state_fields(ctx, offsetof(DevState, R), offsetof(DevState, ell) - offsetof(DevState, R));
    if (rc) return rc;
    rc = push_state_fields(ctx, offsetof(DevState, Rt), offsetof(DevState, used_Rt) - offsetof(DevState, Rt));
    if (rc) return rc;
    rc = push_state_fields(ctx, offsetof(DevState, done), sizeof(int32_t));
    if (rc) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->have_tf = true;
    return CVO_HIP_OK;
}

int cvo_hip_flow(cvo_hip_ctx *ctx, float ell, double out13[13])
{
    cvo_lock::Api api_guard;
    if (!ctx || !out13) return CVO_HIP_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->have_tf) return fail(ctx, CVO_HIP_ERR_INVALID, "transform_pcd not called");
    int rc = mailboxes_usable(ctx);
    if (rc) return rc;
    DevState *h = &ctx->st_host[kPollSlots];
    h->kc = make_kconsts(ctx->dprm, ell);
    h->kc_ell = -1.0f;   // (never equal to an ell: prepare_iteration recomputes)
    rc = fill_filter_geometry(ctx, h);
    if (rc) return rc;
    compute_filter_bounds(h, false);
    rc = push_state_fields(ctx, offsetof(DevState, kc),
                           offsetof(DevState, xi) - offsetof(DevState, kc));
    if (rc) return rc;
    for (bool redo = true; redo;) {
        rc = zero_counters(ctx);
        if (!rc) rc = enqueue_flow(ctx, true, 0, false, nullptr, 0);
        if (!rc) rc = check_overflow_and_grow(ctx, &redo);
        if (rc) return rc;
        if (multi_rank(ctx)) {
            // a list that overflowed on ANY rank poisoned nnz before the sums went over the
            // ranks: every rank sees the NaN and redoes the pass (the one that overflowed with
            // a larger list), so that all of them run the same number of exchanges
            double nnz = 0.0;
            rc = fetch_red(ctx, RED_FLOW + 8, 1, &nnz);
            if (rc) return rc;
            if (nnz != nnz) redo = true;
        }
    }
    rc = fetch_red(ctx, RED_FLOW, 13, out13);
    if (!rc && ctx->profiling) rc = drain_events(ctx);
    return rc;
}

int cvo_hip_step_coeffs(cvo_hip_ctx *ctx, const float omega[3], const float v[3], float ell,
                        double bcde[4])
{
    cvo_lock::Api api_guard;
    if (!ctx || !omega || !v || !bcde) return CVO_HIP_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->have_tf) return fail(ctx, CVO_HIP_ERR_INVALID, "transform_pcd not called");
    int rc = mailboxes_usable(ctx);
    if (rc) return rc;
    DevState *h = &ctx->st_host[kPollSlots];
    h->kc = make_kconsts(ctx->dprm, ell);
    h->kc_ell = -1.0f;   // (never equal to an ell: prepare_iteration recomputes)
    rc = fill_filter_geometry(ctx, h);
    if (rc) return rc;
    compute_filter_bounds(h, false);
    h->xi = cvo_math::make_xi_consts(omega, v);
    rc = push_state_fields(ctx, offsetof(DevState, kc),
                           offsetof(DevState, omega) - offsetof(DevState, kc));
    if (rc) return rc;
    // stand-alone call: rebuild A (filter + PROC_FLOW records the kept weights),
    // then stream it for the coefficient sums
    int rlo, rhi, slo, shi;
    shard_ranges(ctx, rlo, rhi, slo, shi);
    for (bool redo = true; redo;) {
        rc = zero_counters(ctx);
        if (!rc) rc = enqueue_filter(ctx, LIST_XY, ctx->fixed, rlo, rhi, 0, ctx->moving, 1, 0);
        if (!rc) rc = enqueue_process(ctx, PROC_FLOW, LIST_XY, ctx->part_flow, ctx->fixed.pos,
                                      ctx->fixed.feat, 0, ctx->moving.pos, ctx->moving.feat, 1, 0, 0);
        if (!rc) rc = check_overflow_and_grow(ctx, &redo);
        if (rc) return rc;
    }
    rc = enqueue_step(ctx, 0, false, nullptr, 0);
    if (rc) return rc;
    rc = fetch_red(ctx, RED_STEP, 4, bcde);
    if (!rc && ctx->profiling) rc = drain_events(ctx);
    return rc;
}

int cvo_hip_pick_step(const double bcde[4], float min_step, float *step)
{
    if (!bcde || !step) return CVO_HIP_ERR_INVALID;
    *step = cvo_math::pick_step(bcde, min_step);
    return CVO_HIP_OK;
}

int cvo_hip_exp_se3(const float omega[3], const float v[3], float dt, float dR[9], float dT[3])
{
    if (!omega || !v || !dR || !dT) return CVO_HIP_ERR_INVALID;
    cvo_math::exp_se3(omega, v, dt, dR, dT);
    return CVO_HIP_OK;
}

int cvo_hip_dist_se3(const float omega[3], const float v[3], float dt, float *dist)
{
    if (!omega || !v || !dist) return CVO_HIP_ERR_INVALID;
    *dist = cvo_math::dist_se3(omega, v, dt);
    return CVO_HIP_OK;
}

int cvo_hip_function_inner_product(cvo_hip_ctx *ctx, float ell, float *out)
{
    cvo_lock::Api api_guard;
    if (!ctx || !out) return CVO_HIP_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // untransformed positions, colour cut with sp_thres (ref acvo.cpp:391-392)
    DevParams dp = ctx->dprm;
    // the spatial threshold is written log(sp_thres/sigma/sigma) here (ref acvo.cpp:391): two
    // float divisions, not the division by the float product s2 of se_kernel (ref :100)
    dp.log_sp_s2 = (float)std::log((double)(ctx->prm.sp_thres / ctx->prm.sigma / ctx->prm.sigma));
    if (ctx->prm.mode == CVO_HIP_MODE_ACVO) {
        dp.c_sp = ctx->prm.sp_thres;
        dp.tau_c = (float)(-2.0 * ctx->prm.c_ell * ctx->prm.c_ell *
                           (double)(float)std::log((double)(dp.c_sp / ctx->prm.c_sigma / ctx->prm.c_sigma)));
    }
    int rc = mailboxes_usable(ctx);
    if (rc) return rc;
    DevState *h = &ctx->st_host[kPollSlots];
    h->kc = make_kconsts(dp, ell);
    h->kc_ell = -1.0f;   // (never equal to an ell: prepare_iteration recomputes)
    rc = fill_filter_geometry(ctx, h);
    if (rc) return rc;
    compute_filter_bounds(h, true);
    h->done = 0;
    rc = push_state_fields(ctx, offsetof(DevState, kc),
                           offsetof(DevState, xi) - offsetof(DevState, kc));
    if (rc) return rc;
    rc = push_state_fields(ctx, offsetof(DevState, done), sizeof(int32_t));
    if (rc) return rc;
    int rlo, rhi, slo, shi;
    shard_ranges(ctx, rlo, rhi, slo, shi);
    PostFlowArgs pa{};
    pa.st = ctx->st;
    pa.prm = ctx->dprm;
    pa.prm.mode = CVO_HIP_MODE_CVO;   // no self terms here
    pa.nblk = ctx->proc_blocks;
    pa.flags = POST_REDUCE;
    pa.comm = ctx->comm_table;
    for (bool redo = true; redo;) {
        rc = zero_counters(ctx);
        if (!rc) rc = enqueue_filter(ctx, LIST_XY, ctx->fixed, rlo, rhi, 0, ctx->moving, 0, 0);
        if (!rc) rc = enqueue_process(ctx, PROC_FLOW, LIST_XY, ctx->part_flow, ctx->fixed.pos,
                                      ctx->fixed.feat, 0, ctx->moving.pos, ctx->moving.feat, 0, 0, 0);
        if (!rc) rc = check_overflow_and_grow(ctx, &redo);
        if (rc) return rc;
    }
    pa.part_flow = (const double *)ctx->part_flow.p;
    launch_post_flow(pa, ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    rc = reduce_over_ranks(ctx, RED_FLOW, 9);
    if (rc) return rc;
    double red[9];
    rc = fetch_red(ctx, RED_FLOW, 9, red);
    if (rc) return rc;
    *out = (float)(red[6] / red[8]);
    if (ctx->profiling) return drain_events(ctx);
    return CVO_HIP_OK;
}

int cvo_hip_function_inner_product_clouds(cvo_hip_ctx *ctx, float ell, const float *xyz_a, const float *feat_a,
                                          int na, const float *xyz_b, const float *feat_b, int nb,
                                          int feat_layout, float *out)
{
    cvo_lock::Api api_guard;
    if (!ctx || !out) return CVO_HIP_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // the two clouds go to buffers of their own: the registration's fixed / moving clouds, a
    // pending set_pcd() and the transform handed to the low-level calls stay as they are
    // (the reference's function reads its two arguments and `ell`, nothing else: acvo.cpp:385)
    const bool had_tf = ctx->have_tf;
    std::swap(ctx->fixed, ctx->scratch_a);
    std::swap(ctx->moving, ctx->scratch_b);
    int rc = upload_cloud(ctx, ctx->fixed, xyz_a, feat_a, na, feat_layout);
    if (!rc) rc = upload_cloud(ctx, ctx->moving, xyz_b, feat_b, nb, feat_layout);
    if (!rc) rc = cvo_hip_function_inner_product(ctx, ell, out);
    std::swap(ctx->fixed, ctx->scratch_a);
    std::swap(ctx->moving, ctx->scratch_b);
    ctx->have_tf = had_tf;
    return rc;
}

int cvo_hip_get_wave_load(cvo_hip_ctx *ctx, uint32_t *members_per_wave, int capacity, int *waves)
{
    cvo_lock::Api api_guard;
    if (!ctx || !members_per_wave || !waves || capacity < 0) return CVO_HIP_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int n = std::min(capacity, 4 * ctx->proc_blocks);
    *waves = 0;
    if (!ctx->kept_cnt.p || n <= 0) return CVO_HIP_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(members_per_wave, ctx->kept_cnt.p, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    *waves = n;
    return CVO_HIP_OK;
}

int cvo_hip_set_graph_capture(cvo_hip_ctx *ctx, int enable)
{
    cvo_lock::Api api_guard;
    if (!ctx) return CVO_HIP_ERR_INVALID;
    return apply_option(ctx, "graph_capture", enable ? 1.0 : 0.0);
}

int cvo_hip_set_profiling(cvo_hip_ctx *ctx, int enable)
{
    cvo_lock::Api api_guard;
    if (!ctx) return CVO_HIP_ERR_INVALID;
    ctx->profiling = enable != 0;
    return CVO_HIP_OK;
}

int cvo_hip_get_profile(cvo_hip_ctx *ctx, cvo_hip_profile *out, int reset)
{
    cvo_lock::Api api_guard;
    if (!ctx || !out) return CVO_HIP_ERR_INVALID;
    int rc = drain_events(ctx);
    if (rc) return rc;
    *out = ctx->prof;
    if (reset) ctx->prof = cvo_hip_profile{};
    return CVO_HIP_OK;
}

int cvo_hip_get_graph_stats(const cvo_hip_ctx *ctx, long long *launches_from_cache, long long *captures)
{
    if (!ctx || !launches_from_cache || !captures) return CVO_HIP_ERR_INVALID;
    *launches_from_cache = ctx->plans.hits;
    *captures = ctx->plans.captures;
    return CVO_HIP_OK;
}

int cvo_hip_get_run_stats(cvo_hip_ctx *ctx, int *runs, int *declined, int *iterations, int *candidates)
{
    cvo_lock::Api api_guard;   // (runtime calls: not inside another thread's capture window)
    if (!ctx || !ctx->st) return CVO_HIP_ERR_INVALID;
    // (diagnostics: the counters live in the state's tail, which a cvo_hip_align that ended on the mirrored head has not copied;
    // into a buffer of this call's own -- the pinned state copies belong to the align loop)
    struct { int32_t run_count, run_entered, run_iterations, run_candidates; } f{};
    static_assert(offsetof(DevState, run_candidates) - offsetof(DevState, run_count) == 12, "four consecutive words");
    if (hipSetDevice(ctx->device) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess ||
        hipMemcpy(&f, reinterpret_cast<const char *>(ctx->st) + offsetof(DevState, run_count), sizeof(f), hipMemcpyDeviceToHost) != hipSuccess)
        return CVO_HIP_ERR_HIP;
    if (runs) *runs = f.run_entered;
    if (declined) *declined = f.run_count - f.run_entered;
    if (iterations) *iterations = f.run_iterations;
    if (candidates) *candidates = f.run_candidates;
    return CVO_HIP_OK;
}

int cvo_hip_get_run_clocks(cvo_hip_ctx *ctx, long long clocks16[16])
{
    cvo_lock::Api api_guard;
    if (!ctx || !ctx->st || !clocks16) return CVO_HIP_ERR_INVALID;
    long long clk[16];
    static_assert(sizeof(clk) == sizeof(DevState::run_clk), "run_clk");
    if (hipSetDevice(ctx->device) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess ||
        hipMemcpy(clk, reinterpret_cast<const char *>(ctx->st) + offsetof(DevState, run_clk), sizeof(clk), hipMemcpyDeviceToHost) != hipSuccess)
        return CVO_HIP_ERR_HIP;
    for (int q = 0; q < 16; ++q) clocks16[q] = clk[q];
    return CVO_HIP_OK;
}

int cvo_hip_set_option(cvo_hip_ctx *ctx, const char *key, double value)
{
    cvo_lock::Api api_guard;
    if (!ctx || !key) return CVO_HIP_ERR_INVALID;
    if (hipSetDevice(ctx->device) != hipSuccess) return CVO_HIP_ERR_HIP;
    const int rc = apply_option(ctx, key, value);
    if (rc) return fail(ctx, rc, "cvo_hip_set_option: unknown key or value out of range");
    return CVO_HIP_OK;
}

int cvo_hip_get_option(const cvo_hip_ctx *ctx, const char *key, double *value)
{
    if (!ctx || !key || !value) return CVO_HIP_ERR_INVALID;
    auto is = [&](const char *k) { return std::strcmp(key, k) == 0; };
    const CtxOptions &o = ctx->opt;
    if (is("graph_capture")) *value = ctx->use_graphs;
    else if (is("no_graph")) *value = o.no_graph;
    else if (is("head_graphs")) *value = ctx->head_graphs;
    else if (is("head_mode")) *value = ctx->allow_head;
    else if (is("resident_runs")) *value = ctx->allow_run;
    else if (is("run_solvers_max")) *value = ctx->run_g_max;
    else if (is("run_candidates_max")) *value = o.run_cand;
    else if (is("run_timeout_ms")) *value = o.run_timeout_ms;
    else if (is("run_fault")) *value = o.run_fault;
    else if (is("merged_launches")) *value = ctx->allow_merge;
    else if (is("async_builds")) *value = ctx->allow_async;
    else if (is("list_pass_blocks")) *value = ctx->proc_blocks_forced ? ctx->proc_blocks : 0;
    else if (is("post_debug")) *value = o.post_debug;
    else if (is("mailbox_timeout_s")) *value = o.mailbox_timeout_s;
    else if (is("candidate_records")) *value = !o.no_cand;
    else if (is("sync_upload")) *value = o.sync_upload;
    else if (is("engine_debug")) *value = o.engine_debug;
    else if (is("one_launch_hand_over")) *value = !o.no_cloud_one;
    else if (is("small_calls_alone")) *value = !o.no_alone;
    else if (is("fused_groups")) *value = !o.no_fuse;
    else if (is("engines")) *value = o.engines_force;
    else if (is("narrow_merge")) *value = o.narrow_merge;
    else if (is("narrow_blocks")) *value = o.narrow_blocks;
    else if (is("engine_crowd")) *value = o.engine_crowd;
    else if (is("engine_merge_max")) *value = o.engine_merge_max;
    else if (is("list_init")) *value = o.list_init;
    else if (is("kept_pack")) *value = !o.no_pack;
    else if (is("list_margin")) *value = o.list_margin;
    else if (is("final_mirror")) *value = !o.no_final_mirror;
    else if (is("twist_on_shared_gpu")) *value = o.twist_on_shared_gpu;
    else if (is("comm_debug")) *value = o.comm_debug;
    else if (is("wait_policy")) *value = o.wait_policy;
    else if (is("acvo_runs")) *value = !o.no_acvo_run;
    else if (is("tail_alone")) *value = o.tail_alone;
    else if (is("tail_handovers")) *value = (double)ctx->tail_handovers;
    else if (is("run_restart")) *value = !o.no_restart;
    else if (is("side_builds")) *value = !o.no_side_builds;
    else if (is("run_build_at")) *value = o.run_build_at;
    else if (is("side_builds_launched")) *value = (double)ctx->side_builds_launched;
    else if (is("alone_max")) *value = o.alone_max;
    // read-only counters
    else if (is("run_timeouts")) *value = (double)ctx->run_timeouts;
    else if (is("run_aborts")) *value = (double)ctx->run_aborts;
    else if (is("no_run_backoff")) *value = ctx->no_run_backoff;
    else return CVO_HIP_ERR_INVALID;
    return CVO_HIP_OK;
}

long long cvo_hip_get_mirror_retries(void) { return mirror_retries().load(std::memory_order_relaxed); }

int cvo_hip_synchronize(cvo_hip_ctx *ctx)
{
    cvo_lock::Api api_guard;
    if (!ctx) return CVO_HIP_ERR_INVALID;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CVO_HIP_OK;
}


}   // extern "C"
