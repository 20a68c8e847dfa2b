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
#include "cvo_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <memory>
#include <deque>
#include <mutex>
#include <chrono>
#include <atomic>
#include <thread>
#include <vector>

#include "cvo_comm.h"
#include "cvo_cloud.h"
#include "cvo_lock.h"
#include "cvo_device.h"
#include "se3_math.hpp"

using namespace cvo_dev;

namespace {

struct Cloud {
    float4 *pos = nullptr;   // Morton-sorted; .w = the 5th feature
    float *feat = nullptr;   // same order: f0..f4, index in the caller's cloud (int bits), 2 pad
    float4 *seg = nullptr;   // bounding sphere (centre, radius) of every SEG consecutive points
    int n = 0;               // points, as the caller counts them
    int np = 0;              // rows of the device arrays: n padded to CLOUD_PAD (cvo_cloud.h); what kernels get
    int pad_axis = 0;        // where the padding rows are parked (the two clouds of a pair differ)
    int cap = 0;
    float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};   // bounding box
    // A hand-over from host arrays does not wait for the device (round 3): the cloud's own pinned staging and
    // bounding-box words, an event behind the preparation; whoever needs the box or another stream's view of
    // the arrays waits then (cloud_ready: at the next compute entry point, or when the staging is needed again)
    void *stage = nullptr;
    size_t stage_bytes = 0;
    float *bbox_pin = nullptr;        // [6] pinned
    hipEvent_t ready_ev = nullptr;
    hipEvent_t wait_ev = nullptr;     // what `pending` waits for: ready_ev, or the event of a batched hand-over (borrowed)
    bool pending = false;
};

struct EventPair {
    hipEvent_t a, b;
    int kind;      // SweepMode
    int iter_tag;  // align() iteration the launch belongs to, -1 outside align()
    double pairs;
};

struct FilterPlan {
    dim3 grid;
    int jt = 0;
};

struct DevBuf {   // a growable device array
    void *p = nullptr;
    size_t bytes = 0;
};

struct List {     // LIST_XY/XX/YY: TileEntry[cap] in a; LIST_KEPT: uint2[cap] in a, float[cap] in b
    DevBuf a, b;
    uint32_t cap = 0;   // entries, a multiple of NSUB
};

// iterations per captured batch of a registration on its own (paced submission, job_pump: measured with the next
// batch enqueued when the running one has finished -- 6 / 8 / 12 / 16 / 24: 643 / 644 / 632 / 625 / 600 reg/s at 10k x 10k,
// 650 / 668 / 668 / 644 / 672 at 3k x 3k, acvo 388 / 393 / 376 / 392 / 380)
// ... and in a fused group, where a batch boundary is also where a slot that fell free is noticed and
// refilled.  With tables of 16 slots (a batch of 64 pairs = 48 in flight + 16 waiting for a slot) shorter was
// better (3 -> 2857, 4 -> 2917, 6 -> 2693, 8 -> 2621 registrations/s); with tables of 32 a batch of 64 is in
// flight at once and the iterations are cheaper (candidate lists): 4 / 8 / 10 / 12 / 16 / 24 per captured batch:
// 256 pairs per call 3905 / 4188 / 4254 / 4248 / 4277 / 4105, 64 pairs 3604 / 3625 / - / 3650 / 3594 / 3571,
// 32 pairs 2630 / 2734 / - / 2723 / 2453 / 2694, 8 x 20k 1022 / 1029 / 1028 / 1014 / 995 / 961 -> 10
constexpr int kEngineBatch = 10;
// (an even number: a head-mode batch must leave the state's head in its first copy, cvo_kernels.hip "the head")
constexpr int kBatch = 8;

// The kernels of the loop read their argument blocks from a table of Slots in device memory
// (cvo_device.h "Argument tables"): one slot for a registration on its own (cvo_hip_align), up
// to MAXG for a fused group (cvo_hip_align_many).  The host keeps an image of what it last sent
// per slot and sends a slot again only when its image changed -- through a small ring of
// pinned staging buffers, ordered on the stream that runs the loop.
constexpr int kStage = 4;
struct TableBuf {
    Slot *dev = nullptr;
    int nslots = 0;
    std::vector<Slot> image;          // what the device holds (after the queued copies)
    Slot *stage = nullptr;            // pinned [kStage][nslots]
    hipEvent_t stage_ev[kStage] = {};
    bool stage_used[kStage] = {};
    int next = 0;

    // (s: the stream every later copy into the table is ordered on -- the zero fill must be too:
    // a non-blocking stream does not wait for the null stream's memset)
    int init(int n, hipStream_t s)
    {
        if (dev) return 0;
        if (hipMalloc((void **)&dev, (size_t)n * sizeof(Slot)) != hipSuccess) { dev = nullptr; return -1; }
        if (hipMemsetAsync(dev, 0, (size_t)n * sizeof(Slot), s) != hipSuccess) return -1;
        if (hipHostMalloc((void **)&stage, (size_t)kStage * n * sizeof(Slot), hipHostMallocDefault) != hipSuccess) return -1;
        for (int i = 0; i < kStage; ++i)
            if (hipEventCreateWithFlags(&stage_ev[i], hipEventDisableTiming) != hipSuccess) return -1;
        nslots = n;
        image.assign((size_t)n, Slot{});
        return 0;
    }
    void destroy()
    {
        for (int i = 0; i < kStage; ++i)
            if (stage_ev[i]) (void)hipEventDestroy(stage_ev[i]);
        if (stage) (void)hipHostFree(stage);
        if (dev) (void)hipFree(dev);
        dev = nullptr; stage = nullptr; nslots = 0;
        image.clear();
    }
    // Make the device hold want[0 .. nslots): as far as the first `nq` argument blocks of the
    // ACTIVE slots and every slot's `active` flag go.  Whatever differs travels in ONE copy (the
    // span from the first to the last slot that changed), ordered on s -- every copy is a stop of
    // its own between two batches of the stream.
    int sync(const Slot *want, hipStream_t s, int nq = MAX_OPS)
    {
        const size_t head = offsetof(Slot, op);
        int lo = nslots, hi = -1;
        for (int z = 0; z < nslots; ++z) {
            const Slot &img = image[(size_t)z];
            bool same = std::memcmp(&img, &want[z], head) == 0;
            if (same && want[z].active) same = std::memcmp(img.op, want[z].op, (size_t)nq * sizeof(OpArgs)) == 0;
            if (!same) { lo = std::min(lo, z); hi = z; }
        }
        if (hi < 0) return 0;
        const int b = next;
        next = (next + 1) % kStage;
        if (stage_used[b] && hipEventSynchronize(stage_ev[b]) != hipSuccess) return -1;   // (kStage copies ago)
        Slot *st = stage + (size_t)b * nslots;
        const size_t n = (size_t)(hi - lo + 1);
        for (int z = lo; z <= hi; ++z) {
            if (want[z].active) image[(size_t)z] = want[z];
            else image[(size_t)z].active = 0;   // (its argument blocks stay what they were: nobody reads them)
            st[z] = image[(size_t)z];
        }
        if (hipMemcpyAsync(&dev[lo], &st[lo], n * sizeof(Slot), hipMemcpyHostToDevice, s) != hipSuccess) return -1;
        if (hipEventRecord(stage_ev[b], s) != hipSuccess) return -1;
        stage_used[b] = true;
        return 0;
    }
};

// A captured batch of kBatch iterations of a launch plan (hipGraph).  It depends on the table's
// address and on the plan -- kernels, grids, LDS sizes -- not on any argument: one capture
// serves every frame pair (and every membership of a fused group) of the same shape.
struct PlanGraph {
    std::vector<TLaunch> plan;
    int iterations = 0;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    uint64_t stamp = 0;
};
struct PlanCache {
    std::vector<PlanGraph> graphs;
    uint64_t clock = 0;
    long long hits = 0, captures = 0;
    int fails = 0;
    void drop()
    {
        for (auto &g : graphs) {
            if (g.exec) (void)hipGraphExecDestroy(g.exec);
            if (g.graph) (void)hipGraphDestroy(g.graph);
        }
        graphs.clear();
    }
};
constexpr int kPollSlots = 4;
constexpr int kEvProcFlow = 10, kEvProcStep = 11;   // EventPair::kind of the list kernels (0..2: k_filter of list l)
constexpr int kProcStepTwist = 100;   // RecOp::mode of a k_step_twist launch
constexpr int kFlowBuild = 101;       // RecOp::mode of a k_flow_build launch (RecOp::f = the build's arguments)
constexpr int kFilterAhead = 102;     // RecOp::mode of an xx / yy filter that builds ahead (rides in the flow launch)

// One kernel launch of an iteration, recorded instead of launched (fused mode:
// the launches of several registrations are merged slot by slot).
struct RecOp {
    enum Kind { FILTER, PROCESS, POST_FLOW, POST_STEP } kind;
    int mode = 0;   // PROCESS: ProcMode
    FilterArgs f{};
    ProcessArgs p{};
    PostFlowArgs pf{};
    PostStepArgs ps{};
};

}   // namespace

struct cvo_hip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    cvo_hip_params prm{};
    DevParams dprm{};
    Cloud fixed, moving;
    Cloud scratch_a, scratch_b;      // cvo_hip_function_inner_product_clouds: never the registration's clouds
    DevState *st = nullptr;          // device
    DevHead *st2 = nullptr;          // device: second copy of the state's head (head mode, cvo_kernels.hip)
    bool head_mode = false;          // the plan of the align() in progress is a head-mode plan
    bool allow_head = true;          // CVO_HIP_NO_HEAD
    DevState *st_host = nullptr;     // pinned [kPollSlots + 2]
    int32_t *done_mirror = nullptr;  // pinned (in the last slot): the post kernels copy `done` here
    int32_t *progress_mirror = nullptr;   // pinned, next to it: slots the post-step kernel has completed
    std::vector<RecOp> *rec = nullptr;   // not null: record launches instead of issuing them
    int proc_blocks = PROC_BLOCKS;       // blocks of the list kernels (fewer in fused launches)
    int proc_blocks_default = PROC_BLOCKS;
    bool proc_blocks_forced = false;     // CVO_HIP_PROC_BLOCKS
    DevBuf raw_xyz, raw_feat;            // upload_cloud: the caller's arrays as they came
    DevBuf sort_keys[2], sort_idx[2], sort_tmp;   // ... scratch of the device-side Morton sort
    float *bbox_dev = nullptr;           // [6] device, bounding box of a cloud handed over in device memory
    float *bbox_host = nullptr;          // [6] pinned
    // asynchronous xy builds (cvo_device.h plan_xy_async): the k_filter blocks of the xy
    // list ride in the launch of the flow pass of the same slot (k_flow_build) and fill
    // the idle one of two buffers
    FilterArgs xy_build{};               // argument block of those filter blocks (this slot)
    bool have_xy_build = false;
    bool allow_async = true;
    bool crowded = false;                // set by align_many: many registrations share the launches
    DevBuf cand[3], cand_cnt[3];         // the candidate lists of the xy / xx / yy tile lists (ProcessArgs::cand, cand_cnt)
    DevBuf cand_xyb, cand_cnt_xyb;       // head mode: the record of the second buffer of the xy list (ProcessArgs::cand_b)
    DevBuf cand_sfb[2], cand_cnt_sfb[2]; // ... and of the xx / yy lists (acvo)
    int ck_nblk[3] = {0, 0, 0};          // recorded plan: the pass over list l keeps a candidate list with this many blocks (0: no)
    DevBuf pos_bt;                       // crowded: the moving cloud under the iteration's transform (FilterArgs::pos_bt)
    bool lone = true;                    // this registration has its launches to itself
    bool allow_async_self = true;
    bool use_async_self = false;         // acvo, lone: self lists built ahead, PROC_SELF in the flow launch
    bool use_async = false;              // decided per align(): single rank, not profiling
    bool in_loop = false;                // enqueueing iterations of align()
    bool plan_recording = false;         // ... into the RecOp list a table plan is made of (record_iteration)
    bool merge_twist = false;            // inside align(): k_step_twist replaces k_post_flow + PROC_STEP
    bool allow_merge = true;
    cvo_hip_trace *cur_trace = nullptr;  // trace buffer of the iterations being enqueued
    int cur_trace_cap = 0;
    hipEvent_t poll_ev[kPollSlots]{};
    DevBuf part_flow, part_xx, part_yy, part_step;   // [PROC_BLOCKS][NACC_MAX] float64
    List lists[LIST_N];
    DevBuf kept_cnt;                 // uint32[PROC_WAVES]
    cvo_hip_trace *trace_dev = nullptr;
    int trace_dev_cap = 0;
    bool have_tf = false;
    int row_lo = 0, row_hi = -1, srow_lo = 0, srow_hi = -1;
    bool sharded = false;
    cvo_comm *comm = nullptr;
    // mailbox all-reduce (cvo_device.h Mailbox / CommTable)
    Mailbox *mailbox = nullptr;          // this rank's own, device memory (uncached where the runtime offers it)
    CommTable *comm_table = nullptr;     // device copy; not null = connected: the post kernels exchange
    void *mail_opened[MAX_WORLD] = {};   // peers' mailboxes opened from IPC handles (closed at destroy)
    int mail_rank = 0, mail_world = 0;
    bool mail_broken = false;            // an exchange timed out: the ranks' sequence numbers no longer agree (see job_finish)
    cvo_hip_allreduce_fn user_allreduce = nullptr;
    void *user_allreduce_arg = nullptr;
    bool profiling = false;
    long long *post_dbg = nullptr;   // CVO_HIP_POST_DEBUG diagnostics
    TableBuf table;                  // this registration's own argument table (one slot): cvo_hip_align
    PlanCache plans;                 // ... and the batches captured for it
    std::vector<TLaunch> plan;       // launches of one iteration of the align() in progress
    hipStream_t loop_stream = nullptr;   // stream the align() in progress runs on (a fused group's, else `stream`)
    bool warm = false;               // every device buffer of the loop has been allocated
    bool use_graphs = true;
    int iter_tag = -1;
    std::vector<EventPair> events;
    cvo_hip_profile prof{};
    std::string err;
};

namespace {

#define HIP_TRY(ctx, expr)                                                                   \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            if (ctx) (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);         \
            return CVO_HIP_ERR_HIP;                                                          \
        }                                                                                    \
    } while (0)

int fail(cvo_hip_ctx *ctx, int code, const char *msg)
{
    if (ctx) ctx->err = msg;
    return code;
}

// The switches this library reads from the environment, one per mechanism: test switches (read where a plan is
// recorded, so that a test can flip them between two registrations) and diagnostics.
bool env_no_cand() { return getenv("CVO_HIP_NO_CAND") != nullptr; }           // no candidate records: expand the tile list every time
bool env_no_graph() { return getenv("CVO_HIP_NO_GRAPH") != nullptr; }         // no stream captures at all
bool env_sync_upload() { static const bool v = getenv("CVO_HIP_SYNC_UPLOAD") != nullptr; return v; }   // hand-overs wait for the device
bool env_engine_debug() { static const bool v = getenv("CVO_HIP_ENGINE_DEBUG") != nullptr; return v; } // host-side clocks of the engines

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

int ensure_buf(cvo_hip_ctx *ctx, DevBuf &b, size_t bytes);

// the hand-over of `c` has completed on the device; its bounding box is on the host
int cloud_ready(cvo_hip_ctx *ctx, Cloud &c)
{
    if (!c.pending) return CVO_HIP_OK;
    // (pending stays up if the wait fails: the box is still the zeros of upload_cloud, and every later entry
    // point must fail here again instead of building its filter geometry from them)
    HIP_TRY(ctx, hipEventSynchronize(c.wait_ev ? c.wait_ev : c.ready_ev));
    c.pending = false;
    for (int a = 0; a < 3; ++a) { c.lo[a] = c.bbox_pin[a]; c.hi[a] = c.bbox_pin[3 + a]; }
    return CVO_HIP_OK;
}

// The cloud into the kernels' layout (cvo_cloud.hip): Morton order -- consecutive device
// points are spatial neighbours, so a wave's 64 rows and a 16-column MFMA tile are compact
// patches and most (wave, tile) steps see no candidate -- packed rows, bounding spheres
// of the 64-point runs.  `on_device`: xyz / feat are device pointers (same device).
// What every hand-over of cloud `c` begins with: a hand-over of the same cloud that is still on its way ends,
// the arguments are checked, the device arrays hold n points (padded), the cloud's pinned words and event exist.
int cloud_reserve(cvo_hip_ctx *ctx, Cloud &c, const float *xyz, const float *feat, int n, int layout)
{
    const int np = cloud_padded(n);
    const Cloud &other = (&c == &ctx->fixed) ? ctx->moving : ctx->fixed;
    {   // (a hand-over of this cloud that is still on its way uses the staging and the arrays)
        const int rcw = cloud_ready(ctx, c);
        if (rcw) return rcw;
    }
    if (n < 0 || (n > 0 && (!xyz || !feat))) return fail(ctx, CVO_HIP_ERR_INVALID, "null cloud");
    if (n > (1 << 26))   // (the list kernels address a cloud through 32-bit byte offsets: 32 B per point)
        return fail(ctx, CVO_HIP_ERR_INVALID, "cloud too large: at most 2^26 points");
    if (layout != CVO_HIP_FEAT_COLMAJOR && layout != CVO_HIP_FEAT_ROWMAJOR)
        return fail(ctx, CVO_HIP_ERR_INVALID, "bad feat_layout");
    if (np > c.cap) {
        if (c.pos) HIP_TRY(ctx, hipFree(c.pos));
        if (c.feat) HIP_TRY(ctx, hipFree(c.feat));
        if (c.seg) HIP_TRY(ctx, hipFree(c.seg));
        c.pos = nullptr; c.feat = nullptr; c.seg = nullptr; c.cap = 0;
        HIP_TRY(ctx, hipMalloc((void **)&c.pos, (size_t)np * sizeof(float4)));
        HIP_TRY(ctx, hipMalloc((void **)&c.feat, (size_t)np * FEAT_STRIDE * sizeof(float)));
        HIP_TRY(ctx, hipMalloc((void **)&c.seg, (size_t)((np + SEG - 1) / SEG) * sizeof(float4)));
        c.cap = np;
    }
    c.n = n;
    c.np = np;
    c.pad_axis = (other.n > 0) ? 1 - other.pad_axis : 0;
    for (int a = 0; a < 3; ++a) { c.lo[a] = 0.0f; c.hi[a] = 0.0f; }
    if (n == 0) return CVO_HIP_OK;
    if (!ctx->bbox_host) {
        HIP_TRY(ctx, hipHostMalloc((void **)&ctx->bbox_host, 6 * sizeof(float), hipHostMallocDefault));
        HIP_TRY(ctx, hipMalloc((void **)&ctx->bbox_dev, 6 * sizeof(float)));
    }
    if (!c.bbox_pin) {
        HIP_TRY(ctx, hipHostMalloc((void **)&c.bbox_pin, 6 * sizeof(float), hipHostMallocDefault));
        HIP_TRY(ctx, hipEventCreateWithFlags(&c.ready_ev, hipEventDisableTiming));
    }
    return CVO_HIP_OK;
}

int upload_cloud(cvo_hip_ctx *ctx, Cloud &c, const float *xyz, const float *feat, int n,
                 int layout, bool on_device = false)
{
    {
        const int rcr = cloud_reserve(ctx, c, xyz, feat, n, layout);
        if (rcr || n == 0) return rcr;
    }
    const int np = c.np;
    const size_t bytes_xyz = (size_t)n * 3 * sizeof(float), bytes_feat = (size_t)n * CVO_HIP_NFEAT * sizeof(float);
    const float *d_xyz = xyz, *d_feat = feat;
    if (!on_device) {
        // the arrays as they are, through pinned staging kept with the cloud
        if (bytes_xyz + bytes_feat > c.stage_bytes) {
            if (c.stage) (void)hipHostFree(c.stage);
            c.stage = nullptr;
            c.stage_bytes = 0;
            const size_t want = (bytes_xyz + bytes_feat) * 5 / 4 + 4096;
            if (hipHostMalloc(&c.stage, want, hipHostMallocDefault) != hipSuccess)
                return fail(ctx, CVO_HIP_ERR_NOMEM, "hipHostMalloc(upload staging) failed");
            c.stage_bytes = want;
        }
        // (raw_xyz / raw_feat and the sort scratch are shared by the two clouds of a context: stream order
        // keeps one hand-over's kernels ahead of the next one's copies; growing them frees memory a queued
        // kernel may still read, so a growth waits for the stream first)
        if (bytes_xyz > ctx->raw_xyz.bytes || bytes_feat > ctx->raw_feat.bytes) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        int rcb = ensure_buf(ctx, ctx->raw_xyz, bytes_xyz);
        if (!rcb) rcb = ensure_buf(ctx, ctx->raw_feat, bytes_feat);
        if (rcb) return rcb;
        char *hs = reinterpret_cast<char *>(c.stage);
        std::memcpy(hs, xyz, bytes_xyz);
        HIP_TRY(ctx, hipMemcpyAsync(ctx->raw_xyz.p, hs, bytes_xyz, hipMemcpyHostToDevice, ctx->stream));
        std::memcpy(hs + bytes_xyz, feat, bytes_feat);
        HIP_TRY(ctx, hipMemcpyAsync(ctx->raw_feat.p, hs + bytes_xyz, bytes_feat, hipMemcpyHostToDevice, ctx->stream));
        d_xyz = (const float *)ctx->raw_xyz.p;
        d_feat = (const float *)ctx->raw_feat.p;
    }
    // From here on the cloud is in device memory either way.
    const bool no_one = getenv("CVO_HIP_NO_CLOUD_ONE") != nullptr;   // (test switch, read per call: the multi-launch preparation)
    if (n <= CLOUD_ONE_MAX && !no_one) {
        // ONE launch (k_cloud_one: a block does the whole preparation, the sort in LDS); the bounding box goes
        // straight into the cloud's pinned words
        CloudJob jb{};
        jb.xyz = d_xyz; jb.feat = d_feat; jb.n = n; jb.colmajor = layout == CVO_HIP_FEAT_COLMAJOR ? 1 : 0;
        jb.np = np; jb.pad_axis = c.pad_axis;
        jb.pos = c.pos; jb.feat8 = c.feat; jb.seg = c.seg;
        void *bbox_d = nullptr;
        HIP_TRY(ctx, hipHostGetDevicePointer(&bbox_d, c.bbox_pin, 0));
        jb.bbox_out = (float *)bbox_d;
        HIP_TRY(ctx, cloud_prepare_one(jb, ctx->stream));
        HIP_TRY(ctx, hipEventRecord(c.ready_ev, ctx->stream));
        c.wait_ev = nullptr;
        c.pending = true;
        if (on_device || env_sync_upload()) return cloud_ready(ctx, c);
        return CVO_HIP_OK;
    }
    // Larger clouds: bounding box, keys, rocPRIM's radix sort, pack, spheres as launches of their own.  The box is
    // made on the device too and comes back to the host (the filter geometry of align() is made from it)
    // together with the end of the preparation: one synchronisation.
    HIP_TRY(ctx, cloud_bbox_device(d_xyz, n, ctx->bbox_dev, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(c.bbox_pin, ctx->bbox_dev, 6 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    int rc = CVO_HIP_OK;
    const size_t tmp = cloud_sort_scratch_bytes(n);
    if ((size_t)n * sizeof(uint32_t) > ctx->sort_keys[0].bytes || tmp > ctx->sort_tmp.bytes)
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // (see raw_xyz above)
    for (int q = 0; q < 2 && !rc; ++q) {
        rc = ensure_buf(ctx, ctx->sort_keys[q], (size_t)n * sizeof(uint32_t));
        if (!rc) rc = ensure_buf(ctx, ctx->sort_idx[q], (size_t)n * sizeof(int));
    }
    if (!rc) rc = ensure_buf(ctx, ctx->sort_tmp, tmp);
    if (rc) return rc;
    CloudPrep cp{};
    cp.np = np; cp.pad_axis = c.pad_axis;
    cp.xyz = d_xyz; cp.feat = d_feat; cp.n = n; cp.colmajor = layout == CVO_HIP_FEAT_COLMAJOR ? 1 : 0;
    cp.bbox = ctx->bbox_dev;
    for (int q = 0; q < 2; ++q) { cp.keys[q] = (uint32_t *)ctx->sort_keys[q].p; cp.idx[q] = (int *)ctx->sort_idx[q].p; }
    cp.scratch = ctx->sort_tmp.p; cp.scratch_bytes = tmp;
    cp.pos = c.pos; cp.feat8 = c.feat; cp.seg = c.seg;
    HIP_TRY(ctx, cloud_prepare_device(cp, ctx->stream));
    HIP_TRY(ctx, hipEventRecord(c.ready_ev, ctx->stream));
    c.wait_ev = nullptr;
    c.pending = true;
    // Host arrays were copied into the cloud's staging: the caller's are free at once, and the call does not
    // wait for the device (64 x 2 hand-overs of a batch overlap each other instead of costing 0.1 ms of host
    // time apiece).  Device arrays of the caller's are read by the queued kernels: they may be re-used
    // once this returns, so that form waits here.
    if (on_device || env_sync_upload()) return cloud_ready(ctx, c);
    return CVO_HIP_OK;
}

// ---------------------------------------------------------------------------
// The hand-over of a batch (cvo_hip_set_pcd_many): per device one stream, one pinned staging arena, one device
// arena for the caller's arrays as they come, a table of CloudJobs -- kept for the life of the process, like the
// engines.  One transfer per batch (or one per array where the caller's memory is page-locked already), one
// launch of k_cloud_one for all clouds of up to CLOUD_ONE_MAX points, one event the clouds of the batch wait
// for.  Larger clouds take upload_cloud's way.
struct Handover {
    std::mutex mu;
    hipStream_t s = nullptr;
    char *stage = nullptr;                               // pinned, the size of ...
    char *raw = nullptr;     size_t raw_bytes = 0;       // ... the device arena
    CloudJob *jobs_pin = nullptr, *jobs_dev = nullptr;   int jobs_cap = 0;
    hipEvent_t ev[16] = {};
    int next_ev = 0;
    hipEvent_t last = nullptr;                           // the event of the batch that used the arenas last
};
Handover *handover_of(int device)
{
    static Handover *h = new Handover[64];   // (never destroyed: see cvo_lock.h)
    return (device >= 0 && device < 64) ? &h[device] : nullptr;
}

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
        if (const char *e = getenv("CVO_HIP_LIST_INIT")) {
            const double v = atof(e);
            if (v > 0.0) want = v;
        }
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

bool multi_rank(const cvo_hip_ctx *ctx);
DevParams loop_params(const cvo_hip_ctx *ctx);
hipStream_t loop_stream(const cvo_hip_ctx *ctx);

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
    a.need_d2 = (ctx->prm.mode == CVO_HIP_MODE_ACVO || !ctx->in_loop) ? 1 : 0;
    a.weight = ctx->prm.color_scale > 0.0f ? 1 : 0;   // the MATLAB object's weight: its own instantiation
    const bool no_pack = getenv("CVO_HIP_NO_PACK") != nullptr;   // (test switch, read when a plan is recorded: 8 + 4 byte kept entries)
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
        const bool no_cand = env_no_cand();
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
        const bool no_cand = env_no_cand();
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
        const bool no_cand = env_no_cand();
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
int drain_events(cvo_hip_ctx *ctx, int n_exec = -1, const DevState *fin = nullptr)
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
    if (const char *e = getenv("CVO_HIP_LIST_MARGIN")) {   // (test switch; 0 = rebuild every iteration)
        const double m = atof(e);
        if (m >= 0.0 && m <= 4.0) dp.list_margin = (float)m;
    }
    dp.async_xy = ctx->use_async ? 1 : 0;
    dp.async_self = ctx->use_async_self ? 1 : 0;
    // Head mode: a build is named a slot earlier than it is made and costs its launch 10 us; later is better
    // (0.7 / 0.85 / 0.9 / 0.95 of the margin gone: 10k x 10k 711 / 728 / 733 / 732 registrations/s, 14k 432 / 444 / 445 /
    // 444, 6k 694 / 694 / 706 / 705, 3k 788 / 794 / 792 / 792; profiles/r03_ab.txt 18)
    if (ctx->use_async && ctx->lone && ctx->allow_head && !multi_rank(ctx)) dp.build_at = 0.9f;
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
    ctx->merge_twist = ctx->allow_merge && !multi_rank(ctx);
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
TLaunch mk_launch(int kernel, int q, unsigned gx, unsigned gz, unsigned smem = 0)
{
    TLaunch l{};
    l.kernel = kernel; l.q = q; l.gx = gx; l.gz = gz; l.smem = smem;
    return l;
}

long long filter_items(const FilterArgs &f) { return (long long)f.gx * f.gy; }

// One registration with the launches to itself: the flow side of an iteration is merged into
// as few launches as its scheme allows (what enqueue_flow does for eager launches):
//   flow pass + xy build + both self passes + both self builds      -> kt_flow_build6
//   flow pass + xy build + xx / yy filters (self passes afterwards) -> kt_flow_build3, kt_self2
//   flow pass + xy build                                            -> kt_flow_build
//   synchronous lists                                               -> kt_filter(_group), kt_process, kt_self2
// Head mode (cvo_kernels.hip "the head"), where the scheme allows it -- asynchronous builds, step pass with the
// twist in front, one rank: the post-step launch is gone; its argument block rides in the flow launch's entry
// (op[q].ps), every flow / self block runs it as its head.
bool plan_lone(const std::vector<RecOp> &ops, Slot &slot, std::vector<TLaunch> &plan, const bool allow_head, bool *head_mode)
{
    plan.clear();
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
        plan.push_back(mk_launch(head ? TK_HFLOW_BUILD6 : (six ? TK_FLOW_BUILD6 : TK_FLOW_BUILD3), q,
                                 (unsigned)((six ? 3 : 1) * o.np + o.n0 + o.n1 + o.n2), 1, head ? smem_head(jt) : smem_of(jt)));
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
            plan.push_back(mk_launch(head ? TK_HFLOW_BUILD : TK_FLOW_BUILD, q, (unsigned)(o.np + o.n0), 1,
                                     head ? smem_head(build.jt) : smem_of(build.jt)));
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
                const long long cap = std::max<long long>(64, fbmax / (2 * zdim));
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

bool same_plan(const std::vector<TLaunch> &a, const std::vector<TLaunch> &b)
{
    return a.size() == b.size() && (a.empty() || std::memcmp(a.data(), b.data(), a.size() * sizeof(TLaunch)) == 0);
}

void launch_plan_eager(const Slot *tab, const std::vector<TLaunch> &plan, int iterations, hipStream_t s)
{
    for (int k = 0; k < iterations; ++k)
        for (const TLaunch &l : plan) launch_table(tab, l, s, nullptr, nullptr, k & 1);
}

// kBatch iterations of `plan` on table `tab`: through a cached graph when allowed, else eagerly.
int run_plan(const Slot *tab, PlanCache &cache, const std::vector<TLaunch> &plan, hipStream_t s, bool use_graph,
             int iterations)
{
    if (!use_graph || cache.fails >= 64) {
        launch_plan_eager(tab, plan, iterations, s);
        return hipGetLastError() == hipSuccess ? CVO_HIP_OK : CVO_HIP_ERR_HIP;
    }
    PlanGraph *hit = nullptr;
    for (auto &g : cache.graphs)
        if (g.iterations == iterations && same_plan(g.plan, plan)) { hit = &g; break; }
    if (hit) ++cache.hits;
    if (!hit) {
        // The capture window needs the library's lock exclusively (cvo_lock.h).  Not getting it within its
        // millisecond -- other host threads are inside their own entry points -- is neither a capture nor a
        // failed one: this batch goes out eagerly, nothing is counted, the next batch tries again.
        cvo_lock::Capture alone;   // (held: no other thread of this library is inside the runtime)
        if (!alone.ok) {
            launch_plan_eager(tab, plan, iterations, s);
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
        g.iterations = iterations;
        // A capture can be spoilt from outside (another thread's HIP work: cvo_lock.h).  Nothing
        // has been launched then: the batch goes out eagerly and the next one tries again.
        hipError_t e = hipErrorUnknown;
        if (hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed) == hipSuccess) {
            launch_plan_eager(tab, plan, iterations, s);
            e = hipStreamEndCapture(s, &g.graph);
        }
        if (e != hipSuccess || !g.graph || hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0) != hipSuccess) {
            if (g.graph) (void)hipGraphDestroy(g.graph);
            (void)hipGetLastError();
            ++cache.fails;
            launch_plan_eager(tab, plan, iterations, s);
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
    if (!plan_lone(ops, slot, ctx->plan, ctx->allow_head, &ctx->head_mode))
        return fail(ctx, CVO_HIP_ERR_INVALID, "launch plan does not fit the argument table");
    if (ctx->table.sync(&slot, loop_stream(ctx)) != 0) return fail(ctx, CVO_HIP_ERR_HIP, "argument table upload failed");
    return CVO_HIP_OK;
}

// Launch one batch of kBatch iterations: through the context's table (graph or eager table
// launches); profiling and the stream-level all-reduces (RCCL, caller's hook) keep the
// classic by-value launches -- they need their own launches / host calls in between.
int launch_batch(cvo_hip_ctx *ctx, int tag0, int trace_cap)
{
    if (ctx->profiling || host_reduce(ctx)) {
        const int rc = enqueue_iterations(ctx, kBatch, tag0, trace_cap);
        if (!rc) ctx->warm = true;
        return rc;
    }
    const int rc = run_plan(ctx->table.dev, ctx->plans, ctx->plan, loop_stream(ctx), ctx->use_graphs, kBatch);
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

}   // namespace

// ---------------------------------------------------------------------------
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
    if (getenv("CVO_HIP_NO_HEAD")) ctx->allow_head = false;
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
    *ctx->done_mirror = 0;
    *ctx->progress_mirror = 0;
    // Stream capture is a process-wide affair in this runtime (cvo_lock.h): the library's own
    // entry points keep out of each other's captures, but HIP work of OTHER code in the process
    // (torch on another thread, say) cannot be kept out and would fail with "previous error
    // during capture".  So batches are captured into hipGraphs by default only on a stream the
    // library created itself; with a caller-supplied stream the caller opts in
    // (cvo_hip_set_graph_capture, or CVO_HIP_GRAPH=1) once it knows no other thread of the
    // process uses HIP while an align() is being set up.  CVO_HIP_NO_GRAPH=1 forbids captures.
    ctx->use_graphs = ctx->own_stream || getenv("CVO_HIP_GRAPH") != nullptr;
    if (env_no_graph()) ctx->use_graphs = false;
    if (getenv("CVO_HIP_NO_MERGE")) ctx->allow_merge = false;
    if (getenv("CVO_HIP_NO_ASYNC")) ctx->allow_async = ctx->allow_async_self = false;
    if (const char *e = getenv("CVO_HIP_PROC_BLOCKS")) {   // list-kernel blocks of a lone registration
        const int v = atoi(e);
        if (v == 64 || v == 128 || v == 256 || v == 512 || v == 1024) {
            ctx->proc_blocks = ctx->proc_blocks_default = v;
            ctx->proc_blocks_forced = true;
        }
    }
    if (getenv("CVO_HIP_POST_DEBUG")) {
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
                    ctx->part_yy.p, ctx->part_step.p, (void *)ctx->trace_dev, ctx->kept_cnt.p, ctx->pos_bt.p, ctx->cand[0].p, ctx->cand[1].p, ctx->cand[2].p, ctx->cand_xyb.p, ctx->cand_cnt_xyb.p, ctx->cand_sfb[0].p, ctx->cand_sfb[1].p, ctx->cand_cnt_sfb[0].p, ctx->cand_cnt_sfb[1].p, ctx->cand_cnt[0].p,
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

int cvo_hip_set_fixed(cvo_hip_ctx *ctx, const float *xyz, const float *feat, int n, int layout)
{
    cvo_lock::Api api_guard;
    if (!ctx) return CVO_HIP_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return upload_cloud(ctx, ctx->fixed, xyz, feat, n, layout);
}

int cvo_hip_set_moving(cvo_hip_ctx *ctx, const float *xyz, const float *feat, int m, int layout)
{
    cvo_lock::Api api_guard;
    if (!ctx) return CVO_HIP_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->have_tf = false;
    return upload_cloud(ctx, ctx->moving, xyz, feat, m, layout);
}

int cvo_hip_set_fixed_device(cvo_hip_ctx *ctx, const float *d_xyz, const float *d_feat, int n, int layout)
{
    cvo_lock::Api api_guard;
    if (!ctx) return CVO_HIP_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return upload_cloud(ctx, ctx->fixed, d_xyz, d_feat, n, layout, true);
}

int cvo_hip_set_moving_device(cvo_hip_ctx *ctx, const float *d_xyz, const float *d_feat, int m, int layout)
{
    cvo_lock::Api api_guard;
    if (!ctx) return CVO_HIP_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->have_tf = false;
    return upload_cloud(ctx, ctx->moving, d_xyz, d_feat, m, layout, true);
}

int cvo_hip_swap_moving_to_fixed(cvo_hip_ctx *ctx)
{
    cvo_lock::Api api_guard;
    if (!ctx) return CVO_HIP_ERR_INVALID;
    std::swap(ctx->fixed, ctx->moving);
    ctx->moving.n = 0;
    ctx->moving.np = 0;
    ctx->have_tf = false;
    return CVO_HIP_OK;
}

int cvo_hip_set_pcd_many(cvo_hip_ctx *const *ctxs, const float *const *fixed_xyz, const float *const *fixed_feat,
                         const int *n_fixed, const float *const *moving_xyz, const float *const *moving_feat,
                         const int *n_moving, int feat_layout, int count)
{
    cvo_lock::Api api_guard;
    if (count < 0 || (count > 0 && (!ctxs || !moving_xyz || !moving_feat || !n_moving))) return CVO_HIP_ERR_INVALID;
    if (count == 0) return CVO_HIP_OK;
    if (fixed_xyz && (!fixed_feat || !n_fixed)) return CVO_HIP_ERR_INVALID;
    for (int k = 0; k < count; ++k)
        if (!ctxs[k] || ctxs[k]->device != ctxs[0]->device) return CVO_HIP_ERR_INVALID;
    cvo_hip_ctx *c0 = ctxs[0];
    HIP_TRY(c0, hipSetDevice(c0->device));
    Handover *ho = handover_of(c0->device);
    if (!ho) return fail(c0, CVO_HIP_ERR_INVALID, "device index out of range");
    std::lock_guard<std::mutex> lock(ho->mu);
    if (!ho->s) {
        HIP_TRY(c0, hipStreamCreateWithFlags(&ho->s, hipStreamNonBlocking));
        for (auto &e : ho->ev) HIP_TRY(c0, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    // what goes through the one launch, what goes the long way
    struct Item { cvo_hip_ctx *ctx; Cloud *c; const float *xyz, *feat; int n; size_t off_xyz, off_feat; };
    std::vector<Item> small;
    size_t raw_need = 0;
    for (int k = 0; k < count; ++k) {
        cvo_hip_ctx *ctx = ctxs[k];
        for (int which = 0; which < 2; ++which) {
            const float *xyz = which == 0 ? (fixed_xyz ? fixed_xyz[k] : nullptr) : moving_xyz[k];
            const float *feat = which == 0 ? (fixed_xyz ? fixed_feat[k] : nullptr) : moving_feat[k];
            if (which == 0 && !xyz) continue;   // (the fixed cloud stays what it is)
            const int n = which == 0 ? n_fixed[k] : n_moving[k];
            Cloud &c = which == 0 ? ctx->fixed : ctx->moving;
            if (which == 1) ctx->have_tf = false;
            if (n > CLOUD_ONE_MAX || n <= 0) {
                const int rc = upload_cloud(ctx, c, xyz, feat, n, feat_layout);
                if (rc) return rc;
                continue;
            }
            const int rc = cloud_reserve(ctx, c, xyz, feat, n, feat_layout);
            if (rc) return rc;
            Item it{ctx, &c, xyz, feat, n, 0, 0};
            const size_t bx = ((size_t)n * 12 + 255) & ~(size_t)255, bf = ((size_t)n * 20 + 255) & ~(size_t)255;
            it.off_xyz = raw_need; it.off_feat = raw_need + bx;   // (the staging arena mirrors the device arena)
            raw_need += bx + bf;
            small.push_back(it);
        }
    }
    if (small.empty()) return CVO_HIP_OK;
    // the arenas are the previous batch's until its last event has completed
    if (ho->last) HIP_TRY(c0, hipEventSynchronize(ho->last));
    if (raw_need > ho->raw_bytes) {
        if (ho->raw) HIP_TRY(c0, hipFree(ho->raw));
        if (ho->stage) (void)hipHostFree(ho->stage);
        ho->raw = nullptr; ho->stage = nullptr; ho->raw_bytes = 0;
        const size_t want = raw_need + raw_need / 4;
        if (hipMalloc((void **)&ho->raw, want) != hipSuccess || hipHostMalloc((void **)&ho->stage, want, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            return fail(c0, CVO_HIP_ERR_NOMEM, "hand-over arena allocation failed");
        }
        ho->raw_bytes = want;
    }
    if ((int)small.size() > ho->jobs_cap) {
        if (ho->jobs_pin) (void)hipHostFree(ho->jobs_pin);
        if (ho->jobs_dev) (void)hipFree(ho->jobs_dev);
        ho->jobs_pin = nullptr; ho->jobs_dev = nullptr; ho->jobs_cap = 0;
        const int want = (int)small.size() * 2;
        if (hipHostMalloc((void **)&ho->jobs_pin, (size_t)want * sizeof(CloudJob), hipHostMallocDefault) != hipSuccess ||
            hipMalloc((void **)&ho->jobs_dev, (size_t)want * sizeof(CloudJob)) != hipSuccess) {
            (void)hipGetLastError();
            return fail(c0, CVO_HIP_ERR_NOMEM, "hand-over job table allocation failed");
        }
        ho->jobs_cap = want;
    }
    for (size_t q = 0; q < small.size(); ++q) {
        const Item &it = small[q];
        CloudJob jb{};
        jb.xyz = (const float *)(ho->raw + it.off_xyz); jb.feat = (const float *)(ho->raw + it.off_feat);
        jb.n = it.n; jb.colmajor = feat_layout == CVO_HIP_FEAT_COLMAJOR ? 1 : 0;
        jb.np = it.c->np; jb.pad_axis = it.c->pad_axis;
        jb.pos = it.c->pos; jb.feat8 = it.c->feat; jb.seg = it.c->seg;
        void *bbox_d = nullptr;
        HIP_TRY(it.ctx, hipHostGetDevicePointer(&bbox_d, it.c->bbox_pin, 0));
        jb.bbox_out = (float *)bbox_d;
        ho->jobs_pin[q] = jb;
    }
    HIP_TRY(c0, hipMemcpyAsync(ho->jobs_dev, ho->jobs_pin, small.size() * sizeof(CloudJob), hipMemcpyHostToDevice, ho->s));
    // The batch goes out in a few pieces -- the caller's arrays into the staging arena (a few host threads, a share
    // of a piece's clouds each: one thread moves ~10 GB/s, 128 clouds of 10k points are 41 MB), one transfer, one
    // launch, one event per piece -- so that the transfer of a piece runs while the next one is staged, and the
    // registrations of the first contexts can begin while the last clouds are still on their way (a cloud waits
    // for the event of ITS piece, when the next compute entry point of its context needs it).
    constexpr int kPieces = 4;
    const size_t per_piece = std::max<size_t>(raw_need / kPieces + 1, (size_t)4 << 20);
    std::vector<size_t> piece_end;   // index past the last cloud of each piece
    {
        size_t start_off = 0;
        for (size_t q = 0; q < small.size(); ++q) {
            const size_t end_off = small[q].off_feat + (((size_t)small[q].n * 20 + 255) & ~(size_t)255);
            if (end_off - start_off >= per_piece || q + 1 == small.size()) { piece_end.push_back(q + 1); start_off = end_off; }
        }
    }
    static const int max_threads = [] {   // (half the host's cores, at most 8: staging is memory-bound well before that)
        const int hw = (int)std::thread::hardware_concurrency();
        return std::min(8, std::max(2, hw / 2));
    }();
    const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)max_threads, raw_need / ((size_t)2 << 20)));
    std::vector<std::atomic<int>> staged(piece_end.size());
    for (auto &a : staged) a.store(0);
    char *stage = ho->stage;
    auto work = [&](int t) {
        size_t lo = 0;
        for (size_t pc = 0; pc < piece_end.size(); ++pc) {
            for (size_t q = lo + (size_t)t; q < piece_end[pc]; q += (size_t)nt) {
                const Item &it = small[q];
                std::memcpy(stage + it.off_xyz, it.xyz, (size_t)it.n * 12);
                std::memcpy(stage + it.off_feat, it.feat, (size_t)it.n * 20);
            }
            staged[pc].fetch_add(1, std::memory_order_release);
            lo = piece_end[pc];
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work, t);
    struct Join { std::vector<std::thread> &p; ~Join() { for (auto &th : p) if (th.joinable()) th.join(); } } join_guard{pool};
    size_t lo = 0;
    int rc_out = CVO_HIP_OK;
    for (size_t pc = 0; pc < piece_end.size() && rc_out == CVO_HIP_OK; ++pc) {
        // (this thread's share of the piece, then the others')
        for (size_t q = lo; q < piece_end[pc]; q += (size_t)nt) {
            const Item &it = small[q];
            std::memcpy(stage + it.off_xyz, it.xyz, (size_t)it.n * 12);
            std::memcpy(stage + it.off_feat, it.feat, (size_t)it.n * 20);
        }
        while (staged[pc].load(std::memory_order_acquire) < nt - 1) __builtin_ia32_pause();
        const size_t hi = piece_end[pc];
        const size_t b0 = small[lo].off_xyz, b1 = small[hi - 1].off_feat + (((size_t)small[hi - 1].n * 20 + 255) & ~(size_t)255);
        int nmax = 0;
        for (size_t q = lo; q < hi; ++q) nmax = std::max(nmax, small[q].n);
        hipEvent_t ev = ho->ev[ho->next_ev];
        ho->next_ev = (ho->next_ev + 1) % 16;
        if (hipMemcpyAsync(ho->raw + b0, ho->stage + b0, b1 - b0, hipMemcpyHostToDevice, ho->s) != hipSuccess ||
            cloud_prepare_many(ho->jobs_dev + lo, (int)(hi - lo), nmax, ho->s) != hipSuccess ||
            hipEventRecord(ev, ho->s) != hipSuccess) {
            (void)hipGetLastError();
            rc_out = fail(c0, CVO_HIP_ERR_HIP, "batched hand-over: transfer or launch failed");
            break;
        }
        ho->last = ev;
        for (size_t q = lo; q < hi; ++q) { small[q].c->wait_ev = ev; small[q].c->pending = true; }
        lo = hi;
    }
    return rc_out;
}

int cvo_hip_get_device_cloud(cvo_hip_ctx *ctx, int which, float *pos4, float *feat8, float *seg4, int *rows, int *points)
{
    cvo_lock::Api api_guard;
    if (!ctx || (which != 0 && which != 1)) return CVO_HIP_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    Cloud &c = which == 0 ? ctx->fixed : ctx->moving;
    const int rc = cloud_ready(ctx, c);
    if (rc) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (rows) *rows = c.np;
    if (points) *points = c.n;
    if (c.np <= 0) return CVO_HIP_OK;
    if (pos4) HIP_TRY(ctx, hipMemcpy(pos4, c.pos, (size_t)c.np * sizeof(float4), hipMemcpyDeviceToHost));
    if (feat8) HIP_TRY(ctx, hipMemcpy(feat8, c.feat, (size_t)c.np * FEAT_STRIDE * sizeof(float), hipMemcpyDeviceToHost));
    if (seg4) HIP_TRY(ctx, hipMemcpy(seg4, c.seg, (size_t)((c.np + SEG - 1) / SEG) * sizeof(float4), hipMemcpyDeviceToHost));
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
    double secs = 5.0;
    if (const char *e = getenv("CVO_HIP_MAILBOX_TIMEOUT_S")) { const double v = atof(e); if (v > 0.0) secs = v; }
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

// ---- align() as a resumable job, so that one host thread can keep many
// ---- registrations (one context + stream each) in flight: cvo_hip_align_many
namespace {

struct AlignJob {
    cvo_hip_ctx *ctx = nullptr;
    cvo_hip_state *s = nullptr;
    cvo_hip_trace *trace = nullptr;
    int trace_cap = 0;
    int *n_iter = nullptr;
    int enq = 0;            // iterations enqueued in this round
    int batches = 0;        // batches enqueued in this round
    int checked = 0;        // batches whose poll copy has been looked at
    int executed_base = 0;  // iterations completed before this round (after a list grew)
    int phase = 0;          // 0 enqueueing/polling, 1 waiting for the final state, 2 finished
    int rc = CVO_HIP_OK;
    bool in_group = false;  // runs in a fused group (on the group's stream and table)
    bool paced = false;     // cvo_hip_align only: the calling thread has nothing else to pump and may sit in the
                            // paced loop of job_pump (align_many's blocking fall-back must keep its round-robin going:
                            // the other jobs -- the peer ranks of a mailbox world among them -- run dry otherwise)
};

int job_begin(AlignJob &j)
{
    cvo_hip_ctx *ctx = j.ctx;
    cvo_hip_state *s = j.s;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    {
        const int rcm = mailboxes_usable(ctx);
        if (rcm) return rcm;
    }
    const cvo_hip_params &p = ctx->prm;
    if (p.mode == CVO_HIP_MODE_ACVO) {   // tail of acvo::set_pcd (ref src/adaptive_cvo.cpp:476-478)
        s->ell = p.ell_init;
        s->ell_max = p.ell_max_init;
    }
    *ctx->done_mirror = 0;
    *ctx->progress_mirror = 0;
    if (!j.trace) j.trace_cap = 0;
    if (j.trace_cap > p.max_iter) j.trace_cap = p.max_iter;
    if (j.trace_cap > ctx->trace_dev_cap) {
        if (ctx->trace_dev) HIP_TRY(ctx, hipFree(ctx->trace_dev));
        ctx->trace_dev = nullptr; ctx->trace_dev_cap = 0;
        HIP_TRY(ctx, hipMalloc((void **)&ctx->trace_dev, (size_t)j.trace_cap * sizeof(cvo_hip_trace)));
        ctx->trace_dev_cap = j.trace_cap;
    }
    if (j.trace_cap > 0)
        HIP_TRY(ctx, hipMemsetAsync(ctx->trace_dev, 0, (size_t)j.trace_cap * sizeof(cvo_hip_trace),
                                    loop_stream(ctx)));
    // initial device state
    DevState *h = &ctx->st_host[kPollSlots];
    std::memset(h, 0, sizeof(*h));
    std::memcpy(h->R, s->R, sizeof(h->R));
    std::memcpy(h->T, s->T, sizeof(h->T));
    h->ell = s->ell;
    h->ell_max = s->ell_max;
    h->iter = s->iter;
    {
        const int rcg = fill_filter_geometry(ctx, h);
        if (rcg) return rcg;
    }
    if (p.max_iter <= 0) h->done = DONE_MAX_ITER;
    // (everything but the mailbox sequence number, which lives as long as the context)
    HIP_TRY(ctx, hipMemcpyAsync(ctx->st, h, DEVSTATE_INIT_BYTES, hipMemcpyHostToDevice, loop_stream(ctx)));
    // small clouds (the ~3k-point clouds of the reference's front end): 2048 waves do
    // (measured 3k x 3k: 2.11 ms with 512 blocks, 2.19 with 1024; 10k x 10k the other way round)
    const bool small_pair = (double)ctx->fixed.n * (double)ctx->moving.n <= 2.5e7;
    if (!ctx->proc_blocks_forced)
        ctx->proc_blocks = ctx->proc_blocks_default = small_pair ? PROC_BLOCKS / 2 : PROC_BLOCKS;
    // (use_async_self below; from ~20k x 20k on a build is too long to hide beside one flow pass)
    // (the MATLAB weight exists as a classic k_process launch only)
    ctx->use_async = ctx->allow_async && !ctx->crowded && !ctx->profiling && !multi_rank(ctx) &&
                     !(ctx->prm.color_scale > 0.0f) &&
                     (double)ctx->fixed.n * (double)ctx->moving.n <= 2.0e8;
    ctx->use_async_self = ctx->use_async && ctx->allow_async_self && ctx->lone &&
                          ctx->prm.mode == CVO_HIP_MODE_ACVO;
    // acvo with everything in one launch (flow pass + both self passes + the builds = 3 x blocks + filter blocks):
    // a quarter of the blocks per pass do (measured one registration at a time, 1024 / 512 / 256 / 128 blocks per
    // pass: 10k x 10k 390 / 461 / 499 / - registrations/s with the post-step launch, - / 417 / 537 / 499 in head
    // mode; 3k x 3k - / 543 / 619 / - and - / 473 / 703 / 749 -- profiles/r03_ab.txt)
    const double npairs = (double)ctx->fixed.n * (double)ctx->moving.n;
    if (ctx->use_async_self && !ctx->proc_blocks_forced)   // (6k x 6k: 128 / 256 blocks 750 / 700; 14k x 14k 510 / 568)
        ctx->proc_blocks = ctx->proc_blocks_default = npairs <= 6.0e7 ? PROC_BLOCKS / 8 : PROC_BLOCKS / 4;
    // cvo in head mode: every block of the flow launch starts with the head, and with the candidate records the
    // pass behind it is short -- fewer, longer blocks (us per iteration with 256 / 512 / 1024 blocks per pass:
    // 2k x 2k 18.4 / 18.9 / 20.4, 4.5k 19.3 / 19.5 / 22.3, 6k 21.4 / 20.9 / 23.0, 8k 27.2 / 24.7 / 26.6,
    // 10k 31.6 / 26.5 / 26.7, 14k 34.8 / 28.5 / 27.6 -- profiles/r03_ab.txt 17)
    if (!ctx->proc_blocks_forced && !ctx->use_async_self && ctx->use_async && ctx->lone && ctx->allow_head &&
        ctx->prm.mode == CVO_HIP_MODE_CVO)
        ctx->proc_blocks = ctx->proc_blocks_default =
            npairs <= 2.5e7 ? PROC_BLOCKS / 4 : (npairs <= 1.5e8 ? PROC_BLOCKS / 2 : PROC_BLOCKS);
    launch_prepare(ctx->st, loop_params(ctx), loop_stream(ctx));
    HIP_TRY(ctx, hipGetLastError());
    ctx->have_tf = true;
    const int prc = prepare_buffers(ctx);
    if (prc) return prc;
    // (a member of a fused group is planned by the group: its slot is one of many)
    ctx->head_mode = false;   // (set again by prepare_lone_plan if this align() runs a head-mode plan)
    if (!j.in_group && !ctx->profiling && !host_reduce(ctx)) {
        const int rc2 = prepare_lone_plan(ctx, j.trace_cap);
        if (rc2) return rc2;
    }
    j.enq = j.batches = j.checked = 0;
    j.executed_base = 0;
    j.phase = p.max_iter <= 0 ? 1 : 0;
    if (j.phase == 1) {
        HIP_TRY(ctx, hipMemcpyAsync(&ctx->st_host[0], ctx->st, sizeof(DevState), hipMemcpyDeviceToHost,
                                    loop_stream(ctx)));
        HIP_TRY(ctx, hipEventRecord(ctx->poll_ev[0], loop_stream(ctx)));
    }
    return CVO_HIP_OK;
}

// ref src/cvo.cpp:413-415 and the trace / state hand-back
int job_finish(AlignJob &j)
{
    cvo_hip_ctx *ctx = j.ctx;
    cvo_hip_state *s = j.s;
    const DevState &f = ctx->st_host[0];
    ctx->have_tf = false;   // the low-level entry points need their own transform_pcd()
    if (f.done == DONE_COMM_ERROR) {
        // The rank that timed out has advanced its sequence number, a peer that left early or never launched
        // has not, and a late store may still land in a slot of the same generation: from here on every
        // exchange of this world would mismatch or time out.  The mailboxes are unusable until every rank
        // has called cvo_hip_mailbox_create / _connect again; sharded calls are refused until then.
        ctx->mail_broken = true;
        return fail(ctx, CVO_HIP_ERR_COMM, "mailbox all-reduce timed out: a peer rank never delivered its partial sums "
                                           "(the mailboxes must be created and connected again on every rank)");
    }
    if (f.done == RUNNING || f.done == NEED_BIGGER_LIST)
        return fail(ctx, CVO_HIP_ERR_INVALID, "align loop ended without a verdict");
    const int executed = f.n_exec;
    if (j.trace_cap > 0 && executed > 0)
        HIP_TRY(ctx, hipMemcpy(j.trace, ctx->trace_dev,
                               (size_t)std::min(executed, j.trace_cap) * sizeof(cvo_hip_trace),
                               hipMemcpyDeviceToHost));
    // accumulate the transform computed at the TOP of the last executed
    // iteration, then refresh `transform` from the final R,T
    if (executed > 0) cvo_math::tf_to_mat4(f.used_Rt, f.used_t, s->transform);
    std::memcpy(s->R, f.R, sizeof(s->R));
    std::memcpy(s->T, f.T, sizeof(s->T));
    s->ell = f.ell;
    s->ell_max = f.ell_max;
    s->iter = f.iter;
    std::memcpy(s->prev_transform, s->transform, sizeof(s->transform));
    cvo_math::mat4_mul(s->accum_transform, s->transform, s->accum_transform);
    float Rt[9], t[3];
    cvo_math::inverse_tf(s->R, s->T, Rt, t);
    cvo_math::tf_to_mat4(Rt, t, s->transform);
    if (j.n_iter) *j.n_iter = executed;
    if (ctx->profiling) return drain_events(ctx, executed, &f);
    return CVO_HIP_OK;
}

// Advance a job without (block = false) or with (block = true) waiting on the
// GPU.  Returns 1 when the job has finished (j.rc holds its status), else 0.
// At most two batches are in flight; `done` is looked at one batch behind; a
// list that overflows parks the loop with NEED_BIGGER_LIST before any state was
// changed: enlarge it and resume from the same iteration.
int job_pump(AlignJob &j, bool block)
{
    cvo_hip_ctx *ctx = j.ctx;
    if (j.phase == 2) return 1;
    auto finish_with = [&](int rc) { j.rc = rc; j.phase = 2; return 1; };
    if (hipSetDevice(ctx->device) != hipSuccess) return finish_with(CVO_HIP_ERR_HIP);
    // Blocking caller, launches that need no host work in between: PACED mode.  The post-step
    // kernel mirrors its slot count and `done` into pinned memory; this thread watches the two
    // words and enqueues the next batch when the running one has finished -- not a whole batch
    // ahead, which left a registration that converged with (on average) a batch and a half of
    // queued launches to return one by one (~85 us of 1.7 ms, and the next frame's hand-over
    // queues behind them).  The ~10 us the stream idles between two batches cost less than that
    // (CVO_HIP_PACE_LEAD = slots of overlap, 0 / 1 / 2 / 3: 644 / 619 / 627 / 620 registrations/s at
    // 10k x 10k, event-paced two batches ahead: 604).
    if (j.phase == 0 && block && j.paced && !host_reduce(ctx) && !ctx->profiling) {
        const int limit = (ctx->use_async ? 3 : 1) * ctx->prm.max_iter + 4 * kBatch;
        unsigned spins = 0;
        int idle_seen = 0;
        for (;;) {
            if (*(volatile int32_t *)ctx->done_mirror != RUNNING) break;
            const int slots = *(volatile int32_t *)ctx->progress_mirror;
            // (head mode without a flush: the post-step part of a batch's last slot runs in the head of the NEXT
            // batch's first launch, so the next batch must be on its way before the running one ends -- it goes
            // out when the running batch is down to its last slots; the GPU never idles between batches, and
            // a registration that stops in those last slots leaves one batch of launches that return at once)
            const int lead = ctx->head_mode ? 2 : 0;
            if (j.enq - slots <= lead) {
                if (j.enq >= limit) break;   // cannot happen
                const int rc = launch_batch(ctx, j.executed_base + j.enq, j.trace_cap);
                if (rc) return finish_with(rc);
                j.enq += kBatch;
                ++j.batches;
                spins = 0;
                idle_seen = 0;
            } else {
                __builtin_ia32_pause();
                // The two words only move while the queued kernels run.  A fault, a stream in an error state or a
                // post kernel that never ran would leave this thread spinning for ever: now and then ask the
                // stream itself (a batch lasts ~0.25 ms; 2^14 pauses are about that long).
                if ((++spins & 0x3fffu) == 0u) {
                    const hipError_t q = hipStreamQuery(loop_stream(ctx));
                    if (q != hipSuccess && q != hipErrorNotReady)
                        return finish_with(fail(ctx, CVO_HIP_ERR_HIP, "the stream of the align loop reports an error"));
                    // idle, yet the batch has not reported all its slots and nothing stopped: seen twice in a row
                    // (the mirrors are written before a kernel ends, so once is already conclusive; twice is cheap)
                    if (q == hipSuccess && *(volatile int32_t *)ctx->done_mirror == RUNNING &&
                        *(volatile int32_t *)ctx->progress_mirror == slots) {
                        if (++idle_seen >= 2)
                            return finish_with(fail(ctx, CVO_HIP_ERR_HIP, "the align loop's stream went idle without progress"));
                    } else {
                        idle_seen = 0;
                    }
                }
            }
        }
        if (hipMemcpyAsync(&ctx->st_host[0], ctx->st, sizeof(DevState), hipMemcpyDeviceToHost,
                           ctx->stream) != hipSuccess ||
            hipEventRecord(ctx->poll_ev[0], ctx->stream) != hipSuccess)
            return finish_with(fail(ctx, CVO_HIP_ERR_HIP, "state copy failed"));
        j.phase = 1;
    }
    if (j.phase == 0) {
        bool stop = false;
        while (j.batches - j.checked < 2) {   // keep two batches queued
            int rc = launch_batch(ctx, j.executed_base + j.enq, j.trace_cap);
            if (rc) return finish_with(rc);
            j.enq += kBatch;
            const int slot = j.batches % kPollSlots;
            // Single rank: the post kernels mirror `done` into pinned memory, an event
            // per batch is all the polling needs.  With ranks to stay in step with, the
            // state is copied in stream order instead: every rank must see `done` at the
            // same batch, or their all-reduce counts would differ.
            if (host_reduce(ctx) &&
                hipMemcpyAsync(&ctx->st_host[slot], ctx->st, DEVSTATE_HEAD_BYTES, hipMemcpyDeviceToHost,
                               ctx->stream) != hipSuccess)
                return finish_with(fail(ctx, CVO_HIP_ERR_HIP, "poll copy failed"));
            if (hipEventRecord(ctx->poll_ev[slot], ctx->stream) != hipSuccess)
                return finish_with(fail(ctx, CVO_HIP_ERR_HIP, "poll event failed"));
            ++j.batches;
        }
        // look at the oldest batch not yet examined
        const int slot = j.checked % kPollSlots;
        hipError_t q = block ? hipEventSynchronize(ctx->poll_ev[slot]) : hipEventQuery(ctx->poll_ev[slot]);
        if (q == hipErrorNotReady) return 0;
        if (q != hipSuccess) return finish_with(fail(ctx, CVO_HIP_ERR_HIP, "poll event failed"));
        ++j.checked;
        // (mailboxes: a rank that sees `done` one batch after its peers only queues kernels that
        // return at their first load -- no exchange is left half done)
        if (host_reduce(ctx) ? ctx->st_host[slot].done != RUNNING
                             : *(volatile int32_t *)ctx->done_mirror != RUNNING)
            stop = true;
        // (slots, not iterations: asynchronous builds add a stall slot now and then)
        if (j.enq >= (ctx->use_async ? 3 : 1) * ctx->prm.max_iter + 4 * kBatch) stop = true;   // cannot happen
        if (!stop) return 0;
        // everything still queued either runs or returns at once; fetch the full state
        if (hipMemcpyAsync(&ctx->st_host[0], ctx->st, sizeof(DevState), hipMemcpyDeviceToHost,
                           ctx->stream) != hipSuccess ||
            hipEventRecord(ctx->poll_ev[0], ctx->stream) != hipSuccess)
            return finish_with(fail(ctx, CVO_HIP_ERR_HIP, "state copy failed"));
        j.phase = 1;
    }
    // phase 1: wait for the final state
    hipError_t q = block ? hipEventSynchronize(ctx->poll_ev[0]) : hipEventQuery(ctx->poll_ev[0]);
    if (q == hipErrorNotReady) return 0;
    if (q != hipSuccess) return finish_with(fail(ctx, CVO_HIP_ERR_HIP, "state event failed"));
    const DevState &cur = ctx->st_host[0];
    if (cur.done != NEED_BIGGER_LIST) return finish_with(job_finish(j));
    // grow the overflowed list(s) and resume from the parked iteration
    int rc = CVO_HIP_OK;
    if (ctx->profiling) rc = drain_events(ctx, cur.k + 1, &cur);
    j.executed_base = cur.k;
    for (int l = 0; l < LIST_N && !rc; ++l)
        if (cur.ovf[0][l] | cur.ovf[1][l]) {
            uint32_t worst = 0;   // appends are spread evenly: scale by the fullest sub-list
            for (int qq = 0; qq < NSUB; ++qq) worst = std::max(worst, cur.sub[l][qq]);
            const double grown =
                std::min(4.0e9, std::max((double)worst * NSUB, (double)ctx->lists[l].cap) * 1.5 + 1024.0);
            rc = ensure_list(ctx, l, 0, 0, grown);
        }
    for (int q = 0; q < 3 && !rc; ++q) {   // the two buffers of a list share one capacity
        const int la = q == 0 ? LIST_XY : (q == 1 ? LIST_XX : LIST_YY), lb = q == 0 ? LIST_XYB : (q == 1 ? LIST_XXB : LIST_YYB);
        if (!ctx->lists[la].cap && !ctx->lists[lb].cap) continue;
        const double both = (double)std::max(ctx->lists[la].cap, ctx->lists[lb].cap);
        rc = ensure_list(ctx, la, 0, 0, both);
        if (!rc && ctx->lists[lb].cap) rc = ensure_list(ctx, lb, 0, 0, both);
    }
    if (rc) return finish_with(rc);
    int32_t zero = 0;
    if (hipMemcpyAsync(reinterpret_cast<char *>(ctx->st) + offsetof(DevState, done), &zero, sizeof(zero),
                       hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess)
        return finish_with(fail(ctx, CVO_HIP_ERR_HIP, "resume failed"));
    *ctx->done_mirror = 0;   // (the stream is idle: nothing can be writing it)
    *ctx->progress_mirror = 0;
    if (hipMemset(reinterpret_cast<char *>(ctx->st) + offsetof(DevState, n_slots), 0, sizeof(int32_t)) != hipSuccess ||
        hipDeviceSynchronize() != hipSuccess)
        return finish_with(fail(ctx, CVO_HIP_ERR_HIP, "resume failed"));
    launch_prepare(ctx->st, loop_params(ctx), ctx->stream);   // idempotent; re-zeroes the counters
    if (hipGetLastError() != hipSuccess) return finish_with(fail(ctx, CVO_HIP_ERR_HIP, "resume failed"));
    if (!ctx->profiling && !host_reduce(ctx)) {   // the lists moved: new arguments
        rc = prepare_lone_plan(ctx, j.trace_cap);
        if (rc) return finish_with(rc);
    }
    j.enq = j.batches = j.checked = 0;
    j.phase = 0;
    return 0;
}

// ---------------------------------------------------------------------------
// Fused mode: up to MAXG registrations advance through ONE sequence of launches
// (blockIdx.z = registration).  All members run the same launch sequence (same
// mode, single rank, no per-launch events) on the leader's stream; a member
// that has stopped keeps returning at its first load until it is dropped from
// the launches at the next poll.
bool fusable(const cvo_hip_ctx *c) { return !c->profiling && !multi_rank(c) && !(c->prm.color_scale > 0.0f); }

// Engine profiling (cvo_hip_engine_profiling): while it is on, the engines launch eagerly and every
// flow-pass launch (kt_process<PROC_FLOW>, the kernel with the largest share of a batched run)
// carries a HIP event pair; the sums are read with cvo_hip_get_engine_profile.
struct EngineProfile {
    std::mutex mu;
    bool on = false;
    double flow_ms = 0.0, flow_slots = 0.0;
    long long flow_launches = 0;
    // the launches one by one, in launch order per engine (cvo_hip_get_engine_flow_trace): duration, the time from
    // this launch's begin to the next flow launch's begin on the same stream (= one iteration of the engine;
    // 0 for the last of a drain), occupied slots
    std::vector<float> dur_us, period_us;
    std::vector<int> slots;
};
EngineProfile *engine_profile()
{
    static EngineProfile *p = new EngineProfile;
    return p;
}

// A fused group as a long-lived engine: a stream, a table of ENGINE_SLOTS slots and the batches
// captured for it, all of which outlive the cvo_hip_align_many call that uses them.
// Registrations enter a free slot and leave it when they stop -- by stream-ordered copies into
// the table, between two batches of iterations: nothing is drained, nothing is captured again
// (continuous batching).  Slots are kept packed at the low end; the launches serve
// zdim = 1, 2, 4, 8, 16, 24 or 32 slots, the list kernels getting more blocks per registration the
// fewer share the launch.  One host thread keeps several engines in flight: while one group
// sits in its single-block post kernels or between two kernels, the other one has the GPU.
struct Engine {
    int device = 0;
    hipStream_t s = nullptr;
    TableBuf tab;
    PlanCache plans;
    bool in_use = false;

    // state of the call in progress
    AlignJob *member[ENGINE_SLOTS] = {};
    std::vector<RecOp> ops[ENGINE_SLOTS];
    Slot slot[ENGINE_SLOTS];
    struct Retire { hipEvent_t ev = nullptr; std::vector<AlignJob *> jobs; };
    std::vector<Retire> retiring;          // their final state is on its way to the host
    hipEvent_t ev[4] = {};
    long long launched = 0, checked = 0;   // batches
    int zdim = 0;
    bool crowded = true, use_graph = true, dirty = true, failed = false;
    std::vector<TLaunch> plan;
    struct FlowEv { hipEvent_t a, b; int live; };
    std::vector<FlowEv> flow_ev;           // engine profiling: one pair per flow-pass launch
    // diagnostics (CVO_HIP_ENGINE_DEBUG)
    long long n_batches[5] = {}, n_replans = 0, n_inserts = 0, n_sends = 0;
    double t_replan = 0, t_insert = 0, t_launch = 0, t_finish = 0, t_wait = 0, t_collect = 0, t_idle_at = 0;
    static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

    int create(int dev)
    {
        device = dev;
        if (hipSetDevice(dev) != hipSuccess) return -1;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return -1;
        if (tab.init(ENGINE_SLOTS, s) != 0) return -1;
        for (auto &e : ev)
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return -1;
        return 0;
    }

    int live() const { int n = 0; for (AlignJob *j : member) n += j != nullptr; return n; }
    bool idle() const { return live() == 0 && retiring.empty() && launched == checked; }

    static int nblk_for(int z)
    {
        // blocks of a whole fused launch (1024 / 2048 / 4096 measured: 3 806 / 3 808 / 3 615 registrations/s at 64 pairs per
        // call, 4 398 / 4 386 / 4 326 at 256, profiles/r04_ab.txt 1); a registration gets 64, 128, 256, 512 or 1024 of them
        constexpr int budget = 2048;
        int nblk = 64;
        while (nblk < PROC_BLOCKS && nblk * 2 <= (budget + z / 2) / std::max(1, z)) nblk *= 2;
        return nblk;
    }

    void finish_job(AlignJob *j, int rc)
    {
        j->rc = rc;
        j->phase = 2;
        j->in_group = false;
        j->ctx->loop_stream = nullptr;
        j->ctx->crowded = false;
        j->ctx->lone = true;
        j->ctx->proc_blocks = j->ctx->proc_blocks_default;
    }

    void fail_all(const char *msg, std::deque<AlignJob *> &pending)
    {
        failed = true;
        (void)hipStreamSynchronize(s);
        for (AlignJob *&j : member)
            if (j) { finish_job(j, fail(j->ctx, CVO_HIP_ERR_HIP, msg)); j = nullptr; }
        for (auto &r : retiring) {
            for (AlignJob *j : r.jobs) finish_job(j, fail(j->ctx, CVO_HIP_ERR_HIP, msg));
            if (r.ev) (void)hipEventDestroy(r.ev);
        }
        retiring.clear();
        for (AlignJob *j : pending) { j->rc = fail(j->ctx, CVO_HIP_ERR_HIP, msg); j->phase = 2; }
        pending.clear();
        for (int z = 0; z < ENGINE_SLOTS; ++z) slot[z].active = 0;
        (void)tab.sync(slot, s, 0);
        (void)hipStreamSynchronize(s);
        launched = checked = 0;
    }

    // a job takes slot z: its align() begins (or resumes after its lists grew) on this stream
    int insert(AlignJob *j, int z)
    {
        cvo_hip_ctx *c = j->ctx;
        c->loop_stream = s;
        c->crowded = crowded;
        c->lone = false;
        j->in_group = true;
        int rc = CVO_HIP_OK;
        if (j->phase == 3) {   // resuming: the state is where the overflow parked it
            int32_t zero = 0;
            std::memcpy(&c->st_host[kPollSlots].done, &zero, sizeof(zero));
            if (hipMemcpyAsync(reinterpret_cast<char *>(c->st) + offsetof(DevState, done), &c->st_host[kPollSlots].done,
                               sizeof(zero), hipMemcpyHostToDevice, s) != hipSuccess)
                rc = fail(c, CVO_HIP_ERR_HIP, "resume failed");
            *c->done_mirror = 0;
            launch_prepare(c->st, loop_params(c), s);
            j->phase = 0;
        } else {
            rc = job_begin(*j);
        }
        if (rc) { finish_job(j, rc); return rc; }
        if (j->phase != 0) {   // max_iter <= 0: nothing to run; the state copy is already queued
            Retire r;
            if (hipEventCreateWithFlags(&r.ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(r.ev, s) != hipSuccess) {
                finish_job(j, fail(c, CVO_HIP_ERR_HIP, "event failed"));
                return CVO_HIP_ERR_HIP;
            }
            r.jobs.push_back(j);
            retiring.push_back(r);
            return CVO_HIP_OK;
        }
        member[z] = j;
        ops[z].clear();
        dirty = true;
        return CVO_HIP_OK;
    }

    // membership changed: pick zdim (1, 2, 4, 8, 16, 24, 32 >= the members), bring the members that sit
    // above it down into free slots (the others stay where they are: a slot that moves is a slot
    // that has to be sent again), (re)record what needs it, make the plan, send what changed
    int replan()
    {
        int n = 0;
        for (int z = 0; z < ENGINE_SLOTS; ++z) n += member[z] != nullptr;
        int zd = 1;
        while (zd < n) zd *= 2;
        if (n > 16 && n <= 24) zd = 24;   // (three engines sharing 64 registrations hold 21 or 22 each)
        for (int z = ENGINE_SLOTS - 1, hole = 0; z >= zd; --z) {
            if (!member[z]) continue;
            while (member[hole]) ++hole;
            member[hole] = member[z]; member[z] = nullptr;
            ops[hole].swap(ops[z]); ops[z].clear();
        }
        const bool regeom = zd != zdim;
        zdim = zd;
        const int nblk = nblk_for(zdim);
        constexpr int merge_max = 2;
        std::vector<const std::vector<RecOp> *> po;
        std::vector<Slot *> ps;
        for (int z = 0; z < ENGINE_SLOTS; ++z) {
            if (!member[z]) { slot[z].active = 0; continue; }
            cvo_hip_ctx *c = member[z]->ctx;
            if (regeom || ops[z].empty()) {
                c->proc_blocks = nblk;
                // k_step_twist pays for the saved launch with a prologue in every block:
                // a gain while launches are latency-bound, a loss once the GPU is full
                const bool allow = c->allow_merge;
                if (zdim > merge_max) c->allow_merge = false;
                const int rc = record_iteration(c, ops[z], 0);
                c->allow_merge = allow;
                if (rc) return rc;
            }
            std::memset(&slot[z], 0, sizeof(Slot));
            slot[z].active = 1;
            po.push_back(&ops[z]);
            ps.push_back(&slot[z]);
        }
        if (!plan_fused(po, ps, zdim, plan)) return CVO_HIP_ERR_INVALID;
        const int nq = po.empty() ? 0 : (int)po[0]->size();
        if (tab.sync(slot, s, nq) != 0) return CVO_HIP_ERR_HIP;
        ++n_replans;
        dirty = false;
        return CVO_HIP_OK;
    }

    // members whose loop has stopped leave their slots; their final state starts for the host
    void collect_stopped()
    {
        Retire r;
        for (int z = 0; z < ENGINE_SLOTS; ++z) {
            AlignJob *j = member[z];
            if (!j || *(volatile int32_t *)j->ctx->done_mirror == RUNNING) continue;
            if (hipMemcpyAsync(&j->ctx->st_host[0], j->ctx->st, sizeof(DevState), hipMemcpyDeviceToHost, s) != hipSuccess) {
                finish_job(j, fail(j->ctx, CVO_HIP_ERR_HIP, "state copy failed"));
            } else {
                r.jobs.push_back(j);
            }
            member[z] = nullptr;
            ops[z].clear();
            dirty = true;
        }
        if (r.jobs.empty()) return;
        if (hipEventCreateWithFlags(&r.ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(r.ev, s) != hipSuccess) {
            for (AlignJob *j : r.jobs) finish_job(j, fail(j->ctx, CVO_HIP_ERR_HIP, "event failed"));
            if (r.ev) (void)hipEventDestroy(r.ev);
            return;
        }
        retiring.push_back(r);
    }

    // final states that have arrived: hand the registration back, or -- a list overflowed --
    // enlarge it and queue the registration again (it resumes at the iteration it parked at)
    bool finish_arrived(std::deque<AlignJob *> &pending, bool block)
    {
        bool moved = false;
        while (!retiring.empty()) {
            Retire &r = retiring.front();
            const hipError_t q = block ? hipEventSynchronize(r.ev) : hipEventQuery(r.ev);
            if (q == hipErrorNotReady) break;
            block = false;
            for (AlignJob *j : r.jobs) {
                cvo_hip_ctx *c = j->ctx;
                const DevState &cur = c->st_host[0];
                if (q != hipSuccess) { finish_job(j, fail(c, CVO_HIP_ERR_HIP, "state event failed")); continue; }
                if (cur.done != NEED_BIGGER_LIST) { finish_job(j, job_finish(*j)); continue; }
                int rc = CVO_HIP_OK;
                for (int l = 0; l < LIST_N && !rc; ++l)
                    if (cur.ovf[0][l] | cur.ovf[1][l]) {
                        uint32_t worst = 0;
                        for (int qq = 0; qq < NSUB; ++qq) worst = std::max(worst, cur.sub[l][qq]);
                        const double grown = std::min(
                            4.0e9, std::max((double)worst * NSUB, (double)c->lists[l].cap) * 1.5 + 1024.0);
                        rc = ensure_list(c, l, 0, 0, grown);
                    }
                for (int qq = 0; qq < 3 && !rc; ++qq) {   // the two buffers of a list share one capacity
                    const int la = qq == 0 ? LIST_XY : (qq == 1 ? LIST_XX : LIST_YY), lb = qq == 0 ? LIST_XYB : (qq == 1 ? LIST_XXB : LIST_YYB);
                    if (!c->lists[la].cap && !c->lists[lb].cap) continue;
                    const double both = (double)std::max(c->lists[la].cap, c->lists[lb].cap);
                    rc = ensure_list(c, la, 0, 0, both);
                    if (!rc && c->lists[lb].cap) rc = ensure_list(c, lb, 0, 0, both);
                }
                if (rc) { finish_job(j, rc); continue; }
                j->executed_base = cur.k;
                j->phase = 3;   // resume
                pending.push_front(j);
            }
            (void)hipEventDestroy(r.ev);
            retiring.erase(retiring.begin());
            moved = true;
        }
        return moved;
    }

    // one batch of kEngineBatch iterations of the current plan on this engine's stream
    int launch_one_batch()
    {
        const int batch = kEngineBatch;
        if (engine_profile()->on) {   // eager, the flow-pass launches bracketed by events
            for (int k = 0; k < batch; ++k)
                for (const TLaunch &l : plan) {
                    if (l.kernel == TK_FLOW || l.kernel == TK_FLOW_D2) {
                        FlowEv fe{nullptr, nullptr, live()};
                        if (hipEventCreate(&fe.a) == hipSuccess && hipEventCreate(&fe.b) == hipSuccess) {
                            launch_table(tab.dev, l, s, fe.a, fe.b);
                            flow_ev.push_back(fe);
                            continue;
                        }
                    }
                    launch_table(tab.dev, l, s);
                }
            return hipGetLastError() == hipSuccess ? CVO_HIP_OK : CVO_HIP_ERR_HIP;
        }
        return run_plan(tab.dev, plans, plan, s, use_graph, batch);
    }

    // Advance as far as possible without waiting on the GPU.  `want` = how many members this
    // engine should hold at most right now.  Returns true if anything moved.
    bool pump(std::deque<AlignJob *> &pending, int want)
    {
        if (failed) return false;
        if (hipSetDevice(device) != hipSuccess) { fail_all("hipSetDevice failed", pending); return true; }
        bool moved = false;
        // batches that have completed: look for members that stopped
        while (checked < launched) {
            const hipError_t q = hipEventQuery(ev[checked % 4]);
            if (q == hipErrorNotReady) break;
            if (q != hipSuccess) { fail_all("fused poll failed", pending); return true; }
            ++checked;
            { const double t0 = now_ms(); collect_stopped(); t_collect += now_ms() - t0; }
            moved = true;
        }
        { const double t0 = now_ms(); if (finish_arrived(pending, false)) moved = true; t_finish += now_ms() - t0; }
        // free slots take the next registrations
        while (!pending.empty() && live() < std::min(want, (int)ENGINE_SLOTS)) {
            AlignJob *j = pending.front();
            pending.pop_front();
            int z = 0;
            while (member[z]) ++z;
            { const double t0 = now_ms(); insert(j, z); t_insert += now_ms() - t0; }
            moved = true;
        }
        // batches kept queued per engine (the other engines fill the gap between two batches of this one)
        constexpr long long depth = 2;   // (one batch queued per engine instead of two: -17 %, profiles/r02_ab.txt)
        while (live() > 0 && launched - checked < depth) {
            if (dirty) {
                const double t0 = now_ms();
                const int rc = replan();
                t_replan += now_ms() - t0;
                if (rc) { fail_all("fused launch recording failed", pending); return true; }
            }
            const double t_l0 = now_ms();
            const int rc_launch = launch_one_batch();
            if (rc_launch != CVO_HIP_OK ||
                hipEventRecord(ev[launched % 4], s) != hipSuccess) {
                fail_all("fused launch failed", pending);
                return true;
            }
            t_launch += now_ms() - t_l0;
            ++launched;
            ++n_batches[zdim >= 16 ? 4 : (zdim >= 8 ? 3 : (zdim >= 4 ? 2 : (zdim >= 2 ? 1 : 0)))];
            moved = true;
        }
        if (live() == 0 && dirty && launched == checked) {   // the last members left: empty the table
            if (replan() != CVO_HIP_OK) { fail_all("table update failed", pending); return true; }
        }
        return moved;
    }

    // block until the oldest thing in flight has completed
    void wait_oldest(std::deque<AlignJob *> &pending)
    {
        const double t0 = now_ms();
        struct Acc { double &t; double t0; ~Acc() { t += now_ms() - t0; } } acc{t_wait, t0};
        if (checked < launched) {
            if (hipEventSynchronize(ev[checked % 4]) != hipSuccess) fail_all("fused poll failed", pending);
        } else if (!retiring.empty()) {
            (void)hipEventSynchronize(retiring.front().ev);
        }
    }
};

// engines live for the life of the process (like their streams); a call borrows them
std::mutex *engine_mutex()
{
    static std::mutex *mu = new std::mutex;   // (never destroyed: see cvo_lock.h)
    return mu;
}

Engine *engine_checkout(int device)
{
    static std::vector<Engine *> *all = new std::vector<Engine *>();
    std::lock_guard<std::mutex> lock(*engine_mutex());
    for (Engine *e : *all)
        if (!e->in_use && e->device == device && !e->failed) { e->in_use = true; return e; }
    // The runtime deals streams to its (four) hardware queues in the order they are created: engines
    // whose streams share a queue run one behind the other.  An engine created alone, long after its
    // siblings, landed on a queue one of them already had (256 pairs per call after a first call with
    // three engines: 4 130 -> 3 370 registrations/s).  So the first call on a device creates all four
    // streams back to back; the spares cost a table each.
    size_t have = 0;
    for (Engine *e : *all) have += e->device == device && !e->failed;
    Engine *first = nullptr;
    for (size_t k = have; k < std::max<size_t>(have + 1, 4); ++k) {
        Engine *e = new (std::nothrow) Engine();
        if (!e) break;
        if (e->create(device) != 0) { (void)hipGetLastError(); delete e; break; }
        all->push_back(e);
        if (!first) first = e;
    }
    if (first) first->in_use = true;
    return first;
}

void engine_release(Engine *e)
{
    if (!e->flow_ev.empty()) {   // (the engine is idle: every event has completed)
        EngineProfile *pr = engine_profile();
        std::lock_guard<std::mutex> plock(pr->mu);
        for (size_t q = 0; q < e->flow_ev.size(); ++q) {
            auto &fe = e->flow_ev[q];
            float ms = 0.f, gap = 0.f;
            if (hipEventSynchronize(fe.b) == hipSuccess && hipEventElapsedTime(&ms, fe.a, fe.b) == hipSuccess) {
                pr->flow_ms += ms; pr->flow_launches++; pr->flow_slots += fe.live;
                if (q + 1 < e->flow_ev.size() && hipEventElapsedTime(&gap, fe.a, e->flow_ev[q + 1].a) != hipSuccess) gap = 0.f;
                if (pr->dur_us.size() < (size_t)1 << 20) {
                    pr->dur_us.push_back(ms * 1e3f); pr->period_us.push_back(gap * 1e3f); pr->slots.push_back(fe.live);
                }
            }
        }
        for (auto &fe : e->flow_ev) {
            (void)hipEventDestroy(fe.a);
            (void)hipEventDestroy(fe.b);
        }
        e->flow_ev.clear();
    }
    std::lock_guard<std::mutex> lock(*engine_mutex());
    if (env_engine_debug())
        fprintf(stderr, "[cvo_hip] engine %p: batches at zdim 1/2/4/8/16: %lld %lld %lld %lld %lld, replans %lld, "
                "graph captures %lld hits %lld; host ms: insert %.2f replan %.2f launch %.2f collect %.2f finish %.2f wait %.2f\n",
                (void *)e, e->n_batches[0], e->n_batches[1], e->n_batches[2],
                e->n_batches[3], e->n_batches[4], e->n_replans, e->plans.captures, e->plans.hits,
                e->t_insert, e->t_replan, e->t_launch, e->t_collect, e->t_finish, e->t_wait);
    e->t_insert = e->t_replan = e->t_launch = e->t_collect = e->t_finish = e->t_wait = 0;
    for (long long &v : e->n_batches) v = 0;
    e->n_replans = 0;
    e->launched = e->checked = 0;
    e->in_use = false;
}

}   // namespace

int cvo_hip_align(cvo_hip_ctx *ctx, cvo_hip_state *s, cvo_hip_trace *trace, int trace_cap,
                  int *n_iter)
{
    cvo_lock::Api api_guard;
    if (!ctx || !s) return CVO_HIP_ERR_INVALID;
    AlignJob j;
    j.ctx = ctx; j.s = s; j.trace = trace; j.trace_cap = trace_cap; j.n_iter = n_iter;
    j.paced = true;
    int rc = job_begin(j);
    if (rc) return rc;
    while (!job_pump(j, true)) {}
    return j.rc;
}

int cvo_hip_align_many(cvo_hip_ctx **ctxs, cvo_hip_state **states, int *n_iters, int count)
{
    cvo_lock::Api api_guard;
    if (count < 0 || (count > 0 && (!ctxs || !states))) return CVO_HIP_ERR_INVALID;
    const bool dbg_many = env_engine_debug();
    const double t_many0 = Engine::now_ms();
    struct Tell { double t0; int n; ~Tell() { if (env_engine_debug()) fprintf(stderr, "[cvo_hip] align_many(%d): %.2f ms\n", n, Engine::now_ms() - t0); } } tell{t_many0, count};
    std::vector<AlignJob> jobs((size_t)count);
    for (int i = 0; i < count; ++i) {
        if (!ctxs[i] || !states[i]) return CVO_HIP_ERR_INVALID;
        for (int k = 0; k < i; ++k)
            if (ctxs[k] == ctxs[i]) return CVO_HIP_ERR_INVALID;   // one job per context
        jobs[i].ctx = ctxs[i];
        jobs[i].s = states[i];
        jobs[i].n_iter = n_iters ? &n_iters[i] : nullptr;
    }
    int first_err = CVO_HIP_OK;
    std::vector<char> taken((size_t)count, 0);
    // fused groups: same device, same mode, nothing that needs its own launches.  The jobs of
    // a class wait in one queue; one or two engines (two from 8 jobs on: two groups fill each
    // other's bubbles -- single-block post kernels, kernel boundaries) take them into their
    // slots as slots become free.
    static const bool no_fuse = getenv("CVO_HIP_NO_FUSE") != nullptr;
    if (!no_fuse && count > 1) {
        constexpr int gmax = ENGINE_SLOTS;
        for (int i = 0; i < count; ++i) {
            if (taken[i] || jobs[i].phase != 0 || !fusable(jobs[i].ctx)) continue;
            std::deque<AlignJob *> pending;
            for (int k = i; k < count; ++k)
                if (!taken[k] && jobs[k].phase == 0 && fusable(jobs[k].ctx) &&
                    jobs[k].ctx->device == jobs[i].ctx->device &&
                    jobs[k].ctx->prm.mode == jobs[i].ctx->prm.mode)
                    pending.push_back(&jobs[k]);
            if (pending.size() < 2) continue;
            for (AlignJob *j : pending) taken[j - &jobs[0]] = 1;
            const size_t total = pending.size();
            // how many engines share the GPU: one group alone leaves it idle in its single-block post
            // kernels and at every kernel boundary; two fill each other's bubbles (32 pairs: 2079 ->
            // 2428 registrations/s; three: 2273); a third pays once there are enough jobs to keep three
            // groups well filled (64 distinct pairs in engines of 32 slots: 2 x 32 2897, 3 x 22 3149,
            // 4 x 16 2822); a fourth when three tables cannot hold every job at once (128 pairs:
            // 3 x 32 and 32 waiting 3297, 4 x 32 3644 -- the longest registration starts at once)
            constexpr size_t max_engines = 4;   // (the runtime's hardware queues; with 8 queues and 6 engines: -40 % at 64 pairs, r04_ab.txt 1)
            size_t ngroups = total > 3 * ENGINE_SLOTS ? 4 : (total >= 40 ? 3 : (total >= 8 ? 2 : 1));
            ngroups = std::max<size_t>(1, std::min(ngroups, max_engines));
            if (const char *e = getenv("CVO_HIP_ENGINES_FORCE")) ngroups = (size_t)std::max(1, std::min(atoi(e), 8));   // (tuning probe)
            bool graphs_ok = true;   // (capture policy: cvo_hip_set_graph_capture)
            for (AlignJob *j : pending) graphs_ok = graphs_ok && j->ctx->use_graphs;
            // Asynchronous xy builds shorten the launch chain of a registration; once the GPU is
            // shared by many registrations the chain no longer matters and the extra builds cost
            // more than they save: members of large groups keep the synchronous scheme.
            constexpr int crowd = 2;
            std::vector<Engine *> engines;
            for (size_t g = 0; g < ngroups; ++g) {
                Engine *e = engine_checkout(jobs[i].ctx->device);
                if (!e) break;
                e->crowded = (int)total > crowd;
                e->use_graph = graphs_ok;
                e->zdim = 0;
                e->t_idle_at = 0;
                e->dirty = true;
                engines.push_back(e);
            }
            if (engines.empty()) {   // no engine to be had: the jobs run on their own below
                for (AlignJob *j : pending) taken[j - &jobs[0]] = 0;
                continue;
            }
            // the first fill is even (16 + 16 of 32, 4 + 4 of 8); later a free slot takes the next job
            const int share = std::min<int>(gmax, (int)((total + engines.size() - 1) / engines.size()));
            for (;;) {
                bool any = false, moved = false;
                for (Engine *e : engines) {
                    if (e->pump(pending, share)) moved = true;
                    if (!e->idle()) any = true;
                    else if (dbg_many && e->t_idle_at == 0) {
                        e->t_idle_at = Engine::now_ms();
                        fprintf(stderr, "[cvo_hip]   engine %p idle after %.2f ms\n", (void *)e, e->t_idle_at - t_many0);
                    }
                }
                if (!any && pending.empty()) break;
                bool alive = false;
                for (Engine *e : engines) alive = alive || !e->failed;
                if (!alive) break;
                if (!moved)   // everybody waits for the GPU: block on the oldest thing in flight
                    for (Engine *e : engines)
                        if (!e->idle() && !e->failed) { e->wait_oldest(pending); break; }
            }
            for (Engine *e : engines) engine_release(e);
        }
        for (int i = 0; i < count; ++i)
            if (jobs[i].phase == 2 && jobs[i].rc && !first_err) first_err = jobs[i].rc;
    }
    // the others run on their own streams and tables
    for (int i = 0; i < count; ++i) {
        if (taken[i] || jobs[i].phase == 2) continue;
        jobs[i].ctx->crowded = false;
        jobs[i].ctx->lone = true;
        const int rc = job_begin(jobs[i]);
        if (rc) { jobs[i].rc = rc; jobs[i].phase = 2; if (!first_err) first_err = rc; }
    }
    // round-robin: every pass tops up each registration's queue and looks at its
    // poll word without blocking; when nobody moved, block on the oldest job
    for (;;) {
        int live = 0, moved = 0, first_live = -1;
        for (int i = 0; i < count; ++i) {
            if (jobs[i].phase == 2) continue;
            const int before_phase = jobs[i].phase, before_checked = jobs[i].checked;
            if (job_pump(jobs[i], false)) {
                if (jobs[i].rc && !first_err) first_err = jobs[i].rc;
                ++moved;
                continue;
            }
            ++live;
            if (first_live < 0) first_live = i;
            if (jobs[i].phase != before_phase || jobs[i].checked != before_checked) ++moved;
        }
        if (live == 0) break;
        if (!moved) {
            if (job_pump(jobs[first_live], true) && jobs[first_live].rc && !first_err)
                first_err = jobs[first_live].rc;
        }
    }
    return first_err;
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

int cvo_hip_engine_profiling(int enable)
{
    EngineProfile *pr = engine_profile();
    std::lock_guard<std::mutex> lock(pr->mu);
    pr->on = enable != 0;
    return CVO_HIP_OK;
}

int cvo_hip_get_engine_profile(double *flow_ms, long long *flow_launches, double *flow_registrations, int reset)
{
    if (!flow_ms || !flow_launches || !flow_registrations) return CVO_HIP_ERR_INVALID;
    EngineProfile *pr = engine_profile();
    std::lock_guard<std::mutex> lock(pr->mu);
    *flow_ms = pr->flow_ms; *flow_launches = pr->flow_launches; *flow_registrations = pr->flow_slots;
    if (reset) { pr->flow_ms = 0.0; pr->flow_launches = 0; pr->flow_slots = 0.0; pr->dur_us.clear(); pr->period_us.clear(); pr->slots.clear(); }
    return CVO_HIP_OK;
}

int cvo_hip_get_engine_flow_trace(float *dur_us, float *period_us, int *slots, int capacity, int *count, int reset)
{
    if (!count || capacity < 0) return CVO_HIP_ERR_INVALID;
    EngineProfile *pr = engine_profile();
    std::lock_guard<std::mutex> lock(pr->mu);
    const int n = (int)std::min<size_t>(pr->dur_us.size(), (size_t)capacity);
    for (int q = 0; q < n; ++q) {
        if (dur_us) dur_us[q] = pr->dur_us[(size_t)q];
        if (period_us) period_us[q] = pr->period_us[(size_t)q];
        if (slots) slots[q] = pr->slots[(size_t)q];
    }
    *count = (int)pr->dur_us.size();
    if (reset) { pr->dur_us.clear(); pr->period_us.clear(); pr->slots.clear(); }
    return CVO_HIP_OK;
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
    ctx->use_graphs = enable != 0 && !env_no_graph();
    if (!ctx->use_graphs) drop_graphs(ctx);
    return CVO_HIP_OK;
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

int cvo_hip_synchronize(cvo_hip_ctx *ctx)
{
    cvo_lock::Api api_guard;
    if (!ctx) return CVO_HIP_ERR_INVALID;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CVO_HIP_OK;
}

}   // extern "C"
