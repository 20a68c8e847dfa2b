// cvo_internal.h -- what the translation units of the host library share: the context, the growable device
// arrays, argument tables and captured batches, the recorded launches of an iteration, the resumable align() job.
//   cvo_capi.cpp    entry points that are not listed below (create / destroy, parameters, sharding, mailboxes, the
//                   low-level calls of the loop body, function_inner_product, profiling getters)
//   cvo_clouds.cpp  the cloud hand-over (tail of set_pcd(), ref src/cvo.cpp:344-356): one cloud, a batch
//   cvo_plan.cpp    from the launches of ONE iteration (recorded, not issued) to launch plans, argument tables and
//                   captured batches; the eager launches of the low-level entry points
//   cvo_job.cpp     align() as a resumable job: begin / pump / finish; cvo_hip_align
//   cvo_engine.cpp  fused groups as long-lived engines with continuous batching; cvo_hip_align_many
#pragma once
#include "cvo_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "cvo_cloud.h"
#include "cvo_comm.h"
#include "cvo_device.h"
#include "cvo_lock.h"
#include "se3_math.hpp"

namespace cvo_impl {
using namespace cvo_dev;


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
    float *bbox_pin_dev = nullptr;    // ... as the device addresses it (the one-launch preparation writes the box there)
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
// (In front of the slots, in the same allocation: the table's BUILD MASKS, cvo_device.h kTableHeaderBytes -- one word per tile
// list, bit z = slot z may have that list to build in the coming filter launch.  The post-step kernels keep their slot's bits;
// whenever the host changes the table or starts a registration in it, it sets them all: a set bit only costs the full check.)
struct TableBuf {
    Slot *dev = nullptr;
    char *raw = nullptr;
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
        if (hipMalloc((void **)&raw, kTableHeaderBytes + (size_t)n * sizeof(Slot)) != hipSuccess) { raw = nullptr; return -1; }
        dev = reinterpret_cast<Slot *>(raw + kTableHeaderBytes);
        if (hipMemsetAsync(raw, 0, kTableHeaderBytes + (size_t)n * sizeof(Slot), s) != hipSuccess) return -1;
        if (arm(s) != 0) return -1;
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
        if (raw) (void)hipFree(raw);
        dev = nullptr; raw = nullptr; stage = nullptr; nslots = 0;
        image.clear();
    }
    uint32_t *masks() const { return reinterpret_cast<uint32_t *>(raw); }
    // every slot may have every list to build (stream-ordered): after a change of the table, at the start of a registration
    int arm(hipStream_t s) { return hipMemsetAsync(raw, 0xff, 4 * sizeof(uint32_t), s) == hipSuccess ? 0 : -1; }
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
        if (arm(s) != 0) return -1;   // (slots may have moved: the masks' bits with them)
        if (hipEventRecord(stage_ev[b], s) != hipSuccess) return -1;
        stage_used[b] = true;
        return 0;
    }
};

// A captured batch of kBatch iterations of a launch plan (hipGraph).  It depends on the table's
// address and on the plan -- kernels, grids, LDS sizes -- not on any argument: one capture
// serves every frame pair (and every membership of a fused group) of the same shape.
struct PlanGraph {
    std::vector<TLaunch> pre;     // launched once in front of the iterations
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


}   // namespace cvo_impl

// (an internal header: the host translation units all work in these two namespaces)
using namespace cvo_dev;
using namespace cvo_impl;

// Policy and test switches of a context beyond the fields of cvo_hip_ctx that have always carried one (allow_head, allow_run,
// use_graphs ...): set through cvo_hip_set_option (cvo_capi.cpp: the table of keys); their defaults come from the environment,
// read ONCE, when the context is created.  A call that serves many contexts (cvo_hip_align_many, cvo_hip_set_pcd_many) goes by
// its first context's.
struct CtxOptions {
    bool no_cand = false;             // "candidate_records" = 0: expand the tile list every time
    bool no_graph = false;            // "no_graph": no stream captures at all, whatever graph_capture says
    bool sync_upload = false;         // "sync_upload": hand-overs wait for the device
    bool engine_debug = false;        // "engine_debug": host-side clocks of the engines on stderr
    bool no_cloud_one = false;        // "one_launch_hand_over" = 0: the multi-launch cloud preparation
    bool no_alone = false;            // "small_calls_alone" = 0: small align_many calls through the engines
    bool no_fuse = false;             // "fused_groups" = 0: align_many runs every registration on its own stream
    bool no_pack = false;             // "kept_pack" = 0: 8 + 4 byte kept entries
    bool no_final_mirror = false;     // "final_mirror" = 0: the final state comes by a copy in stream order
    bool twist_on_shared_gpu = false; // "twist_on_shared_gpu": in-launch exchange although the ranks share a GPU
    bool comm_debug = false;          // "comm_debug"
    bool post_debug = false;          // "post_debug" (environment only: the buffer is made at create)
    bool no_side_builds = true;       // "side_builds" = 1: a resident run of up to RUN_G_SIDE solvers has its next xy list built beside it (kt_run "side
                                      // builds"; built and measured in round 6, slower than ending the run: off unless asked for, profiles/r06_ab.txt 7)
    float run_build_at = 0.6f;        // "run_build_at": with side builds a run names its next list when this fraction of the room of the list in use is gone
    bool no_restart = false;          // "run_restart" = 0 (diagnostics): a run that times out fails the registration with CVO_HIP_ERR_RUN, its state kept
    bool no_acvo_run = false;         // "acvo_runs" = 0: resident runs for cvo registrations only (round 5's state)
    int tail_alone = 8;               // "tail_alone": when a cvo_hip_align_many call's queue is empty and at most this many registrations are left in its
                                      // engines, all past their wide iterations, they leave the engines for resident runs of their own (0: never)
    int alone_max = 0;                // "alone_max": a call of up to this many registrations leaves them to their own streams (0: by the clouds)
    int engines_force = 0;            // "engines": engines of an align_many call (0: by the call's size)
    int narrow_blocks = 0;            // "narrow_blocks": ... and give every registration this many blocks per list pass (0: by the engine's slots as ever)
    bool narrow_merge = false;        // "narrow_merge": once every registration of a cvo_hip_align_many call is past its wide iterations its engines launch the step pass with the twist in front
    int engine_crowd = 2;             // "engine_crowd" (tuning probe): a call of more registrations than this keeps the synchronous list scheme in its engines
    int engine_merge_max = 2;         // "engine_merge_max" (tuning probe): engines of up to this many slots launch the step pass with the twist in front
    double list_init = 0.0;           // "list_init": first capacity of every list (0: by the clouds)
    float list_margin = -1.0f;        // "list_margin": width of the tile lists (< 0: by the clouds)
    int run_cand = 0;                 // "run_candidates_max" (0: what the runs hold)
    double mailbox_timeout_s = 5.0;   // "mailbox_timeout_s": read by the next cvo_hip_mailbox_connect
    double run_timeout_ms = 0.0;      // "run_timeout_ms": a resident run's exchange gives up after this long (0: 1 s)
    int run_fault = 0;                // "run_fault" (test switch, PostStepArgs::run_fault)
    int wait_policy = 0;              // "wait_policy": how cvo_hip_align's calling thread waits between looks at the pinned words:
                                      // 0 spin (pause), 1 yield (sched_yield), 2 sleep (50 us naps)
};

struct cvo_hip_ctx {
    CtxOptions opt;
    int no_run_backoff = 0;              // registrations to go without resident runs (one of them timed out, job_pump)
    long long side_builds_launched = 0;  // side builds launched by this context ("side_builds_launched")
    long long tail_handovers = 0;        // times this context's registration left an engine for runs of its own ("tail_handovers")
    long long run_aborts = 0;            // resident runs of this context that gave up at their entry hand-shake ("run_aborts")
    long long run_timeouts = 0;          // resident runs of this context that gave up on an exchange (cvo_hip_get_option "run_timeouts")
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
    DevHead *final_mirror = nullptr;      // pinned: the head of a loop that stopped with a verdict (PostStepArgs::final_mirror)
    int32_t *run_mirror = nullptr;        // pinned: resident runs (kt_run) that have ended since align() began
    int32_t *hint_mirror = nullptr;       // pinned: DevHead::run_hint, candidates expected in the record in use (-1: none yet)
    int32_t *side_mirror = nullptr;       // pinned: requests for side builds the runs of the align() in progress have made (PostStepArgs::side_mirror)
    hipStream_t side_stream = nullptr;    // ... and the stream their kernels go out on (made when a plan first has them)
    std::vector<TLaunch> plan_side;       // ... the two launches of a side build (kt_side_filter, kt_side_record); empty: the plan has none
    DevBuf run_mail;                      // RunMail of this registration's resident runs
    bool plan_has_final_mirror = false;   // the plan of the align() in progress publishes its final head to final_mirror
    bool allow_run = true;                // CVO_HIP_NO_RUN
    int run_g_max = RUN_G;                // solver blocks of a resident run at most: a block per compute unit, a few to spare (cvo_hip_create)
    int run_g_call = RUN_G;               // ... and in the call in progress: a small cvo_hip_align_many call leaves k registrations to their own streams, whose
                                          // runs spin side by side -- k (g + 1) blocks must fit the compute units (cvo_engine.cpp)
    bool call_no_spec = false;            // ... and with more than two of them no first run goes out on spec (a record that does not fit costs every neighbour)
    int big_run_backoff = 0;              // registrations to go without runs of more than 32 solvers (one gave up at its entry hand-shake, job_pump)
    bool spec_first_run = true;           // the first run of a registration goes out on spec behind its first two slots (job_pump learns from each try)
    bool head_graphs = false;             // CVO_HIP_RUN_GRAPHS: head-mode plans go out as captured batches too (they launch eagerly by default)
    int run_small_max = 0;                // ... and with the launch of RUN_G_SMALL + 1 blocks up to this many
    int run_nnz_max = 0;                  // a batch begins with a resident run when the record in use is expected to hold at most this many candidates
    std::vector<TLaunch> plan_pre;        // launches in front of a RUN batch's iterations (the kt_run launch); empty: the plan has no run
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
    bool mail_shared_device = false;     // a peer's kernels run on this rank's own GPU (tests, rehearsals): the exchange inside
                                         // k_step_twist -- every block spinning -- would keep the peer's kernels off the GPU
    unsigned long long mail_dev_id = 0;  // this device's Mailbox::owner_dev
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

namespace cvo_impl {

#define HIP_TRY(ctx, expr)                                                                   \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            if (ctx) (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);         \
            return CVO_HIP_ERR_HIP;                                                          \
        }                                                                                    \
    } while (0)

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
    int runs_enq = 0;       // resident runs enqueued in this round
    bool run_waiting = false;   // the last batch began with a resident run that has not reported its end yet
    bool spec_pending = false;  // ... and that run was sent on spec (job_pump: the first run of a registration)
    bool paced_nb = false;  // a registration on its own inside cvo_hip_align_many: the paced steps of job_pump, one look per call
    int idle_seen = 0;      // ... and how often in a row its stream was found idle with the mirrors where they were
    int side_seen = 0;      // the last request for a side build that was answered (the side mirror's word)
    int side_launched = 0;  // side builds launched for this registration
    int restarts = 0;       // times this registration was begun again without resident runs (a run of it timed out)
    bool paced = false;     // cvo_hip_align only: the calling thread has nothing else to pump and may sit in the
                            // paced loop of job_pump (align_many's blocking fall-back must keep its round-robin going:
                            // the other jobs -- the peer ranks of a mailbox world among them -- run dry otherwise)
};

// ---- cvo_capi.cpp
int fail(cvo_hip_ctx *ctx, int code, const char *msg);
bool engine_debug_on();
int apply_option(cvo_hip_ctx *ctx, const char *key, double value);
inline bool runs_allowed(const cvo_hip_ctx *ctx) { return ctx->allow_run && ctx->no_run_backoff <= 0; }
const char *params_problem(const cvo_hip_params &p);
DevParams make_dev_params(const cvo_hip_params &p);
// ---- cvo_clouds.cpp
int cloud_ready(cvo_hip_ctx *ctx, Cloud &c);
int cloud_reserve(cvo_hip_ctx *ctx, Cloud &c, const float *xyz, const float *feat, int n, int layout);
int upload_cloud(cvo_hip_ctx *ctx, Cloud &c, const float *xyz, const float *feat, int n, int layout, bool on_device = false);
// ---- cvo_plan.cpp
FilterPlan plan_filter(int nrows, int nb);
int ensure_buf(cvo_hip_ctx *ctx, DevBuf &b, size_t bytes);
int ensure_list(cvo_hip_ctx *ctx, int list, int nrows, int nb, double at_least);
void shard_ranges(const cvo_hip_ctx *ctx, int &rlo, int &rhi, int &slo, int &shi);
int fill_filter_geometry(cvo_hip_ctx *ctx, DevState *h);
int mailboxes_usable(cvo_hip_ctx *ctx);
bool host_reduce(const cvo_hip_ctx *ctx);
bool multi_rank(const cvo_hip_ctx *ctx);
DevParams loop_params(const cvo_hip_ctx *ctx);
hipStream_t loop_stream(const cvo_hip_ctx *ctx);
int enqueue_filter(cvo_hip_ctx *ctx, int list, const Cloud &ca, int row_lo, int row_hi, int tf_a, const Cloud &cb, int tf_b, int check_done);
int enqueue_process(cvo_hip_ctx *ctx, int mode, int list, DevBuf &part, const float4 *pos_a, const float *feat_a, int tf_a,
                    const float4 *pos_b, const float *feat_b, int tf_b, int first_counted, int check_done);
int drain_events(cvo_hip_ctx *ctx, int n_exec = -1, const DevState *fin = nullptr);
int reduce_over_ranks(cvo_hip_ctx *ctx, int off, int count);
int enqueue_flow(cvo_hip_ctx *ctx, bool tf_moving, int check_done, bool do_math, cvo_hip_trace *trace, int trace_cap);
int enqueue_step(cvo_hip_ctx *ctx, int check_done, bool do_math, cvo_hip_trace *trace, int trace_cap);
int check_overflow_and_grow(cvo_hip_ctx *ctx, bool *redo);
int prepare_buffers(cvo_hip_ctx *ctx);
int enqueue_iterations(cvo_hip_ctx *ctx, int count, int tag0, int trace_cap);
void drop_graphs(cvo_hip_ctx *ctx);
TLaunch mk_launch(int kernel, int q, unsigned gx, unsigned gz, unsigned smem = 0);
bool plan_lone(const std::vector<RecOp> &ops, Slot &slot, std::vector<TLaunch> &plan, const bool allow_head, bool *head_mode,
               std::vector<TLaunch> *pre = nullptr, std::vector<TLaunch> *side = nullptr);
void set_build_masks(Slot &slot, const std::vector<TLaunch> &plan, uint32_t *masks, int z);
bool plan_fused(const std::vector<const std::vector<RecOp> *> &ops, const std::vector<Slot *> &slots, int zdim, std::vector<TLaunch> &plan);
int run_plan(const Slot *tab, PlanCache &cache, const std::vector<TLaunch> &plan, hipStream_t s, bool use_graph, int iterations,
             const std::vector<TLaunch> *pre = nullptr);
int record_iteration(cvo_hip_ctx *ctx, std::vector<RecOp> &ops, int trace_cap);
int prepare_lone_plan(cvo_hip_ctx *ctx, int trace_cap);
int launch_batch(cvo_hip_ctx *ctx, int tag0, int trace_cap, bool with_run = false, int slots = kBatch, bool small_run = false, bool two_runs = false);
constexpr int kNoRunBackoff = 64;    // registrations a context goes without resident runs after one of them timed out
constexpr int kBigRunBackoff = 16;   // registrations a context keeps its large runs away after one of them found the compute units taken
constexpr int kShortBatch = 2;      // classic slots of a batch of a plan that has a resident run (job_pump; an even number, see kBatch)
constexpr int kRunBatchSlots = 2;   // classic slots behind the resident run of a RUN batch (an even number, see kBatch)
int zero_counters(cvo_hip_ctx *ctx);
int push_state_fields(cvo_hip_ctx *ctx, size_t off, size_t bytes);
int fetch_red(cvo_hip_ctx *ctx, int off, int count, double *out);
// ---- cvo_job.cpp
void decide_scheme(cvo_hip_ctx *ctx);
std::atomic<long long> &mirror_retries();
int job_begin(AlignJob &j);
int job_finish(AlignJob &j);
int job_continue_alone(AlignJob &j);
int job_pump(AlignJob &j, bool block);

}   // namespace cvo_impl
