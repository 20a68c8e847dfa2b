// cvo_engine.cpp -- batched mode (BASELINE configs[4]): fused groups as long-lived engines with continuous
// batching, and cvo_hip_align_many, the host loop that pumps them.
#include "cvo_internal.h"

using namespace cvo_dev;
using namespace cvo_impl;

namespace cvo_impl {

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

// A FEW cvo registrations on small clouds are faster each on its own stream than sharing an engine's launches (round 5): on
// its own a registration runs most of its iterations inside resident runs (kt_run: ~10 us per iteration on a few dozen compute
// units when the clouds are small, several of them side by side), in a group of two or four an iteration is a chain of four or
// five dependent launches (~40 us) whatever its members need.  Registrations per second, engines / on their own (one MI355X,
// distinct pairs, profiles/r05_ab.txt 22): 3k x 3k 2 / 4 / 8 / 12 / 16 / 24 per call 1 027 / 1 113 / 1 832 / 2 697 / 2 823 / 4 578 against
// 2 316 / 1 914 / 2 896 / 2 901 / 2 866 / -; 6k x 6k 2 / 4 / 8 / 12 754 / 1 378 / 1 723 / 2 325 against 1 428 / 1 686 / 2 289 / 2 094;
// 10k x 10k 2 / 3 / 4 591 / 824 / 987 against 722 / 735 / 774; 12k 2 / 3 756 / 1 113 against 983 / 1 040; 14k 540 / 777 against 781 / 709.
bool better_alone(const std::deque<AlignJob *> &pending)
{
    const bool off = !pending.empty() && pending.front()->ctx->opt.no_alone;   // (test switch "small_calls_alone" = 0: small calls through the engines as before)
    if (off || engine_profile()->on) return false;   // (... or the caller is measuring the engines: cvo_hip_engine_profiling)
    // (acvo has no runs, but on its own an iteration is two launches against the engines' nine: acvo 3k x 3k 2 / 4 / 8 / 16 per call
    // 507 / 1 120 / 1 336 / 2 632 against 1 318 / 1 508 / 2 063 / 1 924, 6k x 6k 369 / 841 / 1 220 / 2 190 against 857 / 1 016 / 1 214 / 1 266)
    double pairs = 0.0;
    bool acvo = false;
    for (const AlignJob *j : pending) {
        const cvo_hip_ctx *c = j->ctx;
        acvo = c->prm.mode == CVO_HIP_MODE_ACVO;   // (the members of a group share their mode)
        if (!c->allow_head || !c->allow_async || !c->allow_merge || c->opt.no_cand || c->fixed.np > 65536 || c->moving.np > 65536 ||
            (acvo ? !c->allow_async_self : !c->allow_run))
            return false;
        pairs = std::max(pairs, (double)c->fixed.n * (double)c->moving.n);
    }
    // (round 6, with every registration's runs sized to its share of the compute units and acvo in runs as well -- on their own / through the
    // engines, profiles/r06_ab.txt 4: cvo 3k 12 per call 2 758 / 2 181, 16: 2 633 / 2 652; 6k 12: 2 800 / 2 939; 10k 3: 1 140 / 761, 4: 1 277 / 826;
    // acvo 3k 12: 2 836 / 1 653; 6k 4: 2 373 / 972, 8: 1 463 / 1 481; 10k 2 / 3 / 4: 1 316 / 455, 1 073 / 814, 1 102 / 965)
    const size_t few = acvo ? (pairs <= 1.6e7 ? 12 : (pairs <= 5.0e7 ? 4 : (pairs <= 1.2e8 ? 4 : 0)))
                            : (pairs <= 1.6e7 ? 12 : (pairs <= 5.0e7 ? 8 : (pairs <= 2.0e8 ? 4 : 0)));
    if (const int force = pending.front()->ctx->opt.alone_max) return pending.size() <= (size_t)force;   // (tuning probe: "alone_max")
    return pending.size() <= few;
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
    int batch_len = kEngineBatch;          // iterations per batch (shorter when the call's tail is near: what is queued must complete before anybody leaves)
    bool narrow = false;                   // every registration of the call is past its wide iterations (align_many): replan goes by the narrow-phase options
    bool narrow_merge = false;             // ... the step launch carries the twist whatever zdim
    int narrow_blocks = 0;                 // ... blocks of a list pass per registration (0: by zdim as ever)
    bool wind_down = false;                // the call's tail: no further batches; the members that are left go on alone (release_members)
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
            launch_prepare(c->st, loop_params(c), s, tab.masks());   // (the lists are built anew: the build masks' bits up)
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
        (void)tab.arm(s);   // (a registration begins: everything is built; the replan that follows arms again with its copy)
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
        const int nblk = (narrow && narrow_blocks > 0) ? std::min(narrow_blocks, nblk_for(zdim)) : nblk_for(zdim);
        int merge_max = 2;   // (4 / 8 / 16 measured alike on the round-4 kernels, 32 loses 7 %: profiles/r04_ab.txt 13)
        std::vector<const std::vector<RecOp> *> po;
        std::vector<Slot *> ps;
        for (int z = 0; z < ENGINE_SLOTS; ++z)
            if (member[z]) { merge_max = member[z]->ctx->opt.engine_merge_max; break; }
        for (int z = 0; z < ENGINE_SLOTS; ++z) {
            if (!member[z]) { slot[z].active = 0; continue; }
            cvo_hip_ctx *c = member[z]->ctx;
            if (regeom || ops[z].empty()) {
                c->proc_blocks = nblk;
                // k_step_twist pays for the saved launch with a prologue in every block:
                // a gain while launches are latency-bound, a loss once the GPU is full
                const bool allow = c->allow_merge;
                if (zdim > merge_max && !(narrow && narrow_merge)) c->allow_merge = false;
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
        for (int z = 0; z < ENGINE_SLOTS; ++z)
            if (member[z]) set_build_masks(slot[z], plan, tab.masks(), z);
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
        const int batch = batch_len;
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
        while (live() > 0 && launched - checked < depth && !wind_down) {
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

    // The call's tail (wind_down, every queued batch completed): the members that are still running leave -- the caller carries them on
    // alone (job_continue_alone); their slots go out of the table with the next replan.
    void release_members(std::vector<AlignJob *> &out)
    {
        if (!wind_down || launched != checked || failed) return;
        collect_stopped();
        for (int z = 0; z < ENGINE_SLOTS; ++z) {
            AlignJob *j = member[z];
            if (!j) continue;
            out.push_back(j);
            member[z] = nullptr;
            ops[z].clear();
            dirty = true;
        }
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
    if (engine_debug_on())
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


}   // namespace cvo_impl

extern "C" {

int cvo_hip_align_many(cvo_hip_ctx **ctxs, cvo_hip_state **states, int *n_iters, int count)
{
    cvo_lock::Api api_guard;
    if (count < 0 || (count > 0 && (!ctxs || !states))) return CVO_HIP_ERR_INVALID;
    const bool dbg_many = engine_debug_on();
    const double t_many0 = Engine::now_ms();
    struct Tell { double t0; int n; ~Tell() { if (engine_debug_on()) fprintf(stderr, "[cvo_hip] align_many(%d): %.2f ms\n", n, Engine::now_ms() - t0); } } tell{t_many0, count};
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
    std::vector<AlignJob *> leavers;   // registrations that left the engines in a call's tail: they go on alone below
    // fused groups: same device, same mode, nothing that needs its own launches.  The jobs of
    // a class wait in one queue; one or two engines (two from 8 jobs on: two groups fill each
    // other's bubbles -- single-block post kernels, kernel boundaries) take them into their
    // slots as slots become free.
    const bool no_fuse = count > 0 && ctxs[0] && ctxs[0]->opt.no_fuse;   // ("fused_groups" = 0)
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
            if (better_alone(pending)) continue;   // (they run on their own below, resident runs and all)
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
            if (const int force = pending.front()->ctx->opt.engines_force) ngroups = (size_t)std::max(1, std::min(force, 8));   // (tuning probe: "engines")
            bool graphs_ok = true;   // (capture policy: cvo_hip_set_graph_capture)
            for (AlignJob *j : pending) graphs_ok = graphs_ok && j->ctx->use_graphs;
            // Asynchronous xy builds shorten the launch chain of a registration; once the GPU is
            // shared by many registrations the chain no longer matters and the extra builds cost
            // more than they save: members of large groups keep the synchronous scheme.
            const int crowd = pending.front()->ctx->opt.engine_crowd;
            std::vector<Engine *> engines;
            for (size_t g = 0; g < ngroups; ++g) {
                Engine *e = engine_checkout(jobs[i].ctx->device);
                if (!e) break;
                e->crowded = (int)total > crowd;
                e->use_graph = graphs_ok;
                e->zdim = 0;
                e->wind_down = false;
                e->narrow = false;
                e->narrow_merge = pending.front()->ctx->opt.narrow_merge;
                e->narrow_blocks = pending.front()->ctx->opt.narrow_blocks;
                e->batch_len = kEngineBatch;
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
            // THE CALL'S TAIL (round 6): once the queue is empty and few registrations are left in the engines, all past their wide
            // iterations, an iteration is a chain of five dependent launches (~40 us) for a handful of members -- 165 iterations of the
            // longest of 64 registrations against a mean of 68.  They leave: the engines queue no further batches, and when the queued
            // ones have completed the members go on alone, each on its context's stream, most of their iterations inside resident runs
            // sized to their share of the compute units (below).
            const int tail_k = pending.empty() ? 0 : pending.front()->ctx->opt.tail_alone;
            const int narrow_from = pending.empty() ? 0 : (pending.front()->ctx->prm.mode == CVO_HIP_MODE_ACVO ? 8 : 24);
            bool tail_ok = tail_k > 0 && !engine_profile()->on;
            for (AlignJob *j : pending) {
                const cvo_hip_ctx *c = j->ctx;
                tail_ok = tail_ok && c->allow_head && c->allow_async && c->allow_merge && !c->opt.no_cand && runs_allowed(c) && c->fixed.np <= 65536 &&
                          c->moving.np <= 65536 && (c->prm.mode == CVO_HIP_MODE_CVO || (c->allow_async_self && !c->opt.no_acvo_run)) &&
                          (double)c->fixed.n * (double)c->moving.n <= 2.0e8;
            }
            const bool narrow_merge_ok = !pending.empty() && (pending.front()->ctx->opt.narrow_merge || pending.front()->ctx->opt.narrow_blocks > 0) && !engine_profile()->on;
            int dbg_left = -1, dbg_narrow = -1;
            for (;;) {
                bool any = false, moved = false;
                if ((tail_ok || narrow_merge_ok) && pending.empty() && !engines.front()->wind_down) {
                    int left = 0;
                    bool narrow = true, settled = true;
                    for (Engine *e : engines) {
                        settled = settled && e->retiring.empty();
                        for (AlignJob *j : e->member)
                            if (j) { ++left; narrow = narrow && *(volatile int32_t *)j->ctx->progress_mirror >= narrow_from; }
                    }
                    // THE NARROW PHASE: a launch saved per iteration (the twist in front of the step pass, no post-flow launch) is a loss while the
                    // engines' launches fill the GPU and a gain once every registration of the call is past its wide iterations
                    if (narrow_merge_ok && narrow && left > 0 && !engines.front()->narrow)
                        for (Engine *e : engines) {
                            e->narrow = true;
                            e->dirty = true;
                            for (int z = 0; z < ENGINE_SLOTS; ++z) e->ops[z].clear();   // (recorded again by the next replan)
                        }
                    if (dbg_many && (left != dbg_left || (int)narrow != dbg_narrow)) { dbg_left = left; dbg_narrow = (int)narrow; fprintf(stderr, "[cvo_hip]   %.2f ms: %d left (narrow %d settled %d)\n", Engine::now_ms() - t_many0, left, (int)narrow, (int)settled); }
                    (void)settled;
                    // (batches of 3 iterations once the tail is near -- what is queued must complete before anybody leaves, 0.4-0.8 ms with two
                    // batches of ten -- were measured: the engines lose more by the short batches than the leavers gain, profiles/r06_ab.txt 10)
                    if (tail_ok && left > 0 && left <= tail_k && narrow) {
                        for (Engine *e : engines) e->wind_down = true;
                        if (dbg_many) {
                            fprintf(stderr, "[cvo_hip]   tail: %d registrations left after %.2f ms, slots done:", left, Engine::now_ms() - t_many0);
                            for (Engine *e : engines)
                                for (AlignJob *j : e->member)
                                    if (j) fprintf(stderr, " %d", *(volatile int32_t *)j->ctx->progress_mirror);
                            fprintf(stderr, "\n");
                        }
                    }
                }
                for (Engine *e : engines) {
                    if (e->wind_down) {
                        const size_t before = leavers.size();
                        e->release_members(leavers);
                        if (leavers.size() != before) {
                            moved = true;
                            if (dbg_many) fprintf(stderr, "[cvo_hip]   tail: %zu registrations leave engine %p after %.2f ms\n", leavers.size() - before, (void *)e, Engine::now_ms() - t_many0);
                        }
                    }
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
            for (Engine *e : engines) { e->wind_down = false; e->narrow = false; e->batch_len = kEngineBatch; engine_release(e); }
        }
        for (int i = 0; i < count; ++i)
            if (jobs[i].phase == 2 && jobs[i].rc && !first_err) first_err = jobs[i].rc;
    }
    // the others run on their own streams and tables.  Those of them that run resident runs spin side by side, a block per compute
    // unit: with k of them in the call each keeps its runs to (units / k) - 1 solvers, so that k (g + 1) blocks are resident together
    // whatever their records want (four 3k registrations asked for 128-solver runs on spec: two of four entries waited out their
    // 200 us and backed off -- 4 per call 1 808 /s against 2 210 for 2, profiles/r05_ab.txt 22), and with more than two of them no
    // first run goes out on spec
    std::vector<char> cont((size_t)count, 0);
    for (AlignJob *j : leavers) {
        const size_t i = (size_t)(j - &jobs[0]);
        taken[i] = 0;
        cont[i] = 1;
    }
    int k_alone = 0;
    for (int i = 0; i < count; ++i)
        if (!taken[i] && jobs[i].phase != 2) ++k_alone;
    // (at most kAloneLive of them in flight: with more, the share of each no longer holds its first records -- 12 per call were
    // slower than 8 --; the others begin as these end)
    const int kAloneLive = std::max<int>(8, (int)leavers.size());   // (a call's tail leaves the engines together: all of it goes on at once)
    std::vector<char> begun((size_t)count, 0);
    auto begin_job = [&](int i) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, jobs[i].ctx->device) != hipSuccess || cus <= 0) cus = RUN_G + 8;
        const int share = std::min(k_alone, kAloneLive);
        jobs[i].ctx->run_g_call = share > 1 ? std::max(8, (cus - 8) / share - 1) : RUN_G;
        jobs[i].ctx->call_no_spec = share > 2;
        jobs[i].ctx->crowded = false;
        jobs[i].ctx->lone = true;
        jobs[i].paced_nb = true;   // (job_pump's paced steps, one look per call: resident runs for registrations on their own here too)
        begun[(size_t)i] = 1;
        const int rc = cont[(size_t)i] ? job_continue_alone(jobs[i]) : job_begin(jobs[i]);
        if (rc) { jobs[i].rc = rc; jobs[i].phase = 2; if (!first_err) first_err = rc; }
    };
    auto is_alone = [&](int i) { return !taken[i]; };
    // round-robin: every pass tops up each registration's queue and looks at its
    // poll word without blocking; when nobody moved, block on the oldest job
    for (;;) {
        int live = 0, moved = 0, first_live = -1, waiting = 0;
        for (int i = 0; i < count; ++i)
            if (is_alone(i) && begun[(size_t)i] && jobs[i].phase != 2) ++live;
        // (the ranks of a mailbox world wait for each other: all of them begin at once, whatever their number)
        for (int i = 0; i < count; ++i)
            if (is_alone(i) && !begun[(size_t)i] && jobs[i].phase != 2 && (live < kAloneLive || multi_rank(jobs[i].ctx))) {
                begin_job(i); ++moved;
                if (jobs[i].phase != 2) ++live;
            }
        live = 0;
        for (int i = 0; i < count; ++i) {
            if (jobs[i].phase == 2) continue;
            if (is_alone(i) && !begun[(size_t)i]) { ++waiting; continue; }
            const int before_phase = jobs[i].phase, before_checked = jobs[i].checked + jobs[i].batches;
            if (job_pump(jobs[i], false)) {
                if (jobs[i].rc && !first_err) first_err = jobs[i].rc;
                ++moved;
                continue;
            }
            ++live;
            if (first_live < 0) first_live = i;
            if (jobs[i].phase != before_phase || jobs[i].checked + jobs[i].batches != before_checked) ++moved;
        }
        if (live == 0 && waiting == 0) break;
        if (!moved && first_live >= 0) {
            if (job_pump(jobs[first_live], true) && jobs[first_live].rc && !first_err)
                first_err = jobs[first_live].rc;
        }
    }
    return first_err;
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


}   // extern "C"
