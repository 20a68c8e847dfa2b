// cvo_job.cpp -- align() (ref src/cvo.cpp:361-420, src/adaptive_cvo.cpp:490-555) as a resumable job: begin / pump /
// finish, so that one host thread can keep many registrations in flight; cvo_hip_align is one job pumped to its end.
#include "cvo_internal.h"

using namespace cvo_dev;
using namespace cvo_impl;

namespace cvo_impl {

// How this context's align() runs (decided per align() from the clouds' sizes and from who shares the launches): blocks of the
// list kernels, asynchronous list builds.
void decide_scheme(cvo_hip_ctx *ctx)
{
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
    {
        // (with resident runs the narrow iterations no longer run these launches: the wide ones want the blocks -- 10k x 10k
        // 256 / 512 / 1024 blocks per pass 752 / 869 / 913 registrations/s, 6k 950 / 966 / 950, 3k 1 176 / 1 157 / 1 074: profiles/r05_ab.txt 7)
        const bool runs = runs_allowed(ctx) && ctx->fixed.np <= 65536 && ctx->moving.np <= 65536;
        ctx->proc_blocks = ctx->proc_blocks_default =
            npairs <= 2.5e7 ? PROC_BLOCKS / 4 : (npairs <= (runs ? 6.0e7 : 1.5e8) ? PROC_BLOCKS / 2 : PROC_BLOCKS);
    }
}

// (how often the host found the final head's pinned copy incomplete when the `done` word was there: cvo_hip_get_mirror_retries)
std::atomic<long long> &mirror_retries()
{
    static std::atomic<long long> n{0};
    return n;
}

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
    if (ctx->no_run_backoff > 0) --ctx->no_run_backoff;   // (a resident run of this context timed out not long ago: runs_allowed)
    if (p.mode == CVO_HIP_MODE_ACVO) {   // tail of acvo::set_pcd (ref src/adaptive_cvo.cpp:476-478)
        s->ell = p.ell_init;
        s->ell_max = p.ell_max_init;
    }
    *ctx->done_mirror = 0;
    *ctx->progress_mirror = 0;
    *ctx->run_mirror = 0;
    *ctx->hint_mirror = -1;
    *ctx->side_mirror = 0;
    j.side_seen = 0;
    j.side_launched = 0;
    if (ctx->final_mirror) ctx->final_mirror->done = RUNNING;   // (an old verdict must not pass for this registration's)
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
    // initial device state: zeros but for what the object carries and the filter's geometry -- by value into the prepare kernel
    PrepareInit in{};
    {
        DevState *h = &ctx->st_host[kPollSlots];   // (fill_filter_geometry writes a state's head)
        const int rcg = fill_filter_geometry(ctx, h);
        if (rcg) return rcg;
        in.on = 1;
        std::memcpy(in.R, s->R, sizeof(in.R));
        std::memcpy(in.T, s->T, sizeof(in.T));
        in.ell = s->ell;
        in.ell_max = s->ell_max;
        in.iter = s->iter;
        in.n_fixed = h->n_fixed;
        for (int q = 0; q < 3; ++q) in.center[q] = h->center[q];
        in.xmax = h->xmax; in.y0max = h->y0max;
        in.done = p.max_iter <= 0 ? DONE_MAX_ITER : RUNNING;
    }
    decide_scheme(ctx);
    // (a member of a fused group: the group's table, armed by its insert; on its own: this context's table, whose build masks the
    // prepare kernel sets -- it exists from the first align() on, and its first use arms it anyway)
    launch_prepare(ctx->st, loop_params(ctx), loop_stream(ctx), (!j.in_group && ctx->table.raw) ? ctx->table.masks() : nullptr, &in);
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
    j.runs_enq = 0;
    j.run_waiting = false;
    j.spec_pending = false;
    j.executed_base = 0;
    // (a batch begins with a resident run when the record in use is expected to hold at most this many candidates -- DevHead::
    // run_hint: the record's count where the last head knew it, else an estimate with 5 % of room; a run that finds more than it can
    // hold declines, which costs its launch and one head)
    const int g_call = std::max(8, std::min(ctx->run_g_max, ctx->run_g_call));
    ctx->run_nnz_max = g_call * RUN_BLOCK * (RUN_R + RUN_L);
    // (acvo: a run holds RUN_A candidates per lane of EACH of its three records, and the hint speaks of the xy record alone -- the
    // self records of two copies of a surface hold ~1.2 x as many)
    ctx->run_small_max = 3 * RUN_G_SMALL * RUN_BLOCK;
    if (ctx->prm.mode == CVO_HIP_MODE_ACVO) {
        ctx->run_nnz_max = (int)(0.8 * g_call * RUN_BLOCK * RUN_A);
        ctx->run_small_max = (int)(0.8 * 2 * RUN_G_SMALL * RUN_BLOCK);
    }
    if (ctx->big_run_backoff > 0) --ctx->big_run_backoff;
    if (ctx->opt.run_cand > 0) ctx->run_nnz_max = ctx->opt.run_cand;   // (tuning switch "run_candidates_max")
    j.phase = p.max_iter <= 0 ? 1 : 0;
    if (j.phase == 1) {
        HIP_TRY(ctx, hipMemcpyAsync(&ctx->st_host[0], ctx->st, sizeof(DevState), hipMemcpyDeviceToHost,
                                    loop_stream(ctx)));
        HIP_TRY(ctx, hipEventRecord(ctx->poll_ev[0], loop_stream(ctx)));
    }
    return CVO_HIP_OK;
}

// A registration that has been running in an engine (cvo_engine.cpp) goes on alone, on its context's stream: the state stays (R, T, ell, the
// iteration count: the engine's last batch has completed and left it at an iteration's end), the lists are forgotten -- the engine's are
// synchronous, single-buffered and without head-mode records -- and the plan of a registration on its own, resident runs and all, takes
// over: what job_pump does after a list grew, without the growing.  The caller has seen the engine's stream idle.
int job_continue_alone(AlignJob &j)
{
    cvo_hip_ctx *ctx = j.ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    j.in_group = false;
    ctx->loop_stream = nullptr;
    ctx->crowded = false;
    ctx->lone = true;
    ctx->proc_blocks = ctx->proc_blocks_default;
    if (ctx->no_run_backoff > 0) --ctx->no_run_backoff;
    *ctx->done_mirror = 0;
    *ctx->progress_mirror = 0;
    *ctx->run_mirror = 0;
    *ctx->hint_mirror = -1;
    *ctx->side_mirror = 0;
    if (ctx->final_mirror) ctx->final_mirror->done = RUNNING;
    decide_scheme(ctx);
    hipStream_t s = loop_stream(ctx);
    HIP_TRY(ctx, hipMemsetAsync(reinterpret_cast<char *>(ctx->st) + offsetof(DevState, n_slots), 0, sizeof(int32_t), s));
    HIP_TRY(ctx, hipMemsetAsync(reinterpret_cast<char *>(ctx->st) + offsetof(DevState, run_count), 0, 4 * sizeof(int32_t), s));
    HIP_TRY(ctx, hipMemsetAsync(reinterpret_cast<char *>(ctx->st) + offsetof(DevState, run_last_entered), 0, 2 * sizeof(int32_t), s));
    // (k_prepare without the registration's first state: every list to be built, the plan made anew from R, T, ell as they stand)
    launch_prepare(ctx->st, loop_params(ctx), s, ctx->table.raw ? ctx->table.masks() : nullptr);
    HIP_TRY(ctx, hipGetLastError());
    const int prc = prepare_buffers(ctx);
    if (prc) return prc;
    ctx->head_mode = false;
    const int rc = prepare_lone_plan(ctx, 0);
    if (rc) return rc;
    j.enq = j.batches = j.checked = 0;
    j.runs_enq = 0;
    j.run_waiting = false;
    j.spec_pending = false;
    j.side_seen = 0;
    j.side_launched = 0;
    const int g_call = std::max(8, std::min(ctx->run_g_max, ctx->run_g_call));
    ctx->run_nnz_max = g_call * RUN_BLOCK * (RUN_R + RUN_L);
    ctx->run_small_max = 3 * RUN_G_SMALL * RUN_BLOCK;
    if (ctx->prm.mode == CVO_HIP_MODE_ACVO) {
        ctx->run_nnz_max = (int)(0.8 * g_call * RUN_BLOCK * RUN_A);
        ctx->run_small_max = (int)(0.8 * 2 * RUN_G_SMALL * RUN_BLOCK);
    }
    if (ctx->opt.run_cand > 0) ctx->run_nnz_max = ctx->opt.run_cand;
    j.phase = 0;
    ++ctx->tail_handovers;
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
        if (ctx->opt.comm_debug)
            fprintf(stderr, "[cvo_hip] rank %d of %d: exchange timed out at iteration k = %d (executed %d), mail_seq %llu, mail_snap %llu, slots %d, twist in launch %d\n",
                    ctx->mail_rank, ctx->mail_world, f.k, f.n_exec, (unsigned long long)f.mail_seq, (unsigned long long)f.mail_snap, f.n_slots,
                    (int)(ctx->plan.size() == 4));
        return fail(ctx, CVO_HIP_ERR_COMM, "mailbox all-reduce timed out: a peer rank never delivered its partial sums "
                                           "(the mailboxes must be created and connected again on every rank)");
    }
    if (f.done == DONE_RUN_TIMEOUT)   // (job_pump registers again without runs; this is the second time-out in a row, which cannot be a run's)
        return fail(ctx, CVO_HIP_ERR_RUN, "a resident run timed out and the registration could not be redone without runs");
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
    if (j.phase == 0 && (block || j.paced_nb) && (j.paced || j.paced_nb) && !host_reduce(ctx) && !ctx->profiling) {
        // (j.paced: cvo_hip_align, the calling thread sits here until the loop stops.  j.paced_nb: a registration on its own inside
        // cvo_hip_align_many -- the same steps, one look per call without blocking, a short bounded wait with it: the caller has
        // other registrations to pump)
        const int limit = (ctx->use_async ? 3 : 1) * ctx->prm.max_iter + 4 * kBatch;
        unsigned spins = 0;
        int idle_seen = 0;
        const auto t_enter = std::chrono::steady_clock::now();
        for (;;) {
            if (*(volatile int32_t *)ctx->done_mirror != RUNNING) break;
            // (the run counter first: a run publishes its slots before it reports its end)
            const int runs_word = *(volatile int32_t *)ctx->run_mirror;
            const int runs = runs_word & (RUN_MIRROR_ABORTED - 1);
            const int slots = *(volatile int32_t *)ctx->progress_mirror;
            // a run asks for its next list to be built beside it (kt_run "side builds"): two launches on the side stream
            if (!ctx->plan_side.empty()) {
                const int want = *(volatile int32_t *)ctx->side_mirror;
                if (want != 0 && want != j.side_seen) {
                    j.side_seen = want;
                    for (const TLaunch &l : ctx->plan_side) launch_table(ctx->table.dev, l, ctx->side_stream);
                    if (hipGetLastError() != hipSuccess) return finish_with(fail(ctx, CVO_HIP_ERR_HIP, "launching a side build failed"));
                    ++j.side_launched;
                    ++ctx->side_builds_launched;
                }
            }
            // (head mode without a flush: the post-step part of a batch's last slot runs in the head of the NEXT
            // batch's first launch, so the next batch must be on its way before the running one ends -- it goes
            // out when the running batch is down to its last slots; the GPU never idles between batches, and
            // a registration that stops in those last slots leaves one batch of launches that return at once)
            const int lead = ctx->head_mode ? 2 : 0;
            bool go;
            if (j.run_waiting) {
                // a RUN batch: its length is the run's business; the next batch goes out when the run reports its end --
                // the batch's kRunBatchSlots classic slots are what is left then
                go = runs >= j.runs_enq;
                if (go) {
                    // (the run that was sent on spec behind the first two slots: did it carry slots, or decline?)
                    // (a run that gave up at its entry hand-shake: something else -- another thread's registration, most likely -- holds
                    // the compute units.  Runs of more than 32 solvers stay away for the next registrations: each try costs its 200 us wait)
                    if (runs_word & RUN_MIRROR_ABORTED) { ctx->big_run_backoff = kBigRunBackoff; ++ctx->run_aborts; }
                    else if (j.spec_pending) ctx->spec_first_run = (runs_word & RUN_MIRROR_ENTERED) != 0;
                    j.spec_pending = false;
                    j.enq = slots + kRunBatchSlots; j.run_waiting = false;
                }
            } else {
                go = j.enq - slots <= lead;
            }
            if (go) {
                if (j.enq >= limit) break;   // cannot happen
                const int hint = *(volatile int32_t *)ctx->hint_mirror;
                // A registration's first list is its widest.  Where the last registration of this context could run on it (or nothing is
                // known and the clouds are small enough to try) the batch behind the first two slots -- the build, then the iteration
                // whose flow pass records the list's candidates -- begins with a run ON SPEC: the hint that would say so comes two slots
                // later.  A run that finds the record too large declines (its launch and one head: ~5 us) and the context stops trying
                // until a first record fits again.
                const bool run_plan_ = ctx->head_mode && !ctx->plan_pre.empty();
                const bool first_choice = run_plan_ && j.batches == 1 && j.enq == kShortBatch && j.runs_enq == 0;
                if (run_plan_ && j.batches == 2 && j.runs_enq == 0 && !ctx->spec_first_run && hint > 0) ctx->spec_first_run = hint <= ctx->run_nnz_max;
                const bool big_ok = ctx->big_run_backoff <= 0;   // (see above)
                const bool spec = first_choice && ctx->spec_first_run && big_ok && !ctx->call_no_spec;
                if (spec) j.spec_pending = true;
                const bool with_run = run_plan_ && (spec || (hint > 0 && hint <= (big_ok ? ctx->run_nnz_max : ctx->run_small_max)));
                // (a plan with runs is launched eagerly and in the shortest batches: a run can only start at a batch's head, and the
                // slot it may start at is two or three slots after the head that first says so)
                const bool near_run = ctx->head_mode && !ctx->plan_pre.empty() && !with_run;
                const bool small_run = with_run && !spec && hint > 0 && hint <= ctx->run_small_max;   // (a launch of 33 blocks does)
                // (two runs in a row where the record is narrow enough for side builds: RUN_G_SIDE solvers, three candidates per lane)
                const bool two_runs = with_run && !spec && !ctx->plan_side.empty() && hint > 0 &&
                                      hint <= (int)((ctx->prm.mode == CVO_HIP_MODE_ACVO ? 0.8 : 1.0) * RUN_G_SIDE * RUN_BLOCK * 3);
                const int rc = launch_batch(ctx, j.executed_base + j.enq, j.trace_cap, with_run, near_run ? kShortBatch : kBatch, small_run, two_runs);
                if (rc) return finish_with(rc);
                if (with_run) { j.runs_enq += two_runs ? 2 : 1; j.run_waiting = true; }
                else j.enq += near_run ? kShortBatch : kBatch;
                ++j.batches;
                spins = 0;
                idle_seen = 0;
                j.idle_seen = 0;
                if (!j.paced) return 0;   // (one step per call: the caller's other registrations want their turn)
            } else {
                if (!j.paced) {
                    if (!block) return 0;
                    if (std::chrono::steady_clock::now() - t_enter > std::chrono::microseconds(30)) {
                        // (the caller found nobody moving and asked this job to wait: the same look at the stream as below --
                        // an error, or idle twice in a row with the mirrors where they were, ends the job)
                        const hipError_t q = hipStreamQuery(loop_stream(ctx));
                        if (q != hipSuccess && q != hipErrorNotReady)
                            return finish_with(fail(ctx, CVO_HIP_ERR_HIP, "the stream of the align loop reports an error"));
                        if (q == hipSuccess && *(volatile int32_t *)ctx->done_mirror == RUNNING &&
                            *(volatile int32_t *)ctx->progress_mirror == slots && *(volatile int32_t *)ctx->run_mirror == runs_word) {
                            if (++j.idle_seen >= 2)
                                return finish_with(fail(ctx, CVO_HIP_ERR_HIP, "the align loop's stream went idle without progress"));
                        } else {
                            j.idle_seen = 0;
                        }
                        return 0;
                    }
                    __builtin_ia32_pause();
                    continue;
                }
                // (how the calling thread waits between two looks: it spins by default -- one core per concurrent cvo_hip_align, the
                // shortest reaction --; "wait_policy" 1 gives the core away between looks, 2 naps for 50 us)
                if (ctx->opt.wait_policy == 0) __builtin_ia32_pause();
                else if (ctx->opt.wait_policy == 1) std::this_thread::yield();
                else std::this_thread::sleep_for(std::chrono::microseconds(50));
                // The mirrors only move while the queued kernels run.  A fault, a stream in an error state or a
                // post kernel that never ran would leave this thread spinning for ever: now and then ask the
                // stream itself (a batch lasts ~0.25 ms; 2^14 pauses are about that long).
                if ((++spins & (ctx->opt.wait_policy == 0 ? 0x3fffu : (ctx->opt.wait_policy == 1 ? 0x3ffu : 0x7u))) == 0u) {
                    const hipError_t q = hipStreamQuery(loop_stream(ctx));
                    if (q != hipSuccess && q != hipErrorNotReady)
                        return finish_with(fail(ctx, CVO_HIP_ERR_HIP, "the stream of the align loop reports an error"));
                    // idle, yet the batch has not reported all its slots and nothing stopped: seen twice in a row
                    // (the mirrors are written before a kernel ends, so once is already conclusive; twice is cheap)
                    if (q == hipSuccess && *(volatile int32_t *)ctx->done_mirror == RUNNING &&
                        *(volatile int32_t *)ctx->progress_mirror == slots && *(volatile int32_t *)ctx->run_mirror == runs_word) {
                        if (++idle_seen >= 2)
                            return finish_with(fail(ctx, CVO_HIP_ERR_HIP, "the align loop's stream went idle without progress"));
                    } else {
                        idle_seen = 0;
                    }
                }
            }
        }
        {
            // The loop has stopped.  With a verdict and a plan whose heads mirror the final head (one registration on its own):
            // everything job_finish reads is in pinned memory already -- written in front of the `done` word -- and the call
            // returns without a copy behind the launches of the batch that are still queued (they return at their first load;
            // whatever uses the stream next is ordered behind them).  Else: the state comes by a copy in stream order.
            const int32_t verdict = *(volatile int32_t *)ctx->done_mirror;
            const bool mirrored = (verdict == DONE_BREAK_A || verdict == DONE_BREAK_B || verdict == DONE_MAX_ITER) && ctx->final_mirror &&
                                  ctx->table.image.size() == 1 && ctx->plan_has_final_mirror;
            if (mirrored) {
                // (the check word: a piece of the copy that has not landed yet -- seen on this platform although the device fences
                // at system scope between the copy and the `done` word -- is a retry; after ~50 us the copy in stream order below)
                const auto t0 = std::chrono::steady_clock::now();
                for (;;) {
                    std::atomic_thread_fence(std::memory_order_acquire);
                    std::memcpy(&ctx->st_host[0], ctx->final_mirror, sizeof(DevHead));
                    const uint32_t *w = reinterpret_cast<const uint32_t *>(&ctx->st_host[0]);
                    constexpr int pieces = (int)(sizeof(DevHead) / 16);
                    uint32_t sum = 0;
                    for (int q = 0; q < pieces; ++q)
                        sum += head_check_mix(w[4 * q], q == pieces - 1 ? 0u : w[4 * q + 1], w[4 * q + 2], w[4 * q + 3], (unsigned)q);
                    if (ctx->st_host[0].done == verdict && sum == (uint32_t)ctx->st_host[0].head_check_) return finish_with(job_finish(j));
                    mirror_retries().fetch_add(1, std::memory_order_relaxed);
                    if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(50)) break;
                    __builtin_ia32_pause();
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
    if (ctx->opt.comm_debug)
        fprintf(stderr, "[cvo_hip] final state: done %d, side mirror %d (seen %d, launched %d), plan_side %zu\n", cur.done, *(volatile int32_t *)ctx->side_mirror,
                j.side_seen, j.side_launched, ctx->plan_side.size());
    if (cur.done == DONE_RUN_TIMEOUT && j.restarts == 0 && !j.in_group && !ctx->opt.no_restart) {
        // A resident run gave up on an exchange (kt_run: a block of it did not arrive within the limit -- GPU scheduling, most
        // likely another process's kernels on the compute units).  Nothing of the run is in the state and the caller's object
        // has not been touched: the registration is begun again, this time -- and for this context's next registrations --
        // without runs.  Same result, bit for bit (runs change nothing); the frame is not lost.  The run's exchange rows may hold
        // words of the exchanges some blocks were ahead by: the next run's numbers start well past them.
        ++ctx->run_timeouts;
        ++j.restarts;
        ctx->no_run_backoff = kNoRunBackoff + 1;   // (job_begin takes one off)
        const unsigned long long seq = cur.run_seq + 4096ull;
        if ((ctx->side_stream && hipStreamSynchronize(ctx->side_stream) != hipSuccess) || hipStreamSynchronize(loop_stream(ctx)) != hipSuccess ||
            hipMemcpy(reinterpret_cast<char *>(ctx->st) + offsetof(DevState, run_seq), &seq, sizeof(seq), hipMemcpyHostToDevice) != hipSuccess)
            return finish_with(fail(ctx, CVO_HIP_ERR_RUN, "a resident run timed out and the state could not be reset"));
        const int rc = job_begin(j);
        if (rc) return finish_with(rc);
        return 0;
    }
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
    *ctx->run_mirror = 0;
    *ctx->hint_mirror = -1;
    *ctx->side_mirror = 0;
    j.side_seen = 0;
    j.runs_enq = 0;
    j.run_waiting = false;
    j.spec_pending = false;
    if (hipMemset(reinterpret_cast<char *>(ctx->st) + offsetof(DevState, n_slots), 0, sizeof(int32_t)) != hipSuccess ||
        hipMemset(reinterpret_cast<char *>(ctx->st) + offsetof(DevState, run_count), 0, sizeof(int32_t)) != hipSuccess ||
        hipDeviceSynchronize() != hipSuccess)
        return finish_with(fail(ctx, CVO_HIP_ERR_HIP, "resume failed"));
    launch_prepare(ctx->st, loop_params(ctx), ctx->stream, ctx->table.raw ? ctx->table.masks() : nullptr);   // idempotent; re-zeroes the counters
    if (hipGetLastError() != hipSuccess) return finish_with(fail(ctx, CVO_HIP_ERR_HIP, "resume failed"));
    if (!ctx->profiling && !host_reduce(ctx)) {   // the lists moved: new arguments
        rc = prepare_lone_plan(ctx, j.trace_cap);
        if (rc) return finish_with(rc);
    }
    j.enq = j.batches = j.checked = 0;
    j.phase = 0;
    return 0;
}


}   // namespace cvo_impl

extern "C" {

int cvo_hip_align(cvo_hip_ctx *ctx, cvo_hip_state *s, cvo_hip_trace *trace, int trace_cap,
                  int *n_iter)
{
    cvo_lock::Api api_guard;
    if (!ctx || !s) return CVO_HIP_ERR_INVALID;
    AlignJob j;
    j.ctx = ctx; j.s = s; j.trace = trace; j.trace_cap = trace_cap; j.n_iter = n_iter;
    j.paced = true;
    ctx->run_g_call = RUN_G;   // (the GPU is this registration's: no neighbours to leave room for)
    ctx->call_no_spec = false;
    int rc = job_begin(j);
    if (rc) return rc;
    while (!job_pump(j, true)) {}
    return j.rc;
}


}   // extern "C"
