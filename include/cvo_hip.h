/*
 * cvo_hip.h -- C-ABI of the MI355X (gfx950) back end for the CVO / Adaptive-CVO
 * registration inner loop.
 *
 * This is the drop-in boundary: the four private calls
 *     update_tf(); transform_pcd(); compute_flow(); compute_step_size();
 * inside cvo::cvo::align() (ref cpp/rkhs_registration/src/cvo.cpp:366-377;
 * acvo: src/adaptive_cvo.cpp:495-506) and the cloud hand-over at the tail of
 * set_pcd() (ref src/cvo.cpp:344-356).  Plain pointers and sizes only; no
 * Eigen, OpenCV or torch types cross it.  INTEGRATION.md shows the edits a
 * maintainer of the reference makes to bind it.
 *
 * Conventions
 *   - every function returns int: 0 = ok, <0 = cvo_hip_status (never throws);
 *     cvo_hip_error_string() gives text, cvo_hip_last_error(ctx) detail.
 *   - the caller owns all host arrays; the context owns all device memory.
 *   - a context is bound to one device and one HIP stream and is NOT
 *     thread-safe; distinct contexts are independent (batched mode = one
 *     context per stream, ref SURVEY 8e).
 *   - clouds: xyz is AoS n x 3 float32 (std::vector<Eigen::Vector3f>,
 *     ref include/data_type.h:30,63); features n x 5 float32, either
 *     column-major (Eigen::Matrix<float,Dynamic,5>, ref data_type.h:64) or
 *     row-major, selected by feat_layout.
 *   - matrices R (3x3) and the 4x4 transforms are ROW-major in this ABI.
 */
#ifndef CVO_HIP_H
#define CVO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CVO_HIP_NFEAT 5

typedef enum cvo_hip_status {
    CVO_HIP_OK = 0,
    CVO_HIP_ERR_INVALID = -1,   /* bad argument / call order */
    CVO_HIP_ERR_HIP = -2,       /* a HIP runtime call failed */
    CVO_HIP_ERR_NOMEM = -3,
    CVO_HIP_ERR_COMM = -4,      /* RCCL failure */
    CVO_HIP_ERR_NODEVICE = -5,  /* no usable gfx950 device */
    CVO_HIP_ERR_RUN = -6        /* a resident run timed out AND the registration could not be redone without runs
                                 * (a time-out alone is not an error: the pair is registered again on the launch-per-pass
                                 * path, same result) */
} cvo_hip_status;

/* CVO_HIP_MODE_MATLAB is accepted by cvo_hip_default_params() only: it returns mode = CVO with
 * the constants of the reference's MATLAB object and color_scale > 0 (SURVEY 8 a9,
 * ref matlab/@rkhs_se3_registration/rkhs_se3_registration.m:10-28). */
enum { CVO_HIP_MODE_CVO = 0, CVO_HIP_MODE_ACVO = 1, CVO_HIP_MODE_MATLAB = 2 };
enum { CVO_HIP_FEAT_COLMAJOR = 0, CVO_HIP_FEAT_ROWMAJOR = 1 };

/* Hyper-parameters.  Defaults = the reference's constructor initialisers
 * (ref src/cvo.cpp:18-48, src/adaptive_cvo.cpp:18-50). */
typedef struct cvo_hip_params {
    int32_t mode;        /* CVO_HIP_MODE_* */
    int32_t max_iter;    /* MAX_ITER */
    float ell_init;
    float ell_min;
    float ell_max_init;
    float sigma;
    float sp_thres;
    float c_sp_thres;
    float c;
    float d;
    float c_ell;
    float c_sigma;
    float min_step;
    float eps;
    float eps_2;
    float color_scale;   /* 0: the C++ pair weight (ref src/cvo.cpp:143-153).  > 0: the MATLAB object's:
                          * a = color_scale * <c_i, c_j> * K with K = sigma^2 exp(-d2 / 2 ell^2) kept iff
                          * K >= sp_thres, c = features 0..2 -- linear colour inner product, no colour
                          * cut-off (ref rkhs_se3_registration.m:40-73,125-127) */
    double dl_step;
} cvo_hip_params;

/* Per-object state the reference keeps in cvo::cvo members and carries from
 * frame to frame (R, T, ell are never reset in cvo: SURVEY 8a quirks 1-4). */
typedef struct cvo_hip_state {
    float R[9];
    float T[3];
    float ell;
    float ell_max;
    float transform[16];        /* ref cvo.hpp:104 */
    float prev_transform[16];   /* ref cvo.hpp:105 */
    float accum_transform[16];  /* ref cvo.hpp:106 */
    int32_t iter;               /* ref cvo.hpp:103 */
    int32_t pad_;
} cvo_hip_state;

/* One record per executed iteration of align() (the parity artefact; the
 * reference only prints, ref src/cvo.cpp:382,404). */
typedef struct cvo_hip_trace {
    int32_t k;
    int32_t exit_code;   /* 0 continued, 1 break on twist norms, 2 break on se3 distance */
    float ell;
    float step;
    float dist;
    float pad_;
    float omega[3];
    float v[3];
    double omega_d[3];
    double v_d[3];
    double bcde[4];
    double sum_a;
    double dl;
    int64_t nnz;
    int64_t nnz_xx;
    int64_t nnz_yy;
} cvo_hip_trace;

/* Accumulated HIP-event timings (profiling mode: one registration at a time, eager launches,
 * an event pair attached to each dispatch = the kernel's own begin / end timestamps).  Only
 * launches that did work are counted (not the ones queued past convergence, not a k_filter
 * whose list was re-used). */
typedef struct cvo_hip_profile {
    double flow_ms;        /* k_filter on the (fixed x moving) pair set: the all-pairs test */
    int64_t flow_launches;
    double flow_pairs;     /* pair tests those launches stand for (rows x padded columns) */
    double step_ms;        /* the step-size pass over the members of A (k_step_twist / k_process<PROC_STEP>) */
    int64_t step_launches;
    double step_pairs;
    double self_ms;        /* acvo: k_filter on (x, x) and (y, y) */
    int64_t self_launches;
    double self_pairs;
    double proc_flow_ms;   /* k_process<PROC_FLOW>: exact test + flow sums over the candidate list */
    int64_t proc_flow_launches;
} cvo_hip_profile;

typedef struct cvo_hip_ctx cvo_hip_ctx;

const char *cvo_hip_error_string(int status);
const char *cvo_hip_last_error(const cvo_hip_ctx *ctx);
int cvo_hip_device_count(int *count);

/* ref src/cvo.cpp:18-48 / src/adaptive_cvo.cpp:18-50 (constructors). */
int cvo_hip_default_params(int mode, cvo_hip_params *p);
int cvo_hip_init_state(const cvo_hip_params *p, cvo_hip_state *s);

/* stream: a hipStream_t (e.g. torch.cuda.current_stream().cuda_stream) or NULL
 * to let the context create its own non-blocking stream. */
int cvo_hip_create(int device, void *stream, const cvo_hip_params *p, cvo_hip_ctx **out);
int cvo_hip_destroy(cvo_hip_ctx *ctx);
int cvo_hip_set_params(cvo_hip_ctx *ctx, const cvo_hip_params *p);

/* Cloud hand-over: tail of set_pcd() (ref src/cvo.cpp:344-356).  At most 2^26 points per cloud
 * (CVO_HIP_ERR_INVALID beyond).  The arrays are copied before the call returns (they are the caller's
 * again at once); the upload and the preparation on the device (Morton order, bounding spheres) are only
 * queued on the context's stream -- the next call that computes with the cloud waits for them, so the
 * hand-overs of many contexts overlap.  CVO_HIP_SYNC_UPLOAD=1: wait before returning. */
int cvo_hip_set_fixed(cvo_hip_ctx *ctx, const float *xyz, const float *feat, int n,
                      int feat_layout);
int cvo_hip_set_moving(cvo_hip_ctx *ctx, const float *xyz, const float *feat, int m,
                       int feat_layout);
/* ptr_fixed_pcd = std::move(ptr_moving_pcd) (ref src/cvo.cpp:417): device
 * buffers are swapped, nothing is re-uploaded. */
/* The same with the cloud already in device memory (same device as the context), e.g. as
 * the front end leaves it (cvo_fe_device_cloud, cvo_frontend.h): nothing but the bounding
 * box (24 bytes) crosses PCIe.  The arrays may be re-used when the call returns. */
int cvo_hip_set_fixed_device(cvo_hip_ctx *ctx, const float *d_xyz, const float *d_feat, int n,
                             int feat_layout);
int cvo_hip_set_moving_device(cvo_hip_ctx *ctx, const float *d_xyz, const float *d_feat, int m,
                              int feat_layout);
int cvo_hip_swap_moving_to_fixed(cvo_hip_ctx *ctx);

/* The hand-over of a whole batch -- the tail of set_pcd() (ref src/cvo.cpp:344-356) for `count` registration
 * objects at once: context k receives fixed cloud k (fixed_xyz[k], fixed_feat[k], n_fixed[k] points) and moving
 * cloud k; a null fixed_xyz (or a null entry of it) leaves that context's fixed cloud as it is (a streamed
 * sequence: cvo_hip_swap_moving_to_fixed first).  Host arrays, the layouts of cvo_hip_set_fixed.  One transfer
 * and ONE kernel launch serve all clouds of up to 16384 points (a block per cloud: bounding box, Morton keys,
 * sort, packed rows, bounding spheres); larger clouds go the way of cvo_hip_set_fixed / _set_moving.  The
 * arrays are copied into the library's own page-locked staging (by a few host threads, in pieces whose transfers
 * overlap the staging of the next piece) and are free when the call returns.  The call does not wait for the
 * device: a cloud's hand-over ends when the next compute entry point of its context needs it.  All contexts
 * must be on one device.  Every argument of the whole batch is checked before any context is touched; after an error further on
 * (out of memory, a transfer or launch that does not go out) the clouds of the batch that were not filled are left EMPTY -- the
 * next compute entry point of their context fails on them -- and the batch must be handed over again. */
int cvo_hip_set_pcd_many(cvo_hip_ctx *const *ctxs, const float *const *fixed_xyz, const float *const *fixed_feat,
                         const int *n_fixed, const float *const *moving_xyz, const float *const *moving_feat,
                         const int *n_moving, int feat_layout, int count);

/* Inspection (tests): the device arrays of a context's cloud as the kernels read them -- which: 0 fixed,
 * 1 moving; pos4 rows x 4 floats (x, y, z, 5th feature), feat8 rows x 8, seg4 ceil(rows / 64) x 4 (bounding
 * spheres); any of the three may be null.  *rows = padded row count, *points = the caller's count. */
int cvo_hip_get_device_cloud(cvo_hip_ctx *ctx, int which, float *pos4, float *feat8, float *seg4, int *rows,
                             int *points);

/* The cloud preparation of the reference's MATLAB driver, on the device:
 * pcRangeFilter (ref util/pcRangeFilter.m:5-12: points whose float32 range is above max_range or
 * below min_range are dropped; max_range <= 0: no range filter), then
 * pcdownsample(cloud, 'gridAverage', grid_size) (ref data/rgbd_dataset/rgbddataset_rkhs.m:36-39,58:
 * one point per occupied voxel of a box grid anchored at the minimum corner of the kept points = the
 * mean location and the mean colour, voxels in lexicographic (x, y, z) index order; grid_size <= 0: no
 * downsampling, the kept points in their order).  Host arrays in and out: xyz n x 3 float32, rgb n x 3
 * uint8; xyz_out / rgb_out must hold n points, *n_out receives the count.  Context-free.
 * Points with a NaN / Inf coordinate are dropped whatever the filter (pcdownsample drops invalid points; they
 * have no voxel).  A grid so fine that the box of the kept points spans 2^63 voxels or more is refused
 * with CVO_HIP_ERR_INVALID. */
int cvo_hip_range_filter_grid_average(int device, const float *xyz, const unsigned char *rgb, int n,
                                      float max_range, float min_range, double grid_size,
                                      float *xyz_out, unsigned char *rgb_out, int *n_out);

/* Multi-GPU: this context owns target rows [row_lo,row_hi) of the fixed cloud
 * (and rows [srow_lo,srow_hi) of the moving cloud for the acvo Ayy sweep).
 * Default = everything. */
int cvo_hip_set_shard(cvo_hip_ctx *ctx, int row_lo, int row_hi, int srow_lo, int srow_hi);
/* Even contiguous split helper: rows of `n` owned by `rank` of `world`. */
int cvo_hip_shard_range(int n, int rank, int world, int *lo, int *hi);

/* RCCL all-reduce of the per-iteration partial sums (13 + 4 float64).
 * id_bytes = the 128-byte ncclUniqueId made by cvo_hip_comm_unique_id() on
 * rank 0 and shipped to the other ranks by the caller. */
int cvo_hip_comm_unique_id(void *id_bytes_128);
int cvo_hip_comm_init(cvo_hip_ctx *ctx, const void *id_bytes_128, int rank, int world);
/* Alternative: caller-supplied reduction (buf is a DEVICE pointer valid on the
 * context's stream; must be summed over ranks in place, stream-ordered). */
typedef int (*cvo_hip_allreduce_fn)(void *user, double *dev_buf, int count, void *stream);
int cvo_hip_set_allreduce(cvo_hip_ctx *ctx, cvo_hip_allreduce_fn fn, void *user);

/* Peer-to-peer mailbox all-reduce (preferred over RCCL for these <= 104-byte messages): every
 * rank owns a 4 KB mailbox in its own device memory; per reduction each rank stores its partial
 * sums + a sequence number into every rank's mailbox (peer stores over xGMI), polls its OWN
 * mailbox and adds the world's partials in rank order -- inside the kernels that reduce the
 * block partials, so an iteration has no collective launch and no stream-level wait, results
 * are bitwise the same on every rank, and batches of iterations are still captured as hipGraphs.
 * Replaces the mutex-guarded sums of ref src/cvo.cpp:201-204,283-288 and
 * src/adaptive_cvo.cpp:234-238 across GPUs.
 *   1. every rank: cvo_hip_mailbox_create(ctx, rank, world, handle, &ptr)    (world <= 16)
 *   2. the caller ships the 64-byte IPC handles (one process per GPU, e.g. with
 *      torch.distributed.all_gather_object) or the raw pointers (ranks in one process)
 *   3. every rank: cvo_hip_mailbox_connect(ctx, handles (world x 64 bytes, rank order), NULL)
 *      or (ctx, NULL, ptrs (world device pointers)); after it cvo_hip_align / _flow /
 *      _step_coeffs / _function_inner_product all-reduce through the mailboxes.
 * All ranks must issue the same sequence of those calls (SPMD).  A peer that does not show up
 * within CVO_HIP_MAILBOX_TIMEOUT_S (default 5 s) ends the call with CVO_HIP_ERR_COMM.  After
 * CVO_HIP_ERR_COMM the mailboxes of the world are UNUSABLE (the ranks' sequence numbers no longer
 * agree): further cvo_hip_align calls on that context are refused with CVO_HIP_ERR_COMM until
 * steps 1-3 have been repeated on every rank.  Nobody may destroy its context while a peer can
 * still be inside such a call. */
#define CVO_HIP_MAILBOX_HANDLE_BYTES 64
int cvo_hip_mailbox_create(cvo_hip_ctx *ctx, int rank, int world, void *ipc_handle_64, void **dev_ptr);
int cvo_hip_mailbox_connect(cvo_hip_ctx *ctx, const void *ipc_handles, void *const *dev_ptrs);

/* update_tf() + transform_pcd() (ref src/cvo.cpp:83-87,310-315). */
int cvo_hip_transform_pcd(cvo_hip_ctx *ctx, const float R[9], const float T[3]);

/* se_kernel() + compute_flow() (ref src/cvo.cpp:99-161,164-210) on the
 * current transformed cloud: float64 sums of the per-pair float32 terms.
 * out13 = { omega[3], v[3], sum_a, sum_a_d2 (acvo dl term), nnz(A),
 *           sum_xx, nnz(Axx), sum_yy_tail, nnz(Ayy) }  (last four: acvo only,
 * ref src/adaptive_cvo.cpp:154-272); already all-reduced if a communicator is
 * attached. */
int cvo_hip_flow(cvo_hip_ctx *ctx, float ell, double out13[13]);

/* compute_step_size() coefficient sums (ref src/cvo.cpp:213-289). */
int cvo_hip_step_coeffs(cvo_hip_ctx *ctx, const float omega[3], const float v[3], float ell,
                        double bcde[4]);

/* Root selection + clamps (ref src/cvo.cpp:291-307), Exp_SEK3
 * (ref src/LieGroup.cpp:159-186) and dist_se3 (ref src/cvo.cpp:71-81): the O(1)
 * host maths of the loop, exported for tests and for callers that drive the
 * iteration themselves. */
int cvo_hip_pick_step(const double bcde[4], float min_step, float *step);
int cvo_hip_exp_se3(const float omega[3], const float v[3], float dt, float dR[9], float dT[3]);
int cvo_hip_dist_se3(const float omega[3], const float v[3], float dt, float *dist);

/* align() (ref src/cvo.cpp:361-420, src/adaptive_cvo.cpp:490-555): the whole
 * gradient-flow loop on the clouds currently set.  Updates *state; writes at
 * most trace_cap trace records if trace != NULL; *n_iter = loop bodies run. */
int cvo_hip_align(cvo_hip_ctx *ctx, cvo_hip_state *state, cvo_hip_trace *trace,
                  int trace_cap, int *n_iter);

/* Batched mode (BASELINE configs[4], ref SURVEY 8e): `count` independent
 * registrations, one context each, driven concurrently by the calling thread.
 * Equivalent to calling cvo_hip_align(ctxs[i], states[i], NULL, 0, &n_iters[i])
 * for every i -- bit for bit -- but with all of them in flight at once: contexts
 * of the same device and mode share their kernel launches in groups of up to 32
 * (one grid slice per registration; up to four such groups run side by side, and a
 * registration that stops hands its slice to the next one of the call) on streams
 * owned by the library; contexts
 * that profile or are sharded over ranks run on their own streams.  All the
 * contexts' streams are idle when the call returns.  Returns the first
 * non-zero status, 0 if all succeeded. */
int cvo_hip_align_many(cvo_hip_ctx **ctxs, cvo_hip_state **states, int *n_iters, int count);

/* acvo::function_inner_product (ref src/adaptive_cvo.cpp:385-439) between the
 * fixed and the (untransformed) moving cloud at length-scale ell. */
int cvo_hip_function_inner_product(cvo_hip_ctx *ctx, float ell, float *out);
/* The reference's own signature -- float function_inner_product(point_cloud* cloud_a,
 * point_cloud* cloud_b) (ref include/adaptive_cvo.hpp:179): any two clouds (host arrays), the
 * statistic at length-scale `ell`.  Registration state is not touched: the clouds go to
 * buffers of their own, a pending set_moving() stays pending. */
int cvo_hip_function_inner_product_clouds(cvo_hip_ctx *ctx, float ell, const float *xyz_a,
                                          const float *feat_a, int na, const float *xyz_b,
                                          const float *feat_b, int nb, int feat_layout, float *out);

/* hipGraph capture of the loop's batches of iterations.  A stream capture is a process-wide
 * affair in the HIP runtime: while one thread captures, HIP calls of any other thread fail
 * ("previous error during capture") and spoil the capture.  The library serialises its own
 * entry points against its captures; it cannot do that for other HIP users in the process.
 * Default: ON for a context that created its own stream (cvo_hip_create(stream = NULL)), OFF
 * for a caller-supplied stream -- eager launches, same results, a few per cent slower.  Enable
 * it when no other thread of the process issues HIP work while align() runs (or set
 * CVO_HIP_GRAPH=1); CVO_HIP_NO_GRAPH=1 disables every capture of the library.
 * cvo_hip_align_many() captures its shared launches only if every member context allows it.
 * One registration at a time on clouds small enough for the two-launch scheme (cvo, acvo: up to
 * ~14k x 14k points) captures nothing: its few launches per iteration -- for cvo most iterations run
 * inside two or three launches altogether -- go out eagerly, which is faster there. */
int cvo_hip_set_graph_capture(cvo_hip_ctx *ctx, int enable);

/* Policy switches of a context, by name (value: 0 / 1 for switches).  What a host program may want to set:
 *   "graph_capture"      as cvo_hip_set_graph_capture
 *   "head_graphs"        1: registrations that run on their own (two launches per iteration, resident runs) go out as captured
 *                        batches as well (default 0: eager launches are faster there)
 *   "list_pass_blocks"   blocks of the list kernels of a registration on its own: 64 / 128 / 256 / 512 / 1024, 0 = by the clouds
 *   "mailbox_timeout_s"  how long an exchange waits for a peer rank (default 5; read by the next cvo_hip_mailbox_connect)
 *   "wait_policy"        how the thread inside cvo_hip_align waits between two looks at the loop's pinned progress words:
 *                        0 spin (default: one core per concurrent caller, shortest reaction), 1 sched_yield, 2 naps of 50 us
 *   "resident_runs"      0: no resident runs (csrc/cvo_kernels.hip kt_run); "run_solvers_max": their blocks at most
 *   "run_timeout_ms"     how long an exchange inside a resident run waits for a block of its launch (default 1000).  A run that
 *                        times out costs that wait, not the frame: the pair is registered again without runs, and the context
 *                        goes without runs for its next 64 registrations ("run_timeouts" counts them, read-only)
 * and the test switches of tests/ ("head_mode", "merged_launches", "async_builds", "candidate_records", "kept_pack",
 * "list_init", "list_margin", "final_mirror", "one_launch_hand_over", "small_calls_alone", "fused_groups", "engines",
 * "run_candidates_max", "run_fault", "sync_upload", "no_graph", "twist_on_shared_gpu", "comm_debug", "engine_debug"; the
 * measured probes "engine_crowd", "engine_merge_max", "narrow_merge", "narrow_blocks", all off: profiles/r06_ab.txt 16).
 * The environment variables of the same switches only set the DEFAULTS, read once when a context is
 * created.  A call that serves many contexts goes by its first context's switches.  Unknown key / bad value:
 * CVO_HIP_ERR_INVALID. */
int cvo_hip_set_option(cvo_hip_ctx *ctx, const char *key, double value);
int cvo_hip_get_option(const cvo_hip_ctx *ctx, const char *key, double *value);

/* Profiling: HIP events on the context's stream around every sweep launch. */
int cvo_hip_set_profiling(cvo_hip_ctx *ctx, int enable);
int cvo_hip_get_profile(cvo_hip_ctx *ctx, cvo_hip_profile *out, int reset);

/* Profiling of cvo_hip_align_many's shared launches (process-wide switch): while on, the fused groups
 * launch eagerly and every flow-pass launch (kt_process<PROC_FLOW>: one launch serves up to 32
 * registrations) carries a HIP event pair attached to the dispatch.  cvo_hip_get_engine_profile
 * returns the summed kernel time, the launches and the registrations those launches served
 * (sum over launches of the occupied slots); bench.py quotes its roofline on it. */
int cvo_hip_engine_profiling(int enable);
int cvo_hip_get_engine_profile(double *flow_ms, long long *flow_launches, double *flow_registrations, int reset);
/* The same launches one by one, in launch order (one engine at a time: with several, engine after engine as they
 * are released): kernel duration, the time from the launch's begin to the NEXT flow launch's begin on the same
 * stream (= one iteration of the engine: the length of its dependent launch chain; 0 for the last one), occupied
 * slots.  *count = launches recorded (may exceed `capacity`, which bounds what is copied). */
int cvo_hip_get_engine_flow_trace(float *dur_us, float *period_us, int *slots, int capacity, int *count, int reset);

/* Diagnostics: how evenly the last flow pass (cvo_hip_flow / the last iteration of align) spread
 * the members of A over its wavefronts: members_per_wave[w] = pairs kept by wave w (4 waves per
 * block, blocks in launch order); *waves = entries written (<= capacity). */
int cvo_hip_get_wave_load(cvo_hip_ctx *ctx, uint32_t *members_per_wave, int capacity, int *waves);

/* Blocks until everything queued on the context's stream has finished. */
/* Diagnostics: batches of align() launched from a cached hipGraph / batches that had to be
 * captured first (a stream of frames should capture a handful of times, not per frame). */
int cvo_hip_get_graph_stats(const cvo_hip_ctx *ctx, long long *launches_from_cache, long long *captures);
/* Resident runs of the last cvo_hip_align (one cvo registration at a time runs the narrow part of its loop as whole
 * iterations inside one launch, csrc/cvo_kernels.hip kt_run): launches of that kernel that executed iterations, launches
 * that declined, iterations executed inside runs, candidate pairs of the record the last run looked at.  Diagnostics; any
 * pointer may be null. */
int cvo_hip_get_run_stats(cvo_hip_ctx *ctx, int *runs, int *declined, int *iterations, int *candidates);
/* ... and, in builds with -DCVO_RUN_CLOCKS, the ticks the first solver block of the runs spent in each phase of the loop
 * (csrc/cvo_kernels.hip kt_run: RUN_CLK; zeros otherwise). */
int cvo_hip_get_run_clocks(cvo_hip_ctx *ctx, long long clocks16[16]);
/* How often (in this process) a registration that ran on its own found the final state's pinned copy incomplete when the
 * `done` word was already there, and read it again (csrc/cvo_job.cpp job_pump: the copy carries a check word). */
long long cvo_hip_get_mirror_retries(void);
int cvo_hip_synchronize(cvo_hip_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* CVO_HIP_H */
