// cvo_device.h -- device-resident state and argument blocks shared by the HIP
// kernels (cvo_kernels.hip) and their host driver (cvo_capi.cpp).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stddef.h>
#include <stdint.h>

#include "cvo_hip.h"
#include "se3_math.hpp"

namespace cvo_dev {

// Device cloud layout (DESIGN.md "Data layout in HBM"):
//   pos  : float4 per point (x, y, z, caller's index as int bits)  16 B
//   feat : 8 floats per point (f0..f4, 0, 0, 0)                    32 B
// both in Morton order.  Algorithmic bytes per point are 12 + 20 = 32 B
// (SURVEY 8d); the padding is free: a sweep reads each cloud once and reuses it
// ~N-fold on chip.
constexpr int FEAT_STRIDE = 8;       // floats per point in `feat`: f0..f4, caller index bits, 2 pad
constexpr int FEAT_INDEX_SLOT = 5;   // (f4 is also kept in pos.w, which is what the kernels read)

// Filter geometry: a block of 256 threads = 4 waves; every wave owns
// TILES_PER_WAVE MFMA row tiles of 16 target rows, the block one chunk of `jt`
// source columns (a multiple of 16: one MFMA column tile per group).
constexpr int BLOCK = 256;
constexpr int TILES_PER_WAVE = 4;
constexpr int ROWS_PER_WAVE = 16 * TILES_PER_WAVE;
constexpr int ROWS_PER_TILE = 4 * ROWS_PER_WAVE;   // rows per block
constexpr float PAD_BIG = 1.0e30f;  // filter value of padded rows / columns
constexpr int PROC_BLOCKS = 1024;   // largest grid of the list kernels (ProcessArgs::nblk, a multiple of NSUB)
constexpr int PROC_WAVES = PROC_BLOCKS * 4;

// Two kinds of lists live in HBM, both split into NSUB independent sub-lists
// (own counter, own slice of the buffer): one hot atomic counter saturates near
// 90 appends/us on this chip (MI355X_MICROARCH "dequeue"), 256 of them do not,
// and scattering appends over them round-robin balances the consumers.
//   tile list  : what k_filter emits.  One TileEntry per MFMA result register
//                of a 16x16 pair tile that has at least one pair under the
//                filter bound: the tile's first row / column, the register r
//                and the wave ballot: bit l <-> row (l>>4)*4 + r, column l&15.
//                At most 64 pairs per entry = one full wavefront of exact work.
//   kept list  : what PROC_FLOW emits: the members of A as (i, j) + weight,
//                i.e. the reference's triplets (ref src/cvo.cpp:152) in COO form;
//                PROC_STEP streams it.  It needs no atomics: wave w of PROC_FLOW
//                owns slice w (kept_cap / PROC_WAVES entries) and records its count;
//                wave w of PROC_STEP reads slice w back.
// k_filter stages its appends in LDS and reserves an exactly-sized slice of a
// sub-list with one returning atomic when the stage is full or the wave ends.
constexpr int NSUB = 256;
constexpr int PROC_PARTS = PROC_BLOCKS / NSUB;   // most blocks cooperating on one sub-list
constexpr int TILE_STAGE = 128;     // TileEntry slots staged per wave in k_filter
constexpr int DEAL_SHIFT = 2;       // ... and dealt over the sub-lists of a region in runs of 1 << DEAL_SHIFT (flush_tiles)
constexpr int PAIR_QUEUE = 128;     // compaction queue of a k_process wave
constexpr int SEG = 64;             // points per bounding-sphere segment (= rows of a filter wave)
constexpr int MAX_CSEG = 32;        // column segments of a filter block (jt <= 2048)

// x = first row of the tile, y = first column | r << 30, z/w = the 64-bit mask
typedef uint4 TileEntry;

enum ProcMode { PROC_FLOW = 0, PROC_STEP = 1, PROC_SELF = 2 };
// lists: three tile lists + the kept list
// LIST_XYB: the second buffer of the xy tile list (asynchronous builds, see plan_xy_async)
// LIST_XXB / LIST_YYB: second buffers of the acvo self lists (plan_self_async)
enum ListId { LIST_XY = 0, LIST_XX = 1, LIST_YY = 2, LIST_KEPT = 3, LIST_XYB = 4, LIST_XXB = 5, LIST_YYB = 6,
              LIST_N = 7 };
// list id of buffer `buf` of self list l (0: xx, 1: yy)
CVO_HD int self_list_id(int l, int buf) { return buf ? LIST_XXB + l : LIST_XX + l; }

// float64 partial sums a block emits per mode
constexpr int NACC_FLOW = 9;   // omega[3] v[3] sum_a sum_a_d2 nnz
constexpr int NACC_STEP = 4;   // B C D E
constexpr int NACC_SELF = 2;   // sum (1/l^3 a) d2 over counted rows, nnz
constexpr int NACC_MAX = 9;

// Reduced sums (what is all-reduced across ranks): see cvo_hip_flow()
//   red[0..2] omega  [3..5] v  [6] sum_a  [7] sum_a_d2  [8] nnz
//   red[9] sum_xx  [10] nnz_xx  [11] sum_yy_tail  [12] nnz_yy
//   red[13..16] B C D E
constexpr int RED_FLOW = 0, RED_XX = 9, RED_YY = 11, RED_STEP = 13, RED_N = 17;

// DevState::done values
enum { RUNNING = 0, DONE_BREAK_A = 1, DONE_BREAK_B = 2, DONE_MAX_ITER = 3, NEED_BIGGER_LIST = 4,
       DONE_COMM_ERROR = 5,
       DONE_RUN_TIMEOUT = 6 };   // a resident run (kt_run) gave up on an exchange: nothing of it was written to the state but scratch (kt_run
                                 // "what a run writes before its exit"); the host registers again without runs (cvo_job.cpp job_pump)

// Peer-to-peer mailbox all-reduce: every rank owns one Mailbox in ITS OWN device
// memory with one slot per sender (two generations: a sender may be one exchange ahead of
// the slowest reader, never two).  An exchange = every rank stores its partial sums and then
// the sequence number into slot[seq & 1][my rank] of EVERY rank's mailbox (peer stores over
// xGMI, own store locally), then polls the slots of its own mailbox -- local memory, no
// remote reads -- until all carry `seq`, and adds them up in rank order: every rank gets the
// same bits whatever the arrival order.  It runs inside the post kernels: no collective
// launch, no stream-level synchronisation, two one-way link latencies per iteration.
constexpr int MAX_WORLD = 16;
constexpr int MAIL_VALS = 14;      // >= the 13 flow-side sums
struct alignas(128) MailSlot {
    double v[MAIL_VALS];
    unsigned long long seq;
    unsigned long long pad_;
};
struct Mailbox {
    MailSlot slot[2][MAX_WORLD];
    unsigned long long owner_dev;   // which GPU the owner's kernels run on (a hash of its PCI bus id, never 0), written at creation:
                                    // ranks that SHARE a GPU must not all spin in every block of a launch (cvo_hip_mailbox_connect)
    unsigned long long pad_[15];
};
struct CommTable {
    Mailbox *peer[MAX_WORLD];      // peer[rank] = this rank's own mailbox
    int rank, world;
    long long timeout_ticks;       // of the 100 MHz wall clock
};

// Resident runs (cvo_kernels.hip kt_run): RUN_G workgroups of RUN_BLOCK threads carry whole iterations of ONE
// registration in one launch; between the passes they exchange their partial sums through a RunMail in device
// memory -- every double as two 8-byte words (tag32 << 32 | half32), relaxed agent-scope atomic stores; a reader
// polls the words themselves until the tags match (no flag, no fence).  FOUR generations (seq & 3): among the solvers
// two would do -- a solver can post exchange e + 2 only after it has read every row of e + 1, which every solver
// posts after reading e -- but the head block only reads, and the solvers wait for it through the verdict word of
// every SECOND (step) exchange alone: a solver that has seen the verdict of step exchange 2i knows that the head is
// past step exchange 2i - 2, no more, and may post flow exchange 2i + 1 while the head still sweeps flow exchange
// 2i - 1 -- same parity.  With four generations the row a solver overwrites with e + 4 is one the head left at
// least one verdict ago.  1.7 us per exchange among 32 blocks of one XCD
// (tools/microbench/xcd_exchange.hip, profiles/r05_ab.txt 1).
constexpr int RUN_G = 248;           // solver blocks of a run at most (blocks 1 .. RUN_G of the launch; block 0 is the head block): a block
                                     // per compute unit and a few units to spare -- above RUN_G_SMALL a run first proves that all of it is resident
constexpr int RUN_G_SMALL = 32;      // runs of up to this many solvers start without the entry handshake (eight of them fit the GPU side by side)
constexpr int RUN_BLOCK = 512;       // 8 waves = two per SIMD: 256 vector registers per lane, the run's candidates live there
constexpr int RUN_WAVES = RUN_BLOCK / 64;
constexpr int RUN_LANES = RUN_G * RUN_BLOCK;
constexpr int RUN_R = 8;             // candidates per lane in registers: RUN_LANES * RUN_R = 1 015 808 candidates
constexpr int RUN_MIRROR_ABORTED = 1 << 29;   // ... the run that reported last gave up at its entry hand-shake: something else held the compute units
constexpr int RUN_MIRROR_ENTERED = 1 << 30;   // in the host's run mirror: the run that reported last carried slots (it did not decline)
constexpr int RUN_L = 8;             // ... and in LDS behind them (the widest runs only: 2 x 16 bytes per candidate, 128 KB per block)
constexpr int RUN_CAP = RUN_LANES * (RUN_R + RUN_L);   // candidates a run holds at most: 2 031 616
constexpr unsigned RUN_LDS_BYTES = (unsigned)RUN_L * 2u * (unsigned)RUN_BLOCK * 16u;   // dynamic LDS of kt_run
constexpr int RUN_NV = 13;           // doubles per exchange at most (flow: 9, acvo 9 + 2 + 2; step: 4): the stride of a row of the mail
#ifndef CVO_RUN_A
#define CVO_RUN_A 4   // (3: no vector register spilled instead of 34, +1 % at 3k-10k, but 14k x 14k no longer fits its runs: -10 %; profiles/r06_ab.txt 8)
#endif
constexpr int RUN_A = CVO_RUN_A;             // acvo runs (kt_run_acvo): candidates per lane of each of the three records (xy, xx, yy), all in registers
constexpr int RUN_GEN = 4;           // generations of the exchange rows (see above)
constexpr int RUN_CHAINS = 8;        // chains an exchange's rows are added in (rows c, c + 8, ...; then a tree over the chains): fixed -- it is the order of the sums
struct RunMail {
    unsigned long long w[RUN_GEN][RUN_G + 1][2 * RUN_NV];   // (row RUN_G: the head block's verdict word)
    // two-level exchanges of the runs of more than RUN_G_SMALL solvers (cvo_kernels.hip run_exchange_hier): the RUN_CHAINS partial sums of an
    // exchange -- chain c = rows c, c + RUN_CHAINS, ... added by solver c, the same chains in the same order the one-level exchange adds --,
    // tagged and generation-numbered like the rows (a leader posts chain sums e + 4 after it has read every row of e + 4, which a block
    // posts after it has read the chain sums of e + 3; the head block, which only reads, is held by the verdict words as above)
    unsigned long long p[RUN_GEN][RUN_CHAINS][2 * RUN_NV];
    // entry handshake of a large run: every block of every kt_run launch draws a ticket as its first act (launches of one registration
    // follow each other in one stream, so launch L holds tickets [L NB, (L + 1) NB)); the head block waits until the g + 1 blocks
    // that take part have drawn theirs -- blocks start in index order: they are then resident -- and says GO, or ABORT when that
    // does not happen in time (another large run holds the compute units): nothing has been written, the run declines
    unsigned long long entry_ticket;
    unsigned long long entry_go;                      // (launch number + 1) << 32 | RUN_GO / RUN_ABORT
    // side builds (round 6, kt_run "side builds"): blocks of the record pass of side builds that have ended (kt_side_record; never reset),
    // and the requests the runs' head blocks have made (the host's side mirror carries the same number)
    unsigned long long side_done;
    unsigned long long side_req;
};
constexpr int RUN_G_SIDE = 128;      // a run of up to this many solvers has its next xy list built BESIDE it (the side kernels need compute units of their own)

struct KernConsts {
    float tau;        // d2 < tau
    float tau_c;      // d2c < tau_c
    float sp;         // keep iff a > sp
    float inv_c;      // 1/c   (float, ref `1/c*Ai`)
    float inv_d;      // 1/d
    float inv_l3;     // 1/(ell*ell*ell)
    float cb, cg, cd; // step-size scalings: -2t, -t, 2t with t = 1/(2 l^2)
    float cscale;     // > 0: MATLAB weight (linear colour inner product, threshold on K only)
    double s2_d;      // (double)(sigma*sigma)
    double cs2_d;     // (double)(c_sigma*c_sigma)
    double ninv_2l2;  // -1/(2 l^2)
    double ninv_2cl2; // -1/(2 c_l^2)
};

// Constants of a context (functions of cvo_hip_params only).
struct DevParams {
    int32_t mode, max_iter;
    float ell_init, ell_min, ell_max_init;
    float sp, c_sp;
    float c, d, c_ell;
    float min_step, eps, eps_2;
    float log_sp_s2;     // (float)log(sp/s2): tau = (float)(-2.0*l*l*log_sp_s2)
    float tau_c;         // (float)(-2.0*c_ell*c_ell*(float)log(c_sp/c_sigma/c_sigma))
    float list_margin;   // tile lists are built (1 + list_margin) wider than needed and re-used
                         // while they provably still hold every pair; 0 = rebuild every iteration
    int32_t async_xy;    // the xy list is double-buffered and built concurrently (plan_xy_async)
    float build_at;      // ... a new build is scheduled when this fraction of the margin in use is gone
    int32_t async_self;  // acvo: the xx / yy lists are double-buffered too and PROC_SELF rides in the flow launch
    float color_scale;   // cvo_hip_params::color_scale
    float run_cand_cap;  // > 0: the plan has resident runs (kt_run) that hold this many candidates in registers
    double s2_d, cs2_d, dl_step;
};

// The registration state, resident in HBM for the whole align(): a HEAD (what the O(1) maths of an
// iteration reads and writes; the post kernels take a private copy of it into registers; head mode
// keeps two copies in HBM) and a TAIL (counters that the list kernels touch with atomics).
struct alignas(16) DevHead {
    float R[9], T[3];
    float ell, ell_max;
    float Rt[9], t[3];          // inverse transform of the current iteration
    float used_Rt[9], used_t[3]; // the one the last EXECUTED iteration used
    KernConsts kc;              // kernel constants of the current iteration
    float kc_ell, kc_pad_;      // the ell they were made for (they depend on nothing else)
    // MFMA pre-filter (DESIGN.md "Conservative filter"): coordinates are taken
    // relative to `center`; a pair can only pass the exact test if its filter
    // value is below tauf[sel] = tau + rounding margin (sel = LIST_XY/XX/YY)
    float center[3];
    float xmax, y0max;          // max |x - center|, max |y0 - center| (bbox bounds)
    float tauf[3];
    int32_t n_fixed;            // points of the fixed cloud as the caller counts them (acvo Ayy rule);
    uint32_t pad0_;
    // Tile-list re-use (plan_lists): list l was built with every pair closer than
    // list_r[l]; the xy list with the moving cloud at [list_Rt | list_t].
    float list_r[3];
    int32_t list_ok[3];         // list l holds a build of this align()
    int32_t reuse[3];           // this iteration consumes list l as it is: k_filter returns at once
    int32_t ck_nblk[3];         // candidate list of tile list l (ProcessArgs::cand): recorded by a pass of this many
                                // blocks over the tile list as it stands; 0 = none (tile list rebuilt / to be rebuilt)
    float list_Rt[9], list_t[3];
    // Asynchronous xy builds (DevParams::async_xy): two buffers; FLOW consumes `xy_active`; the
    // k_filter blocks of the launch that READS this state build `xy_target` (-1: none) at the
    // transform recorded for it (xy_Rt / xy_t[target]); `xy_fresh`: the buffer whose build ran in
    // the previous flow launch and is judged by the next plan (head mode only, see plan_xy_async);
    // `stall`: no buffer is valid, the slot only builds.
    int32_t xy_active, xy_target, stall, xy_fresh;
    int32_t xy_ok[2];
    int32_t xy_ck[2];           // candidate record of xy buffer b (ProcessArgs::cand / cand_b): recorded by a flow pass of this
                                // many blocks over the buffer as it stands; 0 = none (head mode, where the list is double-buffered)
    float xy_r[2];
    float tauf_build;
    float r_last;               // sqrt(tau) of the plan before this one (0: none): scales the last iteration's member count to this one's radius
    float xy_Rt[2][9], xy_t[2][3];
    // the same for the two self lists of acvo (rigid: only the radius matters), [l][buffer]
    int32_t sf_active[2], sf_target[2], sf_fresh[2];
    int32_t sf_ok[2][2];
    int32_t sf_ck[2][2];        // candidate records of the self lists' buffers, as xy_ck
    float sf_r[2][2];
    float sf_tauf_build[2];
    cvo_math::XiConsts xi;      // twist constants for the step-size pass
    float omega[3], v[3];
    double dl;
    double red[RED_N];
    int32_t k;                  // iteration about to run / running
    int32_t done;               // RUNNING, DONE_*, NEED_BIGGER_LIST
    int32_t iter;               // the reference's `iter` member
    int32_t n_exec;             // loop bodies executed
    int32_t n_slots;            // slots completed (iterations + stall slots): the host paces its batches on it
    int32_t pending;            // head mode: a slot has been started whose post-step part has not run yet
    unsigned long long mail_snap;   // DevState::mail_seq as the last single-block exchange (or k_prepare) left it: what the blocks of a
                                // k_step_twist launch that exchanges number their exchange from (mail_seq itself moves while they start)
    int32_t run_hint;           // candidates expected in the record the slot that begins reads (prepare_iteration): what the host picks
                                // the next batch's plan by (PostStepArgs::hint_mirror)
    int32_t head_check_;        // the check word of a copy that went to the host's pinned mirror (head_check_mix); unused on the device
    int32_t rec_count[2];       // candidates in the record of xy buffer b where a launch has counted them (the publishing block of a head-mode
                                // flow launch, a resident run at entry), 0: not known -- plan_xy_async keeps a wide list whose record fits a run
};
// Check word of a head that went to the host's pinned copy (head_publish -> job_pump): every 16-byte piece mixed with its index,
// summed; DevHead::head_check_ counts as zero and then holds the sum.
CVO_HD unsigned head_check_mix(unsigned x, unsigned y, unsigned z, unsigned w, unsigned piece)
{
    return (x * 0x9E3779B1u + y * 0x85EBCA77u + z * 0xC2B2AE3Du + w * 0x27D4EB2Fu) ^ (piece * 0x165667B1u + 0x9E3779B9u);
}
struct DevState : DevHead {
    // ---- the TAIL: one copy per registration, at a fixed address (the head exists twice in
    // ---- head mode, see cvo_kernels.hip "Head mode"); everything below is only ever touched with
    // ---- atomics / plain stores of single words by the kernels
    // entries appended to every sub-list (the host polls only the part of the state in front of it)
    uint32_t sub[LIST_N][NSUB];
    // overflow flag of list l, raised by the launch that appended past the end (the host grows the
    // list and resumes): ovf[launch parity][l].  Classic launches use parity 0 throughout; in head
    // mode the flow launch of slot h raises ovf[h & 1], which the step launch of the slot and the
    // head of slot h + 1 read while the blocks of THAT launch already raise ovf[(h + 1) & 1].
    uint32_t ovf[2][8];
    // bit k of built[l]: iteration k (mod 2048) rebuilt list l (profiling: which
    // k_filter launches did the work)
    uint32_t built[3][64];
    // resident runs (kt_run) that have ended or declined since align() began: mirrored to the host, which sends the
    // next batch when the run of the batch in flight is over
    int32_t run_count;
    int32_t run_entered;    // ... runs that executed at least one iteration, and the iterations executed inside runs (cvo_hip_get_run_stats)
    int32_t run_iterations;
    int32_t run_candidates; // candidates of the record the last run looked at
    int32_t run_last_entered, run_last_aborted;   // run_count as the last run that carried slots / gave up at its entry left it: a batch may bring two
                            // runs, and the host looks at the mirror when both have reported (run_over)
    long long run_clk[16];  // CVO_RUN_CLOCKS builds: ticks of the first solver block by phase (tools/gpu_run_clocks.py)
    // exchanges done through the mailboxes since the context was created (never reset: the
    // sequence numbers of successive align() calls must keep alternating between the two
    // slot generations) -- kept last, align() re-initialises everything in front of it
    unsigned long long mail_seq;
    // ... and the exchanges of resident runs (RunMail), likewise
    unsigned long long run_seq;
};
static_assert(LIST_N <= 8, "DevState::ovf holds 8 lists");
constexpr size_t DEVSTATE_INIT_BYTES = offsetof(DevState, mail_seq);
constexpr size_t DEVSTATE_HEAD_BYTES = sizeof(DevHead);
static_assert(offsetof(DevState, sub) == sizeof(DevHead) && sizeof(DevHead) % 16 == 0, "the tail follows the head");

// Fused launches: one launch can serve up to MAXG independent registrations
// (blockIdx.z selects the argument block).  Every hot kernel here is latency-
// bound at 10k x 10k, so G registrations per launch cost little more than one.
constexpr int MAXG = 16;
template <class A> struct Grp { A a[MAXG]; };
constexpr int ENGINE_SLOTS = 32;   // slots of an engine's argument table (the by-value groups above stop at MAXG)

// Dense pair filter: rows [row_lo,row_hi) of cloud a against all of cloud b.
struct FilterArgs {
    const float4 *pos_a;
    const float4 *pos_b;
    const float4 *seg_a;   // bounding spheres (centre, radius) of every SEG device points
    const float4 *seg_b;   // ... of the ORIGINAL positions; centres are moved with [Rt|t]
    DevState *st;          // Rt, t, center, tauf, done; sub[list][] is appended to
    DevState *st2;         // head mode: the second copy of the state's head (else null)
    TileEntry *tiles;      // the tile list
    TileEntry *tiles_b;    // async: the second buffer (same capacity); st->xy_target / sf_target picks
    int async_xy;          // 1: the asynchronous xy build; 2, 3: the asynchronous xx / yy build
    uint32_t subcap;       // capacity (entries) of each of its NSUB sub-lists
    int list;              // LIST_XY / LIST_XX / LIST_YY: selects tauf[] and sub[]
    int row_lo, row_hi;
    int nb;
    int jt;                // columns per block chunk (multiple of SEG)
    int tf_a, tf_b;        // apply [Rt|t] to the row / column cloud while staging
    int check_done;        // return at once when st->done != 0
    int gx, gy;            // this registration's own grid (a fused launch may be larger)
    long long *dbg;        // probe only (tools/microbench): per-wave phase clocks, else null
    float4 *pos_bt;        // crowded engines: this launch also leaves [Rt|t] pos_b (all nb rows, .w kept) here for
                           // the list passes of the iteration, which then skip the transform per pair (else null)
};

// Exact evaluation of a tile list (PROC_FLOW, PROC_SELF) or of the kept list (PROC_STEP).
struct ProcessArgs {
    const float4 *pos_a;
    const float *feat_a;
    const float4 *pos_b;
    const float *feat_b;
    const TileEntry *tiles;
    const TileEntry *tiles_b;   // async: second buffer; st->xy_active (PROC_FLOW) / sf_active (PROC_SELF) picks
    int async_xy;               // also: every kernel of the slot returns at once when st->stall
    int async_self;             // PROC_SELF: 1 xx, 2 yy double-buffered
    uint2 *kept_ij;        // kept list: PROC_FLOW writes, PROC_STEP reads
    float *kept_a;
    uint32_t *kept_cnt;    // [PROC_WAVES] members recorded by each PROC_FLOW wave
    double *partials;      // [nacc][nblk] (k_step_twist: [nacc][nblk / 4])
    DevState *st;
    DevState *st2;         // head mode: the second copy of the state's head (else null)
    uint32_t subcap;       // of the tile list
    uint32_t kept_wcap;    // kept-list slice of one wave (entries)
    int nblk;              // blocks of this launch for this registration (NSUB .. PROC_BLOCKS)
    // k_step_twist only (PROC_STEP with the tail of compute_flow in front):
    const double *flow_part;   // [nblk][NACC_FLOW] partial sums of PROC_FLOW
    const double *xx_part;     // acvo: [nblk][NACC_SELF]
    const double *yy_part;
    cvo_hip_trace *trace; int trace_cap;
    int acvo;
    int32_t *done_mirror;
    int list;              // which tile list
    int row_hi, nb;        // valid rows / columns (mask bits beyond are padding)
    int first_counted;     // PROC_SELF: 1 = rows whose caller index is below st->n_fixed contribute 0 to the sum
    int tf_a, tf_b;
    int check_done;
    int weight;            // PROC_FLOW: 0 the C++ pair weight, 1 the MATLAB object's (classic launches only)
    uint2 *cand;           // PROC_FLOW over a synchronous xy list of clouds that pack (kept_packed): the CANDIDATE list --
                           // every pair of the tile list as (i | j << 16, its colour weight), wave by wave in the order
                           // the wave meets them.  The pass after a build expands the tile list and records it; the
                           // passes over the same tile list (st->ck_nblk[list] == nblk) stream the record instead: nothing to
                           // expand, no feature gathers, no colour exp.  PROC_SELF (acvo's xx / yy lists) likewise,
                           // the sign of the recorded weight = the row counts (Ayy rule).  Null: no candidate list.
                           // (8 bytes per candidate: clouds of up to 65536 rows -- 12-byte records for larger clouds were
                           // built and measured slower, profiles/r03_ab.txt 8)
    uint32_t *cand_cnt;    // [PROC_WAVES] candidates recorded by each wave
    uint2 *cand_b;         // head mode (double-buffered xy list): the record of the second buffer (8-byte form only) ...
    uint32_t *cand_cnt_b;  // ... DevHead::xy_ck[b] says whether buffer b's record matches its tile list
    int need_d2;           // PROC_FLOW: accumulate sum a (trace records, cvo_hip_flow) and sum (1/l^3 a) d2 (acvo's dl term; cvo_hip_flow reports it for both modes)
    const CommTable *comm; // k_step_twist with ranks: the flow-side sums are exchanged through the mailboxes inside the launch, every
                           // block reading its rank's own mailbox (null: one rank, or the exchange is a post kernel's / the host's)
    int kept_packed;       // 1: both clouds have <= 65536 rows: a kept entry is 8 bytes (i | j << 16, weight bits)
                           // in kept_ij alone instead of 8 + 4 -- the kept list is the largest HBM stream of a
                           // batched run (written by every flow pass, read back by the step pass).
                           // 2: clouds up to 262144 rows, 8 bytes as well: i and j in 18 bits each, and the weight --
                           // a float32 in (sp, sigma^2 c_sigma^2], positive, within 16 binades -- as 4 bits of
                           // exponent above kept_ebase and its 23 mantissa bits (kept_pack / kept_unpack: lossless).
                           // 0: 8 + 4 bytes (larger clouds, parameter sets whose weights span more, the MATLAB weight)
    unsigned kept_ebase;   // kept_packed == 2: the exponent field of the smallest weight there can be
};

// k_post flags
enum { POST_REDUCE = 1, POST_MATH = 2 };

struct PostFlowArgs {
    DevState *st;
    const double *part_flow;
    const double *part_xx;
    const double *part_yy;
    cvo_hip_trace *trace; int trace_cap;
    int flags;
    int check_done;
    int32_t *done_mirror;  // optional host-visible copy of st->done once the loop has stopped
    int nblk;              // rows of the partial-sum arrays (= ProcessArgs::nblk of the producers)
    const CommTable *comm; // not null: the reduced sums are exchanged with the other ranks (mailboxes)
    DevParams prm;
};

struct PostStepArgs {
    DevState *st;
    DevState *st2;         // head mode: the second copy of the state's head (else null)
    const double *part_step;
    cvo_hip_trace *trace; int trace_cap;
    int flags;
    int check_done;
    long long *dbg;        // diagnostics only (CVO_HIP_POST_DEBUG): phase clocks of thread 0
    int32_t *done_mirror;  // see PostFlowArgs
    int ck_nblk[3];        // the pass over tile list l of this iteration ran with ProcessArgs::cand and this many blocks:
                           // its candidate list now matches the tile list (0: no candidate list)
    // THE HOST'S MIRRORS (pinned host memory written by the kernels): done_mirror, progress_mirror, run_mirror, hint_mirror, side_mirror are
    // SINGLE 4-byte WORDS and must stay so -- stores to host memory were seen to pass each other on this platform although the device fences
    // at system scope in between (profiles/r05_ab.txt 15c), so nothing may be read as "written before the word that says so" except through a
    // check of its own: the one multi-word mirror, final_mirror, carries a check word (head_check_mix) that the host verifies and re-reads
    // on.  A plain store to host memory may also stay in the device's cache until the kernel ENDS (round 6: the first side-build request was
    // never seen while its run was alive): a word the host must see while the kernel runs is stored at system scope (__hip_atomic_store ...
    // __HIP_MEMORY_SCOPE_SYSTEM); progress_mirror and hint_mirror inside a run are plain -- the host reads them when the run has reported.
    int32_t *progress_mirror;   // optional host-visible copy of st->n_slots (pinned): lets the host enqueue the next
                                // batch when the running one is down to its last slot instead of a whole batch ahead
    int nblk;
    const CommTable *comm; // see PostFlowArgs
    DevHead *final_mirror; // pinned (null: none): the head of a loop that has STOPPED with a verdict goes there too, in front of the `done`
                           // mirror: cvo_hip_align then returns without a copy behind the launches that are still queued
    uint32_t *build_mask;  // the table's build masks (kt_filter; null: the plan has no filter launch of its own) and this slot's bit
    uint32_t slot_bit;
    // resident runs (kt_run; null / 0: the plan has none)
    RunMail *run_mail;
    int32_t *run_mirror;   // pinned: DevState::run_count
    int32_t *hint_mirror;  // pinned: DevHead::run_hint (the host picks the next batch's plan by it)
    int run_iters;         // iterations per run at most
    int run_g_max;         // solver blocks of a run at most (RUN_G on an unpartitioned MI355X; fewer compute units: fewer, cvo_hip_create)
    long long run_timeout_ticks;   // how long a poll of a run's exchange waits (100 MHz wall clock; cvo_hip_set_option "run_timeout_ms")
    int32_t *side_mirror;  // pinned (null: no side builds): the head block of a run that wants its next xy list built beside it writes the request's
                           // number here; the host launches kt_side_filter + kt_side_record on the context's side stream (cvo_job.cpp job_pump)
    float run_build_at;    // ... and with side builds a run names its builds when this fraction of the list's room is gone (DevParams::build_at:
                           // the launch-per-pass path, whose builds take one slot)
    int run_fault;         // test switch (cvo_hip_set_option "run_fault" = n > 0): the first solver of every run leaves at the top of its n-th
                           // iteration without a word -- what a block lost to the scheduler looks like to its peers
    DevParams prm;
};

// ---------------------------------------------------------------------------
// Argument tables.  The kernels of the align() loop do not take their argument blocks by
// value: they take a pointer to an array of Slots in device memory and blockIdx.z selects the
// slot; op[q] holds the arguments of the q-th launch of an iteration.  What this buys:
//   * a captured batch of iterations (hipGraph) depends on the launch geometry only, not on
//     any buffer address: it is captured once per table and shape and then serves every
//     frame pair and every membership of a fused group -- a registration enters or leaves a
//     running group by a stream-ordered copy into the table (continuous batching), no
//     drained queue, no new capture;
//   * the merged launches (flow pass + self passes + list builds in one grid) read only the
//     fields of the role a block plays, when it needs them: no scalar-register spills.
// An empty slot (active == 0) costs one scalar load per block.
constexpr size_t kTableHeaderBytes = 256;   // in front of a table's slots: uint32_t build_mask[3] (+ padding), see kt_filter
constexpr int MAX_OPS = 10;   // launches of one iteration (9 for acvo in a fused group: 3 filters, flow, 2 selfs, post, step, post)
struct OpArgs {
    ProcessArgs p;            // PROCESS; the flow pass of a merged launch
    FilterArgs f;             // FILTER; the xy build riding in a flow launch
    union {
        PostFlowArgs pf;
        PostStepArgs ps;
    };
    int np, n0, n1, n2;       // merged launches: blocks of the roles
};
// Merged launches of a registration on its own take the argument blocks of their further roles
// from the NEXT entries: op[q + 1], op[q + 2] hold the xx / yy filters (kt_filter_group,
// kt_flow_build3, kt_flow_build6: .f) and the two self passes (kt_flow_build6: .p; kt_self2:
// op[q], op[q + 1]).
struct Slot {
    int active;
    int pad_[3];
    OpArgs op[MAX_OPS];
};

CVO_HD KernConsts make_kconsts(const DevParams &p, float ell)
{
    KernConsts k;
    const float l = ell;
    k.tau = (float)(-2.0 * l * l * (double)p.log_sp_s2);
    k.tau_c = p.tau_c;
    k.sp = p.sp;
    k.inv_c = 1 / p.c;
    k.inv_d = 1 / p.d;
    const float ell_3 = l * l * l;
    k.inv_l3 = 1 / ell_3;
    const float temp_coef = (float)(1 / (2.0 * l * l));
    k.cb = (float)(-2.0 * temp_coef);
    k.cg = -temp_coef;
    k.cd = (float)(2.0 * temp_coef);
    k.cscale = p.color_scale;
    // MATLAB keeps K >= sp (ref rkhs_se3_registration.m:70): the radius is widened by 1e-5 so
    // that the exact test on K in pair_weight decides, not the rounding of tau
    if (k.cscale > 0.0f) k.tau = (float)((double)k.tau * 1.00001);
    k.s2_d = p.s2_d;
    k.cs2_d = p.cs2_d;
    k.ninv_2l2 = -1.0 / (2.0 * l * l);
    k.ninv_2cl2 = -1.0 / (2.0 * p.c_ell * p.c_ell);
    return k;
}

// Conservative thresholds of the MFMA pre-filter.  The filter evaluates
// q = (|x'|^2 - tauf) + sum_k x'_k (-2 y'_k) + |y'|^2 in float32 with
// x' = x - center, y' = y - center.  All its rounding errors together are below
// 8 u S (u = 2^-24, S = (max|x'| + max|y'|)^2); tauf = tau + 16 u S
// keeps a factor two in hand, so every pair with d2 < tau has q < 0.
// `identity` = the moving cloud is used untransformed (function_inner_product).
CVO_HD void compute_filter_bounds(DevHead *s, bool identity)
{
    // (float32 throughout, as all of the plan below: these are bounds with 1e-6 ... 1e-4 of slack, worked out on
    // the ONE chain of the post-step part, where a float64 operation waits twice as long for its operands as a
    // float32 one and a float64 square root is ~25 dependent instructions instead of one; ~20 float32 operations
    // in a row are good to 2e-6 relative)
    float ymax = s->y0max;
    if (!identity) {
        // |R^T (p - T) - c| = |(p - c) - (T + R c - c)| <= |p - c| + |T + (R - I) c|
        const float *R = s->R, *c = s->center;
        float sh2 = 0.0f;
        for (int r = 0; r < 3; ++r) {
            const float rc = R[3 * r] * c[0] + R[3 * r + 1] * c[1] + R[3 * r + 2] * c[2];
            const float d = s->T[r] + rc - c[r];
            sh2 += d * d;
        }
        ymax = (s->y0max + sqrtf(sh2)) * 1.0001f + 1e-6f;
    }
    const float u16 = 16.0f / 16777216.0f;
    const float sxy = (s->xmax + ymax) * (s->xmax + ymax);
    const float sxx = 4.0f * s->xmax * s->xmax;
    const float syy = 4.0f * ymax * ymax;
    const float tau = s->kc.tau;
    s->tauf[LIST_XY] = (tau + u16 * sxy) * 1.000001f;
    s->tauf[LIST_XX] = (tau + u16 * sxx) * 1.000001f;
    s->tauf[LIST_YY] = (tau + u16 * syy) * 1.000001f;
}

// Tile-list re-use.  k_filter is conservative and membership in A is decided by
// the exact test of k_process, so a list stays valid for as long as it is a
// superset of {pairs with d2 < tau}.  A list is therefore built for the radius
// list_r = (1 + margin) * sqrt(tau) and kept while
//     sqrt(tau_now) + (how far any moving point has travelled since the build) <= list_r
// (triangle inequality; the xx / yy lists of acvo are rigid: only tau matters).
// The travel of y = Rt y0 + t is bounded by |dRt|_2 max|y0 - c| + |dRt c + dt| with
// |dRt|_2 = |dRt|_F / sqrt(2) for a difference of rotations.  A list much wider
// than needed (after ell dropped) is rebuilt as well.  All slack terms are far
// above the float32 rounding of the coordinates and of d2 (<= ~1e-5 m here) and
// far below the margin (>= 1 mm at ell_min): decisions change performance only.
constexpr float LIST_LOOSE = 1.3f;
// (`bulk`: where the transform records and the kernel constants are stored -- the state itself; the post
// kernels run the plan on a private copy in registers, every lane of a wave the same, and send these few
// large, rarely written fields straight to the shared copy instead of carrying them along: `store` =
// this lane does)
CVO_HD void plan_lists(DevHead *s, DevHead *bulk, const bool store, const DevParams &p, const float r_now)
{
    const float ymax = s->y0max;
    const float slack = 1.0e-4f * (1.0f + s->xmax + ymax);
    const float margin = p.list_margin;
    float travel = 0.0f;
    if (s->list_ok[LIST_XY] && !p.async_xy) {
        float f2 = 0.0f, c2 = 0.0f;
        for (int r = 0; r < 3; ++r) {
            float dc = s->t[r] - s->list_t[r];
            for (int q = 0; q < 3; ++q) {
                const float d = s->Rt[3 * r + q] - s->list_Rt[3 * r + q];
                f2 += d * d;
                dc += d * s->center[q];
            }
            c2 += dc * dc;
        }
        travel = sqrtf(0.5f * f2) * 1.001f * ymax + sqrtf(c2);
    }
#pragma unroll
    for (int l = 0; l < 3; ++l) {
        if (l == LIST_XY && p.async_xy) continue;   // planned by plan_xy_async (and tauf[XY] must stay
                                                    // tau + rounding slack: tauf_build is made from it)
        if (l != LIST_XY && p.async_self) continue; // planned by plan_self_async
        if (l != LIST_XY && p.mode != CVO_HIP_MODE_ACVO) { s->reuse[l] = 1; continue; }   // cvo has no self lists
        const float need = (r_now + (l == LIST_XY ? travel : 0.0f)) * 1.0001f + slack;
        const float lr = s->list_r[l];
        const bool keep = margin > 0.0f && s->list_ok[l] && need <= lr &&
                          lr <= LIST_LOOSE * (1.0f + margin) * (r_now * 1.0001f + slack);
        s->reuse[l] = keep ? 1 : 0;
        if (keep) continue;
        s->ck_nblk[l] = 0;   // a new tile list: the candidate list recorded from the old one is void
        const float rb = (r_now * 1.0001f + slack) * (1.0f + margin);
        s->list_r[l] = rb * 1.000001f;   // rounded up: the list holds at least this radius
        s->list_ok[l] = 1;
        if (margin > 0.0f)   // tauf of compute_filter_bounds is tau + rounding slack: widen tau
            s->tauf[l] = (s->list_r[l] * s->list_r[l] + (s->tauf[l] - s->kc.tau)) * 1.000001f;
        if (l == LIST_XY && store) {
            for (int q = 0; q < 9; ++q) bulk->list_Rt[q] = s->Rt[q];
            for (int q = 0; q < 3; ++q) bulk->list_t[q] = s->t[q];
        }
    }
}

// Asynchronous xy builds.  A build needs A transform near the current one, not
// the next one.  The plan step (end of slot s in the classic launches, head of the flow
// launch of slot s + 1 in head mode) names a buffer to build; the k_filter blocks riding in
// a flow launch build the target named by the state that launch STARTS from, at the
// transform the plan recorded for it (xy_Rt / xy_t[target]); a later plan step judges the
// finished build and switches to it if it holds every pair for the transform of that moment.
// The filter is off the launch chain and never waited for.  A build is scheduled when most
// of the margin of the list in use is gone or the list is too wide; if no buffer is valid
// (first slot, a jump) the coming slot is a STALL: nothing but the build runs.
//   classic launches: the target named at the end of slot s is built by the flow launch of
//     slot s + 1 and judged at the end of slot s + 1   (fresh = the state's xy_target, inflight = -1)
//   head mode: the plan runs redundantly in the head of every block of the flow launch of slot
//     s + 1 while the launch's filter blocks are already at work on the target named one slot
//     earlier (inflight = the state's xy_target, not to be judged yet: it becomes xy_fresh);
//     the target named now is built by the flow launch of slot s + 2 and judged at the head of
//     slot s + 3.  One build at a time: the buffers are two.
// No dynamic indexing below: in the post kernels the state lives in registers.
// (rec: where the transform records are read from -- the shared copy of the head, see `bulk` at plan_lists)
template <int B> CVO_HD float xy_travel(const DevHead *s, const DevHead *rec)
{
    float f2 = 0.0f, c2 = 0.0f;
    for (int r = 0; r < 3; ++r) {
        float dc = s->t[r] - rec->xy_t[B][r];
        for (int q = 0; q < 3; ++q) {
            const float d = s->Rt[3 * r + q] - rec->xy_Rt[B][3 * r + q];
            f2 += d * d;
            dc += d * s->center[q];
        }
        c2 += dc * dc;
    }
    return sqrtf(0.5f * f2) * 1.001f * s->y0max + sqrtf(c2);
}

CVO_HD void plan_xy_async(DevHead *s, DevHead *bulk, const bool store, const DevParams &p, const float r_now, const int fresh,
                          const bool fresh_failed, const int inflight)
{
    const float slack = 1.0e-4f * (1.0f + s->xmax + s->y0max);
    const float margin = p.list_margin;
    const float r0 = r_now * 1.0001f + slack;   // radius needed with no travel
    if (fresh == 0) s->xy_ok[0] = fresh_failed ? 0 : 1;   // the build that has just ended
    if (fresh == 1) s->xy_ok[1] = fresh_failed ? 0 : 1;
    // (how far the cloud has travelled since a buffer was built: worked out for the buffer that is looked
    // at first, the other one only if that fails)
    const int act = s->xy_active ? 1 : 0;
    const int first = fresh >= 0 ? (fresh ? 1 : 0) : act;
    float need0 = 0.0f, need1 = 0.0f;
    bool valid0 = false, valid1 = false;
    if (first == 0) { need0 = s->xy_ok[0] ? (r_now + xy_travel<0>(s, bulk)) * 1.0001f + slack : 0.0f; valid0 = s->xy_ok[0] && need0 <= s->xy_r[0]; }
    else { need1 = s->xy_ok[1] ? (r_now + xy_travel<1>(s, bulk)) * 1.0001f + slack : 0.0f; valid1 = s->xy_ok[1] && need1 <= s->xy_r[1]; }
    int use = -1;
    if (first ? valid1 : valid0) use = first;
    else {
        if (first == 0) { need1 = s->xy_ok[1] ? (r_now + xy_travel<1>(s, bulk)) * 1.0001f + slack : 0.0f; valid1 = s->xy_ok[1] && need1 <= s->xy_r[1]; }
        else { need0 = s->xy_ok[0] ? (r_now + xy_travel<0>(s, bulk)) * 1.0001f + slack : 0.0f; valid0 = s->xy_ok[0] && need0 <= s->xy_r[0]; }
        if (first ? valid0 : valid1) use = 1 - first;
    }
    s->stall = use < 0 ? 1 : 0;
    if (use >= 0) s->xy_active = use;
    bool build = use < 0 || !(margin > 0.0f);
    if (!build) {
        const float lr = use ? s->xy_r[1] : s->xy_r[0];
        const float nd = use ? need1 : need0;
        if (nd - r0 > p.build_at * (lr - r0)) build = true;   // most of the margin is gone
        // far wider than needed (the length scale has dropped).  Where the plan has resident runs and the record of the list in use
        // still fits one, "far" is farther: inside a run a candidate costs ~0.03 us per iteration, a rebuild ends the run and costs three
        // classic slots and a new entry (~60 us) -- a list of up to 1.9 x the radius (3.6 x the candidates: one step of the cvo schedule,
        // 0.15 -> 0.10 -> 0.06) serves on, the step 0.06 -> 0.03 (2 x) rebuilds
        const float nnz = (float)s->red[RED_FLOW + 8];
        const float w = lr / r0;
        // (the record's count where a launch has counted it, else an estimate with room)
        const int known = use ? s->rec_count[1] : s->rec_count[0];
        const bool fits = p.run_cand_cap > 0.0f && (known > 0 ? (float)known <= p.run_cand_cap :
                          (nnz > 0.0f && 1.05f * nnz * (r_now / fmaxf(s->r_last, 1.0e-9f)) * (r_now / fmaxf(s->r_last, 1.0e-9f)) * w * w <= 0.9f * p.run_cand_cap));
        if (lr > (fits ? 1.9f : LIST_LOOSE) * (1.0f + margin) * r0) build = true;
    }
    if (inflight >= 0) build = false;   // (its buffer is the only one that is free)
    s->xy_fresh = inflight;
    s->xy_target = -1;
    if (build) {
        const int tgt = use < 0 ? 0 : 1 - use;
        s->xy_target = tgt;
        const float r = r0 * (1.0f + margin) * 1.000001f;   // rounded up
        // tauf[LIST_XY] of compute_filter_bounds is tau + rounding slack: widen tau
        s->tauf_build = (r * r + (s->tauf[LIST_XY] - s->kc.tau)) * 1.000001f;
        if (tgt == 0) { s->xy_ok[0] = 0; s->xy_ck[0] = 0; s->xy_r[0] = r; s->rec_count[0] = 0; }   // (a new tile list: its candidate record is void)
        else { s->xy_ok[1] = 0; s->xy_ck[1] = 0; s->xy_r[1] = r; s->rec_count[1] = 0; }
        if (store) {
            float *dR = tgt == 0 ? bulk->xy_Rt[0] : bulk->xy_Rt[1], *dt = tgt == 0 ? bulk->xy_t[0] : bulk->xy_t[1];
            for (int q = 0; q < 9; ++q) dR[q] = s->Rt[q];
            for (int q = 0; q < 3; ++q) dt[q] = s->t[q];
        }
    }
}

// The acvo self lists the same way (PROC_SELF then rides in the flow launch and needs
// its lists built BEFORE that launch).  They are rigid -- the pair distances do not
// depend on the transform -- so only ell ages them: a list built for radius
// (1 + margin) r serves until r_now outgrows it; the next one is built ahead when
// most of that room is gone or ell has dropped far below.
template <int L> CVO_HD void plan_self_async_one(DevHead *s, const DevParams &p, const float r0, const float margin,
                                                 const int fresh, const bool fresh_failed, const int inflight)
{
    if (fresh == 0) s->sf_ok[L][0] = fresh_failed ? 0 : 1;
    if (fresh == 1) s->sf_ok[L][1] = fresh_failed ? 0 : 1;
    const bool valid0 = s->sf_ok[L][0] && r0 <= s->sf_r[L][0];
    const bool valid1 = s->sf_ok[L][1] && r0 <= s->sf_r[L][1];
    const int act = s->sf_active[L] ? 1 : 0;
    int use = -1;
    if (fresh >= 0 && (fresh ? valid1 : valid0)) use = fresh ? 1 : 0;
    else if (act ? valid1 : valid0) use = act;
    else if (act ? valid0 : valid1) use = 1 - act;
    if (use < 0) s->stall = 1;
    else s->sf_active[L] = use;
    bool build = use < 0 || !(margin > 0.0f);
    if (!build) {
        const float lr = use ? s->sf_r[L][1] : s->sf_r[L][0];
        const float built_for = lr / (1.0f + margin);              // the radius it was built around
        if (r0 > built_for + p.build_at * (lr - built_for)) build = true;
        if (lr > LIST_LOOSE * (1.0f + margin) * r0) build = true;
    }
    if (inflight >= 0) build = false;
    s->sf_fresh[L] = inflight;
    s->sf_target[L] = -1;
    if (build) {
        const int tgt = use < 0 ? 0 : 1 - use;
        s->sf_target[L] = tgt;
        const float r = r0 * (1.0f + margin) * 1.000001f;
        s->sf_tauf_build[L] = (r * r + (s->tauf[LIST_XX + L] - s->kc.tau)) * 1.000001f;
        if (tgt == 0) { s->sf_ok[L][0] = 0; s->sf_ck[L][0] = 0; s->sf_r[L][0] = r; }
        else { s->sf_ok[L][1] = 0; s->sf_ck[L][1] = 0; s->sf_r[L][1] = r; }
    }
}

// What a plan step knows about the asynchronous builds besides the state (see plan_xy_async)
struct PlanBuilds {
    int xy_fresh, xy_inflight;
    int sf_fresh[2], sf_inflight[2];
    bool xy_failed, sf_failed[2];
};
// the classic launches: the target the state names has been built by the launches of the slot that ends
CVO_HD PlanBuilds plan_builds_classic(const DevHead *s, bool xy_failed, bool xx_failed, bool yy_failed)
{
    PlanBuilds b;
    b.xy_fresh = s->xy_target; b.xy_inflight = -1; b.xy_failed = xy_failed;
    b.sf_fresh[0] = s->sf_target[0]; b.sf_fresh[1] = s->sf_target[1];
    b.sf_inflight[0] = b.sf_inflight[1] = -1;
    b.sf_failed[0] = xx_failed; b.sf_failed[1] = yy_failed;
    return b;
}
// head mode: the target the state names is being built by this very launch
CVO_HD PlanBuilds plan_builds_head(const DevHead *s, bool xy_failed, bool xx_failed, bool yy_failed)
{
    PlanBuilds b;
    b.xy_fresh = s->xy_fresh; b.xy_inflight = s->xy_target; b.xy_failed = xy_failed;
    b.sf_fresh[0] = s->sf_fresh[0]; b.sf_fresh[1] = s->sf_fresh[1];
    b.sf_inflight[0] = s->sf_target[0]; b.sf_inflight[1] = s->sf_target[1];
    b.sf_failed[0] = xx_failed; b.sf_failed[1] = yy_failed;
    return b;
}
// first plan of an align() (k_prepare): nothing built, nothing in flight
CVO_HD PlanBuilds plan_builds_none()
{
    PlanBuilds b;
    b.xy_fresh = b.xy_inflight = -1; b.xy_failed = false;
    b.sf_fresh[0] = b.sf_fresh[1] = b.sf_inflight[0] = b.sf_inflight[1] = -1;
    b.sf_failed[0] = b.sf_failed[1] = false;
    return b;
}

// Everything an iteration needs that derives from (R, T, ell).  The caller logs
// the lists that are rebuilt (reuse[l] == 0) in DevState::built and zeroes the sub-list
// counters (and, classic launches, the overflow flags) of what will be built.
CVO_HD void prepare_iteration(DevHead *s, DevHead *bulk, const bool store, const DevParams &p, const PlanBuilds &b)
{
    cvo_math::inverse_tf(s->R, s->T, s->Rt, s->t);
    if (!(s->kc_ell == s->ell)) {   // three float64 divisions and a log-scaled threshold: only when ell moved
        const KernConsts k = make_kconsts(p, s->ell);
        if (store) bulk->kc = k;
        s->kc.tau = k.tau;   // (all the plan reads of it)
        s->kc_ell = s->ell;
    }
    compute_filter_bounds(s, false);
    const float r_now = sqrtf(s->kc.tau);
    plan_lists(s, bulk, store, p, r_now);
    if (p.async_xy) plan_xy_async(s, bulk, store, p, r_now, b.xy_fresh, b.xy_failed, b.xy_inflight);
    {   // Candidates of the record the slot that begins will read -- what a resident run (kt_run) must hold in registers; the host
        // picks the next batch's plan by it.  An estimate: the clouds are surfaces, so pairs within a radius go with its square
        // (members of A per iteration at ell = 0.15 / 0.10 / 0.06 / 0.03: ratios 2.15, 2.73, 4.0), and a record of radius R holds
        // ~1.05 (R / r)^2 candidates per member at radius r (1.64 at R = 1.25 r; profiles/r05_ab.txt 0, 4).
        const float nnz = (float)s->red[RED_FLOW + 8];
        const float q = s->r_last > 0.0f ? r_now / s->r_last : 1.0f;
        // (a build named or in flight: the record that list will give -- the host queues a RUN batch two slots ahead)
        const int coming = p.async_xy ? (s->xy_target >= 0 ? s->xy_target : s->xy_fresh) : -1;
        const float rr = p.async_xy ? ((coming >= 0 ? coming : s->xy_active) ? s->xy_r[1] : s->xy_r[0]) : s->list_r[LIST_XY];
        const float w = rr / r_now;
        // (x 1 / 0.95: the host compares with what a run holds, and an estimate wants room -- the publishing block of a head-mode
        // flow launch replaces it by the record's count where that is known, head_body)
        s->run_hint = (nnz > 0.0f && nnz < 1.0e9f) ? (int32_t)fminf(1.105f * nnz * (q * q) * (w * w), 2.0e9f) : 0;   // (NaN: an overflowed iteration)
        s->r_last = r_now;
    }
    if (p.async_self) {   // (after the xy plan: it may add a stall)
        const float slack = 1.0e-4f * (1.0f + s->xmax + s->y0max);
        const float r0 = r_now * 1.0001f + slack;
        plan_self_async_one<0>(s, p, r0, p.list_margin, b.sf_fresh[0], b.sf_failed[0], b.sf_inflight[0]);
        plan_self_async_one<1>(s, p, r0, p.list_margin, b.sf_fresh[1], b.sf_failed[1], b.sf_inflight[1]);
    }
}

size_t filter_smem_bytes(int jt);
// What the state of a registration that begins holds besides zeros (cvo_job.cpp job_begin): handed to the prepare kernel by value, which
// zeroes the state and sets these -- one stream operation instead of a host-to-device copy and a kernel.
struct PrepareInit {
    int32_t on;          // 0: the state is in place already (a resumption after a list grew)
    float R[9], T[3], ell, ell_max;
    int32_t iter, n_fixed, done;
    float center[3], xmax, y0max;
};
void launch_prepare(DevState *st, const DevParams &prm, hipStream_t s, uint32_t *build_masks = nullptr, const PrepareInit *init = nullptr);
void launch_filter(const FilterArgs &a, dim3 grid, hipStream_t s, hipEvent_t ev_start = nullptr,
                   hipEvent_t ev_stop = nullptr);
void launch_process(int mode, const ProcessArgs &a, hipStream_t s, hipEvent_t ev_start = nullptr,
                    hipEvent_t ev_stop = nullptr);
void launch_post_flow(const PostFlowArgs &a, hipStream_t s);
void launch_post_step(const PostStepArgs &a, hipStream_t s);
// fused: n <= MAXG argument blocks, one launch (FilterArgs::gx/gy must be set)
void launch_filter_group(const FilterArgs *a, int n, hipStream_t s);
void launch_process_group(int mode, const ProcessArgs *a, int n, hipStream_t s, hipEvent_t ev_start = nullptr,
                          hipEvent_t ev_stop = nullptr);
void launch_post_flow_group(const PostFlowArgs *a, int n, hipStream_t s);
void launch_post_step_group(const PostStepArgs *a, int n, hipStream_t s);
void launch_step_twist_group(const ProcessArgs *a, int n, hipStream_t s, hipEvent_t ev_start = nullptr,
                             hipEvent_t ev_stop = nullptr);
constexpr int STEP_TWIST_ROWS_DIV = 4;   // k_step_twist writes nblk / 4 partial rows

// One launch of an iteration through a table: which kernel, its geometry, which op[] it reads.
enum TKernel { TK_FILTER = 0, TK_FILTER_GROUP, TK_FLOW, TK_FLOW_MATLAB, TK_STEP, TK_SELF, TK_SELF2, TK_STEP_TWIST,
               TK_FLOW_BUILD, TK_FLOW_BUILD3, TK_FLOW_BUILD6, TK_POST_FLOW, TK_POST_STEP,
               TK_HFLOW_BUILD, TK_HFLOW_BUILD6, TK_HSTEP_TWIST, TK_RUN /* q = flow op | step op << 4 */, TK_RUN_ACVO /* likewise; the self passes: flow op + 1, + 2 */,
               TK_SIDE_FILTER, TK_SIDE_RECORD /* q = the flow op: the side build of a run's next xy list */,
               TK_FLOW_D2 /* TK_FLOW is built without the sums of a and of a d2 (ProcessArgs::need_d2 == 0 in every slot); this one has them */ };   // head mode (cvo_kernels.hip "Head mode")
// Head-mode launches carry the slot's parity and the mode in the bits above the op index of their
// second kernel argument: qp = q | parity << 8 | QP_HEAD.
constexpr int QP_PARITY = 1 << 8, QP_HEAD = 1 << 9, QP_MASK = 0xff;
struct TLaunch {
    int kernel;          // TKernel
    int q;               // op index in the slots
    unsigned gx, gz;     // grid.x, grid.z (= slots served)
    unsigned smem;       // dynamic LDS bytes
    int list;            // TK_FILTER: which tile list it builds (its word of the table's build masks)
};
// (parity: of the slot within its batch, head-mode kernels only)
void launch_table(const Slot *tab, const TLaunch &l, hipStream_t s, hipEvent_t ev_start = nullptr,
                  hipEvent_t ev_stop = nullptr, int parity = 0);
// geometry helpers shared with the host (what the by-value launchers compute from their arguments)
unsigned filter_grid_cap(long long nitems, long long cap);
long long filter_blocks_cap();
hipError_t run_allow_lds();   // (before the first kt_run launch on the current device: its dynamic LDS is above the default limit)

}   // namespace cvo_dev
