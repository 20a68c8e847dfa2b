// cvo_device.h -- device-resident state and argument blocks shared by the HIP
// kernels (cvo_kernels.hip) and their host driver (cvo_capi.cpp).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cvo_hip.h"
#include "se3_math.hpp"

namespace cvo_dev {

// Device cloud layout (DESIGN.md "Data layout in HBM"):
//   pos  : float4 per point (x, y, z, 0)        16 B, one global_load_dwordx4
//   feat : 8 floats per point (f0..f4, 0, 0, 0) 32 B, two dwordx4 loads
// Algorithmic bytes per point are 12 + 20 = 32 B; the padding is
// free here: a sweep reads each cloud once and reuses it ~N-fold on chip.
constexpr int FEAT_STRIDE = 8;

// Sweep geometry: a block of 256 threads owns ROWS_PER_LANE x 256 target rows
// and one chunk of `jt` source columns.
constexpr int BLOCK = 256;
constexpr int ROWS_PER_LANE = 4;
constexpr int ROWS_PER_TILE = BLOCK * ROWS_PER_LANE;
constexpr int QCAP = 128;           // per-wave candidate queue entries

enum SweepMode { SWEEP_FLOW = 0, SWEEP_STEP = 1, SWEEP_SELF = 2 };

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

struct KernConsts {
    float tau;        // d2 < tau
    float tau_c;      // d2c < tau_c
    float sp;         // keep iff a > sp
    float inv_c;      // 1/c   (float, ref `1/c*Ai`)
    float inv_d;      // 1/d
    float inv_l3;     // 1/(ell*ell*ell)
    float cb, cg, cd; // step-size scalings: -2t, -t, 2t with t = 1/(2 l^2)
    float pad_;
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
    float pad_;
    double s2_d, cs2_d, dl_step;
};

// The registration state, resident in HBM for the whole align().
struct DevState {
    float R[9], T[3];
    float ell, ell_max;
    float Rt[9], t[3];          // inverse transform of the current iteration
    float used_Rt[9], used_t[3]; // the one the last EXECUTED iteration used
    KernConsts kc;              // kernel constants of the current iteration
    cvo_math::XiConsts xi;      // twist constants for the step-size sweep
    float omega[3], v[3];
    double dl;
    double red[RED_N];
    int32_t k;                  // iteration about to run / running
    int32_t done;               // 0 running, 1 break A, 2 break B, 3 MAX_ITER exhausted
    int32_t iter;               // the reference's `iter` member
    int32_t n_exec;             // loop bodies executed
};

struct SweepArgs {
    const float4 *pos_a;   // rows (targets)
    const float *feat_a;
    const float4 *pos_b;   // columns (sources; transformed on the fly if tf_b)
    const float *feat_b;
    double *partials;      // [gridDim.y * gridDim.x][nacc]
    const DevState *st;    // Rt, t, kc, xi, done
    int row_lo, row_hi;    // rows processed
    int nb;                // columns
    int jt;                // columns per chunk
    int first_counted;     // SWEEP_SELF: rows below contribute 0 to the sum
    int tf_a, tf_b;        // apply [Rt|t] to the row / column cloud while staging
    int check_done;        // return at once when st->done != 0
};

// k_post flags
enum { POST_REDUCE = 1, POST_MATH = 2 };

struct PostFlowArgs {
    DevState *st;
    const double *part_flow; int nb_flow;
    const double *part_xx;   int nb_xx;
    const double *part_yy;   int nb_yy;
    cvo_hip_trace *trace; int trace_cap;
    int flags;
    int check_done;
    DevParams prm;
};

struct PostStepArgs {
    DevState *st;
    const double *part_step; int nb_step;
    cvo_hip_trace *trace; int trace_cap;
    int flags;
    int check_done;
    DevParams prm;
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
    k.pad_ = 0.0f;
    k.s2_d = p.s2_d;
    k.cs2_d = p.cs2_d;
    k.ninv_2l2 = -1.0 / (2.0 * l * l);
    k.ninv_2cl2 = -1.0 / (2.0 * p.c_ell * p.c_ell);
    return k;
}

// Everything an iteration needs that derives from (R, T, ell).
CVO_HD void prepare_iteration(DevState *s, const DevParams &p)
{
    cvo_math::inverse_tf(s->R, s->T, s->Rt, s->t);
    s->kc = make_kconsts(p, s->ell);
}

void launch_prepare(DevState *st, const DevParams &prm, hipStream_t s);
void launch_sweep(int mode, const SweepArgs &a, dim3 grid, hipStream_t s);
void launch_post_flow(const PostFlowArgs &a, hipStream_t s);
void launch_post_step(const PostStepArgs &a, hipStream_t s);

}   // namespace cvo_dev
