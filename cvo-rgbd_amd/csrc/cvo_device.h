// cvo_device.h -- argument blocks shared by the HIP kernels and their host
// launchers.  Plain PODs passed by value in the kernarg segment.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cvo_dev {

// Device cloud layout (DESIGN.md "Data layout in HBM"):
//   pos  : float4 per point (x, y, z, 0)        16 B, one global_load_dwordx4
//   feat : 8 floats per point (f0..f4, 0, 0, 0) 32 B, two dwordx4 loads
// Algorithmic bytes per point are 12 + 20 = 32 B; the padding is
// free here: a sweep reads each cloud once and reuses it ~N-fold on chip.
constexpr int FEAT_STRIDE = 8;
constexpr int TAYLOR_STRIDE = 16;   // floats per source point, see k_taylor

// Sweep geometry: a block of 256 threads owns ROWS_PER_LANE x 256 target rows
// and one chunk of `jt` source columns.
constexpr int BLOCK = 256;
constexpr int QCAP = 128;           // per-wave candidate queue entries

enum SweepMode { SWEEP_FLOW = 0, SWEEP_STEP = 1, SWEEP_SELF = 2 };

// number of float64 partial sums a block emits per mode
constexpr int NACC_FLOW = 9;   // omega[3] v[3] sum_a sum_a_d2 nnz
constexpr int NACC_STEP = 4;   // B C D E
constexpr int NACC_SELF = 2;   // sum (1/l^3 a) d2 over counted rows, nnz
constexpr int NACC_MAX = 9;

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

struct SweepArgs {
    const float4 *pos_a;   // rows (targets)
    const float *feat_a;
    const float4 *pos_b;   // columns (sources, already transformed for xy)
    const float *feat_b;
    const float *taylor;   // [nb][TAYLOR_STRIDE], SWEEP_STEP only
    double *partials;      // [gridDim.y * gridDim.x][nacc]
    int row_lo, row_hi;    // rows processed
    int nb;                // columns
    int jt;                // columns per chunk
    int first_counted;     // SWEEP_SELF: rows below contribute 0 to the sum
    int pad_;
    KernConsts kc;
};

struct TransformArgs {
    const float4 *src;
    float4 *dst;
    int n;
    float Rt[9];
    float t[3];
};

struct TaylorArgs {
    const float4 *pos;   // transformed source cloud
    float *taylor;
    int n;
    float omega[3], v[3];
    float W2[9], W3[9], W4[9];
    float u2[3], u3[3], u4[3];
};

void launch_transform(const TransformArgs &a, hipStream_t s);
void launch_taylor(const TaylorArgs &a, hipStream_t s);
// grid = (n_chunks, n_row_tiles); returns rows per block tile via rows_per_tile()
int rows_per_tile();
void launch_sweep(int mode, const SweepArgs &a, dim3 grid, hipStream_t s);
// sums partials[nblocks][nacc] in a fixed order into totals[nacc]
void launch_finalize(const double *partials, int nblocks, int nacc, double *totals,
                     hipStream_t s);
void launch_pack_cloud(const float *xyz, const float *feat, int n, int feat_colmajor,
                       float4 *pos, float *featp, hipStream_t s);

}   // namespace cvo_dev
