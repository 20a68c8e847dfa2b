/*
 * cvo_oracle.h -- CPU restatement of the CVO / Adaptive-CVO inner loop.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped product path
 * (cvo-rgbd_amd/, include/) may include, link or call this.  Allowed users:
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 *
 * PARITY UNPINNED: the reference (MaaniGhaffari/cvo-rgbd) ships no tests and
 * no golden vectors for this path, and its sources need Eigen/TBB/OpenCV/PCL,
 * none of which exist in this image, so cvo.cpp itself cannot be built.  What
 * is pinned: the neighbour-set semantics of se_kernel() against the reference's
 * own vendored nanoflann header (oracle/_ref, see Makefile), and soft sanity
 * against the decoded MATLAB transforms / mocap ground truth (tests/golden).
 *
 * Reference files followed (paths under cpp/rkhs_registration/):
 *   src/cvo.cpp:99-161   se_kernel          src/cvo.cpp:164-210  compute_flow
 *   src/cvo.cpp:213-308  compute_step_size  src/cvo.cpp:310-315  transform_pcd
 *   src/cvo.cpp:361-420  align              src/cvo.cpp:53-87    poly/dist/update_tf
 *   src/adaptive_cvo.cpp:92-151,154-272,490-555  (acvo variants)
 *   src/adaptive_cvo.cpp:385-439  function_inner_product
 *   src/LieGroup.cpp:20-27,159-186  skew, Exp_SEK3
 *   thirdparty/nanoflann.hpp:383-408,249-253  L2 metric, strict '<' result set
 *
 * Canonical arithmetic (what "bit-faithful" means for the HIP path, see
 * DESIGN.md "Arithmetic contract"): geometry, features and kernel values are
 * float32; every cross-pair accumulator is float64; squared distances use a
 * fused-multiply-add chain (what -O3 -march=native contraction makes of the
 * nanoflann loop); the two exponentials are evaluated in float64 and rounded
 * to float32 exactly as the reference's `2.0` literals force; all other
 * float32 expressions are evaluated without contraction in Eigen's
 * coefficient order.
 */
#ifndef CVO_ORACLE_H
#define CVO_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CVO_ORACLE_NFEAT 5

/* MODE_MATLAB is accepted by cvo_oracle_default_params() only: it returns mode = CVO with the
 * MATLAB object's constants and color_scale > 0 (ref matlab/@rkhs_se3_registration/
 * rkhs_se3_registration.m:10-28). */
enum { CVO_ORACLE_MODE_CVO = 0, CVO_ORACLE_MODE_ACVO = 1, CVO_ORACLE_MODE_MATLAB = 2 };
enum { CVO_ORACLE_SEARCH_DENSE = 0, CVO_ORACLE_SEARCH_GRID = 1 };

/* Hyper-parameters: cvo.cpp:18-48, adaptive_cvo.cpp:18-50. */
typedef struct cvo_oracle_params {
    int32_t mode;          /* CVO_ORACLE_MODE_* */
    int32_t max_iter;      /* MAX_ITER = 2000 */
    float ell_init;        /* cvo 0.15 ; acvo 0.1 */
    float ell_min;         /* acvo 0.0391 */
    float ell_max_init;    /* acvo 0.15 (reset in set_pcd, adaptive_cvo.cpp:477) */
    float sigma;           /* 0.1 */
    float sp_thres;        /* cvo 8e-3 ; acvo 8.315e-3 */
    float c_sp_thres;      /* acvo 8.315e-3 (cvo uses sp_thres for the colour cut) */
    float c;               /* 7 */
    float d;               /* 7 */
    float c_ell;           /* cvo 200 ; acvo 0.5 */
    float c_sigma;         /* 1 */
    float min_step;        /* 0.2 */
    float eps;             /* 5e-5 */
    float eps_2;           /* 1e-5 */
    float color_scale;     /* 0: the C++ weight.  > 0: the MATLAB object's weight
                            * a = color_scale * <c_i, c_j> * K, K = s2 exp(-d2/2l^2) kept iff K >= sp
                            * (ref rkhs_se3_registration.m:40-73,125-127), c = features 0..2 */
    double dl_step;        /* acvo 0.3 */
} cvo_oracle_params;

/* One record per executed iteration of align(). */
typedef struct cvo_oracle_trace {
    int32_t k;
    int32_t exit_code;     /* 0 = continued, 1 = break A (norms), 2 = break B (dist) */
    float ell;             /* ell used by this iteration */
    float step;
    float dist;            /* dist_se3(dR,dT); NaN if break A */
    float pad_;
    float omega[3];
    float v[3];
    double omega_d[3];     /* float64 sums before the float cast */
    double v_d[3];
    double bcde[4];
    double sum_a;          /* sum of kept A_ij (function_inner_product numerator) */
    double dl;             /* acvo only */
    int64_t nnz;           /* nnz(A) */
    int64_t nnz_xx;        /* acvo only */
    int64_t nnz_yy;        /* acvo only */
} cvo_oracle_trace;

/* Registration state carried between frames exactly as the reference object
 * carries it. */
typedef struct cvo_oracle_state {
    float R[9];            /* row-major */
    float T[3];
    float ell;
    float ell_max;
    float transform[16];   /* row-major 4x4 */
    float prev_transform[16];
    float accum_transform[16];
    int32_t iter;
    int32_t pad_;
} cvo_oracle_state;

void cvo_oracle_default_params(int mode, cvo_oracle_params *p);
void cvo_oracle_init_state(const cvo_oracle_params *p, cvo_oracle_state *s);
void cvo_oracle_set_threads(int n);   /* 0 = OpenMP default */
int  cvo_oracle_get_threads(void);

/* Deviation study (tests/test_oracle_variants.py): re-run the oracle with another admissible
 * reading of the reference's arithmetic.  0 = the contract (the default, what HIP is held to). */
#define CVO_ORACLE_VAR_ROWSUM_SEQ    1u   /* flow: per-row FLOAT product, one accumulator (cvo.cpp:197-198) */
#define CVO_ORACLE_VAR_ROWSUM_PACKET 2u   /* flow: per-row FLOAT product, 8-lane FMA redux */
#define CVO_ORACLE_VAR_D2_PLAIN      4u   /* squared distance without FMA contraction */
void cvo_oracle_set_variant(unsigned flags);
unsigned cvo_oracle_get_variant(void);

/* Thresholds (cvo.cpp:102-103): tau[0] = d2_thres, tau[1] = d2_c_thres. */
void cvo_oracle_thresholds(const cvo_oracle_params *p, float ell, float tau[2]);

/* y = Rt*y0 + t with [Rt|t] = [R^T | -R^T T] (update_tf + transform_pcd). */
void cvo_oracle_transform(const float R[9], const float T[3], const float *y0,
                          int m, float *y_out);

/* se_kernel on already-transformed clouds.  xyz arrays are AoS n x 3,
 * features are ROW-major n x 5.  Returns CSR (malloc'd; free with
 * cvo_oracle_free).  colour cut uses c_sp (cvo: sp_thres, acvo: c_sp_thres;
 * function_inner_product: sp_thres). */
int cvo_oracle_se_kernel(const cvo_oracle_params *p, float ell, float c_sp,
                         const float *xa, const float *fa, int na,
                         const float *xb, const float *fb, int nb, int search,
                         int64_t **row_ptr, int32_t **col, float **val);
void cvo_oracle_free(void *p);

/* The purely geometric neighbour sets of se_kernel: for every row point a_i
 * all b_j with d2(a_i,b_j) < tau (strict), columns ascending, value = d2.
 * This is what nanoflann's radiusSearch returns (ref cvo.cpp:110-125). */
int cvo_oracle_radius_sets(const float *xa, int na, const float *xb, int nb, float tau,
                           int search, int64_t **row_ptr, int32_t **col, float **d2);

/* compute_flow on a CSR A.  omega_d/v_d are the float64 sums (already scaled by
 * 1/c, 1/d per pair as the reference does); sum_a = sum of A values;
 * sum_a_d2 = sum of (1/ell^3 * A_ij) * ||y_j - x_i||^2 (acvo dl term). */
void cvo_oracle_flow(const cvo_oracle_params *p, float ell, const float *x, int n,
                     const float *y, int m, const int64_t *row_ptr,
                     const int32_t *col, const float *val, double omega_d[3],
                     double v_d[3], double *sum_a, double *sum_a_d2);

/* compute_step_size coefficient sums B,C,D,E. */
void cvo_oracle_step_coeffs(float ell, const float omega[3], const float v[3],
                            const float *x, int n, const float *y, int m,
                            const int64_t *row_ptr, const int32_t *col,
                            const float *val, double bcde[4]);

/* Smallest positive real root rule + min_step + 0.8 clamp (cvo.cpp:291-307). */
float cvo_oracle_pick_step(const double bcde[4], float min_step);

/* Exp_SEK3 for K=1 (Lie.cpp:159-186): dR row-major 3x3, dT 3. */
void cvo_oracle_exp_se3(const float omega[3], const float v[3], float dt,
                        float dR[9], float dT[3]);
/* ||logm([dR dT;0 1])||_F in closed form, from the twist that produced it. */
float cvo_oracle_dist_se3(const float omega[3], const float v[3], float dt);

/* acvo::function_inner_product on UNtransformed positions (acvo.cpp:385-439). */
float cvo_oracle_function_inner_product(const cvo_oracle_params *p, float ell,
                                        const float *xa, const float *fa, int na,
                                        const float *xb, const float *fb, int nb,
                                        int search);

/* One full align() (cvo.cpp:361-420 / acvo.cpp:490-555) on state *s.
 * x = fixed cloud (n), y0 = moving cloud (m).  trace may be NULL; at most
 * trace_cap records are written; returns the number of iterations executed
 * (loop bodies entered), or <0 on error. */
int cvo_oracle_align(const cvo_oracle_params *p, cvo_oracle_state *s,
                     const float *x, const float *fx, int n, const float *y0,
                     const float *fy, int m, int search,
                     cvo_oracle_trace *trace, int trace_cap);

/* Row-sharded variant used by the multi-rank tests: this rank owns target rows
 * [row_lo,row_hi) (and source rows [srow_lo,srow_hi) for the acvo Ayy terms);
 * after each local reduction `allreduce(user, buf, count)` must sum buf over
 * ranks in place.  allreduce == NULL behaves as a single rank. */
typedef void (*cvo_oracle_allreduce_fn)(void *user, double *buf, int count);
int cvo_oracle_align_sharded(const cvo_oracle_params *p, cvo_oracle_state *s,
                             const float *x, const float *fx, int n,
                             const float *y0, const float *fy, int m, int search,
                             int row_lo, int row_hi, int srow_lo, int srow_hi,
                             cvo_oracle_allreduce_fn allreduce, void *user,
                             cvo_oracle_trace *trace, int trace_cap);

#ifdef __cplusplus
}
#endif
#endif
