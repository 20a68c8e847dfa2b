/*
 * cvo_oracle.c -- CPU restatement of the CVO / Adaptive-CVO inner loop.
 * TEST INFRASTRUCTURE ONLY (see cvo_oracle.h).  PARITY UNPINNED (ibid.).
 *
 * Build: gcc -std=c11 -O3 -mavx2 -mfma -ffp-contract=off -fopenmp -fPIC -shared
 * (-ffp-contract=off is part of the arithmetic contract: every fused
 * multiply-add in this file is an explicit fmaf()).
 *
 * All "ref:" comments cite /root/reference/cpp/rkhs_registration/.
 */
#include "cvo_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int g_threads = 0;

void cvo_oracle_set_threads(int n) { g_threads = n > 0 ? n : 0; }

int cvo_oracle_get_threads(void)
{
#ifdef _OPENMP
    return g_threads > 0 ? g_threads : omp_get_max_threads();
#else
    return 1;
#endif
}

static int nthreads(void) { return cvo_oracle_get_threads(); }

/* Arithmetic variants for the deviation study (tests/test_oracle_variants.py): the places
 * where this file fixes ONE of several readings the reference's compiled arithmetic admits.
 * 0 = the contract the HIP path is held to. */
static unsigned g_variant = 0;
void cvo_oracle_set_variant(unsigned flags) { g_variant = flags; }
unsigned cvo_oracle_get_variant(void) { return g_variant; }

/* ------------------------------------------------------------------------ */
/* parameters: ref src/cvo.cpp:18-48, src/adaptive_cvo.cpp:18-50             */
/* ------------------------------------------------------------------------ */
void cvo_oracle_default_params(int mode, cvo_oracle_params *p)
{
    memset(p, 0, sizeof(*p));
    p->mode = mode == CVO_ORACLE_MODE_MATLAB ? CVO_ORACLE_MODE_CVO : mode;
    p->max_iter = 2000;
    p->sigma = 0.1f;
    p->c = 7.0f;
    p->d = 7.0f;
    p->c_sigma = 1.0f;
    p->min_step = (float)(2 * 1.0e-1);
    p->eps = (float)(5 * 1.0e-5);
    p->eps_2 = (float)1.0e-5;
    if (mode == CVO_ORACLE_MODE_ACVO) {
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
        p->c_sp_thres = 8e-3f; /* cvo.cpp:103 uses sp_thres for the colour cut */
        p->c_ell = 200.0f;
        p->dl_step = 0.0;
    }
    if (mode == CVO_ORACLE_MODE_MATLAB) {   /* ref rkhs_se3_registration.m:10-28 */
        p->sp_thres = 1e-3f;
        p->c_sp_thres = 1e-3f;
        p->eps = 5e-4f;
        p->eps_2 = 1e-4f;
        p->color_scale = 1e-5f;
    }
}

static void mat4_identity(float *m)
{
    memset(m, 0, 16 * sizeof(float));
    m[0] = m[5] = m[10] = m[15] = 1.0f;
}

void cvo_oracle_init_state(const cvo_oracle_params *p, cvo_oracle_state *s)
{
    memset(s, 0, sizeof(*s));
    s->R[0] = s->R[4] = s->R[8] = 1.0f;
    s->ell = p->ell_init;
    s->ell_max = p->ell_max_init;
    mat4_identity(s->transform);
    mat4_identity(s->prev_transform);
    mat4_identity(s->accum_transform);
}

/* ------------------------------------------------------------------------ */
/* small float32 helpers in Eigen coefficient order (no contraction)         */
/* ------------------------------------------------------------------------ */
static inline float dot3_seq(const float *a, const float *b)
{ /* dynamic-size redux: ((a0 b0 + a1 b1) + a2 b2) */
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
}

static inline float sqnorm3_fixed(const float *a)
{ /* fixed-size Vector3f redux (Eigen redux_novec_unroller): e0 + (e1 + e2) */
    return a[0] * a[0] + (a[1] * a[1] + a[2] * a[2]);
}

static inline void matvec3(const float *m /*row-major*/, const float *x, float *out)
{
    for (int r = 0; r < 3; ++r)
        out[r] = (m[3 * r] * x[0] + m[3 * r + 1] * x[1]) + m[3 * r + 2] * x[2];
}

static inline void matmul3(const float *a, const float *b, float *out)
{
    float t[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            t[3 * r + c] = (a[3 * r] * b[c] + a[3 * r + 1] * b[3 + c]) + a[3 * r + 2] * b[6 + c];
    memcpy(out, t, sizeof(t));
}

static inline void skew3(const float *w, float *m)
{ /* ref src/LieGroup.cpp:20-27 */
    m[0] = 0.0f;  m[1] = -w[2]; m[2] = w[1];
    m[3] = w[2];  m[4] = 0.0f;  m[5] = -w[0];
    m[6] = -w[1]; m[7] = w[0];  m[8] = 0.0f;
}

static inline void cross3(const float *a, const float *b, float *out)
{ /* Eigen cross(): each component is a difference of two rounded products */
    out[0] = a[1] * b[2] - a[2] * b[1];
    out[1] = a[2] * b[0] - a[0] * b[2];
    out[2] = a[0] * b[1] - a[1] * b[0];
}

/* squared distances: FMA chain in x,y,z order (ref thirdparty/nanoflann.hpp:
 * 403-406 under -O3 -march=native contraction; dim = 3 so only the tail loop
 * of evalMetric runs). */
static inline float d2_pos(const float *a, const float *b)
{
    const float d0 = a[0] - b[0], d1 = a[1] - b[1], d2 = a[2] - b[2];
    if (g_variant & CVO_ORACLE_VAR_D2_PLAIN)      /* no contraction: ((d0^2 + d1^2) + d2^2) */
        return (d0 * d0 + d1 * d1) + d2 * d2;
    return fmaf(d2, d2, fmaf(d1, d1, d0 * d0));
}

static inline float d2_feat(const float *fa, const float *fb)
{ /* ref src/cvo.cpp:145-146 */
    const float e0 = fa[0] - fb[0];
    float r = e0 * e0;
    for (int k = 1; k < CVO_ORACLE_NFEAT; ++k) {
        const float e = fa[k] - fb[k];
        r = fmaf(e, e, r);
    }
    return r;
}

/* ------------------------------------------------------------------------ */
/* kernel constants                                                          */
/* ------------------------------------------------------------------------ */
typedef struct kconsts {
    float tau, tau_c;   /* d2_thres, d2_c_thres */
    float s2, cs2;      /* sigma^2, c_sigma^2 */
    float sp;           /* keep iff a > sp */
    double ninv_2l2;    /* -1/(2 l^2)   (float64) */
    double ninv_2cl2;   /* -1/(2 c_l^2) (float64) */
    int geom_only;      /* radius-set mode: keep every d2 < tau, value = d2 */
    float cscale;       /* > 0: MATLAB weight (linear colour inner product, threshold on K) */
} kconsts;

static float logf_cr(float x) { return (float)log((double)x); }

/* fip: function_inner_product writes the spatial threshold as log(sp_thres/sigma/sigma)
 * (ref src/adaptive_cvo.cpp:391) -- two float divisions -- where se_kernel divides by the
 * float product s2 once (ref src/cvo.cpp:102, src/adaptive_cvo.cpp:100): the float quotients
 * differ in the last bit (0.8315 vs 0.83149993 with the acvo constants). */
static void make_kconsts_ex(const cvo_oracle_params *p, float ell, float c_sp, int fip, kconsts *k)
{
    const float l = ell;
    k->geom_only = 0;
    k->s2 = p->sigma * p->sigma;
    k->cs2 = p->c_sigma * p->c_sigma;
    k->sp = p->sp_thres;
    /* ref src/cvo.cpp:102-103: float = -2.0*l*l*log(sp/s2): float log (std::log
     * overload), the rest in double, stored as float. */
    k->tau = (float)(-2.0 * l * l * (double)logf_cr(fip ? p->sp_thres / p->sigma / p->sigma
                                                        : p->sp_thres / k->s2));
    k->tau_c = (float)(-2.0 * p->c_ell * p->c_ell *
                       (double)logf_cr(c_sp / p->c_sigma / p->c_sigma));
    k->ninv_2l2 = -1.0 / (2.0 * l * l);
    k->ninv_2cl2 = -1.0 / (2.0 * p->c_ell * p->c_ell);
    k->cscale = p->color_scale;
    /* MATLAB keeps K >= sp (ref rkhs_se3_registration.m:70): the radius is widened by 1e-5 so
     * that the exact test on K below decides, not the rounding of tau */
    if (k->cscale > 0.0f) k->tau = (float)((double)k->tau * 1.00001);
}

static void make_kconsts(const cvo_oracle_params *p, float ell, float c_sp, kconsts *k)
{
    make_kconsts_ex(p, ell, c_sp, 0, k);
}

void cvo_oracle_thresholds(const cvo_oracle_params *p, float ell, float tau[2])
{
    kconsts k;
    make_kconsts(p, ell, p->mode == CVO_ORACLE_MODE_ACVO ? p->c_sp_thres : p->sp_thres, &k);
    tau[0] = k.tau;
    tau[1] = k.tau_c;
}

/* pair weight for a pair that already passed d2 < tau.  Returns 0 if dropped.
 * ref src/cvo.cpp:143-153.  The exponent is formed as d2 * (-1/(2 l^2)) in
 * float64 (one rounding apart from the reference's division). */
static inline float pair_weight(const kconsts *kc, float d2, const float *fa, const float *fb)
{
    if (kc->geom_only) return d2 > 0.0f ? d2 : 1e-45f;   /* denormal marks d2 == 0 */
    if (kc->cscale > 0.0f) {   /* ref rkhs_se3_registration.m:40-53,55-73,125-127 */
        const float k = (float)((double)kc->s2 * exp((double)d2 * kc->ninv_2l2));
        if (!(k >= kc->sp)) return 0.0f;
        const float ci = kc->cscale * ((fa[0] * fb[0] + fa[1] * fb[1]) + fa[2] * fb[2]);
        return ci * k;
    }
    const float d2c = d2_feat(fa, fb);
    if (!(d2c < kc->tau_c)) return 0.0f;
    const float k = (float)((double)kc->s2 * exp((double)d2 * kc->ninv_2l2));
    const float ck = (float)((double)kc->cs2 * exp((double)d2c * kc->ninv_2cl2));
    const float a = ck * k;
    return a > kc->sp ? a : 0.0f;
}

/* ------------------------------------------------------------------------ */
/* transform                                                                 */
/* ------------------------------------------------------------------------ */
static void make_tf(const float R[9], const float T[3], float Rt[9], float t[3])
{ /* ref src/cvo.cpp:83-87: transform = [R', -R'*T] */
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) Rt[3 * r + c] = R[3 * c + r];
    for (int r = 0; r < 3; ++r) {
        const float n0 = -Rt[3 * r], n1 = -Rt[3 * r + 1], n2 = -Rt[3 * r + 2];
        t[r] = (n0 * T[0] + n1 * T[1]) + n2 * T[2];
    }
}

static void tf_to_mat4(const float Rt[9], const float t[3], float m[16])
{
    mat4_identity(m);
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) m[4 * r + c] = Rt[3 * r + c];
        m[4 * r + 3] = t[r];
    }
}

static void transform_points(const float Rt[9], const float t[3], const float *y0, int m,
                             float *y)
{ /* ref src/cvo.cpp:310-315 */
#pragma omp parallel for num_threads(nthreads()) schedule(static)
    for (int j = 0; j < m; ++j) {
        const float *p = y0 + 3 * j;
        for (int r = 0; r < 3; ++r)
            y[3 * j + r] =
                ((Rt[3 * r] * p[0] + Rt[3 * r + 1] * p[1]) + Rt[3 * r + 2] * p[2]) + t[r];
    }
}

void cvo_oracle_transform(const float R[9], const float T[3], const float *y0, int m,
                          float *y_out)
{
    float Rt[9], t[3];
    make_tf(R, T, Rt, t);
    transform_points(Rt, t, y0, m, y_out);
}

/* ------------------------------------------------------------------------ */
/* se_kernel: dense-threshold or uniform-grid neighbour search               */
/* ------------------------------------------------------------------------ */
typedef struct rowbuf {
    int32_t *col;
    float *val;
    int64_t n, cap;
} rowbuf;

static int rowbuf_push(rowbuf *b, int32_t c, float v)
{
    if (b->n == b->cap) {
        int64_t nc = b->cap ? b->cap * 2 : 4096;
        int32_t *c2 = (int32_t *)realloc(b->col, (size_t)nc * sizeof(int32_t));
        if (!c2) return -1;
        b->col = c2;
        float *v2 = (float *)realloc(b->val, (size_t)nc * sizeof(float));
        if (!v2) return -1;
        b->val = v2;
        b->cap = nc;
    }
    b->col[b->n] = c;
    b->val[b->n] = v;
    b->n++;
    return 0;
}

typedef struct grid {
    double org[3], inv_h;
    int dim[3];
    int64_t ncell;
    int64_t *start;  /* ncell + 1 */
    int32_t *idx;    /* nb, ascending inside each cell */
} grid;

static inline int64_t cell_coord(const grid *g, int axis, float p)
{
    return (int64_t)floor(((double)p - g->org[axis]) * g->inv_h);
}

static int grid_build(grid *g, const float *xb, int nb, float tau)
{
    memset(g, 0, sizeof(*g));
    double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    for (int a = 0; a < 3; ++a) { lo[a] = 1e300; hi[a] = -1e300; }
    for (int j = 0; j < nb; ++j)
        for (int a = 0; a < 3; ++a) {
            const double v = xb[3 * j + a];
            if (v < lo[a]) lo[a] = v;
            if (v > hi[a]) hi[a] = v;
        }
    /* cell edge >= search radius with margin so that +-1 cell always suffices */
    double h = sqrt((double)(tau > 0 ? tau : 0)) * (1.0 + 1e-4);
    if (!(h > 1e-9)) h = 1e-9;
    for (;;) {
        double nc = 1;
        for (int a = 0; a < 3; ++a) {
            double e = nb > 0 ? (hi[a] - lo[a]) / h : 0;
            g->dim[a] = (int)floor(e) + 1;
            nc *= g->dim[a];
        }
        if (nc <= 16777216.0) break;
        h *= 1.5;
    }
    for (int a = 0; a < 3; ++a) g->org[a] = nb > 0 ? lo[a] : 0;
    g->inv_h = 1.0 / h;
    g->ncell = (int64_t)g->dim[0] * g->dim[1] * g->dim[2];
    g->start = (int64_t *)calloc((size_t)g->ncell + 1, sizeof(int64_t));
    g->idx = (int32_t *)malloc((size_t)(nb > 0 ? nb : 1) * sizeof(int32_t));
    int64_t *cell_of = (int64_t *)malloc((size_t)(nb > 0 ? nb : 1) * sizeof(int64_t));
    if (!g->start || !g->idx || !cell_of) { free(cell_of); return -1; }
    for (int j = 0; j < nb; ++j) {
        int64_t c[3];
        for (int a = 0; a < 3; ++a) {
            c[a] = cell_coord(g, a, xb[3 * j + a]);
            if (c[a] < 0) c[a] = 0;
            if (c[a] >= g->dim[a]) c[a] = g->dim[a] - 1;
        }
        cell_of[j] = (c[2] * g->dim[1] + c[1]) * g->dim[0] + c[0];
        g->start[cell_of[j] + 1]++;
    }
    for (int64_t c = 0; c < g->ncell; ++c) g->start[c + 1] += g->start[c];
    int64_t *fill = (int64_t *)malloc((size_t)g->ncell * sizeof(int64_t));
    if (!fill) { free(cell_of); return -1; }
    memcpy(fill, g->start, (size_t)g->ncell * sizeof(int64_t));
    for (int j = 0; j < nb; ++j) g->idx[fill[cell_of[j]]++] = j;
    free(fill);
    free(cell_of);
    return 0;
}

static void grid_free(grid *g)
{
    free(g->start);
    free(g->idx);
}

static int cmp_i32(const void *a, const void *b)
{
    const int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
    return (x > y) - (x < y);
}

#define DENSE_BLK 512

/* rows [r0,r1) of the Gram matrix between clouds a (rows) and b (columns). */
static int se_kernel_rows(const kconsts *kc, const float *xa, const float *fa, int r0, int r1,
                          const float *xb, const float *fb, int nb, int search,
                          int64_t **row_ptr_out, int32_t **col_out, float **val_out)
{
    const int nrows = r1 - r0;
    const int nt = nthreads();
    rowbuf *bufs = (rowbuf *)calloc((size_t)nt, sizeof(rowbuf));
    int64_t *row_cnt = (int64_t *)calloc((size_t)(nrows > 0 ? nrows : 1), sizeof(int64_t));
    int *row_thr_lo = (int *)calloc((size_t)nt + 1, sizeof(int));
    float *bx = NULL, *by = NULL, *bz = NULL;
    grid g;
    int have_grid = 0, err = 0;
    memset(&g, 0, sizeof(g));
    if (!bufs || !row_cnt || !row_thr_lo) err = -1;

    if (!err && search == CVO_ORACLE_SEARCH_GRID) {
        if (grid_build(&g, xb, nb, kc->tau) != 0) err = -1;
        have_grid = 1;
    } else if (!err) {
        bx = (float *)malloc((size_t)(nb + DENSE_BLK) * sizeof(float));
        by = (float *)malloc((size_t)(nb + DENSE_BLK) * sizeof(float));
        bz = (float *)malloc((size_t)(nb + DENSE_BLK) * sizeof(float));
        if (!bx || !by || !bz) err = -1;
        else
            for (int j = 0; j < nb; ++j) {
                bx[j] = xb[3 * j];
                by[j] = xb[3 * j + 1];
                bz[j] = xb[3 * j + 2];
            }
    }

    if (!err) {
        /* static contiguous row blocks per thread => CSR rows stay in order */
        for (int t = 0; t <= nt; ++t) row_thr_lo[t] = (int)((int64_t)nrows * t / nt);
#pragma omp parallel num_threads(nt)
        {
#ifdef _OPENMP
            const int t = omp_get_thread_num();
#else
            const int t = 0;
#endif
            rowbuf *rb = &bufs[t];
            int32_t *cand = NULL;
            int64_t cand_cap = 0;
            float d2buf[DENSE_BLK];
            int lerr = 0;
            for (int ri = row_thr_lo[t]; ri < row_thr_lo[t + 1] && !lerr; ++ri) {
                const int i = r0 + ri;
                const float *xi = xa + 3 * i;
                const float *fi = fa + CVO_ORACLE_NFEAT * i;
                const int64_t n_before = rb->n;
                if (!have_grid) {
                    const float x0 = xi[0], x1 = xi[1], x2 = xi[2];
                    for (int jb = 0; jb < nb; jb += DENSE_BLK) {
                        const int je = jb + DENSE_BLK < nb ? DENSE_BLK : nb - jb;
                        for (int u = 0; u < je; ++u) {
                            const float e0 = x0 - bx[jb + u], e1 = x1 - by[jb + u],
                                        e2 = x2 - bz[jb + u];
                            d2buf[u] = fmaf(e2, e2, fmaf(e1, e1, e0 * e0));
                        }
                        for (int u = 0; u < je; ++u) {
                            if (d2buf[u] < kc->tau) { /* strict, nanoflann.hpp:250 */
                                const int j = jb + u;
                                const float a = pair_weight(kc, d2buf[u], fi,
                                                            fb + CVO_ORACLE_NFEAT * j);
                                if (a > 0.0f && rowbuf_push(rb, j, a)) lerr = 1;
                            }
                        }
                    }
                } else {
                    int64_t c[3], ncand = 0;
                    for (int a = 0; a < 3; ++a) c[a] = cell_coord(&g, a, xi[a]);
                    for (int64_t cz = c[2] - 1; cz <= c[2] + 1; ++cz) {
                        if (cz < 0 || cz >= g.dim[2]) continue;
                        for (int64_t cy = c[1] - 1; cy <= c[1] + 1; ++cy) {
                            if (cy < 0 || cy >= g.dim[1]) continue;
                            for (int64_t cx = c[0] - 1; cx <= c[0] + 1; ++cx) {
                                if (cx < 0 || cx >= g.dim[0]) continue;
                                const int64_t cell = (cz * g.dim[1] + cy) * g.dim[0] + cx;
                                for (int64_t q = g.start[cell]; q < g.start[cell + 1]; ++q) {
                                    const int32_t j = g.idx[q];
                                    if (d2_pos(xi, xb + 3 * j) < kc->tau) {
                                        if (ncand == cand_cap) {
                                            cand_cap = cand_cap ? cand_cap * 2 : 1024;
                                            int32_t *c2 = (int32_t *)realloc(
                                                cand, (size_t)cand_cap * sizeof(int32_t));
                                            if (!c2) { lerr = 1; break; }
                                            cand = c2;
                                        }
                                        cand[ncand++] = j;
                                    }
                                }
                            }
                        }
                    }
                    if (!lerr) {
                        qsort(cand, (size_t)ncand, sizeof(int32_t), cmp_i32);
                        for (int64_t q = 0; q < ncand && !lerr; ++q) {
                            const int32_t j = cand[q];
                            const float d2 = d2_pos(xi, xb + 3 * j);
                            const float a =
                                pair_weight(kc, d2, fi, fb + CVO_ORACLE_NFEAT * j);
                            if (a > 0.0f && rowbuf_push(rb, j, a)) lerr = 1;
                        }
                    }
                }
                row_cnt[ri] = rb->n - n_before;
            }
            free(cand);
            if (lerr) {
#pragma omp atomic write
                err = -1;
            }
        }
    }

    int64_t *row_ptr = NULL;
    int32_t *col = NULL;
    float *val = NULL;
    if (!err) {
        row_ptr = (int64_t *)malloc((size_t)(nrows + 1) * sizeof(int64_t));
        if (!row_ptr) err = -1;
    }
    if (!err) {
        row_ptr[0] = 0;
        for (int r = 0; r < nrows; ++r) row_ptr[r + 1] = row_ptr[r] + row_cnt[r];
        const int64_t nnz = row_ptr[nrows];
        col = (int32_t *)malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(int32_t));
        val = (float *)malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(float));
        if (!col || !val) err = -1;
        else
            for (int t = 0; t < nt; ++t) {
                const int64_t off = row_ptr[row_thr_lo[t]];
                if (bufs[t].n) {
                    memcpy(col + off, bufs[t].col, (size_t)bufs[t].n * sizeof(int32_t));
                    memcpy(val + off, bufs[t].val, (size_t)bufs[t].n * sizeof(float));
                }
            }
    }
    if (bufs)
        for (int t = 0; t < nt; ++t) { free(bufs[t].col); free(bufs[t].val); }
    free(bufs);
    free(row_cnt);
    free(row_thr_lo);
    free(bx); free(by); free(bz);
    if (have_grid) grid_free(&g);
    if (err) { free(row_ptr); free(col); free(val); return -1; }
    *row_ptr_out = row_ptr;
    *col_out = col;
    *val_out = val;
    return 0;
}

int cvo_oracle_se_kernel(const cvo_oracle_params *p, float ell, float c_sp, const float *xa,
                         const float *fa, int na, const float *xb, const float *fb, int nb,
                         int search, int64_t **row_ptr, int32_t **col, float **val)
{
    kconsts kc;
    make_kconsts(p, ell, c_sp, &kc);
    return se_kernel_rows(&kc, xa, fa, 0, na, xb, fb, nb, search, row_ptr, col, val);
}

int cvo_oracle_radius_sets(const float *xa, int na, const float *xb, int nb, float tau,
                            int search, int64_t **row_ptr, int32_t **col, float **d2)
{
    kconsts kc;
    memset(&kc, 0, sizeof(kc));
    kc.geom_only = 1;
    kc.tau = tau;
    /* features are not read in geom_only mode; pass the positions as dummies */
    const int rc = se_kernel_rows(&kc, xa, xa, 0, na, xb, xb, nb, search, row_ptr, col, d2);
    if (rc == 0) {
        const int64_t nnz = (*row_ptr)[na];
        for (int64_t q = 0; q < nnz; ++q)
            if ((*d2)[q] == 1e-45f) (*d2)[q] = 0.0f;
    }
    return rc;
}

void cvo_oracle_free(void *p) { free(p); }

/* ------------------------------------------------------------------------ */
/* compute_flow (ref src/cvo.cpp:164-210, src/adaptive_cvo.cpp:154-272)      */
/* rows [r0,r1) of x; CSR is local to that row range                         */
/* ------------------------------------------------------------------------ */
/* The literal reading of ref src/cvo.cpp:197-198: per row i, `1/c*Ai` is a float row vector,
 * `(1/c*Ai)*cross_xy` a FLOAT 1 x k by k x 3 product, cast to double only afterwards, and the
 * rows are added in double in whatever order the TBB workers take the lock (here: row order).
 * Eigen evaluates the product as three dot products; the order inside one is the compiler's:
 *   ROWSUM_SEQ     one accumulator, plain multiply-add in column order;
 *   ROWSUM_PACKET  eight lanes (AVX packets) of fused multiply-adds over columns j = l mod 8,
 *                  a scalar tail, lanes reduced pairwise -- the shape of Eigen's vectorised redux.
 * Deviation study only; the dl terms (acvo) keep the float64 per-pair sum. */
static float rowdot_f32(const float *a, const float *b, int k, int packet)
{
    if (!packet) {
        float s = 0;
        for (int j = 0; j < k; ++j) s = s + a[j] * b[j];
        return s;
    }
    float lane[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int kv = k & ~7;
    for (int j = 0; j < kv; j += 8)
        for (int l = 0; l < 8; ++l) lane[l] = fmaf(a[j + l], b[j + l], lane[l]);
    float s = ((lane[0] + lane[4]) + (lane[2] + lane[6])) + ((lane[1] + lane[5]) + (lane[3] + lane[7]));
    for (int j = kv; j < k; ++j) s = fmaf(a[j], b[j], s);
    return s;
}

static void flow_rows_f32(const cvo_oracle_params *p, float ell, const float *x, int r0, int r1,
                          const float *y, const int64_t *row_ptr, const int32_t *col,
                          const float *val, double out[8])
{
    const float inv_c = 1 / p->c, inv_d = 1 / p->d;
    const float ell_3 = ell * ell * ell, inv_l3 = 1 / ell_3;
    const int packet = (g_variant & CVO_ORACLE_VAR_ROWSUM_PACKET) != 0;
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t kmax = 1;
    for (int ri = 0; ri < r1 - r0; ++ri)
        if (row_ptr[ri + 1] - row_ptr[ri] > kmax) kmax = row_ptr[ri + 1] - row_ptr[ri];
    float *buf = (float *)malloc((size_t)kmax * 8 * sizeof(float));
    float *ac = buf, *ad = buf + kmax, *cr = buf + 2 * kmax, *df = buf + 5 * kmax;
    for (int ri = 0; ri < r1 - r0; ++ri) {
        const float *xi = x + 3 * (r0 + ri);
        const int k = (int)(row_ptr[ri + 1] - row_ptr[ri]);
        for (int j = 0; j < k; ++j) {
            const int64_t q = row_ptr[ri] + j;
            const float *yj = y + 3 * col[q];
            float c3[3];
            cross3(xi, yj, c3);
            ac[j] = inv_c * val[q];
            ad[j] = inv_d * val[q];
            for (int a = 0; a < 3; ++a) {                 /* column-major k x 3 */
                cr[a * kmax + j] = c3[a];
                df[a * kmax + j] = yj[a] - xi[a];
            }
            acc[6] += (double)val[q];
            acc[7] += (double)((inv_l3 * val[q]) * d2_pos(xi, yj));
        }
        for (int a = 0; a < 3; ++a) {
            acc[a] += (double)rowdot_f32(ac, cr + a * kmax, k, packet);
            acc[3 + a] += (double)rowdot_f32(ad, df + a * kmax, k, packet);
        }
    }
    free(buf);
    memcpy(out, acc, sizeof(acc));
}

static void flow_rows(const cvo_oracle_params *p, float ell, const float *x, int r0, int r1,
                      const float *y, const int64_t *row_ptr, const int32_t *col,
                      const float *val, double out[8])
{
    const float inv_c = 1 / p->c, inv_d = 1 / p->d;   /* `1/c` is float */
    const float ell_3 = ell * ell * ell;              /* ref acvo.cpp:172 */
    const float inv_l3 = 1 / ell_3;
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int nrows = r1 - r0;
    if (g_variant & (CVO_ORACLE_VAR_ROWSUM_SEQ | CVO_ORACLE_VAR_ROWSUM_PACKET)) {
        flow_rows_f32(p, ell, x, r0, r1, y, row_ptr, col, val, out);
        return;
    }
#pragma omp parallel num_threads(nthreads())
    {
        double w[3] = {0, 0, 0}, vv[3] = {0, 0, 0}, sa = 0, sad2 = 0;
#pragma omp for schedule(static)
        for (int ri = 0; ri < nrows; ++ri) {
            const float *xi = x + 3 * (r0 + ri);
            for (int64_t q = row_ptr[ri]; q < row_ptr[ri + 1]; ++q) {
                const float *yj = y + 3 * col[q];
                const float a = val[q];
                float cr[3], df[3];
                cross3(xi, yj, cr);                       /* x_i x y_j */
                for (int k = 0; k < 3; ++k) df[k] = yj[k] - xi[k];
                const float ac = inv_c * a, ad = inv_d * a; /* (1/c*Ai) first */
                for (int k = 0; k < 3; ++k) {
                    w[k] += (double)(ac * cr[k]);
                    vv[k] += (double)(ad * df[k]);
                }
                sa += (double)a;
                /* acvo dl term: (1/ell^3 * A_ij) * ||y_j - x_i||^2 */
                sad2 += (double)((inv_l3 * a) * d2_pos(xi, yj));
            }
        }
#pragma omp critical
        {
            for (int k = 0; k < 3; ++k) { acc[k] += w[k]; acc[3 + k] += vv[k]; }
            acc[6] += sa;
            acc[7] += sad2;
        }
    }
    memcpy(out, acc, sizeof(acc));
}

void cvo_oracle_flow(const cvo_oracle_params *p, float ell, const float *x, int n,
                     const float *y, int m, const int64_t *row_ptr, const int32_t *col,
                     const float *val, double omega_d[3], double v_d[3], double *sum_a,
                     double *sum_a_d2)
{
    (void)m;
    double out[8];
    flow_rows(p, ell, x, 0, n, y, row_ptr, col, val, out);
    for (int k = 0; k < 3; ++k) { omega_d[k] = out[k]; v_d[k] = out[3 + k]; }
    if (sum_a) *sum_a = out[6];
    if (sum_a_d2) *sum_a_d2 = out[7];
}

/* self-Gram dl term: sum over rows [r0,r1) of (1/ell^3 A_ij) ||b_j - a_i||^2,
 * rows below `first_counted` contribute 0 (acvo quirk: sum_diff_yy_2 is never
 * filled in the first loop, ref acvo.cpp:213-223). */
static double self_rows(float ell, const float *xa, int r0, int r1, int first_counted,
                        const int64_t *row_ptr, const int32_t *col, const float *val)
{
    const float ell_3 = ell * ell * ell;
    const float inv_l3 = 1 / ell_3;
    double tot = 0;
    const int nrows = r1 - r0;
#pragma omp parallel for num_threads(nthreads()) schedule(static) reduction(+ : tot)
    for (int ri = 0; ri < nrows; ++ri) {
        const int i = r0 + ri;
        if (i < first_counted) continue;
        const float *xi = xa + 3 * i;
        for (int64_t q = row_ptr[ri]; q < row_ptr[ri + 1]; ++q)
            tot += (double)((inv_l3 * val[q]) * d2_pos(xi, xa + 3 * col[q]));
    }
    return tot;
}

/* ------------------------------------------------------------------------ */
/* compute_step_size (ref src/cvo.cpp:213-308)                               */
/* ------------------------------------------------------------------------ */
typedef struct taylor {
    float xiz[3], xi2z[3], xi3z[3], xi4z[3];
    float normxiz2, xiz_dot_xi2z, epsil_const;
} taylor;

typedef struct xi_consts {
    float omega[3], v[3];
    float W2[9], W3[9], W4[9];   /* omega_hat^2..^4, left-assoc products */
    float u2[3], u3[3], u4[3];   /* omega_hat v, omega_hat^2 v, omega_hat^3 v */
} xi_consts;

static void make_xi_consts(const float omega[3], const float v[3], xi_consts *c)
{
    float W[9];
    memcpy(c->omega, omega, 3 * sizeof(float));
    memcpy(c->v, v, 3 * sizeof(float));
    skew3(omega, W);
    matmul3(W, W, c->W2);       /* omega_hat*omega_hat */
    matmul3(c->W2, W, c->W3);   /* (omega_hat*omega_hat)*omega_hat */
    matmul3(c->W3, W, c->W4);
    matvec3(W, v, c->u2);
    matvec3(c->W2, v, c->u3);
    matvec3(c->W3, v, c->u4);
}

static inline void taylor_point(const xi_consts *c, const float *y, taylor *t)
{ /* ref src/cvo.cpp:226-238 */
    float cr[3], m[3];
    cross3(c->omega, y, cr);
    for (int k = 0; k < 3; ++k) t->xiz[k] = cr[k] + c->v[k];
    matvec3(c->W2, y, m);
    for (int k = 0; k < 3; ++k) t->xi2z[k] = m[k] + c->u2[k];
    matvec3(c->W3, y, m);
    for (int k = 0; k < 3; ++k) t->xi3z[k] = m[k] + c->u3[k];
    matvec3(c->W4, y, m);
    for (int k = 0; k < 3; ++k) t->xi4z[k] = m[k] + c->u4[k];
    t->normxiz2 = dot3_seq(t->xiz, t->xiz);
    t->xiz_dot_xi2z = -dot3_seq(t->xiz, t->xi2z);
    t->epsil_const = dot3_seq(t->xi2z, t->xi2z) + 2 * dot3_seq(t->xiz, t->xi3z);
}

static inline float sdot3(float s, const float *r, const float *d)
{ /* (s * row) * col, Eigen inner product: sum_k (s r_k) d_k, sequential */
    return ((s * r[0]) * d[0] + (s * r[1]) * d[1]) + (s * r[2]) * d[2];
}

static void step_rows(float ell, const float omega[3], const float v[3], const float *x,
                      int r0, int r1, const float *y, int m, const int64_t *row_ptr,
                      const int32_t *col, const float *val, double bcde[4])
{
    xi_consts xc;
    make_xi_consts(omega, v, &xc);
    taylor *ty = (taylor *)malloc((size_t)(m > 0 ? m : 1) * sizeof(taylor));
    if (!ty) { bcde[0] = bcde[1] = bcde[2] = bcde[3] = NAN; return; }
#pragma omp parallel for num_threads(nthreads()) schedule(static)
    for (int j = 0; j < m; ++j) taylor_point(&xc, y + 3 * j, &ty[j]);

    const float temp_coef = (float)(1 / (2.0 * ell * ell));   /* ref cvo.cpp:241 */
    const float cb = (float)(-2.0 * temp_coef);
    const float cg = -temp_coef;
    const float cd = (float)(2.0 * temp_coef);
    double B = 0, C = 0, D = 0, E = 0;
    const int nrows = r1 - r0;
#pragma omp parallel for num_threads(nthreads()) schedule(static) reduction(+ : B, C, D, E)
    for (int ri = 0; ri < nrows; ++ri) {
        const float *xi = x + 3 * (r0 + ri);
        double Bi = 0, Ci = 0, Di = 0, Ei = 0;
        for (int64_t q = row_ptr[ri]; q < row_ptr[ri + 1]; ++q) {
            const int j = col[q];
            const taylor *t = &ty[j];
            const float *yj = y + 3 * j;
            const float df[3] = {xi[0] - yj[0], xi[1] - yj[1], xi[2] - yj[2]};
            const float beta = sdot3(cb, t->xiz, df);
            const float gamma = cg * (t->normxiz2 + sdot3(2.0f, t->xi2z, df));
            const float delta = cd * (t->xiz_dot_xi2z + sdot3(-1.0f, t->xi3z, df));
            const float epsil = cg * (t->epsil_const + sdot3(2.0f, t->xi4z, df));
            const float A = val[q];
            /* ref src/cvo.cpp:275-280: C promotion rules spelled out */
            Bi += (double)(A * beta);
            Ci += (double)A * ((double)gamma + (double)(beta * beta) / 2.0);
            Di += (double)A * ((double)(delta + beta * gamma) +
                               (double)(beta * beta * beta) / 6.0);
            Ei += (double)A *
                  ((((double)(epsil + beta * delta) + 1 / 2.0 * beta * beta * gamma) +
                    1 / 2.0 * gamma * gamma) +
                   1 / 24.0 * beta * beta * beta * beta);
        }
        B += Bi; C += Ci; D += Di; E += Ei;
    }
    free(ty);
    bcde[0] = B; bcde[1] = C; bcde[2] = D; bcde[3] = E;
}

void cvo_oracle_step_coeffs(float ell, const float omega[3], const float v[3], const float *x,
                            int n, const float *y, int m, const int64_t *row_ptr,
                            const int32_t *col, const float *val, double bcde[4])
{
    step_rows(ell, omega, v, x, 0, n, y, m, row_ptr, col, val, bcde);
}

/* Smallest positive real root of 4E s^3 + 3D s^2 + 2C s + B, else min_step,
 * clamped to 0.8.  ref src/cvo.cpp:53-69,291-307 solves the float companion
 * matrix with Eigen's EigenSolver and accepts roots with imag()==0; here: the
 * float coefficients and the float division by the leading one exactly as the
 * reference forms them, then the real roots of the monic cubic are bracketed
 * between its stationary points and refined by 64-way sectioning in float64
 * until its float value is decided (only +,-,*,/,sqrt: bit-reproducible).
 * Degenerate cubics (0/0 -> NaN eigenvalues in the reference) give min_step. */
static double cubic_eval(double a, double b, double c, double s)
{
    return ((s + a) * s + b) * s + c;
}

typedef struct cubic_bracket {
    double a, b, c, lo, hi;
    int found, increasing;
} cubic_bracket;

static cubic_bracket make_bracket(const double bcde[4])
{
    cubic_bracket B;
    memset(&B, 0, sizeof(B));
    B.increasing = 1;
    const float c3 = (float)(4.0 * (float)bcde[3]);
    const float c2 = (float)(3.0 * (float)bcde[2]);
    const float c1 = (float)(2.0 * (float)bcde[1]);
    const float c0 = (float)bcde[0];
    const int finite = (c3 == c3) && (c2 == c2) && (c1 == c1) && (c0 == c0) &&
                       fabsf(c3) <= 3.0e38f && fabsf(c2) <= 3.0e38f && fabsf(c1) <= 3.0e38f &&
                       fabsf(c0) <= 3.0e38f;
    if (!(c3 != 0.0f && finite)) return B;
    /* companion-matrix first row: -(coef/coef(0)) in float */
    const float qa = c2 / c3, qb = c1 / c3, qc = c0 / c3;
    if (!(fabsf(qa) <= 3.0e38f && fabsf(qb) <= 3.0e38f && fabsf(qc) <= 3.0e38f)) return B;
    const double a = (double)qa, b = (double)qb, c = (double)qc;
    B.a = a; B.b = b; B.c = c;
    double M = fabs(a);
    if (fabs(b) > M) M = fabs(b);
    if (fabs(c) > M) M = fabs(c);
    const double U = 1.0 + M;   /* Cauchy bound */
    const double f0 = c;
    const double disc = a * a - 3.0 * b;
    if (!(disc > 0.0)) {
        if (f0 < 0.0) { B.lo = 0.0; B.hi = U; B.increasing = 1; B.found = 1; }
        return B;
    }
    const double sq = sqrt(disc);
    const double s1 = (-a - sq) / 3.0;   /* local maximum */
    const double s2 = (-a + sq) / 3.0;   /* local minimum */
    if (s1 > 0.0 && f0 < 0.0) {
        const double f1 = cubic_eval(a, b, c, s1);
        if (f1 >= 0.0) { B.lo = 0.0; B.hi = s1; B.increasing = 1; B.found = 1; return B; }
    }
    if (s2 > 0.0) {
        const double lo = s1 > 0.0 ? s1 : 0.0;
        const double fl = cubic_eval(a, b, c, lo);
        const double f2 = cubic_eval(a, b, c, s2);
        if (fl > 0.0 && f2 <= 0.0) { B.lo = lo; B.hi = s2; B.increasing = 0; B.found = 1; return B; }
    }
    {
        const double lo = s2 > 0.0 ? s2 : 0.0;
        const double fl = cubic_eval(a, b, c, lo);
        if (fl < 0.0) { B.lo = lo; B.hi = U; B.increasing = 1; B.found = 1; }
    }
    return B;
}

/* 64-way sectioning until both ends round to the same float (see header comment
 * above; the GPU runs the same rounds with one lane per interior point). */
static double section_root(const cubic_bracket *B)
{
    double lo = B->lo, hi = B->hi;
    for (int round = 0; round < 64; ++round) {
        if ((float)lo == (float)hi) break;
        const double w = hi - lo;
        double nlo = lo, nhi = hi;
        for (int l = 0; l < 64; ++l) {
            const double x = lo + w * ((double)(l + 1) * (1.0 / 65.0));
            const int inside = x > lo && x < hi;
            const double f = cubic_eval(B->a, B->b, B->c, x);
            const int go_right = inside && (B->increasing ? (f < 0.0) : (f > 0.0));
            if (!go_right) {
                if (inside) nhi = x;
                break;
            }
            nlo = x;
        }
        if (nlo == lo && nhi == hi) break;
        lo = nlo;
        hi = nhi;
    }
    return hi;
}

float cvo_oracle_pick_step(const double bcde[4], float min_step)
{
    const cubic_bracket B = make_bracket(bcde);
    float step = min_step;
    if (B.found) {
        const float r = (float)section_root(&B);
        if (r > 0.0f) step = r;
    }
    step = step > 0.8 ? (float)0.8 : step;   /* ref cvo.cpp:307 */
    return step;
}

/* ------------------------------------------------------------------------ */
/* Exp_SEK3 (ref src/LieGroup.cpp:159-186), K = 1                            */
/* ------------------------------------------------------------------------ */
/* sin / cos in float64 from +,-,*,/,floor only (Cody-Waite reduction by pi/2,
 * nested Taylor polynomials on [-pi/4, pi/4]); |error| < 1e-16 for |x| < 1e5.
 * The reference calls float sin/cos (std:: overloads); rounding an accurate
 * float64 value to float is the correctly-rounded float result except with
 * probability ~1e-8, and it is bit-reproducible across CPU and GPU. */
static void sincos_det(double x, double *s_out, double *c_out)
{
    const double two_over_pi = 0.63661977236758134308;
    const double pio2_hi = 1.57079632673412561417e+00;
    const double pio2_lo = 6.07710050650619224932e-11;
    const double kd = floor(x * two_over_pi + 0.5);
    const double r = (x - kd * pio2_hi) - kd * pio2_lo;
    const double r2 = r * r;
    double ps = 1.0;
    ps = 1.0 - r2 * (1.0 / (18.0 * 19.0)) * ps;
    ps = 1.0 - r2 * (1.0 / (16.0 * 17.0)) * ps;
    ps = 1.0 - r2 * (1.0 / (14.0 * 15.0)) * ps;
    ps = 1.0 - r2 * (1.0 / (12.0 * 13.0)) * ps;
    ps = 1.0 - r2 * (1.0 / (10.0 * 11.0)) * ps;
    ps = 1.0 - r2 * (1.0 / (8.0 * 9.0)) * ps;
    ps = 1.0 - r2 * (1.0 / (6.0 * 7.0)) * ps;
    ps = 1.0 - r2 * (1.0 / (4.0 * 5.0)) * ps;
    ps = 1.0 - r2 * (1.0 / (2.0 * 3.0)) * ps;
    const double sr = r * ps;
    double pc = 1.0;
    pc = 1.0 - r2 * (1.0 / (17.0 * 18.0)) * pc;
    pc = 1.0 - r2 * (1.0 / (15.0 * 16.0)) * pc;
    pc = 1.0 - r2 * (1.0 / (13.0 * 14.0)) * pc;
    pc = 1.0 - r2 * (1.0 / (11.0 * 12.0)) * pc;
    pc = 1.0 - r2 * (1.0 / (9.0 * 10.0)) * pc;
    pc = 1.0 - r2 * (1.0 / (7.0 * 8.0)) * pc;
    pc = 1.0 - r2 * (1.0 / (5.0 * 6.0)) * pc;
    pc = 1.0 - r2 * (1.0 / (3.0 * 4.0)) * pc;
    pc = 1.0 - r2 * (1.0 / (1.0 * 2.0)) * pc;
    const double cr = pc;
    const double q4 = kd - 4.0 * floor(kd * 0.25);
    double sn, cs;
    if (q4 == 0.0) { sn = sr; cs = cr; }
    else if (q4 == 1.0) { sn = cr; cs = -sr; }
    else if (q4 == 2.0) { sn = -sr; cs = -cr; }
    else { sn = -cr; cs = sr; }
    *s_out = sn;
    *c_out = cs;
}

void cvo_oracle_exp_se3(const float omega[3], const float v[3], float dt, float dR[9],
                        float dT[3])
{
    const float TOL = 1e-6f;
    const float theta = sqrtf(sqnorm3_fixed(omega));
    float Jl[9];
    static const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (theta < TOL) {
        memcpy(dR, I, sizeof(I));
        memcpy(Jl, I, sizeof(I));   /* quirk: Jl = I, not dt*I */
    } else {
        float A[9], A2[9];
        skew3(omega, A);
        const float theta2 = theta * theta;
        double sd, cd;
        sincos_det((double)(dt * theta), &sd, &cd);
        const float stheta = (float)sd;
        const float ctheta = (float)cd;
        const float omc = (1 - ctheta) / theta2;
        matmul3(A, A, A2);
        const float s1 = stheta / theta;
        const float j3 = (dt * theta - stheta) / (theta2 * theta);
        for (int k = 0; k < 9; ++k) {
            dR[k] = (I[k] + s1 * A[k]) + omc * A2[k];
            Jl[k] = (dt * I[k] + omc * A[k]) + j3 * A2[k];
        }
    }
    matvec3(Jl, v, dT);
}

float cvo_oracle_dist_se3(const float omega[3], const float v[3], float dt)
{ /* ||logm([dR dT;0 1])||_F; ref src/cvo.cpp:71-81 uses Eigen's float matrix
     log.  Closed form: dt*sqrt(2|w|^2+|v|^2); small-angle branch (dR=I, dT=v):
     |v|.  Evaluated in float64, rounded to float. */
    const double w2 = (double)omega[0] * omega[0] + (double)omega[1] * omega[1] +
                      (double)omega[2] * omega[2];
    const double v2 = (double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2];
    const float theta = sqrtf(sqnorm3_fixed(omega));
    if (theta < 1e-6f) return (float)sqrt(v2);
    return (float)((double)dt * sqrt(2.0 * w2 + v2));
}

/* ------------------------------------------------------------------------ */
/* function_inner_product (ref src/adaptive_cvo.cpp:385-439)                 */
/* ------------------------------------------------------------------------ */
float cvo_oracle_function_inner_product(const cvo_oracle_params *p, float ell, const float *xa,
                                        const float *fa, int na, const float *xb,
                                        const float *fb, int nb, int search)
{
    int64_t *rp = NULL;
    int32_t *col = NULL;
    float *val = NULL;
    kconsts kc;
    make_kconsts_ex(p, ell, p->sp_thres, 1, &kc);   /* ref acvo.cpp:391-392: its own threshold lines */
    if (se_kernel_rows(&kc, xa, fa, 0, na, xb, fb, nb, search, &rp, &col, &val))
        return NAN;
    double sum_a = 0;
    const int64_t nnz = rp[na];
    for (int64_t q = 0; q < nnz; ++q) sum_a += (double)val[q];
    free(rp); free(col); free(val);
    return (float)(sum_a / (double)nnz);
}

/* ------------------------------------------------------------------------ */
/* align (ref src/cvo.cpp:361-420, src/adaptive_cvo.cpp:490-555)             */
/* ------------------------------------------------------------------------ */
static void mat4_mul(const float *a, const float *b, float *out)
{
    float t[16];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c)
            t[4 * r + c] = ((a[4 * r] * b[c] + a[4 * r + 1] * b[4 + c]) +
                            a[4 * r + 2] * b[8 + c]) + a[4 * r + 3] * b[12 + c];
    memcpy(out, t, sizeof(t));
}

int cvo_oracle_align_sharded(const cvo_oracle_params *p, cvo_oracle_state *s, const float *x,
                             const float *fx, int n, const float *y0, const float *fy, int m,
                             int search, int row_lo, int row_hi, int srow_lo, int srow_hi,
                             cvo_oracle_allreduce_fn allreduce, void *user,
                             cvo_oracle_trace *trace, int trace_cap)
{
    const int acvo = (p->mode == CVO_ORACLE_MODE_ACVO);
    float *y = (float *)malloc((size_t)(m > 0 ? m : 1) * 3 * sizeof(float));
    if (!y) return -1;
    if (acvo) { /* set_pcd tail, ref acvo.cpp:476-478 */
        s->ell = p->ell_init;
        s->ell_max = p->ell_max_init;
    }
    float Rt[9], t[3];
    int executed = 0;
    for (int k = 0; k < p->max_iter; ++k) {
        cvo_oracle_trace tr;
        memset(&tr, 0, sizeof(tr));
        tr.k = k;
        tr.ell = s->ell;
        tr.dist = NAN;
        /* update_tf + transform_pcd */
        make_tf(s->R, s->T, Rt, t);
        tf_to_mat4(Rt, t, s->transform);
        transform_points(Rt, t, y0, m, y);

        /* compute_flow */
        kconsts kc;
        make_kconsts(p, s->ell, acvo ? p->c_sp_thres : p->sp_thres, &kc);
        int64_t *rp = NULL; int32_t *col = NULL; float *val = NULL;
        if (se_kernel_rows(&kc, x, fx, row_lo, row_hi, y, fy, m, search, &rp, &col, &val)) {
            free(y);
            return -1;
        }
        double red[13];
        memset(red, 0, sizeof(red));
        flow_rows(p, s->ell, x, row_lo, row_hi, y, rp, col, val, red); /* 0..7 */
        red[8] = (double)rp[row_hi - row_lo];                          /* nnz(A) */
        if (acvo) {
            int64_t *rpx = NULL, *rpy = NULL; int32_t *cx = NULL, *cy = NULL;
            float *vx = NULL, *vy = NULL;
            if (se_kernel_rows(&kc, x, fx, row_lo, row_hi, x, fx, n, search, &rpx, &cx, &vx) ||
                se_kernel_rows(&kc, y, fy, srow_lo, srow_hi, y, fy, m, search, &rpy, &cy, &vy)) {
                free(rp); free(col); free(val); free(rpx); free(cx); free(vx); free(y);
                return -1;
            }
            red[9] = self_rows(s->ell, x, row_lo, row_hi, 0, rpx, cx, vx);
            red[10] = (double)rpx[row_hi - row_lo];
            /* Ayy rows below num_fixed contribute 0 (quirk 5) */
            red[11] = self_rows(s->ell, y, srow_lo, srow_hi, n, rpy, cy, vy);
            red[12] = (double)rpy[srow_hi - srow_lo];
            free(rpx); free(cx); free(vx); free(rpy); free(cy); free(vy);
        }
        if (allreduce) allreduce(user, red, 13);
        float omega[3], v[3];
        for (int q = 0; q < 3; ++q) {
            tr.omega_d[q] = red[q];
            tr.v_d[q] = red[3 + q];
            omega[q] = (float)red[q];
            v[q] = (float)red[3 + q];
            tr.omega[q] = omega[q];
            tr.v[q] = v[q];
        }
        tr.sum_a = red[6];
        tr.nnz = (int64_t)red[8];
        double dl = 0;
        if (acvo) {
            tr.nnz_xx = (int64_t)red[10];
            tr.nnz_yy = (int64_t)red[12];
            const double num = (red[11] - 2.0 * red[7]) + red[9];
            dl = num / (double)(tr.nnz_xx + tr.nnz_yy - 2 * tr.nnz);
            tr.dl = dl;
        }

        /* compute_step_size */
        double bcde[4];
        step_rows(s->ell, omega, v, x, row_lo, row_hi, y, m, rp, col, val, bcde);
        free(rp); free(col); free(val);
        if (allreduce) allreduce(user, bcde, 4);
        memcpy(tr.bcde, bcde, sizeof(bcde));
        const float step = cvo_oracle_pick_step(bcde, p->min_step);
        tr.step = step;
        executed = k + 1;

        /* break A */
        int brk = 0;
        if (acvo) {
            const double nw = sqrt((double)omega[0] * omega[0] +
                                   ((double)omega[1] * omega[1] + (double)omega[2] * omega[2]));
            const double nv = sqrt((double)v[0] * v[0] +
                                   ((double)v[1] * v[1] + (double)v[2] * v[2]));
            brk = (nw < (double)p->eps && nv < (double)p->eps);
        } else {
            brk = (sqrtf(sqnorm3_fixed(omega)) < p->eps && sqrtf(sqnorm3_fixed(v)) < p->eps);
        }
        if (brk) {
            s->iter = k;
            tr.exit_code = 1;
            if (trace && k < trace_cap) trace[k] = tr;
            break;
        }

        /* integrate */
        float dR[9], dT[3], RdT[3];
        cvo_oracle_exp_se3(omega, v, step, dR, dT);
        matvec3(s->R, dT, RdT);
        for (int q = 0; q < 3; ++q) s->T[q] = RdT[q] + s->T[q];
        matmul3(s->R, dR, s->R);

        const float dist = cvo_oracle_dist_se3(omega, v, step);
        tr.dist = dist;
        if (dist < p->eps_2) {
            s->iter = k;
            tr.exit_code = 2;
            if (trace && k < trace_cap) trace[k] = tr;
            break;
        }

        /* length-scale update */
        if (acvo) { /* ref acvo.cpp:538-545 */
            s->ell = (float)((double)s->ell + p->dl_step * dl);
            if (s->ell >= s->ell_max) {
                s->ell = (float)(s->ell_max * 0.7);
                s->ell_max = (float)(s->ell_max * 0.7);
            }
            s->ell = (s->ell < p->ell_min) ? p->ell_min : s->ell;
        } else { /* ref cvo.cpp:408-410 */
            s->ell = (k > 2) ? (float)0.10 : s->ell;
            s->ell = (k > 9) ? (float)0.06 : s->ell;
            s->ell = (k > 19) ? (float)0.03 : s->ell;
        }
        if (trace && k < trace_cap) trace[k] = tr;
    }
    /* ref cvo.cpp:413-415 */
    memcpy(s->prev_transform, s->transform, sizeof(s->transform));
    mat4_mul(s->accum_transform, s->transform, s->accum_transform);
    make_tf(s->R, s->T, Rt, t);
    tf_to_mat4(Rt, t, s->transform);
    free(y);
    return executed;
}

int cvo_oracle_align(const cvo_oracle_params *p, cvo_oracle_state *s, const float *x,
                     const float *fx, int n, const float *y0, const float *fy, int m,
                     int search, cvo_oracle_trace *trace, int trace_cap)
{
    return cvo_oracle_align_sharded(p, s, x, fx, n, y0, fy, m, search, 0, n, 0, m, NULL, NULL,
                                    trace, trace_cap);
}
