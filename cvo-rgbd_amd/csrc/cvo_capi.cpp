// cvo_capi.cpp -- implementation of include/cvo_hip.h: context, device memory,
// kernel orchestration and the align() loop of cvo::cvo / acvo::acvo
// (ref src/cvo.cpp:361-420, src/adaptive_cvo.cpp:490-555) on one MI355X.
//
// There is no CPU fallback in this library: every entry point that computes
// needs a gfx950 device and fails with CVO_HIP_ERR_NODEVICE / _HIP otherwise.
#include "cvo_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "cvo_comm.h"
#include "cvo_device.h"
#include "se3_host.hpp"

using namespace cvo_dev;

namespace {

struct Cloud {
    float4 *pos = nullptr;   // original positions
    float *feat = nullptr;
    int n = 0;
    int cap = 0;
};

struct EventPair {
    hipEvent_t a, b;
    int kind;      // SweepMode
    double pairs;
};

}   // namespace

struct cvo_hip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    cvo_hip_params prm{};
    Cloud fixed, moving;
    float4 *moving_tf = nullptr;   // transformed moving cloud (cloud_y)
    int moving_tf_cap = 0;
    float *taylor = nullptr;
    int taylor_cap = 0;
    double *partials = nullptr;
    size_t partials_cap = 0;       // in doubles
    double *totals = nullptr;      // device [32]
    double *totals_host = nullptr; // pinned [32]
    bool have_tf = false;
    int row_lo = 0, row_hi = -1, srow_lo = 0, srow_hi = -1;   // -1 = whole cloud
    bool sharded = false;
    cvo_comm *comm = nullptr;
    cvo_hip_allreduce_fn user_allreduce = nullptr;
    void *user_allreduce_arg = nullptr;
    bool profiling = false;
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

int ensure_f4(cvo_hip_ctx *ctx, float4 **p, int *cap, int n)
{
    if (n <= *cap) return CVO_HIP_OK;
    if (*p) HIP_TRY(ctx, hipFree(*p));
    *p = nullptr;
    *cap = 0;
    HIP_TRY(ctx, hipMalloc((void **)p, (size_t)n * sizeof(float4)));
    *cap = n;
    return CVO_HIP_OK;
}

int upload_cloud(cvo_hip_ctx *ctx, Cloud &c, const float *xyz, const float *feat, int n,
                 int layout)
{
    if (n < 0 || (n > 0 && (!xyz || !feat))) return fail(ctx, CVO_HIP_ERR_INVALID, "null cloud");
    if (layout != CVO_HIP_FEAT_COLMAJOR && layout != CVO_HIP_FEAT_ROWMAJOR)
        return fail(ctx, CVO_HIP_ERR_INVALID, "bad feat_layout");
    if (n > c.cap) {
        if (c.pos) HIP_TRY(ctx, hipFree(c.pos));
        if (c.feat) HIP_TRY(ctx, hipFree(c.feat));
        c.pos = nullptr; c.feat = nullptr; c.cap = 0;
        HIP_TRY(ctx, hipMalloc((void **)&c.pos, (size_t)n * sizeof(float4)));
        HIP_TRY(ctx, hipMalloc((void **)&c.feat, (size_t)n * FEAT_STRIDE * sizeof(float)));
        c.cap = n;
    }
    c.n = n;
    if (n == 0) return CVO_HIP_OK;
    // pack on the host into the device layout, one copy each
    std::vector<float> hp((size_t)n * 4), hf((size_t)n * FEAT_STRIDE, 0.0f);
    for (int i = 0; i < n; ++i) {
        hp[4 * (size_t)i + 0] = xyz[3 * (size_t)i + 0];
        hp[4 * (size_t)i + 1] = xyz[3 * (size_t)i + 1];
        hp[4 * (size_t)i + 2] = xyz[3 * (size_t)i + 2];
        hp[4 * (size_t)i + 3] = 0.0f;
        for (int f = 0; f < CVO_HIP_NFEAT; ++f)
            hf[(size_t)i * FEAT_STRIDE + f] = (layout == CVO_HIP_FEAT_COLMAJOR)
                                                  ? feat[(size_t)f * n + i]
                                                  : feat[(size_t)i * CVO_HIP_NFEAT + f];
    }
    HIP_TRY(ctx, hipMemcpyAsync(c.pos, hp.data(), hp.size() * sizeof(float), hipMemcpyHostToDevice,
                                ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(c.feat, hf.data(), hf.size() * sizeof(float),
                                hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // host staging buffers die here
    return CVO_HIP_OK;
}

KernConsts make_kconsts(const cvo_hip_params &p, float ell, float c_sp)
{
    KernConsts k{};
    const float l = ell;
    const float s2 = p.sigma * p.sigma;
    const float cs2 = p.c_sigma * p.c_sigma;
    k.tau = cvo_host::d2_threshold(l, p.sp_thres, s2);
    k.tau_c = cvo_host::d2c_threshold(p.c_ell, c_sp, p.c_sigma);
    k.sp = p.sp_thres;
    k.inv_c = 1 / p.c;
    k.inv_d = 1 / p.d;
    const float ell_3 = l * l * l;
    k.inv_l3 = 1 / ell_3;
    const float temp_coef = (float)(1 / (2.0 * l * l));
    k.cb = (float)(-2.0 * temp_coef);
    k.cg = -temp_coef;
    k.cd = (float)(2.0 * temp_coef);
    k.s2_d = (double)s2;
    k.cs2_d = (double)cs2;
    k.ninv_2l2 = -1.0 / (2.0 * l * l);
    k.ninv_2cl2 = -1.0 / (2.0 * p.c_ell * p.c_ell);
    return k;
}

// Chunk length so that the grid has enough workgroups to fill 256 CUs a few
// times over while each block still amortises its prologue/epilogue.
int pick_jt(int nrows, int nb)
{
    const int tiles = std::max(1, (nrows + rows_per_tile() - 1) / rows_per_tile());
    const int want_blocks = 2048;
    int chunks = std::max(1, want_blocks / tiles);
    int jt = (nb + chunks - 1) / chunks;
    jt = std::max(jt, 64);
    jt = std::min(jt, 2048);
    jt = (jt + 3) & ~3;
    return jt;
}

struct SweepPlan {
    dim3 grid;
    int jt;
    int nblocks;
};

SweepPlan plan_sweep(int nrows, int nb)
{
    SweepPlan p{};
    p.jt = pick_jt(nrows, nb);
    const int chunks = std::max(1, (nb + p.jt - 1) / p.jt);
    const int tiles = std::max(1, (nrows + rows_per_tile() - 1) / rows_per_tile());
    p.grid = dim3(chunks, tiles);
    p.nblocks = chunks * tiles;
    return p;
}

int ensure_partials(cvo_hip_ctx *ctx, size_t doubles)
{
    if (doubles <= ctx->partials_cap) return CVO_HIP_OK;
    if (ctx->partials) HIP_TRY(ctx, hipFree(ctx->partials));
    ctx->partials = nullptr;
    ctx->partials_cap = 0;
    HIP_TRY(ctx, hipMalloc((void **)&ctx->partials, doubles * sizeof(double)));
    ctx->partials_cap = doubles;
    return CVO_HIP_OK;
}

void shard_ranges(const cvo_hip_ctx *ctx, int &rlo, int &rhi, int &slo, int &shi)
{
    rlo = ctx->sharded ? ctx->row_lo : 0;
    rhi = ctx->sharded ? std::min(ctx->row_hi, ctx->fixed.n) : ctx->fixed.n;
    slo = ctx->sharded ? ctx->srow_lo : 0;
    shi = ctx->sharded ? std::min(ctx->srow_hi, ctx->moving.n) : ctx->moving.n;
    rlo = std::min(rlo, rhi);
    slo = std::min(slo, shi);
}

// Launch one sweep + its finalize; totals land in ctx->totals[off .. off+nacc).
int run_sweep(cvo_hip_ctx *ctx, int mode, const float4 *pos_a, const float *feat_a, int row_lo,
              int row_hi, const float4 *pos_b, const float *feat_b, int nb, int first_counted,
              const KernConsts &kc, int nacc, int off)
{
    const int nrows = row_hi - row_lo;
    if (nrows <= 0 || nb <= 0) {
        HIP_TRY(ctx, hipMemsetAsync(ctx->totals + off, 0, nacc * sizeof(double), ctx->stream));
        return CVO_HIP_OK;
    }
    const SweepPlan pl = plan_sweep(nrows, nb);
    int rc = ensure_partials(ctx, (size_t)pl.nblocks * NACC_MAX);
    if (rc) return rc;
    SweepArgs a{};
    a.pos_a = pos_a; a.feat_a = feat_a;
    a.pos_b = pos_b; a.feat_b = feat_b;
    a.taylor = ctx->taylor;
    a.partials = ctx->partials;
    a.row_lo = row_lo; a.row_hi = row_hi;
    a.nb = nb; a.jt = pl.jt;
    a.first_counted = first_counted;
    a.kc = kc;
    EventPair ev{};
    if (ctx->profiling) {
        HIP_TRY(ctx, hipEventCreate(&ev.a));
        HIP_TRY(ctx, hipEventCreate(&ev.b));
        ev.kind = mode;
        ev.pairs = (double)nrows * (double)nb;
        HIP_TRY(ctx, hipEventRecord(ev.a, ctx->stream));
    }
    launch_sweep(mode, a, pl.grid, ctx->stream);
    if (ctx->profiling) {
        HIP_TRY(ctx, hipEventRecord(ev.b, ctx->stream));
        ctx->events.push_back(ev);
    }
    HIP_TRY(ctx, hipGetLastError());
    launch_finalize(ctx->partials, pl.nblocks, nacc, ctx->totals + off, ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    return CVO_HIP_OK;
}

int drain_events(cvo_hip_ctx *ctx)
{
    for (auto &ev : ctx->events) {
        float ms = 0.f;
        HIP_TRY(ctx, hipEventSynchronize(ev.b));
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ev.a, ev.b));
        if (ev.kind == SWEEP_FLOW) {
            ctx->prof.flow_ms += ms; ctx->prof.flow_launches++; ctx->prof.flow_pairs += ev.pairs;
        } else if (ev.kind == SWEEP_STEP) {
            ctx->prof.step_ms += ms; ctx->prof.step_launches++; ctx->prof.step_pairs += ev.pairs;
        } else {
            ctx->prof.self_ms += ms; ctx->prof.self_launches++; ctx->prof.self_pairs += ev.pairs;
        }
        hipEventDestroy(ev.a);
        hipEventDestroy(ev.b);
    }
    ctx->events.clear();
    return CVO_HIP_OK;
}

// all-reduce `count` doubles at ctx->totals+off over ranks (no-op single rank)
int reduce_over_ranks(cvo_hip_ctx *ctx, int off, int count)
{
    if (ctx->comm) {
        if (cvo_comm_allreduce(ctx->comm, ctx->totals + off, count, ctx->stream) != 0)
            return fail(ctx, CVO_HIP_ERR_COMM, cvo_comm_last_error(ctx->comm));
    } else if (ctx->user_allreduce) {
        if (ctx->user_allreduce(ctx->user_allreduce_arg, ctx->totals + off, count,
                                (void *)ctx->stream) != 0)
            return fail(ctx, CVO_HIP_ERR_COMM, "user all-reduce failed");
    }
    return CVO_HIP_OK;
}

int fetch_totals(cvo_hip_ctx *ctx, int off, int count, double *out)
{
    HIP_TRY(ctx, hipMemcpyAsync(ctx->totals_host + off, ctx->totals + off, count * sizeof(double),
                                hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    std::memcpy(out, ctx->totals_host + off, count * sizeof(double));
    return CVO_HIP_OK;
}

int flow_impl(cvo_hip_ctx *ctx, float ell, double out13[13])
{
    if (!ctx->have_tf) return fail(ctx, CVO_HIP_ERR_INVALID, "transform_pcd not called");
    const bool acvo = ctx->prm.mode == CVO_HIP_MODE_ACVO;
    const KernConsts kc =
        make_kconsts(ctx->prm, ell, acvo ? ctx->prm.c_sp_thres : ctx->prm.sp_thres);
    int rlo, rhi, slo, shi;
    shard_ranges(ctx, rlo, rhi, slo, shi);
    int rc = run_sweep(ctx, SWEEP_FLOW, ctx->fixed.pos, ctx->fixed.feat, rlo, rhi, ctx->moving_tf,
                       ctx->moving.feat, ctx->moving.n, 0, kc, NACC_FLOW, 0);
    if (rc) return rc;
    if (acvo) {
        // Axx rows of this shard vs all of x; Ayy rows of this shard vs all of y
        rc = run_sweep(ctx, SWEEP_SELF, ctx->fixed.pos, ctx->fixed.feat, rlo, rhi, ctx->fixed.pos,
                       ctx->fixed.feat, ctx->fixed.n, 0, kc, NACC_SELF, 9);
        if (rc) return rc;
        rc = run_sweep(ctx, SWEEP_SELF, ctx->moving_tf, ctx->moving.feat, slo, shi, ctx->moving_tf,
                       ctx->moving.feat, ctx->moving.n, ctx->fixed.n, kc, NACC_SELF, 11);
        if (rc) return rc;
    } else {
        HIP_TRY(ctx, hipMemsetAsync(ctx->totals + 9, 0, 4 * sizeof(double), ctx->stream));
    }
    rc = reduce_over_ranks(ctx, 0, 13);
    if (rc) return rc;
    return fetch_totals(ctx, 0, 13, out13);
}

int step_impl(cvo_hip_ctx *ctx, const float omega[3], const float v[3], float ell, double bcde[4])
{
    if (!ctx->have_tf) return fail(ctx, CVO_HIP_ERR_INVALID, "transform_pcd not called");
    const bool acvo = ctx->prm.mode == CVO_HIP_MODE_ACVO;
    const KernConsts kc =
        make_kconsts(ctx->prm, ell, acvo ? ctx->prm.c_sp_thres : ctx->prm.sp_thres);
    const int m = ctx->moving.n;
    if (m > ctx->taylor_cap) {
        if (ctx->taylor) HIP_TRY(ctx, hipFree(ctx->taylor));
        ctx->taylor = nullptr; ctx->taylor_cap = 0;
        HIP_TRY(ctx, hipMalloc((void **)&ctx->taylor, (size_t)m * TAYLOR_STRIDE * sizeof(float)));
        ctx->taylor_cap = m;
    }
    const cvo_host::XiConsts xc = cvo_host::make_xi_consts(omega, v);
    TaylorArgs ta{};
    ta.pos = ctx->moving_tf; ta.taylor = ctx->taylor; ta.n = m;
    std::memcpy(ta.omega, xc.omega, sizeof(ta.omega));
    std::memcpy(ta.v, xc.v, sizeof(ta.v));
    std::memcpy(ta.W2, xc.W2, sizeof(ta.W2));
    std::memcpy(ta.W3, xc.W3, sizeof(ta.W3));
    std::memcpy(ta.W4, xc.W4, sizeof(ta.W4));
    std::memcpy(ta.u2, xc.u2, sizeof(ta.u2));
    std::memcpy(ta.u3, xc.u3, sizeof(ta.u3));
    std::memcpy(ta.u4, xc.u4, sizeof(ta.u4));
    launch_taylor(ta, ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    int rlo, rhi, slo, shi;
    shard_ranges(ctx, rlo, rhi, slo, shi);
    int rc = run_sweep(ctx, SWEEP_STEP, ctx->fixed.pos, ctx->fixed.feat, rlo, rhi, ctx->moving_tf,
                       ctx->moving.feat, m, 0, kc, NACC_STEP, 16);
    if (rc) return rc;
    rc = reduce_over_ranks(ctx, 16, 4);
    if (rc) return rc;
    return fetch_totals(ctx, 16, 4, bcde);
}

int transform_impl(cvo_hip_ctx *ctx, const float R[9], const float T[3])
{
    const int m = ctx->moving.n;
    int rc = ensure_f4(ctx, &ctx->moving_tf, &ctx->moving_tf_cap, std::max(m, 1));
    if (rc) return rc;
    TransformArgs ta{};
    ta.src = ctx->moving.pos; ta.dst = ctx->moving_tf; ta.n = m;
    cvo_host::inverse_tf(R, T, ta.Rt, ta.t);
    launch_transform(ta, ctx->stream);
    HIP_TRY(ctx, hipGetLastError());
    ctx->have_tf = true;
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
    if (!p || (mode != CVO_HIP_MODE_CVO && mode != CVO_HIP_MODE_ACVO)) return CVO_HIP_ERR_INVALID;
    std::memset(p, 0, sizeof(*p));
    p->mode = mode;
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
    if (!p || !out) return CVO_HIP_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n)
        return CVO_HIP_ERR_NODEVICE;
    cvo_hip_ctx *ctx = new (std::nothrow) cvo_hip_ctx();
    if (!ctx) return CVO_HIP_ERR_NOMEM;
    ctx->device = device;
    ctx->prm = *p;
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
    if (hipMalloc((void **)&ctx->totals, 32 * sizeof(double)) != hipSuccess)
        return bail(CVO_HIP_ERR_NOMEM);
    if (hipHostMalloc((void **)&ctx->totals_host, 32 * sizeof(double), hipHostMallocDefault) !=
        hipSuccess)
        return bail(CVO_HIP_ERR_NOMEM);
    if (hipMemsetAsync(ctx->totals, 0, 32 * sizeof(double), ctx->stream) != hipSuccess)
        return bail(CVO_HIP_ERR_HIP);
    *out = ctx;
    return CVO_HIP_OK;
}

int cvo_hip_destroy(cvo_hip_ctx *ctx)
{
    if (!ctx) return CVO_HIP_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (auto &ev : ctx->events) { hipEventDestroy(ev.a); hipEventDestroy(ev.b); }
    if (ctx->comm) cvo_comm_destroy(ctx->comm);
    for (void *p : {(void *)ctx->fixed.pos, (void *)ctx->fixed.feat, (void *)ctx->moving.pos,
                    (void *)ctx->moving.feat, (void *)ctx->moving_tf, (void *)ctx->taylor,
                    (void *)ctx->partials, (void *)ctx->totals})
        if (p) (void)hipFree(p);
    if (ctx->totals_host) (void)hipHostFree(ctx->totals_host);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return CVO_HIP_OK;
}

int cvo_hip_set_params(cvo_hip_ctx *ctx, const cvo_hip_params *p)
{
    if (!ctx || !p) return CVO_HIP_ERR_INVALID;
    ctx->prm = *p;
    return CVO_HIP_OK;
}

int cvo_hip_set_fixed(cvo_hip_ctx *ctx, const float *xyz, const float *feat, int n, int layout)
{
    if (!ctx) return CVO_HIP_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return upload_cloud(ctx, ctx->fixed, xyz, feat, n, layout);
}

int cvo_hip_set_moving(cvo_hip_ctx *ctx, const float *xyz, const float *feat, int m, int layout)
{
    if (!ctx) return CVO_HIP_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->have_tf = false;
    return upload_cloud(ctx, ctx->moving, xyz, feat, m, layout);
}

int cvo_hip_swap_moving_to_fixed(cvo_hip_ctx *ctx)
{
    if (!ctx) return CVO_HIP_ERR_INVALID;
    std::swap(ctx->fixed, ctx->moving);
    ctx->moving.n = 0;
    ctx->have_tf = false;
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
    if (!ctx || !id_bytes_128 || world <= 0 || rank < 0 || rank >= world)
        return CVO_HIP_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->comm) { cvo_comm_destroy(ctx->comm); ctx->comm = nullptr; }
    ctx->comm = cvo_comm_create(id_bytes_128, rank, world);
    if (!ctx->comm) return fail(ctx, CVO_HIP_ERR_COMM, "ncclCommInitRank failed");
    return CVO_HIP_OK;
}

int cvo_hip_set_allreduce(cvo_hip_ctx *ctx, cvo_hip_allreduce_fn fn, void *user)
{
    if (!ctx) return CVO_HIP_ERR_INVALID;
    ctx->user_allreduce = fn;
    ctx->user_allreduce_arg = user;
    return CVO_HIP_OK;
}

int cvo_hip_transform_pcd(cvo_hip_ctx *ctx, const float R[9], const float T[3])
{
    if (!ctx || !R || !T) return CVO_HIP_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return transform_impl(ctx, R, T);
}

int cvo_hip_flow(cvo_hip_ctx *ctx, float ell, double out13[13])
{
    if (!ctx || !out13) return CVO_HIP_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = flow_impl(ctx, ell, out13);
    if (!rc && ctx->profiling) rc = drain_events(ctx);
    return rc;
}

int cvo_hip_step_coeffs(cvo_hip_ctx *ctx, const float omega[3], const float v[3], float ell,
                        double bcde[4])
{
    if (!ctx || !omega || !v || !bcde) return CVO_HIP_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = step_impl(ctx, omega, v, ell, bcde);
    if (!rc && ctx->profiling) rc = drain_events(ctx);
    return rc;
}

int cvo_hip_pick_step(const double bcde[4], float min_step, float *step)
{
    if (!bcde || !step) return CVO_HIP_ERR_INVALID;
    *step = cvo_host::pick_step(bcde, min_step);
    return CVO_HIP_OK;
}

int cvo_hip_exp_se3(const float omega[3], const float v[3], float dt, float dR[9], float dT[3])
{
    if (!omega || !v || !dR || !dT) return CVO_HIP_ERR_INVALID;
    cvo_host::exp_se3(omega, v, dt, dR, dT);
    return CVO_HIP_OK;
}

int cvo_hip_dist_se3(const float omega[3], const float v[3], float dt, float *dist)
{
    if (!omega || !v || !dist) return CVO_HIP_ERR_INVALID;
    *dist = cvo_host::dist_se3(omega, v, dt);
    return CVO_HIP_OK;
}

int cvo_hip_align(cvo_hip_ctx *ctx, cvo_hip_state *s, cvo_hip_trace *trace, int trace_cap,
                  int *n_iter)
{
    if (!ctx || !s) return CVO_HIP_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const cvo_hip_params &p = ctx->prm;
    const bool acvo = p.mode == CVO_HIP_MODE_ACVO;
    if (acvo) {   // tail of acvo::set_pcd (ref src/adaptive_cvo.cpp:476-478)
        s->ell = p.ell_init;
        s->ell_max = p.ell_max_init;
    }
    int executed = 0;
    float Rt[9], t[3];
    for (int k = 0; k < p.max_iter; ++k) {
        cvo_hip_trace tr{};
        tr.k = k;
        tr.ell = s->ell;
        tr.dist = std::nanf("");
        // update_tf(); transform_pcd();
        cvo_host::inverse_tf(s->R, s->T, Rt, t);
        cvo_host::tf_to_mat4(Rt, t, s->transform);
        int rc = transform_impl(ctx, s->R, s->T);
        if (rc) return rc;
        // compute_flow();
        double red[13];
        rc = flow_impl(ctx, s->ell, red);
        if (rc) return rc;
        float omega[3], v[3];
        for (int q = 0; q < 3; ++q) {
            tr.omega_d[q] = red[q];
            tr.v_d[q] = red[3 + q];
            tr.omega[q] = omega[q] = (float)red[q];
            tr.v[q] = v[q] = (float)red[3 + q];
        }
        tr.sum_a = red[6];
        tr.nnz = (int64_t)red[8];
        double dl = 0.0;
        if (acvo) {
            tr.nnz_xx = (int64_t)red[10];
            tr.nnz_yy = (int64_t)red[12];
            const double num = (red[11] - 2.0 * red[7]) + red[9];
            dl = num / (double)(tr.nnz_xx + tr.nnz_yy - 2 * tr.nnz);
            tr.dl = dl;
        }
        // compute_step_size();
        double bcde[4];
        rc = step_impl(ctx, omega, v, s->ell, bcde);
        if (rc) return rc;
        std::memcpy(tr.bcde, bcde, sizeof(bcde));
        const float step = cvo_host::pick_step(bcde, p.min_step);
        tr.step = step;
        executed = k + 1;

        bool brk;
        if (acvo) {   // omega.cast<double>().norm() (ref src/adaptive_cvo.cpp:509)
            const double nw = std::sqrt((double)omega[0] * omega[0] +
                                        ((double)omega[1] * omega[1] + (double)omega[2] * omega[2]));
            const double nv = std::sqrt((double)v[0] * v[0] +
                                        ((double)v[1] * v[1] + (double)v[2] * v[2]));
            brk = nw < (double)p.eps && nv < (double)p.eps;
        } else {
            brk = cvo_host::norm_fixed3(omega) < p.eps && cvo_host::norm_fixed3(v) < p.eps;
        }
        if (brk) {
            s->iter = k;
            tr.exit_code = 1;
            if (trace && k < trace_cap) trace[k] = tr;
            break;
        }
        float dR[9], dT[3], RdT[3];
        cvo_host::exp_se3(omega, v, step, dR, dT);
        cvo_host::Mat3 R{}, dRm{};
        std::memcpy(R.m, s->R, sizeof(R.m));
        std::memcpy(dRm.m, dR, sizeof(dRm.m));
        cvo_host::mul(R, dT, RdT);
        for (int q = 0; q < 3; ++q) s->T[q] = RdT[q] + s->T[q];   // T = R*dT + T
        const cvo_host::Mat3 Rn = cvo_host::mul(R, dRm);            // R = R*dR
        std::memcpy(s->R, Rn.m, sizeof(Rn.m));

        const float dist = cvo_host::dist_se3(omega, v, step);
        tr.dist = dist;
        if (dist < p.eps_2) {
            s->iter = k;
            tr.exit_code = 2;
            if (trace && k < trace_cap) trace[k] = tr;
            break;
        }
        if (acvo) {   // ref src/adaptive_cvo.cpp:538-545
            s->ell = (float)((double)s->ell + p.dl_step * dl);
            if (s->ell >= s->ell_max) {
                s->ell = (float)(s->ell_max * 0.7);
                s->ell_max = (float)(s->ell_max * 0.7);
            }
            s->ell = (s->ell < p.ell_min) ? p.ell_min : s->ell;
        } else {      // ref src/cvo.cpp:408-410
            s->ell = (k > 2) ? (float)0.10 : s->ell;
            s->ell = (k > 9) ? (float)0.06 : s->ell;
            s->ell = (k > 19) ? (float)0.03 : s->ell;
        }
        if (trace && k < trace_cap) trace[k] = tr;
    }
    // ref src/cvo.cpp:413-415
    std::memcpy(s->prev_transform, s->transform, sizeof(s->transform));
    cvo_host::mat4_mul(s->accum_transform, s->transform, s->accum_transform);
    cvo_host::inverse_tf(s->R, s->T, Rt, t);
    cvo_host::tf_to_mat4(Rt, t, s->transform);
    if (n_iter) *n_iter = executed;
    if (ctx->profiling) return drain_events(ctx);
    return CVO_HIP_OK;
}

int cvo_hip_function_inner_product(cvo_hip_ctx *ctx, float ell, float *out)
{
    if (!ctx || !out) return CVO_HIP_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // untransformed positions, colour cut with sp_thres (ref acvo.cpp:391-392)
    const KernConsts kc = make_kconsts(ctx->prm, ell, ctx->prm.sp_thres);
    int rlo, rhi, slo, shi;
    shard_ranges(ctx, rlo, rhi, slo, shi);
    int rc = run_sweep(ctx, SWEEP_FLOW, ctx->fixed.pos, ctx->fixed.feat, rlo, rhi, ctx->moving.pos,
                       ctx->moving.feat, ctx->moving.n, 0, kc, NACC_FLOW, 0);
    if (rc) return rc;
    rc = reduce_over_ranks(ctx, 0, 9);
    if (rc) return rc;
    double red[9];
    rc = fetch_totals(ctx, 0, 9, red);
    if (rc) return rc;
    *out = (float)(red[6] / red[8]);
    if (ctx->profiling) return drain_events(ctx);
    return CVO_HIP_OK;
}

int cvo_hip_set_profiling(cvo_hip_ctx *ctx, int enable)
{
    if (!ctx) return CVO_HIP_ERR_INVALID;
    ctx->profiling = enable != 0;
    return CVO_HIP_OK;
}

int cvo_hip_get_profile(cvo_hip_ctx *ctx, cvo_hip_profile *out, int reset)
{
    if (!ctx || !out) return CVO_HIP_ERR_INVALID;
    int rc = drain_events(ctx);
    if (rc) return rc;
    *out = ctx->prof;
    if (reset) ctx->prof = cvo_hip_profile{};
    return CVO_HIP_OK;
}

int cvo_hip_synchronize(cvo_hip_ctx *ctx)
{
    if (!ctx) return CVO_HIP_ERR_INVALID;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CVO_HIP_OK;
}

}   // extern "C"
