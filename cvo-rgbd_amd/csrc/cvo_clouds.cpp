// cvo_clouds.cpp -- the cloud hand-over: the tail of set_pcd() (ref src/cvo.cpp:344-356) for one cloud
// (cvo_hip_set_fixed / _set_moving[_device]) and for a batch of registration objects (cvo_hip_set_pcd_many).
#include "cvo_internal.h"

#if defined(__SSE2__)
#include <emmintrin.h>
#endif

using namespace cvo_dev;
using namespace cvo_impl;

namespace cvo_impl {

static inline void cpu_relax()
{
#if defined(__SSE2__)
    _mm_pause();
#endif
}

// The caller's array into the staging arena with streaming stores: the destination is written once and read by
// the DMA engine, never by this core -- no read-for-ownership of 41 MB of staging per batch, and the caller's
// arrays stay in the cache that held them.  dst is 16-byte aligned (the arena's pieces are 256-byte aligned).
void stage_copy(void *dst, const void *src, size_t bytes)
{
#if defined(__SSE2__)
    char *d = static_cast<char *>(dst);
    const char *s = static_cast<const char *>(src);
    size_t q = 0;
    for (; q + 64 <= bytes; q += 64) {
        const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i *>(s + q));
        const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i *>(s + q + 16));
        const __m128i c = _mm_loadu_si128(reinterpret_cast<const __m128i *>(s + q + 32));
        const __m128i e = _mm_loadu_si128(reinterpret_cast<const __m128i *>(s + q + 48));
        _mm_stream_si128(reinterpret_cast<__m128i *>(d + q), a);
        _mm_stream_si128(reinterpret_cast<__m128i *>(d + q + 16), b);
        _mm_stream_si128(reinterpret_cast<__m128i *>(d + q + 32), c);
        _mm_stream_si128(reinterpret_cast<__m128i *>(d + q + 48), e);
    }
    if (q < bytes) std::memcpy(d + q, s + q, bytes - q);
    _mm_sfence();
#else
    std::memcpy(dst, src, bytes);
#endif
}

// the hand-over of `c` has completed on the device; its bounding box is on the host
int cloud_ready(cvo_hip_ctx *ctx, Cloud &c)
{
    if (!c.pending) return CVO_HIP_OK;
    // (pending stays up if the wait fails: the box is still the zeros of upload_cloud, and every later entry
    // point must fail here again instead of building its filter geometry from them)
    HIP_TRY(ctx, hipEventSynchronize(c.wait_ev ? c.wait_ev : c.ready_ev));
    c.pending = false;
    for (int a = 0; a < 3; ++a) { c.lo[a] = c.bbox_pin[a]; c.hi[a] = c.bbox_pin[3 + a]; }
    return CVO_HIP_OK;
}

// The cloud into the kernels' layout (cvo_cloud.hip): Morton order -- consecutive device
// points are spatial neighbours, so a wave's 64 rows and a 16-column MFMA tile are compact
// patches and most (wave, tile) steps see no candidate -- packed rows, bounding spheres
// of the 64-point runs.  `on_device`: xyz / feat are device pointers (same device).
// What every hand-over of cloud `c` begins with: a hand-over of the same cloud that is still on its way ends,
// the arguments are checked, the device arrays hold n points (padded), the cloud's pinned words and event exist.
int cloud_reserve(cvo_hip_ctx *ctx, Cloud &c, const float *xyz, const float *feat, int n, int layout)
{
    const int np = cloud_padded(n);
    const Cloud &other = (&c == &ctx->fixed) ? ctx->moving : ctx->fixed;
    {   // (a hand-over of this cloud that is still on its way uses the staging and the arrays)
        const int rcw = cloud_ready(ctx, c);
        if (rcw) return rcw;
    }
    if (n < 0 || (n > 0 && (!xyz || !feat))) return fail(ctx, CVO_HIP_ERR_INVALID, "null cloud");
    if (n > (1 << 26))   // (the list kernels address a cloud through 32-bit byte offsets: 32 B per point)
        return fail(ctx, CVO_HIP_ERR_INVALID, "cloud too large: at most 2^26 points");
    if (layout != CVO_HIP_FEAT_COLMAJOR && layout != CVO_HIP_FEAT_ROWMAJOR)
        return fail(ctx, CVO_HIP_ERR_INVALID, "bad feat_layout");
    if (np > c.cap) {
        if (c.pos) HIP_TRY(ctx, hipFree(c.pos));
        if (c.feat) HIP_TRY(ctx, hipFree(c.feat));
        if (c.seg) HIP_TRY(ctx, hipFree(c.seg));
        c.pos = nullptr; c.feat = nullptr; c.seg = nullptr; c.cap = 0;
        HIP_TRY(ctx, hipMalloc((void **)&c.pos, (size_t)np * sizeof(float4)));
        HIP_TRY(ctx, hipMalloc((void **)&c.feat, (size_t)np * FEAT_STRIDE * sizeof(float)));
        HIP_TRY(ctx, hipMalloc((void **)&c.seg, (size_t)((np + SEG - 1) / SEG) * sizeof(float4)));
        c.cap = np;
    }
    c.n = n;
    c.np = np;
    c.pad_axis = (other.n > 0) ? 1 - other.pad_axis : 0;
    for (int a = 0; a < 3; ++a) { c.lo[a] = 0.0f; c.hi[a] = 0.0f; }
    if (n == 0) return CVO_HIP_OK;
    if (!ctx->bbox_host) {
        HIP_TRY(ctx, hipHostMalloc((void **)&ctx->bbox_host, 6 * sizeof(float), hipHostMallocDefault));
        HIP_TRY(ctx, hipMalloc((void **)&ctx->bbox_dev, 6 * sizeof(float)));
    }
    if (!c.bbox_pin) {
        HIP_TRY(ctx, hipHostMalloc((void **)&c.bbox_pin, 6 * sizeof(float), hipHostMallocDefault));
        void *bbox_d = nullptr;
        HIP_TRY(ctx, hipHostGetDevicePointer(&bbox_d, c.bbox_pin, 0));
        c.bbox_pin_dev = (float *)bbox_d;
        HIP_TRY(ctx, hipEventCreateWithFlags(&c.ready_ev, hipEventDisableTiming));
    }
    return CVO_HIP_OK;
}

int upload_cloud(cvo_hip_ctx *ctx, Cloud &c, const float *xyz, const float *feat, int n,
                 int layout, bool on_device)
{
    {
        const int rcr = cloud_reserve(ctx, c, xyz, feat, n, layout);
        if (rcr || n == 0) return rcr;
    }
    const int np = c.np;
    const size_t bytes_xyz = (size_t)n * 3 * sizeof(float), bytes_feat = (size_t)n * CVO_HIP_NFEAT * sizeof(float);
    const float *d_xyz = xyz, *d_feat = feat;
    if (!on_device) {
        // the arrays as they are, through pinned staging kept with the cloud
        const size_t off_feat = (bytes_xyz + 255) & ~(size_t)255;   // (stage_copy wants its destination 16-byte aligned)
        if (off_feat + bytes_feat > c.stage_bytes) {
            if (c.stage) (void)hipHostFree(c.stage);
            c.stage = nullptr;
            c.stage_bytes = 0;
            const size_t want = (off_feat + bytes_feat) * 5 / 4 + 4096;
            if (hipHostMalloc(&c.stage, want, hipHostMallocDefault) != hipSuccess)
                return fail(ctx, CVO_HIP_ERR_NOMEM, "hipHostMalloc(upload staging) failed");
            c.stage_bytes = want;
        }
        // (raw_xyz / raw_feat and the sort scratch are shared by the two clouds of a context: stream order
        // keeps one hand-over's kernels ahead of the next one's copies; growing them frees memory a queued
        // kernel may still read, so a growth waits for the stream first)
        if (bytes_xyz > ctx->raw_xyz.bytes || bytes_feat > ctx->raw_feat.bytes) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        int rcb = ensure_buf(ctx, ctx->raw_xyz, bytes_xyz);
        if (!rcb) rcb = ensure_buf(ctx, ctx->raw_feat, bytes_feat);
        if (rcb) return rcb;
        char *hs = reinterpret_cast<char *>(c.stage);
        stage_copy(hs, xyz, bytes_xyz);
        HIP_TRY(ctx, hipMemcpyAsync(ctx->raw_xyz.p, hs, bytes_xyz, hipMemcpyHostToDevice, ctx->stream));
        stage_copy(hs + off_feat, feat, bytes_feat);
        HIP_TRY(ctx, hipMemcpyAsync(ctx->raw_feat.p, hs + off_feat, bytes_feat, hipMemcpyHostToDevice, ctx->stream));
        d_xyz = (const float *)ctx->raw_xyz.p;
        d_feat = (const float *)ctx->raw_feat.p;
    }
    // From here on the cloud is in device memory either way.
    const bool no_one = ctx->opt.no_cloud_one;   // (test switch "one_launch_hand_over" = 0: the multi-launch preparation)
    if (n <= CLOUD_ONE_MAX && !no_one) {
        // ONE launch (k_cloud_one: a block does the whole preparation, the sort in LDS); the bounding box goes
        // straight into the cloud's pinned words
        CloudJob jb{};
        jb.xyz = d_xyz; jb.feat = d_feat; jb.n = n; jb.colmajor = layout == CVO_HIP_FEAT_COLMAJOR ? 1 : 0;
        jb.np = np; jb.pad_axis = c.pad_axis;
        jb.pos = c.pos; jb.feat8 = c.feat; jb.seg = c.seg;
        jb.bbox_out = c.bbox_pin_dev;
        HIP_TRY(ctx, cloud_prepare_one(jb, ctx->stream));
        HIP_TRY(ctx, hipEventRecord(c.ready_ev, ctx->stream));
        c.wait_ev = nullptr;
        c.pending = true;
        if (on_device || ctx->opt.sync_upload) return cloud_ready(ctx, c);
        return CVO_HIP_OK;
    }
    // Larger clouds: bounding box, keys, rocPRIM's radix sort, pack, spheres as launches of their own.  The box is
    // made on the device too and comes back to the host (the filter geometry of align() is made from it)
    // together with the end of the preparation: one synchronisation.
    HIP_TRY(ctx, cloud_bbox_device(d_xyz, n, ctx->bbox_dev, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(c.bbox_pin, ctx->bbox_dev, 6 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    int rc = CVO_HIP_OK;
    const size_t tmp = cloud_sort_scratch_bytes(n);
    if ((size_t)n * sizeof(uint32_t) > ctx->sort_keys[0].bytes || tmp > ctx->sort_tmp.bytes)
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // (see raw_xyz above)
    for (int q = 0; q < 2 && !rc; ++q) {
        rc = ensure_buf(ctx, ctx->sort_keys[q], (size_t)n * sizeof(uint32_t));
        if (!rc) rc = ensure_buf(ctx, ctx->sort_idx[q], (size_t)n * sizeof(int));
    }
    if (!rc) rc = ensure_buf(ctx, ctx->sort_tmp, tmp);
    if (rc) return rc;
    CloudPrep cp{};
    cp.np = np; cp.pad_axis = c.pad_axis;
    cp.xyz = d_xyz; cp.feat = d_feat; cp.n = n; cp.colmajor = layout == CVO_HIP_FEAT_COLMAJOR ? 1 : 0;
    cp.bbox = ctx->bbox_dev;
    for (int q = 0; q < 2; ++q) { cp.keys[q] = (uint32_t *)ctx->sort_keys[q].p; cp.idx[q] = (int *)ctx->sort_idx[q].p; }
    cp.scratch = ctx->sort_tmp.p; cp.scratch_bytes = tmp;
    cp.pos = c.pos; cp.feat8 = c.feat; cp.seg = c.seg;
    HIP_TRY(ctx, cloud_prepare_device(cp, ctx->stream));
    HIP_TRY(ctx, hipEventRecord(c.ready_ev, ctx->stream));
    c.wait_ev = nullptr;
    c.pending = true;
    // Host arrays were copied into the cloud's staging: the caller's are free at once, and the call does not
    // wait for the device (64 x 2 hand-overs of a batch overlap each other instead of costing 0.1 ms of host
    // time apiece).  Device arrays of the caller's are read by the queued kernels: they may be re-used
    // once this returns, so that form waits here.
    if (on_device || ctx->opt.sync_upload) return cloud_ready(ctx, c);
    return CVO_HIP_OK;
}

// ---------------------------------------------------------------------------
// The hand-over of a batch (cvo_hip_set_pcd_many): per device one stream, one pinned staging arena, one device
// arena for the caller's arrays as they come, a table of CloudJobs -- kept for the life of the process, like the
// engines.  One transfer per batch (or one per array where the caller's memory is page-locked already -- not implemented: the staged copy measured faster, profiles/r04_ab.txt 7), one
// launch of k_cloud_one for all clouds of up to CLOUD_ONE_MAX points, one event the clouds of the batch wait
// for.  Larger clouds take upload_cloud's way.
struct Handover {
    std::mutex mu;
    hipStream_t s = nullptr;
    char *stage = nullptr;                               // pinned, the size of ...
    char *raw = nullptr;     size_t raw_bytes = 0;       // ... the device arena
    CloudJob *jobs_pin = nullptr, *jobs_dev = nullptr;   int jobs_cap = 0;
    hipEvent_t ev[16] = {};
    int next_ev = 0;
    hipEvent_t last = nullptr;                           // the event of the batch that used the arenas last
};
Handover *handover_of(int device)
{
    static Handover *h = new Handover[64];   // (never destroyed: see cvo_lock.h)
    return (device >= 0 && device < 64) ? &h[device] : nullptr;
}


}   // namespace cvo_impl

extern "C" {

int cvo_hip_set_fixed(cvo_hip_ctx *ctx, const float *xyz, const float *feat, int n, int layout)
{
    cvo_lock::Api api_guard;
    if (!ctx) return CVO_HIP_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return upload_cloud(ctx, ctx->fixed, xyz, feat, n, layout);
}

int cvo_hip_set_moving(cvo_hip_ctx *ctx, const float *xyz, const float *feat, int m, int layout)
{
    cvo_lock::Api api_guard;
    if (!ctx) return CVO_HIP_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->have_tf = false;
    return upload_cloud(ctx, ctx->moving, xyz, feat, m, layout);
}

int cvo_hip_set_fixed_device(cvo_hip_ctx *ctx, const float *d_xyz, const float *d_feat, int n, int layout)
{
    cvo_lock::Api api_guard;
    if (!ctx) return CVO_HIP_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return upload_cloud(ctx, ctx->fixed, d_xyz, d_feat, n, layout, true);
}

int cvo_hip_set_moving_device(cvo_hip_ctx *ctx, const float *d_xyz, const float *d_feat, int m, int layout)
{
    cvo_lock::Api api_guard;
    if (!ctx) return CVO_HIP_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->have_tf = false;
    return upload_cloud(ctx, ctx->moving, d_xyz, d_feat, m, layout, true);
}

int cvo_hip_swap_moving_to_fixed(cvo_hip_ctx *ctx)
{
    cvo_lock::Api api_guard;
    if (!ctx) return CVO_HIP_ERR_INVALID;
    std::swap(ctx->fixed, ctx->moving);
    ctx->moving.n = 0;
    ctx->moving.np = 0;
    ctx->have_tf = false;
    return CVO_HIP_OK;
}

int cvo_hip_set_pcd_many(cvo_hip_ctx *const *ctxs, const float *const *fixed_xyz, const float *const *fixed_feat,
                         const int *n_fixed, const float *const *moving_xyz, const float *const *moving_feat,
                         const int *n_moving, int feat_layout, int count)
{
    cvo_lock::Api api_guard;
    if (count < 0 || (count > 0 && (!ctxs || !moving_xyz || !moving_feat || !n_moving))) return CVO_HIP_ERR_INVALID;
    if (count == 0) return CVO_HIP_OK;
    if (fixed_xyz && (!fixed_feat || !n_fixed)) return CVO_HIP_ERR_INVALID;
    for (int k = 0; k < count; ++k) {
        if (!ctxs[k] || ctxs[k]->device != ctxs[0]->device) return CVO_HIP_ERR_INVALID;
        for (int q = 0; q < k; ++q)
            if (ctxs[q] == ctxs[k]) return CVO_HIP_ERR_INVALID;   // (two clouds of a batch would land in the same device arrays)
    }
    cvo_hip_ctx *c0 = ctxs[0];
    // (every argument of the whole batch before any context is touched: a bad cloud in the middle must not leave the
    // earlier contexts with new counts over old rows)
    if (feat_layout != CVO_HIP_FEAT_COLMAJOR && feat_layout != CVO_HIP_FEAT_ROWMAJOR) return fail(c0, CVO_HIP_ERR_INVALID, "unknown feature layout");
    for (int k = 0; k < count; ++k) {
        if (n_moving[k] < 0 || (n_moving[k] > 0 && (!moving_xyz[k] || !moving_feat[k])))
            return fail(c0, CVO_HIP_ERR_INVALID, "batched hand-over: a moving cloud without its arrays (or a negative count)");
        if (fixed_xyz && fixed_xyz[k] && (n_fixed[k] < 0 || (n_fixed[k] > 0 && !fixed_feat[k])))
            return fail(c0, CVO_HIP_ERR_INVALID, "batched hand-over: a fixed cloud without its features (or a negative count)");
    }
    HIP_TRY(c0, hipSetDevice(c0->device));
    Handover *ho = handover_of(c0->device);
    if (!ho) return fail(c0, CVO_HIP_ERR_INVALID, "device index out of range");
    std::lock_guard<std::mutex> lock(ho->mu);
    if (!ho->s) {
        HIP_TRY(c0, hipStreamCreateWithFlags(&ho->s, hipStreamNonBlocking));
        for (auto &e : ho->ev) HIP_TRY(c0, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    // what goes through the one launch, what goes the long way
    struct Item { cvo_hip_ctx *ctx; Cloud *c; const float *xyz, *feat; int n; size_t off_xyz, off_feat; };
    std::vector<Item> small;
    size_t raw_need = 0;
    // A failure behind this point (out of memory on a grow, a transfer or a launch that does not go out) leaves clouds that were
    // reserved -- new counts, zeroed box -- but never filled: they are emptied, so that the next compute entry point of their
    // context fails on an empty cloud instead of registering stale rows.  (The state of the batch's clouds after an error is
    // otherwise undefined: hand them over again.)
    size_t filled = 0;   // clouds of `small` whose piece has gone out
    auto void_unfilled = [&]() {
        for (size_t q = filled; q < small.size(); ++q) { small[q].c->n = 0; small[q].c->np = 0; small[q].c->pending = false; small[q].c->wait_ev = nullptr; }
    };
    for (int k = 0; k < count; ++k) {
        cvo_hip_ctx *ctx = ctxs[k];
        for (int which = 0; which < 2; ++which) {
            const float *xyz = which == 0 ? (fixed_xyz ? fixed_xyz[k] : nullptr) : moving_xyz[k];
            const float *feat = which == 0 ? (fixed_xyz ? fixed_feat[k] : nullptr) : moving_feat[k];
            if (which == 0 && !xyz) continue;   // (the fixed cloud stays what it is)
            const int n = which == 0 ? n_fixed[k] : n_moving[k];
            Cloud &c = which == 0 ? ctx->fixed : ctx->moving;
            if (which == 1) ctx->have_tf = false;
            if (n > CLOUD_ONE_MAX || n <= 0) {
                const int rc = upload_cloud(ctx, c, xyz, feat, n, feat_layout);
                if (rc) { void_unfilled(); return rc; }
                continue;
            }
            const int rc = cloud_reserve(ctx, c, xyz, feat, n, feat_layout);
            if (rc) { void_unfilled(); return rc; }
            Item it{ctx, &c, xyz, feat, n, 0, 0};
            const size_t bx = ((size_t)n * 12 + 255) & ~(size_t)255, bf = ((size_t)n * 20 + 255) & ~(size_t)255;
            it.off_xyz = raw_need; it.off_feat = raw_need + bx;   // (the staging arena mirrors the device arena)
            raw_need += bx + bf;
            small.push_back(it);
        }
    }
    if (small.empty()) return CVO_HIP_OK;
    // the arenas are the previous batch's until its last event has completed
    if (ho->last && hipEventSynchronize(ho->last) != hipSuccess) { void_unfilled(); return fail(c0, CVO_HIP_ERR_HIP, "batched hand-over: the previous batch's event failed"); }
    if (raw_need > ho->raw_bytes) {
        if (ho->raw) (void)hipFree(ho->raw);
        if (ho->stage) (void)hipHostFree(ho->stage);
        ho->raw = nullptr; ho->stage = nullptr; ho->raw_bytes = 0;
        const size_t want = raw_need + raw_need / 4;
        if (hipMalloc((void **)&ho->raw, want) != hipSuccess || hipHostMalloc((void **)&ho->stage, want, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            void_unfilled();
            return fail(c0, CVO_HIP_ERR_NOMEM, "hand-over arena allocation failed");
        }
        ho->raw_bytes = want;
    }
    if ((int)small.size() > ho->jobs_cap) {
        if (ho->jobs_pin) (void)hipHostFree(ho->jobs_pin);
        if (ho->jobs_dev) (void)hipFree(ho->jobs_dev);
        ho->jobs_pin = nullptr; ho->jobs_dev = nullptr; ho->jobs_cap = 0;
        const int want = (int)small.size() * 2;
        if (hipHostMalloc((void **)&ho->jobs_pin, (size_t)want * sizeof(CloudJob), hipHostMallocDefault) != hipSuccess ||
            hipMalloc((void **)&ho->jobs_dev, (size_t)want * sizeof(CloudJob)) != hipSuccess) {
            (void)hipGetLastError();
            void_unfilled();
            return fail(c0, CVO_HIP_ERR_NOMEM, "hand-over job table allocation failed");
        }
        ho->jobs_cap = want;
    }
    for (size_t q = 0; q < small.size(); ++q) {
        const Item &it = small[q];
        CloudJob jb{};
        jb.xyz = (const float *)(ho->raw + it.off_xyz); jb.feat = (const float *)(ho->raw + it.off_feat);
        jb.n = it.n; jb.colmajor = feat_layout == CVO_HIP_FEAT_COLMAJOR ? 1 : 0;
        jb.np = it.c->np; jb.pad_axis = it.c->pad_axis;
        jb.pos = it.c->pos; jb.feat8 = it.c->feat; jb.seg = it.c->seg;
        jb.bbox_out = it.c->bbox_pin_dev;
        ho->jobs_pin[q] = jb;
    }
    if (hipMemcpyAsync(ho->jobs_dev, ho->jobs_pin, small.size() * sizeof(CloudJob), hipMemcpyHostToDevice, ho->s) != hipSuccess) {
        (void)hipGetLastError();
        void_unfilled();
        return fail(c0, CVO_HIP_ERR_HIP, "batched hand-over: job table transfer failed");
    }
    // The batch goes out in a few pieces -- the caller's arrays into the staging arena (a few host threads, a share
    // of a piece's clouds each: one thread moves ~10 GB/s, 128 clouds of 10k points are 41 MB), one transfer, one
    // launch, one event per piece -- so that the transfer of a piece runs while the next one is staged, and the
    // registrations of the first contexts can begin while the last clouds are still on their way (a cloud waits
    // for the event of ITS piece, when the next compute entry point of its context needs it).
    constexpr int kPieces = 4;
    const size_t per_piece = std::max<size_t>(raw_need / kPieces + 1, (size_t)4 << 20);
    std::vector<size_t> piece_end;   // index past the last cloud of each piece
    {
        size_t start_off = 0;
        for (size_t q = 0; q < small.size(); ++q) {
            const size_t end_off = small[q].off_feat + (((size_t)small[q].n * 20 + 255) & ~(size_t)255);
            if (end_off - start_off >= per_piece || q + 1 == small.size()) { piece_end.push_back(q + 1); start_off = end_off; }
        }
    }
    static const int max_threads = [] {   // (half the host's cores, at most 8: staging is memory-bound well before that)
        const int hw = (int)std::thread::hardware_concurrency();
        return std::min(8, std::max(2, hw / 2));
    }();
    int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)max_threads, raw_need / ((size_t)2 << 20)));
    std::vector<std::atomic<int>> staged(piece_end.size());
    for (auto &a : staged) a.store(0);
    char *stage = ho->stage;
    auto work = [&](int t) {
        size_t lo = 0;
        for (size_t pc = 0; pc < piece_end.size(); ++pc) {
            for (size_t q = lo + (size_t)t; q < piece_end[pc]; q += (size_t)nt) {
                const Item &it = small[q];
                stage_copy(stage + it.off_xyz, it.xyz, (size_t)it.n * 12);
                stage_copy(stage + it.off_feat, it.feat, (size_t)it.n * 20);
            }
            staged[pc].fetch_add(1, std::memory_order_release);
            lo = piece_end[pc];
        }
    };
    std::vector<std::thread> pool;
    struct Join { std::vector<std::thread> &p; ~Join() { for (auto &th : p) if (th.joinable()) th.join(); } } join_guard{pool};
    pool.reserve((size_t)nt);
    try {
        for (int t = 1; t < nt; ++t) pool.emplace_back(work, t);
    } catch (const std::system_error &) {
        // (no thread to be had: the helpers that did start keep their shares -- they stride by the nt they were given --, this
        // thread takes the shares of the ones that did not)
    }
    const int started = (int)pool.size() + 1;
    size_t lo = 0;
    int rc_out = CVO_HIP_OK;
    for (size_t pc = 0; pc < piece_end.size() && rc_out == CVO_HIP_OK; ++pc) {
        // (this thread's share of the piece -- and of the helpers that could not be started --, then the others')
        for (int t = 0; t < nt; t = t == 0 ? started : t + 1)
            for (size_t q = lo + (size_t)t; q < piece_end[pc]; q += (size_t)nt) {
                const Item &it = small[q];
                stage_copy(stage + it.off_xyz, it.xyz, (size_t)it.n * 12);
                stage_copy(stage + it.off_feat, it.feat, (size_t)it.n * 20);
            }
        while (staged[pc].load(std::memory_order_acquire) < started - 1) cpu_relax();
        const size_t hi = piece_end[pc];
        const size_t b0 = small[lo].off_xyz, b1 = small[hi - 1].off_feat + (((size_t)small[hi - 1].n * 20 + 255) & ~(size_t)255);
        int nmax = 0;
        for (size_t q = lo; q < hi; ++q) nmax = std::max(nmax, small[q].n);
        hipEvent_t ev = ho->ev[ho->next_ev];
        ho->next_ev = (ho->next_ev + 1) % 16;
        if (hipMemcpyAsync(ho->raw + b0, ho->stage + b0, b1 - b0, hipMemcpyHostToDevice, ho->s) != hipSuccess ||
            cloud_prepare_many(ho->jobs_dev + lo, (int)(hi - lo), nmax, ho->s) != hipSuccess ||
            hipEventRecord(ev, ho->s) != hipSuccess) {
            (void)hipGetLastError();
            void_unfilled();
            rc_out = fail(c0, CVO_HIP_ERR_HIP, "batched hand-over: transfer or launch failed");
            break;
        }
        ho->last = ev;
        for (size_t q = lo; q < hi; ++q) { small[q].c->wait_ev = ev; small[q].c->pending = true; }
        lo = hi;
        filled = hi;
    }
    return rc_out;
}

int cvo_hip_get_device_cloud(cvo_hip_ctx *ctx, int which, float *pos4, float *feat8, float *seg4, int *rows, int *points)
{
    cvo_lock::Api api_guard;
    if (!ctx || (which != 0 && which != 1)) return CVO_HIP_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    Cloud &c = which == 0 ? ctx->fixed : ctx->moving;
    const int rc = cloud_ready(ctx, c);
    if (rc) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (rows) *rows = c.np;
    if (points) *points = c.n;
    if (c.np <= 0) return CVO_HIP_OK;
    if (pos4) HIP_TRY(ctx, hipMemcpy(pos4, c.pos, (size_t)c.np * sizeof(float4), hipMemcpyDeviceToHost));
    if (feat8) HIP_TRY(ctx, hipMemcpy(feat8, c.feat, (size_t)c.np * FEAT_STRIDE * sizeof(float), hipMemcpyDeviceToHost));
    if (seg4) HIP_TRY(ctx, hipMemcpy(seg4, c.seg, (size_t)((c.np + SEG - 1) / SEG) * sizeof(float4), hipMemcpyDeviceToHost));
    return CVO_HIP_OK;
}


}   // extern "C"
