// cvo_prep.hip -- the cloud preparation of the reference's MATLAB driver on the device
// (SURVEY 8 f2): pcRangeFilter (ref util/pcRangeFilter.m:5-12) followed by
// pcdownsample(cloud, 'gridAverage', gridSize) (ref data/rgbd_dataset/rgbddataset_rkhs.m:36-39,58):
// one point per occupied voxel of a box grid anchored at the cloud's minimum corner = the mean
// location and the mean colour of its points, voxels in lexicographic (x, y, z) index order.
//
// Results are defined by the arithmetic, not by the device (oracle/matlab_prep.py is the same
// sequence in numpy): float32 range as ((x*x + y*y) + z*z) and a correctly rounded sqrt; voxel
// indices floor((x - min) / grid) in float64; a STABLE sort of (voxel key, point index), so
// that every voxel's float64 sums run over its points in their original order; one division
// by the count; colours rounded as floor(mean + 0.5).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <mutex>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "cvo_hip.h"
#include "cvo_lock.h"

namespace {

constexpr int PB = 256;
constexpr unsigned long long KEY_DROPPED = ~0ull;

// keep[i] (range filter) and the bounding box of the kept points: box[0..2] min, [3..5] max, one block
__global__ void __launch_bounds__(1024) k_prep_keep_box(const float *xyz, int n, float max_range, float min_range, int use_range,
                                                        unsigned char *keep, float *box, int *n_kept)
{
    __shared__ float s_lo[16][3], s_hi[16][3];
    __shared__ int s_cnt[16];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    int cnt = 0;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const float x = xyz[3 * (size_t)i], y = xyz[3 * (size_t)i + 1], z = xyz[3 * (size_t)i + 2];
        // (non-finite points never reach a voxel: MATLAB's pcdownsample drops invalid points, and a NaN / Inf
        // coordinate has no voxel index -- the cast of floor(NaN) is undefined and the box would become infinite)
        bool k = isfinite(x) && isfinite(y) && isfinite(z);
        if (k && use_range) {
            const float r = sqrtf((x * x + y * y) + z * z);   // (-ffp-contract=off: no FMA)
            k = !((r > max_range) || (r < min_range));
        }
        keep[i] = k ? 1 : 0;
        if (k) {
            ++cnt;
            const float v[3] = {x, y, z};
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                lo[a] = v[a] < lo[a] ? v[a] : lo[a];
                hi[a] = v[a] > hi[a] ? v[a] : hi[a];
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
        for (int off = 32; off > 0; off >>= 1) {
            const float l2 = __shfl_xor(lo[a], off), h2 = __shfl_xor(hi[a], off);
            lo[a] = l2 < lo[a] ? l2 : lo[a];
            hi[a] = h2 > hi[a] ? h2 : hi[a];
        }
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
    if ((threadIdx.x & 63) == 0) {
        for (int a = 0; a < 3; ++a) { s_lo[threadIdx.x >> 6][a] = lo[a]; s_hi[threadIdx.x >> 6][a] = hi[a]; }
        s_cnt[threadIdx.x >> 6] = cnt;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        float l = INFINITY, h = -INFINITY;
        for (int w = 0; w < 16; ++w) {
            l = s_lo[w][threadIdx.x] < l ? s_lo[w][threadIdx.x] : l;
            h = s_hi[w][threadIdx.x] > h ? s_hi[w][threadIdx.x] : h;
        }
        box[threadIdx.x] = l;
        box[3 + threadIdx.x] = h;
    }
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < 16; ++w) t += s_cnt[w];
        *n_kept = t;
    }
}

__device__ __forceinline__ long long voxel_index(float v, float lo, double grid)
{
    return (long long)floor(((double)v - (double)lo) / grid);
}

// key of every point: its voxel in lexicographic order, or KEY_DROPPED; grid <= 0: the point's own index
// (no downsampling: the kept points in their original order)
__global__ void __launch_bounds__(PB) k_prep_keys(const float *xyz, const unsigned char *keep, int n, const float *box, double grid,
                                                  unsigned long long *keys, int *idx)
{
    const int i = blockIdx.x * PB + threadIdx.x;
    if (i >= n) return;
    idx[i] = i;
    if (!keep[i]) { keys[i] = KEY_DROPPED; return; }
    if (!(grid > 0.0)) { keys[i] = (unsigned long long)i; return; }
    long long q[3], span[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        q[a] = voxel_index(xyz[3 * (size_t)i + a], box[a], grid);
        span[a] = voxel_index(box[3 + a], box[a], grid) + 1;   // (the index is monotone in the coordinate)
    }
    keys[i] = (unsigned long long)((q[0] * span[1] + q[1]) * span[2] + q[2]);
}

__global__ void __launch_bounds__(PB) k_prep_heads(const unsigned long long *keys, int n, int *head)
{
    const int i = blockIdx.x * PB + threadIdx.x;
    if (i >= n) return;
    const unsigned long long k = keys[i];
    head[i] = (k != KEY_DROPPED && (i == 0 || keys[i - 1] != k)) ? 1 : 0;
}

// one thread per voxel head: float64 sums over the voxel's points in their original order
__global__ void __launch_bounds__(PB) k_prep_average(const float *xyz, const unsigned char *rgb, const unsigned long long *keys,
                                                     const int *order, const int *head, const int *seg, int n,
                                                     float *xyz_out, unsigned char *rgb_out)
{
    const int i = blockIdx.x * PB + threadIdx.x;
    if (i >= n || !head[i]) return;
    const unsigned long long k = keys[i];
    double sx = 0.0, sy = 0.0, sz = 0.0, sr = 0.0, sg = 0.0, sb = 0.0, cnt = 0.0;
    for (int j = i; j < n && keys[j] == k; ++j) {
        const int p = order[j];
        sx += (double)xyz[3 * (size_t)p]; sy += (double)xyz[3 * (size_t)p + 1]; sz += (double)xyz[3 * (size_t)p + 2];
        sr += (double)rgb[3 * (size_t)p]; sg += (double)rgb[3 * (size_t)p + 1]; sb += (double)rgb[3 * (size_t)p + 2];
        cnt += 1.0;
    }
    const int o = seg[i] - 1;   // (inclusive scan of the heads)
    xyz_out[3 * (size_t)o] = (float)(sx / cnt);
    xyz_out[3 * (size_t)o + 1] = (float)(sy / cnt);
    xyz_out[3 * (size_t)o + 2] = (float)(sz / cnt);
    const double c[3] = {sr / cnt, sg / cnt, sb / cnt};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        double v = floor(c[a] + 0.5);
        v = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v);
        rgb_out[3 * (size_t)o + a] = (unsigned char)v;
    }
}

// One arena and one stream per device, kept for the life of the process (like the engines' streams): a
// call costs its copies and seven launches, not a dozen allocations.
struct Arena {
    void *p = nullptr;
    size_t bytes = 0;
    hipStream_t s = nullptr;
};
std::mutex *arena_mutex()
{
    static std::mutex *m = new std::mutex;   // (never destroyed: see cvo_lock.h)
    return m;
}
Arena *arena_of(int device)
{
    static Arena *a = new Arena[64];
    return (device >= 0 && device < 64) ? &a[device] : nullptr;
}

}   // namespace

extern "C" int cvo_hip_range_filter_grid_average(int device, const float *xyz, const unsigned char *rgb, int n,
                                                 float max_range, float min_range, double grid_size,
                                                 float *xyz_out, unsigned char *rgb_out, int *n_out)
{
    cvo_lock::Api api_guard;
    if (n < 0 || !n_out || (n > 0 && (!xyz || !rgb || !xyz_out || !rgb_out))) return CVO_HIP_ERR_INVALID;
    *n_out = 0;
    if (n == 0) return CVO_HIP_OK;
    if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return CVO_HIP_ERR_NODEVICE; }
    Arena *ar = arena_of(device);
    if (!ar) return CVO_HIP_ERR_INVALID;
    std::lock_guard<std::mutex> lock(*arena_mutex());
#define PREP_TRY(e) do { if ((e) != hipSuccess) { (void)hipGetLastError(); return CVO_HIP_ERR_HIP; } } while (0)
    if (!ar->s) PREP_TRY(hipStreamCreateWithFlags(&ar->s, hipStreamNonBlocking));
    hipStream_t st = ar->s;
    const size_t N = (size_t)n;
    size_t sort_bytes = 0, scan_bytes = 0;
    {
        unsigned long long *k = nullptr;
        int *v = nullptr;
        PREP_TRY(rocprim::radix_sort_pairs(nullptr, sort_bytes, k, k, v, v, N, 0u, 64u, st));
        PREP_TRY(rocprim::inclusive_scan(nullptr, scan_bytes, v, v, N, rocprim::plus<int>(), st));
    }
    // the arena, carved (256-byte pieces)
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t at = off; off += (bytes + 255) / 256 * 256; return at; };
    const size_t o_xyz = take(N * 12), o_rgb = take(N * 3), o_keep = take(N), o_box = take(64), o_k0 = take(N * 8),
                 o_k1 = take(N * 8), o_i0 = take(N * 4), o_i1 = take(N * 4), o_head = take(N * 4), o_seg = take(N * 4),
                 o_out = take(N * 12), o_cout = take(N * 3), o_tmp = take(sort_bytes > scan_bytes ? sort_bytes : scan_bytes);
    if (off > ar->bytes) {
        if (ar->p) { (void)hipStreamSynchronize(st); (void)hipFree(ar->p); ar->p = nullptr; ar->bytes = 0; }
        const size_t want = off + off / 4;
        if (hipMalloc(&ar->p, want) != hipSuccess) { (void)hipGetLastError(); ar->p = nullptr; return CVO_HIP_ERR_NOMEM; }
        ar->bytes = want;
    }
    char *base = (char *)ar->p;
    float *d_xyz = (float *)(base + o_xyz), *d_box = (float *)(base + o_box), *d_out = (float *)(base + o_out);
    int *d_cnt = (int *)(d_box + 8);
    unsigned char *d_rgb = (unsigned char *)(base + o_rgb), *d_keep = (unsigned char *)(base + o_keep),
                  *d_rgb_out = (unsigned char *)(base + o_cout);
    unsigned long long *d_keys[2] = {(unsigned long long *)(base + o_k0), (unsigned long long *)(base + o_k1)};
    int *d_idx[2] = {(int *)(base + o_i0), (int *)(base + o_i1)}, *d_head = (int *)(base + o_head), *d_seg = (int *)(base + o_seg);
    void *d_tmp = base + o_tmp;
    PREP_TRY(hipMemcpyAsync(d_xyz, xyz, N * 12, hipMemcpyHostToDevice, st));
    PREP_TRY(hipMemcpyAsync(d_rgb, rgb, N * 3, hipMemcpyHostToDevice, st));
    const int use_range = max_range > 0.0f ? 1 : 0;
    hipLaunchKernelGGL(k_prep_keep_box, dim3(1), dim3(1024), 0, st, d_xyz, n, max_range, min_range, use_range, d_keep, d_box, d_cnt);
    const int nb = (n + PB - 1) / PB;
    hipLaunchKernelGGL(k_prep_keys, dim3(nb), dim3(PB), 0, st, d_xyz, d_keep, n, d_box, grid_size, d_keys[0], d_idx[0]);
    // stable: points of one voxel stay in their original order
    PREP_TRY(rocprim::radix_sort_pairs(d_tmp, sort_bytes, d_keys[0], d_keys[1], d_idx[0], d_idx[1], N, 0u, 64u, st));
    hipLaunchKernelGGL(k_prep_heads, dim3(nb), dim3(PB), 0, st, d_keys[1], n, d_head);
    PREP_TRY(rocprim::inclusive_scan(d_tmp, scan_bytes, d_head, d_seg, N, rocprim::plus<int>(), st));
    hipLaunchKernelGGL(k_prep_average, dim3(nb), dim3(PB), 0, st, d_xyz, d_rgb, d_keys[1], d_idx[1], d_head, d_seg, n, d_out, d_rgb_out);
    PREP_TRY(hipGetLastError());
    int n_seg = 0;
    float box[6] = {0, 0, 0, 0, 0, 0};
    PREP_TRY(hipMemcpyAsync(&n_seg, d_seg + (N - 1), sizeof(int), hipMemcpyDeviceToHost, st));
    PREP_TRY(hipMemcpyAsync(box, d_box, sizeof(box), hipMemcpyDeviceToHost, st));
    PREP_TRY(hipStreamSynchronize(st));
    if (grid_size > 0.0 && n_seg > 0) {
        // the lexicographic key (q0 * span1 + q1) * span2 + q2 must fit below KEY_DROPPED: a grid so fine
        // that the box holds 2^63 voxels or more is refused (the keys computed above were garbage)
        long double prod = 1.0L;
        for (int a = 0; a < 3; ++a) prod *= std::floor(((long double)box[3 + a] - (long double)box[a]) / (long double)grid_size) + 1.0L;
        if (!(prod < 9.0e18L)) return CVO_HIP_ERR_INVALID;
    }
    if (n_seg > 0) {
        PREP_TRY(hipMemcpyAsync(xyz_out, d_out, (size_t)n_seg * 12, hipMemcpyDeviceToHost, st));
        PREP_TRY(hipMemcpyAsync(rgb_out, d_rgb_out, (size_t)n_seg * 3, hipMemcpyDeviceToHost, st));
        PREP_TRY(hipStreamSynchronize(st));
    }
#undef PREP_TRY
    *n_out = n_seg;
    return CVO_HIP_OK;
}
