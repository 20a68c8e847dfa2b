// cvo_cloud.hip -- a cloud as the caller hands it over (xyz n x 3, features n x 5)
// into the layout the kernels read, on the device: Morton keys, a stable radix sort,
// the packed position / feature rows and the bounding spheres of the 64-point runs.
// (ref: the tail of cvo::set_pcd, src/cvo.cpp:344-356, only copies the clouds; the
// ordering is this back end's data layout)
//
// Results are defined by the arithmetic below, not by the device: every step is either
// exact (min / max, integer keys, a stable sort) or a fixed float sequence; a host
// implementation of the same steps (the first version of upload_cloud) gave the same
// bytes.
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "cvo_cloud.h"

namespace cvo_dev {

namespace {

constexpr int CB = 256;

__device__ __forceinline__ uint32_t spread10(uint32_t v)
{   // 10 bits -> every third bit
    v &= 1023u;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

// bounding box of n points: bbox[0..2] = min, [3..5] = max (one block)
__global__ void __launch_bounds__(1024) k_cloud_bbox(const float *xyz, int n, float *bbox)
{
    __shared__ float s_lo[16][3], s_hi[16][3];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = threadIdx.x; i < n; i += 1024)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = xyz[3 * (size_t)i + a];
            lo[a] = v < lo[a] ? v : lo[a];   // (NaN never replaces: as the host loop)
            hi[a] = v > hi[a] ? v : hi[a];
        }
#pragma unroll
    for (int a = 0; a < 3; ++a)
        for (int off = 32; off > 0; off >>= 1) {
            const float l2 = __shfl_xor(lo[a], off), h2 = __shfl_xor(hi[a], off);
            lo[a] = l2 < lo[a] ? l2 : lo[a];
            hi[a] = h2 > hi[a] ? h2 : hi[a];
        }
    if ((threadIdx.x & 63) == 0)
        for (int a = 0; a < 3; ++a) { s_lo[threadIdx.x >> 6][a] = lo[a]; s_hi[threadIdx.x >> 6][a] = hi[a]; }
    __syncthreads();
    if (threadIdx.x < 3) {
        float l = INFINITY, h = -INFINITY;
        for (int w = 0; w < 16; ++w) {
            l = s_lo[w][threadIdx.x] < l ? s_lo[w][threadIdx.x] : l;
            h = s_hi[w][threadIdx.x] > h ? s_hi[w][threadIdx.x] : h;
        }
        bbox[threadIdx.x] = l;
        bbox[3 + threadIdx.x] = h;
    }
}


// 30-bit Morton key of every point (10 bits per axis of the bounding box) and the
// identity permutation
__global__ void __launch_bounds__(CB) k_cloud_keys(const float *xyz, int n, const float *bbox, uint32_t *keys, int *idx)
{
    const int i = blockIdx.x * CB + threadIdx.x;
    if (i >= n) return;
    uint32_t q[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        // (the box is read from device memory: a cloud that arrives in device memory is prepared
        // without the host having to see its box first)
        const float lo = bbox[a], ext = bbox[3 + a] - lo;
        const float inv = (ext > 0.0f && ext <= 3.4e38f) ? 1023.0f / ext : 0.0f;   // (finite extent)
        float f = (xyz[3 * (size_t)i + a] - lo) * inv;
        if (!(f >= 0.0f)) f = 0.0f;   // also catches NaN
        if (f > 1023.0f) f = 1023.0f;
        q[a] = (uint32_t)f;
    }
    keys[i] = spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
    idx[i] = i;
}

// device rows in sorted order: pos = (x, y, z, f4), feat = f0..f4, caller's index, 0, 0
__global__ void __launch_bounds__(CB) k_cloud_pack(const float *xyz, const float *feat, int n, int colmajor,
                                                   const int *order, float4 *pos, float *feat8)
{
    const int s = blockIdx.x * CB + threadIdx.x;
    if (s >= n) return;
    const int i = order[s];
    float f[CVO_HIP_NFEAT];
#pragma unroll
    for (int q = 0; q < CVO_HIP_NFEAT; ++q)
        f[q] = colmajor ? feat[(size_t)q * n + i] : feat[(size_t)i * CVO_HIP_NFEAT + q];
    // the 5th feature rides in pos.w: a pair then costs four 16-byte gathers (two
    // positions, two feature quads) and the caller's index (acvo Ayy rule only) moves to feat[5]
    pos[s] = make_float4(xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], f[4]);
    float4 *o = reinterpret_cast<float4 *>(feat8 + (size_t)s * FEAT_STRIDE);
    o[0] = make_float4(f[0], f[1], f[2], f[3]);
    o[1] = make_float4(f[4], __int_as_float(i), 0.0f, 0.0f);
}

// bounding spheres of the Morton runs (culling in k_filter): centre of the run's bounding
// box, radius = farthest point, inflated against rounding.  One wave per run of SEG points.
__global__ void __launch_bounds__(CB) k_cloud_seg(const float4 *pos, int n, int nseg, float4 *seg)
{
    const int g = blockIdx.x * (CB / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (g >= nseg) return;
    const int q = g * SEG + lane;
    const bool live = q < n;
    const float4 p = pos[live ? q : n - 1];
    float lo[3] = {live ? p.x : INFINITY, live ? p.y : INFINITY, live ? p.z : INFINITY};
    float hi[3] = {live ? p.x : -INFINITY, live ? p.y : -INFINITY, live ? p.z : -INFINITY};
#pragma unroll
    for (int a = 0; a < 3; ++a)
        for (int off = 32; off > 0; off >>= 1) {
            const float l2 = __shfl_xor(lo[a], off), h2 = __shfl_xor(hi[a], off);
            lo[a] = l2 < lo[a] ? l2 : lo[a];
            hi[a] = h2 > hi[a] ? h2 : hi[a];
        }
    float c[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) c[a] = (float)(0.5 * ((double)lo[a] + hi[a]));
    double d2 = 0.0;
    if (live) {
        const double dx = (double)p.x - (double)c[0], dy = (double)p.y - (double)c[1], dz = (double)p.z - (double)c[2];
        d2 = (dx * dx + dy * dy) + dz * dz;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(d2, off);
        d2 = o > d2 ? o : d2;
    }
    if (lane == 0) seg[g] = make_float4(c[0], c[1], c[2], (float)(sqrt(d2) * 1.00001 + 1e-6));
}

// padding rows [n, np): parked far away along `axis`, 16 m apart; NaN features
__global__ void __launch_bounds__(CB) k_cloud_pad(float4 *pos, float *feat8, float4 *seg, int n, int np,
                                                  const float *bbox, int axis)
{
    const int q = n + blockIdx.x * CB + threadIdx.x;
    if (q >= np) return;
    const float bx = 0.5f * (bbox[0] + bbox[3]), by = 0.5f * (bbox[1] + bbox[4]), bz = 0.5f * (bbox[2] + bbox[5]);
    const float nanv = __int_as_float(0x7fc00000);
    const float off = 1.0e4f + 16.0f * (float)(q - n);
    const float4 p = axis ? make_float4(bx, by + off, bz, nanv) : make_float4(bx + off, by, bz, nanv);
    pos[q] = p;
    float4 *o = reinterpret_cast<float4 *>(feat8 + (size_t)q * FEAT_STRIDE);
    o[0] = make_float4(nanv, nanv, nanv, nanv);
    o[1] = make_float4(nanv, __int_as_float(-1), 0.0f, 0.0f);
    // runs that hold nothing but padding get a sphere of their own (the last real run keeps the
    // sphere of its real points: k_cloud_seg)
    if ((q & (SEG - 1)) == 0 && q >= ((n + SEG - 1) / SEG) * SEG)
        seg[q / SEG] = make_float4(p.x + (axis ? 0.0f : 8.0f * SEG), p.y + (axis ? 8.0f * SEG : 0.0f), p.z, 8.5f * SEG);
}

}   // namespace

size_t cloud_sort_scratch_bytes(int n)
{
    size_t bytes = 0;
    uint32_t *k = nullptr;
    int *v = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, k, k, v, v, (size_t)(n > 0 ? n : 1), 0u, 30u, (hipStream_t)0);
    return bytes;
}

hipError_t cloud_bbox_device(const float *d_xyz, int n, float *d_bbox6, hipStream_t s)
{
    hipLaunchKernelGGL(k_cloud_bbox, dim3(1), dim3(1024), 0, s, d_xyz, n, d_bbox6);
    return hipGetLastError();
}

hipError_t cloud_prepare_device(const CloudPrep &c, hipStream_t s)
{
    const int n = c.n;
    if (n <= 0) return hipSuccess;
    const int nb = (n + CB - 1) / CB;
    hipLaunchKernelGGL(k_cloud_keys, dim3(nb), dim3(CB), 0, s, c.xyz, n, c.bbox, c.keys[0], c.idx[0]);
    size_t bytes = c.scratch_bytes;
    // stable LSD radix sort over the 30 key bits: the permutation of sorting (key, index) pairs
    hipError_t e = rocprim::radix_sort_pairs(c.scratch, bytes, c.keys[0], c.keys[1], c.idx[0], c.idx[1], (size_t)n,
                                             0u, 30u, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_cloud_pack, dim3(nb), dim3(CB), 0, s, c.xyz, c.feat, n, c.colmajor, c.idx[1], c.pos, c.feat8);
    const int nseg = (n + SEG - 1) / SEG;
    hipLaunchKernelGGL(k_cloud_seg, dim3((nseg + CB / 64 - 1) / (CB / 64)), dim3(CB), 0, s, c.pos, n, nseg, c.seg);
    if (c.np > n)
        hipLaunchKernelGGL(k_cloud_pad, dim3((c.np - n + CB - 1) / CB), dim3(CB), 0, s, c.pos, c.feat8, c.seg, n, c.np,
                           c.bbox, c.pad_axis);
    return hipGetLastError();
}

}   // namespace cvo_dev
