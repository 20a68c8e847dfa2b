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

#include <algorithm>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include <atomic>

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

// ---------------------------------------------------------------------------
// The whole preparation of one cloud of up to CLOUD_ONE_MAX points by ONE block of 1024 threads (16 waves):
// the steps and the arithmetic of k_cloud_bbox / k_cloud_keys / the stable radix sort / k_cloud_pack /
// k_cloud_seg / k_cloud_pad above, with the sort in LDS.  The sort: the 30-bit keys stay where they are
// (keys[i], 4 B per point), what moves is the permutation (16-bit point indices, two buffers): four stable
// counting passes of 8 bits.  Stability without atomics: wave w owns the w-th sixteenth of the positions and
// walks it in order, 64 at a time; lanes with equal digits find each other with eight ballots, take consecutive
// places in lane order, and the last of them advances the wave's counter of that digit.
// LDS: 4 n + 2 * 2 n + 16 * 256 * 4 + 256 * 4 + a few words: 147.6 KB at n = 16384 (one block per CU).
constexpr int ONE_T = 1024, ONE_W = ONE_T / 64;

__device__ __forceinline__ void cloud_one_body(const CloudJob &jb, char *smem, const int ncap)
{
    uint32_t *keys = reinterpret_cast<uint32_t *>(smem);
    uint16_t *ord0 = reinterpret_cast<uint16_t *>(smem + (size_t)ncap * 4);
    uint16_t *ord1 = ord0 + ncap;
    uint32_t *cnt = reinterpret_cast<uint32_t *>(smem + (size_t)ncap * 8);   // [ONE_W][256]
    uint32_t *dbase = cnt + ONE_W * 256;                                      // [256]
    float *s_box = reinterpret_cast<float *>(dbase + 256);                    // [6], then [16][6] scratch
    float *s_red = s_box + 8;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int n = jb.n;
    const float *xyz = jb.xyz;
    // ---- bounding box (k_cloud_bbox)
    {
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int i = tid; i < n; i += ONE_T)
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float v = xyz[3 * (size_t)i + a];
                lo[a] = v < lo[a] ? v : lo[a];
                hi[a] = v > hi[a] ? v : hi[a];
            }
#pragma unroll
        for (int a = 0; a < 3; ++a)
            for (int off = 32; off > 0; off >>= 1) {
                const float l2 = __shfl_xor(lo[a], off), h2 = __shfl_xor(hi[a], off);
                lo[a] = l2 < lo[a] ? l2 : lo[a];
                hi[a] = h2 > hi[a] ? h2 : hi[a];
            }
        if (lane == 0)
            for (int a = 0; a < 3; ++a) { s_red[wid * 6 + a] = lo[a]; s_red[wid * 6 + 3 + a] = hi[a]; }
        __syncthreads();
        if (tid < 3) {
            float l = INFINITY, h = -INFINITY;
            for (int w = 0; w < ONE_W; ++w) {
                l = s_red[w * 6 + tid] < l ? s_red[w * 6 + tid] : l;
                h = s_red[w * 6 + 3 + tid] > h ? s_red[w * 6 + 3 + tid] : h;
            }
            s_box[tid] = l;
            s_box[3 + tid] = h;
            jb.bbox_out[tid] = l;
            jb.bbox_out[3 + tid] = h;
        }
        __syncthreads();
    }
    // ---- Morton keys (k_cloud_keys) and the identity permutation
    for (int i = tid; i < n; i += ONE_T) {
        uint32_t q[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float lo = s_box[a], ext = s_box[3 + a] - lo;
            const float inv = (ext > 0.0f && ext <= 3.4e38f) ? 1023.0f / ext : 0.0f;
            float f = (xyz[3 * (size_t)i + a] - lo) * inv;
            if (!(f >= 0.0f)) f = 0.0f;
            if (f > 1023.0f) f = 1023.0f;
            q[a] = (uint32_t)f;
        }
        keys[i] = spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
        ord0[i] = (uint16_t)i;
    }
    __syncthreads();
    // ---- stable LSD radix sort of the permutation by key, 8 bits a pass
    const int chunk = (((n + ONE_W - 1) / ONE_W) + 63) & ~63;   // positions per wave, whole rounds of 64
    const int p_lo = wid * chunk, p_hi = min(n, p_lo + chunk);
    uint16_t *src = ord0, *dst = ord1;
    for (int shift = 0; shift < 30; shift += 8) {
        for (int q = tid; q < ONE_W * 256; q += ONE_T) cnt[q] = 0u;
        __syncthreads();
        uint32_t *mycnt = cnt + wid * 256;
        // count: how many of this wave's positions carry each digit
        for (int p0 = p_lo; p0 < p_hi; p0 += 64) {
            const int p = p0 + lane;
            const bool live = p < p_hi;
            const unsigned d = live ? ((keys[src[p]] >> shift) & 255u) : 0u;
            unsigned long long peers = __ballot(live);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const unsigned long long bal = __ballot((d >> b) & 1u);
                peers &= ((d >> b) & 1u) ? bal : ~bal;
            }
            if (live && (peers >> lane) == 1ull) mycnt[d] += (unsigned)__popcll(peers);   // (the last lane of a group)
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
        // scan: digit-major, wave-minor -- where each wave's run of each digit starts
        if (tid < 256) {
            unsigned run = 0u;
            for (int w = 0; w < ONE_W; ++w) {
                const unsigned c = cnt[w * 256 + tid];
                cnt[w * 256 + tid] = run;
                run += c;
            }
            dbase[tid] = run;   // the digit's total
        }
        __syncthreads();
        if (tid < 64) {   // exclusive scan of the 256 totals: four per lane, a wave scan in between
            unsigned t[4], sum = 0u;
#pragma unroll
            for (int k = 0; k < 4; ++k) { t[k] = dbase[tid * 4 + k]; sum += t[k]; }
            unsigned inc = sum;
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned o = __shfl_up(inc, off);
                if (lane >= off) inc += o;
            }
            unsigned run = inc - sum;
#pragma unroll
            for (int k = 0; k < 4; ++k) { dbase[tid * 4 + k] = run; run += t[k]; }
        }
        __syncthreads();
        // scatter, in order
        for (int p0 = p_lo; p0 < p_hi; p0 += 64) {
            const int p = p0 + lane;
            const bool live = p < p_hi;
            const unsigned i = live ? src[p] : 0u;
            const unsigned d = live ? ((keys[i] >> shift) & 255u) : 0u;
            unsigned long long peers = __ballot(live);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const unsigned long long bal = __ballot((d >> b) & 1u);
                peers &= ((d >> b) & 1u) ? bal : ~bal;
            }
            if (live) {
                const unsigned below = (unsigned)__popcll(peers & ((1ull << lane) - 1ull));
                const unsigned at = dbase[d] + mycnt[d] + below;
                dst[at] = (uint16_t)i;
            }
            __builtin_amdgcn_wave_barrier();
            if (live && (peers >> lane) == 1ull) mycnt[d] += (unsigned)__popcll(peers);
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
        uint16_t *t = src; src = dst; dst = t;
    }
    // ---- packed rows (k_cloud_pack) and the bounding sphere of every run of SEG (k_cloud_seg): a wave per run
    const int nseg = (n + SEG - 1) / SEG;
    const float *feat = jb.feat;
    for (int g = wid; g < nseg; g += ONE_W) {
        const int s = g * SEG + lane;
        const bool live = s < n;
        const int i = src[live ? s : n - 1];
        float f[CVO_HIP_NFEAT];
#pragma unroll
        for (int q = 0; q < CVO_HIP_NFEAT; ++q)
            f[q] = jb.colmajor ? feat[(size_t)q * n + i] : feat[(size_t)i * CVO_HIP_NFEAT + q];
        const float4 p = make_float4(xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], f[4]);
        if (live) {
            jb.pos[s] = p;
            float4 *o = reinterpret_cast<float4 *>(jb.feat8 + (size_t)s * FEAT_STRIDE);
            o[0] = make_float4(f[0], f[1], f[2], f[3]);
            o[1] = make_float4(f[4], __int_as_float(i), 0.0f, 0.0f);
        }
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
        if (lane == 0) jb.seg[g] = make_float4(c[0], c[1], c[2], (float)(sqrt(d2) * 1.00001 + 1e-6));
    }
    // ---- padding rows (k_cloud_pad)
    {
        const float bx = 0.5f * (s_box[0] + s_box[3]), by = 0.5f * (s_box[1] + s_box[4]), bz = 0.5f * (s_box[2] + s_box[5]);
        const float nanv = __int_as_float(0x7fc00000);
        for (int q = n + tid; q < jb.np; q += ONE_T) {
            const float off = 1.0e4f + 16.0f * (float)(q - n);
            const float4 p = jb.pad_axis ? make_float4(bx, by + off, bz, nanv) : make_float4(bx + off, by, bz, nanv);
            jb.pos[q] = p;
            float4 *o = reinterpret_cast<float4 *>(jb.feat8 + (size_t)q * FEAT_STRIDE);
            o[0] = make_float4(nanv, nanv, nanv, nanv);
            o[1] = make_float4(nanv, __int_as_float(-1), 0.0f, 0.0f);
            if ((q & (SEG - 1)) == 0 && q >= ((n + SEG - 1) / SEG) * SEG)
                jb.seg[q / SEG] = make_float4(p.x + (jb.pad_axis ? 0.0f : 8.0f * SEG), p.y + (jb.pad_axis ? 8.0f * SEG : 0.0f), p.z, 8.5f * SEG);
        }
    }
}

__global__ void __launch_bounds__(ONE_T) k_cloud_one(const CloudJob *jobs, const int ncap)
{
    extern __shared__ __attribute__((aligned(16))) char smem_one[];
    const CloudJob jb = jobs[blockIdx.x];
    if (jb.n <= 0) return;
    cloud_one_body(jb, smem_one, ncap);
}

__global__ void __launch_bounds__(ONE_T) k_cloud_one_value(const CloudJob jb, const int ncap)
{
    extern __shared__ __attribute__((aligned(16))) char smem_one[];
    if (jb.n <= 0) return;
    cloud_one_body(jb, smem_one, ncap);
}

int cloud_one_cap(int nmax) { return std::max(64, (nmax + 63) & ~63); }

}   // namespace

size_t cloud_one_smem_bytes(int nmax)
{
    return (size_t)cloud_one_cap(nmax) * 8 + (size_t)(ONE_W * 256 + 256) * 4 + (8 + ONE_W * 6) * sizeof(float) + 64;
}

namespace {
template <class K> hipError_t allow_smem(K kernel, size_t bytes)
{
    // (more than 64 KB of dynamic LDS per block must be asked for)
    return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}
}   // namespace

hipError_t cloud_prepare_many(const CloudJob *d_jobs, int count, int nmax, hipStream_t s)
{
    if (count <= 0) return hipSuccess;
    if (nmax > CLOUD_ONE_MAX) return hipErrorInvalidValue;
    const size_t smem = cloud_one_smem_bytes(nmax);
    {   // (an attribute of the function on the CURRENT device: once per device, whichever thread comes first)
        static std::atomic<unsigned char> allowed[64];
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
        if (!allowed[dev].load(std::memory_order_acquire)) {
            const hipError_t e = allow_smem(k_cloud_one, cloud_one_smem_bytes(CLOUD_ONE_MAX));
            if (e != hipSuccess) return e;
            allowed[dev].store(1, std::memory_order_release);
        }
    }
    hipLaunchKernelGGL(k_cloud_one, dim3((unsigned)count), dim3(ONE_T), smem, s, d_jobs, cloud_one_cap(nmax));
    return hipGetLastError();
}

hipError_t cloud_prepare_one(const CloudJob &job, hipStream_t s)
{
    if (job.n <= 0) return hipSuccess;
    if (job.n > CLOUD_ONE_MAX) return hipErrorInvalidValue;
    const size_t smem = cloud_one_smem_bytes(job.n);
    {
        static std::atomic<unsigned char> allowed[64];
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
        if (!allowed[dev].load(std::memory_order_acquire)) {
            const hipError_t e = allow_smem(k_cloud_one_value, cloud_one_smem_bytes(CLOUD_ONE_MAX));
            if (e != hipSuccess) return e;
            allowed[dev].store(1, std::memory_order_release);
        }
    }
    hipLaunchKernelGGL(k_cloud_one_value, dim3(1), dim3(ONE_T), smem, s, job, cloud_one_cap(job.n));
    return hipGetLastError();
}

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
