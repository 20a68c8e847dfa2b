// cvo_frontend.hip -- the RGB-D front end on MI355X (gfx950): C-ABI of
// include/cvo_frontend.h.  One frame = ~20 small launches on one stream; every
// kernel is a per-pixel or per-cell pass over a 640x480-class image (0.3 M pixels,
// ~1 MB per plane), i.e. launch- and latency-bound, not bandwidth-bound; the
// data-dependent decisions of the selector (second pass or not, sub-sampling
// threshold) are taken by one-thread kernels on the device so that the host does
// not have to look at the frame before the cloud is ready.
//
// What each kernel restates is cited at the kernel.  The arithmetic contract of the
// library holds here too (-ffp-contract=off, correctly rounded sqrt / divide).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "cvo_frontend.h"
#include "cvo_lock.h"

namespace {

constexpr int FE_LEVELS = 3;
constexpr int FE_BLOCK = 256;
constexpr int FE_CHUNK = 4 * FE_BLOCK;   // pixels per block of the ordered (scan) kernels

// device-resident control words of one frame
struct FeCtrl {
    int n[2][3];        // pixels chosen at level 0 / 1 / 2 by selection pass 0 / 1
    int pot[2];         // potential of pass 0 / 1
    int redo;           // pass 1 runs
    int do_sub;         // the map is sub-sampled
    int char_th;        // ... with this threshold on the random bytes
    int num_have;       // pixels chosen by the pass that counts
    int in_map;         // pixels in the map when the cloud was emitted
    int pad2_[2];
    int num_points;     // selected pixels with depth
    int changed;        // hysteresis sweep flag
    int pad_;
};

struct FeDims {
    int w, h;
    int wl[FE_LEVELS], hl[FE_LEVELS];
    int w32, h32;
};

__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// exclusive scan of one int per thread over a 256-thread block; *total = block sum
__device__ __forceinline__ int block_excl_scan(int v, int *total)
{
    __shared__ int s_w[FE_BLOCK / 64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(inc, off);
        if (lane >= off) inc += o;
    }
    __syncthreads();   // (s_w may still be read by a previous call)
    if (lane == 63) s_w[wid] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int q = 0; q < FE_BLOCK / 64; ++q) {
        if (q < wid) base += s_w[q];
        tot += s_w[q];
    }
    *total = tot;
    return base + inc - v;
}

// sum of cnt[0 .. b) by the whole block
__device__ __forceinline__ int block_base(const int *cnt, int b)
{
    __shared__ int s_r[FE_BLOCK / 64];
    int v = 0;
    for (int q = threadIdx.x; q < b; q += FE_BLOCK) v += cnt[q];
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_r[threadIdx.x >> 6] = v;
    __syncthreads();
    int s = 0;
#pragma unroll
    for (int q = 0; q < FE_BLOCK / 64; ++q) s += s_r[q];
    return s;
}

// Level 0 in one pass.  load_image: cv::cvtColor RGB2GRAY and RGB2HSV on 8-bit data,
// channel 0 taken as R (ref src/pcd_generator.cpp:389-390), OpenCV's fixed-point
// definitions; the grey image as float is level 0 of the pyramid (ref :53-61); its
// central differences on the flattened image for idx in [w, w*(h-1)) (ref :95-113; the
// rest is zero here).  The four neighbours' grey values are recomputed from the colour
// image instead of waiting for another launch.
__global__ void __launch_bounds__(FE_BLOCK) k_fe_level0(const uint8_t *img, int w, int h, const int *sdiv,
                                                        const int *hdiv, uint8_t *gray, uint32_t *hsv, float *I0,
                                                        float *ag, float *dx_out, float *dy_out)
{
    const int i = blockIdx.x * FE_BLOCK + threadIdx.x;
    const int np = w * h;
    if (i >= np) return;
    auto grey = [&](int j) {
        return (img[3 * j] * 4899 + img[3 * j + 1] * 9617 + img[3 * j + 2] * 1868 + (1 << 13)) >> 14;
    };
    const int r = img[3 * i], g = img[3 * i + 1], b = img[3 * i + 2];
    const int gr = (r * 4899 + g * 9617 + b * 1868 + (1 << 13)) >> 14;
    gray[i] = (uint8_t)gr;
    I0[i] = (float)gr;
    int v = max(b, max(g, r)), vmin = min(b, min(g, r));
    const int diff = v - vmin;
    const int vr = v == r ? -1 : 0, vg = v == g ? -1 : 0;
    const int s = (diff * sdiv[v] + (1 << 11)) >> 12;
    int hh = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
    hh = (hh * hdiv[diff] + (1 << 11)) >> 12;
    hh += hh < 0 ? 180 : 0;
    hh = min(max(hh, 0), 255);
    hsv[i] = (uint32_t)hh | ((uint32_t)s << 8) | ((uint32_t)v << 16);
    float dx = 0.0f, dy = 0.0f, a = 0.0f;
    if (i >= w && i < w * (h - 1)) {
        dx = 0.5f * ((float)grey(i + 1) - (float)grey(i - 1));
        dy = 0.5f * ((float)grey(i + w) - (float)grey(i - w));
        a = dx * dx + dy * dy;
    }
    ag[i] = a;
    dx_out[i] = dx;
    dy_out[i] = dy;
}

// Level l > 0 in one pass: the 2x2 mean of level l-1 (ref src/pcd_generator.cpp:79-93)
// and its squared gradient magnitude (ref :95-113); the neighbours' means are recomputed.
__global__ void __launch_bounds__(FE_BLOCK) k_fe_level(const float *prev, int pw, float *cur, int wl, int hl, float *ag)
{
    const int i = blockIdx.x * FE_BLOCK + threadIdx.x;
    if (i >= wl * hl) return;
    auto mean = [&](int j) {
        const int y = j / wl, x = j - y * wl;
        const float *p = prev + 2 * x + 2 * y * pw;
        return 0.25f * (((p[0] + p[1]) + p[pw]) + p[pw + 1]);
    };
    cur[i] = mean(i);
    float a = 0.0f;
    if (i >= wl && i < wl * (hl - 1)) {
        float dx = 0.5f * (mean(i + 1) - mean(i - 1));
        float dy = 0.5f * (mean(i + wl) - mean(i - wl));
        if (!isfinite(dx)) dx = 0.0f;
        if (!isfinite(dy)) dy = 0.0f;
        a = dx * dx + dy * dy;
    }
    ag[i] = a;
}

// PixelSelector::makeHists, first half (ref thirdparty/PixelSelector2.cpp:70-104): one
// block per 32x32 cell; histogram of the integer gradient magnitudes, its median + 7
__global__ void __launch_bounds__(FE_BLOCK) k_fe_hist(const float *ag0, int w, int h, int w32, float *ths)
{
    __shared__ int hist[50];
    const int cx = blockIdx.x % w32, cy = blockIdx.x / w32;
    if (threadIdx.x < 50) hist[threadIdx.x] = 0;
    __syncthreads();
    for (int q = threadIdx.x; q < 1024; q += FE_BLOCK) {
        const int it = (q & 31) + 32 * cx, jt = (q >> 5) + 32 * cy;
        if (it > w - 2 || jt > h - 2 || it < 1 || jt < 1) continue;
        int g = (int)sqrtf(ag0[it + jt * w]);
        if (g > 48) g = 48;
        atomicAdd(&hist[g + 1], 1);
        atomicAdd(&hist[0], 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // computeHistQuantil(hist, 0.5) (ref :58-67; bins past the 50 cleared ones are empty)
        int th = (int)(hist[0] * 0.5f + 0.5f);
        int res = 90;
        for (int i = 0; i < 90; ++i) {
            th -= (i + 1 < 50) ? hist[i + 1] : 0;
            if (th < 0) { res = i; break; }
        }
        ths[blockIdx.x] = (float)(res + 7);
    }
}

// makeHists, second half (ref thirdparty/PixelSelector2.cpp:106-131): squared 3x3 mean
__global__ void __launch_bounds__(FE_BLOCK) k_fe_smooth(const float *ths, int w32, int h32, float *out)
{
    const int c = blockIdx.x * FE_BLOCK + threadIdx.x;
    if (c >= w32 * h32) return;
    const int x = c % w32, y = c / w32;
    float sum = 0, num = 0;
    if (x > 0) {
        if (y > 0) { num++; sum += ths[x - 1 + (y - 1) * w32]; }
        if (y < h32 - 1) { num++; sum += ths[x - 1 + (y + 1) * w32]; }
        num++; sum += ths[x - 1 + y * w32];
    }
    if (x < w32 - 1) {
        if (y > 0) { num++; sum += ths[x + 1 + (y - 1) * w32]; }
        if (y < h32 - 1) { num++; sum += ths[x + 1 + (y + 1) * w32]; }
        num++; sum += ths[x + 1 + y * w32];
    }
    if (y > 0) { num++; sum += ths[x + (y - 1) * w32]; }
    if (y < h32 - 1) { num++; sum += ths[x + (y + 1) * w32]; }
    num++; sum += ths[x + y * w32];
    out[c] = (sum / num) * (sum / num);
}

struct SelectArgs {
    const float *ag0, *ag1, *ag2, *ths;
    float *map;
    FeCtrl *ctrl;
    int *blk_cnt;   // [block][3]: pixels chosen at level 0 / 1 / 2 by that block
    int w, h, w32, ncell, pass;
};

// PixelSelector::select (ref thirdparty/PixelSelector2.cpp:240-437), direction
// distribution off (PixelSelector2.h:31).  The reference walks every block B4 of
// 4pot x 4pot pixels through its four 2pot blocks B3 and their four pot blocks B2 with
// three running maxima and two sticky "-2" marks; with the random directions unused
// that walk computes, for the pixels inside the margin (ref :313):
//   * every B2 holding a pixel with ag0 > th0 keeps the first largest ag0 among them
//     (map 1);
//   * every B3 WITHOUT such a pixel that holds a pixel with ag1 > th1 keeps the first
//     largest ag1 among those (map 2) -- the first level-0 hit of a B3 sets
//     bestIdx3 = -2 for good (ref :325,:328,:366);
//   * a B4 without either kind that holds a pixel with ag2 > th2 keeps the first largest
//     ag2 (map 4) -- any level-0 or level-1 improvement sets bestIdx4 = -2 (ref :325,:338);
// "first" in the reference's visiting order (B3s, their B2s, rows, columns).  One wave
// takes one B4: its lanes test the pixels (coalesced rows) and post (value, order) keys
// to 21 LDS slots with 64-bit atomic max; 21 lanes then write the picks.
__global__ void __launch_bounds__(FE_BLOCK) k_fe_select(const SelectArgs a)
{
    if (a.pass == 1 && !a.ctrl->redo) return;
    __shared__ unsigned long long s_slot[FE_BLOCK / 64][24];
    __shared__ int s_n[FE_BLOCK / 64][3];
    const int pot = a.ctrl->pot[a.pass];
    const int w = a.w, h = a.h, side = 4 * pot;
    const int cw = (w + side - 1) / side, ch = (h + side - 1) / side;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int cell = blockIdx.x * (FE_BLOCK / 64) + wid;
    const bool live = cell < cw * ch;
    unsigned long long *slot = s_slot[wid];
    if (lane < 24) slot[lane] = 0ull;
    __syncthreads();
    const int y4 = live ? (cell / cw) * side : 0, x4 = live ? (cell % cw) * side : 0;
    if (live) {
        const int w1 = w / 2, w2 = w / 4;
        const float dw1 = 0.75f, dw2 = dw1 * dw1, thFactor = 1.0f;
        for (int q = lane; q < side * side; q += 64) {
            const int ly = q / side, lx = q - ly * side;
            const int xf = x4 + lx, yf = y4 + ly;
            if (xf >= w || yf >= h) continue;
            const int idx = xf + w * yf;
            a.map[idx] = 0.0f;   // (ref :276)
            if (xf < 4 || xf >= w - 5 || yf < 4 || yf > h - 4) continue;
            const int bx3 = lx / (2 * pot), by3 = ly / (2 * pot);
            const int rx = lx - bx3 * 2 * pot, ry = ly - by3 * 2 * pot;
            const int bx2 = rx / pot, by2 = ry / pot;
            const int x1 = rx - bx2 * pot, y1 = ry - by2 * pot;
            const int b3 = by3 * 2 + bx3, b2 = b3 * 4 + by2 * 2 + bx2;
            const unsigned order = ~(unsigned)((b2 * pot + y1) * pot + x1);   // earlier pixel = larger key
            const int cellth = min((xf >> 5) + (yf >> 5) * a.w32, a.ncell - 1);
            const float th0 = a.ths[cellth];
            const float th1 = th0 * dw1;
            const float th2 = th1 * dw2;
            const float a0 = a.ag0[idx];
            const float a1 = a.ag1[(int)(xf * 0.5f + 0.25f) + (int)(yf * 0.5f + 0.25f) * w1];
            const float a2 = a.ag2[(int)(xf * 0.25f + 0.125f) + (int)(yf * 0.25f + 0.125f) * w2];
            if (a0 > th0 * thFactor)
                atomicMax(&slot[b2], ((unsigned long long)__float_as_uint(a0) << 32) | order);
            if (a1 > th1 * thFactor)
                atomicMax(&slot[16 + b3], ((unsigned long long)__float_as_uint(a1) << 32) | order);
            if (a2 > th2 * thFactor)
                atomicMax(&slot[20], ((unsigned long long)__float_as_uint(a2) << 32) | order);
        }
    }
    __syncthreads();
    int pick = 0;   // 1 / 2 / 4: this lane writes a pick of that level
    if (live && lane < 21) {
        unsigned long long v = 0ull;
        if (lane < 16) {
            v = slot[lane];
            pick = v ? 1 : 0;
        } else if (lane < 20) {
            const int b3 = lane - 16;
            const bool lvl0 = (slot[4 * b3] | slot[4 * b3 + 1] | slot[4 * b3 + 2] | slot[4 * b3 + 3]) != 0ull;
            v = slot[lane];
            pick = (!lvl0 && v) ? 2 : 0;
        } else {
            unsigned long long any = 0ull;
            for (int q = 0; q < 20; ++q) any |= slot[q];
            v = slot[20];
            pick = (!any && v) ? 4 : 0;
        }
        if (pick) {
            const int r = (int)~(unsigned)v;   // visiting order -> pixel
            const int x1 = r % pot, y1 = (r / pot) % pot, b2 = r / (pot * pot);
            const int b3 = b2 >> 2, bx3 = b3 & 1, by3 = b3 >> 1, bx2 = b2 & 1, by2 = (b2 >> 1) & 1;
            const int xf = x4 + bx3 * 2 * pot + bx2 * pot + x1, yf = y4 + by3 * 2 * pot + by2 * pot + y1;
            a.map[xf + w * yf] = (float)pick;
        }
    }
    const int n2 = __popcll(__ballot(pick == 1)), n3 = __popcll(__ballot(pick == 2)), n4 = __popcll(__ballot(pick == 4));
    if (lane == 0) { s_n[wid][0] = n2; s_n[wid][1] = n3; s_n[wid][2] = n4; }
    __syncthreads();
    if (threadIdx.x < 3)
        a.blk_cnt[3 * blockIdx.x + threadIdx.x] =
            (s_n[0][threadIdx.x] + s_n[1][threadIdx.x]) + (s_n[2][threadIdx.x] + s_n[3][threadIdx.x]);
}

// PixelSelector::makeMaps, the decisions (ref thirdparty/PixelSelector2.cpp:137-207):
// stage 0 after the first selection pass (re-select with another potential?), stage 1
// after the second (sub-sample?).  One block: it first adds up the per-block counts of
// the selection pass that just ran (`nb` blocks).
__global__ void __launch_bounds__(FE_BLOCK) k_fe_decide(FeCtrl *c, const int *blk_cnt, int nb, int stage,
                                                        float num_want_f)
{
    __shared__ int s_r[FE_BLOCK / 64][3];
    __shared__ int s_skip;
    if (threadIdx.x == 0) s_skip = (stage == 1 && !c->redo) ? 1 : 0;
    __syncthreads();
    if (!s_skip) {   // (pass 1 did not run: its counts are stale)
        int v[3] = {0, 0, 0};
        for (int q = threadIdx.x; q < nb; q += FE_BLOCK)
            for (int k = 0; k < 3; ++k) v[k] += blk_cnt[3 * q + k];
        for (int k = 0; k < 3; ++k) v[k] = wave_sum(v[k]);
        if ((threadIdx.x & 63) == 0)
            for (int k = 0; k < 3; ++k) s_r[threadIdx.x >> 6][k] = v[k];
        __syncthreads();
        if (threadIdx.x == 0)
            for (int k = 0; k < 3; ++k) c->n[stage][k] = (s_r[0][k] + s_r[1][k]) + (s_r[2][k] + s_r[3][k]);
    }
    if (threadIdx.x != 0) return;
    if (stage == 0) {
        const int pot = c->pot[0];
        const float numHave = (float)(c->n[0][0] + c->n[0][1] + c->n[0][2]);
        const float quotia = num_want_f / numHave;
        const float K = numHave * (float)(pot + 1) * (float)(pot + 1);
        int ideal = (int)(sqrtf(K / num_want_f) - 1);
        if (ideal < 1) ideal = 1;
        int redo = 0;
        if (quotia > 1.25f && pot > 1) {
            if (ideal >= pot) ideal = pot - 1;
            redo = 1;
        } else if (quotia < 0.25f) {
            if (ideal <= pot) ideal = pot + 1;
            redo = 1;
        }
        c->redo = redo;
        c->pot[1] = ideal;
    } else {
        const int p = c->redo ? 1 : 0;
        const int have = c->n[p][0] + c->n[p][1] + c->n[p][2];
        const float quotia = num_want_f / (float)have;
        c->num_have = have;
        c->do_sub = ((double)quotia < 0.95) ? 1 : 0;
        c->char_th = c->do_sub ? (int)(unsigned char)(255 * quotia) : 255;
    }
}

// Ordered passes over the map.  Block b owns pixels [b*FE_CHUNK, (b+1)*FE_CHUNK),
// thread t of it the 4 consecutive ones from 4t: a count kernel, then a kernel in
// which every block sums the counts of the blocks before it.
// MODE 0: map != 0 (the sub-sampling's running index, ref PixelSelector2.cpp:213-226)
// MODE 1: map != 0 and depth != 0 (the cloud's point index, ref pcd_generator.cpp:304-321)
template <int MODE>
__device__ __forceinline__ int fe_flags(const float *map, const uint16_t *depth, int np, int i0)
{
    int f = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = i0 + q;
        if (i < np && map[i] != 0.0f && (MODE == 0 || depth[i] != 0)) f |= 1 << q;
    }
    return f;
}

template <int MODE>
__global__ void __launch_bounds__(FE_BLOCK) k_fe_count(const float *map, const uint16_t *depth, int np,
                                                       const FeCtrl *c, int *cnt)
{
    if (MODE == 0 && !c->do_sub) return;
    __shared__ int s_w[2][FE_BLOCK / 64];
    const int i0 = blockIdx.x * FE_CHUNK + 4 * threadIdx.x;
    const int f = fe_flags<MODE>(map, depth, np, i0);
    const int v = wave_sum(__popc(f));
    // MODE 1 also counts the pixels in the map, with or without depth (cnt[gridDim.x + b]):
    // their total is the selector's result (num_selected)
    const int u = MODE == 1 ? wave_sum(__popc(fe_flags<0>(map, depth, np, i0))) : 0;
    if ((threadIdx.x & 63) == 0) { s_w[0][threadIdx.x >> 6] = v; s_w[1][threadIdx.x >> 6] = u; }
    __syncthreads();
    if (threadIdx.x == 0) cnt[blockIdx.x] = (s_w[0][0] + s_w[0][1]) + (s_w[0][2] + s_w[0][3]);
    if (MODE == 1 && threadIdx.x == 1) cnt[gridDim.x + blockIdx.x] = (s_w[1][0] + s_w[1][1]) + (s_w[1][2] + s_w[1][3]);
}

// makeMaps, sub-sampling (ref thirdparty/PixelSelector2.cpp:209-226): the rn-th chosen
// pixel in scan order is dropped if the rn-th random byte exceeds the threshold
__global__ void __launch_bounds__(FE_BLOCK) k_fe_subsample(float *map, int np, const uint8_t *pattern,
                                                           const int *cnt, FeCtrl *c)
{
    if (!c->do_sub) return;
    const int i0 = blockIdx.x * FE_CHUNK + 4 * threadIdx.x;
    const int f = fe_flags<0>(map, nullptr, np, i0);
    int total;
    int rn = block_base(cnt, blockIdx.x) + block_excl_scan(__popc(f), &total);
    const int th = c->char_th;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (f & (1 << q)) {
            if ((int)pattern[rn] > th) map[i0 + q] = 0.0f;
            rn++;
        }
    // (num_selected = what is left: counted by the next pass, k_fe_count<1>)
}

struct EmitArgs {
    const float *map;
    const uint16_t *depth;
    const uint8_t *img;
    const uint32_t *hsv;
    const float *dx0, *dy0;
    const int *cnt;
    FeCtrl *ctrl;
    float *pos, *feat;
    int np, w, cap, feature_type, nblocks;
    float cam[5];
};

// get_points_from_pixels + get_features (ref src/pcd_generator.cpp:297-321, 336-380)
__global__ void __launch_bounds__(FE_BLOCK) k_fe_emit(const EmitArgs a)
{
    const int i0 = blockIdx.x * FE_CHUNK + 4 * threadIdx.x;
    const int f = fe_flags<1>(a.map, a.depth, a.np, i0);
    int total;
    int idx = block_base(a.cnt, blockIdx.x) + block_excl_scan(__popc(f), &total);
    if (blockIdx.x == a.nblocks - 1 && threadIdx.x == FE_BLOCK - 1) a.ctrl->num_points = idx + __popc(f);
    if (blockIdx.x == 0) {   // (uniform per block) the size of the map
        const int in_map = block_base(a.cnt + a.nblocks, a.nblocks);
        if (threadIdx.x == 0) a.ctrl->in_map = in_map;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (f & (1 << q)) {
            const int p = i0 + q;
            if (idx < a.cap) {
                const int y = p / a.w, x = p - y * a.w;
                const float z = (float)a.depth[p] / a.cam[0];
                a.pos[3 * idx + 2] = z;
                a.pos[3 * idx + 0] = ((float)x - a.cam[3]) * z / a.cam[1];
                a.pos[3 * idx + 1] = ((float)y - a.cam[4]) * z / a.cam[2];
                float *ft = a.feat + 5 * (size_t)idx;
                if (a.feature_type == 0) {
                    const uint32_t hv = a.hsv[p];
                    ft[0] = (float)((double)(hv & 255u) / 180.0);
                    ft[1] = (float)((double)((hv >> 8) & 255u) / 255.0);
                    ft[2] = (float)((double)((hv >> 16) & 255u) / 255.0);
                    ft[3] = (float)((double)a.dx0[p] / 255.0 * 2);
                    ft[4] = (float)((double)a.dy0[p] / 255.0 * 2);
                } else {
                    ft[0] = (float)a.img[3 * p];
                    ft[1] = (float)a.img[3 * p + 1];
                    ft[2] = (float)a.img[3 * p + 2];
                    ft[3] = a.dx0[p];
                    ft[4] = a.dy0[p];
                }
            }
            idx++;
        }
}

// ---- the Canny top-up (ref src/pcd_generator.cpp:143-175), only when ctrl->canny ----

__device__ __forceinline__ int reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}

// cv::blur 3x3, 8-bit (BORDER_REFLECT_101, rounded to nearest)
__global__ void __launch_bounds__(FE_BLOCK) k_fe_blur(const uint8_t *src, int w, int h, uint8_t *dst)
{
    const int i = blockIdx.x * FE_BLOCK + threadIdx.x;
    if (i >= w * h) return;
    const int y = i / w, x = i - y * w;
    int s = 0;
    for (int j = -1; j <= 1; ++j)
        for (int k = -1; k <= 1; ++k) s += src[reflect101(y + j, h) * w + reflect101(x + k, w)];
    dst[i] = (uint8_t)__double2int_rn((double)s * (1.0 / 9.0));
}

// cv::Canny, gradients: 3x3 Sobel with replicated borders, L1 magnitude
__global__ void __launch_bounds__(FE_BLOCK) k_fe_sobel(const uint8_t *src, int w, int h, short2 *g, int *mag)
{
    const int i = blockIdx.x * FE_BLOCK + threadIdx.x;
    if (i >= w * h) return;
    const int y = i / w, x = i - y * w;
    auto px = [&](int xx, int yy) { return (int)src[min(max(yy, 0), h - 1) * w + min(max(xx, 0), w - 1)]; };
    const int dx = (px(x + 1, y - 1) + 2 * px(x + 1, y) + px(x + 1, y + 1)) -
                   (px(x - 1, y - 1) + 2 * px(x - 1, y) + px(x - 1, y + 1));
    const int dy = (px(x - 1, y + 1) + 2 * px(x, y + 1) + px(x + 1, y + 1)) -
                   (px(x - 1, y - 1) + 2 * px(x, y - 1) + px(x + 1, y - 1));
    g[i] = make_short2((short)dx, (short)dy);
    mag[i] = abs(dx) + abs(dy);
}

// cv::Canny, non-maximum suppression: 0 candidate, 1 not an edge, 2 edge (above `high`)
__global__ void __launch_bounds__(FE_BLOCK) k_fe_nms(const short2 *g, const int *mag, int w, int h, int low, int high,
                                                     uint8_t *st)
{
    const int i = blockIdx.x * FE_BLOCK + threadIdx.x;
    if (i >= w * h) return;
    const int y = i / w, x = i - y * w;
    auto mg = [&](int xx, int yy) { return (xx < 0 || xx >= w || yy < 0 || yy >= h) ? 0 : mag[yy * w + xx]; };
    const int m = mag[i];
    bool is_max = false;
    if (m > low) {
        const int TG22 = 13573;   // (int)(0.41421356... * (1 << 15) + 0.5)
        const int xs = g[i].x, ys = g[i].y;
        const int ax = abs(xs), ay = abs(ys) << 15;
        const int tg22x = ax * TG22;
        if (ay < tg22x) {
            is_max = m > mg(x - 1, y) && m >= mg(x + 1, y);
        } else {
            const int tg67x = tg22x + (ax << 16);
            if (ay > tg67x) {
                is_max = m > mg(x, y - 1) && m >= mg(x, y + 1);
            } else {
                const int s = (xs ^ ys) < 0 ? -1 : 1;
                is_max = m > mg(x - s, y - 1) && m > mg(x + s, y + 1);
            }
        }
    }
    st[i] = !is_max ? 1 : (m > high ? 2 : 0);
}

// cv::Canny, hysteresis: one sweep -- a 32x32 tile (with a one-pixel halo) is brought to
// its local fixed point in LDS; the host repeats sweeps until none changed anything
__global__ void __launch_bounds__(FE_BLOCK) k_fe_hyst(uint8_t *st, int w, int h, int tiles_x, FeCtrl *c)
{
    __shared__ uint8_t t[34][36];
    __shared__ int s_changed, s_any;
    const int tx = (blockIdx.x % tiles_x) * 32, ty = (blockIdx.x / tiles_x) * 32;
    for (int q = threadIdx.x; q < 34 * 34; q += FE_BLOCK) {
        const int lx = q % 34, ly = q / 34, gx = tx + lx - 1, gy = ty + ly - 1;
        t[ly][lx] = (gx < 0 || gx >= w || gy < 0 || gy >= h) ? 1 : st[gy * w + gx];
    }
    if (threadIdx.x == 0) s_any = 0;
    __syncthreads();
    for (;;) {
        if (threadIdx.x == 0) s_changed = 0;
        __syncthreads();
        for (int q = threadIdx.x; q < 1024; q += FE_BLOCK) {
            const int lx = (q & 31) + 1, ly = (q >> 5) + 1;
            if (t[ly][lx] != 0) continue;
            bool strong = false;
            for (int j = -1; j <= 1; ++j)
                for (int k = -1; k <= 1; ++k) strong = strong || t[ly + j][lx + k] == 2;
            if (strong) { t[ly][lx] = 2; s_changed = 1; }
        }
        __syncthreads();
        const int ch = s_changed;
        __syncthreads();
        if (!ch) break;
        if (threadIdx.x == 0) s_any = 1;
    }
    __syncthreads();
    if (s_any) {
        for (int q = threadIdx.x; q < 1024; q += FE_BLOCK) {
            const int lx = (q & 31) + 1, ly = (q >> 5) + 1, gx = tx + lx - 1, gy = ty + ly - 1;
            if (gx < w && gy < h && t[ly][lx] == 2) st[gy * w + gx] = 2;
        }
        if (threadIdx.x == 0) c->changed = 1;
    }
}

// select_point, the top-up itself (ref src/pcd_generator.cpp:153-174): one thread per
// 8x8 block, first free edge pixel in row order
__global__ void __launch_bounds__(FE_BLOCK) k_fe_topup(const uint8_t *st, int w, int h, float *map)
{
    const int bw = (w + 7) / 8, bh = (h + 7) / 8;
    const int b = blockIdx.x * FE_BLOCK + threadIdx.x;
    if (b >= bw * bh) return;
    const int x = (b % bw) * 8, y = (b / bw) * 8;
    for (int j = 0; j < 8; ++j)
        for (int i = 0; i < 8; ++i) {
            if (x + i >= w || y + j >= h) continue;
            const int p = (y + j) * w + x + i;
            if (st[p] == 2 && map[p] == 0.0f) { map[p] = 1.0f; return; }
        }
}

__global__ void __launch_bounds__(FE_BLOCK) k_fe_edges(const uint8_t *st, int np, uint8_t *edges)
{
    const int i = blockIdx.x * FE_BLOCK + threadIdx.x;
    if (i < np) edges[i] = st[i] == 2 ? 255 : 0;
}

// ---- the C library's rand() after srand(seed): additive feedback generator of degree
// 31, separation 3 (glibc random_r.c TYPE_3), restated so that no global state is touched
struct LibcRand {
    int32_t st[34];
    int f, r;
    explicit LibcRand(unsigned seed)
    {
        int32_t word = seed ? (int32_t)seed : 1;
        st[0] = word;
        for (int i = 1; i < 31; ++i) {
            const long hi = word / 127773, lo = word % 127773;
            word = (int32_t)(16807 * lo - 2836 * hi);
            if (word < 0) word += 2147483647;
            st[i] = word;
        }
        f = 3; r = 0;
        for (int i = 0; i < 310; ++i) next();
    }
    int next()
    {
        const uint32_t v = (uint32_t)st[f] + (uint32_t)st[r];
        st[f] = (int32_t)v;
        f = f + 1 == 31 ? 0 : f + 1;
        r = r + 1 == 31 ? 0 : r + 1;
        return (int)(v >> 1);
    }
};

const float kCameras[6][5] = {{1000.0f, 616.368f, 616.745f, 319.935f, 243.639f},   // ref pcd_generator.cpp:243-249
                              {5000.0f, 517.3f, 516.5f, 318.6f, 255.3f},            // fr1 :250-256
                              {5000.0f, 520.9f, 521.0f, 325.1f, 249.7f},            // fr2 :257-263
                              {5000.0f, 535.4f, 539.2f, 320.1f, 247.6f},            // fr3 :264-270
                              {2000.0f, 718.856f, 718.856f, 607.1928f, 185.2157f},  // kitti 15 :271-277
                              {2000.0f, 707.0912f, 707.0912f, 601.8873f, 183.1104f}};   // kitti 05 :279-285

}   // namespace

struct FeGraph {
    uint64_t key = 0;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
};

struct cvo_fe_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    FeDims d{};
    int np = 0, nblk = 0, nchunks = 0;
    int num_want = 3000;
    int cap = 0;
    // device
    uint8_t *img = nullptr, *gray = nullptr, *pattern = nullptr, *tmp8 = nullptr, *st8 = nullptr, *edges = nullptr;
    uint16_t *depth = nullptr;
    uint32_t *hsv = nullptr;
    float *I[FE_LEVELS] = {}, *ag[FE_LEVELS] = {}, *dx0 = nullptr, *dy0 = nullptr, *map = nullptr;
    float *ths = nullptr, *ths_s = nullptr, *pos = nullptr, *feat = nullptr;
    int *sdiv = nullptr, *hdiv = nullptr, *cnt = nullptr, *mag = nullptr, *blk_cnt = nullptr;
    short2 *grad = nullptr;
    FeCtrl *ctrl = nullptr;
    // pinned host
    uint8_t *h_img = nullptr;
    uint16_t *h_depth = nullptr;
    FeCtrl *h_ctrl = nullptr;
    float *h_pos = nullptr, *h_feat = nullptr;
    cvo_fe_info info{};
    std::vector<FeGraph> graphs;   // one frame's device work, captured per (camera, features, ...)
    int copied = 0;            // points of the cloud already on their way to h_pos / h_feat
    bool pending = false;      // a frame was submitted and not collected yet
    bool device_output = false;   // collect_device() will be used: no copy of the cloud to the host
    int p_seq = 0, p_ftype = 0;
    std::string err;
};

namespace {

int fail(cvo_fe_ctx *c, int code, const char *what, hipError_t e = hipSuccess)
{
    if (c) {
        c->err = what;
        if (e != hipSuccess) { c->err += ": "; c->err += hipGetErrorString(e); }
    }
    return code;
}

#define FE_HIP(call)                                                            \
    do {                                                                        \
        const hipError_t e_ = (call);                                           \
        if (e_ != hipSuccess) return fail(ctx, CVO_HIP_ERR_HIP, #call, e_);     \
    } while (0)

template <class T> hipError_t dev_alloc(T **p, size_t n) { return hipMalloc((void **)p, n * sizeof(T)); }
template <class T> hipError_t pin_alloc(T **p, size_t n) { return hipHostMalloc((void **)p, n * sizeof(T), hipHostMallocDefault); }

inline int blocks(int n) { return (n + FE_BLOCK - 1) / FE_BLOCK; }

// the Canny path of a frame whose selection came out short (host loop: rare)
int run_canny(cvo_fe_ctx *ctx)
{
    const int w = ctx->d.w, h = ctx->d.h, np = ctx->np;
    hipStream_t s = ctx->stream;
    hipLaunchKernelGGL(k_fe_blur, dim3(blocks(np)), dim3(FE_BLOCK), 0, s, ctx->gray, w, h, ctx->tmp8);
    hipLaunchKernelGGL(k_fe_sobel, dim3(blocks(np)), dim3(FE_BLOCK), 0, s, ctx->tmp8, w, h, ctx->grad, ctx->mag);
    hipLaunchKernelGGL(k_fe_nms, dim3(blocks(np)), dim3(FE_BLOCK), 0, s, ctx->grad, ctx->mag, w, h, 0, 25, ctx->st8);
    const int tiles_x = (w + 31) / 32, tiles_y = (h + 31) / 32;
    for (int sweep = 0; sweep < w + h; ++sweep) {
        FE_HIP(hipMemsetAsync(&ctx->ctrl->changed, 0, sizeof(int), s));
        hipLaunchKernelGGL(k_fe_hyst, dim3(tiles_x * tiles_y), dim3(FE_BLOCK), 0, s, ctx->st8, w, h, tiles_x, ctx->ctrl);
        FE_HIP(hipMemcpyAsync(ctx->h_ctrl, ctx->ctrl, sizeof(FeCtrl), hipMemcpyDeviceToHost, s));
        FE_HIP(hipStreamSynchronize(s));
        if (!ctx->h_ctrl->changed) break;
    }
    hipLaunchKernelGGL(k_fe_edges, dim3(blocks(np)), dim3(FE_BLOCK), 0, s, ctx->st8, np, ctx->edges);
    hipLaunchKernelGGL(k_fe_topup, dim3(blocks(((w + 7) / 8) * ((h + 7) / 8))), dim3(FE_BLOCK), 0, s, ctx->st8, w, h,
                       ctx->map);
    FE_HIP(hipGetLastError());
    return CVO_HIP_OK;
}

int run_emit(cvo_fe_ctx *ctx, int dataset_seq, int feature_type)
{
    hipStream_t s = ctx->stream;
    hipLaunchKernelGGL(k_fe_count<1>, dim3(ctx->nchunks), dim3(FE_BLOCK), 0, s, ctx->map, ctx->depth, ctx->np, ctx->ctrl,
                       ctx->cnt);
    EmitArgs e{};
    e.map = ctx->map; e.depth = ctx->depth; e.img = ctx->img; e.hsv = ctx->hsv; e.dx0 = ctx->dx0; e.dy0 = ctx->dy0;
    e.cnt = ctx->cnt; e.ctrl = ctx->ctrl; e.pos = ctx->pos; e.feat = ctx->feat;
    e.np = ctx->np; e.w = ctx->d.w; e.cap = ctx->cap; e.feature_type = feature_type; e.nblocks = ctx->nchunks;
    cvo_fe_camera(dataset_seq, e.cam);
    hipLaunchKernelGGL(k_fe_emit, dim3(ctx->nchunks), dim3(FE_BLOCK), 0, s, e);
    FE_HIP(hipMemcpyAsync(ctx->h_ctrl, ctx->ctrl, sizeof(FeCtrl), hipMemcpyDeviceToHost, s));
    // the cloud follows optimistically (its size is not known yet): enough for any normal frame
    // (not when the consumer takes it from device memory: cvo_fe_set_device_output)
    ctx->copied = 0;
    if (ctx->device_output) return CVO_HIP_OK;
    ctx->copied = std::min(ctx->cap, std::max(4096, 2 * ctx->num_want));
    FE_HIP(hipMemcpyAsync(ctx->h_pos, ctx->pos, (size_t)ctx->copied * 3 * sizeof(float), hipMemcpyDeviceToHost, s));
    FE_HIP(hipMemcpyAsync(ctx->h_feat, ctx->feat, (size_t)ctx->copied * 5 * sizeof(float), hipMemcpyDeviceToHost, s));
    return CVO_HIP_OK;
}

}   // namespace

extern "C" {

int cvo_fe_random_pattern(int n, uint8_t *out)
{
    if (n < 0 || (n > 0 && !out)) return CVO_HIP_ERR_INVALID;
    LibcRand g(3141592u);
    for (int i = 0; i < n; ++i) out[i] = (uint8_t)(g.next() & 0xFF);
    return CVO_HIP_OK;
}

int cvo_fe_camera(int dataset_seq, float cam[5])
{
    if (!cam) return CVO_HIP_ERR_INVALID;
    const int k = (dataset_seq >= 0 && dataset_seq <= 5) ? dataset_seq : 0;   // default: RealSense (ref :287-294)
    for (int q = 0; q < 5; ++q) cam[q] = kCameras[k][q];
    return CVO_HIP_OK;
}

const char *cvo_fe_last_error(const cvo_fe_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int cvo_fe_destroy(cvo_fe_ctx *ctx)
{
    cvo_lock::Api api_guard;
    if (!ctx) return CVO_HIP_ERR_INVALID;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    void *dev[] = {ctx->img, ctx->gray, ctx->pattern, ctx->tmp8, ctx->st8, ctx->edges, ctx->depth, ctx->hsv,
                   ctx->I[0], ctx->I[1], ctx->I[2], ctx->ag[0], ctx->ag[1], ctx->ag[2], ctx->dx0, ctx->dy0,
                   ctx->map, ctx->ths, ctx->ths_s, ctx->pos, ctx->feat, ctx->sdiv, ctx->hdiv, ctx->cnt,
                   ctx->mag, ctx->grad, ctx->ctrl, ctx->blk_cnt};
    for (void *p : dev)
        if (p) (void)hipFree(p);
    for (auto &g : ctx->graphs) {
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        if (g.graph) (void)hipGraphDestroy(g.graph);
    }
    void *pin[] = {ctx->h_img, ctx->h_depth, ctx->h_ctrl, ctx->h_pos, ctx->h_feat};
    for (void *p : pin)
        if (p) (void)hipHostFree(p);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return CVO_HIP_OK;
}

int cvo_fe_create(int device, void *stream, int width, int height, cvo_fe_ctx **out)
{
    cvo_lock::Api api_guard;
    if (!out || width < 64 || height < 64 || (long long)width * height > (1ll << 26)) return CVO_HIP_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev)
        return CVO_HIP_ERR_NODEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return CVO_HIP_ERR_NODEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return CVO_HIP_ERR_NODEVICE;
    cvo_fe_ctx *ctx = new (std::nothrow) cvo_fe_ctx;
    if (!ctx) return CVO_HIP_ERR_NOMEM;
    ctx->device = device;
    auto bail = [&](int code) { cvo_fe_destroy(ctx); return code; };
    if (hipSetDevice(device) != hipSuccess) return bail(CVO_HIP_ERR_HIP);
    if (stream) ctx->stream = (hipStream_t)stream;
    else {
        // lowest priority: when a frame is prepared while the previous one is being registered
        // (submit / collect), the registration's latency-bound launch chain goes first
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, least) != hipSuccess)
            return bail(CVO_HIP_ERR_HIP);
        ctx->own_stream = true;
    }
    FeDims &d = ctx->d;
    d.w = width; d.h = height; d.w32 = width / 32; d.h32 = height / 32;
    int wl = width, hl = height;
    for (int l = 0; l < FE_LEVELS; ++l) { d.wl[l] = wl; d.hl[l] = hl; wl /= 2; hl /= 2; }
    const size_t np = (size_t)width * height;
    ctx->np = (int)np;
    ctx->nchunks = (int)((np + FE_CHUNK - 1) / FE_CHUNK);
    ctx->cap = (int)np;   // (a selection can never hold more points than the image has pixels)
    bool ok = true;
    ok = ok && dev_alloc(&ctx->img, np * 3) == hipSuccess && dev_alloc(&ctx->gray, np) == hipSuccess;
    ok = ok && dev_alloc(&ctx->pattern, np) == hipSuccess && dev_alloc(&ctx->tmp8, np) == hipSuccess;
    ok = ok && dev_alloc(&ctx->st8, np) == hipSuccess && dev_alloc(&ctx->edges, np) == hipSuccess;
    ok = ok && dev_alloc(&ctx->depth, np) == hipSuccess && dev_alloc(&ctx->hsv, np) == hipSuccess;
    for (int l = 0; l < FE_LEVELS; ++l) {
        const size_t nl = (size_t)d.wl[l] * d.hl[l] + 8;
        ok = ok && dev_alloc(&ctx->I[l], nl) == hipSuccess && dev_alloc(&ctx->ag[l], nl) == hipSuccess;
    }
    ok = ok && dev_alloc(&ctx->dx0, np) == hipSuccess && dev_alloc(&ctx->dy0, np) == hipSuccess;
    ok = ok && dev_alloc(&ctx->map, np) == hipSuccess;
    ok = ok && dev_alloc(&ctx->ths, (size_t)d.w32 * d.h32 + 1) == hipSuccess;
    ok = ok && dev_alloc(&ctx->ths_s, (size_t)d.w32 * d.h32 + 1) == hipSuccess;
    ok = ok && dev_alloc(&ctx->pos, (size_t)ctx->cap * 3) == hipSuccess;
    ok = ok && dev_alloc(&ctx->feat, (size_t)ctx->cap * 5) == hipSuccess;
    ok = ok && dev_alloc(&ctx->sdiv, 256) == hipSuccess && dev_alloc(&ctx->hdiv, 256) == hipSuccess;
    ok = ok && dev_alloc(&ctx->cnt, 2 * (size_t)ctx->nchunks) == hipSuccess && dev_alloc(&ctx->mag, np) == hipSuccess;
    ok = ok && dev_alloc(&ctx->grad, np) == hipSuccess && dev_alloc(&ctx->ctrl, 1) == hipSuccess;
    ok = ok && dev_alloc(&ctx->blk_cnt, 3 * ((size_t)((width + 3) / 4) * ((height + 3) / 4) / 4 + 2)) == hipSuccess;
    ok = ok && pin_alloc(&ctx->h_img, np * 3) == hipSuccess && pin_alloc(&ctx->h_depth, np) == hipSuccess;
    ok = ok && pin_alloc(&ctx->h_ctrl, 1) == hipSuccess;
    ok = ok && pin_alloc(&ctx->h_pos, (size_t)ctx->cap * 3) == hipSuccess;
    ok = ok && pin_alloc(&ctx->h_feat, (size_t)ctx->cap * 5) == hipSuccess;
    if (!ok) return bail(CVO_HIP_ERR_NOMEM);
    // OpenCV's reciprocal tables of the 8-bit HSV conversion (12-bit fixed point)
    int sdiv[256], hdiv[256];
    sdiv[0] = hdiv[0] = 0;
    for (int i = 1; i < 256; ++i) {
        sdiv[i] = (int)std::lrint((255 << 12) / (1. * i));
        hdiv[i] = (int)std::lrint((180 << 12) / (6. * i));
    }
    // the selector's pattern does not depend on the frame: made once (the reference
    // re-seeds and redraws it for every frame, ref thirdparty/PixelSelector2.cpp:33-37)
    cvo_fe_random_pattern((int)np, ctx->h_img);
    if (hipMemcpy(ctx->pattern, ctx->h_img, np, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(ctx->sdiv, sdiv, sizeof(sdiv), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(ctx->hdiv, hdiv, sizeof(hdiv), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemset(ctx->edges, 0, np) != hipSuccess || hipMemset(ctx->map, 0, np * sizeof(float)) != hipSuccess)
        return bail(CVO_HIP_ERR_HIP);
    *out = ctx;
    return CVO_HIP_OK;
}

int cvo_fe_host_buffers(cvo_fe_ctx *ctx, uint8_t **img, uint16_t **depth)
{
    if (!ctx || !img || !depth) return CVO_HIP_ERR_INVALID;
    *img = ctx->h_img;
    *depth = ctx->h_depth;
    return CVO_HIP_OK;
}

int cvo_fe_set_device_output(cvo_fe_ctx *ctx, int on)
{
    if (!ctx) return CVO_HIP_ERR_INVALID;
    if (ctx->pending) return fail(ctx, CVO_HIP_ERR_INVALID, "set_device_output: a frame is in flight");
    ctx->device_output = on != 0;
    return CVO_HIP_OK;
}

int cvo_fe_set_num_want(cvo_fe_ctx *ctx, int num_want)
{
    if (!ctx || num_want < 1) return CVO_HIP_ERR_INVALID;
    ctx->num_want = num_want;
    return CVO_HIP_OK;
}

namespace {
int enqueue_frame(cvo_fe_ctx *ctx, int dataset_seq, int feature_type);
}

int cvo_fe_submit(cvo_fe_ctx *ctx, const uint8_t *img, size_t img_stride, const uint16_t *depth,
                  size_t depth_stride, int dataset_seq, int feature_type)
{
    cvo_lock::Api api_guard;
    if (!ctx) return CVO_HIP_ERR_INVALID;
    const int w = ctx->d.w, h = ctx->d.h, np = ctx->np;
    if (!img || !depth || img_stride < (size_t)w * 3 || depth_stride < (size_t)w * 2 ||
        (feature_type != CVO_FE_FEATURES_HSV && feature_type != CVO_FE_FEATURES_RGB))
        return fail(ctx, CVO_HIP_ERR_INVALID, "submit: bad argument");
    if (ctx->pending) return fail(ctx, CVO_HIP_ERR_INVALID, "submit: the previous frame was not collected");
    FE_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    // pinned staging; the depth rows are staged while the colour image is on its way
    if (img == ctx->h_img && img_stride == (size_t)w * 3) {
        // (the caller decoded into the staging image itself: cvo_fe_host_buffers)
    } else if (img_stride == (size_t)w * 3) std::memcpy(ctx->h_img, img, (size_t)np * 3);
    else
        for (int y = 0; y < h; ++y)
            std::memcpy(ctx->h_img + (size_t)y * w * 3, img + (size_t)y * img_stride, (size_t)w * 3);
    if (depth == ctx->h_depth && depth_stride == (size_t)w * 2) {
    } else if (depth_stride == (size_t)w * 2) std::memcpy(ctx->h_depth, depth, (size_t)np * 2);
    else
        for (int y = 0; y < h; ++y)
            std::memcpy(ctx->h_depth + (size_t)y * w, (const uint8_t *)depth + (size_t)y * depth_stride, (size_t)w * 2);
    FeCtrl c0{};
    c0.pot[0] = 3;   // a selector starts every frame at potential 3 (ref PixelSelector2.cpp:39)
    *ctx->h_ctrl = c0;
    // Everything from here to the copies back is the same sequence for every frame (fixed
    // buffers, fixed pinned staging): captured once per (camera, feature type, num_want,
    // output mode) and launched as one hipGraph -- 3 copies in, 11 kernels, the copies out.
    static const bool no_graph = getenv("CVO_FE_NO_GRAPH") != nullptr;
    const uint64_t key = ((uint64_t)(uint32_t)dataset_seq << 40) ^ ((uint64_t)feature_type << 36) ^
                         ((uint64_t)ctx->device_output << 32) ^ (uint64_t)(uint32_t)ctx->num_want;
    int rc = CVO_HIP_OK;
    FeGraph *g = nullptr;
    for (auto &e : ctx->graphs)
        if (e.key == key) g = &e;
    if (!no_graph && !g && ctx->graphs.size() < 8) {
        FeGraph ng;
        ng.key = key;
        cvo_lock::Capture alone;   // (see cvo_lock.h)
        if (alone.ok && hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed) == hipSuccess) {
            rc = enqueue_frame(ctx, dataset_seq, feature_type);
            const hipError_t e = hipStreamEndCapture(s, &ng.graph);
            if (!rc && e == hipSuccess && ng.graph &&
                hipGraphInstantiate(&ng.exec, ng.graph, nullptr, nullptr, 0) == hipSuccess) {
                ctx->graphs.push_back(ng);
                g = &ctx->graphs.back();
            } else {
                if (ng.graph) (void)hipGraphDestroy(ng.graph);
                (void)hipGetLastError();
                rc = CVO_HIP_OK;
            }
        }
    }
    ctx->copied = ctx->device_output ? 0 : std::min(ctx->cap, std::max(4096, 2 * ctx->num_want));
    if (g) {
        FE_HIP(hipGraphLaunch(g->exec, s));
    } else {
        rc = enqueue_frame(ctx, dataset_seq, feature_type);
        if (rc) return rc;
    }
    ctx->pending = true;
    ctx->p_seq = dataset_seq;
    ctx->p_ftype = feature_type;
    return CVO_HIP_OK;
}

namespace {
// one frame's device work, in stream order (also what a captured graph holds)
int enqueue_frame(cvo_fe_ctx *ctx, int dataset_seq, int feature_type)
{
    const int w = ctx->d.w, h = ctx->d.h, np = ctx->np;
    hipStream_t s = ctx->stream;
    FE_HIP(hipMemcpyAsync(ctx->img, ctx->h_img, (size_t)np * 3, hipMemcpyHostToDevice, s));
    FE_HIP(hipMemcpyAsync(ctx->depth, ctx->h_depth, (size_t)np * 2, hipMemcpyHostToDevice, s));
    FE_HIP(hipMemcpyAsync(ctx->ctrl, ctx->h_ctrl, sizeof(FeCtrl), hipMemcpyHostToDevice, s));

    const FeDims &d = ctx->d;
    hipLaunchKernelGGL(k_fe_level0, dim3(blocks(np)), dim3(FE_BLOCK), 0, s, ctx->img, w, h, ctx->sdiv, ctx->hdiv,
                       ctx->gray, ctx->hsv, ctx->I[0], ctx->ag[0], ctx->dx0, ctx->dy0);
    for (int l = 1; l < FE_LEVELS; ++l)
        hipLaunchKernelGGL(k_fe_level, dim3(blocks(d.wl[l] * d.hl[l])), dim3(FE_BLOCK), 0, s, ctx->I[l - 1],
                           2 * d.wl[l] /* the reference's row stride of the level above, also when
                                           that level is one pixel wider (ref src/pcd_generator.cpp:82) */,
                           ctx->I[l], d.wl[l], d.hl[l], ctx->ag[l]);
    const int ncell = d.w32 * d.h32;
    hipLaunchKernelGGL(k_fe_hist, dim3(ncell), dim3(FE_BLOCK), 0, s, ctx->ag[0], w, h, d.w32, ctx->ths);
    hipLaunchKernelGGL(k_fe_smooth, dim3(blocks(ncell)), dim3(FE_BLOCK), 0, s, ctx->ths, d.w32, d.h32, ctx->ths_s);
    SelectArgs sa{ctx->ag[0], ctx->ag[1], ctx->ag[2], ctx->ths_s, ctx->map, ctx->ctrl, ctx->blk_cnt, w, h, d.w32, ncell, 0};
    const int nb0 = (((w + 11) / 12) * ((h + 11) / 12) + 3) / 4;   // potential 3: one wave per 12 x 12 block
    hipLaunchKernelGGL(k_fe_select, dim3(nb0), dim3(FE_BLOCK), 0, s, sa);
    hipLaunchKernelGGL(k_fe_decide, dim3(1), dim3(FE_BLOCK), 0, s, ctx->ctrl, ctx->blk_cnt, nb0, 0, (float)ctx->num_want);
    sa.pass = 1;
    const int nb1 = (((w + 3) / 4) * ((h + 3) / 4) + 3) / 4;       // any potential >= 1
    hipLaunchKernelGGL(k_fe_select, dim3(nb1), dim3(FE_BLOCK), 0, s, sa);
    hipLaunchKernelGGL(k_fe_decide, dim3(1), dim3(FE_BLOCK), 0, s, ctx->ctrl, ctx->blk_cnt, nb1, 1, (float)ctx->num_want);
    hipLaunchKernelGGL(k_fe_count<0>, dim3(ctx->nchunks), dim3(FE_BLOCK), 0, s, ctx->map, ctx->depth, np, ctx->ctrl,
                       ctx->cnt);
    hipLaunchKernelGGL(k_fe_subsample, dim3(ctx->nchunks), dim3(FE_BLOCK), 0, s, ctx->map, np, ctx->pattern, ctx->cnt,
                       ctx->ctrl);
    FE_HIP(hipGetLastError());
    // the cloud is emitted at once; a frame that needs the edge top-up (rare: a nearly
    // texture-free image) is noticed at collect(), when the control block has arrived, and
    // emitted again
    return run_emit(ctx, dataset_seq, feature_type);
}

int collect_impl(cvo_fe_ctx *ctx, float *positions, float *features, int capacity, int *num_points, bool to_host);
}

int cvo_fe_collect(cvo_fe_ctx *ctx, float *positions, float *features, int capacity, int *num_points)
{
    cvo_lock::Api api_guard;
    if (!ctx) return CVO_HIP_ERR_INVALID;
    if (!num_points || capacity < 0 || (capacity > 0 && (!positions || !features)))
        return fail(ctx, CVO_HIP_ERR_INVALID, "collect: bad argument");
    return collect_impl(ctx, positions, features, capacity, num_points, true);
}

int cvo_fe_collect_device(cvo_fe_ctx *ctx, const float **d_positions, const float **d_features, int *num_points)
{
    cvo_lock::Api api_guard;
    if (!ctx) return CVO_HIP_ERR_INVALID;
    if (!d_positions || !d_features || !num_points) return fail(ctx, CVO_HIP_ERR_INVALID, "collect_device: bad argument");
    const int rc = collect_impl(ctx, nullptr, nullptr, ctx->cap, num_points, false);
    *d_positions = ctx->pos;
    *d_features = ctx->feat;
    return rc;
}

namespace {
int collect_impl(cvo_fe_ctx *ctx, float *positions, float *features, int capacity, int *num_points, bool to_host)
{
    if (!ctx->pending) return fail(ctx, CVO_HIP_ERR_INVALID, "collect: no frame was submitted");
    ctx->pending = false;
    FE_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    FE_HIP(hipStreamSynchronize(s));
    int rc = CVO_HIP_OK;
    // ref src/pcd_generator.cpp:141-144: fewer than a third of what was asked for?
    const int num_selected = ctx->h_ctrl->in_map;   // (before any top-up: that is what the reference tests)
    const bool canny = num_selected < ctx->num_want / 3;
    if (canny) {
        rc = run_canny(ctx);
        if (!rc) rc = run_emit(ctx, ctx->p_seq, ctx->p_ftype);
        if (rc) return rc;
        FE_HIP(hipStreamSynchronize(s));
    }
    const FeCtrl &c = *ctx->h_ctrl;
    ctx->info.num_selected = num_selected;
    ctx->info.reselected = c.redo;
    ctx->info.pot_used = c.pot[c.redo ? 1 : 0];
    ctx->info.canny_used = canny ? 1 : 0;
    ctx->info.num_points = c.num_points;
    *num_points = c.num_points;
    const int ncopy = std::min(std::min(c.num_points, capacity), ctx->cap);
    if (to_host && ncopy > ctx->copied) {   // an unusually large cloud: fetch the rest
        const size_t from = (size_t)ctx->copied, more = (size_t)(ncopy - ctx->copied);
        FE_HIP(hipMemcpyAsync(ctx->h_pos + from * 3, ctx->pos + from * 3, more * 3 * sizeof(float), hipMemcpyDeviceToHost, s));
        FE_HIP(hipMemcpyAsync(ctx->h_feat + from * 5, ctx->feat + from * 5, more * 5 * sizeof(float), hipMemcpyDeviceToHost, s));
        FE_HIP(hipStreamSynchronize(s));
    }
    if (to_host && ncopy > 0) {
        std::memcpy(positions, ctx->h_pos, (size_t)ncopy * 3 * sizeof(float));
        std::memcpy(features, ctx->h_feat, (size_t)ncopy * 5 * sizeof(float));
    }
    if (c.num_points > ncopy) return fail(ctx, CVO_HIP_ERR_INVALID, "create_pointcloud: more points than capacity");
    return CVO_HIP_OK;
}

}   // namespace

int cvo_fe_create_pointcloud(cvo_fe_ctx *ctx, const uint8_t *img, size_t img_stride, const uint16_t *depth,
                             size_t depth_stride, int dataset_seq, int feature_type, float *positions,
                             float *features, int capacity, int *num_points)
{
    cvo_lock::Api api_guard;
    if (!ctx) return CVO_HIP_ERR_INVALID;
    if (!num_points || capacity < 0 || (capacity > 0 && (!positions || !features)))
        return fail(ctx, CVO_HIP_ERR_INVALID, "create_pointcloud: bad argument");
    const int rc = cvo_fe_submit(ctx, img, img_stride, depth, depth_stride, dataset_seq, feature_type);
    if (rc) return rc;
    return cvo_fe_collect(ctx, positions, features, capacity, num_points);
}

int cvo_fe_get_info(const cvo_fe_ctx *ctx, cvo_fe_info *out)
{
    if (!ctx || !out) return CVO_HIP_ERR_INVALID;
    *out = ctx->info;
    return CVO_HIP_OK;
}

int cvo_fe_read_stage(cvo_fe_ctx *ctx, int stage, void *out, size_t bytes)
{
    cvo_lock::Api api_guard;
    if (!ctx || !out) return CVO_HIP_ERR_INVALID;
    const FeDims &d = ctx->d;
    const size_t np = (size_t)ctx->np;
    const void *src = nullptr;
    size_t need = 0;
    switch (stage) {
    case CVO_FE_STAGE_GRAY: src = ctx->gray; need = np; break;
    case CVO_FE_STAGE_HSV: src = ctx->hsv; need = np * 3; break;
    case CVO_FE_STAGE_MAP: src = ctx->map; need = np * 4; break;
    case CVO_FE_STAGE_AG0: src = ctx->ag[0]; need = np * 4; break;
    case CVO_FE_STAGE_AG1: src = ctx->ag[1]; need = (size_t)d.wl[1] * d.hl[1] * 4; break;
    case CVO_FE_STAGE_AG2: src = ctx->ag[2]; need = (size_t)d.wl[2] * d.hl[2] * 4; break;
    case CVO_FE_STAGE_THS: src = ctx->ths_s; need = (size_t)d.w32 * d.h32 * 4; break;
    case CVO_FE_STAGE_DX0: src = ctx->dx0; need = np * 4; break;
    case CVO_FE_STAGE_DY0: src = ctx->dy0; need = np * 4; break;
    case CVO_FE_STAGE_EDGES: src = ctx->edges; need = np; break;
    default: return fail(ctx, CVO_HIP_ERR_INVALID, "read_stage: unknown stage");
    }
    if (bytes < need) return fail(ctx, CVO_HIP_ERR_INVALID, "read_stage: buffer too small");
    FE_HIP(hipSetDevice(ctx->device));
    FE_HIP(hipStreamSynchronize(ctx->stream));
    if (stage == CVO_FE_STAGE_HSV) {   // packed words on the device, 3 bytes per pixel for the caller
        uint32_t *tmp = new (std::nothrow) uint32_t[np];
        if (!tmp) return fail(ctx, CVO_HIP_ERR_NOMEM, "read_stage");
        const hipError_t e = hipMemcpy(tmp, src, np * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess)
            for (size_t i = 0; i < np; ++i) {
                ((uint8_t *)out)[3 * i] = (uint8_t)(tmp[i] & 255u);
                ((uint8_t *)out)[3 * i + 1] = (uint8_t)((tmp[i] >> 8) & 255u);
                ((uint8_t *)out)[3 * i + 2] = (uint8_t)((tmp[i] >> 16) & 255u);
            }
        delete[] tmp;
        if (e != hipSuccess) return fail(ctx, CVO_HIP_ERR_HIP, "read_stage copy", e);
        return CVO_HIP_OK;
    }
    FE_HIP(hipMemcpy(out, src, need, hipMemcpyDeviceToHost));
    return CVO_HIP_OK;
}

}   // extern "C"
