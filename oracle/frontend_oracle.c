/* frontend_oracle.c -- CPU restatement of the reference's RGB-D front end
 * (SURVEY 8 f3): image -> semi-dense coloured point cloud.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path may call this; it is the
 * checker the HIP front end (cvo-rgbd_amd/csrc/cvo_frontend.hip) is compared with.
 *
 * PARITY UNPINNED: the reference has no tests or vectors for this path, its build
 * needs OpenCV / Eigen / PCL (absent here), and no RGB-D frame ships with it.  The
 * parts the reference writes itself are restated line by line:
 *   pyramid + gradients      ref cpp/rkhs_registration/src/pcd_generator.cpp:33-129
 *   selection + Canny top-up ref src/pcd_generator.cpp:131-176
 *   back-projection          ref src/pcd_generator.cpp:233-327
 *   features                 ref src/pcd_generator.cpp:329-385
 *   pixel selector           ref thirdparty/PixelSelector2.cpp:33-46 (pattern),
 *                            :58-67 (quantile), :70-136 (histograms), :137-236
 *                            (makeMaps), :240-437 (select)
 * The OpenCV calls it makes (cvtColor RGB2GRAY / RGB2HSV on 8-bit data, blur 3x3,
 * Canny(0, 25, 3), ref src/pcd_generator.cpp:151-152,389-390) are restated from
 * OpenCV's published 8-bit fixed-point definitions; see each function.
 *
 * Where the reference reads memory it never wrote (gradients of the first / last
 * image row, `new[]` without a fill, ref src/pcd_generator.cpp:43-44,96) this
 * restatement reads zeros.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define FE_LEVELS 3

/* the selector's random bytes: srand(3141592), rand() & 0xFF per pixel
 * (ref thirdparty/PixelSelector2.cpp:35-37) -- the C library's own generator */
void fe_random_pattern(int n, uint8_t *out)
{
    srand(3141592);
    for (int i = 0; i < n; ++i) out[i] = (uint8_t)(rand() & 0xFF);
}

/* cv::cvtColor(COLOR_RGB2GRAY), 8-bit: channel 0 is taken as R whatever the
 * file held (cv::imread delivers BGR; the reference passes it as RGB,
 * ref src/pcd_generator.cpp:389).  OpenCV: 14-bit fixed point, R2Y 4899, G2Y 9617,
 * B2Y 1868, rounded. */
void fe_gray(const uint8_t *img, int w, int h, uint8_t *gray)
{
    for (int i = 0; i < w * h; ++i) {
        const int c0 = img[3 * i], c1 = img[3 * i + 1], c2 = img[3 * i + 2];
        gray[i] = (uint8_t)((c0 * 4899 + c1 * 9617 + c2 * 1868 + (1 << 13)) >> 14);
    }
}

/* cv::cvtColor(COLOR_RGB2HSV), 8-bit, hue range 180 (ref src/pcd_generator.cpp:390):
 * OpenCV's 12-bit fixed-point form with its two reciprocal tables. */
static int fe_round_half_even(double v) { return (int)lrint(v); }

void fe_hsv(const uint8_t *img, int w, int h, uint8_t *hsv)
{
    int sdiv[256], hdiv[256];
    sdiv[0] = hdiv[0] = 0;
    for (int i = 1; i < 256; ++i) {
        sdiv[i] = fe_round_half_even((255 << 12) / (1. * i));
        hdiv[i] = fe_round_half_even((180 << 12) / (6. * i));
    }
    for (int i = 0; i < w * h; ++i) {
        const int r = img[3 * i], g = img[3 * i + 1], b = img[3 * i + 2];
        int v = b, vmin = b;
        if (g > v) v = g;
        if (r > v) v = r;
        if (g < vmin) vmin = g;
        if (r < vmin) vmin = r;
        const int diff = v - vmin;
        const int vr = v == r ? -1 : 0, vg = v == g ? -1 : 0;
        const int s = (diff * sdiv[v] + (1 << 11)) >> 12;
        int hh = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
        hh = (hh * hdiv[diff] + (1 << 11)) >> 12;
        hh += hh < 0 ? 180 : 0;
        hsv[3 * i] = (uint8_t)(hh < 0 ? 0 : (hh > 255 ? 255 : hh));
        hsv[3 * i + 1] = (uint8_t)s;
        hsv[3 * i + 2] = (uint8_t)v;
    }
}

/* ref src/pcd_generator.cpp:33-129: three levels; level l > 0 is the 2x2 mean of
 * level l-1; central differences on the flattened image (so the first / last
 * column take their neighbour from the adjacent row, as the reference does).
 * I[l], ag[l]: w_l*h_l floats; dx0, dy0: level 0 only (the only ones read later). */
void fe_pyramid(const uint8_t *gray, int w, int h, float *I[FE_LEVELS], float *dx0, float *dy0,
                float *ag[FE_LEVELS])
{
    int wl = w, hl = h;
    for (int i = 0; i < w * h; ++i) I[0][i] = (float)gray[i];
    for (int lvl = 0; lvl < FE_LEVELS; ++lvl) {
        float *cur = I[lvl];
        if (lvl > 0) {
            const float *prev = I[lvl - 1];
            const int pw = wl * 2;
            for (int y = 0; y < hl; ++y)
                for (int x = 0; x < wl; ++x)
                    cur[x + y * wl] = 0.25f * (((prev[2 * x + 2 * y * pw] + prev[2 * x + 1 + 2 * y * pw]) +
                                                prev[2 * x + 2 * y * pw + pw]) +
                                               prev[2 * x + 1 + 2 * y * pw + pw]);
        }
        memset(ag[lvl], 0, sizeof(float) * (size_t)wl * hl);
        if (lvl == 0) {
            memset(dx0, 0, sizeof(float) * (size_t)wl * hl);
            memset(dy0, 0, sizeof(float) * (size_t)wl * hl);
        }
        for (int idx = wl; idx < wl * (hl - 1); ++idx) {
            float dx = 0.5f * (cur[idx + 1] - cur[idx - 1]);
            float dy = 0.5f * (cur[idx + wl] - cur[idx - wl]);
            if (!isfinite(dx)) dx = 0;
            if (!isfinite(dy)) dy = 0;
            if (lvl == 0) { dx0[idx] = dx; dy0[idx] = dy; }
            ag[lvl][idx] = dx * dx + dy * dy;
        }
        wl /= 2;
        hl /= 2;
    }
}

/* ref thirdparty/PixelSelector2.cpp:58-67 (bins past the 50 the caller clears are
 * taken as empty) */
static int fe_hist_quantile(const int *hist, float below)
{
    int th = (int)(hist[0] * below + 0.5f);
    for (int i = 0; i < 90; ++i) {
        th -= (i + 1 < 50) ? hist[i + 1] : 0;
        if (th < 0) return i;
    }
    return 90;
}

/* ref thirdparty/PixelSelector2.cpp:70-136: per 32x32 cell the median gradient
 * magnitude + 7, then the squared 3x3 mean.  ths_smoothed: (w/32)*(h/32) floats. */
void fe_thresholds(const float *ag0, int w, int h, float *ths_smoothed)
{
    const int w32 = w / 32, h32 = h / 32;
    float *ths = (float *)calloc((size_t)w32 * h32 + 1, sizeof(float));
    for (int y = 0; y < h32; ++y)
        for (int x = 0; x < w32; ++x) {
            int hist[50];
            memset(hist, 0, sizeof(hist));
            for (int j = 0; j < 32; ++j)
                for (int i = 0; i < 32; ++i) {
                    const int it = i + 32 * x, jt = j + 32 * y;
                    if (it > w - 2 || jt > h - 2 || it < 1 || jt < 1) continue;
                    int g = (int)sqrtf(ag0[it + jt * w]);
                    if (g > 48) g = 48;
                    hist[g + 1]++;
                    hist[0]++;
                }
            ths[x + y * w32] = (float)(fe_hist_quantile(hist, 0.5f) + 7);
        }
    for (int y = 0; y < h32; ++y)
        for (int x = 0; x < w32; ++x) {
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
            ths_smoothed[x + y * w32] = (sum / num) * (sum / num);
        }
    free(ths);
}

/* ref thirdparty/PixelSelector2.cpp:240-437 with selectDirectionDistribution off
 * (PixelSelector2.h:31): the score of a pixel is its gradient magnitude at the
 * level that admits it, the random directions are never used.  n[3]: pixels
 * chosen at level 0, 1, 2. */
static void fe_select(const float *ag0, const float *ag1, const float *ag2, const float *ths, int w, int h,
                      int pot, float thFactor, float *map_out, int n[3])
{
    const int w1 = w / 2, w2 = w / 4, w32 = w / 32, ncell = (w / 32) * (h / 32);
    const float dw1 = 0.75f, dw2 = dw1 * dw1;
    memset(map_out, 0, sizeof(float) * (size_t)w * h);
    int n3 = 0, n2 = 0, n4 = 0;
    for (int y4 = 0; y4 < h; y4 += 4 * pot)
        for (int x4 = 0; x4 < w; x4 += 4 * pot) {
            const int my3 = 4 * pot < h - y4 ? 4 * pot : h - y4;
            const int mx3 = 4 * pot < w - x4 ? 4 * pot : w - x4;
            int bestIdx4 = -1; float bestVal4 = 0;
            for (int y3 = 0; y3 < my3; y3 += 2 * pot)
                for (int x3 = 0; x3 < mx3; x3 += 2 * pot) {
                    const int x34 = x3 + x4, y34 = y3 + y4;
                    const int my2 = 2 * pot < h - y34 ? 2 * pot : h - y34;
                    const int mx2 = 2 * pot < w - x34 ? 2 * pot : w - x34;
                    int bestIdx3 = -1; float bestVal3 = 0;
                    for (int y2 = 0; y2 < my2; y2 += pot)
                        for (int x2 = 0; x2 < mx2; x2 += pot) {
                            const int x234 = x2 + x34, y234 = y2 + y34;
                            const int my1 = pot < h - y234 ? pot : h - y234;
                            const int mx1 = pot < w - x234 ? pot : w - x234;
                            int bestIdx2 = -1; float bestVal2 = 0;
                            for (int y1 = 0; y1 < my1; ++y1)
                                for (int x1 = 0; x1 < mx1; ++x1) {
                                    const int xf = x1 + x234, yf = y1 + y234;
                                    const int idx = xf + w * yf;
                                    if (xf < 4 || xf >= w - 5 || yf < 4 || yf > h - 4) continue;
                                    int cell = (xf >> 5) + (yf >> 5) * w32;
                                    if (cell >= ncell) cell = ncell - 1;   /* (reference: past the end) */
                                    const float th0 = ths[cell];
                                    const float th1 = th0 * dw1;
                                    const float th2 = th1 * dw2;
                                    const float a0 = ag0[idx];
                                    if (a0 > th0 * thFactor) {
                                        if (a0 > bestVal2) { bestVal2 = a0; bestIdx2 = idx; bestIdx3 = -2; bestIdx4 = -2; }
                                    }
                                    if (bestIdx3 == -2) continue;
                                    const float a1 = ag1[(int)(xf * 0.5f + 0.25f) + (int)(yf * 0.5f + 0.25f) * w1];
                                    if (a1 > th1 * thFactor) {
                                        if (a1 > bestVal3) { bestVal3 = a1; bestIdx3 = idx; bestIdx4 = -2; }
                                    }
                                    if (bestIdx4 == -2) continue;
                                    const float a2 = ag2[(int)(xf * 0.25f + 0.125f) + (int)(yf * 0.25f + 0.125f) * w2];
                                    if (a2 > th2 * thFactor) {
                                        if (a2 > bestVal4) { bestVal4 = a2; bestIdx4 = idx; }
                                    }
                                }
                            if (bestIdx2 > 0) { map_out[bestIdx2] = 1; bestVal3 = 1e10f; n2++; }
                        }
                    if (bestIdx3 > 0) { map_out[bestIdx3] = 2; bestVal4 = 1e10f; n3++; }
                }
            if (bestIdx4 > 0) { map_out[bestIdx4] = 4; n4++; }
        }
    n[0] = n2; n[1] = n3; n[2] = n4;
}

/* ref thirdparty/PixelSelector2.cpp:137-236 (one recursion allowed, thFactor 1,
 * a selector starts every frame at potential 3, ref :39 and pcd_generator.cpp:140).
 * Returns the number of pixels left in the map; *pot_used the potential of the
 * pass that produced it. */
int fe_make_maps(const float *ag0, const float *ag1, const float *ag2, const float *ths, const uint8_t *pattern,
                 int w, int h, float density, float *map_out, int *pot_used)
{
    int pot = 3, recursions = 1;
    float numHave, quotia;
    const float numWant = density;
    for (;;) {
        int n[3];
        fe_select(ag0, ag1, ag2, ths, w, h, pot, 1.0f, map_out, n);
        numHave = (float)(n[0] + n[1] + n[2]);
        quotia = numWant / numHave;
        const float K = numHave * (float)(pot + 1) * (float)(pot + 1);
        int ideal = (int)(sqrtf(K / numWant) - 1);
        if (ideal < 1) ideal = 1;
        if (recursions > 0 && quotia > 1.25 && pot > 1) {
            if (ideal >= pot) ideal = pot - 1;
            pot = ideal; recursions--;
            continue;
        }
        if (recursions > 0 && quotia < 0.25) {
            if (ideal <= pot) ideal = pot + 1;
            pot = ideal; recursions--;
            continue;
        }
        break;
    }
    if (pot_used) *pot_used = pot;
    int numHaveSub = (int)numHave;
    if ((double)quotia < 0.95) {
        const unsigned char charTH = (unsigned char)(255 * quotia);
        int rn = 0;
        for (int i = 0; i < w * h; ++i)
            if (map_out[i] != 0) {
                if (pattern[rn] > charTH) { map_out[i] = 0; numHaveSub--; }
                rn++;
            }
    }
    return numHaveSub;
}

/* cv::blur(src, dst, Size(3,3)) on 8-bit data: box mean, BORDER_REFLECT_101,
 * rounded to nearest (ref src/pcd_generator.cpp:151) */
static int fe_reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}

void fe_blur3(const uint8_t *src, int w, int h, uint8_t *dst)
{
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int s = 0;
            for (int j = -1; j <= 1; ++j)
                for (int i = -1; i <= 1; ++i) s += src[fe_reflect101(y + j, h) * w + fe_reflect101(x + i, w)];
            dst[y * w + x] = (uint8_t)fe_round_half_even(s * (1.0 / 9.0));
        }
}

/* cv::Canny(img, edges, 0, 25, 3) (ref src/pcd_generator.cpp:152): 3x3 Sobel with
 * replicated borders, L1 magnitude, non-maximum suppression with OpenCV's 15-bit
 * tan(22.5 deg) tests, hysteresis over 8-neighbourhoods.  The result: local maxima
 * above `low` that are connected to a local maximum above `high`. */
void fe_canny(const uint8_t *src, int w, int h, int low, int high, uint8_t *edges)
{
    const size_t np = (size_t)w * h;
    int *mag = (int *)calloc((size_t)(w + 2) * (h + 2), sizeof(int));
    int16_t *gx = (int16_t *)malloc(np * sizeof(int16_t)), *gy = (int16_t *)malloc(np * sizeof(int16_t));
    uint8_t *st = (uint8_t *)malloc(np);   /* 0 candidate, 1 not an edge, 2 edge */
    int *stack = (int *)malloc(np * sizeof(int));
    const int ms = w + 2;
#define PX(xx, yy) ((int)src[((yy) < 0 ? 0 : ((yy) >= h ? h - 1 : (yy))) * w + ((xx) < 0 ? 0 : ((xx) >= w ? w - 1 : (xx)))])
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const int dx = (PX(x + 1, y - 1) + 2 * PX(x + 1, y) + PX(x + 1, y + 1)) -
                           (PX(x - 1, y - 1) + 2 * PX(x - 1, y) + PX(x - 1, y + 1));
            const int dy = (PX(x - 1, y + 1) + 2 * PX(x, y + 1) + PX(x + 1, y + 1)) -
                           (PX(x - 1, y - 1) + 2 * PX(x, y - 1) + PX(x + 1, y - 1));
            gx[y * w + x] = (int16_t)dx;
            gy[y * w + x] = (int16_t)dy;
            mag[(y + 1) * ms + x + 1] = abs(dx) + abs(dy);
        }
#undef PX
    const int TG22 = (int)(0.4142135623730950488016887242097 * (1 << 15) + 0.5);
    int top = 0;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const int *m0 = mag + (y + 1) * ms + x + 1;
            const int m = m0[0];
            int is_max = 0;
            if (m > low) {
                const int xs = gx[y * w + x], ys = gy[y * w + x];
                const int ax = abs(xs), ay = abs(ys) << 15;
                const int tg22x = ax * TG22;
                if (ay < tg22x) {
                    is_max = m > m0[-1] && m >= m0[1];
                } else {
                    const int tg67x = tg22x + (ax << 16);
                    if (ay > tg67x) {
                        is_max = m > m0[-ms] && m >= m0[ms];
                    } else {
                        const int s = (xs ^ ys) < 0 ? -1 : 1;
                        is_max = m > m0[-ms - s] && m > m0[ms + s];
                    }
                }
            }
            if (!is_max) st[y * w + x] = 1;
            else if (m > high) { st[y * w + x] = 2; stack[top++] = y * w + x; }
            else st[y * w + x] = 0;
        }
    while (top > 0) {
        const int p = stack[--top], px = p % w, py = p / w;
        for (int j = -1; j <= 1; ++j)
            for (int i = -1; i <= 1; ++i) {
                const int qx = px + i, qy = py + j;
                if (qx < 0 || qx >= w || qy < 0 || qy >= h) continue;
                if (st[qy * w + qx] == 0) { st[qy * w + qx] = 2; stack[top++] = qy * w + qx; }
            }
    }
    for (size_t i = 0; i < np; ++i) edges[i] = st[i] == 2 ? 255 : 0;
    free(mag); free(gx); free(gy); free(st); free(stack);
}

/* ref src/pcd_generator.cpp:143-175: when the selector kept fewer than a third of
 * what was asked for, every 8x8 block gives its first edge pixel that is not in
 * the map yet (rows first; a row is left at the first edge pixel that was free). */
void fe_canny_topup(const uint8_t *gray, int w, int h, float *map)
{
    uint8_t *tmp = (uint8_t *)malloc((size_t)w * h), *edge = (uint8_t *)malloc((size_t)w * h);
    fe_blur3(gray, w, h, tmp);
    fe_canny(tmp, w, h, 0, 25, edge);
    const int bs = 8;
    for (int y = 0; y < h; y += bs)
        for (int x = 0; x < w; x += bs) {
            int got = 0;
            for (int j = 0; j < bs && !got; ++j)
                for (int i = 0; i < bs; ++i) {
                    if (x + i >= w || y + j >= h) continue;   /* (reference: reads past the image) */
                    if (edge[(y + j) * w + x + i] != 0 && map[(y + j) * w + x + i] == 0) {
                        map[(y + j) * w + x + i] = 1;
                        got = 1;
                        break;
                    }
                }
        }
    free(tmp); free(edge);
}

/* camera table, ref src/pcd_generator.cpp:241-295: {scale, fx, fy, cx, cy} */
void fe_camera(int dataset_seq, float cam[5])
{
    static const float tab[6][5] = {{1000.0f, 616.368f, 616.745f, 319.935f, 243.639f},
                                    {5000.0f, 517.3f, 516.5f, 318.6f, 255.3f},
                                    {5000.0f, 520.9f, 521.0f, 325.1f, 249.7f},
                                    {5000.0f, 535.4f, 539.2f, 320.1f, 247.6f},
                                    {2000.0f, 718.856f, 718.856f, 607.1928f, 185.2157f},
                                    {2000.0f, 707.0912f, 707.0912f, 601.8873f, 183.1104f}};
    const int k = (dataset_seq >= 0 && dataset_seq <= 5) ? dataset_seq : 0;
    for (int q = 0; q < 5; ++q) cam[q] = tab[k][q];
}

/* The whole front end, ref src/pcd_generator.cpp:387-420 (load_image +
 * create_pointcloud).  img: h*w*3 bytes as decoded (cv::imread order), depth:
 * h*w uint16.  positions: cap*3, features: cap*5 ROW-major.  map_out (w*h floats,
 * optional) receives the selection map.  Returns the number of points (those
 * beyond `cap` are counted, not stored). */
int fe_create_pointcloud(const uint8_t *img, const uint16_t *depth, int w, int h, int dataset_seq,
                         int feature_type, int num_want, float *positions, float *features, int cap,
                         float *map_out, int *num_selected_out)
{
    const size_t np = (size_t)w * h;
    uint8_t *gray = (uint8_t *)malloc(np), *hsv = (uint8_t *)malloc(np * 3), *pattern = (uint8_t *)malloc(np);
    float *I[FE_LEVELS], *ag[FE_LEVELS];
    int wl = w, hl = h;
    for (int l = 0; l < FE_LEVELS; ++l) {
        I[l] = (float *)calloc((size_t)wl * hl + 1, sizeof(float));
        ag[l] = (float *)calloc((size_t)wl * hl + 1, sizeof(float));
        wl /= 2; hl /= 2;
    }
    float *dx0 = (float *)calloc(np, sizeof(float)), *dy0 = (float *)calloc(np, sizeof(float));
    float *map = (float *)calloc(np, sizeof(float));
    float *ths = (float *)calloc((size_t)(w / 32) * (h / 32) + 1, sizeof(float));
    fe_gray(img, w, h, gray);
    fe_hsv(img, w, h, hsv);
    fe_pyramid(gray, w, h, I, dx0, dy0, ag);
    fe_thresholds(ag[0], w, h, ths);
    fe_random_pattern((int)np, pattern);
    int pot = 0;
    const int num_selected = fe_make_maps(ag[0], ag[1], ag[2], ths, pattern, w, h, (float)num_want, map, &pot);
    if (num_selected < num_want / 3) fe_canny_topup(gray, w, h, map);
    float cam[5];
    fe_camera(dataset_seq, cam);
    int idx = 0;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const uint16_t dep = depth[y * w + x];
            if (map[y * w + x] == 0 || dep == 0) continue;
            if (idx < cap) {
                const float z = (float)dep / cam[0];
                positions[3 * idx + 2] = z;
                positions[3 * idx + 0] = ((float)x - cam[3]) * z / cam[1];
                positions[3 * idx + 1] = ((float)y - cam[4]) * z / cam[2];
                float *f = features + 5 * (size_t)idx;
                const size_t p = (size_t)y * w + x;
                if (feature_type == 0) {
                    f[0] = (float)(hsv[3 * p] / 180.0);
                    f[1] = (float)(hsv[3 * p + 1] / 255.0);
                    f[2] = (float)(hsv[3 * p + 2] / 255.0);
                    f[3] = (float)(dx0[p] / 255.0 * 2);
                    f[4] = (float)(dy0[p] / 255.0 * 2);
                } else {
                    f[0] = (float)img[3 * p];
                    f[1] = (float)img[3 * p + 1];
                    f[2] = (float)img[3 * p + 2];
                    f[3] = dx0[p];
                    f[4] = dy0[p];
                }
            }
            ++idx;
        }
    if (map_out) memcpy(map_out, map, np * sizeof(float));
    if (num_selected_out) *num_selected_out = num_selected;
    for (int l = 0; l < FE_LEVELS; ++l) { free(I[l]); free(ag[l]); }
    free(gray); free(hsv); free(pattern); free(dx0); free(dy0); free(map); free(ths);
    return idx;
}
