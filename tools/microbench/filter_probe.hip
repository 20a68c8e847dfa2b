// filter_probe.hip -- where does a k_filter block spend its time?
// Includes the production kernel source and runs it on a synthetic cloud pair
// with per-wave s_memtime stamps (FilterArgs::dbg).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../include
//        -I../../cvo-rgbd_amd/csrc filter_probe.hip -o filter_probe
#define CVO_FILTER_PROBE 1
#include "../../cvo-rgbd_amd/csrc/cvo_kernels.hip"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

using namespace cvo_dev;

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 10000;
    const float ell = argc > 2 ? atof(argv[2]) : 0.03f;
    const int jt = argc > 3 ? atoi(argv[3]) : 400;
    std::mt19937 rng(1);
    std::uniform_real_distribution<float> U(-0.78f, 0.78f);
    std::vector<float4> ha(n), hb(n);
    auto surf = [&](float x, float y) { return 1.3f + 0.25f * sinf(2.1f * x + 0.3f) * cosf(1.7f * y) + 0.15f * sinf(4.3f * y); };
    for (auto *v : {&ha, &hb}) {
        for (int i = 0; i < n; ++i) { float x = U(rng), y = U(rng); (*v)[i] = make_float4(x, y, surf(x, y), 0.f); }
        // crude spatial sort (by cell) so that tiles are compact like the Morton order
        std::sort(v->begin(), v->end(), [](const float4 &p, const float4 &q) {
            auto mort = [](float x, float y) { unsigned a = (unsigned)((x + 0.8f) * 600.f), b = (unsigned)((y + 0.8f) * 600.f), m = 0; for (int k = 0; k < 10; ++k) m |= ((a >> k) & 1u) << (2 * k) | ((b >> k) & 1u) << (2 * k + 1); return m; }; unsigned cp = mort(p.x, p.y), cq = mort(q.x, q.y);
            return cp < cq; });
    }
    auto spheres = [&](const std::vector<float4> &v) {
        std::vector<float4> sp((v.size() + SEG - 1) / SEG);
        for (size_t g = 0; g < sp.size(); ++g) {
            const size_t s0 = g * SEG, s1 = std::min(v.size(), s0 + SEG);
            float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
            for (size_t q = s0; q < s1; ++q) { const float p[3] = {v[q].x, v[q].y, v[q].z};
                for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); } }
            const float c[3] = {0.5f * (lo[0] + hi[0]), 0.5f * (lo[1] + hi[1]), 0.5f * (lo[2] + hi[2])};
            double r2 = 0; for (size_t q = s0; q < s1; ++q) { double d = 0; const float p[3] = {v[q].x, v[q].y, v[q].z};
                for (int a = 0; a < 3; ++a) d += ((double)p[a] - c[a]) * ((double)p[a] - c[a]); r2 = std::max(r2, d); }
            sp[g] = make_float4(c[0], c[1], c[2], (float)(sqrt(r2) * 1.00001 + 1e-6));
        }
        return sp; };
    const std::vector<float4> sa = spheres(ha), sb = spheres(hb);
    float4 *dsa, *dsb;
    hipMalloc(&dsa, sa.size() * sizeof(float4)); hipMalloc(&dsb, sb.size() * sizeof(float4));
    hipMemcpy(dsa, sa.data(), sa.size() * sizeof(float4), hipMemcpyHostToDevice);
    hipMemcpy(dsb, sb.data(), sb.size() * sizeof(float4), hipMemcpyHostToDevice);
    { double rs = 0; for (auto &q : sa) rs += q.w; printf("mean segment radius %.3f m\n", rs / sa.size()); }
    float4 *da, *db; DevState *st; TileEntry *cand; long long *dbg;
    hipMalloc(&da, n * sizeof(float4)); hipMalloc(&db, n * sizeof(float4));
    hipMemcpy(da, ha.data(), n * sizeof(float4), hipMemcpyHostToDevice);
    hipMemcpy(db, hb.data(), n * sizeof(float4), hipMemcpyHostToDevice);
    DevState h{}; h.R[0] = h.R[4] = h.R[8] = 1.f; h.ell = ell; h.center[2] = 1.3f; h.xmax = h.y0max = 1.3f;
    DevParams prm{}; prm.c = prm.d = 7.f; prm.c_ell = 200.f; prm.log_sp_s2 = logf(0.8f); prm.tau_c = 1e9f;
    prepare_iteration(&h, prm);
    hipMalloc(&st, sizeof(DevState)); hipMemcpy(st, &h, sizeof(h), hipMemcpyHostToDevice);
    const uint32_t subcap = 1u << 12;
    hipMalloc(&cand, (size_t)subcap * NSUB * sizeof(TileEntry));
    const int tiles = (n + ROWS_PER_TILE - 1) / ROWS_PER_TILE, chunks = (n + jt - 1) / jt;
    const size_t nw = (size_t)tiles * chunks * 4;
    hipMalloc(&dbg, nw * 8 * sizeof(long long));
    FilterArgs a{}; a.pos_a = da; a.pos_b = db; a.seg_a = dsa; a.seg_b = dsb; a.st = st; a.tiles = cand; a.subcap = subcap; a.list = LIST_XY;
    a.row_lo = 0; a.row_hi = n; a.nb = n; a.jt = jt; a.tf_a = 0; a.tf_b = 1; a.check_done = 1; a.dbg = dbg;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(&st->sub[0][0], 0, sizeof(h.sub)); hipMemset(st->cnt, 0, sizeof(h.cnt)); hipMemset(dbg, 0, nw * 8 * sizeof(long long));
        hipDeviceSynchronize();
        hipEventRecord(e0);
        launch_filter(a, dim3(chunks, tiles), 0);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> hd(nw * 8); size_t nw_live = nw;
        hipMemcpy(hd.data(), dbg, hd.size() * sizeof(long long), hipMemcpyDeviceToHost);
        hipMemcpy(&h, st, sizeof(h), hipMemcpyDeviceToHost);
        unsigned long long tot = 0; for (int q = 0; q < NSUB; ++q) tot += h.sub[0][q];
        size_t culled = 0; double clife = 0;
        for (size_t w = 0; w < nw; ++w) if (hd[w * 8 + 3] == 0) { culled++; }
        printf("culled waves %zu of %zu (avg life %.0f ticks)\n", culled, (size_t)nw, culled ? clife / culled : 0.0);
        {   size_t k = 0;   // keep only the surviving waves for the statistics below
            for (size_t w = 0; w < nw; ++w) if (hd[w * 8 + 3] != 0) { for (int q = 0; q < 8; ++q) hd[k * 8 + q] = hd[w * 8 + q]; ++k; }
            nw_live = k; }
        if (nw_live == 0) continue;
        const size_t nw_outer = nw; (void)nw_outer;
        {
        const size_t nw = nw_live;
        long long tmin = hd[0], tmax = 0; double pro = 0, loop = 0, tail = 0, life = 0;
        for (size_t w = 0; w < nw; ++w) {
            const long long *o = &hd[w * 8];
            tmin = std::min(tmin, o[0]); tmax = std::max(tmax, o[3]);
            pro += o[1] - o[0]; loop += o[2] - o[1]; tail += o[3] - o[2]; life += o[3] - o[0];
        }
        // when did waves start relative to the first one?
        std::vector<long long> starts(nw); for (size_t w = 0; w < nw; ++w) starts[w] = hd[w * 8] - tmin;
        std::sort(starts.begin(), starts.end());
        if (rep == 2) {   // census: concurrency per CU from the stamps (clocks agree inside an XCD)
            struct W { long long s, e; int cu; };
            std::vector<W> ws(nw);
            for (size_t w = 0; w < nw; ++w) {
                const long long *o = &hd[w * 8];
                const int hw = (int)o[4], xcc = (int)o[5] & 15;
                const int cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
                ws[w] = {o[0], o[3], ((xcc * 8 + se) * 2 + sh) * 16 + cu};
            }
            int maxcu = 0; for (auto &w : ws) maxcu = std::max(maxcu, w.cu);
            std::vector<int> cnt(maxcu + 1, 0), peak(maxcu + 1, 0);
            std::vector<long long> first(maxcu + 1, (long long)9e18), last(maxcu + 1, 0);
            for (auto &w : ws) { cnt[w.cu]++; first[w.cu] = std::min(first[w.cu], w.s); last[w.cu] = std::max(last[w.cu], w.e); }
            for (int c = 0; c <= maxcu; ++c) {
                if (!cnt[c]) continue;
                std::vector<std::pair<long long,int>> ev;
                for (auto &w : ws) if (w.cu == c) { ev.push_back({w.s, 1}); ev.push_back({w.e, -1}); }
                std::sort(ev.begin(), ev.end());
                int cur = 0; for (auto &e : ev) { cur += e.second; peak[c] = std::max(peak[c], cur); }
            }
            {   // per XCD: when do waves start / end relative to the first start on that XCD
                for (int x = 0; x < 8; ++x) {
                    std::vector<long long> st_, en_;
                    for (size_t w = 0; w < nw; ++w) { const long long *o = &hd[w * 8]; if (((int)o[5] & 15) == x) { st_.push_back(o[0]); en_.push_back(o[3]); } }
                    if (st_.empty()) continue;
                    std::sort(st_.begin(), st_.end()); std::sort(en_.begin(), en_.end());
                    const long long t0 = st_[0];
                    printf("xcd %d: %zu waves; start p50 %lld p90 %lld max %lld | end min %lld p50 %lld max %lld ticks\n", x, st_.size(),
                           st_[st_.size() / 2] - t0, st_[st_.size() * 9 / 10] - t0, st_.back() - t0, en_[0] - t0, en_[en_.size() / 2] - t0, en_.back() - t0);
                }
            }
            {   double dm = 0, dw = 0; long long wmin = (long long)9e18, wmax = 0;
                for (size_t w = 0; w < nw; ++w) { const long long *o = &hd[w * 8]; dm += o[3] - o[0]; dw += o[7] - o[6]; wmin = std::min(wmin, o[6]); wmax = std::max(wmax, o[7]); }
                std::vector<long long> ws_, we_; for (size_t w = 0; w < nw; ++w) { ws_.push_back(hd[w * 8 + 6] - wmin); we_.push_back(hd[w * 8 + 7] - wmin); }
                std::sort(ws_.begin(), ws_.end()); std::sort(we_.begin(), we_.end());
                printf("wave START (10 ns units): p10 %lld p50 %lld p90 %lld max %lld | wave END: min %lld p10 %lld p50 %lld p90 %lld max %lld\n",
                       ws_[nw / 10], ws_[nw / 2], ws_[nw * 9 / 10], ws_[nw - 1], we_[0], we_[nw / 10], we_[nw / 2], we_[nw * 9 / 10], we_[nw - 1]);
                printf("shader clock during the kernel: %.0f MHz; first wave start -> last wave end: %.2f us (100 MHz wall clock)\n", dm / dw * 100.0, (wmax - wmin) / 100.0);
            }
            {   std::vector<int> hist(40, 0); std::vector<double> spanby(40, 0);
                for (int c = 0; c <= maxcu; ++c) if (cnt[c]) { hist[std::min(cnt[c], 39)]++; spanby[std::min(cnt[c], 39)] += last[c] - first[c]; }
                printf("waves per CU histogram (waves: #CUs, avg busy ticks):");
                for (int k = 0; k < 40; ++k) if (hist[k]) printf(" %d: %d, %.0f |", k, hist[k], spanby[k] / hist[k]);
                printf("\n");
            }
            {   std::vector<long long> pl, ll, tl;
                for (size_t w = 0; w < nw; ++w) { const long long *o = &hd[w * 8]; pl.push_back(o[1] - o[0]); ll.push_back(o[2] - o[1]); tl.push_back(o[3] - o[2]); }
                std::sort(pl.begin(), pl.end()); std::sort(ll.begin(), ll.end()); std::sort(tl.begin(), tl.end());
                auto pc = [&](std::vector<long long> &v, double f) { return v[(size_t)(f * (v.size() - 1))]; };
                printf("prologue ticks p10 %lld p50 %lld p90 %lld max %lld | loop p10 %lld p50 %lld p90 %lld p99 %lld max %lld | tail p50 %lld p99 %lld max %lld\n",
                       pc(pl, .1), pc(pl, .5), pc(pl, .9), pc(pl, 1), pc(ll, .1), pc(ll, .5), pc(ll, .9), pc(ll, .99), pc(ll, 1), pc(tl, .5), pc(tl, .99), pc(tl, 1));
            }
            int used = 0; double avgw = 0, avgpeak = 0, avgspan = 0;
            for (int c = 0; c <= maxcu; ++c) if (cnt[c]) { used++; avgw += cnt[c]; avgpeak += peak[c]; avgspan += last[c] - first[c]; }
            printf("census: %d CU ids used, waves/CU avg %.1f, peak concurrent waves/CU avg %.1f, CU busy span avg %.0f ticks\n", used, avgw / used, avgpeak / used, avgspan / used);
        }
        printf("n %d ell %.2f jt %d grid %dx%d: kernel %.1f us, first start -> last exit %lld ticks, per wave: prologue %.0f loop %.0f tail %.0f life %.0f ticks; start p50 %lld p90 %lld max %lld; candidates %llu ovf %u\n",
               n, ell, jt, chunks, tiles, ms * 1e3, tmax - tmin, pro / nw, loop / nw, tail / nw, life / nw,
               starts[nw / 2], starts[nw * 9 / 10], starts[nw - 1], tot, h.cnt[1]);
        }
    }
    return 0;
}
