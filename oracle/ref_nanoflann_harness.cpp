// ref_nanoflann_harness.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Thin C-ABI around the REFERENCE'S OWN vendored nanoflann header, compiled
// from where it lies under /root/reference (never copied into this repo; see
// oracle/Makefile target `ref`).  It repeats the call sequence of
// se_kernel() (ref src/cvo.cpp:110-125): build a
// KDTreeVectorOfVectorsAdaptor<cloud_t,float> with leaf size 10 on the column
// cloud, then radiusSearch(query, d2_thres, matches, SearchParams()) per row.
// Used to pin WHICH pairs enter the Gram matrix (strict '<', exact search,
// float32 metric) against the oracle's dense-threshold / grid search.
//
// cloud_t in the reference is std::vector<Eigen::Vector3f>; Eigen is not in
// this image, and the adaptor only needs operator[] and size(), so
// std::array<float,3> stands in for the point type (same 12-byte layout).
#include <nanoflann.hpp>
#include <KDTreeVectorOfVectorsAdaptor.h>

#include <array>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

typedef std::vector<std::array<float, 3>> cloud_t;
typedef KDTreeVectorOfVectorsAdaptor<cloud_t, float> kd_tree_t;

extern "C" {

// Returns 0 on success.  row_ptr has na+1 entries (caller-allocated); *col and
// *d2 are malloc'd (free with ref_free) in nanoflann's own output order
// (sorted by distance, SearchParams default).
int ref_radius_search(const float *b_xyz, int nb, const float *a_xyz, int na, float radius_sq,
                      int64_t *row_ptr, int32_t **col, float **d2)
{
    if (nb <= 0 || na < 0) return -1;
    cloud_t cloud_b((size_t)nb);
    for (int j = 0; j < nb; ++j)
        for (int k = 0; k < 3; ++k) cloud_b[j][k] = b_xyz[3 * j + k];
    kd_tree_t mat_index(3 /*dim*/, cloud_b, 10 /* max leaf */);
    mat_index.index->buildIndex();   // the reference builds twice, too

    std::vector<int32_t> cols;
    std::vector<float> dists;
    row_ptr[0] = 0;
    for (int i = 0; i < na; ++i) {
        std::vector<std::pair<size_t, float>> ret_matches;
        nanoflann::SearchParams params;
        const size_t n = mat_index.index->radiusSearch(a_xyz + 3 * i, radius_sq, ret_matches, params);
        for (size_t q = 0; q < n; ++q) {
            cols.push_back((int32_t)ret_matches[q].first);
            dists.push_back(ret_matches[q].second);
        }
        row_ptr[i + 1] = (int64_t)cols.size();
    }
    *col = (int32_t *)malloc(sizeof(int32_t) * (cols.size() ? cols.size() : 1));
    *d2 = (float *)malloc(sizeof(float) * (dists.size() ? dists.size() : 1));
    if (!*col || !*d2) return -2;
    if (!cols.empty()) {
        memcpy(*col, cols.data(), sizeof(int32_t) * cols.size());
        memcpy(*d2, dists.data(), sizeof(float) * dists.size());
    }
    return 0;
}

void ref_free(void *p) { free(p); }
}
