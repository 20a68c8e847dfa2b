// integration_stubs.hpp -- ten-line stand-ins for the reference's types (Eigen, the point cloud, the image front end), so that the
// code INTEGRATION.md tells a maintainer to paste into the reference's cvo.cpp can be put in front of a compiler
// (tests/test_integration_snippets.py).  A BOUNDARY TYPO CHECK: names, argument counts and types of the C-ABI calls against
// include/cvo_hip.h.  It pins nothing about results and is no parity evidence.  Nothing here is the reference's code.
#pragma once
#include <memory>
#include <stdexcept>
#include <vector>
namespace Eigen {
enum { Dynamic = -1, RowMajor = 1, ColMajor = 0 };
template <class T, int N> struct Vec {
    T v[N];
    Vec() : v{} {}
    Vec(T a, T b, T c) : v{a, b, c} {}
    T *data() { return v; }
    const T *data() const { return v; }
    T &operator()(int i) { return v[i]; }
    template <class U> Vec<U, N> cast() const { Vec<U, N> o; for (int i = 0; i < N; ++i) o.v[i] = (U)v[i]; return o; }
};
typedef Vec<float, 3> Vector3f;
typedef Vec<double, 3> Vector3d;
template <class T, int R, int C, int O = ColMajor> struct Matrix {
    std::vector<T> m;
    T *data() { return m.data(); }
    const T *data() const { return m.data(); }
    template <class X> Matrix &operator=(const X &) { return *this; }
};
typedef Matrix<float, 3, 3> Matrix3f;
template <class M> struct Map { template <class P> explicit Map(P *) {} };
struct Affine3f { Matrix<float, 4, 4> mm; Matrix<float, 4, 4> &matrix() { return mm; } };
}   // namespace Eigen
namespace cv { struct Mat {}; }
namespace cvo {
struct frame {};
struct point_cloud {   // (the reference's cloud_t as its consumers see it: AoS positions, column-major N x 5 features)
    int num_points = 0;
    std::vector<Eigen::Vector3f> positions;
    Eigen::Matrix<float, Eigen::Dynamic, 5> features;
};
struct pcd_generator {
    void load_image(const cv::Mat &, const cv::Mat &, frame *) {}
    void create_pointcloud(int, frame *, point_cloud *) {}
};
}   // namespace cvo
