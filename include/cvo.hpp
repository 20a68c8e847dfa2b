// cvo.hpp -- C++ host-side mirror of the reference's registration objects
// cvo::cvo (ref cpp/rkhs_registration/include/cvo.hpp:55-193) and acvo::acvo
// (ref include/adaptive_cvo.hpp:57-196), backed by the HIP C-ABI (cvo_hip.h).
//
// Same public surface: members init, iter, transform, prev_transform,
// accum_transform; methods set_pcd(), align(), run_cvo(); acvo additionally
// function_inner_product().  The reference's set_pcd()/run_cvo() take
// cv::Mat RGB/depth images and run the pcd_generator front end; here they take
// either an image_view pair (plain pointers in place of cv::Mat; the front end
// then runs on the GPU behind cvo_frontend.h) or directly the
// point_cloud a front end produced (positions + 5 features,
// ref include/data_type.h:59-71).  State carry-over between frames follows
// the reference object exactly (ell and R,T are not reset in cvo; acvo resets
// ell per pair): SURVEY 8a quirks 1-4, 10, 13.
//
// Error behaviour: the reference methods are void and fail by UB; these throw
// std::runtime_error carrying the C-ABI status text.
#pragma once

#include <string>

#include "cvo_frontend.h"
#include "cvo_hip.h"

namespace cvo_hip {

// Stand-in for the cv::Mat arguments of set_pcd() / run_cvo(): what cv::imread gives
// -- rows x cols pixels, `step` bytes per row; the colour image 3 bytes per pixel in
// file (B, G, R) order, the depth image one uint16 per pixel.
struct image_view {
    const void *data;
    int rows, cols;
    size_t step;
};

// Stand-in for Eigen::Affine3f: 4x4 row-major, matrix()(r,c) access.
struct Affine3f {
    float m[16];
    Affine3f();
    struct View {
        float *p;
        float &operator()(int r, int c) { return p[4 * r + c]; }
        float operator()(int r, int c) const { return p[4 * r + c]; }
    };
    View matrix() { return View{m}; }
    const float *data() const { return m; }
    void translation(float t[3]) const;
    void linear(float r[9]) const;
    // unit quaternion (x, y, z, w) of the rotation block, as
    // Eigen::Quaternionf(transform.linear()) gives it (ref src/cvo_main.cpp:61-64)
    void quaternion(float q[4]) const;
};

// A point cloud as the front end hands it over (ref data_type.h:59-71).
struct point_cloud_view {
    int num_points;
    const float *positions;   // n x 3 AoS
    const float *features;    // n x 5
    int feat_layout;          // CVO_HIP_FEAT_COLMAJOR (Eigen default) or _ROWMAJOR
};

class registration {
  public:
    bool init;
    int iter;
    Affine3f transform;
    Affine3f prev_transform;
    Affine3f accum_transform;

    explicit registration(int mode, int device = 0, void *stream = nullptr);
    ~registration();
    registration(const registration &) = delete;
    registration &operator=(const registration &) = delete;

    void set_pcd(const point_cloud_view &pc);
    void align();
    void run_cvo(const point_cloud_view &pc);
    // The reference's own signatures (ref include/cvo.hpp:171-192): images in, the front
    // end (pcd_generator) runs first -- on the GPU.  The two paths are dead parameters
    // there too (ref src/cvo.cpp:332,341).
    void set_pcd(const int dataset_seq, const image_view &RGB_img, const image_view &dep_img,
                 const std::string &pcd_pth = std::string(), const std::string &pcd_dso_pth = std::string());
    void run_cvo(const int dataset_seq, const image_view &RGB_img, const image_view &dep_img,
                 const std::string &pcd_pth = std::string(), const std::string &pcd_dso_pth = std::string());
    int num_points_last_frame() const { return fe_points_; }
    // Batched mode: align() of `count` objects (each with its moving cloud set) in
    // one call, their kernel launches shared (cvo_hip_align_many).  The result of
    // every object is what its own align() would have given.
    static void align_many(registration *const *objects, int count);

    int num_iterations() const { return n_iter_; }   // loop bodies executed by the last align()
    cvo_hip_ctx *context() { return ctx_; }
    const cvo_hip_state &state() const { return state_; }

  protected:
    cvo_hip_ctx *ctx_;
    cvo_hip_params params_;
    cvo_hip_state state_;
    bool have_moving_;
    int n_iter_;
    cvo_fe_ctx *fe_;           // front end, created with the first image (its size is fixed then)
    int fe_w_, fe_h_, fe_points_;
    int device_;
    void check(int status, const char *what);
    void publish();
    void cloud_from_images(int dataset_seq, const image_view &rgb, const image_view &dep);
};

}   // namespace cvo_hip

namespace cvo {
class cvo : public cvo_hip::registration {
  public:
    explicit cvo(int device = 0, void *stream = nullptr)
        : cvo_hip::registration(CVO_HIP_MODE_CVO, device, stream) {}
};
}   // namespace cvo

namespace acvo {
class acvo : public cvo_hip::registration {
  public:
    explicit acvo(int device = 0, void *stream = nullptr)
        : cvo_hip::registration(CVO_HIP_MODE_ACVO, device, stream) {}
    // ref include/adaptive_cvo.hpp:179, src/adaptive_cvo.cpp:385-439 (public, never called in
    // the tree): the inner-product statistic of two arbitrary clouds at the current
    // length-scale.  Reads nothing else and changes nothing: a pending set_pcd() stays pending.
    float function_inner_product(const cvo_hip::point_cloud_view *cloud_a,
                                 const cvo_hip::point_cloud_view *cloud_b);
};
}   // namespace acvo
