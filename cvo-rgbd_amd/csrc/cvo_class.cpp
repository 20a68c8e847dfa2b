// cvo_class.cpp -- C++ registration objects over the C-ABI (see include/cvo.hpp).
#include "cvo.hpp"

#include <cmath>
#include <cstring>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

namespace cvo_hip {

Affine3f::Affine3f()
{
    std::memset(m, 0, sizeof(m));
    m[0] = m[5] = m[10] = m[15] = 1.0f;
}

void Affine3f::translation(float t[3]) const
{
    t[0] = m[3]; t[1] = m[7]; t[2] = m[11];
}

void Affine3f::linear(float r[9]) const
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r[3 * i + j] = m[4 * i + j];
}

void Affine3f::quaternion(float q[4]) const
{   // Eigen's quaternion-from-matrix (Shoemake), float
    const float m00 = m[0], m11 = m[5], m22 = m[10];
    float t = m00 + m11 + m22;
    float x, y, z, w;
    if (t > 0.0f) {
        t = std::sqrt(t + 1.0f);
        w = 0.5f * t;
        t = 0.5f / t;
        x = (m[9] - m[6]) * t;
        y = (m[2] - m[8]) * t;
        z = (m[4] - m[1]) * t;
    } else {
        int i = 0;
        if (m11 > m00) i = 1;
        if (m22 > m[5 * i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(m[5 * i] - m[5 * j] - m[5 * k] + 1.0f);
        float qv[3];
        qv[i] = 0.5f * t;
        t = 0.5f / t;
        w = (m[4 * k + j] - m[4 * j + k]) * t;
        qv[j] = (m[4 * j + i] + m[4 * i + j]) * t;
        qv[k] = (m[4 * k + i] + m[4 * i + k]) * t;
        x = qv[0]; y = qv[1]; z = qv[2];
    }
    q[0] = x; q[1] = y; q[2] = z; q[3] = w;
}

registration::registration(int mode, int device, void *stream)
    : init(false), iter(0), ctx_(nullptr), have_moving_(false), n_iter_(0), fe_(nullptr), fe_w_(0), fe_h_(0),
      fe_points_(0), device_(device)
{
    check(cvo_hip_default_params(mode, &params_), "cvo_hip_default_params");
    check(cvo_hip_init_state(&params_, &state_), "cvo_hip_init_state");
    check(cvo_hip_create(device, stream, &params_, &ctx_), "cvo_hip_create");
}

registration::~registration()
{
    if (fe_) cvo_fe_destroy(fe_);
    cvo_hip_destroy(ctx_);
}

// pcd_generator::load_image + create_pointcloud (ref src/cvo.cpp:321-341): cvo asks for
// the raw colour features (type 1), acvo for the HSV ones (type 0, ref
// src/adaptive_cvo.cpp:451,462).  The cloud never leaves the device: the front end's
// output arrays go straight into cvo_hip_set_*_device.
void registration::cloud_from_images(int dataset_seq, const image_view &rgb, const image_view &dep)
{
    if (!rgb.data || !dep.data || rgb.rows != dep.rows || rgb.cols != dep.cols)
        throw std::runtime_error("set_pcd(): colour and depth image must have the same size");
    if (!fe_) {
        const int rc = cvo_fe_create(device_, nullptr, rgb.cols, rgb.rows, &fe_);
        if (rc != CVO_HIP_OK) throw std::runtime_error(std::string("cvo_fe_create: ") + cvo_hip_error_string(rc));
        fe_w_ = rgb.cols; fe_h_ = rgb.rows;
        cvo_fe_set_device_output(fe_, 1);
    }
    if (rgb.cols != fe_w_ || rgb.rows != fe_h_)
        throw std::runtime_error("set_pcd(): the image size changed within a sequence");
    const int ftype = params_.mode == CVO_HIP_MODE_ACVO ? CVO_FE_FEATURES_HSV : CVO_FE_FEATURES_RGB;
    int rc = cvo_fe_submit(fe_, (const uint8_t *)rgb.data, rgb.step, (const uint16_t *)dep.data, dep.step,
                           dataset_seq, ftype);
    const float *d_pos = nullptr, *d_feat = nullptr;
    if (rc == CVO_HIP_OK) rc = cvo_fe_collect_device(fe_, &d_pos, &d_feat, &fe_points_);
    if (rc != CVO_HIP_OK)
        throw std::runtime_error(std::string("front end: ") + cvo_hip_error_string(rc) + " (" +
                                 cvo_fe_last_error(fe_) + ")");
    if (init == false) {   // ref src/cvo.cpp:325-334
        std::cout << "initializing cvo..." << std::endl;
        check(cvo_hip_set_fixed_device(ctx_, d_pos, d_feat, fe_points_, CVO_HIP_FEAT_ROWMAJOR),
              "cvo_hip_set_fixed_device");
        std::cout << "first pcd generated!" << std::endl;
        init = true;
        return;
    }
    check(cvo_hip_set_moving_device(ctx_, d_pos, d_feat, fe_points_, CVO_HIP_FEAT_ROWMAJOR),
          "cvo_hip_set_moving_device");
    have_moving_ = true;
    std::cout << "num moving: " << fe_points_ << std::endl;   // ref src/cvo.cpp:343-347
}

void registration::set_pcd(const int dataset_seq, const image_view &RGB_img, const image_view &dep_img,
                           const std::string &, const std::string &)
{
    cloud_from_images(dataset_seq, RGB_img, dep_img);
}

void registration::run_cvo(const int dataset_seq, const image_view &RGB_img, const image_view &dep_img,
                           const std::string &, const std::string &)
{   // ref src/cvo.cpp:422-435
    const bool first = !init;
    cloud_from_images(dataset_seq, RGB_img, dep_img);
    if (first) return;
    align();
    std::cout << "Total iterations: " << iter << std::endl;
    std::cout << "RKHS-SE(3) Object Transformation Estimate: \n";
    for (int r = 0; r < 4; ++r)
        std::cout << transform.m[4 * r] << " " << transform.m[4 * r + 1] << " " << transform.m[4 * r + 2] << " "
                  << transform.m[4 * r + 3] << std::endl;
}

void registration::check(int status, const char *what)
{
    if (status == CVO_HIP_OK) return;
    std::string msg = std::string(what) + ": " + cvo_hip_error_string(status);
    if (ctx_) msg += std::string(" (") + cvo_hip_last_error(ctx_) + ")";
    throw std::runtime_error(msg);
}

void registration::publish()
{
    std::memcpy(transform.m, state_.transform, sizeof(transform.m));
    std::memcpy(prev_transform.m, state_.prev_transform, sizeof(prev_transform.m));
    std::memcpy(accum_transform.m, state_.accum_transform, sizeof(accum_transform.m));
    iter = state_.iter;
}

void registration::set_pcd(const point_cloud_view &pc)
{   // ref src/cvo.cpp:319-357
    if (init == false) {
        std::cout << "initializing cvo..." << std::endl;
        check(cvo_hip_set_fixed(ctx_, pc.positions, pc.features, pc.num_points, pc.feat_layout),
              "cvo_hip_set_fixed");
        std::cout << "first pcd generated!" << std::endl;
        init = true;
        return;
    }
    check(cvo_hip_set_moving(ctx_, pc.positions, pc.features, pc.num_points, pc.feat_layout),
          "cvo_hip_set_moving");
    have_moving_ = true;
}

void registration::align()
{   // ref src/cvo.cpp:361-420; the moving cloud becomes the fixed one (:417)
    if (!have_moving_) throw std::runtime_error("align(): set_pcd() must precede each align()");
    check(cvo_hip_align(ctx_, &state_, nullptr, 0, &n_iter_), "cvo_hip_align");
    check(cvo_hip_swap_moving_to_fixed(ctx_), "cvo_hip_swap_moving_to_fixed");
    have_moving_ = false;
    publish();
}

void registration::align_many(registration *const *objects, int count)
{
    std::vector<cvo_hip_ctx *> ctxs((size_t)count);
    std::vector<cvo_hip_state *> states((size_t)count);
    std::vector<int> iters((size_t)count, 0);
    for (int i = 0; i < count; ++i) {
        if (!objects[i] || !objects[i]->have_moving_)
            throw std::runtime_error("align_many(): set_pcd() must precede align() for every object");
        ctxs[(size_t)i] = objects[i]->ctx_;
        states[(size_t)i] = &objects[i]->state_;
    }
    const int rc = cvo_hip_align_many(ctxs.data(), states.data(), iters.data(), count);
    if (rc != CVO_HIP_OK) {
        for (int i = 0; i < count; ++i) {
            const char *d = cvo_hip_last_error(ctxs[(size_t)i]);
            if (d && d[0]) throw std::runtime_error(std::string("cvo_hip_align_many: ") + d);
        }
        throw std::runtime_error(std::string("cvo_hip_align_many: ") + cvo_hip_error_string(rc));
    }
    for (int i = 0; i < count; ++i) {
        registration *o = objects[i];
        o->n_iter_ = iters[(size_t)i];
        o->check(cvo_hip_swap_moving_to_fixed(o->ctx_), "cvo_hip_swap_moving_to_fixed");
        o->have_moving_ = false;
        o->publish();
    }
}

void registration::run_cvo(const point_cloud_view &pc)
{   // ref src/cvo.cpp:422-435
    if (init == false) {
        set_pcd(pc);
    } else {
        set_pcd(pc);
        align();
        std::cout << "Total iterations: " << iter << std::endl;
        std::cout << "RKHS-SE(3) Object Transformation Estimate: \n";
        for (int r = 0; r < 4; ++r)
            std::cout << transform.m[4 * r] << " " << transform.m[4 * r + 1] << " "
                      << transform.m[4 * r + 2] << " " << transform.m[4 * r + 3] << std::endl;
    }
}

}   // namespace cvo_hip

namespace acvo {

float acvo::function_inner_product(const cvo_hip::point_cloud_view *cloud_a,
                                   const cvo_hip::point_cloud_view *cloud_b)
{   // ref src/adaptive_cvo.cpp:385-439: two arbitrary clouds and the current ell; nothing else is read
    if (!cloud_a || !cloud_b || cloud_a->feat_layout != cloud_b->feat_layout)
        check(CVO_HIP_ERR_INVALID, "function_inner_product");
    float out = 0.0f;
    check(cvo_hip_function_inner_product_clouds(ctx_, state_.ell, cloud_a->positions, cloud_a->features,
                                                cloud_a->num_points, cloud_b->positions, cloud_b->features,
                                                cloud_b->num_points, cloud_a->feat_layout, &out),
          "cvo_hip_function_inner_product_clouds");
    return out;
}

}   // namespace acvo
