/* cvo_frontend.h -- C-ABI of the MI355X RGB-D front end: one RGB
 * image + one depth image in, the semi-dense coloured point cloud the
 * registration consumes out.  Part of libcvo_hip.so; status codes are
 * cvo_hip_status (cvo_hip.h).
 *
 * Replaces, behind plain pointers and sizes, what the reference does inside
 * set_pcd() with cv::Mat arguments:
 *   cvo::pcd_generator::load_image          ref cpp/rkhs_registration/src/pcd_generator.cpp:387-398
 *   cvo::pcd_generator::create_pointcloud   ref src/pcd_generator.cpp:400-420
 *     make_pyramid :33-129, select_point :131-176 (dso::PixelSelector::makeMaps,
 *     ref thirdparty/PixelSelector2.cpp:137-236, and the Canny top-up),
 *     get_points_from_pixels :233-327, get_features :329-385
 * as called from cvo::set_pcd (ref src/cvo.cpp:318-341, feature type 1) and
 * acvo::set_pcd (ref src/adaptive_cvo.cpp:440-463, feature type 0).
 *
 * Every stage runs as HIP kernels on the context's stream; the images are
 * copied in, the cloud is copied out.  There is no CPU path: without a gfx950
 * device cvo_fe_create() fails with CVO_HIP_ERR_NODEVICE.
 */
#ifndef CVO_FRONTEND_H
#define CVO_FRONTEND_H

#include <stddef.h>
#include <stdint.h>

#include "cvo_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cvo_fe_ctx cvo_fe_ctx;

/* feature_type of create_pointcloud (ref include/pcd_generator.hpp:96-99) */
enum { CVO_FE_FEATURES_HSV = 0,   /* H/180 S/255 V/255 dx/255*2 dy/255*2 (acvo) */
       CVO_FE_FEATURES_RGB = 1 }; /* raw channel bytes + raw gradients (cvo) */

/* intermediate images, for parity checks (cvo_fe_read_stage) */
enum { CVO_FE_STAGE_GRAY = 0,     /* w*h uint8 */
       CVO_FE_STAGE_HSV = 1,      /* w*h*3 uint8 */
       CVO_FE_STAGE_MAP = 2,      /* w*h float: 0 / 1 / 2 / 4 as the selector writes them */
       CVO_FE_STAGE_AG0 = 3,      /* squared gradient magnitude, level 0: w*h float */
       CVO_FE_STAGE_AG1 = 4,      /* level 1: (w/2)*(h/2) float */
       CVO_FE_STAGE_AG2 = 5,      /* level 2: (w/4)*(h/4) float */
       CVO_FE_STAGE_THS = 6,      /* smoothed cell thresholds: (w/32)*(h/32) float */
       CVO_FE_STAGE_DX0 = 7, CVO_FE_STAGE_DY0 = 8,   /* level-0 gradients: w*h float */
       CVO_FE_STAGE_EDGES = 9 };  /* Canny edges of the last top-up: w*h uint8 (0 / 255) */

typedef struct cvo_fe_info {
    int32_t num_selected;   /* pixels the selector kept (before the depth test), ref pcd_generator.cpp:141 */
    int32_t pot_used;       /* potential of the selection pass that produced the map */
    int32_t reselected;     /* 1: the first pass missed the density band and a second one ran */
    int32_t canny_used;     /* 1: fewer than num_want/3 were kept, edges were added */
    int32_t num_points;     /* points in the cloud (selected and depth != 0) */
    int32_t pad_;
} cvo_fe_info;

/* One context per image size, device and stream.  `stream` as in cvo_hip_create
 * (NULL: a stream of its own).  Images must be at least 64 x 64. */
int cvo_fe_create(int device, void *stream, int width, int height, cvo_fe_ctx **out);
int cvo_fe_destroy(cvo_fe_ctx *ctx);
const char *cvo_fe_last_error(const cvo_fe_ctx *ctx);

/* num_want of pcd_generator (ref src/pcd_generator.cpp:22; default 3000) */
int cvo_fe_set_num_want(cvo_fe_ctx *ctx, int num_want);

/* load_image + create_pointcloud.
 *   img:   height rows of width*3 bytes, `img_stride` bytes apart, channel order as
 *          decoded from the file by cv::imread (B, G, R) -- the reference hands that
 *          to its RGB conversions unchanged, and so does this.
 *   depth: height rows of width uint16, `depth_stride` BYTES apart.
 *   dataset_seq: camera table index (ref src/pcd_generator.cpp:241-295); 1 = TUM fr1.
 *   positions: capacity*3 floats (x y z per point); features: capacity*5 floats,
 *   ROW-major (CVO_HIP_FEAT_ROWMAJOR).  Points are in image scan order.
 *   *num_points: points found; if it exceeds `capacity` only the first `capacity`
 *   are stored and CVO_HIP_ERR_INVALID is returned. */
int cvo_fe_create_pointcloud(cvo_fe_ctx *ctx, const uint8_t *img, size_t img_stride, const uint16_t *depth,
                             size_t depth_stride, int dataset_seq, int feature_type, float *positions,
                             float *features, int capacity, int *num_points);

/* The context's own pinned staging images (width*3 bytes per colour row, width uint16 per
 * depth row, no padding): a decoder that writes straight into them -- e.g. a cv::Mat header
 * over the pointer handed to cv::imdecode -- saves the copy that submit() / create_pointcloud()
 * otherwise make; pass these same pointers (and the dense strides) to them.  They may be
 * refilled once the frame has been collected. */
int cvo_fe_host_buffers(cvo_fe_ctx *ctx, uint8_t **img, uint16_t **depth);

/* The same in two halves, for callers that have other work while the GPU is busy (the
 * drivers register frame k while frame k+1 is in the front end): submit() stages the
 * images, enqueues every kernel and the copies back and returns; collect() waits, runs the
 * rare edge top-up if the frame needs it, and delivers the cloud.  One frame in flight per
 * context; the image buffers may be re-used as soon as submit() returns. */
int cvo_fe_submit(cvo_fe_ctx *ctx, const uint8_t *img, size_t img_stride, const uint16_t *depth,
                  size_t depth_stride, int dataset_seq, int feature_type);
int cvo_fe_collect(cvo_fe_ctx *ctx, float *positions, float *features, int capacity, int *num_points);

/* collect() without the copy to the host: the cloud stays in device memory (positions
 * n x 3, features n x 5 row-major, floats) for cvo_hip_set_fixed_device / _set_moving_device.
 * The pointers are valid until the next submit() / create_pointcloud() on this context. */
int cvo_fe_collect_device(cvo_fe_ctx *ctx, const float **d_positions, const float **d_features, int *num_points);

/* 1: the following frames are collected with cvo_fe_collect_device(): submit() then does not
 * start the (optimistic) copy of the cloud to the host.  cvo_fe_collect() still works. */
int cvo_fe_set_device_output(cvo_fe_ctx *ctx, int on);

/* what the last create_pointcloud / collect did */
int cvo_fe_get_info(const cvo_fe_ctx *ctx, cvo_fe_info *out);
/* copy an intermediate image of the last create_pointcloud to host memory */
int cvo_fe_read_stage(cvo_fe_ctx *ctx, int stage, void *out, size_t bytes);

/* The selector's random bytes (ref thirdparty/PixelSelector2.cpp:35-37:
 * srand(3141592); rand() & 0xFF) from a restatement of the C library's additive
 * feedback generator: host only, no device needed. */
int cvo_fe_random_pattern(int n, uint8_t *out);
/* The camera table: {depth scale, fx, fy, cx, cy}.  Host only. */
int cvo_fe_camera(int dataset_seq, float cam[5]);

#ifdef __cplusplus
}
#endif
#endif
