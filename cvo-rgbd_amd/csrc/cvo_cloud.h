// cvo_cloud.h -- device-side preparation of a cloud (cvo_cloud.hip)
#pragma once
#include <hip/hip_runtime.h>

#include "cvo_device.h"

namespace cvo_dev {

struct CloudPrep {
    const float *xyz;    // device: n x 3 as the caller gave them
    const float *feat;   // device: n x 5, row- or column-major
    int n, colmajor;
    int np;              // rows of the device arrays: n rounded up to CLOUD_PAD, the tail is padding
    int pad_axis;        // 0 / 1: which way the padding rows are parked (differs between the two clouds of a pair)
    const float *bbox;   // device: bounding box of xyz (min xyz, max xyz)
    uint32_t *keys[2];   // device scratch, n each
    int *idx[2];
    void *scratch;       // rocPRIM temporary storage
    size_t scratch_bytes;
    float4 *pos;         // out: Morton-sorted rows (x, y, z, f4)
    float *feat8;        // out: n x FEAT_STRIDE
    float4 *seg;         // out: bounding spheres of the SEG-point runs
};

// Device clouds are padded to a multiple of CLOUD_PAD rows: kernel arguments (and captured
// graphs) then change only when a cloud crosses a bucket, not with every frame of a stream.
// Padding rows are parked ~10 km away, 16 m apart, with NaN features: the filter never lists
// them, and if it did the exact test would drop them.
constexpr int CLOUD_PAD = 256;
inline int cloud_padded(int n) { return n <= 0 ? 0 : (n + CLOUD_PAD - 1) / CLOUD_PAD * CLOUD_PAD; }

// One cloud of a hand-over made by ONE launch (k_cloud_one, cvo_cloud.hip): a block per cloud does what
// cloud_bbox_device + cloud_prepare_device do in ten launches -- bounding box, Morton keys, a stable radix sort
// in LDS, packed rows, bounding spheres, padding rows -- for clouds of up to CLOUD_ONE_MAX points; the arrays
// it writes are the same, bit for bit.
constexpr int CLOUD_ONE_MAX = 16384;
struct CloudJob {
    const float *xyz;    // device: n x 3 as the caller gave them
    const float *feat;   // device: n x 5, row- or column-major
    int n, colmajor;
    int np, pad_axis;    // as CloudPrep
    float4 *pos;         // out
    float *feat8;
    float4 *seg;
    float *bbox_out;     // out: min xyz, max xyz -- device or host-mapped (pinned) memory
};
size_t cloud_one_smem_bytes(int nmax);
// jobs: device array; nmax: the largest n among them (sizes the LDS of the launch)
hipError_t cloud_prepare_many(const CloudJob *d_jobs, int count, int nmax, hipStream_t s);
hipError_t cloud_prepare_one(const CloudJob &job, hipStream_t s);

size_t cloud_sort_scratch_bytes(int n);
// bbox6 (device): min xyz, max xyz
hipError_t cloud_bbox_device(const float *d_xyz, int n, float *d_bbox6, hipStream_t s);
hipError_t cloud_prepare_device(const CloudPrep &c, hipStream_t s);

}   // namespace cvo_dev
