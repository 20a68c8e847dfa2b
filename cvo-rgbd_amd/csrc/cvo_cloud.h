// cvo_cloud.h -- device-side preparation of a cloud (cvo_cloud.hip)
#pragma once
#include <hip/hip_runtime.h>

#include "cvo_device.h"

namespace cvo_dev {

struct CloudPrep {
    const float *xyz;    // device: n x 3 as the caller gave them
    const float *feat;   // device: n x 5, row- or column-major
    int n, colmajor;
    float lo[3], hi[3];  // bounding box of xyz
    uint32_t *keys[2];   // device scratch, n each
    int *idx[2];
    void *scratch;       // rocPRIM temporary storage
    size_t scratch_bytes;
    float4 *pos;         // out: Morton-sorted rows (x, y, z, f4)
    float *feat8;        // out: n x FEAT_STRIDE
    float4 *seg;         // out: bounding spheres of the SEG-point runs
};

size_t cloud_sort_scratch_bytes(int n);
// bbox6 (device): min xyz, max xyz
hipError_t cloud_bbox_device(const float *d_xyz, int n, float *d_bbox6, hipStream_t s);
hipError_t cloud_prepare_device(const CloudPrep &c, hipStream_t s);

}   // namespace cvo_dev
