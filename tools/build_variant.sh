#!/bin/bash
# A/B builds: cvo_kernels.hip compiled once more with extra -D flags and linked with the library's other objects
# into cvo-rgbd_amd/csrc/libcvo_hip_<name>.so (what CVO_LIB of tools/gpu_batch.py / gpu_single.py / gpu_abx_libs.py names).
#   tools/build_variant.sh <name> [-DFLAG ...]
set -e
cd "$(dirname "$0")/../cvo-rgbd_amd/csrc"
name=$1; shift
make -s all
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-result -Wno-invalid-offsetof \
    -I../../include -I. --offload-arch=gfx950 -mllvm -amdgpu-kernarg-preload-count=8 "$@" -c cvo_kernels.hip -o /tmp/cvo_kernels_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libcvo_hip_$name.so /tmp/cvo_kernels_$name.o \
    cvo_capi.o cvo_clouds.o cvo_plan.o cvo_job.o cvo_engine.o cvo_comm.o cvo_class.o cvo_frontend.o cvo_cloud.o cvo_prep.o -ldl
echo built libcvo_hip_$name.so
