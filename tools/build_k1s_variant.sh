#!/bin/bash
# tools/bin/libalignsdf_hip_<name>.so = the shipped library with k1s_kernels.hip rebuilt under extra flags:  build_k1s_variant.sh <name> [flags...]
name=$1; shift
cd "$(dirname "$0")/../alignsdf_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form "$@" -c k1s_kernels.hip -o /tmp/k1s_$name.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC decoder.o k1_kernels.o k1_cls_kernels.o k1h_kernels.o k1h_nerf_kernels.o /tmp/k1s_$name.o k1s_nerf_kernels.o mc33.o icp.o mesh_cc.o -o ../../tools/bin/libalignsdf_hip_$name.so
