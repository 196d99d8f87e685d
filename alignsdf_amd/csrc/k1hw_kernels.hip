// K1h, the W form (round 6): the split-half SeparateDecoder kernels on v_mfma_f32_16x16x32_f16 (sdf_mlp_f16_kernel.h, "the W form") - the
// default arithmetic of every grid sweep of a decoder with affine point features.  Their own translation unit: their own compiler flags
// (alignsdf_amd/build_native.py).
#include "k1_launch.h"
#include "sdf_mlp_f16w_kernel.h"

namespace asdf {

__global__ __launch_bounds__(256, 1) void sdf_mlp_f16w_kernel(const DecodeParams p) { sdf_mlp_f16w_body<0, false>(p); }
// ... over a voxel list (the exact values of the narrow-band fine sweep, the audit picks)
__global__ __launch_bounds__(256, 1) void sdf_mlp_f16w_subset_kernel(const DecodeParams p) { sdf_mlp_f16w_body<0, true>(p); }

hipError_t k1hw_prepare() {
  hipError_t e = hipSuccess;
  for (const void* k : {(const void*)sdf_mlp_f16w_kernel, (const void*)sdf_mlp_f16w_subset_kernel})
    if (e == hipSuccess) e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytesF16);
  return e;
}

void k1hw_launch(const DecodeParams& p, int grid, hipStream_t st) {
  hipLaunchKernelGGL(sdf_mlp_f16w_kernel, dim3(grid), dim3(256), kLdsBytesF16, st, p);
}

void k1hw_subset_launch(const DecodeParams& p, int grid, hipStream_t st) {
  hipLaunchKernelGGL(sdf_mlp_f16w_subset_kernel, dim3(grid), dim3(256), kLdsBytesF16, st, p);
}

}  // namespace asdf
