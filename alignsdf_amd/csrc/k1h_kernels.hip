// K1h: the fused decoder on split-half fp16 MFMAs (sdf_mlp_f16_kernel.h), the default arithmetic of the grid sweeps.
#include "k1_launch.h"
#include "sdf_mlp_f16_kernel.h"

namespace asdf {

__global__ __launch_bounds__(256, 1) void sdf_mlp_f16_kernel(const DecodeParams p) { sdf_mlp_f16_body<false>(p); }
__global__ __launch_bounds__(256, 1) void sdf_mlp_f16_combined_kernel(const DecodeParams p) { sdf_mlp_f16_body<true>(p); }

// the split-half arithmetic on a voxel list (the exact values of the narrow-band fine sweep)
__global__ __launch_bounds__(256, 1) void sdf_mlp_f16_subset_kernel(const DecodeParams p) { sdf_mlp_f16_body<false, 0, 2, 2, 1, true>(p); }
__global__ __launch_bounds__(256, 1) void sdf_mlp_f16_subset_combined_kernel(const DecodeParams p) { sdf_mlp_f16_body<true, 0, 2, 2, 1, true>(p); }

hipError_t k1h_prepare() {
  hipError_t e = k1h_nerf_prepare();
  for (const void* k : {(const void*)sdf_mlp_f16_kernel, (const void*)sdf_mlp_f16_combined_kernel, (const void*)sdf_mlp_f16_subset_kernel,
                        (const void*)sdf_mlp_f16_subset_combined_kernel})
    if (e == hipSuccess) e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytesF16);
  if (e == hipSuccess) e = k1s_prepare();
  return e;
}

void k1h_subset_launch(int kp, bool two_out, const DecodeParams& p, int grid, hipStream_t st) {
  if (kp != 2) { k1h_nerf_subset_launch(kp, two_out, p, grid, st); return; }
  if (two_out) hipLaunchKernelGGL(sdf_mlp_f16_subset_combined_kernel, dim3(grid), dim3(256), kLdsBytesF16, st, p);
  else hipLaunchKernelGGL(sdf_mlp_f16_subset_kernel, dim3(grid), dim3(256), kLdsBytesF16, st, p);
}

void k1h_launch(int kp, bool two_out, const DecodeParams& p, int grid, hipStream_t st) {
  if (kp != 2) { k1h_nerf_launch(kp, two_out, p, grid, st); return; }
  if (two_out) hipLaunchKernelGGL(sdf_mlp_f16_combined_kernel, dim3(grid), dim3(256), kLdsBytesF16, st, p);
  else hipLaunchKernelGGL(sdf_mlp_f16_kernel, dim3(grid), dim3(256), kLdsBytesF16, st, p);
}

}  // namespace asdf
