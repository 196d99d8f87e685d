// K1h for the in-kernel NeRF encoding (PointFeatSize 9 / 15, utils/mesh.py:53-55): the split-half kernel with 16 KiB stages -
// the constants block of these decoders (static point-feature fragments of layers 0 and 2) takes 40 / 75 KiB of the LDS.
#define ASDF16_STAGE_KB 8
#include "k1_launch.h"
#include "sdf_mlp_f16_kernel.h"

namespace asdf {

__global__ __launch_bounds__(256, 1) void sdf_mlp_f16_nerf9_kernel(const DecodeParams p) { sdf_mlp_f16_body<false, 0, 5>(p); }
__global__ __launch_bounds__(256, 1) void sdf_mlp_f16_nerf15_kernel(const DecodeParams p) { sdf_mlp_f16_body<false, 0, 8>(p); }
__global__ __launch_bounds__(256, 1) void sdf_mlp_f16_combined_nerf9_kernel(const DecodeParams p) { sdf_mlp_f16_body<true, 0, 5>(p); }
__global__ __launch_bounds__(256, 1) void sdf_mlp_f16_combined_nerf15_kernel(const DecodeParams p) { sdf_mlp_f16_body<true, 0, 8>(p); }

// ... over a voxel list (the exact values behind the one-plane sweeps of these decoders: k1s_nerf_kernels.hip)
__global__ __launch_bounds__(256, 1) void sdf_mlp_f16_subset_nerf9_kernel(const DecodeParams p) { sdf_mlp_f16_body<false, 0, 5, 2, 1, true>(p); }
__global__ __launch_bounds__(256, 1) void sdf_mlp_f16_subset_nerf15_kernel(const DecodeParams p) { sdf_mlp_f16_body<false, 0, 8, 2, 1, true>(p); }
__global__ __launch_bounds__(256, 1) void sdf_mlp_f16_subset_combined_nerf9_kernel(const DecodeParams p) { sdf_mlp_f16_body<true, 0, 5, 2, 1, true>(p); }
__global__ __launch_bounds__(256, 1) void sdf_mlp_f16_subset_combined_nerf15_kernel(const DecodeParams p) { sdf_mlp_f16_body<true, 0, 8, 2, 1, true>(p); }

hipError_t k1h_nerf_prepare() {
  hipError_t e = hipSuccess;
  for (const void* k : {(const void*)sdf_mlp_f16_nerf9_kernel, (const void*)sdf_mlp_f16_nerf15_kernel,
                        (const void*)sdf_mlp_f16_combined_nerf9_kernel, (const void*)sdf_mlp_f16_combined_nerf15_kernel,
                        (const void*)sdf_mlp_f16_subset_nerf9_kernel, (const void*)sdf_mlp_f16_subset_nerf15_kernel,
                        (const void*)sdf_mlp_f16_subset_combined_nerf9_kernel, (const void*)sdf_mlp_f16_subset_combined_nerf15_kernel})
    if (e == hipSuccess) e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes_f16(kMaxKP));
  return e;
}

void k1h_nerf_launch(int kp, bool two_out, const DecodeParams& p, int grid, hipStream_t st) {
  const int lds = lds_bytes_f16(kp);
  if (kp == 5) {
    if (two_out) hipLaunchKernelGGL(sdf_mlp_f16_combined_nerf9_kernel, dim3(grid), dim3(256), lds, st, p);
    else hipLaunchKernelGGL(sdf_mlp_f16_nerf9_kernel, dim3(grid), dim3(256), lds, st, p);
  } else {
    if (two_out) hipLaunchKernelGGL(sdf_mlp_f16_combined_nerf15_kernel, dim3(grid), dim3(256), lds, st, p);
    else hipLaunchKernelGGL(sdf_mlp_f16_nerf15_kernel, dim3(grid), dim3(256), lds, st, p);
  }
}

void k1h_nerf_subset_launch(int kp, bool two_out, const DecodeParams& p, int grid, hipStream_t st) {
  const int lds = lds_bytes_f16(kp);
  if (kp == 5) {
    if (two_out) hipLaunchKernelGGL(sdf_mlp_f16_subset_combined_nerf9_kernel, dim3(grid), dim3(256), lds, st, p);
    else hipLaunchKernelGGL(sdf_mlp_f16_subset_nerf9_kernel, dim3(grid), dim3(256), lds, st, p);
  } else {
    if (two_out) hipLaunchKernelGGL(sdf_mlp_f16_subset_combined_nerf15_kernel, dim3(grid), dim3(256), lds, st, p);
    else hipLaunchKernelGGL(sdf_mlp_f16_subset_nerf15_kernel, dim3(grid), dim3(256), lds, st, p);
  }
}

}  // namespace asdf
