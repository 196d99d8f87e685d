// K1h: the fused decoder on split-half fp16 MFMAs (sdf_mlp_f16_kernel.h), the default arithmetic of the grid sweeps.
#include <cstdlib>

#include "k1_launch.h"
#include "sdf_mlp_f16_kernel.h"

namespace asdf {

__global__ __launch_bounds__(256, 1) void sdf_mlp_f16_kernel(const DecodeParams p) { sdf_mlp_f16_body<false>(p); }
__global__ __launch_bounds__(256, 1) void sdf_mlp_f16_combined_kernel(const DecodeParams p) { sdf_mlp_f16_body<true>(p); }

// the split-half arithmetic on a voxel list (the exact values of the narrow-band fine sweep)
__global__ __launch_bounds__(256, 1) void sdf_mlp_f16_subset_kernel(const DecodeParams p) { sdf_mlp_f16_body<false, 0, 2, 2, 1, true>(p); }
__global__ __launch_bounds__(256, 1) void sdf_mlp_f16_subset_combined_kernel(const DecodeParams p) { sdf_mlp_f16_body<true, 0, 2, 2, 1, true>(p); }

// Round 6: the W form of the SeparateDecoder kernels - the same GEMMs on v_mfma_f32_16x16x32_f16 (sdf_mlp_f16_kernel.h, "the W form"):
// under the part's power management the matrix pipe sustains ~13 % more of them per second.  The default for affine point features;
// ASDF_K1H_SHAPE=32 selects the 32x32x16 kernels (A/B runs; the CombinedDecoder and NeRF forms are 32x32x16 only).
// (the kernels live in k1hw_kernels.hip: their own translation unit, their own compiler flags)

static int g_k1h_shape = 0;      // 0 = not decided yet (the environment decides at the first question), 16, 32
int k1h_shape() {
  if (g_k1h_shape == 0) {
    const char* e = getenv("ASDF_K1H_SHAPE");
    g_k1h_shape = (e && e[0] == '3' && e[1] == '2') ? 32 : 16;
  }
  return g_k1h_shape;
}
int k1h_set_shape(int shape) {      // 16 / 32; 0 = back to the environment's choice; returns the shape in force before the call
  const int before = k1h_shape();
  if (shape == 16 || shape == 32 || shape == 0) g_k1h_shape = shape;
  return before;
}
static bool k1h_wide() { return k1h_shape() == 16; }

hipError_t k1h_prepare() {
  hipError_t e = k1h_nerf_prepare();
  for (const void* k : {(const void*)sdf_mlp_f16_kernel, (const void*)sdf_mlp_f16_combined_kernel, (const void*)sdf_mlp_f16_subset_kernel,
                        (const void*)sdf_mlp_f16_subset_combined_kernel})
    if (e == hipSuccess) e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytesF16);
  if (e == hipSuccess) e = k1hw_prepare();
  if (e == hipSuccess) e = k1s_prepare();
  return e;
}

void k1h_subset_launch(int kp, bool two_out, const DecodeParams& p, int grid, hipStream_t st) {
  if (kp != 2) { k1h_nerf_subset_launch(kp, two_out, p, grid, st); return; }
  if (two_out) hipLaunchKernelGGL(sdf_mlp_f16_subset_combined_kernel, dim3(grid), dim3(256), kLdsBytesF16, st, p);
  else if (k1h_wide()) k1hw_subset_launch(p, grid, st);
  else hipLaunchKernelGGL(sdf_mlp_f16_subset_kernel, dim3(grid), dim3(256), kLdsBytesF16, st, p);
}

void k1h_launch(int kp, bool two_out, const DecodeParams& p, int grid, hipStream_t st) {
  if (kp != 2) { k1h_nerf_launch(kp, two_out, p, grid, st); return; }
  if (two_out) hipLaunchKernelGGL(sdf_mlp_f16_combined_kernel, dim3(grid), dim3(256), kLdsBytesF16, st, p);
  else if (k1h_wide()) k1hw_launch(p, grid, st);
  else hipLaunchKernelGGL(sdf_mlp_f16_kernel, dim3(grid), dim3(256), kLdsBytesF16, st, p);
}

}  // namespace asdf
