// K1 on the fp32 MFMA: SeparateDecoder / CombinedDecoder, affine xyz features or the in-kernel NeRF encoding (PointFeatSize 9 / 15, utils/mesh.py:53-55).
#include "k1_launch.h"
#include "sdf_mlp_kernel.h"
#include "sdf_mlp_short_kernel.h"

namespace asdf {

__global__ __launch_bounds__(256, 1) void sdf_mlp_kernel(const DecodeParams p) { sdf_mlp_body<0, 2, false>(p); }
__global__ __launch_bounds__(256, 1) void sdf_mlp_combined_kernel(const DecodeParams p) { sdf_mlp_body<0, 2, true>(p); }
__global__ __launch_bounds__(256, 1) void sdf_mlp_nerf9_kernel(const DecodeParams p) { sdf_mlp_body<0, 5, false>(p); }
__global__ __launch_bounds__(256, 1) void sdf_mlp_nerf15_kernel(const DecodeParams p) { sdf_mlp_body<0, 8, false>(p); }
__global__ __launch_bounds__(256, 1) void sdf_mlp_combined_nerf9_kernel(const DecodeParams p) { sdf_mlp_body<0, 5, true>(p); }
__global__ __launch_bounds__(256, 1) void sdf_mlp_combined_nerf15_kernel(const DecodeParams p) { sdf_mlp_body<0, 8, true>(p); }

__global__ __launch_bounds__(256, 1) void sdf_mlp_short_kernel(const DecodeParams p, const ShortParams sp) { sdf_mlp_short_body<false>(p, sp); }
__global__ __launch_bounds__(256, 1) void sdf_mlp_short_combined_kernel(const DecodeParams p, const ShortParams sp) { sdf_mlp_short_body<true>(p, sp); }

hipError_t k1_prepare() {
  hipError_t e = hipSuccess;
  for (const void* k : {(const void*)sdf_mlp_short_kernel, (const void*)sdf_mlp_short_combined_kernel})
    if (e == hipSuccess) e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytesShort);
  for (const void* k : {(const void*)sdf_mlp_kernel, (const void*)sdf_mlp_combined_kernel})
    if (e == hipSuccess) e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
  for (const void* k : {(const void*)sdf_mlp_nerf9_kernel, (const void*)sdf_mlp_nerf15_kernel,
                        (const void*)sdf_mlp_combined_nerf9_kernel, (const void*)sdf_mlp_combined_nerf15_kernel})
    if (e == hipSuccess) e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes(kMaxKP));
  return e;
}

void k1_launch(int kp, bool two_out, const DecodeParams& p, int grid, hipStream_t st) {
  if (kp == 2) {
    if (two_out) hipLaunchKernelGGL(sdf_mlp_combined_kernel, dim3(grid), dim3(256), kLdsBytes, st, p);
    else hipLaunchKernelGGL(sdf_mlp_kernel, dim3(grid), dim3(256), kLdsBytes, st, p);
  } else if (kp == 5) {
    if (two_out) hipLaunchKernelGGL(sdf_mlp_combined_nerf9_kernel, dim3(grid), dim3(256), lds_bytes(5), st, p);
    else hipLaunchKernelGGL(sdf_mlp_nerf9_kernel, dim3(grid), dim3(256), lds_bytes(5), st, p);
  } else {
    if (two_out) hipLaunchKernelGGL(sdf_mlp_combined_nerf15_kernel, dim3(grid), dim3(256), lds_bytes(8), st, p);
    else hipLaunchKernelGGL(sdf_mlp_nerf15_kernel, dim3(grid), dim3(256), lds_bytes(8), st, p);
  }
}

// the short-list form: one workgroup per 32 listed points and MLP, p.short_max / 32 of them per MLP - or, for lists of up to
// sp.cluster_max points, four per block (sdf_mlp_short_kernel.h); workgroups beyond the list return
void k1_short_launch(bool two_out, const DecodeParams& p, const ShortParams& sp, hipStream_t st) {
  const int units = (p.short_max + kWavePts - 1) / kWavePts * p.num_mlps;
  const int clusters = ((sp.cluster_max + kWavePts - 1) / kWavePts * p.num_mlps + 7) / 8 * 8;
  const dim3 grid(units > clusters * kClusterWgs ? units : clusters * kClusterWgs);
  if (two_out) hipLaunchKernelGGL(sdf_mlp_short_combined_kernel, grid, dim3(256), kLdsBytesShort, st, p, sp);
  else hipLaunchKernelGGL(sdf_mlp_short_kernel, grid, dim3(256), kLdsBytesShort, st, p, sp);
}

}  // namespace asdf
