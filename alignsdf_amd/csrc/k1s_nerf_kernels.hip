// K1s for the in-kernel NeRF encoding (PointFeatSize 9 / 15, utils/mesh.py:53-55): the ONE-PLANE kernel of the audited box-only /
// narrow-band sweeps (sdf_mlp_f16_kernel.h, PL = 1) with the point features on the fp32 MFMA (KP = 5 / 8 K-steps, sin / cos generated in
// the kernel).  The 40 / 75 KiB constants block of these decoders leaves room for the 4 x 16 KiB ring of the one-plane weight stream;
// built with their accumulators in VGPRs (build_native.py: TU_FLAGS) the SeparateDecoder forms also hold TWO point groups per wave
// (nerf9: no scratch; nerf15: 44 B outside the MFMA stream) - 63.5 -> 59.7 and 68.7 -> 63.8 ms per N = 256 sample against one group.
// The CombinedDecoder forms carry one group (their second output's last-layer state takes the registers).
// Round 4: until then NeRF-encoded decoders ran ordinary sweeps on both passes.
#include "k1_launch.h"
#include "sdf_mlp_f16_kernel.h"
#ifndef ASDF_NERF15_G
#define ASDF_NERF15_G 2
#endif
#ifndef ASDF_NERF9_G
#define ASDF_NERF9_G 2
#endif

namespace asdf {

__global__ __launch_bounds__(256, 1) void sdf_mlp_f16p1_nerf9_kernel(const DecodeParams p) { sdf_mlp_f16_body<false, 0, 5, 1, ASDF_NERF9_G>(p); }
__global__ __launch_bounds__(256, 1) void sdf_mlp_f16p1_nerf15_kernel(const DecodeParams p) { sdf_mlp_f16_body<false, 0, 8, 1, ASDF_NERF15_G>(p); }
__global__ __launch_bounds__(256, 1) void sdf_mlp_f16p1_combined_nerf9_kernel(const DecodeParams p) { sdf_mlp_f16_body<true, 0, 5, 1, 1>(p); }
__global__ __launch_bounds__(256, 1) void sdf_mlp_f16p1_combined_nerf15_kernel(const DecodeParams p) { sdf_mlp_f16_body<true, 0, 8, 1, 1>(p); }

hipError_t k1s_nerf_prepare() {
  hipError_t e = hipSuccess;
  for (const void* k : {(const void*)sdf_mlp_f16p1_nerf9_kernel, (const void*)sdf_mlp_f16p1_nerf15_kernel,
                        (const void*)sdf_mlp_f16p1_combined_nerf9_kernel, (const void*)sdf_mlp_f16p1_combined_nerf15_kernel})
    if (e == hipSuccess) e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes_f16(kMaxKP, 1));
  return e;
}

void k1s_nerf_launch(int kp, bool two_out, const DecodeParams& p, int grid, hipStream_t st) {
  const int lds = lds_bytes_f16(kp, 1);
  if (kp == 5) {
    if (two_out) hipLaunchKernelGGL(sdf_mlp_f16p1_combined_nerf9_kernel, dim3(grid), dim3(256), lds, st, p);
    else hipLaunchKernelGGL(sdf_mlp_f16p1_nerf9_kernel, dim3(grid), dim3(256), lds, st, p);
  } else {
    if (two_out) hipLaunchKernelGGL(sdf_mlp_f16p1_combined_nerf15_kernel, dim3(grid), dim3(256), lds, st, p);
    else hipLaunchKernelGGL(sdf_mlp_f16p1_nerf15_kernel, dim3(grid), dim3(256), lds, st, p);
  }
}

}  // namespace asdf
