// K1s: the ONE-PLANE kernel of the audited box-only coarse sweep and narrow-band fine sweep (asdf_decode_grid_box / _band) - K1h's body
// (sdf_mlp_f16_kernel.h) with one fp16 plane per operand and one MFMA per product sum; the dominant kernel of a sample.
//
// Its own translation unit since round 4 because it is compiled with `-mllvm -amdgpu-mfma-vgpr-form` (build_native.py: TU_FLAGS): the
// MFMAs then take their accumulators in VGPRs and the fp16 activation planes - their B operands, which only the matrix pipe reads -
// are what the allocator parks in the AGPR half of the register file.  The deferred epilogues (v_cvt_pk / c + |c|) read the
// accumulators where they are instead of through 16 v_accvgpr_read per 32 x 32 tile: 1 862 -> 70 reads per tile body (340 writes
// instead of 232), 26.4 -> 25.2 ms per N = 256 sweep in a same-box A/B.  The split-half kernels (k1h_kernels.hip) are 0.5 % SLOWER in
// that form and stay on the compiler's default.
#include "k1_launch.h"
#include "sdf_mlp_f16_kernel.h"

namespace asdf {

// (SeparateDecoder: two point groups per wave - 256 points per workgroup tile, every A fragment feeds two MFMAs)
__global__ __launch_bounds__(256, 1) void sdf_mlp_f16p1_kernel(const DecodeParams p) { sdf_mlp_f16_body<false, 0, 2, 1, 2>(p); }
__global__ __launch_bounds__(256, 1) void sdf_mlp_f16p1_combined_kernel(const DecodeParams p) { sdf_mlp_f16_body<true, 0, 2, 1>(p); }

hipError_t k1s_prepare() {
  hipError_t e = hipSuccess;
  for (const void* k : {(const void*)sdf_mlp_f16p1_kernel, (const void*)sdf_mlp_f16p1_combined_kernel})
    if (e == hipSuccess) e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytesF16P1);
  return e;
}

void k1h_box_launch(bool two_out, const DecodeParams& p, int grid, hipStream_t st) {
  if (two_out) hipLaunchKernelGGL(sdf_mlp_f16p1_combined_kernel, dim3(grid), dim3(256), kLdsBytesF16P1, st, p);
  else hipLaunchKernelGGL(sdf_mlp_f16p1_kernel, dim3(grid), dim3(256), kLdsBytesF16P1, st, p);
}

}  // namespace asdf
