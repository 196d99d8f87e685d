// K1 with the part classifier riding in the last-layer epilogue: the label pass (utils/mesh.py:137-157).
#include "k1_launch.h"
#include "sdf_mlp_kernel.h"

namespace asdf {

__global__ __launch_bounds__(256, 1) void sdf_mlp_cls_kernel(const DecodeParams p) { sdf_mlp_body<0, 2, false, true>(p); }
__global__ __launch_bounds__(256, 1) void sdf_mlp_combined_cls_kernel(const DecodeParams p) { sdf_mlp_body<0, 2, true, true>(p); }
__global__ __launch_bounds__(256, 1) void sdf_mlp_nerf9_cls_kernel(const DecodeParams p) { sdf_mlp_body<0, 5, false, true>(p); }
__global__ __launch_bounds__(256, 1) void sdf_mlp_nerf15_cls_kernel(const DecodeParams p) { sdf_mlp_body<0, 8, false, true>(p); }
__global__ __launch_bounds__(256, 1) void sdf_mlp_combined_nerf9_cls_kernel(const DecodeParams p) { sdf_mlp_body<0, 5, true, true>(p); }
__global__ __launch_bounds__(256, 1) void sdf_mlp_combined_nerf15_cls_kernel(const DecodeParams p) { sdf_mlp_body<0, 8, true, true>(p); }

hipError_t k1_cls_prepare() {
  hipError_t e = hipSuccess;
  for (const void* k : {(const void*)sdf_mlp_cls_kernel, (const void*)sdf_mlp_combined_cls_kernel,
                        (const void*)sdf_mlp_nerf9_cls_kernel, (const void*)sdf_mlp_nerf15_cls_kernel,
                        (const void*)sdf_mlp_combined_nerf9_cls_kernel, (const void*)sdf_mlp_combined_nerf15_cls_kernel})
    if (e == hipSuccess) e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes_cls(kMaxKP));
  return e;
}

void k1_cls_launch(int kp, bool two_out, const DecodeParams& p, int grid, hipStream_t st) {
  const int lds = lds_bytes_cls(kp);
  if (kp == 2) {
    if (two_out) hipLaunchKernelGGL(sdf_mlp_combined_cls_kernel, dim3(grid), dim3(256), lds, st, p);
    else hipLaunchKernelGGL(sdf_mlp_cls_kernel, dim3(grid), dim3(256), lds, st, p);
  } else if (kp == 5) {
    if (two_out) hipLaunchKernelGGL(sdf_mlp_combined_nerf9_cls_kernel, dim3(grid), dim3(256), lds, st, p);
    else hipLaunchKernelGGL(sdf_mlp_nerf9_cls_kernel, dim3(grid), dim3(256), lds, st, p);
  } else {
    if (two_out) hipLaunchKernelGGL(sdf_mlp_combined_nerf15_cls_kernel, dim3(grid), dim3(256), lds, st, p);
    else hipLaunchKernelGGL(sdf_mlp_nerf15_cls_kernel, dim3(grid), dim3(256), lds, st, p);
  }
}

}  // namespace asdf
