// Timing-only ablation harness for the fused decoder kernel (results are NOT checked here - parity lives
// in tests/).  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Ialignsdf_amd/csrc tools/k1_ablate.hip -o tools/k1_ablate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "sdf_mlp_kernel.h"
using namespace asdf;
#ifndef ABL_LIST
#define ABL_LIST X(0) X(1) X(16) X(4) X(8) X(13)
#endif
#define X(n) __global__ __launch_bounds__(256, 1) void k_abl_##n(const DecodeParams p) { sdf_mlp_body<n, 2, false>(p); }
ABL_LIST
#undef X
int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 128;
  const long long P = (long long)N * N * N;
  float *stream, *cst, *o0, *o1;
  std::vector<float> h((size_t)kStagesAll * kStageFloats);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((int)((i * 2654435761u) >> 20) % 2001 - 1000) * 2e-5f;
  hipMalloc(&stream, h.size() * 4); hipMemcpy(stream, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  std::vector<float> c(kHeads * kCstFloats);
  for (size_t i = 0; i < c.size(); ++i) c[i] = (float)((int)((i * 40503u) >> 4) % 201 - 100) * 1e-3f;
  hipMalloc(&cst, c.size() * 4); hipMemcpy(cst, c.data(), c.size() * 4, hipMemcpyHostToDevice);
  hipMalloc(&o0, P * 4); hipMalloc(&o1, P * 4);
  DecodeParams p{}; p.stream = stream; p.cst = cst; p.sdf0 = o0; p.sdf1 = o1; p.P = P; p.N = N; p.mode = kGridReference;
  p.vs = 2.0f / (N - 1); p.o0 = p.o1 = p.o2 = -1.f; p.num_mlps = 2; p.first_mlp = 0;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double flop = (double)P * 2 * 1057792.0;
#define X(n) { hipFuncSetAttribute((const void*)k_abl_##n, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes); \
    float best = 1e9; for (int it = 0; it < 4; ++it) { hipEventRecord(e0); hipLaunchKernelGGL(k_abl_##n, dim3(256), dim3(256), kLdsBytes, 0, p); \
      hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; } \
    printf("ABL %2d  N=%d  %.3f ms  %.1f TF/s executed-equivalent (%.1f%% of 157.3)  err=%d\n", n, N, best, flop / best / 1e9, flop / best / 1e9 / 1.573, (int)hipGetLastError()); }
  ABL_LIST
#undef X
  return 0;
}
