// Timing-only ablation / schedule sweep for the split-half decoder kernel (results are NOT checked here - parity lives
// in tests/).  Build (per knob setting):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Ialignsdf_amd/csrc [-DASDF16_PREFETCH=2 -DASDF16_BARRIER_KB=5] tools/k1h_ablate.hip -o /tmp/k1h
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "sdf_mlp_f16_kernel.h"
using namespace asdf;
#ifndef ABL_LIST
#define ABL_LIST X(0) X(32) X(1) X(16) X(4)
#endif
#define X(n) __global__ __launch_bounds__(256, 1) void k_abl_##n(const DecodeParams p) { sdf_mlp_f16_body<false, n>(p); }
ABL_LIST
#undef X
int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 128;
  const long long P = (long long)N * N * N;
  float *stream, *cst, *o0, *o1;
  std::vector<uint16_t> h((size_t)kStagesAll * kStageFloats * 2);
  for (size_t i = 0; i < h.size(); ++i) { _Float16 v = (_Float16)((float)((int)((i * 2654435761u) >> 20) % 2001 - 1000) * 2e-5f); h[i] = *(uint16_t*)&v; }   // small weights: activations stay finite in every ablation
  hipMalloc(&stream, h.size() * 2); hipMemcpy(stream, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  std::vector<float> c(kHeads * kCstFloats);
  for (size_t i = 0; i < c.size(); ++i) c[i] = (float)((int)((i * 40503u) >> 4) % 201 - 100) * 1e-3f;
  hipMalloc(&cst, c.size() * 4); hipMemcpy(cst, c.data(), c.size() * 4, hipMemcpyHostToDevice);
  hipMalloc(&o0, P * 4); hipMalloc(&o1, P * 4);
  DecodeParams p{}; p.stream = stream; p.cst = cst; p.sdf0 = o0; p.sdf1 = o1; p.P = P; p.N = N; p.mode = kGridReference;
  p.vs = 2.0f / (N - 1); p.o0 = p.o1 = p.o2 = -1.f; p.num_mlps = 2; p.first_mlp = 0;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double flop = (double)P * 2 * 3145728.0;
  printf("PREFETCH %d  BARRIER_KB %d\n", ASDF16_PREFETCH, ASDF16_BARRIER_KB);
#define X(n) { hipFuncSetAttribute((const void*)k_abl_##n, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytesF16); \
    float best = 1e9; for (int it = 0; it < 4; ++it) { hipEventRecord(e0); hipLaunchKernelGGL(k_abl_##n, dim3(256), dim3(256), kLdsBytesF16, 0, p); \
      hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; } \
    printf("ABL %2d  N=%d  %.3f ms  %.0f TF/s f16 MFMA (%.1f%% of 2516)  err=%d\n", n, N, best, flop / best / 1e9, flop / best / 1e9 / 25.166, (int)hipGetLastError()); }
  ABL_LIST
#undef X
  return 0;
}
