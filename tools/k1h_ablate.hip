// Timing-only ablation / schedule sweep for the split-half decoder kernel (results are NOT checked here - parity lives
// in tests/).  Build (per knob setting):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -w -Ialignsdf_amd/csrc [-DASDF16_PREFETCH=2 ...] tools/k1h_ablate.hip -o tools/bin/k1h_x
// Run:  k1h_x [N] [data]     data = path of a tools/dump_k1h_inputs.py image (the product's real weights + folded
//                            constants), "zero" (all-zero operands: the DVFS upper bound) or "small" (default: small
//                            pseudo-random weights that keep every ablation finite)
// Besides the launch time the tool reports the shader clock the kernel ran at (s_memtime ticks of workgroup 0 / wall).
// Environment: K1H_STATUS=1 passes a status record like the product does (round 5: that is where three 64-lane LDS atomics per tile
// hid - 14 k clocks the tool never saw), K1H_NOBBOX=1 drops the negative-voxel fold.  Round 5: builds with -DASDF16_SEGMENT_TIMES
// fault in the fold's early return (the stamps keep an 8-entry array live across it; only this timing build, the product's kernel
// has no stamps and is covered by the parity tests) - run them with K1H_NOBBOX=1.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#ifdef ASDF16_SEGMENT_TIMES
__device__ unsigned long long g_seg[8];
#endif
#include "sdf_mlp_f16w_kernel.h"
using namespace asdf;
#ifndef ABL_LIST
#define ABL_LIST X(0) X(32) X(1) X(16) X(4)
#endif
#ifndef PLANES
#define PLANES 2        // 1 = the one-plane kernel of the box-only coarse sweep
#endif
#ifndef GROUPS
#define GROUPS 1        // 2 = two point groups per wave (one-plane kernel only)
#endif
#ifndef WIDE
#define WIDE 0          // 1 = the W form (16x16x32 MFMAs; timing only here: it is fed the 32x32x16 image - the same values in another order)
#endif
#if WIDE
#define K1H_BODY(n) sdf_mlp_f16w_body<n, false>(p);
#else
#define K1H_BODY(n) sdf_mlp_f16_body<false, n, 2, PLANES, GROUPS>(p);
#endif
constexpr int kLds = PLANES == 1 ? kLdsBytesF16P1 : kLdsBytesF16;
__device__ unsigned long long g_ticks[2];
#define X(n) __global__ __launch_bounds__(256, 1) void k_abl_##n(const DecodeParams p) { \
    unsigned long long t0 = __builtin_readcyclecounter(); K1H_BODY(n)  \
    if (blockIdx.x == 0 && threadIdx.x == 0) { g_ticks[0] = t0; g_ticks[1] = __builtin_readcyclecounter(); } }
ABL_LIST
#undef X
#ifdef ASDF16_SEGMENT_TIMES
// shader-clock cycles of the segments of one tile of one wave, next to the MFMA time the segment's instructions need
static void seg_report() {
  unsigned long long g[8];
  (void)hipMemcpyFromSymbol(g, HIP_SYMBOL(g_seg), sizeof(g));
  const double f = (PLANES == 1 ? 1.0 / 3.0 : 1.0) * GROUPS;      // (per workgroup tile: 128 x GROUPS points)
  const char* name[5] = {"coordinates + layer-0 tiles 0..7", "layer 1 (+ layer-0 tiles 8..15)", "layer 2", "layer 3", "last epilogue, tanh, stores, box fold"};
  const double ideal[5] = {8 * 2 * 64.0 * GROUPS, 8 * 32 * 96.0 * f + 8 * 2 * 64.0 * GROUPS, 16 * 16 * 96.0 * f + 16 * 2 * 64.0 * GROUPS, 16 * 32 * 96.0 * f, 0.0};
  double tot = 0, toti = 0;
  for (int k = 0; k < 5; ++k) {
    const double c = (double)(g[k + 1] - g[k]);
    printf("    %-40s %8.0f cycles   MFMA time of its instructions %8.0f   (%5.1f %%)\n", name[k], c, ideal[k], ideal[k] > 0 ? 100.0 * ideal[k] / c : 0.0);
    tot += c; toti += ideal[k];
  }
  printf("    (of the first segment: %.0f cycles until the coordinates are done)\n", (double)(g[7] - g[0]));
  printf("    %-40s %8.0f cycles   %8.0f   (%5.1f %%);  tile period (start to start of the next) %.0f cycles\n", "one tile of one MLP", tot, toti, 100.0 * toti / tot,
         (double)(g[6] - g[0]) / 1.0);
}
#define SEG_REPORT seg_report();
#else
#define SEG_REPORT
#endif
int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 128;
  const char* data = argc > 2 ? argv[2] : "small";
  const long long P = (long long)N * N * N;
  float *stream, *cst, *o0, *o1;
  std::vector<uint16_t> h((size_t)kStagesAll * kStageFloats * 2);
  std::vector<float> c(kHeads * kCstFloats);
  if (!strcmp(data, "small")) {
    for (size_t i = 0; i < h.size(); ++i) { _Float16 v = (_Float16)((float)((int)((i * 2654435761u) >> 20) % 2001 - 1000) * 2e-5f); h[i] = *(uint16_t*)&v; }
    for (size_t i = 0; i < c.size(); ++i) c[i] = (float)((int)((i * 40503u) >> 4) % 201 - 100) * 1e-3f;
  } else if (!strcmp(data, "zero")) {
    std::fill(h.begin(), h.end(), 0); std::fill(c.begin(), c.end(), 0.0f);
  } else {
    FILE* f = fopen(data, "rb");
    if (!f || fread(h.data(), 2, h.size(), f) != h.size() || fread(c.data(), 4, c.size(), f) != c.size()) { printf("cannot read %s\n", data); return 1; }
    fclose(f);
  }
  (void)hipMalloc(&stream, h.size() * 4); (void)hipMemcpy(stream, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  (void)hipMemcpy(reinterpret_cast<char*>(stream) + h.size() * 2, h.data(), h.size() * 2, hipMemcpyHostToDevice);      // (where the W form looks for its image)
  (void)hipMalloc(&cst, c.size() * 4); (void)hipMemcpy(cst, c.data(), c.size() * 4, hipMemcpyHostToDevice);
  (void)hipMalloc(&o0, P * 4); (void)hipMalloc(&o1, P * 4);
  int* bbox; (void)hipMalloc(&bbox, 64); (void)hipMemset(bbox, 0, 64);
  // K1H_STATUS=1: the decoder's status record as the product passes it (peak words, range report, clock stamps); K1H_NOBBOX=1: no box fold
  int* status = nullptr;
  if (getenv("K1H_STATUS")) { (void)hipMalloc(&status, 64); (void)hipMemset(status, 0, 64); }
  // one-plane kernels: the fp16 point-feature / bias operands (timing only: small values, T = 1)
  std::vector<uint16_t> a16h((size_t)kHeads * kA16Floats * 2);
  for (size_t i = 0; i < a16h.size(); ++i) { _Float16 v = strcmp(data, "zero") ? (_Float16)((float)((int)((i * 2654435761u) >> 20) % 201 - 100) * 1e-2f) : (_Float16)0.0f; a16h[i] = *(uint16_t*)&v; }
  for (int h = 0; h < kHeads; ++h) { float one = 1.0f; memcpy(&a16h[((size_t)h * kA16Floats + 2 * kA16LayerFloats) * 2], &one, 4); memcpy(&a16h[((size_t)h * kA16Floats + 2 * kA16LayerFloats + 1) * 2], &one, 4); }
  float* a16; (void)hipMalloc(&a16, a16h.size() * 2); (void)hipMemcpy(a16, a16h.data(), a16h.size() * 2, hipMemcpyHostToDevice);
  DecodeParams p{}; p.stream = stream; p.cst = cst; p.sdf0 = o0; p.sdf1 = o1; p.P = P; p.N = N; p.mode = kGridReference;
  p.vs = 2.0f / (N - 1); p.o0 = p.o1 = p.o2 = -1.f; p.num_mlps = 2; p.first_mlp = 0; p.bbox = getenv("K1H_NOBBOX") ? nullptr : bbox; p.a16 = a16; p.status = status;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const double flop = (double)P * 2 * 3145728.0 / (PLANES == 1 ? 3 : 1);
  printf("GROUPS %d  PLANES %d  PREFETCH %d  BARRIER_KB %d  data %s\n", GROUPS, PLANES, ASDF16_PREFETCH, ASDF16_BARRIER_KB, data);
#define X(n) { (void)hipFuncSetAttribute((const void*)k_abl_##n, hipFuncAttributeMaxDynamicSharedMemorySize, kLds); \
    float best = 1e9; double ghz = 0; for (int it = 0; it < 4; ++it) { (void)hipEventRecord(e0); hipLaunchKernelGGL(k_abl_##n, dim3(256), dim3(256), kLds, 0, p); \
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); \
      unsigned long long t[2]; (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(g_ticks), 16); \
      if (ms < best) { best = ms; ghz = (double)(t[1] - t[0]) / (ms * 1e6); } } \
    SEG_REPORT printf("ABL %2d  N=%d  %.3f ms  %.0f TF/s f16 MFMA (%.1f%% of 2516)  wg0 ticks/wall = %.3f GHz  err=%d\n", n, N, best, flop / best / 1e9, flop / best / 1e9 / 25.166, ghz, (int)hipGetLastError()); }
  ABL_LIST
#undef X
  return 0;
}
