// Microbenchmark (round 6): v_mfma_f32_16x16x32_f16 on split-half plane data, TWO accumulator chains that take turns every L MFMAs
// (L = 1, 2, 3, 6, 12, 24, 48, 96; the last = practically one chain): how long must an MFMA chain stay on ONE accumulator before the
// matrix pipe's forwarding pays?  Same harness as tools/mfma_f16_energy_bench.hip (operands in registers, one wave per SIMD, ~45 ms).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_chain_bench.hip -o tools/bin/mfma_chain_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kA = 8, kB = 32;
__device__ unsigned long long g_ticks[2];

template <int L>
__global__ __launch_bounds__(256, 1) void bench(float* out, const h8* a_in, const h8* b_in, int iters) {
  const int lane = threadIdx.x & 63;
  h8 a[kA], b[kB];
#pragma unroll
  for (int i = 0; i < kA; ++i) a[i] = a_in[i * 64 + lane];
#pragma unroll
  for (int i = 0; i < kB; ++i) b[i] = b_in[i * 64 + lane];
  f32x4 c0 = {0}, c1 = {0};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 192; ++k) {      // 192 MFMAs per iteration
#ifdef K1H_PATTERN
      // K1h's operand pattern: per (record, group) three MFMAs (W_hi, x_lo), (W_lo, x_hi), (W_hi, x_hi); a[2 r] / a[2 r + 1] = the planes
      // of record r's A fragment, b[2 m] / b[2 m + 1] = x_hi / x_lo of operand m
      const int rec = k / 3, ph = k % 3;
      const int ai = (rec % (kA / 2)) * 2 + (ph == 1 ? 1 : 0), bi = ((rec * 2) % kB) + (ph == 0 ? 1 : 0);
#else
      const int ai = (k / 3) % kA, bi = (k * 5) % kB;
#endif
      if ((k / L) & 1) c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ai], b[bi], c1, 0, 0, 0);
      else c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[ai], b[bi], c0, 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (blockIdx.x == 0 && threadIdx.x == 0) { g_ticks[0] = t0; g_ticks[1] = t1; }
  out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1];
}

static uint16_t rnd_half(uint32_t& st, bool relu, float lo, float hi) {
  st = st * 1664525u + 1013904223u;
  if (relu && (st >> 31)) return 0;
  st = st * 1664525u + 1013904223u;
  const float mag = lo + (hi - lo) * (float)((st >> 8) & 0xffff) / 65536.0f;
  const _Float16 v = (_Float16)(((st >> 30) & 1) ? -mag : mag);
  return *(const uint16_t*)&v;
}

template <int L>
static void run(float* out, h8* a_d, h8* b_d, int iters) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9; double ghz = 0;
  for (int r = 0; r < 3; ++r) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(bench<L>, dim3(256), dim3(256), 0, 0, out, a_d, b_d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long t[2]; (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(g_ticks), 16);
    if (ms < best) { best = ms; ghz = (double)(t[1] - t[0]) / (ms * 1e6); }
  }
  const double flop = (double)iters * 192 * 16384.0 * 1024;
  printf("turns every %3d MFMAs  %.2f ms  %.0f TFLOP/s (%.1f %% of 2516.6)  clock %.3f GHz  cycles per MFMA %.2f\n", L, best, flop / best / 1e9,
         flop / best / 1e9 / 25.166, ghz, best * 1e6 * ghz / ((double)iters * 192));
}

int main() {
  const int iters = 30000;
  float* out; (void)hipMalloc(&out, 256 * 256 * 4);
  h8 *a_d, *b_d;
  (void)hipMalloc(&a_d, kA * 64 * 16); (void)hipMalloc(&b_d, kB * 64 * 16);
  std::vector<uint16_t> ha(kA * 512), hb(kB * 512);
  uint32_t st = 777u;
  for (auto& v : ha) v = rnd_half(st, false, 8.0f, 1024.0f);
  for (auto& v : hb) v = rnd_half(st, true, 1.0f, 2048.0f);
  (void)hipMemcpy(a_d, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
  (void)hipMemcpy(b_d, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) {
    run<1>(out, a_d, b_d, iters); run<2>(out, a_d, b_d, iters); run<3>(out, a_d, b_d, iters); run<6>(out, a_d, b_d, iters);
    run<12>(out, a_d, b_d, iters); run<24>(out, a_d, b_d, iters); run<48>(out, a_d, b_d, iters); run<96>(out, a_d, b_d, iters);
  }
  return 0;
}
