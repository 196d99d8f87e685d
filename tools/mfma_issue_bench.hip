// Microbenchmark: sustained issue rate of v_mfma_f32_32x32x2_f32 under the operand / interleave patterns the
// fused decoder uses.  One 256-thread workgroup per CU, one wave per SIMD.  Prints cycles per MFMA per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_issue_bench.hip -o tools/mfma_issue_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
constexpr int kIters = 2000, kUnroll = 64;

// V: 0 one chain, const operands | 1 two chains | 2 one chain, A from LDS (ds_read_b128 per 4 MFMA, prefetched 2 ahead)
//    3 one chain, B cycling through 64 registers | 4 = 2 + 3 | 5 = 4 with two chains
template <int V>
__global__ __launch_bounds__(256, 1) void bench(float* out, const float* in) {
  __shared__ __attribute__((aligned(16))) float lds[16384];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = in[i & 1023];
  __syncthreads();
  f32x16 acc0 = {0}, acc1 = {0};
  float b[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) b[i] = in[lane + i];
  float a = in[lane];
  const f32x4* a4 = reinterpret_cast<const f32x4*>(lds) + lane;
  for (int it = 0; it < kIters; ++it) {
    f32x4 abuf[18];
    if (V == 2 || V >= 4) { abuf[0] = a4[0]; abuf[1] = a4[64]; }
#pragma unroll
    for (int g = 0; g < kUnroll / 4; ++g) {
      if (V == 2 || V >= 4) abuf[g + 2] = a4[((g + 2) & 15) * 64];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = g * 4 + j;
        const float av = (V == 2 || V >= 4) ? abuf[g][j] : a;
        const float bv = (V >= 3) ? b[k & 63] : b[0];
        if ((V == 1 || V == 5) && (j & 1)) acc1 = MFMA(av, bv, acc1);
        else acc0 = MFMA(av, bv, acc0);
      }
    }
  }
  f32x16 r = acc0 + acc1;
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += r[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int V>
void run(float* out, float* in, double ghz) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9;
  for (int r = 0; r < 3; ++r) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(bench<V>, dim3(256), dim3(256), 0, 0, out, in);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  const double n = (double)kIters * kUnroll;
  printf("variant %d: %.3f ms  %.2f ns/MFMA/SIMD = %.1f cycles at %.2f GHz;  %.1f TF/s\n", V, best, best * 1e6 / n, best * 1e6 / n * ghz, ghz,
         n * 1024 * 2.0 * 32 * 32 * 2 / (best * 1e-3) / 1e12);
}

int main() {
  float *out, *in;
  (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&in, 4096 * 4);
  float h[4096]; for (int i = 0; i < 4096; ++i) h[i] = (float)((i * 37) % 201 - 100) * 1e-4f;
  (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  const double ghz = 2.4;
  run<0>(out, in, ghz); run<1>(out, in, ghz); run<2>(out, in, ghz); run<3>(out, in, ghz); run<4>(out, in, ghz); run<5>(out, in, ghz);
  return 0;
}
