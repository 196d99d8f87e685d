// Probe for v_mfma_f32_32x32x16_f16 on gfx950: (1) operand layout, (2) fp16 subnormal inputs, (3) issue rate of one
// dependent accumulator chain vs two / four chains with one wave per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_f16_probe tools/mfma_f16_probe.hip && /tmp/mfma_f16_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void layout_kernel(const _Float16* A /*[32][16]*/, const _Float16* B /*[16][32]*/, float* D /*[32][32]*/) {
  const int lane = threadIdx.x, half = lane >> 5, col = lane & 31;
  h8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = A[col * 16 + 8 * half + e];        // assumed: lane holds A[i = lane & 31][k = 8 (lane >> 5) + e]
    b[e] = B[(8 * half + e) * 32 + col];      // assumed: lane holds B[k = 8 (lane >> 5) + e][j = lane & 31]
  }
  f16v c = {};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + col] = c[r];
}

template <int CHAINS>
__global__ __launch_bounds__(256, 1) void rate_kernel(float* out, int iters, long long* cycles) {
  h8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * (threadIdx.x + e)); b[e] = (_Float16)(0.002f * (threadIdx.x + 3 * e)); }
  f16v acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) acc[c] = {};
  const long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 48 / CHAINS; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) {
        a[c & 7] += (_Float16)1e-3f;       // vary the operand a little (constant operands take a slow path)
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[c], 0, 0, 0);
      }
  }
  const long long t1 = clock64();
  float s = 0;
  for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

int main() {
  std::vector<_Float16> A(32 * 16), B(16 * 32);
  for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) A[i * 16 + k] = (_Float16)(float)((i * 7 + k * 3) % 11 - 5);
  for (int k = 0; k < 16; ++k) for (int j = 0; j < 32; ++j) B[k * 32 + j] = (_Float16)(float)((k * 5 + j * 2 + k * j) % 13 - 6);
  _Float16 *dA, *dB; float* dD;
  hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dD, 32 * 32 * 4);
  hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
  layout_kernel<<<1, 64>>>(dA, dB, dD);
  std::vector<float> D(32 * 32);
  hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
    float ref = 0;
    for (int k = 0; k < 16; ++k) ref += (float)A[i * 16 + k] * (float)B[k * 32 + j];
    if (ref != D[i * 32 + j]) ++bad;
  }
  printf("layout: %d mismatches of 1024 (0 = the assumed A/B/D maps are right)\n", bad);

  // subnormal inputs: A = 2^-20 (fp16 subnormal), B = 2^10 -> product 2^-10 if subnormals are honoured, 0 if flushed
  for (auto& v : A) v = (_Float16)0.0f;
  for (auto& v : B) v = (_Float16)0.0f;
  for (int i = 0; i < 32; ++i) A[i * 16 + 0] = (_Float16)9.5367431640625e-07f;
  for (int j = 0; j < 32; ++j) B[0 * 32 + j] = (_Float16)1024.0f;
  hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
  layout_kernel<<<1, 64>>>(dA, dB, dD);
  hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
  printf("subnormal A (2^-20) x 2^10: D = %g (expected %g if fp16 subnormal inputs are honoured)\n", D[0], 9.5367431640625e-07 * 1024.0);
  for (auto& v : A) v = (_Float16)0.0f;
  for (auto& v : B) v = (_Float16)0.0f;
  for (int i = 0; i < 32; ++i) A[i * 16 + 0] = (_Float16)1024.0f;
  for (int j = 0; j < 32; ++j) B[0 * 32 + j] = (_Float16)9.5367431640625e-07f;
  hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
  layout_kernel<<<1, 64>>>(dA, dB, dD);
  hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
  printf("subnormal B (2^-20) x 2^10: D = %g\n", D[0]);

  float* out; long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
  long long h;
  const int iters = 2000;
  rate_kernel<1><<<256, 256>>>(out, iters, cyc); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  rate_kernel<1><<<256, 256>>>(out, iters, cyc); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("1 chain : %.1f cycles / MFMA (clock64 ticks, 100 MHz-class counter scaled? raw %lld)\n", (double)h / (iters * 48.0), h);
  rate_kernel<2><<<256, 256>>>(out, iters, cyc); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("2 chains: %.1f ticks / MFMA\n", (double)h / (iters * 48.0));
  rate_kernel<4><<<256, 256>>>(out, iters, cyc); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("4 chains: %.1f ticks / MFMA\n", (double)h / (iters * 48.0));
  // wall-clock rate over the whole chip
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int chains = 1; chains <= 4; chains *= 2) {
    hipEventRecord(e0);
    if (chains == 1) rate_kernel<1><<<256, 256>>>(out, iters, cyc);
    else if (chains == 2) rate_kernel<2><<<256, 256>>>(out, iters, cyc);
    else rate_kernel<4><<<256, 256>>>(out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = 256.0 * 4 * iters * 48.0 * 2 * 32 * 32 * 16;
    printf("%d chain(s): %.3f ms -> %.0f TFLOP/s f16 MFMA, %.1f ns per MFMA per SIMD\n", chains, ms, flop / ms * 1e-9, ms * 1e6 / (iters * 48.0));
  }
  return 0;
}
