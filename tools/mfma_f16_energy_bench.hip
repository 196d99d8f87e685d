// Microbenchmark (round 6): what the fp16 matrix pipe SUSTAINS under the part's power management, by instruction shape and by
// operand data.  K1h is power-managed (profiles/r02_k1h_power_limit.txt: the same binary runs zero operands at 2.35 GHz and real ones
// at 1.7-1.9 GHz), so the question behind every remaining idea is "does it cost less ENERGY", not "does it cost fewer clocks".
// One 256-thread workgroup per CU, one wave per SIMD, operands in registers only (no LDS, no memory): the bare instruction.
//   shape 0: v_mfma_f32_32x32x16_f16, one accumulator chain (K1h's instruction)
//   shape 1: v_mfma_f32_16x16x32_f16, four accumulator chains (same 16 accumulator registers)
//   shapes 2 / 3: K1h's operand pattern on 16x16x32 (two chains, alternating) / on 32x32x16; 4: 16x16x32 as ONE fully dependent chain;
//   5: the W form's shipped order (a group's three MFMAs back to back, the groups in snake order)
//   data  0: zeros | 1: random fp16 of K1h's plane magnitudes (A: +-[512, 1024) mantissas random; B: half of the values zero = ReLU)
//         2: as 1 with the B operand's low 5 mantissa bits cleared | 3: as 1 with A AND B low 5 mantissa bits cleared
// Each launch runs ~60 ms so that the clock settles; prints ms, TFLOP/s and the shader clock held (s_memtime ticks / wall).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_f16_energy_bench.hip -o tools/bin/mfma_f16_energy_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kA = 8, kB = 32;      // operand registers cycled through (K1h: 2 A fragments live, 32 K-blocks of activations)
__device__ unsigned long long g_ticks[2];

template <int SHAPE>
__global__ __launch_bounds__(256, 1) void bench(float* out, const h8* a_in, const h8* b_in, int iters) {
  const int lane = threadIdx.x & 63;
  h8 a[kA], b[kB];
#pragma unroll
  for (int i = 0; i < kA; ++i) a[i] = a_in[i * 64 + lane];
#pragma unroll
  for (int i = 0; i < kB; ++i) b[i] = b_in[i * 64 + lane];
  f32x16 acc = {0};
  f32x4 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < kB; ++k) {
      if (SHAPE == 0) {
        // three MFMAs per K-block on one accumulator, like K1h (hi.lo, lo.hi, hi.hi -> here: three operand pairs)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(k) % kA], b[k], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(k + 1) % kA], b[(k + 7) % kB], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(k) % kA], b[(k + 7) % kB], acc, 0, 0, 0);
      } else if (SHAPE == 2) {
        // K1h's operand pattern on 16x16x32, TWO chains (one per group of 16 points), alternating: a[2j] = W_hi, a[2j+1] = W_lo of A
        // fragment j; b[2m] = x_hi, b[2m+1] = x_lo of (K-block, group) m.  Per A fragment pair: 3 MFMAs per group.
        const int j = (k % (kA / 2)) * 2, m0 = (2 * k) % kB, m1 = (2 * k + 2) % kB;
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j], b[m0 + 1], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j], b[m1 + 1], c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j + 1], b[m0], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j + 1], b[m1], c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j], b[m0], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j], b[m1], c1, 0, 0, 0);
      } else if (SHAPE == 4) {
        // 16x16x32, ONE fully dependent chain (what does a dependent 4-pass MFMA cost back to back?)
        const int j = (k % (kA / 2)) * 2, m0 = (2 * k) % kB, m1 = (2 * k + 2) % kB;
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j], b[m0 + 1], c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j + 1], b[m0], c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j], b[m0], c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j], b[m1 + 1], c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j + 1], b[m1], c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j], b[m1], c0, 0, 0, 0);
      } else if (SHAPE == 5) {
        // the W form's shipped order: a group's three MFMAs back to back, the groups in snake order (chains of 6 across records)
        const int j = (k % (kA / 2)) * 2, m0 = (2 * k) % kB, m1 = (2 * k + 2) % kB;
        if (k & 1) {
          c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j], b[m1 + 1], c1, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j + 1], b[m1], c1, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j], b[m1], c1, 0, 0, 0);
          c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j], b[m0 + 1], c0, 0, 0, 0);
          c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j + 1], b[m0], c0, 0, 0, 0);
          c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j], b[m0], c0, 0, 0, 0);
        } else {
          c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j], b[m0 + 1], c0, 0, 0, 0);
          c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j + 1], b[m0], c0, 0, 0, 0);
          c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j], b[m0], c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j], b[m1 + 1], c1, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j + 1], b[m1], c1, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j], b[m1], c1, 0, 0, 0);
        }
      } else if (SHAPE == 3) {
        // the same on 32x32x16: (W_hi, x_lo), (W_lo, x_hi), (W_hi, x_hi) - exactly K1h's K-block
        const int j = (k % (kA / 2)) * 2, m0 = (2 * k) % kB;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j], b[m0 + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j + 1], b[m0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j], b[m0], acc, 0, 0, 0);
      } else {
        // the same FLOPs: six 16x16x32 on four chains
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(k) % kA], b[k], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(k) % kA], b[(k + 3) % kB], c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(k + 1) % kA], b[(k + 7) % kB], c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(k + 1) % kA], b[(k + 9) % kB], c3, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(k + 2) % kA], b[(k + 7) % kB], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(k + 2) % kA], b[(k + 9) % kB], c1, 0, 0, 0);
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (blockIdx.x == 0 && threadIdx.x == 0) { g_ticks[0] = t0; g_ticks[1] = t1; }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i];
  s += c0[0] + c1[1] + c2[2] + c3[3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

static uint16_t rnd_half(uint32_t& st, bool relu, int clear_bits, float scale_lo, float scale_hi) {
  st = st * 1664525u + 1013904223u;
  if (relu && (st >> 31)) return 0;
  st = st * 1664525u + 1013904223u;
  const float mag = scale_lo + (scale_hi - scale_lo) * (float)((st >> 8) & 0xffff) / 65536.0f;
  const _Float16 v = (_Float16)(((st >> 30) & 1) ? -mag : mag);
  uint16_t bits = *(const uint16_t*)&v;
  bits &= (uint16_t)~((1u << clear_bits) - 1);
  return bits;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 30000;
  float* out; (void)hipMalloc(&out, 256 * 256 * 4);
  h8 *a_d, *b_d;
  (void)hipMalloc(&a_d, kA * 64 * 16); (void)hipMalloc(&b_d, kB * 64 * 16);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int data = 0; data < 5; ++data) {
    std::vector<uint16_t> ha(kA * 64 * 8), hb(kB * 64 * 8);
    uint32_t st = 12345u;
    for (auto& v : ha) v = data == 0 ? 0 : rnd_half(st, false, data == 3 ? 5 : 0, 8.0f, 1024.0f);
    for (auto& v : hb) v = data == 0 ? 0 : rnd_half(st, true, data >= 2 ? 5 : 0, 1.0f, 2048.0f);
    if (data == 4) {
      // split-half planes as K1h holds them: register 2 j = the high plane (A: |w| S in [8, 1024); B: relu, half zero, up to 2048),
      // register 2 j + 1 = the LOW plane of the same values (|lo| <= half an ulp of hi, random mantissa; zero where hi is zero)
      st = 777u;
      for (int r = 0; r < kA; r += 2)
        for (int i = 0; i < 512; ++i) {
          const uint16_t hi = rnd_half(st, false, 0, 8.0f, 1024.0f); _Float16 hv; *(uint16_t*)&hv = hi;
          const float ulp = (float)hv * (1.0f / 2048.0f);
          st = st * 1664525u + 1013904223u;
          const _Float16 lo = (_Float16)(ulp * ((float)((st >> 8) & 0xffff) / 65536.0f - 0.5f));
          ha[r * 512 + i] = hi; ha[(r + 1) * 512 + i] = *(const uint16_t*)&lo;
        }
      for (int r = 0; r < kB; r += 2)
        for (int i = 0; i < 512; ++i) {
          const uint16_t hi = rnd_half(st, true, 0, 1.0f, 2048.0f); _Float16 hv; *(uint16_t*)&hv = hi;
          const float ulp = (float)hv * (1.0f / 2048.0f);
          st = st * 1664525u + 1013904223u;
          const _Float16 lo = (_Float16)(ulp * ((float)((st >> 8) & 0xffff) / 65536.0f - 0.5f));
          hb[r * 512 + i] = hi; hb[(r + 1) * 512 + i] = hi ? *(const uint16_t*)&lo : 0;
        }
    }
    (void)hipMemcpy(a_d, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(b_d, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep)
      for (int shape = 0; shape < 6; ++shape) {
        float best = 1e9; double ghz = 0;
        for (int r = 0; r < 3; ++r) {
          (void)hipEventRecord(e0);
          if (shape == 0) hipLaunchKernelGGL(bench<0>, dim3(256), dim3(256), 0, 0, out, a_d, b_d, iters);
          else if (shape == 1) hipLaunchKernelGGL(bench<1>, dim3(256), dim3(256), 0, 0, out, a_d, b_d, iters);
          else if (shape == 2) hipLaunchKernelGGL(bench<2>, dim3(256), dim3(256), 0, 0, out, a_d, b_d, iters);
          else if (shape == 3) hipLaunchKernelGGL(bench<3>, dim3(256), dim3(256), 0, 0, out, a_d, b_d, iters);
          else if (shape == 4) hipLaunchKernelGGL(bench<4>, dim3(256), dim3(256), 0, 0, out, a_d, b_d, iters);
          else hipLaunchKernelGGL(bench<5>, dim3(256), dim3(256), 0, 0, out, a_d, b_d, iters);
          (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
          float ms; (void)hipEventElapsedTime(&ms, e0, e1);
          unsigned long long t[2]; (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(g_ticks), 16);
          if (ms < best) { best = ms; ghz = (double)(t[1] - t[0]) / (ms * 1e6); }
        }
        const double flop = (double)iters * kB * 3 * 32768.0 * 1024;      // per launch: 1024 SIMDs
        printf("data %d  %-10s  %.2f ms  %.0f TFLOP/s (%.1f %% of 2516.6)  clock %.3f GHz  cycles per 32 K flop %.1f\n", data,
               shape == 0 ? "32x32x16" : shape == 1 ? "16x16x32" : shape == 2 ? "16x16 K1h" : shape == 3 ? "32x32 K1h" : shape == 4 ? "16x16 chain" : "16x16 ord3", best, flop / best / 1e9, flop / best / 1e9 / 25.166, ghz, best * 1e6 * ghz / ((double)iters * kB * 3));
      }
  }
  return 0;
}
