// Does the wave's TRAPSTS.EXCP field (sticky IEEE exception flags: [0] invalid, [1] input denormal, [2] div0, [3] overflow, [4] underflow,
// [5] inexact) record an fp32 -> fp16 conversion that overflows - and a matrix-pipe inf - inf?  If it does, the one-plane kernels can
// learn about an activation that left the fp16 range from ONE s_getreg per tile instead of a running maximum per register pair.
//   hipcc --offload-arch=gfx950 -O2 tools/trapsts_probe.hip -o tools/bin/trapsts_probe && tools/bin/trapsts_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned trapsts() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_TRAPSTS, 0, 9)" : "=s"(v));
  return v;
}
__device__ __forceinline__ void clear_trapsts() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_TRAPSTS, 0, 9), 0" ::: "memory"); }

__global__ void probe(const float* in, unsigned* out, float* sink) {
  const int lane = threadIdx.x;
  clear_trapsts();
  unsigned t0 = trapsts();
  // 1: a conversion that does not overflow
  f2 a; a[0] = in[0]; a[1] = in[1];
  h2 r = __builtin_convertvector(a, h2);
  asm volatile("" :: "v"(r));
  asm volatile("s_nop 7\n s_nop 7");
  unsigned t1 = trapsts();
  // 2: lane 5 converts 1e6 (overflows to +inf)
  f2 b; b[0] = lane == 5 ? in[2] : in[0]; b[1] = in[1];
  h2 q = __builtin_convertvector(b, h2);
  asm volatile("" :: "v"(q));
  asm volatile("s_nop 7\n s_nop 7");
  unsigned t2 = trapsts();
  clear_trapsts();
  // 3: matrix pipe: inf - inf
  h8 A, B;
  for (int e = 0; e < 8; ++e) { A[e] = (_Float16)(e & 1 ? -1.0f : 1.0f); B[e] = q[0]; }      // lane 5's column: +inf, -inf, ...
  f32x16 acc = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc, 0, 0, 0);
  asm volatile("s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7");
  sink[lane] = acc[0];
  unsigned t3 = trapsts();
  if (lane == 0) { out[0] = t0; out[1] = t1; out[2] = t2; out[3] = t3; }
  if (lane == 5) { out[4] = __float_as_uint(acc[0]); out[5] = __builtin_bit_cast(unsigned short, q[0]); }
}

int main() {
  float h_in[3] = {1.5f, 1000.0f, 1.0e6f};
  float* d_in; unsigned* d_out; float* d_sink;
  hipMalloc(&d_in, sizeof(h_in)); hipMalloc(&d_out, 8 * 4); hipMalloc(&d_sink, 64 * 4);
  hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice);
  hipMemset(d_out, 0, 32);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_in, d_out, d_sink);
  unsigned h_out[8];
  hipMemcpy(h_out, d_out, 32, hipMemcpyDeviceToHost);
  printf("TRAPSTS.EXCP after clear %03x | after in-range cvt_pk %03x | after an overflowing cvt_pk in one lane %03x | after an MFMA whose column holds +inf and -inf %03x\n",
         h_out[0], h_out[1], h_out[2], h_out[3]);
  printf("lane 5: MFMA result bits %08x (a NaN with the sign bit %s), converted half %04x\n", h_out[4], (h_out[4] >> 31) ? "SET" : "clear", h_out[5]);
  return 0;
}
