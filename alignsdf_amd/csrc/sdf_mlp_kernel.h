// K1: fused 5-layer SDF decoder over a dense grid (or an explicit point list), gfx950 only.
//
// Replaces, per chunk of the reference hot loop (utils/mesh.py:46-63,98-115):
//   grid-coordinate construction (utils/mesh.py:27-44,82-96),
//   latent expand + cat (utils/utils.py:568-569),
//   SeparateDecoder.forward - 10 GEMMs, ReLU, tanh (networks/model.py:285-350),
//   the negative-voxel bounding box of get_higher_res_cube (utils/mesh.py:208-237).
//
// Structure (see sdf_layout.h for the operand maps):
//   * one 256-thread workgroup per CU, one wave per SIMD, up to 512 VGPR+AGPR per lane;
//   * every wave owns 32 query points for BOTH heads and ALL layers: the 512-wide activation
//     of a layer lives in 256 registers per lane and is consumed in place as the MFMA B operand
//     of the next layer (no LDS / HBM round trip for activations);
//   * the weights are the MFMA A operand.  They are pre-packed on the host into a linear stream
//     of 16 KiB stages and flow HBM/L2 -> LDS through a 4-slot ring filled by LDS-DMA
//     (global_load_lds_dwordx4), shared by the 4 waves; one s_barrier per stage;
//   * bias / ReLU / final dot-product + tanh are fused epilogues on the accumulator registers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sdf_layout.h"

namespace asdf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kRing = 4;
constexpr int kLdsRingFloats = kRing * kStageFloats;                 // 64 KiB
constexpr int kLdsFloats = kLdsRingFloats + kHeads * kCstFloats;     // + 51 232 B
constexpr int kLdsBytes = kLdsFloats * 4;

enum GridMode : int {
  kGridReference = 0,   // true-division ("sheared") indices of utils/mesh.py:33-34
  kGridInteger = 1,     // integer floor-division indices (what the code presumably intended)
  kPointList = 2,       // explicit xyz list
};

struct DecodeParams {
  const float* stream;      // [kStagesAll][kStageFloats] packed static weights
  const float* cst;         // [kHeads][kCstFloats] per-sample constants
  float* sdf0;              // [P] hand SDF (may be null)
  float* sdf1;              // [P] object SDF (may be null)
  const float* xyz;         // [P][3] when mode == kPointList
  int* bbox;                // [kHeads][8]: min0,min1,min2,max0,max1,max2,count,pad (or null)
  long long P;              // number of query points
  int N;                    // grid resolution (P == N^3 for grid modes)
  int mode;
  float vs;                 // voxel size (fp32, as the reference rounds it)
  float o0, o1, o2;         // origin added to axis-0/1/2 coordinates
  int heads_mask;           // bit h set -> evaluate head h
};

__device__ __forceinline__ void lds_dma16(const float* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}

// Reference grid coordinates, bit-for-bit (utils/mesh.py:32-40): fp32 true division, fp32 fmod,
// then separately rounded multiply and add (no FMA contraction).
__device__ __forceinline__ void grid_point(long long i, int N, int mode, float vs, float o0, float o1, float o2,
                                           float& c0, float& c1, float& c2) {
  float i0, i1, i2;
  if (mode == kGridReference) {
    const float Nf = (float)N;
    const float fi = (float)i;                      // int64 -> fp32 (RNE), as torch does
    const float q1 = __fdiv_rn(fi, Nf);             // overall_index / N
    i2 = (float)(i % N);
    i1 = fmodf(q1, Nf);
    i0 = fmodf(__fdiv_rn(q1, Nf), Nf);
  } else {
    i2 = (float)(i % N);
    i1 = (float)((i / N) % N);
    i0 = (float)((i / N) / N);
  }
  c0 = __fadd_rn(__fmul_rn(i0, vs), o0);
  c1 = __fadd_rn(__fmul_rn(i1, vs), o1);
  c2 = __fadd_rn(__fmul_rn(i2, vs), o2);
}

#define ASDF_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ f32x16 load_bias16(const float* lds_bias) {
  const f32x4* p = reinterpret_cast<const f32x4*>(lds_bias);
  f32x4 a = p[0], b = p[1], c = p[2], d = p[3];
  f32x16 v;
  v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
  v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
  v[8] = c[0]; v[9] = c[1]; v[10] = c[2]; v[11] = c[3];
  v[12] = d[0]; v[13] = d[1]; v[14] = d[2]; v[15] = d[3];
  return v;
}

__device__ __forceinline__ f32x16 relu16(f32x16 v) {
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.0f);
  return v;
}

// One weight-stream stage: wait for it, hand the slot of the previous stage back to the DMA
// engine, then run its 64 K-steps on `acc`.  KT = number of input tiles (registers = 16 KT),
// Q = stage number within the output tile (K-steps 64 Q .. 64 Q + 63).
template <int KT, int Q, int SLOT>
__device__ __forceinline__ void stage(f32x16& acc, const f32x16 (&hin)[KT], const float* ring,
                                      const float* next_src, unsigned lds_ring_base, int lane, int wave) {
  // my 4 pieces of this stage were issued 3 stages ago: at most 8 younger loads may stay in flight
  asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  constexpr int nslot = (SLOT + kRing - 1) % kRing;   // slot of stage (this - 1) == (this + 3)
  const float* src = next_src + wave * 1024 + lane * 4;
  const unsigned dst = lds_ring_base + (nslot * kStageFloats + wave * 1024) * 4;
  const f32x4* a4 = reinterpret_cast<const f32x4*>(ring + SLOT * kStageFloats) + lane;
  // A fragments are read two groups (8 K-steps) ahead of their MFMAs
  f32x4 abuf[18];
  abuf[0] = a4[0];
  abuf[1] = a4[64];
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    if (g + 2 < 16) abuf[g + 2] = a4[(g + 2) * 64];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int s = Q * 64 + g * 4 + j;
      acc = ASDF_MFMA(abuf[g][j], hin[s >> 4][s & 15], acc);
      // the 4 DMA pieces of stage (this + 3) go into the shadow of the first MFMAs
      if (g == 0) lds_dma16(src + j * 256, dst + j * 1024);
    }
  }
}

__global__ __launch_bounds__(256, 1) void sdf_mlp_kernel(const DecodeParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* ring = smem;
  float* cst = smem + kLdsRingFloats;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;

  const long long ntiles = (p.P + kWgPts - 1) / kWgPts;
  if ((long long)blockIdx.x >= ntiles) return;

  // per-sample constants -> LDS (once per workgroup)
  for (int i = tid; i < kHeads * kCstFloats / 4; i += 256)
    reinterpret_cast<f32x4*>(cst)[i] = reinterpret_cast<const f32x4*>(p.cst)[i];
  __syncthreads();

  const unsigned lds_ring_base = (unsigned)(size_t)(__attribute__((address_space(3))) float*)ring;

  // prologue: stages 0..2 in flight
#pragma unroll
  for (int s = 0; s < kRing - 1; ++s) {
    const float* src = p.stream + (size_t)s * kStageFloats + wave * 1024 + lane * 4;
    const unsigned dst = lds_ring_base + (s * kStageFloats + wave * 1024) * 4;
#pragma unroll
    for (int c = 0; c < 4; ++c) lds_dma16(src + c * 256, dst + c * 1024);
  }

  int bmin0 = 0x7fffffff, bmin1 = 0x7fffffff, bmin2 = 0x7fffffff, bmax0 = -1, bmax1 = -1, bmax2 = -1;
  int omin0 = 0x7fffffff, omin1 = 0x7fffffff, omin2 = 0x7fffffff, omax0 = -1, omax1 = -1, omax2 = -1;
  int bcnt = 0, ocnt = 0;

  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long pi = tile * kWgPts + wave * kWavePts + (lane & 31);
    const bool valid = pi < p.P;
    float x0 = 0.f, x1 = 0.f, x2 = 0.f;
    if (p.mode == kPointList) {
      if (valid) { x0 = p.xyz[pi * 3 + 0]; x1 = p.xyz[pi * 3 + 1]; x2 = p.xyz[pi * 3 + 2]; }
    } else {
      grid_point(valid ? pi : 0, p.N, p.mode, p.vs, p.o0, p.o1, p.o2, x0, x1, x2);
    }
    // B operands of the two xyz K-steps: lane half 0 supplies k = 0 / 2, half 1 supplies k = 1 / 3
    const float bx0 = half ? x1 : x0;
    const float bx1 = half ? 0.0f : x2;

#pragma unroll 1
    for (int head = 0; head < kHeads; ++head) {
      const float* hc = cst + head * kCstFloats;
      // source of stage (s + 3) relative to this head's first stage, wrapping to the other head
      const float* sbase = p.stream + (size_t)head * kStagesHead * kStageFloats;
      const float* swrap = p.stream + (size_t)(1 - head) * kStagesHead * kStageFloats;
      auto src_of = [&](int s) -> const float* {   // s = stage index within head + 3
        return s < kStagesHead ? sbase + (size_t)s * kStageFloats : swrap + (size_t)(s - kStagesHead) * kStageFloats;
      };

      // ---- layer 0: K = 4 (xyz + zero pad), per-sample A fragments from LDS
      f32x16 h0[kTilesHidden];
#pragma unroll
      for (int t = 0; t < kTilesHidden; ++t) {
        f32x16 acc = load_bias16(hc + kCstC0 + (t * 2 + half) * 16);
        acc = ASDF_MFMA(hc[kCstA0 + (t * 2 + 0) * 64 + lane], bx0, acc);
        acc = ASDF_MFMA(hc[kCstA0 + (t * 2 + 1) * 64 + lane], bx1, acc);
        h0[t] = relu16(acc);
      }

      // ---- layer 1: 512 -> 256 (rows >= n1 are zero padding)
      f32x16 h1[kTilesL1];
#pragma unroll
      for (int t = 0; t < kTilesL1; ++t) {
        f32x16 acc = load_bias16(hc + kCstB1 + (t * 2 + half) * 16);
        constexpr int S0 = 0;
        stage<16, 0, 0>(acc, h0, ring, src_of(S0 + t * 4 + 0 + 3), lds_ring_base, lane, wave);
        stage<16, 1, 1>(acc, h0, ring, src_of(S0 + t * 4 + 1 + 3), lds_ring_base, lane, wave);
        stage<16, 2, 2>(acc, h0, ring, src_of(S0 + t * 4 + 2 + 3), lds_ring_base, lane, wave);
        stage<16, 3, 3>(acc, h0, ring, src_of(S0 + t * 4 + 3 + 3), lds_ring_base, lane, wave);
        h1[t] = relu16(acc);
      }

      // ---- layer 2: [h1 (256) | xyz (4)] -> 512
      f32x16 h2[kTilesHidden];
#pragma unroll
      for (int t = 0; t < kTilesHidden; ++t) {
        f32x16 acc = load_bias16(hc + kCstC2 + (t * 2 + half) * 16);
        acc = ASDF_MFMA(hc[kCstA2 + (t * 2 + 0) * 64 + lane], bx0, acc);
        acc = ASDF_MFMA(hc[kCstA2 + (t * 2 + 1) * 64 + lane], bx1, acc);
        constexpr int S0 = kStagesL1;
        if (t & 1) {
          stage<8, 0, 2>(acc, h1, ring, src_of(S0 + t * 2 + 0 + 3), lds_ring_base, lane, wave);
          stage<8, 1, 3>(acc, h1, ring, src_of(S0 + t * 2 + 1 + 3), lds_ring_base, lane, wave);
        } else {
          stage<8, 0, 0>(acc, h1, ring, src_of(S0 + t * 2 + 0 + 3), lds_ring_base, lane, wave);
          stage<8, 1, 1>(acc, h1, ring, src_of(S0 + t * 2 + 1 + 3), lds_ring_base, lane, wave);
        }
        h2[t] = relu16(acc);
      }

      // ---- layer 3 (512 -> 512) fused with layer 4 (dot with w4) and tanh
      float part = 0.0f;
#pragma unroll
      for (int t = 0; t < kTilesHidden; ++t) {
        f32x16 acc = load_bias16(hc + kCstB3 + (t * 2 + half) * 16);
        constexpr int S0 = kStagesL1 + kStagesL2;
        stage<16, 0, 0>(acc, h2, ring, src_of(S0 + t * 4 + 0 + 3), lds_ring_base, lane, wave);
        stage<16, 1, 1>(acc, h2, ring, src_of(S0 + t * 4 + 1 + 3), lds_ring_base, lane, wave);
        stage<16, 2, 2>(acc, h2, ring, src_of(S0 + t * 4 + 2 + 3), lds_ring_base, lane, wave);
        stage<16, 3, 3>(acc, h2, ring, src_of(S0 + t * 4 + 3 + 3), lds_ring_base, lane, wave);
        const f32x16 w = load_bias16(hc + kCstW4 + (t * 2 + half) * 16);
#pragma unroll
        for (int r = 0; r < 16; ++r) part = fmaf(fmaxf(acc[r], 0.0f), w[r], part);
      }
      part += __shfl_xor(part, 32);
      const float sdf = tanhf(part + hc[kCstB4]);

      float* out = head == 0 ? p.sdf0 : p.sdf1;
      if (valid && half == 0 && out) out[pi] = sdf;

      if (p.bbox && valid && half == 0 && sdf < 0.0f && p.mode != kPointList) {
        const int i2 = (int)(pi % p.N), i1 = (int)((pi / p.N) % p.N), i0 = (int)((pi / p.N) / p.N);
        if (head == 0) {
          bmin0 = min(bmin0, i0); bmin1 = min(bmin1, i1); bmin2 = min(bmin2, i2);
          bmax0 = max(bmax0, i0); bmax1 = max(bmax1, i1); bmax2 = max(bmax2, i2); ++bcnt;
        } else {
          omin0 = min(omin0, i0); omin1 = min(omin1, i1); omin2 = min(omin2, i2);
          omax0 = max(omax0, i0); omax1 = max(omax1, i1); omax2 = max(omax2, i2); ++ocnt;
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  if (p.bbox) {
    if (bcnt) {
      atomicMin(p.bbox + 0, bmin0); atomicMin(p.bbox + 1, bmin1); atomicMin(p.bbox + 2, bmin2);
      atomicMax(p.bbox + 3, bmax0); atomicMax(p.bbox + 4, bmax1); atomicMax(p.bbox + 5, bmax2);
      atomicAdd(p.bbox + 6, bcnt);
    }
    if (ocnt) {
      atomicMin(p.bbox + 8, omin0); atomicMin(p.bbox + 9, omin1); atomicMin(p.bbox + 10, omin2);
      atomicMax(p.bbox + 11, omax0); atomicMax(p.bbox + 12, omax1); atomicMax(p.bbox + 13, omax2);
      atomicAdd(p.bbox + 14, ocnt);
    }
  }
}

}  // namespace asdf
