// K1 (shared declarations): fused 5-layer SDF decoder over a dense grid (or an explicit point list), gfx950 only.
//
// Replaces, per chunk of the reference hot loop (utils/mesh.py:46-63,98-115):
//   grid-coordinate construction (utils/mesh.py:27-44,82-96),
//   latent expand + cat (utils/utils.py:568-569),
//   SeparateDecoder.forward - 10 GEMMs, ReLU, tanh (networks/model.py:285-350),
//   the negative-voxel bounding box of get_higher_res_cube (utils/mesh.py:208-237).
//
// Structure (see sdf_layout.h for the operand maps):
//   * one 256-thread workgroup per CU, one wave per SIMD, up to 512 VGPR+AGPR per lane;
//   * every wave owns 32 query points through ALL layers of one MLP: the 512-wide activation
//     of a layer lives in 256 registers per lane and is consumed in place as the MFMA B operand
//     of the next layer (no LDS / HBM round trip for activations);
//   * loop order: MLP (head) outer, point tile inner, so one 2 MiB weight stream is live at a time
//     and stays resident in the 4 MiB per-XCD L2;
//   * the weights are the MFMA A operand.  They are pre-packed on the host into a linear stream
//     of 16 KiB stages and flow L2 -> LDS through a 4-slot ring filled by LDS-DMA
//     (global_load_lds_dwordx4) 2.5 stages ahead, shared by the 4 waves; one s_barrier per stage,
//     placed mid-stage in the shadow of an MFMA;
//   * bias / ReLU / final dot-product + tanh are fused epilogues on the accumulator registers,
//     deferred into the MFMA stream of the next output tile.
// Two kernels share this structure: sdf_mlp_kernel.h (v_mfma_f32_32x32x2_f32, the description above) and
// sdf_mlp_f16_kernel.h (split-half: two fp16 planes per operand, three v_mfma_f32_32x32x16_f16 per product sum, 32 KiB
// stages) - the default for the grid sweeps.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sdf_layout.h"

namespace asdf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kRing = 4;
constexpr int kLdsRingFloats = kRing * kStageFloats;                 // 64 KiB
// LDS = weight ring + the constants block of the MLP being evaluated
constexpr int lds_bytes(int kp) { return (kLdsRingFloats + cst_offsets(kp).floats) * 4; }
constexpr int kLdsBytes = lds_bytes(2);
// part classifier of the label pass: up to kMaxClasses rows over the last hidden activation + their biases
constexpr int kMaxClasses = 8;
constexpr int kClsFloats = kMaxClasses * kHidden + kMaxClasses;
constexpr int lds_bytes_cls(int kp) { return lds_bytes(kp) + kClsFloats * 4; }

enum GridMode : int {
  kGridReference = 0,   // true-division ("sheared") indices of utils/mesh.py:33-34
  kGridInteger = 1,     // integer floor-division indices (what the code presumably intended)
  kPointList = 2,       // explicit xyz list
  kGridSubset = 3,      // listed lattice points of a grid (idx / count_dev / grid_mode): coordinates as the sweep, outputs in place
};

struct DecodeParams {
  const float* stream;      // [kStagesAll][kStageFloats] packed static weights
  const float* cst;         // [heads][CstLayout<KP>::kFloats] per-sample constants
  float* sdf0;              // [P] hand SDF (may be null)
  float* sdf1;              // [P] object SDF (may be null)
  const float* xyz;         // [P][3] when mode == kPointList
  int* bbox;                // [kHeads][8]: min0,min1,min2,max0,max1,max2,count,pad (or null)
  const int* idx;           // kGridSubset: [<= P] linear lattice indices
  const int* count_dev;     // kGridSubset: number of listed points (device word; P is the capacity)
  int short_max;            // kGridSubset, fp32 chain: lists of up to this many points belong to the short-list form
                            // (sdf_mlp_short_kernel.h) - it returns for longer ones, the tile form for these; 0 = no such split
  const int* short_fault;   // kGridSubset, fp32 chain: when non-null and *short_fault != 0 the TILE form takes the short lists as well - a
                            // cluster-form launch in front of it timed out waiting for a member and wrote nothing for that block
                            // (sdf_mlp_short_kernel.h: the recoverable failure of round 6); the results are the same bits
  int grid_mode;            // kGridSubset: kGridReference / kGridInteger of the lattice
  int* fixup_flag;          // kGridSubset with bbox: the outputs REPLACE earlier values - the box is patched in place (a voxel
                            // that turns negative extends it) and *fixup_flag is raised when one turns non-negative
  int* status;              // decoder-owned status record: [0] += lanes whose activations left the fp16 range (K1h only)
  int* audit;               // split-half kGridSubset only: the audit record of a one-plane sweep (or null).  List positions
                            // >= *audit_from (all of them when audit_from is null) are AUDIT picks - voxels the one-plane sweep
                            // decided by sign alone, drawn at random and re-evaluated to check that decision: [0] = largest
                            // |new - old| over them (float bits), [1] += picks whose sign changed, [2] += picks evaluated;
                            // the other positions report to status[3] as usual
  const int* audit_from;
  const float* a16;         // one-plane kernels (affine point features): [heads][kA16Floats] - the point-feature columns and bias rows
                            // of layers 0 and 2 as fp16 A operands of ONE v_mfma_f32_32x32x16_f16 per tile (fold_points_f16_kernel)
  float neg_thr;            // a voxel counts as negative for the fused box when sdf < neg_thr: 0 for the ordinary sweeps, -tau
                            // for the one-plane sweep of asdf_decode_grid_box (certainly negative); kGridSubset patches
                            // compare the value they replace against it
  long long P;              // number of query points
  int N;                    // grid resolution (P == N^3 for grid modes)
  int mode;
  float vs;                 // voxel size (fp32, as the reference rounds it)
  float o0, o1, o2;         // origin added to axis-0/1/2 coordinates
  const float* lattice;     // when non-null: {origin0, origin1, origin2, voxel size} in DEVICE memory replace o0 .. o2 / vs - the zoom
                            // cube of a fine pass that is enqueued right behind its coarse pass (asdf_zoom_cube writes it; round 5)
  int num_mlps;             // MLPs to evaluate: 2 = both heads of a SeparateDecoder, 1 = one head or a CombinedDecoder
  int first_mlp;            // index of the first MLP to evaluate (1 = object head only)
  int pf;                   // raw point-feature count (NeRF-feature kernels only)
  // label-pass kernels only
  const float* cls;         // [kMaxClasses][512 in D-layout order] + [kMaxClasses] biases
  float* logits;            // [P][num_class] class scores of MLP 0's last hidden activation (may be null)
  int* labels;              // [P] argmax of the scores (may be null)
  int num_class;
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

// The same with an instruction offset: the immediate is added to BOTH the global address and the LDS address, so the
// four 1 KiB pieces of a wave's share of a stage (equal 1024-byte strides on both sides) need one base address.
template <int OFF>
__device__ __forceinline__ void lds_dma16_off(const float* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off offset:%c3\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst), "i"(OFF)
      : "memory");
}

// Reference grid coordinates, bit-for-bit (utils/mesh.py:32-40): fp32 true division, fp32 fmod,
// then separately rounded multiply and add (no FMA contraction).
// fmodf(a, b) for a >= 0 and an integer-valued b >= 1 with a / b < 2^20, bit for bit: the remainder a - b trunc(a / b) is exactly
// representable, so one FMA produces it without rounding once trunc(a / b) is known, and a quotient estimate that is off by one
// (a rb carries ~2 ulp) leaves a remainder in [-b, 2 b) that one exact +-b repairs.  (The library fmodf is a shift-and-subtract loop:
// two of them, a 64-bit modulo and a 64-bit conversion per point were ~4 % of the one-plane kernel's cycles.)
__device__ __forceinline__ float fmod_small(float a, float b, float rb) {
  const float k = truncf(a * rb);
  float r = fmaf(-k, b, a);
  if (r < 0.0f) r += b;
  else if (r >= b) r -= b;
  return r;
}

__device__ __forceinline__ void grid_point(long long i, int N, int mode, float vs, float o0, float o1, float o2,
                                           float& c0, float& c1, float& c2) {
  float i0, i1, i2;
  if (N <= 1290) {
    // every lattice the C ABI accepts (N <= 1024): N^3 < 2^31, the index arithmetic fits 32 bits
    const unsigned u = (unsigned)i, n = (unsigned)N;
    if (mode == kGridReference) {
      const float Nf = (float)N, rN = __frcp_rn(Nf);
      const float fi = (float)(int)u;                 // int64 -> fp32 (RNE), as torch does (the same value: u < 2^31)
      const float q1 = __fdiv_rn(fi, Nf);             // overall_index / N
      i2 = (float)(u % n);
      i1 = fmod_small(q1, Nf, rN);
      i0 = fmod_small(__fdiv_rn(q1, Nf), Nf, rN);
    } else {
      const unsigned q = u / n;
      i2 = (float)(u - q * n);
      const unsigned q2 = q / n;
      i1 = (float)(q - q2 * n);
      i0 = (float)q2;
    }
  } else if (mode == kGridReference) {
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

// lattice index -> (i0, i1, i2) in 32-bit arithmetic (N <= 1024: every index is below 2^30)
__device__ __forceinline__ void lattice_ijk(long long i, int N, int& i0, int& i1, int& i2) {
  const unsigned u = (unsigned)i, n = (unsigned)N;
  const unsigned q = u / n, q2 = q / n;
  i2 = (int)(u - q * n); i1 = (int)(q - q2 * n); i0 = (int)q2;
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

}  // namespace asdf
