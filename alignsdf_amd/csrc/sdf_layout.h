// Data layout shared by the host-side weight packer and the gfx950 decoder kernel.
//
// The SDF heads of AlignSDF (reference: networks/model.py:191-350, SeparateDecoder) are
// 5-layer, 512-wide MLPs.  The per-sample-constant latent columns of layer 0 and layer 2 are
// folded into per-sample biases (reference cat order: networks/model.py:311,330 and
// utils/utils.py:568-569), and the (affine) point embedding is folded into the 3 xyz columns,
// so one head, per query point, is
//
//   h0 = relu(A0 [512x4]   . (x,y,z,0)        + c0)      K =   4   (per-sample A0, c0)
//   h1 = relu(W1 [256x512] . h0               + b1)      K = 512   (rows >= n1 are zero)
//   h2 = relu(W2a[512x256] . h1 + A2[512x4].p + c2)      K = 256+4 (per-sample A2, c2)
//   h3 = relu(W3 [512x512] . h2               + b3)      K = 512
//   s  = tanh(w4 . h3 + b4)                               (CombinedDecoder: two rows w4, two outputs)
//
// Everything is laid out for v_mfma_f32_32x32x2_f32 with the WEIGHTS as the A operand
// (M = output features) and the POINTS as the B operand (N = 32 points per wave):
//   A: lane l holds A[i = l & 31][k = l >> 5]      (one f32 VGPR)
//   B: lane l holds B[k = l >> 5][j = l & 31]      (one f32 VGPR)
//   D: lane l, reg r holds D[row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][col = l & 31]
// Register r of an output tile is therefore directly a valid B operand of the next layer for
// the K pair {row(r, 0), row(r, 1)}: activations never leave the register file.
#pragma once
#include <stdint.h>

namespace asdf {

constexpr int kHidden = 512;            // width of every hidden layer
constexpr int kLatent = 256;            // latent code size (specs["LatentSize"])
constexpr int kTilesHidden = 16;        // 512 / 32 output tiles
constexpr int kTilesL1 = 8;             // layer-1 output padded to 256 rows
constexpr int kWavePts = 32;            // query points per wave (one 32-wide MFMA column block)
constexpr int kWaves = 4;               // one wave per SIMD
constexpr int kWgPts = kWavePts * kWaves;

// Weight stream: fixed-size stages of 64 K-steps (K = 128) of ONE 32-row output tile.
// stage image = [g = 0..15][lane = 0..63][j = 0..3] floats; K-step s = 4 g + j.
constexpr int kStageKSteps = 64;
constexpr int kStageFloats = kStageKSteps * 64;     // 4096 floats
constexpr int kStageBytes = kStageFloats * 4;       // 16 KiB
constexpr int kStagesL1 = kTilesL1 * 4;             // K = 512 -> 4 stages per tile
constexpr int kStagesL2 = kTilesHidden * 2;         // K = 256 -> 2 stages per tile
constexpr int kStagesL3 = kTilesHidden * 4;
constexpr int kStagesHead = kStagesL1 + kStagesL2 + kStagesL3;   // 128
constexpr int kHeads = 2;
constexpr int kStagesAll = kStagesHead * kHeads;                  // 256 stages = 4 MiB

// Per-head constants block (floats), resident in LDS.  KP = number of point-feature K-steps of layers 0 and 2:
// 2 when the point features are (an affine function of) xyz, ceil(pf / 2) for the NeRF positional encoding.
//   frag arrays: [tile][kstep = 0..KP-1][lane]  (A operand images)
//   bias arrays: [tile][half = 0..1][r = 0..15]  (D-layout order)
template <int KP>
struct CstLayout {
  static constexpr int kA0 = 0;                                   // 16 * KP * 64
  static constexpr int kA2 = kA0 + kTilesHidden * KP * 64;        // 16 * KP * 64
  static constexpr int kC0 = kA2 + kTilesHidden * KP * 64;        // 512
  static constexpr int kB1 = kC0 + kHidden;                       // 256
  static constexpr int kC2 = kB1 + kTilesL1 * 32;                 // 512
  static constexpr int kB3 = kC2 + kHidden;                       // 512
  static constexpr int kW4 = kB3 + kHidden;                       // 512
  static constexpr int kW4b = kW4 + kHidden;                      // 512: second output row (CombinedDecoder), zero otherwise
  static constexpr int kB4 = kW4b + kHidden;                      // 8: b4 of output 0 / 1; split-half image: [2] mul1, [3] mul2,
                                                                  //    [4] mul0 (accumulator -> next planes multipliers), [5..7] pad
  static constexpr int kFloats = kB4 + 8;
};
constexpr int kCstFloats = CstLayout<2>::kFloats;                 // 6920 floats = 27 680 B (xyz features)
constexpr int kMaxKP = 8;                                          // NeRF encoding up to PointFeatSize 15
constexpr int kCstFloatsMax = CstLayout<kMaxKP>::kFloats;

// runtime view of the same offsets (host packer, fold kernel)
struct CstOffsets {
  int a0, a2, c0, b1, c2, b3, w4, w4b, b4, floats;
};
__host__ __device__ constexpr CstOffsets cst_offsets(int kp) {
  CstOffsets o{};
  o.a0 = 0;
  o.a2 = o.a0 + kTilesHidden * kp * 64;
  o.c0 = o.a2 + kTilesHidden * kp * 64;
  o.b1 = o.c0 + kHidden;
  o.c2 = o.b1 + kTilesL1 * 32;
  o.b3 = o.c2 + kHidden;
  o.w4 = o.b3 + kHidden;
  o.w4b = o.w4 + kHidden;
  o.b4 = o.w4b + kHidden;
  o.floats = o.b4 + 8;
  return o;
}

// One-plane kernels, affine point features: layers 0 and 2 take their K = 4 point-feature products AND their bias row from one
// fp16 MFMA (K = 16) instead of a bias load + two fp32 MFMAs.  Per head: [layer 0 | layer 2][tile 0..15][lane 0..63][8 halves],
// then T0, T2 (the power of two the values are divided by; the point operand is multiplied by it) + pad.
//   A lane l (row i = l & 31), half 0 (k 0..7):  w_hi[0..2], w_hi[0..2], c_hi, c_lo        w = point-feature column, c = bias
//                              half 1 (k 8..15): w_lo[0..2], 0, 0, 0, 0, 0
//   B lane l (point j = l & 31), half 0:          x_hi[0..2], x_lo[0..2], T, T              x_hi + x_lo = x T in two fp16 planes
//                              half 1:            x_hi[0..2], 0, 0, 0, 0, 0
// = (w_hi + w_lo) x + c to 2^-22 of each term (the w_lo x_lo products are dropped): more than the kernel's other operands carry.
constexpr int kA16TileFloats = 64 * 4;                                  // 64 lanes x 8 halves
constexpr int kA16LayerFloats = kTilesHidden * kA16TileFloats;          // 4096
constexpr int kA16Floats = 2 * kA16LayerFloats + 8;                     // + T0, T2, pad

// feature row held by (register r, lane half h) of a 32x32 D tile
__host__ __device__ constexpr int tile_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
// input feature consumed by K-step s (s = 16 * in_tile + r) on lane half h
__host__ __device__ constexpr int kstep_feature(int s, int h) { return 32 * (s >> 4) + tile_row(s & 15, h); }

}  // namespace asdf
