// Host-only packing of decoder weights into the layouts of sdf_layout.h (no device calls).
#pragma once
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/alignsdf_hip.h"
#include "sdf_layout.h"

namespace asdf {

struct HostPack {
  std::vector<float> stream;   // [kStagesAll][kStageFloats]
  std::vector<float> wlat;     // [heads][2][512][256]   latent columns of layers 0 and 2
  std::vector<float> wpt;      // [heads][2][512][ASDF_MAX_POINT_FEATS]  point columns of layers 0 and 2
  std::vector<float> b02;      // [heads][2][512]
  std::vector<float> cst;      // [heads][cst_offsets(kp).floats]  static parts filled, per-sample parts zero
  int kp;                      // point-feature K-steps of layers 0 / 2
  std::vector<float> emb;      // [heads][ASDF_MAX_POINT_FEATS][4]  identity-on-xyz default
  // split-half image (ASDF_MATH_F16X3; see sdf_mlp_f16_kernel.h)
  std::vector<uint16_t> stream16;   // [kStagesAll][kStageFloats * 2] fp16 bits: stage = [kblock 8][plane 2][lane 64][8]
  std::vector<uint16_t> stream16w;  // the same weights for the W form (16x16x32 MFMAs): records (tile, K32-block, feature half)
  std::vector<float> cst16;         // the constants block with the scaled entries of the split-half kernel
  float s2[kHeads];                 // scale of the layer-2 accumulator (K0 applies it to c2 / A2)
  float sw[kHeads][3];              // S_w of layers 1..3 (max |w| S_w in [512, 1024))
};

constexpr float kActScale = 8.0f;   // default S_x: activations are carried as x * S_x in the fp16 planes
// target of the calibrated S_x (host side, hip_decoder.py): the largest plane value of a layer over the calibration sweep
// lands in [1024, 2048) - a factor 32 to 64 below the fp16 maximum - and activations down to 2^-13 of the layer maximum keep
// a normal low plane (two full planes = 22 significand bits)
constexpr float kActTarget = 2048.0f;

inline uint16_t f16_bits(float x) {
  const _Float16 h = (_Float16)x;   // round to nearest even
  uint16_t u;
  std::memcpy(&u, &h, 2);
  return u;
}
inline float f16_round(float x) { return (float)(_Float16)x; }

// weight scale of a layer ([rows][cols] with leading dimension ld): the power of two that puts max |w| * S in [512, 1024)
inline float weight_scale(const float* w, int rows, int cols, int ld) {
  float m = 0.0f;
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < cols; ++c) m = std::fmax(m, std::fabs(w[(size_t)r * ld + c]));
  if (!(m > 0.0f) || !std::isfinite(m)) return 1.0f;
  return std::exp2(std::floor(std::log2(1000.0f / m)));
}

// The static constants of the split-half image for given scales.  cst = the fp32 constants image (unscaled b1, b3, w4, w4b and,
// for the NeRF encoding, the static point fragments A2), sw = S_w of layers 1..3, sx = S_x of the three activation vectors
// h0, h1, h2 (powers of two).  With acc_l = S_w,l S_x,l-1 (W_l x + b_l):
//   b1 * (sw1 sx0),  b3 * (sw3 sx2),  w4 / (sw3 sx2),  s2 = sw2 sx1 (K0 scales the folded c2 / A2 by it),
//   mul0 = sx0,  mul1 = sx1 / (sw1 sx0),  mul2 = sx2 / (sw2 sx1)      (accumulator -> next planes, exact powers of two).
// Per-sample entries (A0, A2, c0, c2) are K0's and are left alone here (A2 of the NeRF image is static: scaled here).
inline void scale_constants_f16(const asdf_decoder_spec_t& spec, int kp, const float* cst, const float (*sw)[3], const float (*sx)[3],
                                float* cst16, float* s2) {
  const CstOffsets co = cst_offsets(kp);
  for (int h = 0; h < spec.num_heads; ++h) {
    const float* c = cst + (size_t)h * co.floats;
    float* d = cst16 + (size_t)h * co.floats;
    const float s1 = sw[h][0] * sx[h][0], s3 = sw[h][2] * sx[h][2];
    s2[h] = sw[h][1] * sx[h][1];
    for (int i = 0; i < kTilesL1 * 32; ++i) d[co.b1 + i] = c[co.b1 + i] * s1;
    for (int i = 0; i < kHidden; ++i) {
      d[co.b3 + i] = c[co.b3 + i] * s3;
      d[co.w4 + i] = c[co.w4 + i] / s3;
      d[co.w4b + i] = c[co.w4b + i] / s3;
    }
    d[co.b4] = c[co.b4];
    d[co.b4 + 1] = c[co.b4 + 1];
    d[co.b4 + 2] = sx[h][1] / s1;
    d[co.b4 + 3] = sx[h][2] / s2[h];
    d[co.b4 + 4] = sx[h][0];
    if (spec.feature_mode == ASDF_FEATURES_NERF)
      for (int i = 0; i < kTilesHidden * kp * 64; ++i) d[co.a2 + i] = c[co.a2 + i] * s2[h];
  }
}

// Split-half image of the hidden layers: every weight w of layers 1-3 is carried as two fp16 planes of w * S_w
// (hi = fp16(w S_w), lo = fp16(w S_w - hi)); the activations are carried the same way as x * S_x.  Three fp16 MFMAs
// (hi.lo + lo.hi + hi.hi, the lo.lo term is below fp32 resolution) accumulate S_w S_x (W x) in fp32; biases and the
// fp32 point-feature products enter the same accumulator pre-multiplied by S_w S_x, and the (exact, power-of-two)
// rescale happens when the accumulator is turned into the next layer's planes.
inline bool pack_decoder_f16(const asdf_decoder_spec_t& spec, const asdf_head_params_t* heads, HostPack& hp) {
  const CstOffsets co = cst_offsets(hp.kp);
  try {
    hp.stream16.assign((size_t)kStagesAll * kStageFloats * 2, 0);
    hp.stream16w.assign((size_t)kStagesAll * kStageFloats * 2, 0);
    hp.cst16 = hp.cst;
  } catch (...) {
    return false;
  }
  for (int h = 0; h < spec.num_heads; ++h) {
    const int in = kLatent + spec.point_feats[h];
    const int n1 = kHidden - in;
    const float* W1 = heads[h].w[1]; const float* W2 = heads[h].w[2]; const float* W3 = heads[h].w[3]; const float* W4 = heads[h].w[4];
    const float sw1 = weight_scale(W1, n1, kHidden, kHidden), sw2 = weight_scale(W2, kHidden, n1, kHidden),
                sw3 = weight_scale(W3, kHidden, kHidden, kHidden);
    uint16_t* sp = &hp.stream16[(size_t)h * kStagesHead * kStageFloats * 2];
    auto pack = [&](int ntiles, int stages_per_tile, float sw, auto weight_at) {
      for (int t = 0; t < ntiles; ++t)
        for (int q = 0; q < stages_per_tile; ++q) {
          for (int kb = 0; kb < 8; ++kb)
            for (int lane = 0; lane < 64; ++lane)
              for (int e = 0; e < 8; ++e) {
                // K-block kbg of the input = half of input tile kbg >> 1: D registers 8 (kbg & 1) + e of that tile
                const int kbg = q * 8 + kb;
                const int feat = 32 * (kbg >> 1) + tile_row(8 * (kbg & 1) + e, lane >> 5);
                const int row = 32 * t + (lane & 31);
                const float w = weight_at(row, feat) * sw;
                const float hi = f16_round(w);
                sp[((kb * 2 + 0) * 64 + lane) * 8 + e] = f16_bits(hi);
                sp[((kb * 2 + 1) * 64 + lane) * 8 + e] = f16_bits(w - hi);
              }
          sp += kStageFloats * 2;
        }
    };
    pack(kTilesL1, 4, sw1, [&](int row, int feat) { return row < n1 ? W1[(size_t)row * kHidden + feat] : 0.0f; });
    pack(kTilesHidden, 2, sw2, [&](int row, int feat) { return feat < n1 ? W2[(size_t)row * kHidden + feat] : 0.0f; });
    pack(kTilesHidden, 4, sw3, [&](int row, int feat) { return W3[(size_t)row * kHidden + feat]; });
    // The W form's image (sdf_mlp_f16_kernel.h, "the W form"): the same planes, the same record count and stage structure, but a
    // record i of a tile is (feature half fh, K32-block j) = (i / 16, i % 16) in layers 1 and 3, (i % 2, i / 2) in layer 2, as the A operand of
    // v_mfma_f32_16x16x32_f16 - lane l: row 32 t + 16 fh + (l & 15), slot (q = l >> 4, e) = input feature 32 j + 16 (e >> 2) + 4 q
    // + (e & 3), the feature the producing layer's epilogue leaves in that slot of the B operand.
    uint16_t* wp = &hp.stream16w[(size_t)h * kStagesHead * kStageFloats * 2];
    auto pack_w = [&](int ntiles, int records_per_tile, float sw, auto weight_at) {
      for (int t = 0; t < ntiles; ++t)
        for (int i = 0; i < records_per_tile; ++i) {
          for (int lane = 0; lane < 64; ++lane)
            for (int e = 0; e < 8; ++e) {
              // (feature half outer for the 32-record tiles of layers 1 and 3, K32-block outer for layer 2's 16)
              const int fh = records_per_tile == 32 ? i / 16 : (i & 1), j = records_per_tile == 32 ? i % 16 : (i >> 1);
              const int row = 32 * t + 16 * fh + (lane & 15);
              const int feat = 32 * j + 16 * (e >> 2) + 4 * (lane >> 4) + (e & 3);
              const float w = weight_at(row, feat) * sw;
              const float hi = f16_round(w);
              wp[(0 * 64 + lane) * 8 + e] = f16_bits(hi);
              wp[(1 * 64 + lane) * 8 + e] = f16_bits(w - hi);
            }
          wp += 1024;
        }
    };
    pack_w(kTilesL1, 32, sw1, [&](int row, int feat) { return row < n1 ? W1[(size_t)row * kHidden + feat] : 0.0f; });
    pack_w(kTilesHidden, 16, sw2, [&](int row, int feat) { return feat < n1 ? W2[(size_t)row * kHidden + feat] : 0.0f; });
    pack_w(kTilesHidden, 32, sw3, [&](int row, int feat) { return W3[(size_t)row * kHidden + feat]; });
    hp.sw[h][0] = sw1; hp.sw[h][1] = sw2; hp.sw[h][2] = sw3;
  }
  const float sx_default[kHeads][3] = {{kActScale, kActScale, kActScale}, {kActScale, kActScale, kActScale}};
  scale_constants_f16(spec, hp.kp, hp.cst.data(), hp.sw, sx_default, hp.cst16.data(), hp.s2);
  return true;
}

// K-steps the point features occupy in layers 0 and 2: 2 for (affine) xyz, ceil(pf / 2) for the NeRF encoding
inline int point_ksteps(const asdf_decoder_spec_t& spec) {
  return spec.feature_mode == ASDF_FEATURES_NERF ? (spec.point_feats[0] + 1) / 2 : 2;
}

inline bool pack_decoder(const asdf_decoder_spec_t& spec, const asdf_head_params_t* heads, HostPack& hp) {
  hp.kp = point_ksteps(spec);
  const CstOffsets co = cst_offsets(hp.kp);
  try {
    hp.stream.assign((size_t)kStagesAll * kStageFloats, 0.f);
    hp.wlat.assign((size_t)kHeads * 2 * kHidden * kLatent, 0.f);
    hp.wpt.assign((size_t)kHeads * 2 * kHidden * ASDF_MAX_POINT_FEATS, 0.f);
    hp.b02.assign((size_t)kHeads * 2 * kHidden, 0.f);
    hp.cst.assign((size_t)kHeads * co.floats, 0.f);
    hp.emb.assign((size_t)kHeads * ASDF_MAX_POINT_FEATS * 4, 0.f);
  } catch (...) {
    return false;
  }
  for (int h = 0; h < spec.num_heads; ++h) {
    const int pf = spec.point_feats[h];
    const int in = kLatent + pf;
    const int n1 = kHidden - in;       // layer-1 width: dims[1] - dims[0] (networks/model.py:244-245)
    const float* W0 = heads[h].w[0]; const float* W1 = heads[h].w[1]; const float* W2 = heads[h].w[2];
    const float* W3 = heads[h].w[3]; const float* W4 = heads[h].w[4];
    // latent / point columns of layer 0 ([latent | pts]) and layer 2 ([x1 (n1) | latent | pts])
    for (int o = 0; o < kHidden; ++o) {
      float* l0 = &hp.wlat[((size_t)(h * 2 + 0) * kHidden + o) * kLatent];
      float* l2 = &hp.wlat[((size_t)(h * 2 + 1) * kHidden + o) * kLatent];
      for (int k = 0; k < kLatent; ++k) { l0[k] = W0[(size_t)o * in + k]; l2[k] = W2[(size_t)o * kHidden + n1 + k]; }
      float* p0 = &hp.wpt[((size_t)(h * 2 + 0) * kHidden + o) * ASDF_MAX_POINT_FEATS];
      float* p2 = &hp.wpt[((size_t)(h * 2 + 1) * kHidden + o) * ASDF_MAX_POINT_FEATS];
      for (int f = 0; f < pf; ++f) { p0[f] = W0[(size_t)o * in + kLatent + f]; p2[f] = W2[(size_t)o * kHidden + n1 + kLatent + f]; }
      hp.b02[(h * 2 + 0) * kHidden + o] = heads[h].b[0][o];
      hp.b02[(h * 2 + 1) * kHidden + o] = heads[h].b[2][o];
    }
    // weight stream: L1 (8 tiles x 4 stages), L2 (16 x 2), L3 (16 x 4)
    float* sp = &hp.stream[(size_t)h * kStagesHead * kStageFloats];
    auto pack = [&](int ntiles, int stages_per_tile, auto weight_at) {
      for (int t = 0; t < ntiles; ++t)
        for (int q = 0; q < stages_per_tile; ++q) {
          for (int g = 0; g < 16; ++g)
            for (int lane = 0; lane < 64; ++lane)
              for (int j = 0; j < 4; ++j) {
                const int s = q * kStageKSteps + g * 4 + j;
                const int row = 32 * t + (lane & 31);
                const int feat = kstep_feature(s, lane >> 5);
                sp[(g * 64 + lane) * 4 + j] = weight_at(row, feat);
              }
          sp += kStageFloats;
        }
    };
    pack(kTilesL1, 4, [&](int row, int feat) { return row < n1 ? W1[(size_t)row * kHidden + feat] : 0.0f; });
    pack(kTilesHidden, 2, [&](int row, int feat) { return feat < n1 ? W2[(size_t)row * kHidden + feat] : 0.0f; });
    pack(kTilesHidden, 4, [&](int row, int feat) { return W3[(size_t)row * kHidden + feat]; });
    // static constants in D-layout order
    float* c = &hp.cst[(size_t)h * co.floats];
    for (int t = 0; t < kTilesHidden; ++t)
      for (int hh = 0; hh < 2; ++hh)
        for (int r = 0; r < 16; ++r) {
          const int row = 32 * t + tile_row(r, hh);
          if (t < kTilesL1) c[co.b1 + (t * 2 + hh) * 16 + r] = row < n1 ? heads[h].b[1][row] : 0.0f;
          c[co.b3 + (t * 2 + hh) * 16 + r] = heads[h].b[3][row];
          c[co.w4 + (t * 2 + hh) * 16 + r] = W4[row];
          c[co.w4b + (t * 2 + hh) * 16 + r] = spec.outputs[h] > 1 ? W4[kHidden + row] : 0.0f;
        }
    c[co.b4] = heads[h].b[4][0];
    c[co.b4 + 1] = spec.outputs[h] > 1 ? heads[h].b[4][1] : 0.0f;
    // default embedding: identity on xyz (PointFeatSize 3)
    for (int f = 0; f < 3 && f < pf; ++f) hp.emb[((size_t)h * ASDF_MAX_POINT_FEATS + f) * 4 + f] = 1.0f;
    // NeRF encoding: the point columns are static weights, not a per-sample fold -> A fragments packed here
    if (spec.feature_mode == ASDF_FEATURES_NERF)
      for (int layer = 0; layer < 2; ++layer)
        for (int t = 0; t < kTilesHidden; ++t)
          for (int s = 0; s < hp.kp; ++s)
            for (int lane = 0; lane < 64; ++lane) {
              const int row = 32 * t + (lane & 31), f = 2 * s + (lane >> 5);
              const float* wp = &hp.wpt[((size_t)(h * 2 + layer) * kHidden + row) * ASDF_MAX_POINT_FEATS];
              c[(layer ? co.a2 : co.a0) + (t * hp.kp + s) * 64 + lane] = f < pf ? wp[f] : 0.0f;
            }
  }
  return true;
}

}  // namespace asdf
