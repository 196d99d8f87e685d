// K1: the fused SDF decoder kernel (operand maps and structure: sdf_mlp_common.h / sdf_layout.h).
// Schedule (second generation - profiles/r01_k1_tuning_log.txt has the measurements against the first):
//   * the per-stage barrier sits in the MIDDLE of a stage (in the shadow of an in-flight MFMA) and certifies
//     the NEXT stage, so the A-fragment prefetch runs across stage boundaries and never drains;
//   * the ReLU / dot-product epilogue of output tile t is deferred into the MFMA stream of tile t+1;
//   * ReLU is an integer max on the float bits (one VALU op, no canonicalisation).
#pragma once
#include "sdf_mlp_common.h"

namespace asdf {

__device__ __forceinline__ f32x16 relu16i(f32x16 v) {
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = __int_as_float(max(__float_as_int(v[r]), 0));
  return v;
}

struct NoEpilogue {
  __device__ __forceinline__ void operator()() const {}
};

// One 64-K-step stage of the weight stream.  On entry (a0, a1) hold the A fragments of groups 0 and 1 of THIS
// stage; on exit they hold those of the next stage in stream order.  `epi` (the previous tile's epilogue) is
// issued behind the first eight MFMAs.
// ABL is an ablation mask for tools/k1_ablate.hip (0 in the product; timing only, results invalid):
// 1 = no DMA / vmcnt wait / barrier in the loop (ring filled once), 4 = tile epilogues skipped (accumulators kept
// live), 8 = no layer 0, 16 = no s_barrier (DMA and waits kept).
template <int KT, int Q, int SLOT, int ABL, class Epi>
__device__ __forceinline__ void stage(f32x16& acc, const f32x16 (&hin)[KT], const float* ring, const float* next_src,
                                       unsigned lds_ring_base, int lane, int wave, f32x4& a0, f32x4& a1, Epi&& epi) {
  constexpr int nslot = (SLOT + kRing - 1) % kRing;   // slot of stage (this - 1), refilled with stage (this + 3)
  const float* src = next_src + wave * 1024 + lane * 4;
  const unsigned dst = lds_ring_base + (nslot * kStageFloats + wave * 1024) * 4;
  const f32x4* cur = reinterpret_cast<const f32x4*>(ring + SLOT * kStageFloats) + lane;
  const f32x4* nxt = reinterpret_cast<const f32x4*>(ring + ((SLOT + 1) % kRing) * kStageFloats) + lane;
  f32x4 abuf[18];
  abuf[0] = a0;
  abuf[1] = a1;
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    if (g == 8 && !(ABL & 1)) {
      // my pieces of stage (this + 1) were issued 2.5 stages ago; only those of (this + 2) may stay in flight
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      if (!(ABL & 16)) __builtin_amdgcn_s_barrier();
    }
    abuf[g + 2] = g + 2 < 16 ? cur[(g + 2) * 64] : nxt[(g + 2 - 16) * 64];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int s = Q * 64 + g * 4 + j;
      // one dependent chain per tile: tools/mfma_issue_bench shows a chain fed with varying operands already
      // issues at the pipe rate (65 cycles), and splitting it into two chains measured slower end to end
      acc = ASDF_MFMA(abuf[g][j], hin[s >> 4][s & 15], acc);
      if (g == 8 && !(ABL & 1)) {
        // one DMA piece per MFMA shadow (an LDS-DMA issue costs about one 64-cycle MFMA slot); pinned so the
        // scheduler cannot cluster the four pieces behind a single MFMA
        if (j == 0) lds_dma16_off<0>(src, dst);
        else if (j == 1) lds_dma16_off<1024>(src, dst);
        else if (j == 2) lds_dma16_off<2048>(src, dst);
        else lds_dma16_off<3072>(src, dst);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (g == 1) epi();
  }
  a0 = abuf[16];
  a1 = abuf[17];
}

// k-th NeRF positional-encoding feature of a point (get_nerf_embedder, utils/utils.py:433-463,521-533):
// [x0 x1 x2 | sin(2^0 x) (3) cos(2^0 x) (3) | sin(2^1 x) (3) cos(2^1 x) (3) | ...]; fp32 like torch (x * freq, then sin).
__device__ __forceinline__ float nerf_feature(int f, float x0, float x1, float x2) {
  if (f < 3) return f == 0 ? x0 : (f == 1 ? x1 : x2);
  const int j = f - 3, r = j % 6, d = r % 3;
  const float xd = d == 0 ? x0 : (d == 1 ? x1 : x2);
  const float arg = __fmul_rn(xd, (float)(1 << (j / 6)));
  return r < 3 ? sinf(arg) : cosf(arg);
}

// p.num_mlps = 2: SeparateDecoder (two MLPs, one output each); 1: CombinedDecoder (one MLP, two outputs).  The head loop
// keeps a runtime trip count on purpose (with a compile-time single trip the compiler hoists ~1400 loop-invariant
// values out of the tile loop and spills them); TWO_OUT selects, at compile time, whether the second last-layer row
// is accumulated - it costs the single-output path 3.5 % when merely left in with zero weights.
// KP = K-steps taken by the point features in layers 0 and 2: 2 = (affine) xyz, 5 / 8 = NeRF encoding of 9 / 15 features.
// CLS adds the part classifier of the label pass (classifier_head = Linear(512, num_class) on the last hidden activation
// of MLP 0, networks/model.py:134-137,161-162 / :257-259,306-307): kMaxClasses more rows of the fused last layer.
template <int ABL, int KP, bool TWO_OUT, bool CLS = false>
__device__ __forceinline__ void sdf_mlp_body(const DecodeParams& p) {
  using CL = CstLayout<KP>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* ring = smem;
  float* cst = smem + kLdsRingFloats;
  float* clsw = cst + CL::kFloats;         // CLS only: [kMaxClasses][512 in D-layout order] + [kMaxClasses] biases

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;

  // the lattice: by value, or - a fine pass enqueued behind its coarse pass - from the device words asdf_zoom_cube wrote
  float lat_vs = p.vs, lat_o0 = p.o0, lat_o1 = p.o1, lat_o2 = p.o2;
  if (p.lattice) { lat_o0 = p.lattice[0]; lat_o1 = p.lattice[1]; lat_o2 = p.lattice[2]; lat_vs = p.lattice[3]; }
  // kGridSubset: the point count lives on the device (the list was compacted by a kernel just in front of this launch)
  long long npts = p.P;
  if (p.mode == kGridSubset) {
    const long long c = *p.count_dev;
    npts = c < npts ? c : npts;
    // the short-list form's (sdf_mlp_short_kernel.h) - unless its cluster form has reported a member that never arrived: then this
    // launch, which is enqueued behind it anyway, evaluates the list (bit-identical results, no extra launch)
    if (KP == 2 && !CLS && p.short_max > 0 && npts <= (long long)p.short_max && !(p.short_fault && *p.short_fault != 0)) return;
  }
  const long long ntiles = (npts + kWgPts - 1) / kWgPts;
  if ((long long)blockIdx.x >= ntiles) return;

  const unsigned lds_ring_base = (unsigned)(size_t)(__attribute__((address_space(3))) float*)ring;

  // Head-outer, tile-inner: one MLP's 2 MiB weight stream is live at a time, so it stays resident in the 4 MiB
  // per-XCD L2 (both heads interleaved per tile thrash it: ~10 GB of refills per N=256 launch).
#pragma unroll 1
  for (int slot = 0; slot < p.num_mlps; ++slot) {
    const int head = p.first_mlp + slot;                 // which MLP of the decoder this iteration evaluates
    const float* hc = cst;
    // negative-voxel bounding box of this MLP's output(s): per-lane accumulators, flushed once per head
    int bmin0 = 0x7fffffff, bmin1 = 0x7fffffff, bmin2 = 0x7fffffff, bmax0 = -1, bmax1 = -1, bmax2 = -1, bcnt = 0;
    int omin0 = 0x7fffffff, omin1 = 0x7fffffff, omin2 = 0x7fffffff, omax0 = -1, omax1 = -1, omax2 = -1, ocnt = 0;   // TWO_OUT only
    // (re)start: every wave is done with the previous head's constants and ring slots, and its own wrapped
    // prefetches have landed -> load this head's constants, restart the DMA ring on this head's stream
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    {
      const f32x4* src4 = reinterpret_cast<const f32x4*>(p.cst + (size_t)head * CL::kFloats);
      for (int i = tid; i < CL::kFloats / 4; i += 256) reinterpret_cast<f32x4*>(cst)[i] = src4[i];
      if (CLS) {
        const f32x4* c4 = reinterpret_cast<const f32x4*>(p.cls);
        for (int i = tid; i < kClsFloats / 4; i += 256) reinterpret_cast<f32x4*>(clsw)[i] = c4[i];
      }
    }
    const float* sbase0 = p.stream + (size_t)head * kStagesHead * kStageFloats;
#pragma unroll
    for (int s = 0; s < ((ABL & 1) ? kRing : kRing - 1); ++s) {
      const float* src = sbase0 + (size_t)s * kStageFloats + wave * 1024 + lane * 4;
      const unsigned dst = lds_ring_base + (s * kStageFloats + wave * 1024) * 4;
#pragma unroll
      for (int c = 0; c < 4; ++c) lds_dma16(src + c * 256, dst + c * 1024);
    }
    if (ABL & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // my pieces of stage 0 (and my constants loads)
    __syncthreads();                                      // everybody's pieces of stage 0, and the constants
    f32x4 a0 = (reinterpret_cast<const f32x4*>(ring) + lane)[0];
    f32x4 a1 = (reinterpret_cast<const f32x4*>(ring) + lane)[64];

#pragma unroll 1
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
      const long long pi = tile * kWgPts + wave * kWavePts + (lane & 31);
      const bool valid = pi < npts;
      float x0 = 0.f, x1 = 0.f, x2 = 0.f;
      long long po = pi;                        // where this point's outputs go
      if (p.mode == kPointList) {
        if (valid) { x0 = p.xyz[pi * 3 + 0]; x1 = p.xyz[pi * 3 + 1]; x2 = p.xyz[pi * 3 + 2]; }
      } else if (p.mode == kGridSubset) {
        // listed lattice points: the same coordinate function as the sweep, outputs scattered back into the volumes
        po = valid ? (long long)p.idx[pi] : 0;
        grid_point(po, p.N, p.grid_mode, lat_vs, lat_o0, lat_o1, lat_o2, x0, x1, x2);
      } else {
        grid_point(valid ? pi : 0, p.N, p.mode, lat_vs, lat_o0, lat_o1, lat_o2, x0, x1, x2);
      }
      // B operands of the point-feature K-steps: lane half h supplies feature 2 s + h of K-step s
      float bp[KP];
      if (KP == 2) {
        bp[0] = half ? x1 : x0;
        bp[1] = half ? 0.0f : x2;
      } else {
#pragma unroll
        for (int s = 0; s < KP; ++s) bp[s] = 2 * s + half < p.pf ? nerf_feature(2 * s + half, x0, x1, x2) : 0.0f;
      }
      // source of stage (s + 3) of this head's stream, wrapping to its start for the next tile; the base is made
      // opaque per tile so that the 128 x 4 per-lane source addresses are not hoisted out of the tile loop (spills)
      const float* sbase = sbase0;
      asm volatile("" : "+s"(sbase));
      auto src_of = [&](int s) -> const float* {   // s = stage index within the head + 3
        return sbase + (size_t)(s < kStagesHead ? s : s - kStagesHead) * kStageFloats;
      };

      // ---- layer 0: K = 2 KP point features (xyz + zero pad, or the NeRF encoding), per-sample A fragments from LDS
      f32x16 h0[kTilesHidden];
#pragma unroll
      for (int t = 0; t < kTilesHidden; ++t) {
        f32x16 acc = load_bias16(hc + CL::kC0 + (t * 2 + half) * 16);
        if (!(ABL & 8)) {
#pragma unroll
          for (int s = 0; s < KP; ++s) acc = ASDF_MFMA(hc[CL::kA0 + (t * KP + s) * 64 + lane], bp[s], acc);
        }
        h0[t] = (ABL & 8) ? acc : relu16i(acc);
      }

#define ASDF_STAGE(KT, Q, SLOT, ACC, HIN, SIDX, EPI) \
  stage<KT, Q, SLOT, ABL>(ACC, HIN, ring, src_of((SIDX) + 3), lds_ring_base, lane, wave, a0, a1, EPI)

      // ---- layer 1: 512 -> 256 (rows >= n1 are zero padding); epilogue of tile t-1 rides in tile t
      f32x16 h1[kTilesL1];
      if (ABL & 4) for (int t = 0; t < kTilesL1; ++t) h1[t] = h0[t];
      f32x16 acc1[2];
#pragma unroll
      for (int t = 0; t < kTilesL1; ++t) {
        f32x16& acc = acc1[t & 1];
        acc = load_bias16(hc + CL::kB1 + (t * 2 + half) * 16);
        auto epi = [&]() {
          if (t == 0) return;
          if (ABL & 4) { asm volatile("" :: "v"(acc1[(t - 1) & 1])); return; }
          h1[t - 1] = relu16i(acc1[(t - 1) & 1]);
        };
        ASDF_STAGE(16, 0, 0, acc, h0, t * 4 + 0, epi);
        ASDF_STAGE(16, 1, 1, acc, h0, t * 4 + 1, NoEpilogue());
        ASDF_STAGE(16, 2, 2, acc, h0, t * 4 + 2, NoEpilogue());
        ASDF_STAGE(16, 3, 3, acc, h0, t * 4 + 3, NoEpilogue());
      }

      // ---- layer 2: [h1 (256) | xyz (4)] -> 512
      f32x16 h2[kTilesHidden];
      if (ABL & 4) for (int t = 0; t < kTilesHidden; ++t) h2[t] = h0[t];
      f32x16 acc2[2];
#pragma unroll
      for (int t = 0; t < kTilesHidden; ++t) {
        f32x16& acc = acc2[t & 1];
        acc = load_bias16(hc + CL::kC2 + (t * 2 + half) * 16);
#pragma unroll
        for (int s = 0; s < KP; ++s) acc = ASDF_MFMA(hc[CL::kA2 + (t * KP + s) * 64 + lane], bp[s], acc);
        auto epi = [&]() {
          if (ABL & 4) { asm volatile("" :: "v"(acc2[(t + 1) & 1]), "v"(acc1[1])); return; }
          if (t > 0) h2[t - 1] = relu16i(acc2[(t - 1) & 1]);
          else h1[kTilesL1 - 1] = relu16i(acc1[(kTilesL1 - 1) & 1]);   // consumed by K-steps >= 112
        };
        constexpr int S0 = kStagesL1;
        if (t & 1) {
          ASDF_STAGE(8, 0, 2, acc, h1, S0 + t * 2 + 0, epi);
          ASDF_STAGE(8, 1, 3, acc, h1, S0 + t * 2 + 1, NoEpilogue());
        } else {
          ASDF_STAGE(8, 0, 0, acc, h1, S0 + t * 2 + 0, epi);
          ASDF_STAGE(8, 1, 1, acc, h1, S0 + t * 2 + 1, NoEpilogue());
        }
      }

      // ---- layer 3 (512 -> 512) fused with layer 4 (dot with w4) and tanh
      float part = 0.0f, partb = 0.0f;      // partb: second output row (CombinedDecoder; its weights are 0 otherwise)
      float pc[kMaxClasses];                // CLS: classifier logits (rows >= num_class have zero weights)
#pragma unroll
      for (int k = 0; k < kMaxClasses; ++k) pc[k] = 0.0f;
      f32x16 acc3[2];
      auto dot_w4 = [&](const f32x16 a, int t) {
        if (CLS) {
          const f32x4* wc = reinterpret_cast<const f32x4*>(clsw + (t * 2 + half) * 16);
#pragma unroll
          for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int k = 0; k < kMaxClasses; ++k) {
              const f32x4 w = wc[k * (kHidden / 4) + c];
#pragma unroll
              for (int r = 0; r < 4; ++r)
                pc[k] = fmaf(__int_as_float(max(__float_as_int(a[c * 4 + r]), 0)), w[r], pc[k]);
            }
          }
          // as for partb below: pin the chains to this tile, or they are deferred to the head end through scratch
          asm volatile("" : "+v"(pc[0]), "+v"(pc[1]), "+v"(pc[2]), "+v"(pc[3]), "+v"(pc[4]), "+v"(pc[5]), "+v"(pc[6]), "+v"(pc[7]));
        }
        // 4 registers at a time: keeps the two weight rows out of long-lived registers
        const f32x4* w4 = reinterpret_cast<const f32x4*>(hc + CL::kW4 + (t * 2 + half) * 16);
        const f32x4* w4b = reinterpret_cast<const f32x4*>(hc + CL::kW4b + (t * 2 + half) * 16);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f32x4 w = w4[c];
          f32x4 wb = w;
          if (TWO_OUT) wb = w4b[c];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = __int_as_float(max(__float_as_int(a[c * 4 + r]), 0));
            part = fmaf(v, w[r], part);
            if (TWO_OUT) partb = fmaf(v, wb[r], partb);
          }
        }
        // keep both chains where they are written: left alone, the scheduler defers the second chain to the head end and
        // spills every tile's activations and weights to scratch to do so
        if (TWO_OUT) asm volatile("" : "+v"(part), "+v"(partb));
      };
#pragma unroll
      for (int t = 0; t < kTilesHidden; ++t) {
        f32x16& acc = acc3[t & 1];
        acc = load_bias16(hc + CL::kB3 + (t * 2 + half) * 16);
        auto epi = [&]() {
          if (ABL & 4) { asm volatile("" :: "v"(acc3[(t + 1) & 1]), "v"(acc2[1])); return; }
          if (t > 0) dot_w4(acc3[(t - 1) & 1], t - 1);
          else h2[kTilesHidden - 1] = relu16i(acc2[(kTilesHidden - 1) & 1]);   // consumed by K-steps >= 240
        };
        constexpr int S0 = kStagesL1 + kStagesL2;
        ASDF_STAGE(16, 0, 0, acc, h2, S0 + t * 4 + 0, epi);
        ASDF_STAGE(16, 1, 1, acc, h2, S0 + t * 4 + 1, NoEpilogue());
        ASDF_STAGE(16, 2, 2, acc, h2, S0 + t * 4 + 2, NoEpilogue());
        ASDF_STAGE(16, 3, 3, acc, h2, S0 + t * 4 + 3, NoEpilogue());
      }
      dot_w4(acc3[(kTilesHidden - 1) & 1], kTilesHidden - 1);
#undef ASDF_STAGE
      part += __shfl_xor(part, 32);
      const float sdf = tanhf(part + hc[CL::kB4]);
      const bool combined = TWO_OUT;
      float sdfb = 1.0f;
      if (combined) {
        partb += __shfl_xor(partb, 32);
        sdfb = tanhf(partb + hc[CL::kB4 + 1]);
      }
      // output 0 of MLP 0 is the hand SDF; the object SDF is output 0 of MLP 1 or output 1 of a combined MLP
      const bool is_hand = head == 0;
      if (valid && half == 0) {
        float* out = is_hand ? p.sdf0 : p.sdf1;
        if (p.mode == kGridSubset && p.status && !p.bbox) {
          // re-evaluation without a box to patch (the narrow-band fine sweep): only the measured error of what it replaces
          if (out) atomicMax(p.status + 3, __float_as_int(fabsf(sdf - out[po])));
          if (combined && p.sdf1) atomicMax(p.status + 3, __float_as_int(fabsf(sdfb - p.sdf1[po])));
        }
        if (p.mode == kGridSubset && p.bbox) {
          // refinement of an existing volume: patch the negative-voxel box for every sign change instead of recounting
          auto patch = [&](float* vol, float now, int* rec) {
            const float before = vol[po];
            const bool was = before < p.neg_thr, is = now < 0.0f;
            // the largest change this pass made to a value (float bits): the measured error of the arithmetic it corrects
            if (p.status) atomicMax(p.status + 3, __float_as_int(fabsf(now - before)));
            if (was == is) return;
            if (is) {
              int i0, i1, i2;
              lattice_ijk(po, p.N, i0, i1, i2);
              atomicMin(rec + 0, i0); atomicMin(rec + 1, i1); atomicMin(rec + 2, i2);
              atomicMax(rec + 3, i0); atomicMax(rec + 4, i1); atomicMax(rec + 5, i2);
              atomicAdd(rec + 6, 1);
            } else {
              atomicExch(p.fixup_flag, 1);        // the box may have to shrink: the caller recounts
            }
          };
          if (out) patch(out, sdf, p.bbox + (is_hand ? 0 : 8));
          if (combined && p.sdf1) patch(p.sdf1, sdfb, p.bbox + 8);
        }
        if (out) out[po] = sdf;
        if (combined && p.sdf1) p.sdf1[po] = sdfb;
      }
      if (CLS) {
        int best = 0;
        float best_v = 0.0f;
#pragma unroll
        for (int k = 0; k < kMaxClasses; ++k) {
          pc[k] += __shfl_xor(pc[k], 32);
          pc[k] += clsw[kMaxClasses * kHidden + k];
          if (k == 0) best_v = pc[0];
          else if (k < p.num_class && pc[k] > best_v) { best_v = pc[k]; best = k; }   // first maximum wins (torch.argmax)
        }
        if (valid && half == 0 && is_hand) {
          if (p.logits) {
#pragma unroll
            for (int k = 0; k < kMaxClasses; ++k)
              if (k < p.num_class) p.logits[pi * p.num_class + k] = pc[k];
          }
          if (p.labels) p.labels[pi] = best;
        }
      }
      if (p.bbox && valid && half == 0 && p.mode != kPointList && p.mode != kGridSubset) {
        int i0, i1, i2;
        lattice_ijk(po, p.N, i0, i1, i2);
        if (sdf < 0.0f) {
          bmin0 = min(bmin0, i0); bmin1 = min(bmin1, i1); bmin2 = min(bmin2, i2);
          bmax0 = max(bmax0, i0); bmax1 = max(bmax1, i1); bmax2 = max(bmax2, i2); ++bcnt;
        }
        if (combined && sdfb < 0.0f) {
          omin0 = min(omin0, i0); omin1 = min(omin1, i1); omin2 = min(omin2, i2);
          omax0 = max(omax0, i0); omax1 = max(omax1, i1); omax2 = max(omax2, i2); ++ocnt;
        }
      }
    }   // tiles

    if (p.bbox) {
      // one set of atomics per wave: record 0 = hand (MLP 0 / first row), record 1 = object (MLP 1 / second row)
      auto flush = [&](int* rec, int a0_, int a1_, int a2_, int b0_, int b1_, int b2_, int n) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
          a0_ = min(a0_, __shfl_xor(a0_, m)); a1_ = min(a1_, __shfl_xor(a1_, m)); a2_ = min(a2_, __shfl_xor(a2_, m));
          b0_ = max(b0_, __shfl_xor(b0_, m)); b1_ = max(b1_, __shfl_xor(b1_, m)); b2_ = max(b2_, __shfl_xor(b2_, m));
          n += __shfl_xor(n, m);
        }
        if (lane == 0 && n) {
          atomicMin(rec + 0, a0_); atomicMin(rec + 1, a1_); atomicMin(rec + 2, a2_);
          atomicMax(rec + 3, b0_); atomicMax(rec + 4, b1_); atomicMax(rec + 5, b2_);
          atomicAdd(rec + 6, n);
        }
      };
      flush(p.bbox + (head == 0 ? 0 : 8), bmin0, bmin1, bmin2, bmax0, bmax1, bmax2, bcnt);
      if (TWO_OUT) flush(p.bbox + 8, omin0, omin1, omin2, omax0, omax1, omax2, ocnt);
    }
  }   // MLPs
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// The __global__ instantiations live in k1_kernels.hip / k1_cls_kernels.hip (one translation unit per family, so that
// they compile side by side); tools/k1_ablate.hip instantiates its own.

}  // namespace asdf
