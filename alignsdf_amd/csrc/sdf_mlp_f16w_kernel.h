// K1h, the W form (round 6): the split-half SeparateDecoder kernel of sdf_mlp_f16_kernel.h with its three hidden GEMMs on
// v_mfma_f32_16x16x32_f16 instead of v_mfma_f32_32x32x16_f16 - affine point features, two fp16 planes per operand, 32 points per
// wave, full lattice or voxel list (SUB).  The default arithmetic of every sweep of such a decoder since round 6.
//
// Why.  Under real operands the part is power-managed, and what the matrix pipe SUSTAINS depends on the instruction: on K1h's operand
// pattern and split-half data 2.04-2.08 PFLOP/s with 16x16x32 against 1.80-1.82 with 32x32x16 (tools/mfma_f16_energy_bench.hip,
// profiles/r06_mfma_shape_energy.txt) - the 16-wide instruction moves half the accumulator words per FLOP through the register file.
// In the kernel: 69.8 against 74.3 ms per N = 256 sweep, the two forms interleaved on one box (profiles/r06_k1h_shape_ab.txt).
//
// What changes.  Operand maps of the instruction: A lane l holds A[i = l & 15][k-slot (l >> 4, e)], B lane l holds
// B[k-slot (l >> 4, e)][j = l & 15], e = 0..7 - the SAME slot in both, so which k the hardware gives a slot never matters: the host
// packs the weight of the input feature that sits in that slot of B; D lane l, register r holds D[row = 4 (l >> 4) + r][col = l & 15].
// A wave's 32 points are two GROUPS of 16 (g = 0, 1); an output tile stays 32 features = two HALVES of 16 (fh = 0, 1); its
// accumulator stays ONE f32x16 = four 16x16 tiles, register 8 g + 4 fh + r <-> (feature 32 T + 16 fh + 4 (l >> 4) + r, point
// 16 g + (l & 15)).  split_part's access pattern (acc[e], acc[8 + e] -> element e of two operands) then needs NO change: element
// e = 4 fh + r of xh[2 T + g] IS the B operand of K32-block T of the next layer for group g, whose slot (q, e) holds feature
// 32 T + 16 (e >> 2) + 4 q + (e & 3).  A 2 KiB record of the stream ([plane][lane][8 halves]) is (tile, feature half fh, K32-block j):
// feature half OUTER in layers 1 and 3 (32 records per tile: fh = i / 16, j = i % 16), K32-block outer in layer 2 (16 records:
// j = i / 2, fh = i % 2 - the deferred epilogue of layer 1's last tile finishes the last K32-block's operands only in K-block 8 of
// layer 2's first tile).  Record i reads the B operands xh / xl[2 j + g] and feeds accumulator registers 8 g + 4 fh .. + 3 with six
// 16-clock MFMAs: the same 96 matrix-pipe clocks, the same LDS and L2 -> LDS traffic, the same 16-record stages, ring, barriers and
// epilogue slots as the 32-wide form.  The point-feature products of layers 0 / 2 run on v_mfma_f32_16x16x4_f32 (K = 4 = x, y, z,
// pad in ONE instruction per feature half and group).  Biases, w4 and the point fragments are other gathers of the SAME constants
// image (K0 unchanged); only the weight stream has a second image (pack.h: pack_decoder_f16, behind the first in one allocation).
//
// What stays.  Everything that is not the matrix instruction - plane split (split_part / split_tile), LDS-DMA ring and stage sizes
// (S16<2>), constants layout, range guard, activation peaks, box fold, status record - is sdf_mlp_f16_kernel.h's and is used from
// there.  This header holds its own copy of the stage and of the body because the 32-wide forms sit at the 512-register limit: with
// the W paths as a template parameter of their body - every one of them `if constexpr` - the CombinedDecoder form took 60 B and the
// subset form 20 B of scratch.  The copy is specialised: two planes, one 32-point group, K = 4 point features, one output, the
// shipped schedule (deferred epilogue parts in three pieces, one LDS-DMA piece per K-block, preloads half a tile ahead).
#pragma once
#include "sdf_mlp_f16_kernel.h"

namespace asdf {

static_assert(ASDF16_STAGE_KB == 16 && ASDF16_MIX_SPLIT && ASDF16_EPI_STEPS && ASDF16_DMA_PER_KB && ASDF16_PRELOAD && ASDF16_LOADS_FIRST &&
                  ASDF16_PIN_ACC,
              "the W form is written for the shipped schedule of the split-half kernel");

#define ASDF_MFMA16W(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)
#define ASDF_MFMA4W(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ f32x4 acc_get4(const f32x16& a, int o) {
  f32x4 v;
  v[0] = a[o]; v[1] = a[o + 1]; v[2] = a[o + 2]; v[3] = a[o + 3];
  return v;
}
__device__ __forceinline__ void acc_set4(f32x16& a, int o, const f32x4& v) { a[o] = v[0]; a[o + 1] = v[1]; a[o + 2] = v[2]; a[o + 3] = v[3]; }

// a tile's 32 bias-like words ([lane half][16 registers] of the 32x32 D layout) gathered for the W form's accumulator: feature
// 16 fh + 4 q + r of the tile sits in register 4 (2 fh + (q >> 1)) + r of lane half q & 1
__device__ __forceinline__ f32x16 load_bias16w(const float* tile_words, int lane) {
  const int q = lane >> 4;
  // FOUR reads, one per accumulator quad (the two groups start from the same words): a read lands in the quad it is for, where two
  // reads + copies cost 16 v_accvgpr_write per tile.  (The second address carries an opaque ZERO so that the reads are not merged -
  // an opaque pointer would lose its address space and come out as a flat load.)
  const float* w0 = tile_words + (q & 1) * 16 + 4 * (q >> 1);
  int dup = 0;
  asm volatile("" : "+v"(dup));
  const float* w1 = w0 + dup;
  f32x16 r;
  acc_set4(r, 0, *reinterpret_cast<const f32x4*>(w0)); acc_set4(r, 4, *reinterpret_cast<const f32x4*>(w0 + 8));
  acc_set4(r, 8, *reinterpret_cast<const f32x4*>(w1)); acc_set4(r, 12, *reinterpret_cast<const f32x4*>(w1 + 8));
  return r;
}

// The accumulator is four independent quads: pinned one by one (see pin_acc) - the 512-bit constraint of the 32-wide form makes the
// compiler gather them into one aligned tuple through 16 v_accvgpr_read / write pairs per tile.
__device__ __forceinline__ void pin_acc_w(f32x16& acc) {
#pragma unroll
  for (int o = 0; o < 16; o += 4) {
    f32x4 q;
    q[0] = acc[o]; q[1] = acc[o + 1]; q[2] = acc[o + 2]; q[3] = acc[o + 3];
    asm volatile("" : "+a"(q));
    acc[o] = q[0]; acc[o + 1] = q[1]; acc[o + 2] = q[2]; acc[o + 3] = q[3];
  }
}

// One LDS-DMA piece with M0 written WITHOUT saving and restoring it around the instruction (3 instead of 5 instructions per piece).
// The 32-wide form hides a piece in the 32 clocks of an MFMA; under 16-clock MFMAs the two extra scalar moves of every piece showed:
// 71.1 against 72.8 ms per sweep (profiles/r06_k1h_shape_ab.txt (5)).  M0 is a reserved register - the compiler does not track the
// clobber - so this is allowed only while nothing else in these kernels touches M0: tests/test_kernel_resources.py pins exactly that.
// (ASDF16_W_M0_CLOBBER=0: the saving form of sdf_mlp_common.h, for A/B runs.)
#ifndef ASDF16_W_M0_CLOBBER
#define ASDF16_W_M0_CLOBBER 1
#endif
template <int P>
__device__ __forceinline__ void dma_piece_w(const float* src, unsigned dst) {
#if ASDF16_W_M0_CLOBBER
  asm volatile(
      "s_mov_b32 m0, %1\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, off offset:%c2"
      :
      : "v"(src + (P >> 2) * 1024), "s"(dst + (P >> 2) * 4096), "i"((P & 3) * 1024)
      : "memory", "m0");
#else
  dma_piece<P>(src, dst);
#endif
}

// One stage = 16 records of one 32-feature output tile; KB = records per tile (32: layers 1 / 3, 16: layer 2), Q = the stage's
// index within the tile.  The structure of stage16 (sdf_mlp_f16_kernel.h): on entry (ah, al) hold the A fragments of the stage's
// first record(s), on exit those of the next stage in stream order; every record is one scheduling region
//     [A-fragment reads of record kb + PREFETCH, pre(kb)]  fence  [2 MFMAs, piece 0] [2 MFMAs, piece 1] [2 MFMAs, DMA piece, piece 2]
// The order of the six MFMAs is what decides this kernel's speed: a group's three MFMAs - (W_hi, x_lo), (W_lo, x_hi), (W_hi, x_hi),
// small terms first - sit back to back, and the groups take turns in SNAKE order from record to record, so that five of six MFMAs
// continue the accumulator of the MFMA right in front of them (the matrix pipe forwards it).  Product sum outer / group inner - no
// MFMA continuing its predecessor - was 2.5 % slower with FEWER clocks.  Both groups take the three products in the SAME order: a
// voxel's bits must not depend on the lane it sits in (the voxel lists of the subset form place it anywhere; with the order mirrored
// for the second group the full sweep and the list form differed in the last bit - tests/test_gpu_default_sweeps.py).
// ABL (timing only, tools/k1h_ablate.hip): 1 = no DMA / wait / barrier, 16 = no barrier, 32 = no DMA instructions.
template <int KB, int Q, int SLOT, int ABL, class Pre, class Epi>
__device__ __forceinline__ void stage16w(f32x16& acc, const h8 (&xh)[KB], const h8 (&xl)[KB], const float* ring, const float* next_src,
                                         unsigned lds_ring_base, int lane, int wave, h8 (&ah)[S16<2>::kPrefetch], h8 (&al)[S16<2>::kPrefetch],
                                         Pre&& pre, Epi&& epi) {
  using SG = S16<2>;
  constexpr int PF = SG::kPrefetch;
  constexpr int BKB = ASDF16_BARRIER_KB;
  constexpr int nslot = (SLOT + kRing - 1) % kRing;   // slot of stage (this - 1), refilled with stage (this + 3)
  static_assert(KB == 32 || KB == 16, "records per tile");
  static_assert(kS16Kb - BKB >= SG::kPieces, "one LDS-DMA piece per record behind the barrier");
  const float* src = next_src + wave * SG::kWaveFloats + lane * 4;
  const unsigned dst = lds_ring_base + (nslot * SG::kFloats + wave * SG::kWaveFloats) * 4;
  const h8* cur = reinterpret_cast<const h8*>(ring + SLOT * SG::kFloats) + lane;
  const h8* nxt = reinterpret_cast<const h8*>(ring + ((SLOT + 1) % kRing) * SG::kFloats) + lane;
  h8 bufh[kS16Kb + PF], bufl[kS16Kb + PF];
#pragma unroll
  for (int i = 0; i < PF; ++i) { bufh[i] = ah[i]; bufl[i] = al[i]; }
#pragma unroll
  for (int kb = 0; kb < kS16Kb; ++kb) {
    if (kb == BKB && !(ABL & 1)) {
      // my pieces of stage (this + 1) were issued 2.5 stages ago; only those of (this + 2) may stay in flight
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      if (!(ABL & 16)) __builtin_amdgcn_s_barrier();
    }
    bufh[kb + PF] = kb + PF < kS16Kb ? cur[((kb + PF) * 2 + 0) * 64] : nxt[((kb + PF - kS16Kb) * 2 + 0) * 64];
    bufl[kb + PF] = kb + PF < kS16Kb ? cur[((kb + PF) * 2 + 1) * 64] : nxt[((kb + PF - kS16Kb) * 2 + 1) * 64];
    pre(kb);
    __builtin_amdgcn_sched_barrier(0);
    // record ri of the tile: feature half fh, K32-block j (see the header); B operands xh / xl[2 j + g], accumulator quads 4 fh, 8 + 4 fh
    const int ri = Q * kS16Kb + kb;
    const int xb = KB == 32 ? 2 * (ri % 16) : (ri & ~1), o = (KB == 32 ? ri / 16 : (ri & 1)) * 4;
    const bool snake = ri & 1;
    const int xa = xb + (snake ? 1 : 0), xc = xb + (snake ? 0 : 1);      // the group that goes first / second in this record
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      f32x4 s0 = acc_get4(acc, o), s1 = acc_get4(acc, 8 + o);
      f32x4& sa = snake ? s1 : s0;
      f32x4& sb = snake ? s0 : s1;
      if (j == 0) { sa = ASDF_MFMA16W(bufh[kb], xl[xa], sa); sa = ASDF_MFMA16W(bufl[kb], xh[xa], sa); }
      if (j == 1) { sa = ASDF_MFMA16W(bufh[kb], xh[xa], sa); sb = ASDF_MFMA16W(bufh[kb], xl[xc], sb); }
      if (j == 2) { sb = ASDF_MFMA16W(bufl[kb], xh[xc], sb); sb = ASDF_MFMA16W(bufh[kb], xh[xc], sb); }
      acc_set4(acc, o, s0); acc_set4(acc, 8 + o, s1);
      // one DMA piece per record, behind its LAST MFMA pair - the gap that carries the least of a deferred epilogue part
      const int m = j == 2 ? kb - BKB : -1;
      if (!(ABL & 1) && !(ABL & 32) && m >= 0 && m < SG::kPieces) {
        if (m == 0) dma_piece_w<0>(src, dst);
        else if (m == 1) dma_piece_w<1>(src, dst);
        else if (m == 2) dma_piece_w<2>(src, dst);
        else if (m == 3) dma_piece_w<3>(src, dst);
        else if (m == 4) dma_piece_w<4>(src, dst);
        else if (m == 5) dma_piece_w<5>(src, dst);
        else if (m == 6) dma_piece_w<6>(src, dst);
        else dma_piece_w<7>(src, dst);
        __builtin_amdgcn_sched_barrier(0);
      }
      epi(kb, j);      // piece j of the record's deferred epilogue part
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int i = 0; i < PF; ++i) { ah[i] = bufh[kS16Kb + i]; al[i] = bufl[kS16Kb + i]; }
}

// p.stream / p.cst are the split-half images (pack_decoder_f16; the W form's stream image lies behind the 32-wide one).
// SUB: the points are the lattice voxels listed in p.idx (p.count_dev of them, a device word; p.P is the list's capacity),
// coordinates from the voxel index, outputs scattered in place, no box - see sdf_mlp_f16_body.
template <int ABL = 0, bool SUB = false>
__device__ __forceinline__ void sdf_mlp_f16w_body(const DecodeParams& p) {
  using CL = CstLayout<2>;
  using SG = S16<2>;
  static_assert(lds_bytes_f16(2, 2) <= 160 * 1024, "LDS budget");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* ring = smem;
  float* cst = smem + SG::kRingFloats;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;

  // the lattice: by value, or - a fine pass enqueued behind its coarse pass - from the device words asdf_zoom_cube wrote
  float lat_vs = p.vs, lat_o0 = p.o0, lat_o1 = p.o1, lat_o2 = p.o2;
  if (p.lattice) { lat_o0 = p.lattice[0]; lat_o1 = p.lattice[1]; lat_o2 = p.lattice[2]; lat_vs = p.lattice[3]; }
  long long npts = p.P;
  if (SUB) { const long long c = *p.count_dev; npts = c < npts ? c : npts; }
  const long long ntiles = (npts + kWgPts - 1) / kWgPts;
  if ((long long)blockIdx.x >= ntiles) return;

  const unsigned lds_ring_base = (unsigned)(size_t)(__attribute__((address_space(3))) float*)ring;
  // shader-clock stamps of workgroup 0 around a whole-lattice sweep (status words 12..13 begin, 14..15 end): bench.py's clock and
  // matrix-pipe utilisation
  if (!SUB && p.status && p.mode != kPointList && blockIdx.x == 0 && tid == 0) reinterpret_cast<long long*>(p.status + 12)[0] = clock64();

#pragma unroll 1
  for (int slot = 0; slot < p.num_mlps; ++slot) {
    const int head = p.first_mlp + slot;
    const float* hc = cst;
    // negative-voxel bounding box of this MLP's output + the range report, one record per wave in LDS: [0..2] min index, [3..5] max
    // index, [6] count, [7] lanes whose activations left the fp16 range (or whose output is not in [-1, 1]); [16..18] activation peaks
    int* wrec = reinterpret_cast<int*>(cst + CL::kFloats) + wave * kWrecInts;
    if (lane < kWrecInts) wrec[lane] = lane >= 16 ? 0 : ((lane & 7) < 3 ? 0x7fffffff : ((lane & 7) < 6 ? -1 : 0));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    {
      const f32x4* src4 = reinterpret_cast<const f32x4*>(p.cst + (size_t)head * CL::kFloats);
      for (int i = tid; i < CL::kFloats / 4; i += 256) reinterpret_cast<f32x4*>(cst)[i] = src4[i];
    }
    // (the W form's image lies behind the 32x32x16 one in the same allocation)
    const float* sbase0 = p.stream + (size_t)kStagesAll * kStageFloats + (size_t)head * kS16Head * SG::kFloats;
#pragma unroll
    for (int s = 0; s < ((ABL & 33) ? kRing : kRing - 1); ++s) {
      const float* src = sbase0 + (size_t)s * SG::kFloats + wave * SG::kWaveFloats + lane * 4;
      const unsigned dst = lds_ring_base + (s * SG::kFloats + wave * SG::kWaveFloats) * 4;
#pragma unroll
      for (int c = 0; c < SG::kPieces; ++c) lds_dma16(src + c * 256, dst + c * 1024);
    }
    if (ABL & 33) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // my pieces of stage 0 (and my constants loads): those of stages 1 and 2 may stay in flight
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __syncthreads();                                      // everybody's pieces of stage 0, and the constants
    h8 ah[SG::kPrefetch], al[SG::kPrefetch];
#pragma unroll
    for (int i = 0; i < SG::kPrefetch; ++i) {
      ah[i] = (reinterpret_cast<const h8*>(ring) + lane)[(i * 2 + 0) * 64];
      al[i] = (reinterpret_cast<const h8*>(ring) + lane)[(i * 2 + 1) * 64];
    }
    // accumulator -> next layer's planes: S_x of the produced activations / (S_w S_x) of the accumulator (powers of two)
    const float mul1 = hc[CL::kB4 + 2], mul2 = hc[CL::kB4 + 3], mul0 = hc[CL::kB4 + 4];
    // subset mode: list positions from here on are audit picks (see DecodeParams::audit)
    int audit_from = 0x7fffffff;
    if (SUB && p.audit) audit_from = p.audit_from ? *p.audit_from : 0;
    // a tile's bias row as the accumulator's initial value, gathered for the 16x16 tiles
    auto bias_at = [&](int off, int t) -> f32x16 { return load_bias16w(hc + off + t * 32, lane); };
    // the fp32 A-fragment word of the point features for feature half fh of tile t: lane l holds row 16 fh + (l & 15), k = l >> 4 (the
    // image holds [tile][K-step k >> 1][lane half k & 1][row]: the fold kernel's layout for v_mfma_f32_32x32x2_f32)
    auto pt_word = [&](int off, int t, int fh) -> float {
      return hc[off + (t * 2 + (lane >> 5)) * 64 + ((lane >> 4) & 1) * 32 + 16 * fh + (lane & 15)];
    };

#pragma unroll 1
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
      const long long pi = tile * kWgPts + wave * kWavePts + (lane & 31);
      const bool valid = pi < npts;
      const long long po = SUB ? (valid ? (long long)p.idx[pi] : 0) : pi;      // where the point lives in the lattice / the outputs
      float x0 = 0.f, x1 = 0.f, x2 = 0.f;
      if (SUB) {
        grid_point(po, p.N, p.grid_mode, lat_vs, lat_o0, lat_o1, lat_o2, x0, x1, x2);
      } else if (p.mode == kPointList) {
        if (valid) { x0 = p.xyz[pi * 3 + 0]; x1 = p.xyz[pi * 3 + 1]; x2 = p.xyz[pi * 3 + 2]; }
      } else {
        grid_point(valid ? pi : 0, p.N, p.mode, lat_vs, lat_o0, lat_o1, lat_o2, x0, x1, x2);
      }
      // largest plane value (x S_x) this lane hands to the fp16 conversion, per activation vector h0 / h1 / h2: >= 65504 is an
      // overflow (range report); the maxima themselves go to the decoder's status record (the host calibrates S_x from them)
      float amax = 0.0f, amax1 = 0.0f, amax2 = 0.0f;
      // the point operand of v_mfma_f32_16x16x4_f32 per group: lane l carries component l >> 4 of (x, y, z, 0) of point
      // 16 g + (l & 15); a lane computed the coordinates of ITS point l & 31, so the source lane is (l & 15) + 16 g
      float bq[2];
      {
        const int q = lane >> 4;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const int src = (lane & 15) + 16 * g;
          const float c0 = __shfl(x0, src), c1 = __shfl(x1, src), c2 = __shfl(x2, src);
          bq[g] = q == 0 ? c0 : q == 1 ? c1 : q == 2 ? c2 : 0.0f;
        }
      }
      // point-feature products of one tile into its accumulator: af[fh] = pt_word(.., t, fh)
      auto pt_mfma = [&](f32x16& a, const float* af) {
#pragma unroll
        for (int fh = 0; fh < 2; ++fh)
#pragma unroll
          for (int g = 0; g < 2; ++g) acc_set4(a, 8 * g + 4 * fh, ASDF_MFMA4W(af[fh], bq[g], acc_get4(a, 8 * g + 4 * fh)));
      };
      const float* sbase = sbase0;
      asm volatile("" : "+s"(sbase));
      auto src_of = [&](int s) -> const float* {   // s = stage index within the head + 3
        return sbase + (size_t)(s < kS16Head ? s : s - kS16Head) * SG::kFloats;
      };

      // LDS reads that feed a tile are issued half a tile (or one record) ahead of their first use - `pre` slots of stage16w - into
      // registers that are dead at that point: the OTHER accumulator of the double buffer takes the next tile's bias row, `pf2` its
      // point-feature fragments, `w4n` the last-layer weights of the next epilogue part.
      f32x16 acc1[2], acc2[2], acc3[2];
      float pf2[2];                   // A fragments (fp32 MFMA) of the next layer-2 tile
      float w4c[2], w4n[2];           // last-layer weights of the current / next part of the layer-3 epilogue
      constexpr int kPreKb = ASDF16_PRE_KB;
      static_assert(kPreKb >= kEpiShift + kEpiChunks, "preload K-block inside the epilogue slots");
      auto load_pf2 = [&](int t) { pf2[0] = pt_word(CL::kA2, t, 0); pf2[1] = pt_word(CL::kA2, t, 1); };
      auto load_w4 = [&](int t, int c) {        // accumulator registers 2 c, 2 c + 1 of tile t: group c >> 2, feature half (c >> 1) & 1,
        const int wq = lane >> 4;               // r = 2 (c & 1) - feature 16 fh + 4 q + r sits in register 4 (2 fh + (q >> 1)) + r of lane half q & 1
        const float* w4 = hc + CL::kW4 + (t * 2 + (wq & 1)) * 16 + 4 * (2 * ((c >> 1) & 1) + (wq >> 1)) + 2 * (c & 1);
        w4n[0] = w4[0]; w4n[1] = w4[1];
      };
      auto next_w4 = [&]() { w4c[0] = w4n[0]; w4c[1] = w4n[1]; };

      // ---- layer 0 (fp32 MFMA, K = 4 point features): planes of relu(.) * S_x
      h8 h0h[2 * kTilesHidden], h0l[2 * kTilesHidden];
      f32x16 acc0[2];
      float pf0[2][2];
      auto l0_load = [&](int t) {
        acc0[t & 1] = bias_at(CL::kC0, t);
        pf0[t & 1][0] = pt_word(CL::kA0, t, 0); pf0[t & 1][1] = pt_word(CL::kA0, t, 1);
      };
      // the split of a tile is ~100 VALU instructions against 128 cycles of fp32 MFMA: layer 0 is VALU-bound when it runs on its
      // own.  Only the tiles the first stage of layer 1 consumes are computed up front; the others ride in the epilogue slots of
      // that stage, under its fp16 MFMAs.
      constexpr int kL0Front = kS16Kb / 2;
      l0_load(0);
      acc1[0] = bias_at(CL::kB1, 0);
#pragma unroll
      for (int t = 0; t < kL0Front; ++t) {
        l0_load(t + 1);
        __builtin_amdgcn_sched_barrier(0);
        f32x16 acc = acc0[t & 1];
        pt_mfma(acc, pf0[t & 1]);
        split_tile<2, 1, true>(acc, acc, mul0, h0h[2 * t], h0l[2 * t], h0h[2 * t + 1], h0l[2 * t + 1], amax);
        __builtin_amdgcn_sched_barrier(0);
      }

#define ASDF_STAGE16W(KB, Q, SLOT, ACC, XH, XL, SIDX, PRE, EPI) \
  stage16w<KB, Q, SLOT, ABL>(ACC, XH, XL, ring, src_of((SIDX) + 3), lds_ring_base, lane, wave, ah, al, PRE, EPI)

      // (the second argument of an epilogue callback is the PIECE of the part - stage16w calls it behind each of the record's three
      // MFMA pairs - and `er` carries a part's two ReLUs from piece to piece)
      float er[2] = {0.0f, 0.0f};
      // ---- layer 1: 512 -> 256; epilogue of tile t-1 rides in tile t
      h8 h1h[2 * kTilesL1], h1l[2 * kTilesL1];
      if (ABL & 4) for (int t = 0; t < 2 * kTilesL1; ++t) { h1h[t] = h0h[t]; h1l[t] = h0l[t]; }
#pragma unroll
      for (int t = 0; t < kTilesL1; ++t) {
        f32x16& acc = acc1[t & 1];
        auto pre = [&](int kb) {       // first stage: the layer-0 tile of the NEXT record's epilogue slot
          const int c = kb - kEpiShift;
          if (t == 0 && c >= 0 && c + 1 < kEpiChunks) l0_load(kL0Front + c + 1);
        };
        auto epi = [&](int kb, int g) {
          const int c = kb - kEpiShift;
          if (c < 0 || c >= kEpiChunks) return;
          if (t == 0) {
            // a whole layer-0 tile per record: its fp32 MFMAs behind the first pair, its eight parts over the three gaps (3 + 3 + 2)
            const int T = kL0Front + c;
            if (g == 0) pt_mfma(acc0[T & 1], pf0[T & 1]);
#pragma unroll
            for (int e = 0; e < 8; ++e)
              if (e / 3 == g) split_part<2, 1, true>(acc0[T & 1], acc0[T & 1], mul0, h0h[2 * T], h0l[2 * T], h0h[2 * T + 1], h0l[2 * T + 1], amax, e);
            return;
          }
          if (ABL & 4) { asm volatile("" :: "v"(acc1[(t - 1) & 1])); return; }
          if (g == 0) pin_acc_w(acc1[(t - 1) & 1]);
          split_part<2, 1, true>(acc1[(t - 1) & 1], acc1[(t - 1) & 1], mul1, h1h[2 * (t - 1)], h1l[2 * (t - 1)], h1h[2 * (t - 1) + 1],
                                 h1l[2 * (t - 1) + 1], amax1, c, -1, g, er);
        };
        auto pre_last = [&](int c) {   // last stage: bias row (and point fragments) of the next tile
          if (c != kPreKb) return;
          if (t + 1 < kTilesL1) acc1[(t + 1) & 1] = bias_at(CL::kB1, t + 1);
          else { acc2[0] = bias_at(CL::kC2, 0); load_pf2(0); }
        };
        if (t & 1) {
          ASDF_STAGE16W(32, 0, 2, acc, h0h, h0l, t * 2 + 0, pre, epi);
          ASDF_STAGE16W(32, 1, 3, acc, h0h, h0l, t * 2 + 1, pre_last, NoOp16());
        } else {
          ASDF_STAGE16W(32, 0, 0, acc, h0h, h0l, t * 2 + 0, pre, epi);
          ASDF_STAGE16W(32, 1, 1, acc, h0h, h0l, t * 2 + 1, pre_last, NoOp16());
        }
      }

      // ---- layer 2: [h1 (256) | xyz (4, fp32 MFMA, pre-scaled A fragments)] -> 512
      h8 h2h[2 * kTilesHidden], h2l[2 * kTilesHidden];
      if (ABL & 4) for (int t = 0; t < 2 * kTilesHidden; ++t) { h2h[t] = h0h[t]; h2l[t] = h0l[t]; }
      // one tile of layer 2; SLOT is the ring slot of its stage (a tag type: the slot must be a compile-time constant)
      auto l2_tile = [&](int t, auto slot_tag) {
        constexpr int SLOT = decltype(slot_tag)::value;
        f32x16& acc = acc2[t & 1];
        pt_mfma(acc, pf2);
        auto epi = [&](int kb, int g) {
          const int c = kb - kEpiShift;
          if (c < 0 || c >= kEpiChunks) return;
          if (ABL & 4) { asm volatile("" :: "v"(acc2[(t + 1) & 1]), "v"(acc1[1])); return; }
          if (t > 0) {
            if (g == 0) pin_acc_w(acc2[(t - 1) & 1]);
            split_part<2, 1, true>(acc2[(t - 1) & 1], acc2[(t - 1) & 1], mul2, h2h[2 * (t - 1)], h2l[2 * (t - 1)], h2h[2 * (t - 1) + 1],
                                   h2l[2 * (t - 1) + 1], amax2, c, -1, g, er);
          } else {
            if (g == 0) pin_acc_w(acc1[(kTilesL1 - 1) & 1]);
            split_part<2, 1, true>(acc1[(kTilesL1 - 1) & 1], acc1[(kTilesL1 - 1) & 1], mul1, h1h[2 * kTilesL1 - 2], h1l[2 * kTilesL1 - 2],
                                   h1h[2 * kTilesL1 - 1], h1l[2 * kTilesL1 - 1], amax1, c, -1, g, er);      // the last K32-block's operands: consumed from record 14 on
          }
        };
        auto pre_last = [&](int c) {
          if (c != kPreKb) return;
          if (t + 1 < kTilesHidden) { acc2[(t + 1) & 1] = bias_at(CL::kC2, t + 1); load_pf2(t + 1); }
          else acc3[0] = bias_at(CL::kB3, 0);
        };
        constexpr int S0 = 256 / kS16Kb;         // stages of layer 1
        ASDF_STAGE16W(16, 0, SLOT, acc, h1h, h1l, S0 + t, pre_last, epi);      // one stage per tile
      };
#pragma unroll
      for (int tt = 0; tt < kTilesHidden / 4; ++tt) {
        l2_tile(4 * tt + 0, std::integral_constant<int, 0>());
        l2_tile(4 * tt + 1, std::integral_constant<int, 1>());
        l2_tile(4 * tt + 2, std::integral_constant<int, 2>());
        l2_tile(4 * tt + 3, std::integral_constant<int, 3>());
      }

      // ---- layer 3 (512 -> 512) fused with layer 4 (dot with w4 / (S_w3 S_x)) and tanh
      float part = 0.0f, partg = 0.0f;      // the two point groups' dot products (accumulator registers 0..7 / 8..15)
      // accumulator registers 2 c, 2 c + 1 of a finished tile into the last-layer dot products, weights from w4c (only: 0 / 1 = that
      // register of the pair)
      auto dot_w4_part = [&](const f32x16& a, int c, int only = -1) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          if (only >= 0 && only != r) continue;
          const float v = __int_as_float(max(__float_as_int(a[2 * c + r]), 0));
          if (c < 4) part = fmaf(v, w4c[r], part);
          else partg = fmaf(v, w4c[r], partg);
        }
      };
#pragma unroll
      for (int t = 0; t < kTilesHidden; ++t) {
        f32x16& acc = acc3[t & 1];
        auto pre = [&](int kb) {       // first stage: w4 of the next epilogue part
          const int c = kb - kEpiShift;
          if (t > 0 && c >= 0 && c + 1 < kEpiChunks) load_w4(t - 1, c + 1);
        };
        auto epi = [&](int kb, int g) {
          const int c = kb - kEpiShift;
          if (c < 0 || c >= kEpiChunks) return;
          if (ABL & 4) { asm volatile("" :: "v"(acc3[(t + 1) & 1]), "v"(acc2[1])); return; }
          if (t > 0) {      // piece 0 / 1: one register of the pair each; piece 2: the next pair's weights
            if (g == 0) pin_acc_w(acc3[(t - 1) & 1]);
            if (g < 2) dot_w4_part(acc3[(t - 1) & 1], c, g);
            else next_w4();
          } else {
            if (g == 0) pin_acc_w(acc2[(kTilesHidden - 1) & 1]);
            split_part<2, 1, true>(acc2[(kTilesHidden - 1) & 1], acc2[(kTilesHidden - 1) & 1], mul2, h2h[2 * kTilesHidden - 2],
                                   h2l[2 * kTilesHidden - 2], h2h[2 * kTilesHidden - 1], h2l[2 * kTilesHidden - 1], amax2, c, -1, g, er);  // the last K32-block's operands: record 15
          }
        };
        auto pre_last = [&](int c) {   // last stage: bias row of the next tile, w4 of the first part of this tile's epilogue
          if (c != kPreKb) return;
          if (t + 1 < kTilesHidden) acc3[(t + 1) & 1] = bias_at(CL::kB3, t + 1);
          load_w4(t, 0);
          next_w4();
        };
        constexpr int S0 = 512 / kS16Kb;         // stages of layers 1 and 2
        if (t & 1) {
          ASDF_STAGE16W(32, 0, 2, acc, h2h, h2l, S0 + t * 2 + 0, pre, epi);
          ASDF_STAGE16W(32, 1, 3, acc, h2h, h2l, S0 + t * 2 + 1, pre_last, NoOp16());
        } else {
          ASDF_STAGE16W(32, 0, 0, acc, h2h, h2l, S0 + t * 2 + 0, pre, epi);
          ASDF_STAGE16W(32, 1, 1, acc, h2h, h2l, S0 + t * 2 + 1, pre_last, NoOp16());
        }
      }
      // the last tile's epilogue has no MFMA stream to hide under
#pragma unroll
      for (int c = 0; c < kEpiChunks; ++c) {
        if (c + 1 < kEpiChunks) load_w4(kTilesHidden - 1, c + 1);
        dot_w4_part(acc3[(kTilesHidden - 1) & 1], c);
        next_w4();
      }
#undef ASDF_STAGE16W
      // the four lane groups hold the four quarters of every feature half: sum them, then every lane takes the sum of ITS point
      // l & 31 (group (l >> 4) & 1) - everything below is the 32-wide form's
      part += __shfl_xor(part, 16); partg += __shfl_xor(partg, 16);
      part += __shfl_xor(part, 32); partg += __shfl_xor(partg, 32);
      part = (lane & 16) ? partg : part;
      const float pre = part + hc[CL::kB4];
      const float sdf = tanhf(pre);
      const bool is_hand = head == 0;
      if (SUB) {
        // subset mode replaces values: the largest change is the measured error of the arithmetic it corrects.  Marked voxels report
        // to status[3]; audit picks (voxels the one-plane sweep decided by sign alone) to the audit record, together with the number
        // of picks whose sign the exact value contradicts.  Reduced over the wave on the float BITS (a NaN is a huge pattern and must
        // survive the reduction), then one set of atomics per wave.
        int dm = 0, da = 0, flips = 0, na = 0;
        float sq = 0.0f;           // sum of squared audit errors: sigma of the one-plane error, next to its maximum
        if (valid && half == 0) {
          const bool aud = p.audit && pi >= (long long)audit_from;
          float* out = is_hand ? p.sdf0 : p.sdf1;
          if (out) {
            const float before = out[po];
            const int d = __float_as_int(fabsf(sdf - before));
            if (aud) { da = max(da, d); flips += ((before < 0.0f) != (sdf < 0.0f)) ? 1 : 0; ++na; sq += (sdf - before) * (sdf - before); }
            else dm = max(dm, d);
            // (a list of audit picks alone - the box sweep's - leaves the scratch volume as the one-plane kernel wrote it: that audit
            // runs BESIDE the collection of the sweep's candidates, which reads the same volume)
            if (!aud || p.audit_from) out[po] = sdf;
          }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
          dm = max(dm, __shfl_xor(dm, m)); da = max(da, __shfl_xor(da, m));
          flips += __shfl_xor(flips, m); na += __shfl_xor(na, m);
          if (p.audit) sq += __shfl_xor(sq, m);
        }
        if (lane == 0) {
          if (p.status && dm) atomicMax(p.status + 3, dm);
          if (p.audit && na) {
            atomicMax(p.audit + 0, da); if (flips) atomicAdd(p.audit + 1, flips); atomicAdd(p.audit + 2, na);
            atomicAdd(reinterpret_cast<float*>(p.audit + 3), sq);
          }
        }
      } else if (valid && half == 0) {
        float* out = is_hand ? p.sdf0 : p.sdf1;
        if (out) out[po] = sdf;
      }
      // every lane reports its own activations (the lane groups of a wave hold different features of the same points): a lane is
      // out of range when a value handed to the fp16 conversion reached 65504 or an output left [-1, 1] (NaN / infinity downstream
      // of an overflow).  The count goes to the decoder's status word - always - and, for grid sweeps with a bbox, to word 7 / 15 of
      // that record as well.
      amax *= mul0; amax1 *= mul1; amax2 *= mul2;      // (the maxima were taken in front of the rescale)
      const bool act_over = !(fmaxf(amax, fmaxf(amax1, amax2)) < 65504.0f);
      const int bad = (valid && (act_over || !(fabsf(sdf) <= 1.0f))) ? 1 : 0;
      if (p.status) {
        // per-wave peaks of the three activation vectors: non-negative floats order like their bit patterns, a NaN is a huge pattern
        // and reads as overflow.  One DPP reduction per value and a plain read-modify-write by one lane (the record is the wave's own).
        const int m0 = wave_max_i32(valid ? __float_as_int(amax) : 0), m1 = wave_max_i32(valid ? __float_as_int(amax1) : 0),
                  m2 = wave_max_i32(valid ? __float_as_int(amax2) : 0);
        if (lane == 0) { wrec[16] = max(wrec[16], m0); wrec[17] = max(wrec[17], m1); wrec[18] = max(wrec[18], m2); }
      }
      if (!SUB && p.bbox && p.mode != kPointList) {
        const bool neg = valid && half == 0 && sdf < p.neg_thr;
        int extra = bad;
        if (__any(neg || extra != 0)) {      // (wave-uniform: most tiles of a sweep hold no negative voxel)
          int i0, i1, i2;                    // (behind the early return: two integer divisions per lane)
          lattice_ijk(pi, p.N, i0, i1, i2);
          int a0 = neg ? i0 : 0x7fffffff, a1 = neg ? i1 : 0x7fffffff, a2 = neg ? i2 : 0x7fffffff;
          int b0 = neg ? i0 : -1, b1 = neg ? i1 : -1, b2 = neg ? i2 : -1, n = neg ? 1 : 0;
#pragma unroll
          for (int m = 32; m >= 1; m >>= 1) {
            a0 = min(a0, __shfl_xor(a0, m)); a1 = min(a1, __shfl_xor(a1, m)); a2 = min(a2, __shfl_xor(a2, m));
            b0 = max(b0, __shfl_xor(b0, m)); b1 = max(b1, __shfl_xor(b1, m)); b2 = max(b2, __shfl_xor(b2, m));
            n += __shfl_xor(n, m);
            extra += __shfl_xor(extra, m);
          }
          if (lane == 0 && (n | extra)) {
            wrec[0] = min(wrec[0], a0); wrec[1] = min(wrec[1], a1); wrec[2] = min(wrec[2], a2);
            wrec[3] = max(wrec[3], b0); wrec[4] = max(wrec[4], b1); wrec[5] = max(wrec[5], b2);
            wrec[6] += n; wrec[7] += extra;
          }
        }
      } else {
        const unsigned long long m = __ballot(bad);
        if (m && lane == 0) wrec[7] += __popcll(m);
      }
    }   // tiles

    if (lane == 0) {
      // one set of atomics per wave: record 0 = hand (MLP 0), record 1 = object (MLP 1); word 7: lanes out of range (0 unless the
      // fp16 planes overflowed; the host then re-calibrates)
      if (p.bbox && p.mode != kPointList) {
        int* out = p.bbox + (head == 0 ? 0 : 8);
        if (wrec[6]) {
          atomicMin(out + 0, wrec[0]); atomicMin(out + 1, wrec[1]); atomicMin(out + 2, wrec[2]);
          atomicMax(out + 3, wrec[3]); atomicMax(out + 4, wrec[4]); atomicMax(out + 5, wrec[5]);
          atomicAdd(out + 6, wrec[6]);
        }
        if (wrec[7]) atomicAdd(p.bbox + (head == 0 ? 7 : 15), wrec[7]);
      }
      if (p.status) {
        if (wrec[7]) atomicAdd(p.status, wrec[7]);
        int* peak = p.status + 4 + 4 * head;       // [4..6] MLP 0, [8..10] MLP 1: largest plane value of h0 / h1 / h2 (float bits)
        atomicMax(peak + 0, wrec[16]); atomicMax(peak + 1, wrec[17]); atomicMax(peak + 2, wrec[18]);
      }
    }
  }   // MLPs
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (!SUB && p.status && p.mode != kPointList && blockIdx.x == 0 && tid == 0) reinterpret_cast<long long*>(p.status + 12)[1] = clock64();
}

// The __global__ instantiations live in k1hw_kernels.hip; tools/k1h_ablate.hip (-DWIDE=1) instantiates its own.

}  // namespace asdf
