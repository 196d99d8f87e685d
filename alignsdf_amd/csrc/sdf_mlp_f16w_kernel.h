// K1h, the W form (round 6): the split-half SeparateDecoder kernel of sdf_mlp_f16_kernel.h with its three hidden GEMMs on
// v_mfma_f32_16x16x32_f16 instead of v_mfma_f32_32x32x16_f16 - affine point features (KP = 2), two fp16 planes, one group of 32
// points per wave, full lattice or voxel list.  Its OWN copy of the stage and of the body: the 32-wide forms sit at the 512-register
// limit, and the same W paths as a template parameter of their body - every one of them `if constexpr` - still cost the
// CombinedDecoder form 60 B and the subset form 20 B of scratch.  Everything that is not the matrix instruction (plane split,
// LDS-DMA ring, constants image, range guard, box fold, status record) is sdf_mlp_f16_kernel.h's and is used from there; the
// dead branches of the other forms (PL = 1, G = 2, TWO_OUT, NeRF features) are still spelled out in the copy - PL, G, KP, TWO_OUT are
// constants here - so that the two bodies can be compared line by line.
#pragma once
#include "sdf_mlp_f16_kernel.h"

namespace asdf {

// ---- round 6: the W form (template parameter W of the body) - the same GEMMs on v_mfma_f32_16x16x32_f16 ----------------------------
// Under real operands the part is power-managed and a launch takes the time its ENERGY takes (profiles/r06_k1h_front_ab.txt), and the
// bare instruction streams differ: on K1h's operand pattern and split-half data the matrix pipe SUSTAINS 2.04-2.08 PFLOP/s with
// 16x16x32 where it sustains 1.80-1.82 with 32x32x16 (tools/mfma_f16_energy_bench.hip, profiles/r06_mfma_shape_energy.txt).
// Operand maps: A lane l holds A[i = l & 15][k-slot (l >> 4, e)], B lane l holds B[k-slot (l >> 4, e)][j = l & 15], e = 0..7 (the
// SAME slot in both, so which k the hardware gives a slot never matters: the host packs the weight of the input feature that sits in
// that slot of B); D lane l, register r holds D[row = 4 (l >> 4) + r][col = l & 15].
// A wave's 32 points are two GROUPS of 16 (g = 0, 1); an output tile stays 32 features = two HALVES of 16 (fh = 0, 1); its accumulator
// stays ONE f32x16 = four 16x16 tiles, register 8 g + 4 fh + r <-> (feature 32 T + 16 fh + 4 (l >> 4) + r, point 16 g + (l & 15)).
// split_part's access pattern (acc[e], acc[8 + e] -> element e of two operands) then needs NO change: element e = 4 fh + r of
// xh[2 T + g] = the B operand of K32-block T of the next layer for group g, whose slot (q, e) holds feature 32 T + 16 (e >> 2) + 4 q
// + (e & 3).  A record of the stream (2 KiB, [plane][lane][8 halves]) = (tile, feature half fh, K32-block j), feature half OUTER in
// layers 1 and 3 (32 records per tile: fh = i / 16, j = i % 16) and K32-block outer in layer 2 (16 records: j = i / 2, fh = i % 2 - the
// deferred epilogue of layer 1's last tile finishes the last K32-block's operands only in K-block 8 of layer 2's first tile); record i reads the B operands xh / xl[2 j + g] and feeds accumulator
// registers 8 g + 4 fh .. + 3 with six 16-clock MFMAs - three per group back to back (W_hi x_lo, W_lo x_hi, W_hi x_hi), the groups in
// snake order from record to record, so that five of six MFMAs continue the accumulator of the MFMA in front of them.  The same 96
// matrix-pipe clocks, the same LDS and L2 -> LDS traffic, the same stage structure as the 32x32x16 form.  The point-feature products of layers 0 / 2 run on
// v_mfma_f32_16x16x4_f32 (K = 4 = x, y, z, pad in ONE instruction per feature half and group).  Biases, w4 and the point fragments
// are read from the SAME constants image (another gather of the same words); only the weight stream has its own image
// (pack.h: pack_decoder_f16w).
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
  // reads + copies cost 16 v_accvgpr_write per tile.  (The second pointer is opaque so that the reads are not merged.)
  const float* w0 = tile_words + (q & 1) * 16 + 4 * (q >> 1);
  int dup = 0;
  asm volatile("" : "+v"(dup));      // (an opaque ZERO, not an opaque pointer: the address stays an LDS address)
  const float* w1 = w0 + dup;
  f32x16 r;
  acc_set4(r, 0, *reinterpret_cast<const f32x4*>(w0)); acc_set4(r, 4, *reinterpret_cast<const f32x4*>(w0 + 8));
  acc_set4(r, 8, *reinterpret_cast<const f32x4*>(w1)); acc_set4(r, 12, *reinterpret_cast<const f32x4*>(w1 + 8));
  return r;
}


#ifndef ASDF16_W_ORDER
#define ASDF16_W_ORDER 3         // W form: order of a tile's records and of a record's six MFMAs.  3 (shipped): feature half outer, the
                                 // groups' three MFMAs back to back, snake order of the groups; 0 / 1 / 2: (K32-block, feature half) records with
                                 // product sum outer + group inner / group outer / A-operand reuse - timing experiments against the image of
                                 // order 3 (wrong results), profiles/r06_k1h_shape_ab.txt
#endif

// (W form: the accumulator is four independent quads - pinned one by one, or the 512-bit constraint makes the compiler gather them
// into one aligned tuple through 16 v_accvgpr_read / write pairs per tile)
template <int PL = 2, bool W = true>
__device__ __forceinline__ void pin_acc_w(f32x16& acc) {
  if constexpr (W) {
    if (ASDF16_PIN_ACC) {
#pragma unroll
      for (int o = 0; o < 16; o += 4) {
        f32x4 q;
        q[0] = acc[o]; q[1] = acc[o + 1]; q[2] = acc[o + 2]; q[3] = acc[o + 3];
        asm volatile("" : "+a"(q));
        acc[o] = q[0]; acc[o + 1] = q[1]; acc[o + 2] = q[2]; acc[o + 3] = q[3];
      }
    }
    return;
  }
  if (PL == 2 ? ASDF16_PIN_ACC : ASDF16_PIN_ACC_P1) asm volatile("" : "+a"(acc));
}


// One LDS-DMA piece with M0 declared CLOBBERED instead of saved and restored around the instruction (5 -> 3 instructions per piece).
// The 32-wide form hides a piece in the 32 clocks of an MFMA; under the W form's 16-clock MFMAs the two extra scalar moves of every
// piece showed (without its LDS-DMA instructions the W form ran 10-11 % faster, the 32-wide form 6 %).
#ifndef ASDF16_W_M0_CLOBBER
#define ASDF16_W_M0_CLOBBER 1
#endif
template <int P>
__device__ __forceinline__ void dma_piece_w(const float* src, unsigned dst) {
#if ASDF16_W_M0_CLOBBER == 2
  // (timing experiment: M0 written once per four pieces - it would have to survive between asm statements)
  if ((P & 3) == 0)
    asm volatile(
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, off offset:%c2"
        :
        : "v"(src + (P >> 2) * 1024), "s"(dst + (P >> 2) * 4096), "i"((P & 3) * 1024)
        : "memory", "m0");
  else
    asm volatile("global_load_lds_dwordx4 %0, off offset:%c1" : : "v"(src + (P >> 2) * 1024), "i"((P & 3) * 1024) : "memory");
#elif ASDF16_W_M0_CLOBBER
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

// One stage = kS16Kb K-blocks of one 32-row output tile: [kblock][plane hi / lo][lane][8 halves].
// On entry (ah[i], al[i]) hold the A fragments of K-blocks 0 .. PREFETCH-1 of THIS stage; on exit those of the next
// stage in stream order.  Every K-block is one scheduling region
//     [A-fragment reads of K-block kb + PREFETCH, pre(kb)]  fence  [3 MFMAs (+ DMA pieces), epi(kb)]  fence
// pre(kb) carries LDS reads whose results are wanted a K-block (or half a tile) later, epi(kb) the VALU work of the
// deferred epilogue.  The fence behind the reads is what keeps them AHEAD of the K-block's MFMAs: left to itself the
// scheduler sinks every LDS read to one MFMA (32 cycles) in front of its first use and the wave then sits in
// s_waitcnt for the rest of the LDS latency - 26 % of the wave cycles in the round-1 kernel (SQ_WAIT_ANY).
// ABL (timing only, tools/k1h_ablate.hip): 1 = no DMA / wait / barrier, 16 = no barrier,
// 32 = no DMA instructions (waits and barriers kept; the ring keeps the four stages loaded at the head start).
template <int KB, int Q, int SLOT, int ABL, int PL, int G, bool STEPS, bool W, class Pre, class Epi>
__device__ __forceinline__ void stage16w(f32x16& acc, f32x16& accb, const h8 (&xh)[KB], const h8 (&xl)[KB], const float* ring,
                                        const float* next_src, unsigned lds_ring_base, int lane, int wave,
                                        h8 (&ah)[S16<PL, G>::kPrefetch], h8 (&al)[S16<PL, G>::kPrefetch], Pre&& pre, Epi&& epi) {
  constexpr int PF = S16<PL, G>::kPrefetch;
  constexpr int BKB = PL == 2 ? ASDF16_BARRIER_KB : ASDF16_BARRIER_KB_P1;
  constexpr int nslot = (SLOT + kRing - 1) % kRing;   // slot of stage (this - 1), refilled with stage (this + 3)
  using SG = S16<PL, G>;
  const float* src = next_src + wave * SG::kWaveFloats + lane * 4;
  const unsigned dst = lds_ring_base + (nslot * SG::kFloats + wave * SG::kWaveFloats) * 4;
  const h8* cur = reinterpret_cast<const h8*>(ring + SLOT * SG::kFloats) + lane;
  const h8* nxt = reinterpret_cast<const h8*>(ring + ((SLOT + 1) % kRing) * SG::kFloats) + lane;
  h8 bufh[kS16Kb + PF], bufl[kS16Kb + PF];
#pragma unroll
  for (int i = 0; i < PF; ++i) { bufh[i] = ah[i]; if (PL == 2) bufl[i] = al[i]; }
#pragma unroll
  for (int kb = 0; kb < kS16Kb; ++kb) {
    if (kb == BKB && !(ABL & 1)) {
      // my pieces of stage (this + 1) were issued 2.5 stages ago; only those of (this + 2) may stay in flight
      if (SG::kPieces == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      if (!(ABL & 16)) __builtin_amdgcn_s_barrier();
    }
    bufh[kb + PF] = kb + PF < kS16Kb ? cur[((kb + PF) * PL + 0) * 64] : nxt[((kb + PF - kS16Kb) * PL + 0) * 64];
    if (PL == 2) bufl[kb + PF] = kb + PF < kS16Kb ? cur[((kb + PF) * 2 + 1) * 64] : nxt[((kb + PF - kS16Kb) * 2 + 1) * 64];
    pre(kb);
#if ASDF16_LOADS_FIRST
    __builtin_amdgcn_sched_barrier(0);
#endif
    constexpr int base = Q * kS16Kb;
#pragma unroll
    for (int j = 0; j < SG::kMfmas; ++j) {
      if constexpr (W) {
        // W form: record kb = feature half kb & 1 of K32-block (base + kb) >> 1; (W_hi, x_lo), (W_lo, x_hi), (W_hi, x_hi) for the two
        // point groups in turn - every MFMA has an independent one between itself and the next on its accumulator
#if ASDF16_W_ORDER == 3
        // feature-half OUTER over the tile's records, the groups' three MFMAs back to back, the groups in snake order from record to
        // record: record i = (fh = i / (KB / 2), K32-block i % (KB / 2)); every MFMA but one per record continues the accumulator of
        // the MFMA right in front of it (the matrix pipe forwards it: no accumulator read), which the part rewards with a higher
        // clock - measured against product-sum-outer / group-inner: 71.9 against 73.8 ms per N = 256 sweep, same box
        // (Layer 2 - 16 records per tile - keeps K32-block outer: the epilogue of layer 1's LAST tile rides in K-blocks 1 .. 8 of layer
        // 2's first tile and finishes the operands of the last K32-block there; feature half outer would read them at record 7.)
        const int ri = base + kb;
        const int xb = KB == 32 ? 2 * (ri % 16) : (ri & ~1), o = (KB == 32 ? ri / 16 : (ri & 1)) * 4;
        f32x4 s0 = acc_get4(acc, o), s1 = acc_get4(acc, 8 + o);
        {
          const bool snake = ri & 1;
          f32x4& sa = snake ? s1 : s0; f32x4& sb = snake ? s0 : s1;
          const int xa = xb + (snake ? 1 : 0), xc = xb + (snake ? 0 : 1);
          if (j == 0) { sa = ASDF_MFMA16W(bufh[kb], xl[xa], sa); sa = ASDF_MFMA16W(bufl[kb], xh[xa], sa); }
          // (the SAME order of the three products for both groups and every record: a voxel's bits must not depend on the lane it
          // happens to sit in - the voxel lists of the subset form place it anywhere)
          if (j == 1) { sa = ASDF_MFMA16W(bufh[kb], xh[xa], sa); sb = ASDF_MFMA16W(bufh[kb], xl[xc], sb); }
          if (j == 2) { sb = ASDF_MFMA16W(bufl[kb], xh[xc], sb); sb = ASDF_MFMA16W(bufh[kb], xh[xc], sb); }
        }
#else
        const int xb = (base + kb) & ~1, o = (kb & 1) * 4;
        f32x4 s0 = acc_get4(acc, o), s1 = acc_get4(acc, 8 + o);
#endif
#if ASDF16_W_ORDER == 3
#elif ASDF16_W_ORDER == 0
        const h8& af = j == 1 ? bufl[kb] : bufh[kb];
        s0 = ASDF_MFMA16W(af, j == 0 ? xl[xb] : xh[xb], s0);
        s1 = ASDF_MFMA16W(af, j == 0 ? xl[xb + 1] : xh[xb + 1], s1);
#elif ASDF16_W_ORDER == 2
        // (W_hi, x_lo), (W_hi, x_hi), (W_lo, x_hi): four MFMAs in a row on the same A operand (timing experiment)
        const h8& af = j == 2 ? bufl[kb] : bufh[kb];
        s0 = ASDF_MFMA16W(af, j == 0 ? xl[xb] : xh[xb], s0);
        s1 = ASDF_MFMA16W(af, j == 0 ? xl[xb + 1] : xh[xb + 1], s1);
#else
        // group-outer (timing experiment): the three MFMAs of a group back to back on its accumulator
        if (j == 0) { s0 = ASDF_MFMA16W(bufh[kb], xl[xb], s0); s0 = ASDF_MFMA16W(bufl[kb], xh[xb], s0); }
        if (j == 1) { s0 = ASDF_MFMA16W(bufh[kb], xh[xb], s0); s1 = ASDF_MFMA16W(bufh[kb], xl[xb + 1], s1); }
        if (j == 2) { s1 = ASDF_MFMA16W(bufl[kb], xh[xb + 1], s1); s1 = ASDF_MFMA16W(bufh[kb], xh[xb + 1], s1); }
#endif
        acc_set4(acc, o, s0); acc_set4(acc, 8 + o, s1);
      } else {
      if (PL == 1 && j == 0) acc = ASDF_MFMA16(bufh[kb], xh[base + kb], acc);
      else if (PL == 1) accb = ASDF_MFMA16(bufh[kb], xl[base + kb], accb);       // the second point group, same A fragment
      else
#if ASDF16_MFMA_ORDER == 0
      // W_hi . x_lo, W_lo . x_hi, W_hi . x_hi - small terms first
      acc = ASDF_MFMA16(j == 1 ? bufl[kb] : bufh[kb], j == 0 ? xl[base + kb] : xh[base + kb], acc);
#elif ASDF16_MFMA_ORDER == 1
      // W_hi . x_lo, W_hi . x_hi, W_lo . x_hi - consecutive MFMAs share an operand
      acc = ASDF_MFMA16(j == 2 ? bufl[kb] : bufh[kb], j == 0 ? xl[base + kb] : xh[base + kb], acc);
#else
      // W_lo . x_hi, W_hi . x_hi, W_hi . x_lo
      acc = ASDF_MFMA16(j == 0 ? bufl[kb] : bufh[kb], j == 2 ? xl[base + kb] : xh[base + kb], acc);
#endif
      }
      // one DMA piece per MFMA shadow behind the barrier; split-half kernel (round 5): one per K-block, behind its LAST MFMA - the
      // gap that carries the least of a deferred epilogue part
      constexpr bool kDmaPerKb = STEPS && ASDF16_DMA_PER_KB && kS16Kb - BKB >= SG::kPieces;
      const int m = kDmaPerKb ? (j == 2 ? kb - BKB : -1) : (kb - BKB) * SG::kMfmas + j;
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
#ifdef ASDF16_FENCE_EVERY_MFMA
      __builtin_amdgcn_sched_barrier(0);
#endif
      if (PL == 1 && G == 2) {
        // two point groups: each group's share of the deferred epilogue goes behind ONE of the two MFMAs (left to itself the
        // scheduler issues both MFMAs back to back - the second waits a whole MFMA for the pipe - and then all the VALU work)
        epi(kb, j);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (STEPS) {
        // split-half kernel (round 5): the three pieces of a part behind the three MFMAs (split_part).  (The fence sits BEHIND the piece,
        // so a scheduling region is {MFMA j, piece j} and the scheduler puts the dependency-free piece in front of an MFMA that waits for
        // its LDS read now and then - every odd element: pieces 0 + 1 in one gap of 7 VALU instructions, still inside the MFMA's 32
        // clocks.  A fence on BOTH sides gives exactly 4 / 3 / 2 per gap, needs -pragma-unroll-threshold raised for the extra IR, and
        // measured the same: pipe busy 0.788 against 0.785 - not kept.)
        epi(kb, j);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (!((PL == 1 && G == 2) || STEPS)) epi(kb);
    // K-block = scheduling region (hoisted, 16 K-blocks of A fragments do not fit the register file either)
    if (ASDF16_LOADS_FIRST || (kS16Kb > 8 && (kb % ASDF16_SCHED_KB) == ASDF16_SCHED_KB - 1)) __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int i = 0; i < PF; ++i) { ah[i] = bufh[kS16Kb + i]; if (PL == 2) al[i] = bufl[kS16Kb + i]; }
}


// p.stream / p.cst are the split-half images here (pack_decoder_f16).  KP = K-steps of the point features on the fp32
// MFMA in layers 0 and 2: 2 = affine xyz, 5 / 8 = NeRF encoding of 9 / 15 features (those need 16 KiB stages: their
// constants block is 40 / 75 KiB).
// SUB: the kGridSubset form of sdf_mlp_kernel.h - the points are the lattice voxels listed in p.idx (p.count_dev of them, a
// device word; p.P is the list's capacity), coordinates from the voxel index, outputs scattered in place, no box.
template <int ABL = 0, bool SUB = false>
__device__ __forceinline__ void sdf_mlp_f16w_body(const DecodeParams& p) {
  constexpr bool TWO_OUT = false, W = true;
  constexpr int KP = 2, PL = 2, G = 1;
  static_assert(!W || (PL == 2 && G == 1 && KP == 2 && !TWO_OUT), "W form: split-half, affine point features, SeparateDecoder");
  using CL = CstLayout<KP>;
  using SG = S16<PL, G>;
  static_assert(G == 1 || !TWO_OUT, "two point groups: SeparateDecoder");
  constexpr int kTilePts = kWgPts * G;           // points per workgroup tile
  // the plane split on v_fma_mix (split_part), issued in three pieces, one behind each MFMA of the part's K-block.  SeparateDecoder
  // forms only: the CombinedDecoder forms sit at the register limit with their second dot product - the asm blocks' simultaneous
  // destinations cost them 20 .. 48 B of scratch, and with the pieces their unrolled layer-3 loop exceeds the compiler's full-unroll
  // budget and comes out ROLLED, with indexed registers - and keep the round-2 form (same bits).
  constexpr bool kMix = ASDF16_MIX_SPLIT && PL == 2 && !TWO_OUT;
  constexpr bool kSteps = ASDF16_EPI_STEPS && kMix;
  static_assert(lds_bytes_f16(KP, PL) <= 160 * 1024, "LDS budget");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* ring = smem;
  float* cst = smem + SG::kRingFloats;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;

  static_assert(!SUB || G == 1, "subset mode: one point group");
  // the lattice: by value, or - a fine pass enqueued behind its coarse pass - from the device words asdf_zoom_cube wrote
  float lat_vs = p.vs, lat_o0 = p.o0, lat_o1 = p.o1, lat_o2 = p.o2;
  if (p.lattice) { lat_o0 = p.lattice[0]; lat_o1 = p.lattice[1]; lat_o2 = p.lattice[2]; lat_vs = p.lattice[3]; }
  long long npts = p.P;
  if (SUB) { const long long c = *p.count_dev; npts = c < npts ? c : npts; }
  const long long ntiles = (npts + kTilePts - 1) / kTilePts;
  if ((long long)blockIdx.x >= ntiles) return;

  const unsigned lds_ring_base = (unsigned)(size_t)(__attribute__((address_space(3))) float*)ring;
  // shader-clock stamps of workgroup 0 around a whole-lattice sweep (status words 12..13 begin, 14..15 end; s_memtime counts shader
  // clocks on gfx950): ticks / the launch's HIP-event time = the clock the part actually held under this kernel, which bench.py
  // reports next to the matrix-pipe utilisation.  Stored at once, so nothing stays live across the kernel.
  if (!SUB && p.status && p.mode != kPointList && blockIdx.x == 0 && tid == 0) reinterpret_cast<long long*>(p.status + 12)[0] = clock64();

#pragma unroll 1
  for (int slot = 0; slot < p.num_mlps; ++slot) {
    const int head = p.first_mlp + slot;
    const float* hc = cst;
    // negative-voxel bounding box of this MLP's output(s) + the range report, one record per wave and output in LDS:
    // [0..2] min index, [3..5] max index, [6] count, [7] lanes whose activations left the fp16 range (or whose output is
    // not in [-1, 1]); the second output of a CombinedDecoder uses the record 8 ints further
    int* wrec = reinterpret_cast<int*>(cst + CL::kFloats) + wave * kWrecInts;
    if (lane < kWrecInts) wrec[lane] = lane >= 16 ? 0 : ((lane & 7) < 3 ? 0x7fffffff : ((lane & 7) < 6 ? -1 : 0));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    {
      const f32x4* src4 = reinterpret_cast<const f32x4*>(p.cst + (size_t)head * CL::kFloats);
      for (int i = tid; i < CL::kFloats / 4; i += 256) reinterpret_cast<f32x4*>(cst)[i] = src4[i];
    }
    // (one-plane kernels: the fp16 point-feature / bias operands of layers 0 and 2, behind the per-wave records)
    constexpr bool kPt16 = pt16(KP, PL);
    static_assert(!kPt16 || ASDF16_L0_PIPE, "the fp16 point operands are wired into the pipelined layer 0");
    float* a16s = cst + CL::kFloats + kWaves * kWrecInts;
    if (kPt16) {
      const f32x4* src4 = reinterpret_cast<const f32x4*>(p.a16 + (size_t)head * kA16Floats);
      for (int i = tid; i < kA16Floats / 4; i += 256)
        if (i < kA16LayerFloats / 4 || i >= 2 * kA16LayerFloats / 4) reinterpret_cast<f32x4*>(a16s)[i] = src4[i];      // (layer 2's operands are not used: see the tuning log)
    }
    // (W form: its image lies behind the 32x32x16 one in the same allocation)
    const float* sbase0 = p.stream + (W ? (size_t)kStagesAll * kStageFloats : 0) + (size_t)head * kS16Head * SG::kFloats;
#pragma unroll
    for (int s = 0; s < ((ABL & 33) ? kRing : kRing - 1); ++s) {
      const float* src = sbase0 + (size_t)s * SG::kFloats + wave * SG::kWaveFloats + lane * 4;
      const unsigned dst = lds_ring_base + (s * SG::kFloats + wave * SG::kWaveFloats) * 4;
#pragma unroll
      for (int c = 0; c < SG::kPieces; ++c) lds_dma16(src + c * 256, dst + c * 1024);
    }
    if (ABL & 33) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // my pieces of stage 0 (and my constants loads): those of stages 1 and 2 may stay in flight
    if (SG::kPieces == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __syncthreads();                                      // everybody's pieces of stage 0, and the constants
    h8 ah[SG::kPrefetch], al[SG::kPrefetch];
#pragma unroll
    for (int i = 0; i < SG::kPrefetch; ++i) {
      ah[i] = (reinterpret_cast<const h8*>(ring) + lane)[(i * PL + 0) * 64];
      if (PL == 2) al[i] = (reinterpret_cast<const h8*>(ring) + lane)[(i * 2 + 1) * 64];
    }
    // accumulator -> next layer's planes: S_x of the produced activations / (S_w S_x) of the accumulator (powers of two)
    const float mul1 = hc[CL::kB4 + 2], mul2 = hc[CL::kB4 + 3], mul0 = hc[CL::kB4 + 4];
    // subset mode: list positions from here on are audit picks (see DecodeParams::audit)
    int audit_from = 0x7fffffff;
    if (SUB && p.audit) audit_from = p.audit_from ? *p.audit_from : 0;
    // a tile's bias row as the accumulator's initial value (W form: gathered for the 16x16 tiles)
    auto bias_at = [&](int off, int t) -> f32x16 {
      if constexpr (W) return load_bias16w(hc + off + t * 32, lane);
      else return load_bias16(hc + off + (t * 2 + half) * 16);
    };
    // the fp32 A-fragment word of point-feature step s of tile t (W form: s = the feature half, k = lane >> 4 in ONE K = 4 step)
    auto pt_word = [&](int off, int t, int s) -> float {
      if constexpr (W) return hc[off + (t * 2 + (lane >> 5)) * 64 + ((lane >> 4) & 1) * 32 + 16 * s + (lane & 15)];
      else return hc[off + (t * KP + s) * 64 + lane];
    };

#pragma unroll 1
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
#ifdef ASDF16_SEGMENT_TIMES
      long long seg_t[8];
#endif
      ASDF16_MARK(0);
      const long long pi = tile * kTilePts + wave * (kWavePts * G) + (lane & 31);
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
      // second point group (G == 2): the next 32 points
      const long long pib = pi + kWavePts;
      const bool validb = G == 2 && pib < npts;
      float bpb[KP];
#pragma unroll
      for (int s = 0; s < KP; ++s) bpb[s] = 0.0f;
      float y0 = 0.f, y1 = 0.f, y2 = 0.f;
      if (G == 2) {
        if (p.mode == kPointList) {
          if (validb) { y0 = p.xyz[pib * 3 + 0]; y1 = p.xyz[pib * 3 + 1]; y2 = p.xyz[pib * 3 + 2]; }
        } else {
          grid_point(validb ? pib : 0, p.N, p.mode, lat_vs, lat_o0, lat_o1, lat_o2, y0, y1, y2);
        }
        if (KP == 2) {
          bpb[0] = half ? y1 : y0;
          bpb[1] = half ? 0.0f : y2;
        } else {
#pragma unroll
          for (int s = 0; s < KP; ++s) bpb[s] = 2 * s + half < p.pf ? nerf_feature(2 * s + half, y0, y1, y2) : 0.0f;
        }
      }
      // largest plane value (x S_x) this lane hands to the fp16 conversion, per activation vector h0 / h1 / h2: >= 65504
      // is an overflow (range report); the maxima themselves go to the decoder's status record, from which the host
      // calibrates the S_x of each layer
      float amax = 0.0f, amax1 = 0.0f, amax2 = 0.0f;
      float bp[KP];
      if (KP == 2) {
        bp[0] = half ? x1 : x0;
        bp[1] = half ? 0.0f : x2;
      } else {
#pragma unroll
        for (int s = 0; s < KP; ++s) bp[s] = 2 * s + half < p.pf ? nerf_feature(2 * s + half, x0, x1, x2) : 0.0f;
      }
      // W form: the point operand of v_mfma_f32_16x16x4_f32 per group - lane l carries component l >> 4 of (x, y, z, 0) of point
      // 16 g + (l & 15); a lane computed the coordinates of ITS point l & 31, the other group's come from lane l ^ 16
      float bq[2] = {0.0f, 0.0f};
      if constexpr (W) {
        // component q = l >> 4 of point 16 g + (l & 15): the source lane is (l & 15) + 16 g (any lane with that l & 31)
        const int q = lane >> 4;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const int src = (lane & 15) + 16 * g;
          const float c0 = __shfl(x0, src), c1 = __shfl(x1, src), c2 = __shfl(x2, src);
          bq[g] = q == 0 ? c0 : q == 1 ? c1 : q == 2 ? c2 : 0.0f;
        }
      }
      // point-feature products of one tile into its accumulator: af[s] = pt_word(.., t, s)
      auto pt_mfma = [&](f32x16& a, const float* af) {
        if constexpr (W) {
#pragma unroll
          for (int fh = 0; fh < 2; ++fh)
#pragma unroll
            for (int g = 0; g < 2; ++g) acc_set4(a, 8 * g + 4 * fh, ASDF_MFMA4W(af[fh], bq[g], acc_get4(a, 8 * g + 4 * fh)));
        } else {
#pragma unroll
          for (int s = 0; s < KP; ++s) a = ASDF_MFMA(af[s], bp[s], a);
        }
      };
      // one-plane kernels: the points as the fp16 B operand of layers 0 / 2 (sdf_layout.h: kA16Floats) - x T in two planes, T twice
      // (the bias planes' multiplier) on lane half 0; the high planes again on lane half 1 (they meet the weights' low planes)
      auto point_operand = [&](float c0, float c1, float c2, float T) -> h8 {
        const float s0 = c0 * T, s1 = c1 * T, s2 = c2 * T;
        const _Float16 a0 = (_Float16)s0, a1 = (_Float16)s1, a2 = (_Float16)s2;
        h8 r;
        r[0] = a0; r[1] = a1; r[2] = a2;
        if (half == 0) {
          r[3] = (_Float16)(s0 - (float)a0); r[4] = (_Float16)(s1 - (float)a1); r[5] = (_Float16)(s2 - (float)a2);
          r[6] = (_Float16)T; r[7] = (_Float16)T;
        } else {
          r[3] = r[4] = r[5] = r[6] = r[7] = (_Float16)0.0f;
        }
        return r;
      };
      ASDF16_MARK(7);      // (coordinates done)
      const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      h8 bq0, bq0b;
      if (kPt16) {
        bq0 = point_operand(x0, x1, x2, a16s[2 * kA16LayerFloats]);
        if (G == 2) bq0b = point_operand(y0, y1, y2, a16s[2 * kA16LayerFloats]);
      }
      auto a16_frag = [&](int layer, int t) -> h8 {      // this lane's 8 halves of tile t's operand
        return *reinterpret_cast<const h8*>(a16s + layer * kA16LayerFloats + t * kA16TileFloats + lane * 4);
      };
      const float* sbase = sbase0;
      asm volatile("" : "+s"(sbase));
      auto src_of = [&](int s) -> const float* {   // s = stage index within the head + 3
        return sbase + (size_t)(s < kS16Head ? s : s - kS16Head) * SG::kFloats;
      };

      // LDS reads that feed a tile are issued half a tile (or one K-block) ahead of their first use - `pre` slots of
      // stage16 - into registers that are dead at that point: the OTHER accumulator of the double buffer takes the next
      // tile's bias row, `pf2` its point-feature fragments, `w4n` the last-layer weights of the next epilogue part.
      f32x16 acc1[2], acc2[2], acc3[2];
      f32x16 acc1b[2], acc2b[2], acc3b[2];      // G == 2: the accumulators of the second point group
      float pf2[KP];                            // A fragments (fp32 MFMA) of the next layer-2 tile
      float w4c[2], w4n[2], w4bc[2], w4bn[2];   // last-layer weights of the current / next part of the layer-3 epilogue
      // one-plane kernel: all 16 of a tile's, read half a tile ahead (a K-block of 64 cycles is shorter than the LDS latency)
      constexpr bool kW4Tile = ASDF16_W4_TILE && PL == 1 && !TWO_OUT;
      constexpr int kPreKb = PL == 2 ? ASDF16_PRE_KB : ASDF16_PRE_KB_P1;
      // layer 2 has ONE stage per tile: the preload of the next tile's bias row lands in the accumulator the deferred epilogue of
      // the previous tile is still reading until its last part (K-block kEpiShift + kEpiChunks - 1)
      static_assert(kS16Kb == 8 || kPreKb >= kEpiShift + kEpiChunks, "preload K-block inside the epilogue slots");
      f32x16 w4t;
      // (the 8 K-step form of PointFeatSize 15 has no registers to spare for these: it reads its fragments at the point of use)
      constexpr bool kPreloadPf = KP <= 5;
      auto load_pf2 = [&](int t) {
        if (!kPreloadPf) return;
#pragma unroll
        for (int s = 0; s < KP; ++s) pf2[s] = pt_word(CL::kA2, t, s);
      };
      auto load_w4 = [&](int t, int c) {        // accumulator registers 2 c, 2 c + 1 of tile t
        // (W form: register 2 c = group c >> 2, feature half (c >> 1) & 1, r = 2 (c & 1): feature 16 fh + 4 q + r of the tile sits in
        // register 4 (2 fh + (q >> 1)) + r of lane half q & 1 of the 32x32 D-layout image)
        const float* w4 = hc + CL::kW4 + (t * 2 + half) * 16 + 2 * c;
        if constexpr (W) {
          const int wq = lane >> 4;
          w4 = hc + CL::kW4 + (t * 2 + (wq & 1)) * 16 + 4 * (2 * ((c >> 1) & 1) + (wq >> 1)) + 2 * (c & 1);
        }
        const float* w4b = hc + CL::kW4b + (t * 2 + half) * 16 + 2 * c;
        w4n[0] = w4[0]; w4n[1] = w4[1];
        if (TWO_OUT) { w4bn[0] = w4b[0]; w4bn[1] = w4b[1]; }
      };
      auto next_w4 = [&]() {
        w4c[0] = w4n[0]; w4c[1] = w4n[1];
        if (TWO_OUT) { w4bc[0] = w4bn[0]; w4bc[1] = w4bn[1]; }
      };

      // ---- layer 0 (fp32 MFMA, K = 4 point features): planes of relu(.) * S_x
      h8 h0h[2 * kTilesHidden], h0l[2 * kTilesHidden];
      f32x16 acc0[2];
      float pf0[2][KP];
      auto l0_load = [&](int t) {
        if (!kPreloadPf) return;
        acc0[t & 1] = bias_at(CL::kC0, t);
#pragma unroll
        for (int s = 0; s < KP; ++s) pf0[t & 1][s] = pt_word(CL::kA0, t, s);
      };
      auto l0_compute = [&](int t, int g = -1) {
        f32x16 acc = kPreloadPf ? acc0[t & 1] : bias_at(CL::kC0, t);
        f32x16 accb = acc;
        if constexpr (W) pt_mfma(acc, pf0[t & 1]);
        else {
#pragma unroll
        for (int s = 0; s < KP; ++s) {
          const float af = kPreloadPf ? pf0[t & 1][s] : hc[CL::kA0 + (t * KP + s) * 64 + lane];
          if (g != 1) acc = ASDF_MFMA(af, bp[s], acc);
          if (G == 2 && g != 0) accb = ASDF_MFMA(af, bpb[s], accb);
        }
        }
        split_tile<PL, G, kMix>(acc, accb, mul0, h0h[2 * t], h0l[2 * t], h0h[2 * t + 1], h0l[2 * t + 1], amax, g);
      };
      // the split of a tile is ~100 VALU instructions against 128 cycles of fp32 MFMA: layer 0 is VALU-bound when it runs
      // on its own.  Only the tiles the first stage of layer 1 consumes (K-blocks 0 .. kS16Kb-1) are computed up front;
      // the others ride in the epilogue slots of that stage, under its fp16 MFMAs.
      constexpr int kL0Front = (PL == 1 && ASDF16_L0_PIPE) ? kTilesHidden : (kS16Kb / 2 < kEpiChunks ? kTilesHidden : kS16Kb / 2);
      if (PL == 1 && ASDF16_L0_PIPE) {
        // One-plane kernel: all 16 tiles up front, software-pipelined over three accumulator sets - the bias / fragment reads
        // of tile t + 2, then the fp32 MFMAs of tile t + 1, then the fp16 conversions of tile t.  (Tile by tile, and with the
        // second half squeezed into the 64-cycle K-blocks of layer 1's first stage, layer 0 took 21.6 k of this kernel's
        // 124 k cycles per 256-point tile where its VALU work is 8 k: every tile waited for its LDS reads, then for its
        // MFMAs, then converted.)
        f32x16 la[3], lb[3];
        float lf[3][KP];
        h8 lq[3];
        auto l0p_load = [&](int t) {
          if (kPt16) { lq[t % 3] = a16_frag(0, t); return; }      // bias and point-feature columns in one fp16 operand
          la[t % 3] = bias_at(CL::kC0, t);
          if (G == 2) lb[t % 3] = bias_at(CL::kC0, t);
#pragma unroll
          for (int s = 0; s < KP; ++s) lf[t % 3][s] = hc[CL::kA0 + (t * KP + s) * 64 + lane];
        };
        auto l0p_mfma = [&](int t) {
          if (kPt16) {
            la[t % 3] = ASDF_MFMA16(lq[t % 3], bq0, zero16);
            if (G == 2) lb[t % 3] = ASDF_MFMA16(lq[t % 3], bq0b, zero16);
            return;
          }
#pragma unroll
          for (int s = 0; s < KP; ++s) {
            la[t % 3] = ASDF_MFMA(lf[t % 3][s], bp[s], la[t % 3]);
            if (G == 2) lb[t % 3] = ASDF_MFMA(lf[t % 3][s], bpb[s], lb[t % 3]);
          }
        };
        l0p_load(0);
        l0p_load(1);
        acc1[0] = bias_at(CL::kB1, 0);
        if (G == 2) acc1b[0] = acc1[0];
        __builtin_amdgcn_sched_barrier(0);
        l0p_mfma(0);
#pragma unroll
        for (int t = 0; t < kTilesHidden; ++t) {
          if (t + 2 < kTilesHidden) l0p_load(t + 2);
          __builtin_amdgcn_sched_barrier(0);
          if (t + 1 < kTilesHidden) l0p_mfma(t + 1);
          split_tile<PL, G, kMix>(la[t % 3], lb[t % 3], mul0, h0h[2 * t], h0l[2 * t], h0h[2 * t + 1], h0l[2 * t + 1], amax);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
      l0_load(0);
      acc1[0] = bias_at(CL::kB1, 0);
      if (G == 2) acc1b[0] = acc1[0];
#pragma unroll
      for (int t = 0; t < kL0Front; ++t) {
        if (ASDF16_PRELOAD && t + 1 < kTilesHidden) { l0_load(t + 1); __builtin_amdgcn_sched_barrier(0); }
        l0_compute(t);
        if (!ASDF16_PRELOAD && t + 1 < kTilesHidden) l0_load(t + 1);
        if (ASDF16_PRELOAD) __builtin_amdgcn_sched_barrier(0);
      }
      }

#define ASDF_STAGE16(KB, Q, SLOT, ACC, XH, XL, SIDX, PRE, EPI) \
  stage16w<KB, Q, SLOT, ABL, PL, G, kSteps, W>(ACC, ACC##b, XH, XL, ring, src_of((SIDX) + 3), lds_ring_base, lane, wave, ah, al, PRE, EPI)

      ASDF16_MARK(1);
      // (split-half kernel: the second argument of an epilogue callback is the PIECE of the part - stage16 calls it behind each of
      // the K-block's three MFMAs - and `er` carries a part's two ReLUs from piece to piece)
      float er[2] = {0.0f, 0.0f};
      // ---- layer 1: 512 -> 256; epilogue of tile t-1 rides in tile t
      h8 h1h[2 * kTilesL1], h1l[2 * kTilesL1];
      if (ABL & 4) for (int t = 0; t < 2 * kTilesL1; ++t) { h1h[t] = h0h[t]; h1l[t] = h0l[t]; }
#pragma unroll
      for (int t = 0; t < kTilesL1; ++t) {
        f32x16& acc = acc1[t & 1];
        f32x16& accb = (G == 2 ? acc1b : acc1)[t & 1];
        if (!ASDF16_PRELOAD && t > 0) acc = bias_at(CL::kB1, t);
        auto pre = [&](int kb) {       // first stage: the layer-0 tile of the NEXT K-block's epilogue slot
          const int c = kb - kEpiShift;
          if (t == 0 && ASDF16_PRELOAD && kL0Front < kTilesHidden && c >= 0 && c + 1 < kEpiChunks) l0_load(kL0Front + c + 1);
        };
        auto epi = [&](int kb, int g = -1) {
          const int c = kb - kEpiShift;
          if (c < 0 || c >= kEpiChunks) return;
          if (t == 0) {
            if (kSteps && kPreloadPf && kL0Front < kTilesHidden) {
              // a whole layer-0 tile per K-block: its fp32 MFMAs behind the first MFMA, its eight parts over the three gaps (3 + 3 + 2)
              const int T = kL0Front + c;
              if (g == 0) {
                if constexpr (W) pt_mfma(acc0[T & 1], pf0[T & 1]);
                else {
#pragma unroll
                  for (int s = 0; s < KP; ++s) acc0[T & 1] = ASDF_MFMA(pf0[T & 1][s], bp[s], acc0[T & 1]);
                }
              }
#pragma unroll
              for (int e = 0; e < 8; ++e)
                if (e / 3 == g) split_part<PL, G, kMix>(acc0[T & 1], acc0[T & 1], mul0, h0h[2 * T], h0l[2 * T], h0h[2 * T + 1], h0l[2 * T + 1], amax, e);
              return;
            }
            if (kSteps && g > 0) return;
            if (kL0Front < kTilesHidden) {       // layer-0 tiles 8 .. 15: consumed by the next stage
              l0_compute(kL0Front + c, kSteps ? -1 : g);
              if (!ASDF16_PRELOAD && c + 1 < kEpiChunks) l0_load(kL0Front + c + 1);
            }
            return;
          }
          if (ABL & 4) { asm volatile("" :: "v"(acc1[(t - 1) & 1])); return; }
          if (kSteps ? g <= 0 : g != 1) pin_acc_w<PL, W>(acc1[(t - 1) & 1]);
          if (G == 2 && g != 0) pin_acc_w<PL, W>(acc1b[(t - 1) & 1]);
          split_part<PL, G, kMix>(acc1[(t - 1) & 1], acc1b[(t - 1) & 1], mul1, h1h[2 * (t - 1)], h1l[2 * (t - 1)], h1h[2 * (t - 1) + 1],
                            h1l[2 * (t - 1) + 1], amax1, c, kSteps ? -1 : g, kSteps ? g : -1, er);
        };
        auto pre_last = [&](int c) {   // last stage: bias row (and point fragments) of the next tile
          if (!ASDF16_PRELOAD || c != kPreKb) return;
          if (t + 1 < kTilesL1) { acc1[(t + 1) & 1] = bias_at(CL::kB1, (t + 1)); if (G == 2) acc1b[(t + 1) & 1] = acc1[(t + 1) & 1]; }
          else { acc2[0] = bias_at(CL::kC2, 0); if (G == 2) acc2b[0] = acc2[0]; load_pf2(0); }
        };
#if ASDF16_STAGE_KB == 8
        ASDF_STAGE16(32, 0, 0, acc, h0h, h0l, t * 4 + 0, pre, epi);
        ASDF_STAGE16(32, 1, 1, acc, h0h, h0l, t * 4 + 1, NoOp16(), NoOp16());
        ASDF_STAGE16(32, 2, 2, acc, h0h, h0l, t * 4 + 2, NoOp16(), NoOp16());
        ASDF_STAGE16(32, 3, 3, acc, h0h, h0l, t * 4 + 3, pre_last, NoOp16());
#else
        if (t & 1) {
          ASDF_STAGE16(32, 0, 2, acc, h0h, h0l, t * 2 + 0, pre, epi);
          ASDF_STAGE16(32, 1, 3, acc, h0h, h0l, t * 2 + 1, pre_last, NoOp16());
        } else {
          ASDF_STAGE16(32, 0, 0, acc, h0h, h0l, t * 2 + 0, pre, epi);
          ASDF_STAGE16(32, 1, 1, acc, h0h, h0l, t * 2 + 1, pre_last, NoOp16());
        }
#endif
      }

      ASDF16_MARK(2);
      // ---- layer 2: [h1 (256) | xyz (4, fp32 MFMA, pre-scaled A fragments)] -> 512
      h8 h2h[2 * kTilesHidden], h2l[2 * kTilesHidden];
      if (ABL & 4) for (int t = 0; t < 2 * kTilesHidden; ++t) { h2h[t] = h0h[t]; h2l[t] = h0l[t]; }
      // one tile of layer 2; SLOT is the ring slot of its first stage (a tag type: the slot must be a compile-time constant)
      auto l2_tile = [&](int t, auto slot_tag) {
        constexpr int SLOT = decltype(slot_tag)::value;
        f32x16& acc = acc2[t & 1];
        f32x16& accb = (G == 2 ? acc2b : acc2)[t & 1];
        if (!ASDF16_PRELOAD) { acc = bias_at(CL::kC2, t); load_pf2(t); }
        if constexpr (W) pt_mfma(acc, pf2);
        else {
#pragma unroll
        for (int s = 0; s < KP; ++s) {
          const float af = kPreloadPf ? pf2[s] : hc[CL::kA2 + (t * KP + s) * 64 + lane];
          acc = ASDF_MFMA(af, bp[s], acc);
          if (G == 2) accb = ASDF_MFMA(af, bpb[s], accb);
        }
        }
        auto epi = [&](int kb, int g = -1) {
          const int c = kb - kEpiShift;
          if (c < 0 || c >= kEpiChunks) return;
          if (ABL & 4) { asm volatile("" :: "v"(acc2[(t + 1) & 1]), "v"(acc1[1])); return; }
          if (t > 0) {
            if (kSteps ? g <= 0 : g != 1) pin_acc_w<PL, W>(acc2[(t - 1) & 1]);
            if (G == 2 && g != 0) pin_acc_w<PL, W>(acc2b[(t - 1) & 1]);
            split_part<PL, G, kMix>(acc2[(t - 1) & 1], acc2b[(t - 1) & 1], mul2, h2h[2 * (t - 1)], h2l[2 * (t - 1)], h2h[2 * (t - 1) + 1],
                              h2l[2 * (t - 1) + 1], amax2, c, kSteps ? -1 : g, kSteps ? g : -1, er);
          } else {
            if (kSteps ? g <= 0 : g != 1) pin_acc_w<PL, W>(acc1[(kTilesL1 - 1) & 1]);
            if (G == 2 && g != 0) pin_acc_w<PL, W>(acc1b[(kTilesL1 - 1) & 1]);
            split_part<PL, G, kMix>(acc1[(kTilesL1 - 1) & 1], acc1b[(kTilesL1 - 1) & 1], mul1, h1h[2 * kTilesL1 - 2], h1l[2 * kTilesL1 - 2],
                              h1h[2 * kTilesL1 - 1], h1l[2 * kTilesL1 - 1], amax1, c, kSteps ? -1 : g, kSteps ? g : -1, er);      // K-blocks 14, 15: consumed at the end of this tile, after chunk 7
          }
        };
        auto pre_last = [&](int c) {
          if (!ASDF16_PRELOAD || c != kPreKb) return;
          if (t + 1 < kTilesHidden) {
            acc2[(t + 1) & 1] = bias_at(CL::kC2, (t + 1));
            if (G == 2) acc2b[(t + 1) & 1] = acc2[(t + 1) & 1];
            load_pf2(t + 1);
          } else {
            acc3[0] = bias_at(CL::kB3, 0);
            if (G == 2) acc3b[0] = acc3[0];
          }
        };
        constexpr int S0 = 256 / kS16Kb;         // stages of layer 1
#if ASDF16_STAGE_KB == 8
        ASDF_STAGE16(16, 0, SLOT, acc, h1h, h1l, S0 + t * 2 + 0, NoOp16(), epi);
        ASDF_STAGE16(16, 1, SLOT + 1, acc, h1h, h1l, S0 + t * 2 + 1, pre_last, NoOp16());
#else
        ASDF_STAGE16(16, 0, SLOT, acc, h1h, h1l, S0 + t, pre_last, epi);      // one stage per tile
#endif
      };
#pragma unroll
      for (int tt = 0; tt < kTilesHidden / 4; ++tt) {
#if ASDF16_STAGE_KB == 8
        l2_tile(4 * tt + 0, std::integral_constant<int, 0>());
        l2_tile(4 * tt + 1, std::integral_constant<int, 2>());
        l2_tile(4 * tt + 2, std::integral_constant<int, 0>());
        l2_tile(4 * tt + 3, std::integral_constant<int, 2>());
#else
        l2_tile(4 * tt + 0, std::integral_constant<int, 0>());
        l2_tile(4 * tt + 1, std::integral_constant<int, 1>());
        l2_tile(4 * tt + 2, std::integral_constant<int, 2>());
        l2_tile(4 * tt + 3, std::integral_constant<int, 3>());
#endif
      }

      ASDF16_MARK(3);
      // ---- layer 3 (512 -> 512) fused with layer 4 (dot with w4 / (S_w3 S_x)) and tanh
      float part = 0.0f, partb = 0.0f, partg = 0.0f;      // partg: the second point group's dot product (G == 2)
      // accumulator registers 2 c, 2 c + 1 of a finished tile into the last-layer dot product(s), weights from (w4c, w4bc)
      auto dot_w4_part = [&](const f32x16& a, const f32x16& ab, int c, int g = -1, int only = -1) {      // only: 0 / 1 = that register of the pair
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          if (only >= 0 && only != r) continue;
          const float w = kW4Tile ? w4t[2 * c + r] : w4c[r];
          if (PL == 1 && ASDF16_P1_FOLD) {
            // relu(a) w = a (w / 2) + |a| (w / 2): two FMAs like max + FMA, and a NaN / infinity of either sign stays one (the image's
            // last-layer weights carry the 1 / 2)
            if (g != 1) {
              const float v = a[2 * c + r];
              part = fmaf(fabsf(v), w, fmaf(v, w, part));
              if (TWO_OUT) partb = fmaf(fabsf(v), w4bc[r], fmaf(v, w4bc[r], partb));
            }
            if (G == 2 && g != 0) { const float v = ab[2 * c + r]; partg = fmaf(fabsf(v), w, fmaf(v, w, partg)); }
            continue;
          }
          if constexpr (W) {        // registers 8 .. 15 are the second point group's: its own dot product
            const float v = __int_as_float(max(__float_as_int(a[2 * c + r]), 0));
            if (c < 4) part = fmaf(v, w, part);
            else partg = fmaf(v, w, partg);
            continue;
          }
          if (g != 1) {
            const float v = __int_as_float(max(__float_as_int(a[2 * c + r]), 0));
            part = fmaf(v, w, part);
            if (TWO_OUT) partb = fmaf(v, w4bc[r], partb);
          }
          if (G == 2 && g != 0) partg = fmaf(__int_as_float(max(__float_as_int(ab[2 * c + r]), 0)), w, partg);
        }
        if (TWO_OUT) asm volatile("" : "+v"(part), "+v"(partb));
        if (G == 2) asm volatile("" : "+v"(part), "+v"(partg));
      };
#pragma unroll
      for (int t = 0; t < kTilesHidden; ++t) {
        f32x16& acc = acc3[t & 1];
        f32x16& accb = (G == 2 ? acc3b : acc3)[t & 1];
        if (!ASDF16_PRELOAD) acc = bias_at(CL::kB3, t);
        auto pre = [&](int kb) {       // first stage: w4 of the next epilogue part
          const int c = kb - kEpiShift;
          if (!kW4Tile && t > 0 && c >= 0 && c + 1 < kEpiChunks) load_w4(t - 1, c + 1);
        };
        auto epi = [&](int kb, int g = -1) {
          const int c = kb - kEpiShift;
          if (c < 0 || c >= kEpiChunks) return;
          if (ABL & 4) { asm volatile("" :: "v"(acc3[(t + 1) & 1]), "v"(acc2[1])); return; }
          if (t > 0) {
            if (kSteps ? g <= 0 : g != 1) pin_acc_w<PL, W>(acc3[(t - 1) & 1]);
            if (G == 2 && g != 0) pin_acc_w<PL, W>(acc3b[(t - 1) & 1]);
            if (kSteps) {        // piece 0 / 1: one register of the pair each; piece 2: the next pair's weights
              if (g < 2) dot_w4_part(acc3[(t - 1) & 1], acc3b[(t - 1) & 1], c, -1, g);
              if (!kW4Tile && (g == 2 || g < 0)) next_w4();
            } else {
              dot_w4_part(acc3[(t - 1) & 1], acc3b[(t - 1) & 1], c, g, -1);
              if (!kW4Tile && g != 0) next_w4();            // (both groups use the same weights: rotate behind the second)
            }
          } else {
            if (kSteps ? g <= 0 : g != 1) pin_acc_w<PL, W>(acc2[(kTilesHidden - 1) & 1]);
            if (G == 2 && g != 0) pin_acc_w<PL, W>(acc2b[(kTilesHidden - 1) & 1]);
            split_part<PL, G, kMix>(acc2[(kTilesHidden - 1) & 1], acc2b[(kTilesHidden - 1) & 1], mul2, h2h[2 * kTilesHidden - 2],
                              h2l[2 * kTilesHidden - 2], h2h[2 * kTilesHidden - 1], h2l[2 * kTilesHidden - 1], amax2, c, kSteps ? -1 : g,
                              kSteps ? g : -1, er);  // K-blocks 30, 31: end of this tile
          }
        };
        auto pre_last = [&](int c) {   // last stage: bias row of the next tile, w4 of the first part of this tile's epilogue
          if (c != kPreKb) return;
          if (ASDF16_PRELOAD && t + 1 < kTilesHidden) {
            acc3[(t + 1) & 1] = bias_at(CL::kB3, (t + 1));
            if (G == 2) acc3b[(t + 1) & 1] = acc3[(t + 1) & 1];
          }
          if (kW4Tile) { w4t = load_bias16(hc + CL::kW4 + (t * 2 + half) * 16); return; }
          load_w4(t, 0);
          next_w4();
        };
        constexpr int S0 = 512 / kS16Kb;         // stages of layers 1 and 2
#if ASDF16_STAGE_KB == 8
        ASDF_STAGE16(32, 0, 0, acc, h2h, h2l, S0 + t * 4 + 0, pre, epi);
        ASDF_STAGE16(32, 1, 1, acc, h2h, h2l, S0 + t * 4 + 1, NoOp16(), NoOp16());
        ASDF_STAGE16(32, 2, 2, acc, h2h, h2l, S0 + t * 4 + 2, NoOp16(), NoOp16());
        ASDF_STAGE16(32, 3, 3, acc, h2h, h2l, S0 + t * 4 + 3, pre_last, NoOp16());
#else
        if (t & 1) {
          ASDF_STAGE16(32, 0, 2, acc, h2h, h2l, S0 + t * 2 + 0, pre, epi);
          ASDF_STAGE16(32, 1, 3, acc, h2h, h2l, S0 + t * 2 + 1, pre_last, NoOp16());
        } else {
          ASDF_STAGE16(32, 0, 0, acc, h2h, h2l, S0 + t * 2 + 0, pre, epi);
          ASDF_STAGE16(32, 1, 1, acc, h2h, h2l, S0 + t * 2 + 1, pre_last, NoOp16());
        }
#endif
      }
      ASDF16_MARK(4);
      // the last tile's epilogue has no MFMA stream to hide under
#pragma unroll
      for (int c = 0; c < kEpiChunks; ++c) {
        if (!kW4Tile && c + 1 < kEpiChunks) load_w4(kTilesHidden - 1, c + 1);
        dot_w4_part(acc3[(kTilesHidden - 1) & 1], acc3b[(kTilesHidden - 1) & 1], c);
        if (!kW4Tile) next_w4();
      }
#undef ASDF_STAGE16
      auto tanh_out = [&](float x) -> float {
#if ASDF16_FAST_TANH
        if (PL == 1) return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x));
#endif
        return tanhf(x);
      };
      if constexpr (W) {
        // the four lane groups hold the four quarters of every feature half: sum them, then every lane takes the sum of ITS point
        // l & 31 (group (l >> 4) & 1) - everything below is the 32x32 form's
        part += __shfl_xor(part, 16); partg += __shfl_xor(partg, 16);
        part += __shfl_xor(part, 32); partg += __shfl_xor(partg, 32);
        part = (lane & 16) ? partg : part;
      } else {
        part += __shfl_xor(part, 32);
      }
      const float pre = part + hc[CL::kB4];           // (pre-activations: the strict range report below looks at THESE)
      const float sdf = tanh_out(pre);
      float sdfb = 1.0f, preb = 0.0f, preg = 0.0f;
      if (TWO_OUT) {
        partb += __shfl_xor(partb, 32);
        preb = partb + hc[CL::kB4 + 1];
        sdfb = tanh_out(preb);
      }
      float sdfg = 0.0f;           // the second point group's output (G == 2)
      if (G == 2) {
        partg += __shfl_xor(partg, 32);
        preg = partg + hc[CL::kB4];
        sdfg = tanh_out(preg);
      }
      const bool is_hand = head == 0;
      if (SUB) {
        // subset mode replaces values: the largest change is the measured error of the arithmetic it corrects.  Marked voxels
        // report to status[3]; audit picks (voxels the one-plane sweep decided by sign alone) to the audit record, together
        // with the number of picks whose sign the exact value contradicts.  Reduced over the wave on the float BITS (a NaN
        // is a huge pattern and must survive the reduction), then one set of atomics per wave.
        int dm = 0, da = 0, flips = 0, na = 0;
        float sq = 0.0f;           // sum of squared audit errors: sigma of the one-plane error, next to its maximum
        if (valid && half == 0) {
          const bool aud = p.audit && pi >= (long long)audit_from;
          auto replace = [&](float* vol, float now) {
            const float before = vol[po];
            const int d = __float_as_int(fabsf(now - before));
            if (aud) { da = max(da, d); flips += ((before < 0.0f) != (now < 0.0f)) ? 1 : 0; ++na; sq += (now - before) * (now - before); }
            else dm = max(dm, d);
            // (a list of audit picks alone - the box sweep's - leaves the scratch volume as the one-plane kernel wrote it: that audit
            // runs BESIDE the collection of the sweep's candidates, which reads the same volume; round 5)
            if (!aud || p.audit_from) vol[po] = now;
          };
          float* out = is_hand ? p.sdf0 : p.sdf1;
          if (out) replace(out, sdf);
          if (TWO_OUT && p.sdf1) replace(p.sdf1, sdfb);
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
        if (TWO_OUT && p.sdf1) p.sdf1[po] = sdfb;
      }
      if (G == 2 && validb && half == 1) {       // lanes 32..63 store the second group: 64 consecutive floats per wave
        float* out = is_hand ? p.sdf0 : p.sdf1;
        if (out) out[pib] = sdfg;
      }
      // every lane reports its own activations (the two halves of a wave hold different features of the same point):
      // a lane is out of range when a value handed to the fp16 conversion reached 65504 (|x| >= 8188) or an output left
      // [-1, 1] (NaN / infinity downstream of an overflow).  The count goes to the decoder's status word - always, not only
      // when the caller passed a bbox buffer - and, for grid sweeps with a bbox, to word 7 / 15 of that record as well.
      // (one-plane kernels, folded image: no running maximum - every ReLU preserves a poisoned value, so an activation that left the
      // fp16 range arrives HERE as a non-finite PRE-activation of the tanh: a NaN, or an infinity.  The check is on that value, not
      // on the output - tanhf returns exactly +-1 for every finite argument beyond ~9, so a decoder that merely saturates in the far
      // field must not read as a range violation: ADVICE r04, it drove such a decoder through four re-calibrations to the fp32 chain)
      constexpr bool kStrict = PL == 1 && ASDF16_P1_FOLD;
      if (kMix) { amax *= mul0; amax1 *= mul1; amax2 *= mul2; }      // (the maxima were taken in front of the rescale)
      const bool act_over = !(fmaxf(amax, fmaxf(amax1, amax2)) < 65504.0f);
      auto out_ok = [&](float v, float pre_v) { return kStrict ? fabsf(pre_v) < INFINITY : fabsf(v) <= 1.0f; };
      const int bad = ((valid || validb) && (act_over || !out_ok(sdf, pre) || (TWO_OUT && !out_ok(sdfb, preb)) ||
                                            (G == 2 && !out_ok(sdfg, preg)))) ? 1 : 0;
      if (!kStrict && p.status) {
        // per-wave peaks of the three activation vectors: non-negative floats order like their bit patterns, a NaN is a huge
        // pattern and reads as overflow.  One DPP reduction per value and a plain read-modify-write by one lane (the record is the
        // wave's own).  Round 5: this was three LDS atomicMax by all 64 lanes on ONE word each - 14 k clocks per tile, a tenth of
        // the kernel, and only in the product (the timing tool passed no status record: tools/k1h_ablate.hip K1H_STATUS=1).
        const bool v_ = valid || validb;
        const int m0 = wave_max_i32(v_ ? __float_as_int(amax) : 0), m1 = wave_max_i32(v_ ? __float_as_int(amax1) : 0),
                  m2 = wave_max_i32(v_ ? __float_as_int(amax2) : 0);
        if (lane == 0) { wrec[16] = max(wrec[16], m0); wrec[17] = max(wrec[17], m1); wrec[18] = max(wrec[18], m2); }
      }
      if (!SUB && p.bbox && p.mode != kPointList) {
        // with two point groups every lane folds ITS point: lanes 0..31 the first group's, lanes 32..63 the second's
        const long long pf_ = (G == 2 && half == 1) ? pib : pi;
        auto fold = [&](bool neg, int* rec, int extra) {
#if ASDF16_FOLD_BALLOT
          if ((PL == 1 || kMix) && !__any(neg || extra != 0)) return;      // (wave-uniform: most tiles of a sweep hold no negative voxel; round 5: the split-half SeparateDecoder forms too)
#endif
          int i0, i1, i2;                                        // (behind the early return: two integer divisions per lane)
          lattice_ijk(pf_, p.N, i0, i1, i2);
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
            rec[0] = min(rec[0], a0); rec[1] = min(rec[1], a1); rec[2] = min(rec[2], a2);
            rec[3] = max(rec[3], b0); rec[4] = max(rec[4], b1); rec[5] = max(rec[5], b2);
            rec[6] += n; rec[7] += extra;
          }
        };
        if (G == 2) fold(half == 0 ? (valid && sdf < p.neg_thr) : (validb && sdfg < p.neg_thr), wrec, bad);
        else fold(valid && half == 0 && sdf < p.neg_thr, wrec, bad);
        if (TWO_OUT) fold(valid && half == 0 && sdfb < p.neg_thr, wrec + 8, 0);
      } else {
        const unsigned long long m = __ballot(bad);
        if (m && lane == 0) wrec[7] += __popcll(m);
      }
      ASDF16_MARK(5);
#ifdef ASDF16_SEGMENT_TIMES
      if (blockIdx.x == 0 && wave == 0 && lane == 0 && slot == 0 && tile == (long long)blockIdx.x + 2 * gridDim.x)
        { for (int k = 0; k < 6; ++k) g_seg[k] = (unsigned long long)seg_t[k]; g_seg[7] = (unsigned long long)seg_t[7]; }
      if (blockIdx.x == 0 && wave == 0 && lane == 0 && slot == 0 && tile == (long long)blockIdx.x + 3 * gridDim.x) g_seg[6] = (unsigned long long)seg_t[0];
#endif
    }   // tiles

    if (lane == 0) {
      // one set of atomics per wave: record 0 = hand (MLP 0 / first row), record 1 = object (MLP 1 / second row);
      // word 7: lanes out of range (0 unless the fp16 planes overflowed; the host then falls back to fp32)
      auto flush = [&](int* out, const int* rec) {
        if (rec[6]) {
          atomicMin(out + 0, rec[0]); atomicMin(out + 1, rec[1]); atomicMin(out + 2, rec[2]);
          atomicMax(out + 3, rec[3]); atomicMax(out + 4, rec[4]); atomicMax(out + 5, rec[5]);
          atomicAdd(out + 6, rec[6]);
        }
      };
      if (p.bbox && p.mode != kPointList) {
        flush(p.bbox + (head == 0 ? 0 : 8), wrec);
        if (TWO_OUT) flush(p.bbox + 8, wrec + 8);
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


}  // namespace asdf
