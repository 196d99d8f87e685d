// K1n: the fp32 MFMA chain for SHORT voxel lists - the near-level refinement of every split-half sweep (a few dozen to a few hundred
// voxels), the candidates of a box-only coarse sweep - with the output tiles of a layer spread over the four waves of a workgroup.
//
// The tile form (sdf_mlp_kernel.h) gives each wave 32 points through ALL 16 output tiles of every layer: 8 256 dependent MFMAs,
// 0.23 ms, however few points the list holds - at N = 64 two such launches were a quarter of a sample.  Here a
// workgroup owns ONE block of 32 points of one MLP and its four waves take a quarter of the output tiles each (2 064 MFMAs per
// wave); between layers the activations cross through LDS in the accumulator's own register layout, [tile][lane][register], so
// that K-step s of the next layer reads, per lane, exactly the value the tile form holds in register s & 15 of tile s >> 4.  Every
// output is therefore accumulated by the SAME instruction sequence as in the tile form - bias, then the K-steps in stream order on
// v_mfma_f32_32x32x2_f32, the last layer's fmaf chain over tiles 0..15, registers 0..15, one cross-half add, tanhf - and the
// results are bit-identical (tests/test_gpu_short_list.py compares the two forms on the same lists).
//
// Weights: each A fragment is used by one wave once, so it goes straight from L2 into registers (the packed stream's stage image
// [group][lane][4] is one coalesced 1 KiB load per wave and 4 K-steps), two input tiles (32 K-steps) ahead.  kGridSubset only,
// affine point features (KP = 2); lists longer than p.short_max are left to the tile form, which skips the short ones.
//
// Round 5, the CLUSTER form of the same launch (lists of up to ShortParams::cluster_max points - the near-level list of a sweep and
// the candidates of a box sweep on the small lattices hold a few dozen to a few hundred): the four-wave form is bound by ONE compute
// unit's four matrix pipes, 2 064 dependent 64-clock MFMAs per wave = 0.1 ms per launch, two launches per sample, a sixth of a
// 64^3 sample.  A block of 32 points is therefore given to FOUR workgroups of the same XCD (16 waves): every wave owns one output
// tile of a layer - the unit that cannot be split without changing the order of its sum - so the longest chain is 256 + 130 + 256
// MFMAs.  Layer 0 is computed by every member (2 MFMAs per tile); the activations of layers 1..3 cross between the members
// through a per-cluster exchange buffer in device memory in the same [tile][lane][register] layout, published with an agent-scope
// release on an arrival counter and read behind an acquire (members run on different compute units: release = L2 write-back,
// acquire = invalidate, the memory model's own instructions).  The counters only ever grow: four arrivals per live cluster per
// launch, so between launches each is a multiple of four and a member waits for (value it saw & ~3) + 4 - no reset launch.
// Members of a cluster are 8 workgroup ids apart (ids are dealt round-robin over the 8 XCDs) inside one aligned group of 32 ids:
// workgroups are dispatched in id order, so the members a waiting workgroup spins on are resident or next in line whatever else
// holds compute units (the audit of a box sweep runs beside this kernel on another stream).  Results are bit-identical to both
// other forms: same per-tile instruction sequence, same last layer (tests/test_gpu_short_list.py).
// Round 6: a member that waits longer than ShortParams::timeout_ticks (1 s) no longer traps - it raises a sticky fault word, the
// cluster writes nothing, and the tile form enqueued behind the launch evaluates the list (see await below); the host switches the
// cluster form off for the decoder when it reads the word (HipSdfDecoder._note_cluster_fault).  The arrival counters and the exchange
// buffer belong to the decoder: a decoder is driven from ONE stream at a time (handles are not re-entrant).
#pragma once
#include "k1_launch.h"
#include "sdf_mlp_kernel.h"

namespace asdf {

constexpr int kShortBufFloats = kTilesHidden * 64 * 16;                       // one 512-wide activation of 32 points: 64 KiB
constexpr int kLdsBytesShort = (2 * kShortBufFloats + kCstFloats) * 4;
static_assert(kLdsBytesShort <= 160 * 1024, "LDS budget");

__device__ __forceinline__ void short_store16(float* buf, int t, int lane, const f32x16& a) {
  f32x4* d = reinterpret_cast<f32x4*>(buf + (t * 64 + lane) * 16);
#pragma unroll
  for (int c = 0; c < 4; ++c) { f32x4 v; v[0] = a[4 * c]; v[1] = a[4 * c + 1]; v[2] = a[4 * c + 2]; v[3] = a[4 * c + 3]; d[c] = v; }
}

// acc += sum over K-steps 0 .. 16 IN_TILES - 1, in order: A from the packed stream of one output tile (global), B from the
// previous layer's activations in LDS.  The A fragments run PF input tiles (16 PF K-steps) ahead; short_chain_begin issues the
// first PF of them - the cluster form does that BEFORE it waits for the other members' activations, so that the first weights
// (cold in this XCD's L2: the sweeps around this launch stream the fp16 images) arrive behind the wait, not in front of the MFMAs
template <int PF>
struct ShortAhead { f32x4 a[PF][4]; };

template <int IN_TILES, int PF>
__device__ __forceinline__ void short_chain_begin(ShortAhead<PF>& h, const float* __restrict__ wsrc, int lane) {
  const f32x4* w = reinterpret_cast<const f32x4*>(wsrc) + lane;               // group g of the tile's stream: w[g * 64]
#pragma unroll
  for (int i = 0; i < PF && i < IN_TILES; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) h.a[i][g] = w[(i * 4 + g) * 64];
}

template <int IN_TILES, int PF>
__device__ __forceinline__ void short_chain_run(f32x16& acc, const ShortAhead<PF>& h, const float* __restrict__ wsrc, const float* hin, int lane) {
  const f32x4* w = reinterpret_cast<const f32x4*>(wsrc) + lane;
  f32x4 abuf[IN_TILES + PF][4];
#pragma unroll
  for (int i = 0; i < PF && i < IN_TILES; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) abuf[i][g] = h.a[i][g];
#pragma unroll
  for (int it = 0; it < IN_TILES; ++it) {
    if (it + PF < IN_TILES) {
#pragma unroll
      for (int g = 0; g < 4; ++g) abuf[it + PF][g] = w[((it + PF) * 4 + g) * 64];
    }
    const f32x4* b4 = reinterpret_cast<const f32x4*>(hin + (it * 64 + lane) * 16);
    f32x4 b[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) b[c] = b4[c];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = ASDF_MFMA(abuf[it][g][j], b[g][j], acc);        // K-step 16 it + 4 g + j
  }
}

template <int IN_TILES>
__device__ __forceinline__ void short_chain(f32x16& acc, const float* __restrict__ wsrc, const float* hin, int lane) {
  ShortAhead<2> h;
  short_chain_begin<IN_TILES, 2>(h, wsrc, lane);
  short_chain_run<IN_TILES, 2>(acc, h, wsrc, hin, lane);
}

// layer 4 + tanh + what the tile form does with a kGridSubset result - one wave, the raw layer-3 accumulators of its 32 points in
// `bufY` ([tile][lane][16])
template <bool TWO_OUT>
__device__ __forceinline__ void short_last_layer(const DecodeParams& p, const float* hc, const float* bufY, int lane, int half, int head,
                                                 bool valid, long long po) {
  using CL = CstLayout<2>;
  // ---- layer 4 + tanh, in the tile form's order: tiles 0..15, registers 0..15, then the cross-half add
  float part = 0.0f, partb = 0.0f;
#pragma unroll 1
  for (int t = 0; t < kTilesHidden; ++t) {
    const f32x4* a4 = reinterpret_cast<const f32x4*>(bufY + (t * 64 + lane) * 16);
    const f32x4* w4 = reinterpret_cast<const f32x4*>(hc + CL::kW4 + (t * 2 + half) * 16);
    const f32x4* w4b = reinterpret_cast<const f32x4*>(hc + CL::kW4b + (t * 2 + half) * 16);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const f32x4 a = a4[c], w = w4[c];
      f32x4 wb = w;
      if (TWO_OUT) wb = w4b[c];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = __int_as_float(max(__float_as_int(a[r]), 0));
        part = fmaf(v, w[r], part);
        if (TWO_OUT) partb = fmaf(v, wb[r], partb);
      }
    }
  }
  part += __shfl_xor(part, 32);
  const float sdf = tanhf(part + hc[CL::kB4]);
  float sdfb = 1.0f;
  if (TWO_OUT) {
    partb += __shfl_xor(partb, 32);
    sdfb = tanhf(partb + hc[CL::kB4 + 1]);
  }
  const bool is_hand = head == 0;
  if (valid && half == 0) {
    // what the tile form does with a kGridSubset result (sdf_mlp_kernel.h): measured change, box patch, value in place
    float* out = is_hand ? p.sdf0 : p.sdf1;
    if (p.status && !p.bbox) {
      if (out) atomicMax(p.status + 3, __float_as_int(fabsf(sdf - out[po])));
      if (TWO_OUT && p.sdf1) atomicMax(p.status + 3, __float_as_int(fabsf(sdfb - p.sdf1[po])));
    }
    if (p.bbox) {
      auto patch = [&](float* vol, float now, int* rec) {
        const float before = vol[po];
        const bool was = before < p.neg_thr, is = now < 0.0f;
        if (p.status) atomicMax(p.status + 3, __float_as_int(fabsf(now - before)));
        if (was == is) return;
        if (is) {
          int i0, i1, i2;
              lattice_ijk(po, p.N, i0, i1, i2);
          atomicMin(rec + 0, i0); atomicMin(rec + 1, i1); atomicMin(rec + 2, i2);
          atomicMax(rec + 3, i0); atomicMax(rec + 4, i1); atomicMax(rec + 5, i2);
          atomicAdd(rec + 6, 1);
        } else {
          atomicExch(p.fixup_flag, 1);
        }
      };
      if (out) patch(out, sdf, p.bbox + (is_hand ? 0 : 8));
      if (TWO_OUT && p.sdf1) patch(p.sdf1, sdfb, p.bbox + 8);
    }
    if (out) out[po] = sdf;
    if (TWO_OUT && p.sdf1) p.sdf1[po] = sdfb;
  }
}

template <bool TWO_OUT>
__device__ __forceinline__ void sdf_mlp_short_body(const DecodeParams& p, const ShortParams& sp) {
  using CL = CstLayout<2>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* bufX = smem;                          // h0, then h2
  float* bufY = smem + kShortBufFloats;        // h1, then the raw layer-3 accumulators
  float* cst = smem + 2 * kShortBufFloats;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;

  long long npts = *p.count_dev;
  if (npts > p.P) npts = p.P;
  if (npts > (long long)p.short_max) return;                   // the tile form's list
  // one launch, two forms: the list length (a device word) decides.  Workgroup ids are one-dimensional; beyond the list they return
  const bool clustered = npts <= (long long)sp.cluster_max;
  const int wg = blockIdx.x;
  int cluster = 0, member = 0, unit = wg;                      // unit = (block of 32 points, MLP)
  if (clustered) {
    const int within = wg & 31;
    cluster = (wg >> 5) * 8 + (within & 7);                    // members 8 ids apart: the same XCD
    member = within >> 3;
    unit = cluster;
  }
  const int slot = unit % p.num_mlps;                          // (MLP innermost: the live units come first in dispatch order)
  const long long pb = (long long)(unit / p.num_mlps) * kWavePts;
  if (pb >= npts) return;
  const int head = p.first_mlp + slot;
  const float* sbase = p.stream + (size_t)head * kStagesHead * kStageFloats;
  constexpr int kAhead = 6;                                    // cluster form: input tiles of weights in flight per wave (6 KiB)
  ShortAhead<kAhead> ahead;
  const float* w1 = sbase + (size_t)((member * 2 + (wave & 1)) * 4) * kStageFloats;
  if (clustered && wave < 2) short_chain_begin<16, kAhead>(ahead, w1, lane);      // (layer 1's first weights travel with the constants)
  {
    const f32x4* src4 = reinterpret_cast<const f32x4*>(p.cst + (size_t)head * CL::kFloats);
    for (int i = tid; i < CL::kFloats / 4; i += 256) reinterpret_cast<f32x4*>(cst)[i] = src4[i];
  }
  __syncthreads();
  const float* hc = cst;
  const long long pi = pb + (lane & 31);
  const bool valid = pi < npts;
  const long long po = valid ? (long long)p.idx[pi] : 0;
  float x0, x1, x2;
  float lat_vs = p.vs, lat_o0 = p.o0, lat_o1 = p.o1, lat_o2 = p.o2;          // (the lattice by value, or from the words asdf_zoom_cube wrote)
  if (p.lattice) { lat_o0 = p.lattice[0]; lat_o1 = p.lattice[1]; lat_o2 = p.lattice[2]; lat_vs = p.lattice[3]; }
  grid_point(po, p.N, p.grid_mode, lat_vs, lat_o0, lat_o1, lat_o2, x0, x1, x2);
  const float bp0 = half ? x1 : x0, bp1 = half ? 0.0f : x2;

  // ---- layer 0: tiles 4 wave .. 4 wave + 3
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int t = wave * 4 + k;
    f32x16 acc = load_bias16(hc + CL::kC0 + (t * 2 + half) * 16);
    acc = ASDF_MFMA(hc[CL::kA0 + (t * 2 + 0) * 64 + lane], bp0, acc);
    acc = ASDF_MFMA(hc[CL::kA0 + (t * 2 + 1) * 64 + lane], bp1, acc);
    short_store16(bufX, t, lane, relu16i(acc));
  }
  __syncthreads();
  if (clustered) {
    // ---- the cluster form: one output tile per wave and layer, activations through the cluster's exchange buffer
    float* xb = sp.xchg + (size_t)cluster * kXchgFloats;
    float* x1 = xb, *x2 = xb + kTilesL1 * 1024, *x3 = xb + (kTilesL1 + kTilesHidden) * 1024;
    unsigned* arrivals = reinterpret_cast<unsigned*>(sp.arrivals) + cluster * 4;
    const int gw = member * kWaves + wave;                     // 0 .. 15
    // publish: this workgroup's tile stores become visible device-wide, then one arrival; await: all four have arrived.  (Between the
    // two a wave issues the first weights of its next tile: they travel while it waits.)
    unsigned all = 0;                                          // (unsigned: the counters wrap after 2^30 launches - days of 64^3 samples - and the comparison below survives that)
    __shared__ int s_gave_up;                                  // this workgroup has stopped waiting (see await)
    if (tid == 0) s_gave_up = 0;
    auto publish = [&](int which) {
      __threadfence();
      __syncthreads();
      if (tid == 0) {
        const unsigned seen = __hip_atomic_fetch_add(arrivals + which, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        all = (seen & ~(unsigned)(kClusterWgs - 1)) + kClusterWgs;
      }
    };
    auto await = [&](int which) {
      if (tid == 0 && !s_gave_up) {
        // The other members are resident or next in line (see the header), so this takes microseconds.  Forward progress of an
        // ordinary launch whose workgroups wait for each other is an ASSUMPTION (in-order dispatch, enough free workgroup slots: a CU
        // mask, a partitioned device or another process's persistent kernels can break it), so the wait is bounded and its failure is
        // RECOVERABLE (round 6; it used to trap, which takes the whole HIP context along): the member raises the decoder's sticky
        // fault word - BEFORE its remaining arrivals, which it still makes, so that the counters stay multiples of four and nobody
        // else waits for it - and stops waiting; whoever runs the last layer of a cluster sees the word behind its acquire and writes
        // nothing; the tile form enqueued behind this launch then evaluates the list (DecodeParams::short_fault), same bits.
        // (A bound of ONE tick is the test hook: the member gives up at its first wait WITHOUT looking at the counter - whether four
        // members that start together ever find each other missing is a matter of timing, and a test must not depend on that.)
        const bool forced = sp.timeout_ticks == 1;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (forced || (int)(__hip_atomic_load(arrivals + which, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - all) < 0) {
          if (!forced) __builtin_amdgcn_s_sleep(1);
          if (forced || __builtin_amdgcn_s_memrealtime() - t0 > sp.timeout_ticks) {
            __hip_atomic_store(sp.fault, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            s_gave_up = 1;
            break;
          }
        }
      }
      __syncthreads();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    };
    auto fetch = [&](float* dst, const float* src, int tiles) {
      const f32x4* s4 = reinterpret_cast<const f32x4*>(src);
      f32x4* d4 = reinterpret_cast<f32x4*>(dst);
      for (int i = tid; i < tiles * 256; i += 256) d4[i] = s4[i];
      __syncthreads();
    };
    const float* w2 = sbase + (size_t)(kStagesL1 + gw * 2) * kStageFloats;
    const float* w3 = sbase + (size_t)(kStagesL1 + kStagesL2 + gw * 4) * kStageFloats;
    if (wave < 2) {                                            // layer 1: 8 tiles, two per member
      const int t = member * 2 + wave;
      f32x16 acc = load_bias16(hc + CL::kB1 + (t * 2 + half) * 16);
      short_chain_run<16, kAhead>(acc, ahead, w1, bufX, lane);
      short_store16(x1, t, lane, relu16i(acc));
    }
    publish(0);
    short_chain_begin<8, kAhead>(ahead, w2, lane);
    await(0);
    fetch(bufY, x1, kTilesL1);
    {                                                          // layer 2: tile gw
      const int t = gw;
      f32x16 acc = load_bias16(hc + CL::kC2 + (t * 2 + half) * 16);
      acc = ASDF_MFMA(hc[CL::kA2 + (t * 2 + 0) * 64 + lane], bp0, acc);
      acc = ASDF_MFMA(hc[CL::kA2 + (t * 2 + 1) * 64 + lane], bp1, acc);
      short_chain_run<8, kAhead>(acc, ahead, w2, bufY, lane);
      short_store16(x2, t, lane, relu16i(acc));
    }
    publish(1);
    short_chain_begin<16, kAhead>(ahead, w3, lane);
    await(1);
    fetch(bufX, x2, kTilesHidden);
    {                                                          // layer 3: tile gw, raw accumulators
      const int t = gw;
      f32x16 acc = load_bias16(hc + CL::kB3 + (t * 2 + half) * 16);
      short_chain_run<16, kAhead>(acc, ahead, w3, bufX, lane);
      short_store16(x3, t, lane, acc);
    }
    publish(2);
    if (member != 0) return;                                   // the last layer is member 0's
    await(2);
    // a member of this cluster that gave up raised the fault word before its last arrival, and the await above acquired all four:
    // this block's activations may be stale - nothing is written (the word is sticky: until the host has switched the cluster form
    // off every cluster launch leaves its list to the tile form behind it)
    if (__hip_atomic_load(sp.fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
    fetch(bufY, x3, kTilesHidden);
    if (wave != 0) return;
    short_last_layer<TWO_OUT>(p, hc, bufY, lane, half, head, valid, po);
    return;
  }
  // ---- layer 1 (512 -> 256): tiles 2 wave, 2 wave + 1; four stages (K = 512) per tile
#pragma unroll 1
  for (int k = 0; k < 2; ++k) {
    const int t = wave * 2 + k;
    f32x16 acc = load_bias16(hc + CL::kB1 + (t * 2 + half) * 16);
    short_chain<16>(acc, sbase + (size_t)(t * 4) * kStageFloats, bufX, lane);
    short_store16(bufY, t, lane, relu16i(acc));
  }
  __syncthreads();
  // ---- layer 2 ([h1 (256) | xyz] -> 512): the point-feature K-steps first, then two stages per tile
#pragma unroll 1
  for (int k = 0; k < 4; ++k) {
    const int t = wave * 4 + k;
    f32x16 acc = load_bias16(hc + CL::kC2 + (t * 2 + half) * 16);
    acc = ASDF_MFMA(hc[CL::kA2 + (t * 2 + 0) * 64 + lane], bp0, acc);
    acc = ASDF_MFMA(hc[CL::kA2 + (t * 2 + 1) * 64 + lane], bp1, acc);
    short_chain<8>(acc, sbase + (size_t)(kStagesL1 + t * 2) * kStageFloats, bufY, lane);
    short_store16(bufX, t, lane, relu16i(acc));
  }
  __syncthreads();
  // ---- layer 3 (512 -> 512): raw accumulators to LDS; the last layer applies the ReLU as it reads them
#pragma unroll 1
  for (int k = 0; k < 4; ++k) {
    const int t = wave * 4 + k;
    f32x16 acc = load_bias16(hc + CL::kB3 + (t * 2 + half) * 16);
    short_chain<16>(acc, sbase + (size_t)(kStagesL1 + kStagesL2 + t * 4) * kStageFloats, bufX, lane);
    short_store16(bufY, t, lane, acc);
  }
  __syncthreads();
  if (wave != 0) return;
  short_last_layer<TWO_OUT>(p, hc, bufY, lane, half, head, valid, po);
}

}  // namespace asdf
