// Host side of the decoder C ABI (include/alignsdf_hip.h): weight packing, the per-sample fold
// kernel (K0) and the launches of the fused MLP kernel (K1, sdf_mlp_kernel.h).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/alignsdf_hip.h"
#include "common.h"
#include "pack.h"
#include "k1_launch.h"

namespace asdf {

thread_local int g_last_hip_error = 0;

// ---------------------------------------------------------------------------------------------
// K0: fold the per-sample constants.  One wave per (head, layer in {0, 2}, output row):
//   c[o]    = b[o] + W_lat[o,:] . latent + W_pt[o,:] . E[:,3]
//   A[o][d] = W_pt[o,:] . E[:,d]            d = 0..2   (column 3 of the K = 4 operand is zero)
// and scatter them into the LDS constants image (sdf_layout.h).
// Reference arithmetic being folded: utils/utils.py:568-569 (latent expand + cat),
// networks/model.py:311-312 (layer-2 skip concat) and utils/utils.py:376-430 (kinematic embedding).
// ---------------------------------------------------------------------------------------------
struct FoldParams {
  const float* wlat;     // [heads][2][512][256]
  const float* wpt;      // [heads][2][512][ASDF_MAX_POINT_FEATS]
  const float* bias02;   // [heads][2][512]
  const float* embed;    // [heads][ASDF_MAX_POINT_FEATS][4]
  const float* latent;   // [256]
  // up to three constants images are folded by ONE launch (blockIdx.y picks the image): the fp32 image, the split-half image and
  // the one-plane image differ in the scales only (they were three launches per sample)
  float* cst[3];         // [heads][cst_offsets(kp).floats]
  int pf[ASDF_MAX_HEADS];
  int kp;                // point-feature K-steps (2 = affine xyz: the A fragments are folded here; > 2 = NeRF: static)
  float s2[3][ASDF_MAX_HEADS];   // scale of the layer-2 constants: 1 for the fp32 image, S_w2 S_x for the split-half image
  float s0[3][ASDF_MAX_HEADS];   // scale of the layer-0 constants: 1, except in the one-plane image (S_x of h0: its accumulators need no rescale)
};

__global__ __launch_bounds__(256) void fold_sample_kernel(const FoldParams p) {
  const int lane = threadIdx.x & 63;
  const int job = blockIdx.x * 4 + (threadIdx.x >> 6);   // (head, layer, row)
  const int row = job & (kHidden - 1);
  const int layer = (job >> 9) & 1;
  const int head = job >> 10;
  if (head >= kHeads) return;

  const float* wl = p.wlat + ((size_t)(head * 2 + layer) * kHidden + row) * kLatent;
  float dot = 0.0f;
#pragma unroll
  for (int k = 0; k < kLatent / 64; ++k) dot = fmaf(wl[lane + 64 * k], p.latent[lane + 64 * k], dot);
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) dot += __shfl_xor(dot, m);

  const float* wp = p.wpt + ((size_t)(head * 2 + layer) * kHidden + row) * ASDF_MAX_POINT_FEATS;
  const float* E = p.embed + (size_t)head * ASDF_MAX_POINT_FEATS * 4;
  float a = 0.0f;
  if (lane < 4)
    for (int f = 0; f < p.pf[head]; ++f) a = fmaf(wp[f], E[f * 4 + lane], a);
  const float a3 = __shfl(a, 3);

  const CstOffsets co = cst_offsets(p.kp);
  const int img = blockIdx.y;
  float* cst = p.cst[img] + (size_t)head * co.floats;
  const int t = row >> 5, rr = row & 31;
  const float sc = layer ? p.s2[img][head] : p.s0[img][head];      // a power of two: the scaled values are exact
  if (lane == 0) {
    const float c = (dot + p.bias02[(head * 2 + layer) * kHidden + row]) + (p.kp == 2 ? a3 : 0.0f);
    const int hh = (rr >> 2) & 1, r = (rr & 3) + 4 * (rr >> 3);
    cst[(layer ? co.c2 : co.c0) + (t * 2 + hh) * 16 + r] = c * sc;
  }
  if (lane < 4 && p.kp == 2) {
    const int step = lane >> 1, hh = lane & 1;
    cst[(layer ? co.a2 : co.a0) + (t * 2 + step) * 64 + hh * 32 + rr] = (lane < 3) ? a * sc : 0.0f;
  }
}

// K0b: the point-feature columns and bias rows of layers 0 / 2 of the split-half constants image -> fp16 A operands of the
// one-plane kernels (sdf_layout.h: kA16Floats).  One workgroup per (head, layer), one thread per output row: the largest
// magnitude of the layer's 4 x 512 values picks the power of two T = 2^e (0 <= e <= 15) that brings them under 2^14; every
// value is stored as the two fp16 planes of v / T and the kernel multiplies the point operand by T.  More than 2^29 cannot be
// carried: counted in status[0] like any other fp16 range violation (the sweep is then refused and repeated as an ordinary one).
__global__ __launch_bounds__(512) void fold_points_f16_kernel(const float* __restrict__ cst_all, float* __restrict__ a16_all, int* status) {
  const int head = blockIdx.x >> 1, layer = blockIdx.x & 1;
  const CstOffsets co = cst_offsets(2);
  const float* cst = cst_all + (size_t)head * co.floats;
  float* a16 = a16_all + (size_t)head * kA16Floats;
  const int row = threadIdx.x, t = row >> 5, rr = row & 31;
  const int abase = layer ? co.a2 : co.a0, cbase = layer ? co.c2 : co.c0;
  float v[4];
  v[0] = cst[abase + (t * 2 + 0) * 64 + 0 * 32 + rr];      // feature d = 2 step + lane half
  v[1] = cst[abase + (t * 2 + 0) * 64 + 1 * 32 + rr];
  v[2] = cst[abase + (t * 2 + 1) * 64 + 0 * 32 + rr];
  v[3] = cst[cbase + (t * 2 + ((rr >> 2) & 1)) * 16 + (rr & 3) + 4 * (rr >> 3)];
  float m = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
  if (!(m == m)) m = INFINITY;
#pragma unroll
  for (int k = 32; k >= 1; k >>= 1) m = fmaxf(m, __shfl_xor(m, k));
  __shared__ float wmax[8];
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  m = wmax[0];
#pragma unroll
  for (int k = 1; k < 8; ++k) m = fmaxf(m, wmax[k]);
  int e = 0;
  if (m > 16384.0f) { int ex; (void)frexpf(m, &ex); e = ex - 14; }       // m in [2^(ex-1), 2^ex): m / 2^e < 2^14
  if (e > 15 || !(m < INFINITY)) { e = 15; if (threadIdx.x == 0 && status) atomicAdd(status, 1); }
  const float T = ldexpf(1.0f, e), inv = ldexpf(1.0f, -e);
  _Float16 hi[4], lo[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) { const float sv = v[k] * inv; hi[k] = (_Float16)sv; lo[k] = (_Float16)(sv - (float)hi[k]); }
  typedef _Float16 h8v __attribute__((ext_vector_type(8)));
  h8v a, b;
  a[0] = hi[0]; a[1] = hi[1]; a[2] = hi[2]; a[3] = hi[0]; a[4] = hi[1]; a[5] = hi[2]; a[6] = hi[3]; a[7] = lo[3];
  b[0] = lo[0]; b[1] = lo[1]; b[2] = lo[2]; b[3] = b[4] = b[5] = b[6] = b[7] = (_Float16)0.0f;
  h8v* out = reinterpret_cast<h8v*>(a16 + layer * kA16LayerFloats + t * kA16TileFloats);
  out[rr] = a;              // lane rr:      lane half 0
  out[32 + rr] = b;         // lane 32 + rr: lane half 1
  if (threadIdx.x == 0) a16[2 * kA16LayerFloats + layer] = T;
}

__global__ void bbox_init_kernel(int* bbox) {   // 16 ints: two records of {min[3], max[3], count, pad}
  const int i = threadIdx.x;
  if (i < 16) {
    const int j = i & 7;
    bbox[i] = j < 3 ? 0x7fffffff : (j < 6 ? -1 : 0);
  }
}

// Start of a one-plane sweep: the box record plus every small counter / flag word the sweep's kernels accumulate into, in ONE
// launch (they were a bbox_init launch and four memsets: seven launches per sample that the small lattices notice).
struct ClearRange { int* p; int n; };
__global__ void sweep_init_kernel(int* bbox, ClearRange a, ClearRange b, ClearRange c, ClearRange e) {
  const int i = threadIdx.x;
  if (i < 16) {
    const int j = i & 7;
    bbox[i] = j < 3 ? 0x7fffffff : (j < 6 ? -1 : 0);
  }
  if (i < a.n) a.p[i] = 0;
  if (i < b.n) b.p[i] = 0;
  if (i < c.n) c.p[i] = 0;
  if (i < e.n) e.p[i] = 0;
}

// K2 (standalone form; K1 fuses the same reduction into its epilogue): bounding box of the voxels with
// sdf < 0 - torch.nonzero + per-axis min/max of get_higher_res_cube (utils/mesh.py:208-237).
// One wave per (i0, i1) row: the row index is wave-uniform (no per-voxel division), lanes stride over axis 2 with float4
// loads when the row length allows; HBM-bound streaming read (4 n bytes).  `flag`: when non-null the kernel is a no-op
// unless *flag != 0 (the conditional recount behind the near-level refinement).
__global__ __launch_bounds__(256) void neg_bbox_kernel(const float* __restrict__ vol, int n0, int n1, int n2, int* bbox, const int* flag) {
  if (flag && *flag == 0) return;
  const int lane = threadIdx.x & 63;
  const long long rows = (long long)n0 * n1;
  const long long wave0 = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * (blockDim.x >> 6);
  int lo0 = 0x7fffffff, lo1 = 0x7fffffff, lo2 = 0x7fffffff, hi0 = -1, hi1 = -1, hi2 = -1, cnt = 0;
  const bool vec = (n2 & 3) == 0 && ((size_t)vol & 15) == 0;
  for (long long r = wave0; r < rows; r += nwaves) {
    const float* row = vol + r * n2;
    int rlo = 0x7fffffff, rhi = -1, rc = 0;
    if (vec) {
      for (int x = lane * 4; x < n2; x += 256) {
        const float4 q = *reinterpret_cast<const float4*>(row + x);
        if (q.x < 0.0f) { rlo = min(rlo, x); rhi = max(rhi, x); ++rc; }
        if (q.y < 0.0f) { rlo = min(rlo, x + 1); rhi = max(rhi, x + 1); ++rc; }
        if (q.z < 0.0f) { rlo = min(rlo, x + 2); rhi = max(rhi, x + 2); ++rc; }
        if (q.w < 0.0f) { rlo = min(rlo, x + 3); rhi = max(rhi, x + 3); ++rc; }
      }
    } else {
      for (int x = lane; x < n2; x += 64)
        if (row[x] < 0.0f) { rlo = min(rlo, x); rhi = max(rhi, x); ++rc; }
    }
    if (rc) {
      const int i0 = (int)(r / n1), i1 = (int)(r % n1);
      lo0 = min(lo0, i0); hi0 = max(hi0, i0); lo1 = min(lo1, i1); hi1 = max(hi1, i1);
      lo2 = min(lo2, rlo); hi2 = max(hi2, rhi); cnt += rc;
    }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    lo0 = min(lo0, __shfl_xor(lo0, m)); lo1 = min(lo1, __shfl_xor(lo1, m)); lo2 = min(lo2, __shfl_xor(lo2, m));
    hi0 = max(hi0, __shfl_xor(hi0, m)); hi1 = max(hi1, __shfl_xor(hi1, m)); hi2 = max(hi2, __shfl_xor(hi2, m));
    cnt += __shfl_xor(cnt, m);
  }
  // one set of atomics per WORKGROUP (a set per wave on seven hot words serialised 16 k waves: 0.8 ms)
  __shared__ int s_rec[4][7];
  const int w = threadIdx.x >> 6;
  if (lane == 0) { s_rec[w][0] = lo0; s_rec[w][1] = lo1; s_rec[w][2] = lo2; s_rec[w][3] = hi0; s_rec[w][4] = hi1; s_rec[w][5] = hi2; s_rec[w][6] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 4; ++k) {
      lo0 = min(lo0, s_rec[k][0]); lo1 = min(lo1, s_rec[k][1]); lo2 = min(lo2, s_rec[k][2]);
      hi0 = max(hi0, s_rec[k][3]); hi1 = max(hi1, s_rec[k][4]); hi2 = max(hi2, s_rec[k][5]); cnt += s_rec[k][6];
    }
    if (cnt) {
      atomicMin(bbox + 0, lo0); atomicMin(bbox + 1, lo1); atomicMin(bbox + 2, lo2);
      atomicMax(bbox + 3, hi0); atomicMax(bbox + 4, hi1); atomicMax(bbox + 5, hi2);
      atomicAdd(bbox + 6, cnt);
    }
  }
}

// Voxels whose value lies within tau of the iso level, of one or two volumes swept on the same lattice: their linear
// indices are appended to idx (order arbitrary), *count counts all of them (also those beyond cap: status[1] then counts
// the voxels that could not be listed).  float4 path when n is a multiple of 4.
__global__ __launch_bounds__(256) void collect_near_level_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n,
                                                                 float tau, int* idx, int* count, int cap, int* status) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long t0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  auto hit = [&](long long i) {
    const int k = atomicAdd(count, 1);
    if (k < cap) idx[k] = (int)i;
    else if (status) atomicAdd(status + 1, 1);
  };
  if ((n & 3) == 0) {
    for (long long q = t0; q < n / 4; q += stride) {
      float4 va = a ? reinterpret_cast<const float4*>(a)[q] : make_float4(1.f, 1.f, 1.f, 1.f);
      float4 vb = b ? reinterpret_cast<const float4*>(b)[q] : make_float4(1.f, 1.f, 1.f, 1.f);
      if (fabsf(va.x) < tau || fabsf(vb.x) < tau) hit(4 * q + 0);
      if (fabsf(va.y) < tau || fabsf(vb.y) < tau) hit(4 * q + 1);
      if (fabsf(va.z) < tau || fabsf(vb.z) < tau) hit(4 * q + 2);
      if (fabsf(va.w) < tau || fabsf(vb.w) < tau) hit(4 * q + 3);
    }
  } else {
    for (long long i = t0; i < n; i += stride)
      if ((a && fabsf(a[i]) < tau) || (b && fabsf(b[i]) < tau)) hit(i);
  }
}

// The candidates of the box-only sweep: voxels whose one-plane value v is not decided by the error bound (-tau <= v < tau)
// for a head, and which lie outside that head's box of certainly negative voxels (v < -tau) - only those can move the box.
// Both heads of a listed voxel are re-evaluated exactly.
__global__ __launch_bounds__(256) void collect_box_candidates_kernel(const float* __restrict__ a, const float* __restrict__ b, int N, float tau,
                                                                     const int* __restrict__ bbox, int* idx, int* count, int cap, int* status) {
  const long long n = (long long)N * N * N;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long t0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  int box[2][7];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int k = 0; k < 7; ++k) box[h][k] = bbox[8 * h + k];
  auto outside = [&](int h, int i0, int i1, int i2) {
    return box[h][6] == 0 || i0 < box[h][0] || i0 > box[h][3] || i1 < box[h][1] || i1 > box[h][4] || i2 < box[h][2] || i2 > box[h][5];
  };
  auto open = [&](float v) { return v >= -tau && v < tau; };
  // The list is gathered per workgroup in LDS and handed over with ONE reservation per 8 steps: a pose-aligned decoder lists up
  // to 1e6 voxels, and one atomic per wave and step on the single count word was 0.3 ms of same-address atomics.
  constexpr int kSteps = 8, kLocal = kSteps * 256 * 4;
  __shared__ int s_list[kLocal];
  __shared__ int s_n, s_base;
  const int lane = threadIdx.x & 63;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  auto append = [&](bool h, long long i) {
    const unsigned long long m = __ballot(h);
    if (!m) return;
    const int leader = __ffsll((long long)m) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(&s_n, __popcll(m));
    base = __shfl(base, leader);
    if (h) s_list[base + __popcll(m & ((1ull << lane) - 1ull))] = (int)i;      // (at most kLocal voxels between two flushes)
  };
  auto flush = [&]() {
    __syncthreads();
    const int m = s_n;
    if (threadIdx.x == 0 && m > 0) s_base = atomicAdd(count, m);
    __syncthreads();
    for (int j = threadIdx.x; j < m; j += blockDim.x) {
      const int k = s_base + j;
      if (k < cap) idx[k] = s_list[j];
      else if (status) atomicAdd(status + 1, 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
  };
  const bool vec = (N & 3) == 0;
  const long long items = vec ? n / 4 : n;
  const long long rounds = (items + stride - 1) / stride;              // (every thread takes every step: appends and flushes are collective)
  for (long long r = 0; r < rounds; ++r) {
    const long long q = t0 + r * stride;
    const bool live = q < items;
    const long long i = vec ? 4 * q : q;
    const int i2 = (int)(i % N), i1 = (int)((i / N) % N), i0 = (int)((i / N) / N);
    float va[4] = {1.f, 1.f, 1.f, 1.f}, vb[4] = {1.f, 1.f, 1.f, 1.f};
    if (live) {
      if (vec) {
        if (a) { const float4 t = reinterpret_cast<const float4*>(a)[q]; va[0] = t.x; va[1] = t.y; va[2] = t.z; va[3] = t.w; }
        if (b) { const float4 t = reinterpret_cast<const float4*>(b)[q]; vb[0] = t.x; vb[1] = t.y; vb[2] = t.z; vb[3] = t.w; }
      } else {
        if (a) va[0] = a[i];
        if (b) vb[0] = b[i];
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (!vec && k > 0) break;
      append(live && ((a && open(va[k]) && outside(0, i0, i1, i2 + k)) || (b && open(vb[k]) && outside(1, i0, i1, i2 + k))), i + k);
    }
    if ((r + 1) % kSteps == 0) flush();
  }
  flush();
}

// The candidate count decides how the candidates are re-evaluated (the host cannot know it without a wait, so both forms are
// enqueued and the one whose count word is zero returns at once): up to `direct` voxels straight on the fp32 chain - one round of
// 128-point tiles over the CUs is the latency floor of that chain anyway; more than that (pose-aligned decoders list up to 1e6)
// through the split-half kernel first, 3 x the fp32 chain's rate, and the fp32 chain only where those values lie within refine_tau
// of the level - the rule of every split-half sweep.  out[0] / out[1]: the count as the direct / the two-step form sees it,
// out[2] = 0: the near-level count of the two-step form.
__global__ void split_candidate_count_kernel(const int* __restrict__ count, int cap, int direct, int* out) {
  if (threadIdx.x == 0) {
    int c = *count;
    if (c > cap) c = cap;
    out[0] = c <= direct ? c : 0;
    out[1] = c > direct ? c : 0;
    out[2] = 0;
  }
}

// ... and the two-step form's last step: every listed voxel whose exact value is negative extends its head's box (a voxel the
// sweep kernel had counted already changes nothing: min / max; words 6 / 14 only have to be non-zero iff there is a negative voxel)
__global__ __launch_bounds__(256) void extend_box_from_list_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                   const int* __restrict__ list, const int* __restrict__ count, int N, int* bbox) {
  const int n = *count;
  const float* vols[2] = {a, b};
  int lo[2][3], hi[2][3], cnt[2] = {0, 0};
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int k = 0; k < 3; ++k) { lo[h][k] = 0x7fffffff; hi[h][k] = -1; }
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int po = list[i];
    const int i2 = po % N, i1 = (po / N) % N, i0 = (po / N) / N;
#pragma unroll
    for (int h = 0; h < 2; ++h)
      if (vols[h] && vols[h][po] < 0.0f) {
        lo[h][0] = min(lo[h][0], i0); lo[h][1] = min(lo[h][1], i1); lo[h][2] = min(lo[h][2], i2);
        hi[h][0] = max(hi[h][0], i0); hi[h][1] = max(hi[h][1], i1); hi[h][2] = max(hi[h][2], i2);
        ++cnt[h];
      }
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (!vols[h]) continue;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        lo[h][k] = min(lo[h][k], __shfl_xor(lo[h][k], off));
        hi[h][k] = max(hi[h][k], __shfl_xor(hi[h][k], off));
      }
      cnt[h] += __shfl_xor(cnt[h], off);
    }
    if ((threadIdx.x & 63) == 0 && cnt[h] > 0) {
      int* rec = bbox + 8 * h;
      atomicMin(rec + 0, lo[h][0]); atomicMin(rec + 1, lo[h][1]); atomicMin(rec + 2, lo[h][2]);
      atomicMax(rec + 3, hi[h][0]); atomicMax(rec + 4, hi[h][1]); atomicMax(rec + 5, hi[h][2]);
      atomicAdd(rec + 6, cnt[h]);
    }
  }
}

// The narrow-band fine sweep (asdf_decode_grid_band).  Marching cubes reads a cell's eight corner values only if the cell is
// active, and otherwise only their signs.  With one-plane values v known to lie within tau of the exact ones, a cell CAN be
// active only if its corners' one-plane signs are mixed or one of them is undecided (|v| < tau): for those cells all eight
// corners are marked for exact re-evaluation; every other cell is inactive whatever the error, and its corners are never read.
// One thread per 4 x-consecutive cells, like mc_classify: 4 corner rows of 5 values.
// (round 5: z from blockIdx.y, (y, group) from a 32-bit index - no 64-bit divisions - and the four corner rows as one 16-byte load
// plus one word each where the row length allows: 80 -> about 30 us per 256^3 volume; same marks)
__global__ __launch_bounds__(256) void band_mark_kernel(const float* __restrict__ vol, int N, float tau, unsigned char* __restrict__ mark) {
  const int cx = N - 1, groups = (cx + 3) >> 2;
  const int per_slab = groups * cx;                   // (y, group) pairs of one z
  const bool vec = (N & 3) == 0;                      // rows start 16-byte aligned and x0 is a multiple of 4
  for (int z = blockIdx.y; z < cx; z += gridDim.y) {
    for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < per_slab; g += gridDim.x * blockDim.x) {
      const int y = g / groups, x0 = (g - y * groups) * 4;
      const float* r00 = vol + ((size_t)z * N + y) * N;
      const float* rows[4] = {r00, r00 + N, r00 + (size_t)N * N, r00 + (size_t)N * N + N};
      unsigned pos[4], neg[4];           // bit i: value i of the row is certainly positive / certainly negative
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v[5];
        if (vec) {
          const float4 t = *reinterpret_cast<const float4*>(rows[q] + x0);
          v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
          v[4] = x0 + 4 < N ? rows[q][x0 + 4] : 0.0f;
        } else {
#pragma unroll
          for (int i = 0; i < 5; ++i) v[i] = x0 + i < N ? rows[q][x0 + i] : 0.0f;
        }
        pos[q] = neg[q] = 0;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
          if (x0 + i < N) {
            pos[q] |= (v[i] >= tau ? 1u : 0u) << i;
            neg[q] |= (v[i] < -tau ? 1u : 0u) << i;
          } else {
            pos[q] |= 1u << i; neg[q] |= 1u << i;      // beyond the row: neutral for the AND reductions below, cells there are masked off
          }
        }
      }
      // a cell is certainly inactive iff all 8 corners are certainly positive, or all certainly negative
      const unsigned ap = pos[0] & pos[1] & pos[2] & pos[3], an = neg[0] & neg[1] & neg[2] & neg[3];
      const int ncell = min(4, cx - x0);
      const unsigned inactive = (ap & (ap >> 1)) | (an & (an >> 1));
      const unsigned cand = ~inactive & ((1u << ncell) - 1);
      if (!cand) continue;
      // corners of the candidate cells: columns x0 + i and x0 + i + 1 of the four rows
      const unsigned cols = (cand | (cand << 1)) & 0x1fu;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        unsigned char* m = mark + (rows[q] - vol);
#pragma unroll
        for (int i = 0; i < 5; ++i)
          if ((cols >> i) & 1) m[x0 + i] = 1;
      }
    }
  }
}

// marked voxels -> index list (order arbitrary); *count counts all of them, also those beyond cap.  One reservation on the count
// word per WORKGROUP and round (wave scans + an LDS hand-over): half a million marked voxels used to be ~1e5 same-address atomics.
// `done` / `count_copy` (optional): the last workgroup to finish copies the final count to *count_copy - the first audit position of the
// list (the audit picks are appended behind the marked voxels); *done must be zero at launch (sweep_init_kernel).  Was a separate
// 4-byte device-to-device copy per head.
__global__ __launch_bounds__(256) void band_compact_kernel(const unsigned char* __restrict__ mark, long long n, int* idx, int* count, int cap,
                                                           int* done, int* count_copy) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long items = (n + 15) / 16;
  const long long rounds = (items + stride - 1) / stride;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ int s_wave[4], s_base;
  for (long long r = 0; r < rounds; ++r) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x + r * stride;
    unsigned ww[4] = {0u, 0u, 0u, 0u};
    if (q < items) {
      if (q * 16 + 16 <= n) {
        const uint4 w = reinterpret_cast<const uint4*>(mark)[q];
        ww[0] = w.x; ww[1] = w.y; ww[2] = w.z; ww[3] = w.w;
      } else {
        for (long long i = q * 16; i < n; ++i)
          if (mark[i]) ww[(i - q * 16) >> 2] |= 1u << (8 * ((i - q * 16) & 3));
      }
    }
    int c = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) c += __popc(ww[k] & 0x01010101u);
    // exclusive position of this thread's entries inside the workgroup's reservation
    int incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d); if (lane >= d) incl += t; }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
      const int total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
      s_base = total ? atomicAdd(count, total) : 0;
    }
    __syncthreads();
    int at = s_base + (incl - c);
    for (int w = 0; w < wave; ++w) at += s_wave[w];
    if (c) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          if ((ww[k] >> (8 * b)) & 1) { if (at < cap) idx[at] = (int)(q * 16 + 4 * k + b); ++at; }
    }
    __syncthreads();
  }
  // (no fence: thread 0 issued this workgroup's reservations on *count itself and has their return values - they are complete -
  // before it arrives here, and device-scope atomics are coherent across the XCDs.  A __threadfence() per workgroup is an L2
  // write-back each: 19 -> 131 us per launch at N = 256, seen in the kernel statistics)
  if (done && threadIdx.x == 0 && atomicAdd(done, 1) == (int)gridDim.x - 1) *count_copy = atomicAdd(count, 0);
}

// near-level voxels among the LISTED ones of one volume (the narrow-band sweep refines only where it re-evaluated)
__global__ __launch_bounds__(256) void collect_near_level_list_kernel(const float* __restrict__ vol, const int* __restrict__ list,
                                                                      const int* __restrict__ list_count, int list_cap, float tau, int* idx,
                                                                      int* count, int cap, int* status) {
  const int n = *list_count < list_cap ? *list_count : list_cap;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int v = list[i];
    if (fabsf(vol[v]) < tau) {
      const int k = atomicAdd(count, 1);
      if (k < cap) idx[k] = v;
      else if (status) atomicAdd(status + 1, 1);
    }
  }
}

// The audit of a one-plane sweep: n voxels drawn uniformly at random (splitmix64 of seed + k, with replacement) from those the
// sweep DECIDED BY SIGN ALONE - not marked for re-evaluation (band sweep: mark[v] == 0, which implies |value| >= tau), or
// outside [-tau, tau) for every evaluated head (box sweep) - are appended to a voxel list; the caller re-evaluates them with
// the split-half kernel, which reports the largest |exact - one-plane| over them and the number whose sign was wrong.
// One reservation per wave.
__device__ __forceinline__ unsigned long long splitmix64_dev(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__global__ __launch_bounds__(256) void audit_pick_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         const unsigned char* __restrict__ mark, long long P, float tau,
                                                         unsigned long long seed, int n, int* list, int* count, int cap);

// The AT-RISK SHELL of a one-plane sweep (round 4): of the voxels a sweep decides by sign alone, only those whose one-plane value
// lies close to the decision threshold can be decided wrongly by an error of the allowance's order - tau <= |v| < 2 tau.  A uniform
// draw spends a fraction of a percent of its picks there; half of the audit is therefore drawn FROM the shell: it is counted
// (shell_count_kernel) and then listed - every shell voxel while the shell is smaller than the budget (an exhaustive check), a
// hash-thinned uniform subset of expected size `budget` otherwise (shell_pick_kernel).  Band sweep: unmarked voxels of one head's
// volume (unmarked implies |v| >= tau); box sweep: voxels every evaluated head leaves outside [-tau, tau), some head inside 2 tau.
__device__ __forceinline__ bool in_audit_shell(const float* __restrict__ a, const float* __restrict__ b,
                                               const unsigned char* __restrict__ mark, long long v, float tau) {
  if (mark) return mark[v] == 0 && (fabsf(a[v]) < 2.0f * tau || (b && fabsf(b[v]) < 2.0f * tau));      // (b: a CombinedDecoder's second column)
  bool decided = true, close = false;
  if (a) { const float t = a[v]; decided = decided && !(t >= -tau && t < tau); close = close || fabsf(t) < 2.0f * tau; }
  if (b) { const float t = b[v]; decided = decided && !(t >= -tau && t < tau); close = close || fabsf(t) < 2.0f * tau; }
  return decided && close;
}
__device__ __forceinline__ void audit_pick_block(int k, const float* __restrict__ a, const float* __restrict__ b, const unsigned char* __restrict__ mark,
                                                 long long P, float tau, unsigned long long seed, int n, int* list, int* count, int cap) {
  bool ok = k < n;
  long long v = 0;
  if (ok) {
    v = (long long)(splitmix64_dev(seed + (unsigned long long)k * 0xD1342543DE82EF95ull) % (unsigned long long)P);
    if (mark) ok = mark[v] == 0;
    else {
      if (a) { const float t = a[v]; ok = ok && !(t >= -tau && t < tau); }
      if (b) { const float t = b[v]; ok = ok && !(t >= -tau && t < tau); }
    }
  }
  const unsigned long long m = __ballot(ok);
  if (!m) return;
  const int lane = threadIdx.x & 63;
  int base = 0;
  if (lane == 0) base = atomicAdd(count, __popcll(m));
  base = __shfl(base, 0);
  const int at = base + __popcll(m & ((1ull << lane) - 1));
  if (ok && at < cap) list[at] = (int)v;
}
__global__ __launch_bounds__(256) void audit_pick_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         const unsigned char* __restrict__ mark, long long P, float tau,
                                                         unsigned long long seed, int n, int* list, int* count, int cap) {
  audit_pick_block(blockIdx.x * blockDim.x + threadIdx.x, a, b, mark, P, tau, seed, n, list, count, cap);
}
// workgroups 0 .. shell_blocks - 1 count the shell; the ones behind them draw the `uniform_n` uniform picks (audit_pick_kernel's draw:
// the two were separate launches - the small lattices notice every launch)
__global__ __launch_bounds__(256) void shell_count_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          const unsigned char* __restrict__ mark, long long P, float tau, int* shell_n,
                                                          int shell_blocks, unsigned long long seed, int uniform_n, int* list, int* count, int cap) {
  if ((int)blockIdx.x >= shell_blocks) {
    audit_pick_block(((int)blockIdx.x - shell_blocks) * blockDim.x + threadIdx.x, a, b, mark, P, tau, seed, uniform_n, list, count, cap);
    return;
  }
  const long long stride = (long long)shell_blocks * blockDim.x;
  int c = 0;
  for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < P; v += stride) c += in_audit_shell(a, b, mark, v, tau) ? 1 : 0;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) c += __shfl_xor(c, m);
  __shared__ int s_c[4];
  if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) { const int t = s_c[0] + s_c[1] + s_c[2] + s_c[3]; if (t) atomicAdd(shell_n, t); }
}
__global__ __launch_bounds__(256) void shell_pick_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         const unsigned char* __restrict__ mark, long long P, float tau,
                                                         unsigned long long seed, int budget, const int* __restrict__ shell_n,
                                                         int* list, int* count, int cap, int* audit_rec) {
  const int population = *shell_n;
  // keep every shell voxel while they fit the budget, else each with probability budget / population (a hash of the voxel index)
  const unsigned long long thr = population <= budget ? (1ull << 32) : (unsigned long long)(((double)budget / (double)population) * 4294967296.0);
  const long long stride = (long long)gridDim.x * blockDim.x;
  const int lane = threadIdx.x & 63;
  const long long rounds = (P + stride - 1) / stride;
  // the picks of a workgroup are gathered in LDS and handed over with ONE reservation on the list's count word (32 k picks were
  // 32 k same-address atomics: 0.19 ms per launch); what does not fit the LDS list goes straight to the global one
  constexpr int kLocal = 2048;
  __shared__ int s_list[kLocal];
  __shared__ int s_n, s_base;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  int kept = 0;
  for (long long r = 0; r < rounds; ++r) {
    const long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x + r * stride;
    const bool ok = v < P && in_audit_shell(a, b, mark, v, tau) &&
                    (splitmix64_dev(seed ^ ((unsigned long long)v * 0x9E3779B97F4A7C15ull)) >> 32) < thr;
    const unsigned long long m = __ballot(ok);
    if (!m) continue;
    int base = 0;
    if (lane == 0) base = atomicAdd(&s_n, __popcll(m));
    base = __shfl(base, 0);
    const int at = base + __popcll(m & ((1ull << lane) - 1));
    if (ok) {
      if (at < kLocal) s_list[at] = (int)v;
      else { const int k = atomicAdd(count, 1); if (k < cap) { list[k] = (int)v; ++kept; } }
    }
  }
  __syncthreads();
  const int nl = s_n < kLocal ? s_n : kLocal;
  if (threadIdx.x == 0) s_base = nl ? atomicAdd(count, nl) : 0;
  __syncthreads();
  for (int j = threadIdx.x; j < nl; j += blockDim.x) {
    const int k = s_base + j;
    if (k < cap) { list[k] = s_list[j]; ++kept; }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) kept += __shfl_xor(kept, m);
  if (lane == 0 && kept) atomicAdd(audit_rec + 6, kept);
  if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(audit_rec + 7, population);
}

// more voxels lay within the refinement threshold of the level than the list holds (status[1] != 0): bit 30 of the range word of
// the bbox record tells the caller, who reads that record anyway
// ... and bit 29 that the cluster form of the short-list kernel has reported a member that never arrived (status[11], sticky: the
// list was evaluated by the tile form instead - the result is complete - and the host switches the cluster form off)
__global__ void near_overflow_to_bbox_kernel(const int* status, int* bbox) {
  if (threadIdx.x == 0 && status[1] != 0) atomicOr(bbox + 7, 0x40000000);
  if (threadIdx.x == 0 && status[11] != 0) atomicOr(bbox + 7, 0x20000000);
}

__global__ void bbox_reinit_keep_flags_kernel(int* bbox, const int* flag) {   // words 7 / 15 (the fp16 range report) survive
  if (flag && *flag == 0) return;
  const int i = threadIdx.x;
  if (i < 16 && (i & 7) != 7) {
    const int j = i & 7;
    bbox[i] = j < 3 ? 0x7fffffff : (j < 6 ? -1 : 0);
  }
}

}  // namespace asdf

using namespace asdf;

struct asdf_decoder {
  asdf_decoder_spec_t spec;
  int device;
  int num_cus;
  float* stream;    // [kStagesAll][kStageFloats]
  float* wlat;      // [heads][2][512][256]
  float* wpt;       // [heads][2][512][MAXPF]
  float* bias02;    // [heads][2][512]
  float* cst;       // [heads][cst_offsets(kp).floats]  (static parts written at create time)
  int kp;           // point-feature K-steps: 2 (affine xyz) or ceil(pf / 2) (NeRF encoding)
  float* embed;     // [heads][MAXPF][4]
  float* latent_stage;   // [kLatent]: the latent of a sample whose codes came from host memory (asdf_decoder_set_sample_host)
  float* cls;       // [kMaxClasses][512 in D-layout order] + [kMaxClasses]; null until asdf_decoder_set_classifier
  int num_class;
  // split-half image (pack_decoder_f16) and the arithmetic in use
  float* stream16;
  float* stream16_hi;   // the high planes alone (1 KiB records): weight stream of the one-plane kernel (asdf_decode_grid_box)
  float* cst16;
  float* a16;       // [heads][kA16Floats] fp16 point-feature / bias operands of the one-plane kernels (K0b, per sample; affine features only)
  // The one-plane kernels' OWN image (round 4): weight scales chosen so that every accumulator already carries its activation's
  // plane scale - S_w1' = S_x1 / S_x0, S_w2' = S_x2 / S_x1, layer 0's constants times S_x0 - and the epilogue needs no rescale
  // (a v_pk_mul_f32 per register pair: with the range maximum, a third of the kernel's VALU issue).  cst16p1 = its constants
  // block, stream16_hi's layer-1 / layer-2 records = the high planes times S_w' / S_w (exact: powers of two); both are rebuilt
  // from the host copies whenever the activation scales change.
  float* cst16p1;
  float* cst16p1_host;
  uint16_t* hi_host;      // high planes at the split-half scales (the base of every rebuild), hi_count halves
  uint16_t* hi_scaled;    // staging buffer of the rescaled image
  size_t hi_count;
  float s0p[ASDF_MAX_HEADS], s2p[ASDF_MAX_HEADS];
  int p1_usable;          // 0: a rescaled weight would leave the fp16 range - the one-plane sweeps are not offered for this decoder
  float s2[ASDF_MAX_HEADS];
  int math;
  bool sample_bound;
  int* status;      // [16] device words: [0] = lanes whose activations left the fp16 range in split-half launches since the
                    // last clear, [1] = near-level voxels that did not fit the refinement list, [2] = scratch flag of the
                    // refinement (a voxel left the negative set: recount the box), [4..6] / [8..10] = largest
                    // plane value (float bits) of h0 / h1 / h2 of MLP 0 / MLP 1
  // activation scales of the split-half image (asdf_decoder_set_act_scales) and what is needed to rebuild its constants
  float sw[ASDF_MAX_HEADS][3];
  float sx[ASDF_MAX_HEADS][3];
  float* cst_host;    // the fp32 constants image (static parts), host copy
  float* cst16_host;  // staging buffer of the rebuilt split-half constants
  // near-level refinement of split-half sweeps (asdf_decoder_set_refine)
  void* ev_start;   // one-shot hipEvent_t pair recorded around the dominant kernel of the next sweep (asdf_decoder_time_next_sweep)
  void* ev_stop;
  float refine_tau;
  int* near_idx;    // [kNearCap] lattice indices
  int* near_count;  // device word
  // narrow-band fine sweep (asdf_decode_grid_band): voxel marks and the per-head re-evaluation lists, allocated on first use
  unsigned char* band_mark;
  size_t band_mark_bytes;
  int* box_aux;     // box-only sweep, large candidate lists: [0..2] count words, [4 ..] near-level voxels among the candidates (kNearCap)
  int* band_idx;    // [2][kBandCap]
  int* band_count;  // [2] device words
  // the exact re-evaluation of a short voxel list is latency-bound (one 128-point tile of one MLP on the fp32 chain takes
  // 0.24 ms whatever the list holds): the two MLPs of a SeparateDecoder run side by side, the second on this stream
  hipStream_t side;
  hipEvent_t ev_fork, ev_join;
  // audit of the one-plane sweeps (asdf_decoder_set_audit): voxels per sweep and head, the seed of the next draw, the record
  // ([0] largest |exact - one-plane| bits, [1] sign contradictions, [2] evaluations, [4] / [5] first audit position of the band
  // lists), and the voxel list of the box sweep's audit
  int audit_n;
  unsigned long long audit_seed;
  int* audit_rec;   // [12]: [0..7] the record, [8] / [9] shell population per use, [10] / [11] band compaction done-counters
  int* audit_idx;   // [kAuditCap]
  int* audit_count;
  int short_max;    // voxel lists of up to this many points are re-evaluated by the short-list form of the fp32 chain (0 = never)
  // ... and the shortest ones by its cluster form (four workgroups per 32 points): exchange buffers and arrival counters of
  // kShortClusters clusters (asdf_decoder_set_cluster_list)
  ShortParams shortp;
  // the audit of a box sweep runs beside the exact re-evaluation of its candidates (round 5)
  hipStream_t audit_side;
  hipEvent_t ev_audit_fork, ev_audit_join;
};
static constexpr int kShortClusters = kClusterCap / kWavePts * kHeads;      // 128
static constexpr int kNearCap = 1 << 16;      // near-level refinement list of a split-half sweep
static constexpr int kCandCap = 1 << 21;      // box candidates of asdf_decode_grid_box (a head without a certainly negative voxel - a thin
                                              // or tiny shape - lists its whole surface shell); shares near_idx
static constexpr int kAuditCap = 1 << 18;

namespace asdf {
// K1 in its subset mode over one voxel list: both MLPs of a SeparateDecoder concurrently (one launch each, the second on the
// decoder's side stream, joined back into `st`), everything else as one launch
static int launch_subset(asdf_decoder* d, const DecodeParams& q, bool two_out, int grid, hipStream_t st);
}
static constexpr int kCandDirect = 1 << 15;  // candidates of a box-only sweep that go straight to the fp32 chain (more: split-half kernel first)
static constexpr int kBandCap = 1 << 22;     // voxels per head the narrow-band sweep re-evaluates at most (25 % of 256^3)

// SeparateDecoder: 2 MLPs x 1 output; CombinedDecoder: 1 MLP x 2 outputs
// (Re)build the one-plane image for the activation scales in force: constants block + the layer-1 / layer-2 records of the
// high-plane stream (see asdf_decoder::cst16p1).  Host work on ~1 M halves + two uploads; the caller has synchronised.
static hipError_t rebuild_one_plane(asdf_decoder* d) {
  if (!d->stream16_hi || !d->hi_host) return hipSuccess;
  const CstOffsets co = cst_offsets(d->kp);
  const size_t n = (size_t)kHeads * co.floats;
  std::memcpy(d->cst16p1_host, d->cst_host, n * sizeof(float));
  d->p1_usable = 1;
  std::memcpy(d->hi_scaled, d->hi_host, d->hi_count * sizeof(uint16_t));
  for (int h = 0; h < d->spec.num_heads; ++h) {
    const float sx0 = d->sx[h][0], sx1 = d->sx[h][1], sx2 = d->sx[h][2];
    // (the kernels' ReLU is c + |c| = 2 relu(c) - it keeps NaNs and infinities alive, sdf_mlp_f16_kernel.h - so the accumulators of
    // layers 0..2 carry HALF their activation's plane scale, and the last layer's weights the other 1 / 2 of its a w + |a| w form)
    const float sw1p = 0.5f * sx1 / sx0, sw2p = 0.5f * sx2 / sx1, sw3 = d->sw[h][2];
    const float* c = d->cst_host + (size_t)h * co.floats;
    float* o = d->cst16p1_host + (size_t)h * co.floats;
    for (int i = 0; i < kTilesL1 * 32; ++i) o[co.b1 + i] = c[co.b1 + i] * (sw1p * sx0);
    const float s3 = sw3 * sx2;
    for (int i = 0; i < kHidden; ++i) { o[co.b3 + i] = c[co.b3 + i] * s3; o[co.w4 + i] = 0.5f * c[co.w4 + i] / s3; o[co.w4b + i] = 0.5f * c[co.w4b + i] / s3; }
    o[co.b4] = c[co.b4]; o[co.b4 + 1] = c[co.b4 + 1];
    o[co.b4 + 2] = o[co.b4 + 3] = o[co.b4 + 4] = 1.0f;                       // (the multipliers of the split-half image: unused here)
    d->s0p[h] = 0.5f * sx0;
    d->s2p[h] = sw2p * sx1;
    if (d->spec.feature_mode == ASDF_FEATURES_NERF) {
      // the NeRF encoding's point-feature fragments are static weights (pack_decoder), not a per-sample fold: scaled here
      for (int i = 0; i < kTilesHidden * d->kp * 64; ++i) { o[co.a0 + i] = c[co.a0 + i] * d->s0p[h]; o[co.a2 + i] = c[co.a2 + i] * d->s2p[h]; }
    }
    // records of one head: [0, 256) layer 1, [256, 512) layer 2, [512, 1024) layer 3 (unchanged), 512 halves each
    const float f[2] = {sw1p / d->sw[h][0], sw2p / d->sw[h][1]};
    for (int l = 0; l < 2; ++l) {
      uint16_t* rec = d->hi_scaled + ((size_t)h * 1024 + (size_t)l * 256) * 512;
      float top = 0.0f;
      for (size_t i = 0; i < (size_t)256 * 512; ++i) {
        _Float16 v;
        std::memcpy(&v, rec + i, 2);
        const float w = (float)v * f[l];
        top = std::fmax(top, std::fabs(w));
        v = (_Float16)w;
        std::memcpy(rec + i, &v, 2);
      }
      // the largest weight must stay a comfortable fp16 number: below 2^15, and high enough that weights down to 2^-8 of it are
      // still normal (they carry the product sums; smaller ones go subnormal, 2^-25 absolute - far below the one-plane error)
      if (!(top < 32768.0f) || !(top >= 0.015625f)) d->p1_usable = 0;
    }
  }
  hipError_t e = hipMemcpy(d->cst16p1, d->cst16p1_host, n * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d->stream16_hi, d->hi_scaled, d->hi_count * sizeof(uint16_t), hipMemcpyHostToDevice);
  return e;
}

static bool spec_supported(const asdf_decoder_spec_t* s) {
  if (s->latent_size != kLatent || s->hidden != kHidden) return false;
  if (s->feature_mode == ASDF_FEATURES_NERF) {
    if (s->point_feats[0] != 9 && s->point_feats[0] != 15) return false;
    if (s->num_heads == 2 && s->point_feats[1] != s->point_feats[0]) return false;
  } else if (s->feature_mode != ASDF_FEATURES_AFFINE) {
    return false;
  }
  if (s->num_heads == 2) return s->outputs[0] == 1 && s->outputs[1] == 1;
  if (s->num_heads == 1) return s->outputs[0] == 2;
  return false;
}

extern "C" {

int asdf_version(void) { return 129; }

int asdf_set_mfma_shape(int shape) {
  if (shape != 0 && shape != 16 && shape != 32) return ASDF_EINVAL;
  return k1h_set_shape(shape);
}
int asdf_get_mfma_shape(void) { return k1h_shape(); }

const char* asdf_strerror(int code) {
  switch (code) {
    case ASDF_OK: return "ok";
    case ASDF_EINVAL: return "invalid argument or unsupported decoder shape";
    case ASDF_ENOMEM: return "out of memory";
    case ASDF_EHIP: return "HIP runtime error";
    case ASDF_ENODEV: return "no gfx950 device";
    case ASDF_ENOSPC: return "buffer or workspace too small";
    case ASDF_ERANGE: return "Surface level must be within volume data range.";
    case ASDF_ENOSURF: return "No surface found at the given iso value.";
    default: return "unknown error";
  }
}

int asdf_last_hip_error(void) { return g_last_hip_error; }

int asdf_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  int good = 0;
  for (int d = 0; d < n; ++d) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, d) == hipSuccess && std::strncmp(prop.gcnArchName, "gfx950", 6) == 0) ++good;
  }
  return good;
}

void asdf_decoder_destroy(asdf_decoder_t* d) {
  if (!d) return;
  float* bufs[] = {d->stream, d->wlat, d->wpt, d->bias02, d->cst, d->embed, d->cls, d->stream16, d->cst16, d->stream16_hi, d->a16, d->cst16p1};
  if (d->side) { (void)hipStreamSynchronize(d->side); (void)hipStreamDestroy(d->side); }
  if (d->audit_side) { (void)hipStreamSynchronize(d->audit_side); (void)hipStreamDestroy(d->audit_side); }
  if (d->ev_audit_fork) (void)hipEventDestroy(d->ev_audit_fork);
  if (d->ev_audit_join) (void)hipEventDestroy(d->ev_audit_join);
  if (d->shortp.xchg) (void)hipFree(d->shortp.xchg);
  if (d->shortp.arrivals) (void)hipFree(d->shortp.arrivals);
  if (d->ev_fork) (void)hipEventDestroy(d->ev_fork);
  if (d->ev_join) (void)hipEventDestroy(d->ev_join);
  if (d->band_mark) (void)hipFree(d->band_mark);
  if (d->box_aux) (void)hipFree(d->box_aux);
  if (d->band_idx) (void)hipFree(d->band_idx);
  if (d->band_count) (void)hipFree(d->band_count);
  for (float* b : bufs) (void)hipFree(b);
  (void)hipFree(d->status);
  if (d->latent_stage) (void)hipFree(d->latent_stage);
  (void)hipFree(d->near_idx);
  (void)hipFree(d->near_count);
  (void)hipFree(d->audit_rec);
  (void)hipFree(d->audit_idx);
  (void)hipFree(d->audit_count);
  std::free(d->cst_host);
  std::free(d->cst16_host);
  std::free(d->cst16p1_host);
  std::free(d->hi_host);
  std::free(d->hi_scaled);
  delete d;
}

int asdf_decoder_create(const asdf_decoder_spec_t* spec, const asdf_head_params_t* heads, asdf_decoder_t** out) {
  if (!spec || !heads || !out) return ASDF_EINVAL;
  *out = nullptr;
  if (!spec_supported(spec)) return ASDF_EINVAL;
  for (int h = 0; h < spec->num_heads; ++h) {
    if (spec->point_feats[h] < 1 || spec->point_feats[h] > ASDF_MAX_POINT_FEATS) return ASDF_EINVAL;
    if (kHidden - kLatent - spec->point_feats[h] < 1) return ASDF_EINVAL;
    for (int l = 0; l < 5; ++l)
      if (!heads[h].w[l] || !heads[h].b[l]) return ASDF_EINVAL;
  }
  int dev = 0;
  ASDF_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  ASDF_HIP(hipGetDeviceProperties(&prop, dev));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return ASDF_ENODEV;

  asdf_decoder* d = new (std::nothrow) asdf_decoder();
  if (!d) return ASDF_ENOMEM;
  std::memset(d, 0, sizeof(*d));
  d->spec = *spec;
  d->device = dev;
  d->num_cus = prop.multiProcessorCount;

  HostPack hp;
  if (!pack_decoder(*spec, heads, hp)) { delete d; return ASDF_ENOMEM; }
  d->kp = hp.kp;
  std::vector<float>&stream = hp.stream, &wlat = hp.wlat, &wpt = hp.wpt, &b02 = hp.b02, &cst = hp.cst, &emb = hp.emb;

  hipError_t e = hipSuccess;
  auto up = [&](float** dst, const std::vector<float>& src) {
    if (e != hipSuccess) return;
    e = hipMalloc((void**)dst, src.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(*dst, src.data(), src.size() * sizeof(float), hipMemcpyHostToDevice);
  };
  up(&d->stream, stream); up(&d->wlat, wlat); up(&d->wpt, wpt); up(&d->bias02, b02); up(&d->cst, cst); up(&d->embed, emb);
  d->math = ASDF_MATH_F32;
  if (e == hipSuccess) {
    if (!pack_decoder_f16(*spec, heads, hp)) { asdf_decoder_destroy(d); return ASDF_ENOMEM; }
    up(&d->cst16, hp.cst16);
    // (one allocation: the 32x32x16 image, and behind it the W form's - its kernels add the offset themselves)
    if (e == hipSuccess) e = hipMalloc((void**)&d->stream16, (hp.stream16.size() + hp.stream16w.size()) * sizeof(uint16_t));
    if (e == hipSuccess) e = hipMemcpy(d->stream16, hp.stream16.data(), hp.stream16.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(reinterpret_cast<uint16_t*>(d->stream16) + hp.stream16.size(), hp.stream16w.data(),
                                       hp.stream16w.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
      // records of [plane hi / lo][lane][8 halves] -> the hi halves alone, same record order
      const size_t nrec = hp.stream16.size() / 1024;
      std::vector<uint16_t> hi(nrec * 512);
      for (size_t r = 0; r < nrec; ++r) std::memcpy(&hi[r * 512], &hp.stream16[r * 1024], 512 * sizeof(uint16_t));
      e = hipMalloc((void**)&d->stream16_hi, hi.size() * sizeof(uint16_t));
      if (e == hipSuccess) e = hipMemcpy(d->stream16_hi, hi.data(), hi.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
      d->hi_count = hi.size();
      d->hi_host = (uint16_t*)std::malloc(hi.size() * sizeof(uint16_t));
      d->hi_scaled = (uint16_t*)std::malloc(hi.size() * sizeof(uint16_t));
      d->cst16p1_host = (float*)std::malloc(hp.cst.size() * sizeof(float));
      if (!d->hi_host || !d->hi_scaled || !d->cst16p1_host) { asdf_decoder_destroy(d); return ASDF_ENOMEM; }
      std::memcpy(d->hi_host, hi.data(), hi.size() * sizeof(uint16_t));
      if (e == hipSuccess) e = hipMalloc((void**)&d->cst16p1, hp.cst.size() * sizeof(float));
      if (e == hipSuccess && d->kp == 2) {      // (affine point features: layer 0's bias row and point columns as one fp16 A operand, K0b)
        e = hipMalloc((void**)&d->a16, (size_t)kHeads * kA16Floats * sizeof(float));
        if (e == hipSuccess) e = hipMemset(d->a16, 0, (size_t)kHeads * kA16Floats * sizeof(float));
      }
    }
    for (int h = 0; h < kHeads; ++h) {
      d->s2[h] = h < spec->num_heads ? hp.s2[h] : 1.0f;
      for (int l = 0; l < 3; ++l) { d->sw[h][l] = h < spec->num_heads ? hp.sw[h][l] : 1.0f; d->sx[h][l] = kActScale; }
    }
    d->cst_host = (float*)std::malloc(hp.cst.size() * sizeof(float));
    d->cst16_host = (float*)std::malloc(hp.cst.size() * sizeof(float));
    if (!d->cst_host || !d->cst16_host) { asdf_decoder_destroy(d); return ASDF_ENOMEM; }
    std::memcpy(d->cst_host, hp.cst.data(), hp.cst.size() * sizeof(float));
    if (e == hipSuccess) e = rebuild_one_plane(d);          // the one-plane image at the default activation scales
    if (e == hipSuccess) e = k1h_prepare();
    if (e == hipSuccess) e = k1s_nerf_prepare();
  }
  if (e == hipSuccess) e = k1_prepare();
  if (e == hipSuccess) e = k1_cls_prepare();
  if (e == hipSuccess) e = hipMalloc((void**)&d->latent_stage, kLatent * sizeof(float));
  if (e == hipSuccess) e = hipMalloc((void**)&d->status, 16 * sizeof(int));
  if (e == hipSuccess) e = hipMemset(d->status, 0, 16 * sizeof(int));
  if (e == hipSuccess) e = hipMalloc((void**)&d->near_idx, kCandCap * sizeof(int));
  if (e == hipSuccess) e = hipMalloc((void**)&d->near_count, sizeof(int));
  if (e == hipSuccess) e = hipMalloc((void**)&d->audit_rec, 12 * sizeof(int));
  if (e == hipSuccess) e = hipMalloc((void**)&d->audit_idx, kAuditCap * sizeof(int));
  if (e == hipSuccess) e = hipMalloc((void**)&d->audit_count, sizeof(int));
  d->audit_n = 1 << 16;
  d->audit_seed = 0x5DF5A11D00000000ull;
  d->short_max = 8192;
  d->shortp.cluster_max = kClusterCap;
  d->shortp.fault = d->status ? d->status + 11 : nullptr;
  d->shortp.timeout_ticks = kClusterTimeoutTicks;
  if (e == hipSuccess) e = hipMalloc((void**)&d->shortp.xchg, (size_t)kShortClusters * kXchgFloats * sizeof(float));
  if (e == hipSuccess) e = hipMalloc((void**)&d->shortp.arrivals, (size_t)kShortClusters * 4 * sizeof(int));
  if (e == hipSuccess) e = hipMemset(d->shortp.arrivals, 0, (size_t)kShortClusters * 4 * sizeof(int));
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&d->audit_side, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&d->ev_audit_fork, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&d->ev_audit_join, hipEventDisableTiming);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&d->side, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&d->ev_fork, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&d->ev_join, hipEventDisableTiming);
  d->refine_tau = 4e-6f;

  if (e != hipSuccess) {
    g_last_hip_error = (int)e;
    asdf_decoder_destroy(d);
    return e == hipErrorOutOfMemory ? ASDF_ENOMEM : ASDF_EHIP;
  }
  *out = d;
  return ASDF_OK;
}

int asdf_neg_bbox(const float* vol_dev, int32_t n0, int32_t n1, int32_t n2, int32_t* bbox_dev, void* stream) {
  if (!vol_dev || !bbox_dev || n0 < 1 || n1 < 1 || n2 < 1) return ASDF_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(bbox_init_kernel, dim3(1), dim3(64), 0, st, bbox_dev);
  const long long rows = (long long)n0 * n1;
  const int grid = (int)((rows + 3) / 4 < 1024 ? (rows + 3) / 4 : 1024);
  hipLaunchKernelGGL(neg_bbox_kernel, dim3(grid), dim3(256), 0, st, vol_dev, n0, n1, n2, bbox_dev, (const int*)nullptr);
  ASDF_HIP(hipGetLastError());
  return ASDF_OK;
}

int asdf_debug_pack_host(const asdf_decoder_spec_t* spec, const asdf_head_params_t* heads, float* stream,
                         float* wlat, float* wpt, float* bias02, float* cst, float* embed) {
  if (!spec || !heads || !spec_supported(spec)) return ASDF_EINVAL;
  HostPack hp;
  if (!pack_decoder(*spec, heads, hp)) return ASDF_ENOMEM;
  auto cp = [](float* dst, const std::vector<float>& v) { if (dst) std::memcpy(dst, v.data(), v.size() * sizeof(float)); };
  cp(stream, hp.stream); cp(wlat, hp.wlat); cp(wpt, hp.wpt); cp(bias02, hp.b02); cp(cst, hp.cst); cp(embed, hp.emb);
  return ASDF_OK;
}

int asdf_debug_pack_host_f16w(const asdf_decoder_spec_t* spec, const asdf_head_params_t* heads, uint16_t* stream16w) {
  if (!spec || !heads || !stream16w || !spec_supported(spec)) return ASDF_EINVAL;
  HostPack hp;
  if (!pack_decoder(*spec, heads, hp)) return ASDF_ENOMEM;
  if (!pack_decoder_f16(*spec, heads, hp)) return ASDF_ENOMEM;
  std::memcpy(stream16w, hp.stream16w.data(), hp.stream16w.size() * sizeof(uint16_t));
  return ASDF_OK;
}

int asdf_debug_pack_host_f16(const asdf_decoder_spec_t* spec, const asdf_head_params_t* heads, uint16_t* stream16, float* cst16,
                             float* s2) {
  if (!spec || !heads || !spec_supported(spec)) return ASDF_EINVAL;
  HostPack hp;
  if (!pack_decoder(*spec, heads, hp)) return ASDF_ENOMEM;
  if (!pack_decoder_f16(*spec, heads, hp)) return ASDF_ENOMEM;
  if (stream16) std::memcpy(stream16, hp.stream16.data(), hp.stream16.size() * sizeof(uint16_t));
  if (cst16) std::memcpy(cst16, hp.cst16.data(), hp.cst16.size() * sizeof(float));
  if (s2) for (int h = 0; h < kHeads; ++h) s2[h] = h < spec->num_heads ? hp.s2[h] : 1.0f;
  return ASDF_OK;
}

// the per-sample codes of a caller whose codes live on the HOST (pinned, device-addressable memory): one workgroup reads them over
// the link into the decoder's own device words, in stream order, right in front of the fold - no copy engine, no blit kernel, no
// side stream (round 6, VERDICT r05 item 3: the runtime's copy of a 1 KB latent was a shader blit that could not get a wave slot
// while a persistent sweep owned every compute unit and sat resident for the whole sweep - 21.5 ms per sample in the eval-mode trace)
__global__ __launch_bounds__(256) void stage_sample_kernel(const float* __restrict__ latent_host, const float* __restrict__ embed_host,
                                                           float* __restrict__ latent_dev, float* __restrict__ embed_dev, int n_embed) {
  const int i = threadIdx.x;
  if (i < kLatent) latent_dev[i] = latent_host[i];
  if (embed_host)
    for (int k = i; k < n_embed; k += 256) embed_dev[k] = embed_host[k];
}

static int set_sample_fold(asdf_decoder_t* d, const float* latent_dev, hipStream_t st);

int asdf_decoder_set_sample(asdf_decoder_t* d, const float* latent_dev, const float* embed_host, void* stream) {
  if (!d || !latent_dev) return ASDF_EINVAL;
  if (embed_host && d->spec.feature_mode != ASDF_FEATURES_AFFINE) return ASDF_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (embed_host) {
    ASDF_HIP(hipMemcpyAsync(d->embed, embed_host, sizeof(float) * kHeads * ASDF_MAX_POINT_FEATS * 4,
                            hipMemcpyHostToDevice, st));
  }
  return set_sample_fold(d, latent_dev, st);
}

int asdf_decoder_set_sample_host(asdf_decoder_t* d, const float* latent_pinned, const float* embed_pinned, void* stream) {
  if (!d || !latent_pinned || !d->latent_stage) return ASDF_EINVAL;
  if (embed_pinned && d->spec.feature_mode != ASDF_FEATURES_AFFINE) return ASDF_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(stage_sample_kernel, dim3(1), dim3(256), 0, st, latent_pinned, embed_pinned, d->latent_stage, d->embed,
                     kHeads * ASDF_MAX_POINT_FEATS * 4);
  ASDF_HIP(hipGetLastError());
  return set_sample_fold(d, d->latent_stage, st);
}

static int set_sample_fold(asdf_decoder_t* d, const float* latent_dev, hipStream_t st) {
  FoldParams fp;
  fp.wlat = d->wlat; fp.wpt = d->wpt; fp.bias02 = d->bias02; fp.embed = d->embed; fp.latent = latent_dev;
  fp.kp = d->kp;
  int images = 1;
  fp.cst[0] = d->cst; fp.cst[1] = fp.cst[2] = nullptr;
  for (int h = 0; h < kHeads; ++h) {
    fp.pf[h] = h < d->spec.num_heads ? d->spec.point_feats[h] : 0;
    for (int i = 0; i < 3; ++i) { fp.s2[i][h] = 1.0f; fp.s0[i][h] = 1.0f; }
  }
  if (d->cst16) {      // the same fold into the split-half image, layer-2 constants scaled
    fp.cst[1] = d->cst16;
    for (int h = 0; h < kHeads; ++h) fp.s2[1][h] = d->s2[h];
    images = 2;
    if (d->cst16p1) {
      // ... and into the one-plane image (its own scales), whose layer-0 constants then become the kernels' fp16 point-feature /
      // bias operands (affine features; the NeRF-encoded kernels read them from the constants block as they are)
      fp.cst[2] = d->cst16p1;
      for (int h = 0; h < kHeads; ++h) { fp.s2[2][h] = d->s2p[h]; fp.s0[2][h] = d->s0p[h]; }
      images = 3;
    }
  }
  hipLaunchKernelGGL(fold_sample_kernel, dim3(d->spec.num_heads * 2 * kHidden / 4, images), dim3(256), 0, st, fp);
  if (images == 3 && d->a16) hipLaunchKernelGGL(fold_points_f16_kernel, dim3(d->spec.num_heads * 2), dim3(512), 0, st, d->cst16p1, d->a16, d->status);
  ASDF_HIP(hipGetLastError());
  d->sample_bound = true;
  return ASDF_OK;
}

namespace asdf {
static int launch_subset(asdf_decoder* d, const DecodeParams& q_in, bool two_out, int grid, hipStream_t st) {
  DecodeParams q = q_in;
  if (d->kp == 2 && d->short_max > 0 && q.mode == kGridSubset) {
    // short lists (the usual case of a near-level refinement: a few dozen to a few hundred voxels) take the short-list form - a
    // quarter of the tile form's latency; both are enqueued, the device-side count decides which one runs (bit-identical results)
    q.short_max = d->short_max;
    ShortParams sp = d->shortp;
    if (sp.cluster_max > d->short_max) sp.cluster_max = d->short_max;
    if (d->num_cus < 64) sp.cluster_max = 0;       // (its members wait for each other: they need workgroup slots side by side)
    // four workgroups per block and MLP, one per compute unit (155 KB of LDS each): beyond 64 blocks x MLPs the cluster form needs a
    // second round of the chip and loses to the short form's single one (N = 128 trace: 1 400 candidates x 2 MLPs, 0.22 -> 0.36 ms per
    // sample) - lists of up to 2048 / MLPs voxels take it
    if (sp.cluster_max > kClusterCap / q.num_mlps) sp.cluster_max = kClusterCap / q.num_mlps;
    if (!sp.fault) sp.cluster_max = 0;
    // a cluster launch that timed out waiting for a member writes nothing and raises *sp.fault: the tile form below - enqueued anyway,
    // it returns at once for a short list - then evaluates the list (same bits)
    q.short_fault = sp.cluster_max > 0 ? sp.fault : nullptr;
    k1_short_launch(two_out, q, sp, st);
  }
  if (two_out || q.num_mlps != 2 || !q.sdf0 || !q.sdf1 || !d->side) {
    k1_launch(d->kp, two_out, q, grid, st);
    return ASDF_OK;
  }
  DecodeParams q0 = q, q1 = q;
  q0.num_mlps = 1; q0.first_mlp = 0; q0.sdf1 = nullptr;
  q1.num_mlps = 1; q1.first_mlp = 1; q1.sdf0 = nullptr;
  ASDF_HIP(hipEventRecord(d->ev_fork, st));
  ASDF_HIP(hipStreamWaitEvent(d->side, d->ev_fork, 0));
  k1_launch(d->kp, false, q0, grid, st);
  k1_launch(d->kp, false, q1, grid, d->side);
  ASDF_HIP(hipEventRecord(d->ev_join, d->side));
  ASDF_HIP(hipStreamWaitEvent(st, d->ev_join, 0));
  return ASDF_OK;
}
}  // namespace asdf

static int launch_decode(asdf_decoder_t* d, DecodeParams& p, hipStream_t st) {
  if (!d->sample_bound) return ASDF_EINVAL;
  p.stream = d->stream;
  p.cst = d->cst;
  p.status = d->status;
  // a SeparateDecoder head whose output pointer is NULL is not evaluated at all (the reference always runs both,
  // networks/model.py:304-344, and discards one when HandBranch / ObjectBranch is off)
  p.first_mlp = 0;
  p.num_mlps = d->spec.num_heads;
  const bool two_out = d->spec.num_heads == 1;     // CombinedDecoder: one MLP, two last-layer rows
  if (p.bbox) {
    hipLaunchKernelGGL(bbox_init_kernel, dim3(1), dim3(64), 0, st, p.bbox);
    ASDF_HIP(hipGetLastError());
  }
  const bool want_cls = p.logits || p.labels;
  if (!two_out) {
    if (!p.sdf0 && !p.sdf1 && !want_cls) return ASDF_OK;
    if (!p.sdf1) p.num_mlps = 1;
    else if (!p.sdf0) { p.first_mlp = 1; p.num_mlps = 1; }
  }
  const long long ntiles = (p.P + kWgPts - 1) / kWgPts;
  if (ntiles == 0) return ASDF_OK;
  const int grid = (int)(ntiles < d->num_cus ? ntiles : d->num_cus);
  p.pf = d->spec.point_feats[0];
  if (p.logits || p.labels) {
    // label pass: the classifier reads MLP 0's last hidden activation, so MLP 0 always runs
    if (!d->cls) return ASDF_EINVAL;
    p.cls = d->cls;
    p.num_class = d->num_class;
    if (!two_out && p.first_mlp != 0) { p.first_mlp = 0; p.num_mlps = 2; }
    k1_cls_launch(d->kp, two_out, p, grid, st);
    ASDF_HIP(hipGetLastError());
    return ASDF_OK;
  }
  // split-half arithmetic: grid sweeps only; the fp16 range report goes to the decoder's status word (asdf_decoder_status)
  // and, when the caller passed one, to word 7 / 15 of the bbox record it reads back anyway.  Explicit point lists (the
  // label pass, a few 10^4 points) stay on the fp32 chain
  if (d->math == ASDF_MATH_F16X3 && d->stream16 && p.mode != kPointList) {
    p.stream = d->stream16;
    p.cst = d->cst16;
    if (d->ev_start) ASDF_HIP(hipEventRecord((hipEvent_t)d->ev_start, st));
    k1h_launch(d->kp, two_out, p, grid, st);
    if (d->ev_stop) ASDF_HIP(hipEventRecord((hipEvent_t)d->ev_stop, st));
    d->ev_start = d->ev_stop = nullptr;
    if (d->refine_tau > 0.0f && p.P <= 0x7fffffffLL) {
      // Near-level refinement: the split-half result and the exact fp32 FMA chain differ by a few 1e-7, so the SIGN of a
      // voxel within that distance of the iso level - all that marching cubes and the negative-voxel box look at - could
      // depend on the arithmetic.  Every voxel with |sdf| < tau (a few hundred of 16.7 M at N = 256) is re-evaluated on
      // the fp32 MFMA chain (the same kernel as ASDF_MATH_F32, coordinates from the same device function) and written
      // back in place: surfaces and boxes are those of the fp32 chain.
      ASDF_HIP(hipMemsetAsync(d->near_count, 0, sizeof(int), st));
      ASDF_HIP(hipMemsetAsync(d->status + 1, 0, 2 * sizeof(int), st));  // this sweep's list-overflow count and "box may shrink" flag
      const long long n4 = (p.P + 3) / 4;
      const int cgrid = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
      hipLaunchKernelGGL(collect_near_level_kernel, dim3(cgrid), dim3(256), 0, st, p.sdf0, p.sdf1, p.P, d->refine_tau, d->near_idx,
                         d->near_count, kNearCap, d->status);
      DecodeParams q = p;
      q.stream = d->stream; q.cst = d->cst; q.bbox = p.bbox; q.fixup_flag = p.bbox ? d->status + 2 : nullptr;
      q.mode = kGridSubset; q.grid_mode = p.mode; q.idx = d->near_idx; q.count_dev = d->near_count; q.P = kNearCap;
      const int rgrid = kNearCap / kWgPts < d->num_cus ? kNearCap / kWgPts : d->num_cus;
      { const int rc = launch_subset(d, q, two_out, rgrid, st); if (rc != ASDF_OK) return rc; }
      if (p.bbox) {
        // the fused box saw the unrefined values.  The refinement pass patches it in place - a voxel that became negative
        // extends the box and the count exactly - and raises a flag when one became non-negative (the box may have to
        // shrink): only then do the three kernels below recount on the refined volumes (the range report words stay).
        const int* flag = d->status + 2;
        hipLaunchKernelGGL(bbox_reinit_keep_flags_kernel, dim3(1), dim3(64), 0, st, p.bbox, flag);
        const long long rows = (long long)p.N * p.N;
        const int bgrid = (int)((rows + 3) / 4 < 1024 ? (rows + 3) / 4 : 1024);
        if (p.sdf0) hipLaunchKernelGGL(neg_bbox_kernel, dim3(bgrid), dim3(256), 0, st, p.sdf0, p.N, p.N, p.N, p.bbox, flag);
        if (p.sdf1) hipLaunchKernelGGL(neg_bbox_kernel, dim3(bgrid), dim3(256), 0, st, p.sdf1, p.N, p.N, p.N, p.bbox + 8, flag);
        hipLaunchKernelGGL(near_overflow_to_bbox_kernel, dim3(1), dim3(64), 0, st, d->status, p.bbox);
      }
    }
  } else {
    if (d->ev_start) ASDF_HIP(hipEventRecord((hipEvent_t)d->ev_start, st));
    k1_launch(d->kp, two_out, p, grid, st);
    if (d->ev_stop) ASDF_HIP(hipEventRecord((hipEvent_t)d->ev_stop, st));
    d->ev_start = d->ev_stop = nullptr;
  }
  ASDF_HIP(hipGetLastError());
  return ASDF_OK;
}

// (origin / voxel_size by value, or - lattice_dev non-null - {origin0, origin1, origin2, voxel} read by the kernels from device memory)
static int decode_grid_impl(asdf_decoder_t* d, int32_t N, const float* origin, float voxel_size, const float* lattice_dev, int32_t grid_mode,
                            float* sdf_hand_dev, float* sdf_obj_dev, int32_t* bbox_dev, void* stream) {
  if (!d || (!origin && !lattice_dev) || N < 2 || N > 1024) return ASDF_EINVAL;
  if (grid_mode != ASDF_GRID_REFERENCE && grid_mode != ASDF_GRID_INTEGER) return ASDF_EINVAL;
  DecodeParams p;
  std::memset(&p, 0, sizeof(p));
  p.sdf0 = sdf_hand_dev; p.sdf1 = sdf_obj_dev; p.bbox = bbox_dev;
  p.P = (long long)N * N * N; p.N = N; p.mode = grid_mode == ASDF_GRID_REFERENCE ? kGridReference : kGridInteger;
  if (lattice_dev) p.lattice = lattice_dev;
  else { p.vs = voxel_size; p.o0 = origin[0]; p.o1 = origin[1]; p.o2 = origin[2]; }
  return launch_decode(d, p, (hipStream_t)stream);
}

int asdf_decode_grid(asdf_decoder_t* d, int32_t N, const float origin[3], float voxel_size, int32_t grid_mode,
                     float* sdf_hand_dev, float* sdf_obj_dev, int32_t* bbox_dev, void* stream) {
  if (!origin) return ASDF_EINVAL;
  return decode_grid_impl(d, N, origin, voxel_size, nullptr, grid_mode, sdf_hand_dev, sdf_obj_dev, bbox_dev, stream);
}

int asdf_decode_grid_dev(asdf_decoder_t* d, int32_t N, const float* lattice_dev, int32_t grid_mode,
                         float* sdf_hand_dev, float* sdf_obj_dev, int32_t* bbox_dev, void* stream) {
  if (!lattice_dev) return ASDF_EINVAL;
  return decode_grid_impl(d, N, nullptr, 0.0f, lattice_dev, grid_mode, sdf_hand_dev, sdf_obj_dev, bbox_dev, stream);
}

// get_higher_res_cube's arithmetic (utils/mesh.py:239-254) on the boxes of a coarse pass' record, in fp32 like the reference's CPU
// tensors: min / max over the enabled branches (an empty branch contributes zeros, :209-211, :225-227), every operation rounded
// separately (the file is built with -ffp-contract=off; the _rn intrinsics say so again).
__global__ void zoom_cube_kernel(const int* __restrict__ bbox, int N, float vs, int use_hand, int use_obj, float* __restrict__ lattice) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float lo[3], hi[3];
  bool first = true;
  for (int h = 0; h < 2; ++h) {
    if (!(h == 0 ? use_hand : use_obj)) continue;
    const int* b = bbox + 8 * h;
    const bool any = b[6] != 0;
    for (int a = 0; a < 3; ++a) {
      const float l = any ? (float)b[a] : 0.0f, u = any ? (float)b[3 + a] : 0.0f;
      lo[a] = first ? l : fminf(lo[a], l);
      hi[a] = first ? u : fmaxf(hi[a], u);
    }
    first = false;
  }
  if (first) { for (int a = 0; a < 3; ++a) lo[a] = hi[a] = 0.0f; }
  float span = __fsub_rn(hi[0], lo[0]);
  span = fmaxf(span, __fsub_rn(hi[1], lo[1]));
  span = fmaxf(span, __fsub_rn(hi[2], lo[2]));
  const float cube = __fmul_rn(__fadd_rn(span, 4.0f), vs);
  lattice[3] = __fdiv_rn(cube, (float)(N - 1));
  for (int a = 0; a < 3; ++a) lattice[a] = __fsub_rn(__fmul_rn(__fsub_rn(lo[a], 2.0f), vs), 1.0f);
}

int asdf_zoom_cube(const int32_t* bbox_dev, int32_t N, float voxel_size, int32_t hand_branch, int32_t obj_branch, float* lattice_dev,
                   void* stream) {
  if (!bbox_dev || !lattice_dev || N < 2 || (!hand_branch && !obj_branch)) return ASDF_EINVAL;
  hipLaunchKernelGGL(zoom_cube_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, bbox_dev, N, voxel_size, hand_branch, obj_branch, lattice_dev);
  ASDF_HIP(hipGetLastError());
  return ASDF_OK;
}

// the audit of a one-plane sweep: min(d->audit_n, P / 64) voxels the sweep decided by sign alone are appended to list[*count ..] -
// half of them drawn uniformly (audit_pick_kernel), half from the at-risk shell (shell_count_kernel / shell_pick_kernel) - and the
// seed advances.  The small lattices pay a share of their own size, not a sample sized for 256^3 (round 5: a
// 64th - still four times the share of the lattice that 65 536 picks are at N = 256.  At a sixteenth a 128^3 lattice paid the full
// 256^3 sample, 5 % of its sample time.  The shell half stays exhaustive wherever the shell is smaller than its budget - hundreds to
// thousands of voxels on these lattices - and the uniform half's size enters the tail ratio the estimate is corrected with).
static int audit_size(const asdf_decoder* d, long long P) {
  if (d->audit_n <= 0) return 0;
  const long long cap = P / 64 > 64 ? P / 64 : 64;
  return (long long)d->audit_n < cap ? d->audit_n : (int)cap;
}
static int enqueue_audit_picks(asdf_decoder* d, const float* a, const float* b, const unsigned char* mark, long long P, float tau,
                               int* list, int* count, int cap, int use, hipStream_t st) {
  const int n = audit_size(d, P);
  if (n <= 0) return ASDF_OK;
  const int uniform = n - n / 2, budget = n / 2;
  if (budget > 0) {
    // (the shell population of this use - head `use` of a band sweep, 0 for a box sweep - is audit_rec[8 + use]: cleared with the
    // record by sweep_init_kernel)
    int* shell_n = d->audit_rec + 8 + use;
    const int sgrid = (int)((P + 255) / 256 < 2048 ? (P + 255) / 256 : 2048);
    hipLaunchKernelGGL(shell_count_kernel, dim3(sgrid + (uniform + 255) / 256), dim3(256), 0, st, a, b, mark, P, tau, shell_n, sgrid,
                       d->audit_seed, uniform, list, count, cap);
    hipLaunchKernelGGL(shell_pick_kernel, dim3(sgrid), dim3(256), 0, st, a, b, mark, P, tau, d->audit_seed ^ 0xA5A5A5A55A5A5A5Aull, budget,
                       shell_n, list, count, cap, d->audit_rec);
  } else {
    hipLaunchKernelGGL(audit_pick_kernel, dim3((uniform + 255) / 256), dim3(256), 0, st, a, b, mark, P, tau, d->audit_seed, uniform,
                       list, count, cap);
  }
  ASDF_HIP(hipGetLastError());
  d->audit_seed = d->audit_seed * 6364136223846793005ull + 1442695040888963407ull;
  return ASDF_OK;
}

// words 32..41 of the record of a one-plane sweep, gathered on the device behind the call
__global__ void sweep_record_kernel(int* rec, const int* status, const int* near_count, const int* audit_rec) {
  const int i = threadIdx.x;
  if (i < 16) rec[16 + i] = status[i];
  if (i == 0) {
    rec[32] = near_count ? *near_count : 0;
    rec[33] = audit_rec[4];          // band sweep: voxels marked for the hand / object head (0 for the box sweep)
    rec[34] = audit_rec[5];
    rec[35] = audit_rec[0]; rec[36] = audit_rec[1]; rec[37] = audit_rec[2];
    rec[38] = status[1];
    rec[39] = audit_rec[6];          // audit picks drawn from the at-risk shell (tau <= |one-plane| < 2 tau)
    rec[40] = audit_rec[7];          // population of that shell (summed over the heads of a band sweep)
    rec[41] = audit_rec[3];          // sum of squared audit errors (float bits): sigma of the one-plane error = sqrt([41] / [37])
    for (int k = 42; k < 48; ++k) rec[k] = 0;
  }
}

int asdf_decode_grid_box(asdf_decoder_t* d, int32_t N, const float origin[3], float voxel_size, int32_t grid_mode, float tau,
                         float* scratch_hand_dev, float* scratch_obj_dev, int32_t* bbox_dev, void* stream) {
  if (!d || !origin || !bbox_dev || N < 2 || N > 1024 || !(tau > 0.0f) || !(tau < 0.5f)) return ASDF_EINVAL;
  if (grid_mode != ASDF_GRID_REFERENCE && grid_mode != ASDF_GRID_INTEGER) return ASDF_EINVAL;
  if (!d->stream16_hi || !d->sample_bound || !d->p1_usable) return ASDF_EINVAL;
  const bool two_out = d->spec.num_heads == 1;
  if (two_out ? !(scratch_hand_dev && scratch_obj_dev) : !(scratch_hand_dev || scratch_obj_dev)) return ASDF_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  DecodeParams p;
  std::memset(&p, 0, sizeof(p));
  p.sdf0 = scratch_hand_dev; p.sdf1 = scratch_obj_dev; p.bbox = bbox_dev;
  p.P = (long long)N * N * N; p.N = N; p.mode = grid_mode == ASDF_GRID_REFERENCE ? kGridReference : kGridInteger;
  p.vs = voxel_size; p.o0 = origin[0]; p.o1 = origin[1]; p.o2 = origin[2];
  p.stream = d->stream16_hi; p.cst = d->cst16p1; p.a16 = d->a16; p.status = d->status;
  p.first_mlp = 0; p.num_mlps = d->spec.num_heads; p.pf = d->spec.point_feats[0];
  if (!two_out) {
    if (!p.sdf1) p.num_mlps = 1;
    else if (!p.sdf0) { p.first_mlp = 1; p.num_mlps = 1; }
  }
  p.neg_thr = -tau;                                                   // the fused box takes the CERTAINLY negative voxels
  // the box record, the candidate count, status [1] candidates beyond the list / [2] contradiction flag / [3] largest
  // |exact - one-plane|, the audit record and the audit pick count (none of them is touched by the sweep kernel itself)
  hipLaunchKernelGGL(sweep_init_kernel, dim3(1), dim3(64), 0, st, p.bbox, ClearRange{d->near_count, 1}, ClearRange{d->status + 1, 3},
                     ClearRange{d->audit_rec, 12}, ClearRange{d->audit_count, 1});
  const long long ntiles = (p.P + kWgPts - 1) / kWgPts;
  const int grid = (int)(ntiles < d->num_cus ? ntiles : d->num_cus);
  if (d->ev_start) ASDF_HIP(hipEventRecord((hipEvent_t)d->ev_start, st));
  if (d->kp == 2) k1h_box_launch(two_out, p, grid, st);
  else k1s_nerf_launch(d->kp, two_out, p, grid, st);
  if (d->ev_stop) ASDF_HIP(hipEventRecord((hipEvent_t)d->ev_stop, st));
  d->ev_start = d->ev_stop = nullptr;
  // audit: voxels both heads decided by sign alone, drawn at random, through the split-half kernel (the arithmetic of the
  // ordinary sweep) - they report the error of the one-plane values where nothing else looks.  The picks are drawn here, in
  // order; their evaluation (a latency-bound launch on the small lattices: one 128-point tile per compute unit or less) runs on
  // the decoder's audit stream beside the candidates below - it reads the volume at the picks (never candidates: |v| >= tau) and
  // writes only the audit record, which nothing reads before the join in front of sweep_record_kernel
  bool audit_forked = false;
  // (a scope guard: whichever way this function is left behind the fork - an ASDF_HIP early return, a failed launch_subset - the
  // caller's stream waits for the audit's kernels, so the next call's sweep_init_kernel cannot clear the audit record under them:
  // ADVICE r05)
  struct AuditJoin {
    asdf_decoder* d; hipStream_t st; const bool* forked; bool done = false;
    void join() { if (*forked && !done) { done = true; (void)hipStreamWaitEvent(st, d->ev_audit_join, 0); } }
    ~AuditJoin() { join(); }
  } audit_join{d, st, &audit_forked};
  if (d->audit_n > 0) {
    { const int rc = enqueue_audit_picks(d, p.sdf0, p.sdf1, nullptr, p.P, tau, d->audit_idx, d->audit_count, kAuditCap, 0, st); if (rc != ASDF_OK) return rc; }
    DecodeParams a = p;
    a.stream = d->stream16; a.cst = d->cst16; a.bbox = nullptr; a.neg_thr = 0.0f;
    a.mode = kGridSubset; a.grid_mode = p.mode; a.idx = d->audit_idx; a.count_dev = d->audit_count; a.P = kAuditCap;
    a.audit = d->audit_rec; a.audit_from = nullptr;
    const int agrid = kAuditCap / kWgPts < d->num_cus ? kAuditCap / kWgPts : d->num_cus;
    hipStream_t as = st;
    if (d->audit_side && !std::getenv("ASDF_AUDIT_INLINE")) {
      ASDF_HIP(hipEventRecord(d->ev_audit_fork, st));
      ASDF_HIP(hipStreamWaitEvent(d->audit_side, d->ev_audit_fork, 0));
      as = d->audit_side;
      audit_forked = true;
    }
    k1h_subset_launch(d->kp, two_out, a, agrid, as);
    if (audit_forked && hipEventRecord(d->ev_audit_join, d->audit_side) != hipSuccess) {
      // no event to wait for: drain the side stream here, then report
      (void)hipStreamSynchronize(d->audit_side);
      audit_forked = false;
      return ASDF_EHIP;
    }
  }
  // candidates -> exact values (fp32 MFMA chain) -> the box is extended by every candidate that is negative
  const long long items = (N & 3) == 0 ? p.P / 4 : p.P;
  const int cgrid = (int)((items + 255) / 256 < 2048 ? (items + 255) / 256 : 2048);
  // A lattice of up to 8 x kCandDirect voxels (N <= 64) gets no two-step form: more than kCandDirect candidates there would mean an
  // eighth of the lattice within tau of the level - such a sweep is refused as a list overflow (status[1], the caller repeats it as an
  // ordinary sweep) and its five launches, which did nothing in every other sweep, are not enqueued (a 64^3 sample is ~40 launches)
  const bool direct_only = p.P <= 8LL * kCandDirect;
  const int cand_cap = direct_only ? kCandDirect : kCandCap;
  hipLaunchKernelGGL(collect_box_candidates_kernel, dim3(cgrid), dim3(256), 0, st, p.sdf0, p.sdf1, N, tau, p.bbox, d->near_idx,
                     d->near_count, cand_cap, d->status);
  if (!d->box_aux) ASDF_HIP(hipMalloc((void**)&d->box_aux, (4 + (size_t)kNearCap) * sizeof(int)));
  int* direct_count = d->box_aux, *twostep_count = d->box_aux + 1, *twostep_near = d->box_aux + 2, *twostep_idx = d->box_aux + 4;
  hipLaunchKernelGGL(split_candidate_count_kernel, dim3(1), dim3(64), 0, st, d->near_count, cand_cap, kCandDirect, d->box_aux);
  const int rgrid = kCandCap / kWgPts < d->num_cus ? kCandCap / kWgPts : d->num_cus;
  {
    // up to kCandDirect candidates: the fp32 chain patches values and boxes in one step
    DecodeParams q = p;
    q.stream = d->stream; q.cst = d->cst; q.fixup_flag = d->status + 2;
    q.mode = kGridSubset; q.grid_mode = p.mode; q.idx = d->near_idx; q.count_dev = direct_count; q.P = kCandCap;
    { const int rc = launch_subset(d, q, two_out, rgrid, st); if (rc != ASDF_OK) return rc; }
  }
  if (!direct_only) {
    // more: the values of the ordinary sweep (split-half kernel; it reports the largest |exact - one-plane| to status[3]) ...
    DecodeParams e = p;
    e.stream = d->stream16; e.cst = d->cst16; e.bbox = nullptr; e.neg_thr = 0.0f;
    e.mode = kGridSubset; e.grid_mode = p.mode; e.idx = d->near_idx; e.count_dev = twostep_count; e.P = kCandCap;
    e.audit = nullptr; e.audit_from = nullptr;
    k1h_subset_launch(d->kp, two_out, e, rgrid, st);
    if (d->refine_tau > 0.0f) {
      // ... the fp32 chain where they lie within refine_tau of the level ...
      const float* vols[2] = {p.sdf0, p.sdf1};
      for (int h = 0; h < 2; ++h)
        if (vols[h])
          hipLaunchKernelGGL(collect_near_level_list_kernel, dim3(256), dim3(256), 0, st, vols[h], d->near_idx, twostep_count, kCandCap,
                             d->refine_tau, twostep_idx, twostep_near, kNearCap, d->status);
      DecodeParams f = p;
      f.stream = d->stream; f.cst = d->cst; f.bbox = nullptr; f.neg_thr = 0.0f; f.status = nullptr;
      f.mode = kGridSubset; f.grid_mode = p.mode; f.idx = twostep_idx; f.count_dev = twostep_near; f.P = kNearCap;
      const int ngrid = kNearCap / kWgPts < d->num_cus ? kNearCap / kWgPts : d->num_cus;
      { const int rc = launch_subset(d, f, two_out, ngrid, st); if (rc != ASDF_OK) return rc; }
    }
    // ... and the boxes take every candidate that is negative
    hipLaunchKernelGGL(extend_box_from_list_kernel, dim3(256), dim3(256), 0, st, p.sdf0, p.sdf1, d->near_idx, twostep_count, N, p.bbox);
  }
  // the record of this call travels with the boxes: one read-back for the caller
  audit_join.join();
  hipLaunchKernelGGL(sweep_record_kernel, dim3(1), dim3(64), 0, st, bbox_dev, d->status, d->near_count, d->audit_rec);
  ASDF_HIP(hipGetLastError());
  return ASDF_OK;
}

static int decode_grid_band_impl(asdf_decoder_t* d, int32_t N, const float* origin, float voxel_size, const float* lattice_dev,
                                 int32_t grid_mode, float tau, float* sdf_hand_dev, float* sdf_obj_dev, int32_t* rec_dev, void* stream) {
  if (!d || (!origin && !lattice_dev) || !rec_dev || N < 2 || N > 1024 || !(tau > 0.0f) || !(tau < 0.5f)) return ASDF_EINVAL;
  if (grid_mode != ASDF_GRID_REFERENCE && grid_mode != ASDF_GRID_INTEGER) return ASDF_EINVAL;
  if (!d->stream16_hi || !d->sample_bound || !d->p1_usable) return ASDF_EINVAL;
  const bool two_out = d->spec.num_heads == 1;       // CombinedDecoder: one MLP, both columns from every evaluation
  if (two_out ? !(sdf_hand_dev && sdf_obj_dev) : (!sdf_hand_dev && !sdf_obj_dev)) return ASDF_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const long long P = (long long)N * N * N;
  if (!d->band_idx) {
    ASDF_HIP(hipMalloc((void**)&d->band_idx, 2 * (size_t)kBandCap * sizeof(int)));
    ASDF_HIP(hipMalloc((void**)&d->band_count, 2 * sizeof(int)));
  }
  if (d->band_mark_bytes < (size_t)P + 16) {
    if (d->band_mark) ASDF_HIP(hipFree(d->band_mark));
    d->band_mark = nullptr; d->band_mark_bytes = 0;
    ASDF_HIP(hipMalloc((void**)&d->band_mark, (size_t)P + 16));
    d->band_mark_bytes = (size_t)P + 16;
  }
  DecodeParams p;
  std::memset(&p, 0, sizeof(p));
  p.sdf0 = sdf_hand_dev; p.sdf1 = sdf_obj_dev; p.bbox = rec_dev;      // (the box words are by-products; 7 / 15 carry the range report)
  p.P = P; p.N = N; p.mode = grid_mode == ASDF_GRID_REFERENCE ? kGridReference : kGridInteger;
  if (lattice_dev) p.lattice = lattice_dev;        // (every subset launch below copies p: the listed voxels get the same lattice)
  else { p.vs = voxel_size; p.o0 = origin[0]; p.o1 = origin[1]; p.o2 = origin[2]; }
  p.stream = d->stream16_hi; p.cst = d->cst16p1; p.a16 = d->a16; p.status = d->status;
  p.first_mlp = 0; p.num_mlps = d->spec.num_heads; p.pf = d->spec.point_feats[0];
  if (!two_out) {
    if (!p.sdf1) p.num_mlps = 1;
    else if (!p.sdf0) { p.first_mlp = 1; p.num_mlps = 1; }
  }
  // the record, the two band counts, status [1] near-level voxels beyond the list / [3] largest |exact - one-plane| of this call,
  // the audit record and the near-level count
  hipLaunchKernelGGL(sweep_init_kernel, dim3(1), dim3(64), 0, st, p.bbox, ClearRange{d->band_count, 2}, ClearRange{d->status + 1, 3},
                     ClearRange{d->audit_rec, 12}, ClearRange{d->near_count, 1});
  const long long ntiles = (P + kWgPts - 1) / kWgPts;
  const int grid = (int)(ntiles < d->num_cus ? ntiles : d->num_cus);
  if (d->ev_start) ASDF_HIP(hipEventRecord((hipEvent_t)d->ev_start, st));
  if (d->kp == 2) k1h_box_launch(two_out, p, grid, st);
  else k1s_nerf_launch(d->kp, two_out, p, grid, st);
  if (d->ev_stop) ASDF_HIP(hipEventRecord((hipEvent_t)d->ev_stop, st));
  d->ev_start = d->ev_stop = nullptr;
  float* vols[2] = {sdf_hand_dev, sdf_obj_dev};
  // SeparateDecoder: one list per head (an MLP is evaluated for ITS marked voxels only).  CombinedDecoder (networks/model.py:149-188):
  // every evaluation yields both columns, so the corners of the cells that can be active in EITHER volume are marked into one list
  // (h = 0) and re-evaluated once.
  for (int h = 0; h < (two_out ? 1 : 2); ++h) {
    if (!vols[h]) continue;
    ASDF_HIP(hipMemsetAsync(d->band_mark, 0, (size_t)P, st));
    const int per_slab = (((N - 1) + 3) >> 2) * (N - 1);          // (y, group of four x-consecutive cells) pairs per z
    const dim3 mgrid((per_slab + 255) / 256, N - 1 < 1024 ? N - 1 : 1024);
    hipLaunchKernelGGL(band_mark_kernel, mgrid, dim3(256), 0, st, vols[h], N, tau, d->band_mark);
    if (two_out) hipLaunchKernelGGL(band_mark_kernel, mgrid, dim3(256), 0, st, vols[1], N, tau, d->band_mark);
    const long long items = (P + 15) / 16;
    const int cgrid = (int)((items + 255) / 256 < 4096 ? (items + 255) / 256 : 4096);
    int* list = d->band_idx + (size_t)h * kBandCap;
    // the audit picks of this head - unmarked voxels, i.e. voxels marching cubes will read the SIGN of and nothing else - ride
    // behind the marked ones in the same list (positions >= audit_rec[4 + h]: the compaction's last workgroup writes that word)
    // (... on the small lattices, where a launch is what counts.  On a large one the 4096 arrivals on one word cost more than the
    // 4-byte copy they replace - 19 -> 51 us per launch at N = 256 in the kernel statistics - so there the copy stays)
    const bool publish = cgrid <= 256;
    hipLaunchKernelGGL(band_compact_kernel, dim3(cgrid), dim3(256), 0, st, d->band_mark, P, list, d->band_count + h, kBandCap,
                       publish ? d->audit_rec + 10 + h : nullptr, d->audit_rec + 4 + h);
    if (!publish) ASDF_HIP(hipMemcpyAsync(d->audit_rec + 4 + h, d->band_count + h, sizeof(int), hipMemcpyDeviceToDevice, st));
    { const int rc = enqueue_audit_picks(d, vols[h], two_out ? vols[1] : nullptr, d->band_mark, P, tau, list, d->band_count + h, kBandCap, h, st); if (rc != ASDF_OK) return rc; }
    // the values of the ordinary sweep at the listed voxels of this head: the split-half kernel over the list ...
    DecodeParams q = p;
    q.stream = d->stream16; q.cst = d->cst16; q.bbox = nullptr; q.neg_thr = 0.0f;
    if (!two_out) { q.sdf0 = h == 0 ? vols[0] : nullptr; q.sdf1 = h == 1 ? vols[1] : nullptr; }
    q.first_mlp = two_out ? 0 : h; q.num_mlps = 1;
    q.mode = kGridSubset; q.grid_mode = p.mode; q.idx = list; q.count_dev = d->band_count + h; q.P = kBandCap;
    q.audit = d->audit_n > 0 ? d->audit_rec : nullptr; q.audit_from = d->audit_rec + 4 + h;
    const int rgrid = kBandCap / kWgPts < d->num_cus ? kBandCap / kWgPts : d->num_cus;
    k1h_subset_launch(d->kp, two_out, q, rgrid, st);
  }
  if (d->refine_tau > 0.0f) {
    // ... and, as behind every split-half sweep, the fp32 chain where those values lie within refine_tau of the level (both
    // MLPs over the union of the two near-level lists: an extra exact value is harmless)
    for (int h = 0; h < 2; ++h)
      if (vols[h]) {
        const int l = two_out ? 0 : h;       // (a CombinedDecoder's one list serves both volumes)
        hipLaunchKernelGGL(collect_near_level_list_kernel, dim3(256), dim3(256), 0, st, vols[h], d->band_idx + (size_t)l * kBandCap,
                           d->band_count + l, kBandCap, d->refine_tau, d->near_idx, d->near_count, kNearCap, d->status);
      }
    DecodeParams q = p;
    q.stream = d->stream; q.cst = d->cst; q.bbox = nullptr; q.neg_thr = 0.0f; q.status = nullptr;
    q.mode = kGridSubset; q.grid_mode = p.mode; q.idx = d->near_idx; q.count_dev = d->near_count; q.P = kNearCap;
    const int ngrid = kNearCap / kWgPts < d->num_cus ? kNearCap / kWgPts : d->num_cus;
    { const int rc = launch_subset(d, q, two_out, ngrid, st); if (rc != ASDF_OK) return rc; }
  }
  hipLaunchKernelGGL(sweep_record_kernel, dim3(1), dim3(64), 0, st, rec_dev, d->status, d->near_count, d->audit_rec);
  ASDF_HIP(hipGetLastError());
  return ASDF_OK;
}

int asdf_decode_grid_band(asdf_decoder_t* d, int32_t N, const float origin[3], float voxel_size, int32_t grid_mode, float tau,
                          float* sdf_hand_dev, float* sdf_obj_dev, int32_t* rec_dev, void* stream) {
  if (!origin) return ASDF_EINVAL;
  return decode_grid_band_impl(d, N, origin, voxel_size, nullptr, grid_mode, tau, sdf_hand_dev, sdf_obj_dev, rec_dev, stream);
}

int asdf_decode_grid_band_dev(asdf_decoder_t* d, int32_t N, const float* lattice_dev, int32_t grid_mode, float tau,
                              float* sdf_hand_dev, float* sdf_obj_dev, int32_t* rec_dev, void* stream) {
  if (!lattice_dev) return ASDF_EINVAL;
  return decode_grid_band_impl(d, N, nullptr, 0.0f, lattice_dev, grid_mode, tau, sdf_hand_dev, sdf_obj_dev, rec_dev, stream);
}

int asdf_decoder_one_plane_usable(const asdf_decoder_t* d) { return d && d->stream16_hi && d->p1_usable ? 1 : 0; }

int asdf_decoder_set_short_list(asdf_decoder_t* d, int32_t max_points) {
  if (!d || max_points < 0 || max_points > (1 << 16)) return ASDF_EINVAL;
  d->short_max = max_points;
  return ASDF_OK;
}

int asdf_decoder_set_cluster_list(asdf_decoder_t* d, int32_t max_points) {
  if (!d || max_points < 0 || max_points > kClusterCap) return ASDF_EINVAL;
  d->shortp.cluster_max = max_points;
  // switching the form (back) on withdraws an earlier fault report (status[11] is sticky otherwise: asdf_decoder_status leaves it)
  if (max_points > 0 && d->shortp.fault) { ASDF_HIP(hipDeviceSynchronize()); ASDF_HIP(hipMemset(d->shortp.fault, 0, sizeof(int))); }
  return ASDF_OK;
}

int asdf_decoder_set_cluster_timeout(asdf_decoder_t* d, uint64_t ticks) {
  if (!d) return ASDF_EINVAL;
  d->shortp.timeout_ticks = ticks ? ticks : kClusterTimeoutTicks;
  return ASDF_OK;
}

int asdf_decoder_set_audit(asdf_decoder_t* d, int32_t voxels, uint64_t seed) {
  if (!d || voxels < 0 || voxels > kAuditCap) return ASDF_EINVAL;
  d->audit_n = voxels;
  d->audit_seed = seed;
  return ASDF_OK;
}

int asdf_decode_points(asdf_decoder_t* d, const float* xyz_dev, int64_t M, float* sdf_hand_dev, float* sdf_obj_dev,
                       void* stream) {
  if (!d || M < 0 || (M > 0 && !xyz_dev)) return ASDF_EINVAL;
  DecodeParams p;
  std::memset(&p, 0, sizeof(p));
  p.sdf0 = sdf_hand_dev; p.sdf1 = sdf_obj_dev; p.xyz = xyz_dev; p.P = M; p.N = 1; p.mode = kPointList;
  return launch_decode(d, p, (hipStream_t)stream);
}

int asdf_decoder_set_math(asdf_decoder_t* d, int32_t math) {
  if (!d || (math != ASDF_MATH_F32 && math != ASDF_MATH_F16X3)) return ASDF_EINVAL;
  if (math == ASDF_MATH_F16X3 && !d->stream16) return ASDF_EINVAL;
  d->math = math;
  return ASDF_OK;
}

int asdf_decoder_get_math(const asdf_decoder_t* d) { return d ? d->math : ASDF_EINVAL; }

int asdf_decoder_time_next_sweep(asdf_decoder_t* d, void* event_start, void* event_stop) {
  if (!d || (!event_start) != (!event_stop)) return ASDF_EINVAL;
  d->ev_start = event_start;
  d->ev_stop = event_stop;
  return ASDF_OK;
}

int asdf_decoder_set_refine(asdf_decoder_t* d, float tau) {
  if (!d || !(tau >= 0.0f) || tau > 1.0f) return ASDF_EINVAL;
  d->refine_tau = tau;
  return ASDF_OK;
}

int asdf_decoder_status(asdf_decoder_t* d, int32_t out_host[16], int32_t clear, void* stream) {
  if (!d || !out_host) return ASDF_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  ASDF_HIP(hipMemcpyAsync(out_host, d->status, 16 * sizeof(int), hipMemcpyDeviceToHost, st));
  if (clear) {       // (word 11, the cluster form's fault report, is sticky: asdf_decoder_set_cluster_list withdraws it)
    ASDF_HIP(hipMemsetAsync(d->status, 0, 11 * sizeof(int), st));
    ASDF_HIP(hipMemsetAsync(d->status + 12, 0, 4 * sizeof(int), st));
  }
  ASDF_HIP(hipStreamSynchronize(st));
  return ASDF_OK;
}

int asdf_decoder_get_act_scales(const asdf_decoder_t* d, float sx_out[ASDF_MAX_HEADS][3]) {
  if (!d || !sx_out) return ASDF_EINVAL;
  std::memcpy(sx_out, d->sx, sizeof(d->sx));
  return ASDF_OK;
}

int asdf_decoder_set_act_scales(asdf_decoder_t* d, const float sx[ASDF_MAX_HEADS][3], void* stream) {
  if (!d || !sx || !d->cst16) return ASDF_EINVAL;
  for (int h = 0; h < d->spec.num_heads; ++h)
    for (int l = 0; l < 3; ++l) {
      int e = 0;
      const float m = std::frexp(sx[h][l], &e);
      if (!(sx[h][l] > 0.0f) || m != 0.5f || e < -23 || e > 25) return ASDF_EINVAL;      // powers of two in [2^-24, 2^24]
    }
  hipStream_t st = (hipStream_t)stream;
  ASDF_HIP(hipStreamSynchronize(st));                      // sweeps in flight still read the old constants
  std::memcpy(d->sx, sx, sizeof(d->sx));
  const size_t n = (size_t)kHeads * cst_offsets(d->kp).floats;
  std::memcpy(d->cst16_host, d->cst_host, n * sizeof(float));
  scale_constants_f16(d->spec, d->kp, d->cst_host, d->sw, d->sx, d->cst16_host, d->s2);
  ASDF_HIP(hipMemcpy(d->cst16, d->cst16_host, n * sizeof(float), hipMemcpyHostToDevice));
  ASDF_HIP(rebuild_one_plane(d));
  d->sample_bound = false;                                 // the folded per-sample constants carried the old layer-2 scale
  return ASDF_OK;
}

// Debug hook: the coordinates grid_point() produces, so that the in-kernel lattice can be compared bit for bit with the
// reference's (utils/mesh.py:27-40) - the decoder kernels call the very same device function.
namespace asdf {
__global__ void grid_coords_kernel(float* out, long long first, long long count, int N, int mode, float vs, float o0, float o1, float o2) {
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= count) return;
  float c0, c1, c2;
  grid_point(first + k, N, mode, vs, o0, o1, o2, c0, c1, c2);
  out[k * 3 + 0] = c0; out[k * 3 + 1] = c1; out[k * 3 + 2] = c2;
}
}  // namespace asdf

int asdf_debug_grid_coords(int32_t N, const float origin[3], float voxel_size, int32_t grid_mode, int64_t first, int64_t count,
                           float* coords_dev, void* stream) {
  if (!origin || !coords_dev || N < 2 || N > 1024 || first < 0 || count < 0 || first + count > (int64_t)N * N * N) return ASDF_EINVAL;
  if (grid_mode != ASDF_GRID_REFERENCE && grid_mode != ASDF_GRID_INTEGER) return ASDF_EINVAL;
  if (count == 0) return ASDF_OK;
  hipLaunchKernelGGL(grid_coords_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, coords_dev,
                     (long long)first, (long long)count, N, grid_mode == ASDF_GRID_REFERENCE ? kGridReference : kGridInteger,
                     voxel_size, origin[0], origin[1], origin[2]);
  ASDF_HIP(hipGetLastError());
  return ASDF_OK;
}

int asdf_decoder_set_classifier(asdf_decoder_t* d, const float* w_host, const float* b_host, int32_t num_class) {
  if (!d || !w_host || !b_host || num_class < 1 || num_class > kMaxClasses) return ASDF_EINVAL;
  std::vector<float> img(kClsFloats, 0.0f);
  for (int k = 0; k < num_class; ++k) {
    for (int row = 0; row < kHidden; ++row) {
      const int t = row >> 5, rr = row & 31, hh = (rr >> 2) & 1, r = (rr & 3) + 4 * (rr >> 3);   // D-layout order
      img[(size_t)k * kHidden + (t * 2 + hh) * 16 + r] = w_host[(size_t)k * kHidden + row];
    }
    img[(size_t)kMaxClasses * kHidden + k] = b_host[k];
  }
  if (!d->cls) ASDF_HIP(hipMalloc((void**)&d->cls, img.size() * sizeof(float)));
  ASDF_HIP(hipMemcpy(d->cls, img.data(), img.size() * sizeof(float), hipMemcpyHostToDevice));
  d->num_class = num_class;
  return ASDF_OK;
}

int asdf_decode_points_cls(asdf_decoder_t* d, const float* xyz_dev, int64_t M, float* sdf_hand_dev, float* sdf_obj_dev,
                           float* logits_dev, int32_t* labels_dev, void* stream) {
  if (!d || M < 0 || (M > 0 && !xyz_dev) || (!logits_dev && !labels_dev)) return ASDF_EINVAL;
  DecodeParams p;
  std::memset(&p, 0, sizeof(p));
  p.sdf0 = sdf_hand_dev; p.sdf1 = sdf_obj_dev; p.xyz = xyz_dev; p.P = M; p.N = 1; p.mode = kPointList;
  p.logits = logits_dev; p.labels = labels_dev;
  return launch_decode(d, p, (hipStream_t)stream);
}

}  // extern "C"
