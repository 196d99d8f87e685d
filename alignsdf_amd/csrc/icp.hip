// K7: translate + scale ICP of the reference's eval mode (deep_sdf/metrics/icp_trans_scale.py:33-113, ICP_T_S.run_icp_f,
// called from utils/mesh.py:385-395 with 30 000 surface samples per mesh and up to 100 iterations).
//
// Per iteration the reference does two exact nearest-neighbour sweeps through static KD-trees and a 4-unknown linear
// least-squares solve.  Here both sweeps are ONE brute-force launch in fp64 (the reference's arithmetic type, so the
// neighbour assignments are the KD-tree's) and the whole iteration stays on the device:
//   K7a icp_nn_kernel      (query tile x reference split) blocks: 2 queries per thread, the split's reference points
//                          streamed through LDS as SoA tiles (broadcast ds_read_b128), best (distance, index) per
//                          (query, split).  30k x 30k is only 118 query tiles; the 8-way reference split is what puts
//                          ~1900 waves on the 1024 SIMDs.
//   K7b icp_update_kernel  per query: first minimum over the splits, the squared-error and least-squares terms,
//                          block sums; the last block to finish adds the block sums in a fixed order, applies the
//                          reference's stopping rules and solves the 4 unknowns in closed form into the device state.
// The host enqueues iterations in batches and reads the 64-byte state between batches; kernels of iterations past
// convergence return at once.  VALU-bound (1.8e9 distance evaluations per iteration); HBM traffic is negligible.
#include <hip/hip_runtime.h>

#include <cmath>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "../../include/alignsdf_hip.h"
#include "common.h"

namespace asdf {

constexpr int kIcpThreads = 256;
constexpr int kIcpQpt = 2;              // queries per thread
constexpr int kIcpQBlock = kIcpThreads * kIcpQpt;
constexpr int kIcpTile = 1024;          // reference points per LDS tile (24 KiB of fp64, SoA)
constexpr int kIcpSplits = 8;           // reference-set splits per query tile
constexpr int kIcpSums = 9;             // err, sum X (3), sum Y (3), sum X.Y, sum X.X
constexpr int kIcpBatch = 8;            // iterations enqueued between two reads of the state
constexpr int kIcpUpdateGrid = 64;      // workgroups of the reduction kernel (K7b)

struct IcpState {
  double scale, t[3];
  double previous, error;
  int iters, done;
  unsigned ticket, pad;
};

// dir 0: queries = source samples p,  q = p * s + t,      reference set = target;  X = p,        Y = nearest target
// dir 1: queries = target samples P,  q = (P - t) / s,    reference set = source;  X = nearest p, Y = P
__global__ __launch_bounds__(kIcpThreads) void icp_nn_kernel(const double* __restrict__ src, int ns,
                                                             const double* __restrict__ tgt, int nt,
                                                             const IcpState* __restrict__ state, double* __restrict__ cand_d,
                                                             int* __restrict__ cand_i) {
  if (state->done) return;
  __shared__ __attribute__((aligned(16))) double tx[kIcpTile], ty[kIcpTile], tz[kIcpTile];
  const int qb_s = (ns + kIcpQBlock - 1) / kIcpQBlock;
  int b = blockIdx.x;
  const int dir = b >= qb_s * kIcpSplits;
  if (dir) b -= qb_s * kIcpSplits;
  const int split = b % kIcpSplits, qblock = b / kIcpSplits;
  const double* qpts = dir ? tgt : src;
  const double* rpts = dir ? src : tgt;
  const int nq = dir ? nt : ns, nr = dir ? ns : nt;
  const int chunk = (nr + kIcpSplits - 1) / kIcpSplits;
  const int r_lo = split * chunk, r_hi = min(nr, r_lo + chunk);
  const double s = state->scale, t0 = state->t[0], t1 = state->t[1], t2 = state->t[2];

  double q0[kIcpQpt], q1[kIcpQpt], q2[kIcpQpt], best[kIcpQpt];
  int bidx[kIcpQpt];
#pragma unroll
  for (int k = 0; k < kIcpQpt; ++k) {
    const int i = qblock * kIcpQBlock + k * kIcpThreads + threadIdx.x;
    double p0 = 0, p1 = 0, p2 = 0;
    if (i < nq) { p0 = qpts[3 * (size_t)i]; p1 = qpts[3 * (size_t)i + 1]; p2 = qpts[3 * (size_t)i + 2]; }
    if (dir == 0) { q0[k] = p0 * s + t0; q1[k] = p1 * s + t1; q2[k] = p2 * s + t2; }
    else { q0[k] = (p0 - t0) / s; q1[k] = (p1 - t1) / s; q2[k] = (p2 - t2) / s; }
    best[k] = INFINITY;
    bidx[k] = r_lo;
  }
  for (int base = r_lo; base < r_hi; base += kIcpTile) {
    const int n = min(kIcpTile, r_hi - base);
    __syncthreads();
    for (int k = threadIdx.x; k < n; k += kIcpThreads) {
      tx[k] = rpts[3 * (size_t)(base + k)]; ty[k] = rpts[3 * (size_t)(base + k) + 1]; tz[k] = rpts[3 * (size_t)(base + k) + 2];
    }
    __syncthreads();
#pragma unroll 4
    for (int j = 0; j < n; ++j) {
      const double r0 = tx[j], r1 = ty[j], r2 = tz[j];
#pragma unroll
      for (int k = 0; k < kIcpQpt; ++k) {
        const double d0 = q0[k] - r0, d1 = q1[k] - r1, d2 = q2[k] - r2;
        const double d = d0 * d0 + d1 * d1 + d2 * d2;
        if (d < best[k]) { best[k] = d; bidx[k] = base + j; }      // first minimum wins on exact ties
      }
    }
  }
  const size_t row = (size_t)split * ((size_t)ns + nt) + (dir ? ns : 0);
#pragma unroll
  for (int k = 0; k < kIcpQpt; ++k) {
    const int i = qblock * kIcpQBlock + k * kIcpThreads + threadIdx.x;
    if (i < nq) { cand_d[row + i] = best[k]; cand_i[row + i] = bidx[k]; }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Exact nearest neighbours through a uniform grid (round 3).  Both reference sets of the iteration are STATIC - the target
// for the source's queries, and the untransformed source for the target's queries (a uniform scale + translation of the
// reference set is the inverse transform of the query: q = (P - t) / s, as the brute-force kernel already does) - so each
// gets a grid once per run: cells of edge h over its bounding box, the points counting-sorted by cell.  A query walks the
// Chebyshev shells around its cell and stops as soon as the best distance found is smaller than the distance to the nearest
// face of the explored block that still has unexplored cells behind it.  The result is the brute-force result: the same fp64
// expression per pair, and on exact ties the lowest index, compared explicitly (the brute-force kernel gets that from its scan
// order).  30k x 30k: ~60-300 distance evaluations per query instead of 30 000.
constexpr int kGridMaxRes = 64;
constexpr int kGridMaxCells = kGridMaxRes * kGridMaxRes * kGridMaxRes;

struct IcpGrid {
  double lo[3], h, inv_h;
  int g[3], ncell;
};

__global__ __launch_bounds__(1024) void grid_bbox_kernel(const double* __restrict__ pts, int n, IcpGrid* grid) {
  __shared__ double smin[3][16], smax[3][16];
  double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = threadIdx.x; i < n; i += blockDim.x)
#pragma unroll
    for (int a = 0; a < 3; ++a) { const double v = pts[3 * (size_t)i + a]; mn[a] = fmin(mn[a], v); mx[a] = fmax(mx[a], v); }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { mn[a] = fmin(mn[a], __shfl_xor(mn[a], m)); mx[a] = fmax(mx[a], __shfl_xor(mx[a], m)); }
    if ((threadIdx.x & 63) == 0) { smin[a][threadIdx.x >> 6] = mn[a]; smax[a][threadIdx.x >> 6] = mx[a]; }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ext = 0.0;
    for (int a = 0; a < 3; ++a) {
      for (int k = 1; k < (int)(blockDim.x >> 6); ++k) { smin[a][0] = fmin(smin[a][0], smin[a][k]); smax[a][0] = fmax(smax[a][0], smax[a][k]); }
      grid->lo[a] = smin[a][0];
      ext = fmax(ext, smax[a][0] - smin[a][0]);
    }
    // resolution: ~1.5 cbrt(n) cells along the longest axis (surface samples fill a 2-d sheet of them: a few points per occupied cell)
    int res = (int)(1.5 * cbrt((double)n) + 0.5);                  // (grid_resolution on the host: the workspace holds (res + 1)^3 cells)
    res = res < 4 ? 4 : (res > kGridMaxRes ? kGridMaxRes : res);
    double h = ext > 0.0 ? ext / res : 1.0;
    h *= 1.0 + 1e-9;                                     // the maximum lands inside the last cell
    grid->h = h; grid->inv_h = 1.0 / h;
    int ncell = 1;
    for (int a = 0; a < 3; ++a) {
      int ga = (int)floor((smax[a][0] - smin[a][0]) * grid->inv_h) + 1;
      ga = ga < 1 ? 1 : (ga > res + 1 ? res + 1 : ga);
      grid->g[a] = ga; ncell *= ga;
    }
    grid->ncell = ncell;
  }
}

__device__ __forceinline__ int grid_cell_axis(const IcpGrid* g, int a, double v) {
  int c = (int)floor((v - g->lo[a]) * g->inv_h);
  return c < 0 ? 0 : (c >= g->g[a] ? g->g[a] - 1 : c);
}

__global__ __launch_bounds__(256) void grid_count_kernel(const double* __restrict__ pts, int n, const IcpGrid* grid, int* cell_of, int* counts) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c0 = grid_cell_axis(grid, 0, pts[3 * (size_t)i]), c1 = grid_cell_axis(grid, 1, pts[3 * (size_t)i + 1]),
            c2 = grid_cell_axis(grid, 2, pts[3 * (size_t)i + 2]);
  const int c = (c0 * grid->g[1] + c1) * grid->g[2] + c2;
  cell_of[i] = c;
  atomicAdd(counts + c, 1);
}

// exclusive scan of the cell counts (<= 262 144 cells) by one workgroup; cursor = a second copy for the scatter
__global__ __launch_bounds__(1024) void grid_scan_kernel(const IcpGrid* grid, const int* counts, int* starts, int* cursor) {
  __shared__ int wsum[16];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  const int ncell = grid->ncell, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  // four consecutive cells per thread as ONE 16-byte load (coalesced): 110 k cells are 27 rounds of this one workgroup, not 108.
  // (counts is zeroed up to the workspace's capacity, a multiple of 64 ints: reading the vector that straddles ncell is in bounds)
  for (int base = 0; base < ncell; base += 4096) {
    const int i0 = base + 4 * threadIdx.x;
    int4 v = make_int4(0, 0, 0, 0);
    if (i0 < ncell) v = *reinterpret_cast<const int4*>(counts + i0);
    if (i0 + 1 >= ncell) v.y = 0;
    if (i0 + 2 >= ncell) v.z = 0;
    if (i0 + 3 >= ncell) v.w = 0;
    const int run = v.x + v.y + v.z + v.w;
    int x = run;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) { const int y = __shfl_up(x, m); if (lane >= m) x += y; }
    if (lane == 63) wsum[w] = x;
    __syncthreads();
    int off = carry + x - run;
    for (int k = 0; k < w; ++k) off += wsum[k];
    const int4 o = make_int4(off, off + v.x, off + v.x + v.y, off + v.x + v.y + v.z);
    if (i0 + 3 < ncell) {
      *reinterpret_cast<int4*>(starts + i0) = o;
      *reinterpret_cast<int4*>(cursor + i0) = o;
    } else if (i0 < ncell) {
      const int e[4] = {o.x, o.y, o.z, o.w};
      for (int k = 0; k < 4 && i0 + k < ncell; ++k) { starts[i0 + k] = e[k]; cursor[i0 + k] = e[k]; }
    }
    __syncthreads();
    if (threadIdx.x == 1023) carry = off + run;
    __syncthreads();
  }
  if (threadIdx.x == 0) starts[ncell] = carry;
}

__global__ __launch_bounds__(256) void grid_scatter_kernel(const double* __restrict__ pts, int n, const int* __restrict__ cell_of, int* cursor,
                                                           double* sorted, int* sorted_idx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int at = atomicAdd(cursor + cell_of[i], 1);
  sorted[3 * (size_t)at] = pts[3 * (size_t)i]; sorted[3 * (size_t)at + 1] = pts[3 * (size_t)i + 1]; sorted[3 * (size_t)at + 2] = pts[3 * (size_t)i + 2];
  sorted_idx[at] = i;
}

struct IcpGridRef {
  const IcpGrid* grid;
  const int* starts;
  const double* sorted;
  const int* sorted_idx;
};

// kGridLanes lanes per query (the scattered loads of a grid walk are latency-bound: 60 000 queries at one lane each are one wave
// per SIMD); row 0 of the candidate arrays receives the result.  Cells along axis 2 are contiguous in the sorted array, so a
// walk visits ROWS: two loads for a row's point range, then consecutive points.
//   1. the 3 x 3 rows around the query's cell (then 5 x 5, 9 x 9, ... while nothing has been found);
//   2. with D = the best distance so far, every point within D of the query lies in the cells
//      [cell(q - D), cell(q + D)] per axis: if that box is not inside the block already scanned, scan the box.
constexpr int kGridLanes = 8;
__global__ __launch_bounds__(kIcpThreads) void icp_nn_grid_kernel(const double* __restrict__ src, int ns, const double* __restrict__ tgt, int nt,
                                                                  const IcpState* __restrict__ state, IcpGridRef gt, IcpGridRef gs,
                                                                  double* __restrict__ cand_d, int* __restrict__ cand_i) {
  if (state->done) return;
  const int t = blockIdx.x * kIcpThreads + threadIdx.x;
  const int gidx = t / kGridLanes, sub = t % kGridLanes;
  const bool valid = gidx < ns + nt;              // (all lanes of a query agree; invalid lanes walk query 0 and write nothing)
  const int qid = valid ? gidx : 0;
  const int dir = qid >= ns;
  // queries are taken in the CELL-SORTED order of their own set's grid (the source set has one for the target's queries, and vice
  // versa): the lanes of a wave then walk neighbouring cells of the other grid - similar trip counts, shared cache lines - instead
  // of eight unrelated places each (the samples are in random order).  Results go to the query's own slot: nothing downstream changes
  const int k = dir ? qid - ns : qid;
  const int i = (dir ? gt : gs).sorted_idx[k];
  const double* pp = (dir ? gt : gs).sorted + 3 * (size_t)k;
  const double s = state->scale, t0 = state->t[0], t1 = state->t[1], t2 = state->t[2];
  double q[3];
  if (dir == 0) { q[0] = pp[0] * s + t0; q[1] = pp[1] * s + t1; q[2] = pp[2] * s + t2; }
  else { q[0] = (pp[0] - t0) / s; q[1] = (pp[1] - t1) / s; q[2] = (pp[2] - t2) / s; }
  const IcpGridRef R = dir ? gs : gt;
  const IcpGrid* G = R.grid;
  const int g0 = G->g[0], g1 = G->g[1], g2 = G->g[2];
  const double h = G->h;
  int c[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) c[a] = grid_cell_axis(G, a, q[a]);
  double best = INFINITY;
  int bidx = 0x7fffffff;
  // the rows (z, y) of a cell box, dealt round-robin to the lanes of the query; then the lanes agree on the minimum
  auto scan_box = [&](int z_lo, int z_hi, int y_lo, int y_hi, int x_lo, int x_hi) {
    const int ny = y_hi - y_lo + 1, nrows = (z_hi - z_lo + 1) * ny;
    for (int row = sub; row < nrows; row += kGridLanes) {
      const int z = z_lo + row / ny, y = y_lo + row % ny;
      const int cell0 = (z * g1 + y) * g2 + x_lo;
      const int b = R.starts[cell0], e = R.starts[cell0 + (x_hi - x_lo) + 1];
      for (int k = b; k < e; ++k) {
        const double d0 = q[0] - R.sorted[3 * (size_t)k], d1 = q[1] - R.sorted[3 * (size_t)k + 1], d2 = q[2] - R.sorted[3 * (size_t)k + 2];
        const double d = d0 * d0 + d1 * d1 + d2 * d2;
        const int id = R.sorted_idx[k];
        if (d < best || (d == best && id < bidx)) { best = d; bidx = id; }      // first minimum of the brute-force scan = lowest index
      }
    }
#pragma unroll
    for (int m = 1; m < kGridLanes; m <<= 1) {
      const double od = __shfl_xor(best, m);
      const int oi = __shfl_xor(bidx, m);
      if (od < best || (od == best && oi < bidx)) { best = od; bidx = oi; }
    }
  };
  int r = 1;
  int lo[3], hi[3];
  for (;;) {
    lo[0] = max(c[0] - r, 0); hi[0] = min(c[0] + r, g0 - 1);
    lo[1] = max(c[1] - r, 0); hi[1] = min(c[1] + r, g1 - 1);
    lo[2] = max(c[2] - r, 0); hi[2] = min(c[2] + r, g2 - 1);
    scan_box(lo[0], hi[0], lo[1], hi[1], lo[2], hi[2]);
    if (best < INFINITY || (lo[0] == 0 && hi[0] == g0 - 1 && lo[1] == 0 && hi[1] == g1 - 1 && lo[2] == 0 && hi[2] == g2 - 1)) break;
    r *= 2;
  }
  if (best < INFINITY) {
    const double D = sqrt(best) * (1.0 + 1e-12) + 1e-7 * h;      // (cell assignment rounds: a hair of slack)
    int blo[3], bhi[3];
    bool inside = true;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      blo[a] = grid_cell_axis(G, a, q[a] - D);
      bhi[a] = grid_cell_axis(G, a, q[a] + D);
      inside = inside && blo[a] >= lo[a] && bhi[a] <= hi[a];
    }
    if (!inside) scan_box(blo[0], bhi[0], blo[1], bhi[1], blo[2], bhi[2]);
  }
  if (valid && sub == 0) { const int slot = dir ? ns + i : i; cand_d[slot] = best; cand_i[slot] = bidx; }
}

__global__ __launch_bounds__(kIcpThreads) void icp_update_kernel(const double* __restrict__ src, int ns,
                                                                 const double* __restrict__ tgt, int nt, IcpState* state,
                                                                 const double* __restrict__ cand_d, const int* __restrict__ cand_i,
                                                                 double* partials, int iteration, double stop_error,
                                                                 double stop_improvement, int nsplits) {
  if (state->done) return;
  __shared__ double red[kIcpThreads / 64][kIcpSums];
  __shared__ bool last;
  const int total = ns + nt;
  const double s = state->scale, t0 = state->t[0], t1 = state->t[1], t2 = state->t[2];
  double v[kIcpSums] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  // at most kIcpUpdateGrid workgroups, every thread adds its queries in index order (a fixed order: the sums are reproducible).  It was
  // one workgroup per 256 queries - 235 of them at 30k + 30k, each paying an L2 write-back for its fence before the ticket: 61 us per
  // launch, more than the nearest-neighbour search it follows (round 5)
  for (int g = blockIdx.x * kIcpThreads + threadIdx.x; g < total; g += gridDim.x * kIcpThreads) {
    const int dir = g >= ns;
    const int i = dir ? g - ns : g;
    double best = cand_d[g];
    int idx = cand_i[g];
    for (int r = 1; r < nsplits; ++r) {
      const double d = cand_d[(size_t)r * total + g];
      if (d < best) { best = d; idx = cand_i[(size_t)r * total + g]; }    // splits are in index order: first minimum
    }
    const double* pp = (dir ? tgt : src) + 3 * (size_t)i;
    const double* bb = (dir ? src : tgt) + 3 * (size_t)idx;
    const double p0 = pp[0], p1 = pp[1], p2 = pp[2], b0 = bb[0], b1 = bb[1], b2 = bb[2];
    if (dir == 0) {
      v[0] += best;                                  // |q - ct|^2
      v[1] += p0; v[2] += p1; v[3] += p2; v[4] += b0; v[5] += b1; v[6] += b2;
      v[7] += p0 * b0 + p1 * b1 + p2 * b2; v[8] += p0 * p0 + p1 * p1 + p2 * p2;
    } else {
      const double c0 = b0 * s + t0, c1 = b1 * s + t1, c2 = b2 * s + t2;     // nearest source sample, transformed
      v[0] += (p0 - c0) * (p0 - c0) + (p1 - c1) * (p1 - c1) + (p2 - c2) * (p2 - c2);
      v[1] += b0; v[2] += b1; v[3] += b2; v[4] += p0; v[5] += p1; v[6] += p2;
      v[7] += b0 * p0 + b1 * p1 + b2 * p2; v[8] += b0 * b0 + b1 * b1 + b2 * b2;
    }
  }
#pragma unroll
  for (int k = 0; k < kIcpSums; ++k) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v[k] += __shfl_xor(v[k], m);
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0)
    for (int k = 0; k < kIcpSums; ++k) red[w][k] = v[k];
  __syncthreads();
  if (threadIdx.x < kIcpSums) {
    double a = 0;
    for (int k = 0; k < kIcpThreads / 64; ++k) a += red[k][threadIdx.x];
    partials[(size_t)blockIdx.x * kIcpSums + threadIdx.x] = a;
  }
  // the last block to arrive closes the iteration (the partial sums were written by wave 0: its fence, then the ticket)
  if (threadIdx.x < 64) __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) last = atomicAdd(&state->ticket, 1u) == gridDim.x - 1;
  __syncthreads();
  if (!last) return;
  if (threadIdx.x < 64) {
    // the block sums (at most kIcpUpdateGrid = 64 of them: one per lane), added by a fixed butterfly - the same tree every run.
    // (nine serial loops over the partial sums by nine threads were 19 of the kernel's 20 us)
    __threadfence();
    static_assert(kIcpUpdateGrid <= 64, "one partial sum per lane");
#pragma unroll
    for (int k = 0; k < kIcpSums; ++k) {
      double a = threadIdx.x < gridDim.x ? ((volatile double*)partials)[(size_t)threadIdx.x * kIcpSums + k] : 0.0;
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) a += __shfl_xor(a, m);
      if (threadIdx.x == 0) red[0][k] = a;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double* sum = red[0];
    const double n = (double)ns + (double)nt;
    const double error = sqrt(sum[0] / n);
    state->error = error;
    state->iters = iteration + 1;
    state->ticket = 0;
    // stopping rules of run_icp_f (:66-72)
    if (state->previous - error < stop_improvement) { state->done = 1; return; }
    state->previous = error;
    if (error < stop_error) { state->done = 1; return; }
    // argmin_{s,t} sum |s X + t - Y|^2 over the stacked system (:76-107)
    const double xm[3] = {sum[1] / n, sum[2] / n, sum[3] / n}, ym[3] = {sum[4] / n, sum[5] / n, sum[6] / n};
    const double num = sum[7] - n * (xm[0] * ym[0] + xm[1] * ym[1] + xm[2] * ym[2]);
    const double den = sum[8] - n * (xm[0] * xm[0] + xm[1] * xm[1] + xm[2] * xm[2]);
    const double scale = num / den;
    state->scale = scale;
    for (int k = 0; k < 3; ++k) state->t[k] = ym[k] - scale * xm[k];
  }
}

// Symmetric Chamfer distance of compute_trimesh_chamfer (deep_sdf/metrics/chamfer.py:217-229) on top of K7a's
// candidates (identity transform): mean squared nearest-neighbour distance per direction.  Blocks [0, blocks_a) own the
// a -> b queries, the rest the b -> a queries; the last block adds the block sums in a fixed order.
__global__ __launch_bounds__(kIcpThreads) void chamfer_reduce_kernel(int na, int nb, IcpState* state,
                                                                     const double* __restrict__ cand_d, double* partials,
                                                                     double* out, int nsplits) {
  __shared__ double red[kIcpThreads / 64];
  __shared__ bool last;
  const int blocks_a = (na + kIcpThreads - 1) / kIcpThreads;
  const int dir = (int)blockIdx.x >= blocks_a;
  const int i = (dir ? (int)blockIdx.x - blocks_a : (int)blockIdx.x) * kIcpThreads + threadIdx.x;
  const int total = na + nb;
  double v = 0.0;
  if (i < (dir ? nb : na)) {
    const int g = dir ? na + i : i;
    double best = cand_d[g];
    for (int r = 1; r < nsplits; ++r) best = fmin(best, cand_d[(size_t)r * total + g]);
    v = best;
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0;
    for (int k = 0; k < kIcpThreads / 64; ++k) a += red[k];
    partials[blockIdx.x] = a;
  }
  if (threadIdx.x < 64) __threadfence();           // (the writing wave's fence only: each one is an L2 write-back)
  __syncthreads();
  if (threadIdx.x == 0) last = atomicAdd(&state->ticket, 1u) == gridDim.x - 1;
  __syncthreads();
  if (!last) return;
  if (threadIdx.x < 64) __threadfence();
  if (threadIdx.x < 2) {
    const int lo = threadIdx.x ? blocks_a : 0, hi = threadIdx.x ? (int)gridDim.x : blocks_a;
    double a = 0;
    for (int b = lo; b < hi; ++b) a += ((volatile double*)partials)[b];
    out[threadIdx.x] = a / (double)(threadIdx.x ? nb : na);
    if (threadIdx.x == 0) state->ticket = 0;
  }
}

__global__ void icp_init_kernel(IcpState* state, const IcpState init) { *state = init; }

// result layout of the C ABI, written straight into device-accessible host memory
__global__ void icp_publish_kernel(const IcpState* state, double* out) {
  out[0] = state->scale; out[1] = state->t[0]; out[2] = state->t[1]; out[3] = state->t[2];
  out[4] = (double)state->iters; out[5] = state->error;
  out[6] = (double)state->done;          // 0 = the stopping rules have not fired yet (asdf_icp_ts_enqueue_range)
}

}  // namespace asdf

using namespace asdf;

namespace {
int g_icp_search = 0;       // asdf_icp_set_search: 0 = grid when both sets have >= 1024 points, 1 = brute force, 2 = grid

struct IcpGridLayout {      // byte offsets of one set's grid inside the workspace
  size_t hdr, cell_of, counts, starts, cursor, sorted, sorted_idx;
  int ncell_max;
};
// upper bound of the cell count grid_bbox_kernel can choose for n points: (resolution + 1)^3
int grid_resolution(int n) {
  int res = (int)(1.5 * std::cbrt((double)n) + 0.5);
  return res < 4 ? 4 : (res > kGridMaxRes ? kGridMaxRes : res);
}
size_t grid_max_cells(int n) {
  const size_t r = (size_t)grid_resolution(n) + 1;
  return r * r * r;
}
struct IcpLayout {
  size_t state, partials, cand_d, cand_i, bytes;
  IcpGridLayout grid[2];    // [0] = grid of the target set (queries: source), [1] = grid of the source set
  int update_blocks, nn_blocks;
};
size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
bool use_grid(int ns, int nt) { return g_icp_search == 2 || (g_icp_search == 0 && ns >= 1024 && nt >= 1024); }
// The search mode of a run is fixed when the run BEGINS and remembered per workspace on the host: a continuation
// (asdf_icp_ts_enqueue_range with first_iter > 0) attaches with the layout and the kernels its run was begun with, whatever
// asdf_icp_set_search has been told in between (the process-wide setting used to be re-read on every attach).
std::mutex g_run_mu;
std::unordered_map<const void*, bool> g_run_grid;
// An entry lives from icp_begin until the run's outcome is read (asdf_icp_ts_result / the end of asdf_icp_ts / asdf_chamfer) - it does
// not outlive the run, so a recycled workspace address cannot inherit a stale mode and the map does not grow with every address the
// caching allocator hands out; a run that is never read is dropped when the map passes 4096 entries.
void remember_run(const void* ws, bool grid) {
  std::lock_guard<std::mutex> g(g_run_mu);
  if (g_run_grid.size() > 4096) g_run_grid.clear();
  g_run_grid[ws] = grid;
}
void forget_run(const void* ws) { std::lock_guard<std::mutex> g(g_run_mu); g_run_grid.erase(ws); }
bool run_uses_grid(const void* ws, int ns, int nt) {
  std::lock_guard<std::mutex> g(g_run_mu);
  auto it = g_run_grid.find(ws);
  return it != g_run_grid.end() ? it->second : use_grid(ns, nt);
}
IcpLayout icp_layout(int ns, int nt, bool grid) {
  IcpLayout l;
  const size_t total = (size_t)ns + nt;
  l.update_blocks = (int)((total + kIcpThreads - 1) / kIcpThreads);
  l.nn_blocks = ((ns + kIcpQBlock - 1) / kIcpQBlock + (nt + kIcpQBlock - 1) / kIcpQBlock) * kIcpSplits;
  l.state = 0;
  l.partials = 256;
  l.cand_d = l.partials + (size_t)l.update_blocks * kIcpSums * sizeof(double);
  l.cand_i = l.cand_d + (size_t)kIcpSplits * total * sizeof(double);
  size_t at = align256(l.cand_i + (size_t)kIcpSplits * total * sizeof(int));
  for (int k = 0; k < 2 && grid; ++k) {
    const size_t n = k == 0 ? nt : ns;
    IcpGridLayout& g = l.grid[k];
    g.hdr = at; at = align256(at + sizeof(IcpGrid));
    g.cell_of = at; at = align256(at + n * sizeof(int));
    const size_t cells = grid_max_cells((int)n);
    g.ncell_max = (int)cells;
    g.counts = at; at = align256(at + cells * sizeof(int));
    g.starts = at; at = align256(at + (cells + 1) * sizeof(int));
    g.cursor = at; at = align256(at + cells * sizeof(int));
    g.sorted = at; at = align256(at + 3 * n * sizeof(double));
    g.sorted_idx = at; at = align256(at + n * sizeof(int));
  }
  l.bytes = at;
  return l;
}
}  // namespace

extern "C" {

int asdf_icp_set_search(int32_t mode) {
  if (mode < 0 || mode > 2) return ASDF_EINVAL;
  g_icp_search = mode;
  return ASDF_OK;
}

int asdf_icp_workspace_bytes(int32_t ns, int32_t nt, size_t* bytes) {
  if (!bytes || ns < 1 || nt < 1) return ASDF_EINVAL;
  *bytes = icp_layout(ns, nt, use_grid(ns, nt)).bytes;
  return ASDF_OK;
}

namespace {
struct IcpRun {
  IcpLayout l;
  IcpState* state;
  double* partials;
  double* cand_d;
  int* cand_i;
  bool grid;
  IcpGridRef gt, gs;
};
// counting-sort one reference set into its grid (five small launches, once per run)
int build_grid(char* ws, const IcpGridLayout& g, const double* pts, int n, hipStream_t st, IcpGridRef& ref) {
  IcpGrid* hdr = (IcpGrid*)(ws + g.hdr);
  int* counts = (int*)(ws + g.counts);
  ASDF_HIP(hipMemsetAsync(counts, 0, (size_t)g.ncell_max * sizeof(int), st));
  hipLaunchKernelGGL(grid_bbox_kernel, dim3(1), dim3(1024), 0, st, pts, n, hdr);
  const int blocks = (n + 255) / 256;
  hipLaunchKernelGGL(grid_count_kernel, dim3(blocks), dim3(256), 0, st, pts, n, hdr, (int*)(ws + g.cell_of), counts);
  hipLaunchKernelGGL(grid_scan_kernel, dim3(1), dim3(1024), 0, st, hdr, counts, (int*)(ws + g.starts), (int*)(ws + g.cursor));
  hipLaunchKernelGGL(grid_scatter_kernel, dim3(blocks), dim3(256), 0, st, pts, n, (const int*)(ws + g.cell_of), (int*)(ws + g.cursor),
                     (double*)(ws + g.sorted), (int*)(ws + g.sorted_idx));
  ASDF_HIP(hipGetLastError());
  ref.grid = hdr; ref.starts = (const int*)(ws + g.starts); ref.sorted = (const double*)(ws + g.sorted); ref.sorted_idx = (const int*)(ws + g.sorted_idx);
  return ASDF_OK;
}
void grid_ref(char* ws, const IcpGridLayout& g, IcpGridRef& ref) {
  ref.grid = (const IcpGrid*)(ws + g.hdr); ref.starts = (const int*)(ws + g.starts); ref.sorted = (const double*)(ws + g.sorted);
  ref.sorted_idx = (const int*)(ws + g.sorted_idx);
}
// the pointers of a run inside its workspace (no device work): a run begun earlier is continued through this
int icp_attach(int32_t ns, int32_t nt, void* workspace_dev, size_t workspace_bytes, IcpRun& r, bool grid) {
  if (!workspace_dev || ns < 1 || nt < 1) return ASDF_EINVAL;
  r.l = icp_layout(ns, nt, grid);
  if (workspace_bytes < r.l.bytes) return ASDF_ENOSPC;
  char* ws = (char*)workspace_dev;
  r.state = (IcpState*)(ws + r.l.state);
  r.partials = (double*)(ws + r.l.partials);
  r.cand_d = (double*)(ws + r.l.cand_d);
  r.cand_i = (int*)(ws + r.l.cand_i);
  r.grid = grid;
  if (r.grid) { grid_ref(ws, r.l.grid[0], r.gt); grid_ref(ws, r.l.grid[1], r.gs); }
  return ASDF_OK;
}
int icp_begin(const double* src_dev, int32_t ns, const double* tgt_dev, int32_t nt, int32_t max_iter, void* workspace_dev,
              size_t workspace_bytes, hipStream_t st, IcpRun& r) {
  if (!src_dev || !tgt_dev || !workspace_dev || ns < 1 || nt < 1 || max_iter < 1) return ASDF_EINVAL;
  const bool grid = use_grid(ns, nt);            // the setting in force NOW belongs to this run until its workspace begins another
  { const int rc = icp_attach(ns, nt, workspace_dev, workspace_bytes, r, grid); if (rc != ASDF_OK) return rc; }
  remember_run(workspace_dev, grid);
  char* ws = (char*)workspace_dev;
  // the initial state is a kernel argument of a fill, not a host buffer: nothing the caller must keep alive
  IcpState h;
  h.scale = 1.0; h.t[0] = h.t[1] = h.t[2] = 0.0;
  h.previous = 1e8; h.error = 1e8;
  h.iters = 0; h.done = 0; h.ticket = 0; h.pad = 0;
  hipLaunchKernelGGL(icp_init_kernel, dim3(1), dim3(1), 0, st, r.state, h);
  ASDF_HIP(hipGetLastError());
  if (r.grid) {
    int rc = build_grid(ws, r.l.grid[0], tgt_dev, nt, st, r.gt);
    if (rc == ASDF_OK) rc = build_grid(ws, r.l.grid[1], src_dev, ns, st, r.gs);
    if (rc != ASDF_OK) return rc;
  }
  return ASDF_OK;
}
void icp_nn(const IcpRun& r, const double* src_dev, int ns, const double* tgt_dev, int nt, hipStream_t st) {
  if (r.grid)
    hipLaunchKernelGGL(icp_nn_grid_kernel, dim3((unsigned)(((size_t)ns + nt) * kGridLanes + kIcpThreads - 1) / kIcpThreads), dim3(kIcpThreads), 0, st, src_dev, ns, tgt_dev, nt, r.state, r.gt, r.gs,
                       r.cand_d, r.cand_i);
  else
    hipLaunchKernelGGL(icp_nn_kernel, dim3(r.l.nn_blocks), dim3(kIcpThreads), 0, st, src_dev, ns, tgt_dev, nt, r.state, r.cand_d, r.cand_i);
}
void icp_iterations(const IcpRun& r, const double* src_dev, int ns, const double* tgt_dev, int nt, int first, int last,
                    double stop_error, double stop_improvement, hipStream_t st) {
  for (int it = first; it < last; ++it) {
    icp_nn(r, src_dev, ns, tgt_dev, nt, st);
    hipLaunchKernelGGL(icp_update_kernel, dim3(r.l.update_blocks < kIcpUpdateGrid ? r.l.update_blocks : kIcpUpdateGrid), dim3(kIcpThreads), 0, st, src_dev, ns, tgt_dev, nt, r.state,
                       r.cand_d, r.cand_i, r.partials, it, stop_error, stop_improvement, r.grid ? 1 : kIcpSplits);
  }
}
void icp_unpack(const IcpState& h, double* result) {
  result[0] = h.scale; result[1] = h.t[0]; result[2] = h.t[1]; result[3] = h.t[2];
  result[4] = (double)h.iters; result[5] = h.error;
}
}  // namespace

int asdf_icp_ts(const double* src_dev, int32_t ns, const double* tgt_dev, int32_t nt, int32_t max_iter, double stop_error,
                double stop_improvement, void* workspace_dev, size_t workspace_bytes, double* result, void* stream) {
  if (!result) return ASDF_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  IcpRun r;
  const int rc = icp_begin(src_dev, ns, tgt_dev, nt, max_iter, workspace_dev, workspace_bytes, st, r);
  if (rc != ASDF_OK) return rc;
  IcpState h;
  h.done = 0;
  for (int it = 0; it < max_iter && !h.done;) {
    const int stop = it + kIcpBatch < max_iter ? it + kIcpBatch : max_iter;
    icp_iterations(r, src_dev, ns, tgt_dev, nt, it, stop, stop_error, stop_improvement, st);
    it = stop;
    ASDF_HIP(hipGetLastError());
    ASDF_HIP(hipMemcpyAsync(&h, r.state, sizeof(h), hipMemcpyDeviceToHost, st));
    ASDF_HIP(hipStreamSynchronize(st));
  }
  icp_unpack(h, result);
  forget_run(workspace_dev);
  return ASDF_OK;
}

int asdf_icp_ts_enqueue(const double* src_dev, int32_t ns, const double* tgt_dev, int32_t nt, int32_t max_iter,
                        double stop_error, double stop_improvement, void* workspace_dev, size_t workspace_bytes,
                        double* result_mapped, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  IcpRun r;
  const int rc = icp_begin(src_dev, ns, tgt_dev, nt, max_iter, workspace_dev, workspace_bytes, st, r);
  if (rc != ASDF_OK) return rc;
  icp_iterations(r, src_dev, ns, tgt_dev, nt, 0, max_iter, stop_error, stop_improvement, st);
  if (result_mapped) hipLaunchKernelGGL(icp_publish_kernel, dim3(1), dim3(1), 0, st, r.state, result_mapped);
  ASDF_HIP(hipGetLastError());
  return ASDF_OK;
}

int asdf_icp_ts_enqueue_range(const double* src_dev, int32_t ns, const double* tgt_dev, int32_t nt, int32_t first_iter, int32_t last_iter,
                              double stop_error, double stop_improvement, void* workspace_dev, size_t workspace_bytes,
                              double* result_mapped, void* stream) {
  if (first_iter < 0 || last_iter <= first_iter) return ASDF_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  IcpRun r;
  int rc;
  if (first_iter == 0) rc = icp_begin(src_dev, ns, tgt_dev, nt, last_iter, workspace_dev, workspace_bytes, st, r);
  else rc = (!src_dev || !tgt_dev) ? ASDF_EINVAL : icp_attach(ns, nt, workspace_dev, workspace_bytes, r, run_uses_grid(workspace_dev, ns, nt));
  if (rc != ASDF_OK) return rc;
  icp_iterations(r, src_dev, ns, tgt_dev, nt, first_iter, last_iter, stop_error, stop_improvement, st);
  if (result_mapped) hipLaunchKernelGGL(icp_publish_kernel, dim3(1), dim3(1), 0, st, r.state, result_mapped);
  ASDF_HIP(hipGetLastError());
  return ASDF_OK;
}

int asdf_icp_ts_result(const void* workspace_dev, double* result, void* stream) {
  if (!workspace_dev || !result) return ASDF_EINVAL;
  IcpState h;
  ASDF_HIP(hipMemcpyAsync(&h, workspace_dev, sizeof(h), hipMemcpyDeviceToHost, (hipStream_t)stream));
  ASDF_HIP(hipStreamSynchronize((hipStream_t)stream));
  icp_unpack(h, result);
  if (h.done) forget_run(workspace_dev);        // the run is over: a continuation (first_iter > 0) is only ever enqueued for an unfinished one
  return ASDF_OK;
}

int asdf_chamfer(const double* a_dev, int32_t na, const double* b_dev, int32_t nb, void* workspace_dev, size_t workspace_bytes,
                 double* result, void* stream) {
  if (!result) return ASDF_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  IcpRun r;
  const int rc = icp_begin(a_dev, na, b_dev, nb, 1, workspace_dev, workspace_bytes, st, r);     // identity transform
  if (rc != ASDF_OK) return rc;
  icp_nn(r, a_dev, na, b_dev, nb, st);
  const int blocks = (na + kIcpThreads - 1) / kIcpThreads + (nb + kIcpThreads - 1) / kIcpThreads;   // <= update_blocks + 1
  double* out = (double*)((char*)workspace_dev + 128);     // inside the 256-byte state slot
  hipLaunchKernelGGL(chamfer_reduce_kernel, dim3(blocks), dim3(kIcpThreads), 0, st, na, nb, r.state, r.cand_d, r.partials, out, r.grid ? 1 : kIcpSplits);
  ASDF_HIP(hipGetLastError());
  ASDF_HIP(hipMemcpyAsync(result, out, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
  ASDF_HIP(hipStreamSynchronize(st));
  forget_run(workspace_dev);
  return ASDF_OK;
}

}  // extern "C"
