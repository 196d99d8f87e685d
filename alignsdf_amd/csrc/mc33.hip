// K3-K6: Lewiner marching cubes (MC33) on gfx950 - replaces the host call
// skimage.measure.marching_cubes_lewiner (utils/mesh.py:354, deep_sdf/mesh.py:81).
//
// The sequential routine creates a vertex the first time a cell's triangle list references a grid
// edge and shares it through per-layer lookup arrays.  Every cell adjacent to an intersected edge
// references it, so "first reference" is decidable locally: the owner of an edge is the adjacent
// in-bounds cell that comes first in scan order (axis 0 slowest).  That makes the output - vertex
// order, face order, vertex ids - reproducible in parallel, element for element.
//
// Round-2 chain (three streaming kernels + one single-block reduction; round 1 wrote a 4-byte code for EVERY cell and
// scanned 16 k block totals in one workgroup - 262 us per 256^3 volume, 3x the algorithmic traffic):
//   K3 mc_classify     one thread per 4 x-consecutive cells (float4 corner-row loads): sign patterns, tiling of an active
//                      cell from one table word (mc33_direct.h) or, for the ambiguous MC33 cases, the deciders of
//                      mc33_common.h.  Only ACTIVE cells leave the kernel: each workgroup compacts its active cells (in
//                      scan order) into its own 1024 slots of a list and writes one record per block (triangle / vertex
//                      totals, active count, min / max).  No atomics.  HBM traffic = the 4 N^3 volume read.
//   K4 mc_finalize     ONE workgroup over the block records: totals per super-block (~sqrt(#blocks) blocks each), their
//                      exclusive bases, grand totals, min / max; publishes (V, F, min, max) to the header and - without
//                      any host synchronisation - to mapped pinned host memory when the caller passed some.
//   K5 mc_emit_verts   one wave per non-empty block: base = super-block base + the totals of the blocks in front of it
//                      inside the super-block (a 128-term wave sum), wave scan over the block's active cells -> vertex
//                      ids; fp64 interpolation of the owned vertices (as the routine does); ids published per grid edge
//   K6 mc_emit_faces   the same walk for triangles; three id look-ups per face
// K5 / K6 touch only active cells (about 1 % of the cells of an SDF volume): 12 V + 12 F bytes written + the id table.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>

#include "../../include/alignsdf_hip.h"
#include "common.h"

#define MC33_TABLE_QUAL __device__ const
#include "mc33_common.h"
#include "mc33_direct.h"

namespace asdf {

constexpr int kMcThreads = 256;
constexpr int kMcCellsPerThread = 4;
constexpr int kMcChunk = kMcThreads * kMcCellsPerThread;   // cells per workgroup

// cell code: [13:0] tiling offset in kMcTiles, [17:14] #triangles, [21:18] #owned (new) vertices; in the compacted list
// bits [31:22] carry the cell's slot inside its workgroup (0 .. 1023)
static_assert(kMcTilesSize < (1 << 14), "tiling offsets must fit 14 bits");
__device__ __forceinline__ unsigned code_pack(int off, int nt, int nv) { return (unsigned)off | (nt << 14) | (nv << 18); }
__device__ __forceinline__ int code_off(unsigned c) { return c & 0x3fff; }
__device__ __forceinline__ int code_nt(unsigned c) { return (c >> 14) & 15; }
__device__ __forceinline__ int code_nv(unsigned c) { return (c >> 18) & 15; }

struct McHeader {          // first 64 bytes of the workspace
  unsigned total_tris;
  unsigned total_verts;
  unsigned min_key;        // order-preserving keys of the volume's min / max
  unsigned max_key;
  unsigned pad[12];
};

struct McDims {
  int nx, ny, nz;          // nx = fastest axis (axis 2)
  int cx, cy, cz;          // cells per axis
  int cxp;                 // cells per row padded to a multiple of 4 (one thread classifies 4 x-consecutive cells)
  long long nslots;        // cxp * cy * cz cell slots, row-major = scan order; padding slots are never active
  int nblocks;
  int sb_shift;            // a super-block = 2^sb_shift consecutive blocks (about sqrt(nblocks))
  int nsuper;
  double inv_cxp, inv_cy;  // reciprocals for cell_coords
};

__device__ __forceinline__ unsigned float_key(float f) {
  const unsigned b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ inline float key_float(unsigned k) {
  const unsigned b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  float f;
  std::memcpy(&f, &b, 4);
  return f;
}

// n / dv for n < 2^32 through one fp64 multiply with the host-computed reciprocal and a one-step correction (a 32-bit
// integer division is a ~25-instruction VALU sequence here, a 64-bit one ~100; two per thread were most of mc_classify's
// instruction stream in round 1).  (double)n is exact and the product is within 2^-52 relative of n / dv, so the truncated
// quotient is off by at most one, and only next to a multiple of dv.
__device__ __forceinline__ unsigned div_u32(unsigned n, unsigned dv, double inv, unsigned& rem) {
  unsigned q = (unsigned)((double)n * inv);
  int r = (int)(n - q * dv);
  if (r < 0) { --q; r += (int)dv; }
  else if (r >= (int)dv) { ++q; r -= (int)dv; }
  rem = (unsigned)r;
  return q;
}

__device__ __forceinline__ void cell_coords(const McDims& d, long long slot, int& x, int& y, int& z) {
  unsigned ux, uy;
  const unsigned r = div_u32((unsigned)slot, (unsigned)d.cxp, d.inv_cxp, ux);
  z = (int)div_u32(r, (unsigned)d.cy, d.inv_cy, uy);
  x = (int)ux;
  y = (int)uy;
}

// wave64 inclusive scan on the DPP network (row_shr 1 / 2 / 4 / 8 inside the rows of 16 lanes, then row_bcast:15 into rows
// 1 and 3 and row_bcast:31 into rows 2 and 3): six VALU instructions with a DPP operand, where the __shfl form is six
// ds_bpermute (LDS pipe) + address arithmetic + select.  Lane 63 holds the wave total.
#define ASDF_DPP(old, v, ctrl, rows) (unsigned)__builtin_amdgcn_update_dpp((int)(old), (int)(v), (ctrl), (rows), 0xf, false)
__device__ __forceinline__ unsigned wave_incl_sum(unsigned v) {
  v += ASDF_DPP(0, v, 0x111, 0xf);
  v += ASDF_DPP(0, v, 0x112, 0xf);
  v += ASDF_DPP(0, v, 0x114, 0xf);
  v += ASDF_DPP(0, v, 0x118, 0xf);
  v += ASDF_DPP(0, v, 0x142, 0xa);
  v += ASDF_DPP(0, v, 0x143, 0xc);
  return v;
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {       // result in lane 63
  v = min(v, ASDF_DPP(0xffffffffu, v, 0x111, 0xf));
  v = min(v, ASDF_DPP(0xffffffffu, v, 0x112, 0xf));
  v = min(v, ASDF_DPP(0xffffffffu, v, 0x114, 0xf));
  v = min(v, ASDF_DPP(0xffffffffu, v, 0x118, 0xf));
  v = min(v, ASDF_DPP(0xffffffffu, v, 0x142, 0xa));
  v = min(v, ASDF_DPP(0xffffffffu, v, 0x143, 0xc));
  return v;
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {       // result in lane 63
  v = max(v, ASDF_DPP(0, v, 0x111, 0xf));
  v = max(v, ASDF_DPP(0, v, 0x112, 0xf));
  v = max(v, ASDF_DPP(0, v, 0x114, 0xf));
  v = max(v, ASDF_DPP(0, v, 0x118, 0xf));
  v = max(v, ASDF_DPP(0, v, 0x142, 0xa));
  v = max(v, ASDF_DPP(0, v, 0x143, 0xc));
  return v;
}

// Is cell (x,y,z) the first cell in scan order that touches cell-local edge e ?
__device__ __forceinline__ bool owns_edge(int e, int x, int y, int z) {
  const int axis = MC33_EDGE_AXIS(e);
  // the edge's own position relative to the cell; the other adjacent cells lie at -1 along the two
  // axes perpendicular to it, whenever that index is >= 0
  const int dx = MC33_EDGE_DX(e), dy = MC33_EDGE_DY(e), dz = MC33_EDGE_DZ(e);
  bool first = true;
  if (axis != 0) first = first && (dx == 1 || x == 0);   // dx == 0: cell x-1 shares it (if it exists)
  if (axis != 1) first = first && (dy == 1 || y == 0);
  if (axis != 2) first = first && (dz == 1 || z == 0);
  return first;
}

// bit e set <=> cell (x,y,z) owns its edge e (bit 12: the cell-centre vertex, always the cell's own)
__device__ __forceinline__ unsigned owned_edge_mask(int x, int y, int z) {
  unsigned own = 1u << 12;
#pragma unroll
  for (int e = 0; e < 12; ++e) own |= owns_edge(e, x, y, z) ? 1u << e : 0u;
  return own;
}

// The 3 * nt edge ids of a tiling (nt <= 12) as 4-bit fields, 12 per 64-bit word (= 4 triangles), unused fields 15.
// All byte loads of a word are issued together: walking kMcTiles entry by entry is a chain of dependent ~1 us memory
// round trips (up to 15 of them for an ordinary cell), which is what an active cell cost in all three kernels.
struct TilePack {
  unsigned long long w[3];
};
__device__ __forceinline__ TilePack load_tiling(int off, int nt) {
  TilePack t;
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    unsigned long long w = ~0ull;
    if (nt > 4 * g) {
      unsigned e[12];
#pragma unroll
      for (int k = 0; k < 12; ++k) e[k] = 12 * g + k < 3 * nt ? (unsigned)kMcTiles[off + 12 * g + k] : 15u;
      unsigned lo = 0, hi = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) lo |= e[k] << (4 * k);
#pragma unroll
      for (int k = 8; k < 12; ++k) hi |= e[k] << (4 * (k - 8));
      w = ((unsigned long long)(hi | 0xffff0000u) << 32) | lo;
    }
    t.w[g] = w;
  }
  return t;
}
__device__ __forceinline__ unsigned tile_edge(unsigned long long w, int k) { return (unsigned)(w >> (4 * k)) & 15u; }

__device__ __forceinline__ void load_corners(const float* vol, const McDims& d, int x, int y, int z, double level, double* v) {
  const float* p0 = vol + ((size_t)z * d.ny + y) * d.nx + x;
  const float* p1 = p0 + (size_t)d.ny * d.nx;
  v[0] = (double)p0[0] - level; v[1] = (double)p0[1] - level;
  v[2] = (double)p0[d.nx + 1] - level; v[3] = (double)p0[d.nx] - level;
  v[4] = (double)p1[0] - level; v[5] = (double)p1[1] - level;
  v[6] = (double)p1[d.nx + 1] - level; v[7] = (double)p1[d.nx] - level;
}

// 5 consecutive values of one volume row starting at x0 (a multiple of 4); entries beyond the row are not used
__device__ __forceinline__ void load_row5(const float* __restrict__ row, int x0, int nx, bool vec, float* v) {
  if (vec) {
    const float4 q = *reinterpret_cast<const float4*>(row + x0);
    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    v[4] = x0 + 4 < nx ? row[x0 + 4] : 0.0f;
  } else {
#pragma unroll
    for (int i = 0; i < 5; ++i) v[i] = x0 + i < nx ? row[x0 + i] : 0.0f;
  }
}

#ifndef ASDF_MC_WAVES
#define ASDF_MC_WAVES 0
#endif
__global__ __launch_bounds__(kMcThreads)
#if ASDF_MC_WAVES
__attribute__((amdgpu_waves_per_eu(ASDF_MC_WAVES, ASDF_MC_WAVES)))
#endif
void mc_classify(const float* __restrict__ vol, McDims d, double level, float level_f, uint2* __restrict__ block_tot,
                 uint2* __restrict__ block_seg, uint2* __restrict__ block_minmax, unsigned* __restrict__ compact) {
  __shared__ unsigned s_tri[kMcThreads / 64], s_vert[kMcThreads / 64], s_min[kMcThreads / 64], s_max[kMcThreads / 64], s_act[kMcThreads / 64];
  unsigned ntri = 0, nvert = 0, nact = 0;
  float lo = INFINITY, hi = -INFINITY;
  unsigned cc[4] = {0, 0, 0, 0};
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long long group = (long long)blockIdx.x * kMcThreads + threadIdx.x;     // 4 x-consecutive cell slots
  const bool live = group * 4 < d.nslots;
  int x0 = 0, y = 0, z = 0;
  if (live) cell_coords(d, group * 4, x0, y, z);
  {
    // four corner rows (z,y) (z,y+1) (z+1,y) (z+1,y+1), 5 values each: a float4 per row, and the fifth value is the NEXT
    // lane's first (same row, next 4 cells) - a cross-lane move instead of a second, 4-byte-per-lane load instruction; only
    // the last lane of the wave and the last group of a row load it themselves
    const bool vec = (d.nx & 3) == 0 && ((size_t)vol & 15) == 0;
    const float* r00 = vol + ((size_t)z * d.ny + y) * d.nx;
    const float* rows[4] = {r00, r00 + d.nx, r00 + (size_t)d.ny * d.nx, r00 + (size_t)d.ny * d.nx + d.nx};
    float a[4][5];
    const bool neighbour = vec && lane < 63 && x0 + 4 < d.cxp;     // lane + 1 holds (x0 + 4 .. x0 + 7) of the same rows
    if (vec) {
      // ALL loads first, then the cross-lane moves: with one `if (vec)` per row the compiler put an s_waitcnt vmcnt(0) in
      // front of every row's cross-lane move, i.e. four memory latencies in series per thread - that, not bandwidth or the
      // instruction count, was what kept this kernel at ~100 us in the first half of round 2
      float4 q[4];
      float t5[4];
      const bool own5 = live && !neighbour && x0 + 4 < d.nx;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        q[r] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live) q[r] = *reinterpret_cast<const float4*>(rows[r] + x0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) t5[r] = own5 ? rows[r][x0 + 4] : 0.0f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        a[r][0] = q[r].x; a[r][1] = q[r].y; a[r][2] = q[r].z; a[r][3] = q[r].w;
        // wave_shl:1 - lane l receives lane l + 1's value (one v_mov with a DPP operand; lane 63 keeps `old`)
        const float nxt = __uint_as_float(ASDF_DPP(0, __float_as_uint(q[r].x), 0x130, 0xf));
        a[r][4] = neighbour ? nxt : t5[r];
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 5; ++i) a[r][i] = (live && x0 + i < d.nx) ? rows[r][x0 + i] : 0.0f;
    }
    if (live) {
      // Sign bits of the 20 loaded values in fp32: c > level (in double, as the routine compares) <=> c > level_f, with
      // level_f the largest float <= level (computed on the host).
      unsigned bits[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        bits[r] = 0;
#pragma unroll
        for (int i = 0; i < 5; ++i) bits[r] |= (a[r][i] > level_f ? 1u : 0u) << i;
      }
      // The volume's min / max: every value is element 0..3 of corner row 0 of exactly one thread, except the last row of
      // a slice / the last slice (rows 1 / 2 / 3 of the threads next to them) and, when the row length is not a multiple of
      // 4, possibly its last value (element 4 of the row's last thread).  (This kernel is VALU-bound: min + max over all 20
      // loaded values - each value seen by four threads - and the per-corner fp64 form of round 1 before it were most of
      // the instructions of a wave without an active cell.)
      const bool tail = x0 + 4 < d.nx && x0 + 4 >= d.cxp;
      auto row_minmax = [&](int r) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (vec || x0 + i < d.nx) { lo = fminf(lo, a[r][i]); hi = fmaxf(hi, a[r][i]); }
        if (tail) { lo = fminf(lo, a[r][4]); hi = fmaxf(hi, a[r][4]); }
      };
      row_minmax(0);
      if (y == d.cy - 1) row_minmax(1);
      if (z == d.cz - 1) { row_minmax(2); if (y == d.cy - 1) row_minmax(3); }
      // a cell is active iff its 8 corner bits are mixed: for the four cells at once, bit i of `any | any >> 1` = a corner of
      // cell i is above the level, bit i of `all & all >> 1` = all of them are
      const unsigned any = bits[0] | bits[1] | bits[2] | bits[3], all = bits[0] & bits[1] & bits[2] & bits[3];
      const int ncell = min(4, d.cx - x0);                   // <= 0 for the padding group of a row
      const unsigned active = ncell > 0 ? ((any | (any >> 1)) & ~(all & (all >> 1)) & ((1u << ncell) - 1)) : 0u;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int x = x0 + i;
        if ((active >> i) & 1) {
          // corner-sign pattern in the routine's corner order v0..v7: (x,y,z) (x+1,y,z) (x+1,y+1,z) (x,y+1,z), the same at z+1
          const unsigned index = ((bits[0] >> i) & 1) | (((bits[0] >> (i + 1)) & 1) << 1) | (((bits[1] >> (i + 1)) & 1) << 2) |
                                 (((bits[1] >> i) & 1) << 3) | (((bits[2] >> i) & 1) << 4) | (((bits[2] >> (i + 1)) & 1) << 5) |
                                 (((bits[3] >> (i + 1)) & 1) << 6) | (((bits[3] >> i) & 1) << 7);
          // The MC33 cases without a test - nearly every active cell of a smooth surface - take their tiling from one
          // table word.  The deciders of the other cases are ~3000 instructions of divergent control flow (4 unrolled
          // copies here) that a wave walks through even if one lane needs one branch of it: with every active cell sent
          // through them, the waves that hold an active cell (30 % of them) cost as much as all the rest of the kernel.
          const uint2 dm = *reinterpret_cast<const uint2*>(kMcDirect[index]);
          const unsigned direct = dm.x;
          unsigned used = dm.y;                 // cell-local edges the tiling references
          int off = (int)(direct & 0x3fffu), nt = (int)(direct >> 14);
          if (direct == 0xffffffffu) {
            const float c[8] = {a[0][i], a[0][i + 1], a[1][i + 1], a[1][i], a[2][i], a[2][i + 1], a[3][i + 1], a[3][i]};
            // the deciders index the corner array dynamically (it lives in scratch): only these cells pay for it
            double v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = (double)c[k] - level;
            nt = mc33_select_tiling(v, &off);
            const TilePack tp = load_tiling(off, nt);
#pragma unroll
            for (int g = 0; g < 3; ++g) {
              if (nt <= 4 * g) break;
#pragma unroll
              for (int k = 0; k < 12; ++k) used |= 1u << tile_edge(tp.w[g], k);
            }
          }
          // new vertices of this cell = the distinct edges its triangles reference that the cell owns
          const int nv = __popc(used & 0x1fffu & owned_edge_mask(x, y, z));      // new vertices = referenced edges the cell owns
          if (nt > 0) {                       // ("impossible case 13" cells emit nothing and are not listed)
            cc[i] = code_pack(off, nt, nv) | ((unsigned)(4 * threadIdx.x + i) << 22);
            ntri += nt; nvert += nv; ++nact;
          }
        }
      }
    }
  }
  // workgroup totals + exclusive scan of the active counts (thread order = scan order)
  const unsigned klo = wave_min_u32(float_key(lo)), khi = wave_max_u32(float_key(hi));      // lane 63
  unsigned inc = 0;
  if (__ballot(nact != 0)) {            // (wave-uniform) about 70 % of the waves of an SDF volume hold no active cell
    inc = wave_incl_sum(nact);
    ntri = wave_incl_sum(ntri);
    nvert = wave_incl_sum(nvert);
  }
  if (lane == 63) { s_act[w] = inc; s_tri[w] = ntri; s_vert[w] = nvert; s_min[w] = klo; s_max[w] = khi; }
  __syncthreads();
  // No atomics: a block's totals, its min / max and its active-cell count go to per-block records (mc_finalize reduces
  // them), and its compacted cells to the block's OWN 1024 slots of the list.  (The first round-2 form reserved list
  // segments and accumulated totals / min / max with atomics on ~sqrt(#blocks) super-block slots: the two min / max
  // atomics of every block alone were 20 of its 52 us, the three of a non-empty block another 14 of the 37 us that the
  // active cells of a volume cost.)
  if (threadIdx.x == 0) {
    unsigned t = 0, vv = 0, a2 = 0xffffffffu, b2 = 0, act = 0;
    for (int i = 0; i < kMcThreads / 64; ++i) { t += s_tri[i]; vv += s_vert[i]; a2 = min(a2, s_min[i]); b2 = max(b2, s_max[i]); act += s_act[i]; }
    block_tot[blockIdx.x] = make_uint2(t, vv);
    block_seg[blockIdx.x] = make_uint2(blockIdx.x * (unsigned)kMcChunk, act);
    block_minmax[blockIdx.x] = make_uint2(a2, b2);
  }
  if (nact) {
    unsigned pos = blockIdx.x * (unsigned)kMcChunk + inc - nact;
    for (int k = 0; k < w; ++k) pos += s_act[k];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (cc[i]) compact[pos++] = cc[i];
  }
}

// One workgroup over the per-block records: totals per super-block (a wave each), their exclusive bases, the grand totals
// and the volume's min / max.
__global__ __launch_bounds__(1024) void mc_finalize(const uint2* __restrict__ block_tot, const uint2* __restrict__ block_minmax, McDims d,
                                                    uint2* __restrict__ super_base, McHeader* hdr, unsigned* result_mapped) {
  __shared__ uint2 s_sum[1024];
  __shared__ uint2 s_wave[16];
  __shared__ uint2 s_carry;
  __shared__ unsigned s_lo[16], s_hi[16];
  if (threadIdx.x == 0) s_carry = make_uint2(0, 0);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  unsigned klo = 0xffffffffu, khi = 0;
  for (int start = 0; start < d.nsuper; start += 1024) {
    // totals of super-blocks start .. start + 1023: wave w takes every 16th
    for (int k = w; k < 1024 && start + k < d.nsuper; k += 16) {
      const int first = (start + k) << d.sb_shift, last = min(first + (1 << d.sb_shift), d.nblocks);
      unsigned t = 0, v = 0;
      for (int b = first + lane; b < last; b += 64) {
        const uint2 tv = block_tot[b], mm = block_minmax[b];
        t += tv.x; v += tv.y;
        klo = min(klo, mm.x); khi = max(khi, mm.y);
      }
      t = wave_incl_sum(t); v = wave_incl_sum(v);
      if (lane == 63) s_sum[k] = make_uint2(t, v);
    }
    __syncthreads();
    const int i = start + threadIdx.x;
    const uint2 v = i < d.nsuper ? s_sum[threadIdx.x] : make_uint2(0, 0);
    const uint2 inc = make_uint2(wave_incl_sum(v.x), wave_incl_sum(v.y));
    if (lane == 63) s_wave[w] = inc;
    __syncthreads();
    uint2 pre = s_carry;
    for (int k = 0; k < w; ++k) { pre.x += s_wave[k].x; pre.y += s_wave[k].y; }
    if (i < d.nsuper) super_base[i] = make_uint2(pre.x + inc.x - v.x, pre.y + inc.y - v.y);
    __syncthreads();
    if (threadIdx.x == 1023) s_carry = make_uint2(pre.x + inc.x, pre.y + inc.y);
    __syncthreads();
  }
  klo = wave_min_u32(klo); khi = wave_max_u32(khi);
  if (lane == 63) { s_lo[w] = klo; s_hi[w] = khi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    klo = s_lo[0]; khi = s_hi[0];
    for (int k = 1; k < 16; ++k) { klo = min(klo, s_lo[k]); khi = max(khi, s_hi[k]); }
    hdr->total_tris = s_carry.x; hdr->total_verts = s_carry.y; hdr->min_key = klo; hdr->max_key = khi;
    if (result_mapped) {
      result_mapped[0] = s_carry.y; result_mapped[1] = s_carry.x; result_mapped[2] = klo; result_mapped[3] = khi;
      __threadfence_system();
    }
  }
}

constexpr int kEmitThreads = 64;      // one wave per block of 1024 cell slots: a non-empty block holds ~30 active cells

// first output id of `block`: the super-block's base + the totals of the blocks in front of it inside the super-block
__device__ __forceinline__ uint2 block_first_ids(const McDims& d, int block, const uint2* __restrict__ super_base,
                                                 const uint2* __restrict__ block_tot) {
  const int lane = threadIdx.x & 63;
  const int first = (block >> d.sb_shift) << d.sb_shift;
  uint2 acc = make_uint2(0, 0);
  for (int b = first + lane; b < block; b += 64) { const uint2 t = block_tot[b]; acc.x += t.x; acc.y += t.y; }
  acc.x = __builtin_amdgcn_readlane(wave_incl_sum(acc.x), 63);
  acc.y = __builtin_amdgcn_readlane(wave_incl_sum(acc.y), 63);
  const uint2 sb = super_base[block >> d.sb_shift];
  return make_uint2(sb.x + acc.x, sb.y + acc.y);
}

// exclusive scan of `val` over one wave plus running carry
__device__ __forceinline__ unsigned wave_excl_scan(unsigned val, unsigned& carry) {
  const unsigned inc = wave_incl_sum(val);
  const unsigned out = carry + inc - val;
  carry += __builtin_amdgcn_readlane(inc, 63);
  return out;
}

__device__ __forceinline__ size_t vid_slot(const McDims& d, int e, int x, int y, int z) {
  if (e == 12) return 4 * (((size_t)z * d.ny + y) * d.nx + x) + 3;
  const int ex = x + MC33_EDGE_DX(e), ey = y + MC33_EDGE_DY(e), ez = z + MC33_EDGE_DZ(e);
  return 4 * (((size_t)ez * d.ny + ey) * d.nx + ex) + MC33_EDGE_AXIS(e);
}

__global__ __launch_bounds__(kEmitThreads) void mc_emit_verts(const float* __restrict__ vol, McDims d, double level,
                                                              const unsigned* __restrict__ compact, const uint2* __restrict__ block_tot,
                                                              const uint2* __restrict__ block_seg, const uint2* __restrict__ super_base,
                                                              unsigned* __restrict__ vid, float* __restrict__ verts, unsigned cap_verts) {
  if (block_tot[blockIdx.x].y == 0) return;             // no vertex is owned by this block's 1024 cell slots
  const uint2 seg = block_seg[blockIdx.x];
  unsigned carry = block_first_ids(d, blockIdx.x, super_base, block_tot).y;
  const long long base = (long long)blockIdx.x * kMcChunk;
  for (unsigned i0 = 0; i0 < seg.y; i0 += kEmitThreads) {
    const unsigned i = i0 + threadIdx.x;
    const unsigned cc = i < seg.y ? compact[seg.x + i] : 0;
    unsigned id = wave_excl_scan(code_nv(cc), carry);
    if (code_nv(cc) == 0) continue;
    int x, y, z;
    cell_coords(d, base + (cc >> 22), x, y, z);
    double v[8];
    load_corners(vol, d, x, y, z, level, v);
    const int off = code_off(cc), nt = code_nt(cc);
    const TilePack tp = load_tiling(off, nt);
    unsigned todo = owned_edge_mask(x, y, z);           // owned edges not emitted yet: ids follow the order of first reference
#pragma unroll
    for (int g = 0; g < 3; ++g) {
     if (nt <= 4 * g) break;
     const unsigned long long tw = tp.w[g];
     for (int k = 0; k < 12; ++k) {
      const int e = (int)tile_edge(tw, k);
      if (!((todo >> e) & 1)) continue;                 // (15 = unused field: never owned)
      todo &= ~(1u << e);
      double fx = 0, fy = 0, fz = 0, ff = 0;
      if (e == 12) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const double w = 1.0 / (MC33_EPS + fabs(v[c]));
          fx += (double)((0x66 >> c) & 1) * w;    // corners 1,2,5,6 at x+1
          fy += (double)((0xCC >> c) & 1) * w;    // corners 2,3,6,7 at y+1
          fz += (double)((0xF0 >> c) & 1) * w;    // corners 4..7 at z+1
          ff += w;
        }
      } else {
        const int a = e < 8 ? e : e - 8;                                  // first end point
        const int b = e < 8 ? ((e & 4) | ((e + 1) & 3)) : e - 4;          // second end point
        const double wa = 1.0 / (MC33_EPS + fabs(v[a])), wb = 1.0 / (MC33_EPS + fabs(v[b]));
        fx += (double)((0x66 >> a) & 1) * wa; fy += (double)((0xCC >> a) & 1) * wa; fz += (double)((0xF0 >> a) & 1) * wa; ff += wa;
        fx += (double)((0x66 >> b) & 1) * wb; fy += (double)((0xCC >> b) & 1) * wb; fz += (double)((0xF0 >> b) & 1) * wb; ff += wb;
      }
      if (id < cap_verts) {                     // (asdf_mc_emit_bounded: a buffer sized before the count was known)
        float* o = verts + 3 * (size_t)id;
        o[0] = (float)((double)z + fz / ff);      // (axis0, axis1, axis2) = (z, y, x)
        o[1] = (float)((double)y + fy / ff);
        o[2] = (float)((double)x + fx / ff);
      }
      vid[vid_slot(d, e, x, y, z)] = id;
      ++id;
     }
    }
  }
}

__global__ __launch_bounds__(kEmitThreads) void mc_emit_faces(McDims d, const unsigned* __restrict__ compact,
                                                              const uint2* __restrict__ block_tot, const uint2* __restrict__ block_seg,
                                                              const uint2* __restrict__ super_base, const unsigned* __restrict__ vid,
                                                              int* __restrict__ faces, unsigned cap_faces) {
  const uint2 seg = block_seg[blockIdx.x];
  if (seg.y == 0) return;                               // no active cell in this block's 1024 cell slots
  unsigned carry = block_first_ids(d, blockIdx.x, super_base, block_tot).x;
  const long long base = (long long)blockIdx.x * kMcChunk;
  for (unsigned i0 = 0; i0 < seg.y; i0 += kEmitThreads) {
    const unsigned i = i0 + threadIdx.x;
    const unsigned cc = i < seg.y ? compact[seg.x + i] : 0;
    const int nt = code_nt(cc);
    const unsigned tri0 = wave_excl_scan(nt, carry);
    if (nt == 0) continue;
    int x, y, z;
    cell_coords(d, base + (cc >> 22), x, y, z);
    const TilePack tp = load_tiling(code_off(cc), nt);
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      if (nt <= 4 * g) break;
      // the (up to) 12 id look-ups of four triangles are issued together
      unsigned ids[12];
#pragma unroll
      for (int k = 0; k < 12; ++k) {
        const int e = (int)tile_edge(tp.w[g], k);
        ids[k] = e != 15 ? vid[vid_slot(d, e, x, y, z)] : 0u;
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (4 * g + t >= nt) break;
        if (tri0 + 4 * g + t >= cap_faces) break;
        int* f = faces + 3 * (size_t)(tri0 + 4 * g + t);
        // 'descent' orientation: the routine reverses every face
        f[2] = (int)ids[3 * t + 0];
        f[1] = (int)ids[3 * t + 1];
        f[0] = (int)ids[3 * t + 2];
      }
    }
  }
}

struct McLayout {
  McDims d;
  size_t off_sbase, off_tot, off_seg, off_minmax, off_compact, off_vid, total;
};

static bool mc_layout(int n0, int n1, int n2, McLayout& L) {
  if (n0 < 2 || n1 < 2 || n2 < 2) return false;
  if ((long long)n0 * n1 * n2 > (1ll << 31) || (long long)((n2 + 2) & ~3) * n1 * n0 >= (1ll << 32)) return false;
  McDims& d = L.d;
  d.nx = n2; d.ny = n1; d.nz = n0;
  d.cx = n2 - 1; d.cy = n1 - 1; d.cz = n0 - 1;
  d.cxp = (d.cx + 3) & ~3;
  d.nslots = (long long)d.cxp * d.cy * d.cz;
  d.nblocks = (int)((d.nslots + kMcChunk - 1) / kMcChunk);
  d.sb_shift = 0;
  while ((1ll << (2 * d.sb_shift)) < d.nblocks) ++d.sb_shift;          // 2^shift >= sqrt(nblocks)
  d.nsuper = (d.nblocks + (1 << d.sb_shift) - 1) >> d.sb_shift;
  d.inv_cxp = 1.0 / (double)d.cxp;
  d.inv_cy = 1.0 / (double)d.cy;
  auto align = [](size_t v) { return (v + 255) & ~(size_t)255; };
  size_t o = align(sizeof(McHeader));
  L.off_sbase = o; o = align(o + sizeof(uint2) * (size_t)d.nsuper);
  L.off_tot = o; o = align(o + sizeof(uint2) * (size_t)d.nblocks);
  L.off_seg = o; o = align(o + sizeof(uint2) * (size_t)d.nblocks);
  L.off_minmax = o; o = align(o + sizeof(uint2) * (size_t)d.nblocks);
  L.off_compact = o; o = align(o + sizeof(unsigned) * (size_t)d.nblocks * kMcChunk);      // worst case: every cell active
  L.off_vid = o; o = align(o + sizeof(unsigned) * 4 * (size_t)n0 * n1 * n2);
  L.total = o;
  return true;
}

static int mc_count_enqueue(const float* vol, const McLayout& L, double level, void* ws, unsigned* result_mapped, hipStream_t st) {
  char* w = (char*)ws;
  float level_f = (float)level;                 // largest float <= level: for a float c, (double)c > level <=> c > level_f
  if ((double)level_f > level) level_f = std::nextafterf(level_f, -INFINITY);
  hipLaunchKernelGGL(mc_classify, dim3(L.d.nblocks), dim3(kMcThreads), 0, st, vol, L.d, level, level_f, (uint2*)(w + L.off_tot),
                     (uint2*)(w + L.off_seg), (uint2*)(w + L.off_minmax), (unsigned*)(w + L.off_compact));
  hipLaunchKernelGGL(mc_finalize, dim3(1), dim3(1024), 0, st, (const uint2*)(w + L.off_tot), (const uint2*)(w + L.off_minmax), L.d,
                     (uint2*)(w + L.off_sbase), (McHeader*)w, result_mapped);
  ASDF_HIP(hipGetLastError());
  return ASDF_OK;
}

}  // namespace asdf

using namespace asdf;

extern "C" {

int asdf_mc_workspace_bytes(int32_t n0, int32_t n1, int32_t n2, size_t* bytes) {
  McLayout L;
  if (!bytes || !mc_layout(n0, n1, n2, L)) return ASDF_EINVAL;
  *bytes = L.total;
  return ASDF_OK;
}

int asdf_mc_count_enqueue(const float* vol, int32_t n0, int32_t n1, int32_t n2, double level, void* ws, size_t ws_bytes,
                          uint32_t* result_mapped, void* stream) {
  McLayout L;
  if (!vol || !ws || !mc_layout(n0, n1, n2, L)) return ASDF_EINVAL;
  if (ws_bytes < L.total) return ASDF_ENOSPC;
  return mc_count_enqueue(vol, L, level, ws, result_mapped, (hipStream_t)stream);
}

int asdf_mc_result_status(const uint32_t result[4], double level) {
  if (!result) return ASDF_EINVAL;
  // skimage: ValueError when level is outside [min, max]; RuntimeError when nothing was produced
  const double lo = key_float(result[2]), hi = key_float(result[3]);
  if (level < lo || level > hi) return ASDF_ERANGE;
  if (result[0] == 0) return ASDF_ENOSURF;
  return ASDF_OK;
}

int asdf_mc_count(const float* vol, int32_t n0, int32_t n1, int32_t n2, double level, void* ws, size_t ws_bytes,
                  uint32_t* num_verts, uint32_t* num_faces, void* stream) {
  McLayout L;
  if (!vol || !ws || !num_verts || !num_faces || !mc_layout(n0, n1, n2, L)) return ASDF_EINVAL;
  if (ws_bytes < L.total) return ASDF_ENOSPC;
  hipStream_t st = (hipStream_t)stream;
  const int rc = mc_count_enqueue(vol, L, level, ws, nullptr, st);
  if (rc != ASDF_OK) return rc;
  McHeader h;
  ASDF_HIP(hipMemcpyAsync(&h, ws, sizeof(h), hipMemcpyDeviceToHost, st));
  ASDF_HIP(hipStreamSynchronize(st));
  *num_verts = h.total_verts;
  *num_faces = h.total_tris;
  const uint32_t r[4] = {h.total_verts, h.total_tris, h.min_key, h.max_key};
  return asdf_mc_result_status(r, level);
}

int asdf_mc_emit(const float* vol, int32_t n0, int32_t n1, int32_t n2, double level, void* ws, size_t ws_bytes,
                 float* verts, int32_t* faces, void* stream) {
  return asdf_mc_emit_bounded(vol, n0, n1, n2, level, ws, ws_bytes, verts, 0xffffffffu, faces, 0xffffffffu, stream);
}

int asdf_mc_emit_bounded(const float* vol, int32_t n0, int32_t n1, int32_t n2, double level, void* ws, size_t ws_bytes,
                         float* verts, uint32_t cap_verts, int32_t* faces, uint32_t cap_faces, void* stream) {
  McLayout L;
  if (!vol || !ws || !verts || !faces || !mc_layout(n0, n1, n2, L)) return ASDF_EINVAL;
  if (ws_bytes < L.total) return ASDF_ENOSPC;
  hipStream_t st = (hipStream_t)stream;
  char* w = (char*)ws;
  const unsigned* compact = (const unsigned*)(w + L.off_compact);
  const uint2* tot = (const uint2*)(w + L.off_tot);
  const uint2* seg = (const uint2*)(w + L.off_seg);
  const uint2* sbase = (const uint2*)(w + L.off_sbase);
  unsigned* vid = (unsigned*)(w + L.off_vid);
  hipLaunchKernelGGL(mc_emit_verts, dim3(L.d.nblocks), dim3(kEmitThreads), 0, st, vol, L.d, level, compact, tot, seg, sbase, vid, verts, cap_verts);
  hipLaunchKernelGGL(mc_emit_faces, dim3(L.d.nblocks), dim3(kEmitThreads), 0, st, L.d, compact, tot, seg, sbase, vid, faces, cap_faces);
  ASDF_HIP(hipGetLastError());
  return ASDF_OK;
}

}  // extern "C"
