// K3-K6: Lewiner marching cubes (MC33) on gfx950 - replaces the host call
// skimage.measure.marching_cubes_lewiner (utils/mesh.py:354, deep_sdf/mesh.py:81).
//
// The sequential routine creates a vertex the first time a cell's triangle list references a grid
// edge and shares it through per-layer lookup arrays.  Every cell adjacent to an intersected edge
// references it, so "first reference" is decidable locally: the owner of an edge is the adjacent
// in-bounds cell that comes first in scan order (axis 0 slowest).  That makes the output - vertex
// order, face order, vertex ids - reproducible in parallel, element for element:
//   K3 mc_classify    one thread per 4 x-consecutive cells (float4 corner-row loads): MC33 case selection
//                     (mc33_common.h) -> 32-bit cell codes (tiling offset, #triangles, #owned vertices) in
//                     row-padded scan order; per-block totals; volume min/max
//   K4 mc_scan_blocks exclusive scan of the per-block totals (one workgroup)
//   K5 mc_emit_verts  in-block scan + block base -> vertex ids; interpolate owned vertices (fp64, as
//                     the routine does), publish their ids in a per-grid-edge table
//   K6 mc_emit_faces  same scan for triangles; look the three vertex ids up and write the face
//                     (K5 / K6 blocks whose 1024 cell slots hold nothing return immediately)
// All four are HBM-bound streaming kernels: 4 B read per voxel in K3, 4 B per cell code in K5/K6.
#include <hip/hip_runtime.h>

#include <cstring>

#include "../../include/alignsdf_hip.h"
#include "common.h"

#define MC33_TABLE_QUAL __device__ const
#include "mc33_common.h"

namespace asdf {

constexpr int kMcThreads = 256;
constexpr int kMcCellsPerThread = 4;
constexpr int kMcChunk = kMcThreads * kMcCellsPerThread;   // cells per workgroup

// cell code: [13:0] tiling offset in kMcTiles, [17:14] #triangles, [21:18] #owned (new) vertices
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
  long long nslots;        // cxp * cy * cz cell slots, row-major = scan order; padding slots carry code 0
  int nblocks;
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

__device__ __forceinline__ void cell_coords(const McDims& d, long long slot, int& x, int& y, int& z) {
  x = (int)(slot % d.cxp);
  const long long r = slot / d.cxp;
  y = (int)(r % d.cy);
  z = (int)(r / d.cy);
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

__global__ __launch_bounds__(kMcThreads) void mc_classify(const float* __restrict__ vol, McDims d, double level,
                                                          uint4* __restrict__ code4, uint2* __restrict__ block_tot,
                                                          uint2* __restrict__ block_minmax) {
  __shared__ unsigned s_tri[kMcThreads / 64], s_vert[kMcThreads / 64], s_min[kMcThreads / 64], s_max[kMcThreads / 64];
  unsigned ntri = 0, nvert = 0;
  float lo = INFINITY, hi = -INFINITY;
  const long long group = (long long)blockIdx.x * kMcThreads + threadIdx.x;     // 4 x-consecutive cell slots
  if (group * 4 < d.nslots) {
    int x0, y, z;
    cell_coords(d, group * 4, x0, y, z);
    const bool vec = (d.nx & 3) == 0 && ((size_t)vol & 15) == 0;
    const float* r00 = vol + ((size_t)z * d.ny + y) * d.nx;
    float a[4][5];      // rows (z,y) (z,y+1) (z+1,y) (z+1,y+1)
    load_row5(r00, x0, d.nx, vec, a[0]);
    load_row5(r00 + d.nx, x0, d.nx, vec, a[1]);
    load_row5(r00 + (size_t)d.ny * d.nx, x0, d.nx, vec, a[2]);
    load_row5(r00 + (size_t)d.ny * d.nx + d.nx, x0, d.nx, vec, a[3]);
    unsigned cc[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int x = x0 + i;
      if (x >= d.cx) break;
      // corners v0..v7: (x,y,z) (x+1,y,z) (x+1,y+1,z) (x,y+1,z) and the same at z+1
      const float c[8] = {a[0][i], a[0][i + 1], a[1][i + 1], a[1][i], a[2][i], a[2][i + 1], a[3][i + 1], a[3][i]};
      bool any_hi = false, any_lo = false;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        lo = fminf(lo, c[k]); hi = fmaxf(hi, c[k]);
        const bool above = (double)c[k] - level > 0.0;
        any_hi |= above; any_lo |= !above;
      }
      if (any_hi && any_lo) {
        // the deciders index the corner array dynamically (it lives in scratch): only active cells pay for it
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (double)c[k] - level;
        int off;
        const int nt = mc33_select_tiling(v, &off);
        int nv = 0;
        unsigned seen = 0;
        for (int k = 0; k < 3 * nt; ++k) {
          const int e = kMcTiles[off + k];
          if (seen & (1u << e)) continue;
          seen |= 1u << e;
          if (e == 12 || owns_edge(e, x, y, z)) ++nv;
        }
        cc[i] = code_pack(off, nt, nv);
        ntri += nt; nvert += nv;
      }
    }
    code4[group] = make_uint4(cc[0], cc[1], cc[2], cc[3]);
  }
  // workgroup totals
  unsigned klo = float_key(lo), khi = float_key(hi);
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    ntri += __shfl_xor(ntri, m); nvert += __shfl_xor(nvert, m);
    klo = min(klo, (unsigned)__shfl_xor((int)klo, m)); khi = max(khi, (unsigned)__shfl_xor((int)khi, m));
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s_tri[w] = ntri; s_vert[w] = nvert; s_min[w] = klo; s_max[w] = khi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned t = 0, vv = 0, a2 = 0xffffffffu, b2 = 0;
    for (int i = 0; i < kMcThreads / 64; ++i) { t += s_tri[i]; vv += s_vert[i]; a2 = min(a2, s_min[i]); b2 = max(b2, s_max[i]); }
    block_tot[blockIdx.x] = make_uint2(t, vv);
    block_minmax[blockIdx.x] = make_uint2(a2, b2);     // reduced by mc_scan_blocks (one hot atomic word would serialise)
  }
}

// exclusive scan of the per-block totals into block_base by one workgroup; grand totals into the header
__global__ __launch_bounds__(1024) void mc_scan_blocks(const uint2* __restrict__ block_sums, uint2* __restrict__ block_base,
                                                       const uint2* __restrict__ block_minmax, int nblocks, McHeader* hdr) {
  __shared__ uint2 s_wave[16];
  __shared__ uint2 s_carry;
  __shared__ unsigned s_lo[16], s_hi[16];
  if (threadIdx.x == 0) s_carry = make_uint2(0, 0);
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  unsigned klo = 0xffffffffu, khi = 0;
  for (int start = 0; start < nblocks; start += 1024) {
    const int i = start + threadIdx.x;
    uint2 v = i < nblocks ? block_sums[i] : make_uint2(0, 0);
    if (i < nblocks) { const uint2 mm = block_minmax[i]; klo = min(klo, mm.x); khi = max(khi, mm.y); }
    uint2 inc = v;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
      const unsigned a = __shfl_up(inc.x, m), b = __shfl_up(inc.y, m);
      if (lane >= m) { inc.x += a; inc.y += b; }
    }
    if (lane == 63) s_wave[w] = inc;
    __syncthreads();
    uint2 pre = s_carry;
    for (int k = 0; k < w; ++k) { pre.x += s_wave[k].x; pre.y += s_wave[k].y; }
    if (i < nblocks) block_base[i] = make_uint2(pre.x + inc.x - v.x, pre.y + inc.y - v.y);
    __syncthreads();
    if (threadIdx.x == 1023) s_carry = make_uint2(pre.x + inc.x, pre.y + inc.y);
    __syncthreads();
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    klo = min(klo, (unsigned)__shfl_xor((int)klo, m)); khi = max(khi, (unsigned)__shfl_xor((int)khi, m));
  }
  if (lane == 0) { s_lo[w] = klo; s_hi[w] = khi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 16; ++k) { klo = min(klo, s_lo[k]); khi = max(khi, s_hi[k]); }
    hdr->total_tris = s_carry.x; hdr->total_verts = s_carry.y; hdr->min_key = klo; hdr->max_key = khi;
  }
}

// exclusive scan of `val` over the workgroup's 256 threads, plus running carry
__device__ __forceinline__ unsigned block_excl_scan(unsigned val, unsigned* s_wave, unsigned& carry) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  unsigned inc = val;
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) {
    const unsigned a = __shfl_up(inc, m);
    if (lane >= m) inc += a;
  }
  if (lane == 63) s_wave[w] = inc;
  __syncthreads();
  unsigned pre = carry;
  for (int k = 0; k < w; ++k) pre += s_wave[k];
  const unsigned total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
  __syncthreads();
  carry += total;
  return pre + inc - val;
}

__device__ __forceinline__ size_t vid_slot(const McDims& d, int e, int x, int y, int z) {
  if (e == 12) return 4 * (((size_t)z * d.ny + y) * d.nx + x) + 3;
  const int ex = x + MC33_EDGE_DX(e), ey = y + MC33_EDGE_DY(e), ez = z + MC33_EDGE_DZ(e);
  return 4 * (((size_t)ez * d.ny + ey) * d.nx + ex) + MC33_EDGE_AXIS(e);
}

__global__ __launch_bounds__(kMcThreads) void mc_emit_verts(const float* __restrict__ vol, McDims d, double level,
                                                            const unsigned* __restrict__ code,
                                                            const uint2* __restrict__ block_tot,
                                                            const uint2* __restrict__ block_base, unsigned* __restrict__ vid,
                                                            float* __restrict__ verts) {
  __shared__ unsigned s_wave[kMcThreads / 64];
  if (block_tot[blockIdx.x].y == 0) return;             // no vertex is owned by this block's 1024 cell slots
  unsigned carry = block_base[blockIdx.x].y;
  const long long base = (long long)blockIdx.x * kMcChunk;
  for (int i = 0; i < kMcCellsPerThread; ++i) {
    const long long cell = base + i * kMcThreads + threadIdx.x;
    const unsigned cc = cell < d.nslots ? code[cell] : 0;
    unsigned id = block_excl_scan(code_nv(cc), s_wave, carry);
    if (code_nv(cc) == 0) continue;
    int x, y, z;
    cell_coords(d, cell, x, y, z);
    double v[8];
    load_corners(vol, d, x, y, z, level, v);
    const int off = code_off(cc), nt = code_nt(cc);
    unsigned seen = 0;
    for (int k = 0; k < 3 * nt; ++k) {
      const int e = kMcTiles[off + k];
      if (seen & (1u << e)) continue;
      seen |= 1u << e;
      if (!(e == 12 || owns_edge(e, x, y, z))) continue;
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
      float* o = verts + 3 * (size_t)id;
      o[0] = (float)((double)z + fz / ff);      // (axis0, axis1, axis2) = (z, y, x)
      o[1] = (float)((double)y + fy / ff);
      o[2] = (float)((double)x + fx / ff);
      vid[vid_slot(d, e, x, y, z)] = id;
      ++id;
    }
  }
}

__global__ __launch_bounds__(kMcThreads) void mc_emit_faces(McDims d, const unsigned* __restrict__ code,
                                                            const uint2* __restrict__ block_tot,
                                                            const uint2* __restrict__ block_base,
                                                            const unsigned* __restrict__ vid, int* __restrict__ faces) {
  __shared__ unsigned s_wave[kMcThreads / 64];
  if (block_tot[blockIdx.x].x == 0) return;             // no triangle in this block's 1024 cell slots
  unsigned carry = block_base[blockIdx.x].x;
  const long long base = (long long)blockIdx.x * kMcChunk;
  for (int i = 0; i < kMcCellsPerThread; ++i) {
    const long long cell = base + i * kMcThreads + threadIdx.x;
    const unsigned cc = cell < d.nslots ? code[cell] : 0;
    const unsigned tri0 = block_excl_scan(code_nt(cc), s_wave, carry);
    const int nt = code_nt(cc);
    if (nt == 0) continue;
    int x, y, z;
    cell_coords(d, cell, x, y, z);
    const int off = code_off(cc);
    for (int t = 0; t < nt; ++t) {
      int* f = faces + 3 * (size_t)(tri0 + t);
      // 'descent' orientation: the routine reverses every face
      f[2] = (int)vid[vid_slot(d, kMcTiles[off + 3 * t + 0], x, y, z)];
      f[1] = (int)vid[vid_slot(d, kMcTiles[off + 3 * t + 1], x, y, z)];
      f[0] = (int)vid[vid_slot(d, kMcTiles[off + 3 * t + 2], x, y, z)];
    }
  }
}

struct McLayout {
  McDims d;
  size_t off_code, off_sums, off_base, off_minmax, off_vid, total;
};

static bool mc_layout(int n0, int n1, int n2, McLayout& L) {
  if (n0 < 2 || n1 < 2 || n2 < 2) return false;
  if ((long long)n0 * n1 * n2 > (1ll << 31)) return false;
  McDims& d = L.d;
  d.nx = n2; d.ny = n1; d.nz = n0;
  d.cx = n2 - 1; d.cy = n1 - 1; d.cz = n0 - 1;
  d.cxp = (d.cx + 3) & ~3;
  d.nslots = (long long)d.cxp * d.cy * d.cz;
  d.nblocks = (int)((d.nslots + kMcChunk - 1) / kMcChunk);
  auto align = [](size_t v) { return (v + 255) & ~(size_t)255; };
  size_t o = align(sizeof(McHeader));
  L.off_code = o; o = align(o + sizeof(unsigned) * (size_t)d.nblocks * kMcChunk);
  L.off_sums = o; o = align(o + sizeof(uint2) * (size_t)d.nblocks);
  L.off_base = o; o = align(o + sizeof(uint2) * (size_t)d.nblocks);
  L.off_minmax = o; o = align(o + sizeof(uint2) * (size_t)d.nblocks);
  L.off_vid = o; o = align(o + sizeof(unsigned) * 4 * (size_t)n0 * n1 * n2);
  L.total = o;
  return true;
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

int asdf_mc_count(const float* vol, int32_t n0, int32_t n1, int32_t n2, double level, void* ws, size_t ws_bytes,
                  uint32_t* num_verts, uint32_t* num_faces, void* stream) {
  McLayout L;
  if (!vol || !ws || !num_verts || !num_faces || !mc_layout(n0, n1, n2, L)) return ASDF_EINVAL;
  if (ws_bytes < L.total) return ASDF_ENOSPC;
  hipStream_t st = (hipStream_t)stream;
  char* w = (char*)ws;
  McHeader* hdr = (McHeader*)w;
  uint4* code4 = (uint4*)(w + L.off_code);
  uint2* sums = (uint2*)(w + L.off_sums);
  uint2* base = (uint2*)(w + L.off_base);
  uint2* minmax = (uint2*)(w + L.off_minmax);
  hipLaunchKernelGGL(mc_classify, dim3(L.d.nblocks), dim3(kMcThreads), 0, st, vol, L.d, level, code4, sums, minmax);
  hipLaunchKernelGGL(mc_scan_blocks, dim3(1), dim3(1024), 0, st, sums, base, minmax, L.d.nblocks, hdr);
  ASDF_HIP(hipGetLastError());
  McHeader h;
  ASDF_HIP(hipMemcpyAsync(&h, hdr, sizeof(h), hipMemcpyDeviceToHost, st));
  ASDF_HIP(hipStreamSynchronize(st));
  *num_verts = h.total_verts;
  *num_faces = h.total_tris;
  // skimage: ValueError when level is outside [min, max]; RuntimeError when nothing was produced
  const double lo = key_float(h.min_key), hi = key_float(h.max_key);
  if (level < lo || level > hi) return ASDF_ERANGE;
  if (h.total_verts == 0) return ASDF_ENOSURF;
  return ASDF_OK;
}

int asdf_mc_emit(const float* vol, int32_t n0, int32_t n1, int32_t n2, double level, void* ws, size_t ws_bytes,
                 float* verts, int32_t* faces, void* stream) {
  McLayout L;
  if (!vol || !ws || !verts || !faces || !mc_layout(n0, n1, n2, L)) return ASDF_EINVAL;
  if (ws_bytes < L.total) return ASDF_ENOSPC;
  hipStream_t st = (hipStream_t)stream;
  char* w = (char*)ws;
  unsigned* code = (unsigned*)(w + L.off_code);
  uint2* sums = (uint2*)(w + L.off_sums);
  uint2* base = (uint2*)(w + L.off_base);
  unsigned* vid = (unsigned*)(w + L.off_vid);
  hipLaunchKernelGGL(mc_emit_verts, dim3(L.d.nblocks), dim3(kMcThreads), 0, st, vol, L.d, level, code, sums, base, vid, verts);
  hipLaunchKernelGGL(mc_emit_faces, dim3(L.d.nblocks), dim3(kMcThreads), 0, st, L.d, code, sums, base, vid, faces);
  ASDF_HIP(hipGetLastError());
  return ASDF_OK;
}

}  // extern "C"
