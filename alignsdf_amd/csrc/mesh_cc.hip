// K8: the largest-component filter of utils/mesh.py:371-381 on the device.
//
// The reference builds a trimesh of the marching-cubes output, calls trimesh.graph.split (only_watertight=True) and, when
// more than one sub-mesh comes back, keeps the one with the largest area.  alignsdf_amd/mesh_post.py restates those
// semantics on the host (numpy / scipy, ~30 ms per 200 k-face surface - more than a third of a decoder pass); this is the
// same computation as seven small kernels and two hipCUB primitives, without any host synchronisation:
//   faces are adjacent when they share an edge that belongs to exactly two faces; components are the connected
//   components of that adjacency; a component qualifies if it has >= 4 faces and none of its edges is shared by a number
//   of faces other than two (watertight); with fewer than two qualifying components the mesh is returned unchanged,
//   otherwise the qualifying component of largest area (first one on ties) is returned with its vertices compacted in
//   ascending original order and its faces in original order.
// Steps: edge keys (lo * V + hi) per face -> radix sort (hipCUB) -> runs of equal keys: a run of two unites its faces
// (lock-free union-find, larger root hooked under the smaller, so a component's root is its first face), any other run
// length marks its faces open -> per-root area / size / open flags -> selection (one workgroup) -> keep flags ->
// exclusive scans (hipCUB) -> compaction.  The area is taken on the placed vertices of utils/mesh.py:360-363
// (fp32 spacing * v + origin, as alignsdf_amd.utils.mesh.place_vertices computes them) in fp64.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdint>

#include "../../include/alignsdf_hip.h"
#include "common.h"

namespace asdf {

struct CcHeader {
  int valid;        // qualifying components
  int best;         // root (= first face) of the component that is kept when valid > 1
  int out_v, out_f;
  int open_dropped;   // components of >= 4 faces that do not qualify because an edge is shared by a number of faces other than two -
                      // the only place where trimesh's fill_holes (not reproduced: PARITY UNPINNED) could have changed the outcome
  int small_dropped;  // components of fewer than 4 faces (graph.split's min_len)
};

__global__ void cc_edges_kernel(const int* __restrict__ faces, int F, unsigned long long V, unsigned long long* keys, int* owner,
                                int* parent, int* open_face, double* area, int* size, int* open_comp) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  const int v[3] = {faces[3 * f], faces[3 * f + 1], faces[3 * f + 2]};
#pragma unroll
  for (int e = 0; e < 3; ++e) {
    const unsigned long long a = (unsigned)v[e], b = (unsigned)v[(e + 1) % 3];
    keys[3 * (size_t)f + e] = (a < b ? a : b) * V + (a < b ? b : a);
    owner[3 * (size_t)f + e] = f;
  }
  parent[f] = f;
  open_face[f] = 0;
  area[f] = 0.0;
  size[f] = 0;
  open_comp[f] = 0;
}

__device__ __forceinline__ int cc_find(int* parent, int x) {
  // path halving; parents only ever decrease, so a stale read still leads towards the root
  while (true) {
    const int p = ((volatile int*)parent)[x];
    if (p == x) return x;
    const int g = ((volatile int*)parent)[p];
    if (g != p) parent[x] = g;
    x = p;
  }
}

__device__ __forceinline__ void cc_union(int* parent, int a, int b) {
  while (true) {
    a = cc_find(parent, a);
    b = cc_find(parent, b);
    if (a == b) return;
    if (a < b) { const int t = a; a = b; b = t; }      // hook the larger root under the smaller
    if (atomicCAS(&parent[a], a, b) == a) return;
  }
}

__global__ void cc_union_kernel(const unsigned long long* __restrict__ keys, const int* __restrict__ owner, long long E, int* parent,
                                int* open_face) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= E) return;
  const unsigned long long k = keys[i];
  if (i > 0 && keys[i - 1] == k) return;               // not the start of a run
  long long j = i + 1;
  while (j < E && keys[j] == k) ++j;
  if (j - i == 2) {
    cc_union(parent, owner[i], owner[i + 1]);
  } else {
    for (long long t = i; t < j; ++t) open_face[owner[t]] = 1;   // boundary edge or non-manifold edge
  }
}

__global__ void cc_stats_kernel(const float* __restrict__ verts, const int* __restrict__ faces, int F, float vs, float o0, float o1,
                                float o2, int* parent, const int* __restrict__ open_face, int* label, double* area, int* size,
                                int* open_comp) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = f < F;
  int root = -1, open = 0;
  double a = 0.0;
  if (live) {
  root = cc_find(parent, f);
  label[f] = root;
  double p[3][3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float* v = verts + 3 * (size_t)faces[3 * f + c];
    // place_vertices: fp32 spacing multiply, fp32 origin add - then the fp64 area of trimesh
    p[c][0] = (double)__fadd_rn(o0, __fmul_rn(v[0], vs));
    p[c][1] = (double)__fadd_rn(o1, __fmul_rn(v[1], vs));
    p[c][2] = (double)__fadd_rn(o2, __fmul_rn(v[2], vs));
  }
  const double ux = p[1][0] - p[0][0], uy = p[1][1] - p[0][1], uz = p[1][2] - p[0][2];
  const double wx = p[2][0] - p[0][0], wy = p[2][1] - p[0][1], wz = p[2][2] - p[0][2];
  const double cx = uy * wz - uz * wy, cy = uz * wx - ux * wz, cz = ux * wy - uy * wx;
  a = 0.5 * sqrt(cx * cx + cy * cy + cz * cz);
  open = open_face[f];
  }
  // A surface is mostly ONE component: 200 k atomics on a single root's words serialise for milliseconds.  Faces of a
  // block that share the root of the block's first face are reduced in the block first; the others go straight to memory.
  __shared__ int s_root;
  __shared__ double s_a[256 / 64];
  __shared__ int s_n[256 / 64], s_o[256 / 64];
  if (threadIdx.x == 0) s_root = root;
  __syncthreads();
  const bool mine = live && root == s_root;
  if (live && !mine) {
    atomicAdd(&area[root], a);
    atomicAdd(&size[root], 1);
    if (open) atomicOr(&open_comp[root], 1);
  }
  double ra = mine ? a : 0.0;
  int rn = mine ? 1 : 0, ro = mine ? open : 0;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { ra += __shfl_xor(ra, m); rn += __shfl_xor(rn, m); ro |= __shfl_xor(ro, m); }
  if ((threadIdx.x & 63) == 0) { s_a[threadIdx.x >> 6] = ra; s_n[threadIdx.x >> 6] = rn; s_o[threadIdx.x >> 6] = ro; }
  __syncthreads();
  if (threadIdx.x == 0 && s_root >= 0) {
    double ta = 0.0;
    int tn = 0, to = 0;
    for (int w = 0; w < 256 / 64; ++w) { ta += s_a[w]; tn += s_n[w]; to |= s_o[w]; }
    atomicAdd(&area[s_root], ta);
    atomicAdd(&size[s_root], tn);
    if (to) atomicOr(&open_comp[s_root], 1);
  }
}

__global__ __launch_bounds__(1024) void cc_select_kernel(const int* __restrict__ label, int F, const double* __restrict__ area,
                                                         const int* __restrict__ size, const int* __restrict__ open_comp,
                                                         CcHeader* hdr) {
  __shared__ double s_area[1024];
  __shared__ int s_root[1024], s_valid[1024];
  __shared__ int s_open, s_small;
  if (threadIdx.x == 0) { s_open = 0; s_small = 0; }
  __syncthreads();
  double best_a = -1.0;
  int best_r = 0x7fffffff, valid = 0, n_open = 0, n_small = 0;
  for (int f = threadIdx.x; f < F; f += 1024) {
    if (label[f] != f) continue;                        // roots only
    if (size[f] < 4) { ++n_small; continue; }           // graph.split: min_len 4 ...
    if (open_comp[f]) { ++n_open; continue; }           // ... only_watertight
    ++valid;
    const double a = area[f];
    if (a > best_a || (a == best_a && f < best_r)) { best_a = a; best_r = f; }   // first maximum in order of first faces
  }
  s_area[threadIdx.x] = best_a; s_root[threadIdx.x] = best_r; s_valid[threadIdx.x] = valid;
  if (n_open) atomicAdd(&s_open, n_open);
  if (n_small) atomicAdd(&s_small, n_small);
  __syncthreads();
  for (int m = 512; m >= 1; m >>= 1) {
    if ((int)threadIdx.x < m) {
      const double a = s_area[threadIdx.x + m];
      const int r = s_root[threadIdx.x + m];
      if (a > s_area[threadIdx.x] || (a == s_area[threadIdx.x] && r < s_root[threadIdx.x])) { s_area[threadIdx.x] = a; s_root[threadIdx.x] = r; }
      s_valid[threadIdx.x] += s_valid[threadIdx.x + m];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { hdr->valid = s_valid[0]; hdr->best = s_root[0]; hdr->open_dropped = s_open; hdr->small_dropped = s_small; }
}

__global__ void cc_fill_kernel(int* a, int n, int value) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = value;
}

__global__ void cc_mark_kernel(const int* __restrict__ faces, const int* __restrict__ label, int F, const CcHeader* hdr, int* keep,
                               int* used) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  const bool all = hdr->valid <= 1;                     // `if len(split_mesh) > 1` of the reference: otherwise unchanged
  const int k = all || label[f] == hdr->best;
  keep[f] = k;
  if (k && !all) { used[faces[3 * f]] = 1; used[faces[3 * f + 1]] = 1; used[faces[3 * f + 2]] = 1; }
}

__global__ void cc_fill_all_if_unchanged_kernel(int* used, int V, const CcHeader* hdr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < V && hdr->valid <= 1) used[i] = 1;
}

__global__ void cc_counts_kernel(const CcHeader* hdr, int* counts) {
  counts[0] = hdr->out_v; counts[1] = hdr->out_f; counts[2] = hdr->valid; counts[3] = hdr->best;
  counts[4] = hdr->open_dropped; counts[5] = hdr->small_dropped; counts[6] = 0; counts[7] = 0;
}

__global__ void cc_compact_kernel(const float* __restrict__ verts, const int* __restrict__ faces, int V, int F,
                                  const int* __restrict__ keep, const int* __restrict__ fpos, const int* __restrict__ used,
                                  const int* __restrict__ vpos, float* out_verts, int* out_faces, CcHeader* hdr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < F && keep[i]) {
    const int o = fpos[i];
    out_faces[3 * o] = vpos[faces[3 * i]]; out_faces[3 * o + 1] = vpos[faces[3 * i + 1]]; out_faces[3 * o + 2] = vpos[faces[3 * i + 2]];
  }
  if (i < V && used[i]) {
    const int o = vpos[i];
    out_verts[3 * o] = verts[3 * i]; out_verts[3 * o + 1] = verts[3 * i + 1]; out_verts[3 * o + 2] = verts[3 * i + 2];
  }
  if (i == 0) {
    hdr->out_f = F ? fpos[F - 1] + keep[F - 1] : 0;
    hdr->out_v = V ? vpos[V - 1] + used[V - 1] : 0;
  }
}

struct CcLayout {
  size_t keys_in, keys_out, owner_in, owner_out, parent, open_face, label, area, size, open_comp, keep, fpos, used, vpos, hdr, cub,
      cub_bytes, bytes;
};

static CcLayout cc_layout(int V, int F) {
  CcLayout l;
  size_t off = 0;
  auto take = [&](size_t n) { const size_t o = off; off += (n + 255) & ~(size_t)255; return o; };
  const size_t E = 3 * (size_t)F;
  l.hdr = take(sizeof(CcHeader));
  l.keys_in = take(E * 8); l.keys_out = take(E * 8); l.owner_in = take(E * 4); l.owner_out = take(E * 4);
  l.parent = take((size_t)F * 4); l.open_face = take((size_t)F * 4); l.label = take((size_t)F * 4); l.area = take((size_t)F * 8);
  l.size = take((size_t)F * 4); l.open_comp = take((size_t)F * 4); l.keep = take((size_t)F * 4); l.fpos = take((size_t)F * 4);
  l.used = take((size_t)V * 4); l.vpos = take((size_t)V * 4);
  size_t sort_bytes = 0, scan_f = 0, scan_v = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, (const unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                     (const int*)nullptr, (int*)nullptr, (int)E, 0, 64, (hipStream_t)0);
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_f, (const int*)nullptr, (int*)nullptr, F, (hipStream_t)0);
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_v, (const int*)nullptr, (int*)nullptr, V, (hipStream_t)0);
  l.cub_bytes = sort_bytes > scan_f ? sort_bytes : scan_f;
  if (scan_v > l.cub_bytes) l.cub_bytes = scan_v;
  l.cub = take(l.cub_bytes + 256);
  l.bytes = off;
  return l;
}

}  // namespace asdf

using namespace asdf;

extern "C" {

int asdf_mesh_cc_workspace_bytes(int32_t num_verts, int32_t num_faces, size_t* bytes) {
  if (!bytes || num_verts < 1 || num_faces < 1) return ASDF_EINVAL;
  *bytes = cc_layout(num_verts, num_faces).bytes;
  return ASDF_OK;
}

int asdf_mesh_largest_component(const float* verts_dev, int32_t V, const int32_t* faces_dev, int32_t F, float voxel_size,
                                const float origin[3], void* workspace_dev, size_t workspace_bytes, float* out_verts_dev,
                                int32_t* out_faces_dev, int32_t* counts_dev, void* stream) {
  if (!verts_dev || !faces_dev || !origin || !workspace_dev || !out_verts_dev || !out_faces_dev || !counts_dev || V < 1 || F < 1)
    return ASDF_EINVAL;
  const CcLayout l = cc_layout(V, F);
  if (workspace_bytes < l.bytes) return ASDF_ENOSPC;
  hipStream_t st = (hipStream_t)stream;
  char* ws = (char*)workspace_dev;
  auto at = [&](size_t o) { return (void*)(ws + o); };
  unsigned long long* keys_in = (unsigned long long*)at(l.keys_in);
  unsigned long long* keys_out = (unsigned long long*)at(l.keys_out);
  int *owner_in = (int*)at(l.owner_in), *owner_out = (int*)at(l.owner_out), *parent = (int*)at(l.parent);
  int *open_face = (int*)at(l.open_face), *label = (int*)at(l.label), *size = (int*)at(l.size), *open_comp = (int*)at(l.open_comp);
  int *keep = (int*)at(l.keep), *fpos = (int*)at(l.fpos), *used = (int*)at(l.used), *vpos = (int*)at(l.vpos);
  double* area = (double*)at(l.area);
  CcHeader* hdr = (CcHeader*)at(l.hdr);
  const int T = 256;
  const int gf = (F + T - 1) / T, gv = (V + T - 1) / T;
  const long long E = 3LL * F;
  hipLaunchKernelGGL(cc_edges_kernel, dim3(gf), dim3(T), 0, st, faces_dev, F, (unsigned long long)V, keys_in, owner_in, parent, open_face,
                     area, size, open_comp);
  int end_bit = 1;
  while (end_bit < 64 && ((unsigned long long)V * (unsigned long long)V >> end_bit)) ++end_bit;
  size_t cub_bytes = l.cub_bytes;
  ASDF_HIP(hipcub::DeviceRadixSort::SortPairs(at(l.cub), cub_bytes, keys_in, keys_out, owner_in, owner_out, (int)E, 0, end_bit, st));
  hipLaunchKernelGGL(cc_union_kernel, dim3((unsigned)((E + T - 1) / T)), dim3(T), 0, st, keys_out, owner_out, E, parent, open_face);
  hipLaunchKernelGGL(cc_stats_kernel, dim3(gf), dim3(T), 0, st, verts_dev, faces_dev, F, voxel_size, origin[0], origin[1], origin[2],
                     parent, open_face, label, area, size, open_comp);
  hipLaunchKernelGGL(cc_select_kernel, dim3(1), dim3(1024), 0, st, label, F, area, size, open_comp, hdr);
  // vertices: all kept when the mesh is returned unchanged, otherwise only those the kept faces reference
  hipLaunchKernelGGL(cc_fill_kernel, dim3(gv), dim3(T), 0, st, used, V, 0);
  hipLaunchKernelGGL(cc_mark_kernel, dim3(gf), dim3(T), 0, st, faces_dev, label, F, hdr, keep, used);
  hipLaunchKernelGGL(cc_fill_all_if_unchanged_kernel, dim3(gv), dim3(T), 0, st, used, V, hdr);
  cub_bytes = l.cub_bytes;
  ASDF_HIP(hipcub::DeviceScan::ExclusiveSum(at(l.cub), cub_bytes, keep, fpos, F, st));
  cub_bytes = l.cub_bytes;
  ASDF_HIP(hipcub::DeviceScan::ExclusiveSum(at(l.cub), cub_bytes, used, vpos, V, st));
  const int gm = gf > gv ? gf : gv;
  hipLaunchKernelGGL(cc_compact_kernel, dim3(gm), dim3(T), 0, st, verts_dev, faces_dev, V, F, keep, fpos, used, vpos, out_verts_dev,
                     out_faces_dev, hdr);
  ASDF_HIP(hipGetLastError());
  // counts[8]: [0] kept vertices, [1] kept faces, [2] qualifying components, [3] root (first face) of the kept component, [4] open
  // components of >= 4 faces that did not qualify, [5] components of < 4 faces, [6..7] zero
  hipLaunchKernelGGL(cc_counts_kernel, dim3(1), dim3(1), 0, st, hdr, counts_dev);
  ASDF_HIP(hipGetLastError());
  return ASDF_OK;
}

}  // extern "C"
