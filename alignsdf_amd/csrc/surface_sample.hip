// K9: the sampling half of the reference's eval mode on the device - `trimesh.sample.sample_surface(mesh, 30000)` of the predicted
// hand surface (utils/mesh.py:386-389, deep_sdf/metrics/icp_trans_scale.py:19-23) and the normalisation of the source samples onto the
// target's centroid / RMS radius (ICP_T_S.sample_mesh, icp_trans_scale.py:25-31) - as four launches behind the largest-component
// filter (K8) and in front of the ICP (K7), none of which waits for the host.
//
// The sampler is the seeded, area-QUANTISED sampler of alignsdf_amd/surface_sampling.py (the reference's is unseeded; see there), and
// every output is bit-identical to that host sampler: q = rint(area / max(area) * 2^31) per face - fp64 areas from products and sums
// that are rounded one by one, in the host's order (this file is built with -ffp-contract=off and says _rn again where it matters) -
// an exact integer inclusive sum, one binary search per draw and the barycentric point, again in the host's order of operations.
// The vertices arrive in LATTICE units as marching cubes / K8 leave them (fp32) and are placed - v * voxel + origin, an fp32 multiply
// and an fp32 add per coordinate, the exporter's arithmetic (utils/mesh.py:360-369) - on the fly; faces beyond the device-side face
// count (K8's output keeps the input's capacity) have area zero and are never picked.
//
// Until round 5 this was ~50 torch launches for the sampler and ~25 for the normalisation per hand mesh (alignsdf_amd/icp.py keeps
// that form for inputs of other types; tests/test_gpu_icp.py compares all three).  HBM traffic is a few MB: launch-bound work.
#include <hip/hip_runtime.h>

#include <cmath>

#include "../../include/alignsdf_hip.h"
#include "common.h"

namespace asdf {

constexpr double kAreaQuantum = 2147483648.0;       // surface_sampling._AREA_QUANTUM

struct Placement { float vs, o[3]; int place; };

__device__ __forceinline__ double placed_coord(const float* __restrict__ verts, int v, int a, const Placement& pl) {
  const float x = verts[3 * (size_t)v + a];
  return (double)(pl.place ? __fadd_rn(__fmul_rn(x, pl.vs), pl.o[a]) : x);
}

// fp64 area per face (0 beyond the live faces) + the maximum (non-negative doubles order like their bit patterns)
__global__ __launch_bounds__(256) void face_area_kernel(const float* __restrict__ verts, const int* __restrict__ faces, int faces_cap,
                                                        const int* __restrict__ num_faces, Placement pl, double* __restrict__ area,
                                                        unsigned long long* amax_bits) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  const int live = num_faces ? *num_faces : faces_cap;
  double ar = 0.0;
  if (f < faces_cap) {
    if (f < live) {
      const int i0 = faces[3 * (size_t)f], i1 = faces[3 * (size_t)f + 1], i2 = faces[3 * (size_t)f + 2];
      double a[3], e1[3], e2[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        a[k] = placed_coord(verts, i0, k, pl);
        e1[k] = __dsub_rn(placed_coord(verts, i1, k, pl), a[k]);
        e2[k] = __dsub_rn(placed_coord(verts, i2, k, pl), a[k]);
      }
      const double cx = __dsub_rn(__dmul_rn(e1[1], e2[2]), __dmul_rn(e1[2], e2[1]));
      const double cy = __dsub_rn(__dmul_rn(e1[2], e2[0]), __dmul_rn(e1[0], e2[2]));
      const double cz = __dsub_rn(__dmul_rn(e1[0], e2[1]), __dmul_rn(e1[1], e2[0]));
      ar = __dmul_rn(0.5, __dsqrt_rn(__dadd_rn(__dadd_rn(__dmul_rn(cx, cx), __dmul_rn(cy, cy)), __dmul_rn(cz, cz))));
    }
    area[f] = ar;
  }
  unsigned long long b = (unsigned long long)__double_as_longlong(ar);
  if (ar != ar) b = 0x7ff8000000000000ull;          // (a NaN area poisons the maximum, as torch's max does)
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { const unsigned long long o = __shfl_xor(b, m); b = o > b ? o : b; }
  if ((threadIdx.x & 63) == 0 && b) atomicMax(amax_bits, b);
}

// q = rint(area / max * 2^31) and its inclusive sum, by ONE workgroup (a few 10^5 faces: ~0.05 ms; a multi-workgroup scan would be three
// launches of about that much dispatch).  cum[f] = q[0] + .. + q[f]
__global__ __launch_bounds__(1024) void quantise_scan_kernel(const double* __restrict__ area, int faces_cap, const unsigned long long* __restrict__ amax_bits,
                                                             long long* __restrict__ cum) {
  __shared__ long long wsum[16];
  __shared__ long long carry;
  const double amax = __longlong_as_double((long long)*amax_bits);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  constexpr int kPer = 4;
  for (int base = 0; base < faces_cap; base += 1024 * kPer) {
    long long q[kPer], run = 0;
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const int f = base + threadIdx.x * kPer + k;
      long long v = 0;
      if (f < faces_cap && amax > 0.0) v = (long long)rint(__dmul_rn(__ddiv_rn(area[f], amax), kAreaQuantum));     // (rint: half to even, torch.round; a mesh of zero area has no weights)
      run += v;
      q[k] = run;
    }
    long long incl = run;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const long long t = __shfl_up(incl, d); if (lane >= d) incl += t; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    long long before = carry + (incl - run);
    for (int w = 0; w < wave; ++w) before += wsum[w];
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const int f = base + threadIdx.x * kPer + k;
      if (f < faces_cap) cum[f] = before + q[k];
    }
    __syncthreads();
    if (threadIdx.x == 1023) carry = before + run;
    __syncthreads();
  }
}

// one draw per thread: face = first f with cum[f] > floor(u * total) (searchsorted right=True), clamped; point = a + (b - a) r0 + (c - a) r1
__global__ __launch_bounds__(256) void draw_points_kernel(const float* __restrict__ verts, const int* __restrict__ faces, int faces_cap,
                                                          const int* __restrict__ num_faces, Placement pl, const long long* __restrict__ cum,
                                                          const double* __restrict__ u, const double* __restrict__ r, int count,
                                                          double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const long long total = cum[faces_cap - 1];
  const long long target = (long long)floor(__dmul_rn(u[i], (double)total));
  int lo = 0, hi = faces_cap;                          // first index with cum > target
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (cum[mid] > target) hi = mid; else lo = mid + 1;
  }
  int f = lo < faces_cap - 1 ? lo : faces_cap - 1;
  const int live = num_faces ? *num_faces : faces_cap;
  int i0 = 0, i1 = 0, i2 = 0;                           // (a pick beyond the live faces - only when total == 0 - is the degenerate face the host form uses)
  if (f < live) { i0 = faces[3 * (size_t)f]; i1 = faces[3 * (size_t)f + 1]; i2 = faces[3 * (size_t)f + 2]; }
  const double r0 = r[2 * (size_t)i], r1 = r[2 * (size_t)i + 1];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double a = placed_coord(verts, i0, k, pl), b = placed_coord(verts, i1, k, pl), c = placed_coord(verts, i2, k, pl);
    out[3 * (size_t)i + k] = __dadd_rn(__dadd_rn(a, __dmul_rn(__dsub_rn(b, a), r0)), __dmul_rn(__dsub_rn(c, a), r1));
  }
}

// ---- ICP_T_S.sample_mesh's normalisation (icp_trans_scale.py:25-31): offset = mean, scale = sqrt(sum |p - offset|^2 / n) of both
// sets; the source set is moved onto the target's.  One workgroup, sums in a fixed order (thread-strided partial sums, then a
// fixed tree): reproducible run to run; against numpy's pairwise sums a 1e-16-class difference in the four statistics.
__device__ __forceinline__ double block_sum_1024(double v, double* sh) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < 16; ++w) t += sh[w];
  return t;
}
__device__ void set_statistics(const double* __restrict__ p, int n, double* sh, double offset[3], double& scale) {
  double s[3] = {0.0, 0.0, 0.0};
  for (int i = threadIdx.x; i < n; i += 1024)
#pragma unroll
    for (int a = 0; a < 3; ++a) s[a] += p[3 * (size_t)i + a];
#pragma unroll
  for (int a = 0; a < 3; ++a) offset[a] = block_sum_1024(s[a], sh) / (double)n;
  double q = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024)
#pragma unroll
    for (int a = 0; a < 3; ++a) { const double d = p[3 * (size_t)i + a] - offset[a]; q += d * d; }
  scale = sqrt(block_sum_1024(q, sh) / (double)n);
}
__global__ __launch_bounds__(1024) void normalise_source_kernel(const double* __restrict__ ps, int ns, const double* __restrict__ pt, int nt,
                                                                double* __restrict__ src_out, double* stats_mapped) {
  __shared__ double sh[16];
  double os[3], ot[3], ss, st;
  set_statistics(ps, ns, sh, os, ss);
  set_statistics(pt, nt, sh, ot, st);
  for (int i = threadIdx.x; i < ns; i += 1024)
#pragma unroll
    for (int a = 0; a < 3; ++a) src_out[3 * (size_t)i + a] = (ps[3 * (size_t)i + a] - os[a]) / ss * st + ot[a];
  if (threadIdx.x == 0 && stats_mapped) {
    stats_mapped[0] = os[0]; stats_mapped[1] = os[1]; stats_mapped[2] = os[2]; stats_mapped[3] = ss;
    stats_mapped[4] = ot[0]; stats_mapped[5] = ot[1]; stats_mapped[6] = ot[2]; stats_mapped[7] = st;
  }
}

}  // namespace asdf

using namespace asdf;

extern "C" {

int asdf_sample_surface_workspace_bytes(int32_t faces_cap, size_t* bytes) {
  if (!bytes || faces_cap < 1) return ASDF_EINVAL;
  *bytes = 256 + (size_t)faces_cap * (sizeof(double) + sizeof(long long));
  return ASDF_OK;
}

int asdf_sample_surface(const float* verts_dev, const int32_t* faces_dev, int32_t faces_cap, const int32_t* num_faces_dev,
                        int32_t place, float voxel_size, const float origin[3], const double* u_dev, const double* r_dev, int32_t count,
                        double* points_dev, void* workspace_dev, size_t workspace_bytes, void* stream) {
  if (!verts_dev || !faces_dev || !u_dev || !r_dev || !points_dev || !workspace_dev || faces_cap < 1 || count < 1) return ASDF_EINVAL;
  if (place && !origin) return ASDF_EINVAL;
  size_t need = 0;
  asdf_sample_surface_workspace_bytes(faces_cap, &need);
  if (workspace_bytes < need || ((size_t)workspace_dev & 7)) return ASDF_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  Placement pl;
  pl.place = place ? 1 : 0; pl.vs = voxel_size;
  for (int a = 0; a < 3; ++a) pl.o[a] = place ? origin[a] : 0.0f;
  char* ws = (char*)workspace_dev;
  unsigned long long* amax = (unsigned long long*)ws;
  double* area = (double*)(ws + 256);
  long long* cum = (long long*)(ws + 256 + (size_t)faces_cap * sizeof(double));
  ASDF_HIP(hipMemsetAsync(amax, 0, sizeof(unsigned long long), st));
  hipLaunchKernelGGL(face_area_kernel, dim3((faces_cap + 255) / 256), dim3(256), 0, st, verts_dev, faces_dev, faces_cap, num_faces_dev, pl, area, amax);
  hipLaunchKernelGGL(quantise_scan_kernel, dim3(1), dim3(1024), 0, st, area, faces_cap, amax, cum);
  hipLaunchKernelGGL(draw_points_kernel, dim3((count + 255) / 256), dim3(256), 0, st, verts_dev, faces_dev, faces_cap, num_faces_dev, pl, cum,
                     u_dev, r_dev, count, points_dev);
  ASDF_HIP(hipGetLastError());
  return ASDF_OK;
}

int asdf_icp_normalise(const double* src_dev, int32_t ns, const double* tgt_dev, int32_t nt, double* src_out_dev, double* stats_mapped,
                       void* stream) {
  if (!src_dev || !tgt_dev || !src_out_dev || ns < 1 || nt < 1) return ASDF_EINVAL;
  hipLaunchKernelGGL(normalise_source_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, src_dev, ns, tgt_dev, nt, src_out_dev, stats_mapped);
  ASDF_HIP(hipGetLastError());
  return ASDF_OK;
}

}  // extern "C"
