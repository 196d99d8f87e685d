// K7: translate + scale ICP of the reference's eval mode (deep_sdf/metrics/icp_trans_scale.py:33-113, ICP_T_S.run_icp_f,
// called from utils/mesh.py:385-395 with 30 000 surface samples per mesh and up to 100 iterations).
//
// Per iteration the reference does two exact nearest-neighbour sweeps through static KD-trees and a 4-unknown linear
// least-squares solve.  Here a sweep is one brute-force kernel in fp64 (same arithmetic type as the reference, so the
// neighbour assignments are the KD-tree's): one query per thread, the reference set streamed through LDS in tiles, the
// squared-error and least-squares sums reduced per workgroup.  The host sums the per-workgroup partials in a fixed
// order, applies the reference's stopping rules and solves the 4 unknowns in closed form.
// Work per sweep: nq x nr distance evaluations (9e8 at 30k x 30k): VALU-bound, ~1 ms; HBM traffic is negligible.
#include <hip/hip_runtime.h>

#include <cmath>
#include <vector>

#include "../../include/alignsdf_hip.h"
#include "common.h"

namespace asdf {

constexpr int kIcpThreads = 128;
constexpr int kIcpTile = 1024;          // reference points per LDS tile (24 KiB of fp64)
constexpr int kIcpSums = 9;             // err, sum X (3), sum Y (3), sum X.Y, sum X.X

// dir 0: queries = source samples p,  q = p * s + t,      reference set = target;  X = p,        Y = nearest target
// dir 1: queries = target samples P,  q = (P - t) / s,    reference set = source;  X = nearest p, Y = P
__global__ __launch_bounds__(kIcpThreads) void icp_sweep_kernel(const double* __restrict__ qpts, int nq,
                                                                const double* __restrict__ rpts, int nr, int dir, double s,
                                                                double t0, double t1, double t2, double* __restrict__ partials) {
  __shared__ double tile[kIcpTile * 3];
  __shared__ double red[kIcpThreads / 64][kIcpSums];
  const int i = blockIdx.x * kIcpThreads + threadIdx.x;
  const bool live = i < nq;
  double p0 = 0, p1 = 0, p2 = 0;
  if (live) { p0 = qpts[3 * i]; p1 = qpts[3 * i + 1]; p2 = qpts[3 * i + 2]; }
  double q0, q1, q2;
  if (dir == 0) { q0 = p0 * s + t0; q1 = p1 * s + t1; q2 = p2 * s + t2; }
  else { q0 = (p0 - t0) / s; q1 = (p1 - t1) / s; q2 = (p2 - t2) / s; }
  double best = INFINITY, b0 = 0, b1 = 0, b2 = 0;
  for (int base = 0; base < nr; base += kIcpTile) {
    const int n = min(kIcpTile, nr - base);
    __syncthreads();
    for (int k = threadIdx.x; k < n * 3; k += kIcpThreads) tile[k] = rpts[(size_t)base * 3 + k];
    __syncthreads();
#pragma unroll 4
    for (int j = 0; j < n; ++j) {
      const double r0 = tile[3 * j], r1 = tile[3 * j + 1], r2 = tile[3 * j + 2];
      const double d0 = q0 - r0, d1 = q1 - r1, d2 = q2 - r2;
      const double d = d0 * d0 + d1 * d1 + d2 * d2;
      if (d < best) { best = d; b0 = r0; b1 = r1; b2 = r2; }      // first minimum wins on exact ties
    }
  }
  double v[kIcpSums] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (live) {
    if (dir == 0) {
      v[0] = best;                                   // |q - ct|^2
      v[1] = p0; v[2] = p1; v[3] = p2; v[4] = b0; v[5] = b1; v[6] = b2;
      v[7] = p0 * b0 + p1 * b1 + p2 * b2; v[8] = p0 * p0 + p1 * p1 + p2 * p2;
    } else {
      const double c0 = b0 * s + t0, c1 = b1 * s + t1, c2 = b2 * s + t2;     // nearest source sample, transformed
      v[0] = (p0 - c0) * (p0 - c0) + (p1 - c1) * (p1 - c1) + (p2 - c2) * (p2 - c2);
      v[1] = b0; v[2] = b1; v[3] = b2; v[4] = p0; v[5] = p1; v[6] = p2;
      v[7] = b0 * p0 + b1 * p1 + b2 * p2; v[8] = b0 * b0 + b1 * b1 + b2 * b2;
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
}

}  // namespace asdf

using namespace asdf;

extern "C" {

int asdf_icp_workspace_bytes(int32_t ns, int32_t nt, size_t* bytes) {
  if (!bytes || ns < 1 || nt < 1) return ASDF_EINVAL;
  const size_t blocks = (size_t)(ns + kIcpThreads - 1) / kIcpThreads + (size_t)(nt + kIcpThreads - 1) / kIcpThreads;
  *bytes = blocks * kIcpSums * sizeof(double);
  return ASDF_OK;
}

int asdf_icp_ts(const double* src_dev, int32_t ns, const double* tgt_dev, int32_t nt, int32_t max_iter, double stop_error,
                double stop_improvement, void* workspace_dev, size_t workspace_bytes, double* result, void* stream) {
  if (!src_dev || !tgt_dev || !workspace_dev || !result || ns < 1 || nt < 1 || max_iter < 1) return ASDF_EINVAL;
  size_t need = 0;
  asdf_icp_workspace_bytes(ns, nt, &need);
  if (workspace_bytes < need) return ASDF_ENOSPC;
  hipStream_t st = (hipStream_t)stream;
  const int bs = (ns + kIcpThreads - 1) / kIcpThreads, bt = (nt + kIcpThreads - 1) / kIcpThreads;
  double* part = (double*)workspace_dev;
  std::vector<double> host((size_t)(bs + bt) * kIcpSums);
  double scale = 1.0, t[3] = {0, 0, 0};
  double previous = 1e8, error = 1e8;
  int it = 0;
  for (it = 0; it < max_iter; ++it) {
    hipLaunchKernelGGL(icp_sweep_kernel, dim3(bs), dim3(kIcpThreads), 0, st, src_dev, ns, tgt_dev, nt, 0, scale, t[0], t[1], t[2], part);
    hipLaunchKernelGGL(icp_sweep_kernel, dim3(bt), dim3(kIcpThreads), 0, st, tgt_dev, nt, src_dev, ns, 1, scale, t[0], t[1], t[2],
                       part + (size_t)bs * kIcpSums);
    ASDF_HIP(hipGetLastError());
    ASDF_HIP(hipMemcpyAsync(host.data(), part, host.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    ASDF_HIP(hipStreamSynchronize(st));
    double sum[kIcpSums] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = 0; b < bs + bt; ++b)
      for (int k = 0; k < kIcpSums; ++k) sum[k] += host[(size_t)b * kIcpSums + k];
    const double n = (double)ns + (double)nt;
    error = std::sqrt(sum[0] / n);
    // stopping rules of run_icp_f (:66-72)
    if (previous - error < stop_improvement) { ++it; break; }
    previous = error;
    if (error < stop_error) { ++it; break; }
    // argmin_{s,t} sum |s X + t - Y|^2 over the stacked system (:76-107)
    const double xm[3] = {sum[1] / n, sum[2] / n, sum[3] / n}, ym[3] = {sum[4] / n, sum[5] / n, sum[6] / n};
    const double num = sum[7] - n * (xm[0] * ym[0] + xm[1] * ym[1] + xm[2] * ym[2]);
    const double den = sum[8] - n * (xm[0] * xm[0] + xm[1] * xm[1] + xm[2] * xm[2]);
    scale = num / den;
    for (int k = 0; k < 3; ++k) t[k] = ym[k] - scale * xm[k];
  }
  result[0] = scale; result[1] = t[0]; result[2] = t[1]; result[3] = t[2];
  result[4] = (double)(it > max_iter ? max_iter : it); result[5] = error;
  return ASDF_OK;
}

}  // extern "C"
