// Launch interface of the fused decoder kernels: each family is its own translation unit (k1_kernels.hip: fp32 MFMA,
// k1_cls_kernels.hip: fp32 MFMA + part classifier, k1h_kernels.hip: split-half fp16 MFMA) and compiles on its own.
#pragma once
#include <hip/hip_runtime.h>

#include "sdf_mlp_common.h"

namespace asdf {

// the cluster form of the short-list kernel (sdf_mlp_short_kernel.h)
constexpr int kClusterWgs = 4;                                                 // workgroups per 32-point block in the cluster form
constexpr int kClusterCap = 2048;                                              // longest list the cluster form takes (64 blocks per MLP)
constexpr int kXchgTiles = kTilesL1 + 2 * kTilesHidden;                        // h1 (8 tiles), h2 (16), raw layer-3 accumulators (16)
constexpr int kXchgFloats = kXchgTiles * 64 * 16;                              // 160 KiB per cluster
struct ShortParams {
  float* xchg;          // [clusters][kXchgFloats]
  int* arrivals;        // [clusters][4] (three in use): multiples of 4 between launches
  int cluster_max;      // lists of up to this many points take the cluster form (0 = never)
  int* fault;           // device word (the decoder's status[11]), sticky: a member waited longer than `timeout_ticks` for another one.
                        // A cluster that sees it writes NO output; the tile form enqueued behind the launch evaluates the list instead
                        // (DecodeParams::short_fault) and the host switches the cluster form off when it reads the word
  unsigned long long timeout_ticks;   // s_memrealtime ticks (100 MHz) a member waits for the others before it gives up
};
constexpr unsigned long long kClusterTimeoutTicks = 100000000ull;              // 1 s: a healthy wait is microseconds

// raise the dynamic-LDS limit of the family's kernels (once per process; cheap)
hipError_t k1_prepare();
hipError_t k1_cls_prepare();
hipError_t k1h_prepare();            // includes the NeRF-encoded family
hipError_t k1h_nerf_prepare();
hipError_t k1s_prepare();           // the one-plane kernels (k1s_kernels.hip)
hipError_t k1s_nerf_prepare();      // the one-plane kernels of the NeRF-encoded decoders (k1s_nerf_kernels.hip)

// kp = point-feature K-steps (2 affine xyz, 5 / 8 NeRF encoding of 9 / 15 features); two_out = CombinedDecoder
void k1_launch(int kp, bool two_out, const DecodeParams& p, int grid, hipStream_t st);
// the fp32 chain over a SHORT voxel list (kGridSubset, kp == 2): one workgroup per 32 points and MLP, output tiles spread over its
// waves - or, for the shortest lists, four workgroups per block (sdf_mlp_short_kernel.h); returns at once for lists longer than p.short_max
void k1_short_launch(bool two_out, const DecodeParams& p, const ShortParams& sp, hipStream_t st);
void k1_cls_launch(int kp, bool two_out, const DecodeParams& p, int grid, hipStream_t st);
void k1h_launch(int kp, bool two_out, const DecodeParams& p, int grid, hipStream_t st);
// the W form (16x16x32 MFMAs) of the SeparateDecoder kernels with affine point features: k1hw_kernels.hip
hipError_t k1hw_prepare();
void k1hw_launch(const DecodeParams& p, int grid, hipStream_t st);
void k1hw_subset_launch(const DecodeParams& p, int grid, hipStream_t st);
int k1h_shape();      // 16 = the W form is selected (default), 32 = ASDF_K1H_SHAPE=32 / asdf_set_mfma_shape(32)
int k1h_set_shape(int shape);
void k1h_box_launch(bool two_out, const DecodeParams& p, int grid, hipStream_t st);   // one-plane kernel (k1s_kernels.hip), kp == 2 only; p.stream = high planes
void k1h_subset_launch(int kp, bool two_out, const DecodeParams& p, int grid, hipStream_t st);   // split-half kernel over a voxel list
void k1h_nerf_subset_launch(int kp, bool two_out, const DecodeParams& p, int grid, hipStream_t st);      // ... kp 5 / 8 (k1h_nerf_kernels.hip)
void k1h_nerf_launch(int kp, bool two_out, const DecodeParams& p, int grid, hipStream_t st);      // kp 5 / 8 (k1h_nerf_kernels.hip)
void k1s_nerf_launch(int kp, bool two_out, const DecodeParams& p, int grid, hipStream_t st);      // one-plane kernel, kp 5 / 8; p.stream = high planes

}  // namespace asdf
