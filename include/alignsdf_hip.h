/* alignsdf_hip.h - C ABI of libalignsdf_hip.so (MI355X / gfx950 only).
 *
 * The drop-in boundary for AlignSDF's reconstruction hot path.  The reference has no native
 * interface of its own (it is pure PyTorch + skimage); each entry point below names the reference
 * code it replaces (paths relative to the zerchen/AlignSDF tree).  Plain pointers and sizes only:
 * device pointers are raw HIP device addresses (e.g. torch.Tensor.data_ptr()), `stream` is a
 * hipStream_t passed as void* (NULL = default stream).  All functions return 0 on success or a
 * negative ASDF_E* code, never throw, and never fall back to a CPU implementation.
 */
#ifndef ALIGNSDF_HIP_H_
#define ALIGNSDF_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ASDF_OK 0
#define ASDF_EINVAL (-1)     /* bad argument / unsupported decoder shape */
#define ASDF_ENOMEM (-2)     /* device or host allocation failed */
#define ASDF_EHIP (-3)       /* a HIP runtime call failed (see asdf_last_hip_error) */
#define ASDF_ENODEV (-4)     /* no gfx950 device visible */
#define ASDF_ENOSPC (-5)     /* caller-provided buffer / workspace too small */
#define ASDF_ERANGE (-6)     /* iso level outside the volume's data range (skimage ValueError) */
#define ASDF_ENOSURF (-7)    /* no surface found (skimage RuntimeError) */

#define ASDF_MAX_HEADS 2
#define ASDF_MAX_POINT_FEATS 64

/* How a head's point features are obtained from a normalised query point. */
#define ASDF_FEATURES_AFFINE 0 /* feat = E xyz + t (plain xyz, or utils.utils.kinematic_embedding folded in) */
#define ASDF_FEATURES_NERF 1   /* [x, sin(2^k x), cos(2^k x)], k < (pf-3)/6: get_nerf_embedder, utils/utils.py:433-463,
                                  521-533; point_feats must be 9 or 15 and equal for all heads */

/* Grid index modes of asdf_decode_grid. */
#define ASDF_GRID_REFERENCE 0 /* true-division indices exactly as utils/mesh.py:32-34 computes them */
#define ASDF_GRID_INTEGER 1   /* floor-division indices (an axis-aligned lattice) */

typedef struct asdf_decoder asdf_decoder_t;

/* Shape of the SDF MLP heads: networks/model.py:191-282 (SeparateDecoder.__init__) with
 * dims = [512,512,512,512], latent_in = [2], weight_norm on layers 0-3 (already folded by the
 * caller: W = g * v / ||v||, networks/model.py:249-250).
 * Head h consumes [latent (latent_size) | point features (point_feats[h])]. */
typedef struct asdf_decoder_spec {
  int32_t latent_size;                    /* 256 */
  int32_t hidden;                         /* 512 */
  int32_t num_heads;                      /* independent MLPs: 2 = SeparateDecoder (hand, object),
                                             1 = CombinedDecoder (networks/model.py:79-188) */
  int32_t point_feats[ASDF_MAX_HEADS];    /* 3 ("nerf", PointFeatSize 3) or 6 ("both", PointFeatSize 9) ... */
  int32_t outputs[ASDF_MAX_HEADS];        /* rows of the last layer: 1 (SeparateDecoder) or 2 (CombinedDecoder:
                                             row 0 = hand, row 1 = object, networks/model.py:99,185-188) */
  int32_t feature_mode;                   /* ASDF_FEATURES_AFFINE or ASDF_FEATURES_NERF */
} asdf_decoder_spec_t;

/* Host-side effective parameters of one head, row-major [out][in] like nn.Linear.weight:
 *   w[0] [hidden][latent+pf]   b[0] [hidden]        lin{h,o}0   (networks/model.py:249)
 *   w[1] [hidden-latent-pf][hidden]  b[1]           lin{h,o}1   (out = dims[1] - dims[0], :244-245)
 *   w[2] [hidden][hidden]      b[2] [hidden]        lin{h,o}2   (input = cat(x1, head_input), :311)
 *   w[3] [hidden][hidden]      b[3] [hidden]        lin{h,o}3
 *   w[4] [outputs][hidden]     b[4] [outputs]       lin{h,o}4   (plain Linear, then tanh :324-325) */
typedef struct asdf_head_params {
  const float* w[5];
  const float* b[5];
} asdf_head_params_t;

int asdf_version(void);
const char* asdf_strerror(int code);
/* hipError_t of the most recent failing HIP call on this thread (0 if none). */
int asdf_last_hip_error(void);
/* Number of visible gfx950 devices (0 if none); does not initialise a context. */
int asdf_device_count(void);

/* Build a decoder on the current HIP device: packs the weights into the MFMA streaming layout and
 * uploads them.  Replaces module construction + per-forward weight-norm recomputation
 * (networks/model.py:249-250 forward pre-hook). */
int asdf_decoder_create(const asdf_decoder_spec_t* spec, const asdf_head_params_t* heads /*[num_heads]*/,
                        asdf_decoder_t** out);
void asdf_decoder_destroy(asdf_decoder_t* dec);

/* Bind the per-sample inputs: the latent code (device, [latent_size]) and, per head, the affine map
 * from a normalised query point to that head's point features,
 *     feat[f] = embed[h][f][0..2] . xyz + embed[h][f][3]
 * (host, [num_heads][ASDF_MAX_POINT_FEATS][4]; NULL = identity, i.e. PointFeatSize 3).
 * Replaces latent.expand + cat (utils/utils.py:568-569) and utils.utils.kinematic_embedding
 * (utils/utils.py:376-430), which is affine in xyz. */
int asdf_decoder_set_sample(asdf_decoder_t* dec, const float* latent_dev, const float* embed_host, void* stream);
/* The same for a caller whose codes are on the HOST (the reference's codes come out of the encoder once per sample,
 * reconstruct.py:83-84; a run from saved codes has them in host memory): latent_pinned [latent_size] and embed_pinned (or NULL) are
 * PINNED, device-addressable host memory (hipHostMalloc / a pinned torch tensor) that stays untouched until the call's kernels have
 * run; one workgroup reads them over the link, in stream order, in front of the fold.  No copy engine and no runtime blit kernel
 * takes part (such a blit cannot get a wave slot while a persistent sweep owns every compute unit: 21.5 ms per sample resident in
 * the round-5 eval-mode trace). */
int asdf_decoder_set_sample_host(asdf_decoder_t* dec, const float* latent_pinned, const float* embed_pinned, void* stream);

/* Evaluate both heads on the N^3 lattice
 *     coord[a] = idx[a] * voxel_size + origin[a]     (fp32 mul then add, a = 0,1,2; axis 2 fastest)
 * writing sdf_hand[N^3], sdf_obj[N^3] (device; either may be NULL - a SeparateDecoder head whose output is NULL is
 * not evaluated, and its bbox record stays empty) and, if bbox_dev != NULL, the
 * per-head bounding box of negative voxels as int32[16]:
 *   [h*8 + 0..2] = min index per axis, [h*8 + 3..5] = max index per axis, [h*8 + 6] = #negative voxels,
 *   [h*8 + 7] = #points whose hidden activations left the fp16 range of the split-half planes (|x| >= 8188) - always
 *   0 under ASDF_MATH_F32; if non-zero under ASDF_MATH_F16X3 the caller should switch to ASDF_MATH_F32 and repeat.
 *   Bit 30 of word [7] is set when more voxels lay within the refinement threshold of the level than the refinement list
 *   holds (asdf_decoder_set_refine): the signs next to the level are then the split-half arithmetic's, not the fp32 chain's.
 *   With bbox_dev == NULL the same count is available from asdf_decoder_status.
 * Replaces one pass of utils/mesh.py:27-63 (or :82-115) plus the nonzero/min/max of
 * get_higher_res_cube (utils/mesh.py:208-237); deep_sdf/mesh.py:24-54 for the legacy entry point. */
int asdf_decode_grid(asdf_decoder_t* dec, int32_t N, const float origin[3], float voxel_size, int32_t grid_mode,
                     float* sdf_hand_dev, float* sdf_obj_dev, int32_t* bbox_dev, void* stream);

/* The bounding boxes of asdf_decode_grid WITHOUT its volumes: the coarse pass of the reference's two-pass flow
 * (utils/mesh.py:27-63) is consumed only through get_higher_res_cube (utils/mesh.py:198-256), i.e. through the per-head
 * box of its negative voxels, so it need not be evaluated to fp32 accuracy everywhere.  This entry point
 *   1. sweeps the lattice with ONE fp16 plane per operand (one MFMA per product sum: about a third of the time of the
 *      split-half sweep; values good to a few 1e-4) and takes the box of the voxels that are negative by more than tau;
 *   2. lists the voxels a head leaves undecided (-tau <= value < tau) that lie outside that head's box - only those can
 *      move it - and re-evaluates them on the fp32 MFMA chain (the arithmetic of ASDF_MATH_F32); every one that is
 *      negative extends the box.
 * bbox_dev (int32[48], required): words 0..15 then hold exactly the min / max words asdf_decode_grid would have produced,
 * PROVIDED the one-plane values are within tau of the exact ones; word [6] / [14] is non-zero iff the head has a negative
 * voxel (it is not the count), [7] / [15] the fp16 range report as usual.  The proviso is checked on every call: on the
 * re-evaluated candidates, and on an AUDIT sample - asdf_decoder_set_audit voxels (default 65536) drawn at random from those
 * both heads decided by sign alone, re-evaluated with the split-half arithmetic of the ordinary sweep (at most half the lattice).  The outcome travels in
 * words 16..31, a copy of the decoder's status record taken behind the call ([16 + 3] = largest |exact - one-plane| over the
 * candidates (float bits), [16 + 2] != 0 = a voxel taken as certainly negative was not), and in words 32..39:
 *   [32] number of candidates (more than 2^21: not all were re-evaluated),
 *   [33] / [34] asdf_decode_grid_band only: voxels marked for the hand / object head (more than 2^22: not all re-evaluated),
 *   [35] largest |exact - one-plane| over the audit sample (float bits), [36] audit voxels whose SIGN the exact value
 *   contradicts, [37] audit evaluations (voxels x heads), [38] near-level voxels beyond the refinement list (band sweep),
 *   [39] audit voxels drawn from the AT-RISK SHELL (decided by sign alone with tau <= |one-plane| < 2 tau: half of the audit goes
 *   there - all of the shell while it is smaller than that half), [40] the shell's population, [41] sum of squared audit errors
 *   (float bits; sigma = sqrt([41] / [37])); [28..31] shader-clock stamps of the sweep kernel (asdf_decoder_status [12..15]).
 * A caller must treat [7] / [15] / [18] / [36] != 0, [32] > 2^21, or [19] or [35] > tau / 2 as "repeat with asdf_decode_grid"
 * (alignsdf_amd/hip_decoder.py: coarse_begin / coarse_finish, which also re-estimate tau from the audit of every sample).
 * scratch_*_dev: N^3 floats per evaluated head (same NULL rules as asdf_decode_grid); contents afterwards: one-plane
 * values, exact ones at the re-evaluated voxels.  ASDF_EINVAL before a sample is bound or when the one-plane image does not exist (asdf_decoder_one_plane_usable). */
int asdf_decode_grid_box(asdf_decoder_t* dec, int32_t N, const float origin[3], float voxel_size, int32_t grid_mode, float tau,
                         float* scratch_hand_dev, float* scratch_obj_dev, int32_t* bbox_dev, void* stream);

/* asdf_decode_grid for a volume that is consumed by marching cubes ONLY (the fine pass of utils/mesh.py:82-121 followed by
 * utils/mesh.py:354): marching cubes reads the eight corner values of a cell only if the cell is active, and otherwise only
 * their signs.  This entry point
 *   1. sweeps the lattice with ONE fp16 plane per operand (as asdf_decode_grid_box);
 *   2. per head, marks the corners of every cell that can be active given |one-plane - exact| < tau - the cells whose corners
 *      are not all >= tau or all < -tau - (a few percent of the lattice next to the surface) and
 *   3. re-evaluates exactly those voxels of that head the way asdf_decode_grid evaluates every voxel - the decoder's split-half
 *      arithmetic over the list, then the fp32 MFMA chain where the result lies within the refinement threshold of the level -
 *      in place.
 * Afterwards every voxel a marching-cubes pass at level 0 reads the VALUE of holds exactly what asdf_decode_grid delivers
 * there, and every other voxel has the right SIGN: the meshes are identical, vertex for vertex and face for face;
 * the volume away from the surface holds fp16-class values and must not be used for anything else.
 * rec_dev (int32[48], required; layout as asdf_decode_grid_box): [7] / [15] fp16 range report, [16 + 3] largest
 * |re-evaluated - one-plane| over the marked voxels (float bits), [33] / [34] voxels marked for the hand / object head (more
 * than 2^22: not all were re-evaluated), [35] .. [37] the audit: per head, asdf_decoder_set_audit UNMARKED voxels - voxels whose
 * sign is all marching cubes will read - drawn at random and re-evaluated with the marked ones; [38] near-level voxels beyond
 * the refinement list.  A caller must repeat with asdf_decode_grid when [7] / [15] / [36] / [38] != 0, [33] or [34] > 2^22, or
 * [19] or [35] > tau / 2 (alignsdf_amd/hip_decoder.py: fine_begin / fine_needs_repeat).  Affine point features.  A CombinedDecoder
 * (networks/model.py:149-188: both columns from one MLP) marks the cells that can be active in EITHER volume into ONE list ([33];
 * [34] stays 0) and re-evaluates both columns of every listed voxel. */
int asdf_decode_grid_band(asdf_decoder_t* dec, int32_t N, const float origin[3], float voxel_size, int32_t grid_mode, float tau,
                          float* sdf_hand_dev, float* sdf_obj_dev, int32_t* rec_dev, void* stream);

/* The zoom cube of get_higher_res_cube (utils/mesh.py:239-254) ON THE DEVICE, from the boxes a coarse pass left in bbox_dev (the
 * int32[16] / [48] record of asdf_decode_grid / asdf_decode_grid_box): per enabled branch the box of its negative voxels, zeros for a
 * branch without one (utils/mesh.py:209-211); min / max over the enabled branches; then, in fp32 with separately rounded operations
 * like the reference's CPU tensors,
 *     new_cube_size = (max(max_index - min_index) + 4) * voxel_size;   new_voxel_size = new_cube_size / (N - 1);
 *     new_origin    = (min_index - 2) * voxel_size - 1
 * written to lattice_dev[4] = {origin0, origin1, origin2, new_voxel_size} (bit-equal to the host arithmetic the reference runs;
 * tests/test_gpu_zoom_handoff.py).  With it the fine pass can be ENQUEUED right behind the coarse pass - asdf_decode_grid_band_dev /
 * asdf_decode_grid_dev read their lattice from these words - and the host judges the coarse pass' record afterwards instead of
 * standing between the two passes (round 5; the reference synchronises twice per chunk, utils/mesh.py:46-63). */
int asdf_zoom_cube(const int32_t* bbox_dev, int32_t N, float voxel_size, int32_t hand_branch, int32_t obj_branch, float* lattice_dev,
                   void* stream);

/* asdf_decode_grid_band / asdf_decode_grid with the lattice {origin0, origin1, origin2, voxel_size} taken from DEVICE memory at
 * kernel run time (asdf_zoom_cube's output) instead of from the host arguments.  Everything else as the host-argument forms. */
int asdf_decode_grid_band_dev(asdf_decoder_t* dec, int32_t N, const float* lattice_dev, int32_t grid_mode, float tau,
                              float* sdf_hand_dev, float* sdf_obj_dev, int32_t* rec_dev, void* stream);
int asdf_decode_grid_dev(asdf_decoder_t* dec, int32_t N, const float* lattice_dev, int32_t grid_mode,
                         float* sdf_hand_dev, float* sdf_obj_dev, int32_t* bbox_dev, void* stream);

/* 1 when asdf_decode_grid_box / asdf_decode_grid_band are available for this decoder under its current activation scales.  The
 * one-plane kernels keep their own weight image, scaled so that every accumulator already carries its activation's plane scale
 * (no rescale in the epilogue); a decoder whose consecutive layers differ by more than 2^15 in activation magnitude cannot be
 * carried that way (0: the caller runs ordinary sweeps).  Changes with asdf_decoder_set_act_scales. */
int asdf_decoder_one_plane_usable(const asdf_decoder_t* dec);

/* The audit sample of the one-plane sweeps above: min(`voxels`, lattice / 64) per sweep and head (0 switches it off, at most
 * 262144; default 65536) - half drawn uniformly from the voxels decided by sign alone, half from the at-risk shell among them -
 * with a splitmix64 stream that starts at `seed` and advances with every sweep - so a run is reproducible and
 * no two sweeps look at the same voxels. */
int asdf_decoder_set_audit(asdf_decoder_t* dec, int32_t voxels, uint64_t seed);

/* Bounding box of the voxels with value < 0 of one [n0][n1][n2] fp32 device volume, as int32[16] on the
 * device (record 0 only: [0..2] min index per axis, [3..5] max index per axis, [6] count; min = INT_MAX and
 * max = -1 when the count is 0).  Replaces torch.nonzero + min/max in get_higher_res_cube
 * (utils/mesh.py:208-237) for callers that hold volumes rather than a decoder. */
int asdf_neg_bbox(const float* vol_dev, int32_t n0, int32_t n1, int32_t n2, int32_t* bbox_dev, void* stream);

/* Evaluate both heads on an explicit list of M normalised points xyz_dev[M][3] (device).
 * Replaces utils.utils.decode_sdf_multi_output (utils/utils.py:561-572) / deep_sdf.utils.decode_sdf
 * (deep_sdf/utils.py:64-75) for one chunk. */
int asdf_decode_points(asdf_decoder_t* dec, const float* xyz_dev, int64_t M, float* sdf_hand_dev,
                       float* sdf_obj_dev, void* stream);

/* ---- Arithmetic of the three hidden GEMMs of the grid sweeps (asdf_decode_grid; explicit point lists always run
 * ASDF_MATH_F32 - they have no bbox record to carry the range report below):
 *   ASDF_MATH_F32    v_mfma_f32_32x32x2_f32: a k-ordered fp32 FMA chain (the default after create);
 *   ASDF_MATH_F16X3  split-half: every operand carried as two fp16 planes (22 significand bits) of a power-of-two
 *                    scaled value, a product sum = three v_mfma_f32_32x32x16_f16 into one fp32 accumulator.  fp32-class
 *                    results (same error against fp64 as the fp32 chain on the test decoders, well inside the 1e-5 bar)
 *                    at 3/16 of the matrix-pipe time.  Operands must stay inside the fp16 range (hidden
 *                    activations |x| < 8188): violations are counted in word 7 of the bbox record of asdf_decode_grid,
 *                    and a caller that sees a non-zero count switches to ASDF_MATH_F32 and repeats the sweep (pass a
 *                    bbox buffer to every sweep whose range is not known to be safe).
 * May be switched at any time between launches. */
#define ASDF_MATH_F32 0
#define ASDF_MATH_F16X3 1
int asdf_decoder_set_math(asdf_decoder_t* dec, int32_t math);
int asdf_decoder_get_math(const asdf_decoder_t* dec);

/* Near-level refinement of ASDF_MATH_F16X3 grid sweeps (on by default, tau = 4e-6; 0 switches it off): after the sweep every
 * voxel with |sdf| < tau in either output is re-evaluated on the fp32 MFMA chain - the arithmetic of ASDF_MATH_F32 - and
 * written back, and the negative-voxel box is recounted on the refined volumes.  The two arithmetics agree to a few 1e-7
 * (well inside the 1e-5 bar), but the SIGN of a voxel that close to the level is all that the zoom cube
 * (utils/mesh.py:208-237) and marching cubes (utils/mesh.py:354) look at: with the refinement, boxes and surfaces are
 * those of the fp32 chain, voxel for voxel.  Costs one compaction pass over the volumes and one fp32 launch over a few
 * hundred points (about 0.25 ms per sweep, whatever N).  At most 65536 voxels are refined per sweep; asdf_decoder_status
 * word [1] counts any beyond that. */
int asdf_decoder_set_refine(asdf_decoder_t* dec, float tau);

/* Exact re-evaluation of SHORT voxel lists.  The fp32 MFMA chain over a voxel list - the near-level refinement above, the candidates
 * of asdf_decode_grid_box - takes 0.23 ms in its tile form however few voxels the list holds (one wave carries 32 points through
 * all 16 output tiles of every layer).  Lists of up to max_points voxels (default 8192: two rounds of 32-point workgroups per MLP; 0 = always the tile form; at most 65536)
 * run a second form of the same chain in which the four waves of a workgroup share the output tiles of ONE block of 32 points:
 * the same instruction sequence per output, bit-identical results, a quarter of the latency.  SeparateDecoder / CombinedDecoder
 * with affine point features. */
int asdf_decoder_set_short_list(asdf_decoder_t* dec, int32_t max_points);

/* ... and the SHORTEST lists (up to max_points voxels, default and at most 2048; 0 = never; effective limit min(max_points, the
 * short-list limit above, 2048 / MLPs evaluated - beyond 64 blocks x MLPs the form needs a second round of the chip): the same launch gives each block of 32 points to a cluster of four workgroups on one XCD - one output
 * tile per wave and layer, the layers' activations exchanged through device memory behind agent-scope release / acquire - so the
 * longest dependent chain is 642 MFMAs instead of 2064 (0.10 -> about 0.04 ms per launch; two launches per sample, which is a
 * sixth of a 64^3 sample).  Bit-identical to the other two forms (per-tile instruction sequence unchanged). */
int asdf_decoder_set_cluster_list(asdf_decoder_t* dec, int32_t max_points);
/* The cluster form's members wait for each other inside ONE ordinary launch; that they all become resident is an assumption (in-order
 * dispatch, free workgroup slots), so the wait is bounded: a member that waits longer than `ticks` of the 100 MHz s_memrealtime
 * counter (0 = the default, 1 s) raises word [11] of the decoder's status record - sticky until asdf_decoder_set_cluster_list
 * switches the form on again - the cluster writes nothing, and the tile form enqueued behind the launch evaluates the list instead
 * (bit-identical results, about 0.2 ms later).  Bit 29 of word 7 of a sweep's bbox record and word [16 + 11] of a one-plane sweep's
 * record carry the report to the host, which switches the form off (asdf_decoder_set_cluster_list(dec, 0)).  Until round 5 the wave
 * trapped instead, which costs the whole HIP context.  `ticks` = 1 is the test hook that provokes the failure deterministically:
 * every member gives up at its first wait without looking at the counter. */
int asdf_decoder_set_cluster_timeout(asdf_decoder_t* dec, uint64_t ticks);

/* Round 6: the matrix instruction of the split-half kernels of a SeparateDecoder with affine point features (every sweep and every
 * voxel list of the default arithmetic).  16 (the default) = v_mfma_f32_16x16x32_f16, 32 = v_mfma_f32_32x32x16_f16 - the same GEMMs,
 * both within 1e-5 of the reference (their bits differ: the order of the partial sums does); under the part's power management the
 * 16-wide form is ~5 % faster per sweep.  Process-wide (a launch-time choice, not decoder state); 0 = back to the environment's choice
 * (ASDF_K1H_SHAPE=32 selects 32).  Returns the shape in force BEFORE the call, or ASDF_EINVAL.  CombinedDecoder and NeRF-encoded
 * decoders always run the 32-wide form.  Kept for A/B measurements and as the fallback of a maintainer who distrusts the new kernels. */
int asdf_set_mfma_shape(int shape);
int asdf_get_mfma_shape(void);

/* Measurement hook: the next asdf_decode_grid / asdf_decode_points call of this decoder records the two hipEvent_t (passed
 * as void*, created by the caller with timing enabled) immediately before and after the launch of its dominant kernel
 * (sdf_mlp_f16_kernel or sdf_mlp_kernel) on the call's stream - not around the small kernels next to it (bbox
 * initialisation, near-level refinement).  One shot: the pair is forgotten after that call.  bench.py times the kernel the
 * roofline record is about this way, inside the timed region. */
int asdf_decoder_time_next_sweep(asdf_decoder_t* dec, void* event_start, void* event_stop);

/* Range report of the split-half arithmetic that does NOT depend on a bbox buffer: every ASDF_MATH_F16X3 launch of this
 * decoder adds the number of (point, lane-half) pairs whose hidden activations left the fp16 range (or whose output is not
 * in [-1, 1]) to a device word the decoder owns.  Copies the record to out_host[16] ([0] = that count, [1] = near-level
 * voxels (asdf_decode_grid_box: candidates) beyond the re-evaluation list's capacity, [2] = scratch flag of the last
 * re-evaluation (a voxel left the negative set), [3] = largest |new - old| value of the last re-evaluation (float bits), [4..6] / [8..10] = the largest fp16-plane value x S_x handed to the
 * conversion for the activation vectors h0 / h1 / h2 of MLP 0 / MLP 1, as float bit patterns, [12..13] / [14..15] = shader-clock stamps (s_memtime, 64 bit) of workgroup 0 at the first / last instruction of the
 * last whole-lattice split-half or one-plane sweep, [11] = non-zero once the short-list kernel's cluster form has reported a member that
 * never arrived (sticky: `clear` leaves it, see asdf_decoder_set_cluster_timeout), the rest reserved),
 * optionally clears it, and synchronises `stream`.  A caller that sweeps without a bbox buffer (deep_sdf/mesh.py:14-61
 * has no zoom pass) checks this once per volume and repeats the sweep under ASDF_MATH_F32 when the count is non-zero. */
int asdf_decoder_status(asdf_decoder_t* dec, int32_t out_host[16], int32_t clear, void* stream);

/* Activation scales S_x of the split-half image, per MLP and activation vector (h0, h1, h2): every hidden activation x is
 * carried as the two fp16 planes of x S_x.  The default 8 suits activations of order 1e-2 .. 1e3; a caller calibrates them
 * from the peak plane values asdf_decoder_status reports after a sweep over the whole cube (pass 1): S_x such that the peak
 * lands in [1024, 2048) keeps a factor 32 below the fp16 maximum while every activation down to 2^-13 of the layer's peak
 * keeps two full planes (22 significand bits).  Powers of two in [2^-24, 2^24]; the static constants of the image are
 * rebuilt (stream is synchronised first) and the per-sample constants are invalidated: call asdf_decoder_set_sample again. */
int asdf_decoder_set_act_scales(asdf_decoder_t* dec, const float sx[ASDF_MAX_HEADS][3], void* stream);
int asdf_decoder_get_act_scales(const asdf_decoder_t* dec, float sx_out[ASDF_MAX_HEADS][3]);

/* ---- Part classifier (specs["ClassifierBranch"]): classifier_head = nn.Linear(512, num_class) applied to the last
 * hidden activation of the hand MLP (SeparateDecoder, networks/model.py:257-259,306-307) or of the single MLP
 * (CombinedDecoder, networks/model.py:134-137,161-162).  w_host [num_class][512] / b_host [num_class] are host
 * fp32; num_class <= ASDF_MAX_CLASSES.  May be called again to replace the weights. */
#define ASDF_MAX_CLASSES 8
int asdf_decoder_set_classifier(asdf_decoder_t* dec, const float* w_host, const float* b_host, int32_t num_class);

/* asdf_decode_points plus the classifier: logits_dev [M][num_class] fp32 = `predicted_class` of the reference's
 * decoder.forward, labels_dev [M] int32 = predicted_class.argmax(dim=1) (first maximum).  Either may be NULL, not
 * both; the SDF outputs may be NULL.  Replaces one chunk of the label pass over the mesh vertices
 * (utils/mesh.py:146-157).  ASDF_EINVAL when no classifier was set. */
int asdf_decode_points_cls(asdf_decoder_t* dec, const float* xyz_dev, int64_t M, float* sdf_hand_dev,
                           float* sdf_obj_dev, float* logits_dev, int32_t* labels_dev, void* stream);

/* ---- Lewiner marching cubes (replaces skimage.measure.marching_cubes_lewiner as called at
 * utils/mesh.py:354 and deep_sdf/mesh.py:81: level given, step_size 1, allow_degenerate True,
 * use_classic False, gradient_direction 'descent', no mask).  Volume is [n0][n1][n2] fp32 on the
 * device.  Two-phase: count (classify + scan, returns V and F), then emit into caller buffers.
 * Output matches skimage element for element before the `* spacing` step:
 *   verts[V][3] fp32 in (axis0, axis1, axis2) voxel units, faces[F][3] int32. */
int asdf_mc_workspace_bytes(int32_t n0, int32_t n1, int32_t n2, size_t* bytes);
int asdf_mc_count(const float* vol_dev, int32_t n0, int32_t n1, int32_t n2, double level, void* workspace_dev,
                  size_t workspace_bytes, uint32_t* num_verts, uint32_t* num_faces, void* stream);
int asdf_mc_emit(const float* vol_dev, int32_t n0, int32_t n1, int32_t n2, double level, void* workspace_dev,
                 size_t workspace_bytes, float* verts_dev, int32_t* faces_dev, void* stream);
/* asdf_mc_emit into buffers sized BEFORE the counts are known (verts_dev [cap_verts][3], faces_dev [cap_faces][3]): entries beyond a
 * capacity are not written.  With it the emit phase is enqueued right behind asdf_mc_count_enqueue, the host reads V / F afterwards
 * and repeats the emit (or calls asdf_mc_emit) only when a capacity was too small - no host round trip between count and emit. */
int asdf_mc_emit_bounded(const float* vol_dev, int32_t n0, int32_t n1, int32_t n2, double level, void* workspace_dev,
                         size_t workspace_bytes, float* verts_dev, uint32_t cap_verts, int32_t* faces_dev, uint32_t cap_faces, void* stream);
/* The count phase without any host synchronisation: classify + reduce are enqueued and the last kernel writes
 * result[0] = V, result[1] = F, result[2..3] = order-preserving keys of the volume's min / max into result_mapped -
 * device-accessible HOST memory (hipHostMalloc / a pinned torch tensor; NULL = header only).  The caller records an event,
 * keeps queuing (the count of the next volume, ...), waits on the event when it needs the sizes, maps them to skimage's
 * two failure modes with asdf_mc_result_status (ASDF_ERANGE / ASDF_ENOSURF / ASDF_OK), allocates and calls asdf_mc_emit. */
int asdf_mc_count_enqueue(const float* vol_dev, int32_t n0, int32_t n1, int32_t n2, double level, void* workspace_dev,
                          size_t workspace_bytes, uint32_t* result_mapped, void* stream);
int asdf_mc_result_status(const uint32_t result[4], double level);

/* ---- Largest-component filter of utils/mesh.py:371-381 (trimesh.graph.split(only_watertight=True) + largest area) on
 * a marching-cubes surface: verts_dev [V][3] fp32 lattice-unit vertices, faces_dev [F][3] int32.  Faces are adjacent when
 * they share an edge that belongs to exactly two faces; a component qualifies with >= 4 faces and no edge shared by a
 * number of faces other than two; with fewer than two qualifying components the mesh comes back unchanged, otherwise the
 * qualifying component of largest area - measured, like the reference, on origin + voxel_size * v - with its vertices
 * compacted in ascending original order and its faces in original order.  Outputs (device): out_verts_dev [V][3],
 * out_faces_dev [F][3] (capacity of the input), counts_dev int32[8] (round 6; it was [4]) = kept vertices, kept faces, qualifying
 * components, first face of the kept component, [4] components of >= 4 faces that did NOT qualify because they are open or
 * non-manifold - the only place where trimesh's `fill_holes` (submesh(repair=True); not reproduced, this row is PARITY UNPINNED)
 * could have changed the outcome: expected 0 on every closed marching-cubes surface, and a run reports the sum next to its meshes -,
 * [5] components of fewer than 4 faces, [6..7] zero.  No host synchronisation.  Workspace: asdf_mesh_cc_workspace_bytes(V, F). */
int asdf_mesh_cc_workspace_bytes(int32_t num_verts, int32_t num_faces, size_t* bytes);
int asdf_mesh_largest_component(const float* verts_dev, int32_t num_verts, const int32_t* faces_dev, int32_t num_faces,
                                float voxel_size, const float origin[3], void* workspace_dev, size_t workspace_bytes,
                                float* out_verts_dev, int32_t* out_faces_dev, int32_t* counts_dev, void* stream);

/* ---- K9: eval mode's surface sampling and sample normalisation on the device (csrc/surface_sample.hip).
 * asdf_sample_surface replaces `trimesh.sample.sample_surface(mesh, count)` (utils/mesh.py:386-389 through
 * deep_sdf/metrics/icp_trans_scale.py:19-23) with the seeded, area-quantised sampler of alignsdf_amd/surface_sampling.py, bit for bit:
 * verts_dev [V][3] fp32 - lattice units when place != 0 (placed on the fly: v * voxel_size + origin, an fp32 multiply and an fp32 add, the
 * exporter's arithmetic of utils/mesh.py:360-369), positions otherwise - faces_dev [faces_cap][3], of which the first *num_faces_dev (a
 * device word; NULL = all) are faces (K8's output), u_dev [count] / r_dev [count][2] the draws (surface_sampling._uniforms) and
 * points_dev [count][3] fp64 the samples.  Four launches, no host synchronisation.
 * asdf_icp_normalise is ICP_T_S.sample_mesh's normalisation (icp_trans_scale.py:25-31): src_out_dev = (src - mean_s) / rms_s * rms_t +
 * mean_t; stats_mapped (optional, device-accessible host memory, 8 doubles) receives mean_s[3], rms_s, mean_t[3], rms_t. */
int asdf_sample_surface_workspace_bytes(int32_t faces_cap, size_t* bytes);
int asdf_sample_surface(const float* verts_dev, const int32_t* faces_dev, int32_t faces_cap, const int32_t* num_faces_dev,
                        int32_t place, float voxel_size, const float origin[3], const double* u_dev, const double* r_dev, int32_t count,
                        double* points_dev, void* workspace_dev, size_t workspace_bytes, void* stream);
int asdf_icp_normalise(const double* src_dev, int32_t ns, const double* tgt_dev, int32_t nt, double* src_out_dev, double* stats_mapped,
                       void* stream);

/* ---- Translate + scale ICP of the reference's eval mode: ICP_T_S.run_icp_f (deep_sdf/metrics/icp_trans_scale.py:33-113),
 * called from utils/mesh.py:385-395.  src_dev [ns][3] are the ALREADY NORMALISED source samples (sample_mesh :25-31),
 * tgt_dev [nt][3] the target samples, both fp64 on the device.  Runs until the reference's stopping rules fire
 * (error < stop_error, or improvement < stop_improvement, or max_iter) and writes, on the host,
 *   result[0] = scale, result[1..3] = trans, result[4] = iterations executed, result[5] = last RMS error.
 * The iteration (both nearest-neighbour sweeps, the sums, the stopping rules and the solve) stays on the device;
 * asdf_icp_ts enqueues 8 iterations at a time and synchronises the stream between batches to read the 64-byte state. */
int asdf_icp_workspace_bytes(int32_t ns, int32_t nt, size_t* bytes);
/* Nearest-neighbour search of the ICP / Chamfer kernels (process-wide; for tests and measurements): 0 = automatic - a uniform grid
 * per reference set, built once per run, when both sets have at least 1024 points, brute force otherwise; 1 = brute force;
 * 2 = grid.  Both are exact and return identical neighbours (on exact distance ties: the lowest index). */
int asdf_icp_set_search(int32_t mode);
int asdf_icp_ts(const double* src_dev, int32_t ns, const double* tgt_dev, int32_t nt, int32_t max_iter, double stop_error,
                double stop_improvement, void* workspace_dev, size_t workspace_bytes, double* result, void* stream);

/* The same run without any host synchronisation: every one of the max_iter iterations is enqueued (iterations past
 * convergence return at once), the outcome stays in the workspace.  asdf_icp_ts_result copies it to result[6] (layout
 * above) and synchronises the stream; the workspace and both point sets must stay alive until then.
 * result_mapped (optional) is device-accessible HOST memory (hipHostMalloc / hipHostRegister, 7 doubles: [6] = converged) that the last
 * kernel of the run fills: a caller that has queued further work behind the ICP waits on an event recorded after this
 * call and reads it - any copy, even from a side stream, would be a kernel that cannot start while a decoder pass
 * holds the whole register file of every SIMD. */
int asdf_icp_ts_enqueue(const double* src_dev, int32_t ns, const double* tgt_dev, int32_t nt, int32_t max_iter,
                        double stop_error, double stop_improvement, void* workspace_dev, size_t workspace_bytes,
                        double* result_mapped, void* stream);
int asdf_icp_ts_result(const void* workspace_dev, double* result, void* stream);
/* asdf_icp_ts_enqueue for iterations [first_iter, last_iter) only: first_iter == 0 begins the run (initial state, search grids),
 * first_iter > 0 continues the run that lives in the workspace.  result_mapped takes 7 doubles here: the six of the layout above
 * and [6] = 1 if the reference's stopping rules have fired.  A run converges in a handful of iterations, and every enqueued
 * iteration past convergence is still two (empty) launches: a caller enqueues the first 16 behind its other work and, only when
 * [6] is still 0 at the time it needs the result, the remaining ones (alignsdf_amd/icp.py: start_icp_device / finish_icp). */
int asdf_icp_ts_enqueue_range(const double* src_dev, int32_t ns, const double* tgt_dev, int32_t nt, int32_t first_iter, int32_t last_iter,
                              double stop_error, double stop_improvement, void* workspace_dev, size_t workspace_bytes,
                              double* result_mapped, void* stream);

/* Symmetric Chamfer distance of the reference's evaluation (compute_trimesh_chamfer, deep_sdf/metrics/chamfer.py:217-229):
 * exact nearest neighbours both ways between a_dev [na][3] and b_dev [nb][3] (fp64, device), result[0] = mean squared
 * distance a -> b (`gen_to_gt_chamfer` when a is the generated mesh's samples), result[1] = mean squared distance b -> a;
 * the metric is their sum.  Workspace: asdf_icp_workspace_bytes(na, nb).  Synchronises the stream. */
int asdf_chamfer(const double* a_dev, int32_t na, const double* b_dev, int32_t nb, void* workspace_dev,
                 size_t workspace_bytes, double* result, void* stream);

/* ---- Test hook (device): coordinates of lattice points first .. first + count - 1 exactly as the decoder kernels
 * generate them in registers (the same device function), coords_dev [count][3] fp32 in (axis 0, axis 1, axis 2) order.
 * Lets the tests compare the in-kernel lattice with utils/mesh.py:27-40,82-96 bit for bit. */
int asdf_debug_grid_coords(int32_t N, const float origin[3], float voxel_size, int32_t grid_mode, int64_t first, int64_t count,
                           float* coords_dev, void* stream);

/* ---- Test hook (host only, needs no device): run the weight packer of asdf_decoder_create and copy
 * its images out (any pointer may be NULL).  Sizes in floats: stream 256*4096, wlat 2*2*512*256,
 * wpt 2*2*512*ASDF_MAX_POINT_FEATS, bias02 2*2*512, cst 2*(6920 + 2048*(KP-2)) with KP = 2 (affine) or ceil(pf/2), embed 2*ASDF_MAX_POINT_FEATS*4. */
int asdf_debug_pack_host(const asdf_decoder_spec_t* spec, const asdf_head_params_t* heads, float* stream,
                         float* wlat, float* wpt, float* bias02, float* cst, float* embed);
/* The split-half image of the same decoder (ASDF_MATH_F16X3): stream16 2*128*8192 fp16 bit
 * patterns (stage = [kblock 8][plane hi/lo][lane 64][8]), cst16 = the constants block with the scaled entries,
 * s2[2] = the layer-2 accumulator scale per head that K0 applies to the per-sample constants. */
int asdf_debug_pack_host_f16(const asdf_decoder_spec_t* spec, const asdf_head_params_t* heads, uint16_t* stream16,
                             float* cst16, float* s2);
/* The same weights in the record order of the W form (v_mfma_f32_16x16x32_f16; sdf_mlp_f16w_kernel.h): stream16w 2*128*8192 fp16 bit
 * patterns, 2 KiB records [plane hi/lo][lane 64][8] = (tile, feature half, K32-block) - lane l: output row 32 t + 16 fh + (l & 15),
 * slot (l >> 4, e): input feature 32 j + 16 (e >> 2) + 4 (l >> 4) + (e & 3); layers 1 / 3: record i = (fh i / 16, j i % 16), layer 2:
 * (fh i % 2, j i / 2). */
int asdf_debug_pack_host_f16w(const asdf_decoder_spec_t* spec, const asdf_head_params_t* heads, uint16_t* stream16w);

#ifdef __cplusplus
}
#endif
#endif /* ALIGNSDF_HIP_H_ */
