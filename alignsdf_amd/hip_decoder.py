"""Host-side handle of the HIP SDF decoder: turns a reference-shaped decoder module into the packed
MFMA weight stream (once) and exposes the grid / point-list sweeps.

The module is recognised by its parameters (`lin{h,o}{0..4}.weight_g|weight_v|weight|bias`, the
state-dict layout of networks/model.py:191-282) and attributes (`latent_size`, `point_feat_size`,
`encode_style`).  PyTorch is used for device memory and streams only; all arithmetic runs in
libalignsdf_hip.so.
"""
import collections
import ctypes
import os

import numpy as np
import torch

from . import _native


DEFAULT_MATH = "f16x3"
# Round 6: the DEFAULT of every entry point is the reference's arithmetic class on EVERY voxel of both passes - ordinary sweeps
# (the split-half kernel: 22-bit operands, fp32 accumulate, <= 1e-5; or the fp32 MFMA chain under ASDF_MATH=f32).  The audited
# one-plane sweeps (one fp16 plane for the SIGNS of a pass with one consumer, every value that consumer reads
# re-evaluated exactly, a statistical certificate per sweep) are an OPT-IN: ASDF_FAST=1, `--fast` on the CLIs and bench.py,
# HipSdfDecoder.set_fast(True), or ASDF_COARSE=box / ASDF_FINE=band one by one.  Every volume-returning call runs ordinary sweeps
# whatever these say.
DEFAULT_COARSE = "exact"      # coarse pass of the two-pass flow: "exact" | "box" (HipSdfDecoder.coarse_begin)
DEFAULT_FINE = "exact"        # fine pass (feeds marching cubes only): "exact" | "band" (HipSdfDecoder.fine_begin)
FAST_COARSE = "box"           # what --fast / ASDF_FAST=1 select
FAST_FINE = "band"


def fast_requested():
    """ASDF_FAST set to anything but '' / '0': the audited one-plane sweeps wherever a pass has one consumer."""
    return os.environ.get("ASDF_FAST", "0") not in ("", "0")


BAND_CAP = 1 << 22           # voxels per head the narrow-band sweep can re-evaluate (csrc/decoder.hip: kBandCap)
NEAR_CAP = 1 << 16           # near-level refinement list of a split-half sweep
CAND_CAP = 1 << 21           # box candidates of one coarse sweep
AUDIT_VOXELS = 1 << 16        # per one-plane sweep and head (asdf_decoder_set_audit)
NEAR_OVERFLOW_BIT = 0x40000000
CLUSTER_FAULT_BIT = 0x20000000   # bit 29 of word 7 of a sweep's bbox record: the short-list kernel's cluster form reported a member that never arrived
CODE_SLOTS = 16               # pinned staging slots for the per-sample codes of set_sample's host path
REC_WORDS = 48                # record of a one-plane sweep (include/alignsdf_hip.h: asdf_decode_grid_box)
# allowance tau = TAU_FACTOR x the estimated lattice maximum of |one-plane - exact| of recent sweeps; a sweep is accepted while ITS
# estimate (and the error on its re-evaluated voxels) stays within TAU_ACCEPT x the tau it was launched with - a factor
# TAU_FACTOR x TAU_ACCEPT = 3 of sample-to-sample head room (pose-aligned decoders vary by 2 x), 1 / TAU_ACCEPT = 1.67 of safety
# between the tail-corrected estimate and the bound that must hold for every voxel
TAU_FACTOR = 5.0
TAU_ACCEPT = 0.6
# The whole-lattice comparison that calibrates the allowance (one ordinary + one plain one-plane sweep of the coarse lattice, all
# 2 N^3 values compared) is REPEATED every RECAL_EVERY coarse passes of a decoder - the tail ratio (lattice maximum / maximum of a
# sample of the audit's size), i.e. how far the largest error can hide from a sample, is a measured quantity that is never older
# than that - and on the coarse pass that follows any sweep refused for its error.  A tail ratio above TAIL_MAX means the error has
# rare large outliers a sample cannot see: the one-plane sweeps are switched off for that decoder.
# Round 5: the comparison ALTERNATES between the coarse lattice ([-1, 1]^3) and the ZOOM lattice of the fine pass - the lattice whose
# signs marching cubes actually consumes: another voxel pitch, other coordinates, values ten times closer to the level
# (utils/mesh.py:82-121).  A decoder's first coarse AND first fine pass are compared; afterwards one of the two every RECAL_EVERY
# samples, in turn.  A band sweep needs a valid fine-lattice comparison, a box sweep a valid coarse-lattice one.
RECAL_EVERY = 64
TAIL_MAX = 3.0


def _effective(module_sd, name):
    """Effective [out, in] weight of layer `name`: g * v / ||v|| for weight-normed layers, computed
    with the same torch primitive the module's forward hook uses (networks/model.py:249-250)."""
    if name + ".weight_v" in module_sd:
        v = module_sd[name + ".weight_v"].detach().float().cpu()
        g = module_sd[name + ".weight_g"].detach().float().cpu()
        return torch._weight_norm(v, g, 0).contiguous()
    return module_sd[name + ".weight"].detach().float().cpu().contiguous()


def head_point_feats(point_feat_size, encode_style):
    """(hand, obj) point-feature counts per encode style (networks/model.py:212-223, :288-299)."""
    return {
        "nerf": (point_feat_size, point_feat_size),
        "hand": (point_feat_size, 3),
        "obj": (3, point_feat_size),
        "both": (point_feat_size - 3, 6),
    }[encode_style]


def kinematic_affine(point_feat_size, encode_style, scale_factor, mano_results, obj_results, combined=False):
    """Affine form of utils.utils.kinematic_embedding (utils/utils.py:376-430): returns, per head,
    E [pf_head, 4] with  feat = E[:, :3] @ xyz + E[:, 3].  Computed in float64 from the same inputs.

    wrist = xyz * 2 / sf;  mano = wrist + rot_center;  q_j = (G_j^-1 [mano; 1])[:3] / w;
    o = (T_obj^-1 [wrist; 1])[:3] / w;  all scaled back by sf / 2.  (The homogeneous divisor w is the
    constant 1 for rigid transforms; it is folded in as computed.)"""
    sf = float(scale_factor)
    a = 2.0 / sf
    rows_hand, rows_obj_tail = [], []
    if encode_style in ("hand", "both"):
        rc = mano_results["rot_center"].detach().double().cpu().reshape(3).numpy()
        # mano_xyz * sf/2 = xyz + rc * sf/2
        m = np.concatenate([np.eye(3), (rc * sf / 2.0)[:, None]], 1)
        rows_hand.append(m)
        G = mano_results["global_trans"].detach().double().cpu().reshape(16, 4, 4).numpy()
        joints = 1 if ((point_feat_size == 6 and encode_style == "hand") or
                       (point_feat_size == 9 and encode_style == "both")) else 16
        for j in range(joints):
            Gi = np.linalg.inv(G[j])
            # [mano;1] = [a*xyz + rc; 1];  q = Gi[:3,:3] (a xyz + rc) + Gi[:3,3], divided by w
            w = Gi[3, :3] @ rc + Gi[3, 3]          # xyz-dependent part of w is 0 for rigid transforms
            lin = Gi[:3, :3] * a / w
            off = (Gi[:3, :3] @ rc + Gi[:3, 3]) / w
            rows_hand.append(np.concatenate([lin * sf / 2.0, (off * sf / 2.0)[:, None]], 1))
    if encode_style in ("obj", "both"):
        T = obj_results["obj_trans"].detach().double().cpu().reshape(4, 4).numpy()
        Ti = np.linalg.inv(T)
        w = Ti[3, 3]
        lin = Ti[:3, :3] * a / w
        off = Ti[:3, 3] / w
        rows_obj_tail.append(np.concatenate([lin * sf / 2.0, (off * sf / 2.0)[:, None]], 1))
    ident = np.concatenate([np.eye(3), np.zeros((3, 1))], 1)
    if combined:      # CombinedDecoder consumes the whole embedding (networks/model.py:150-157)
        if encode_style == "hand":
            return (np.concatenate(rows_hand, 0),)
        if encode_style == "obj":
            return (np.concatenate([ident] + rows_obj_tail, 0),)
        return (np.concatenate(rows_hand + rows_obj_tail, 0),)
    if encode_style == "hand":
        return np.concatenate(rows_hand, 0), rows_hand[0]            # obj head sees input[:, :L+3] = mano_xyz*sf/2
    if encode_style == "obj":
        return ident, np.concatenate([ident] + rows_obj_tail, 0)     # hand head sees input[:, :L+3] = xyz
    hand = np.concatenate(rows_hand, 0)
    return hand, np.concatenate([rows_hand[0]] + rows_obj_tail, 0)   # networks/model.py:297-299


def unsupported_reason(state_dict, latent_size, point_feat_size, encode_style):
    """Why (a string) the fused kernels cannot evaluate a decoder with these parameters, or None when they can.  The kernels are
    built for the shape every shipped AlignSDF config uses (experiments/*/*.json: LatentSize 256, dims [512] * 4, latent_in [2] -
    networks/model.py:192-282): five layers per MLP, 512-wide, the skip concatenation in front of layer 2; any other legal
    NetworkSpecs runs on the module path (alignsdf_amd.torch_decoder).  `state_dict` maps parameter names (module.decoder. prefix
    already stripped) to tensors / arrays; only shapes are looked at."""
    shape = lambda k: tuple(state_dict[k].shape)
    keys = set(state_dict)
    if any(k.startswith("bn") for k in keys):
        return "LayerNorm form (weight_norm false)"
    combined = "lin0.bias" in keys
    prefixes = ("lin",) if combined else ("linh", "lino")
    if not combined and "linh0.bias" not in keys:
        return "not a SeparateDecoder / CombinedDecoder shaped module"
    if int(latent_size) != 256:
        return "LatentSize %d (the kernels fold a 256-wide latent)" % int(latent_size)
    if encode_style not in ("nerf", "hand", "obj", "both"):
        return "EncodeStyle %r" % (encode_style,)
    pf = (int(point_feat_size),) if combined else head_point_feats(int(point_feat_size), encode_style)
    if encode_style == "nerf" and int(point_feat_size) > 3 and int(point_feat_size) not in (9, 15):
        return "NeRF positional encoding with PointFeatSize %d (9 and 15 are built)" % int(point_feat_size)
    n_out = 2 if combined else 1
    for prefix, f in zip(prefixes, pf):
        if not 1 <= f <= _native.MAX_POINT_FEATS or 512 - 256 - f < 1:
            return "%d point features per head" % f
        if (prefix + "5.bias") in keys or (prefix + "4.bias") not in keys:
            return "not five layers per MLP (dims other than [512, 512, 512, 512])"
        n_in = 256 + f
        want = [(512, n_in), (512 - n_in, 512), (512, 512), (512, 512), (n_out, 512)]
        for layer in range(5):
            name = "%s%d" % (prefix, layer)
            wkey = name + (".weight_v" if name + ".weight_v" in keys else ".weight")
            if wkey not in keys or name + ".bias" not in keys:
                return "layer %s missing" % name
            if shape(wkey) != want[layer]:
                return "layer %s is %s, the kernels need %s (dims [512] * 4, latent_in [2])" % (name, shape(wkey), want[layer])
    if "classifier_head.weight" in keys:
        cw = shape("classifier_head.weight")
        if len(cw) != 2 or cw[1] != 512 or not 1 <= cw[0] <= _native.MAX_CLASSES:
            return "classifier_head of shape %s" % (cw,)
    return None


class HipSdfDecoder:
    """Device-resident packed decoder.  One instance per (module, device)."""

    def __init__(self, module_or_state_dict, latent_size=None, point_feat_size=None, encode_style=None, device=None):
        if isinstance(module_or_state_dict, torch.nn.Module):
            m = module_or_state_dict
            sd = m.state_dict()
            latent_size = latent_size if latent_size is not None else getattr(m, "latent_size", 256)
            point_feat_size = point_feat_size if point_feat_size is not None else m.point_feat_size
            encode_style = encode_style if encode_style is not None else m.encode_style
            if getattr(m, "use_tanh", False) or getattr(m, "xyz_in_all", False):
                raise NotImplementedError("use_tanh / xyz_in_all decoder variants are outside the HIP path")
        else:
            sd = {k: torch.as_tensor(v) for k, v in module_or_state_dict.items()}
        sd = {k[len("module.decoder."):] if k.startswith("module.decoder.") else k: v for k, v in sd.items()}
        why = unsupported_reason(sd, latent_size if latent_size is not None else 256, point_feat_size, encode_style)
        if why is not None:
            raise NotImplementedError("the fused kernels do not cover this decoder: %s - utils.utils.decoder_for routes it to the "
                                      "module path (alignsdf_amd.torch_decoder)" % why)
        self.combined = "lin0.bias" in sd and "lin4.bias" in sd
        self.latent_size = int(latent_size)
        self.point_feat_size = int(point_feat_size)
        self.encode_style = encode_style
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        L = _native.lib()
        if L.asdf_device_count() < 1:
            raise _native.NativeError(-4, "no gfx950 (MI355X) device visible - the HIP path has no CPU fallback")
        # NeRF positional encoding: PointFeatSize > 3 with EncodeStyle "nerf" (utils/mesh.py:49-55) - not affine in xyz
        self.nerf_features = self.point_feat_size > 3 and encode_style == "nerf"
        if self.nerf_features and self.point_feat_size not in (9, 15):
            raise NotImplementedError("NeRF positional encoding is supported for PointFeatSize 9 and 15")
        mode = _native.FEATURES_NERF if self.nerf_features else _native.FEATURES_AFFINE
        if self.combined:
            pf, prefixes, n_out = (self.point_feat_size,), ("lin",), 2
            spec = _native.DecoderSpec(self.latent_size, 512, 1, (ctypes.c_int32 * 2)(pf[0], 0), (ctypes.c_int32 * 2)(2, 0), mode)
        else:
            pf, prefixes, n_out = head_point_feats(self.point_feat_size, encode_style), ("linh", "lino"), 1
            spec = _native.DecoderSpec(self.latent_size, 512, 2, (ctypes.c_int32 * 2)(*pf), (ctypes.c_int32 * 2)(1, 1), mode)
        heads = (_native.HeadParams * 2)()
        keep = []
        for hi, prefix in enumerate(prefixes):
            n_in = self.latent_size + pf[hi]
            shapes = [(512, n_in), (512 - n_in, 512), (512, 512), (512, 512), (n_out, 512)]
            for layer in range(5):
                name = "%s%d" % (prefix, layer)
                w = _effective(sd, name)
                b = sd[name + ".bias"].detach().float().cpu().contiguous()
                if tuple(w.shape) != shapes[layer] or b.numel() != shapes[layer][0]:
                    raise NotImplementedError("unsupported layer shape %s %s (expected %s)" % (
                        name, tuple(w.shape), shapes[layer]))
                keep += [w, b]
                heads[hi].w[layer] = w.data_ptr()
                heads[hi].b[layer] = b.data_ptr()
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _native.check(L.asdf_decoder_create(ctypes.byref(spec), heads, ctypes.byref(handle)), "asdf_decoder_create")
        self._h = handle
        self._L = L
        # part classifier on the last hidden activation of the hand / single MLP (specs["ClassifierBranch"])
        self.num_class = 0
        if "classifier_head.weight" in sd:
            cw = sd["classifier_head.weight"].detach().float().cpu().contiguous()
            cb = sd["classifier_head.bias"].detach().float().cpu().contiguous()
            if cw.dim() != 2 or cw.shape[1] != 512 or cb.numel() != cw.shape[0] or not 1 <= cw.shape[0] <= _native.MAX_CLASSES:
                raise NotImplementedError("unsupported classifier_head shape %s" % (tuple(cw.shape),))
            with torch.cuda.device(self.device):
                _native.check(L.asdf_decoder_set_classifier(handle, cw.data_ptr(), cb.data_ptr(), int(cw.shape[0])),
                              "asdf_decoder_set_classifier")
            self.num_class = int(cw.shape[0])
        self._pf = pf
        self._init_sweep_state()

    def _init_sweep_state(self):
        """Everything the sweep state machine keeps on the host (arithmetic in use, calibration of the activation scales, modes,
        allowance, counters): no device involved beyond set_math, so tests/test_sweep_state_machine.py can drive it on the CPU."""
        # arithmetic of the hidden GEMMs: "f32" (fp32 MFMA chain) or "f16x3" (split-half fp16 MFMA, fp32-class results);
        # ASDF_MATH overrides the default
        self.math = "f32"
        want = os.environ.get("ASDF_MATH", DEFAULT_MATH)
        # (PointFeatSize 15 with two MLPs: until round 4 this shape kept the fp32 chain because its 8-K-step split-half instantiation
        # holds 400 B per lane in scratch.  Measured at N = 256 - profiles/r04_nerf_one_plane.txt - the fp32 chain takes 464 ms per
        # sample, that split-half kernel 194 ms and the audited one-plane sweeps in front of it 63 ms: the default is the default.)
        if want == "f16x3":
            self.set_math("f16x3")
        elif want not in ("f32", "f16x3"):
            raise ValueError("ASDF_MATH must be 'f32' or 'f16x3', not %r" % want)
        self._latent = None
        self._bound = None           # (latent, embed) of the sample the decoder is bound to
        self._cal_done = np.zeros((2, 3), dtype=bool)     # activation scale of (MLP, layer) calibrated from a recorded peak
        self._cal_attempts = 0
        self._recalibrations = 0     # epoch of the activation scales: tickets and allowances carry the epoch they were made under
        self.refine_tau = 4e-6
        # coarse pass of the two-pass flow: "box" = the one-plane box-only sweep (asdf_decode_grid_box) with exact re-evaluation of
        # the voxels that can move the box and a random audit of the others, "exact" = an ordinary sweep; ASDF_COARSE overrides
        fast = fast_requested()
        self.coarse_mode = os.environ.get("ASDF_COARSE", FAST_COARSE if fast else DEFAULT_COARSE)
        if self.coarse_mode not in ("exact", "box"):
            raise ValueError("ASDF_COARSE must be 'exact' or 'box', not %r" % self.coarse_mode)
        # fine pass: "band" = one-plane sweep + re-evaluation (as the ordinary sweep would) of the corners of every cell that can be
        # active (asdf_decode_grid_band) + the audit - for volumes that go to marching cubes and nowhere else; "exact" = ordinary
        self.fine_mode = os.environ.get("ASDF_FINE", FAST_FINE if fast else DEFAULT_FINE)
        if self.fine_mode not in ("exact", "band"):
            raise ValueError("ASDF_FINE must be 'exact' or 'band', not %r" % self.fine_mode)
        self._cluster_off = False    # the cluster form of the short-list kernel has been switched off after a fault report
        self._band_failures = 0
        self._fine_compare_in_flight = False     # a whole-zoom-lattice comparison has been enqueued and not judged yet
        self._band_skip = False      # the next fine_begin runs an ordinary sweep (a band sweep was just refused)
        self._force_f32_once = False # ... and on the fp32 chain (its near-level list overflowed)
        self.band_stats = self._new_stats("band")
        self.box_stats = self._new_stats("box")
        # error allowance tau of the one-plane values: TAU_FACTOR x the estimated largest |one-plane - exact| over a lattice.  Calibrated
        # per decoder and scale set from a whole-lattice comparison, then RE-ESTIMATED FROM EVERY SWEEP'S AUDIT: tau in use =
        # TAU_FACTOR x the largest estimate of the last 16 sweeps (8 samples), where a sweep's estimate is its audit's largest error x
        # the tail ratio (whole-lattice maximum / audit-sample maximum, measured in the calibration sweep)
        self._box_tau = None
        self._box_epoch = -1
        self._box_failures = 0
        self._tail = 1.0
        self._tail_by_n = {}         # tail ratio per uniform sample size (a ladder of powers of two), from the last calibration
        self._cal_points = 0         # lattice size of the last calibration: a much larger lattice needs its own
        self._coarse_since_cal = 0
        # the same for the zoom lattice of the fine pass: its own epoch, lattice size and tail ladder
        self._fine_epoch = -1
        self._fine_cal_points = 0
        self._tail_fine = 1.0
        self._tail_by_n_fine = {}
        self._next_recal = "coarse"  # which lattice the next periodic whole-lattice comparison measures
        self._err_window = collections.deque(maxlen=16)
        self.audit_voxels = AUDIT_VOXELS
        # what the certificate of the one-plane sweeps rests on, as measured (bench.py prints it as the `certificate` block)
        self.cert = {"calibrations": 0, "tail_ratio": None, "tail_ratio_max": None, "lattice_max_error": None, "lattice_sigma": None,
                     "lattice_max_over_sigma": None, "neighbour_correlation": None, "audited_sweeps": 0, "shell_picks": 0,
                     "shell_population_max": 0, "uniform_picks": 0, "min_margin_tau_over_estimate": None, "min_tau_over_sigma": None,
                     "audit_sigma_max": None, "refusals_for_error": 0,
                     # the same whole-lattice comparison on the zoom lattice of a fine pass
                     "fine_calibrations": 0, "fine_lattice_max_error": None, "fine_lattice_sigma": None, "fine_max_over_sigma": None,
                     "fine_tail_ratio": None, "fine_tail_ratio_max": None, "fine_neighbour_correlation": None}
        # what a run did, for the `sweeps.json` next to its meshes (reconstruct / dist_reconstruct): repeats and mode switches
        self.events = {"repeated_sweeps": 0, "modes_switched_off": [], "fp32_fallback": False, "samples_in_one_go": 0}
        self.event_log = None      # set to a list to collect (start, end) torch.cuda.Event pairs around every K1 launch
        self.box_event_log = None  # the same for the one-plane kernel of the box / band sweeps, plus the sweep's record tensor
        self.box_event_stride = 1  # ... around every box_event_stride-th of them (four event records per sweep are a few percent of a 64^3 sample)
        self._box_event_tick = 0

    @staticmethod
    def _new_stats(kind):
        return {kind: 0, "exact": 0, "fallback": 0, "max_err": 0.0, "audit_max_err": 0.0, "audit_evals": 0, "audit_flips": 0,
                "max_candidates" if kind == "box" else "max_marked": 0, "tau_min": None, "tau_max": None}

    def set_audit(self, voxels, seed=None):
        """Audit sample of the one-plane sweeps: `voxels` decided voxels per sweep and head are re-evaluated exactly (0 = off);
        `seed` restarts the draw (asdf_decoder_set_audit)."""
        seed = 0x5DF5A11D00000000 if seed is None else int(seed)
        _native.check(self._L.asdf_decoder_set_audit(self._h, int(voxels), ctypes.c_uint64(seed & 0xFFFFFFFFFFFFFFFF)), "asdf_decoder_set_audit")
        self.audit_voxels = int(voxels)

    def set_fast(self, on=True):
        """Opt in to (or out of) the audited one-plane sweeps for mesh-producing calls: coarse "box" + fine "band" (what ASDF_FAST=1 /
        `--fast` select when the decoder is packed), or ordinary sweeps in both passes - the default."""
        self.coarse_mode, self.fine_mode = (FAST_COARSE, FAST_FINE) if on else ("exact", "exact")

    def set_math(self, math):
        """Select the arithmetic of the hidden GEMMs ("f32" / "f16x3"); raises for NeRF-encoded decoders and f16x3."""
        code = {"f32": _native.MATH_F32, "f16x3": _native.MATH_F16X3}[math]
        _native.check(self._L.asdf_decoder_set_math(self._h, code), "asdf_decoder_set_math")
        self.math = math

    def set_refine(self, tau):
        """Near-level refinement of split-half grid sweeps (asdf_decoder_set_refine): voxels with |sdf| < tau are re-evaluated
        on the fp32 chain so that boxes and surfaces are sign-for-sign those of the fp32 kernel.  Default 4e-6; 0 = off."""
        _native.check(self._L.asdf_decoder_set_refine(self._h, ctypes.c_float(float(tau))), "asdf_decoder_set_refine")
        self.refine_tau = float(tau)

    @staticmethod
    def _range_words(rec):
        """(fp16 range violations, near-level list overflowed) of a bbox / sweep record: words 7 / 15, bit 30 = the flag."""
        w = (int(rec[7]), int(rec[15]))
        return sum(v & (CLUSTER_FAULT_BIT - 1) for v in w), bool((w[0] | w[1]) & NEAR_OVERFLOW_BIT)

    def _note_cluster_fault(self, rec):
        """Bit 29 of word 7 of a bbox record / word 16 + 11 of a one-plane sweep's record: a member of the short-list kernel's cluster
        form waited longer than its bound for another one (csrc/sdf_mlp_short_kernel.h).  The sweep the record belongs to is COMPLETE -
        the tile form behind the launch evaluated the list, same bits - and the form is switched off for this decoder, once, here
        (a recoverable failure instead of a trap that takes the HIP context along)."""
        hit = bool(int(rec[7]) & CLUSTER_FAULT_BIT) or (len(rec) > 27 and int(rec[27]) != 0)
        if hit and not self._cluster_off:
            import logging
            logging.warning("short-list kernel: a member of a cluster did not arrive in time; the tile form evaluated the list, the cluster "
                            "form is switched off for this decoder")
            self._cluster_off = True
            self.events["modes_switched_off"].append("cluster form of the short-list kernel: a member did not arrive within the bound")
            if self._h is not None:
                _native.check(self._L.asdf_decoder_set_cluster_list(self._h, 0), "asdf_decoder_set_cluster_list")
        return hit

    def fall_back_if_overflowed(self, bbox_host, epoch=None):
        """bbox words 7 / 15 count points whose activations left the fp16 range of the split-half planes; they are non-zero
        only for a sweep that ran under f16x3 (the fp32 kernel leaves them 0), so the decision rests on the record alone -
        whatever arithmetic the decoder has been switched to since that sweep was queued.  Returns True when the caller has
        to repeat the sweep the record belongs to: the activation scales were re-calibrated from the peaks that sweep left
        in the status record (first resort), or - when the scales cannot be lowered any further - the decoder was switched
        to the fp32 MFMA chain for good.  `epoch` = the scale epoch (`_recalibrations`) the sweep was LAUNCHED under: a
        sweep queued before a re-calibration that has happened since is simply repeated under the current scales - its
        overflow says nothing about them.  Bit 30 (the near-level refinement list overflowed: the signs next to the level are not
        certified to be the fp32 chain's) makes the NEXT sweep of this decoder run on the fp32 chain, once."""
        bad, near_over = self._range_words(bbox_host)
        self._note_cluster_fault(bbox_host)
        if near_over:
            import logging
            logging.warning("split-half sweep: more than %d voxels within %.1e of the level; repeated on the fp32 chain", NEAR_CAP, self.refine_tau)
            self._force_f32_once = True
        if not bad:
            return near_over
        if epoch is not None and epoch != self._recalibrations:
            return True
        self._recover(bad)
        return True

    def _recover(self, bad, status=None):
        if self.math == "f16x3" and self._recalibrations < 4:
            if self.calibrate(status):
                return
        self._to_f32(bad)

    def _to_f32(self, bad):
        if self.math != "f32":
            import logging
            logging.warning("split-half decoder: %d activations left the fp16 range; falling back to the fp32 MFMA kernel", bad)
            self.set_math("f32")
            self.events["fp32_fallback"] = True
            self.events["modes_switched_off"].append("arithmetic: %d activations outside the fp16 range after %d re-calibrations - fp32 chain" % (bad, self._recalibrations))

    def _status(self, clear):
        out = (ctypes.c_int32 * 16)()
        with torch.cuda.device(self.device):
            _native.check(self._L.asdf_decoder_status(self._h, out, 1 if clear else 0, self._stream()), "asdf_decoder_status")
        return np.frombuffer(out, dtype=np.int32).copy()

    def range_violations(self, clear=True):
        """Number of (point, lane-half) pairs whose activations left the fp16 range in split-half sweeps of this decoder
        since the last clear - the decoder-owned status word, independent of any bbox buffer.  Synchronises the stream."""
        return int(self._status(clear)[0])

    def act_scales(self):
        out = (ctypes.c_float * 6)()
        _native.check(self._L.asdf_decoder_get_act_scales(self._h, out), "asdf_decoder_get_act_scales")
        return np.frombuffer(out, dtype=np.float32).reshape(2, 3).copy()

    def set_act_scales(self, sx):
        """S_x per MLP and activation vector (h0, h1, h2), powers of two; re-binds the current sample (the folded constants
        carry the layer-2 scale)."""
        arr = (ctypes.c_float * 6)(*[float(v) for v in np.asarray(sx, np.float32).reshape(-1)])
        with torch.cuda.device(self.device):
            _native.check(self._L.asdf_decoder_set_act_scales(self._h, arr, self._stream()), "asdf_decoder_set_act_scales")
        self._recalibrations += 1          # a new scale epoch: allowances and tickets made under the old scales are stale
        if self._bound is not None:
            self.set_sample(*self._bound)

    def calibrate(self, status=None):
        """Choose every S_x from the peak plane values of the sweeps since the last status clear (asdf_decoder_status words
        4..6 / 8..10): the power of two that puts the layer's peak in [1024, 2048) - a factor 32..64 of headroom under the fp16
        maximum for later samples, while every activation down to 2^-13 of the peak keeps two full planes.  A peak that overflowed (inf / NaN
        pattern) moves that scale down by 2^6 instead.  A (MLP, layer) whose peak is zero - the MLP did not run in those sweeps -
        stays uncalibrated and is picked up by the first later sweep that evaluates it.  Returns True when a scale changed (the
        caller repeats its sweep); every change starts a new scale epoch (set_act_scales)."""
        st = self._status(clear=True) if status is None else status
        cur = self.act_scales()
        new = cur.copy()
        for h in range(1 if self.combined else 2):
            for l in range(3):
                bits = int(st[4 + 4 * h + l])
                if bits == 0:
                    continue                                  # this MLP did not run (or produced only zeros)
                self._cal_done[h, l] = True
                peak = float(np.int32(bits).view(np.float32))
                if not np.isfinite(peak) or peak >= 65504.0:
                    new[h, l] = max(cur[h, l] / 64.0, 2.0 ** -24)      # (at the floor already: nothing changes, the caller goes to fp32)
                else:
                    want = cur[h, l] * 2.0 ** np.floor(np.log2(2048.0 / peak))      # peak / cur = the activation itself
                    new[h, l] = float(np.clip(want, 2.0 ** -24, 2.0 ** 24))
        if np.array_equal(new, cur):
            return False
        import logging
        logging.info("split-half decoder: activation scales %s -> %s", cur.tolist(), new.tolist())
        self.set_act_scales(new)
        return True

    @property
    def _calibrated(self):
        return bool(self._cal_done.all())

    @_calibrated.setter
    def _calibrated(self, value):
        """True = treat every activation scale as calibrated (tests of the range guard keep the default S_x = 8 this way)."""
        self._cal_done[:] = bool(value)

    def _needs_calibration(self, hand, obj):
        if self.math != "f16x3" or self._cal_attempts >= 4 or self._recalibrations >= 4:
            return False
        heads = [0] if self.combined else [h for h, on in ((0, hand), (1, obj)) if on]
        return any(not self._cal_done[h].all() for h in heads)

    def close(self):
        if getattr(self, "_h", None):
            self._L.asdf_decoder_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _code_slot(self):
        """The next pinned staging slot for one sample's codes: (latent [L] fp32, embed [2, MAX_POINT_FEATS, 4] fp32, event).  A slot is
        reused CODE_SLOTS set_sample calls later, after the event recorded behind ITS staging launch (long done: two or three samples
        are in flight, each bound at most three times)."""
        if not hasattr(self, "_code_ring"):
            self._code_ring = [[torch.zeros(self.latent_size, dtype=torch.float32).pin_memory(),
                                torch.zeros((2, _native.MAX_POINT_FEATS, 4), dtype=torch.float32).pin_memory(), None] for _ in range(CODE_SLOTS)]
            self._code_next = 0
        slot = self._code_ring[self._code_next % CODE_SLOTS]
        self._code_next += 1
        if slot[2] is not None:
            slot[2].synchronize()
        return slot

    def set_sample(self, latent_vec, embed=None):
        """Bind a latent code [1, L] and optional per-head affine embeddings (hand_E, obj_E).

        A latent on the DEVICE (the reference's case: it comes out of the encoder, reconstruct.py:83-84) is read where it lies.  A latent
        on the HOST (codes saved by an encoder run, synthetic codes) is NOT uploaded with a copy: it is placed in a pinned staging slot
        and the fold's staging launch reads it over the link in stream order (asdf_decoder_set_sample_host) - the runtime's copy of
        1 KB is a shader blit that cannot get a wave slot while a persistent sweep owns every compute unit (round 5's eval-mode
        trace: 21.5 ms per sample resident), and a plain `.to(device)` of pageable memory makes the caller wait
        for everything queued."""
        if latent_vec.numel() != self.latent_size:
            raise ValueError("latent has %d elements, expected %d" % (latent_vec.numel(), self.latent_size))
        host = latent_vec.device.type == "cpu"
        slot = None
        emb_ptr = None
        if self.nerf_features:
            if embed is not None:
                raise ValueError("a NeRF-encoded decoder takes raw xyz; no affine embedding applies")
        elif embed is not None:
            slot = self._code_slot()
            buf = slot[1].numpy()
            buf[:] = 0
            for h in range(len(self._pf)):
                e = np.asarray(embed[h], dtype=np.float64)
                if e.shape != (self._pf[h], 4):
                    raise ValueError("embedding of head %d has shape %s, expected %s" % (h, e.shape, (self._pf[h], 4)))
                buf[h, :self._pf[h]] = e.astype(np.float32)
            emb_ptr = ctypes.c_void_p(slot[1].data_ptr())
        elif not self.nerf_features and any(f != 3 for f in self._pf):
            raise ValueError("this decoder needs a point embedding (point features per head: %s)" % (self._pf,))
        self._bound = (latent_vec, embed)
        with torch.cuda.device(self.device):
            if host:
                slot = slot or self._code_slot()
                slot[0].copy_(latent_vec.detach().reshape(-1).to(torch.float32))
                self._latent = None
                _native.check(self._L.asdf_decoder_set_sample_host(self._h, slot[0].data_ptr(), emb_ptr, self._stream()),
                              "asdf_decoder_set_sample_host")
            else:
                lat = latent_vec.detach().reshape(-1).to(device=self.device, dtype=torch.float32).contiguous()
                self._latent = lat   # keep the device buffer alive until the next set_sample
                # (the embedding of a device-side latent still travels by hipMemcpyAsync from the pinned slot, in stream order)
                _native.check(self._L.asdf_decoder_set_sample(self._h, lat.data_ptr(), emb_ptr, self._stream()),
                              "asdf_decoder_set_sample")
            if slot is not None:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.device))
                slot[2] = ev

    def decode_grid(self, N, origin3, voxel_size, grid_mode=_native.GRID_REFERENCE, want_bbox=True, hand=True, obj=True,
                    check_range=None, lattice=None):
        """Heads on the N^3 lattice. Returns (sdf_hand [N,N,N], sdf_obj [N,N,N], bbox int32[16] or None), all device
        tensors; a head switched off (`hand` / `obj` False, SeparateDecoder only) is not evaluated and returns None.

        Under the split-half arithmetic a sweep WITH a bbox reports fp16 range violations in words 7 / 15 of the record
        (the caller reads it anyway: fall_back_if_overflowed).  A sweep WITHOUT one is guarded here: the decoder's status
        word is read behind the launch (one stream synchronisation) and, if it is non-zero, the decoder switches to the
        fp32 kernel and the sweep is repeated.  check_range=False skips that for callers that know the range is safe.

        `lattice` (device float32[4] = origin, voxel size; origin3 / voxel_size are then ignored): the lattice is read on the device
        (asdf_decode_grid_dev) - the fine pass of a sample enqueued in one go, behind asdf_zoom_cube."""
        if self.combined:
            hand = obj = True
        want_hand, want_obj = hand, obj
        hand = torch.empty((N, N, N), dtype=torch.float32, device=self.device) if hand else None
        obj = torch.empty((N, N, N), dtype=torch.float32, device=self.device) if obj else None
        bbox = torch.empty(16, dtype=torch.int32, device=self.device) if want_bbox else None
        org = (ctypes.c_float * 3)(*[float(np.float32(o)) for o in origin3]) if lattice is None else None
        guard = self.math == "f16x3" and not want_bbox and (check_range is None or check_range)
        once_f32 = self._force_f32_once and self.math == "f16x3"
        self._force_f32_once = False
        if once_f32:
            self.set_math("f32")          # one sweep on the fp32 chain (the previous one's near-level list overflowed)
            guard = False

        def launch():
            ev = None
            with torch.cuda.device(self.device):
                if self.event_log is not None:
                    # the events bracket the decoder kernel itself (asdf_decoder_time_next_sweep), not the bbox / refinement
                    # kernels around it; a first record() makes torch create the handles
                    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    ev[0].record()
                    ev[1].record()
                    _native.check(self._L.asdf_decoder_time_next_sweep(self._h, ctypes.c_void_p(ev[0].cuda_event),
                                                                       ctypes.c_void_p(ev[1].cuda_event)), "asdf_decoder_time_next_sweep")
                if lattice is not None:
                    _native.check(self._L.asdf_decode_grid_dev(self._h, int(N), lattice.data_ptr(), int(grid_mode),
                                                               hand.data_ptr() if hand is not None else None,
                                                               obj.data_ptr() if obj is not None else None,
                                                               bbox.data_ptr() if want_bbox else None, self._stream()),
                                  "asdf_decode_grid_dev")
                else:
                    _native.check(self._L.asdf_decode_grid(self._h, int(N), org, ctypes.c_float(float(np.float32(voxel_size))),
                                                           int(grid_mode), hand.data_ptr() if hand is not None else None,
                                                           obj.data_ptr() if obj is not None else None,
                                                           bbox.data_ptr() if want_bbox else None, self._stream()),
                                  "asdf_decode_grid")
            return ev

        # A split-half sweep that evaluates an MLP whose activation scales are still at their default calibrates them from
        # the peaks it leaves in the status record (one stream synchronisation, once per decoder and MLP); a bbox-less
        # sweep is additionally guarded.  Only the events of the launch whose volumes are returned are logged.
        first = check_range is not False and self._needs_calibration(want_hand, want_obj)
        if guard or first:
            self._status(clear=True)                # earlier sweeps answer for themselves
        ev = launch()
        if first:
            self._cal_attempts += 1
            st = self._status(clear=False)
            if self.calibrate(st):
                self._status(clear=True)
                ev = launch()                       # the calibrated image; its own report is read below / by the caller
        if guard:
            for _ in range(5):
                st = self._status(clear=True)
                if int(st[1]) and self.math == "f16x3":
                    # more near-level voxels than the refinement list holds: this sweep on the fp32 chain
                    self.set_math("f32")
                    ev = launch()
                    self.set_math("f16x3")
                    break
                if not int(st[0]):
                    break
                self._recover(int(st[0]), st)
                ev = launch()
        if once_f32:
            self.set_math("f16x3")
        if ev is not None:
            self.event_log.append(ev)
        return hand, obj, bbox

    # ---- the one-plane sweeps (asdf_decode_grid_box / asdf_decode_grid_band): allowance, record, acceptance -------------------
    def _tau_current(self):
        return float(np.clip(TAU_FACTOR * max(self._err_window), 1e-6, 0.05)) if self._err_window else None

    def _one_plane_launch(self, fn, name, N, origin3, voxel_size, grid_mode, hand, obj, tau, lattice=None):
        """`lattice` (device float32[4]): the _dev form of the entry point - origin and voxel size are read on the device."""
        if self.combined:
            hand = obj = True
        sh = torch.empty((N, N, N), dtype=torch.float32, device=self.device) if hand else None
        so = torch.empty((N, N, N), dtype=torch.float32, device=self.device) if obj else None
        rec = torch.empty(REC_WORDS, dtype=torch.int32, device=self.device)
        org = (ctypes.c_float * 3)(*[float(np.float32(o)) for o in origin3]) if lattice is None else None
        with torch.cuda.device(self.device):
            ev = None
            self._box_event_tick += 1
            if self.box_event_log is not None and self._box_event_tick % self.box_event_stride == 0:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
                ev[1].record()
                _native.check(self._L.asdf_decoder_time_next_sweep(self._h, ctypes.c_void_p(ev[0].cuda_event),
                                                                   ctypes.c_void_p(ev[1].cuda_event)), "asdf_decoder_time_next_sweep")
            if lattice is not None:
                _native.check(fn(self._h, int(N), lattice.data_ptr(), int(grid_mode), ctypes.c_float(float(tau)),
                                 sh.data_ptr() if sh is not None else None, so.data_ptr() if so is not None else None, rec.data_ptr(),
                                 self._stream()), name)
            else:
                _native.check(fn(self._h, int(N), org, ctypes.c_float(float(np.float32(voxel_size))), int(grid_mode),
                                 ctypes.c_float(float(tau)), sh.data_ptr() if sh is not None else None,
                                 so.data_ptr() if so is not None else None, rec.data_ptr(), self._stream()), name)
            if ev is not None:
                self.box_event_log.append(ev + (rec,))      # (words 28..31 of the record: shader-clock stamps of the sweep kernel)
        return rec, sh, so

    @staticmethod
    def audit_sizes(audit_voxels, points):
        """(uniform picks, shell budget) per sweep and head on a lattice of `points` voxels - csrc/decoder.hip: audit_size."""
        if audit_voxels <= 0:
            return 0, 0
        n = int(min(audit_voxels, max(points // 64, 64)))
        return n - n // 2, n // 2

    def _tail_for(self, points, which="coarse"):
        """Tail ratio for the uniform half of an audit on a lattice of `points` voxels: the last whole-lattice comparison's entry
        (of the coarse lattice, or of the zoom lattice for a band sweep once one has been measured) for the largest sample size not
        above it (a smaller sample sees less of the tail: conservative).  The ladder starts at 1, so every audit size has an entry
        (below 32 picks the fallback used to be the stale scalar)."""
        uniform, _ = self.audit_sizes(self.audit_voxels, points)
        ladder, scalar = (self._tail_by_n_fine, self._tail_fine) if which == "fine" and self._tail_by_n_fine else (self._tail_by_n, self._tail)
        best = None
        for n, t in ladder.items():
            if n <= max(uniform, 1) and (best is None or n > best[0]):
                best = (n, t)
        if best is None and ladder:
            best = min(ladder.items())            # below the ladder: its smallest entry, the conservative one
        return best[1] if best is not None else scalar

    def _judge(self, r, tau, stats, cap_word, cap, points=None, which="coarse"):
        """Verdict on the record of one one-plane sweep launched with allowance tau: (accepted, range violations, reason).
        Accepted = no fp16 range violation, every list within its capacity, no contradiction, the largest |exact - one-plane| over
        the re-evaluated voxels AND the audit's estimate of the lattice maximum within TAU_ACCEPT x tau, and no audit voxel whose sign the
        exact value contradicts.  The estimate of an ACCEPTED sweep enters the allowance of the sweeps that follow; a sweep refused for
        its error invalidates the allowance instead: the next coarse pass measures the whole lattice again (a refusal
        must not inflate the allowance by itself)."""
        bad, near_over = self._range_words(r)
        self._note_cluster_fault(r)
        f = lambda w: float(np.int32(r[w]).view(np.float32))
        err, audit = f(19), f(35)
        flips, evals = int(r[36]), int(r[37])
        shell_picks, shell_pop, sumsq = int(r[39]), int(r[40]), f(41)
        listed = max(int(r[w]) for w in cap_word)
        tail = self._tail_for(points, which) if points else self._tail
        est = audit * tail
        sigma = float(np.sqrt(sumsq / evals)) if evals > 0 and np.isfinite(sumsq) and sumsq >= 0 else float("nan")
        stats["max_err"] = max(stats["max_err"], err)
        stats["audit_max_err"] = max(stats["audit_max_err"], audit)
        stats["audit_evals"] += evals
        stats["audit_flips"] += flips
        key = "max_candidates" if "max_candidates" in stats else "max_marked"
        stats[key] = max(stats[key], listed)
        stats["tau_min"] = tau if stats["tau_min"] is None else min(stats["tau_min"], tau)
        stats["tau_max"] = tau if stats["tau_max"] is None else max(stats["tau_max"], tau)
        reason, for_error = None, False
        if bad:
            reason = "%d activations left the fp16 range" % bad
        elif listed > cap:
            reason = "%d voxels listed, capacity %d" % (listed, cap)
        elif near_over or int(r[38]):
            reason = "near-level list overflowed"
        elif int(r[18]):
            reason, for_error = "a voxel taken as certainly negative was not", True
        elif flips:
            reason, for_error = "%d of %d audit voxels have the other sign" % (flips, evals), True
        elif not (err <= TAU_ACCEPT * tau):
            reason, for_error = "error %.3g on the re-evaluated voxels against allowance %.3g" % (err, tau), True
        elif not (est <= TAU_ACCEPT * tau):
            reason, for_error = "audit error %.3g (x tail %.2f) against allowance %.3g" % (audit, tail, tau), True
        elif self.audit_voxels and evals == 0:
            reason = "the audit evaluated nothing"
        c = self.cert
        if reason is None:
            if np.isfinite(est) and np.isfinite(err):
                # per-sweep re-estimate of the lattice maximum: the next sweeps' allowance follows the samples
                self._err_window.append(max(est, err, 2.5e-7))
                self._box_tau = self._tau_current()
            c["audited_sweeps"] += 1
            c["shell_picks"] += shell_picks
            c["shell_population_max"] = max(c["shell_population_max"], shell_pop)
            c["uniform_picks"] += max(evals - shell_picks, 0)
            worst = max(est, err, 2.5e-7)
            c["min_margin_tau_over_estimate"] = tau / worst if c["min_margin_tau_over_estimate"] is None else min(c["min_margin_tau_over_estimate"], tau / worst)
            if np.isfinite(sigma) and sigma > 0:
                c["min_tau_over_sigma"] = tau / sigma if c["min_tau_over_sigma"] is None else min(c["min_tau_over_sigma"], tau / sigma)
                c["audit_sigma_max"] = sigma if c["audit_sigma_max"] is None else max(c["audit_sigma_max"], sigma)
        elif for_error:
            c["refusals_for_error"] += 1
            # the allowance is void until the whole lattice has been measured again - the coarse lattice by the next coarse pass, the
            # zoom lattice by the next fine pass
            self._box_epoch = self._fine_epoch = -1
        return reason is None, bad, near_over, reason

    # ---- records travel to the host behind their own sweep: an asynchronous copy into pinned memory + an event recorded right behind
    # it.  The reader waits for THAT event - a `.cpu()` would wait for everything queued on the stream since (the next sample's
    # pass 1, marching cubes ...), which at N = 64 was a tenth of a sample of idle GPU (profiles/r04_small_lattice_traces.txt).
    def _record_to_host(self, rec):
        if rec is None or rec.device.type != "cuda":
            return None
        if not hasattr(self, "_rec_ring"):
            self._rec_ring = [torch.zeros(REC_WORDS, dtype=torch.int32).pin_memory() for _ in range(32)]
            self._rec_turn = 0
        slot = self._rec_ring[self._rec_turn % len(self._rec_ring)]
        self._rec_turn += 1
        n = rec.numel()
        slot[:n].copy_(rec, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        return slot, n, ev

    @staticmethod
    def _record_of(ticket):
        """The ticket's record as a host array (waits for the sweep it belongs to, and for nothing queued behind it)."""
        host = ticket.get("host")
        if host is not None:
            slot, n, ev = host
            ev.synchronize()
            return slot[:n].numpy().copy()
        return ticket["rec"].cpu().numpy()

    # ---- the coarse pass of the two-pass flow (utils/mesh.py:27-63): consumed only through get_higher_res_cube, i.e. the
    # per-head boxes of its negative voxels.  begin() enqueues, finish() reads the record back (the one host
    # synchronisation the zoom cube needs anyway) and repeats the sweep where a guard asks for it.
    def _one_plane_ok(self):
        """The one-plane kernels' own weight image exists under the current activation scales (asdf_decoder_one_plane_usable)."""
        return self._h is None or bool(self._L.asdf_decoder_one_plane_usable(self._h))

    def _box_usable(self):
        return self.coarse_mode == "box" and self.math == "f16x3" and self._one_plane_ok()

    def _box_launch(self, N, origin3, voxel_size, grid_mode, hand, obj, tau):
        return self._one_plane_launch(self._L.asdf_decode_grid_box, "asdf_decode_grid_box", N, origin3, voxel_size, grid_mode, hand, obj, tau)

    def _allowance_valid(self, N=None):
        """The allowance was measured under the current activation scales, no sweep has been refused for its error since, and the
        lattice it was measured on is not much smaller than the one to be swept (the largest of 64 x as many errors is larger)."""
        return (self._box_tau is not None and self._box_epoch == self._recalibrations and
                (N is None or int(N) ** 3 <= 8 * self._cal_points))

    def _fine_valid(self, N=None):
        """A zoom lattice has been compared as a whole under the current activation scales, no sweep has been refused for its error
        since, and it was not much smaller than the one to be swept."""
        return (self._fine_epoch == self._recalibrations and (N is None or int(N) ** 3 <= 8 * self._fine_cal_points))

    def coarse_begin(self, N, origin3, voxel_size, grid_mode=_native.GRID_REFERENCE, hand=True, obj=True):
        """Enqueue the coarse pass of one sample; returns a ticket for coarse_finish."""
        args = (N, origin3, voxel_size, grid_mode, hand, obj)
        self._coarse_since_cal += 1
        # the periodic whole-lattice re-measurement: this pass runs both ways - when it is the coarse lattice's turn (the zoom
        # lattice's turn is taken by this sample's fine pass: fine_begin)
        due = self._coarse_since_cal > RECAL_EVERY and (self._next_recal == "coarse" or not self._band_usable() or
                                                        self._coarse_since_cal > 2 * RECAL_EVERY)      # (no fine pass took its turn)
        if self._box_usable() and self._allowance_valid(N) and not self._force_f32_once and not due:
            rec, sh, so = self._box_launch(*args, self._box_tau)
            return {"kind": "box", "args": args, "rec": rec, "keep": (sh, so), "tau": self._box_tau, "epoch": self._recalibrations,
                    "host": self._record_to_host(rec)}
        h, o, bbox = self.decode_grid(N, origin3, voxel_size, grid_mode, hand=hand, obj=obj)
        return {"kind": "exact", "args": args, "rec": bbox, "keep": (h, o), "epoch": self._recalibrations, "recalibrate": due,
                "host": self._record_to_host(bbox)}

    def coarse_judge(self, ticket):
        """Verdict on a box-only coarse sweep WITHOUT launching anything: (accepted, int32[16] record or None, calibrate_allowance).
        Waits for the sweep's record (and for nothing queued behind it).  A refusal is booked here - counters, failure streak, new
        activation scales after a range violation - so that the caller only has to repeat the pass as an ordinary sweep
        (coarse_finish(ticket, judged=...)) while the decoder is bound to the ticket's sample."""
        import logging
        N = ticket["args"][0]
        r = self._record_of(ticket)
        if ticket["kind"] == "exact":
            # the ORDINARY coarse sweep of a sample that was enqueued in one go (round 6): its only guards are the split-half
            # arithmetic's own - fp16 range (re-calibrate, or at last the fp32 chain) and the near-level list (one sweep on the fp32
            # chain).  A clean record is the pass; otherwise the recovery is booked here and the caller repeats the sample step by step
            if self.fall_back_if_overflowed(r, ticket["epoch"]):
                self.events["repeated_sweeps"] += 1
                return False, None, False
            self.box_stats["exact"] += 1
            b = r[:16].copy()
            b[7] &= CLUSTER_FAULT_BIT - 1
            b[15] &= CLUSTER_FAULT_BIT - 1
            return True, b, False
        if ticket["epoch"] != self._recalibrations:
            ok, bad, reason = False, 0, "launched under activation scales that have been re-calibrated since"
        else:
            ok, bad, _, reason = self._judge(r, ticket["tau"], self.box_stats, (32,), CAND_CAP, points=int(N) ** 3)
        if ok:
            self.box_stats["box"] += 1
            self._box_failures = 0              # (three refusals IN A ROW switch the mode off)
            return True, r[:16].copy(), False
        calibrate_allowance = True
        self.box_stats["fallback"] += 1
        self.events["repeated_sweeps"] += 1
        if bad:
            self._recover(bad)                  # new activation scales: the allowance is re-calibrated with them
        else:
            logging.warning("box-only coarse sweep not accepted (%s): repeated as an ordinary sweep", reason)
            if ticket["epoch"] == self._recalibrations:
                self._box_failures += 1
                # refused for its error: the allowance is void (_judge) and is measured again by the repeat, on this very lattice;
                # refused for a capacity / list overflow: the allowance stands
                calibrate_allowance = not self._allowance_valid(N)
            if self._box_failures >= 3:
                logging.warning("box-only coarse sweep switched off for this decoder")
                self.coarse_mode = "exact"
                self.events["modes_switched_off"].append("coarse: three refusals in a row (%s)" % reason)
        return False, None, calibrate_allowance

    def coarse_finish(self, ticket, judged=None):
        """int32[16] host record of the coarse pass (words 0..5 / 8..13: boxes of the negative voxels; 6 / 14: non-zero
        iff there is one).  Synchronises with the sweep; a sweep whose guards fired is repeated here - the decoder must
        still be bound to the ticket's sample.  `judged` = the result of an earlier coarse_judge(ticket) (the sample pipeline judges
        a speculative coarse pass first and re-binds the sample only when it has to be repeated)."""
        N, origin3, voxel_size, grid_mode, hand, obj = ticket["args"]
        calibrate_allowance = True
        if ticket["kind"] == "box" or judged is not None:
            # (judged is given for an ordinary speculative coarse pass as well: refused there = its range / near-level guards fired and
            # the recovery has been booked by coarse_judge - launch it again)
            ok, b, calibrate_allowance = judged if judged is not None else self.coarse_judge(ticket)
            if ok:
                return b
            h, o, bbox = self.decode_grid(N, origin3, voxel_size, grid_mode, hand=hand, obj=obj)
            ticket = {"kind": "exact", "args": ticket["args"], "rec": bbox, "keep": (h, o), "epoch": self._recalibrations}
        b = self._record_of(ticket)
        keep, epoch = ticket["keep"], ticket["epoch"]
        while self.fall_back_if_overflowed(b, epoch):      # split-half planes out of fp16 range: re-calibrated, or at last fp32
            self.events["repeated_sweeps"] += 1
            h, o, bbox = self.decode_grid(N, origin3, voxel_size, grid_mode, hand=hand, obj=obj)
            b, keep, epoch = bbox.cpu().numpy(), (h, o), self._recalibrations
        self.box_stats["exact"] += 1
        # the error allowance of the one-plane kernel (shared by the box-only coarse sweep and the narrow-band fine sweep) is
        # calibrated here, on the coarse lattice, while the decoder is bound to this sample
        if (self._box_usable() or self._band_usable()) and calibrate_allowance and (not self._allowance_valid(N) or ticket.get("recalibrate")):
            self._calibrate_box(ticket["args"], keep)
            if ticket.get("recalibrate") and self._band_usable():
                self._next_recal = "fine"           # the next periodic comparison measures a zoom lattice
        b = b.copy()
        b[7] &= CLUSTER_FAULT_BIT - 1
        b[15] &= CLUSTER_FAULT_BIT - 1
        return b

    # ---- both passes of a sample ENQUEUED in one go (round 5): the box-only coarse sweep, the zoom cube computed on
    # the device from its boxes (asdf_zoom_cube: the six fp32 operations of utils/mesh.py:250-254, bit-equal to the host's), and the
    # narrow-band fine sweep reading its lattice from those device words.  The host judges both records afterwards; a refused coarse
    # sweep makes the caller repeat the sample step by step (coarse_finish(ticket, judged) -> host zoom cube -> fine_begin).
    def can_speculate(self, N):
        """Both passes may be enqueued back to back: both one-plane modes on and usable, both whole-lattice comparisons valid for this
        lattice size, no comparison due, nothing that forces the next sweep onto another path."""
        return (self._box_usable() and self._band_usable() and self._allowance_valid(N) and self._fine_valid(N) and
                not self._force_f32_once and not self._band_skip and self._coarse_since_cal + 1 <= RECAL_EVERY)

    def can_speculate_ordinary(self, hand=True, obj=True):
        """Round 6: ORDINARY sweeps in both passes (the product's default) may be enqueued back to back as well - coarse sweep, zoom cube
        on the device, fine sweep on that lattice (asdf_decode_grid_dev), marching cubes - whenever nothing can ask for a host decision
        in between: both passes are ordinary, the activation scales of the MLPs that run are calibrated, and no sweep has been ordered
        onto the fp32 chain.  What can still go wrong is reported in the two bbox records (fp16 range, near-level list) and handled
        when the sample is finished: the sample is then repeated step by step."""
        if self.coarse_mode == "box" and self.math == "f16x3" and self._one_plane_ok():
            return False
        if self._band_usable() or self._force_f32_once or self._band_skip:
            return False
        return not self._needs_calibration(hand, obj) or self.math == "f32"

    def two_pass_begin(self, N, voxel_size, grid_mode=_native.GRID_REFERENCE, hand=True, obj=True):
        """Enqueue coarse pass -> device zoom cube -> fine pass (mc_only) of the bound sample.  Returns None when the sample has to go
        step by step (can_speculate / can_speculate_ordinary), else a ticket: `coarse` / `fine` (tickets for coarse_judge /
        fine_needs_repeat), `lattice` (device float32[4]: origin, voxel size), `lattice_host` (pinned copy + event), `vol_hand` /
        `vol_obj` (the fine volumes)."""
        if not self.can_speculate(N):
            if not self.can_speculate_ordinary(hand, obj):
                return None
            # ---- ordinary sweeps in both passes, enqueued in one go
            self.events["samples_in_one_go"] += 1
            org = [-1.0, -1.0, -1.0]
            args = (N, org, voxel_size, grid_mode, hand, obj)
            h, o, bbox = self.decode_grid(N, org, voxel_size, grid_mode, hand=hand, obj=obj)
            coarse = {"kind": "exact", "args": args, "rec": bbox, "keep": (h, o), "epoch": self._recalibrations, "host": self._record_to_host(bbox)}
            lattice = torch.empty(4, dtype=torch.float32, device=self.device)
            with torch.cuda.device(self.device):
                _native.check(self._L.asdf_zoom_cube(bbox.data_ptr(), int(N), ctypes.c_float(float(np.float32(voxel_size))), int(bool(hand)),
                                                     int(bool(obj)), lattice.data_ptr(), self._stream()), "asdf_zoom_cube")
            lattice_host = self._record_to_host(lattice.view(torch.int32))
            vh, vo, bbox2 = self.decode_grid(N, None, None, grid_mode, want_bbox=self.math == "f16x3", hand=hand, obj=obj, lattice=lattice)
            fine = {"kind": "exact", "args": (N, None, None, grid_mode, hand, obj), "rec": bbox2, "epoch": self._recalibrations,
                    "host": self._record_to_host(bbox2)}
            return {"coarse": coarse, "fine": fine, "lattice": lattice, "lattice_host": lattice_host, "vol_hand": vh, "vol_obj": vo}
        # (a CombinedDecoder evaluates both columns whatever the flags say - the launches force that themselves - but the ZOOM CUBE and
        # the marching-cubes parts follow the caller's HandBranch / ObjectBranch like get_higher_res_cube, utils/mesh.py:239-247, and the
        # step-by-step path: ADVICE r05)
        self._coarse_since_cal += 1
        self.events["samples_in_one_go"] += 1
        tau = self._box_tau
        org = [-1.0, -1.0, -1.0]
        args = (N, org, voxel_size, grid_mode, hand, obj)
        rec, sh, so = self._box_launch(*args, tau)
        coarse = {"kind": "box", "args": args, "rec": rec, "keep": (sh, so), "tau": tau, "epoch": self._recalibrations,
                  "host": self._record_to_host(rec)}
        lattice = torch.empty(4, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _native.check(self._L.asdf_zoom_cube(rec.data_ptr(), int(N), ctypes.c_float(float(np.float32(voxel_size))), int(bool(hand)),
                                                 int(bool(obj)), lattice.data_ptr(), self._stream()), "asdf_zoom_cube")
        lattice_host = self._record_to_host(lattice.view(torch.int32))      # (behind the zoom kernel, IN FRONT of the fine sweep: its reader must not wait for that)
        rec2, vh, vo = self._one_plane_launch(self._L.asdf_decode_grid_band_dev, "asdf_decode_grid_band_dev", N, None, None, grid_mode,
                                              hand, obj, tau, lattice=lattice)
        fine = {"kind": "band", "args": (N, None, None, grid_mode, hand, obj), "rec": rec2, "tau": tau, "epoch": self._recalibrations,
                "host": self._record_to_host(rec2)}
        return {"coarse": coarse, "fine": fine, "lattice": lattice, "lattice_host": lattice_host, "vol_hand": vh, "vol_obj": vo}

    @classmethod
    def lattice_of(cls, ticket):
        """(origin [3 python floats], voxel size as a 0-dim fp32 CPU tensor) of a two_pass_begin ticket - what
        utils.mesh.zoom_cube_from_bboxes returns for the same boxes, bit for bit."""
        host = ticket.get("lattice_host")
        w = (cls._record_of({"host": host}) if host is not None else ticket["lattice"].view(torch.int32).cpu().numpy()).view(np.float32)
        return [float(w[0]), float(w[1]), float(w[2])], torch.tensor(w[3], dtype=torch.float32)

    # ---- the fine pass of the two-pass flow when its volumes go to marching cubes and nowhere else
    def _band_usable(self):
        return self.fine_mode == "band" and self.math == "f16x3" and self._one_plane_ok()

    def fine_begin(self, N, origin3, voxel_size, grid_mode=_native.GRID_REFERENCE, hand=True, obj=True, mc_only=False):
        """Enqueue the fine pass; returns (sdf_hand, sdf_obj, ticket).  The caller hands the ticket to fine_needs_repeat
        before it trusts the volumes (one read-back, at the point where it synchronises for marching cubes anyway).
        mc_only=True declares that the volumes go to marching cubes at level 0 and nowhere else: only then may a decoder
        set to fine_mode "band" deliver them exact next to the surface and sign-correct elsewhere."""
        args = (N, origin3, voxel_size, grid_mode, hand, obj)
        band = mc_only and self._band_usable() and not self._force_f32_once
        fine_due = self._coarse_since_cal > RECAL_EVERY and self._next_recal == "fine" and not self._fine_compare_in_flight
        if band and not self._band_skip and self._allowance_valid(N) and self._fine_valid(N) and not fine_due:
            rec, vh, vo = self._one_plane_launch(self._L.asdf_decode_grid_band, "asdf_decode_grid_band", N, origin3, voxel_size,
                                                 grid_mode, hand, obj, self._box_tau)
            return vh, vo, {"kind": "band", "args": args, "rec": rec, "tau": self._box_tau, "epoch": self._recalibrations,
                            "host": self._record_to_host(rec)}
        self._band_skip = False
        vh, vo, bbox2 = self.decode_grid(N, origin3, voxel_size, grid_mode, want_bbox=self.math == "f16x3", hand=hand, obj=obj)
        ticket = {"kind": "exact", "args": args, "rec": bbox2, "epoch": self._recalibrations, "host": self._record_to_host(bbox2)}
        if band and self.math == "f16x3" and (fine_due or not self._fine_valid(N)) and not self._fine_compare_in_flight:
            # the zoom lattice is compared as a whole: this ordinary sweep (whose volumes the caller gets) against a plain one-plane
            # sweep of the same lattice, enqueued here while the decoder is bound to the sample and evaluated in fine_needs_repeat
            # (ONE comparison in flight: with the software pipeline the next samples' fine_begin run before this one is judged -
            # they used to enqueue a comparison each, 4 x N^3 fp32 volumes apiece: ADVICE r05)
            ticket["compare"] = self._plain_one_plane(args) + (vh, vo)
            ticket["periodic"] = fine_due
            self._fine_compare_in_flight = True
        return vh, vo, ticket

    def fine_needs_repeat(self, ticket):
        """True when the fine pass has to be repeated (the decoder must be bound to its sample again first): its range or
        error guards fired and the decoder has been re-calibrated / switched so that the repeat is trustworthy."""
        if ticket is None:
            return False
        if ticket["kind"] == "band":
            import logging
            r = self._record_of(ticket)
            if ticket["epoch"] != self._recalibrations:
                # not judged: its error says nothing about the current scales.  (A near-level list that overflowed is a property of
                # the SAMPLE - refine_tau is absolute - and is honoured: the repeat runs on the fp32 chain.)
                ok, bad, near_over, reason = False, 0, self._range_words(r)[1], "launched under activation scales that have been re-calibrated since"
            else:
                ok, bad, near_over, reason = self._judge(r, ticket["tau"], self.band_stats, (33, 34), BAND_CAP, points=int(ticket["args"][0]) ** 3,
                                                         which="fine")
            if ok:
                self.band_stats["band"] += 1
                self._band_failures = 0
                return False
            self.band_stats["fallback"] += 1
            self.events["repeated_sweeps"] += 1
            if bad:
                self._recover(bad)
            else:
                logging.warning("narrow-band fine sweep not accepted (%s): repeated as an ordinary sweep", reason)
                if near_over or int(r[38]):
                    self._force_f32_once = True
                elif ticket["epoch"] == self._recalibrations:
                    self._band_failures += 1
                if self._band_failures >= 3:
                    logging.warning("narrow-band fine sweep switched off for this decoder")
                    self.fine_mode = "exact"
                    self.events["modes_switched_off"].append("fine: three refusals in a row (%s)" % reason)
            self._band_skip = True
            return True
        if ticket.get("compare") is not None:
            self._fine_compare_in_flight = False        # judged now, whatever comes of it
        if ticket["rec"] is not None and self.fall_back_if_overflowed(self._record_of(ticket), ticket.get("epoch")):
            self.events["repeated_sweeps"] += 1
            return True
        self.band_stats["exact"] += 1
        if ticket.get("compare") is not None and ticket["epoch"] == self._recalibrations:
            rec, sh, so, vh, vo = ticket.pop("compare")
            self._calibrate_fine(ticket["args"], (sh, so), (vh, vo), ticket.get("periodic", False))
        return False

    def _plain_one_plane(self, args):
        """One plain one-plane sweep of a lattice for a whole-lattice comparison: a tiny allowance (next to no candidates are
        re-evaluated) and no audit (it would overwrite its picks with exact values).  Returns (record, sdf_hand, sdf_obj)."""
        audit = self.audit_voxels
        self.set_audit(0)
        out = self._box_launch(*args, 1e-7)
        self._audit_restarts = getattr(self, "_audit_restarts", 0) + 1
        self.set_audit(audit, seed=0x5DF5A11D00000000 + 0x9E3779B9 * self._audit_restarts)
        return out

    LADDER = [1 << k for k in range(0, 18)]

    def _compare_lattices(self, fast_vols, exact_vols, N):
        """All 2 x N^3 one-plane values against the ordinary sweep's: lattice maximum of |one-plane - split-half|, its rms (sigma), the
        correlation of the signed error between x-neighbours (noise-like = near 0), and the TAIL RATIO - lattice maximum / maximum
        of a uniform sample - for a ladder of sample sizes 1 .. 2^17."""
        err, sumsq, count, corr = 0.0, 0.0, 0, 0.0
        ladder = self.LADDER
        sample_max = np.zeros(len(ladder))
        gen = torch.Generator(device=self.device)
        gen.manual_seed(12345 + getattr(self, "_audit_restarts", 0))
        for fast, exact in zip(fast_vols, exact_vols):
            if fast is not None and exact is not None:
                signed = (fast - exact).reshape(-1)
                diff = signed.abs()
                err = max(err, float(diff.max().item()))
                sumsq += float((signed.double() ** 2).sum().item())
                count += diff.numel()
                e3 = signed.reshape(N, N, N)
                a, b = e3[:, :, :-1].reshape(-1).double(), e3[:, :, 1:].reshape(-1).double()
                va, vb = a - a.mean(), b - b.mean()
                den = float((va.norm() * vb.norm()).item())
                corr = max(corr, abs(float((va * vb).sum().item()) / den) if den > 0 else 0.0)
                pick = torch.randint(0, diff.numel(), (ladder[-1],), device=self.device, generator=gen)
                running = torch.cummax(diff[pick], 0).values
                sample_max = np.maximum(sample_max, running[torch.tensor([n - 1 for n in ladder], device=self.device)].cpu().numpy())
        sigma = float(np.sqrt(sumsq / max(count, 1)))
        tails = {n: float(np.clip(err / m, 1.0, 1e3)) if m > 0 and np.isfinite(m) else 2.0 for n, m in zip(ladder, sample_max)}
        return {"err": err, "sigma": sigma, "corr": corr, "tails": tails}

    def _apply_comparison(self, m, N, which):
        """Take a whole-lattice comparison (_compare_lattices) into the allowance, the tail ladder and the certificate.  `which` =
        "coarse" | "fine".  Returns False when it switched one-plane sweeps off."""
        import logging
        err, sigma, corr = m["err"], m["sigma"], m["corr"]
        c = self.cert
        fresh = c["calibrations"] + c["fine_calibrations"] == 0 or not self._err_window
        if fresh:
            self._err_window.clear()
        if not np.isfinite(err) or TAU_FACTOR * err > 0.05:
            logging.warning("one-plane sweeps: error %.3g on the %s lattice too large, switched off for this decoder", err, which)
            self.events["modes_switched_off"].append("both: %s-lattice error %.3g too large" % (which, err))
            self.coarse_mode = self.fine_mode = "exact"
            self._box_tau = None
            return False
        uniform, _ = self.audit_sizes(self.audit_voxels, N ** 3)
        if which == "coarse":
            self._box_epoch, self._cal_points, self._tail_by_n = self._recalibrations, N ** 3, m["tails"]
            tail = self._tail = self._tail_for(N ** 3, "coarse") if uniform else 1.0
            c["calibrations"] += 1
            c["tail_ratio"] = tail
            c["tail_ratio_max"] = tail if c["tail_ratio_max"] is None else max(c["tail_ratio_max"], tail)
            c["lattice_max_error"], c["lattice_sigma"] = err, sigma
            c["lattice_max_over_sigma"] = err / sigma if sigma > 0 else None
            c["neighbour_correlation"] = corr
        else:
            self._fine_epoch, self._fine_cal_points, self._tail_by_n_fine = self._recalibrations, N ** 3, m["tails"]
            tail = self._tail_fine = self._tail_for(N ** 3, "fine") if uniform else 1.0
            c["fine_calibrations"] += 1
            c["fine_tail_ratio"] = tail
            c["fine_tail_ratio_max"] = tail if c["fine_tail_ratio_max"] is None else max(c["fine_tail_ratio_max"], tail)
            c["fine_lattice_max_error"], c["fine_lattice_sigma"] = err, sigma
            c["fine_max_over_sigma"] = err / sigma if sigma > 0 else None
            c["fine_neighbour_correlation"] = corr
        if tail > TAIL_MAX:
            logging.warning("one-plane sweeps: the %s lattice's maximum %.3g is %.1f x what a sample of %d sees - rare large errors a "
                            "sample cannot certify; switched off for this decoder", which, err, tail, uniform)
            self.events["modes_switched_off"].append("both: %s-lattice tail ratio %.2f > %.1f" % (which, tail, TAIL_MAX))
            self.coarse_mode = self.fine_mode = "exact"
            self._box_tau = None
            return False
        self._err_window.append(max(err, 2.5e-7))
        self._box_tau = self._tau_current()
        logging.info("one-plane sweeps, %s lattice: error max %.3g sigma %.3g (max / sigma %.1f, x-neighbour correlation %.3f), sample of "
                     "%d per head: tail ratio %.2f, allowance %.3g", which, err, sigma, err / max(sigma, 1e-30), corr, uniform, tail, self._box_tau)
        return True

    def _calibrate_box(self, args, exact_vols):
        """Error allowance of the one-plane sweep for this decoder and scale set, from one whole COARSE sweep run both ways: every one
        of the 2 x N^3 one-plane values against the ordinary sweep's (_compare_lattices).  Runs on the first coarse pass of a decoder
        and scale epoch, on the coarse pass after a sweep refused for its error, and - alternating with the zoom lattice
        (_calibrate_fine) - every RECAL_EVERY samples (tail regularity is re-measured, not assumed)."""
        N = int(args[0])
        rec, sh, so = self._plain_one_plane(args)
        self._coarse_since_cal = 0
        self._apply_comparison(self._compare_lattices((sh, so), exact_vols, N), N, "coarse")

    def _calibrate_fine(self, args, fast_vols, exact_vols, periodic=False):
        """The same comparison on the ZOOM lattice of a fine pass - the lattice whose signs marching cubes consumes (utils/mesh.py:82-121,
        :354): the pass ran as an ordinary sweep (its volumes went to the caller) with a plain one-plane sweep of the same lattice
        enqueued behind it (fine_begin).  First fine pass of a decoder and scale epoch, the fine pass after a refusal for error, and
        every other periodic re-measurement."""
        N = int(args[0])
        if periodic:
            self._coarse_since_cal = 0
            self._next_recal = "coarse"
        self._apply_comparison(self._compare_lattices(fast_vols, exact_vols, N), N, "fine")

    def certificate(self):
        """What the default sweeps' certificate rests on, as measured so far on this decoder."""
        out = dict(self.cert)
        out.update(tau_factor=TAU_FACTOR, tau_accept=TAU_ACCEPT, recalibrate_every=RECAL_EVERY, tail_max=TAIL_MAX,
                   allowance_now=self._box_tau, audit_voxels=self.audit_voxels)
        return out

    # ---- what a run's sweeps did, for the `sweeps.json` a reconstruction writes next to its meshes (a refused
    # or switched-off mode must be visible after the fact, not only in a log line)
    _ADDITIVE_CERT = ("calibrations", "fine_calibrations", "audited_sweeps", "shell_picks", "uniform_picks", "refusals_for_error")

    def sweep_snapshot(self):
        """Counters of this decoder's sweeps now; hand it to sweep_report() for what happened since."""
        return {"box": dict(self.box_stats), "band": dict(self.band_stats), "repeated": self.events["repeated_sweeps"],
                "one_go": self.events["samples_in_one_go"],
                "switched": len(self.events["modes_switched_off"]), "cert": {k: self.cert[k] for k in self._ADDITIVE_CERT},
                "math": self.math, "modes": (self.coarse_mode, self.fine_mode)}

    @property
    def split_half_kernel(self):
        """Name of the split-half kernel an ordinary sweep of this decoder launches: the W form (v_mfma_f32_16x16x32_f16, round 6)
        for a SeparateDecoder with affine point features unless the 32-wide instruction is selected (asdf_set_mfma_shape /
        ASDF_K1H_SHAPE=32), the 32x32x16 form for CombinedDecoder and NeRF-encoded decoders."""
        wide = not self.combined and not self.nerf_features and int(self._L.asdf_get_mfma_shape()) == 16
        return "sdf_mlp_f16w_kernel" if wide else "sdf_mlp_f16_kernel"

    def sweep_report(self, since=None):
        """Which sweeps produced the volumes behind a run's meshes: one-plane / ordinary / refused-and-repeated counts per pass,
        audits, whole-lattice comparisons, the margins of the statistical certificate, and every mode switch."""
        z = since or {"box": self._new_stats("box"), "band": self._new_stats("band"), "repeated": 0, "switched": 0, "one_go": 0,
                      "cert": {k: 0 for k in self._ADDITIVE_CERT}, "math": self.math, "modes": (self.coarse_mode, self.fine_mode)}
        d = lambda now, then, k: int(now[k]) - int(then[k])
        c = self.cert
        return {
            "evaluator": "hip kernels (libalignsdf_hip.so)",
            "arithmetic": {"at_start": z["math"], "now": self.math,
                           "fell_back_to_fp32_chain": z["math"] == "f16x3" and self.math == "f32"},
            "coarse_pass": {"mode_at_start": z["modes"][0], "mode_now": self.coarse_mode,
                            "one_plane_box_sweeps_accepted": d(self.box_stats, z["box"], "box"),
                            "ordinary_sweeps": d(self.box_stats, z["box"], "exact"), "refused_and_repeated": d(self.box_stats, z["box"], "fallback"),
                            "audit_evaluations": d(self.box_stats, z["box"], "audit_evals"), "audit_sign_flips": d(self.box_stats, z["box"], "audit_flips")},
            "fine_pass": {"mode_at_start": z["modes"][1], "mode_now": self.fine_mode,
                          "one_plane_band_sweeps_accepted": d(self.band_stats, z["band"], "band"),
                          "ordinary_sweeps": d(self.band_stats, z["band"], "exact"), "refused_and_repeated": d(self.band_stats, z["band"], "fallback"),
                          "audit_evaluations": d(self.band_stats, z["band"], "audit_evals"), "audit_sign_flips": d(self.band_stats, z["band"], "audit_flips")},
            "sweeps_audited": c["audited_sweeps"] - z["cert"]["audited_sweeps"],
            "sweeps_refused": d(self.box_stats, z["box"], "fallback") + d(self.band_stats, z["band"], "fallback"),
            "sweeps_repeated": self.events["repeated_sweeps"] - z["repeated"],
            "samples_enqueued_in_one_go": self.events["samples_in_one_go"] - z.get("one_go", 0),
            "refusals_for_error": c["refusals_for_error"] - z["cert"]["refusals_for_error"],
            "whole_lattice_comparisons": {"coarse_lattice": c["calibrations"] - z["cert"]["calibrations"],
                                          "zoom_lattice": c["fine_calibrations"] - z["cert"]["fine_calibrations"]},
            "modes_switched_off": list(self.events["modes_switched_off"][z["switched"]:]),
            # margins of the certificate over this decoder's lifetime (minima / maxima, not per run)
            "min_tau_over_sigma": c["min_tau_over_sigma"], "min_tau_over_estimate": c["min_margin_tau_over_estimate"],
            "tail_ratio_max": {"coarse_lattice": c["tail_ratio_max"], "zoom_lattice": c["fine_tail_ratio_max"]},
            "lattice_max_error": {"coarse_lattice": c["lattice_max_error"], "zoom_lattice": c["fine_lattice_max_error"]},
            "allowance_now": self._box_tau, "audit_voxels_per_sweep_and_head": self.audit_voxels,
            "rule": {"tau_factor": TAU_FACTOR, "tau_accept": TAU_ACCEPT, "recalibrate_every": RECAL_EVERY, "tail_max": TAIL_MAX},
        }

    def decode_points(self, xyz):
        """Both heads on explicit normalised points [M,3]. Returns (hand [M], obj [M]) device tensors."""
        xyz = xyz.detach().to(device=self.device, dtype=torch.float32).contiguous()
        M = xyz.shape[0]
        hand = torch.empty(M, dtype=torch.float32, device=self.device)
        obj = torch.empty(M, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _native.check(self._L.asdf_decode_points(self._h, xyz.data_ptr(), M, hand.data_ptr(), obj.data_ptr(),
                                                     self._stream()), "asdf_decode_points")
        return hand, obj

    def classify_points(self, xyz, want_sdf=True):
        """decode_points plus the part classifier: (hand [M], obj [M], scores [M, num_class], labels [M] int64) - the
        scores are `predicted_class` of the reference forward, labels their argmax (utils/mesh.py:156-157)."""
        if not self.num_class:
            raise ValueError("this decoder has no classifier_head (specs['ClassifierBranch'] is off)")
        xyz = xyz.detach().to(device=self.device, dtype=torch.float32).contiguous()
        M = xyz.shape[0]
        hand = torch.empty(M, dtype=torch.float32, device=self.device) if want_sdf else None
        obj = torch.empty(M, dtype=torch.float32, device=self.device) if want_sdf else None
        scores = torch.empty((M, self.num_class), dtype=torch.float32, device=self.device)
        labels = torch.empty(M, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            _native.check(self._L.asdf_decode_points_cls(
                self._h, xyz.data_ptr(), M, hand.data_ptr() if want_sdf else None, obj.data_ptr() if want_sdf else None,
                scores.data_ptr(), labels.data_ptr(), self._stream()), "asdf_decode_points_cls")
        return hand, obj, scores, labels.long()
