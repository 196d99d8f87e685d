"""The stock-module path of the decoder operator: decoder variants the HIP kernels do not cover are evaluated
by calling the nn.Module itself on PyTorch-ROCm device tensors, exactly like the reference's chunk loop does
(utils/mesh.py:46-63: slice, embed, `decode_sdf_multi_output`, write back) - only the lattice coordinates come from the
native library (the same device function the HIP kernels use, so the lattice is the reference's bit for bit), and the
negative-voxel box / marching cubes behind it are the native kernels as always.

Variants served here: `use_tanh`, the LayerNorm form (`weight_norm` false), `xyz_in_all` (networks/model.py:118-119,131-132,
166-176,314-315), a pose-aligned model evaluated without `mano_results` (the NeRF branch of utils/mesh.py:53-55),
`PixelAlign` (utils/utils.py:536-566), and any module whose parameters are not SeparateDecoder / CombinedDecoder shaped.
It is a GPU path like the rest of the package: CPU tensors raise.
"""
import copy
import itertools
import ctypes
import logging
import weakref

import numpy as np
import torch

from . import _native

CHUNK = 2 ** 18          # max_batch of reconstruct.py:93


def kinematic_embedding(xyz, mano_results, point_feat_size, scale_factor, obj_results, encode_style):
    """utils.utils.kinematic_embedding (utils/utils.py:376-430) on device tensors, op for op (batch of one sample)."""
    M = xyz.shape[0]
    wrist = xyz * 2 / scale_factor
    hand = obj = None
    if encode_style in ("hand", "both"):
        mano_xyz = wrist + mano_results["rot_center"].reshape(1, 3).to(xyz)
        homo = torch.cat([mano_xyz, torch.ones(M, 1, device=xyz.device)], 1)
        inv_g = torch.linalg.inv(mano_results["global_trans"].reshape(16, 4, 4).to(xyz))
        inv_pts = torch.matmul(inv_g.unsqueeze(0), homo.reshape(M, 1, 4, 1)).squeeze(-1)
        inv_xyz = inv_pts[:, :, :3] / inv_pts[:, :, 3:4]
        if (point_feat_size == 6 and encode_style == "hand") or (point_feat_size == 9 and encode_style == "both"):
            inv_xyz = inv_xyz[:, :1, :]
        hand = torch.cat([mano_xyz.unsqueeze(1), inv_xyz], 1).reshape(M, -1) * scale_factor / 2
    if encode_style in ("obj", "both"):
        homo_w = torch.cat([wrist, torch.ones(M, 1, device=xyz.device)], 1)
        inv_o = torch.linalg.inv(obj_results["obj_trans"].reshape(4, 4).to(xyz))
        o = torch.matmul(inv_o, homo_w.t()).t()
        obj = (o[:, :3] / o[:, 3:4]) * scale_factor / 2
    if encode_style == "hand":
        return hand
    if encode_style == "obj":
        return torch.cat([xyz, obj], 1)
    return torch.cat([hand, obj], 1)


def nerf_embedding(xyz, multires):
    """get_nerf_embedder(multires) (utils/utils.py:433-463,521-533): [x, sin(2^k x), cos(2^k x)]."""
    outs = [xyz]
    for freq in 2.0 ** torch.linspace(0.0, multires - 1, steps=multires):
        outs += [torch.sin(xyz * freq), torch.cos(xyz * freq)]
    return torch.cat(outs, -1)


def pixel_alignment(img_feat, xyz, cam_intr, mano_results, image_size, scale_factor):
    """utils.utils.pixel_alignment (utils/utils.py:536-558): per-point bicubic samples of the image feature map at the
    projection of the point (camera space = wrist space + root joint); points that project outside the image take the
    feature map's mean."""
    pred_root = mano_results["joints"][:, [0]].to(xyz)
    x = xyz.reshape((img_feat.shape[0], -1, 3))
    xyz_cam = (x * 2 / scale_factor) + pred_root
    B, P = img_feat.shape[0], x.shape[1]
    homo = torch.cat([xyz_cam, torch.ones([B, P, 1], device=xyz.device)], 2)
    xy_img = torch.bmm(cam_intr.to(xyz), homo.transpose(1, 2)).transpose(1, 2)
    xy_img = (xy_img[:, :, :2] / xy_img[:, :, [2]]).unsqueeze(2)
    uv = xy_img / image_size * 2 - 1
    feat = torch.nn.functional.grid_sample(img_feat, uv, align_corners=True, mode="bicubic")[:, :, :, 0].transpose(1, 2)
    uv = uv.squeeze().reshape((-1, 2))
    inside = (uv[:, 0] >= -1.0) & (uv[:, 0] <= 1.0) & (uv[:, 1] >= -1.0) & (uv[:, 1] <= 1.0)
    outside = (~inside).reshape((B, P, -1))
    feat[torch.where(outside)[:2]] = img_feat.mean(3).mean(2)[torch.where(outside)[:1]]
    return feat.reshape((B * P, -1))


def needs_module_path(decoder, specs=None, mano_results=None):
    """Why (a string) this decoder / sample cannot run on the HIP kernels, or None when it can."""
    if specs is not None and specs.get("PixelAlign", False):
        return "PixelAlign: the latent is a per-point sample of an image feature map"
    if isinstance(decoder, torch.nn.Module):
        if getattr(decoder, "use_tanh", False):
            return "use_tanh"
        if getattr(decoder, "xyz_in_all", False):
            return "xyz_in_all"
        # shapes: NetworkSpecs other than the shipped dims [512] * 4 / latent_in [2] / LatentSize 256 are legal
        # (networks/model.py:192-282) and run on the module
        from .hip_decoder import unsupported_reason
        sd = decoder.state_dict()
        if not hasattr(decoder, "point_feat_size") or not hasattr(decoder, "encode_style"):
            return "not a SeparateDecoder / CombinedDecoder shaped module"
        why = unsupported_reason(sd, getattr(decoder, "latent_size", (specs or {}).get("LatentSize", 256)), decoder.point_feat_size,
                                 decoder.encode_style)
        if why is not None:
            return why
    if specs is not None and specs["PointFeatSize"] > 3 and specs["EncodeStyle"] != "nerf" and mano_results is None:
        return "pose-aligned model without mano_results: NeRF branch of utils/mesh.py:53-55"
    return None


class TorchModuleDecoder:
    """The interface of HipSdfDecoder (set_sample / decode_grid / decode_points / classify_points) on a plain module call.

    The caller's module is neither moved nor switched: it is held through a weak reference, called in eval mode with its
    training flag restored afterwards, and - when its parameters live on another device (the reference keeps checkpoints on the
    CPU until `.cuda()`) - evaluated through a device COPY that is refreshed whenever a parameter tensor is replaced or written
    (the reference reconstructs from inside its training loop, train.py:668).  `specs` None = the legacy single-output
    contract of deep_sdf.utils.decode_sdf (deep_sdf/utils.py:64-75): `module(cat(latent, xyz))` or `module(xyz)` when there is
    no latent, returning [M, 1] (the first element if the module returns a tuple)."""

    math = "torch"
    combined = False
    nerf_features = False
    event_log = None
    coarse_mode = fine_mode = "exact"

    def __init__(self, module, specs, reason, device=None):
        if not isinstance(module, torch.nn.Module):
            raise TypeError("the module path needs the nn.Module itself (%s)" % reason)
        if not torch.cuda.is_available() or _native.lib().asdf_device_count() < 1:
            raise _native.NativeError(-4, "no gfx950 (MI355X) device visible - there is no CPU path")
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        self._module_ref = weakref.ref(module)
        self._copy = None            # (fingerprint, device copy) when the caller's parameters are not on self.device
        self.specs = specs
        self.reason = reason
        self.num_class = int(getattr(module, "num_class", 0)) if getattr(module, "use_classifier", False) else 0
        self._sample = None
        logging.warning("decoder runs on the PyTorch-ROCm module path (%s); the HIP kernels do not cover this variant", reason)

    @property
    def module(self):
        """The module to call: the caller's own when it lives on this device, otherwise an up-to-date device copy."""
        m = self._module_ref()
        if m is None:
            raise ReferenceError("the decoder module has been garbage-collected")
        params = list(m.parameters())
        if not params or all(p.device == self.device for p in params):
            return m
        # parameters AND buffers (a LayerNorm-free module has none, a BatchNorm-style one does): a change of either refreshes the copy
        fp = tuple((t.data_ptr(), t._version) for t in itertools.chain(params, m.buffers()))
        if self._copy is None or self._copy[0] != fp:
            # old-style weight_norm leaves the effective weight of its last forward on the module as a NON-leaf tensor, which
            # deepcopy refuses.  The caller's module is not touched: the copy is made with a memo that hands deepcopy a
            # detached clone for every such tensor - the forward pre-hook recomputes it on every call, so the stand-in loses nothing.
            # Cost: one deepcopy + upload per CHANGE of the caller's CPU parameters - inside a training loop that is every step;
            # keep the module on the device there.
            memo = {}
            for sub in m.modules():
                for val in vars(sub).values():
                    if torch.is_tensor(val) and not val.is_leaf:
                        memo[id(val)] = val.detach().clone()
            self._copy = (fp, copy.deepcopy(m, memo).to(self.device).eval())
        return self._copy[1]

    def set_sample(self, latent_vec, mano_results=None, obj_results=None, cam_intr=None):
        # (codes that arrive on the host - the code sources leave them there - go through the pinned ring + side stream: a plain
        # .to(device) of pageable memory waits for everything queued on the stream)
        def up(t):
            if t is None:
                return None
            t = t.detach()
            if t.device.type == "cpu" and self.device.type == "cuda":
                if getattr(self, "_uploader", None) is None:
                    from .reconstruct import CodeUploader
                    self._uploader = CodeUploader(self.device)
                return self._uploader(t.to(torch.float32).numpy())
            return t.to(self.device)
        dev = lambda d: None if d is None else {k: up(v) for k, v in d.items()}
        lat = up(latent_vec)
        self._sample = (None if lat is None else lat.to(torch.float32), dev(mano_results), dev(obj_results), up(cam_intr))

    def fall_back_if_overflowed(self, bbox_host, epoch=None):
        return False

    def range_violations(self, clear=True):
        return 0

    def close(self):
        self._sample = None
        self._copy = None

    # -- the reference's chunk loop body (utils/mesh.py:47-56 + utils/utils.py:561-572; deep_sdf/utils.py:64-75 without specs) ---
    def _decode_chunk(self, module, xyz):
        latent, mano, obj, cam = self._sample
        specs = self.specs
        if specs is None:
            out = module(xyz if latent is None else torch.cat([latent.reshape(1, -1).expand(xyz.shape[0], -1), xyz], 1))
            return (out[0], out[1] if len(out) > 1 and torch.is_tensor(out[1]) and out[1].dim() == 2 else None, None) \
                if isinstance(out, (tuple, list)) else (out, None, None)
        q = xyz
        if specs["PointFeatSize"] > 3:
            if mano is not None and specs["EncodeStyle"] != "nerf":
                q = kinematic_embedding(xyz, mano, specs["PointFeatSize"], specs["SdfScaleFactor"], obj, specs["EncodeStyle"])
            else:
                q = nerf_embedding(xyz, (specs["PointFeatSize"] - 3) // 6)
        if specs.get("PixelAlign", False):
            lat = pixel_alignment(latent, q[:, :3], cam, mano, specs["ImageSize"][0], specs["SdfScaleFactor"])
        else:
            lat = latent.reshape(1, -1).expand(q.shape[0], -1)
        return module(torch.cat([lat, q], 1))

    def _decode(self, xyz, want_scores=False):
        if self._sample is None:
            raise ValueError("no sample bound: call set_sample first")
        M = xyz.shape[0]
        hand = torch.empty(M, dtype=torch.float32, device=self.device)
        obj = torch.empty(M, dtype=torch.float32, device=self.device)
        has_obj = True
        scores = torch.empty((M, self.num_class), dtype=torch.float32, device=self.device) if want_scores else None
        module = self.module
        was_training = module.training
        module.eval()
        try:
            with torch.no_grad():
                for head in range(0, M, CHUNK):
                    h, o, c = self._decode_chunk(module, xyz[head:head + CHUNK])
                    hand[head:head + CHUNK] = h.reshape(-1)
                    if o is None:
                        has_obj = False
                    else:
                        obj[head:head + CHUNK] = o.reshape(-1)
                    if want_scores:
                        scores[head:head + CHUNK] = c
        finally:
            module.train(was_training)
        return hand, (obj if has_obj else None), scores

    def coarse_begin(self, N, origin3, voxel_size, grid_mode=_native.GRID_REFERENCE, hand=True, obj=True):
        """The coarse pass of the two-pass flow (same interface as HipSdfDecoder.coarse_begin; always an ordinary sweep)."""
        h, o, bbox = self.decode_grid(N, origin3, voxel_size, grid_mode, hand=hand, obj=obj)
        return {"rec": bbox, "keep": (h, o)}

    def coarse_finish(self, ticket):
        return ticket["rec"].cpu().numpy()

    def fine_begin(self, N, origin3, voxel_size, grid_mode=_native.GRID_REFERENCE, hand=True, obj=True, mc_only=False):
        """The fine pass (same interface as HipSdfDecoder.fine_begin; always an ordinary sweep, nothing to guard)."""
        h, o, _ = self.decode_grid(N, origin3, voxel_size, grid_mode, want_bbox=False, hand=hand, obj=obj)
        return h, o, None

    def fine_needs_repeat(self, ticket):
        return False

    def sweep_snapshot(self):
        return None

    def sweep_report(self, since=None):
        """The module path runs ordinary fp32 sweeps only (same interface as HipSdfDecoder.sweep_report)."""
        return {"evaluator": "module on PyTorch-ROCm (%s)" % self.reason, "arithmetic": {"at_start": "f32 (torch)", "now": "f32 (torch)",
                "fell_back_to_fp32_chain": False}, "sweeps_audited": 0, "sweeps_refused": 0, "sweeps_repeated": 0, "modes_switched_off": []}

    def decode_points(self, xyz):
        h, o, _ = self._decode(xyz.detach().to(self.device, torch.float32).contiguous())
        return h, o

    def classify_points(self, xyz, want_sdf=True):
        if not self.num_class:
            raise ValueError("this decoder has no classifier_head (specs['ClassifierBranch'] is off)")
        h, o, s = self._decode(xyz.detach().to(self.device, torch.float32).contiguous(), want_scores=True)
        return (h if want_sdf else None), (o if want_sdf else None), s, s.argmax(dim=1)

    def decode_grid(self, N, origin3, voxel_size, grid_mode=_native.GRID_REFERENCE, want_bbox=True, hand=True, obj=True, check_range=None):
        """Both heads on the N^3 lattice (the module always evaluates both, like the reference); lattice coordinates from
        asdf_debug_grid_coords, i.e. the device function of the HIP kernels."""
        L = _native.lib()
        coords = torch.empty((N ** 3, 3), dtype=torch.float32, device=self.device)
        org = (ctypes.c_float * 3)(*[float(np.float32(o)) for o in origin3])
        with torch.cuda.device(self.device):
            stream = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            _native.check(L.asdf_debug_grid_coords(int(N), org, ctypes.c_float(float(np.float32(voxel_size))), int(grid_mode), 0, N ** 3,
                                                   coords.data_ptr(), stream), "asdf_debug_grid_coords")
            vh, vo, _ = self._decode(coords)
            vh = vh.reshape(N, N, N)
            vo = vo.reshape(N, N, N) if vo is not None else None
            obj = obj and vo is not None
            bbox = None
            if want_bbox:
                bbox = torch.empty(16, dtype=torch.int32, device=self.device)
                tmp = torch.empty(16, dtype=torch.int32, device=self.device)
                for k, (on, vol) in enumerate(((hand, vh), (obj, vo))):
                    if on:
                        _native.check(L.asdf_neg_bbox(vol.data_ptr(), N, N, N, tmp.data_ptr(), stream), "asdf_neg_bbox")
                        bbox[8 * k:8 * k + 8] = tmp[:8]
                    else:
                        bbox[8 * k:8 * k + 8] = torch.tensor([0x7fffffff] * 3 + [-1] * 3 + [0, 0], dtype=torch.int32, device=self.device)
                bbox[7] = 0
                bbox[15] = 0
        return (vh if hand else None), (vo if obj else None), bbox
