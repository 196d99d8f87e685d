"""Drop-in counterparts of the hot-path helpers of the reference's utils/utils.py.

`decode_sdf_multi_output` keeps the reference signature (utils/utils.py:561) but evaluates the decoder
through the HIP path (or, for the variants the kernels do not cover, through the module on PyTorch-ROCm).  Because the reference feeds it already-embedded queries, it accepts either raw
normalised xyz [M,3] (PointFeatSize 3, or with `raw_xyz=True` plus pose dicts so the affine embedding is
folded into the kernel) - embedded [M,pf] queries of a pose-aligned model cannot be un-embedded and raise.
"""
import weakref

import torch

from ..hip_decoder import HipSdfDecoder, kinematic_affine

_cache = weakref.WeakKeyDictionary()


def _param_fingerprint(module):
    """Changes whenever a parameter tensor is replaced or written in place (optimizer step, load_state_dict)."""
    return tuple((p.data_ptr(), p._version) for p in module.parameters())


def hip_decoder_for(decoder, device=None):
    """Packed HIP decoder of an nn.Module, built once per (module, device) and cached; re-packed when the module's
    parameters have changed since (the reference reconstructs from inside the training loop, train.py:668)."""
    if isinstance(decoder, HipSdfDecoder):
        return decoder
    dev = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
    per_mod = _cache.setdefault(decoder, {})
    fp = _param_fingerprint(decoder)
    hit = per_mod.get(str(dev))
    if hit is None or hit[0] != fp:
        if hit is not None:
            hit[1].close()
        per_mod[str(dev)] = (fp, HipSdfDecoder(decoder, device=dev))
    return per_mod[str(dev)][1]


def decoder_for(decoder, specs=None, mano_results=None, device=None):
    """The evaluator of this decoder for this kind of sample: the packed HIP decoder wherever the kernels cover the variant,
    otherwise the module itself on PyTorch-ROCm (alignsdf_amd.torch_decoder: use_tanh / LayerNorm / xyz_in_all / PixelAlign /
    pose-aligned model without mano_results) - the stock-module path of SURVEY 8 b2."""
    from ..torch_decoder import TorchModuleDecoder, needs_module_path
    if isinstance(decoder, (HipSdfDecoder, TorchModuleDecoder)):
        if isinstance(decoder, HipSdfDecoder):
            why = needs_module_path(None, specs, mano_results)
            if why:
                raise NotImplementedError("%s - pass the nn.Module so that it can be called" % why)
        return decoder
    why = needs_module_path(decoder, specs, mano_results)
    if why is None:
        return hip_decoder_for(decoder, device)
    dev = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
    per_mod = _cache.setdefault(decoder, {})
    key = "torch:%s:%s" % (dev, why)
    if key not in per_mod:
        per_mod[key] = (None, TorchModuleDecoder(decoder, specs, why, dev))
    per_mod[key][1].specs = specs
    return per_mod[key][1]


def bind_sample(dec, specs, latent_vec, mano_results, obj_results, cam_intr=None):
    """Bind one sample's codes to an evaluator returned by decoder_for."""
    if isinstance(dec, HipSdfDecoder):
        dec.set_sample(latent_vec, sample_embedding(specs, mano_results, obj_results, dec.combined))
    else:
        dec.set_sample(latent_vec, mano_results, obj_results, cam_intr)


def sample_embedding(specs, mano_results, obj_results, combined=False):
    """Per-head affine embeddings for this sample, or None for plain xyz (utils/mesh.py:49-55)."""
    if specs["PointFeatSize"] <= 3:
        return None
    if mano_results is not None and specs["EncodeStyle"] != "nerf":
        return kinematic_affine(specs["PointFeatSize"], specs["EncodeStyle"], specs["SdfScaleFactor"], mano_results,
                                obj_results, combined)
    if specs["EncodeStyle"] == "nerf":
        return None      # NeRF positional encoding: computed inside the kernel (decoder packed with FEATURES_NERF)
    raise NotImplementedError("a pose-aligned decoder (EncodeStyle %r) evaluated without mano_results falls back to the NeRF "
                              "encoding in the reference (utils/mesh.py:53-55); that combination is not supported" % specs["EncodeStyle"])


def kinematic_embedding(xyz, mano_results, num_points_per_scene, point_feat_size, scale_factor, obj_results, encode_style):
    """Pose-aligned point features [M, pf] (utils/utils.py:376-430) via their affine form, on xyz's device.
    Host-side helper for callers that need the features themselves; the mesh path folds the same affine
    map into the decoder instead of materialising them."""
    Eh, Eo = kinematic_affine(point_feat_size, encode_style, scale_factor, mano_results, obj_results)
    x = xyz.reshape(-1, 3)
    th = torch.as_tensor(Eh, dtype=torch.float32, device=x.device)
    to = torch.as_tensor(Eo, dtype=torch.float32, device=x.device)
    hand = x @ th[:, :3].t() + th[:, 3]
    obj = x @ to[:, :3].t() + to[:, 3]
    if encode_style == "hand":
        return hand
    if encode_style == "obj":
        return obj
    return torch.cat([hand, obj[:, 3:]], 1)


def decode_sdf_multi_output(decoder, latent_vector, queries, mano_results, cam_intr, specs, obj_results=None):
    """(sdf_hand [M,1], sdf_obj [M,1], predicted_class) for normalised xyz queries [M,3]."""
    if queries.shape[1] != 3:
        raise NotImplementedError("pass raw normalised xyz [M,3]; the pose embedding is folded into the decoder")
    hip = decoder_for(decoder, specs, mano_results)
    bind_sample(hip, specs, latent_vector, mano_results, obj_results, cam_intr)
    if hip.num_class:
        h, o, scores, _ = hip.classify_points(queries)
        return h.unsqueeze(1), o.unsqueeze(1), scores
    h, o = hip.decode_points(queries)
    return h.unsqueeze(1), o.unsqueeze(1), torch.zeros(1, device=h.device)
