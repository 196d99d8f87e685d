"""Drop-in counterparts of the hot-path helpers of the reference's utils/utils.py.

`decode_sdf_multi_output` keeps the reference signature (utils/utils.py:561) but evaluates the decoder
through the HIP path.  Because the reference feeds it already-embedded queries, it accepts either raw
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
    if specs.get("PixelAlign", False):
        raise NotImplementedError("PixelAlign is false in every shipped config and is outside the HIP path")
    if queries.shape[1] != 3:
        raise NotImplementedError("pass raw normalised xyz [M,3]; the pose embedding is folded into the HIP decoder")
    hip = hip_decoder_for(decoder)
    hip.set_sample(latent_vector, sample_embedding(specs, mano_results, obj_results, hip.combined))
    if hip.num_class:
        h, o, scores, _ = hip.classify_points(queries)
        return h.unsqueeze(1), o.unsqueeze(1), scores
    h, o = hip.decode_points(queries)
    return h.unsqueeze(1), o.unsqueeze(1), torch.zeros(1, device=h.device)
