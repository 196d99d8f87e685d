"""CPU restatement of AlignSDF's dense-grid SDF decoding.  TEST INFRASTRUCTURE ONLY.

This module is the parity oracle for the HIP decoder path: only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import it; nothing under alignsdf_amd/ does.  It re-states,
op for op and in fp32 on the CPU (torch), what the reference executes, each function citing the
reference lines it follows (paths relative to zerchen/AlignSDF).

Pinning: tests/golden/ref_*.npz hold outputs of the reference itself (imported in the build
container by tests/golden/make_ref_goldens.py); tests/test_oracle_decoder.py checks this module
against them bit for bit (grid, zoom cube) or to 1e-6 (decoder outputs).
"""
import numpy as np
import torch
import torch.nn.functional as F


def effective_weight(weight_v, weight_g):
    """W = g * v / ||v||_row, exactly what nn.utils.weight_norm's hook computes
    (networks/model.py:249-250; torch `_weight_norm(v, g, dim=0)`)."""
    return torch._weight_norm(torch.as_tensor(weight_v), torch.as_tensor(weight_g), 0)


def effective_head_params(state_dict, head):
    """[(W, b)] * 5 for head 'h' or 'o' from a SeparateDecoder state dict (numpy or torch values)."""
    out = []
    for layer in range(5):
        name = "lin%s%d" % (head, layer)
        if name + ".weight_v" in state_dict:
            w = effective_weight(state_dict[name + ".weight_v"], state_dict[name + ".weight_g"])
        else:
            w = torch.as_tensor(state_dict[name + ".weight"])
        out.append((w.float().contiguous(), torch.as_tensor(state_dict[name + ".bias"]).float().contiguous()))
    return out


def grid_indices(N):
    """Per-axis fp32 index columns (idx0, idx1, idx2) of the N^3 lattice as the reference computes
    them: utils/mesh.py:27-34 (deep_sdf/mesh.py:24-31).  `overall_index.long() / N` is TRUE division
    under every PyTorch >= 1.6, so axis 1 and axis 0 carry fractional (sheared) indices."""
    overall = torch.arange(0, N ** 3, 1, dtype=torch.int64)
    cols = torch.zeros(N ** 3, 3)
    cols[:, 2] = overall % N
    cols[:, 1] = (overall / N) % N
    cols[:, 0] = ((overall / N) / N) % N
    return cols


def grid_coords(N, voxel_size, origin3, integer_mode=False):
    """[N^3, 3] fp32 query coordinates.  `voxel_size` is a python float (pass 1, utils/mesh.py:25) or a
    0-dim fp32 tensor (pass 2, utils/mesh.py:93-96); `origin3` is added per axis (pass 1 adds
    voxel_origin[2], [1], [0] = -1 to axes 0, 1, 2; pass 2 adds new_origin[0..2])."""
    if integer_mode:
        overall = torch.arange(0, N ** 3, 1, dtype=torch.int64)
        cols = torch.stack([(overall // N) // N, (overall // N) % N, overall % N], 1).float()
    else:
        cols = grid_indices(N)
    out = torch.zeros(N ** 3, 3)
    for a in range(3):
        out[:, a] = (cols[:, a] * voxel_size) + origin3[a]
    return out


def grid_coords_window(N, voxel_size, origin3, first, count, integer_mode=False):
    """Rows [first, first + count) of grid_coords(N, ...) without building the N^3 lattice (N = 512 / 1024): the same operations on
    the same index values (utils/mesh.py:27-40)."""
    overall = torch.arange(first, first + count, 1, dtype=torch.int64)
    if integer_mode:
        cols = torch.stack([(overall // N) // N, (overall // N) % N, overall % N], 1).float()
    else:
        cols = torch.zeros(count, 3)
        cols[:, 2] = overall % N
        cols[:, 1] = (overall / N) % N
        cols[:, 0] = ((overall / N) / N) % N
    out = torch.zeros(count, 3)
    for a in range(3):
        out[:, a] = (cols[:, a] * voxel_size) + origin3[a]
    return out


def kinematic_embedding(xyz, mano_results, point_feat_size, scale_factor, obj_results, encode_style):
    """Pose-aligned point features, following utils/utils.py:376-430 step by step (batch of one sample).
    xyz [M,3] normalised -> [M, point_feat_size]."""
    M = xyz.shape[0]
    wrist = (xyz * 2 / scale_factor)                                   # :384
    pieces = {}
    if encode_style in ("hand", "both"):
        mano_xyz = wrist + mano_results["rot_center"].reshape(1, 3)   # :387
        homo = torch.cat([mano_xyz, torch.ones(M, 1)], 1)             # :389-390
        inv_g = torch.linalg.inv(mano_results["global_trans"].reshape(16, 4, 4))   # :393
        # [16,4,4] @ [M,4,1] -> [M,16,4]
        inv_pts = torch.matmul(inv_g.unsqueeze(0), homo.reshape(M, 1, 4, 1)).squeeze(-1)   # :394-395
        inv_xyz = inv_pts[:, :, :3] / inv_pts[:, :, 3:4]              # :396
        if (point_feat_size == 6 and encode_style == "hand") or (point_feat_size == 9 and encode_style == "both"):
            inv_xyz = inv_xyz[:, :1, :]                               # :399-400
        hand = torch.cat([mano_xyz.unsqueeze(1), inv_xyz], 1).reshape(M, -1)   # :403-407
        pieces["hand"] = hand * scale_factor / 2                      # :408
    if encode_style in ("obj", "both"):
        homo_w = torch.cat([wrist, torch.ones(M, 1)], 1)              # :411-412
        inv_o = torch.linalg.inv(obj_results["obj_trans"].reshape(4, 4))   # :414
        o = torch.matmul(inv_o, homo_w.t()).t()                       # :415
        o = o[:, :3] / o[:, 3:4]                                      # :416
        pieces["obj"] = o * scale_factor / 2                          # :417
    if encode_style == "hand":
        return pieces["hand"]
    if encode_style == "obj":
        return torch.cat([xyz, pieces["obj"]], 1)                     # :418,424
    return torch.cat([pieces["hand"], pieces["obj"]], 1)              # :427


def nerf_embedding(xyz, multires):
    """[x, sin(2^k x), cos(2^k x)] (utils/utils.py:433-463,521-533)."""
    outs = [xyz]
    for freq in 2.0 ** torch.linspace(0.0, multires - 1, steps=multires):
        outs += [torch.sin(xyz * freq), torch.cos(xyz * freq)]
    return torch.cat(outs, -1)


def point_features(xyz, specs, mano_results, obj_results):
    """The embedding branch of the chunk loop (utils/mesh.py:49-55)."""
    if specs["PointFeatSize"] > 3:
        if mano_results is not None and specs["EncodeStyle"] != "nerf":
            return kinematic_embedding(xyz, mano_results, specs["PointFeatSize"], specs["SdfScaleFactor"], obj_results,
                                       specs["EncodeStyle"])
        return nerf_embedding(xyz, (specs["PointFeatSize"] - 3) // 6)
    return xyz


def _run_head(params, x, head_input, latent_in=(2,), stop_before_last=False):
    """One MLP head: networks/model.py:304-325 (eval mode: dropout is the identity).
    stop_before_last=True returns the 512-wide input of the last layer instead."""
    for layer, (w, b) in enumerate(params):
        if stop_before_last and layer == len(params) - 1:
            return x
        if layer in latent_in:
            x = torch.cat([x, head_input], 1)          # :310-311
        x = F.linear(x, w, b)                           # :312
        if layer < len(params) - 1:
            x = torch.relu(x)                           # :316-320
    return torch.tanh(x)                                # :324-325


def separate_decoder(hand_params, obj_params, inputs, latent_size, point_feat_size, encode_style):
    """SeparateDecoder.forward (networks/model.py:285-350): inputs [M, latent+pf] -> (hand [M,1], obj [M,1])."""
    if encode_style == "nerf":
        xh = xo = inputs
    elif encode_style == "hand":
        xh, xo = inputs, inputs[:, :latent_size + 3]
    elif encode_style == "obj":
        xh, xo = inputs[:, :latent_size + 3], inputs
    elif encode_style == "both":
        xh = inputs[:, :-3]
        xo = torch.cat([inputs[:, :latent_size + 3], inputs[:, -3:]], 1)
    else:
        raise ValueError(encode_style)
    return _run_head(hand_params, xh, xh)[:, 0:1], _run_head(obj_params, xo, xo)[:, 0:1]


def combined_decoder(params, inputs):
    """CombinedDecoder.forward (networks/model.py:149-188; no classifier, xyz_in_all False): one MLP whose last
    layer has two rows - column 0 is the hand SDF, column 1 the object SDF."""
    x = _run_head(params, inputs, inputs)
    return x[:, 0:1], x[:, 1:2]


def combined_params(state_dict):
    """[(W, b)] * 5 of a CombinedDecoder state dict (keys `lin{k}.*`)."""
    out = []
    for layer in range(5):
        name = "lin%d" % layer
        if name + ".weight_v" in state_dict:
            w = effective_weight(state_dict[name + ".weight_v"], state_dict[name + ".weight_g"])
        else:
            w = torch.as_tensor(state_dict[name + ".weight"])
        out.append((w.float().contiguous(), torch.as_tensor(state_dict[name + ".bias"]).float().contiguous()))
    return out


def decode_sdf_multi_output(hand_params, obj_params, latent, queries, specs):
    """latent expand + cat + decoder (utils/utils.py:561-572, PixelAlign False).  obj_params None => hand_params
    are those of a CombinedDecoder."""
    inputs = torch.cat([latent.expand(queries.shape[0], -1), queries], 1)
    if obj_params is None:
        return combined_decoder(hand_params, inputs)
    return separate_decoder(hand_params, obj_params, inputs, latent.shape[1], specs["PointFeatSize"], specs["EncodeStyle"])


def decode_points(state_dict, latent, xyz, specs, mano_results=None, obj_results=None, max_batch=2 ** 18):
    """Chunked decode of explicit points [M,3] -> (hand [M], obj [M]) fp32 tensors."""
    if "lin0.bias" in state_dict:
        hp, op = combined_params(state_dict), None
    else:
        hp, op = effective_head_params(state_dict, "h"), effective_head_params(state_dict, "o")
    latent = torch.as_tensor(latent).float().reshape(1, -1)
    xyz = torch.as_tensor(xyz).float()
    hand, obj = torch.zeros(xyz.shape[0]), torch.zeros(xyz.shape[0])
    with torch.no_grad():
        for head in range(0, xyz.shape[0], max_batch):
            sub = xyz[head:head + max_batch]
            feats = point_features(sub, specs, mano_results, obj_results)
            h, o = decode_sdf_multi_output(hp, op, latent, feats, specs)
            hand[head:head + max_batch] = h.squeeze(1)
            obj[head:head + max_batch] = o.squeeze(1)
    return hand, obj


def classify_points(state_dict, latent, xyz, specs, mano_results=None, obj_results=None, max_batch=2 ** 18):
    """The label pass over explicit points (utils/mesh.py:146-157): `predicted_class` = classifier_head applied to the
    input of the last layer of the hand MLP (SeparateDecoder, networks/model.py:306-307) or of the single MLP
    (CombinedDecoder, networks/model.py:161-162).  Returns (scores [M, num_class] fp32, labels [M] int64 = argmax)."""
    params = combined_params(state_dict) if "lin0.bias" in state_dict else effective_head_params(state_dict, "h")
    wc = torch.as_tensor(state_dict["classifier_head.weight"]).float()
    bc = torch.as_tensor(state_dict["classifier_head.bias"]).float()
    latent = torch.as_tensor(latent).float().reshape(1, -1)
    xyz = torch.as_tensor(xyz).float()
    scores = torch.zeros(xyz.shape[0], wc.shape[0])
    with torch.no_grad():
        for head in range(0, xyz.shape[0], max_batch):
            sub = xyz[head:head + max_batch]
            feats = point_features(sub, specs, mano_results, obj_results)
            inputs = torch.cat([latent.expand(sub.shape[0], -1), feats], 1)        # utils/utils.py:568-569
            if "lin0.bias" not in state_dict:                                       # the hand head's slice, model.py:288-299
                style, L = specs["EncodeStyle"], latent.shape[1]
                inputs = inputs[:, :L + 3] if style == "obj" else (inputs[:, :-3] if style == "both" else inputs)
            hidden = _run_head(params, inputs, inputs, stop_before_last=True)
            scores[head:head + max_batch] = F.linear(hidden, wc, bc)
    return scores, scores.argmax(dim=1)


def get_higher_res_cube(hand_branch, obj_branch, vol_hand, vol_obj, N, voxel_size):
    """Zoom cube from the negative voxels (utils/mesh.py:198-256).  Returns (new_voxel_size 0-dim fp32,
    new_origin [3] fp32, bbox int64 [2][6] with -1 where a branch has no negative voxel)."""
    lo, hi = [], []
    bbox = -np.ones((2, 6), dtype=np.int64)
    for k, (on, vol) in enumerate(((hand_branch, vol_hand), (obj_branch, vol_obj))):
        if not on:
            continue
        idx = torch.nonzero(vol < 0).float()            # :208 / :224
        if idx.shape[0] == 0:
            lo.append(torch.zeros(3)); hi.append(torch.zeros(3))   # :209-211
        else:
            lo.append(idx.min(0).values); hi.append(idx.max(0).values)
            bbox[k, :3] = lo[-1].numpy(); bbox[k, 3:] = hi[-1].numpy()
    min_index = lo[0] if len(lo) == 1 else torch.min(lo[0], lo[1])   # :239-247
    max_index = hi[0] if len(hi) == 1 else torch.max(hi[0], hi[1])
    new_cube_size = (torch.max(max_index - min_index) + 4) * voxel_size   # :250
    new_voxel_size = new_cube_size / (N - 1)                              # :252
    new_origin = (min_index - 2) * voxel_size - 1.0                       # :254
    return new_voxel_size, new_origin, bbox


def two_pass_volumes(state_dict, latent, specs, N, mano_results=None, obj_results=None, hand_branch=True,
                     obj_branch=True, max_batch=2 ** 18, integer_mode=False):
    """Everything create_mesh_combined_decoder does before marching cubes (utils/mesh.py:17-121)."""
    voxel_size = 2.0 / (N - 1)                                                  # :25
    c1 = grid_coords(N, voxel_size, [-1, -1, -1], integer_mode)                 # :27-40
    h1, o1 = decode_points(state_dict, latent, c1, specs, mano_results, obj_results, max_batch)   # :46-63
    vh1, vo1 = h1.reshape(N, N, N), o1.reshape(N, N, N)                         # :65-75
    nvs, norg, bbox = get_higher_res_cube(hand_branch, obj_branch, vh1, vo1, N, voxel_size)      # :78-80
    c2 = grid_coords(N, nvs, norg, integer_mode)                                # :82-96
    h2, o2 = decode_points(state_dict, latent, c2, specs, mano_results, obj_results, max_batch)   # :98-115
    return {
        "coords1": c1, "vol_hand1": vh1, "vol_obj1": vo1, "bbox": bbox, "new_voxel_size": nvs, "new_origin": norg,
        "coords2": c2, "vol_hand2": h2.reshape(N, N, N), "vol_obj2": o2.reshape(N, N, N),
    }


def legacy_volume(decoder_fn, latent, N, max_batch=32 ** 3):
    """deep_sdf.mesh.create_mesh before marching cubes (deep_sdf/mesh.py:14-54): one pass on [-1,1]^3,
    `decoder_fn(inputs[M, L+3]) -> [M,1]`, latent None => inputs = queries (deep_sdf/utils.py:64-75)."""
    coords = grid_coords(N, 2.0 / (N - 1), [-1, -1, -1])
    out = torch.zeros(N ** 3)
    with torch.no_grad():
        for head in range(0, N ** 3, max_batch):
            q = coords[head:head + max_batch]
            inputs = q if latent is None else torch.cat([latent.expand(q.shape[0], -1), q], 1)
            out[head:head + max_batch] = decoder_fn(inputs).squeeze(1)
    return out.reshape(N, N, N)
