"""Generate golden vectors by RUNNING THE REFERENCE (zerchen/AlignSDF at /root/reference).

Runs only in the build container (the reference tree does not travel to the GPU box); the outputs
are committed as small .npz fixtures next to this script:

  synth_last_layer.npz   ridge-fitted last layers of the synthetic decoders (an input, see
                         alignsdf_amd/synthetic.py)
  ref_grid.npz           grid index / coordinate columns from utils/mesh.py:27-40 for several N
  ref_decoder_<tag>.npz  probe SDF values of both passes, zoom-cube parameters, negative-voxel bbox,
                         full 32^3 pass-2 volumes, kinematic-embedding probes - all produced by the
                         reference's own create_mesh_combined_decoder / SeparateDecoder /
                         kinematic_embedding, imported with stubbed third-party modules.
  ref_legacy.npz         volume of deep_sdf.mesh.create_mesh (legacy entry point)

Usage:  python tests/golden/make_ref_goldens.py
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)

from alignsdf_amd import synthetic as syn  # noqa: E402
from oracle import sdf_oracle as orc  # noqa: E402

REF = "/root/reference"


def import_reference():
    for name in ["torchvision", "torchvision.models", "torchvision.transforms", "cv2", "trimesh", "lmdb", "plyfile",
                 "skimage", "skimage.measure"]:
        sys.modules[name] = types.ModuleType(name)
    sys.modules["torchvision"].models = sys.modules["torchvision.models"]
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["skimage"].measure = sys.modules["skimage.measure"]
    # the reference's `utils` / `networks` / `deep_sdf` packages must win over anything else
    sys.path.insert(0, REF)
    torch.Tensor.cuda = lambda self, *a, **k: self   # networks/model.py:350, utils/mesh.py:48,100
    import networks.model as arch
    import utils.mesh as um
    import utils.utils as uu
    import deep_sdf.mesh as dm
    return arch, um, uu, dm


def fit_last_layers():
    """Ridge-fit lin{h,o}4 so the heads approximate sphere / box SDFs; save synth_last_layer.npz."""
    out = {}
    pts = syn.uniform((60000, 3), 424242, -1.0, 1.0).astype(np.float32)
    hand_t, obj_t = syn.analytic_sdf(pts)
    targets = {"h": np.arctanh(np.clip(0.5 * hand_t, -0.05, 0.05)), "o": np.arctanh(np.clip(0.5 * obj_t, -0.05, 0.05))}
    for tag in ("nerf3", "both9", "nerf9"):
        specs = syn.specs_for(tag)
        sd = syn.hidden_state_dict(256, specs["PointFeatSize"], specs["EncodeStyle"], 0)
        latent = torch.from_numpy(syn.latent_code(0))
        mano, obj = (None, None)
        if tag == "both9":
            m, o = syn.pose_inputs(0)
            mano = {k: torch.from_numpy(v) for k, v in m.items()}
            obj = {k: torch.from_numpy(v) for k, v in o.items()}
        with torch.no_grad():
            feats = orc.point_features(torch.from_numpy(pts), specs, mano, obj)
            inputs = torch.cat([latent.expand(pts.shape[0], -1), feats], 1)
            if specs["EncodeStyle"] == "nerf":
                xin = {"h": inputs, "o": inputs}
            else:
                xin = {"h": inputs[:, :-3], "o": torch.cat([inputs[:, :256 + 3], inputs[:, -3:]], 1)}
            for head in "ho":
                params = []
                for layer in range(4):
                    name = "lin%s%d" % (head, layer)
                    params.append((orc.effective_weight(sd[name + ".weight_v"], sd[name + ".weight_g"]),
                                   torch.from_numpy(sd[name + ".bias"])))
                params.append((torch.zeros(1, 512), torch.zeros(1)))
                hid = orc._run_head(params, xin[head], xin[head], stop_before_last=True).double().numpy()
                X = np.concatenate([hid, np.ones((hid.shape[0], 1))], 1)
                A = X.T @ X + 1e-3 * np.eye(X.shape[1])
                w = np.linalg.solve(A, X.T @ targets[head])
                rms = np.sqrt(np.mean((X @ w - targets[head]) ** 2))
                print("fit %s head %s: rms %.3e" % (tag, head, rms))
                out["%s.lin%s4.weight" % (tag, head)] = w[:-1].astype(np.float32).reshape(1, 512)
                out["%s.lin%s4.bias" % (tag, head)] = w[-1:].astype(np.float32)
    # CombinedDecoder ("comb3"): one MLP, the two-row last layer fitted to both targets
    sd = syn.combined_hidden_state_dict(256, 3, 0)
    latent = torch.from_numpy(syn.latent_code(0))
    with torch.no_grad():
        inputs = torch.cat([latent.expand(pts.shape[0], -1), torch.from_numpy(pts)], 1)
        params = [(orc.effective_weight(sd["lin%d.weight_v" % k], sd["lin%d.weight_g" % k]), torch.from_numpy(sd["lin%d.bias" % k]))
                  for k in range(4)] + [(torch.zeros(2, 512), torch.zeros(2))]
        hid = orc._run_head(params, inputs, inputs, stop_before_last=True).double().numpy()
    X = np.concatenate([hid, np.ones((hid.shape[0], 1))], 1)
    A = X.T @ X + 1e-3 * np.eye(X.shape[1])
    W = np.stack([np.linalg.solve(A, X.T @ targets[h]) for h in "ho"])
    print("fit comb3: rms", [float(np.sqrt(np.mean((X @ W[i] - targets[h]) ** 2))) for i, h in enumerate("ho")])
    out["comb3.lin4.weight"] = W[:, :-1].astype(np.float32)
    out["comb3.lin4.bias"] = W[:, -1].astype(np.float32)
    np.savez(os.path.join(HERE, "synth_last_layer.npz"), **out)


def probe_indices(P, n=8192):
    """Fixed pseudo-random probe positions in [0, P)."""
    return np.sort((syn.splitmix64(np.arange(n, dtype=np.uint64), 99) % np.uint64(P)).astype(np.int64))


def main():
    fit_last_layers()
    arch, um, uu, dm = import_reference()

    # ---- grid columns (utils/mesh.py:27-40) for power-of-two and awkward N
    grid = {}
    for N in (4, 5, 7, 16, 31, 32, 64, 100):
        overall = torch.arange(0, N ** 3, 1, out=torch.LongTensor())
        s = torch.zeros(N ** 3, 3)
        s[:, 2] = overall % N
        s[:, 1] = (overall.long() / N) % N
        s[:, 0] = ((overall.long() / N) / N) % N
        vs = 2.0 / (N - 1)
        c = torch.zeros(N ** 3, 3)
        c[:, 0] = (s[:, 0] * vs) + -1
        c[:, 1] = (s[:, 1] * vs) + -1
        c[:, 2] = (s[:, 2] * vs) + -1
        sel = np.arange(N ** 3) if N <= 32 else probe_indices(N ** 3, 20000)
        grid["idx_%d" % N] = s.numpy()[sel]
        grid["coord_%d" % N] = c.numpy()[sel]
        grid["sel_%d" % N] = sel
    np.savez_compressed(os.path.join(HERE, "ref_grid.npz"), **grid)

    # ---- full two-pass runs of the reference
    for tag in ("nerf3", "both9", "comb3", "nerf9"):
        specs = syn.specs_for(tag)
        sd = syn.full_state_dict(tag)
        cls = arch.CombinedDecoder if tag == "comb3" else arch.SeparateDecoder
        dec = cls(256, specs["PointFeatSize"], specs["EncodeStyle"], **specs["NetworkSpecs"], use_classifier=False).eval()
        dec.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        latent = torch.from_numpy(syn.latent_code(0))
        mano, obj = None, None
        if tag == "both9":
            m, o = syn.pose_inputs(0)
            mano = {k: torch.from_numpy(v) for k, v in m.items()}
            obj = {k: torch.from_numpy(v) for k, v in o.items()}
        gold = {}
        for N in (32, 64):
            cap = {"mc": [], "zoom": []}

            def fake_convert(vol, origin, vs, path, offset=None, scale=None, eval_mode=False, task="obman"):
                cap["mc"].append((vol.numpy().copy(), np.array(origin, dtype=np.float64), vs))
                return None, None, np.array([0, 0, 0]), np.array([1])

            real_zoom = um.get_higher_res_cube

            def spy_zoom(hb, ob, vh, vo, n, org, vs):
                r = real_zoom(hb, ob, vh, vo, n, org, vs)
                cap["zoom"].append((vh.numpy().copy(), vo.numpy().copy(), r[0].clone(), r[1].clone()))
                return r

            um.convert_sdf_samples_to_ply = fake_convert
            um.get_higher_res_cube = spy_zoom
            with torch.no_grad():
                um.create_mesh_combined_decoder(True, True, False, dec, latent, mano, obj, None, specs, "/tmp/x", N=N,
                                                max_batch=2 ** 18)
            um.get_higher_res_cube = real_zoom
            vh1, vo1, nvs, norg = cap["zoom"][0]
            (vh2, org_h, vs_h), (vo2, org_o, vs_o) = cap["mc"]
            sel = probe_indices(N ** 3)
            gold["probe_sel_%d" % N] = sel
            gold["p1_hand_%d" % N] = vh1.reshape(-1)[sel]
            gold["p1_obj_%d" % N] = vo1.reshape(-1)[sel]
            gold["p2_hand_%d" % N] = vh2.reshape(-1)[sel]
            gold["p2_obj_%d" % N] = vo2.reshape(-1)[sel]
            gold["new_voxel_size_%d" % N] = nvs.numpy().reshape(1)
            gold["new_origin_%d" % N] = norg.numpy()
            gold["mc_origin_%d" % N] = org_h
            gold["mc_voxel_size_%d" % N] = np.float32(vs_h.item()).reshape(1)
            bbox = -np.ones((2, 6), dtype=np.int64)
            for k, v in enumerate((vh1, vo1)):
                nz = np.argwhere(v < 0)
                if len(nz):
                    bbox[k, :3] = nz.min(0)
                    bbox[k, 3:] = nz.max(0)
            gold["bbox_%d" % N] = bbox
            gold["neg_count_%d" % N] = np.array([(vh1 < 0).sum(), (vo1 < 0).sum(), (vh2 < 0).sum(), (vo2 < 0).sum()])
            if N == 32:
                gold["vol1_hand_32"], gold["vol1_obj_32"] = vh1, vo1
                gold["vol2_hand_32"], gold["vol2_obj_32"] = vh2, vo2
            print(tag, N, "new_vs", nvs.item(), "origin", norg.numpy(), "neg", gold["neg_count_%d" % N])
        # decoder on random points in [-1,1]^3 through the reference's own helpers
        pts = torch.from_numpy(syn.uniform((4096, 3), 777, -1.0, 1.0).astype(np.float32))
        with torch.no_grad():
            q = pts
            if specs["PointFeatSize"] > 3 and mano is not None and specs["EncodeStyle"] != "nerf":
                q = uu.kinematic_embedding(pts, mano, pts.shape[0], specs["PointFeatSize"], specs["SdfScaleFactor"], obj,
                                           specs["EncodeStyle"])
                gold["embed_pts"] = q.numpy()
            elif specs["PointFeatSize"] > 3:      # utils/mesh.py:53-55
                nerf_embedding, _ = uu.get_nerf_embedder((specs["PointFeatSize"] - 3) // 6)
                q = nerf_embedding(pts)
                gold["embed_pts"] = q.numpy()
            h, o, _ = uu.decode_sdf_multi_output(dec, latent, q, mano, None, specs)
        gold["rand_pts"] = pts.numpy()
        gold["rand_hand"] = h.squeeze(1).numpy()
        gold["rand_obj"] = o.squeeze(1).numpy()
        # effective weights as the module's hook computes them (row 0 of each layer, for the fold check)
        for head in ("",) if tag == "comb3" else "ho":
            for layer in range(4):
                gold["effw_%s%d_rows" % (head, layer)] = getattr(dec, "lin%s%d" % (head, layer)).weight.detach().numpy()[:4]
        np.savez_compressed(os.path.join(HERE, "ref_decoder_%s.npz" % tag), **gold)

    # ---- legacy deep_sdf.mesh.create_mesh (deep_sdf/mesh.py:14-61): single-output decoder
    specs = syn.specs_for("nerf3")
    sd = syn.full_state_dict("nerf3")
    dec = arch.SeparateDecoder(256, 3, "nerf", **specs["NetworkSpecs"], use_classifier=False).eval()
    dec.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    cap = []
    dm.convert_sdf_samples_to_ply = lambda vol, origin, vs, path: cap.append((vol.numpy().copy(), origin, vs))
    latent = torch.from_numpy(syn.latent_code(0))

    class HandOnly(torch.nn.Module):
        def forward(self, x):
            return dec(x)[0]

    with torch.no_grad():
        dm.create_mesh(HandOnly(), latent, "/tmp/y", N=32, max_batch=32 ** 3)
    np.savez_compressed(os.path.join(HERE, "ref_legacy.npz"), vol_32=cap[0][0], origin=np.array(cap[0][1], dtype=np.float64),
                        voxel_size=np.array([cap[0][2]]))
    print("done")


if __name__ == "__main__":
    main()
