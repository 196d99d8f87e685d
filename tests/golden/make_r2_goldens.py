"""Round-2 golden vectors, produced by RUNNING THE REFERENCE (zerchen/AlignSDF at /root/reference) in the build container.

Sections (argv; default = all of fit small variants full):
  fit       ridge-fitted last layers of the additional synthetic configurations -> synth_last_layer_r2.npz (an input of
            alignsdf_amd/synthetic.py)
  small     the reference's create_mesh_combined_decoder at N = 32 / 64 + decode_sdf_multi_output on random points for the
            configurations the round-1 fixtures did not cover: NeRF encoding with PointFeatSize 15, EncodeStyle "hand" with
            the wrist joint (pf 6) and with all 16 joints (pf 51, utils/utils.py:399-400), EncodeStyle "obj" (pf 6)
            -> ref_decoder_<tag>.npz
  variants  decoder variants the HIP kernels do not cover (they run the module on PyTorch-ROCm): use_tanh, LayerNorm
            (weight_norm false), CombinedDecoder with xyz_in_all, a pose-aligned model evaluated without mano_results
            (NeRF branch of utils/mesh.py:53-55), PixelAlign (utils/utils.py:536-566) -> ref_variant_<name>.npz
  full      BASELINE.json configs at their full sizes: the DexYCB MANO-aligned decoder ("both9", configs[4]) at
            N = 128 and 256, and the hand-only run of configs[0] at N = 64 -> ref_fullsize_both9.npz, ref_fullsize_hand64.npz
            (the skimage step is `--mc` under /opt/conda/bin/python3.9, like make_fullsize_goldens.py)

Usage:  python tests/golden/make_r2_goldens.py [sections...]
        /opt/conda/bin/python3.9 tests/golden/make_r2_goldens.py --mc
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
TMP = "/tmp/asdf_r2_%s_%d_%s.npy"
FULL = tuple(c for c in (("both9", (128, 256), True, True), ("hand64", (64,), True, False))
             if not os.environ.get("ASDF_ONLY_FULL") or c[0] == os.environ["ASDF_ONLY_FULL"])


def _torch_inputs(tag, syn, torch, sample=0):
    lat = torch.from_numpy(syn.latent_code(sample))
    mano = obj = None
    if syn.specs_for(tag)["EncodeStyle"] != "nerf":
        m, o = syn.pose_inputs(sample)
        mano = {k: torch.from_numpy(v) for k, v in m.items()}
        obj = {k: torch.from_numpy(v) for k, v in o.items()}
    return lat, mano, obj


def fit():
    import torch
    from alignsdf_amd import synthetic as syn
    from oracle import sdf_oracle as orc
    out = {}
    pts = syn.uniform((60000, 3), 424242, -1.0, 1.0).astype(np.float32)
    hand_t, obj_t = syn.analytic_sdf(pts)
    targets = {"h": np.arctanh(np.clip(0.5 * hand_t, -0.05, 0.05)), "o": np.arctanh(np.clip(0.5 * obj_t, -0.05, 0.05))}
    for tag in ("nerf15", "hand6", "hand51", "obj6"):
        specs = syn.specs_for(tag)
        pf, style = specs["PointFeatSize"], specs["EncodeStyle"]
        sd = syn.hidden_state_dict(256, pf, style, 0)
        latent, mano, obj = _torch_inputs(tag, syn, torch)
        with torch.no_grad():
            feats = orc.point_features(torch.from_numpy(pts), specs, mano, obj)
            inputs = torch.cat([latent.expand(pts.shape[0], -1), feats], 1)
            xin = {"nerf": {"h": inputs, "o": inputs}, "hand": {"h": inputs, "o": inputs[:, :259]},
                   "obj": {"h": inputs[:, :259], "o": inputs}}[style]
            for head in "ho":
                params = [(orc.effective_weight(sd["lin%s%d.weight_v" % (head, k)], sd["lin%s%d.weight_g" % (head, k)]),
                           torch.from_numpy(sd["lin%s%d.bias" % (head, k)])) for k in range(4)] + [(torch.zeros(1, 512), torch.zeros(1))]
                hid = orc._run_head(params, xin[head], xin[head], stop_before_last=True).double().numpy()
                X = np.concatenate([hid, np.ones((hid.shape[0], 1))], 1)
                w = np.linalg.solve(X.T @ X + 1e-3 * np.eye(X.shape[1]), X.T @ targets[head])
                print("fit %s head %s: rms %.3e  max hidden %.1f" % (tag, head, np.sqrt(np.mean((X @ w - targets[head]) ** 2)), hid.max()))
                out["%s.lin%s4.weight" % (tag, head)] = w[:-1].astype(np.float32).reshape(1, 512)
                out["%s.lin%s4.bias" % (tag, head)] = w[-1:].astype(np.float32)
    np.savez(os.path.join(HERE, "synth_last_layer_r2.npz"), **out)


def two_pass(um, dec, latent, mano, obj, specs, N, hand=True, objb=True, cam=None, keep_full=False):
    """One run of the reference's create_mesh_combined_decoder with its post-processing captured."""
    import torch
    cap = {"mc": [], "zoom": []}

    def fake_convert(vol, origin, vs, path, offset=None, scale=None, eval_mode=False, task="obman"):
        cap["mc"].append((path, vol.numpy().copy(), np.array(origin, dtype=np.float64), vs))
        return None, None, np.array([0, 0, 0]), np.array([1])

    real_zoom = um.get_higher_res_cube

    def spy_zoom(hb, ob, vh, vo, n, org, vs):
        r = real_zoom(hb, ob, vh, vo, n, org, vs)
        cap["zoom"].append((None if vh is None else vh.numpy().copy(), None if vo is None else vo.numpy().copy(), r[0].clone(), r[1].clone()))
        return r

    um.convert_sdf_samples_to_ply, um.get_higher_res_cube = fake_convert, spy_zoom
    try:
        with torch.no_grad():
            um.create_mesh_combined_decoder(hand, objb, False, dec, latent, mano, obj, cam, specs, "/tmp/x", N=N, max_batch=2 ** 18)
    finally:
        um.get_higher_res_cube = real_zoom
    vh1, vo1, nvs, norg = cap["zoom"][0]
    vols2 = {p.split("_")[-1].split(".")[0]: (v, o, s) for p, v, o, s in cap["mc"]}
    import make_ref_goldens as mrg
    sel = mrg.probe_indices(N ** 3)
    g = {"probe_sel_%d" % N: sel, "new_voxel_size_%d" % N: nvs.numpy().reshape(1), "new_origin_%d" % N: norg.numpy()}
    first = next(iter(vols2.values()))
    g["mc_origin_%d" % N] = first[1]
    g["mc_voxel_size_%d" % N] = np.float32(first[2].item() if hasattr(first[2], "item") else first[2]).reshape(1)
    bbox = -np.ones((2, 6), dtype=np.int64)
    negs = []
    for k, (on, v1, part) in enumerate(((hand, vh1, "hand"), (objb, vo1, "obj"))):
        if not on:
            negs += [0, 0]
            continue
        g["p1_%s_%d" % (part, N)] = v1.reshape(-1)[sel]
        g["p2_%s_%d" % (part, N)] = vols2[part][0].reshape(-1)[sel]
        nz = np.argwhere(v1 < 0)
        if len(nz):
            bbox[k, :3], bbox[k, 3:] = nz.min(0), nz.max(0)
        negs += [int((v1 < 0).sum()), int((vols2[part][0] < 0).sum())]
        g["near_zero_%s_%d" % (part, N)] = np.array([(np.abs(vols2[part][0]) < 1e-6).sum()])
        if keep_full:
            g["vol1_%s_%d" % (part, N)], g["vol2_%s_%d" % (part, N)] = v1, vols2[part][0]
    g["bbox_%d" % N] = bbox
    g["neg_count_%d" % N] = np.array([negs[0], negs[2], negs[1], negs[3]])       # p1 hand, p1 obj, p2 hand, p2 obj
    return g, vols2


def random_points(uu, dec, latent, mano, obj, specs, syn, torch, cam=None):
    """decode_sdf_multi_output on 4096 random points through the reference's own embedding helpers."""
    pts = torch.from_numpy(syn.uniform((4096, 3), 777, -1.0, 1.0).astype(np.float32))
    g = {"rand_pts": pts.numpy()}
    with torch.no_grad():
        q = pts
        if specs["PointFeatSize"] > 3 and mano is not None and specs["EncodeStyle"] != "nerf":
            q = uu.kinematic_embedding(pts, mano, pts.shape[0], specs["PointFeatSize"], specs["SdfScaleFactor"], obj, specs["EncodeStyle"])
            g["embed_pts"] = q.numpy()
        elif specs["PointFeatSize"] > 3:
            emb, _ = uu.get_nerf_embedder((specs["PointFeatSize"] - 3) // 6)
            q = emb(pts)
            g["embed_pts"] = q.numpy()
        h, o, _ = uu.decode_sdf_multi_output(dec, latent, q, mano, cam, specs)
    g["rand_hand"], g["rand_obj"] = h.squeeze(1).numpy(), o.squeeze(1).numpy()
    return g


def small(ref):
    import torch
    from alignsdf_amd import synthetic as syn
    arch, um, uu, _ = ref
    for tag in ("nerf15", "hand6", "hand51", "obj6"):
        specs, sd = syn.specs_for(tag), syn.full_state_dict(tag)
        dec = arch.SeparateDecoder(256, specs["PointFeatSize"], specs["EncodeStyle"], **specs["NetworkSpecs"], use_classifier=False).eval()
        dec.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        latent, mano, obj = _torch_inputs(tag, syn, torch)
        gold = {}
        for N in (32, 64):
            g, _ = two_pass(um, dec, latent, mano, obj, specs, N, keep_full=(N == 32))
            gold.update(g)
            print(tag, N, "new_vs", g["new_voxel_size_%d" % N], "neg", g["neg_count_%d" % N], flush=True)
        gold.update(random_points(uu, dec, latent, mano, obj, specs, syn, torch))
        np.savez_compressed(os.path.join(HERE, "ref_decoder_%s.npz" % tag), **gold)


def variant_state(name, syn, torch):
    """alignsdf_amd.synthetic.variant_config as torch tensors."""
    specs, cls, sd, mano, obj, cam, latent = syn.variant_config(name)
    t = lambda d: None if d is None else {k: torch.from_numpy(v) for k, v in d.items()}
    return specs, cls, sd, t(mano), t(obj), None if cam is None else torch.from_numpy(cam), torch.from_numpy(latent)


VARIANTS = ("tanh", "layernorm", "xyzall", "nomano", "pixelalign", "narrow")     # (narrow: round 3)


def variants(ref):
    import torch
    from alignsdf_amd import synthetic as syn
    arch, um, uu, _ = ref
    for name in [v for v in VARIANTS if not os.environ.get("ASDF_ONLY_VARIANT") or v == os.environ["ASDF_ONLY_VARIANT"]]:
        specs, cls, sd, mano, obj, cam, latent = variant_state(name, syn, torch)
        dec = getattr(arch, cls)(specs["LatentSize"], specs["PointFeatSize"], specs["EncodeStyle"], **specs["NetworkSpecs"], use_classifier=False).eval()
        dec.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
        g, _ = two_pass(um, dec, latent, mano, obj, specs, 32, cam=cam, keep_full=True)
        g.update(random_points(uu, dec, latent, mano, obj, specs, syn, torch, cam=cam))
        print(name, "new_vs", g["new_voxel_size_32"], "neg", g["neg_count_32"], "rand hand range", g["rand_hand"].min(), g["rand_hand"].max(), flush=True)
        np.savez_compressed(os.path.join(HERE, "ref_variant_%s.npz" % name), **g)


def full(ref):
    import torch
    from alignsdf_amd import synthetic as syn
    arch, um, uu, _ = ref
    for name, sizes, hb, ob in FULL:
        tag = "both9" if name == "both9" else "nerf3"
        specs, sd = syn.specs_for(tag), syn.full_state_dict(tag)
        dec = arch.SeparateDecoder(256, specs["PointFeatSize"], specs["EncodeStyle"], **specs["NetworkSpecs"], use_classifier=False).eval()
        dec.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        latent, mano, obj = _torch_inputs(tag, syn, torch)
        gold = {}
        for N in sizes:
            g, vols2 = two_pass(um, dec, latent, mano, obj, specs, N, hand=hb, objb=ob)
            gold.update(g)
            for part, (v, _, _) in vols2.items():
                np.save(TMP % (name, N, part), v)
            print(name, N, "zoom", g["new_voxel_size_%d" % N], g["new_origin_%d" % N], "neg", g["neg_count_%d" % N], flush=True)
        np.savez_compressed(os.path.join(HERE, "ref_fullsize_%s.npz" % name), **gold)


def mc():
    from skimage.measure import marching_cubes_lewiner
    for name, sizes, hb, ob in FULL:
        out = os.path.join(HERE, "ref_fullsize_%s.npz" % name)
        gold = dict(np.load(out))
        for N in sizes:
            vs = np.float32(gold["new_voxel_size_%d" % N][0])
            for part, on in (("hand", hb), ("obj", ob)):
                if not on:
                    continue
                vol = np.load(TMP % (name, N, part))
                v, f, _, _ = marching_cubes_lewiner(vol, level=0.0, spacing=[vs] * 3)
                gold["mc_%s_%d" % (part, N)] = np.array([len(v), len(f)])
                gold["mc_%s_%d_vsum" % (part, N)] = v.astype(np.float64).sum(0)
                gold["mc_%s_%d_fsum" % (part, N)] = f.astype(np.int64).sum(0)
                print(name, N, part, "V", len(v), "F", len(f), flush=True)
        np.savez_compressed(out, **gold)


if __name__ == "__main__":
    if "--mc" in sys.argv:
        mc()
    else:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, HERE)
        want = [a for a in sys.argv[1:] if not a.startswith("-")] or ["fit", "small", "variants", "full"]
        if "fit" in want:
            fit()
        if set(want) - {"fit"}:
            import make_ref_goldens as mrg
            ref = mrg.import_reference()
            for sec in ("small", "variants", "full"):
                if sec in want:
                    {"small": small, "variants": variants, "full": full}[sec](ref)
