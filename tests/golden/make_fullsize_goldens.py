"""Full-size goldens (N = 128, 256: BASELINE.json configs[1], configs[2]) produced by RUNNING THE REFERENCE.

Step 1 (python3, this script):   the reference's create_mesh_combined_decoder on the synthetic "nerf3" decoder at
    N = 128 and 256 - probes of both passes, zoom cube, negative-voxel boxes and counts; the pass-2 volumes are
    parked in /tmp.
Step 2 (/opt/conda/bin/python3.9 this script --mc):   skimage 0.18.3 marching_cubes_lewiner on those volumes with
    the reference's call (utils/mesh.py:354): V, F and coordinate / index checksums per surface.
Both steps write into tests/golden/ref_fullsize.npz.  Takes ~6 minutes of CPU (240 s for the N=256 sample).

Usage:  python tests/golden/make_fullsize_goldens.py && /opt/conda/bin/python3.9 tests/golden/make_fullsize_goldens.py --mc
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
OUT = os.path.join(HERE, "ref_fullsize.npz")
TMP = "/tmp/asdf_fullsize_%d_%s.npy"
SIZES = (128, 256)


def decode():
    import torch
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    from alignsdf_amd import synthetic as syn
    import make_ref_goldens as mrg
    arch, um, uu, _ = mrg.import_reference()
    specs, sd = syn.specs_for("nerf3"), syn.full_state_dict("nerf3")
    dec = arch.SeparateDecoder(256, 3, "nerf", **specs["NetworkSpecs"], use_classifier=False).eval()
    dec.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    latent = torch.from_numpy(syn.latent_code(0))
    gold = {}
    for N in SIZES:
        cap = {"mc": [], "zoom": []}

        def fake_convert(vol, origin, vs, path, offset=None, scale=None, eval_mode=False, task="obman"):
            cap["mc"].append((vol.numpy().copy(), np.array(origin, dtype=np.float64), vs))
            return None, None, np.array([0, 0, 0]), np.array([1])

        real_zoom = um.get_higher_res_cube

        def spy_zoom(hb, ob, vh, vo, n, org, vs):
            r = real_zoom(hb, ob, vh, vo, n, org, vs)
            cap["zoom"].append((vh.numpy().copy(), vo.numpy().copy(), r[0].clone(), r[1].clone()))
            return r

        um.convert_sdf_samples_to_ply, um.get_higher_res_cube = fake_convert, spy_zoom
        with torch.no_grad():
            um.create_mesh_combined_decoder(True, True, False, dec, latent, None, None, None, specs, "/tmp/x", N=N, max_batch=2 ** 18)
        um.get_higher_res_cube = real_zoom
        vh1, vo1, nvs, norg = cap["zoom"][0]
        (vh2, org_h, vs_h), (vo2, _, _) = cap["mc"]
        sel = mrg.probe_indices(N ** 3)
        gold["probe_sel_%d" % N] = sel
        for name, v in (("p1_hand", vh1), ("p1_obj", vo1), ("p2_hand", vh2), ("p2_obj", vo2)):
            gold["%s_%d" % (name, N)] = v.reshape(-1)[sel]
        gold["new_voxel_size_%d" % N] = nvs.numpy().reshape(1)
        gold["new_origin_%d" % N] = norg.numpy()
        gold["mc_origin_%d" % N] = org_h
        bbox = -np.ones((2, 6), dtype=np.int64)
        for k, v in enumerate((vh1, vo1)):
            nz = np.argwhere(v < 0)
            bbox[k, :3], bbox[k, 3:] = nz.min(0), nz.max(0)
        gold["bbox_%d" % N] = bbox
        gold["neg_count_%d" % N] = np.array([(vh1 < 0).sum(), (vo1 < 0).sum(), (vh2 < 0).sum(), (vo2 < 0).sum()])
        # how many voxels sit so close to the level that a 1e-6 SDF difference could change their sign
        gold["near_zero_%d" % N] = np.array([(np.abs(vh2) < 1e-6).sum(), (np.abs(vo2) < 1e-6).sum()])
        np.save(TMP % (N, "hand"), vh2)
        np.save(TMP % (N, "obj"), vo2)
        print(N, "zoom", nvs.item(), norg.numpy(), "neg", gold["neg_count_%d" % N], "near zero", gold["near_zero_%d" % N], flush=True)
    np.savez_compressed(OUT, **gold)


def mc():
    from skimage.measure import marching_cubes_lewiner
    gold = dict(np.load(OUT))
    for N in SIZES:
        vs = np.float32(gold["new_voxel_size_%d" % N][0])
        for part in ("hand", "obj"):
            vol = np.load(TMP % (N, part))
            v, f, _, _ = marching_cubes_lewiner(vol, level=0.0, spacing=[vs] * 3)
            gold["mc_%s_%d" % (part, N)] = np.array([len(v), len(f)])
            gold["mc_%s_%d_vsum" % (part, N)] = v.astype(np.float64).sum(0)
            gold["mc_%s_%d_fsum" % (part, N)] = f.astype(np.int64).sum(0)
            print(N, part, "V", len(v), "F", len(f), flush=True)
    np.savez_compressed(OUT, **gold)


if __name__ == "__main__":
    mc() if "--mc" in sys.argv else decode()
