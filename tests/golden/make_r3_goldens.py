"""Round-3 golden vectors: MORE SAMPLES at BASELINE.json's full sizes, produced by RUNNING THE REFERENCE (zerchen/AlignSDF at
/root/reference) in the build container.

Rounds 1 / 2 pinned one sample (latent 0) per configuration at N = 128 / 256; one sign flip at 256^3 changes the vertex and face
counts, so one sample is thin evidence for "identical triangle counts".  This script runs the reference's
create_mesh_combined_decoder (utils/mesh.py:17-195) on

    nerf3 (ObMan decoder)              N = 128: synthetic samples 1..8,   N = 256: samples 1..6
    both9 (DexYCB MANO-aligned)        N = 128: samples 1..8 (each with its own pose_inputs(s)),   N = 256: samples 1..6

and records, per (tag, N, sample): the negative-voxel boxes and counts of pass 1, the zoom cube (new_voxel_size, new_origin),
8192 probes per head and pass, the number of voxels within 1e-6 of the level, and - step 2, skimage 0.18.3 under
/opt/conda/bin/python3.9 with the reference's call (utils/mesh.py:354) - V, F and coordinate / index checksums of both surfaces.
-> tests/golden/ref_fullsize_r3_<tag>.npz, keys "<N>/s<sample>/<name>".

Step 3 (`--signs`, from the pass-2 volumes step 1 parked in /tmp): the SIGN of every voxel of both pass-2 volumes (packed bits) and
the list of voxels within 2e-6 of the level with their values - so that a test can compare every voxel's sign with the reference's
and attribute each disagreement to a voxel whose sign is undefined at the 1e-5 bar.

Usage:  python tests/golden/make_r3_goldens.py nerf3 [both9]          (~20 min of CPU per tag with 4 threads)
        /opt/conda/bin/python3.9 tests/golden/make_r3_goldens.py --mc nerf3 [both9]
        python tests/golden/make_r3_goldens.py --signs nerf3 [both9]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
TMP = "/tmp/asdf_r3_%s_%d_s%d_%s.npy"
PLAN = ((128, tuple(range(1, 9))), (256, (1, 2, 3, 4, 5, 6)))


def out_path(tag):
    return os.path.join(HERE, "ref_fullsize_r3_%s.npz" % tag)


def decode(tags):
    import torch
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    from alignsdf_amd import synthetic as syn
    import make_r2_goldens as r2
    import make_ref_goldens as mrg
    arch, um, uu, _ = mrg.import_reference()
    for tag in tags:
        specs, sd = syn.specs_for(tag), syn.full_state_dict(tag)
        dec = arch.SeparateDecoder(256, specs["PointFeatSize"], specs["EncodeStyle"], **specs["NetworkSpecs"], use_classifier=False).eval()
        dec.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        gold = dict(np.load(out_path(tag))) if os.path.exists(out_path(tag)) else {}
        for N, samples in PLAN:
            for s in samples:
                if "%d/s%d/bbox" % (N, s) in gold:
                    continue
                latent, mano, obj = r2._torch_inputs(tag, syn, torch, s)
                g, vols2 = r2.two_pass(um, dec, latent, mano, obj, specs, N)
                for k, v in g.items():
                    name = k[:-len("_%d" % N)]
                    if name == "probe_sel":
                        gold["%d/probe_sel" % N] = v            # the same index set for every sample of a size
                    else:
                        gold["%d/s%d/%s" % (N, s, name)] = v
                for part, (v, _, _) in vols2.items():
                    np.save(TMP % (tag, N, s, part), v)
                print(tag, N, "sample", s, "zoom", g["new_voxel_size_%d" % N], g["new_origin_%d" % N], "neg", g["neg_count_%d" % N], flush=True)
                np.savez_compressed(out_path(tag), **gold)


def mc(tags):
    from skimage.measure import marching_cubes_lewiner
    for tag in tags:
        gold = dict(np.load(out_path(tag)))
        for N, samples in PLAN:
            for s in samples:
                if "%d/s%d/bbox" % (N, s) not in gold or "%d/s%d/mc_hand" % (N, s) in gold:
                    continue
                vs = np.float32(gold["%d/s%d/new_voxel_size" % (N, s)][0])
                for part in ("hand", "obj"):
                    vol = np.load(TMP % (tag, N, s, part))
                    v, f, _, _ = marching_cubes_lewiner(vol, level=0.0, spacing=[vs] * 3)
                    gold["%d/s%d/mc_%s" % (N, s, part)] = np.array([len(v), len(f)])
                    gold["%d/s%d/mc_%s_vsum" % (N, s, part)] = v.astype(np.float64).sum(0)
                    gold["%d/s%d/mc_%s_fsum" % (N, s, part)] = f.astype(np.int64).sum(0)
                    print(tag, N, "sample", s, part, "V", len(v), "F", len(f), flush=True)
        np.savez_compressed(out_path(tag), **gold)


def signs(tags):
    for tag in tags:
        gold = dict(np.load(out_path(tag)))
        for N, samples in PLAN:
            for s in samples:
                if "%d/s%d/bbox" % (N, s) not in gold:
                    continue
                for part in ("hand", "obj"):
                    vol = np.load(TMP % (tag, N, s, part)).reshape(-1)
                    gold["%d/s%d/sign_bits_%s" % (N, s, part)] = np.packbits(vol < 0)
                    near = np.flatnonzero(np.abs(vol) < 2e-6)
                    gold["%d/s%d/near_idx_%s" % (N, s, part)] = near.astype(np.int32)
                    gold["%d/s%d/near_val_%s" % (N, s, part)] = vol[near]
                    print(tag, N, "sample", s, part, "negative", int((vol < 0).sum()), "within 2e-6 of the level", len(near), flush=True)
        np.savez_compressed(out_path(tag), **gold)


if __name__ == "__main__":
    tags = [a for a in sys.argv[1:] if not a.startswith("-")] or ["nerf3", "both9"]
    if "--legacy-free" in sys.argv:
        pass                                   # handled at the end of the file
    elif "--mc" in sys.argv:
        mc(tags)
    elif "--signs" in sys.argv:
        signs(tags)
    else:
        decode(tags)


def legacy_free():
    """The legacy entry points on a latent-FREE single-output module (deep_sdf/utils.py:64-75 with latent_vector None,
    deep_sdf/mesh.py:14-61): the reference's create_mesh volume at N = 32 and its decode_sdf on 4096 random points
    -> ref_legacy_free.npz.      python tests/golden/make_r3_goldens.py --legacy-free"""
    import torch
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    from alignsdf_amd import synthetic as syn
    import make_ref_goldens as mrg
    _, _, _, dm = mrg.import_reference()
    import deep_sdf.utils as du
    net = syn.latent_free_module()
    cap = []
    dm.convert_sdf_samples_to_ply = lambda vol, origin, vs, path: cap.append((vol.numpy().copy(), origin, vs))
    pts = torch.from_numpy(syn.uniform((4096, 3), 778, -1.0, 1.0).astype(np.float32))
    with torch.no_grad():
        dm.create_mesh(net, None, "/tmp/z", N=32, max_batch=32 ** 3)
        rand = du.decode_sdf(net, None, pts)
    vol = cap[0][0]
    print("latent-free volume range", vol.min(), vol.max(), "negative voxels", int((vol < 0).sum()))
    np.savez_compressed(os.path.join(HERE, "ref_legacy_free.npz"), vol_32=vol, origin=np.array(cap[0][1], dtype=np.float64),
                        voxel_size=np.array([cap[0][2]]), rand_pts=pts.numpy(), rand_sdf=rand.squeeze(1).numpy())


if "--legacy-free" in sys.argv and __name__ == "__main__":
    legacy_free()
