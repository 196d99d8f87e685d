"""Print the max |HIP - reference| SDF difference on the committed reference goldens (GPU box)."""
import sys
import numpy as np, torch
sys.path.insert(0, '.')
from alignsdf_amd import synthetic as syn
from alignsdf_amd.hip_decoder import HipSdfDecoder, kinematic_affine
for tag in ("nerf3", "both9"):
    g = np.load("tests/golden/ref_decoder_%s.npz" % tag)
    specs = syn.specs_for(tag)
    dec = HipSdfDecoder(syn.full_state_dict(tag), 256, specs["PointFeatSize"], specs["EncodeStyle"], device="cuda:0")
    emb = None
    if tag == "both9":
        m, o = syn.pose_inputs(0)
        emb = kinematic_affine(9, "both", specs["SdfScaleFactor"], {k: torch.from_numpy(v) for k, v in m.items()},
                               {k: torch.from_numpy(v) for k, v in o.items()})
    dec.set_sample(torch.from_numpy(syn.latent_code(0)), emb)
    h, o, _ = dec.decode_grid(32, [-1, -1, -1], 2.0 / 31)
    print(tag, "pass1 32^3 max|d| hand %.3e obj %.3e" % (np.abs(h.cpu().numpy() - g["vol1_hand_32"]).max(), np.abs(o.cpu().numpy() - g["vol1_obj_32"]).max()))
    h, o, _ = dec.decode_grid(32, g["new_origin_32"], g["new_voxel_size_32"][0])
    dh, do = np.abs(h.cpu().numpy() - g["vol2_hand_32"]), np.abs(o.cpu().numpy() - g["vol2_obj_32"])
    print(tag, "pass2 32^3 max|d| hand %.3e obj %.3e   mean %.3e %.3e  max|sdf| %.3f" % (dh.max(), do.max(), dh.mean(), do.mean(), np.abs(g["vol2_hand_32"]).max()))
    for N in (64,):
        h, o, _ = dec.decode_grid(N, g["new_origin_%d" % N], g["new_voxel_size_%d" % N][0])
        sel = g["probe_sel_%d" % N]
        print(tag, "pass2 N=%d probes max|d| hand %.3e obj %.3e" % (N, np.abs(h.cpu().numpy().reshape(-1)[sel] - g["p2_hand_%d" % N]).max(), np.abs(o.cpu().numpy().reshape(-1)[sel] - g["p2_obj_%d" % N]).max()))
    hp, op = dec.decode_points(torch.from_numpy(g["rand_pts"]))
    print(tag, "random pts max|d| hand %.3e obj %.3e" % (np.abs(hp.cpu().numpy() - g["rand_hand"]).max(), np.abs(op.cpu().numpy() - g["rand_obj"]).max()))
