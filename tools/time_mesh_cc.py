"""Device time of K8 (largest-component filter) on an N=256 decoder surface."""
import sys, torch, numpy as np
sys.path.insert(0, ".")
from alignsdf_amd import synthetic as syn, mesh_post
from alignsdf_amd.networks.model import build_decoder
from alignsdf_amd.utils.mesh import decode_two_pass
from alignsdf_amd.marching_cubes import marching_cubes_device
specs = syn.specs_for("nerf3")
dec = build_decoder(specs, {k: torch.from_numpy(v) for k, v in syn.full_state_dict("nerf3").items()})
r = decode_two_pass(True, True, dec, torch.from_numpy(syn.latent_code(1)).cuda(), None, None, specs, 256)
for part in ("hand", "obj"):
    v, f = marching_cubes_device(r["vol_" + part], 0.0)
    mesh_post.keep_largest_component_device(v, f, r["voxel_size"], r["origin"])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        out = mesh_post.keep_largest_component_device(v, f, r["voxel_size"], r["origin"])
    e1.record(); torch.cuda.synchronize()
    print(part, "V %d F %d: %.3f ms per filter, counts %s" % (v.shape[0], f.shape[0], e0.elapsed_time(e1) / 10, out[2].cpu().tolist()))
