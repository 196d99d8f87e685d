"""End-to-end timing of alignsdf_amd.reconstruct.reconstruct() writing PLY files (N=256), with a per-stage host breakdown."""
import json, os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, ".")
from alignsdf_amd import synthetic as syn, reconstruct as rc
from alignsdf_amd.networks.model import build_decoder
from alignsdf_amd import mesh_post
from alignsdf_amd.utils import mesh as mu

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n_samples = int(sys.argv[2]) if len(sys.argv) > 2 else 6
specs = syn.specs_for("nerf3")
dec = build_decoder(specs, {k: torch.from_numpy(v) for k, v in syn.full_state_dict("nerf3").items()})
tmp = tempfile.mkdtemp()
split = os.path.join(tmp, "split.json")
json.dump({"filenames": ["x/%08d.jpg" % i for i in range(n_samples + 1)]}, open(split, "w"))
rc.reconstruct(dec, specs, split, tmp, 0, 1, cube_dim=N)          # warm-up
torch.cuda.synchronize()
t = time.perf_counter()
recs = rc.reconstruct(dec, specs, split, tmp, 1, n_samples + 1, cube_dim=N)
torch.cuda.synchronize()
dt = time.perf_counter() - t
print("reconstruct() with PLY export: %d samples, %.1f ms/sample (N=%d), F_hand %d F_obj %d" % (n_samples, 1e3 * dt / n_samples, N, recs[-1]["F_hand"], recs[-1]["F_obj"]))
# host tail breakdown on the last sample's hand mesh
from alignsdf_amd.ply import read_ply
r = next(iter(rc.pipelined_two_pass(dec, specs, [(0, torch.from_numpy(syn.latent_code(1)).cuda(), None, None)], N)))[1]
torch.cuda.synchronize()
t0 = time.perf_counter(); v, f, mp = mu.place_vertices(r["verts_hand"], r["faces_hand"], r["origin"], r["voxel_size"]); t1 = time.perf_counter()
ov, of = mesh_post.keep_largest_component(mp, f); t2 = time.perf_counter()
from alignsdf_amd.ply import write_ply
write_ply(os.path.join(tmp, "t.ply"), ov, of); t3 = time.perf_counter()
print("host tail per surface (V=%d F=%d): D2H+place %.1f ms, largest component %.1f ms, PLY write %.1f ms" % (len(v), len(f), 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2)))
