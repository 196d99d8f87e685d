"""Where the host time of reconstruct() goes per sample (wrappers around the host-tail functions)."""
import collections, json, os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, ".")
from alignsdf_amd import synthetic as syn, reconstruct as rc, icp, mesh_post
from alignsdf_amd.networks.model import build_decoder
from alignsdf_amd.utils import mesh as mu

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n_samples = int(sys.argv[2]) if len(sys.argv) > 2 else 6
EVAL = len(sys.argv) > 3 and sys.argv[3] == "eval"
acc = collections.defaultdict(float)


calls = collections.defaultdict(list)
T0 = time.perf_counter()


def timed(mod, name, label=None):
    f = getattr(mod, name)
    def w(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[label or name] += time.perf_counter() - t
            calls[label or name].append((1e3 * (t - T0), 1e3 * (time.perf_counter() - t)))
    setattr(mod, name, w)


timed(mu, "export_surface")
timed(mu, "begin_export_surface")
timed(mu, "end_export_surface")
timed(mu, "write_ply")
timed(mu, "place_vertices")
timed(icp, "run_icp_f")
timed(icp, "sample_surface")
timed(icp, "load_obj")
from alignsdf_amd import marching_cubes as mcmod
timed(rc, "marching_cubes_device") if hasattr(rc, "marching_cubes_device") else None
timed(mcmod, "marching_cubes_device")
from alignsdf_amd import hip_decoder as hd
timed(hd.HipSdfDecoder, "decode_grid")
timed(hd.HipSdfDecoder, "set_sample")
timed(mu, "zoom_cube_from_bboxes")
timed(torch.cuda.Event, "synchronize", "event.synchronize")
_cpu = torch.Tensor.cpu
def cpu(self, *a, **k):
    t = time.perf_counter()
    try:
        return _cpu(self, *a, **k)
    finally:
        if self.is_cuda:
            acc["tensor.cpu (bbox / MC sizes)"] += time.perf_counter() - t
torch.Tensor.cpu = cpu

specs = syn.specs_for("nerf3")
dec = build_decoder(specs, {k: torch.from_numpy(v) for k, v in syn.full_state_dict("nerf3").items()})
tmp = tempfile.mkdtemp()
split = os.path.join(tmp, "split.json")
json.dump({"filenames": ["x/%08d.jpg" % i for i in range(n_samples + 1)]}, open(split, "w"))
kw = {}
if EVAL:
    gt_dir = os.path.join(tmp, "data", "obman", "test", "mesh_hand")
    os.makedirs(gt_dir)
    nu, nv = 96, 48
    th, ph = np.meshgrid(np.arange(nu) * 2 * np.pi / nu, (np.arange(nv) + 0.5) * np.pi / nv, indexing="ij")
    P = np.stack([np.sin(ph) * np.cos(th), np.sin(ph) * np.sin(th), np.cos(ph)], -1).reshape(-1, 3)
    P = (P * 0.35 + np.array([-0.25, 0, 0])) * 1.08 + np.array([0.03, -0.02, 0.015])
    idx = lambda i, j: (i % nu) * nv + j
    F = [(idx(i, j), idx(i + 1, j), idx(i + 1, j + 1)) for i in range(nu) for j in range(nv - 1)]
    for i in range(n_samples + 1):
        with open(os.path.join(gt_dir, "%08d.obj" % i), "w") as f:
            f.write("".join("v %.6f %.6f %.6f\n" % tuple(p) for p in P) + "".join("f %d %d %d\n" % (a + 1, b + 1, c + 1) for a, b, c in F))
    kw = dict(eval_mode=True, data_root=os.path.join(tmp, "data"))
kw["code_source"] = rc.synthetic_code_source("nerf3")
rc.reconstruct(dec, specs, split, tmp, 0, 1, cube_dim=N, **kw)
torch.cuda.synchronize()
acc.clear()
from alignsdf_amd.utils.utils import hip_decoder_for
hip = hip_decoder_for(dec)
hip.event_log = []
t = time.perf_counter()
rc.reconstruct(dec, specs, split, tmp, 1, n_samples + 1, cube_dim=N, **kw)
torch.cuda.synchronize()
dt = time.perf_counter() - t
print("%.1f ms/sample wall; per-sample host time inside:" % (1e3 * dt / n_samples))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("  %-32s %7.1f ms" % (k, 1e3 * v / n_samples))

ev = hip.event_log
dur = [a.elapsed_time(b) for a, b in ev]
gaps = [ev[i][1].elapsed_time(ev[i + 1][0]) for i in range(len(ev) - 1)]
print("K1 launches %d, mean %.1f ms; gaps after pass 1 (zoom readback): %s; gaps after pass 2 (next sample): %s" % (
    len(ev), sum(dur) / len(dur), ["%.1f" % g for g in gaps[0::2]], ["%.1f" % g for g in gaps[1::2]]))

for name in ("begin_export_surface", "end_export_surface", "marching_cubes_device", "export_surface", "decode_grid", "set_sample", "event.synchronize", "tensor.cpu (bbox / MC sizes)"):
    print(name, ["%.0f+%.0f" % c for c in calls[name][-14:]])
