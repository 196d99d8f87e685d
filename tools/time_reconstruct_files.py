"""End-to-end timing of alignsdf_amd.reconstruct.reconstruct() writing PLY files (N=256), with a per-stage host breakdown."""
import json, os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, ".")
from alignsdf_amd import synthetic as syn, reconstruct as rc
from alignsdf_amd.networks.model import build_decoder
from oracle import mesh_oracle as mesh_post   # host checker, timed here for comparison only
from alignsdf_amd.utils import mesh as mu

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n_samples = int(sys.argv[2]) if len(sys.argv) > 2 else 6
EVAL = len(sys.argv) > 3 and sys.argv[3] == "eval"      # eval mode: translate+scale ICP of every hand mesh to a ground truth
specs = syn.specs_for("nerf3")
dec = build_decoder(specs, {k: torch.from_numpy(v) for k, v in syn.full_state_dict("nerf3").items()})
tmp = tempfile.mkdtemp()
split = os.path.join(tmp, "split.json")
json.dump({"filenames": ["x/%08d.jpg" % i for i in range(n_samples + 1)]}, open(split, "w"))
if EVAL:
    # synthetic ground truth: a UV sphere near the hand surface (the analytic target, scaled and shifted so the ICP has work)
    gt_dir = os.path.join(tmp, "data", "obman", "test", "mesh_hand")
    os.makedirs(gt_dir)
    nu, nv = 96, 48
    th, ph = np.meshgrid(np.arange(nu) * 2 * np.pi / nu, (np.arange(nv) + 0.5) * np.pi / nv, indexing="ij")
    P = np.stack([np.sin(ph) * np.cos(th), np.sin(ph) * np.sin(th), np.cos(ph)], -1).reshape(-1, 3)
    P = (P * 0.35 + np.array([-0.25, 0, 0])) * 1.08 + np.array([0.03, -0.02, 0.015])
    idx = lambda i, j: (i % nu) * nv + j
    F = [(idx(i, j), idx(i + 1, j), idx(i + 1, j + 1)) for i in range(nu) for j in range(nv - 1)] + \
        [(idx(i, j), idx(i + 1, j + 1), idx(i, j + 1)) for i in range(nu) for j in range(nv - 1)]
    for i in range(n_samples + 1):
        with open(os.path.join(gt_dir, "%08d.obj" % i), "w") as f:
            f.write("".join("v %.6f %.6f %.6f\n" % tuple(p) for p in P) + "".join("f %d %d %d\n" % (a + 1, b + 1, c + 1) for a, b, c in F))
kw = dict(eval_mode=True, data_root=os.path.join(tmp, "data")) if EVAL else {}
kw["code_source"] = rc.synthetic_code_source("nerf3")
rc.reconstruct(dec, specs, split, tmp, 0, 1, cube_dim=N, **kw)          # warm-up
REPS = int(os.environ.get('ASDF_TIMING_REPS', '5'))      # the boxes are shared hosts: the worker threads (and the main one) get descheduled now and then - median of 5 runs
if os.environ.get("ASDF_TIMING_PROFILE"):          # host profile of one timed call (cProfile roughly doubles the interpreter's share)
    import cProfile, pstats, io
    pr = cProfile.Profile()
    pr.enable()
    rc.reconstruct(dec, specs, split, tmp, 1, n_samples + 1, cube_dim=N, **kw)
    torch.cuda.synchronize()
    pr.disable()
    for key, pat in (("tottime", None), ("cumulative", "alignsdf_amd")):
        out = io.StringIO()
        st = pstats.Stats(pr, stream=out).sort_stats(key)
        st.print_stats(pat, 40) if pat else st.print_stats(40)
        print(out.getvalue())
    sys.exit(0)
runs = []
for _ in range(REPS):
    torch.cuda.synchronize()
    t = time.perf_counter()
    recs = rc.reconstruct(dec, specs, split, tmp, 1, n_samples + 1, cube_dim=N, **kw)
    torch.cuda.synchronize()
    runs.append(time.perf_counter() - t)
dt = float(np.median(runs))
print("reconstruct(%s) with PLY export: %d samples, %.1f ms/sample (N=%d; median of %d runs: %s), F_hand %d F_obj %d" % (
    "eval_mode" if EVAL else "", n_samples, 1e3 * dt / n_samples, N, REPS, " ".join("%.1f" % (1e3 * r / n_samples) for r in runs), recs[-1]["F_hand"], recs[-1]["F_obj"]))
if os.environ.get("ASDF_TIMING_FLOW_ONLY"):          # (tools/trace_eval_flow.sh: the trace then ends with the timed flow)
    sys.exit(0)
if EVAL:
    from alignsdf_amd import icp
    from alignsdf_amd.ply import read_ply
    print("icp scale / trans of the last sample:", recs[-1]["icp_scale"], recs[-1]["icp_trans"])
    gv, gf = icp.load_obj(os.path.join(gt_dir, "%08d.obj" % 1))
    hv, hf = read_ply(os.path.join(tmp, "meshes", "%08d_obj.ply" % 1))
    src, tgt = icp.sample_surface(np.asarray(hv, np.float64), hf, 30000, 0), icp.sample_surface(gv, gf, 30000, 1)
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = icp.icp_trans_scale(src, tgt, src)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    t = float(np.median(ts))
    print("stand-alone ICP 30k x 30k: %d iterations, %.1f ms total, %.2f ms / iteration (median of 5: %s)" % (
        out["iterations"], 1e3 * t, 1e3 * t / out["iterations"], " ".join("%.1f" % (1e3 * x) for x in ts)))
# the GPU-side reference of the same configuration: the product's sample pipeline without files (what bench.py times)
def pipeline_ms(n):
    src = rc.synthetic_code_source("nerf3")
    items = [(i,) + src("s", i) for i in range(1, n + 1)]
    list(rc.pipelined_two_pass(dec, specs, iter(items[:2]), N))
    out = []
    for _ in range(REPS):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        list(rc.pipelined_two_pass(dec, specs, iter(items), N))
        torch.cuda.synchronize()
        out.append(1e3 * (time.perf_counter() - t0) / n)
    return float(np.median(out))
base = pipeline_ms(n_samples)
print("sample pipeline without files (bench.py's step): %.1f ms/sample -> file flow / pipeline = %.2f" % (base, 1e3 * dt / n_samples / base))
# host tail breakdown on the last sample's hand mesh
from alignsdf_amd.ply import read_ply
r = next(iter(rc.pipelined_two_pass(dec, specs, [(0, torch.from_numpy(syn.latent_code(1)).cuda(), None, None)], N)))[1]
torch.cuda.synchronize()
t0 = time.perf_counter(); v, f, mp = mu.place_vertices(r["verts_hand"], r["faces_hand"], r["origin"], r["voxel_size"]); t1 = time.perf_counter()
ov, of = mesh_post.keep_largest_component(mp, f); t2 = time.perf_counter()
from alignsdf_amd.ply import write_ply
write_ply(os.path.join(tmp, "t.ply"), ov, of); t3 = time.perf_counter()
print("host tail per surface (V=%d F=%d): D2H+place %.1f ms, largest component %.1f ms, PLY write %.1f ms" % (len(v), len(f), 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2)))
