"""Per-iteration time of the translate + scale ICP (K7) on two 30k-sample sets, grid search: python tools/time_icp.py"""
import ctypes, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from alignsdf_amd import _native, icp
from alignsdf_amd import synthetic as syn

rng = np.random.default_rng(3)
# a hand-sized closed surface: an ellipsoid with bumps, sampled like the product does
nu, nv = 160, 80
th, ph = np.meshgrid(np.arange(nu) * 2 * np.pi / nu, (np.arange(nv) + 0.5) * np.pi / nv, indexing="ij")
rad = 0.3 + 0.05 * np.sin(3 * th) * np.sin(2 * ph)
P = np.stack([rad * np.sin(ph) * np.cos(th), 0.6 * rad * np.sin(ph) * np.sin(th), 1.4 * rad * np.cos(ph)], -1).reshape(-1, 3)
idx = lambda i, j: (i % nu) * nv + j
F = np.array([(idx(i, j), idx(i + 1, j), idx(i + 1, j + 1)) for i in range(nu) for j in range(nv - 1)] +
             [(idx(i, j), idx(i + 1, j + 1), idx(i, j + 1)) for i in range(nu) for j in range(nv - 1)])
src = icp.sample_surface(P, F, 30000, 0)
tgt = icp.sample_surface(P * 1.08 + np.array([0.03, -0.02, 0.015]), F, 30000, 1)
ps, _ = icp.normalise_source(src, tgt)
for _ in range(3):
    icp.run_icp_f(ps, tgt)
ts = []
for _ in range(20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s, t, iters, err = icp.run_icp_f(ps, tgt)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print("lanes %s: %d iterations, median %.3f ms per run = %.3f ms per iteration (min %.3f ms per run); scale %.6f error %.3e" % (
    os.environ.get("ASDF_ICP_GRID_LANES", "8"), iters, 1e3 * np.median(ts), 1e3 * np.median(ts) / iters, 1e3 * min(ts), s, err))
