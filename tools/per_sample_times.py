"""Wall-clock time between consecutive samples leaving the product's sample pipeline (pipelined_two_pass) on a small lattice: where the
periodic whole-lattice comparisons and the step-by-step samples around them show.   python tools/per_sample_times.py [N] [hand|both] [samples]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from alignsdf_amd import synthetic as syn
from alignsdf_amd.hip_decoder import HipSdfDecoder
from alignsdf_amd.reconstruct import pipelined_two_pass

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
hand_only = (sys.argv[2] if len(sys.argv) > 2 else "hand") == "hand"
count = int(sys.argv[3]) if len(sys.argv) > 3 else 300
specs = dict(syn.specs_for("nerf3"), ObjectBranch=not hand_only)
dec = HipSdfDecoder(syn.full_state_dict("nerf3"), 256, specs["PointFeatSize"], specs["EncodeStyle"], device=torch.device("cuda:0"))
codes = []
for s in range(64):
    lat, m, o = syn.sample_inputs("nerf3", s)
    codes.append((torch.from_numpy(lat).cuda(), None, None))


def stream(n):
    for i in range(n):
        yield (i,) + codes[i % 64]


t, last = [], time.perf_counter()
for i, r in pipelined_two_pass(dec, specs, stream(count), N):
    now = time.perf_counter()
    t.append((now - last) * 1e3)
    last = now
torch.cuda.synchronize()
t = np.array(t)
print("N=%d %s: %d samples, median %.3f ms, mean %.3f ms, mean of samples 8.. %.3f ms" % (N, "hand" if hand_only else "both", count, np.median(t), t.mean(), t[8:].mean()))
slow = [(i, round(float(x), 2)) for i, x in enumerate(t) if i >= 8 and x > 2.0 * np.median(t)]
print("samples slower than twice the median (index, ms):", slow)
print("their excess over the median, summed: %.2f ms = %.3f ms per sample over %d samples" % (sum(x - np.median(t) for _, x in slow), sum(x - np.median(t) for _, x in slow) / (count - 8), count - 8))
print("events:", {k: (v if not isinstance(v, list) else len(v)) for k, v in dec.events.items()}, "box", dict(dec.box_stats), "band", dict(dec.band_stats))
