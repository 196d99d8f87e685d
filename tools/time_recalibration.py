"""What the periodic whole-lattice re-calibration of the default sweeps costs (hip_decoder.RECAL_EVERY): wall time of a coarse pass that
re-measures the one-plane error on the whole lattice (ordinary sweep + one-plane sweep + statistics) against an ordinary default one.
    gpurun -- python tools/time_recalibration.py
"""
import sys, time, torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from alignsdf_amd import synthetic as syn, hip_decoder as hd
from alignsdf_amd.hip_decoder import HipSdfDecoder
hip = HipSdfDecoder(syn.full_state_dict("nerf3"), 256, 3, "nerf")
N = 256; vs = 2.0 / (N - 1)
def one(sample, force):
    hip.set_sample(torch.from_numpy(syn.latent_code(sample)).cuda(), None)
    if force: hip._coarse_since_cal = hd.RECAL_EVERY
    torch.cuda.synchronize(); t = time.perf_counter()
    hip.coarse_finish(hip.coarse_begin(N, [-1.0, -1.0, -1.0], vs))
    torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3
one(0, False)
for s in range(1, 4): print("ordinary coarse pass %.1f ms" % one(s, False))
for s in range(4, 8): print("recalibrating coarse pass %.1f ms" % one(s, True), hip.certificate()["calibrations"])
