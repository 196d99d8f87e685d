"""First look at the split-half kernel: agreement with the fp32 kernel / goldens / oracle, and its speed."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from alignsdf_amd import synthetic as syn
from alignsdf_amd.networks.model import build_decoder
from alignsdf_amd.utils.utils import hip_decoder_for, sample_embedding

for tag in ("nerf3", "both9", "comb3"):
    specs = syn.specs_for(tag)
    dec = build_decoder(specs, {k: torch.from_numpy(v) for k, v in syn.full_state_dict(tag).items()})
    hip = hip_decoder_for(dec)
    lat = torch.from_numpy(syn.latent_code(0)).cuda()
    mano = obj = None
    if tag == "both9":
        m, o = syn.pose_inputs(0)
        mano = {k: torch.from_numpy(v).cuda() for k, v in m.items()}; obj = {k: torch.from_numpy(v).cuda() for k, v in o.items()}
    hip.set_sample(lat, sample_embedding(specs, mano, obj, hip.combined))
    g = np.load("tests/golden/ref_decoder_%s.npz" % tag)
    pts = torch.from_numpy(g["rand_pts"]).cuda()
    out = {}
    for math in ("f32", "f16x3"):
        hip.set_math(math)
        h, o = hip.decode_points(pts)
        out[math] = (h.cpu().numpy(), o.cpu().numpy())
        print(tag, math, "vs reference golden: hand %.3e obj %.3e" % (np.abs(out[math][0] - g["rand_hand"]).max(), np.abs(out[math][1] - g["rand_obj"]).max()))
    print(tag, "f16x3 vs f32: %.3e %.3e" % (np.abs(out["f16x3"][0] - out["f32"][0]).max(), np.abs(out["f16x3"][1] - out["f32"][1]).max()))
    if tag == "nerf3":
        for math in ("f32", "f16x3"):
            hip.set_math(math)
            for N in (128, 256):
                hip.decode_grid(N, [-1, -1, -1], 2.0 / (N - 1))
                torch.cuda.synchronize(); t = time.perf_counter()
                for _ in range(3):
                    vh, vo, bb = hip.decode_grid(N, [-1, -1, -1], 2.0 / (N - 1))
                torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
                print(math, "N=%d pass: %.2f ms  (bbox %s)" % (N, 1e3 * dt, bb.cpu().numpy()[[0, 1, 2, 3, 4, 5, 6, 14]].tolist()))
