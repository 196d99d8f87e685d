import sys, torch, numpy as np
sys.path.insert(0, ".")
from alignsdf_amd import synthetic as syn
from alignsdf_amd.hip_decoder import HipSdfDecoder
specs = syn.specs_for("nerf3")
dec = HipSdfDecoder(syn.full_state_dict("nerf3"), 256, 3, "nerf")
dec.set_sample(torch.from_numpy(syn.latent_code(0)).cuda())
N = 256
for label, kw in (("bbox", dict(want_bbox=True)), ("no bbox", dict(want_bbox=False))):
    dec.decode_grid(N, [-1, -1, -1], 2.0 / (N - 1), **kw)
    dec.event_log = []
    for _ in range(4):
        dec.decode_grid(N, [-1, -1, -1], 2.0 / (N - 1), **kw)
    torch.cuda.synchronize()
    print(label, ["%.2f" % a.elapsed_time(b) for a, b in dec.event_log])
    dec.event_log = None
