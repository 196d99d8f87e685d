"""Time of the box-only coarse sweep against the ordinary one at N = 256 (events on the stream, whole call chains)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alignsdf_amd import synthetic as syn
from alignsdf_amd.hip_decoder import HipSdfDecoder
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
specs = syn.specs_for("nerf3")
hip = HipSdfDecoder(syn.full_state_dict("nerf3"), 256, 3, "nerf")
hip.coarse_mode = "box"
vs = 2.0 / (N - 1)
for sample in range(6):
    hip.set_sample(torch.from_numpy(syn.latent_code(sample)).cuda())
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    b = hip.coarse_finish(hip.coarse_begin(N, [-1.0, -1.0, -1.0], vs))
    e[1].record()
    w = hip.decode_grid(N, [-1.0, -1.0, -1.0], vs)[2].cpu().numpy()
    e[2].record(); e[2].synchronize()
    print("sample %d: coarse %.2f ms, ordinary %.2f ms, boxes equal %s, stats %s tau %.3g" % (
        sample, e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2]), list(b[:6]) == list(w[:6]) and list(b[8:14]) == list(w[8:14]), hip.box_stats, hip._box_tau or 0), flush=True)
