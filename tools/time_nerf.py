import sys, time, torch
sys.path.insert(0, ".")
from alignsdf_amd import synthetic as syn
from alignsdf_amd.hip_decoder import HipSdfDecoder
dec = HipSdfDecoder(syn.full_state_dict("nerf9"), 256, 9, "nerf")
dec.set_sample(torch.from_numpy(syn.latent_code(0)).cuda())
N = 256
for math in ("f32", "f16x3"):
    dec.set_math(math)
    dec.decode_grid(N, [-1, -1, -1], 2.0 / (N - 1))
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(3):
        dec.decode_grid(N, [-1, -1, -1], 2.0 / (N - 1))
    torch.cuda.synchronize()
    print("nerf9", math, "N=256 pass %.1f ms" % (1e3 * (time.perf_counter() - t) / 3))
