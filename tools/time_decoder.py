"""Wall-clock one asdf_decode_grid pass (both heads) at several grid sizes on cuda:0."""
import sys
import time

import torch

sys.path.insert(0, ".")
from alignsdf_amd import synthetic as syn
from alignsdf_amd.hip_decoder import HipSdfDecoder

tag = sys.argv[1] if len(sys.argv) > 1 else "nerf3"
specs = syn.specs_for(tag)
dec = HipSdfDecoder(syn.full_state_dict(tag), 256, specs["PointFeatSize"], specs["EncodeStyle"], device="cuda:0")
emb = None
if tag == "both9":
    from alignsdf_amd.hip_decoder import kinematic_affine
    m, o = syn.pose_inputs(0)
    emb = kinematic_affine(9, "both", specs["SdfScaleFactor"], {k: torch.from_numpy(v) for k, v in m.items()},
                           {k: torch.from_numpy(v) for k, v in o.items()})
dec.set_sample(torch.from_numpy(syn.latent_code(0)), emb)
for N in [int(a) for a in sys.argv[2:]] or [64, 128, 256]:
    for it in range(2):
        torch.cuda.synchronize()
        t = time.time()
        h, o, b = dec.decode_grid(N, [-1, -1, -1], 2.0 / (N - 1))
        torch.cuda.synchronize()
        dt = time.time() - t
        ex = N ** 3 * 2 * 1057792 / dt / 1e12
        print("%s N=%d pass %.4f s  alg %.1f TF/s  exec %.1f TF/s (%.1f%% of 157.3)  neg=%s" % (
            tag, N, dt, N ** 3 * 2 * 1573888 / dt / 1e12, ex, ex / 1.573, (int(b[6]), int(b[14]))), flush=True)
