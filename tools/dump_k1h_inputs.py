"""Dump the split-half weight stream and the folded per-sample constants of the synthetic decoder (what K1h reads in the
product) for tools/k1h_ablate.hip:  python tools/dump_k1h_inputs.py [tag]  ->  tools/bin/k1h_<tag>.bin
Layout: stream16 (uint16, 2 heads x 128 x 8192) followed by cst16 (float32, 2 x 6920)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from alignsdf_amd import synthetic as syn  # noqa: E402
import kernel_emulator as ke  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "nerf3"
specs = syn.specs_for(tag)
pk = ke.pack_host(syn.full_state_dict(tag), specs["PointFeatSize"], specs["EncodeStyle"])
cst16 = ke.fold16(pk, syn.latent_code(0).reshape(-1))
out = os.path.join(ROOT, "tools", "bin", "k1h_%s.bin" % tag)
os.makedirs(os.path.dirname(out), exist_ok=True)
with open(out, "wb") as f:
    f.write(pk["stream16"].tobytes())
    f.write(np.asarray(cst16, np.float32).tobytes())
print(out, pk["stream16"].size * 2, np.asarray(cst16).size * 4)
