"""NeRF-encoded decoders (PointFeatSize 9 / 15, EncodeStyle "nerf"; or the tags named on the command line, e.g. comb3) through the sample pipeline at N = 256: ms per sample under the
product's default (audited one-plane) sweeps against ordinary sweeps, and whether every mesh is the same vertex for vertex.
    gpurun -- 'python tools/nerf_one_plane_timing.py > gpurun_out/r4/nerf_one_plane.txt'
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alignsdf_amd import synthetic as syn                                  # noqa: E402
from alignsdf_amd.networks.model import build_decoder                       # noqa: E402
from alignsdf_amd.reconstruct import pipelined_two_pass, synthetic_code_source   # noqa: E402
from alignsdf_amd.utils.utils import decoder_for                            # noqa: E402

N, SAMPLES, WARM = 256, 24, 4


def run(tag, mode, math="f16x3"):
    for k in ("ASDF_COARSE", "ASDF_FINE"):
        if mode == "exact":
            os.environ[k] = "exact"
        else:
            os.environ.pop(k, None)
    specs = syn.specs_for(tag)
    src = synthetic_code_source(tag, "cuda")
    dec = build_decoder(specs, {k: torch.from_numpy(v) for k, v in syn.full_state_dict(tag).items()})
    samples = [(i,) + src("s%d" % i, i) for i in range(SAMPLES + WARM)]
    hip = decoder_for(dec, specs, samples[0][2])
    if hip.math != math:
        hip.set_math(math)
    meshes, t0 = {}, None
    for n, (k, r) in enumerate(pipelined_two_pass(dec, specs, iter(samples), N)):
        if n == WARM - 1:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        meshes[k] = {p: (r["verts_" + p].cpu(), r["faces_" + p].cpu()) for p in ("hand", "obj") if "verts_" + p in r}
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / SAMPLES
    return ms, meshes, dict(box=dict(hip.box_stats), band=dict(hip.band_stats)), hip.math


for tag in ([a for a in sys.argv[1:] if not a.startswith("-")] or ["nerf9", "nerf15"]):
    ms_e, m_e, _, math = run(tag, "exact")
    ms_d, m_d, st, _ = run(tag, "default")
    same = sum(int(torch.equal(m_e[k][p][0], m_d[k][p][0]) and torch.equal(m_e[k][p][1], m_d[k][p][1])) for k in m_e for p in m_e[k])
    total = sum(len(m_e[k]) for k in m_e)
    print("%s (%s) N=%d: ordinary sweeps %.2f ms/sample, default sweeps %.2f ms/sample (%.2fx); meshes identical %d of %d; one-plane "
          "coarse sweeps %d (refused %d), fine %d (refused %d)" % (tag, math, N, ms_e, ms_d, ms_e / ms_d, same, total, st["box"]["box"],
                                                                     st["box"]["fallback"], st["band"]["band"], st["band"]["fallback"]), flush=True)
    ms_f, m_f, _, _ = run(tag, "exact", "f32")
    same = sum(int(torch.equal(m_f[k][p][1], m_d[k][p][1])) for k in m_f for p in m_f[k])
    print("%s fp32 chain N=%d: %.2f ms/sample; faces identical to the default sweeps' %d of %d" % (tag, N, ms_f, same, total), flush=True)
