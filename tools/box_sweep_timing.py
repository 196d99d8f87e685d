"""Untraced timing of asdf_decode_grid_box / the band sweep on a small lattice: audit beside the candidates or in line
(ASDF_AUDIT_INLINE=1), short-list kernel in its cluster form or not.   python tools/box_sweep_timing.py [N] [hand|both]"""
import ctypes, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from alignsdf_amd import _native
from alignsdf_amd import synthetic as syn
from alignsdf_amd.hip_decoder import HipSdfDecoder
from alignsdf_amd.utils.utils import sample_embedding

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
both = (sys.argv[2] if len(sys.argv) > 2 else "hand") == "both"
specs = syn.specs_for("nerf3")
hip = HipSdfDecoder(syn.full_state_dict("nerf3"), 256, specs["PointFeatSize"], specs["EncodeStyle"])
lat, m, o = syn.sample_inputs("nerf3", 1)
hip.set_sample(torch.from_numpy(lat).cuda(), sample_embedding(specs, None, None, hip.combined))
origin, vs = [-1.0, -1.0, -1.0], 2.0 / (N - 1)
hip.decode_grid(N, origin, vs)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
org = (ctypes.c_float * 3)(*origin)
vh = torch.empty((N, N, N), dtype=torch.float32, device="cuda")
vo = torch.empty((N, N, N), dtype=torch.float32, device="cuda") if both else None
rec = torch.zeros(48, dtype=torch.int32, device="cuda")
tau = 4e-4
AUDIT = hip.audit_voxels


def run(kind, reps=200):
    fn = hip._L.asdf_decode_grid_box if kind == "box" else hip._L.asdf_decode_grid_band
    for _ in range(10):
        _native.check(fn(hip._h, N, org, ctypes.c_float(vs), 0, ctypes.c_float(tau), vh.data_ptr(), vo.data_ptr() if both else None, rec.data_ptr(), st), kind)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        _native.check(fn(hip._h, N, org, ctypes.c_float(vs), 0, ctypes.c_float(tau), vh.data_ptr(), vo.data_ptr() if both else None, rec.data_ptr(), st), kind)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6, rec.cpu().numpy()


for cluster in (2048, 0):
    _native.check(hip._L.asdf_decoder_set_cluster_list(hip._h, cluster), "set_cluster_list")
    for inline in (False, True):
        if inline:
            os.environ["ASDF_AUDIT_INLINE"] = "1"
        else:
            os.environ.pop("ASDF_AUDIT_INLINE", None)
        for audit in (AUDIT, 0):
            hip.set_audit(audit)
            us, r = run("box")
            usb, rb = run("band")
            print("N=%d %s cluster<=%d audit %s %-7s: box sweep %.1f us (candidates %d)   band sweep %.1f us (near-level %d, band %d)" % (
                N, "both" if both else "hand", cluster, audit, "in line" if inline else "beside", us, int(r[32]), usb, int(rb[32]), int(rb[33])))
