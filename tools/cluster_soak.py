"""Soak of the short-list kernel's cluster form: many box sweeps with candidate lists of changing length (clusters come and go between
launches, the audit runs beside them), every 500th compared with the tile form.   python tools/cluster_soak.py [sweeps]"""
import ctypes, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from alignsdf_amd import _native
from alignsdf_amd import synthetic as syn
from alignsdf_amd.hip_decoder import HipSdfDecoder
from alignsdf_amd.utils.utils import sample_embedding

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
N = 64
specs = syn.specs_for("nerf3")
hip = HipSdfDecoder(syn.full_state_dict("nerf3"), 256, specs["PointFeatSize"], specs["EncodeStyle"])
lat, m, o = syn.sample_inputs("nerf3", 2)
hip.set_sample(torch.from_numpy(lat).cuda(), sample_embedding(specs, None, None, hip.combined))
origin, vs = [-1.0, -1.0, -1.0], 2.0 / (N - 1)
hip.decode_grid(N, origin, vs)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
org = (ctypes.c_float * 3)(*origin)
vh = torch.empty((N, N, N), dtype=torch.float32, device="cuda")
vo = torch.empty((N, N, N), dtype=torch.float32, device="cuda")
rec = torch.zeros(48, dtype=torch.int32, device="cuda")
taus = [3e-5, 2e-3, 1e-4, 6e-4, 1e-5, 1.5e-3, 3e-4, 5e-5]


def sweep(tau):
    _native.check(hip._L.asdf_decode_grid_box(hip._h, N, org, ctypes.c_float(vs), 0, ctypes.c_float(tau), vh.data_ptr(), vo.data_ptr(),
                                              rec.data_ptr(), st), "asdf_decode_grid_box")


t0 = time.perf_counter()
checked = 0
for i in range(n):
    tau = taus[i % len(taus)]
    sweep(tau)
    if i % 500 == 499:
        a = (vh.clone(), vo.clone(), rec.cpu().numpy().copy())
        _native.check(hip._L.asdf_decoder_set_cluster_list(hip._h, 0), "set_cluster_list")
        _native.check(hip._L.asdf_decoder_set_short_list(hip._h, 0), "set_short_list")
        hip.set_audit(hip.audit_voxels, seed=i)
        sweep(tau)
        b = (vh.clone(), vo.clone(), rec.cpu().numpy().copy())
        _native.check(hip._L.asdf_decoder_set_short_list(hip._h, 8192), "set_short_list")
        _native.check(hip._L.asdf_decoder_set_cluster_list(hip._h, 2048), "set_cluster_list")
        hip.set_audit(hip.audit_voxels, seed=i)
        sweep(tau)
        c = (vh.clone(), vo.clone(), rec.cpu().numpy().copy())
        assert torch.equal(b[0], c[0]) and torch.equal(b[1], c[1]) and np.array_equal(b[2][:16], c[2][:16]) and int(b[2][32]) == int(c[2][32]), i
        checked += 1
torch.cuda.synchronize()
dt = time.perf_counter() - t0
fault = int(hip._status(clear=False)[11])          # round 6: the cluster form's sticky report of a member that did not arrive within its bound
print("%d box sweeps at N = %d in %.1f s (%.1f us each), %d comparisons with the tile form equal; last candidate count %d; cluster fault word %d" % (
    n, N, dt, 1e6 * dt / n, checked, int(rec.cpu()[32]), fault))
assert fault == 0, "a member of a cluster did not arrive within the default bound during the soak"
