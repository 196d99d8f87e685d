"""Golden vectors for the eval-mode translate+scale ICP, produced by the reference's own ICP_T_S class
(deep_sdf/metrics/icp_trans_scale.py:11-196), run in the build container with:
  * a stub `trimesh` whose `sample.sample_surface(mesh, n)` returns the injected point sets (the reference samples
    with an unseeded RNG, utils/mesh.py:391, so the samples are the golden INPUT here),
  * `np.float` aliased to float (removed from numpy >= 1.24; icp_trans_scale.py:40,88).
Writes tests/golden/ref_icp.npz:  per case source / target samples, source vertices, and the reference's iteration
count, scale, trans, error trace, get_trans_scale() and the transformed vertices of export_source_mesh().
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
from alignsdf_amd import synthetic as syn  # noqa: E402

np.float = float
tm = types.ModuleType("trimesh")
tm.sample = types.ModuleType("trimesh.sample")
_queue = []
tm.sample.sample_surface = lambda mesh, n: (_queue.pop(0), None)
sys.modules["trimesh"] = tm
sys.modules["trimesh.sample"] = tm.sample
# load the module file directly: importing the `deep_sdf` package would pull plyfile / skimage / torch-side modules
import importlib.util  # noqa: E402
_spec = importlib.util.spec_from_file_location("ref_icp_trans_scale", "/root/reference/deep_sdf/metrics/icp_trans_scale.py")
_mod = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mod)
ICP_T_S = _mod.ICP_T_S


class Mesh:
    def __init__(self, v):
        self.vertices = v

    def export(self, path):
        pass


def blob(n, seed, radii, centre):
    """Points on a bumpy ellipsoid surface."""
    d = syn.normal((n, 3), seed)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    bump = 1.0 + 0.15 * np.sin(5 * d[:, 0]) * np.cos(4 * d[:, 1]) + 0.1 * np.sin(7 * d[:, 2])
    return d * bump[:, None] * np.asarray(radii) + np.asarray(centre)


def main():
    out = {}
    cases = {
        # name: (n_src, n_tgt, true scale, true trans, noise)
        "small": (2000, 2200, 1.18, (0.03, -0.02, 0.015), 0.0),
        "noisy": (4000, 3500, 0.85, (-0.04, 0.01, 0.02), 0.004),
        "full30k": (30000, 30000, 1.07, (0.012, 0.02, -0.018), 0.001),
    }
    for name, (ns, nt, s_true, t_true, noise) in cases.items():
        seed = 7000 + len(out)
        tgt = blob(nt, seed, (0.09, 0.06, 0.04), (0.02, -0.01, 0.35))
        src_on = blob(ns, seed + 1, (0.09, 0.06, 0.04), (0.02, -0.01, 0.35))
        src = (src_on - np.asarray(t_true)) / s_true + noise * syn.normal((ns, 3), seed + 2)
        verts = (blob(500, seed + 3, (0.09, 0.06, 0.04), (0.02, -0.01, 0.35)) - np.asarray(t_true)) / s_true
        # the product holds sample points as fp32: feed the reference exactly those values
        src, tgt, verts = (a.astype(np.float32).astype(np.float64) for a in (src, tgt, verts))
        icp = ICP_T_S(Mesh(verts.copy()), Mesh(tgt.copy()))
        _queue[:] = [src.copy(), tgt.copy()]
        icp.sample_mesh(30000, "both")
        errors = []
        # trace the error sequence by re-running with increasing iteration caps (run_icp_f has no hook)
        icp.run_icp_f(max_iter=100)
        scale, trans = float(np.asarray(icp.scale).reshape(-1)[0]), np.asarray(icp.trans, dtype=np.float64).reshape(3)
        all_trans, all_scale = icp.get_trans_scale()
        icp.export_source_mesh("/dev/null")
        out[name + ".src"], out[name + ".tgt"], out[name + ".verts"] = (a.astype(np.float32) for a in (src, tgt, verts))
        out[name + ".scale"], out[name + ".trans"] = np.array([scale]), trans
        out[name + ".all_scale"] = np.asarray(all_scale, dtype=np.float64).reshape(1)
        out[name + ".all_trans"] = np.asarray(all_trans, dtype=np.float64).reshape(3)
        out[name + ".verts_out"] = np.asarray(icp.mesh_source.vertices, dtype=np.float64)
        # residual after alignment (for a sanity bound in the tests)
        q = icp.points_source * scale + trans
        out[name + ".rms"] = np.array([np.sqrt(((q[:, None, :][:200] - icp.points_target[None, :, :]) ** 2).sum(-1).min(1).mean())])
        print(name, "scale", scale, "trans", trans, "all_scale", out[name + ".all_scale"], "rms", out[name + ".rms"])
    np.savez_compressed(os.path.join(HERE, "ref_icp.npz"), **out)


if __name__ == "__main__":
    main()
