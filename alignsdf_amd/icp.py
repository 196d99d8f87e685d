"""Eval-mode translate + scale ICP (utils/mesh.py:385-395, deep_sdf/metrics/icp_trans_scale.py) on the GPU.

The reference aligns the predicted hand mesh to the ground-truth mesh before export: 30 000 area-weighted surface
samples per mesh (`trimesh.sample.sample_surface`, UNSEEDED in the reference - here a seeded counter-based generator,
so runs are reproducible), normalisation of the source samples onto the target's centroid / RMS radius, then up to
100 iterations of {nearest target of every source sample, nearest source of every target sample, 4-unknown least
squares}.  The iteration runs in libalignsdf_hip.so (asdf_icp_ts: fp64 brute-force nearest neighbours, K7); sampling,
the normalisation and the final vertex transform are O(30k) host numpy.
"""
import ctypes

import numpy as np
import torch

from . import _native, synthetic


def load_obj(path):
    """Minimal Wavefront OBJ reader: (verts float64 [V,3], faces int64 [F,3]); polygons are fan-triangulated."""
    verts, faces = [], []
    with open(path, "r") as f:
        for line in f:
            if line.startswith("v "):
                verts.append([float(x) for x in line.split()[1:4]])
            elif line.startswith("f "):
                idx = [int(tok.split("/")[0]) for tok in line.split()[1:]]
                idx = [i - 1 if i > 0 else len(verts) + i for i in idx]
                for k in range(1, len(idx) - 1):
                    faces.append([idx[0], idx[k], idx[k + 1]])
    return np.asarray(verts, dtype=np.float64).reshape(-1, 3), np.asarray(faces, dtype=np.int64).reshape(-1, 3)


def sample_surface(verts, faces, count, seed=0):
    """`count` area-weighted uniform samples of a triangle mesh (the scheme of trimesh.sample.sample_surface: pick faces
    by cumulative area, reflect barycentric pairs whose sum exceeds 1), driven by the repo's seeded generator."""
    v = np.asarray(verts, dtype=np.float64)
    f = np.asarray(faces, dtype=np.int64)
    # areas of all faces (190 k per hand surface at N = 256) on column vectors: the same products, differences and sums as
    # 0.5 * norm(cross(b - a, c - a)) - bit for bit - at a third of the time of the [F, 3] gathers + np.cross (this runs on
    # the host between two decoder passes of the sample pipeline)
    vx, vy, vz = np.ascontiguousarray(v[:, 0]), np.ascontiguousarray(v[:, 1]), np.ascontiguousarray(v[:, 2])
    i0, i1, i2 = f[:, 0], f[:, 1], f[:, 2]
    ax, ay, az = vx[i0], vy[i0], vz[i0]
    e1x, e1y, e1z = vx[i1] - ax, vy[i1] - ay, vz[i1] - az
    e2x, e2y, e2z = vx[i2] - ax, vy[i2] - ay, vz[i2] - az
    cx, cy, cz = e1y * e2z - e1z * e2y, e1z * e2x - e1x * e2z, e1x * e2y - e1y * e2x
    area = 0.5 * np.sqrt(cx * cx + cy * cy + cz * cz)
    cum = np.cumsum(area)
    pick = np.searchsorted(cum, synthetic.uniform((count,), 9100 + seed) * cum[-1])
    pick = np.minimum(pick, len(f) - 1)
    r = synthetic.uniform((count, 2), 9200 + seed)
    flip = r.sum(1) > 1.0
    r[flip] = np.abs(r[flip] - 1.0)
    a, b, c = v[f[pick, 0]], v[f[pick, 1]], v[f[pick, 2]]          # only the picked faces
    return a + (b - a) * r[:, :1] + (c - a) * r[:, 1:]


def normalise_source(points_source, points_target):
    """ICP_T_S.sample_mesh's normalisation (icp_trans_scale.py:25-31)."""
    ps, pt = np.asarray(points_source, np.float64), np.asarray(points_target, np.float64)
    offset_s = ps.mean(0)
    scale_s = np.sqrt(((ps - offset_s) ** 2).sum() / len(ps))
    offset_t = pt.mean(0)
    scale_t = np.sqrt(((pt - offset_t) ** 2).sum() / len(pt))
    return (ps - offset_s) / scale_s * scale_t + offset_t, (offset_s, scale_s, offset_t, scale_t)


def run_icp_f(points_source, points_target, max_iter=100, stop_error=1e-3, stop_improvement=1e-5, device="cuda"):
    """ICP_T_S.run_icp_f on normalised source samples, on the GPU.  Returns (scale, trans [3], iterations, last error)."""
    dev = torch.device(device)
    src = torch.as_tensor(np.ascontiguousarray(points_source, dtype=np.float64)).to(dev)
    tgt = torch.as_tensor(np.ascontiguousarray(points_target, dtype=np.float64)).to(dev)
    L = _native.lib()
    nbytes = ctypes.c_size_t()
    _native.check(L.asdf_icp_workspace_bytes(src.shape[0], tgt.shape[0], ctypes.byref(nbytes)), "asdf_icp_workspace_bytes")
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
    res = (ctypes.c_double * 8)()
    with torch.cuda.device(dev):
        _native.check(L.asdf_icp_ts(src.data_ptr(), src.shape[0], tgt.data_ptr(), tgt.shape[0], int(max_iter), float(stop_error),
                                    float(stop_improvement), ws.data_ptr(), ws.numel(), res,
                                    ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "asdf_icp_ts")
    return res[0], np.array([res[1], res[2], res[3]]), int(res[4]), res[5]


def icp_trans_scale(points_source, points_target, vertices, max_iter=100, device="cuda"):
    """sample normalisation + run_icp_f + get_trans_scale (:188-191) + the vertex transform of export_source_mesh
    (:193-196).  Returns a dict with scale, trans, iterations, error, all_scale, all_trans, vertices."""
    return finish_icp(start_icp(points_source, points_target, max_iter, device), vertices)


class IcpJob:
    """An ICP run in flight on the device (start_icp); finish_icp waits for it."""
    __slots__ = ("src", "tgt", "ws", "host", "norm", "stream", "device", "done", "result")





def start_icp(points_source, points_target, max_iter=100, device="cuda", stop_error=1e-3, stop_improvement=1e-5):
    """Normalise the source samples and enqueue the whole ICP on the current stream without synchronising: pinned
    staging + asynchronous uploads + asdf_icp_ts_enqueue.  The caller may queue other work behind it."""
    dev = torch.device(device)
    ps, norm = normalise_source(points_source, points_target)
    job = IcpJob()
    job.norm, job.device = norm, dev
    job.stream = torch.cuda.current_stream(dev)
    job.host = [torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).pin_memory() for a in (ps, points_target)]
    job.src, job.tgt = (h.to(dev, non_blocking=True) for h in job.host)
    L = _native.lib()
    nbytes = ctypes.c_size_t()
    _native.check(L.asdf_icp_workspace_bytes(job.src.shape[0], job.tgt.shape[0], ctypes.byref(nbytes)), "asdf_icp_workspace_bytes")
    job.ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
    # the run publishes its outcome into pinned (device-accessible) host memory itself: any read-back copy would be a
    # blit kernel, which cannot start while a decoder pass (one wave per SIMD holding the whole register file) is on
    # the machine - the caller would wait for whatever it queued behind the ICP
    job.result = torch.zeros(8, dtype=torch.float64).pin_memory()
    with torch.cuda.device(dev):
        _native.check(L.asdf_icp_ts_enqueue(job.src.data_ptr(), job.src.shape[0], job.tgt.data_ptr(), job.tgt.shape[0], int(max_iter),
                                            float(stop_error), float(stop_improvement), job.ws.data_ptr(), job.ws.numel(),
                                            job.result.data_ptr(), ctypes.c_void_p(job.stream.cuda_stream)), "asdf_icp_ts_enqueue")
        job.done = torch.cuda.Event()
        job.done.record(job.stream)
    return job


def finish_icp(job, vertices):
    """Wait for a start_icp job; returns the dict of icp_trans_scale for `vertices`."""
    job.done.synchronize()            # the ICP only - not whatever was queued behind it
    res = job.result.tolist()
    scale, trans, iters, error = res[0], np.array([res[1], res[2], res[3]]), int(res[4]), res[5]
    offset_s, scale_s, offset_t, scale_t = job.norm
    v = (np.asarray(vertices, np.float64) - offset_s) / scale_s * scale_t + offset_t
    return dict(scale=scale, trans=trans, iterations=iters, error=error, all_scale=scale_t * scale / scale_s,
                all_trans=trans + offset_t * scale - offset_s * scale_t * scale / scale_s, vertices=v * scale + trans)


def start_alignment(verts, faces, gt_verts, gt_faces, samples=30000, max_iter=100, seed=0, device="cuda"):
    """First half of the eval-mode block of utils/mesh.py:385-395: sample both meshes, enqueue the ICP."""
    ps = sample_surface(verts, faces, samples, seed)
    pt = sample_surface(gt_verts, gt_faces, samples, seed + 1)
    return start_icp(ps, pt, max_iter, device)


def align_to_ground_truth(verts, faces, gt_verts, gt_faces, samples=30000, max_iter=100, seed=0, device="cuda"):
    """The eval-mode block of utils/mesh.py:385-395: sample both meshes, ICP, return (aligned verts, trans, scale)."""
    r = finish_icp(start_alignment(verts, faces, gt_verts, gt_faces, samples, max_iter, seed, device), verts)
    return r["vertices"], r["all_trans"], r["all_scale"], r
