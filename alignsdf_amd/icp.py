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

from . import _native


from .surface_sampling import _AREA_QUANTUM, _uniforms, load_obj, sample_surface  # noqa: E402,F401  (torch-free host part)


_uniform_cache = {}


def _draws(count, seed, dev):
    key = (count, seed, str(dev))
    if key not in _uniform_cache:
        u, r = _uniforms(count, seed)
        _uniform_cache[key] = (torch.from_numpy(u).to(dev), torch.from_numpy(np.ascontiguousarray(r)).to(dev))
    return _uniform_cache[key]


def sample_surface_native(verts_d, faces_d, count, seed=0, num_faces_dev=None, placement=None):
    """sample_surface on the device through K9 (csrc/surface_sample.hip: asdf_sample_surface - four launches, nothing waited for):
    verts_d [V,3] float32, faces_d [F,3] int32, both contiguous device tensors; `placement` = (voxel size, origin[3]) when the
    vertices are in lattice units (marching cubes' / K8's output: placed on the fly in the exporter's fp32 arithmetic), None when
    they are positions.  Returns [count,3] fp64, bit for bit the host sampler's points."""
    dev = verts_d.device
    assert verts_d.dtype == torch.float32 and faces_d.dtype == torch.int32 and verts_d.is_contiguous() and faces_d.is_contiguous()
    u, r = _draws(count, seed, dev)
    L = _native.lib()
    F = int(faces_d.shape[0])
    nbytes = ctypes.c_size_t()
    _native.check(L.asdf_sample_surface_workspace_bytes(F, ctypes.byref(nbytes)), "asdf_sample_surface_workspace_bytes")
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
    out = torch.empty((count, 3), dtype=torch.float64, device=dev)
    nf = None
    if num_faces_dev is not None:
        nf = num_faces_dev.reshape(-1)[:1]
        nf = nf if nf.dtype == torch.int32 else nf.to(torch.int32)
    vs, org = (np.float32(placement[0]), (ctypes.c_float * 3)(*[float(np.float32(o)) for o in placement[1]])) if placement is not None else (np.float32(0), None)
    with torch.cuda.device(dev):
        _native.check(L.asdf_sample_surface(verts_d.data_ptr(), faces_d.data_ptr(), F, nf.data_ptr() if nf is not None else None,
                                            1 if placement is not None else 0, ctypes.c_float(float(vs)), org, u.data_ptr(), r.data_ptr(), int(count),
                                            out.data_ptr(), ws.data_ptr(), ws.numel(), ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
                      "asdf_sample_surface")
    return out


def sample_surface_device(verts_d, faces_d, count, seed=0, num_faces_dev=None):
    """sample_surface on the device, bit for bit: verts_d [V,3] (any float dtype, converted to fp64 exactly), faces_d [F,3] integer
    device tensors -> [count,3] fp64 device tensor.  `num_faces_dev` (optional 0-dim / 1-element device tensor): only the first
    that many rows of faces_d are faces - the others get area zero and can never be picked - so a mesh whose size is known only on
    the device (the largest-component filter's output) is sampled without a host synchronisation.  float32 vertices with int32
    faces (what marching cubes and K8 produce) go through K9 (sample_surface_native); other types through the elementwise torch form
    below - fp64 products and sums in the host sampler's order (torch does not contract them), integer cumulative sums: same picks,
    same points (tests/test_gpu_icp.py compares the two and the host sampler)."""
    if (verts_d.dtype == torch.float32 and faces_d.dtype == torch.int32 and verts_d.is_contiguous() and faces_d.is_contiguous()
            and faces_d.shape[0] > 0):
        return sample_surface_native(verts_d, faces_d, count, seed, num_faces_dev)
    return sample_surface_torch(verts_d, faces_d, count, seed, num_faces_dev)


def sample_surface_torch(verts_d, faces_d, count, seed=0, num_faces_dev=None):
    """The elementwise torch form of sample_surface_device (about fifty launches)."""
    dev = verts_d.device
    v = verts_d.to(torch.float64)
    f = faces_d.to(torch.int64)
    vx, vy, vz = v[:, 0].contiguous(), v[:, 1].contiguous(), v[:, 2].contiguous()
    i0, i1, i2 = f[:, 0], f[:, 1], f[:, 2]
    if num_faces_dev is not None:
        live = torch.arange(f.shape[0], device=dev) < num_faces_dev.reshape(-1)[0].to(torch.int64)
        i0, i1, i2 = torch.where(live, i0, 0), torch.where(live, i1, 0), torch.where(live, i2, 0)      # (padding rows: a degenerate face)
    ax, ay, az = vx[i0], vy[i0], vz[i0]
    e1x, e1y, e1z = vx[i1] - ax, vy[i1] - ay, vz[i1] - az
    e2x, e2y, e2z = vx[i2] - ax, vy[i2] - ay, vz[i2] - az
    cx, cy, cz = e1y * e2z - e1z * e2y, e1z * e2x - e1x * e2z, e1x * e2y - e1y * e2x
    area = 0.5 * torch.sqrt(cx * cx + cy * cy + cz * cz)
    q = torch.round(area / area.max() * _AREA_QUANTUM).to(torch.int64)       # torch.round = rint (half to even)
    cum = torch.cumsum(q, 0)
    u, r = _draws(count, seed, dev)
    target = torch.floor(u * cum[-1].to(torch.float64)).to(torch.int64)
    pick = torch.clamp(torch.searchsorted(cum, target, right=True), max=f.shape[0] - 1)
    a = torch.stack([vx[i0[pick]], vy[i0[pick]], vz[i0[pick]]], 1)
    b = torch.stack([vx[i1[pick]], vy[i1[pick]], vz[i1[pick]]], 1)
    c = torch.stack([vx[i2[pick]], vy[i2[pick]], vz[i2[pick]]], 1)
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


FIRST_BATCH = 16      # iterations enqueued up front; the run is continued at finish time only if it has not converged by then


def _enqueue_range(job, first, last):
    max_iter, stop_error, stop_improvement = job.args
    with torch.cuda.device(job.device):
        _native.check(_native.lib().asdf_icp_ts_enqueue_range(
            job.src.data_ptr(), job.src.shape[0], job.tgt.data_ptr(), job.tgt.shape[0], int(first), int(last), float(stop_error),
            float(stop_improvement), job.ws.data_ptr(), job.ws.numel(), job.result.data_ptr(),
            ctypes.c_void_p(job.stream.cuda_stream)), "asdf_icp_ts_enqueue_range")


class IcpJob:
    """An ICP run in flight on the device (start_icp); finish_icp waits for it."""
    __slots__ = ("src", "tgt", "ws", "host", "norm", "stream", "device", "done", "result", "args")





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
    job.args = (int(max_iter), float(stop_error), float(stop_improvement))
    with torch.cuda.device(dev):
        _enqueue_range(job, 0, min(FIRST_BATCH, int(max_iter)))
        job.done = torch.cuda.Event()
        job.done.record(job.stream)
    return job


def start_icp_device(points_source_dev, points_target_dev, max_iter=100, stop_error=1e-3, stop_improvement=1e-5):
    """start_icp for sample sets that are ALREADY on the device (sample_surface_device): the normalisation of
    ICP_T_S.sample_mesh (icp_trans_scale.py:25-31) runs there too (fp64 reductions; the sums are associated differently from
    numpy's, a 1e-16-class difference in the statistics), its four statistics travel to pinned host memory behind the run, and
    nothing here waits for the device.  Both inputs [n,3] fp64 on the same device."""
    ps, pt = points_source_dev.contiguous(), points_target_dev.contiguous()
    dev = ps.device
    job = IcpJob()
    job.device = dev
    job.stream = torch.cuda.current_stream(dev)
    job.src = torch.empty_like(ps)
    job.tgt = pt
    # the four statistics go straight into pinned (device-accessible) host memory from the kernel that computes them (K9:
    # asdf_icp_normalise, one launch - it was about 25 elementwise / reduction launches and a copy); finish_icp reads them behind the run
    job.host = [torch.zeros(8, dtype=torch.float64).pin_memory()]
    with torch.cuda.device(dev):
        _native.check(_native.lib().asdf_icp_normalise(ps.data_ptr(), ps.shape[0], pt.data_ptr(), pt.shape[0], job.src.data_ptr(),
                                                       job.host[0].data_ptr(), ctypes.c_void_p(job.stream.cuda_stream)), "asdf_icp_normalise")
    job.host.append(ps)               # (keeps the un-normalised samples alive until the kernel has run)
    job.norm = None                   # read from job.host[0] once the run is done (finish_icp)
    L = _native.lib()
    nbytes = ctypes.c_size_t()
    _native.check(L.asdf_icp_workspace_bytes(job.src.shape[0], job.tgt.shape[0], ctypes.byref(nbytes)), "asdf_icp_workspace_bytes")
    job.ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
    job.result = torch.zeros(8, dtype=torch.float64).pin_memory()
    job.args = (int(max_iter), float(stop_error), float(stop_improvement))
    with torch.cuda.device(dev):
        _enqueue_range(job, 0, min(FIRST_BATCH, int(max_iter)))
        job.done = torch.cuda.Event()
        job.done.record(job.stream)
    return job


_CONTINUATION_STREAMS = {}


def _continuation_stream(device):
    """A side stream per device for the rare continuation of a non-converged run: finish_icp is called when job.stream already holds
    pass 2 of the next sample and pass 1 of the one after - queued THERE the remaining iterations would wait behind both; on a side
    stream (ordered behind the first batch) they start at the next kernel boundary."""
    key = (device.type, device.index)
    if key not in _CONTINUATION_STREAMS:
        _CONTINUATION_STREAMS[key] = torch.cuda.Stream(device=device)
    return _CONTINUATION_STREAMS[key]


def finish_icp(job, vertices):
    """Wait for a start_icp / start_icp_device job; returns the dict of icp_trans_scale for `vertices`."""
    job.done.synchronize()            # the ICP only - not whatever was queued behind it
    res = job.result.tolist()
    if res[6] == 0.0 and int(res[4]) < job.args[0]:
        # not converged within the first batch (rare: runs take a handful of iterations): the remaining ones, now, on the side
        # stream.  The run continues with the search mode it was begun with (csrc/icp.hip: remember_run).
        side = _continuation_stream(job.device)
        side.wait_event(job.done)
        first_stream, job.stream = job.stream, side
        with torch.cuda.stream(side):
            _enqueue_range(job, int(res[4]), job.args[0])
            job.done.record(side)
        job.stream = first_stream
        job.done.synchronize()
        res = job.result.tolist()
    scale, trans, iters, error = res[0], np.array([res[1], res[2], res[3]]), int(res[4]), res[5]
    if job.norm is None:
        n = job.host[0].numpy()
        job.norm = (n[0:3].copy(), float(n[3]), n[4:7].copy(), float(n[7]))
    offset_s, scale_s, offset_t, scale_t = job.norm
    v = (np.asarray(vertices, np.float64) - offset_s) / scale_s * scale_t + offset_t
    return dict(scale=scale, trans=trans, iterations=iters, error=error, all_scale=scale_t * scale / scale_s,
                all_trans=trans + offset_t * scale - offset_s * scale_t * scale / scale_s, vertices=v * scale + trans)


def start_alignment(verts, faces, gt_verts, gt_faces, samples=30000, max_iter=100, seed=0, device="cuda"):
    """First half of the eval-mode block of utils/mesh.py:385-395: sample both meshes, enqueue the ICP."""
    ps = sample_surface(verts, faces, samples, seed)
    pt = sample_surface(gt_verts, gt_faces, samples, seed + 1)
    return start_icp(ps, pt, max_iter, device)


def start_alignment_device(kept_verts_dev, kept_faces_dev, counts_dev, origin, voxel_size, target_points, samples=30000, max_iter=100, seed=0):
    """start_alignment for a surface that lives on the device - the largest-component filter's output (lattice-unit vertices and
    faces at input capacity, counts_dev[1] = number of kept faces): vertex placement in the exporter's fp32 arithmetic
    (utils/mesh.py:360-369), the area-weighted sampling, the normalisation and the ICP are all enqueued on the current stream;
    no step waits for the device.  `target_points` [samples,3] fp64: the ground-truth mesh's samples (host array or pinned
    tensor; sampled with seed + 1 like start_alignment does)."""
    dev = kept_verts_dev.device
    vs = np.float32(voxel_size.item() if hasattr(voxel_size, "item") else voxel_size)
    if kept_verts_dev.dtype == torch.float32 and kept_faces_dev.dtype == torch.int32 and kept_verts_dev.is_contiguous() and kept_faces_dev.is_contiguous():
        # K9 places the vertices itself (fp32 multiply, fp32 add: place_vertices) as it reads them
        ps = sample_surface_native(kept_verts_dev, kept_faces_dev, samples, seed, counts_dev[1:2], placement=(vs, [np.float32(o) for o in origin]))
    else:
        org = torch.tensor([np.float32(o) for o in origin], dtype=torch.float32, device=dev)
        placed = kept_verts_dev * float(vs) + org                       # fp32 multiply, fp32 add: place_vertices on the device
        ps = sample_surface_device(placed, kept_faces_dev, samples, seed, counts_dev[1:2])
    pt = target_points if torch.is_tensor(target_points) else torch.from_numpy(np.ascontiguousarray(target_points, dtype=np.float64)).pin_memory()
    job = start_icp_device(ps, pt.to(dev, non_blocking=True), max_iter)
    job.host.append(pt)               # keep the pinned staging buffer alive until the run is done
    return job


def align_to_ground_truth(verts, faces, gt_verts, gt_faces, samples=30000, max_iter=100, seed=0, device="cuda"):
    """The eval-mode block of utils/mesh.py:385-395: sample both meshes, ICP, return (aligned verts, trans, scale)."""
    r = finish_icp(start_alignment(verts, faces, gt_verts, gt_faces, samples, max_iter, seed, device), verts)
    return r["vertices"], r["all_trans"], r["all_scale"], r
