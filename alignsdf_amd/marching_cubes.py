"""Device-side Lewiner marching cubes with the call shape of skimage.measure.marching_cubes_lewiner
as the reference uses it (utils/mesh.py:354, deep_sdf/mesh.py:81).

All arithmetic runs in libalignsdf_hip.so (asdf_mc_count / asdf_mc_emit); torch only owns the buffers.
"""
import ctypes

import numpy as np
import torch

from . import _native

_workspaces = {}
_generation = {}     # workspace key -> count phases enqueued into it so far: a ticket whose workspace has served another volume since
                     # its own count phase is counted again before it is emitted (its scan results are gone)
_SLOTS = 2           # volumes whose count phase may be in flight at once (hand + object of one sample)
_results = []        # ring of pinned result records (V, F, min key, max key): one per ticket, so that a count phase enqueued for
_result_turn = 0     # the NEXT sample cannot overwrite sizes the host has not read yet (round 5: whole samples are enqueued ahead)


def _workspace(shape, device, slot=0):
    key = (tuple(shape), str(device), slot)
    ws = _workspaces.get(key)
    if ws is None:
        nbytes = ctypes.c_size_t()
        _native.check(_native.lib().asdf_mc_workspace_bytes(shape[0], shape[1], shape[2], ctypes.byref(nbytes)),
                      "asdf_mc_workspace_bytes")
        for k in [k for k in _workspaces if k[:2] != key[:2]]:      # keep the workspaces of one shape alive (~340 MB each at 256^3)
            del _workspaces[k]
        ws = torch.empty(nbytes.value, dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def _result_record():
    global _result_turn
    if not _results:
        _results.extend(torch.zeros(4, dtype=torch.int32).pin_memory() for _ in range(16))
    r = _results[_result_turn % len(_results)]
    _result_turn += 1
    return r


def marching_cubes_begin(volume, level=0.0, slot=0, capacity=None):
    """Enqueue the count phase (classify + reduce) of one volume WITHOUT synchronising: the sizes land in pinned host memory
    behind an event.  Returns a ticket for marching_cubes_finish; `slot` (0 / 1) picks the workspace, so that the hand and
    the object volume of a sample can both be in flight.

    capacity = (max vertices, max faces): the EMIT phase is enqueued right behind the count phase into buffers of that size
    (asdf_mc_emit_bounded) - no host round trip between count and emit; marching_cubes_finish hands out the filled prefix, or runs
    the two phases again when a capacity turned out too small."""
    if not isinstance(volume, torch.Tensor) or not volume.is_cuda:
        raise TypeError("marching cubes needs a CUDA tensor (there is no CPU fallback)")
    if volume.dim() != 3:
        raise ValueError("Input volume should be a 3D numpy array.")
    if min(volume.shape) < 2:
        raise ValueError("Input array must be at least 2x2x2.")
    vol = volume.detach().to(torch.float32).contiguous()
    dev = vol.device
    ws, result = _workspace(vol.shape, dev, slot % _SLOTS), _result_record()
    key = (tuple(vol.shape), str(dev), slot % _SLOTS)
    gen = _generation[key] = _generation.get(key, 0) + 1
    L = _native.lib()
    bufs = None
    with torch.cuda.device(dev):
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _native.check(L.asdf_mc_count_enqueue(vol.data_ptr(), vol.shape[0], vol.shape[1], vol.shape[2], ctypes.c_double(float(level)),
                                              ws.data_ptr(), ws.numel(), result.data_ptr(), stream), "asdf_mc_count_enqueue")
        if capacity is not None:
            cv, cf = int(capacity[0]), int(capacity[1])
            bufs = (torch.empty((cv, 3), dtype=torch.float32, device=dev), torch.empty((cf, 3), dtype=torch.int32, device=dev))
            _native.check(L.asdf_mc_emit_bounded(vol.data_ptr(), vol.shape[0], vol.shape[1], vol.shape[2], ctypes.c_double(float(level)),
                                                 ws.data_ptr(), ws.numel(), bufs[0].data_ptr(), cv, bufs[1].data_ptr(), cf, stream),
                          "asdf_mc_emit_bounded")
        done = torch.cuda.Event()
        done.record()
    return vol, float(level), ws, result, done, bufs, key, gen


def marching_cubes_finish(ticket):
    """Wait for the sizes of marching_cubes_begin (an event, not the stream), allocate, emit.  Returns (verts [V,3] fp32,
    faces [F,3] int32) device tensors; raises ValueError / RuntimeError with skimage's messages."""
    vol, level, ws, result, done, bufs, key, gen = ticket
    done.synchronize()
    L = _native.lib()
    r = result.numpy().view(np.uint32)
    rc = L.asdf_mc_result_status(ctypes.c_void_p(result.data_ptr()), ctypes.c_double(level))
    if rc == _native.ERANGE:
        raise ValueError("Surface level must be within volume data range.")
    if rc == _native.ENOSURF:
        raise RuntimeError("No surface found at the given iso value.")
    _native.check(rc, "asdf_mc_result_status")
    V, F = int(r[0]), int(r[1])
    if bufs is not None:
        if V <= bufs[0].shape[0] and F <= bufs[1].shape[0]:
            return bufs[0][:V], bufs[1][:F]          # (emitted behind the count phase: nothing left to do)
        # a capacity was too small; the workspace may have served another volume since: both phases again, sizes known
        return marching_cubes_finish(marching_cubes_begin(vol, level, key[2]))
    if _generation.get(key) != gen:
        # another volume's count phase has used this workspace since (a caller that interleaves tickets on one slot): the emit below
        # would read ITS scan results with THIS ticket's sizes - count again, emit right away
        return marching_cubes_finish(marching_cubes_begin(vol, level, key[2]))
    dev = vol.device
    verts = torch.empty((V, 3), dtype=torch.float32, device=dev)
    faces = torch.empty((F, 3), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _native.check(L.asdf_mc_emit(vol.data_ptr(), vol.shape[0], vol.shape[1], vol.shape[2], ctypes.c_double(level),
                                     ws.data_ptr(), ws.numel(), verts.data_ptr(), faces.data_ptr(), stream), "asdf_mc_emit")
    return verts, faces


def marching_cubes_device(volume, level=0.0):
    """volume: [n0,n1,n2] fp32 CUDA tensor.  Returns (verts [V,3] fp32, faces [F,3] int32) device tensors
    in voxel units, element-for-element what skimage returns before its `* spacing` step.
    Raises ValueError / RuntimeError with skimage's messages (the reference catches them, utils/mesh.py:353-358)."""
    return marching_cubes_finish(marching_cubes_begin(volume, level))


def marching_cubes_lewiner(volume, level=0.0, spacing=(1.0, 1.0, 1.0)):
    """numpy-returning variant: (verts, faces) on the host with skimage's spacing / dtype semantics
    (`vertices * np.r_[spacing]` unless spacing == (1,1,1))."""
    verts, faces = marching_cubes_device(volume, level)
    verts, faces = verts.cpu().numpy(), faces.cpu().numpy()
    if not np.array_equal(spacing, (1, 1, 1)):
        verts = verts * np.r_[spacing]
    return verts, faces


def time_chain(vol_hand, vol_obj, peak_gbs, repeats=5):
    """HBM roofline record of the marching-cubes kernel chain (classify / scan / emit) on the two volumes of one sample:
    achieved = algorithmic bytes (4 N^3 read + 12 V + 12 F written per volume) / duration of the chain between
    two events on the launch stream (best of `repeats`; the V / F read-back between count and emit is inside, as in the
    product)."""
    best, nbytes = None, 0
    for _ in range(repeats):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        nbytes = 0
        tickets = [marching_cubes_begin(vol, 0.0, slot) for slot, vol in enumerate((vol_hand, vol_obj))]     # as the pipeline does
        for vol, t in zip((vol_hand, vol_obj), tickets):
            v, f = marching_cubes_finish(t)
            nbytes += 4 * vol.numel() + 12 * v.shape[0] + 12 * f.shape[0]
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1)
        best = ms if best is None else min(best, ms)
    gbs = nbytes / (best * 1e-3) / 1e9
    return {"bound": "hbm", "kernels": "mc_classify + mc_finalize + mc_emit_verts + mc_emit_faces (both volumes of one sample)",
            "achieved": gbs, "peak": peak_gbs, "unit": "GB/s", "frac": gbs / peak_gbs, "traffic": None,
            "chain_ms_both_volumes": best, "algorithmic_bytes": nbytes}
