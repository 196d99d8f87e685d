"""ctypes front end of the sequential MC33 oracle (oracle/mc33_oracle.c).  TEST INFRASTRUCTURE ONLY.

`marching_cubes_lewiner(volume, level, spacing)` mirrors the call the reference makes
(utils/mesh.py:354): returns (verts, faces) where verts = fp32 voxel-unit vertices times `spacing`
(numpy promotion rules apply, exactly as skimage's `vertices * np.r_[spacing]`) and faces int32 [F,3].
Raises ValueError / RuntimeError with skimage's messages for the two failure modes the reference
catches (utils/mesh.py:353-358).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "mc33_oracle.c")
_LIB = os.path.join(_HERE, "_build", "libmc33_oracle.so")
_lib = None


def build(force=False):
    deps = [_SRC, os.path.join(_HERE, "mc33_oracle_tables.h")]      # nothing of the product's is compiled into the oracle
    if force or not os.path.exists(_LIB) or any(os.path.getmtime(d) > os.path.getmtime(_LIB) for d in deps):
        os.makedirs(os.path.dirname(_LIB), exist_ok=True)
        subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-std=c99", _SRC, "-o", _LIB, "-lm"], check=True)
    return _LIB


def _load():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(build())
        L.mc33_lewiner.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                   ctypes.POINTER(ctypes.POINTER(ctypes.c_float)), ctypes.POINTER(ctypes.c_int),
                                   ctypes.POINTER(ctypes.POINTER(ctypes.c_int)), ctypes.POINTER(ctypes.c_int)]
        L.mc33_free.argtypes = [ctypes.c_void_p]
        L.mc33_free.restype = None
        _lib = L
    return _lib


def marching_cubes_raw(volume, level=0.0):
    """(verts fp32 [V,3] in voxel units, faces int32 [F,3])."""
    L = _load()
    vol = np.ascontiguousarray(volume, dtype=np.float32)
    if vol.ndim != 3:
        raise ValueError("Input volume should be a 3D numpy array.")
    if min(vol.shape) < 2:
        raise ValueError("Input array must be at least 2x2x2.")
    vp, fp = ctypes.POINTER(ctypes.c_float)(), ctypes.POINTER(ctypes.c_int)()
    nv, nf = ctypes.c_int(), ctypes.c_int()
    rc = L.mc33_lewiner(vol.ctypes.data, vol.shape[0], vol.shape[1], vol.shape[2], float(level), ctypes.byref(vp),
                        ctypes.byref(nv), ctypes.byref(fp), ctypes.byref(nf))
    if rc == -6:
        raise ValueError("Surface level must be within volume data range.")
    if rc == -7:
        raise RuntimeError("No surface found at the given iso value.")
    if rc != 0:
        raise MemoryError("mc33 oracle failed with code %d" % rc)
    verts = np.ctypeslib.as_array(vp, shape=(nv.value, 3)).copy()
    faces = np.ctypeslib.as_array(fp, shape=(nf.value, 3)).astype(np.int32)
    L.mc33_free(vp)
    L.mc33_free(fp)
    return verts, faces


def marching_cubes_lewiner(volume, level=0.0, spacing=(1.0, 1.0, 1.0)):
    verts, faces = marching_cubes_raw(volume, level)
    if not np.array_equal(spacing, (1, 1, 1)):
        verts = verts * np.r_[spacing]
    return verts, faces
