"""Golden vectors for marching cubes, produced by the INSTALLED skimage 0.18.3 binary.

Run in the build container with the interpreter that has scikit-image:
    /opt/conda/bin/python3.9 tests/golden/make_mc_goldens.py
Calls `skimage.measure.marching_cubes_lewiner(vol, level=..., spacing=...)` - the same function and
signature the reference uses at utils/mesh.py:354 - and stores inputs + outputs:

  mc_cells.npz     every non-trivial 2x2x2 sign pattern x K magnitude draws: corner values, V, F and the
                   full (padded) vertex / face arrays
  mc_cells_ambiguous.npz  150 draws for each value-dependent sign pattern: corners, V, F, checksums
  mc_volumes.npz   analytic shapes, exact-zero corners, noise (ambiguous cases), non-cubic shapes,
                   the 32^3 pass-2 volumes of the synthetic decoders: full verts / faces
"""
import os
import sys
import warnings

import numpy as np

warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
from skimage.measure import marching_cubes_lewiner  # noqa: E402

from alignsdf_amd import synthetic as syn  # noqa: E402


def mc(vol, level=0.0, spacing=(1.0, 1.0, 1.0)):
    v, f, _, _ = marching_cubes_lewiner(np.asarray(vol, np.float32), level=level, spacing=spacing)
    return v, f


def cells():
    K = 20
    vals, V, F = [], [], []
    verts = np.full((254 * K, 13, 3), np.nan, np.float32)
    faces = np.full((254 * K, 12, 3), -1, np.int32)
    n = 0
    for pattern in range(1, 255):
        sign = np.array([1.0 if (pattern >> c) & 1 else -1.0 for c in range(8)])
        for k in range(K):
            if k < 8:
                mag = syn.uniform((8,), 10_000 + pattern * 64 + k, 0.05, 1.0)
            elif k < 16:   # wide dynamic range: drives the interior / face deciders both ways
                mag = 10.0 ** syn.uniform((8,), 10_000 + pattern * 64 + k, -3.0, 1.0)
            else:          # one dominant diagonal pair
                mag = syn.uniform((8,), 10_000 + pattern * 64 + k, 0.05, 0.2)
                mag[(k * 3) % 8] = 5.0
                mag[7 - (k * 3) % 8] = 4.0
            c = (sign * mag).astype(np.float32)
            # corner c -> array position: v0=(0,0,0) v1=(0,0,1) v2=(0,1,1) v3=(0,1,0) v4..v7 at z=1
            vol = np.zeros((2, 2, 2), np.float32)
            for ci, (z, y, x) in enumerate([(0, 0, 0), (0, 0, 1), (0, 1, 1), (0, 1, 0), (1, 0, 0), (1, 0, 1), (1, 1, 1),
                                            (1, 1, 0)]):
                vol[z, y, x] = c[ci]
            v, f = mc(vol)
            vals.append(c); V.append(len(v)); F.append(len(f))
            verts[n, :len(v)] = v
            faces[n, :len(f)] = f
            n += 1
    return dict(corners=np.array(vals), V=np.array(V), F=np.array(F), verts=verts, faces=faces)


CORNER_POS = [(0, 0, 0), (0, 0, 1), (0, 1, 1), (0, 1, 0), (1, 0, 0), (1, 0, 1), (1, 1, 1), (1, 1, 0)]


def ambiguous_cells():
    """Many more magnitude draws for the sign patterns whose tiling is value dependent (MC33 cases
    3, 4, 6, 7, 10, 12, 13), so that the rare sub-cases (7.4.x, 10.x, 12.x, 13.1-13.5) are pinned.
    Stores corners + (V, F) + an order-sensitive checksum of faces and verts."""
    K = 150
    # find value-dependent patterns empirically from the first table
    corners, V, F, fsum, vsum = [], [], [], [], []
    for pattern in range(1, 255):
        sign = np.array([1.0 if (pattern >> c) & 1 else -1.0 for c in range(8)])
        counts = set()
        rows = []
        for k in range(K):
            if k % 3 == 0:
                mag = syn.uniform((8,), 900_000 + pattern * 1024 + k, 0.02, 1.0)
            elif k % 3 == 1:
                mag = 10.0 ** syn.uniform((8,), 900_000 + pattern * 1024 + k, -2.5, 0.5)
            else:
                mag = syn.uniform((8,), 900_000 + pattern * 1024 + k, 0.02, 0.3)
                sel = (syn.splitmix64(np.arange(3, dtype=np.uint64), pattern * 1024 + k) % np.uint64(8)).astype(int)
                mag[sel] *= 12.0
            c = (sign * mag).astype(np.float32)
            vol = np.zeros((2, 2, 2), np.float32)
            for ci, (z, y, x) in enumerate(CORNER_POS):
                vol[z, y, x] = c[ci]
            v, f = mc(vol)
            counts.add(len(f))
            w = np.arange(1, f.size + 1, dtype=np.int64)
            rows.append((c, len(v), len(f), int((f.reshape(-1).astype(np.int64) * w).sum()), float(v.astype(np.float64).sum())))
        if len(counts) > 1:
            for c, nv, nf, fs, vs in rows:
                corners.append(c); V.append(nv); F.append(nf); fsum.append(fs); vsum.append(vs)
    print("ambiguous patterns kept:", len(corners) // K)
    return dict(corners=np.array(corners), V=np.array(V), F=np.array(F), fsum=np.array(fsum), vsum=np.array(vsum))


def volumes():
    out = {}

    def add(name, vol, level=0.0, spacing=(1.0, 1.0, 1.0)):
        vol = np.asarray(vol, np.float32)
        v, f = mc(vol, level, spacing)
        out[name + ".vol"] = vol
        out[name + ".level"] = np.array([level])
        out[name + ".spacing"] = np.array(spacing, dtype=np.float64)
        out[name + ".verts"] = v
        out[name + ".faces"] = f
        print("%-22s shape %-14s V %6d F %6d verts dtype %s" % (name, vol.shape, len(v), len(f), v.dtype))

    g = lambda n: np.stack(np.meshgrid(*[np.linspace(-1, 1, n)] * 3, indexing="ij"), -1)
    p = g(24)
    add("sphere24", np.linalg.norm(p, axis=-1) - 0.6)
    add("sphere24_spacing", np.linalg.norm(p, axis=-1) - 0.6, 0.0, (0.1, 0.25, 0.5))
    q = np.abs(g(20)) - np.array([0.5, 0.3, 0.7])
    add("box20", np.linalg.norm(np.maximum(q, 0), axis=-1) + np.minimum(q.max(-1), 0))
    p = g(28)
    add("torus28", np.sqrt((np.sqrt(p[..., 0] ** 2 + p[..., 1] ** 2) - 0.6) ** 2 + p[..., 2] ** 2) - 0.25)
    idx = np.stack(np.meshgrid(np.arange(16), np.arange(16), np.arange(16), indexing="ij"), -1).astype(np.float64)
    add("plane_x16", idx[..., 2] / 15.0 * 2 - 1 - 0.03)
    idx = np.stack(np.meshgrid(np.arange(17), np.arange(17), np.arange(17), indexing="ij"), -1).astype(np.float64)
    add("plane_nodes17", idx[..., 0] - 8.0)              # iso surface exactly through grid nodes
    z = np.ones((3, 3, 3)); z[1, 1, 1] = 0.0
    add("zero_corner3", z)
    tb = np.ones((4, 4, 4)); tb[0, 0, 0] = -1; tb[3, 3, 3] = -1
    add("two_blobs4", tb)
    add("noise12", syn.uniform((12, 12, 12), 31337, -1.0, 1.0))
    add("noise20", syn.uniform((20, 20, 20), 31338, -1.0, 1.0))
    add("noise_9x13x17", syn.uniform((9, 13, 17), 31339, -1.0, 1.0))
    add("noise_2x2x9", syn.uniform((2, 2, 9), 31340, -1.0, 1.0))
    add("noise16_level", syn.uniform((16, 16, 16), 31341, 0.0, 1.0), 0.3)
    add("smooth_noise24", np.sin(3.1 * g(24)[..., 0]) * np.cos(2.3 * g(24)[..., 1]) + 0.4 * np.sin(5 * g(24)[..., 2]))
    quant = np.round(syn.uniform((14, 14, 14), 31342, -3.0, 3.0))   # many exact zeros and ties
    add("quantised14", quant)
    for tag in ("nerf3", "both9"):
        r = np.load(os.path.join(HERE, "ref_decoder_%s.npz" % tag))
        vs = np.float32(r["mc_voxel_size_32"][0])
        add("dec_%s_hand32" % tag, r["vol2_hand_32"], 0.0, (vs, vs, vs))
        add("dec_%s_obj32" % tag, r["vol2_obj_32"], 0.0, (vs, vs, vs))
    # a large noise volume by count + checksum only (the volume is regenerated from the PRNG at test time)
    big = syn.uniform((48, 48, 48), 31400, -1.0, 1.0).astype(np.float32)
    v, f = mc(big)
    w = np.arange(1, f.size + 1, dtype=np.int64)
    out["noise48.V"] = np.array([len(v)]); out["noise48.F"] = np.array([len(f)])
    out["noise48.fsum"] = np.array([int((f.reshape(-1).astype(np.int64) * w).sum())])
    out["noise48.vsum"] = np.array([float(v.astype(np.float64).sum())])
    print("noise48 V %d F %d" % (len(v), len(f)))
    # failure modes the reference catches (utils/mesh.py:353-358)
    for name, vol, level in (("allpos", np.ones((4, 4, 4)), 0.0), ("level_is_max", syn.uniform((4, 4, 4), 5, 0.0, 1.0), None)):
        vol = np.asarray(vol, np.float32)
        if level is None:
            level = float(vol.max())
        try:
            mc(vol, level)
            msg = "ok"
        except Exception as e:   # noqa: BLE001
            msg = "%s: %s" % (type(e).__name__, e)
        out["fail_" + name + ".vol"] = vol
        out["fail_" + name + ".level"] = np.array([level])
        out["fail_" + name + ".error"] = np.array(msg)
        print("fail case", name, "->", msg)
    return out


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "mc_cells.npz"), **cells())
    np.savez_compressed(os.path.join(HERE, "mc_cells_ambiguous.npz"), **ambiguous_cells())
    np.savez_compressed(os.path.join(HERE, "mc_volumes.npz"), **volumes())
