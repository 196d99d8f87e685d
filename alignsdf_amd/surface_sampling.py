"""Host side of eval mode's surface sampling: the Wavefront OBJ reader and the seeded area-weighted sampler that stands in for
`trimesh.load` / `trimesh.sample.sample_surface` (utils/mesh.py:336, 386-389; deep_sdf/metrics/icp_trans_scale.py:19-23).
numpy only - the ground-truth worker process (gt_worker.py) imports this module and must start in a fraction of a second, so
nothing here may import torch.  alignsdf_amd.icp re-exports every name."""
import numpy as np

from . import synthetic


def _load_obj_streaming(lines):
    """Line-by-line OBJ reader: polygons fan-triangulated, v/vt/vn corners, and RELATIVE (negative) indices resolved against the
    vertices read SO FAR at that face line - what the format specifies when v and f blocks are interleaved."""
    verts, out = [], []
    for ln in lines:
        if ln.startswith("v "):
            verts.append([float(x) for x in ln.split()[1:4]])
        elif ln.startswith("f "):
            idx = [int(tok.split("/")[0]) for tok in ln.split()[1:]]
            idx = [i - 1 if i > 0 else len(verts) + i for i in idx]
            for k in range(1, len(idx) - 1):
                out.append([idx[0], idx[k], idx[k + 1]])
    return np.asarray(verts, dtype=np.float64).reshape(-1, 3), np.asarray(out, dtype=np.int64).reshape(-1, 3)


def load_obj(path):
    """Minimal Wavefront OBJ reader: (verts float64 [V,3], faces int64 [F,3]); polygons are fan-triangulated.
    Fast path (~10 x the loop on a 10 k-face mesh): vertex lines parsed in one numpy call, triangle faces too - taken ONLY when every
    vertex line has the same number of columns, every face is a plain or slashed triple and every index is positive (what mesh
    exporters write).  Anything else - polygons, relative (negative) indices, which count from the vertices read so far at that
    line, ragged vertex lines - goes through the streaming reader."""
    with open(path, "r") as f:
        lines = f.read().split("\n")
    vlines = [ln for ln in lines if ln.startswith("v ")]
    flines = [ln for ln in lines if ln.startswith("f ")]
    if not vlines or not flines:
        return _load_obj_streaming(lines)
    import warnings
    columns = {len(ln.split()) for ln in vlines[:: max(1, len(vlines) // 64)]} | {len(vlines[-1].split())}
    text = " ".join(ln[2:] for ln in flines)
    if len(columns) != 1 or "-" in text:
        return _load_obj_streaming(lines)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")            # (np.fromstring's text mode is deprecated, and still the fastest parser here)
        flat = np.fromstring(" ".join(ln[2:] for ln in vlines), dtype=np.float64, sep=" ")
    width = next(iter(columns)) - 1
    if width < 3 or flat.size != width * len(vlines):
        return _load_obj_streaming(lines)
    verts = flat.reshape(len(vlines), width)[:, :3]
    if "/" in text:                                            # v/vt/vn forms: keep the vertex index of every corner
        import re
        text = re.sub(r"/[^ ]*", "", text)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            idx = np.fromstring(text, dtype=np.int64, sep=" ")
    except ValueError:
        return _load_obj_streaming(lines)
    if idx.size != 3 * len(flines) or idx.min() < 1:           # a polygon somewhere, or an index that is not plain positive
        return _load_obj_streaming(lines)
    return np.ascontiguousarray(verts, dtype=np.float64), idx.reshape(-1, 3) - 1


# The sampler picks faces by cumulative area.  A cumulative sum of floating-point areas depends on the order of the additions - a
# sequential numpy cumsum and a parallel scan on the device round differently and would, rarely, pick different faces.  So the
# areas are QUANTISED first: q = rint(area / max(area) * 2^31), an exact function of each area and of the (order-independent)
# maximum; integer sums are exact in any order, and a draw u selects the first face whose cumulative q exceeds floor(u * total).
# Host (numpy) and device (torch) then produce the same picks bit for bit (tests/test_gpu_icp.py), the distribution is
# area-weighted to 2^-31 of the largest face, and the sampler stays what it stands in for: trimesh.sample.sample_surface,
# which the reference calls UNSEEDED (utils/mesh.py:336).
_AREA_QUANTUM = 2147483648.0


def _uniforms(count, seed):
    """The draws of one sampling call: u for the face pick [count], (r1, r2) barycentric pair [count, 2] (reflected)."""
    u = synthetic.uniform((count,), 9100 + seed)
    r = synthetic.uniform((count, 2), 9200 + seed)
    flip = r.sum(1) > 1.0
    r[flip] = np.abs(r[flip] - 1.0)
    return u, r


def sample_surface(verts, faces, count, seed=0):
    """`count` area-weighted uniform samples of a triangle mesh (the scheme of trimesh.sample.sample_surface: pick faces
    by cumulative area, reflect barycentric pairs whose sum exceeds 1), driven by the repo's seeded generator."""
    v = np.asarray(verts, dtype=np.float64)
    f = np.asarray(faces, dtype=np.int64)
    # areas of all faces (190 k per hand surface at N = 256) on column vectors: the same products, differences and sums as
    # 0.5 * norm(cross(b - a, c - a)) at a third of the time of the [F, 3] gathers + np.cross
    vx, vy, vz = np.ascontiguousarray(v[:, 0]), np.ascontiguousarray(v[:, 1]), np.ascontiguousarray(v[:, 2])
    i0, i1, i2 = f[:, 0], f[:, 1], f[:, 2]
    ax, ay, az = vx[i0], vy[i0], vz[i0]
    e1x, e1y, e1z = vx[i1] - ax, vy[i1] - ay, vz[i1] - az
    e2x, e2y, e2z = vx[i2] - ax, vy[i2] - ay, vz[i2] - az
    cx, cy, cz = e1y * e2z - e1z * e2y, e1z * e2x - e1x * e2z, e1x * e2y - e1y * e2x
    area = 0.5 * np.sqrt(cx * cx + cy * cy + cz * cz)
    q = np.rint(area / area.max() * _AREA_QUANTUM).astype(np.int64)
    cum = np.cumsum(q)
    u, r = _uniforms(count, seed)
    target = np.floor(u * float(cum[-1])).astype(np.int64)
    pick = np.minimum(np.searchsorted(cum, target, side="right"), len(f) - 1)
    a, b, c = v[f[pick, 0]], v[f[pick, 1]], v[f[pick, 2]]          # only the picked faces
    return a + (b - a) * r[:, :1] + (c - a) * r[:, 1:]
