"""Minimal binary PLY writer (little endian): float32 x,y,z vertices + `list uchar int` triangle faces -
the layout both writers of the reference produce (trimesh export at utils/mesh.py:397, plyfile at
deep_sdf/mesh.py:107-112)."""
import numpy as np


def write_ply(path, verts, faces):
    verts = np.ascontiguousarray(verts, dtype="<f4").reshape(-1, 3)
    faces = np.ascontiguousarray(faces, dtype="<i4").reshape(-1, 3)
    header = ("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\n"
              "property float z\nelement face %d\nproperty list uchar int vertex_indices\nend_header\n" % (len(verts), len(faces)))
    rec = np.empty(len(faces), dtype=[("n", "u1"), ("idx", "<i4", (3,))])
    rec["n"] = 3
    rec["idx"] = faces
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(verts.tobytes())
        f.write(rec.tobytes())


def read_ply(path):
    """Inverse of write_ply (for tests)."""
    with open(path, "rb") as f:
        data = f.read()
    end = data.index(b"end_header\n") + len(b"end_header\n")
    nv = nf = 0
    for line in data[:end].decode("ascii").splitlines():
        if line.startswith("element vertex"):
            nv = int(line.split()[-1])
        if line.startswith("element face"):
            nf = int(line.split()[-1])
    verts = np.frombuffer(data, dtype="<f4", count=nv * 3, offset=end).reshape(nv, 3)
    rec = np.frombuffer(data, dtype=[("n", "u1"), ("idx", "<i4", (3,))], count=nf, offset=end + nv * 12)
    return verts.copy(), rec["idx"].copy()


def write_ply_ascii(path, verts, faces=None, vertex_colors=None):
    """ASCII PLY with optional per-vertex RGB(A) colours: the text layout of the reference's customized_export_ply
    for the (v, f, v_c) combination it is called with (utils/customized_export_ply.py:49-119 - `%f` coordinates,
    `uchar` red/green/blue/alpha with alpha 255 when only RGB is given, `3 i j k` faces)."""
    verts = np.asarray(verts).reshape(-1, 3)
    faces = np.zeros((0, 3), dtype=np.int64) if faces is None else np.asarray(faces).reshape(-1, 3)
    head = ["ply", "format ascii 1.0", "element vertex %d" % len(verts), "property float x", "property float y", "property float z"]
    if vertex_colors is not None:
        vc = np.asarray(vertex_colors)
        if vc.shape[1] == 3:
            vc = np.hstack([vc, np.full((len(vc), 1), 255, dtype=np.uint8)])
        head += ["property uchar red", "property uchar green", "property uchar blue", "property uchar alpha"]
        body = ["%f %f %f %d %d %d %d" % (v[0], v[1], v[2], c[0], c[1], c[2], c[3]) for v, c in zip(verts.tolist(), vc.tolist())]
    else:
        body = ["%f %f %f" % tuple(v) for v in verts.tolist()]
    head += ["element face %d" % len(faces), "property list uchar int vertex_indices", "end_header"]
    body += ["3 %d %d %d" % tuple(f) for f in faces.tolist()]
    with open(path, "w") as f:
        f.write("\n".join(head + body) + "\n")
