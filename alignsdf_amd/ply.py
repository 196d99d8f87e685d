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
