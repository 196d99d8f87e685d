"""Statistics of a level-0 surface that say how hard a volume is for marching cubes (numpy only - also imported by
tests/golden/make_r4_goldens.py under the skimage interpreter): the number of cells whose sign pattern is one of Lewiner's AMBIGUOUS
configurations (cases 3, 6, 7, 10, 12, 13: a face whose diagonal corners agree and whose neighbours differ; case 4: the two minority
corners on a body diagonal) - where MC33 reads VALUES to pick a tiling, so an error in a value, not only in a sign, changes the
mesh - and the number of connected components of a triangle mesh."""
import numpy as np

_FACES = ((0, 1, 2, 3), (4, 5, 6, 7), (0, 1, 5, 4), (3, 2, 6, 7), (0, 3, 7, 4), (1, 2, 6, 5))       # corner cycles of the six faces
_DIAGONALS = ((0, 6), (1, 7), (2, 4), (3, 5))


def ambiguous_cells(volume, level=0.0):
    """Number of cells of `volume` (3-D array) with an ambiguous MC33 sign configuration at `level` (corner above iff v > level)."""
    above = np.asarray(volume) > level
    z, y, x = (slice(0, -1), slice(1, None)), (slice(0, -1), slice(1, None)), (slice(0, -1), slice(1, None))
    # corner order of the Lewiner tables: v0 = (z, y, x), v1 = x + 1, v2 = x + 1, y + 1, v3 = y + 1, v4 .. v7 the same at z + 1
    order = ((0, 0, 0), (0, 0, 1), (0, 1, 1), (0, 1, 0), (1, 0, 0), (1, 0, 1), (1, 1, 1), (1, 1, 0))
    c = [above[z[dz], y[dy], x[dx]] for dz, dy, dx in order]
    amb = np.zeros(c[0].shape, dtype=bool)
    for a, b, cc, d in _FACES:
        amb |= (c[a] == c[cc]) & (c[b] == c[d]) & (c[a] != c[b])
    count = sum(k.astype(np.int8) for k in c)
    for a, b in _DIAGONALS:
        others_low = count == 2
        others_high = count == 6
        amb |= (others_low & c[a] & c[b]) | (others_high & ~c[a] & ~c[b])
    return int(amb.sum())


def mesh_components(faces, num_verts):
    """Connected components of a triangle mesh (faces [F, 3] vertex indices), vertices as nodes."""
    faces = np.asarray(faces, dtype=np.int64)
    if len(faces) == 0:
        return 0
    try:
        from scipy.sparse import coo_matrix
        from scipy.sparse.csgraph import connected_components
        i = np.concatenate([faces[:, 0], faces[:, 1]])
        j = np.concatenate([faces[:, 1], faces[:, 2]])
        g = coo_matrix((np.ones(len(i), dtype=np.int8), (i, j)), shape=(num_verts, num_verts))
        used = np.zeros(num_verts, dtype=bool)
        used[faces.reshape(-1)] = True
        n, lab = connected_components(g, directed=False)
        return int(len(np.unique(lab[used])))
    except ImportError:
        parent = np.arange(num_verts)

        def find(a):
            while parent[a] != a:
                parent[a] = parent[parent[a]]
                a = parent[a]
            return a
        for f in faces:
            r = [find(int(k)) for k in f]
            parent[r[1]] = r[0]
            parent[r[2]] = r[0]
        return int(len({find(int(k)) for k in np.unique(faces)}))
