"""Host restatement of the largest-component filter of utils/mesh.py:371-381.  TEST INFRASTRUCTURE ONLY: the checker of
the device filter (K8, alignsdf_amd/csrc/mesh_cc.hip); nothing under alignsdf_amd/ imports it.

The reference builds `trimesh.Trimesh(vertices, faces, process=False)`, calls `trimesh.graph.split` and, if
more than one sub-mesh comes back, keeps the one with the largest area.  trimesh is an un-pinned, un-vendored
dependency that is not installable in this environment, so this module re-states the documented semantics of
`graph.split(mesh, only_watertight=True)`:
  * faces are adjacent when they share an edge that belongs to exactly two faces (`face_adjacency`);
  * connected components with fewer than 4 faces are dropped (`min_len = 4`);
  * only watertight components (every edge shared by exactly two faces) are returned;
  * a sub-mesh keeps the referenced vertices in ascending original order.
PARITY UNPINNED: there is no trimesh here to check against (trimesh 3.x additionally tries `fill_holes` on
open components before dropping them; that repair step is not reproduced).
"""
import numpy as np
from scipy.sparse import coo_matrix
from scipy.sparse.csgraph import connected_components


def split_watertight(verts, faces):
    """List of (face_index_array) for the watertight components with >= 4 faces, in order of their first face."""
    faces = np.asarray(faces, dtype=np.int64)
    F = len(faces)
    if F == 0:
        return []
    e = np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]], 0)
    e.sort(axis=1)
    owner = np.tile(np.arange(F), 3)
    key = e[:, 0] * (int(faces.max()) + 1) + e[:, 1]
    order = np.argsort(key, kind="stable")
    key_s, owner_s = key[order], owner[order]
    start = np.flatnonzero(np.r_[True, key_s[1:] != key_s[:-1]])
    count = np.diff(np.r_[start, len(key_s)])
    pair = start[count == 2]
    adj = coo_matrix((np.ones(len(pair), dtype=np.int8), (owner_s[pair], owner_s[pair + 1])), shape=(F, F))
    _, label = connected_components(adj, directed=False)
    # an edge that is not shared by exactly two faces makes every face on it non-watertight
    edge_count = np.repeat(count, count)
    bad_face = np.zeros(F, dtype=bool)
    bad_face[owner_s[edge_count != 2]] = True
    ncomp = label.max() + 1
    size = np.bincount(label, minlength=ncomp)
    bad = np.bincount(label, weights=bad_face, minlength=ncomp) > 0
    first = np.full(ncomp, F, dtype=np.int64)
    np.minimum.at(first, label, np.arange(F))
    keep = [c for c in np.argsort(first) if size[c] >= 4 and not bad[c]]
    return [np.flatnonzero(label == c) for c in keep]


def component_census(faces):
    """(qualifying, open_dropped, small_dropped): components of >= 4 faces that are watertight; components of >= 4 faces with an edge
    that is not shared by exactly two faces (where trimesh's fill_holes - not restated - could have repaired and kept them); components of
    fewer than 4 faces.  Same adjacency as split_watertight (utils/mesh.py:371 -> trimesh.graph.split)."""
    faces = np.asarray(faces, dtype=np.int64)
    F = len(faces)
    if F == 0:
        return 0, 0, 0
    e = np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]], 0)
    e.sort(axis=1)
    owner = np.tile(np.arange(F), 3)
    key = e[:, 0] * (int(faces.max()) + 1) + e[:, 1]
    order = np.argsort(key, kind="stable")
    key_s, owner_s = key[order], owner[order]
    start = np.flatnonzero(np.r_[True, key_s[1:] != key_s[:-1]])
    count = np.diff(np.r_[start, len(key_s)])
    pair = start[count == 2]
    adj = coo_matrix((np.ones(len(pair), dtype=np.int8), (owner_s[pair], owner_s[pair + 1])), shape=(F, F))
    _, label = connected_components(adj, directed=False)
    bad_face = np.zeros(F, dtype=bool)
    bad_face[owner_s[np.repeat(count, count) != 2]] = True
    ncomp = label.max() + 1
    size = np.bincount(label, minlength=ncomp)
    bad = np.bincount(label, weights=bad_face, minlength=ncomp) > 0
    return int(((size >= 4) & ~bad).sum()), int(((size >= 4) & bad).sum()), int((size < 4).sum())


def face_areas(verts, faces):
    v = np.asarray(verts, dtype=np.float64)
    a, b, c = v[faces[:, 0]], v[faces[:, 1]], v[faces[:, 2]]
    return 0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1)


def keep_largest_component(verts, faces):
    """(verts, faces) of the component utils/mesh.py:373-381 would export.  The input is returned unchanged when
    the split yields fewer than two sub-meshes, exactly like the `if len(split_mesh) > 1` of the reference."""
    comps = split_watertight(verts, faces)
    if len(comps) <= 1:
        return verts, faces
    area = face_areas(verts, np.asarray(faces))
    best = max(comps, key=lambda idx: area[idx].sum())     # first maximum wins, as in the reference's loop
    sub = np.asarray(faces)[best]
    used = np.unique(sub)
    remap = np.full(len(verts), -1, dtype=np.int64)
    remap[used] = np.arange(len(used))
    return np.asarray(verts)[used], remap[sub].astype(np.asarray(faces).dtype)
