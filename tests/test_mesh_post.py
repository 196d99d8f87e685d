"""Largest-component filter (utils/mesh.py:371-381 semantics) on meshes produced by the MC oracle."""
import numpy as np

from oracle.mesh_oracle import face_areas, keep_largest_component, split_watertight
from alignsdf_amd.ply import read_ply, write_ply
from oracle import mc33


def _grid(n):
    return np.stack(np.meshgrid(*[np.linspace(-1, 1, n)] * 3, indexing="ij"), -1)


def test_two_closed_blobs_keeps_larger_area():
    p = _grid(40)
    a = np.linalg.norm(p - np.array([-0.45, 0, 0]), axis=-1) - 0.3
    b = np.linalg.norm(p - np.array([0.5, 0, 0]), axis=-1) - 0.2
    v, f = mc33.marching_cubes_raw(np.minimum(a, b).astype(np.float32))
    comps = split_watertight(v, f)
    assert len(comps) == 2 and sum(len(c) for c in comps) == len(f)
    v2, f2 = keep_largest_component(v, f)
    assert len(f2) == max(len(c) for c in comps)
    assert f2.max() == len(v2) - 1 and len(np.unique(f2)) == len(v2)
    # the kept blob is the big sphere around x = -0.45 (voxel units: centre ~ 0.275 * 39)
    assert abs(v2[:, 0].mean() - (-0.45 + 1) / 2 * 39) < 0.5
    assert face_areas(v2, f2).sum() > 0.6 * face_areas(v, f).sum()


def test_single_component_and_open_surfaces_are_left_alone():
    p = _grid(24)
    v, f = mc33.marching_cubes_raw((np.linalg.norm(p, axis=-1) - 0.5).astype(np.float32))
    v2, f2 = keep_largest_component(v, f)
    assert v2 is v and f2 is f                                    # one sub-mesh -> reference keeps source_mesh
    # a plane cut by the volume boundary is not watertight: split returns nothing, mesh unchanged
    v, f = mc33.marching_cubes_raw((p[..., 0] - 0.03).astype(np.float32))
    assert split_watertight(v, f) == []
    assert keep_largest_component(v, f)[1] is f
    # closed blob + open sheet: only ONE watertight sub-mesh -> `len(split) > 1` is false -> unchanged
    both = np.minimum(np.linalg.norm(p - np.array([0.5, 0, 0]), axis=-1) - 0.3, p[..., 0] + 0.7)
    v, f = mc33.marching_cubes_raw(both.astype(np.float32))
    assert len(split_watertight(v, f)) == 1 and keep_largest_component(v, f)[1] is f


def test_ply_roundtrip(tmp_path):
    p = _grid(16)
    v, f = mc33.marching_cubes_raw((np.linalg.norm(p, axis=-1) - 0.5).astype(np.float32))
    write_ply(str(tmp_path / "a.ply"), v, f)
    v2, f2 = read_ply(str(tmp_path / "a.ply"))
    assert np.array_equal(v2, v) and np.array_equal(f2, f)
    head = open(tmp_path / "a.ply", "rb").read(200).decode("ascii", "ignore")
    assert head.startswith("ply\nformat binary_little_endian 1.0") and "property list uchar int vertex_indices" in head


def test_surface_sampler_is_area_weighted_and_uniform_inside_faces():
    """The seeded stand-in for trimesh.sample.sample_surface (utils/mesh.py:336 of the reference samples 30 000 points
    with it, unseeded): faces picked in proportion to their area, points uniform inside a face, always on the face."""
    from alignsdf_amd.icp import sample_surface
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [3, 0, 5], [3, 3, 5], [0, 3, 5.0]])
    f = np.array([[0, 1, 2], [3, 4, 5]])                  # areas 0.5 and 4.5, planes z = 0 and z = 5
    n = 200000
    p = sample_surface(v, f, n, 3)
    on0, on1 = p[:, 2] == 0.0, p[:, 2] == 5.0
    assert np.all(on0 | on1)
    assert abs(on0.mean() - 0.1) < 4 * np.sqrt(0.1 * 0.9 / n)
    q = p[on0][:, :2]                                     # uniform on the unit right triangle: E[x] = E[y] = 1/3, E[xy] = 1/12
    assert np.all(q >= 0) and np.all(q.sum(1) <= 1 + 1e-12)
    m = len(q)
    assert np.all(np.abs(q.mean(0) - 1 / 3) < 5 * np.sqrt(1 / 18) / np.sqrt(m))
    assert abs((q[:, 0] * q[:, 1]).mean() - 1 / 12) < 5 * 0.08 / np.sqrt(m)
    assert np.array_equal(p, sample_surface(v, f, n, 3)) and not np.array_equal(p, sample_surface(v, f, n, 4))
