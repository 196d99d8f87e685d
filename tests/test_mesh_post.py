"""Largest-component filter (utils/mesh.py:371-381 semantics) on meshes produced by the MC oracle."""
import numpy as np
import pytest

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


def test_load_obj_relative_indices_and_ragged_lines(tmp_path):
    """relative (negative) face indices count from the vertices read SO FAR at that face line - with interleaved v / f
    blocks that is not the final vertex count -, and vertex lines with differing column counts must not be reshaped blindly."""
    from alignsdf_amd.surface_sampling import load_obj
    p = tmp_path / "interleaved.obj"
    p.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf -3 -2 -1\nv 0 0 1\nv 1 0 1\nv 0 1 1\nf -3 -2 -1\nf 1 2 3 4\n")
    v, f = load_obj(str(p))
    assert v.shape == (6, 3) and f.tolist() == [[0, 1, 2], [3, 4, 5], [0, 1, 2], [0, 2, 3]]
    p = tmp_path / "ragged.obj"
    p.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0 0.5 0.5 0.5\nf 1/1/1 2/2/2 3/3/3\n")          # 3 + 3 + 6 columns: 12 = 3 x 4 would reshape
    v, f = load_obj(str(p))
    assert v.tolist() == [[0, 0, 0], [1, 0, 0], [0, 1, 0]] and f.tolist() == [[0, 1, 2]]
    p = tmp_path / "plain.obj"
    p.write_text("# exporter output\nv 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\nf 1 2 3\nf 2//1 4//1 3//1\n")
    v, f = load_obj(str(p))
    assert v.shape == (4, 3) and f.tolist() == [[0, 1, 2], [1, 3, 2]]


def test_file_writer_surfaces_a_failed_write_one_sample_later(tmp_path, caplog):
    """a PLY write that fails (disk full, permissions) is raised at the NEXT write, not after the whole shard has been
    decoded; close(failing=True) - the path of a `finally` that is unwinding another exception - logs instead of raising, so that
    the original error survives; the interpreter's switch interval is put back."""
    import logging
    import sys
    import time
    from alignsdf_amd.reconstruct import FileWriter
    before = sys.getswitchinterval()
    v = np.zeros((3, 3)), np.array([[0, 1, 2]])
    w = FileWriter()
    blocker = tmp_path / "not_a_dir"
    blocker.write_text("x")
    w.write_ply(str(tmp_path / "ok.ply"), *v)
    w.jobs.append((str(blocker / "bad.ply"), w.pool.submit(open, str(blocker / "bad.ply"), "wb")))      # a write that fails
    for _ in range(200):
        if all(j.done() for _, j in w.jobs):
            break
        time.sleep(0.01)
    with pytest.raises(OSError):
        w.write_ply(str(tmp_path / "next.ply"), *v)                      # surfaces here
    w.close()
    assert (tmp_path / "ok.ply").exists() and sys.getswitchinterval() == before
    w = FileWriter()
    w.jobs.append((str(blocker / "bad.ply"), w.pool.submit(open, str(blocker / "bad.ply"), "wb")))
    with caplog.at_level(logging.ERROR):
        w.close(failing=True)                                            # does not raise
    assert any("bad.ply" in r.getMessage() for r in caplog.records)
    w = FileWriter()
    w.jobs.append((str(blocker / "bad.ply"), w.pool.submit(open, str(blocker / "bad.ply"), "wb")))
    with pytest.raises(OSError):
        w.close()
