"""Chamfer metric (asdf_chamfer behind alignsdf_amd.deep_sdf.metrics.chamfer) against the reference's own computation
(scipy cKDTree both ways, deep_sdf/metrics/chamfer.py:217-229) on identical point sets."""
import os

import numpy as np
import pytest

from alignsdf_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _sphere_mesh(nu, nv, centre, radius):
    th, ph = np.meshgrid(np.arange(nu) * 2 * np.pi / nu, (np.arange(nv) + 0.5) * np.pi / nv, indexing="ij")
    P = np.stack([np.sin(ph) * np.cos(th), np.sin(ph) * np.sin(th), np.cos(ph)], -1).reshape(-1, 3) * radius + centre
    idx = lambda i, j: (i % nu) * nv + j
    F = [(idx(i, j), idx(i + 1, j), idx(i + 1, j + 1)) for i in range(nu) for j in range(nv - 1)] + \
        [(idx(i, j), idx(i + 1, j + 1), idx(i, j + 1)) for i in range(nu) for j in range(nv - 1)]
    return P, np.array(F)


@pytest.mark.parametrize("na,nb", [(1, 1), (1, 700), (513, 511), (4097, 3000), (30000, 30000)])
def test_chamfer_matches_ckdtree(na, nb):
    from alignsdf_amd.deep_sdf.metrics.chamfer import chamfer_distance
    from oracle.icp_oracle import chamfer_sum
    a = syn.normal((na, 3), 31 + na) * 7.0
    b = syn.normal((nb, 3), 32 + nb) * 7.0 + 0.5
    a_to_b, b_to_a = chamfer_distance(a, b)
    gt_to_gen, gen_to_gt = chamfer_sum(a, b)             # source = a: gen_to_gt is a -> b
    assert abs(a_to_b - gen_to_gt) <= 1e-12 * max(1.0, gen_to_gt)
    assert abs(b_to_a - gt_to_gen) <= 1e-12 * max(1.0, gt_to_gen)


def test_chamfer_of_identical_and_shifted_sets():
    from alignsdf_amd.deep_sdf.metrics.chamfer import chamfer_distance
    a = syn.uniform((5000, 3), 77, -1.0, 1.0)
    assert chamfer_distance(a, a) == (0.0, 0.0)
    d = np.array([1e-3, 0.0, 0.0])                        # far below the sample spacing: every neighbour is its own image
    x, y = chamfer_distance(a, a + d)
    assert abs(x - 1e-6) < 1e-15 and abs(y - 1e-6) < 1e-15


def test_compute_trimesh_chamfer_on_files(tmp_path):
    """File-level entry point: a sphere against a scaled + shifted copy; with `optim` the ICP removes the misfit."""
    from alignsdf_amd.deep_sdf.metrics.chamfer import compute_trimesh_chamfer
    from alignsdf_amd.icp import load_obj, sample_surface
    from alignsdf_amd.ply import read_ply, write_ply
    from oracle.icp_oracle import chamfer_sum
    gv, gf = _sphere_mesh(64, 32, np.array([0.0, 0.0, 0.0]), 0.10)              # metres: a 10 cm sphere
    pv = gv * 1.1 + np.array([0.01, -0.005, 0.0])
    gt = str(tmp_path / "gt.obj")
    with open(gt, "w") as f:
        f.write("".join("v %.9f %.9f %.9f\n" % tuple(p) for p in gv) + "".join("f %d %d %d\n" % tuple(t + 1) for t in gf))
    pred = str(tmp_path / "pred_hand.ply")
    write_ply(pred, pv, gf)
    plain = compute_trimesh_chamfer(gt, pred, optim=False)
    ps = sample_surface(np.asarray(read_ply(pred)[0], np.float64), gf, 30000, 0) * 100.0      # what the files hold
    pt = sample_surface(load_obj(gt)[0], gf, 30000, 1) * 100.0
    want = sum(chamfer_sum(ps, pt))
    assert abs(plain - want) <= 1e-9 * want
    aligned = compute_trimesh_chamfer(gt, pred, optim=True)
    assert aligned < 0.05 * plain and aligned > 0.0
    with pytest.raises(NotImplementedError):
        compute_trimesh_chamfer(gt, pred, optim=True, rot=True)


def test_evaluate_writes_reference_summary(tmp_path):
    """evaluate.py's Chamfer branch over a directory of meshes: per-mesh lines sorted by decreasing distance, then the
    mean / median / failure-count lines of evaluate.py:302-316."""
    from alignsdf_amd import evaluate as ev
    from alignsdf_amd.ply import write_ply
    gv, gf = _sphere_mesh(48, 24, np.zeros(3), 0.08)
    meshes = tmp_path / "exp" / "Eval_obman" / "meshes"
    gt_dir = tmp_path / "data" / "obman" / "test" / "mesh_hand"
    os.makedirs(meshes), os.makedirs(gt_dir)
    for k, shift in enumerate((0.0, 0.004, 0.002)):
        write_ply(str(meshes / ("%08d_hand.ply" % k)), gv + np.array([shift, 0, 0]), gf)
        write_ply(str(meshes / ("%08d_obj.ply" % k)), gv, gf)
        if k < 2 or shift == 0.002:
            with open(gt_dir / ("%08d.obj" % k), "w") as f:
                f.write("".join("v %.9f %.9f %.9f\n" % tuple(p) for p in gv) + "".join("f %d %d %d\n" % tuple(t + 1) for t in gf))
    write_ply(str(meshes / "00000009_hand.ply"), gv, gf)                      # no ground truth: counted as a failure
    summary, n_pred = ev.evaluate(str(tmp_path / "exp"), str(tmp_path / "data" / "obman" / "test"), "obman")
    assert n_pred == 4 and len(summary) == 3
    path = ev.write_summary(str(tmp_path / "exp"), "obman", summary, n_pred)
    lines = open(path).read().splitlines()
    assert lines[0] == "summary of chamfer_dist" and path.endswith("chamfer_hand.txt")
    order = [l.split(",")[0] for l in lines[1:4]]
    assert order == ["00000001", "00000002", "00000000"]                       # 4 mm, 2 mm, 0 mm shift
    vals = [float(l.split(",")[1]) for l in lines[1:4]]
    assert lines[4] == "mean chamfer distance:{}".format(np.mean(vals)) and lines[5] == "median chamfer distance:{}".format(np.median(vals))
    assert lines[-1] == "failure count:1"
    # the reference's worker signature (evaluate.py:19): one queue.put([(name, chamfer, joints, verts)]) per mesh
    import queue
    q = queue.Queue()
    ev.evaluate_queue(q, str(tmp_path / "exp"), str(tmp_path / "data" / "obman" / "test"), 0, None, False, False, False, False, False, False, "obman")
    got = []
    while not q.empty():
        got += q.get()
    assert sorted(r[0] for r in got) == ["00000000", "00000001", "00000002"] and {r[0]: r[1] for r in got} == {r[0]: r[1] for r in summary}
    with pytest.raises(NotImplementedError):
        ev.evaluate_queue(q, str(tmp_path / "exp"), str(tmp_path / "data"), 0, None, False, True, False, False, False, False, "obman")


def test_chamfer_abi_errors(native_lib):
    import ctypes
    import torch
    a = torch.zeros((4, 3), dtype=torch.float64, device="cuda")
    ws = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
    res = (ctypes.c_double * 2)()
    L = native_lib
    assert L.asdf_chamfer(None, 4, a.data_ptr(), 4, ws.data_ptr(), ws.numel(), res, None) == -1
    assert L.asdf_chamfer(a.data_ptr(), 0, a.data_ptr(), 4, ws.data_ptr(), ws.numel(), res, None) == -1
    assert L.asdf_chamfer(a.data_ptr(), 4, a.data_ptr(), 4, ws.data_ptr(), 16, res, None) == -5
    assert L.asdf_chamfer(a.data_ptr(), 4, a.data_ptr(), 4, ws.data_ptr(), ws.numel(), None, None) == -1
    assert L.asdf_chamfer(a.data_ptr(), 4, a.data_ptr(), 4, ws.data_ptr(), ws.numel(), res, None) == 0 and res[0] == 0.0
