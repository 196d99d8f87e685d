"""Eval mode's ground-truth worker (alignsdf_amd/gt_worker.py + reconstruct.GroundTruthPrefetcher): the worker process returns the
samples the in-process sampler returns, reports missing and malformed files like it, never imports torch, and is replaced when
it has died.  (The reference reads and samples the ground-truth mesh inline: utils/mesh.py:386-389, icp_trans_scale.py:19-23.)"""
import os
import subprocess
import sys

import numpy as np
import pytest

from alignsdf_amd import reconstruct as rc
from alignsdf_amd.surface_sampling import load_obj, sample_surface

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def write_tetra(path, scale=1.0):
    with open(path, "w") as f:
        for v in ((0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1)):
            f.write("v %r %r %r\n" % tuple(scale * c for c in v))
        f.write("f 1 2 3\nf 1 2 4\nf 1 3 4\nf 2 3 4\n")


@pytest.fixture()
def data_root(tmp_path):
    d = tmp_path / "obman" / "test" / "mesh_hand"
    d.mkdir(parents=True)
    for i in range(4):
        write_tetra(str(d / ("%08d.obj" % i)), 1.0 + i)
    (d / "00000009.obj").write_text("v 0 0 0\nv 1 0 x\nf 1 2 3\n")
    return str(tmp_path)


@pytest.mark.parametrize("mode", ["process", "thread"])
def test_worker_returns_the_in_process_samples(data_root, monkeypatch, mode):
    monkeypatch.setenv("ASDF_GT_WORKER", mode)
    g = rc.GroundTruthPrefetcher("obman", data_root, samples=2000, seed=1)
    try:
        assert (g.proc is not None) == (mode == "process")
        names = ["/out/%08d_hand.ply" % i for i in range(4)]
        for n in names:                       # all four in flight at once: replies come back in request order
            g.prefetch(n)
        for i, n in enumerate(names):
            want = sample_surface(*load_obj(os.path.join(data_root, "obman", "test", "mesh_hand", "%08d.obj" % i)), 2000, 1)
            got = g.get(n).numpy()
            assert got.dtype == np.float64 and np.array_equal(got, want)
        with pytest.raises(FileNotFoundError):
            g.get("/out/00000007_hand.ply")
        with pytest.raises((ValueError, RuntimeError), match="could not convert"):
            g.get("/out/00000009_hand.ply")
        assert np.array_equal(g.get(names[2]).numpy(), sample_surface(*load_obj(os.path.join(data_root, "obman", "test", "mesh_hand", "00000002.obj")), 2000, 1))
    finally:
        g.close()


def test_missing_file_allowed(data_root, monkeypatch):
    monkeypatch.setenv("ASDF_GT_WORKER", "process")
    g = rc.GroundTruthPrefetcher("obman", data_root, allow_missing_gt=True, samples=100)
    try:
        assert g.get("/out/00000007_hand.ply") is None
    finally:
        g.close()


def test_worker_is_shared_and_replaced_when_dead(data_root, monkeypatch):
    monkeypatch.setenv("ASDF_GT_WORKER", "process")
    a = rc.GroundTruthPrefetcher("obman", data_root, samples=100)
    b = rc.GroundTruthPrefetcher("obman", data_root, samples=100)
    assert a.proc is b.proc                                   # one worker per process, started once
    first = a.get("/out/00000001_hand.ply").clone()
    a.close(); b.close()
    assert a.proc.poll() is None                              # close() leaves the shared worker running
    a.proc.kill(); a.proc.wait()
    c = rc.GroundTruthPrefetcher("obman", data_root, samples=100)
    try:
        assert c.proc is not a.proc and c.proc.poll() is None
        assert np.array_equal(c.get("/out/00000001_hand.ply").numpy(), first.numpy())
    finally:
        c.close()


def test_worker_process_does_not_import_torch():
    """The worker must start in a fraction of a second: numpy only."""
    code = ("import sys; import alignsdf_amd.gt_worker as g; g.load_samples('/nonexistent', 10, 1); "
            "import alignsdf_amd.surface_sampling; print('torch' in sys.modules)")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip() == "False", out.stderr
