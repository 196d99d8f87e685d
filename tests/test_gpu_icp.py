"""GPU parity of the eval-mode ICP (K7) against the reference class' goldens and the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", ["small", "noisy", "full30k"])
def test_icp_vs_reference_goldens(case, golden_dir):
    from alignsdf_amd.icp import icp_trans_scale
    g = np.load(golden_dir + "/ref_icp.npz")
    r = icp_trans_scale(g[case + ".src"], g[case + ".tgt"], g[case + ".verts"])
    assert abs(r["scale"] - g[case + ".scale"][0]) <= 1e-9
    assert np.abs(r["trans"] - g[case + ".trans"]).max() <= 1e-9
    assert abs(r["all_scale"] - g[case + ".all_scale"][0]) <= 1e-9
    assert np.abs(r["all_trans"] - g[case + ".all_trans"]).max() <= 1e-9
    assert np.abs(r["vertices"] - g[case + ".verts_out"]).max() <= 1e-9


def test_icp_iteration_count_and_ragged_sizes():
    from alignsdf_amd import synthetic as syn
    from alignsdf_amd.icp import icp_trans_scale
    from oracle import icp_oracle
    for ns, nt, seed in ((2, 5, 1), (129, 1023, 2), (1025, 77, 3), (3000, 2999, 4)):
        tgt = syn.normal((nt, 3), 50 + seed) * np.array([0.1, 0.06, 0.04]) + 0.3
        src = (syn.normal((ns, 3), 60 + seed) * np.array([0.1, 0.06, 0.04]) + 0.3 - 0.02) / 1.1
        verts = syn.normal((40, 3), 70 + seed)
        r = icp_trans_scale(src, tgt, verts, max_iter=25)
        ref = icp_oracle.icp_trans_scale(src, tgt, verts, max_iter=25)
        assert r["iterations"] == ref["iterations"]
        assert abs(r["scale"] - ref["scale"]) <= 1e-9 and np.abs(r["trans"] - ref["trans"]).max() <= 1e-9
        assert abs(r["error"] - ref["errors"][-1]) <= 1e-12


def test_eval_mode_alignment_recovers_known_transform(tmp_path):
    """Mesh-level flow: OBJ ground truth, seeded surface sampling, ICP, transformed vertices."""
    from alignsdf_amd.icp import align_to_ground_truth, load_obj
    from oracle import mc33
    n = 40
    g = np.stack(np.meshgrid(*[np.linspace(-1, 1, n)] * 3, indexing="ij"), -1)
    vol = (np.linalg.norm(g * np.array([1.0, 1.4, 2.0]), axis=-1) - 0.6).astype(np.float32)
    v, f = mc33.marching_cubes_raw(vol)
    gt_v = v * 0.01 + np.array([0.1, -0.05, 0.4])
    with open(tmp_path / "gt.obj", "w") as fh:
        for p in gt_v:
            fh.write("v %.9f %.9f %.9f\n" % tuple(p))
        for t in f:
            fh.write("f %d %d %d\n" % tuple(t + 1))
    lv, lf = load_obj(str(tmp_path / "gt.obj"))
    assert lv.shape == gt_v.shape and np.array_equal(lf, f)
    pred = (gt_v - np.array([0.01, 0.02, -0.015])) / 1.12           # predicted mesh = GT under a similarity
    aligned, trans, scale, info = align_to_ground_truth(pred, f, lv, lf, samples=20000)
    assert abs(scale - 1.12) < 5e-3 and np.abs(aligned - gt_v).max() < 1e-3
    assert info["iterations"] <= 100
