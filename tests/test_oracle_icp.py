"""Pin the ICP oracle against the reference's own ICP_T_S (tests/golden/ref_icp.npz)."""
import numpy as np
import pytest

from oracle import icp_oracle


@pytest.mark.parametrize("case", ["small", "noisy", "full30k"])
def test_icp_matches_reference_class(case, golden_dir):
    g = np.load(golden_dir + "/ref_icp.npz")
    r = icp_oracle.icp_trans_scale(g[case + ".src"], g[case + ".tgt"], g[case + ".verts"])
    assert abs(r["scale"] - g[case + ".scale"][0]) <= 1e-9
    assert np.abs(r["trans"] - g[case + ".trans"]).max() <= 1e-9
    assert abs(r["all_scale"] - g[case + ".all_scale"][0]) <= 1e-9
    assert np.abs(r["all_trans"] - g[case + ".all_trans"]).max() <= 1e-9
    assert np.abs(r["vertices"] - g[case + ".verts_out"]).max() <= 1e-9
    assert 2 <= r["iterations"] <= 100 and r["errors"][-1] < 0.01
