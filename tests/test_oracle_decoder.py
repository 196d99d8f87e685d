"""Pin the CPU decoder oracle (oracle/sdf_oracle.py) against outputs of the reference itself
(tests/golden/ref_*.npz, made by tests/golden/make_ref_goldens.py in the build container)."""
import numpy as np
import pytest
import torch

from alignsdf_amd import synthetic as syn
from oracle import sdf_oracle as orc


def _inputs(tag):
    specs, sd = syn.specs_for(tag), syn.full_state_dict(tag)
    lat = torch.from_numpy(syn.latent_code(0))
    mano = obj = None
    if specs["EncodeStyle"] != "nerf":
        m, o = syn.pose_inputs(0)
        mano = {k: torch.from_numpy(v) for k, v in m.items()}
        obj = {k: torch.from_numpy(v) for k, v in o.items()}
    return specs, sd, lat, mano, obj


# round-1 fixtures + the round-2 ones (tests/golden/make_r2_goldens.py): two NeRF octaves, EncodeStyle "hand" with the wrist
# joint only / all 16 joints (utils/utils.py:399-400), EncodeStyle "obj"
TAGS = ["nerf3", "both9", "comb3", "nerf9", "nerf15", "hand6", "hand51", "obj6"]


def test_grid_columns_bit_exact(golden_dir):
    g = np.load(golden_dir + "/ref_grid.npz")
    for N in (4, 5, 7, 16, 31, 32, 64, 100):
        sel = g["sel_%d" % N]
        assert np.array_equal(orc.grid_indices(N).numpy()[sel], g["idx_%d" % N]), N
        assert np.array_equal(orc.grid_coords(N, 2.0 / (N - 1), [-1, -1, -1]).numpy()[sel], g["coord_%d" % N]), N


def test_sheared_closed_form():
    """under true division axis 1 = y + z/N and axis 0 = x + y/N + z/N^2 (exact in fp32 for 2^k)."""
    N = 32
    idx = orc.grid_indices(N).double().numpy()
    i = np.arange(N ** 3)
    z, y, x = i % N, (i // N) % N, i // (N * N)
    assert np.array_equal(idx[:, 1], y + z / N) and np.array_equal(idx[:, 0], x + y / N + z / N ** 2)


@pytest.mark.parametrize("tag", ["nerf3", "both9", "comb3", "nerf9"])
def test_effective_weights_match_module_hook(tag, golden_dir):
    g = np.load("%s/ref_decoder_%s.npz" % (golden_dir, tag))
    sd = syn.full_state_dict(tag)
    for head in ("",) if tag == "comb3" else "ho":
        params = orc.combined_params(sd) if tag == "comb3" else orc.effective_head_params(sd, head)
        for layer in range(4):
            assert np.array_equal(params[layer][0].numpy()[:4], g["effw_%s%d_rows" % (head, layer)])


@pytest.mark.parametrize("tag", TAGS)
def test_decoder_and_embedding_vs_reference(tag, golden_dir):
    g = np.load("%s/ref_decoder_%s.npz" % (golden_dir, tag))
    specs, sd, lat, mano, obj = _inputs(tag)
    pts = torch.from_numpy(g["rand_pts"])
    if "embed_pts" in g.files:       # kinematic (both9) or NeRF (nerf9) point features
        e = orc.point_features(pts, specs, mano, obj).numpy()
        assert e.shape == g["embed_pts"].shape and np.abs(e - g["embed_pts"]).max() <= 1e-6
    h, o = orc.decode_points(sd, lat, pts, specs, mano, obj)
    assert np.abs(h.numpy() - g["rand_hand"]).max() <= 1e-6
    assert np.abs(o.numpy() - g["rand_obj"]).max() <= 1e-6


@pytest.mark.parametrize("tag", TAGS)
def test_two_pass_flow_vs_reference(tag, golden_dir):
    """Full create_mesh_combined_decoder restatement at N=32: volumes, bbox, zoom cube."""
    g = np.load("%s/ref_decoder_%s.npz" % (golden_dir, tag))
    specs, sd, lat, mano, obj = _inputs(tag)
    r = orc.two_pass_volumes(sd, lat, specs, 32, mano, obj)
    assert np.abs(r["vol_hand1"].numpy() - g["vol1_hand_32"]).max() <= 1e-6
    assert np.abs(r["vol_obj1"].numpy() - g["vol1_obj_32"]).max() <= 1e-6
    assert np.array_equal(r["bbox"], g["bbox_32"])
    assert np.array_equal(r["new_voxel_size"].numpy().reshape(1), g["new_voxel_size_32"])
    assert np.array_equal(r["new_origin"].numpy(), g["new_origin_32"])
    assert np.abs(r["vol_hand2"].numpy() - g["vol2_hand_32"]).max() <= 1e-6
    assert np.abs(r["vol_obj2"].numpy() - g["vol2_obj_32"]).max() <= 1e-6
    assert np.allclose(g["mc_origin_32"], r["new_origin"].numpy().astype(np.float64), rtol=0, atol=0)


def test_zoom_cube_empty_and_single_branch():
    """Empty coarse pass -> zero bbox (utils/mesh.py:209-211); single-branch selection (:239-244)."""
    N = 16
    vs = 2.0 / (N - 1)
    pos = torch.ones(N, N, N)
    nvs, norg, bbox = orc.get_higher_res_cube(True, True, pos, pos, N, vs)
    assert np.isclose(nvs.item(), np.float32(4 * np.float32(vs)) / 15, rtol=1e-6)
    assert np.allclose(norg.numpy(), np.float32(-2 * vs - 1.0))
    assert (bbox == -1).all()
    neg = pos.clone()
    neg[3:6, 4:9, 10:12] = -1
    nvs, norg, bbox = orc.get_higher_res_cube(True, False, neg, None, N, vs)
    assert list(bbox[0]) == [3, 4, 10, 5, 8, 11]
    assert np.allclose(norg.numpy(), (np.array([3, 4, 10], np.float32) - 2) * np.float32(vs) - 1)


def test_legacy_create_mesh_volume(golden_dir):
    g = np.load(golden_dir + "/ref_legacy.npz")
    specs, sd, lat, _, _ = _inputs("nerf3")
    hp, op = orc.effective_head_params(sd, "h"), orc.effective_head_params(sd, "o")
    fn = lambda x: orc.separate_decoder(hp, op, x, 256, 3, "nerf")[0]
    vol = orc.legacy_volume(fn, lat, 32)
    assert np.abs(vol.numpy() - g["vol_32"]).max() <= 1e-6
    assert list(g["origin"]) == [-1, -1, -1] and g["voxel_size"][0] == 2.0 / 31
