"""GPU parity of the HIP decoder (K0 + K1) against the CPU oracle and the reference goldens.

Tolerance: north_star asks for SDF values within 1e-5 (fp32) of the reference decoder."""
import numpy as np
import pytest
import torch

from alignsdf_amd import synthetic as syn

pytestmark = pytest.mark.gpu
TOL = 1e-5
# every decoder configuration with reference goldens: ObMan (nerf3), DexYCB MANO-aligned (both9), CombinedDecoder (comb3), NeRF
# encoding with one / two octaves (nerf9 / nerf15), hand alignment with the wrist joint / all 16 joints (hand6 / hand51,
# utils/utils.py:399-400), object alignment (obj6)
TAGS = ["nerf3", "both9", "comb3", "nerf9", "nerf15", "hand6", "hand51", "obj6"]


def _setup(tag):
    from alignsdf_amd.hip_decoder import HipSdfDecoder, kinematic_affine
    specs = syn.specs_for(tag)
    sd = syn.full_state_dict(tag)
    dec = HipSdfDecoder(sd, 256, specs["PointFeatSize"], specs["EncodeStyle"], device="cuda:0")
    lat = torch.from_numpy(syn.latent_code(0))
    mano = obj = emb = None
    if specs["EncodeStyle"] != "nerf":
        m, o = syn.pose_inputs(0)
        mano = {k: torch.from_numpy(v) for k, v in m.items()}
        obj = {k: torch.from_numpy(v) for k, v in o.items()}
        emb = kinematic_affine(specs["PointFeatSize"], specs["EncodeStyle"], specs["SdfScaleFactor"], mano, obj)
    dec.set_sample(lat, emb)
    return dec, specs, sd, lat, mano, obj


@pytest.mark.parametrize("tag", TAGS)
def test_points_vs_reference_golden(tag, golden_dir):
    dec, *_ = _setup(tag)
    g = np.load("%s/ref_decoder_%s.npz" % (golden_dir, tag))
    h, o = dec.decode_points(torch.from_numpy(g["rand_pts"]))
    assert np.abs(h.cpu().numpy() - g["rand_hand"]).max() <= TOL
    assert np.abs(o.cpu().numpy() - g["rand_obj"]).max() <= TOL


@pytest.mark.parametrize("tag", ["nerf3", "both9", "comb3", "nerf9"])
@pytest.mark.parametrize("M", [0, 1, 31, 32, 33, 127, 129, 1000, 40000])
def test_points_ragged_vs_oracle(tag, M):
    from oracle import sdf_oracle as orc
    dec, specs, sd, lat, mano, obj = _setup(tag)
    pts = torch.from_numpy(syn.uniform((M, 3), 31 + M, -1.0, 1.0).astype(np.float32))
    h, o = dec.decode_points(pts)
    torch.cuda.synchronize()
    assert h.shape == (M,) and o.shape == (M,)
    if M:
        rh, ro = orc.decode_points(sd, lat, pts, specs, mano, obj)
        assert (h.cpu() - rh).abs().max().item() <= TOL
        assert (o.cpu() - ro).abs().max().item() <= TOL


@pytest.mark.parametrize("tag", TAGS)
def test_grid_pass1_vs_reference_golden(tag, golden_dir):
    """Pass 1 on [-1,1]^3 at N=32 (full volume) and N=64 (8192 probes), incl. the negative-voxel bbox."""
    dec, *_ = _setup(tag)
    g = np.load("%s/ref_decoder_%s.npz" % (golden_dir, tag))
    for N in (32, 64):
        h, o, bbox = dec.decode_grid(N, [-1.0, -1.0, -1.0], 2.0 / (N - 1))
        h, o, bbox = h.cpu().numpy(), o.cpu().numpy(), bbox.cpu().numpy()
        sel = g["probe_sel_%d" % N]
        assert np.abs(h.reshape(-1)[sel] - g["p1_hand_%d" % N]).max() <= TOL
        assert np.abs(o.reshape(-1)[sel] - g["p1_obj_%d" % N]).max() <= TOL
        if N == 32:
            assert np.abs(h - g["vol1_hand_32"]).max() <= TOL
            assert np.abs(o - g["vol1_obj_32"]).max() <= TOL
        # bbox of negative voxels must equal the reference's nonzero/min/max unless a value sits within TOL of 0
        for k, vol in enumerate((h, o)):
            nz = np.argwhere(vol < 0)
            assert bbox[k * 8 + 6] == len(nz)
            assert list(bbox[k * 8: k * 8 + 3]) == list(nz.min(0)) and list(bbox[k * 8 + 3: k * 8 + 6]) == list(nz.max(0))
        ref_bbox = g["bbox_%d" % N]
        got = np.stack([bbox[0:6], bbox[8:14]])
        assert np.array_equal(got, ref_bbox), (got, ref_bbox)


@pytest.mark.parametrize("tag", TAGS)
def test_grid_pass2_vs_reference_golden(tag, golden_dir):
    """Pass 2 inside the reference's own zoom cube (so only the decoder is compared)."""
    dec, *_ = _setup(tag)
    g = np.load("%s/ref_decoder_%s.npz" % (golden_dir, tag))
    for N in (32, 64):
        h, o, _ = dec.decode_grid(N, g["new_origin_%d" % N], g["new_voxel_size_%d" % N][0])
        h, o = h.cpu().numpy(), o.cpu().numpy()
        sel = g["probe_sel_%d" % N]
        assert np.abs(h.reshape(-1)[sel] - g["p2_hand_%d" % N]).max() <= TOL
        assert np.abs(o.reshape(-1)[sel] - g["p2_obj_%d" % N]).max() <= TOL
        if N == 32:
            assert np.abs(h - g["vol2_hand_32"]).max() <= TOL
            assert np.abs(o - g["vol2_obj_32"]).max() <= TOL


def test_grid_modes_and_odd_sizes():
    """Non-power-of-two N, both index modes, vs the oracle's coordinate generator + decoder."""
    from oracle import sdf_oracle as orc
    dec, specs, sd, lat, mano, obj = _setup("nerf3")
    for N, integer in ((5, False), (7, True), (31, False), (20, True)):
        vs = 2.0 / (N - 1)
        h, o, _ = dec.decode_grid(N, [-1.0, -0.5, 0.25], vs, grid_mode=1 if integer else 0)
        c = orc.grid_coords(N, vs, [-1.0, -0.5, 0.25], integer_mode=integer)
        rh, ro = orc.decode_points(sd, lat, c, specs)
        assert (h.cpu().reshape(-1) - rh).abs().max().item() <= TOL
        assert (o.cpu().reshape(-1) - ro).abs().max().item() <= TOL


def test_linearity_of_lattice_full_size():
    """Size-independent property at N=128: the grid sweep equals the point-list sweep on the same coordinates - bit for bit
    on the fp32 chain (the same kernel either way), within 2e-6 for the split-half grid sweep (point lists stay fp32)."""
    from oracle import sdf_oracle as orc
    dec, specs, sd, lat, *_ = _setup("nerf3")
    N = 128
    c = orc.grid_coords(N, 2.0 / (N - 1), [-1, -1, -1])
    sel = torch.from_numpy(np.sort((syn.splitmix64(np.arange(50000, dtype=np.uint64), 5) % np.uint64(N ** 3)).astype(np.int64)))
    hp, op = dec.decode_points(c[sel])
    default = dec.math
    for math in ("f32", "f16x3"):
        dec.set_math(math)
        h, o, bbox = dec.decode_grid(N, [-1.0, -1.0, -1.0], 2.0 / (N - 1))
        if math == "f32":
            assert torch.equal(h.reshape(-1)[sel.cuda()], hp) and torch.equal(o.reshape(-1)[sel.cuda()], op)
        else:
            assert (h.reshape(-1)[sel.cuda()] - hp).abs().max().item() <= 2e-6 and (o.reshape(-1)[sel.cuda()] - op).abs().max().item() <= 2e-6
        assert int(bbox[6]) == int((h < 0).sum()) and int(bbox[14]) == int((o < 0).sum())
        assert int(bbox[7]) == 0 and int(bbox[15]) == 0
    dec.set_math(default)


@pytest.mark.parametrize("tag", ["nerf3", "both9", "nerf9"])
def test_single_head_evaluation_matches_both_heads(tag):
    """A head whose output is not requested is skipped; the other one must be bit-identical to the two-head run."""
    dec, *_ = _setup(tag)
    N = 40
    h, o, b = dec.decode_grid(N, [-1.0, -1.0, -1.0], 2.0 / (N - 1))
    h1, none_o, b1 = dec.decode_grid(N, [-1.0, -1.0, -1.0], 2.0 / (N - 1), obj=False)
    none_h, o2, b2 = dec.decode_grid(N, [-1.0, -1.0, -1.0], 2.0 / (N - 1), hand=False)
    assert none_o is None and none_h is None
    assert torch.equal(h1, h) and torch.equal(o2, o)
    b, b1, b2 = b.cpu().numpy(), b1.cpu().numpy(), b2.cpu().numpy()
    assert list(b1[0:7]) == list(b[0:7]) and b1[14] == 0 and list(b1[8:11]) == [0x7fffffff] * 3
    assert list(b2[8:15]) == list(b[8:15]) and b2[6] == 0


def test_hand_only_two_pass_matches_oracle():
    """HandBranch only: zoom cube from the hand volume alone (utils/mesh.py:239-241), object head not evaluated."""
    from alignsdf_amd.networks.model import build_decoder
    from alignsdf_amd.utils.mesh import decode_two_pass
    from oracle import sdf_oracle as orc
    specs, sd = syn.specs_for("nerf3"), syn.full_state_dict("nerf3")
    dec = build_decoder(specs, {k: torch.from_numpy(v) for k, v in sd.items()})
    lat = torch.from_numpy(syn.latent_code(2))
    r = decode_two_pass(True, False, dec, lat.cuda(), None, None, specs, 32)
    ref = orc.two_pass_volumes(sd, lat, specs, 32, hand_branch=True, obj_branch=False)
    assert r["vol_obj"] is None
    assert torch.equal(r["voxel_size"], ref["new_voxel_size"]) and r["origin"] == ref["new_origin"].tolist()
    assert (r["vol_hand"].cpu() - ref["vol_hand2"]).abs().max().item() <= TOL
