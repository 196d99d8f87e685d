"""The box-only coarse sweep (asdf_decode_grid_box, HipSdfDecoder.coarse_begin / coarse_finish with coarse_mode "box"):
the coarse pass of the two-pass flow (utils/mesh.py:27-63) is consumed only through get_higher_res_cube
(utils/mesh.py:198-256), i.e. through the box of its negative voxels.  One fp16 plane per operand + exact re-evaluation
of every voxel that could move the box must give the SAME boxes - hence the same zoom cube, hence bit-identical
pass-2 volumes - as the ordinary sweep, and must notice when it cannot."""
import numpy as np
import pytest
import torch

from alignsdf_amd import synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _fast_sweeps(monkeypatch):
    """This file is about the OPT-IN audited one-plane sweeps (round 6: the product's default is ordinary sweeps on every voxel;
    ASDF_FAST=1 / --fast / HipSdfDecoder.set_fast select these)."""
    monkeypatch.setenv("ASDF_FAST", "1")


def _decoder(tag):
    from alignsdf_amd.hip_decoder import HipSdfDecoder
    specs = syn.specs_for(tag)
    hip = HipSdfDecoder(syn.full_state_dict(tag), 256, specs["PointFeatSize"], specs["EncodeStyle"])
    hip._test_tag = tag if tag in syn.GRASP_TAGS else None      # (_bind: the grasp family has its own trained latent codes)
    return hip, specs


def _bind(hip, specs, sample):
    from alignsdf_amd.utils.utils import sample_embedding
    tag = {(3, "nerf", "1encoder1decoder"): "comb3"}.get((specs["PointFeatSize"], specs["EncodeStyle"], specs["ModelType"]))
    if tag is None:
        tag = getattr(hip, "_test_tag", None) or ("nerf3" if specs["EncodeStyle"] == "nerf" else "both9")
    lat, m, o = (syn.latent_code(sample), None, None) if tag == "comb3" else syn.sample_inputs(tag, sample)
    if specs["EncodeStyle"] != "nerf" and m is None:
        m, o = syn.pose_inputs(sample)
    mano = {k: torch.from_numpy(v).cuda() for k, v in m.items()} if m is not None else None
    obj = {k: torch.from_numpy(v).cuda() for k, v in o.items()} if o is not None else None
    hip.set_sample(torch.from_numpy(lat).cuda(), sample_embedding(specs, mano, obj, hip.combined))


def _prime_fine(hip, N, origin, voxel):
    """A decoder's FIRST fine pass (and the first after a voided allowance) runs as an ordinary sweep plus a plain one-plane sweep of
    the same zoom lattice, compared voxel for voxel when the record is read (round 5: the whole-lattice comparison on the lattice
    marching cubes consumes); band sweeps follow."""
    _, _, t = hip.fine_begin(N, origin, voxel, mc_only=True)
    assert t["kind"] == "exact" and "compare" in t and not hip.fine_needs_repeat(t)
    assert hip._fine_valid(N) and hip.cert["fine_calibrations"] >= 1


def _boxes(b):
    return [int(v) for v in b[0:6]] + [int(v) for v in b[8:14]] + [int(b[6] != 0), int(b[14] != 0)]


@pytest.mark.parametrize("tag,N", [("nerf3", 64), ("nerf3", 128), ("nerf3", 256), ("comb3", 96), ("hand6", 64), ("obj6", 64)])
def test_boxes_equal_the_ordinary_sweep(tag, N):
    hip, specs = _decoder(tag)
    hip.coarse_mode = "box"
    vs = 2.0 / (N - 1)
    for sample in range(5):
        _bind(hip, specs, sample)
        got = hip.coarse_finish(hip.coarse_begin(N, [-1.0, -1.0, -1.0], vs))
        want = hip.decode_grid(N, [-1.0, -1.0, -1.0], vs)[2].cpu().numpy()
        assert _boxes(got) == _boxes(want), (sample, got, want)
    # the first sample calibrated the allowance on an ordinary sweep, the others ran the one-plane kernel and were accepted
    assert hip.box_stats["box"] == 4 and hip.box_stats["exact"] == 1 and hip.box_stats["fallback"] == 0, hip.box_stats
    assert 0.0 < hip._box_tau < 0.05 and hip.box_stats["max_err"] <= 0.6 * hip._box_tau
    assert hip.range_violations() == 0
    hip.close()


def test_single_branch_and_a_zoom_lattice():
    hip, specs = _decoder("nerf3")
    hip.coarse_mode = "box"
    N = 96
    for sample, (hand, obj) in enumerate([(True, True), (True, False), (False, True), (True, False)]):
        _bind(hip, specs, sample)
        for origin, vs in (([-1.0, -1.0, -1.0], 2.0 / (N - 1)), ([-0.62, -0.36, -0.37], 1.21 / (N - 1))):
            got = hip.coarse_finish(hip.coarse_begin(N, origin, vs, hand=hand, obj=obj))
            want = hip.decode_grid(N, origin, vs, hand=hand, obj=obj)[2].cpu().numpy()
            assert _boxes(got) == _boxes(want)
    assert hip.box_stats["box"] >= 6 and hip.box_stats["fallback"] == 0
    hip.close()


def test_two_pass_flow_is_bit_identical():
    from alignsdf_amd.networks.model import build_decoder
    from alignsdf_amd.utils.mesh import decode_two_pass
    specs = syn.specs_for("nerf3")
    N = 128
    out = {}
    for mode in ("exact", "box"):
        dec = build_decoder(specs, {k: torch.from_numpy(v) for k, v in syn.full_state_dict("nerf3").items()})
        from alignsdf_amd.utils.utils import decoder_for
        hip = decoder_for(dec, specs)
        hip.coarse_mode = mode
        res = []
        for sample in range(3):
            lat = torch.from_numpy(syn.latent_code(sample)).cuda()
            r = decode_two_pass(True, True, dec, lat, None, None, specs, N)
            res.append((r["origin"], float(r["voxel_size"]), r["vol_hand"].clone(), r["vol_obj"].clone()))
        out[mode] = res
        if mode == "box":
            assert hip.box_stats["box"] == 2 and hip.box_stats["fallback"] == 0
    for a, b in zip(out["exact"], out["box"]):
        assert a[0] == b[0] and a[1] == b[1]
        assert torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])


def test_an_allowance_that_is_too_small_is_noticed_and_the_sweep_repeated():
    hip, specs = _decoder("nerf3")
    hip.coarse_mode = "box"
    N = 96
    vs = 2.0 / (N - 1)
    _bind(hip, specs, 0)
    hip.coarse_finish(hip.coarse_begin(N, [-1.0, -1.0, -1.0], vs))          # calibration
    honest = hip._box_tau
    hip._box_tau = honest / 64.0            # pretend the arithmetic were 64 x better than it is
    for sample in range(1, 4):
        _bind(hip, specs, sample)
        got = hip.coarse_finish(hip.coarse_begin(N, [-1.0, -1.0, -1.0], vs))
        want = hip.decode_grid(N, [-1.0, -1.0, -1.0], vs)[2].cpu().numpy()
        assert _boxes(got) == _boxes(want)
        if hip.box_stats["fallback"]:
            break
    # the error seen on the re-evaluated voxels exceeded the allowance's share: that sweep was not trusted, and the allowance was
    # measured again on the whole lattice in the same pass (round 4: a refusal voids the allowance instead of inflating it)
    assert hip.box_stats["fallback"] >= 1 and 0.25 * honest <= hip._box_tau <= 4.0 * honest and hip.certificate()["calibrations"] >= 2
    hip.close()


def test_box_entry_point_argument_checks(native_lib):
    import ctypes
    hip, specs = _decoder("nerf9")          # no sample bound yet: refused (NeRF-encoded decoders have a one-plane kernel since round 4)
    rec = torch.zeros(48, dtype=torch.int32, device="cuda")
    vol = torch.zeros(32 ** 3, dtype=torch.float32, device="cuda")
    org = (ctypes.c_float * 3)(-1.0, -1.0, -1.0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    L = native_lib
    assert L.asdf_decode_grid_box(hip._h, 32, org, ctypes.c_float(2 / 31), 0, ctypes.c_float(1e-3), vol.data_ptr(), vol.data_ptr(), rec.data_ptr(), st) == -1
    _bind(hip, specs, 0)
    assert L.asdf_decode_grid_box(hip._h, 32, org, ctypes.c_float(2 / 31), 0, ctypes.c_float(1e-3), vol.data_ptr(), vol.data_ptr(), rec.data_ptr(), st) == 0
    hip.coarse_mode = "box"
    assert hip._box_usable()
    hip.set_math("f32")
    assert not hip._box_usable()            # the fp32 chain keeps ordinary sweeps
    hip.close()
    hip, specs = _decoder("nerf3")
    _bind(hip, specs, 0)
    for tau in (0.0, -1.0, 0.5, float("nan")):
        assert L.asdf_decode_grid_box(hip._h, 32, org, ctypes.c_float(2 / 31), 0, ctypes.c_float(tau), vol.data_ptr(), vol.data_ptr(), rec.data_ptr(), st) == -1
    assert L.asdf_decode_grid_box(hip._h, 32, org, ctypes.c_float(2 / 31), 0, ctypes.c_float(1e-3), vol.data_ptr(), vol.data_ptr(), None, st) == -1
    assert L.asdf_decode_grid_box(hip._h, 32, org, ctypes.c_float(2 / 31), 0, ctypes.c_float(1e-3), None, None, rec.data_ptr(), st) == -1
    hip.close()


@pytest.mark.parametrize("tag", ["nerf3", "both9", "nerf9"])
def test_sample_pipeline_with_the_box_coarse_pass(tag, monkeypatch):
    """ASDF_COARSE=box through the product's entry points (decoder_for -> pipelined_two_pass): cubes, volumes and meshes of
    every sample are those of the ordinary coarse pass - for the ObMan decoder, the MANO-aligned DexYCB decoder (affine point
    features) and a NeRF-encoded decoder (its own one-plane instantiation since round 4: csrc/k1s_nerf_kernels.hip)."""
    from alignsdf_amd.networks.model import build_decoder
    from alignsdf_amd.reconstruct import pipelined_two_pass, synthetic_code_source
    from alignsdf_amd.utils.utils import decoder_for
    specs = syn.specs_for(tag)
    src = synthetic_code_source(tag, "cuda")
    N = 64
    samples = [(i,) + src("s%d" % i, i) for i in (0, 3, 7, 11, 20, 21)]
    out = {}
    monkeypatch.setenv("ASDF_FINE", "exact")      # (ordinary fine sweeps: the volumes themselves are compared below)
    for mode in ("exact", "box"):
        monkeypatch.setenv("ASDF_COARSE", mode)
        dec = build_decoder(specs, {k: torch.from_numpy(v) for k, v in syn.full_state_dict(tag).items()})
        out[mode] = {k: r for k, r in pipelined_two_pass(dec, specs, iter(samples), N)}
        hip = decoder_for(dec, specs, samples[0][2])
        assert hip.coarse_mode == mode
        if mode == "box":
            assert hip.box_stats["box"] == len(samples) - 1 and hip.box_stats["fallback"] == 0, hip.box_stats
    for k, a in out["exact"].items():
        b = out["box"][k]
        assert a["origin"] == b["origin"] and float(a["voxel_size"]) == float(b["voxel_size"])
        for part in ("hand", "obj"):
            assert torch.equal(a["vol_" + part], b["vol_" + part])
            assert torch.equal(a["verts_" + part], b["verts_" + part]) and torch.equal(a["faces_" + part], b["faces_" + part])


# ---- the narrow-band fine sweep (asdf_decode_grid_band, fine_mode "band"): for volumes that go to marching cubes only

@pytest.mark.parametrize("tag,N", [("nerf3", 64), ("nerf3", 128), ("both9", 96), ("nerf3", 256), ("comb3", 96), ("comb3", 128), ("grasp3", 128)])
def test_band_volumes_give_the_identical_meshes(tag, N):
    """Exact values at every corner of every cell that can be active, the right sign everywhere else: marching cubes must
    return the very same vertices and faces as on the volumes of the ordinary sweep, whose values (split-half arithmetic, fp32
    chain next to the level) the re-evaluated voxels hold.  comb3: the CombinedDecoder (networks/model.py:149-188) - one MLP, both
    columns, ONE list of the cells that can be active in either volume (round 4)."""
    from alignsdf_amd.marching_cubes import marching_cubes_device
    hip, specs = _decoder(tag)
    hip.coarse_mode, hip.fine_mode = "box", "band"
    vs = 2.0 / (N - 1)
    lattice = ([-0.62, -0.36, -0.37], 1.21 / (N - 1))
    for sample in range(4):
        _bind(hip, specs, sample)
        hip.coarse_finish(hip.coarse_begin(N, [-1.0, -1.0, -1.0], vs))            # sample 0: calibrates the allowance
        bh, bo, ticket = hip.fine_begin(N, lattice[0], lattice[1], mc_only=True)
        repeat = hip.fine_needs_repeat(ticket)
        if sample == 0:
            assert ticket["kind"] == "exact" or not repeat
        assert not repeat, hip.band_stats
        sh, so, _ = hip.decode_grid(N, lattice[0], lattice[1])                   # the ordinary (default) volumes
        for b, e in ((bh, sh), (bo, so)):
            assert int(((b < 0) != (e < 0)).sum()) == 0                          # every sign
            vb, fb = marching_cubes_device(b, 0.0)
            ve, fe = marching_cubes_device(e, 0.0)
            assert torch.equal(fb, fe) and torch.equal(vb, ve)                    # the ordinary sweep's mesh, bit for bit
    assert hip.band_stats["band"] >= 3
    assert hip.band_stats["fallback"] == 0 and hip.band_stats["max_err"] <= 0.6 * hip._box_tau
    assert 0 < hip.band_stats["max_marked"] < (1 << 21)
    hip.close()


def test_band_is_used_only_where_the_volumes_go_to_marching_cubes():
    from alignsdf_amd.networks.model import build_decoder
    from alignsdf_amd.utils.mesh import decode_two_pass
    from alignsdf_amd.utils.utils import decoder_for
    specs = syn.specs_for("nerf3")
    dec = build_decoder(specs, {k: torch.from_numpy(v) for k, v in syn.full_state_dict("nerf3").items()})
    hip = decoder_for(dec, specs)
    hip.coarse_mode, hip.fine_mode = "box", "band"
    N = 96
    res = {}
    for mc_only in (False, True, False, True):
        for sample in (0, 1):
            lat = torch.from_numpy(syn.latent_code(sample)).cuda()
            r = decode_two_pass(True, True, dec, lat, None, None, specs, N, mc_only=mc_only)
            res[(mc_only, sample)] = r                      # (the LAST call of each kind: the first mc_only pass compares the zoom lattice as a whole)
    assert hip.band_stats["band"] == 3                      # the mc_only calls (after the two whole-lattice comparisons) ran the band sweep
    for sample in (0, 1):
        a, b = res[(False, sample)], res[(True, sample)]
        assert a["origin"] == b["origin"] and float(a["voxel_size"]) == float(b["voxel_size"])
        # a caller that did not declare mc_only got ordinary volumes: bit-equal to a decoder that never heard of the mode
        plain = decoder_for(build_decoder(specs, {k: torch.from_numpy(v) for k, v in syn.full_state_dict("nerf3").items()}), specs)
        plain_r = decode_two_pass(True, True, plain._module if hasattr(plain, "_module") else dec, torch.from_numpy(syn.latent_code(sample)).cuda(),
                                  None, None, specs, N)
        assert torch.equal(a["vol_hand"], plain_r["vol_hand"]) and torch.equal(a["vol_obj"], plain_r["vol_obj"])
        assert not torch.equal(a["vol_hand"], b["vol_hand"])                      # the band volume differs away from the surface
        assert int(((a["vol_hand"] < 0) != (b["vol_hand"] < 0)).sum()) == 0


def test_band_refused_when_the_allowance_is_understated():
    hip, specs = _decoder("nerf3")
    hip.coarse_mode, hip.fine_mode = "box", "band"
    N = 96
    vs = 2.0 / (N - 1)
    _bind(hip, specs, 0)
    hip.coarse_finish(hip.coarse_begin(N, [-1.0, -1.0, -1.0], vs))
    _prime_fine(hip, N, [-0.62, -0.36, -0.37], 1.21 / (N - 1))
    honest = hip._box_tau
    hip._box_tau = honest / 64.0
    _bind(hip, specs, 1)
    bh, bo, ticket = hip.fine_begin(N, [-0.62, -0.36, -0.37], 1.21 / (N - 1), mc_only=True)
    assert ticket["kind"] == "band" and hip.fine_needs_repeat(ticket)             # error seen > half the allowance: refused
    bh, bo, ticket = hip.fine_begin(N, [-0.62, -0.36, -0.37], 1.21 / (N - 1), mc_only=True)
    assert ticket["kind"] == "exact" and not hip.fine_needs_repeat(ticket)        # the repeat is an ordinary sweep
    eh, eo, _ = hip.decode_grid(N, [-0.62, -0.36, -0.37], 1.21 / (N - 1))
    assert torch.equal(bh, eh) and torch.equal(bo, eo)
    assert hip.band_stats["fallback"] == 1 and not hip._allowance_valid()       # void until the next coarse pass measures again
    hip.close()


def test_band_single_branch_and_through_the_sample_pipeline(monkeypatch):
    from alignsdf_amd.marching_cubes import marching_cubes_device
    from alignsdf_amd.networks.model import build_decoder
    from alignsdf_amd.reconstruct import pipelined_two_pass, synthetic_code_source
    from alignsdf_amd.utils.utils import decoder_for
    # one head only
    hip, specs = _decoder("nerf3")
    hip.coarse_mode, hip.fine_mode = "box", "band"
    N = 96
    lattice = ([-0.62, -0.36, -0.37], 1.21 / (N - 1))
    _bind(hip, specs, 0)
    hip.coarse_finish(hip.coarse_begin(N, [-1.0, -1.0, -1.0], 2.0 / (N - 1)))
    _prime_fine(hip, N, lattice[0], lattice[1])
    for hand, obj in ((True, False), (False, True)):
        _bind(hip, specs, 2)
        bh, bo, ticket = hip.fine_begin(N, lattice[0], lattice[1], hand=hand, obj=obj, mc_only=True)
        assert ticket["kind"] == "band" and not hip.fine_needs_repeat(ticket)
        assert (bh is None) == (not hand) and (bo is None) == (not obj)
        eh, eo, _ = hip.decode_grid(N, lattice[0], lattice[1], hand=hand, obj=obj)
        b, e = (bh, eh) if hand else (bo, eo)
        vb, fb = marching_cubes_device(b, 0.0)
        ve, fe = marching_cubes_device(e, 0.0)
        assert torch.equal(vb, ve) and torch.equal(fb, fe)
    hip.close()
    # the product's pipeline with both one-plane sweeps against the ordinary run (named "fp32" below for history's sake)
    specs = syn.specs_for("nerf3")
    src = synthetic_code_source("nerf3", "cuda")
    samples = [(i,) + src("s%d" % i, i) for i in (0, 4, 8, 15, 16)]
    out = {}
    for name, env in (("fp32", {"ASDF_COARSE": "exact", "ASDF_FINE": "exact"}), ("one_plane", {"ASDF_COARSE": "box", "ASDF_FINE": "band"})):
        for k in ("ASDF_MATH", "ASDF_COARSE", "ASDF_FINE"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        dec = build_decoder(specs, {k: torch.from_numpy(v) for k, v in syn.full_state_dict("nerf3").items()})
        out[name] = {k: r for k, r in pipelined_two_pass(dec, specs, iter(samples), 64)}
        if name == "one_plane":
            hipd = decoder_for(dec, specs)
            # (the coarse pass of sample 0 calibrated on the coarse lattice, its fine pass on the zoom lattice: an ordinary sweep)
            assert hipd.band_stats["band"] == len(samples) - 1 and hipd.band_stats["exact"] == 1 and hipd.band_stats["fallback"] == 0, hipd.band_stats
            assert hipd.cert["fine_calibrations"] == 1 and hipd.cert["calibrations"] == 1
    for k, a in out["fp32"].items():
        b = out["one_plane"][k]
        assert a["origin"] == b["origin"] and float(a["voxel_size"]) == float(b["voxel_size"])
        for part in ("hand", "obj"):
            assert torch.equal(a["verts_" + part], b["verts_" + part]) and torch.equal(a["faces_" + part], b["faces_" + part])


def test_one_plane_sweeps_over_all_64_synthetic_samples():
    """Every synthetic sample (the 64 the bench cycles through): box-only coarse sweep = the ordinary boxes, narrow-band
    fine sweep = the ordinary sweep's meshes, no sweep refused."""
    from alignsdf_amd.marching_cubes import marching_cubes_device
    from alignsdf_amd.utils.mesh import zoom_cube_from_bboxes
    hip, specs = _decoder("nerf3")
    hip.coarse_mode, hip.fine_mode = "box", "band"
    N = 96
    vs = 2.0 / (N - 1)
    checked = 0
    for sample in range(64):
        _bind(hip, specs, sample)
        b = hip.coarse_finish(hip.coarse_begin(N, [-1.0, -1.0, -1.0], vs))
        w = hip.decode_grid(N, [-1.0, -1.0, -1.0], vs)[2].cpu().numpy()
        assert _boxes(b) == _boxes(w), sample
        nvs, norg = zoom_cube_from_bboxes([(b[0:3], b[3:6], int(b[6])), (b[8:11], b[11:14], int(b[14]))], N, vs)
        bh, bo, ticket = hip.fine_begin(N, norg.tolist(), nvs.item(), mc_only=True)
        assert not hip.fine_needs_repeat(ticket), (sample, hip.band_stats)
        eh, eo, _ = hip.decode_grid(N, norg.tolist(), nvs.item())
        for bv, ev in ((bh, eh), (bo, eo)):
            vb, fb = marching_cubes_device(bv, 0.0)
            ve, fe = marching_cubes_device(ev, 0.0)
            assert torch.equal(vb, ve) and torch.equal(fb, fe), sample
            checked += 1
    assert checked == 128 and hip.box_stats["fallback"] == 0 and hip.band_stats["fallback"] == 0
    assert hip.box_stats["box"] == 63 and hip.band_stats["band"] == 63 and hip.band_stats["exact"] == 1      # sample 0: both lattices compared as a whole
    c = hip.certificate()
    assert c["calibrations"] == 1 and c["fine_calibrations"] == 1 and c["fine_tail_ratio"] <= 3.0 and c["fine_lattice_max_error"] > 0
    hip.close()


@pytest.mark.parametrize("name", ["spread1e4", "huge", "tiny", "heavy_tails", "gain30", "latent_x10"])
def test_one_plane_sweeps_on_adversarial_decoders(name):
    """The decoders that attack the split-half arithmetic (tests/test_gpu_split_half_adversarial.py), through the one-plane
    sweeps: whatever the one-plane error is on them, the outcome is either accepted AND right (boxes of the ordinary sweep,
    meshes of the ordinary sweep) or refused and repeated as an ordinary sweep - never a silently different cube or mesh."""
    from alignsdf_amd.hip_decoder import HipSdfDecoder
    from alignsdf_amd.marching_cubes import marching_cubes_device
    from tests.test_gpu_split_half_adversarial import variant
    sd, lat = variant(name)
    hip = HipSdfDecoder(sd, 256, 3, "nerf")
    hip.coarse_mode, hip.fine_mode = "box", "band"
    hip.set_sample(torch.from_numpy(lat).cuda())
    N = 64
    vs = 2.0 / (N - 1)
    for rep in range(3):
        b = hip.coarse_finish(hip.coarse_begin(N, [-1.0, -1.0, -1.0], vs))
        w = hip.decode_grid(N, [-1.0, -1.0, -1.0], vs)[2].cpu().numpy()
        assert _boxes(b) == _boxes(w), (name, rep, hip.box_stats)
        for lattice in (([-1.0, -1.0, -1.0], vs), ([-0.9, -0.8, -0.85], 1.7 / (N - 1))):
            for _ in range(4):
                bh, bo, ticket = hip.fine_begin(N, lattice[0], lattice[1], mc_only=True)
                if not hip.fine_needs_repeat(ticket):
                    break
            eh, eo, _ = hip.decode_grid(N, lattice[0], lattice[1])
            for bv, ev in ((bh, eh), (bo, eo)):
                assert int(((bv < 0) != (ev < 0)).sum()) == 0, (name, ticket["kind"])
                try:
                    ve, fe = marching_cubes_device(ev, 0.0)
                except (ValueError, RuntimeError):
                    continue                          # no surface on this lattice
                vb, fb = marching_cubes_device(bv, 0.0)
                assert torch.equal(fb, fe), (name, ticket["kind"])
                if ticket["kind"] == "band":
                    assert torch.equal(vb, ve)        # the ordinary sweep's vertices exactly
    print(name, "box", hip.box_stats, "band", hip.band_stats, "allowance", hip._box_tau, "modes", hip.coarse_mode, hip.fine_mode, hip.math)
    hip.close()


@pytest.mark.parametrize("N", [17, 31, 50, 66])
def test_one_plane_sweeps_on_lattices_that_are_not_multiples_of_four(N):
    """Row lengths that are not a multiple of 4 (scalar paths of the candidate / mark / compact kernels, ragged last tiles)."""
    from alignsdf_amd.marching_cubes import marching_cubes_device
    hip, specs = _decoder("nerf3")
    hip.coarse_mode, hip.fine_mode = "box", "band"
    lattice = ([-0.62, -0.36, -0.37], 1.21 / (N - 1))
    for sample in range(3):
        _bind(hip, specs, sample)
        b = hip.coarse_finish(hip.coarse_begin(N, [-1.0, -1.0, -1.0], 2.0 / (N - 1)))
        w = hip.decode_grid(N, [-1.0, -1.0, -1.0], 2.0 / (N - 1))[2].cpu().numpy()
        assert _boxes(b) == _boxes(w)
        bh, bo, ticket = hip.fine_begin(N, lattice[0], lattice[1], mc_only=True)
        assert not hip.fine_needs_repeat(ticket)
        eh, eo, _ = hip.decode_grid(N, lattice[0], lattice[1])
        for bv, ev in ((bh, eh), (bo, eo)):
            assert int(((bv < 0) != (ev < 0)).sum()) == 0
            vb, fb = marching_cubes_device(bv, 0.0)
            ve, fe = marching_cubes_device(ev, 0.0)
            assert torch.equal(vb, ve) and torch.equal(fb, fe)
    assert hip.box_stats["box"] == 2 and hip.band_stats["band"] == 2 and hip.box_stats["fallback"] == 0 and hip.band_stats["fallback"] == 0
    hip.close()


@pytest.mark.parametrize("tag,N,hand,obj", [("nerf3", 96, True, True), ("both9", 96, True, True), ("comb3", 72, True, True),
                                            ("nerf3", 68, True, False), ("nerf3", 70, False, True)])      # (N <= 64 has no two-step form: round 5)
def test_large_candidate_lists_take_the_two_step_form(tag, N, hand, obj):
    """Up to 2^15 candidates go straight to the fp32 chain; more than that (pose-aligned decoders list up to 1e6 at N = 256) are
    evaluated by the split-half kernel first and only the near-level ones among them by the fp32 chain, with the boxes extended
    from the list afterwards (csrc/decoder.hip: split_candidate_count_kernel).  A large allowance forces the second form on small
    lattices: both forms must deliver the ordinary sweep's boxes, and the record the same error measure."""
    hip, specs = _decoder(tag)
    hip.coarse_mode = "box"
    vs = 2.0 / (N - 1)
    forms = set()
    for sample in range(3):
        _bind(hip, specs, sample)
        want = hip.decode_grid(N, [-1.0, -1.0, -1.0], vs, hand=hand, obj=obj)[2].cpu().numpy()
        hip.coarse_finish(hip.coarse_begin(N, [-1.0, -1.0, -1.0], vs, hand=hand, obj=obj))      # (calibrates on the first sample)
        errs = {}
        for tau in (hip._box_tau, 0.02, 0.045):
            rec, _, _ = hip._box_launch(N, [-1.0, -1.0, -1.0], vs, 0, hand, obj, tau)
            r = rec.cpu().numpy()
            listed = int(r[32])
            forms.add(listed > (1 << 15))
            assert _boxes(r) == _boxes(want), (tag, sample, tau, listed)
            assert int(r[16]) == 0 and int(r[18]) == 0 and int(r[36]) == 0          # no range violation, no contradiction, no audit flip
            errs[tau] = float(np.int32(r[19]).view(np.float32))
        # the larger lists contain the smaller ones: the largest |exact - one-plane| over the re-evaluated voxels cannot shrink
        taus = sorted(errs)
        assert errs[taus[0]] <= errs[taus[1]] + 1e-9 <= errs[taus[2]] + 2e-9, errs
        assert 0.0 < errs[taus[2]] < 0.01
    assert forms == {False, True}, "both forms of re-evaluation should have run (%s)" % forms
    hip.close()


@pytest.mark.parametrize("tag", ["nerf3", "grasp3", "comb3"])
def test_one_plane_kernel_reports_an_fp16_overflow_through_its_outputs(tag):
    """The one-plane kernels keep no running maximum of their activations (round 4: their own weight image, three VALU instructions per
    register pair): an activation that leaves the fp16 range becomes an infinity, which must reach the output as exactly +-1 or a NaN
    and be REPORTED (words 7 / 15 of the record) - with activation scales 2^7 .. 2^9 too large for the decoder the sweep has to say so,
    and with the calibrated scales it must not."""
    import ctypes
    from alignsdf_amd import _native
    hip, specs = _decoder(tag)
    _bind(hip, specs, 1)
    N = 48
    org, vs = [-1.0, -1.0, -1.0], 2.0 / (N - 1)
    hip.decode_grid(N, org, vs)                                  # calibrates the activation scales (peaks land in [1024, 2048))
    good = hip.act_scales().copy()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def sweep():
        vh = torch.empty((N, N, N), dtype=torch.float32, device="cuda")
        vo = torch.empty((N, N, N), dtype=torch.float32, device="cuda")
        rec = torch.zeros(48, dtype=torch.int32, device="cuda")
        o3 = (ctypes.c_float * 3)(*org)
        _native.check(hip._L.asdf_decode_grid_box(hip._h, N, o3, ctypes.c_float(vs), 0, ctypes.c_float(2e-3), vh.data_ptr(), vo.data_ptr(),
                                                  rec.data_ptr(), st), "asdf_decode_grid_box")
        r = rec.cpu().numpy()
        return (int(r[7]) & 0x3fffffff) + (int(r[15]) & 0x3fffffff)

    assert sweep() == 0
    for layer in range(3):
        for boost in (128.0, 512.0):
            sx = good.copy()
            sx[:, layer] *= boost                              # peak x boost >= 1.3e5 > 65504: that layer's planes overflow
            hip.set_act_scales(sx)
            if not hip._one_plane_ok():
                continue                                        # (the image cannot be built for such a jump between layers: not offered at all)
            _bind(hip, specs, 1)
            assert sweep() > 0, (tag, layer, boost)
    hip.set_act_scales(good)
    _bind(hip, specs, 1)
    assert sweep() == 0
    hip.close()


@pytest.mark.parametrize("tag", ["nerf3", "comb3"])
def test_a_decoder_that_saturates_is_not_a_range_violation(tag):
    """fp32 tanhf returns EXACTLY +-1 for every finite argument beyond ~9, so a decoder whose output merely
    saturates in the far field (here: the last layer x 400 - the same zero set, a steeper field) produced `bad` on every one-plane
    sweep while the range report looked at the tanh OUTPUT: four futile re-calibrations later the decoder sat on the fp32 chain for
    good.  The report now looks at the pre-activation (non-finite = an fp16 overflow upstream): saturated outputs, no violation, the
    box sweep's boxes equal the ordinary sweep's, the arithmetic in force is still the split-half one."""
    from alignsdf_amd.hip_decoder import HipSdfDecoder
    specs = syn.specs_for(tag)
    sd = {k: np.array(v, copy=True) for k, v in syn.full_state_dict(tag).items()}
    for k in list(sd):
        if k.split(".")[0] in ("linh4", "lino4", "lin4"):
            sd[k] = sd[k] * np.float32(400.0)
    hip = HipSdfDecoder(sd, 256, specs["PointFeatSize"], specs["EncodeStyle"])
    hip.coarse_mode = "box"
    N, vs = 64, 2.0 / 63
    for sample in range(3):
        _bind(hip, specs, sample)
        vh, vo, want = hip.decode_grid(N, [-1.0, -1.0, -1.0], vs)
        assert float((vh.abs() == 1.0).float().mean()) > 0.5                    # most of the cube IS saturated
        rec, sh, so = hip._box_launch(N, [-1.0, -1.0, -1.0], vs, 0, True, True, 0.02)
        r = rec.cpu().numpy()
        assert (int(r[7]) & 0x3fffffff) == 0 and (int(r[15]) & 0x3fffffff) == 0, (tag, sample, int(r[7]), int(r[15]))
        assert float((sh.abs() == 1.0).float().mean()) > 0.5 and bool(torch.isfinite(sh).all())
        got = hip.coarse_finish(hip.coarse_begin(N, [-1.0, -1.0, -1.0], vs))
        assert _boxes(got) == _boxes(want.cpu().numpy()), (tag, sample)
    assert hip.math == "f16x3" and hip._recalibrations <= 1, (hip.math, hip._recalibrations)
    hip.close()
