"""Both arithmetic modes of the decoder sweeps against the reference goldens: the fp32 MFMA chain ("f32") and the
split-half fp16 MFMA kernel ("f16x3", the default).  Everything else in the GPU suite runs on the default."""
import numpy as np
import pytest
import torch

from alignsdf_amd import _native
from alignsdf_amd import synthetic as syn

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _decoder(tag):
    from alignsdf_amd.networks.model import build_decoder
    from alignsdf_amd.utils.utils import hip_decoder_for, sample_embedding
    specs = syn.specs_for(tag)
    dec = build_decoder(specs, {k: torch.from_numpy(v) for k, v in syn.full_state_dict(tag).items()})
    hip = hip_decoder_for(dec)
    lat = torch.from_numpy(syn.latent_code(0)).cuda()
    mano = obj = None
    if specs["EncodeStyle"] != "nerf":
        m, o = syn.pose_inputs(0)
        mano = {k: torch.from_numpy(v).cuda() for k, v in m.items()}
        obj = {k: torch.from_numpy(v).cuda() for k, v in o.items()}
    hip.set_sample(lat, sample_embedding(specs, mano, obj, hip.combined))
    return hip


def test_default_is_split_half():
    assert _decoder("nerf3").math == "f16x3"


@pytest.mark.parametrize("tag", ["nerf3", "both9", "comb3", "nerf9", "nerf15", "hand6", "hand51", "obj6"])
def test_both_modes_match_reference_points_and_grids(tag, golden_dir):
    g = np.load("%s/ref_decoder_%s.npz" % (golden_dir, tag))
    hip = _decoder(tag)
    pts = torch.from_numpy(g["rand_pts"]).cuda()
    out = {}
    for math in ("f32", "f16x3"):
        hip.set_math(math)
        assert hip.math == math
        h, o = hip.decode_points(pts)             # point lists run the fp32 chain in either mode
        assert np.abs(h.cpu().numpy() - g["rand_hand"]).max() <= TOL and np.abs(o.cpu().numpy() - g["rand_obj"]).max() <= TOL
        for N in (32, 64):
            vh, vo, bbox = hip.decode_grid(N, [-1.0, -1.0, -1.0], 2.0 / (N - 1))
            sel = g["probe_sel_%d" % N]
            assert np.abs(vh.cpu().numpy().reshape(-1)[sel] - g["p1_hand_%d" % N]).max() <= TOL
            assert np.abs(vo.cpu().numpy().reshape(-1)[sel] - g["p1_obj_%d" % N]).max() <= TOL
            b = bbox.cpu().numpy()
            assert np.array_equal(np.stack([b[0:6], b[8:14]]), g["bbox_%d" % N])
            out[(math, N)] = (vh, vo)
        out[math] = (h, o)
    # the two arithmetics agree far inside the bar (both are fp32-class) - and they are two different kernels
    assert torch.equal(out["f32"][0], out["f16x3"][0])
    for N in (32, 64):
        for k in (0, 1):
            d = (out[("f32", N)][k] - out[("f16x3", N)][k]).abs().max().item()
            assert 0.0 < d <= 2e-6, (N, k, d)


def test_split_half_layer_scales_vs_oracle():
    """The same network re-parametrised so that its layers have very different weight magnitudes (layer 1 scaled by g,
    its consumers' columns by 1/g: ReLU is positively homogeneous, the function is unchanged) - the per-layer power-of-two
    scales of the split-half image then differ by orders of magnitude.  Ragged point counts, against the CPU oracle."""
    from alignsdf_amd.hip_decoder import HipSdfDecoder
    from oracle import sdf_oracle as orc
    specs = syn.specs_for("nerf3")
    base = syn.full_state_dict("nerf3")
    lat = syn.latent_code(2)
    for gain in (1.0, 1.0 / 64.0, 48.0):
        sd = {}
        for head in "ho":
            for layer, (w, b) in enumerate(orc.effective_head_params(base, head)):
                sd["lin%s%d.weight" % (head, layer)], sd["lin%s%d.bias" % (head, layer)] = w.numpy().copy(), b.numpy().copy()
            n1 = sd["lin%s1.weight" % head].shape[0]
            sd["lin%s1.weight" % head] *= np.float32(gain)
            sd["lin%s1.bias" % head] *= np.float32(gain)
            sd["lin%s2.weight" % head][:, :n1] /= np.float32(gain)
        hip = HipSdfDecoder(sd, 256, 3, "nerf")
        assert hip.math == "f16x3"
        hip.set_sample(torch.from_numpy(lat).cuda())
        for N, origin, vs in ((8, [-1.0, -1.0, -1.0], 2.0 / 7), (21, [-0.6, -0.4, -0.35], 0.031)):
            vh, vo, bbox = hip.decode_grid(N, origin, vs, _native.GRID_INTEGER)
            assert bbox.cpu().numpy()[7] == 0 and bbox.cpu().numpy()[15] == 0
            idx = np.stack(np.meshgrid(np.arange(N), np.arange(N), np.arange(N), indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
            pts = idx * np.float32(vs) + np.array(origin, np.float32)
            wh, wo = orc.decode_points(sd, lat, pts, specs)
            assert np.abs(vh.cpu().numpy().reshape(-1) - wh.numpy()).max() <= TOL, (gain, N)
            assert np.abs(vo.cpu().numpy().reshape(-1) - wo.numpy()).max() <= TOL, (gain, N)
        hip.close()


def test_fp16_overflow_is_detected_and_recovered(tmp_path):
    """Activations beyond the fp16 range of the planes at the current scale (|x| S_x >= 65504) cannot be carried: the kernel
    counts them in bbox words 7 / 15 and in the status word; the two-pass drivers re-calibrate the activation scales from
    the recorded peaks and repeat the pass (the fp32 kernel is the last resort, test_gpu_split_half_adversarial.py)."""
    from alignsdf_amd.utils.mesh import decode_two_pass
    specs = syn.specs_for("nerf3")
    hip = _overflowing_decoder()
    hip._calibrated = True                     # skip the first-sweep calibration: the guard itself is under test
    lat = torch.from_numpy(syn.latent_code(0)).cuda()
    hip.set_sample(lat)
    _, _, bbox = hip.decode_grid(32, [-1.0, -1.0, -1.0], 2.0 / 31)
    b = bbox.cpu().numpy()
    assert b[7] > 0 and b[15] > 0 and hip.math == "f16x3"
    r = decode_two_pass(True, True, hip, lat, None, None, specs, 32)
    assert hip.math == "f16x3" and hip.act_scales()[0, 0] < 8.0 and r["bbox"][7] == 0 and r["bbox"][15] == 0
    g = np.load(str(__import__("pathlib").Path(__file__).parent / "golden" / "ref_decoder_nerf3.npz"))
    assert np.array_equal(np.stack([r["bbox"][0:6], r["bbox"][8:14]]), g["bbox_32"])
    assert np.abs(r["vol_hand"].cpu().numpy() - g["vol2_hand_32"]).max() <= TOL
    hip.close()


def test_fp16_overflow_in_the_sample_pipeline(golden_dir):
    """The same decoder through pipelined_two_pass, once with the first-sweep calibration and once with the range report
    tripping in pass 1 of the first sample: every sample comes out as the reference's volumes, on the split-half kernel."""
    from alignsdf_amd.reconstruct import pipelined_two_pass
    specs = syn.specs_for("nerf3")
    g = np.load(golden_dir + "/ref_decoder_nerf3.npz")
    for skip_calibration in (False, True):
        hip = _overflowing_decoder()
        hip._calibrated = skip_calibration
        hip.fine_mode = "exact"          # the pass-2 VOLUMES are compared with the reference's below: ordinary fine sweeps
        lat = torch.from_numpy(syn.latent_code(0)).cuda()
        out = list(pipelined_two_pass(hip, specs, [(k, lat, None, None) for k in range(3)], 32))
        assert hip.math == "f16x3" and len(out) == 3 and hip.act_scales()[0, 0] < 8.0
        for _, r in out:
            assert np.abs(r["vol_hand"].cpu().numpy() - g["vol2_hand_32"]).max() <= TOL
            assert np.abs(r["vol_obj"].cpu().numpy() - g["vol2_obj_32"]).max() <= TOL
            assert r["V_hand"] > 0 and r["V_obj"] > 0
        hip.close()


def test_set_math_argument_checks(native_lib):
    hip = _decoder("nerf9")
    assert hip.math == "f16x3"                    # NeRF-encoded decoders run the split-half kernel too (16 KiB stages)
    assert native_lib.asdf_decoder_set_math(hip._h, 7) == -1
    assert native_lib.asdf_decoder_set_math(None, 0) == -1
    assert native_lib.asdf_decoder_set_math(hip._h, 0) == 0 and native_lib.asdf_decoder_get_math(hip._h) == 0
    assert native_lib.asdf_decoder_set_math(hip._h, 1) == 0 and native_lib.asdf_decoder_get_math(hip._h) == 1


@pytest.mark.parametrize("tag", ["nerf3", "both9"])
def test_modes_agree_over_the_sample_set(tag):
    """All 64 synthetic samples (latent codes, poses) at N=64: the two arithmetics give the same negative-voxel boxes and
    counts up to voxels within 1e-6 of the level, and volumes that agree to 4e-6 (measured: 2.1e-6 at worst)."""
    from alignsdf_amd.networks.model import build_decoder
    from alignsdf_amd.utils.utils import hip_decoder_for, sample_embedding
    specs = syn.specs_for(tag)
    dec = build_decoder(specs, {k: torch.from_numpy(v) for k, v in syn.full_state_dict(tag).items()})
    hip = hip_decoder_for(dec)
    N, worst, box_mismatch = 64, 0.0, 0
    for s in range(64):
        lat = torch.from_numpy(syn.latent_code(s)).cuda()
        mano = obj = None
        if tag == "both9":
            m, o = syn.pose_inputs(s)
            mano = {k: torch.from_numpy(v).cuda() for k, v in m.items()}
            obj = {k: torch.from_numpy(v).cuda() for k, v in o.items()}
        hip.set_sample(lat, sample_embedding(specs, mano, obj, hip.combined))
        out = {}
        for math in ("f32", "f16x3"):
            hip.set_math(math)
            out[math] = hip.decode_grid(N, [-1.0, -1.0, -1.0], 2.0 / (N - 1))
        for k in (0, 1):
            worst = max(worst, (out["f32"][k] - out["f16x3"][k]).abs().max().item())
        a, b = out["f32"][2].cpu().numpy(), out["f16x3"][2].cpu().numpy()
        assert b[7] == 0 and b[15] == 0
        near = int((out["f32"][0].abs() < 1e-6).sum()) + int((out["f32"][1].abs() < 1e-6).sum())
        assert abs(int(a[6]) - int(b[6])) + abs(int(a[14]) - int(b[14])) <= near
        box_mismatch += int(not np.array_equal(a[[0, 1, 2, 3, 4, 5, 8, 9, 10, 11, 12, 13]], b[[0, 1, 2, 3, 4, 5, 8, 9, 10, 11, 12, 13]]))
    assert worst <= 4e-6 and box_mismatch == 0, (worst, box_mismatch)      # two independent fp32-class roundings


def _overflowing_decoder(factor=4096.0):
    """The nerf3 network with layer 0 blown up by `factor` and layer 1 shrunk by it: the same function (ReLU is positively
    homogeneous), hidden activations of order 1e4 - beyond what the split-half planes carry at the default scale."""
    from alignsdf_amd.hip_decoder import HipSdfDecoder
    from oracle import sdf_oracle as orc
    base = syn.full_state_dict("nerf3")
    sd = {}
    for head in "ho":
        for layer, (w, b) in enumerate(orc.effective_head_params(base, head)):
            sd["lin%s%d.weight" % (head, layer)], sd["lin%s%d.bias" % (head, layer)] = w.numpy().copy(), b.numpy().copy()
        sd["lin%s0.weight" % head] *= np.float32(factor)
        sd["lin%s0.bias" % head] *= np.float32(factor)
        sd["lin%s1.weight" % head] /= np.float32(factor)
    return HipSdfDecoder(sd, 256, 3, "nerf")


def test_range_report_does_not_need_a_bbox(native_lib, golden_dir):
    """A C-ABI caller that passes bbox_dev = NULL still learns about an fp16 range violation: the decoder-owned status word
    (asdf_decoder_status) counts it.  The host wrapper's bbox-less sweep reads it, re-calibrates and repeats."""
    import ctypes
    hip = _overflowing_decoder()
    hip.set_sample(torch.from_numpy(syn.latent_code(0)).cuda())
    assert hip.math == "f16x3" and hip.range_violations(clear=True) == 0
    N = 32
    vh = torch.empty((N, N, N), device="cuda")
    vo = torch.empty((N, N, N), device="cuda")
    org = (ctypes.c_float * 3)(-1.0, -1.0, -1.0)
    rc = native_lib.asdf_decode_grid(hip._h, N, org, ctypes.c_float(2.0 / (N - 1)), 0, vh.data_ptr(), vo.data_ptr(), None,
                                     ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    bad = hip.range_violations(clear=False)
    assert bad > 0 and hip.range_violations(clear=True) == bad and hip.range_violations() == 0     # read, read + clear, cleared
    assert hip.math == "f16x3"                       # the raw call decides nothing; the caller does
    # the guarded wrapper: same sweep, no bbox -> falls back and returns the reference's volumes
    g = np.load(golden_dir + "/ref_decoder_nerf3.npz")
    hip._calibrated = True                       # no first-sweep calibration: the guard has to catch it
    h2, o2, none = hip.decode_grid(N, [-1.0, -1.0, -1.0], 2.0 / (N - 1), want_bbox=False)
    assert none is None and hip.math == "f16x3" and hip.act_scales()[0, 0] < 8.0      # recovered by re-calibration
    assert np.abs(h2.cpu().numpy() - g["vol1_hand_32"]).max() <= TOL and np.abs(o2.cpu().numpy() - g["vol1_obj_32"]).max() <= TOL
    hip.close()


def test_range_report_is_zero_on_the_shipped_decoders():
    for tag in ("nerf3", "both9"):
        hip = _decoder(tag)
        hip.range_violations(clear=True)
        hip.decode_grid(48, [-1.0, -1.0, -1.0], 2.0 / 47, want_bbox=False)
        assert hip.math == "f16x3" and hip.range_violations() == 0


def test_legacy_create_mesh_is_guarded(tmp_path, golden_dir, monkeypatch):
    """deep_sdf.mesh.create_mesh (deep_sdf/mesh.py:14-61) sweeps without a bbox buffer: with a decoder whose activations
    overflow the fp16 planes it must still return the reference's volume (it used to return garbage silently)."""
    from alignsdf_amd.deep_sdf import mesh as legacy
    g = np.load(golden_dir + "/ref_legacy.npz")
    hip = _overflowing_decoder()
    seen = {}
    real = legacy.convert_sdf_samples_to_ply

    def spy(vol, origin, vs, path):
        seen["vol"] = vol.cpu().numpy().copy()
        return real(vol, origin, vs, path)

    monkeypatch.setattr(legacy, "convert_sdf_samples_to_ply", spy)
    hip._calibrated = True                       # the guard of the bbox-less sweep, not the first-sweep calibration
    pts, faces = legacy.create_mesh(hip, torch.from_numpy(syn.latent_code(0)).cuda(), str(tmp_path / "legacy"), N=32)
    assert hip.math == "f16x3" and hip.act_scales()[0, 0] < 8.0
    assert np.abs(seen["vol"] - g["vol_32"]).max() <= TOL
    assert len(pts) > 0 and len(faces) > 0 and (tmp_path / "legacy.ply").exists()
    hip.close()


def test_fallback_decision_rests_on_the_record_alone():
    """once the decoder has switched to fp32, a bbox record of an EARLIER split-half sweep must still trigger the
    repeat of that sweep (the pipeline queues pass 1 of sample k+1 before it reads pass 2 of sample k)."""
    hip = _overflowing_decoder()
    hip._calibrated = True
    hip.set_sample(torch.from_numpy(syn.latent_code(0)).cuda())
    _, _, bbox = hip.decode_grid(32, [-1.0, -1.0, -1.0], 2.0 / 31)            # queued under f16x3
    hip.set_math("f32")                                                        # ... another sweep's report arrives first
    assert hip.fall_back_if_overflowed(bbox.cpu().numpy()) is True             # the stale record still says: repeat me
    _, _, clean = hip.decode_grid(32, [-1.0, -1.0, -1.0], 2.0 / 31)
    assert hip.fall_back_if_overflowed(clean.cpu().numpy()) is False
    hip.close()


# ---- round 6: the matrix instruction of the split-half SeparateDecoder kernels (asdf_set_mfma_shape) ------------------------------
@pytest.fixture
def mfma_shape():
    """Sets the process-wide shape for a test and puts the previous one back."""
    L = _native.lib()
    before = L.asdf_get_mfma_shape()
    yield lambda shape: L.asdf_set_mfma_shape(int(shape))
    L.asdf_set_mfma_shape(before)


@pytest.mark.parametrize("tag", ["nerf3", "both9", "hand6", "obj6"])
def test_both_mfma_shapes_match_the_reference_and_each_other(tag, golden_dir, mfma_shape):
    """The W form (v_mfma_f32_16x16x32_f16, the default since round 6) and the 32x32x16 form of the split-half kernel evaluate the
    same GEMMs: both within 1e-5 of the REFERENCE's own run on the probes, identical boxes, and within 2e-6 of each other on every
    voxel - yet different bits (other partial-sum order), i.e. the switch really selects another kernel.  (That a voxel's bits do
    not depend on the lane it sits in - the W form's two point groups take a record's MFMAs in snake order, the voxel lists of the
    subset form place a voxel anywhere - is pinned by tests/test_gpu_default_sweeps.py: the --fast sweeps' meshes, whose values come
    from the subset form, are the ordinary sweeps' vertex for vertex; a group-dependent product order failed exactly there.)"""
    g = np.load("%s/ref_decoder_%s.npz" % (golden_dir, tag))
    hip = _decoder(tag)
    assert _native.lib().asdf_get_mfma_shape() == 16 and hip.split_half_kernel == "sdf_mlp_f16w_kernel"
    vols = {}
    for shape in (16, 32):
        assert mfma_shape(shape) in (16, 32) and _native.lib().asdf_get_mfma_shape() == shape
        assert hip.split_half_kernel == ("sdf_mlp_f16w_kernel" if shape == 16 else "sdf_mlp_f16_kernel")
        for N in (32, 64):
            vh, vo, bbox = hip.decode_grid(N, [-1.0, -1.0, -1.0], 2.0 / (N - 1))
            sel = g["probe_sel_%d" % N]
            assert np.abs(vh.cpu().numpy().reshape(-1)[sel] - g["p1_hand_%d" % N]).max() <= TOL
            assert np.abs(vo.cpu().numpy().reshape(-1)[sel] - g["p1_obj_%d" % N]).max() <= TOL
            b = bbox.cpu().numpy()
            assert np.array_equal(np.stack([b[0:6], b[8:14]]), g["bbox_%d" % N]) and b[7] == 0 and b[15] == 0
            vols[(shape, N)] = (vh, vo)
    for N in (32, 64):
        for k in (0, 1):
            d = (vols[(16, N)][k] - vols[(32, N)][k]).abs().max().item()
            assert 0.0 < d <= 2e-6, (N, k, d)
    assert _native.lib().asdf_set_mfma_shape(7) == _native.lib().asdf_set_mfma_shape(7) < 0        # EINVAL, nothing changed
    hip.close()


def test_combined_and_nerf_decoders_keep_the_32_wide_kernels(mfma_shape):
    for tag in ("comb3", "nerf9"):
        hip = _decoder(tag)
        assert hip.split_half_kernel == "sdf_mlp_f16_kernel"
        mfma_shape(32)
        a = hip.decode_grid(32, [-1.0, -1.0, -1.0], 2.0 / 31)
        mfma_shape(16)
        b = hip.decode_grid(32, [-1.0, -1.0, -1.0], 2.0 / 31)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])      # the switch does not reach them
        hip.close()
