"""The split-half arithmetic (two fp16 planes per operand, three fp16 MFMAs per product sum - the default of the grid
sweeps) on decoders chosen to break it: per-layer magnitude spreads of 1e4, activations far beyond and far below the fp16
range of the default scale, heavy-tailed weight_v, weight_g = 30, latent x 10, and the exact boundary of the range guard.

What keeps it safe: (1) the activation scales S_x are calibrated per layer from the peak plane values of the
first sweep (asdf_decoder_status / asdf_decoder_set_act_scales), so the planes use the fp16 range whatever the magnitudes;
(2) every sweep reports range violations - in the bbox record and in a decoder-owned status word - and the wrappers
re-calibrate, then fall back to the fp32 MFMA chain; (3) voxels within 4e-6 of the level are recomputed on the fp32 chain.
The yardsticks: the fp64 evaluation of the same weights (truth), the fp32 CPU oracle (the reference's arithmetic) and
the fp32 MFMA kernel."""
import numpy as np
import pytest
import torch

from alignsdf_amd import _native
from alignsdf_amd import synthetic as syn

pytestmark = pytest.mark.gpu
N = 24
ORIGIN, VS = [-0.9, -0.8, -0.85], 1.7 / (N - 1)


def plain_weights(tag="nerf3"):
    from oracle import sdf_oracle as orc
    base, sd = syn.full_state_dict(tag), {}
    for head in "ho":
        for layer, (w, b) in enumerate(orc.effective_head_params(base, head)):
            sd["lin%s%d.weight" % (head, layer)], sd["lin%s%d.bias" % (head, layer)] = w.numpy().copy(), b.numpy().copy()
    return sd


def forward64(sd, latent, pts):
    """fp64 evaluation of both heads (networks/model.py:285-350, PointFeatSize 3)."""
    x0 = np.concatenate([np.repeat(latent.reshape(1, -1).astype(np.float64), len(pts), 0), pts.astype(np.float64)], 1)
    out, peaks = [], []
    for head in "ho":
        W = [sd["lin%s%d.weight" % (head, k)].astype(np.float64) for k in range(5)]
        b = [sd["lin%s%d.bias" % (head, k)].astype(np.float64) for k in range(5)]
        h0 = np.maximum(x0 @ W[0].T + b[0], 0)
        h1 = np.maximum(h0 @ W[1].T + b[1], 0)
        h2 = np.maximum(np.concatenate([h1, x0], 1) @ W[2].T + b[2], 0)
        h3 = np.maximum(h2 @ W[3].T + b[3], 0)
        out.append(np.tanh(h3 @ W[4].T + b[4])[:, 0])
        peaks.append((h0.max(), h1.max(), h2.max(), h3.max()))
    return out[0], out[1], peaks


def lattice():
    idx = np.stack(np.meshgrid(np.arange(N), np.arange(N), np.arange(N), indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    return idx * np.float32(VS) + np.array(ORIGIN, np.float32)


def rescale_last_layer(sd, latent, target=0.08):
    """Keep the pre-tanh outputs of an altered network of order 0.1 (where the 1e-5 bar means something)."""
    pts = syn.uniform((2000, 3), 55, -1.0, 1.0).astype(np.float32)
    x0 = np.concatenate([np.repeat(latent.reshape(1, -1).astype(np.float64), len(pts), 0), pts.astype(np.float64)], 1)
    for head in "ho":
        W = [sd["lin%s%d.weight" % (head, k)].astype(np.float64) for k in range(5)]
        b = [sd["lin%s%d.bias" % (head, k)].astype(np.float64) for k in range(5)]
        h = np.maximum(x0 @ W[0].T + b[0], 0)
        h = np.maximum(h @ W[1].T + b[1], 0)
        h = np.maximum(np.concatenate([h, x0], 1) @ W[2].T + b[2], 0)
        h = np.maximum(h @ W[3].T + b[3], 0)
        pre = h @ W[4].T
        k = target / max(float(np.std(pre)), 1e-30)
        sd["lin%s4.weight" % head] = (W[4] * k).astype(np.float32)
        sd["lin%s4.bias" % head] = np.float32([-float(np.mean(pre)) * k])
    return sd


def variant(name):
    sd, lat = plain_weights(), syn.latent_code(3).reshape(-1)
    rng = np.random.RandomState(7)
    if name == "spread1e4":                    # function-preserving (ReLU is positively homogeneous): h0 x 100, h1 x 0.01, h2 x 100
        for head in "ho":
            n1 = sd["lin%s1.weight" % head].shape[0]
            sd["lin%s0.weight" % head] *= np.float32(100.0); sd["lin%s0.bias" % head] *= np.float32(100.0)
            sd["lin%s1.weight" % head] *= np.float32(1e-4); sd["lin%s1.bias" % head] *= np.float32(1e-2)
            sd["lin%s2.weight" % head][:, :n1] *= np.float32(1e4); sd["lin%s2.weight" % head][:, n1:] *= np.float32(100.0)
            sd["lin%s2.bias" % head] *= np.float32(100.0)
            sd["lin%s3.weight" % head] *= np.float32(1e-2)
    elif name == "huge":                       # activations of order 1e4: beyond the default scale's fp16 range
        for head in "ho":
            sd["lin%s0.weight" % head] *= np.float32(4096.0); sd["lin%s0.bias" % head] *= np.float32(4096.0)
            sd["lin%s1.weight" % head] /= np.float32(4096.0)
    elif name == "tiny":                       # activations of order 1e-4: low planes subnormal at the default scale
        for head in "ho":
            sd["lin%s0.weight" % head] /= np.float32(16384.0); sd["lin%s0.bias" % head] /= np.float32(16384.0)
            sd["lin%s1.weight" % head] *= np.float32(16384.0)
    elif name == "heavy_tails":                # log-normal multipliers on every hidden weight: a few weights dominate their rows
        for head in "ho":
            for k in range(4):
                sd["lin%s%d.weight" % (head, k)] *= np.exp(1.2 * rng.randn(*sd["lin%s%d.weight" % (head, k)].shape)).astype(np.float32)
        sd = rescale_last_layer(sd, lat)
    elif name == "gain30":                     # weight_g = 30 instead of 1.5 / 3 on every normed layer
        for head in "ho":
            for k, g in enumerate((3.0, 1.5, 1.5, 1.5)):
                sd["lin%s%d.weight" % (head, k)] *= np.float32(30.0 / g)
        sd = rescale_last_layer(sd, lat)
    elif name == "latent_x10":
        lat = lat * np.float32(10.0)
        sd = rescale_last_layer(sd, lat)
    else:
        raise ValueError(name)
    return sd, lat.astype(np.float32)


@pytest.mark.parametrize("name", ["spread1e4", "huge", "tiny", "heavy_tails", "gain30", "latent_x10"])
def test_adversarial_decoders_stay_fp32_class(name):
    from alignsdf_amd.hip_decoder import HipSdfDecoder
    from oracle import sdf_oracle as orc
    sd, lat = variant(name)
    pts = lattice()
    t_h, t_o, peaks = forward64(sd, lat, pts)
    hip = HipSdfDecoder(sd, 256, 3, "nerf")
    hip.set_refine(0.0)                        # the arithmetic itself, not the near-level repair
    hip.set_sample(torch.from_numpy(lat).cuda())
    vh, vo, bbox = hip.decode_grid(N, ORIGIN, VS, _native.GRID_INTEGER)
    assert hip.math == "f16x3", "calibration should have kept %s on the split-half kernel (peaks %s)" % (name, peaks)
    b = bbox.cpu().numpy()
    assert b[7] == 0 and b[15] == 0
    e16 = max(np.abs(vh.cpu().numpy().reshape(-1) - t_h).max(), np.abs(vo.cpu().numpy().reshape(-1) - t_o).max())
    hip.set_math("f32")
    fh, fo, _ = hip.decode_grid(N, ORIGIN, VS, _native.GRID_INTEGER)
    e32 = max(np.abs(fh.cpu().numpy().reshape(-1) - t_h).max(), np.abs(fo.cpu().numpy().reshape(-1) - t_o).max())
    oh, oo = orc.decode_points(sd, lat, pts, syn.specs_for("nerf3"))
    eor = max(np.abs(oh.numpy() - t_h).max(), np.abs(oo.numpy() - t_o).max())
    print("%s: |split-half - fp64| %.2e, |fp32 MFMA - fp64| %.2e, |fp32 CPU oracle - fp64| %.2e, scales %s, peaks %s" % (
        name, e16, e32, eor, hip.act_scales().tolist(), [tuple(float("%.3g" % v) for v in p) for p in peaks]))
    # fp32-class: within a small factor of what the two fp32 evaluations themselves lose against fp64 ...
    assert e16 <= 3.0 * max(e32, eor) + 5e-7, (name, e16, e32, eor)
    # ... and inside the 1e-5 bar against the reference's arithmetic wherever that arithmetic is itself that well defined
    if max(e32, eor) <= 2e-6:
        d = max(np.abs(vh.cpu().numpy().reshape(-1) - oh.numpy()).max(), np.abs(vo.cpu().numpy().reshape(-1) - oo.numpy()).max())
        assert d <= 1e-5, (name, d)
    hip.close()


def _boundary_decoder(x):
    """nerf3 with row 7 of layer 0 turned into the constant activation x (weights 0, bias x) that nothing consumes."""
    from alignsdf_amd.hip_decoder import HipSdfDecoder
    sd = plain_weights()
    for head in "ho":
        sd["lin%s0.weight" % head][7, :] = 0.0
        sd["lin%s0.bias" % head][7] = np.float32(x)
        sd["lin%s1.weight" % head][:, 7] = 0.0
    hip = HipSdfDecoder(sd, 256, 3, "nerf")
    hip._calibrated = True                     # keep the default scale S_x = 8: the guard's own boundary is under test
    hip.set_sample(torch.from_numpy(syn.latent_code(0)).cuda())
    return hip


def test_range_guard_boundary_8187_passes_8189_is_flagged():
    ok = _boundary_decoder(8187.0)             # 8187 * 8 = 65496 < 65504
    _, _, b = ok.decode_grid(16, [-1.0, -1.0, -1.0], 2.0 / 15)
    assert int(b[7]) == 0 and int(b[15]) == 0 and ok.range_violations() == 0
    st = ok._status(clear=False)
    ok.close()
    bad = _boundary_decoder(8189.0)            # 8189 * 8 = 65512 >= 65504
    vh, vo, b = bad.decode_grid(16, [-1.0, -1.0, -1.0], 2.0 / 15)
    b = b.cpu().numpy()
    assert b[7] > 0 and b[15] > 0
    assert bad.fall_back_if_overflowed(b) is True          # first resort: re-calibration from the recorded peaks
    assert bad.math == "f16x3" and bad.act_scales()[0, 0] < 8.0
    _, _, b2 = bad.decode_grid(16, [-1.0, -1.0, -1.0], 2.0 / 15)
    assert int(b2[7]) == 0 and int(b2[15]) == 0
    bad.close()


def test_unrepresentable_range_ends_on_the_fp32_chain(golden_dir):
    """Activations of order 1e16 cannot be scaled into fp16 at all (S_x is bounded at 2^-24): after the re-calibrations are
    exhausted the decoder runs the fp32 MFMA chain - and still returns the reference's volumes (same function)."""
    from alignsdf_amd.hip_decoder import HipSdfDecoder
    from alignsdf_amd.utils.mesh import decode_two_pass
    sd = plain_weights()
    for head in "ho":
        sd["lin%s0.weight" % head] *= np.float32(2.0 ** 50); sd["lin%s0.bias" % head] *= np.float32(2.0 ** 50)
        sd["lin%s1.weight" % head] *= np.float32(2.0 ** -50)
    hip = HipSdfDecoder(sd, 256, 3, "nerf")
    lat = torch.from_numpy(syn.latent_code(0)).cuda()
    r = decode_two_pass(True, True, hip, lat, None, None, syn.specs_for("nerf3"), 32)
    assert hip.math == "f32"
    g = np.load(golden_dir + "/ref_decoder_nerf3.npz")
    assert np.array_equal(np.stack([r["bbox"][0:6], r["bbox"][8:14]]), g["bbox_32"])
    assert np.abs(r["vol_hand"].cpu().numpy() - g["vol2_hand_32"]).max() <= 1e-5
    hip.close()


def test_calibration_is_a_no_op_on_the_shipped_decoders_and_reported():
    from alignsdf_amd.hip_decoder import HipSdfDecoder
    for tag in ("nerf3", "both9"):
        specs = syn.specs_for(tag)
        hip = HipSdfDecoder(syn.full_state_dict(tag), 256, specs["PointFeatSize"], specs["EncodeStyle"])
        assert np.array_equal(hip.act_scales(), np.full((2, 3), 8.0, np.float32))
        hip.close()
    with pytest.raises(_native.NativeError):
        hip = HipSdfDecoder(syn.full_state_dict("nerf3"), 256, 3, "nerf")
        hip.set_act_scales(np.full((2, 3), 3.0))          # not a power of two
