"""The audited one-plane sweeps of NeRF-ENCODED decoders (PointFeatSize 9 / 15 with EncodeStyle "nerf": utils/mesh.py:53-55 - the
positional encoding of the query points is generated inside the kernel).  Until round 4 these decoders had no one-plane kernel and kept
ordinary sweeps on both passes (VERDICT r03 missing #4); csrc/k1s_nerf_kernels.hip holds the PL = 1 instantiations with 5 / 8 fp32-MFMA
K-steps of point features, csrc/k1h_nerf_kernels.hip the split-half kernel over a voxel list that produces the exact values behind them.

The bar is the one of tests/test_gpu_coarse_box.py: the SAME boxes as the ordinary sweep from the box-only coarse pass, the SAME mesh -
vertex for vertex, face for face - from the narrow-band fine pass, zero refused sweeps, and the product's entry points picking the
one-plane sweeps by default."""
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

# SeparateDecoder tags of alignsdf_amd.synthetic, and CombinedDecoders built on the same hidden layers (networks/model.py:79-188: one
# MLP, a 2-row last layer): column 0 = the fitted hand row of the SeparateDecoder, column 1 = the same shape shrunk by 0.04
CASES = ["nerf9", "nerf15", "comb9", "comb15"]


def _state_dict(case):
    if not case.startswith("comb"):
        return syn.full_state_dict(case), False
    pf = int(case[4:])
    sd = syn.combined_hidden_state_dict(256, pf, 0)
    last = syn.load_last_layers("nerf%d" % pf)
    w, b = last["linh4.weight"], last["linh4.bias"]
    sd["lin4.weight"] = np.concatenate([w, w], axis=0).astype(np.float32)
    sd["lin4.bias"] = np.concatenate([b, b + np.float32(0.04)]).astype(np.float32)
    return sd, True


def _decoder(case):
    from alignsdf_amd.hip_decoder import HipSdfDecoder
    sd, combined = _state_dict(case)
    hip = HipSdfDecoder(sd, 256, int(case[4:]), "nerf")
    assert hip.nerf_features and hip.combined == combined
    if hip.math != "f16x3":          # (nerf15, two MLPs: the fp32 chain by default - hip_decoder._init_sweep_state)
        hip.set_math("f16x3")
    return hip


def _bind(hip, sample):
    hip.set_sample(torch.from_numpy(syn.latent_code(sample)).cuda(), None)


def _boxes(b):
    return [int(v) for v in b[0:6]] + [int(v) for v in b[8:14]] + [int(b[6] != 0), int(b[14] != 0)]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("N", [64, 128])
def test_nerf_box_and_band_sweeps_give_the_ordinary_sweeps_boxes_and_meshes(case, N):
    from alignsdf_amd.marching_cubes import marching_cubes_device
    hip = _decoder(case)
    hip.coarse_mode, hip.fine_mode = "box", "band"
    assert hip._box_usable() and hip._band_usable()
    vs = 2.0 / (N - 1)
    lattice = ([-0.62, -0.36, -0.37], 1.21 / (N - 1))
    for sample in range(4):
        _bind(hip, sample)
        got = hip.coarse_finish(hip.coarse_begin(N, [-1.0, -1.0, -1.0], vs))            # sample 0: calibrates the allowance
        want = hip.decode_grid(N, [-1.0, -1.0, -1.0], vs)[2].cpu().numpy()
        assert _boxes(got) == _boxes(want), (sample, got, want)
        bh, bo, ticket = hip.fine_begin(N, lattice[0], lattice[1], mc_only=True)
        assert not hip.fine_needs_repeat(ticket), hip.band_stats
        sh, so, _ = hip.decode_grid(N, lattice[0], lattice[1])
        for b, e in ((bh, sh), (bo, so)):
            assert int(((b < 0) != (e < 0)).sum()) == 0
            if bool((e < 0).any()) and bool((e > 0).any()):
                vb, fb = marching_cubes_device(b, 0.0)
                ve, fe = marching_cubes_device(e, 0.0)
                assert torch.equal(fb, fe) and torch.equal(vb, ve)
    assert hip.box_stats["box"] == 3 and hip.box_stats["exact"] == 1 and hip.box_stats["fallback"] == 0, hip.box_stats
    assert hip.band_stats["band"] >= 3 and hip.band_stats["fallback"] == 0, hip.band_stats
    assert 0.0 < hip._box_tau < 0.05 and max(hip.box_stats["max_err"], hip.band_stats["max_err"]) <= 0.6 * hip._box_tau
    assert hip.range_violations() == 0
    hip.close()


def test_nerf9_through_the_sample_pipeline_uses_the_one_plane_sweeps_by_default(monkeypatch):
    """decoder_for -> pipelined_two_pass with the product's defaults: after the calibrating sample every sweep of a NeRF-encoded decoder
    is a one-plane sweep, and cubes and meshes are those of ordinary sweeps."""
    from alignsdf_amd.networks.model import build_decoder
    from alignsdf_amd.reconstruct import pipelined_two_pass, synthetic_code_source
    from alignsdf_amd.utils.utils import decoder_for
    tag, N = "nerf9", 96
    specs = syn.specs_for(tag)
    src = synthetic_code_source(tag, "cuda")
    samples = [(i,) + src("s%d" % i, i) for i in (0, 3, 7, 11, 20, 21)]
    out = {}
    for mode in ("exact", "default"):
        for k in ("ASDF_COARSE", "ASDF_FINE"):
            if mode == "exact":
                monkeypatch.setenv(k, "exact")
            else:
                monkeypatch.delenv(k, raising=False)
        dec = build_decoder(specs, {k: torch.from_numpy(v) for k, v in syn.full_state_dict(tag).items()})
        out[mode] = {k: r for k, r in pipelined_two_pass(dec, specs, iter(samples), N)}
        hip = decoder_for(dec, specs, samples[0][2])
        if mode == "default":
            assert hip.coarse_mode == "box" and hip.fine_mode == "band"
            assert hip.box_stats["box"] == len(samples) - 1 and hip.box_stats["fallback"] == 0, hip.box_stats
            assert hip.band_stats["band"] >= len(samples) - 1 and hip.band_stats["fallback"] == 0, hip.band_stats
    for k, a in out["exact"].items():
        b = out["default"][k]
        assert a["origin"] == b["origin"] and float(a["voxel_size"]) == float(b["voxel_size"])
        for part in ("hand", "obj"):
            assert torch.equal(a["verts_" + part], b["verts_" + part]) and torch.equal(a["faces_" + part], b["faces_" + part])
