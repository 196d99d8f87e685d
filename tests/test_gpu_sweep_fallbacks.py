"""Every fallback of the sweep state machine ON THE GPU (the CPU table is tests/test_sweep_state_machine.py):
a near-level refinement list that overflows - in an ordinary sweep, a box-only coarse sweep and a narrow-band fine sweep - is answered
by ONE repeat on the fp32 chain and the volumes / boxes / meshes are the fp32 run's; three refused band sweeps in a row switch the
mode off; lists beyond BAND_CAP / CAND_CAP at N = 256 through the sample pipeline cost time, never correctness.  Whatever path ran,
what comes out is what utils/mesh.py:46-63, :98-115 produce: boxes from pass 1, two volumes' meshes from pass 2."""
import numpy as np
import pytest
import torch

from alignsdf_amd import hip_decoder as hd
from alignsdf_amd import synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _fast_sweeps(monkeypatch):
    """This file is about the OPT-IN audited one-plane sweeps (round 6: the product's default is ordinary sweeps on every voxel;
    ASDF_FAST=1 / --fast / HipSdfDecoder.set_fast select these)."""
    monkeypatch.setenv("ASDF_FAST", "1")


def _module(tag="nerf3"):
    from alignsdf_amd.networks.model import build_decoder
    specs = syn.specs_for(tag)
    return build_decoder(specs, {k: torch.from_numpy(v) for k, v in syn.full_state_dict(tag).items()}), specs


def _hip(dec, specs):
    from alignsdf_amd.utils.utils import decoder_for
    return decoder_for(dec, specs, None)


def _two_pass(dec, specs, sample, N, mc_only):
    from alignsdf_amd.utils.mesh import decode_two_pass
    lat = torch.from_numpy(syn.latent_code(sample)).cuda()
    return decode_two_pass(True, True, dec, lat, None, None, specs, N, mc_only=mc_only)


def test_near_level_overflow_in_an_ordinary_sweep_is_repeated_once_on_the_fp32_chain(monkeypatch):
    """refine_tau = 0.03 puts far more than 2^16 voxels of the zoom lattice into the near-level list: bit 30 of the range word ->
    one repeat of that sweep on the fp32 chain -> the volumes ARE the fp32 run's, and the decoder is back on f16x3 afterwards."""
    N = 128
    dec, specs = _module()
    hip = _hip(dec, specs)
    hip.set_math("f32")
    want = _two_pass(dec, specs, 2, N, mc_only=False)
    hip.set_math("f16x3")
    hip.set_refine(0.03)
    launches = []
    real = hip._L.asdf_decode_grid

    class Spy:
        def __getattr__(self, name):
            return getattr(hip_lib, name)

        def asdf_decode_grid(self, *a):
            launches.append(hip.math)
            return real(*a)

    hip_lib, hip._L = hip._L, Spy()
    got = _two_pass(dec, specs, 2, N, mc_only=False)
    hip._L = hip_lib
    assert got["origin"] == want["origin"] and float(got["voxel_size"]) == float(want["voxel_size"])
    assert torch.equal(got["vol_hand"], want["vol_hand"]) and torch.equal(got["vol_obj"], want["vol_obj"])
    # the sweep that overflowed was repeated ONCE on the fp32 chain (pass 1 - a box-only sweep here, whose ordinary repeat then
    # overflows too - and pass 2 alike), and nothing else ran on it
    assert launches[-2:] == ["f16x3", "f32"] and launches.count("f32") <= 2, launches
    assert hip.math == "f16x3" and not hip._force_f32_once
    hip.set_refine(4e-6)
    hip.close()


def test_near_level_overflow_in_a_band_sweep(monkeypatch):
    """The same under the default sweeps: the band sweep's near-level list (listed voxels within refine_tau of the level) overflows
    -> refused -> the repeat is an ordinary sweep on the fp32 chain; meshes = the fp32 run's, vertex for vertex."""
    from alignsdf_amd.marching_cubes import marching_cubes_device
    N = 256
    dec, specs = _module()
    hip = _hip(dec, specs)
    hip.coarse_mode = hip.fine_mode = "exact"
    hip.set_math("f32")
    want = _two_pass(dec, specs, 3, N, mc_only=True)
    hip.set_math("f16x3")
    hip.coarse_mode, hip.fine_mode = "box", "band"
    _two_pass(dec, specs, 1, N, mc_only=True)                       # compares the coarse and the zoom lattice as a whole (ordinary sweeps)
    assert hip.band_stats["band"] == 0 and hip.cert["fine_calibrations"] == 1
    _two_pass(dec, specs, 2, N, mc_only=True)                       # band sweeps from here on
    assert hip.band_stats["band"] == 1
    hip.set_refine(0.03)
    hip.coarse_mode = "exact"                                       # (pass 1 is not under test here; it would overflow as well)
    got = _two_pass(dec, specs, 3, N, mc_only=True)
    assert hip.band_stats["fallback"] == 1 and hip._band_failures == 0 and hip.math == "f16x3" and not hip._force_f32_once
    assert got["origin"] == want["origin"] and float(got["voxel_size"]) == float(want["voxel_size"])
    for part in ("hand", "obj"):
        va, fa = marching_cubes_device(want["vol_" + part], 0.0)
        vb, fb = marching_cubes_device(got["vol_" + part], 0.0)
        assert torch.equal(fa, fb) and torch.equal(va, vb)
        assert torch.equal(got["vol_" + part], want["vol_" + part])     # (the repeat was an ordinary sweep: the whole volume is exact)
    hip.set_refine(4e-6)
    _two_pass(dec, specs, 4, N, mc_only=True)                       # and the mode is still on
    assert hip.band_stats["band"] == 2
    hip.close()


def test_near_level_overflow_in_a_box_sweep():
    """Box-only coarse sweep with an absurd allowance (0.2: every voxel is a candidate) and refine_tau = 0.04: the candidates take
    the two-step form, far more than 2^16 of them lie within refine_tau of the level -> the sweep is refused, its ordinary repeat
    overflows its own near-level list, and the boxes come from ONE sweep on the fp32 chain."""
    N = 128
    dec, specs = _module()
    hip = _hip(dec, specs)
    lat = lambda s: torch.from_numpy(syn.latent_code(s)).cuda()
    vs = 2.0 / (N - 1)
    hip.set_sample(lat(0))
    hip.coarse_finish(hip.coarse_begin(N, [-1.0, -1.0, -1.0], vs))       # calibration
    hip.set_sample(lat(5))
    hip.set_math("f32")
    want = hip.decode_grid(N, [-1.0, -1.0, -1.0], vs)[2].cpu().numpy()
    hip.set_math("f16x3")
    hip.set_refine(0.04)
    hip._box_tau = 0.2
    t = hip.coarse_begin(N, [-1.0, -1.0, -1.0], vs)
    assert t["kind"] == "box"
    got = hip.coarse_finish(t)
    assert hip.box_stats["fallback"] == 1 and hip.math == "f16x3" and not hip._force_f32_once
    assert got[:6].tolist() == want[:6].tolist() and got[8:14].tolist() == want[8:14].tolist() and (got[6], got[14]) == (want[6], want[14])
    hip.set_refine(4e-6)
    hip.close()


def test_candidate_overflow_on_a_small_lattice_is_refused_and_repeated():
    """N = 64 (round 5): a lattice of up to 8 x 32 768 voxels has no two-step candidate form - its five launches did nothing in any
    real sweep of such a lattice - so more than 32 768 candidates (an absurd allowance of 0.2: every voxel) are a list overflow: the box
    sweep is refused and repeated as an ordinary sweep, and the boxes are the ordinary sweep's."""
    N = 64
    dec, specs = _module()
    hip = _hip(dec, specs)
    lat = lambda s: torch.from_numpy(syn.latent_code(s)).cuda()
    vs = 2.0 / (N - 1)
    hip.set_sample(lat(0))
    hip.coarse_finish(hip.coarse_begin(N, [-1.0, -1.0, -1.0], vs))       # calibration
    hip.set_sample(lat(5))
    want = hip.decode_grid(N, [-1.0, -1.0, -1.0], vs)[2].cpu().numpy()
    ok = hip.coarse_finish(hip.coarse_begin(N, [-1.0, -1.0, -1.0], vs))  # an ordinary allowance: accepted, the same boxes
    assert hip.box_stats["fallback"] == 0 and ok[:6].tolist() == want[:6].tolist() and ok[8:14].tolist() == want[8:14].tolist()
    assert (ok[6] != 0, ok[14] != 0) == (want[6] != 0, want[14] != 0)        # (a box sweep's count words only say whether there is a negative voxel)
    hip._box_tau = 0.2
    t = hip.coarse_begin(N, [-1.0, -1.0, -1.0], vs)
    assert t["kind"] == "box"
    got = hip.coarse_finish(t)
    assert hip.box_stats["fallback"] == 1 and hip.box_stats["max_candidates"] > 32768
    assert got[:7].tolist() == want[:7].tolist() and got[8:15].tolist() == want[8:15].tolist()
    hip.close()


def test_three_refused_band_sweeps_in_a_row_switch_the_mode_off_and_lists_beyond_capacity_cost_only_time(monkeypatch):
    """N = 256 through the SAMPLE PIPELINE with an absurd allowance (0.2: every voxel is 'undecided'): the box sweep lists
    more than CAND_CAP candidates, the band sweep marks more than BAND_CAP voxels - every such sweep is refused and repeated as an
    ordinary one, after three in a row each mode switches itself off, and every mesh is the ordinary run's."""
    from alignsdf_amd.reconstruct import pipelined_two_pass, synthetic_code_source
    N = 256
    samples = [0, 1, 2, 3, 4]
    src = synthetic_code_source("nerf3", "cuda")

    def run(force):
        for k in ("ASDF_COARSE", "ASDF_FINE", "ASDF_MATH"):
            monkeypatch.delenv(k, raising=False)
        monkeypatch.setenv("ASDF_FAST", "1")            # (the undo() below also undoes this file's autouse fixture)
        dec, specs = _module()
        hip = _hip(dec, specs)
        if force:
            monkeypatch.setattr(hd.HipSdfDecoder, "_tau_current", lambda self: 0.2)
        else:
            hip.coarse_mode = hip.fine_mode = "exact"
        items = [(s,) + src("s%d" % s, s) for s in samples]
        out = {s: r for s, r in pipelined_two_pass(dec, specs, iter(items), N)}
        monkeypatch.undo()
        return out, hip

    want, _ = run(False)
    got, hip = run(True)
    for s in samples:
        assert got[s]["origin"] == want[s]["origin"] and float(got[s]["voxel_size"]) == float(want[s]["voxel_size"])
        for part in ("hand", "obj"):
            assert torch.equal(got[s]["verts_" + part], want[s]["verts_" + part]) and torch.equal(got[s]["faces_" + part], want[s]["faces_" + part])
    assert hip.band_stats["max_marked"] > hd.BAND_CAP and hip.box_stats["max_candidates"] > hd.CAND_CAP
    # three refusals IN A ROW switch a mode off; a sample that was already enqueued in one go when the third refusal was judged
    # (round 5: a sample is judged when it is finished) adds one more refused sweep - never a delivered one
    assert 3 <= hip.band_stats["fallback"] <= 4 and hip.fine_mode == "exact"
    assert 3 <= hip.box_stats["fallback"] <= 4 and hip.coarse_mode == "exact"
    assert hip.band_stats["band"] == 0 and hip.box_stats["box"] == 0
    hip.close()
