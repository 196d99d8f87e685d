"""Round 5: the host is out of the coarse -> fine hand-off.

asdf_zoom_cube computes get_higher_res_cube's six fp32 operations (utils/mesh.py:239-254) on the device from the boxes of a coarse
pass; asdf_decode_grid_band_dev / asdf_decode_grid_dev read their lattice from those device words, so the fine pass is enqueued right
behind the coarse pass; asdf_mc_emit_bounded emits into buffers sized before the counts are known.  Everything must be bit-equal to the
step-by-step path: the zoom cube to utils.mesh.zoom_cube_from_bboxes (= the reference's CPU tensors, pinned by tests/golden/ref_*),
the volumes to the host-argument entry points, the meshes to ASDF_SPECULATE=0."""
import ctypes

import numpy as np
import pytest
import torch

from alignsdf_amd import _native
from alignsdf_amd import synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _fast_sweeps(monkeypatch):
    """This file is about the OPT-IN audited one-plane sweeps (round 6: the product's default is ordinary sweeps on every voxel;
    ASDF_FAST=1 / --fast / HipSdfDecoder.set_fast select these)."""
    monkeypatch.setenv("ASDF_FAST", "1")


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def test_zoom_cube_on_the_device_is_the_host_arithmetic_bit_for_bit():
    from alignsdf_amd.utils.mesh import zoom_cube_from_bboxes
    L = _native.lib()
    rng = np.random.default_rng(5)
    cases = []
    for N in (16, 32, 64, 96, 128, 256, 257, 512, 1024):
        for _ in range(60):
            lo = rng.integers(0, N, size=(2, 3))
            hi = np.minimum(lo + rng.integers(0, N, size=(2, 3)), N - 1)
            counts = [int(rng.integers(0, 3) > 0) * 7, int(rng.integers(0, 3) > 0) * 9]      # an empty branch now and then
            for hand, obj in ((1, 1), (1, 0), (0, 1)):
                cases.append((N, lo, hi, counts, hand, obj))
    # the boxes every shipped test lattice produces are small integers; the edge cases: whole cube, single voxel, empty both
    cases.append((256, np.zeros((2, 3), int), np.full((2, 3), 255), [1, 1], 1, 1))
    cases.append((256, np.full((2, 3), 77), np.full((2, 3), 77), [1, 1], 1, 1))
    cases.append((64, np.zeros((2, 3), int), np.zeros((2, 3), int), [0, 0], 1, 1))
    bbox = torch.zeros(16, dtype=torch.int32, device="cuda")
    lat = torch.zeros(4, dtype=torch.float32, device="cuda")
    for N, lo, hi, counts, hand, obj in cases:
        voxel = float(np.float32(2.0 / (N - 1)))
        rec = np.zeros(16, dtype=np.int32)
        for h in range(2):
            rec[8 * h:8 * h + 3] = lo[h] if counts[h] else 0x7fffffff
            rec[8 * h + 3:8 * h + 6] = hi[h] if counts[h] else -1
            rec[8 * h + 6] = counts[h]
        bbox.copy_(torch.from_numpy(rec))
        _native.check(L.asdf_zoom_cube(bbox.data_ptr(), N, ctypes.c_float(voxel), hand, obj, lat.data_ptr(), _stream()), "asdf_zoom_cube")
        got = lat.cpu().numpy()
        boxes = [(rec[8 * h:8 * h + 3], rec[8 * h + 3:8 * h + 6], int(rec[8 * h + 6])) for h, on in ((0, hand), (1, obj)) if on]
        nvs, norg = zoom_cube_from_bboxes(boxes, N, 2.0 / (N - 1))
        want = np.array(norg.tolist() + [nvs.item()], dtype=np.float32)
        assert got.tobytes() == want.tobytes(), (N, lo, hi, counts, hand, obj, got, want)


@pytest.mark.parametrize("tag", ["nerf3", "both9", "comb3"])
def test_lattice_from_device_memory_gives_the_same_volumes(tag):
    """asdf_decode_grid_dev / asdf_decode_grid_band_dev against asdf_decode_grid / asdf_decode_grid_band on the same lattice: every
    voxel, every word of the records."""
    from tests.test_gpu_coarse_box import _bind, _decoder
    hip, specs = _decoder(tag)
    N = 64
    L = hip._L
    origin, voxel = [-0.62, -0.36, -0.37], float(np.float32(1.21 / (N - 1)))
    lat = torch.tensor(origin + [voxel], dtype=torch.float32, device="cuda")
    _bind(hip, specs, 1)
    hip.decode_grid(N, [-1.0, -1.0, -1.0], 2.0 / (N - 1))           # calibrates the activation scales
    _bind(hip, specs, 1)
    a_h, a_o, a_b = hip.decode_grid(N, origin, voxel)
    b_h, b_o = torch.empty_like(a_h), torch.empty_like(a_o)
    b_b = torch.empty(16, dtype=torch.int32, device="cuda")
    _native.check(L.asdf_decode_grid_dev(hip._h, N, lat.data_ptr(), 0, b_h.data_ptr(), b_o.data_ptr(), b_b.data_ptr(), _stream()), "asdf_decode_grid_dev")
    assert torch.equal(a_h, b_h) and torch.equal(a_o, b_o) and torch.equal(a_b, b_b)
    # the band sweep (audit off: its picks are a random stream that advances per call)
    hip.set_audit(0)
    org = (ctypes.c_float * 3)(*origin)
    recs, vols = [], []
    for dev in (False, True):
        vh, vo = torch.empty_like(a_h), torch.empty_like(a_o)
        rec = torch.zeros(48, dtype=torch.int32, device="cuda")
        if dev:
            _native.check(L.asdf_decode_grid_band_dev(hip._h, N, lat.data_ptr(), 0, ctypes.c_float(2e-3), vh.data_ptr(), vo.data_ptr(),
                                                      rec.data_ptr(), _stream()), "asdf_decode_grid_band_dev")
        else:
            _native.check(L.asdf_decode_grid_band(hip._h, N, org, ctypes.c_float(voxel), 0, ctypes.c_float(2e-3), vh.data_ptr(), vo.data_ptr(),
                                                  rec.data_ptr(), _stream()), "asdf_decode_grid_band")
        r = rec.cpu().numpy().copy()
        r[28:32] = 0                                                  # (shader-clock stamps of the sweep kernel)
        recs.append(r)
        vols.append((vh, vo))
    assert np.array_equal(recs[0], recs[1]) and torch.equal(vols[0][0], vols[1][0]) and torch.equal(vols[0][1], vols[1][1])
    hip.close()


@pytest.mark.parametrize("tag,N", [("nerf3", 64), ("both9", 96), ("grasp3", 128), ("nerf9", 64)])
def test_samples_enqueued_in_one_go_give_the_step_by_step_meshes(tag, N, monkeypatch):
    """The sample pipeline with and without the speculation: zoom cubes bit-equal, vertices and faces torch.equal, for 12 samples; the
    speculative path really ran (every sample after the decoder's first), nothing was refused."""
    from tests.test_gpu_default_sweeps import _meshes
    samples = list(range(12))
    monkeypatch.setenv("ASDF_SPECULATE", "0")
    want, hip0 = _meshes(tag, N, samples, monkeypatch)
    assert hip0.events["samples_in_one_go"] == 0
    monkeypatch.setenv("ASDF_SPECULATE", "1")
    got, hip = _meshes(tag, N, samples, monkeypatch)
    # sample 0 compares both lattices as a whole; sample 1's first pass is queued before sample 0's fine pass has been judged (the zoom
    # lattice's comparison is evaluated there), so it still goes step by step; from sample 2 on every sample is enqueued in one go
    assert hip.events["samples_in_one_go"] == len(samples) - 2, hip.events
    assert hip.box_stats["fallback"] == 0 and hip.band_stats["fallback"] == 0 and hip.box_stats["box"] == len(samples) - 1
    for s in samples:
        a, b = want[s], got[s]
        assert a[0] == b[0] and a[1] == b[1], (s, "zoom cube", a[:2], b[:2])
        for k in (2, 3, 4, 5):
            assert torch.equal(a[k], b[k]), (s, k)


def test_bounded_emit_and_its_fallback():
    """asdf_mc_emit_bounded: with room it is asdf_mc_emit; with a capacity that is too small marching_cubes_finish runs both phases
    again and returns the same mesh."""
    from alignsdf_amd.marching_cubes import marching_cubes_begin, marching_cubes_device, marching_cubes_finish
    from tests.test_gpu_coarse_box import _bind, _decoder
    hip, specs = _decoder("nerf3")
    N = 96
    _bind(hip, specs, 3)
    vh, vo, _ = hip.decode_grid(N, [-0.62, -0.36, -0.37], 1.21 / (N - 1))
    direct = {id(vol): marching_cubes_device(vol, 0.0) for vol in (vh, vo)}
    for vol in (vh, vo):
        v0, f0 = direct[id(vol)]
        for cap in ((v0.shape[0] + 100, f0.shape[0] + 100), (v0.shape[0], f0.shape[0]), (v0.shape[0] // 2, f0.shape[0] + 5), (v0.shape[0] + 5, 10)):
            # another volume's count phase in between (the next sample's): the ticket's sizes must survive it
            t = marching_cubes_begin(vol, 0.0, 0, capacity=cap)
            other_vol = vo if vol is vh else vh
            other = marching_cubes_begin(other_vol, 0.0, 0)
            v1, f1 = marching_cubes_finish(t)
            assert torch.equal(v0, v1) and torch.equal(f0, f1), cap
            # ... and the OTHER ticket - counted only, on the same workspace, which a too-small capacity above has just used again for
            # its second attempt - notices that (the workspace's generation moved on) and counts again instead of emitting the
            # first volume's scan results with its own sizes (found as an abort of this test, round 5)
            v2, f2 = marching_cubes_finish(other)
            assert torch.equal(v2, direct[id(other_vol)][0]) and torch.equal(f2, direct[id(other_vol)][1]), cap
    hip.close()


def test_a_second_run_on_a_calibrated_decoder_starts_speculating_without_sizes(monkeypatch):
    """bench.py's pattern (warm-up run, then the timed run on the SAME decoder): the second run's very first sample is enqueued in one
    go, but no surface of this run has been seen yet, so there are no sizes for its emit buffers.  Its marching cubes must wait for
    surfaces() - a count phase left alone would be emitted after the NEXT sample's count phase had reused the workspace (round 5: a
    GPU memory fault at N = 128).  Meshes of both runs = the step-by-step run's."""
    from alignsdf_amd.reconstruct import pipelined_two_pass, synthetic_code_source
    from alignsdf_amd.utils.utils import decoder_for
    from tests.test_gpu_default_sweeps import _module
    N, tag = 128, "nerf3"
    src = synthetic_code_source(tag, "cuda")

    def run(dec, specs, samples):
        items = [(s,) + src("s%d" % s, s) for s in samples]
        return {s: (r["origin"], float(r["voxel_size"]), r["verts_hand"], r["faces_hand"], r["verts_obj"], r["faces_obj"])
                for s, r in pipelined_two_pass(dec, specs, iter(items), N)}

    monkeypatch.setenv("ASDF_SPECULATE", "0")
    dec0, specs = _module(tag)
    want = run(dec0, specs, range(10))
    monkeypatch.setenv("ASDF_SPECULATE", "1")
    dec, specs = _module(tag)
    run(dec, specs, range(4))                                   # the warm-up: whole-lattice comparisons, first sizes
    hip = decoder_for(dec, specs)
    before = hip.events["samples_in_one_go"]
    got = run(dec, specs, range(10))
    assert hip.events["samples_in_one_go"] - before == 10       # every sample of the second run, from its first
    for s in range(10):
        a, b = want[s], got[s]
        assert a[0] == b[0] and a[1] == b[1]
        for k in (2, 3, 4, 5):
            assert torch.equal(a[k], b[k]), (s, k)
    assert hip.box_stats["fallback"] == 0 and hip.band_stats["fallback"] == 0


@pytest.mark.parametrize("branches", [(True, False), (False, True)])
def test_combined_decoder_with_one_branch_off_keeps_its_zoom_cube_when_enqueued_in_one_go(branches, monkeypatch):
    """two_pass_begin forced hand = obj = True for a CombinedDecoder BEFORE asdf_zoom_cube, so samples enqueued in
    one go took the zoom cube over BOTH columns' boxes even with HandBranch / ObjectBranch off - while the step-by-step path and the
    reference (get_higher_res_cube, utils/mesh.py:239-247) use the enabled branches only: within one run the first samples and the
    later ones got different fine lattices.  Here: a CombinedDecoder, one branch off, samples that ARE enqueued in one go - zoom cubes
    and meshes equal the step-by-step run's (ASDF_SPECULATE=0), and differ from the both-branches cube (so the test can tell)."""
    from alignsdf_amd.networks.model import build_decoder
    from alignsdf_amd.reconstruct import pipelined_two_pass
    from alignsdf_amd.utils.utils import decoder_for
    N, tag = 64, "comb3"
    hb, ob = branches
    part = "hand" if hb else "obj"
    sd = {k: torch.from_numpy(v) for k, v in syn.full_state_dict(tag).items()}

    def run(specs, samples):
        dec = build_decoder(specs, sd)
        items = [(s, torch.from_numpy(syn.latent_code(s)), None, None) for s in samples]
        out = {s: (r["origin"], float(r["voxel_size"]), r["verts_" + part], r["faces_" + part]) for s, r in pipelined_two_pass(dec, specs, iter(items), N)}
        return out, decoder_for(dec, specs)

    one = dict(syn.specs_for(tag), HandBranch=hb, ObjectBranch=ob)
    monkeypatch.setenv("ASDF_SPECULATE", "0")
    want, hip0 = run(one, range(8))
    assert hip0.events["samples_in_one_go"] == 0
    both, _ = run(syn.specs_for(tag), range(8))
    monkeypatch.setenv("ASDF_SPECULATE", "1")
    got, hip = run(one, range(8))
    assert hip.combined and hip.events["samples_in_one_go"] >= 4, hip.events
    assert any(want[s][:2] != both[s][:2] for s in range(8)), "the one-branch and the two-branch zoom cubes coincide: the test cannot tell"
    for s in range(8):
        assert want[s][0] == got[s][0] and want[s][1] == got[s][1], (s, want[s][:2], got[s][:2])
        assert torch.equal(want[s][2], got[s][2]) and torch.equal(want[s][3], got[s][3]), s
    assert hip.box_stats["fallback"] == 0 and hip.band_stats["fallback"] == 0
