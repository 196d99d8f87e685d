"""The product's DEFAULT sweeps for mesh-producing calls - the audited box-only coarse sweep and the audited narrow-band fine sweep
(alignsdf_amd.hip_decoder: DEFAULT_COARSE / DEFAULT_FINE) - at the size the headline metric is quoted on.

What `create_mesh_combined_decoder` (utils/mesh.py:17-195) delivers is meshes; under the defaults they must be the meshes of ordinary
sweeps bit for bit (and have the faces of the fp32 chain's), for every one of the 64 synthetic samples the benchmark cycles through,
with no sweep refused; and the audit - a random sample of the voxels whose SIGN is all that is trusted, re-evaluated exactly in every
sweep - must actually run, find no contradiction, and be what refuses a sweep whose allowance is wrong."""
import ctypes

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


def _module(tag):
    from alignsdf_amd.networks.model import build_decoder
    specs = syn.specs_for(tag)
    return build_decoder(specs, {k: torch.from_numpy(v) for k, v in syn.full_state_dict(tag).items()}), specs


def _meshes(tag, N, samples, monkeypatch, coarse=None, fine=None, math=None):
    """{sample: (origin, voxel_size, verts_hand, faces_hand, verts_obj, faces_obj)} through the product's sample pipeline."""
    from alignsdf_amd.reconstruct import pipelined_two_pass, synthetic_code_source
    from alignsdf_amd.utils.utils import decoder_for
    for k, v in (("ASDF_COARSE", coarse), ("ASDF_FINE", fine), ("ASDF_MATH", math)):
        monkeypatch.delenv(k, raising=False)
        if v is not None:
            monkeypatch.setenv(k, v)
    dec, specs = _module(tag)
    src = synthetic_code_source(tag, "cuda")
    items = [(s,) + src("s%d" % s, s) for s in samples]
    out = {}
    for s, r in pipelined_two_pass(dec, specs, iter(items), N):
        out[s] = (r["origin"], float(r["voxel_size"]), r["verts_hand"], r["faces_hand"], r["verts_obj"], r["faces_obj"])
    hip = decoder_for(dec, specs, items[0][2])
    return out, hip


@pytest.mark.parametrize("tag", ["nerf3", "both9", "grasp3", "grasp9", "nerf9", "nerf15"])
def test_default_meshes_are_the_ordinary_sweeps_meshes_all_64_samples_n256(tag, monkeypatch):
    """N = 256 (BASELINE configs[2] / configs[4]'s decoder), all 64 synthetic samples - and all 16 scenes of the GRASP family
    (decoders with every layer trained, hands closing on objects in contact), and the NeRF-encoded decoders (their own one-plane
    kernels, round 4): torch.equal on vertices and faces."""
    N = 256
    samples = list(range(syn.GRASP_SAMPLES if tag in syn.GRASP_TAGS else 64))
    want, _ = _meshes(tag, N, samples, monkeypatch, coarse="exact", fine="exact")
    got, hip = _meshes(tag, N, samples, monkeypatch)                     # the defaults
    assert (hip.coarse_mode, hip.fine_mode) == ("box", "band") and hip.math == "f16x3"
    for s in samples:
        a, b = want[s], got[s]
        assert a[0] == b[0] and a[1] == b[1], (s, "zoom cube")
        for k in (2, 3, 4, 5):
            assert torch.equal(a[k], b[k]), (s, k)
    # sample 0's coarse pass calibrated the allowance on an ordinary sweep of the coarse lattice, its fine pass on one of the zoom
    # lattice (round 5); everything else ran on the one-plane kernel
    n = len(samples)
    assert hip.box_stats["box"] == n - 1 and hip.band_stats["band"] == n - 1, (hip.box_stats, hip.band_stats)
    c = hip.certificate()
    assert c["calibrations"] == 1 and c["fine_calibrations"] == 1 and c["fine_tail_ratio"] <= 3.0, c
    print(tag, "zoom lattice: max error %.3g, sigma %.3g, max / sigma %.1f, tail ratio %.2f (TAIL_MAX 3.0: margin %.2f) | coarse: %.3g, %.1f, %.2f" % (
        c["fine_lattice_max_error"], c["fine_lattice_sigma"], c["fine_max_over_sigma"], c["fine_tail_ratio"], 3.0 / c["fine_tail_ratio"],
        c["lattice_max_error"], c["lattice_max_over_sigma"], c["tail_ratio"]))
    assert hip.box_stats["fallback"] == 0 and hip.band_stats["fallback"] == 0
    # the audit ran in every sweep (voxels x heads), found no sign contradiction, and its error stayed inside the allowance
    # (half of the 65 536 picks per head are uniform, half come from the at-risk shell - as many as it holds)
    assert hip.box_stats["audit_evals"] >= (n - 1) * 2 * 30000 and hip.band_stats["audit_evals"] >= n * 2 * 30000
    cert = hip.certificate()
    assert cert["shell_picks"] > 0 and cert["calibrations"] == 1 and cert["refusals_for_error"] == 0
    assert cert["min_margin_tau_over_estimate"] >= 1.0 / 0.6 and cert["tail_ratio"] <= 3.0
    assert hip.box_stats["audit_flips"] == 0 and hip.band_stats["audit_flips"] == 0
    assert 0.0 < hip.band_stats["audit_max_err"] * hip._tail <= 0.6 * hip.band_stats["tau_max"]
    print(tag, "box", hip.box_stats, "band", hip.band_stats, "certificate", cert)


def test_default_meshes_have_the_faces_of_the_fp32_chain(monkeypatch):
    """Against the fp32 MFMA chain (the strict-reading arithmetic): identical faces and counts; vertices within the two arithmetics'
    difference (a few 1e-7 in the SDF values moves an interpolated vertex by 1e-7 / |v0 - v1| of a voxel: up to a few 1e-2 on the
    rare edges whose two corner values are both within 1e-5 of the level, 1e-4 typically)."""
    N = 128
    samples = list(range(12))
    want, _ = _meshes("nerf3", N, samples, monkeypatch, coarse="exact", fine="exact", math="f32")
    got, hip = _meshes("nerf3", N, samples, monkeypatch)
    for s in samples:
        a, b = want[s], got[s]
        assert a[0] == b[0] and a[1] == b[1]
        assert torch.equal(a[3], b[3]) and torch.equal(a[5], b[5]), s
        assert float((a[2] - b[2]).abs().max()) <= 5e-2 and float((a[4] - b[4]).abs().max()) <= 5e-2
        assert float((a[2] - b[2]).abs().mean()) <= 1e-4 and float((a[4] - b[4]).abs().mean()) <= 1e-4
    assert hip.band_stats["fallback"] == 0 and hip.box_stats["fallback"] == 0


def _raw_band(hip, N, origin, vs, tau, entry="asdf_decode_grid_band"):
    from alignsdf_amd import _native
    vh = torch.empty((N, N, N), dtype=torch.float32, device="cuda")
    vo = torch.empty((N, N, N), dtype=torch.float32, device="cuda")
    rec = torch.zeros(48, dtype=torch.int32, device="cuda")
    org = (ctypes.c_float * 3)(*origin)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _native.check(getattr(hip._L, entry)(hip._h, N, org, ctypes.c_float(vs), 0, ctypes.c_float(tau), vh.data_ptr(), vo.data_ptr(),
                                         rec.data_ptr(), st), entry)
    return vh, vo, rec.cpu().numpy()


def test_audit_record_of_the_c_abi():
    """The audit through the C ABI: it evaluates the requested number of UNMARKED voxels per head, its error is a lower bound of the
    lattice maximum of |one-plane - exact| and of the same order, it is reproducible under a seed, off when set to 0 - and with a
    deliberately understated allowance it reports an error above tau / 2 (what makes the caller refuse the sweep)."""
    from alignsdf_amd.hip_decoder import HipSdfDecoder
    hip = HipSdfDecoder(syn.full_state_dict("nerf3"), 256, 3, "nerf")
    hip.set_sample(torch.from_numpy(syn.latent_code(3)).cuda())
    N = 96
    origin, vs = [-0.62, -0.36, -0.37], 1.21 / (N - 1)
    eh, eo, _ = hip.decode_grid(N, origin, vs)                      # ordinary volumes (this also calibrates the activation scales)
    f = lambda w: float(np.int32(w).view(np.float32))
    # one-plane values: the scratch volumes of a box sweep with a tiny allowance (next to no candidates) and no audit
    hip.set_audit(0)
    ph, po, r0 = _raw_band(hip, N, origin, vs, 1e-7, "asdf_decode_grid_box")
    assert int(r0[37]) == 0 and int(r0[35]) == 0 and int(r0[32]) < 64
    lattice_max = max(float((ph - eh).abs().max()), float((po - eo).abs().max()))
    assert 1e-5 < lattice_max < 5e-3
    # the audit of an honest sweep
    tau = 5.0 * lattice_max
    hip.set_audit(1 << 16, seed=1234)
    _, _, r1 = _raw_band(hip, N, origin, vs, tau)
    marked = (int(r1[33]), int(r1[34]))
    assert all(0 < m < N ** 3 // 2 for m in marked)
    assert int(r1[36]) == 0                                         # no audited voxel has the other sign
    # the audit of a 96^3 lattice: 96^3 / 64 = 13 824 voxels per head - half drawn uniformly (picks that fell on marked voxels are
    # dropped), half from the at-risk shell (unmarked, tau <= |one-plane| < 2 tau: all of it while it is smaller than the budget)
    uniform, budget = HipSdfDecoder.audit_sizes(1 << 16, N ** 3)
    assert (uniform, budget) == (6912, 6912)
    shell_picks, shell_population = int(r1[39]), int(r1[40])
    # (this zoom lattice is steep against the allowance - 6e-3 of SDF per voxel, tau 2e-3: a voxel within 2 tau of the level sits in
    # a cell with a sign change and is MARKED, so the shell of unmarked voxels is next to empty: an exhaustive check of nothing)
    assert 0 <= shell_picks == shell_population <= 2 * budget
    assert 2 * uniform * 0.75 + shell_picks <= int(r1[37]) <= 2 * uniform + shell_picks
    assert 0.2 * lattice_max <= f(r1[35]) <= lattice_max + 1e-6
    sigma = (f(r1[41]) / int(r1[37])) ** 0.5                        # rms of the audit's errors next to their maximum
    assert 0.0 < sigma < f(r1[35]) and f(r1[35]) / sigma < 12.0
    # a wide allowance (one voxel of SDF): the shell tau <= |v| < 2 tau now holds thousands of unmarked voxels per head.  With the
    # full audit (budget 6 912 per head) every one of them is re-evaluated - an exhaustive check; with a small audit (budget 2048)
    # the shell is thinned to about that many picks per head
    for mult in np.arange(16.0, 30.5, 0.5):     # (the shell grows steeply with the allowance: take the first one that fills it as wanted)
        wide = float(mult) * lattice_max
        hip.set_audit(1 << 16, seed=77)
        _, _, rw = _raw_band(hip, N, origin, vs, wide)
        if 4096 < int(rw[40]) <= budget:
            break
    # (the population word sums the heads: at most `budget` in all means at most `budget` per head - every shell voxel is re-evaluated)
    assert 4096 < int(rw[40]) <= budget and int(rw[39]) == int(rw[40]) and int(rw[36]) == 0, (mult, int(rw[40]), int(rw[39]))
    hip.set_audit(4096, seed=78)
    assert HipSdfDecoder.audit_sizes(4096, N ** 3) == (2048, 2048)
    _, _, rt = _raw_band(hip, N, origin, vs, wide)
    # thinned: fewer picks than shell voxels, at most about the two budgets, at least about one (the larger head's shell exceeds its budget)
    assert int(rt[40]) == int(rw[40]) and 2048 * 0.85 <= int(rt[39]) <= min(2 * 2048 * 1.15, int(rt[40]) - 1) and int(rt[36]) == 0
    assert int(rt[37]) >= int(rt[39]) + 2 * 2048 * 0.5
    hip.set_audit(1 << 16, seed=1234)
    assert f(r1[19]) <= lattice_max + 1e-6
    hip.set_audit(1 << 16, seed=1234)
    _, _, r2 = _raw_band(hip, N, origin, vs, tau)
    assert int(r2[35]) == int(r1[35]) and int(r2[37]) == int(r1[37])      # same seed, same draw
    _, _, r3 = _raw_band(hip, N, origin, vs, tau)
    assert int(r3[37]) > 0 and (int(r3[35]), int(r3[37])) != (int(r1[35]), int(r1[37]))      # the seed advances with every sweep
    # an allowance 16 x too small: fewer voxels are marked, and the audit - looking where nothing else does - sees an error > tau / 2
    small = lattice_max / 4.0
    _, _, r4 = _raw_band(hip, N, origin, vs, small)
    assert f(r4[35]) > 0.5 * small
    hip.close()


def test_python_layer_refuses_on_the_audit_alone():
    """The re-evaluated (marked) voxels sit next to the surface; make THEIR check pass trivially and the allowance wrong: only the
    audit can notice.  The sweep must be refused and repeated as an ordinary one."""
    from alignsdf_amd.hip_decoder import HipSdfDecoder
    hip = HipSdfDecoder(syn.full_state_dict("nerf3"), 256, 3, "nerf")
    N = 96
    lat = lambda s: torch.from_numpy(syn.latent_code(s)).cuda()
    hip.set_sample(lat(0))
    hip.coarse_finish(hip.coarse_begin(N, [-1.0, -1.0, -1.0], 2.0 / (N - 1)))      # calibrates the allowance on the coarse lattice
    _, _, ticket = hip.fine_begin(N, [-0.62, -0.36, -0.37], 1.21 / (N - 1), mc_only=True)      # ... and this on the zoom lattice
    assert ticket["kind"] == "exact" and "compare" in ticket and not hip.fine_needs_repeat(ticket) and hip._fine_valid(N)
    honest = hip._box_tau
    hip.set_sample(lat(1))
    real_judge = hip._judge

    def judge_without_marked_error(r, tau, stats, cap_word, cap, **kw):
        r = r.copy()
        r[19] = 0                              # pretend the re-evaluated voxels showed no error at all
        return real_judge(r, tau, stats, cap_word, cap, **kw)

    hip._judge = judge_without_marked_error
    hip._box_tau = honest / 32.0
    bh, bo, ticket = hip.fine_begin(N, [-0.62, -0.36, -0.37], 1.21 / (N - 1), mc_only=True)
    assert ticket["kind"] == "band" and hip.fine_needs_repeat(ticket)
    assert hip.band_stats["fallback"] == 1 and hip.band_stats["audit_max_err"] * max(hip._tail, hip._tail_fine) > 0.6 * honest / 32.0
    assert not hip._fine_valid(N)
    bh, bo, ticket = hip.fine_begin(N, [-0.62, -0.36, -0.37], 1.21 / (N - 1), mc_only=True)
    # the repeat is an ordinary sweep - and, the comparison of the zoom lattice being void, it measures it again on this very lattice
    assert ticket["kind"] == "exact" and "compare" in ticket and not hip.fine_needs_repeat(ticket)
    assert hip._fine_valid(N) and hip.certificate()["fine_calibrations"] == 2
    # a sweep refused for its error does not inflate the allowance by itself: the allowance is VOID until the next
    # coarse pass has compared the whole coarse lattice again
    assert not hip._allowance_valid() and hip.certificate()["refusals_for_error"] == 1
    _, _, ticket = hip.fine_begin(N, [-0.62, -0.36, -0.37], 1.21 / (N - 1), mc_only=True)
    assert ticket["kind"] == "exact"
    hip._judge = real_judge
    t = hip.coarse_begin(N, [-1.0, -1.0, -1.0], 2.0 / (N - 1))
    assert t["kind"] == "exact"
    hip.coarse_finish(t)
    assert hip._allowance_valid() and hip.certificate()["calibrations"] == 2 and 0.25 * honest <= hip._box_tau <= 4.0 * honest
    _, _, ticket = hip.fine_begin(N, [-0.62, -0.36, -0.37], 1.21 / (N - 1), mc_only=True)
    assert ticket["kind"] == "band" and not hip.fine_needs_repeat(ticket)
    hip.close()


def test_whole_lattice_recalibration_is_periodic(monkeypatch):
    """the tail ratio is re-measured every RECAL_EVERY samples (here 5), not once per decoder, in
    turn on the coarse lattice (that coarse pass runs an ordinary sweep next to a plain one-plane one and compares all 2 N^3 values;
    its boxes are the ordinary sweep's) and on the ZOOM lattice (the fine pass does the same; its volumes are the ordinary sweep's)."""
    from alignsdf_amd import hip_decoder as hd
    monkeypatch.setattr(hd, "RECAL_EVERY", 5)
    hip = hd.HipSdfDecoder(syn.full_state_dict("nerf3"), 256, 3, "nerf")
    N = 64
    kinds, fine_kinds = [], []
    for s in range(13):
        hip.set_sample(torch.from_numpy(syn.latent_code(s)).cuda())
        t = hip.coarse_begin(N, [-1.0, -1.0, -1.0], 2.0 / (N - 1))
        kinds.append(t["kind"])
        hip.coarse_finish(t)
        _, _, f = hip.fine_begin(N, [-0.62, -0.36, -0.37], 1.21 / (N - 1), mc_only=True)
        fine_kinds.append(f["kind"] + ("+compare" if "compare" in f else ""))
        assert not hip.fine_needs_repeat(f)
    assert kinds == ["exact"] + ["box"] * 5 + ["exact"] + ["box"] * 6
    assert fine_kinds == ["exact+compare"] + ["band"] * 11 + ["exact+compare"]
    cert = hip.certificate()
    assert cert["calibrations"] == 2 and 1.0 <= cert["tail_ratio"] <= cert["tail_ratio_max"] <= 3.0
    assert cert["fine_calibrations"] == 2 and 1.0 <= cert["fine_tail_ratio"] <= cert["fine_tail_ratio_max"] <= 3.0
    assert cert["lattice_max_over_sigma"] > 1.0 and 0.0 <= cert["neighbour_correlation"] < 1.0
    assert cert["fine_max_over_sigma"] > 1.0 and cert["fine_lattice_max_error"] > 0.0 and 0.0 <= cert["fine_neighbour_correlation"] < 1.0
    assert hip.box_stats["fallback"] == 0 and hip.band_stats["fallback"] == 0 and hip.events["repeated_sweeps"] == 0
    print("certificate after 2 + 2 whole-lattice comparisons:", cert)
    hip.close()


def test_allowance_follows_the_samples():
    """tau is re-estimated from every sweep's audit: after a run of samples it is 4 x the largest recent estimate, not the value the
    first sample calibrated."""
    from alignsdf_amd.hip_decoder import HipSdfDecoder
    hip = HipSdfDecoder(syn.full_state_dict("nerf3"), 256, 3, "nerf")
    N = 64
    taus = []
    for s in range(12):
        hip.set_sample(torch.from_numpy(syn.latent_code(s)).cuda())
        hip.coarse_finish(hip.coarse_begin(N, [-1.0, -1.0, -1.0], 2.0 / (N - 1)))
        taus.append(hip._box_tau)
        _, _, t = hip.fine_begin(N, [-0.62, -0.36, -0.37], 1.21 / (N - 1), mc_only=True)
        assert not hip.fine_needs_repeat(t)
    assert len(hip._err_window) == 16 and abs(hip._box_tau - 5.0 * max(hip._err_window)) <= 1e-12
    assert len(set(taus)) > 1 and all(1e-5 < t < 0.05 for t in taus)
    assert hip.box_stats["fallback"] == 0 and hip.band_stats["fallback"] == 0
    hip.close()


@pytest.mark.parametrize("name", ["spread1e4", "huge", "tiny", "heavy_tails", "gain30", "latent_x10"])
def test_default_sweeps_on_adversarial_decoders_give_the_ordinary_sweeps_meshes(name):
    """The decoders built to break the fp16 arithmetic (tests/test_gpu_split_half_adversarial.py: magnitude spreads of 1e4,
    activations far outside the default scale's range, heavy-tailed weights, weight_g = 30, latent x 10) under the DEFAULT sweeps:
    whatever the one-plane error is there - small, large enough to widen the lists to their capacity, or too large for the
    allowance so that the mode switches itself off - boxes, zoom cube and both meshes are those of ordinary sweeps, bit for bit."""
    from alignsdf_amd import _native
    from alignsdf_amd.hip_decoder import HipSdfDecoder
    from alignsdf_amd.marching_cubes import marching_cubes_device
    from alignsdf_amd.utils.mesh import zoom_cube_from_bboxes
    from tests.test_gpu_split_half_adversarial import variant
    sd, lat = variant(name)
    N, G = 48, _native.GRID_REFERENCE
    voxel = 2.0 / (N - 1)

    def run(hip, latent):
        hip.set_sample(torch.from_numpy(latent).cuda())
        b = hip.coarse_finish(hip.coarse_begin(N, [-1.0, -1.0, -1.0], voxel, G))
        nvs, norg = zoom_cube_from_bboxes([(b[0:3], b[3:6], int(b[6])), (b[8:11], b[11:14], int(b[14]))], N, voxel)
        vh, vo, t = hip.fine_begin(N, norg.tolist(), nvs.item(), G, mc_only=True)
        while hip.fine_needs_repeat(t):
            vh, vo, t = hip.fine_begin(N, norg.tolist(), nvs.item(), G, mc_only=True)
        # (words 6 / 14 are the negative-voxel COUNT of an ordinary sweep and merely non-zero iff there is one after a box-only sweep)
        out = [[int(x) for x in b[0:6]] + [int(b[6]) != 0], [int(x) for x in b[8:14]] + [int(b[14]) != 0], float(nvs), norg.tolist()]
        for vol in (vh, vo):
            try:
                out.append(marching_cubes_device(vol, 0.0))
            except (ValueError, RuntimeError) as e:          # no surface in this volume: the same message either way
                out.append(str(e))
        return out

    ordinary, default = HipSdfDecoder(sd, 256, 3, "nerf"), HipSdfDecoder(sd, 256, 3, "nerf")
    ordinary.coarse_mode = ordinary.fine_mode = "exact"
    assert (default.coarse_mode, default.fine_mode) == ("box", "band")
    surfaces = 0
    # a stream whose one-plane error jumps between neighbours: codes scaled by 1, 0.93, 4, 0.86, 0.25, 0.79, 4, ... - a sweep whose
    # error leaves the allowance the previous samples set is refused and repeated, never trusted
    scales = [1.0, 0.93, 4.0, 0.86, 0.25, 0.79, 4.0, 0.72, 0.25, 0.65] if name != "latent_x10" else [1.0 - 0.07 * k for k in range(6)]
    for k, sc in enumerate(scales):
        latent = (lat * np.float32(sc)).astype(np.float32)
        want, got = run(ordinary, latent), run(default, latent)
        assert want[:4] == got[:4], (name, k, want[:4], got[:4])
        for a, b in zip(want[4:], got[4:]):
            if isinstance(a, str) or isinstance(b, str):
                assert a == b, (name, k, a, b)
            else:
                surfaces += 1
                assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), (name, k)
    print(name, "surfaces compared", surfaces, "modes now", default.coarse_mode, default.fine_mode, "math", default.math,
          "box", {k: default.box_stats[k] for k in ("box", "exact", "fallback", "max_err", "tau_max")},
          "band", {k: default.band_stats[k] for k in ("band", "exact", "fallback", "max_err", "max_marked")})
    assert default.box_stats["audit_flips"] == 0 and default.band_stats["audit_flips"] == 0
    ordinary.close(); default.close()


@pytest.mark.parametrize("tag", ["grasp3", "grasp9"])
@pytest.mark.parametrize("N", [64, 96, 128])
def test_grasp_family_at_other_sizes_and_single_branches(tag, N, monkeypatch):
    """The trained grasp decoders away from the headline size: N = 64 / 96 (not a multiple of 64: ragged tiles) / 128, both branches
    and each branch alone (HandBranch / ObjectBranch off: the other head is not evaluated, utils/mesh.py:239-247), 8 scenes - among them
    the ones with a detached blob / piece: zoom cube and every mesh of the default sweeps equal those of ordinary sweeps, no sweep
    refused for its error."""
    from alignsdf_amd.marching_cubes import marching_cubes_device
    from alignsdf_amd.utils.mesh import decode_two_pass
    from alignsdf_amd.utils.utils import decoder_for
    for k in ("ASDF_COARSE", "ASDF_FINE", "ASDF_MATH"):
        monkeypatch.delenv(k, raising=False)
    scenes = [0, 1, 3, 5, 7, 10, 13, 15]

    def run(exact):
        dec, specs = _module(tag)
        out = {}
        hip = None
        for hand, obj in ((True, True), (True, False), (False, True)):
            for s in scenes:
                lat, m, o = syn.sample_inputs(tag, s)
                lat = torch.from_numpy(lat).cuda()
                mano = {k: torch.from_numpy(v).cuda() for k, v in m.items()} if m is not None else None
                objr = {k: torch.from_numpy(v).cuda() for k, v in o.items()} if o is not None else None
                if hip is None:
                    hip = decoder_for(dec, specs, mano)
                    if exact:
                        hip.coarse_mode = hip.fine_mode = "exact"
                r = decode_two_pass(hand, obj, dec, lat, mano, objr, specs, N, mc_only=True)
                meshes = []
                for part, on in (("hand", hand), ("obj", obj)):
                    if on:
                        try:
                            meshes.append(marching_cubes_device(r["vol_" + part], 0.0))
                        except (ValueError, RuntimeError) as e:
                            meshes.append(str(e))
                out[(hand, obj, s)] = (r["origin"], float(r["voxel_size"]), meshes)
        return out, hip

    want, _ = run(True)
    got, hip = run(False)
    assert (hip.coarse_mode, hip.fine_mode) == ("box", "band")
    for key in want:
        a, b = want[key], got[key]
        assert a[0] == b[0] and a[1] == b[1], key
        for x, y in zip(a[2], b[2]):
            if isinstance(x, str) or isinstance(y, str):
                assert x == y, key
            else:
                assert torch.equal(x[0], y[0]) and torch.equal(x[1], y[1]), key
    c = hip.certificate()
    assert c["refusals_for_error"] == 0 and hip.band_stats["audit_flips"] == 0 and hip.box_stats["audit_flips"] == 0
    assert hip.band_stats["band"] >= 20, hip.band_stats
    print(tag, N, "box", {k: hip.box_stats[k] for k in ("box", "exact", "fallback", "max_candidates")},
          "band", {k: hip.band_stats[k] for k in ("band", "exact", "fallback", "max_marked", "tau_max")}, "tail", c["tail_ratio"])


def test_grasp_decoder_on_latent_codes_it_was_not_trained_on(monkeypatch):
    """The certificate of the default sweeps was calibrated on the 16 trained codes; a deployed decoder sees codes an ENCODER produces.
    Stand-ins: midpoints of trained codes, extrapolations beyond them (1.6 a - 0.6 b) and trained codes with noise of the codes' own
    scale - shapes the decoder never fitted (merged, thinned, partly dissolved surfaces).  N = 128, ObMan-shaped trained decoder: whatever
    the ordinary sweeps deliver (a mesh, or marching cubes' refusal of a volume without a surface), the default sweeps deliver the same,
    vertex for vertex; a sweep may be REFUSED and repeated as an ordinary one (counted, reported) but never deliver anything else."""
    from alignsdf_amd.marching_cubes import marching_cubes_device
    from alignsdf_amd.utils.mesh import decode_two_pass
    from alignsdf_amd.utils.utils import decoder_for
    for k in ("ASDF_COARSE", "ASDF_FINE", "ASDF_MATH"):
        monkeypatch.delenv(k, raising=False)
    tag, N = "grasp3", 128
    codes = np.concatenate([syn.sample_inputs(tag, s)[0] for s in range(syn.GRASP_SAMPLES)], 0)
    scale = float(codes.std())
    latents = []
    for i in range(8):
        a, b = codes[i], codes[(5 * i + 3) % syn.GRASP_SAMPLES]
        latents += [0.5 * (a + b), 1.6 * a - 0.6 * b,
                    a + scale * syn.uniform((codes.shape[1],), 4100 + i, -1.0, 1.0).astype(np.float32)]
    latents = [torch.from_numpy(np.asarray(v, dtype=np.float32).reshape(1, -1)).cuda() for v in latents]

    def run(exact):
        dec, specs = _module(tag)
        hip = decoder_for(dec, specs, None)
        if exact:
            hip.coarse_mode = hip.fine_mode = "exact"
        out = []
        for lat in latents:
            r = decode_two_pass(True, True, dec, lat, None, None, specs, N, mc_only=True)
            meshes = []
            for part in ("hand", "obj"):
                try:
                    meshes.append(marching_cubes_device(r["vol_" + part], 0.0))
                except (ValueError, RuntimeError) as e:
                    meshes.append(str(e))
            out.append((r["origin"], float(r["voxel_size"]), meshes))
        return out, hip

    want, _ = run(True)
    got, hip = run(False)
    surfaces = 0
    for i, (a, b) in enumerate(zip(want, got)):
        assert a[0] == b[0] and a[1] == b[1], (i, "zoom cube")
        for ma, mb in zip(a[2], b[2]):
            if isinstance(ma, str):
                assert ma == mb, (i, ma, mb)
            else:
                surfaces += 1
                assert torch.equal(ma[0], mb[0]) and torch.equal(ma[1], mb[1]), i
    assert surfaces >= len(latents)                       # (most of these codes still decode to two surfaces)
    assert hip.box_stats["audit_flips"] == 0 and hip.band_stats["audit_flips"] == 0
    print("unseen codes: %d, surfaces %d, one-plane coarse sweeps %d (refused %d), fine %d (refused %d), certificate %s" % (
        len(latents), surfaces, hip.box_stats["box"], hip.box_stats["fallback"], hip.band_stats["band"], hip.band_stats["fallback"],
        hip.certificate()))
    hip.close()
