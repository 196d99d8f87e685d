"""The sweep state machine of alignsdf_amd.hip_decoder.HipSdfDecoder as a TABLE, on the CPU.

Which kernels evaluate a lattice is decided on the host from 48-word records the sweeps leave behind: coarse_begin / coarse_finish,
fine_begin / fine_needs_repeat, _judge, fall_back_if_overflowed, decode_grid's one-sweep-on-the-fp32-chain switch - 250 lines of
interacting flags (_band_skip, _force_f32_once, scale epochs, allowance epochs, failure counters).  Here the REAL methods run
against a stand-in for libalignsdf_hip.so that logs every launch (entry point, arithmetic in force) and hands back scripted
records: every refusal reason of _judge x {fresh ticket, ticket launched under activation scales re-calibrated since}, for the
coarse and the fine pass, and what must come out whatever path ran - the reference's flow needs boxes from pass 1 and two volumes
from pass 2 (utils/mesh.py:46-63, :98-115) - i.e. which launch comes NEXT."""
import contextlib
import ctypes

import numpy as np
import pytest
import torch

from alignsdf_amd import hip_decoder as hd


@pytest.fixture(autouse=True)
def _fast_sweeps(monkeypatch):
    """This file is about the OPT-IN audited one-plane sweeps (round 6: the product's default is ordinary sweeps on every voxel;
    ASDF_FAST=1 / --fast / HipSdfDecoder.set_fast select these)."""
    monkeypatch.setenv("ASDF_FAST", "1")


N = 32
ARGS = (N, [-1.0, -1.0, -1.0], 2.0 / (N - 1))
F = lambda x: int(np.float32(x).view(np.int32))


def good_record(tau):
    """Record of an accepted one-plane sweep: boxes present, error on the re-evaluated voxels and audit error a tenth of tau."""
    r = np.zeros(48, dtype=np.int32)
    r[0:3], r[3:6], r[6] = (3, 4, 5), (20, 21, 22), 1
    r[8:11], r[11:14], r[14] = (6, 7, 8), (17, 18, 19), 1
    r[19], r[35] = F(0.1 * tau), F(0.1 * tau)
    r[32], r[33], r[34] = 900, 5000, 4000
    r[37], r[39], r[40], r[41] = 2000, 600, 700, F(2000 * (0.03 * tau) ** 2)
    return r


# reason -> (mutation of the good record, refused for its ERROR (allowance void)?, counts as a failure of the mode?)
def _range(r, tau): r[7] = 5
def _listed_box(r, tau): r[32] = hd.CAND_CAP + 1
def _listed_band(r, tau): r[33] = hd.BAND_CAP + 1
def _near(r, tau): r[38] = 3
def _near_bit(r, tau): r[15] |= hd.NEAR_OVERFLOW_BIT
def _contradiction(r, tau): r[18] = 1
def _flips(r, tau): r[36] = 2
def _err(r, tau): r[19] = F(0.7 * tau)
def _audit(r, tau): r[35] = F(0.7 * tau)
def _no_audit(r, tau): r[37] = 0


REASONS = {"range": (_range, False), "near": (_near, False), "near_bit": (_near_bit, False), "contradiction": (_contradiction, True),
           "flips": (_flips, True), "error": (_err, True), "audit": (_audit, True), "no_audit": (_no_audit, False)}


class FakeLib:
    """Logs (entry point, arithmetic in force) and writes the next scripted record into the caller's buffer."""

    def __init__(self, dec):
        self.dec, self.log, self.script = dec, [], []

    def _write(self, ptr, words):
        buf = (ctypes.c_int32 * len(words)).from_address(ptr)
        buf[:] = [int(w) for w in words]

    def asdf_decode_grid(self, h, n, org, vs, mode, hand, obj, bbox, stream):
        self.log.append(("grid", self.dec.math))
        if bbox:
            self._write(bbox, good_record(1.0)[:16])
        return 0

    def asdf_decode_grid_dev(self, h, n, lattice, mode, hand, obj, bbox, stream):
        self.log.append(("grid_dev", self.dec.math))
        if bbox:
            self._write(bbox, self.grid_dev_record if getattr(self, "grid_dev_record", None) is not None else good_record(1.0)[:16])
        return 0

    def _one_plane(self, name, rec):
        self.log.append((name, self.dec.math))
        self._write(rec, self.script.pop(0))
        return 0

    def asdf_decode_grid_box(self, h, n, org, vs, mode, tau, sh, so, rec, stream):
        return self._one_plane("box", rec)

    def asdf_decode_grid_band(self, h, n, org, vs, mode, tau, sh, so, rec, stream):
        return self._one_plane("band", rec)

    def asdf_decode_grid_band_dev(self, h, n, lattice, mode, tau, sh, so, rec, stream):
        return self._one_plane("band_dev", rec)

    def asdf_zoom_cube(self, bbox, n, vs, hand, obj, lattice, stream):
        self.log.append(("zoom", self.dec.math))
        (ctypes.c_float * 4).from_address(lattice)[:] = [-0.5, -0.25, -0.125, 0.0078125]
        return 0

    def asdf_decoder_set_math(self, h, code):
        return 0

    def asdf_decoder_set_audit(self, h, n, seed):
        return 0


@pytest.fixture()
def machine(monkeypatch):
    monkeypatch.setattr(torch.cuda, "device", lambda d: contextlib.nullcontext())
    for k in ("ASDF_MATH", "ASDF_COARSE", "ASDF_FINE"):
        monkeypatch.delenv(k, raising=False)
    dec = hd.HipSdfDecoder.__new__(hd.HipSdfDecoder)
    dec.combined, dec.nerf_features, dec.point_feat_size = False, False, 3
    dec.device, dec._h = torch.device("cpu"), None
    dec._L = FakeLib(dec)
    dec._stream = lambda: None
    dec._init_sweep_state()
    dec._calibrated = True                          # the activation scales: not under test here
    calls = []

    def fake_calibrate(args, vols):                 # the whole-lattice comparison: sets the allowance, as the real one does
        calls.append("calibrate")
        dec._coarse_since_cal, dec._box_epoch, dec._cal_points = 0, dec._recalibrations, int(args[0]) ** 3
        dec._err_window.append(2e-4)
        dec._box_tau = dec._tau_current()
        dec.cert["calibrations"] += 1

    def fake_recover(bad, status=None):             # fp16 range violation -> new activation scales -> a new scale epoch
        calls.append("recover")
        dec._recalibrations += 1

    def fake_plain(args):                           # the plain one-plane sweep of a whole-lattice comparison
        dec._L.log.append(("plain", dec.math))
        return (None, None, None)

    def fake_calibrate_fine(args, fast, exact, periodic=False):      # the same comparison on the zoom lattice of a fine pass
        calls.append("calibrate_fine")
        if periodic:
            dec._coarse_since_cal, dec._next_recal = 0, "coarse"
        dec._fine_epoch, dec._fine_cal_points = dec._recalibrations, int(args[0]) ** 3
        dec._err_window.append(2e-4)
        dec._box_tau = dec._tau_current()
        dec.cert["fine_calibrations"] += 1

    dec._calibrate_box, dec._recover, dec._plain_one_plane, dec._calibrate_fine = fake_calibrate, fake_recover, fake_plain, fake_calibrate_fine
    dec._status = lambda clear: np.zeros(16, dtype=np.int32)
    dec.close = lambda: None
    # first coarse pass: ordinary sweep + calibration
    t = dec.coarse_begin(*ARGS)
    assert t["kind"] == "exact"
    dec.coarse_finish(t)
    assert calls == ["calibrate"] and dec._allowance_valid(N) and dec._L.log == [("grid", "f16x3")]
    # first fine pass: ordinary sweep + a plain one-plane sweep of the same ZOOM lattice, compared when the record is read
    assert not dec._fine_valid(N)
    _, _, t = dec.fine_begin(*ARGS, mc_only=True)
    assert t["kind"] == "exact" and "compare" in t and dec._L.log[1:] == [("grid", "f16x3"), ("plain", "f16x3")]
    assert not dec.fine_needs_repeat(t) and calls == ["calibrate", "calibrate_fine"] and dec._fine_valid(N)
    dec._L.log.clear()
    calls.clear()
    dec.band_stats["exact"] = 0
    return dec, calls


def prime_fine(dec, calls):
    """The fine pass that follows a voided allowance: ordinary + whole-lattice comparison of the zoom lattice."""
    _, _, t = dec.fine_begin(*ARGS, mc_only=True)
    assert t["kind"] == "exact" and "compare" in t and not dec.fine_needs_repeat(t)
    assert calls[-1] == "calibrate_fine" and dec._fine_valid(N)


def test_accepted_sweeps_stay_on_the_one_plane_kernel(machine):
    dec, calls = machine
    tau = dec._box_tau
    dec._L.script = [good_record(tau), good_record(tau)]
    t = dec.coarse_begin(*ARGS)
    b = dec.coarse_finish(t)
    assert t["kind"] == "box" and b[:16].tolist() == good_record(tau)[:16].tolist()
    _, _, t = dec.fine_begin(*ARGS, mc_only=True)
    assert t["kind"] == "band" and not dec.fine_needs_repeat(t)
    assert dec._L.log == [("box", "f16x3"), ("band", "f16x3")] and calls == []
    c = dec.certificate()
    assert c["audited_sweeps"] == 2 and c["shell_picks"] == 1200 and c["shell_population_max"] == 700
    assert abs(c["min_margin_tau_over_estimate"] - 10.0) < 1e-3 and abs(c["min_tau_over_sigma"] - 1 / 0.03) < 0.1
    # a fine pass whose caller wants VOLUMES is an ordinary sweep whatever the mode
    _, _, t = dec.fine_begin(*ARGS)
    assert t["kind"] == "exact" and dec._L.log[-1] == ("grid", "f16x3")


@pytest.mark.parametrize("stale", [False, True])
@pytest.mark.parametrize("reason", sorted(REASONS) + ["listed"])
def test_coarse_pass_refusals(machine, reason, stale):
    """A refused box-only sweep is answered by an ordinary sweep IN coarse_finish (the zoom cube needs boxes now); what the next
    coarse pass launches depends on why."""
    dec, calls = machine
    tau = dec._box_tau
    r = good_record(tau)
    (REASONS[reason][0] if reason != "listed" else _listed_box)(r, tau)
    dec._L.script = [r, good_record(tau), good_record(tau)]
    t = dec.coarse_begin(*ARGS)
    assert t["kind"] == "box"
    if stale:
        dec._recalibrations += 1                     # the activation scales were re-calibrated while the sweep was in flight
    b = dec.coarse_finish(t)
    assert b[:6].tolist() == [3, 4, 5, 20, 21, 22]                       # boxes come out whatever path ran
    assert dec._L.log[:2] == [("box", "f16x3"), ("grid", "f16x3")]       # ... from an ordinary sweep
    assert dec.box_stats["fallback"] == 1
    for_error = reason != "listed" and REASONS[reason][1]
    if stale:
        # not judged at all: nothing is learnt from a sweep under scales that no longer exist; the new epoch needs its own allowance
        assert dec._box_failures == 0 and dec.cert["refusals_for_error"] == 0 and calls == ["calibrate"]
    elif reason == "range":
        assert calls == ["recover", "calibrate"]     # new scales, then the allowance for them, on this very lattice
        assert dec._box_failures == 0
    elif for_error:
        assert calls == ["calibrate"] and dec.cert["refusals_for_error"] == 1 and dec._box_failures == 1
    else:
        assert calls == [] and dec._box_failures == 1 and dec._allowance_valid(N)
    if reason in ("near", "near_bit") and not stale:
        pass                                         # (the box sweep's near-level overflow is answered by the ordinary repeat itself)
    # the next coarse pass is a box-only sweep again (one refusal does not switch the mode off)
    dec._L.log.clear()
    t = dec.coarse_begin(*ARGS)
    assert t["kind"] == "box", (reason, stale)
    dec.coarse_finish(t)
    assert dec._box_failures == 0


@pytest.mark.parametrize("stale", [False, True])
@pytest.mark.parametrize("reason", sorted(REASONS) + ["listed"])
def test_fine_pass_refusals(machine, reason, stale):
    """A refused narrow-band sweep makes fine_needs_repeat return True; the repeat is an ordinary sweep - on the fp32 chain, once,
    when the near-level list overflowed - and the pass after that is a band sweep again unless the allowance was voided."""
    dec, calls = machine
    tau = dec._box_tau
    r = good_record(tau)
    (REASONS[reason][0] if reason != "listed" else _listed_band)(r, tau)
    dec._L.script = [r, good_record(tau), good_record(tau)]
    _, _, t = dec.fine_begin(*ARGS, mc_only=True)
    assert t["kind"] == "band"
    if stale:
        dec._recalibrations += 1
    assert dec.fine_needs_repeat(t)
    assert dec.band_stats["fallback"] == 1
    dec._L.log.clear()
    _, _, t2 = dec.fine_begin(*ARGS, mc_only=True)                        # the repeat
    assert t2["kind"] == "exact"
    near = reason in ("near", "near_bit")           # (a property of the sample, honoured for a stale ticket too)
    for_error = reason != "listed" and REASONS[reason][1]
    # a repeat whose zoom-lattice comparison is void (refused for its error, or a new scale epoch) measures it again on this very lattice
    remeasured = (stale or reason == "range" or for_error) and not near
    assert dec._L.log == [("grid", "f32" if near else "f16x3")] + ([("plain", "f16x3")] if remeasured else []), (reason, stale, dec._L.log)
    assert dec.math == "f16x3" and not dec._force_f32_once               # ... for that one sweep only
    assert not dec.fine_needs_repeat(t2)
    assert (calls[-1:] == ["calibrate_fine"]) == remeasured
    if stale:
        assert dec._band_failures == 0 and [c for c in calls if c != "calibrate_fine"] == []
    elif reason == "range":
        assert [c for c in calls if c != "calibrate_fine"] == ["recover"] and dec._band_failures == 0
    elif near:
        assert dec._band_failures == 0               # a capacity of the fp32 refinement, not a failure of the band sweep
    else:
        assert dec._band_failures == 1
    # the pass after the repeat
    dec._L.log.clear()
    _, _, t3 = dec.fine_begin(*ARGS, mc_only=True)
    if stale or reason == "range" or (for_error and not stale):
        # no valid allowance (new scale epoch, or voided by the refusal): the zoom lattice was measured again by the repeat, but the
        # passes stay ordinary until the coarse lattice has been compared as a whole again too - by the next coarse pass
        assert t3["kind"] == "exact" and not dec._allowance_valid(N)
        assert (near and not dec._fine_valid(N)) or ("compare" not in t3 and dec._fine_valid(N))
        assert not dec.fine_needs_repeat(t3)
        t = dec.coarse_begin(*ARGS)
        assert t["kind"] == "exact"
        dec.coarse_finish(t)
        assert calls[-1] == "calibrate" and dec._allowance_valid(N)
        _, _, t3 = dec.fine_begin(*ARGS, mc_only=True)
    assert t3["kind"] == "band" and not dec.fine_needs_repeat(t3)
    assert dec._band_failures == 0


def test_three_refusals_in_a_row_switch_the_mode_off(machine):
    dec, calls = machine
    tau = dec._box_tau
    bad = good_record(tau)
    _listed_band(bad, tau)
    dec._L.script = [bad.copy(), bad.copy(), bad.copy()]
    for k in range(3):
        _, _, t = dec.fine_begin(*ARGS, mc_only=True)
        assert t["kind"] == "band" and dec.fine_needs_repeat(t)
        _, _, t = dec.fine_begin(*ARGS, mc_only=True)
        assert t["kind"] == "exact" and not dec.fine_needs_repeat(t)
    assert dec.fine_mode == "exact" and dec._band_failures == 3
    _, _, t = dec.fine_begin(*ARGS, mc_only=True)
    assert t["kind"] == "exact"
    # an accepted sweep in between resets the count: refusals must be CONSECUTIVE
    dec.fine_mode, dec._band_failures = "band", 0
    dec._L.script = [bad.copy(), bad.copy(), good_record(tau), bad.copy()]
    for k, refused in enumerate((True, True, False, True)):
        _, _, t = dec.fine_begin(*ARGS, mc_only=True)
        assert t["kind"] == "band" and dec.fine_needs_repeat(t) == refused
        if refused:
            _, _, t = dec.fine_begin(*ARGS, mc_only=True)
            assert not dec.fine_needs_repeat(t)
    assert dec.fine_mode == "band" and dec._band_failures == 1
    # the coarse mode the same way
    badc = good_record(tau)
    _listed_box(badc, tau)
    dec._L.script = [badc.copy(), badc.copy(), badc.copy()]
    for k in range(3):
        t = dec.coarse_begin(*ARGS)
        assert t["kind"] == "box"
        dec.coarse_finish(t)
    assert dec.coarse_mode == "exact"
    assert dec.coarse_begin(*ARGS)["kind"] == "exact"


def test_ordinary_sweep_reports(machine):
    """An ORDINARY split-half sweep answers for itself through words 7 / 15 of its box record: range violations -> new scales and
    a repeat; bit 30 (near-level list overflowed) -> one repeat on the fp32 chain."""
    dec, calls = machine
    dec.coarse_mode = dec.fine_mode = "exact"
    real = dec._L.asdf_decode_grid
    state = {"n": 0}

    def overflow_once(h, n, org, vs, mode, hand, obj, bbox, stream):
        rc = real(h, n, org, vs, mode, hand, obj, bbox, stream)
        if state["n"] == 0 and bbox:
            (ctypes.c_int32 * 16).from_address(bbox)[7] = hd.NEAR_OVERFLOW_BIT
        state["n"] += 1
        return rc

    dec._L.asdf_decode_grid = overflow_once
    t = dec.coarse_begin(*ARGS)
    b = dec.coarse_finish(t)
    assert dec._L.log == [("grid", "f16x3"), ("grid", "f32")] and dec.math == "f16x3"
    assert int(b[7]) == 0 and int(b[15]) == 0
    # the same in the fine pass: fine_needs_repeat says so, the repeat runs on the fp32 chain
    dec._L.log.clear()
    state["n"] = 0
    _, _, t = dec.fine_begin(*ARGS, mc_only=True)
    assert t["kind"] == "exact" and dec.fine_needs_repeat(t)
    _, _, t = dec.fine_begin(*ARGS, mc_only=True)
    assert not dec.fine_needs_repeat(t)
    assert dec._L.log == [("grid", "f16x3"), ("grid", "f32")] and dec.math == "f16x3"
    # range violations: re-calibrated and repeated under the new scales
    dec._L.log.clear()
    state["n"] = 1

    def range_once(h, n, org, vs, mode, hand, obj, bbox, stream):
        rc = real(h, n, org, vs, mode, hand, obj, bbox, stream)
        if state["n"] == 1 and bbox:
            (ctypes.c_int32 * 16).from_address(bbox)[15] = 7
        state["n"] += 1
        return rc

    dec._L.asdf_decode_grid = range_once
    t = dec.coarse_begin(*ARGS)
    dec.coarse_finish(t)
    assert calls[-1] == "recover" and dec._L.log == [("grid", "f16x3"), ("grid", "f16x3")]


def test_periodic_recalibration_and_lattice_growth(machine, monkeypatch):
    dec, calls = machine
    monkeypatch.setattr(hd, "RECAL_EVERY", 3)
    tau = dec._box_tau
    dec._L.script = [good_record(tau) for _ in range(8)]
    kinds = []
    for _ in range(8):
        t = dec.coarse_begin(*ARGS)
        kinds.append(t["kind"])
        dec.coarse_finish(t)
    # coarse passes only: the zoom lattice's turn (the second periodic comparison) is never taken by a fine pass, so the coarse
    # lattice takes it once the comparison is a whole period overdue
    assert kinds == ["box", "box", "box", "exact", "box", "box", "box", "box"] and calls == ["calibrate"] and dec._next_recal == "fine"
    dec._L.script = [good_record(tau) for _ in range(3)]
    kinds = []
    for _ in range(3):
        t = dec.coarse_begin(*ARGS)
        kinds.append(t["kind"])
        dec.coarse_finish(t)
    assert kinds == ["box", "box", "exact"] and calls == ["calibrate", "calibrate"]
    # a lattice more than 8 x the calibrated one needs its own whole-lattice comparison
    big = (2 * N + 8,) + ARGS[1:]
    t = dec.coarse_begin(*big)
    assert t["kind"] == "exact"
    dec.coarse_finish(t)
    assert calls[-1] == "calibrate" and dec._allowance_valid(2 * N + 8) and dec._allowance_valid(N)


def test_periodic_comparison_alternates_between_the_coarse_and_the_zoom_lattice(machine, monkeypatch):
    """every RECAL_EVERY samples ONE whole-lattice comparison, in turn of the coarse lattice (the coarse pass
    runs as an ordinary sweep + a plain one-plane sweep) and of the zoom lattice (the FINE pass does) - the lattice whose signs
    marching cubes consumes (utils/mesh.py:82-121).  The sample's meshes come from ordinary sweeps in either case."""
    dec, calls = machine
    monkeypatch.setattr(hd, "RECAL_EVERY", 3)
    tau = dec._box_tau
    dec._L.script = [good_record(tau) for _ in range(40)]
    seq = []
    for _ in range(13):
        t = dec.coarse_begin(*ARGS)
        dec.coarse_finish(t)
        _, _, f = dec.fine_begin(*ARGS, mc_only=True)
        assert not dec.fine_needs_repeat(f)
        seq.append((t["kind"], f["kind"] + ("+compare" if f.get("periodic") else "")))
    normal = ("box", "band")
    assert seq == [normal] * 3 + [("exact", "band")] + [normal] * 3 + [("box", "exact+compare")] + [normal] * 3 + [("exact", "band")] + [normal]
    assert calls == ["calibrate", "calibrate_fine", "calibrate"]
    assert dec.box_stats["fallback"] == 0 and dec.band_stats["fallback"] == 0 and dec.events["repeated_sweeps"] == 0


def test_tail_ladder_covers_small_audits(machine):
    """an audit below 64 picks per head (a uniform half below 32) used to find no ladder entry and fell back to the
    stale scalar 1.0; the ladder now starts at 1 and a lattice below it takes its smallest entry."""
    dec, calls = machine
    dec._tail_by_n = {1 << k: 3.0 - 0.1 * k for k in range(0, 18)}
    dec._tail = 1.0
    dec.audit_voxels = 16                            # 8 uniform picks
    assert abs(dec._tail_for(N ** 3) - (3.0 - 0.3)) < 1e-12
    dec.audit_voxels = 1                             # one pick in all: the uniform half holds it
    assert abs(dec._tail_for(N ** 3) - 3.0) < 1e-12
    dec._tail_by_n = {32: 1.5, 64: 1.4}
    dec.audit_voxels = 16
    assert dec._tail_for(N ** 3) == 1.5              # below the ladder: its smallest entry
    assert len(hd.HipSdfDecoder.LADDER) == 18 and hd.HipSdfDecoder.LADDER[0] == 1


# ---- round 5: both passes of a sample enqueued in one go (two_pass_begin), judged afterwards -----------------------------------------------
def test_two_pass_begin_enqueues_both_passes_and_is_judged_afterwards(machine):
    dec, calls = machine
    tau = dec._box_tau
    assert dec.can_speculate(N)
    dec._L.script = [good_record(tau), good_record(tau)]
    since = dec._coarse_since_cal
    t = dec.two_pass_begin(N, ARGS[2])
    assert dec._L.log == [("box", "f16x3"), ("zoom", "f16x3"), ("band_dev", "f16x3")] and dec._coarse_since_cal == since + 1
    assert t["coarse"]["kind"] == "box" and t["fine"]["kind"] == "band" and t["coarse"]["tau"] == t["fine"]["tau"] == tau
    ok, b, _ = dec.coarse_judge(t["coarse"])
    assert ok and b[:6].tolist() == [3, 4, 5, 20, 21, 22] and dec._L.log[3:] == []          # judging launches nothing
    origin, nvs = dec.lattice_of(t)
    assert origin == [-0.5, -0.25, -0.125] and float(nvs) == 0.0078125 and nvs.dtype == torch.float32
    assert not dec.fine_needs_repeat(t["fine"])
    assert dec.box_stats["box"] == 1 and dec.band_stats["band"] == 1 and dec.events["samples_in_one_go"] == 1
    rep = dec.sweep_report()
    assert rep["samples_enqueued_in_one_go"] == 1 and rep["sweeps_refused"] == 0 and rep["sweeps_audited"] == 2


def test_a_refused_speculative_coarse_sweep_is_repeated_step_by_step(machine):
    dec, calls = machine
    tau = dec._box_tau
    bad = good_record(tau)
    _err(bad, tau)                                        # refused for its error: the allowance is void
    dec._L.script = [bad, good_record(tau)]
    t = dec.two_pass_begin(N, ARGS[2])
    judged = dec.coarse_judge(t["coarse"])
    assert judged[0] is False and judged[1] is None and dec.box_stats["fallback"] == 1 and dec.events["repeated_sweeps"] == 1
    assert dec._L.log == [("box", "f16x3"), ("zoom", "f16x3"), ("band_dev", "f16x3")]          # still nothing launched by the verdict
    b = dec.coarse_finish(t["coarse"], judged=judged)     # the caller has re-bound the sample: the ordinary repeat + a new comparison
    assert dec._L.log[3:] == [("grid", "f16x3")] and calls == ["calibrate"] and b[:6].tolist() == [3, 4, 5, 20, 21, 22]
    # the speculative fine sweep ran on a lattice nobody vouches for: its ticket is dropped unjudged, the fine pass goes step by step -
    # and, the refusal having voided both comparisons, measures the zoom lattice again
    assert not dec.can_speculate(N)
    _, _, f = dec.fine_begin(*ARGS, mc_only=True)
    assert f["kind"] == "exact" and "compare" in f and not dec.fine_needs_repeat(f) and calls[-1] == "calibrate_fine"
    assert dec.can_speculate(N) and dec.band_stats["band"] == 0


def test_no_speculation_while_a_comparison_is_due_or_a_mode_is_off(machine, monkeypatch):
    dec, calls = machine
    monkeypatch.setattr(hd, "RECAL_EVERY", 3)
    tau = dec._box_tau
    dec._L.script = [good_record(tau) for _ in range(12)]
    took = []
    for _ in range(5):
        t = dec.two_pass_begin(N, ARGS[2])
        took.append(t is not None)
        if t is None:                                     # the step-by-step path takes the sample (and runs the comparison that is due)
            c = dec.coarse_begin(*ARGS)
            dec.coarse_finish(c)
            _, _, f = dec.fine_begin(*ARGS, mc_only=True)
            assert not dec.fine_needs_repeat(f)
        else:
            assert dec.coarse_judge(t["coarse"])[0] and not dec.fine_needs_repeat(t["fine"])
    assert took == [True, True, True, False, True] and calls == ["calibrate"]      # (RECAL_EVERY = 3: the fourth sample runs the coarse comparison)
    for attr, val in (("coarse_mode", "exact"), ("fine_mode", "exact"), ("_band_skip", True), ("_force_f32_once", True)):
        keep = getattr(dec, attr)
        setattr(dec, attr, val)
        assert not dec.can_speculate(N) and dec.two_pass_begin(N, ARGS[2]) is None, attr
        setattr(dec, attr, keep)
    assert dec.can_speculate(N) and not dec.can_speculate(4 * N)          # a lattice 64 x the compared ones needs its own comparisons


def test_a_speculative_ticket_launched_under_old_scales_is_not_judged(machine):
    dec, calls = machine
    tau = dec._box_tau
    dec._L.script = [good_record(tau), good_record(tau)]
    t = dec.two_pass_begin(N, ARGS[2])
    dec._recalibrations += 1                              # the activation scales were re-calibrated while the sample was in flight
    ok, b, cal = dec.coarse_judge(t["coarse"])
    assert not ok and cal and dec._box_failures == 0 and dec.cert["refusals_for_error"] == 0
    assert dec.fine_needs_repeat(t["fine"]) and dec._band_failures == 0


# ---- round 6: ordinary sweeps are the default, the audited one-plane sweeps an opt-in ---------------------
def _bare_decoder():
    dec = hd.HipSdfDecoder.__new__(hd.HipSdfDecoder)
    dec.combined, dec.nerf_features, dec.point_feat_size = False, False, 3
    dec.device, dec._h = torch.device("cpu"), None
    dec._L = FakeLib(dec)
    dec._stream = lambda: None
    dec._init_sweep_state()
    dec._calibrated = True
    dec._status = lambda clear: np.zeros(16, dtype=np.int32)
    return dec


def test_default_modes_are_ordinary_sweeps_and_fast_is_an_opt_in(monkeypatch):
    monkeypatch.setattr(torch.cuda, "device", lambda d: contextlib.nullcontext())
    for k in ("ASDF_MATH", "ASDF_COARSE", "ASDF_FINE", "ASDF_FAST"):
        monkeypatch.delenv(k, raising=False)
    assert (hd.DEFAULT_COARSE, hd.DEFAULT_FINE) == ("exact", "exact")
    dec = _bare_decoder()
    assert (dec.coarse_mode, dec.fine_mode, dec.math) == ("exact", "exact", "f16x3")
    assert not dec._box_usable() and not dec._band_usable() and not dec.can_speculate(N)
    # a mesh-producing sample under the defaults: two ordinary sweeps, no one-plane launch, no whole-lattice comparison
    b = dec.coarse_finish(dec.coarse_begin(*ARGS))
    vh, vo, t = dec.fine_begin(N, [-0.5] * 3, 0.01, mc_only=True)
    assert not dec.fine_needs_repeat(t)
    assert [k for k, _ in dec._L.log] == ["grid", "grid"] and dec.cert["calibrations"] == 0 and dec.cert["fine_calibrations"] == 0
    assert dec.box_stats["exact"] == 1 and dec.band_stats["exact"] == 1 and dec.box_stats["box"] == 0
    dec.set_fast(True)
    assert (dec.coarse_mode, dec.fine_mode) == ("box", "band") and dec._box_usable() and dec._band_usable()
    dec.set_fast(False)
    assert (dec.coarse_mode, dec.fine_mode) == ("exact", "exact")
    for value, want in (("1", ("box", "band")), ("0", ("exact", "exact")), ("", ("exact", "exact")), ("yes", ("box", "band"))):
        monkeypatch.setenv("ASDF_FAST", value)
        d2 = _bare_decoder()
        assert (d2.coarse_mode, d2.fine_mode) == want, value
    monkeypatch.setenv("ASDF_FAST", "1")
    monkeypatch.setenv("ASDF_FINE", "exact")                        # the pass-by-pass switches override the pair
    d3 = _bare_decoder()
    assert (d3.coarse_mode, d3.fine_mode) == ("box", "exact")


def test_one_zoom_lattice_comparison_in_flight(machine):
    """with the software pipeline the next samples' fine_begin run before the first whole-zoom-lattice comparison has been
    judged - each used to enqueue its own plain one-plane sweep (4 x N^3 fp32 volumes apiece) and calibrate again."""
    dec, calls = machine
    dec._fine_epoch = -1                                             # (what a refusal for error does: the zoom lattice is measured again)
    tickets = [dec.fine_begin(N, [-0.5] * 3, 0.01, mc_only=True)[2] for _ in range(3)]      # three samples' fine passes in flight
    assert [k for k, _ in dec._L.log].count("plain") == 1 and "compare" in tickets[0] and "compare" not in tickets[1] and "compare" not in tickets[2]
    for t in tickets:
        assert not dec.fine_needs_repeat(t)
    assert calls.count("calibrate_fine") == 1 and not dec._fine_compare_in_flight and dec._fine_valid(N)
    dec._L.script = [good_record(dec._box_tau)]
    assert dec.fine_begin(N, [-0.5] * 3, 0.01, mc_only=True)[2]["kind"] == "band"      # the certificate is there: band sweeps from now on


def test_cluster_fault_report_switches_the_form_off_once(machine):
    """Bit 29 of word 7 of a bbox record (or word 27 of a one-plane record): a member of the short-list kernel's cluster form did not
    arrive in time.  The sweep is complete (the tile form evaluated the list); the host switches the form off, once, and says so."""
    dec, calls = machine
    off = []
    dec._L.asdf_decoder_set_cluster_list = lambda h, n: off.append(n) or 0
    dec._h = 1
    r = np.zeros(16, dtype=np.int32)
    r[7] = hd.CLUSTER_FAULT_BIT
    assert dec._range_words(r) == (0, False)                         # not a range violation, not a list overflow
    assert dec.fall_back_if_overflowed(r) is False and off == [0] and dec._cluster_off
    assert dec.fall_back_if_overflowed(r) is False and off == [0]    # once
    assert any("cluster form" in m for m in dec.events["modes_switched_off"])
    dec._h = None


def test_ordinary_sweeps_are_enqueued_in_one_go_too(monkeypatch):
    """Round 6: the product's default - ordinary sweeps in both passes - takes the one-go form as well: coarse sweep, zoom cube on the
    device, fine sweep reading its lattice from those words (asdf_decode_grid_dev).  Nothing is read in between; the two bbox records are
    judged when the sample is finished."""
    monkeypatch.setattr(torch.cuda, "device", lambda d: contextlib.nullcontext())
    for k in ("ASDF_MATH", "ASDF_COARSE", "ASDF_FINE", "ASDF_FAST"):
        monkeypatch.delenv(k, raising=False)
    dec = _bare_decoder()
    assert not dec.can_speculate(N) and dec.can_speculate_ordinary()
    t = dec.two_pass_begin(N, 2.0 / (N - 1))
    assert [k for k, _ in dec._L.log] == ["grid", "zoom", "grid_dev"] and dec.events["samples_in_one_go"] == 1
    assert t["coarse"]["kind"] == "exact" and t["fine"]["kind"] == "exact" and t["fine"]["rec"] is not None
    ok, b, _ = dec.coarse_judge(t["coarse"])
    assert ok and b[:7].tolist() == good_record(1.0)[:7].tolist() and dec.box_stats["exact"] == 1
    assert dec.lattice_of(t)[0] == [-0.5, -0.25, -0.125] and float(dec.lattice_of(t)[1]) == 0.0078125
    assert not dec.fine_needs_repeat(t["fine"]) and dec.band_stats["exact"] == 1
    # an uncalibrated MLP, a sweep ordered onto the fp32 chain, or a one-plane mode switched on: step by step (resp. the audited form)
    dec._calibrated = False
    assert not dec.can_speculate_ordinary() and dec.two_pass_begin(N, 2.0 / (N - 1)) is None
    dec._calibrated = True
    dec._force_f32_once = True
    assert not dec.can_speculate_ordinary()
    dec._force_f32_once = False
    dec.fine_mode = "band"
    assert not dec.can_speculate_ordinary()
    dec.fine_mode = "exact"
    dec.set_math("f32")
    dec._calibrated = False                                          # (the fp32 chain has no activation scales to calibrate)
    t = dec.two_pass_begin(N, 2.0 / (N - 1))
    assert t is not None and t["fine"]["rec"] is None and not dec.fine_needs_repeat(t["fine"])


def test_a_range_violation_in_a_speculative_ordinary_coarse_pass_is_recovered_once(monkeypatch):
    """The coarse record of a sample enqueued in one go reports activations outside the fp16 range: coarse_judge books the recovery
    (new activation scales - once), the caller repeats the pass through coarse_finish(ticket, judged), which launches it again and does
    NOT recover a second time for the same record."""
    monkeypatch.setattr(torch.cuda, "device", lambda d: contextlib.nullcontext())
    for k in ("ASDF_MATH", "ASDF_COARSE", "ASDF_FINE", "ASDF_FAST"):
        monkeypatch.delenv(k, raising=False)
    dec = _bare_decoder()
    recovered = []

    def fake_recover(bad, status=None):
        recovered.append(bad)
        dec._recalibrations += 1

    dec._recover = fake_recover
    bad = good_record(1.0)[:16].copy()
    bad[7] = 12
    real = dec._L.asdf_decode_grid
    first = []

    def grid(h, n, org, vs, mode, hand, obj, bbox, stream):
        rc = real(h, n, org, vs, mode, hand, obj, bbox, stream)
        if not first and bbox:
            first.append(1)
            dec._L._write(bbox, bad)
        return rc

    dec._L.asdf_decode_grid = grid
    t = dec.two_pass_begin(N, 2.0 / (N - 1))
    judged = dec.coarse_judge(t["coarse"])
    assert judged[0] is False and recovered == [12] and dec.events["repeated_sweeps"] == 1
    dec._L.log.clear()
    b = dec.coarse_finish(t["coarse"], judged=judged)
    assert [k for k, _ in dec._L.log] == ["grid"] and recovered == [12] and b[7] == 0 and dec.box_stats["exact"] == 1
    # the fine pass of that sample was launched under the old scales: its record is judged as stale and it is repeated
    assert dec.fine_needs_repeat(dict(t["fine"], rec=torch.from_numpy(bad.copy()), host=None)) is True
