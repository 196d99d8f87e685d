"""Order of the launches pipelined_two_pass issues (alignsdf_amd/reconstruct.py), checked on the CPU with a recording stand-in for
the decoder: per sample  pass 1 -> [box read-back] -> pass 2 -> marching cubes (utils/mesh.py:27-121, 351-369), with pass 1 of the
next sample queued ahead of this sample's marching cubes, pass 2 of the next sample ahead of this sample's post-processing (label
pass), the eval hook between pass 2 of k+1 and pass 1 of k+2 - and a refused fine sweep repeated before marching cubes reads it."""
import numpy as np
import pytest
import torch

from alignsdf_amd import reconstruct as rc


class RecordingDecoder:
    device = torch.device("cpu")

    def __init__(self, log, refuse_fine_of=()):
        self.log, self.bound, self.refuse = log, None, set(refuse_fine_of)
        self.fine_count = {}

    def coarse_begin(self, N, origin, voxel, mode, hand=True, obj=True):
        self.log.append(("pass1", self.bound))
        return {"key": self.bound}

    def coarse_finish(self, ticket):
        self.log.append(("boxes", ticket["key"]))
        b = np.zeros(16, dtype=np.int64)
        b[0:3], b[3:6], b[6] = (3, 4, 5), (9, 11, 12), 100
        b[8:11], b[11:14], b[14] = (2, 6, 5), (8, 10, 13), 50
        return b

    def fine_begin(self, N, origin, voxel, mode, hand=True, obj=True, mc_only=False):
        assert mc_only
        k = self.bound
        self.fine_count[k] = self.fine_count.get(k, 0) + 1
        self.log.append(("pass2", k))
        return ("vol_hand", k), ("vol_obj", k), {"key": k, "nth": self.fine_count[k]}

    def fine_needs_repeat(self, ticket):
        return ticket is not None and ticket["key"] in self.refuse and ticket["nth"] == 1

    def classify_points(self, pts, want_sdf=False):
        self.log.append(("labels", self.bound))
        return None, None, None, torch.zeros(pts.shape[0], dtype=torch.int64)


@pytest.fixture()
def harness(monkeypatch):
    log = []
    state = {}

    def install(refuse=(), cls=RecordingDecoder, **kw):
        dec = cls(log, refuse, **kw)
        state["dec"] = dec
        import alignsdf_amd.marching_cubes as mc
        import alignsdf_amd.utils.utils as uu
        monkeypatch.setattr(uu, "decoder_for", lambda decoder, specs, mano: dec)
        monkeypatch.setattr(uu, "bind_sample", lambda hip, specs, latent, mano, obj: setattr(hip, "bound", int(latent)))
        monkeypatch.setattr(mc, "marching_cubes_begin", lambda vol, level, slot, capacity=None: (
            log.append(("mc_count", vol[1], vol[0])), log.append(("mc_emit_bounded", vol[1], vol[0], capacity)) if capacity else None, vol)[2])
        monkeypatch.setattr(mc, "marching_cubes_finish", lambda t: (log.append(("mc_emit", t[1], t[0])),
                                                                   (torch.zeros(6, 3), torch.zeros(8, 3, dtype=torch.int32)))[1])
        return dec
    return log, install


def samples(n):
    return [(k, k, None, None) for k in range(n)]          # (key, latent stand-in, mano, obj)


SPECS = {"HandBranch": True, "ObjectBranch": True}


def first(log, event):
    return log.index(event)


def test_order_of_the_launches(harness):
    log, install = harness
    install()
    hooks = []
    out = list(rc.pipelined_two_pass(object(), SPECS, samples(4), 16, label_out=True,
                                     midpoint=lambda key, r: (hooks.append(key), log.append(("hook", key)))))
    assert [k for k, _ in out] == [0, 1, 2, 3] and hooks == [0, 1, 2, 3]
    for k in range(4):
        # the reference's order within a sample
        assert first(log, ("pass1", k)) < first(log, ("boxes", k)) < first(log, ("pass2", k)) < first(log, ("mc_count", k, "vol_hand"))
        assert first(log, ("mc_count", k, "vol_obj")) < first(log, ("mc_emit", k, "vol_hand")) < first(log, ("mc_emit", k, "vol_obj"))
        assert first(log, ("mc_emit", k, "vol_obj")) < first(log, ("labels", k)) < first(log, ("hook", k))
    for k in range(3):
        # pass 1 of k+1 ahead of the marching cubes of k; pass 2 of k+1 behind its emits and AHEAD of k's post-processing
        assert first(log, ("pass1", k + 1)) < first(log, ("mc_count", k, "vol_hand"))
        assert first(log, ("mc_emit", k, "vol_obj")) < first(log, ("pass2", k + 1)) < first(log, ("labels", k))
    for k in range(2):
        # the hook of k sits between pass 2 of k+1 and pass 1 of k+2
        assert first(log, ("pass2", k + 1)) < first(log, ("hook", k)) < first(log, ("pass1", k + 2))
    # every result carries the zoom cube of its boxes (union of the two branches' boxes: utils/mesh.py:239-254)
    from alignsdf_amd.utils.mesh import zoom_cube_from_bboxes
    nvs, norg = zoom_cube_from_bboxes([((3, 4, 5), (9, 11, 12), 100), ((2, 6, 5), (8, 10, 13), 50)], 16, 2.0 / 15)
    for _, r in out:
        assert float(r["voxel_size"]) == float(nvs) and r["origin"] == norg.tolist()
        assert r["V_hand"] == 6 and r["F_obj"] == 8 and r["labels_hand"].shape[0] == 6


def test_refused_fine_sweep_is_repeated_before_marching_cubes(harness):
    log, install = harness
    dec = install(refuse=(1,))
    out = list(rc.pipelined_two_pass(object(), SPECS, samples(3), 16))
    assert [k for k, _ in out] == [0, 1, 2]
    assert dec.fine_count == {0: 1, 1: 2, 2: 1}
    second = [i for i, e in enumerate(log) if e == ("pass2", 1)][1]
    # the count phases are queued ahead of the guard record's read-back (round 4: one host wait for record + sizes); a refused sweep
    # is repeated and COUNTED AGAIN, and the emits read the repeat's volumes
    counts = [i for i, e in enumerate(log) if e == ("mc_count", 1, "vol_hand")]
    assert len(counts) == 2 and counts[0] < second < counts[1] < first(log, ("mc_emit", 1, "vol_hand"))
    assert len([e for e in log if e == ("mc_count", 0, "vol_hand")]) == 1
    # the decoder was re-bound to sample 1 for the repeat, and to sample 2 again before ITS pass 2
    assert first(log, ("pass2", 2)) > second and log[first(log, ("pass2", 2))] == ("pass2", 2)


def test_single_sample_and_empty_stream(harness):
    log, install = harness
    install()
    assert list(rc.pipelined_two_pass(object(), SPECS, [], 16)) == []
    out = list(rc.pipelined_two_pass(object(), SPECS, samples(1), 16, midpoint=lambda key, r: log.append(("hook", key))))
    assert [k for k, _ in out] == [0]
    assert [e[0] for e in log] == ["pass1", "boxes", "pass2", "mc_count", "mc_count", "mc_emit", "mc_emit", "hook"]


# ---- round 5: a sample enqueued in one go (HipSdfDecoder.two_pass_begin) ---------------------------------------------------------------
class SpeculatingDecoder(RecordingDecoder):
    """two_pass_begin enqueues pass 1 -> device zoom cube -> pass 2; coarse_judge reads the coarse record afterwards."""

    def __init__(self, log, refuse_fine_of=(), stepwise=(), refuse_coarse_of=()):
        super().__init__(log, refuse_fine_of)
        self.stepwise, self.refuse_coarse = set(stepwise), set(refuse_coarse_of)

    def two_pass_begin(self, N, voxel, mode, hand=True, obj=True):
        k = self.bound
        if k in self.stepwise:
            return None
        self.fine_count[k] = self.fine_count.get(k, 0) + 1
        self.log.extend([("pass1", k), ("zoom_dev", k), ("pass2", k)])
        return {"coarse": {"key": k}, "fine": {"key": k, "nth": self.fine_count[k]}, "vol_hand": ("vol_hand", k), "vol_obj": ("vol_obj", k)}

    def coarse_judge(self, ticket):
        k = ticket["key"]
        self.log.append(("judge", k))
        if k in self.refuse_coarse:
            return False, None, True
        b = np.zeros(16, dtype=np.int64)
        b[0:3], b[3:6], b[6] = (3, 4, 5), (9, 11, 12), 100
        b[8:11], b[11:14], b[14] = (2, 6, 5), (8, 10, 13), 50
        return True, b, False

    def coarse_finish(self, ticket, judged=None):
        if judged is not None:
            assert not judged[0] and self.bound == ticket["key"]       # repeated while the decoder is bound to THAT sample
            self.log.append(("pass1_again", ticket["key"]))
        return super().coarse_finish(ticket)

    @staticmethod
    def lattice_of(ticket):
        from alignsdf_amd.utils.mesh import zoom_cube_from_bboxes
        nvs, norg = zoom_cube_from_bboxes([((3, 4, 5), (9, 11, 12), 100), ((2, 6, 5), (8, 10, 13), 50)], 16, 2.0 / 15)
        return norg.tolist(), nvs


def test_a_sample_is_enqueued_in_one_go_and_judged_afterwards(harness):
    """pass 1, the zoom cube (on the device), pass 2 and the marching-cubes counts of sample k + 1 are all enqueued
    BEFORE the host reads anything of sample k + 1 - and before it finishes sample k; from the second surface on the emit is enqueued
    with the counts (capacity = 1.5 x the largest surface so far).  The results are those of the step-by-step path."""
    log, install = harness
    install(cls=SpeculatingDecoder, stepwise=(0,))
    out = list(rc.pipelined_two_pass(object(), SPECS, samples(4), 16, label_out=True, midpoint=lambda key, r: log.append(("hook", key))))
    assert [k for k, _ in out] == [0, 1, 2, 3]
    # sample 0 (the decoder's first: whole-lattice comparisons) goes step by step
    assert first(log, ("pass1", 0)) < first(log, ("boxes", 0)) < first(log, ("pass2", 0)) < first(log, ("mc_count", 0, "vol_hand"))
    assert ("zoom_dev", 0) not in log and ("judge", 0) not in log
    for k in (1, 2, 3):
        i = first(log, ("pass1", k))
        assert log[i:i + 3] == [("pass1", k), ("zoom_dev", k), ("pass2", k)] and ("boxes", k) not in log
        assert first(log, ("judge", k)) < first(log, ("mc_emit", k, "vol_hand")) < first(log, ("labels", k)) < first(log, ("hook", k))
    # sample 1 is queued before any surface has been seen: no sizes for its emit buffers, so its marching cubes waits for surfaces()
    # (a count phase must never be left alone across the next sample's count phase: they share the workspace)
    assert first(log, ("judge", 1)) < first(log, ("mc_count", 1, "vol_hand"))
    assert not [e for e in log if e[0] == "mc_emit_bounded" and e[1] == 1]
    for k in (2, 3):
        # ... from then on the counts AND the bounded emits of both volumes ride right behind pass 2, before anything of k is read
        i = first(log, ("pass2", k))
        cap = (int(1.5 * 6) + 4096, int(1.5 * 8) + 8192)                 # 1.5 x the largest surface so far + the floor
        assert log[i + 1:i + 5] == [("mc_count", k, "vol_hand"), ("mc_emit_bounded", k, "vol_hand", cap), ("mc_count", k, "vol_obj"),
                                    ("mc_emit_bounded", k, "vol_obj", cap)]
        assert i + 4 < first(log, ("judge", k))
    for k in (1, 2):
        # all of sample k + 1 is in the queue before the host waits for the surfaces of sample k
        assert first(log, ("mc_emit_bounded", k + 1, "vol_obj", cap)) < first(log, ("mc_emit", k, "vol_hand"))
    assert first(log, ("pass2", 1)) < first(log, ("mc_emit", 0, "vol_hand"))
    from alignsdf_amd.utils.mesh import zoom_cube_from_bboxes
    nvs, norg = zoom_cube_from_bboxes([((3, 4, 5), (9, 11, 12), 100), ((2, 6, 5), (8, 10, 13), 50)], 16, 2.0 / 15)
    for _, r in out:
        assert float(r["voxel_size"]) == float(nvs) and r["origin"] == norg.tolist() and r["V_hand"] == 6 and r["F_obj"] == 8


def test_refusals_of_a_speculative_sample(harness):
    """A refused coarse sweep: pass 1 again as an ordinary sweep while the decoder is bound to that sample, the zoom cube on the host,
    pass 2 and the counts again.  A refused fine sweep: re-bound, pass 2 and the counts again (as on the step-by-step path)."""
    log, install = harness
    dec = install(cls=SpeculatingDecoder, refuse=(3,), refuse_coarse_of=(2,))
    out = list(rc.pipelined_two_pass(object(), SPECS, samples(5), 16))
    assert [k for k, _ in out] == [0, 1, 2, 3, 4]
    j = first(log, ("judge", 2))
    assert log[j:j + 4] == [("judge", 2), ("pass1_again", 2), ("boxes", 2), ("pass2", 2)]
    assert dec.fine_count == {0: 1, 1: 1, 2: 2, 3: 2, 4: 1}
    # (samples 2 .. 4 had their marching cubes enqueued behind their speculative fine pass: counted there, and counted AGAIN on the repeat)
    counts = [i for i, e in enumerate(log) if e == ("mc_count", 2, "vol_hand")]
    assert len(counts) == 2 and counts[0] < j < counts[1] < first(log, ("mc_emit", 2, "vol_hand"))
    counts = [i for i, e in enumerate(log) if e == ("mc_count", 3, "vol_hand")]
    second = [i for i, e in enumerate(log) if e == ("pass2", 3)][1]
    assert len(counts) == 2 and counts[0] < second < counts[1] < first(log, ("mc_emit", 3, "vol_hand"))
    assert len([e for e in log if e == ("mc_count", 4, "vol_hand")]) == 1


def test_speculation_can_be_switched_off(harness, monkeypatch):
    log, install = harness
    install(cls=SpeculatingDecoder)
    monkeypatch.setenv("ASDF_SPECULATE", "0")
    out = list(rc.pipelined_two_pass(object(), SPECS, samples(3), 16))
    assert [k for k, _ in out] == [0, 1, 2] and not [e for e in log if e[0] in ("zoom_dev", "judge")]
