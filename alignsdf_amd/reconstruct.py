"""Per-GPU reconstruction driver: counterpart of the reference's reconstruct.py (reconstruct.py:33-95,
CLI :107-178) for the accelerated hot path.

The reference's per-image front end (ResNet-18 encoder, MANO branch: utils.decode_model_output,
utils/utils.py:575-625) is out of scope of this build; what it hands to the hot path - a latent code
[1,256] and, for pose-aligned models, `global_trans` / `rot_center` / `obj_trans` - comes from a
`code_source(sample_name, index) -> (latent, mano_results, obj_results)` callable instead.  Two sources
are provided: precomputed `.npz` codes on disk and the deterministic synthetic codes used by the tests
and the benchmark.  Everything after that point (two-pass grid decode, zoom cube, marching cubes, PLY)
is the HIP path.
"""
import argparse
import json
import os
import threading
import time

import numpy as np
import torch

from . import synthetic
from .networks.model import build_decoder
from .utils import mesh as mesh_utils


class CodeUploader:
    """(Round 6: the code sources below no longer upload - host-side codes stay on the host and the HIP decoder reads them from pinned
    memory, see _code_converter; this class serves the module path and callers that ask for device tensors.)
    Per-sample codes (a latent vector, a few pose matrices: a few KB) -> device WITHOUT making the host wait for the stream.  A
    plain `.to(device)` of pageable memory is a synchronous copy in stream order: with two decoder passes of the next sample queued
    (round 5: samples are enqueued in one go) the caller sat 50 ms behind them, the queue ran dry, and everything the host did next -
    K8, the ground-truth hand-over, the sampler and the ICP of eval mode - ran with the GPU idle in between (4.4 ms per sample in
    the eval-mode trace).  Here: a ring of pinned staging buffers, the copy on a side stream, and the compute stream waits for the
    copy's event on the DEVICE."""

    SLOTS = 8

    def __init__(self, device="cuda"):
        self.device = torch.device(device)
        self.stream = None
        self.ring = {}        # (shape, dtype) -> [turn, [(pinned buffer, event of the copy that last read it)]]

    def __call__(self, array):
        t = torch.from_numpy(np.ascontiguousarray(array))
        if self.device.type != "cuda":
            return t.to(self.device)
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=self.device)
        key = (tuple(t.shape), t.dtype)
        entry = self.ring.setdefault(key, [0, []])
        if len(entry[1]) < self.SLOTS:
            entry[1].append([torch.empty(t.shape, dtype=t.dtype).pin_memory(), None])
            slot = entry[1][-1]
        else:
            slot = entry[1][entry[0] % self.SLOTS]
            entry[0] += 1
            slot[1].synchronize()                          # (eight uploads ago: long done)
        slot[0].copy_(t)
        compute = torch.cuda.current_stream(self.device)
        with torch.cuda.stream(self.stream):
            # (allocated from the SIDE stream's pool: a block of the compute stream's pool may still be in use by kernels queued
            # there, and this copy does not wait for them)
            out = torch.empty(t.shape, dtype=t.dtype, device=self.device)
            out.copy_(slot[0], non_blocking=True)
            done = torch.cuda.Event()
            done.record(self.stream)
        out.record_stream(compute)
        compute.wait_event(done)
        slot[1] = done
        return out


def _code_converter(device, on_host):
    """What a code source does with an array that is on the host.  on_host=True (the default, round 6): a float32 CPU tensor - the HIP
    decoder's set_sample stages it in pinned memory and its fold reads it from there in stream order (asdf_decoder_set_sample_host): no
    copy engine, no runtime blit kernel behind a persistent sweep, and the pose matrices are consumed on the
    host anyway (hip_decoder.kinematic_affine).  The module path uploads what it needs itself (torch_decoder.TorchModuleDecoder.
    set_sample, through a CodeUploader).  on_host=False: a device tensor through CodeUploader."""
    if on_host:
        return lambda array: torch.from_numpy(np.ascontiguousarray(array, dtype=np.float32))
    return CodeUploader(device)


def synthetic_code_source(tag="nerf3", device="cuda", on_host=True):
    """Deterministic per-sample codes (64 distinct samples, cycled; the grasp family: its 16 trained scenes)."""
    up = _code_converter(device, on_host)

    def source(name, index):
        s = index % (synthetic.GRASP_SAMPLES if tag in synthetic.GRASP_TAGS else 64)
        lat, m, o = synthetic.sample_inputs(tag, s)
        lat = up(lat)
        if m is None:
            return lat, None, None
        return lat, {k: up(v) for k, v in m.items()}, {k: up(v) for k, v in o.items()}
    return source


def npz_code_source(code_dir, device="cuda", on_host=True):
    """Codes saved by an external encoder run: <code_dir>/<sample>.npz with `latent` [1,256] and optionally
    `global_trans` [1,16,4,4], `rot_center` [1,1,3], `obj_trans` [1,4,4]."""
    up = _code_converter(device, on_host)

    def source(name, index):
        z = np.load(os.path.join(code_dir, name + ".npz"))
        lat = up(np.asarray(z["latent"], dtype=np.float32))
        mano = obj = None
        if "global_trans" in z.files:
            mano = {"global_trans": up(np.asarray(z["global_trans"], dtype=np.float32)),
                    "rot_center": up(np.asarray(z["rot_center"], dtype=np.float32))}
        if "obj_trans" in z.files:
            obj = {"obj_trans": up(np.asarray(z["obj_trans"], dtype=np.float32))}
        return lat, mano, obj
    return source


def pipelined_two_pass(decoder, specs, samples, N, grid_mode="reference", host_copy=False, label_out=False, midpoint=None, report=None,
                       fast=None):
    """Software pipeline over independent samples.  `samples` yields (key, latent, mano_results, obj_results); the
    generator yields (key, result) in order, where result holds the pass-2 volumes (device), the zoom cube and the
    marching-cubes output per enabled branch (`verts_*`, `faces_*` device tensors, absent when MC found no surface).

    The pipeline exists to produce meshes: it runs the coarse pass through `coarse_begin` / `coarse_finish` and the fine pass
    through `fine_begin(..., mc_only=True)`.  By default (round 6) both are ORDINARY sweeps: every voxel of both lattices in the
    reference's arithmetic class (<= 1e-5).  `fast=True` (or ASDF_FAST=1 / `--fast`; `fast=None` leaves the decoder as it is
    configured) opts in to the audited one-plane sweeps wherever the decoder supports them - the yielded `vol_*` are then exact only
    where marching cubes reads values and must not be used as SDF volumes.

    Per sample the GPU work is  pass 1 -> [64-byte bbox readback, zoom cube on the host] -> pass 2 -> marching cubes,
    and only the bracketed step and the MC size readbacks synchronise with the host.  Pass 1 of sample k+1 is queued
    right behind pass 2 of sample k, i.e. before sample k's marching cubes and before the consumer's host work
    (D2H copy, component filter, PLY export), so the GPU never waits for the host between samples.

    host_copy=True additionally runs the largest-component filter (K8) behind each marching cubes and copies its result
    to pinned host memory on a side stream (`host_kept_verts_*`, `host_kept_faces_*` at input capacity, `host_kept_counts_*`
    = kept vertices / faces, valid after `copy_done_*`.synchronize(); with label_out also the whole surface as
    `host_verts_hand` / `host_faces_hand`): a plain `.cpu()` on the compute stream would wait behind the NEXT sample's
    queued passes.

    label_out=True runs the label pass (utils/mesh.py:137-157) over the hand mesh vertices right behind the hand's
    marching cubes (`labels_hand`, int64 device tensor; `host_labels_hand` with host_copy).  The decoder holds one
    sample's folded constants at a time, so sample k is re-bound for it and sample k+1 bound again afterwards.

    midpoint(key, result), if given, is called for sample k between queuing pass 2 of sample k+1 and pass 1 of sample
    k+2: GPU work it enqueues (the eval-mode ICP of sample k's hand mesh) lands between two decoder passes instead of
    behind both, and its host part is covered by the pass that is already running.

    report, if given (a dict), receives the evaluator that ran and a snapshot of its sweep counters (see write_sweeps_json)."""
    from .marching_cubes import marching_cubes_begin, marching_cubes_finish
    from .utils.mesh import GRID_MODES, zoom_cube_from_bboxes
    from .utils.utils import bind_sample, decoder_for
    it = iter(samples)
    cur = next(it, None)
    if cur is None:
        return
    hip = decoder_for(decoder, specs, cur[2])      # the HIP kernels, or the module on PyTorch-ROCm for variants they do not cover
    if fast is not None and hasattr(hip, "set_fast"):
        hip.set_fast(bool(fast))
    if report is not None:                         # (the caller's `sweeps.json`: which evaluator ran, and its counters at the start)
        report["evaluator"], report["snapshot"] = hip, hip.sweep_snapshot()
    hb, ob = specs.get("HandBranch", True), specs.get("ObjectBranch", True)
    mode = GRID_MODES[grid_mode]
    voxel = 2.0 / (N - 1)
    copy_stream = torch.cuda.Stream(device=hip.device) if host_copy else None

    def bind(sample):
        _, latent, mano, obj = sample
        bind_sample(hip, specs, latent, mano, obj)

    def to_host(r, key, t):
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(hip.device))
        h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(ready)
            h.copy_(t, non_blocking=True)
            done = torch.cuda.Event()
            done.record(copy_stream)
        t.record_stream(copy_stream)
        r["host_" + key] = h
        return done

    # Round 5: wherever the decoder's state allows it (HipSdfDecoder.can_speculate: both one-plane modes on, both
    # whole-lattice comparisons valid, none due) a sample is ENQUEUED IN ONE GO - coarse box sweep, zoom cube on the device, narrow-band
    # fine sweep on that lattice, marching-cubes counts and a capacity-bounded emit - and the host only reads records afterwards: no
    # host round trip between the coarse and the fine pass, none between count and emit.  A refused sweep is repeated step by step, as
    # before.  ASDF_SPECULATE=0 keeps every sample on the step-by-step path.
    speculate = os.environ.get("ASDF_SPECULATE", "1") != "0" and hasattr(hip, "two_pass_begin")
    seen = {}            # part -> (most vertices, most faces) of the surfaces so far: sizes the next sample's emit buffers

    def capacity(part):
        c = seen.get(part)
        return None if c is None else (int(1.5 * c[0]) + 4096, int(1.5 * c[1]) + 8192)

    def first_pass(sample):
        bind(sample)
        t = hip.two_pass_begin(N, voxel, mode, hand=hb, obj=ob) if speculate else None
        if t is None:
            return hip.coarse_begin(N, [-1.0, -1.0, -1.0], voxel, mode, hand=hb, obj=ob)
        # marching cubes right behind the fine pass - count AND capacity-bounded emit - once the sizes of earlier surfaces are known.
        # (Never the count alone: its emit would run when the sample is finished, after the NEXT sample's count phase has reused the
        # workspace - found as a memory fault at N = 128.  Without sizes to go by the sample's marching cubes waits for surfaces().)
        parts = [(slot, part) for slot, (part, on) in enumerate((("hand", hb), ("obj", ob))) if on]
        t["mc"] = ({part: marching_cubes_begin(t["vol_" + part], 0.0, slot, capacity=capacity(part)) for slot, part in parts}
                   if all(capacity(part) is not None for _, part in parts) else None)
        return t

    def second_pass(ticket, judged=None):
        if "coarse" in ticket and judged is None:
            # both passes are in flight already: nothing to launch, and nothing is READ here either - the records are judged when the
            # sample is finished (surfaces), so the host never waits in the middle of the previous sample's post-processing
            return {"pending": ticket}
        if "coarse" in ticket:
            b = hip.coarse_finish(ticket["coarse"], judged=judged)      # refused: an ordinary sweep now (the decoder is bound to this sample)
        else:
            # waits for pass 1 (the zoom cube is data dependent); a coarse sweep whose guards fired (fp16 range, or the error
            # check of the box-only sweep) is repeated in there - the decoder is still bound to this sample
            b = hip.coarse_finish(ticket)
        boxes = ([(b[0:3], b[3:6], int(b[6]))] if hb else []) + ([(b[8:11], b[11:14], int(b[14]))] if ob else [])
        nvs, norg = zoom_cube_from_bboxes(boxes, N, voxel)
        # the fine pass carries a guard record (fp16 range report; error check of the narrow-band sweep): read behind the
        # marching-cubes size read-back in surfaces().  The volumes go to marching cubes only (mc_only).
        vh, vo, ticket = hip.fine_begin(N, norg.tolist(), nvs.item(), mode, hand=hb, obj=ob, mc_only=True)
        return {"vol_hand": vh, "vol_obj": vo, "voxel_size": nvs, "origin": norg.tolist(), "bbox": b, "fine_ticket": ticket}

    def surfaces(r, sample, between=None):
        """Marching cubes (and the label pass) of one sample.  between(rebound), if given, is called once the marching-cubes
        launches are queued and before the component filter, label pass and host copies are: the caller queues pass 2 of the
        next sample there (it may overwrite the volumes only behind the emits), so the ~40 small launches of the post-processing
        never sit between a pass-1 read-back and the pass-2 launch (eval-mode trace: 1.3 ms of idle GPU per sample).  `rebound`
        tells it whether the decoder was re-bound to this sample."""
        rebound = False
        spec = r.pop("pending", None)
        if spec is not None:
            # a sample that was enqueued in one go: judge its coarse record now (everything of it has long run).  Accepted - the usual
            # case - means the fine pass ran on exactly the lattice the host arithmetic gives for these boxes (asdf_zoom_cube).
            judged = hip.coarse_judge(spec["coarse"])
            if judged[0]:
                origin, nvs = hip.lattice_of(spec)
                r.update({"vol_hand": spec["vol_hand"], "vol_obj": spec["vol_obj"], "voxel_size": nvs, "origin": origin, "bbox": judged[1],
                          "fine_ticket": spec["fine"], "mc_tickets": spec["mc"]})
            else:
                bind(sample)                       # refused: this sample again, step by step
                rebound = True
                r.update(second_pass(spec, judged=judged))
        ticket = r.pop("fine_ticket", None)

        def begin_counts():
            # count phases of both volumes (no host synchronisation)
            return {part: marching_cubes_begin(r["vol_" + part], 0.0, slot) for slot, (part, on) in enumerate((("hand", hb), ("obj", ob))) if on}

        # the count phases are queued BEFORE the fine pass's guard record is read (it is accepted all but never refused): one
        # host wait then covers the record and the sizes, instead of record -> launch -> sizes with the GPU idle in between
        # (a sample enqueued in one go brings its marching-cubes tickets along: counted, and usually emitted, behind its fine pass)
        tickets = r.pop("mc_tickets", None) or begin_counts()
        while hip.fine_needs_repeat(ticket):
            # pass 2 left the fp16 range (the decoder has been re-calibrated, or switched to the fp32 kernel) or its
            # narrow-band form was not accepted: repeat this sample's pass 2 - and its count phases
            bind(sample)
            rebound = True
            r["vol_hand"], r["vol_obj"], ticket = hip.fine_begin(N, r["origin"], float(r["voxel_size"]), mode, hand=hb, obj=ob, mc_only=True)
            tickets = begin_counts()
        for part, on in (("hand", hb), ("obj", ob)):
            r["V_" + part] = r["F_" + part] = 0
            if on:
                try:
                    v, f = marching_cubes_finish(tickets[part])
                except (ValueError, RuntimeError) as e:         # the reference logs and skips (utils/mesh.py:353-358)
                    r["mc_error_" + part] = str(e)
                    continue
                r["verts_" + part], r["faces_" + part] = v, f
                r["V_" + part], r["F_" + part] = v.shape[0], f.shape[0]
                c = seen.get(part, (0, 0))
                seen[part] = (max(c[0], v.shape[0]), max(c[1], f.shape[0]))
        if between is not None:
            between(rebound)
        for part, on in (("hand", hb), ("obj", ob)):
            if on and "verts_" + part in r:
                v, f = r["verts_" + part], r["faces_" + part]
                if label_out and part == "hand":
                    # the vertex arithmetic of utils/mesh.py:138-141 in fp32, on the device
                    pts = v * float(r["voxel_size"]) + torch.tensor(r["origin"], dtype=torch.float32, device=v.device)
                    bind(sample)
                    r["labels_hand"] = hip.classify_points(pts, want_sdf=False)[3]
                if host_copy:
                    # K8 right behind marching cubes: only the largest component (what the file holds) crosses to the
                    # host, plus the whole surface when the label pass needs every vertex
                    from .mesh_post import keep_largest_component_device
                    kv, kf, counts = keep_largest_component_device(v, f, r["voxel_size"], r["origin"])
                    r["kept_dev_" + part] = (kv, kf, counts)       # (the eval-mode hook samples the kept surface on the device)
                    to_host(r, "kept_verts_" + part, kv)
                    to_host(r, "kept_faces_" + part, kf)
                    done = to_host(r, "kept_counts_" + part, counts)
                    if "labels_" + part in r:
                        to_host(r, "verts_" + part, v)
                        to_host(r, "faces_" + part, f)
                        done = to_host(r, "labels_" + part, r["labels_" + part])
                    r["copy_done_" + part] = done           # the side stream is in order: the last event covers all

    r = second_pass(first_pass(cur))
    nxt = next(it, None)
    bbox_next = first_pass(nxt) if nxt is not None else None
    while True:
        if nxt is not None:
            queued = []

            def queue_next_pass2(rebound):
                if rebound:
                    bind(nxt)
                queued.append(second_pass(bbox_next))

            # fetch sample k+2 while the queue is short: a source that uploads its codes with a blocking copy would
            # otherwise sit behind pass 2 of sample k+1 and hold the consumer back for a whole pass
            after = next(it, None)
            surfaces(r, cur, queue_next_pass2)          # MC of sample k, queued behind pass 1 of sample k+1
            r_next = queued[0]
            if midpoint is not None:
                midpoint(cur[0], r)
            bbox_after = first_pass(after) if after is not None else None
        else:
            surfaces(r, cur)
            if midpoint is not None:
                midpoint(cur[0], r)
        yield cur[0], r
        if nxt is None:
            return
        cur, r, nxt, bbox_next = nxt, r_next, after, bbox_after


_gt_proc = None
_gt_lock = threading.Lock()


def _ground_truth_process():
    """The ground-truth worker process (gt_worker.py), started on first use and shared by every reconstruct() call of this
    process: its start (interpreter + numpy, ~0.2 s) is paid once, not per call.  A worker that has died is replaced."""
    global _gt_proc
    with _gt_lock:
        if _gt_proc is None or _gt_proc.poll() is not None:
            import atexit
            import subprocess
            import sys
            env = dict(os.environ)
            root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
            env["PYTHONPATH"] = root + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
            first = _gt_proc is None
            _gt_proc = subprocess.Popen([sys.executable, "-m", "alignsdf_amd.gt_worker"], stdin=subprocess.PIPE,
                                        stdout=subprocess.PIPE, env=env)
            if first:
                atexit.register(_stop_ground_truth_process)
        return _gt_proc


def _stop_ground_truth_process():
    global _gt_proc
    proc, _gt_proc = _gt_proc, None
    if proc is not None and proc.poll() is None:
        try:
            proc.stdin.close()
            proc.wait(timeout=5)
        except Exception:
            proc.kill()


class GroundTruthPrefetcher:
    """Eval mode reads one ground-truth mesh per sample (utils/mesh.py:386-389) and samples 30 000 points from it
    (deep_sdf/metrics/icp_trans_scale.py:19-23): file parsing and sampling run in a worker process, one or two samples ahead of
    the consumer, so that neither sits between two decoder passes.  get() returns the pinned [samples, 3] fp64 target points, or
    None when the file is missing and allow_missing_gt is set; a missing file otherwise raises like the reference's trimesh.load.
    The work itself runs in a PROCESS (alignsdf_amd/gt_worker.py, numpy only - like the reference's DataLoader worker): 15 ms of
    parsing and sampling per sample in a thread would hold the interpreter lock exactly when the main thread has to turn a coarse
    pass's boxes into the next launch (measured: 1.3 ms of GPU idle per pass in eval mode).  The thread here only moves requests
    and replies over the pipes (blocking reads release the lock).  ASDF_GT_WORKER=thread keeps everything in-process."""

    def __init__(self, task, data_root, allow_missing_gt=False, samples=30000, seed=1):
        from concurrent.futures import ThreadPoolExecutor
        from .frontend import quick_gil_handover
        self._switch_interval = quick_gil_handover()
        self.task, self.data_root, self.allow_missing, self.samples, self.seed = task, data_root, allow_missing_gt, samples, seed
        self.pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="asdf-gt")
        self.proc = None if os.environ.get("ASDF_GT_WORKER", "process") == "thread" else _ground_truth_process()
        self.jobs = {}
        # pinned staging for the target samples, allocated once (pinning per sample takes a runtime lock that the main thread's
        # launches queue behind); a slot is reused four samples later, long after its ICP has been waited for
        self.ring = [torch.empty((samples, 3), dtype=torch.float64).pin_memory() for _ in range(4)] if torch.cuda.is_available() else []
        self.turn = 0

    def _load(self, path):
        from . import gt_worker
        if self.proc is None:
            return gt_worker.load_samples(path, self.samples, self.seed)
        with _gt_lock:                                  # (one request / reply pair at a time on the shared pipes)
            gt_worker.write_message(self.proc.stdin, (path, self.samples, self.seed))
            reply = gt_worker.read_message(self.proc.stdout)
        if reply is None and self.proc.poll() is not None:
            raise RuntimeError("ground-truth worker process ended with code %s" % self.proc.returncode)
        if isinstance(reply, tuple) and reply and reply[0] == "error":
            raise RuntimeError("ground-truth mesh %s: %s" % (path, reply[1]))
        return reply

    def prefetch(self, ply_filename_out):
        if ply_filename_out not in self.jobs:
            path = mesh_utils.ground_truth_mesh_path(ply_filename_out, self.task, self.data_root)
            self.jobs[ply_filename_out] = (path, self.pool.submit(self._load, path))

    def get(self, ply_filename_out):
        self.prefetch(ply_filename_out)
        path, job = self.jobs.pop(ply_filename_out)
        pts = job.result()
        if pts is not None:
            pts = torch.from_numpy(pts)
            if self.ring:
                slot = self.ring[self.turn % len(self.ring)]
                self.turn += 1
                slot.copy_(pts)
                pts = slot
        if pts is None:
            if not self.allow_missing:
                raise FileNotFoundError("eval_mode: ground-truth mesh %s not found (data_root=%r); pass allow_missing_gt to write "
                                        "unaligned meshes instead" % (path, self.data_root))
            import logging
            logging.warning("eval_mode: ground-truth mesh %s not found; writing the unaligned mesh (allow_missing_gt)" % path)
        return pts

    def discard(self, ply_filename_out):
        """A prefetched sample the consumer does not need after all (no hand surface: nothing to align): drop its job - a failure
        of a mesh nobody reads is not an error, and nothing stays behind in self.jobs."""
        job = self.jobs.pop(ply_filename_out, None)
        if job is not None:
            job[1].cancel()

    def close(self):
        from .frontend import restore_gil_handover
        for _, job in self.jobs.values():           # (prefetched, never asked for)
            job.cancel()
        self.jobs.clear()
        self.pool.shutdown(wait=True)               # (the worker process is shared by later calls and ends with the interpreter)
        restore_gil_handover(self._switch_interval)


class FileWriter:
    """PLY files are written on a worker thread (tobytes + write release the GIL): the consumer hands over host arrays and
    moves on to the next sample.  A failed write (disk full, permissions) surfaces at the NEXT write_ply / poll - one sample later,
    not after the whole shard has been decoded - and close() waits for every file; close(failing=True), from a `finally` that is
    unwinding another exception, only logs what the writer still has to report so that the original error survives."""

    def __init__(self):
        from concurrent.futures import ThreadPoolExecutor
        from .frontend import quick_gil_handover
        self._switch_interval = quick_gil_handover()
        self.pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="asdf-ply")
        self.jobs = []

    def poll(self):
        """Raise the failure of any write that has finished; forget the ones that succeeded."""
        pending, first = [], None
        for path, j in self.jobs:
            if not j.done():
                pending.append((path, j))
                continue
            err = j.exception()
            if err is not None:
                import logging
                logging.error("PLY write of %s failed: %s", path, err)      # every failed path is named, the first is raised
                if first is None:
                    first = err
        self.jobs = pending                        # (a failure is reported once)
        if first is not None:
            raise first

    def write_ply(self, path, verts, faces):
        from .ply import write_ply
        self.poll()
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        self.jobs.append((path, self.pool.submit(write_ply, path, verts, faces)))

    def close(self, failing=False):
        from .frontend import restore_gil_handover
        self.pool.shutdown(wait=True)
        restore_gil_handover(self._switch_interval)
        jobs, self.jobs = self.jobs, []
        first = None
        for path, j in jobs:
            err = j.exception()
            if err is not None:
                if failing:
                    import logging
                    logging.error("PLY write of %s failed: %s", path, err)
                elif first is None:
                    first = err
        if first is not None:
            raise first


def write_sweeps_json(out_dir, start_point, end_point, report, samples, cube_dim, components=None, stride=1):
    """`<output_dir>/sweeps_<start>_<end>.json` - next to `meshes/`, whose listing stays the reference's (reconstruct.py:34-35) -: which
    sweeps produced the volumes behind this shard's meshes.
    The default sweeps of a mesh-producing run rest on a measured, statistical certificate; a run whose sweeps
    were refused and repeated, whose modes were switched off, or which ended on the fp32 chain must say so NEXT TO ITS FILES, not only
    in a log line.  The reference writes nothing of the kind (utils/mesh.py:82-121 evaluates every voxel in fp32)."""
    hip = report.get("evaluator")
    body = {"range": [int(start_point), int(end_point)], "samples": int(samples), "cube_dim": int(cube_dim),
            "sweeps": hip.sweep_report(report.get("snapshot")) if hip is not None else None}
    if int(stride) != 1:
        body["stride"] = int(stride)               # (a strided shard: samples start, start + stride, ... below end)
    if components is not None:
        # f1, the largest-component filter (K8; PARITY UNPINNED against trimesh): how many components of >= 4 faces it dropped as open /
        # non-manifold - the only place where trimesh's fill_holes (utils/mesh.py:371 -> graph.split -> submesh(repair=True); not
        # reproduced) could have changed which component is written.  Expected 0 on marching-cubes surfaces
        body["dropped_open_components"] = int(components.get("open", 0))
        body["dropped_small_components"] = int(components.get("small", 0))
        body["surfaces_filtered"] = int(components.get("surfaces", 0))
    path = os.path.join(out_dir, "sweeps_%d_%d.json" % (int(start_point), int(end_point)))
    with open(path, "w") as f:
        json.dump(body, f, indent=1)
    return path


def merge_sweeps_json(out_dir, out_name="sweeps.json", ranges=None):
    """One `sweeps.json` for the whole run from the per-shard files (dist_reconstruct: rank 0, behind the gather): the per-shard
    records side by side plus the totals a reader looks for first.  `ranges` = the [start, end) shard ranges of THIS run: only their
    `sweeps_<start>_<end>.json` are merged (a glob also picked up the shards an earlier run with another world size had
    left in the directory); None = every shard file present.  On a multi-node run without a shared file system rank 0 sees its own
    host's shards only - `shards_missing` in the totals says how many of `ranges` had no file."""
    shards, missing = [], 0
    if ranges is None:
        wanted = sorted(n for n in os.listdir(out_dir) if n.startswith("sweeps_") and n.endswith(".json"))
    else:
        wanted = ["sweeps_%d_%d.json" % (int(a), int(b)) for a, b in ranges]
    for name in wanted:
        try:
            with open(os.path.join(out_dir, name)) as f:
                shards.append(json.load(f))
        except (OSError, ValueError):
            missing += 1                               # (a shard that failed before it wrote one, or another host's)
    shards.sort(key=lambda b: b["range"][0])
    tot = {"samples": sum(b["samples"] for b in shards), "sweeps_audited": 0, "sweeps_refused": 0, "sweeps_repeated": 0,
           "ordinary_sweeps": 0, "modes_switched_off": [], "fell_back_to_fp32_chain": False, "shards_missing": missing,
           "min_tau_over_sigma": None, "min_tau_over_estimate": None, "tail_ratio_max": None, "dropped_open_components": 0}
    for b in shards:
        sw = b.get("sweeps") or {}
        for k in ("sweeps_audited", "sweeps_refused", "sweeps_repeated"):
            tot[k] += int(sw.get(k, 0))
        tot["ordinary_sweeps"] += int(sw.get("coarse_pass", {}).get("ordinary_sweeps", 0)) + int(sw.get("fine_pass", {}).get("ordinary_sweeps", 0))
        tot["dropped_open_components"] += int(b.get("dropped_open_components", 0) or 0)
        for k in ("min_tau_over_sigma", "min_tau_over_estimate"):
            v = sw.get(k)
            if v is not None:
                tot[k] = v if tot[k] is None else min(tot[k], v)
        for v in (sw.get("tail_ratio_max") or {}).values():
            if v is not None:
                tot["tail_ratio_max"] = v if tot["tail_ratio_max"] is None else max(tot["tail_ratio_max"], v)
        tot["modes_switched_off"] += ["samples %d..%d: %s" % (b["range"][0], b["range"][1], m) for m in sw.get("modes_switched_off", [])]
        tot["fell_back_to_fp32_chain"] = tot["fell_back_to_fp32_chain"] or bool(sw.get("arithmetic", {}).get("fell_back_to_fp32_chain"))
    path = os.path.join(out_dir, out_name)
    with open(path, "w") as f:
        json.dump({"totals": tot, "shards": shards}, f, indent=1)
    return path


def sweeps_summary_line(tot):
    """The one line about the sweeps a user of dist_reconstruct actually reads, from merge_sweeps_json's totals."""
    if not tot.get("sweeps_audited") and not tot.get("sweeps_refused"):
        line = "sweeps: %d ordinary (every voxel of both passes at <= 1e-5; --fast selects the audited one-plane sweeps)" % tot.get("ordinary_sweeps", 0)
    else:
        f = lambda v: "n/a" if v is None else "%.1f" % v
        line = ("sweeps (--fast): %d audited one-plane, %d REFUSED and repeated, %d ordinary; certificate: min allowance / sigma %s, "
                "min allowance / estimate %s, largest tail ratio %s" % (
                    tot["sweeps_audited"], tot["sweeps_refused"], tot.get("ordinary_sweeps", 0), f(tot.get("min_tau_over_sigma")),
                    f(tot.get("min_tau_over_estimate")), f(tot.get("tail_ratio_max"))))
    if tot.get("modes_switched_off"):
        line += "; MODES SWITCHED OFF: %s" % "; ".join(tot["modes_switched_off"])
    if tot.get("fell_back_to_fp32_chain"):
        line += "; a decoder FELL BACK to the fp32 chain"
    if tot.get("dropped_open_components"):
        line += "; %d open components dropped by the largest-component filter (where trimesh's fill_holes could have differed)" % tot["dropped_open_components"]
    if tot.get("shards_missing"):
        line += "; %d shard report(s) missing" % tot["shards_missing"]
    return line


def reconstruct_sample(decoder, specs, latent, mano_results, obj_results, N, mesh_filename=None, grid_mode="reference",
                       eval_mode=False, task="obman", scale=None):
    """One sample through the hot path.  Returns a record dict; writes <mesh_filename>_hand.ply / _obj.ply when
    a filename is given, otherwise leaves the meshes on the device (`verts_*`, `faces_*` tensors)."""
    from .marching_cubes import marching_cubes_device
    hand_branch, obj_branch = specs.get("HandBranch", True), specs.get("ObjectBranch", True)
    t0 = time.perf_counter()
    r = mesh_utils.decode_two_pass(hand_branch, obj_branch, decoder, latent, mano_results, obj_results, specs, N, grid_mode,
                                   mc_only=True)
    rec = {"V_hand": 0, "F_hand": 0, "V_obj": 0, "F_obj": 0}
    if mesh_filename is not None:
        stats = {}
        offset = None
        if hand_branch:
            v, f, offset, scale = mesh_utils.convert_sdf_samples_to_ply(r["vol_hand"], r["origin"], r["voxel_size"],
                                                                        mesh_filename + "_hand.ply", None, None, eval_mode, task)
            stats["hand"] = (0, 0) if v is None else (len(v), len(f))
        if obj_branch:
            v, f, _, _ = mesh_utils.convert_sdf_samples_to_ply(r["vol_obj"], r["origin"], r["voxel_size"],
                                                                mesh_filename + "_obj.ply", offset, scale, False)
            stats["obj"] = (0, 0) if v is None else (len(v), len(f))
        for part, (nv, nf) in stats.items():
            rec["V_" + part], rec["F_" + part] = nv, nf
    else:
        for part, on in (("hand", hand_branch), ("obj", obj_branch)):
            if not on:
                continue
            try:
                v, f = marching_cubes_device(r["vol_" + part], 0.0)
            except (ValueError, RuntimeError):
                continue          # the reference logs and skips (utils/mesh.py:353-358)
            rec["V_" + part], rec["F_" + part] = v.shape[0], f.shape[0]
            rec["verts_" + part], rec["faces_" + part] = v, f
    rec["voxel_size"], rec["origin"] = float(r["voxel_size"]), r["origin"]
    rec["seconds"] = time.perf_counter() - t0
    return rec


def reconstruct(loaded_model, specs, split_filename, output_dir, start_point, end_point, task="obman", device="cuda", scale=None,
                cube_dim=128, label_out=False, viz=False, eval_mode=False, code_source=None, grid_mode="reference",
                data_root="data", allow_missing_gt=False, fast=None, stride=1, on_record=None):
    """Reconstruct samples [start_point, end_point) of a split file (reconstruct.py:33-95).  `loaded_model` is the
    decoder module, or any wrapper exposing it as `.module.decoder` / `.decoder` like the reference's DataParallel model.
    `fast`: see pipelined_two_pass (default: ordinary sweeps, every voxel at <= 1e-5).  `stride` > 1: every stride-th sample of the
    range (the strided shards of dist_reconstruct --shard strided).  `on_record(rec)` is called with every finished sample's record
    (dist_reconstruct keeps its shard's records file current with it).  Returns the list of per-sample records."""
    mesh_dir = os.path.join(output_dir, "meshes")
    os.makedirs(mesh_dir, exist_ok=True)
    stride = max(1, int(stride))
    with open(split_filename, "r") as f:
        names = json.load(f)["filenames"][int(start_point):int(end_point):stride]
    decoder = loaded_model
    for attr in ("module", "decoder"):
        decoder = getattr(decoder, attr, decoder)
    if code_source is None:
        # the encoder front end is outside this build: without codes there is nothing to reconstruct.  (Round 1 substituted
        # synthetic latents here, which wrote meaningless <real sample>_hand.ply files that the evaluation then scored.)
        raise ValueError("reconstruct() needs a code_source: npz_code_source(<dir of per-sample .npz codes>), "
                         "model_output_code_source(<encoder callable>), or synthetic_code_source(...) for tests and benchmarks")
    hand_on = specs.get("HandBranch", True)
    gt = GroundTruthPrefetcher(task, data_root, allow_missing_gt) if eval_mode and hand_on else None
    writer = FileWriter()

    def samples():
        for k, path in enumerate(names):
            name = path.split("/")[-1].split(".")[0]                       # reconstruct.py:78
            if gt is not None:
                gt.prefetch(os.path.join(mesh_dir, "%s_hand.ply" % name))  # parsed + sampled by the time the ICP hook wants it
            latent, mano_results, obj_results = code_source(name, int(start_point) + k * stride)
            yield (int(start_point) + k * stride, name), latent, mano_results, obj_results

    records = []
    sweeps = {}
    components = {"open": 0, "small": 0, "surfaces": 0}      # what K8 dropped, and why (write_sweeps_json)
    try:
        with torch.no_grad():
            t_prev = time.perf_counter()

            def hand_path(name):
                return os.path.join(mesh_dir, "%s_hand" % name)

            def kept(r, part):
                c = r["host_kept_counts_" + part].numpy()
                components["open"] += int(c[4])
                components["small"] += int(c[5])
                components["surfaces"] += 1
                return r["host_kept_verts_" + part][:c[0]], r["host_kept_faces_" + part][:c[1]]

            def begin_hand(key, r):
                """Eval mode: the ICP of the hand mesh, slotted between two decoder passes (see pipelined_two_pass).  Everything
                it needs is on the device already - the largest component straight from K8, sampled there (bit-identical to the
                host sampler), normalised there - and the ground truth's samples come from the prefetch worker: the hook only
                enqueues, the consumer below only waits for the result."""
                if "verts_hand" in r:
                    from .icp import start_alignment_device
                    target = gt.get(hand_path(key[1]) + ".ply")
                    r["icp_job"] = None if target is None else start_alignment_device(*r["kept_dev_hand"], r["origin"], r["voxel_size"], target)
                else:
                    gt.discard(hand_path(key[1]) + ".ply")          # no hand surface: the prefetched ground truth is not needed

            for (index, name), r in pipelined_two_pass(decoder, specs, samples(), cube_dim, grid_mode, host_copy=True,
                                                        label_out=label_out and hand_on, midpoint=begin_hand if gt is not None else None,
                                                        report=sweeps, fast=fast):
                rec = {"index": index, "name": name, "V_hand": r["V_hand"], "F_hand": r["F_hand"], "V_obj": r["V_obj"],
                       "F_obj": r["F_obj"], "voxel_size": float(r["voxel_size"]), "origin": r["origin"]}
                # the object is written with the hand's ICP translation / scale as offset / scale whenever the hand branch
                # is on - zeros / one outside eval mode or when the hand has no surface (utils/mesh.py:123-133,186-195)
                offset, sc = (np.array([0, 0, 0]), np.array([1])) if hand_on else (None, scale)
                for part in ("hand", "obj"):
                    if "verts_" + part in r:
                        r["copy_done_" + part].synchronize()          # side-stream D2H of this mesh only
                        base = os.path.join(mesh_dir, "%s_%s" % (name, part))
                        kv, kf = kept(r, part)
                        _, faces, points = mesh_utils.place_vertices(kv, kf, r["origin"], r["voxel_size"], None if part == "hand" else offset,
                                                                     None if part == "hand" else sc)
                        trans, icp_scale = np.array([0, 0, 0]), np.array([1])
                        if part == "hand" and gt is not None:
                            job = r.pop("icp_job", None)
                            if job is None:
                                rec["icp_skipped"] = True          # allow_missing_gt: no ground-truth mesh, written unaligned
                            else:
                                from .icp import finish_icp
                                out = finish_icp(job, points)
                                points = out["vertices"]
                                trans, icp_scale = np.asarray(out["all_trans"]).reshape(1, 3), np.asarray(out["all_scale"]).reshape(1)
                        writer.write_ply(base + ".ply", points, faces)
                        if part == "hand":
                            offset, sc = trans, icp_scale
                            rec["icp_trans"], rec["icp_scale"] = np.asarray(trans).reshape(-1).tolist(), float(np.asarray(icp_scale).reshape(-1)[0])
                            if "host_labels_hand" in r:
                                verts, faces, vertices = mesh_utils.place_vertices(r["host_verts_hand"], r["host_faces_hand"], r["origin"],
                                                                                   r["voxel_size"])
                                labels = r["host_labels_hand"].float()
                                mesh_utils.write_label_outputs(vertices, faces, labels, base, offset, sc, viz)
                                rec["labels_hand"] = np.bincount(labels.long().numpy(), minlength=1).tolist()
                    elif "mc_error_" + part in r:
                        import logging
                        logging.warning("Cannot reconstruct mesh from '{}'".format(os.path.join(mesh_dir, "%s_%s.ply" % (name, part))))
                        print(r["mc_error_" + part])
                now = time.perf_counter()
                rec["seconds"], t_prev = now - t_prev, now
                records.append(rec)
                if on_record is not None:
                    on_record(rec)
    except BaseException as exc:
        # what was finished before the failure travels with the exception (dist_reconstruct.run_sharded writes it to the shard's
        # records file and hands it to the gather: VERDICT r05 item 2)
        exc.partial_records = list(records)
        if sweeps:
            try:                                   # the shard's sweep report exists even when the shard did not finish
                write_sweeps_json(output_dir, start_point, end_point, sweeps, len(records), cube_dim, components, stride)
            except Exception as e:
                import logging
                logging.error("writing the sweep report of a failed shard failed: %s", e)
        # unwinding: the helpers are closed without letting THEIR errors replace the one in flight
        if gt is not None:
            try:
                gt.close()
            except Exception as e:          # (a failing close must not mask the error being raised)
                import logging
                logging.error("closing the ground-truth prefetcher failed: %s", e)
        writer.close(failing=True)
        raise
    if gt is not None:
        gt.close()
    writer.close()                      # every file is on disk (or its error raised) before reconstruct() returns
    if sweeps:
        write_sweeps_json(output_dir, start_point, end_point, sweeps, len(records), cube_dim, components, stride)
    return records


def code_source_from_args(args, specs, parser):
    """--codes DIR | --synthetic; neither is an error (there is no silent default)."""
    if args.code_dir:
        return npz_code_source(args.code_dir)
    if getattr(args, "synthetic", False):
        return synthetic_code_source("nerf3" if specs["PointFeatSize"] == 3 else "both9")
    parser.error("no latent codes: pass --codes <dir of <sample>.npz written by the encoder run> (or --synthetic for a benchmark run)")


def load_experiment(model_directory, device="cuda"):
    """specs.json + ModelParameters/latest.pth -> (specs, decoder module) (reconstruct.py:166-170,
    networks/model_utils.py:40-47; only the `module.decoder.*` tensors are read)."""
    specs = json.load(open(os.path.join(model_directory, "specs.json")))
    ckpt = torch.load(os.path.join(model_directory, "ModelParameters", "latest.pth"), map_location="cpu")
    return specs, build_decoder(specs, ckpt.get("model_state_dict", ckpt))


def add_sweep_arguments(p):
    """--fast / --coarse / --fine of the reconstruction CLIs.  Round 6: meshes that get scored are produced
    by ORDINARY sweeps unless the caller opts out - every voxel of both passes in the reference's arithmetic class, like
    utils/mesh.py:27-115 evaluates every voxel in fp32."""
    p.add_argument("--fast", action="store_true",
                   help="opt in to the audited one-plane sweeps (2.8 x the throughput at N = 256): one fp16 plane decides the SIGNS "
                        "of both lattices, every value the zoom cube or marching cubes reads is re-evaluated at <= 1e-5, and a "
                        "statistical certificate (audit sample per sweep, periodic whole-lattice comparisons) refuses and repeats a "
                        "sweep that does not hold; identical meshes on everything measured (DESIGN.md 3c).  Default: ordinary sweeps "
                        "- every voxel at <= 1e-5")
    p.add_argument("--coarse", choices=["exact", "box"], default=None,
                   help="coarse pass one by one: an ordinary sweep (default) or the audited box-only one-plane sweep with exact "
                        "re-evaluation of the voxels that can move the zoom cube")
    p.add_argument("--fine", choices=["exact", "band"], default=None,
                   help="fine pass one by one: an ordinary sweep (default) or the audited narrow-band sweep (one fp16 plane, the "
                        "corners of every cell that can be active re-evaluated as an ordinary sweep would)")


def apply_sweep_arguments(args):
    if args.coarse:
        os.environ["ASDF_COARSE"] = args.coarse          # read when the decoder is packed
    if args.fine:
        os.environ["ASDF_FINE"] = args.fine


def main(argv=None):
    p = argparse.ArgumentParser(description="Reconstruct hand / object meshes with the MI355X-native hot path.")
    p.add_argument("--model", "-e", dest="model_directory", default="./pretrained_model")
    p.add_argument("--split", "-s", dest="split_filename", default=None)
    p.add_argument("--task", "-t", dest="task", default="obman", choices=["obman", "dexycb"])
    p.add_argument("--start_point", dest="start_point")
    p.add_argument("--end_point", dest="end_point")
    p.add_argument("--eval_mode", dest="eval_mode", action="store_true")
    p.add_argument("--label", dest="label_out", action="store_true")
    p.add_argument("--viz", dest="viz", action="store_true")
    p.add_argument("--codes", dest="code_dir", default=None, help="directory of precomputed <sample>.npz codes")
    p.add_argument("--synthetic", action="store_true", help="deterministic synthetic codes (tests / benchmarks only: the meshes mean nothing)")
    p.add_argument("--allow_missing_gt", action="store_true", help="eval mode: write unaligned meshes when a ground-truth mesh is missing instead of aborting")
    p.add_argument("--cube_dim", type=int, default=128, help="grid resolution (reference CLI hard-codes 128, reconstruct.py:178)")
    add_sweep_arguments(p)
    args = p.parse_args(argv)
    apply_sweep_arguments(args)
    split = args.split_filename or {"obman": "input/obman.json", "dexycb": "input/dexycb.json"}[args.task]
    output_dir = os.path.join(args.model_directory, "Eval_" + args.task)
    os.makedirs(output_dir, exist_ok=True)
    specs, decoder = load_experiment(args.model_directory)
    if args.start_point is None or args.end_point is None:
        with open(split) as f:
            args.start_point, args.end_point = 0, len(json.load(f)["filenames"])
    source = code_source_from_args(args, specs, p)
    return reconstruct(decoder, specs, split, output_dir, args.start_point, args.end_point, task=args.task, cube_dim=args.cube_dim,
                       label_out=args.label_out, viz=args.viz, eval_mode=args.eval_mode, code_source=source,
                       allow_missing_gt=args.allow_missing_gt, fast=True if args.fast else None)


if __name__ == "__main__":
    main()
