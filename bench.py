#!/usr/bin/env python3
"""Benchmark of the AlignSDF reconstruction hot path on MI355X.

A "step" is one sample through the path BASELINE.json names: bind latent (K0 fold) -> dense-grid SDF decode of
both heads on [-1,1]^3 (K1, with the negative-voxel bbox fused) -> zoom cube -> second N^3 decode (K1) ->
Lewiner marching cubes on the hand and the object volume (K3-K6).  Weights and codes are resident in HBM
before the timed region; the meshes stay on the device.  One step yields 2 meshes (hand + object).  The K timed
steps run through the product's own sample pipeline (alignsdf_amd.reconstruct.pipelined_two_pass), started empty
and drained inside the timed region.

    python bench.py --gpus N --steps K --warmup W [--grid 256] [--tag nerf3|both9]

Multi-GPU: samples are independent, so ranks share nothing on the data path (weak scaling: K samples per GPU); the
per-sample records are gathered to rank 0 over RCCL at the end.  One process per GPU.  Either the caller launches the
ranks (torch.distributed.run: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment), or - when `--gpus N` is given
with N > 1 and no WORLD_SIZE is set - this script spawns the N ranks itself (one child process per GPU, rendezvous on
127.0.0.1) and relays rank 0's line.  `n_gpus` in the line is the value of an all_reduce of ones over the process group,
i.e. the number of ranks that actually took part.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_POINT_HEAD = 1_573_888      # dense formulation the reference executes
EXEC_FLOP_PER_POINT_HEAD = 2 * (4 * 512 + 512 * 256 + 260 * 512 + 512 * 512 + 512)   # after folding the latent columns
PEAK_FP32_MFMA_TFLOPS = 157.3        # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_F16_MFMA_TFLOPS = 2516.6        # MI355X_MICROARCH.md: dense BF16/F16 MFMA (v_mfma_f32_32x32x16_f16: 32 cycles/SIMD at 2.4 GHz)
PEAK_HBM_GBS = 8000.0                # MI355X_MICROARCH.md: HBM3E spec peak
# split-half kernel: the three hidden GEMMs issue 3 fp16 MFMAs per product sum (padded shapes), layers 0 / 2's point
# features stay on the fp32 MFMA
EXEC_F16_FLOP_PER_POINT_HEAD = 3 * 2 * (512 * 256 + 256 * 512 + 512 * 512)
EXEC_F16_SIDE_F32_FLOP_PER_POINT_HEAD = 2 * (4 * 512 + 4 * 512 + 512)
CHUNK = 2 ** 18                      # reconstruct.py:93


# ------------------------------------------------------------------------------------------------------------------
# launching: --gpus N without an external launcher
# ------------------------------------------------------------------------------------------------------------------
def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_environments(n, base_env=None, port=None):
    """Environment of each of the n ranks this script spawns (what torch.distributed.run would export)."""
    port = port or free_port()
    envs = []
    for r in range(n):
        e = dict(base_env if base_env is not None else os.environ)
        e.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                 MASTER_PORT=str(port))
        e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # the host driver only supports dmabuf IPC (RCCL needs it)
        envs.append(e)
    return envs


def spawn_ranks(n, argv, timeout=None):
    """Run this script once per rank; relay rank 0's stdout (the JSON line).  Returns the exit code."""
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=e, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL)
             for r, e in enumerate(rank_environments(n))]
    out, _ = procs[0].communicate(timeout=timeout)
    codes = [procs[0].returncode] + [p.wait(timeout=timeout) for p in procs[1:]]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    return max(abs(c) for c in codes)


# ------------------------------------------------------------------------------------------------------------------
# CPU baseline: the oracle = the port of the reference's CPU op sequence, timed on this box's host cores
# ------------------------------------------------------------------------------------------------------------------
def cpu_description():
    """(model string, physical cores, logical CPUs) from lscpu / /proc/cpuinfo."""
    model, sockets, cores_per_socket, logical = "unknown", 1, None, os.cpu_count() or 1
    try:
        for line in subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout.splitlines():
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "Model name":
                model = v
            elif k == "Socket(s)":
                sockets = int(v)
            elif k == "Core(s) per socket":
                cores_per_socket = int(v)
    except (OSError, ValueError, subprocess.SubprocessError):
        pass
    physical = sockets * cores_per_socket if cores_per_socket else logical
    return model, physical, logical


def cpu_baseline(tag, N, gpu, golden_dir=os.path.join(ROOT, "tests", "golden")):
    """One sample of the reference's CPU path, leg by leg, with the oracle (oracle/sdf_oracle.py, oracle/mc33.py: the
    restatement of utils/mesh.py:27-115,198-256 + networks/model.py:285-350 + skimage's Lewiner MC):

      grid build [P,3] (x2 passes) + chunk loop (slice, embed, expand + cat, 10 GEMMs, write-back; 2 x P / 2^18 chunks)
      + nonzero / min / max zoom cube + sequential MC on both volumes.

    Bounded sample: the chunk leg is the MEDIAN of 5 whole chunks (after 3 warm-up chunks) of the real pass-2 lattice,
    every other leg is run in full.  The thread count is swept on a whole chunk.  What it computes is checked in-run:
    chunk outputs against the GPU volumes (1e-5), zoom cube and V / F against the GPU's, probes against the committed
    reference goldens when the configuration has them.  `gpu` = dict(vol1_hand, vol1_obj, vol_hand, vol_obj, voxel_size,
    origin, V/F) of one sample on the host."""
    import torch
    from alignsdf_amd import synthetic as syn
    from oracle import mc33, sdf_oracle as orc
    specs, sd = syn.specs_for(tag), syn.full_state_dict(tag)
    lat = torch.from_numpy(syn.latent_code(gpu["sample"]))
    mano = obj = None
    if tag == "both9":
        m, o = syn.pose_inputs(gpu["sample"])
        mano = {k: torch.from_numpy(v) for k, v in m.items()}
        obj = {k: torch.from_numpy(v) for k, v in o.items()}
    model, physical, logical = cpu_description()
    checks = {}

    # leg 1: grid construction of pass 2 (pass 1 costs the same): index columns + coordinates for all P points
    t = time.perf_counter()
    coords2 = orc.grid_coords(N, gpu["voxel_size"], gpu["origin"])
    t_grid = time.perf_counter() - t
    P = N ** 3
    nchunks = (P + CHUNK - 1) // CHUNK

    # thread sweep on one WHOLE chunk (torch's default of one thread per logical CPU is rarely the fastest)
    def one_chunk(c):
        sub = coords2[c * CHUNK:(c + 1) * CHUNK]
        t0 = time.perf_counter()
        h, o = orc.decode_points(sd, lat, sub, specs, mano, obj, max_batch=CHUNK)
        return time.perf_counter() - t0, h, o

    one_chunk(0)                                                     # first touch: allocator, weight-norm, thread pool
    default_threads = torch.get_num_threads()
    sweep = {}
    for nt in sorted({logical, physical, max(1, physical // 2), max(1, physical // 4), 32, 16}):
        if nt > logical:
            continue
        torch.set_num_threads(nt)
        sweep[nt] = one_chunk(0)[0]
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)

    # leg 2: the chunk loop - 3 warm-up chunks, then 5 timed whole chunks spread over the lattice
    picks = sorted({int(round(x)) for x in np.linspace(0, nchunks - 1, 5)}) if nchunks >= 5 else list(range(nchunks))
    for c in picks[:3]:
        one_chunk(c)
    times, worst = [], 0.0
    vh, vo = gpu["vol_hand"].reshape(-1), gpu["vol_obj"].reshape(-1)
    for c in picks:
        dt, h, o = one_chunk(c)
        times.append(dt)
        worst = max(worst, float(np.abs(h.numpy() - vh[c * CHUNK:(c + 1) * CHUNK]).max()),
                    float(np.abs(o.numpy() - vo[c * CHUNK:(c + 1) * CHUNK]).max()))
    t_chunk = float(np.median(times))
    checks["chunks_vs_gpu_max_abs"] = worst
    assert worst <= 1e-5, "CPU baseline and GPU volumes differ by %.3e" % worst

    # leg 3: zoom cube from the pass-1 volumes (torch.nonzero + min / max, utils/mesh.py:198-256)
    t = time.perf_counter()
    nvs, norg, _ = orc.get_higher_res_cube(True, True, torch.from_numpy(gpu["vol1_hand"]), torch.from_numpy(gpu["vol1_obj"]), N, 2.0 / (N - 1))
    t_zoom = time.perf_counter() - t
    checks["zoom_cube_equal"] = bool(float(nvs) == float(gpu["voxel_size"]) and norg.tolist() == list(gpu["origin"]))
    assert checks["zoom_cube_equal"], "zoom cube of the CPU baseline differs from the GPU's"

    # leg 4: sequential Lewiner marching cubes on both pass-2 volumes
    t = time.perf_counter()
    counts = []
    for vol in (gpu["vol_hand"], gpu["vol_obj"]):
        try:
            v, f = mc33.marching_cubes_raw(vol, 0.0)
            counts.append([len(v), len(f)])
        except (ValueError, RuntimeError):
            counts.append([0, 0])
    t_mc = time.perf_counter() - t
    checks["mc_counts_equal_gpu"] = counts == [[gpu["V_hand"], gpu["F_hand"]], [gpu["V_obj"], gpu["F_obj"]]]
    assert checks["mc_counts_equal_gpu"], "marching-cubes counts differ: CPU %s" % counts

    # the committed reference goldens of this configuration (sample 0): probes of pass 2 and V / F
    gfile = os.path.join(golden_dir, "ref_fullsize.npz" if tag == "nerf3" else "ref_fullsize_%s.npz" % tag)
    if gpu["sample"] == 0 and os.path.exists(gfile):
        g = np.load(gfile)
        if "probe_sel_%d" % N in g.files:
            sel = g["probe_sel_%d" % N]
            h, o = orc.decode_points(sd, lat, coords2[torch.from_numpy(sel)], specs, mano, obj, max_batch=CHUNK)
            checks["probes_vs_reference_golden_max_abs"] = max(float(np.abs(h.numpy() - g["p2_hand_%d" % N]).max()),
                                                               float(np.abs(o.numpy() - g["p2_obj_%d" % N]).max()))
            assert checks["probes_vs_reference_golden_max_abs"] <= 1e-5
            if "mc_hand_%d" % N in g.files:
                checks["mc_counts_equal_reference_golden"] = counts == [g["mc_hand_%d" % N].tolist(), g["mc_obj_%d" % N].tolist()]
                assert checks["mc_counts_equal_reference_golden"]
    torch.set_num_threads(default_threads)

    t_sample = 2 * t_grid + 2 * nchunks * t_chunk + t_zoom + t_mc
    return {
        "value": 2.0 / t_sample, "unit": "meshes/s", "cores": best, "kind": "port",
        "cpu_model": model, "physical_cores": physical, "logical_cpus": logical,
        "sample": "one 2-pass sample = 2 x grid build (%.2f s each, run in full) + 2 x %d chunks of 2^18 points at %.3f s/chunk "
                  "(median of %d whole chunks after 3 warm-up chunks; both heads, torch CPU fp32, the reference's op sequence "
                  "incl. embed / expand+cat / write-back) + nonzero zoom cube (%.2f s, in full) + sequential Lewiner MC on both "
                  "%d^3 volumes (%.2f s, in full) = %.1f s; %d threads = fastest of the sweep %s on a whole chunk" % (
                      t_grid, nchunks, t_chunk, len(times), t_zoom, N, t_mc, t_sample, best,
                      {k: round(v, 3) for k, v in sorted(sweep.items())}),
        "seconds_per_sample": t_sample,
        "legs_seconds": {"grid_build_per_pass": t_grid, "chunk_median": t_chunk, "chunk_times": times, "chunks_per_pass": nchunks,
                         "zoom_cube": t_zoom, "marching_cubes_both": t_mc},
        "thread_sweep_seconds_per_chunk": {str(k): v for k, v in sorted(sweep.items())},
        "checked_in_run": checks, "mc_counts": counts,
    }


# ------------------------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--grid", type=int, default=256, help="grid resolution N (BASELINE metric is quoted at 256)")
    ap.add_argument("--tag", default="nerf3", choices=["nerf3", "both9"], help="nerf3 = ObMan config, both9 = DexYCB MANO-aligned")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-math", action="store_true", help="skip the second full record under the other arithmetic")
    ap.add_argument("--math", default=None, choices=["f32", "f16x3"],
                    help="arithmetic of the hidden GEMMs (default: the product's default, split-half fp16 MFMA)")
    ap.add_argument("--coarse", default=None, choices=["exact", "box"],
                    help="coarse pass of the two-pass flow in the timed region: an ordinary sweep, or the box-only one-plane "
                         "sweep with exact re-evaluation of the box candidates (default: the product's default)")
    ap.add_argument("--fine", default=None, choices=["exact", "band"],
                    help="fine pass in the timed region: an ordinary sweep, or the narrow-band sweep (one-plane values, exact "
                         "re-evaluation of the corners of every cell that can be active; for marching cubes only)")
    ap.add_argument("--no-other-coarse", action="store_true", help="skip the full records under the other coarse / fine passes")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no external launcher: one child per GPU, rendezvous on 127.0.0.1
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist
    from alignsdf_amd import synthetic as syn

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("ASDF_BENCH_DRYRUN"):
        # test hook (CPU test suite): rendezvous + the all_reduce of ones on gloo, no GPU work - proves that `--gpus N`
        # really starts N ranks that see each other
        n_seen = 1
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
            ones = torch.ones(1, dtype=torch.int32)
            dist.all_reduce(ones)
            n_seen = int(ones.item())
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"dryrun": True, "n_gpus": n_seen, "ranks_requested": args.gpus, "world_size_env": world}), flush=True)
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    # test hooks: ASDF_BENCH_BACKEND=gloo and ASDF_BENCH_SHARE_DEVICE=1 let several ranks share one GPU so that the
    # multi-rank path can be exercised on a single-GPU box; the driver's runs use RCCL with one GPU per rank
    backend = os.environ.get("ASDF_BENCH_BACKEND", "nccl")
    share = bool(os.environ.get("ASDF_BENCH_SHARE_DEVICE"))
    if local_rank >= torch.cuda.device_count() and not share:
        raise SystemExit("rank %d has no GPU of its own (%d visible); one process per GPU" % (rank, torch.cuda.device_count()))
    dev_index = local_rank % torch.cuda.device_count() if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    n_seen = 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # the GPU boxes export NCCL_DEBUG=VERSION: RCCL then prints a five-line banner on STDOUT of every rank (at WARN as
        # well), next to the one JSON line the caller parses
        os.environ.pop("NCCL_DEBUG", None)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
        ones = torch.ones(1, dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(ones)                       # every rank that takes part adds one
        n_seen = int(ones.item())

    from alignsdf_amd.hip_decoder import HipSdfDecoder
    from alignsdf_amd.marching_cubes import marching_cubes_device
    from alignsdf_amd.reconstruct import pipelined_two_pass
    from alignsdf_amd.utils.mesh import decode_two_pass
    from alignsdf_amd.utils.utils import sample_embedding

    N = args.grid
    specs = syn.specs_for(args.tag)
    dec = HipSdfDecoder(syn.full_state_dict(args.tag), 256, specs["PointFeatSize"], specs["EncodeStyle"], device=dev)
    if args.math is not None:
        dec.set_math(args.math)
    if args.coarse is not None:
        dec.coarse_mode = args.coarse
    if args.fine is not None:
        dec.fine_mode = args.fine
    # 64 distinct synthetic samples, resident on the device before timing
    codes = []
    for s in range(64):
        lat = torch.from_numpy(syn.latent_code(s)).to(dev)
        mano = obj = None
        if args.tag == "both9":
            m, o = syn.pose_inputs(s)
            mano = {k: torch.from_numpy(v).to(dev) for k, v in m.items()}
            obj = {k: torch.from_numpy(v).to(dev) for k, v in o.items()}
        codes.append((lat, mano, obj))

    def sample_id(i):
        return (rank * 7919 + i) % 64

    def sample_stream(first, count):
        for i in range(first, first + count):
            lat, mano, obj = codes[sample_id(i)]
            yield i, lat, mano, obj

    # the product's sample pipeline (alignsdf_amd.reconstruct.pipelined_two_pass): pass 1 of sample k+1 is queued
    # before the host-synchronous marching cubes of sample k; each call below runs `count` whole samples to completion
    def run(first, count):
        out = []
        for i, r in pipelined_two_pass(dec, specs, sample_stream(first, count), N):
            out.append((i, r["V_hand"], r["F_hand"], r["V_obj"], r["F_obj"], tuple(r["origin"]), float(r["voxel_size"])))
        return out

    def timed(first, count, warmup):
        """(elapsed seconds over `count` steps, per-launch ms of the decoder kernel, records, per-launch ms of the one-plane
        kernel if any ran) with the barrier + synchronize bracket of the bench contract on both sides."""
        run(0, warmup)
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        dec.event_log, dec.box_event_log = [], []
        t0 = time.perf_counter()
        done = run(first, count)
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        events, dec.event_log = dec.event_log, None
        box_events, dec.box_event_log = dec.box_event_log, None
        return elapsed, [e[0].elapsed_time(e[1]) for e in events], done, [e[0].elapsed_time(e[1]) for e in box_events]

    elapsed, k1_ms, done, main_box_ms = timed(args.warmup, args.steps, args.warmup)
    main_coarse = dec.coarse_mode if dec._box_usable() else "exact"
    main_fine = dec.fine_mode if dec._band_usable() else "exact"
    records = [dict(index=rank * args.steps + k, V_hand=d[1], F_hand=d[2], V_obj=d[3], F_obj=d[4], milliseconds=0.0)
               for k, d in enumerate(done)]
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        from alignsdf_amd.dist_reconstruct import gather_records
        merged = gather_records(records)            # the path's only collective: per-sample records -> rank 0
    else:
        merged = records

    # dominant kernel: the fused decoder (2 launches per step), timed with HIP events on the launch stream
    k1_avg_s = float(np.mean(k1_ms)) * 1e-3
    alg_flop = N ** 3 * 2 * FLOP_PER_POINT_HEAD
    split = dec.math == "f16x3"
    kernel_name = "sdf_mlp_f16_kernel" if split else "sdf_mlp_kernel"
    peak = PEAK_F16_MFMA_TFLOPS if split else PEAK_FP32_MFMA_TFLOPS
    exec_flop = N ** 3 * 2 * (EXEC_F16_FLOP_PER_POINT_HEAD if split else EXEC_FLOP_PER_POINT_HEAD)

    other = parity = mc_line = other_coarse = narrow_band = None
    if world == 1:
        last_sample = sample_id(args.warmup + args.steps - 1)
        lat, mano, obj = codes[last_sample]
        # ---- parity in the run: the last timed sample under both arithmetics, every voxel of both pass-2 volumes
        math0 = dec.math
        vols, counts = {}, {}
        for m in ("f16x3", "f32"):
            dec.set_math(m)
            r = decode_two_pass(True, True, dec, lat, mano, obj, specs, N)
            vols[m] = r
            counts[m] = []
            for part in ("hand", "obj"):
                try:
                    v, f = marching_cubes_device(r["vol_" + part], 0.0)
                    counts[m].append([int(v.shape[0]), int(f.shape[0])])
                except (ValueError, RuntimeError):
                    counts[m].append([0, 0])
        dec.set_math(math0)
        if len(vols) == 2:
            a, b = vols["f16x3"], vols["f32"]
            same_cube = bool(float(a["voxel_size"]) == float(b["voxel_size"]) and a["origin"] == b["origin"])
            parity = {
                "sample": last_sample, "zoom_cube_equal": same_cube,
                "max_abs_f16x3_minus_f32": max(float((a["vol_hand"] - b["vol_hand"]).abs().max()),
                                               float((a["vol_obj"] - b["vol_obj"]).abs().max())) if same_cube else None,
                "sign_differences": int(((a["vol_hand"] < 0) != (b["vol_hand"] < 0)).sum() + ((a["vol_obj"] < 0) != (b["vol_obj"] < 0)).sum()) if same_cube else None,
                "V_F_f16x3": counts["f16x3"], "V_F_f32": counts["f32"], "V_F_equal": counts["f16x3"] == counts["f32"],
                "voxels_compared": 2 * N ** 3, "bar": 1e-5,
            }
        # ---- the other arithmetic as a full record: same steps, same bracket
        if not args.no_other_math and len(vols) == 2:
            dec.set_math("f32" if split else "f16x3")
            e2, k2_ms, _, _ = timed(1, args.steps, 1)
            other = {"math": dec.math, "steps": args.steps, "warmup": 1, "ms_per_step": 1e3 * e2 / args.steps, "value": 2 * args.steps / e2,
                     "unit": "meshes/s", "launch_ms": float(np.mean(k2_ms)), "dtype": "f32" if split else "f32 as 2 x f16 planes"}
            dec.set_math(math0)
        # ---- the other coarse pass as a full record over the SAME samples: zoom cubes and surfaces must come out identical
        if not args.no_other_coarse and split and not dec.nerf_features:
            dec.coarse_mode = "exact" if main_coarse == "box" else "box"
            e3, k3_ms, done3, box3_ms = timed(args.warmup, args.steps, max(args.warmup, 2))
            same_cubes = sum(1 for a, b in zip(done, done3) if a[5] == b[5] and a[6] == b[6])
            same_vf = sum(1 for a, b in zip(done, done3) if a[1:5] == b[1:5])
            other_coarse = {"coarse": dec.coarse_mode if dec._box_usable() or dec.coarse_mode == "exact" else "exact (box switched itself off)",
                            "steps": args.steps, "ms_per_step": 1e3 * e3 / args.steps, "value": 2 * args.steps / e3, "unit": "meshes/s",
                            "launch_ms_pass2_kernel": float(np.mean(k3_ms)) if k3_ms else None,
                            "launch_ms_one_plane_kernel": float(np.mean(box3_ms)) if box3_ms else None,
                            "zoom_cubes_equal_to_main_run": "%d of %d" % (same_cubes, len(done)),
                            "V_F_equal_to_main_run": "%d of %d" % (same_vf, len(done)),
                            "box_stats": dict(dec.box_stats), "allowance": dec._box_tau}
            dec.coarse_mode = main_coarse
            # ---- both passes on the one-plane kernel (box-only coarse + narrow-band fine sweep), same samples
            if main_fine == "exact" and not dec.combined:
                dec.coarse_mode, dec.fine_mode = "box", "band"
                e4, k4_ms, done4, box4_ms = timed(args.warmup, args.steps, max(args.warmup, 2))
                same_cubes = sum(1 for a, b in zip(done, done4) if a[5] == b[5] and a[6] == b[6])
                same_vf = sum(1 for a, b in zip(done, done4) if a[1:5] == b[1:5])
                narrow_band = {"coarse": "box", "fine": "band" if dec._band_usable() else "exact (band switched itself off)",
                               "steps": args.steps, "ms_per_step": 1e3 * e4 / args.steps, "value": 2 * args.steps / e4, "unit": "meshes/s",
                               "launch_ms_one_plane_kernel": float(np.mean(box4_ms)) if box4_ms else None,
                               "zoom_cubes_equal_to_main_run": "%d of %d" % (same_cubes, len(done)),
                               "V_F_equal_to_main_run": "%d of %d" % (same_vf, len(done)),
                               "band_stats": dict(dec.band_stats), "box_stats": dict(dec.box_stats), "allowance": dec._box_tau}
                # the meshes of the band volumes against those of the ordinary sweeps (whose values the band re-evaluates to),
                # vertex for vertex, on the parity sample
                if "f16x3" in vols and dec._band_usable():
                    rb = decode_two_pass(True, True, dec, lat, mano, obj, specs, N, mc_only=True)
                    eq = []
                    for part in ("hand", "obj"):
                        vb, fb = marching_cubes_device(rb["vol_" + part], 0.0)
                        ve, fe = marching_cubes_device(vols["f16x3"]["vol_" + part], 0.0)
                        eq.append(bool(torch.equal(vb, ve) and torch.equal(fb, fe)))
                    narrow_band["meshes_bit_identical_to_ordinary_sweeps"] = eq
                    narrow_band["sign_differences_to_ordinary_sweeps"] = int(((rb["vol_hand"] < 0) != (vols["f16x3"]["vol_hand"] < 0)).sum() +
                                                                             ((rb["vol_obj"] < 0) != (vols["f16x3"]["vol_obj"] < 0)).sum())
                dec.coarse_mode, dec.fine_mode = main_coarse, main_fine
        # ---- marching cubes chain (K3-K6) on the last sample's volumes: HBM roofline of the second kernel family
        from alignsdf_amd import marching_cubes as mcmod
        r = vols.get(math0) or decode_two_pass(True, True, dec, lat, mano, obj, specs, N)
        if hasattr(mcmod, "time_chain"):
            mc_line = mcmod.time_chain(r["vol_hand"], r["vol_obj"], PEAK_HBM_GBS)

    # HBM traffic of the dominant kernel comes from PMC passes (rocprofv3 cannot run inside the timed region); the
    # committed summary of the last collection is reported only while it still describes this kernel: same grid, same
    # kernel name and the same source digest of the kernel headers
    traffic, traffic_src = None, None
    try:
        import hashlib
        digest = hashlib.sha256()
        for name in ("sdf_mlp_f16_kernel.h" if split else "sdf_mlp_kernel.h", "sdf_mlp_common.h", "sdf_layout.h"):
            with open(os.path.join(ROOT, "alignsdf_amd", "csrc", name), "rb") as f:
                digest.update(f.read())
        for cand in sorted([p for p in os.listdir(os.path.join(ROOT, "profiles")) if p.endswith(".json") and "hbm_traffic" in p], reverse=True):
            with open(os.path.join(ROOT, "profiles", cand)) as f:
                t = json.load(f)
            if t.get("grid") == N and t.get("kernel") == kernel_name and t.get("source_sha256", digest.hexdigest()) == digest.hexdigest():
                if "source_sha256" in t:
                    traffic, traffic_src = t["hbm_bytes_per_launch"], "profiles/" + cand
                    break
    except (OSError, ValueError, KeyError):
        pass

    if rank == 0:
        total_meshes = 2 * args.steps * world
        result = {
            "metric": "meshes_per_sec_hand_plus_obj_N%d" % N,
            "value": total_meshes / elapsed,
            "unit": "meshes/s",
            "n_gpus": n_seen,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 as 2 x f16 planes (3 x v_mfma_f32_32x32x16_f16 per product sum, fp32 accumulate)" if split else "f32",
            "data": "synthetic",
            "config": {
                "workload": "single sample, hand+object dual SDF decoder, N=%d grid, 2 passes (coarse + zoom cube) + HIP "
                            "marching cubes; %s decoder (PointFeatSize %d, EncodeStyle %s)" % (
                                N, "ObMan" if args.tag == "nerf3" else "DexYCB MANO-aligned", specs["PointFeatSize"],
                                specs["EncodeStyle"]),
                "grid": N, "samples_per_gpu": args.steps, "meshes_per_sample": 2,
                "parallelism": "sample-sharded x%d (one process per GPU, %s)" % (world, backend if world > 1 else "no collective"),
                "ranks_requested": args.gpus, "world_size_env": world,
                "samples_per_sec": args.steps * world / elapsed,
                "mesh_sizes_last_sample": merged[-1] if merged else None,
            },
            "roofline": {
                "bound": "mfma", "kernel": kernel_name,
                "achieved": alg_flop / k1_avg_s / 1e12, "peak": peak, "unit": "TFLOP/s",
                "frac": alg_flop / k1_avg_s / 1e12 / peak, "traffic": traffic, "traffic_source": traffic_src,
                "launch_ms": 1e3 * k1_avg_s, "launches_timed": len(k1_ms),
                "algorithmic_flop_per_launch": alg_flop,
                "executed_flop_per_launch": exec_flop,
                "achieved_executed": exec_flop / k1_avg_s / 1e12,
                "frac_executed": exec_flop / k1_avg_s / 1e12 / peak,
                "note": ("achieved counts the reference's dense fp32 FLOPs (1,573,888 per point per head) against the fp16 MFMA "
                         "peak, because that is the instruction issued: each fp32 product sum is carried as two fp16 planes per "
                         "operand and costs three v_mfma_f32_32x32x16_f16 (%d MFMA FLOPs per point per head, plus %d on the fp32 "
                         "MFMA for the point features); frac_executed is the matrix-pipe utilisation against the 2.5 PFLOP/s "
                         "spec.  On these operands the part is power-limited (it holds ~1.85-1.95 GHz under this kernel; the same "
                         "binary on all-zero operands runs at 2.36 GHz and 21 %% faster: profiles/r02_k1h_power_limit.txt)" % (
                             EXEC_F16_FLOP_PER_POINT_HEAD, EXEC_F16_SIDE_F32_FLOP_PER_POINT_HEAD)) if split else (
                    "achieved counts the reference's dense FLOPs (1,573,888 per point per head); the kernel folds the "
                    "per-sample-constant latent columns into a bias and issues %d, so frac can exceed 1; "
                    "frac_executed is the MFMA pipe utilisation" % EXEC_FLOP_PER_POINT_HEAD),
            },
        }
        if split:
            result["roofline"]["fp32_mfma_equivalent"] = {
                "note": "the same algorithmic FLOPs against the fp32 MFMA peak the reference arithmetic would be priced at",
                "peak": PEAK_FP32_MFMA_TFLOPS, "frac": alg_flop / k1_avg_s / 1e12 / PEAK_FP32_MFMA_TFLOPS}
        if mc_line is not None:
            result["roofline_marching_cubes"] = mc_line
        if parity is not None:
            result["parity_in_run"] = parity
        if other is not None:
            result["other_math"] = other
        result["config"]["coarse_pass"] = main_coarse
        result["config"]["fine_pass"] = main_fine
        if main_box_ms:
            result["roofline"]["launch_ms_one_plane_kernel"] = float(np.mean(main_box_ms))
        if other_coarse is not None:
            result["other_coarse_pass"] = other_coarse
        if narrow_band is not None:
            result["one_plane_sweeps"] = narrow_band
        if world == 1 and not args.no_cpu_baseline:
            # one extra GPU sample (outside every timed region) with its pass-1 volumes kept for the CPU zoom-cube leg
            sid = 0
            lat, mano, obj = codes[sid]
            dec.set_sample(lat, sample_embedding(specs, mano, obj, dec.combined))
            v1h, v1o, _ = dec.decode_grid(N, [-1.0, -1.0, -1.0], 2.0 / (N - 1))
            r = decode_two_pass(True, True, dec, lat, mano, obj, specs, N)
            gpu = {"sample": sid, "vol1_hand": v1h.cpu().numpy(), "vol1_obj": v1o.cpu().numpy(), "vol_hand": r["vol_hand"].cpu().numpy(),
                   "vol_obj": r["vol_obj"].cpu().numpy(), "voxel_size": r["voxel_size"], "origin": r["origin"]}
            for part in ("hand", "obj"):
                v, f = marching_cubes_device(r["vol_" + part], 0.0)
                gpu["V_" + part], gpu["F_" + part] = int(v.shape[0]), int(f.shape[0])
            result["cpu_baseline"] = cpu_baseline(args.tag, N, gpu)
            result["cpu_baseline"]["gpu_over_cpu"] = result["value"] / result["cpu_baseline"]["value"]
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
