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
