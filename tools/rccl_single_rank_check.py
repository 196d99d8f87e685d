"""The RCCL leg of the multi-GPU path on ONE MI355X: a world-size-1 process group on backend "nccl" (= RCCL on ROCm) through
every collective the product and the benchmark issue - the all_reduce of ones that counts the ranks, the MAX all_reduce of the
elapsed time, the barrier, and dist_reconstruct.gather_records (all_gather of counts + padded gather of [n, 7] fp64 records).
An 8-GPU node is the driver's to launch; this proves the code path executes on the hardware backend.   python tools/rccl_single_rank_check.py
"""
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.pop("NCCL_DEBUG", None)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
from alignsdf_amd.dist_reconstruct import gather_records, limit_host_threads  # noqa: E402

threads = limit_host_threads(1)
ones = torch.ones(1, dtype=torch.int32, device="cuda")
dist.all_reduce(ones)
t = torch.tensor([1.25], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
recs = [dict(index=3, V_hand=10, F_hand=20, V_obj=5, F_obj=6, milliseconds=1.5, icp_skipped=1),
        dict(index=1, V_hand=1, F_hand=2, V_obj=3, F_obj=4, milliseconds=0.5)]
merged = gather_records(recs)
empty = gather_records([])
backend = dist.get_backend()
dist.barrier()
dist.destroy_process_group()
print(json.dumps({"rccl_ok": True, "backend": backend, "n_ranks": int(ones.item()), "max_elapsed": t.item(), "merged": merged,
                  "empty": empty, "host_threads": threads, "device": torch.cuda.get_device_name(0)}))
