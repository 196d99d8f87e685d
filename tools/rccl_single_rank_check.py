import os, sys, torch, torch.distributed as dist
sys.path.insert(0, ".")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
torch.cuda.set_device(0)
os.environ.pop("NCCL_DEBUG", None)
dist.init_process_group("nccl", rank=0, world_size=1)
from alignsdf_amd.dist_reconstruct import gather_records, RECORD_FIELDS
recs = [dict(index=3, V_hand=10, F_hand=20, V_obj=5, F_obj=6, milliseconds=1.5), dict(index=1, V_hand=1, F_hand=2, V_obj=3, F_obj=4, milliseconds=0.5)]
t = torch.tensor([1.25], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
print("rccl ok:", gather_records(recs), t.item())
dist.destroy_process_group()
