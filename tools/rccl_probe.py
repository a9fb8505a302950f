"""Checks that torch.distributed over RCCL initialises on this box with the pattern bench.py uses (one rank)."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
t = torch.tensor([3.5], dtype=torch.float64, device="cuda:0")
dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier(); torch.cuda.synchronize()
g = [torch.empty(4, dtype=torch.float64, device="cuda:0")]
dist.all_gather(g, torch.arange(4, dtype=torch.float64, device="cuda:0"))
print("rccl ok", t.item(), g[0].tolist()); dist.destroy_process_group()
