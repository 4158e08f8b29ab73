"""RCCL smoke with one rank: the torch.distributed calls bench.py makes for N > 1 (init with device_id, barrier,
all_reduce MAX, all_gather_into_tensor, all_gather of int64), world_size 1 (a gpurun box has one GPU)."""
import os
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
dist.barrier()
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
shard = torch.randn((2, 64, 100, 7), device=dev)
full = torch.empty((1,) + tuple(shard.shape), device=dev)
dist.all_gather_into_tensor(full.view((2, 64, 100, 7)), shard)
mine = torch.tensor([7], dtype=torch.int64, device=dev)
sums = [torch.zeros_like(mine)]
dist.all_gather(sums, mine)
torch.cuda.synchronize()
assert torch.equal(full[0], shard) and int(sums[0].item()) == 7 and float(t.item()) == 1.5
print("rccl world-1 ok, backend", dist.get_backend())
dist.destroy_process_group()
