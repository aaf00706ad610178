#!/usr/bin/env python3
"""tools/nccl_probe.py — single-rank RCCL smoke of the collectives bench.py uses at N > 1 (init with device_id, gather,
all_gather_into_tensor, all_reduce MAX, barrier). Development aid: the 1-GPU box cannot run the real N > 1 path."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29655")
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
slab = torch.arange(1024, dtype=torch.float32, device=dev)
out = [torch.empty_like(slab)]
dist.gather(slab, out, dst=0)
assert torch.equal(out[0], slab)
allg = torch.empty(1024, dtype=torch.float32, device=dev)
dist.all_gather_into_tensor(allg, slab)
assert torch.equal(allg, slab)
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
print("nccl probe ok", float(t.item()))
dist.destroy_process_group()
