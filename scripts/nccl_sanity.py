"""RCCL API sanity on ONE GPU (world_size 1): the exact calls bench.py / pybo_amd.dist make for N > 1."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
dist.init_process_group('nccl', device_id=dev)
mine = torch.arange(20, dtype=torch.float64, device=dev)
out = torch.empty(20, dtype=torch.float64, device=dev)
dist.all_gather_into_tensor(out, mine)
dist.barrier()
torch.cuda.synchronize(dev)
assert torch.equal(out, mine)
te = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(te, op=dist.ReduceOp.MAX)
print('rccl ok', out[:3].tolist(), te.item(), dist.get_backend())
dist.destroy_process_group()
