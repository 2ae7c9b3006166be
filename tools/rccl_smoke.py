#!/usr/bin/env python3
"""One-rank RCCL smoke test of the collectives the data-parallel train step uses (init with device_id, barrier,
broadcast, async averaging all-reduce of a flat gradient buffer, MAX / SUM all-reduce of the step statistics).
A 1-GPU box cannot run world_size > 1 on RCCL; this checks that the calls themselves work on this image.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 tools/rccl_smoke.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from histogan_amd import ddp  # noqa: E402
from histogan_amd.optim import FlatParams  # noqa: E402

rank, world, local = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1)), int(os.environ.get('LOCAL_RANK', 0))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29533')
dev = torch.device('cuda', local)
torch.cuda.set_device(dev)
dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
ddp.is_dist = lambda: True                     # exercise the collective code paths even at world size 1 (incl. ReduceOp.AVG)
ps = [torch.nn.Parameter(torch.randn(1000, 1000, device=dev)), torch.nn.Parameter(torch.randn(77, device=dev))]
flat = FlatParams(ps)
ddp.broadcast_flat(flat)
flat.zero_grad()
for p in ps:
    p.grad = torch.ones_like(p) * (rank + 1)
red = ddp.GradAllReduce(flat, chunks=2)
red.start()
x = torch.randn(4096, 4096, device=dev) @ torch.randn(4096, 4096, device=dev)     # compute overlapping the collective
red.finish()
torch.cuda.synchronize()
want = sum(r + 1 for r in range(world)) / world
assert abs(float(flat.grad.mean()) - want) < 1e-6, float(flat.grad.mean())
stats = torch.tensor([1.0, 2.0, 0.0], device=dev, dtype=torch.float64)
dist.all_reduce(stats[:2], op=dist.ReduceOp.SUM)
dist.all_reduce(stats[2:], op=dist.ReduceOp.MAX)
dist.barrier()
assert ddp.all_reduce_scalar(3.0, 'mean', dev) == 3.0
dist.destroy_process_group()
print('rccl smoke ok: world', world, 'backend nccl, x', float(x[0, 0]) != 0)
