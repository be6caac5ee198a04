#!/usr/bin/env python3
"""Does RCCL come up on this box?  torch.distributed's nccl backend, one all_reduce; with WORLD_SIZE > 1
(under torch.distributed.run) every rank uses cuda:0, which RCCL may refuse."""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
t0 = time.time()
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=rank, world_size=world)
x = torch.ones(8, device="cuda")
dist.all_reduce(x)
torch.cuda.synchronize()
print("rank %d of %d: all_reduce ok in %.1f s:" % (rank, world, time.time() - t0), x.tolist(), flush=True)
dist.destroy_process_group()
