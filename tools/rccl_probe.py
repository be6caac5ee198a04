#!/usr/bin/env python3
"""Does RCCL come up on this box at all?  One rank, torch.distributed's nccl backend, one all_reduce."""
import os
import sys
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
t0 = time.time()
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
x = torch.ones(8, device="cuda")
dist.all_reduce(x)
torch.cuda.synchronize()
print("rccl one-rank all_reduce ok in %.1f s:" % (time.time() - t0), x.tolist(), flush=True)
dist.destroy_process_group()
