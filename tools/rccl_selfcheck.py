#!/usr/bin/env python3
"""usage (GPU box): python3 tools/rccl_selfcheck.py
The data-parallel step over a REAL RCCL process group at the only world size one GPU allows (1; RCCL refuses two ranks on one
device): backend "nccl" initialises, the flat-bucket all-reduce / broadcast / barrier / all_gather calls of bench.py's N > 1
path go through RCCL, and the step's result equals the fused in-backward update.  Not a scaling measurement."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "falcon-ttdforgnns_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from FBTT.tt_embeddings_ops import TTEmbeddingBag
from ttemb_dist import TTDataParallel

P, Q, R, N_EMB, D, N = [125, 140, 140], [4, 5, 5], [16, 16], 2449029, 100, 65536
torch.manual_seed(3)
a = TTEmbeddingBag(N_EMB, D, R, P, Q, sparse=False, use_cache=False, weight_dist="normal", learning_rate=0.05)
b = TTEmbeddingBag(N_EMB, D, R, P, Q, sparse=True, use_cache=False, weight_dist="normal", learning_rate=0.05)
for ca, cb in zip(a.tt_cores, b.tt_cores):
    ca.data.mul_(300.0)
    cb.data.copy_(ca.data)
dp = TTDataParallel(a)
dp.broadcast_parameters(0)
rng = np.random.default_rng(0)
offs = torch.arange(N + 1, device="cuda")
for step in range(3):
    ids = torch.from_numpy(rng.choice(N_EMB, size=N, replace=False).astype(np.int64)).cuda()
    d = (torch.rand(N, D, device="cuda") - 0.5) * 0.1
    a(ids, offs).backward(d)
    dp.step(overlap=True)
    b(ids, offs).backward(d)
dp.flush()
dist.barrier()
t = torch.tensor([1.0], dtype=torch.float64, device="cuda")
every = [torch.zeros_like(t)]
dist.all_gather(every, t)
torch.cuda.synchronize()
err = max(float((ca.data - cb.data).abs().max()) for ca, cb in zip(a.tt_cores, b.tt_cores))
scale = max(float(cb.data.abs().max()) for cb in b.tt_cores)
print(f"backend {dist.get_backend()} world {dist.get_world_size()}: 3 data-parallel steps over RCCL; max |dense-bucket step - fused step| = {err:.3e} (weights up to {scale:.2f})")
assert err <= 1e-5 * max(scale, 1.0) + 1e-6
dist.destroy_process_group()
print("ok")
