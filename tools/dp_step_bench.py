#!/usr/bin/env python3
"""usage (GPU box): [TTEMB_LIB=...] python3 tools/dp_step_bench.py [--ids 409600] [--dist uniform|windows]
The data-parallel step at world size 1 (bench.py's dp_mode_1gpu leg alone): sparse=False, gradients into the bucket,
TTDataParallel.step(overlap=True), guarded SGD over the flat weights -- per-step GPU times by HIP events."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "falcon-ttdforgnns_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
from FBTT.tt_embeddings_ops import TTEmbeddingBag
from ttemb_dist import TTDataParallel

ap = argparse.ArgumentParser()
ap.add_argument("--ids", type=int, default=409600)
ap.add_argument("--dist", default="uniform", choices=["uniform", "windows"])
ap.add_argument("--steps", type=int, default=200)
a = ap.parse_args()
P, Q, R, N_EMB, D, N = [125, 140, 140], [4, 5, 5], [16, 16], 2449029, 100, a.ids
rng = np.random.default_rng(2)
emb = TTEmbeddingBag(N_EMB, D, R, P, Q, sparse=False, use_cache=False, weight_dist="normal", learning_rate=0.01, batch_count=N)
dp = TTDataParallel(emb)


def frontier():
    if a.dist == "uniform":
        return rng.choice(N_EMB, size=N, replace=False)
    st = rng.choice(N_EMB // 200 - 1, size=(N + 199) // 200, replace=False) * 200
    return (st[:, None] + np.arange(200)[None, :]).reshape(-1)[:N]


sets = [torch.from_numpy(frontier().astype(np.int64)).cuda() for _ in range(4)]
offs = torch.arange(N + 1, dtype=torch.int64, device="cuda")
d_out = (torch.rand(N, D, device="cuda") - 0.5) * 0.1


def step(i):
    emb(sets[i % 4], offs).backward(d_out)
    dp.step(overlap=True)


for i in range(50):
    step(i)
dp.flush()
torch.cuda.synchronize()
evs = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
evs[0].record()
for i in range(a.steps):
    step(i)
    evs[i + 1].record()
dp.flush()
torch.cuda.synchronize()
per = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(a.steps)])
print(f"dp step, {a.dist} {N} ids: median {np.median(per) * 1e3:.1f} us  min {per.min() * 1e3:.1f}  p90 {np.quantile(per, 0.9) * 1e3:.1f}  "
      f"mean {evs[0].elapsed_time(evs[a.steps]) / a.steps * 1e3:.1f} us")
