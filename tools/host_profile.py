#!/usr/bin/env python3
"""Where the host time of a small eager step goes: cProfile over fwd + bwd + SGD steps of 2048 unique ids through the class
(GPU box; the kernels of such a step take ~30 us, the step ~100)."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "falcon-ttdforgnns_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
from FBTT.tt_embeddings_ops import TTEmbeddingBag

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
N_EMB = 2449029
emb = TTEmbeddingBag(N_EMB, 100, [16, 16], [125, 140, 140], [4, 5, 5], sparse=True, weight_dist="normal", learning_rate=0.01).cuda()
rng = np.random.default_rng(0)
ids = torch.from_numpy(rng.choice(N_EMB, size=n, replace=False).astype(np.int64)).cuda()
offs = torch.arange(n + 1, dtype=torch.int64, device="cuda")
d = torch.randn(n, 100, device="cuda")
for _ in range(50):
    emb(ids, offs).backward(d)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(steps):
    emb(ids, offs).backward(d)
torch.cuda.synchronize()
print(f"eager step: {(time.perf_counter() - t) / steps * 1e6:.1f} us")
t = time.perf_counter()
for _ in range(steps):
    out = emb(ids, offs)
torch.cuda.synchronize()
print(f"eager forward alone (graph node built, never run): {(time.perf_counter() - t) / steps * 1e6:.1f} us")
with torch.no_grad():
    t = time.perf_counter()
    for _ in range(steps):
        out = emb(ids, offs)
    torch.cuda.synchronize()
    print(f"forward under no_grad: {(time.perf_counter() - t) / steps * 1e6:.1f} us")
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    emb(ids, offs).backward(d)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
