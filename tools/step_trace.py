#!/usr/bin/env python3
"""Per-step host enqueue time and GPU completion interval of the bench.py step, from cold: shows where the
first steps of a run are host-bound (allocator / launch-path warm-up) and where the GPU takes over."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "falcon-ttdforgnns_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
from FBTT.tt_embeddings_ops import TTEmbeddingBag

N = int(sys.argv[1]) if len(sys.argv) > 1 else 409600
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
emb = TTEmbeddingBag(2449029, 100, [16, 16], [125, 140, 140], [4, 5, 5], sparse=True, use_cache=False,
                     weight_dist="normal", learning_rate=0.01, batch_count=N)
rng = np.random.default_rng(2)
ids = [torch.from_numpy(rng.choice(2449029, size=N, replace=False).astype(np.int64)).cuda() for _ in range(4)]
offs = torch.arange(N + 1, device="cuda")
d = (torch.rand(N, 100, device="cuda") - 0.5) * 0.1
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
host = np.zeros(steps + 1)
ev[0].record()
host[0] = time.perf_counter()
for i in range(steps):
    emb(ids[i % 4], offs).backward(d)
    ev[i + 1].record()
    host[i + 1] = time.perf_counter()
torch.cuda.synchronize()
gpu = np.array([ev[0].elapsed_time(e) for e in ev]) * 1e3   # us since the start
host = (host - host[0]) * 1e6
w = 25
print(f"{N} ids, {steps} steps; per window of {w} steps: host enqueue us/step | GPU completion us/step | GPU lag behind host (us)")
for a in range(0, steps, w):
    b = min(a + w, steps)
    print(f"  steps {a:4d}-{b:4d}: {(host[b] - host[a]) / (b - a):7.1f} | {(gpu[b] - gpu[a]) / (b - a):7.1f} | {gpu[b] - host[b]:9.1f}")
