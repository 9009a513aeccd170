#!/usr/bin/env python3
"""Host time per step (enqueue only) vs GPU time per step, sparse mode and the data-parallel path (world 1)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "falcon-ttdforgnns_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
from FBTT.tt_embeddings_ops import TTEmbeddingBag
from ttemb_dist import TTDataParallel

N = 409600
rng = np.random.default_rng(0)
ids = torch.from_numpy(rng.choice(2449029, size=N, replace=False).astype(np.int64)).cuda()
offs = torch.arange(N + 1, device="cuda")
d_out = (torch.rand(N, 100, device="cuda") - 0.5) * 0.1
for mode in ("sparse", "dp", "dp-overlap"):
    emb = TTEmbeddingBag(2449029, 100, [16, 16], [125, 140, 140], [4, 5, 5], sparse=(mode == "sparse"), use_cache=False,
                         weight_dist="normal", learning_rate=0.01)
    dp = TTDataParallel(emb) if mode != "sparse" else None
    def step():
        emb(ids, offs).backward(d_out)
        if dp is not None:
            dp.step(overlap=(mode == "dp-overlap"))
    for _ in range(10): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100): step()
    host = (time.perf_counter() - t0) / 100 * 1e6
    if dp is not None: dp.flush()
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) / 100 * 1e6
    print(f"{mode:10s}: host enqueue {host:6.1f} us/step, wall {total:6.1f} us/step", flush=True)
