#!/usr/bin/env python3
"""num_tables > 1 through the class: per-table window calls (no host synchronisation) against the host-side split of the id list."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "falcon-ttdforgnns_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
from FBTT.tt_embeddings_ops import TableBatchedTTEmbeddingBag

T = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n_ids = int(sys.argv[2]) if len(sys.argv) > 2 else 409600
N_EMB, D = 2449029, 100
rng = np.random.default_rng(0)
B = n_ids // T
ids = torch.from_numpy(rng.integers(0, N_EMB, size=T * B).astype(np.int64)).cuda()
offs = torch.arange(T * B + 1, dtype=torch.int64, device="cuda")
d = torch.randn(T, B, D, device="cuda")
for sparse in (True, False):
    for windows in (True, False):
        emb = TableBatchedTTEmbeddingBag(T, N_EMB, D, [16, 16], [125, 140, 140], [4, 5, 5], sparse=sparse, use_cache=False,
                                         weight_dist="normal", learning_rate=0.01).cuda()
        emb._use_windows = windows
        for _ in range(5):
            emb(ids, offs).backward(d)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(30):
            emb(ids, offs).backward(d)
        torch.cuda.synchronize()
        print(f"T={T} ids={T * B} sparse={sparse} {'windows (no host sync)' if windows else 'host-side split      '}: "
              f"{(time.perf_counter() - t) / 30 * 1e3:.3f} ms per fwd+bwd", flush=True)
