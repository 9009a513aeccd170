#!/usr/bin/env python3
"""cProfile of the host side of one small-batch step (where launch/Python overhead dominates)."""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "falcon-ttdforgnns_amd")):
    sys.path.insert(0, p)
import torch
from FBTT.tt_embeddings_ops import TTEmbeddingBag

mode = sys.argv[1] if len(sys.argv) > 1 else "sparse"       # sparse | dp (data-parallel path at world 1)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
emb = TTEmbeddingBag(2449029, 100, [16, 16], [125, 140, 140], [4, 5, 5], sparse=(mode == "sparse"), use_cache=False,
                     weight_dist="normal", learning_rate=0.01)
dp = None
if mode != "sparse":
    from ttemb_dist import TTDataParallel
    dp = TTDataParallel(emb)
ids = torch.randperm(2449029)[:n].cuda()
offs = torch.arange(n + 1).cuda()
d = torch.rand(n, 100, device="cuda")


def step():
    emb(ids, offs).backward(d)
    if dp is not None:
        dp.step(overlap=True)


for _ in range(50):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(500):
    step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
