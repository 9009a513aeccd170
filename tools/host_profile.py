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

emb = TTEmbeddingBag(2449029, 100, [16, 16], [125, 140, 140], [4, 5, 5], sparse=True, use_cache=False,
                     weight_dist="normal", learning_rate=0.01)
ids = torch.randperm(2449029)[:2048].cuda()
offs = torch.arange(2049).cuda()
d = torch.rand(2048, 100, device="cuda")
for _ in range(50):
    emb(ids, offs).backward(d)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(500):
    emb(ids, offs).backward(d)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
