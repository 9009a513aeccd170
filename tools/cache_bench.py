#!/usr/bin/env python3
"""Cache-on configuration (BASELINE.json configs[2], SURVEY.md §8d cfg-C): ogbn-products shapes, LFU cache of
10 % of the rows, frontiers made of 200-id windows whose starts follow a Zipf law (hot regions recur), one
warm-up epoch, cache_populate(), then timed fwd + bwd + SGD steps.  Prints hit rate and step times with the
cache off / in warm-up / live."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "falcon-ttdforgnns_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
from FBTT.tt_embeddings_ops import TTEmbeddingBag

N_EMB, D, N = 2449029, 100, 409600
P, Q, R = [125, 140, 140], [4, 5, 5], [16, 16]
rng = np.random.default_rng(4)
n_win = N_EMB // 200
zipf_p = 1.0 / np.arange(1, n_win + 1) ** 1.05
zipf_p /= zipf_p.sum()
perm_win = rng.permutation(n_win)


def frontier():
    starts = perm_win[rng.choice(n_win, size=N // 200, replace=False, p=zipf_p)] * 200
    return torch.from_numpy((starts[:, None] + np.arange(200)[None, :]).reshape(-1).astype(np.int64)).cuda()


def timed(emb, batches, d_out, offs):
    for b in batches[:3]:
        emb(b, offs).backward(d_out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in batches:
        emb(b, offs).backward(d_out)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / len(batches) * 1e3


def main():
    offs = torch.arange(N + 1, device="cuda")
    d_out = (torch.rand(N, D, device="cuda") - 0.5) * 0.1
    batches = [frontier() for _ in range(24)]
    plain = TTEmbeddingBag(N_EMB, D, R, P, Q, sparse=True, use_cache=False, weight_dist="normal")
    print(f"cache off        : {timed(plain, batches, d_out, offs):.3f} ms/step", flush=True)
    emb = TTEmbeddingBag(N_EMB, D, R, P, Q, sparse=True, use_cache=True, cache_size=int(0.1 * N_EMB),
                         hashtbl_size=N_EMB, weight_dist="normal")
    print(f"cache warm-up    : {timed(emb, batches, d_out, offs):.3f} ms/step (ids counted, nothing served yet)", flush=True)
    for _ in range(72):  # rest of a 96-step epoch
        emb.update_cache(frontier())
    t0 = time.perf_counter()
    emb.cache_populate()
    torch.cuda.synchronize()
    print(f"cache_populate   : {(time.perf_counter() - t0) * 1e3:.2f} ms for {emb.cache_weight.shape[0]} rows", flush=True)
    test = [frontier() for _ in range(24)]
    hits = []
    for b in test[:4]:
        slot_keys = emb.hashtbl
        cached = torch.isin(b, slot_keys[emb.cache_state >= 0])
        hits.append(float(cached.float().mean()))
    print(f"cache live       : {timed(emb, test, d_out, offs):.3f} ms/step, hit rate {np.mean(hits) * 100:.1f} %", flush=True)
    emb._fused_probe = lambda: False
    print(f"cache live, update + preprocess as two calls: {timed(emb, test, d_out, offs):.3f} ms/step", flush=True)


if __name__ == "__main__":
    main()
