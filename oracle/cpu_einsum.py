"""CPU baseline for bench.py: the TT lookup as gather + einsum in plain PyTorch.

TEST / MEASUREMENT INFRASTRUCTURE ONLY (see oracle/tt_oracle.py's header).  The
reference has no CPU implementation of this path; this is the
"pure-PyTorch einsum" port BASELINE.json asks to be timed on the host cores of the
GPU box.  It is checked against oracle/tt_oracle.py in tests/test_oracle_golden.py.
"""
from __future__ import annotations

import time

import torch


def einsum_rows(indices: torch.Tensor, cores, p, q, R) -> torch.Tensor:
    """[n, D] rows of a 3-core table: index split, index_select, one einsum."""
    assert len(p) == 3
    i0 = indices // (p[1] * p[2])
    rem = indices - i0 * (p[1] * p[2])
    i1 = rem // p[2]
    i2 = rem - i1 * p[2]
    n = indices.numel()
    g0 = cores[0].index_select(0, i0).view(n, q[0], R[1])
    g1 = cores[1].index_select(0, i1).view(n, R[1], q[1], R[2])
    g2 = cores[2].index_select(0, i2).view(n, R[2], q[2])
    return torch.einsum("bia,bajc,bck->bijk", g0, g1, g2).reshape(n, -1)


def train_step(indices, d_out, cores, p, q, R, lr):
    """forward + autograd backward + SGD on the cores (what one hot-path step does)."""
    for c in cores:
        c.grad = None
    out = einsum_rows(indices, cores, p, q, R)
    out.backward(d_out)
    with torch.no_grad():
        for c in cores:
            c -= lr * c.grad
    return out


def time_baseline(p, q, R, n_ids, n_emb, seed, budget_s=15.0, threads=None, train=True):
    """lookups/s of the einsum port on the host cores, bounded to ~budget_s seconds."""
    if threads:
        torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(seed)
    cores = [(torch.randn(p[t], R[t] * q[t] * R[t + 1], generator=g) * 0.3).requires_grad_(train)
             for t in range(3)]
    ids = torch.randperm(n_emb, generator=g)[:n_ids]
    d_out = torch.rand(n_ids, q[0] * q[1] * q[2], generator=g) * 0.1
    step = (lambda: train_step(ids, d_out, cores, p, q, R, 0.01)) if train else \
           (lambda: einsum_rows(ids, cores, p, q, R))
    step()  # warm-up
    t0 = time.perf_counter()
    iters = 0
    while True:
        step()
        iters += 1
        el = time.perf_counter() - t0
        if el > budget_s or iters >= 50:
            break
    return {"lookups_per_s": n_ids * iters / el, "iters": iters, "seconds": el,
            "threads": torch.get_num_threads()}
