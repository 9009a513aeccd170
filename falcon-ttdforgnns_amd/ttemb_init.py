"""Weight initialisers for the TT cores.

Same five distributions as the reference's ``reset_parameters``
(FBTT/tt_embeddings_ops.py:629-808) -- ``uniform``, ``naive-uniform``, ``normal``
(the one the GNN drivers use, gnn_model.py:123), ``approx-normal`` and
``approx-uniform`` -- written as array operations (the reference's
``approx-normal`` is a per-element Python rejection loop).  Random draws come from
numpy's / torch's global generators, so ``np.random.seed`` / ``torch.manual_seed``
make them reproducible.
"""
from __future__ import annotations

import numpy as np
import torch

DISTS = ("uniform", "naive-uniform", "normal", "approx-uniform", "approx-normal")


def _assign(param: torch.nn.Parameter, values: np.ndarray) -> None:
    param.data = torch.as_tensor(np.ascontiguousarray(values, dtype=np.float32)).to(param.device)


def _tail_normal(size, threshold: float = 2.0) -> np.ndarray:
    """N(0,1) conditioned on |x| >= threshold (vectorised rejection sampling)."""
    out = np.random.normal(size=size).astype(np.float32)
    flat = out.reshape(-1)
    todo = np.flatnonzero(np.abs(flat) < threshold)
    while todo.size:
        flat[todo] = np.random.normal(size=todo.size).astype(np.float32)
        todo = todo[np.abs(flat[todo]) < threshold]
    return out


def _saw_tooth(grid_points: int, width: float, n: int) -> np.ndarray:
    """Flat teeth of the given width centred on j/grid_points, j in (-grid_points, grid_points)."""
    centre = np.random.randint(-(grid_points - 1), grid_points, n) / float(grid_points)
    return centre + width * (np.random.rand(n) - 0.5)


def _to_row_layout(core4: np.ndarray) -> np.ndarray:
    """[R, p, q, R'] -> [1, p, R*q*R'] (the layer's storage layout)."""
    R, p, q, R2 = core4.shape
    return core4.transpose(1, 0, 2, 3).reshape(1, p, R * q * R2)


def _approx_uniform(mod) -> None:
    assert mod.tt_ndim == 3, "approx-uniform is defined for 3 cores"
    assert mod.num_tables == 1, "approx_uniform only supported for num_tables == 1"
    sigma, grid, width = 0.01, 15, 0.7 / 30.0
    scale = 1.0 / (np.sqrt(mod.num_embeddings) ** (1.0 / 3.0))
    dims = [[mod.tt_ranks[t], mod.tt_p_shapes[t], mod.tt_q_shapes[t], mod.tt_ranks[t + 1]] for t in range(3)]
    # head: everything close to 1/sqrt(r1)
    head = 1.0 / np.sqrt(dims[0][3]) + sigma * np.random.randn(*dims[0])
    # middle: close to 1/sqrt(r1); per (p,q) one even output-rank column is made tiny
    # except for one input-rank entry drawn from the saw tooth
    r_in, p1, q1, r_out = dims[1]
    base = 1.0 / np.sqrt(r_in)
    mid = (base + sigma * np.random.randn(*dims[1])).reshape(r_in, p1 * q1, r_out)
    cols = np.arange(p1 * q1)
    even = np.arange(0, r_out, 2)
    pick_out = even[np.random.randint(0, even.size, p1 * q1)]
    mid[:, cols, pick_out] = np.random.randn(r_in, p1 * q1) * (sigma * sigma / base)
    pick_in = np.random.randint(0, r_in, p1 * q1)
    mid[pick_in, cols, pick_out] = _saw_tooth(grid, width, p1 * q1) / base
    mid = mid.reshape(dims[1])
    # tail: small noise; per (p,q) one odd rank entry drawn from the saw tooth
    r3, p2, q2, _ = dims[2]
    tail = (sigma * np.random.randn(*dims[2])).reshape(r3, p2 * q2)
    odd = np.arange(1, r3, 2)
    if odd.size:
        cols2 = np.arange(p2 * q2)
        tail[odd[np.random.randint(0, odd.size, p2 * q2)], cols2] = _saw_tooth(grid, width, p2 * q2)
    tail = tail.reshape(dims[2])
    for t, core in enumerate((head, mid, tail)):
        _assign(mod.tt_cores[t], _to_row_layout(core * scale))


def init_cores(mod, weight_dist: str) -> None:
    assert weight_dist in DISTS, f"unknown weight_dist {weight_dist!r}"
    T, n, D = mod.tt_ndim, mod.num_embeddings, mod.embedding_dim
    with torch.no_grad():
        if weight_dist == "uniform":
            stddev = np.sqrt(2.0 / (n + D))
            rank_term = float(np.prod(np.array(mod.tt_ranks, dtype=np.float64) ** (-1.0 / (2 * T))))
            hi = stddev ** (1.0 / T) * rank_term
            for c in mod.tt_cores:
                c.uniform_(0.0, float(hi))
        elif weight_dist == "naive-uniform":
            for c in mod.tt_cores:
                c.uniform_(0.0, float(1.0 / np.sqrt(n)))
        elif weight_dist == "normal":
            for c in mod.tt_cores:
                c.normal_(0.0, float(1.0 / np.sqrt(n)))
                c.mul_(1.0 / mod.tt_ranks[0])
        elif weight_dist == "approx-normal":
            scale = np.power(1.0 / np.sqrt(3.0 * n), 1.0 / 3.0)
            for c in mod.tt_cores:
                _assign(c, _tail_normal(tuple(c.shape)) * np.float32(scale))
        else:
            _approx_uniform(mod)
