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


# ---------------------------------------------------------------------------------------------
# Initialisers the GNN drivers apply on top of the layer (tt_utils.py:117-201, selected by
# `--init ortho|dortho|eigen` in gnn_model.py:127-181).  Written for any 2..4-core shape (the reference
# hard-codes 3 cores and q = [4, 5, 5]) and as torch linear algebra on whatever device the caller asks
# for, so that the 2.45 M x 100 decomposition runs on the MI355X instead of in numpy.
# ---------------------------------------------------------------------------------------------


def _full_ranks(tt_ranks, T):
    r = [int(x) for x in tt_ranks]
    if len(r) == T - 1:
        r = [1] + r + [1]
    assert len(r) == T + 1 and r[0] == 1 and r[-1] == 1, "tt_ranks must be the T-1 inner ranks or [1, ..., 1]"
    return r


def _rows_from_core4(core4: torch.Tensor) -> torch.Tensor:
    """[R, p, q, R'] -> [1, p, R*q*R'] (storage layout of `tt_cores`)."""
    R, p, q, R2 = core4.shape
    return core4.permute(1, 0, 2, 3).reshape(1, p, R * q * R2).contiguous()


def ortho_cores(tt_ranks, tt_p_shapes, tt_q_shapes, device="cpu", generator=None):
    """Cores whose (r_in, q) slices are orthonormal vectors over (p, r_out)  (`get_ortho`, tt_utils.py:117-157).

    For core t the reference takes the first R_t*q_t rows of the Q factor of a random
    (p_t R_{t+1}) x (p_t R_{t+1}) matrix; an orthonormal R_t*q_t-frame of that space has the same
    distribution and only needs the reduced QR of a tall (p_t R_{t+1}) x (R_t q_t) matrix.
    Returns a list of float32 tensors [1, p_t, R_t q_t R_{t+1}].
    """
    T = len(tt_p_shapes)
    R = _full_ranks(tt_ranks, T)
    out = []
    for t in range(T):
        p, q = int(tt_p_shapes[t]), int(tt_q_shapes[t])
        n, k = p * R[t + 1], R[t] * q
        if k > n:
            raise ValueError(f"core {t}: needs {k} orthonormal vectors in a {n}-dimensional space")
        m = torch.randn(n, k, device=device, dtype=torch.float32, generator=generator)
        frame, _ = torch.linalg.qr(m, mode="reduced")            # n x k, orthonormal columns
        core4 = frame.t().reshape(R[t], q, p, R[t + 1]).permute(0, 2, 1, 3)   # [R, p, q, R']
        out.append(_rows_from_core4(core4))
    return out


def tt_svd_cores(matrix, tt_ranks, tt_p_shapes, tt_q_shapes):
    """TT-SVD of a dense [prod(p), prod(q)] table into cores  (`tt_matrix_decomp`, tt_utils.py:159-201).

    The table is viewed as a tensor with modes (p_t q_t); each step takes the rank-truncated SVD of the
    current unfolding, keeps U as the core and carries diag(s)·V^T forward.  Runs where `matrix` lives.
    Returns (cores, ranks) with cores float32 [1, p_t, R_t q_t R_{t+1}] and the ranks actually used.
    """
    T = len(tt_p_shapes)
    want = _full_ranks(tt_ranks, T)
    p = [int(x) for x in tt_p_shapes]
    q = [int(x) for x in tt_q_shapes]
    x = torch.as_tensor(matrix, dtype=torch.float32)
    assert x.shape == (int(np.prod(p)), int(np.prod(q))), "matrix must be [prod(p), prod(q)]"
    order = [i for t in range(T) for i in (t, T + t)]
    x = x.reshape(p + q).permute(order)
    ranks = [1] * (T + 1)
    cores = []
    for t in range(T - 1):
        rows = ranks[t] * p[t] * q[t]
        x = x.reshape(rows, -1)
        ranks[t + 1] = 1 if want[t + 1] == 1 else min(want[t + 1], rows, x.shape[1])
        if x.shape[1] > 4 * rows:
            # wide unfolding: eigen-decompose the small Gram matrix instead of a full SVD
            evals, u = torch.linalg.eigh((x @ x.t()).double())
            u = u[:, -ranks[t + 1]:].flip(1).float()
            x_next = u.t() @ x
        else:
            u, s, vh = torch.linalg.svd(x, full_matrices=False)
            u = u[:, :ranks[t + 1]]
            x_next = s[:ranks[t + 1], None] * vh[:ranks[t + 1]]
        cores.append(_rows_from_core4(u.reshape(ranks[t], p[t], q[t], ranks[t + 1])))
        x = x_next
    cores.append(_rows_from_core4(x.reshape(ranks[T - 1], p[T - 1], q[T - 1], 1)))
    return cores, ranks


def prefix_locality(indices: torch.Tensor, tt_p_shapes):
    """How well a batch of ids shares TT prefixes (what METIS reordering buys).

    Returns a dict: ids, distinct ids, distinct prefixes (all index digits but the last), ids per
    prefix, and the fraction of first-stage products a prefix-sharing kernel skips.  Runs on the
    tensor's device.
    """
    ids = indices.reshape(-1).long()
    n = int(ids.numel())
    last = int(tt_p_shapes[-1])
    prefixes = int(torch.unique(torch.div(ids, last, rounding_mode="floor")).numel()) if n else 0
    distinct = int(torch.unique(ids).numel()) if n else 0
    return {
        "ids": n,
        "distinct_ids": distinct,
        "distinct_prefixes": prefixes,
        "ids_per_prefix": (n / prefixes) if prefixes else 0.0,
        "stage1_reuse": (1.0 - prefixes / n) if n else 0.0,
    }
