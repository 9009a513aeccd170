"""Module-level drop-in for the reference's compiled extension ``tt_embeddings``
(the 11 functions of FBTT/tt_embeddings.cpp:131-161), so code written against the
extension API keeps working.  Same names, argument order, return values and error
type (``RuntimeError``); the work is done by ``libttemb_hip.so`` through the C ABI.

Differences that do not change results: ``batch_count`` is validated (``> 0``) and
otherwise ignored -- there are no HBM-resident partial products to chunk; scratch
comes from one cached workspace tensor per device instead of per-call ``at::empty``.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch

import ttemb_native as _nat

_WS = {}


def _ws(dev: torch.device) -> _nat.Workspace:
    return _WS.setdefault((dev.type, dev.index), _nat.Workspace())


def _need(cond: bool, msg: str) -> None:
    if not cond:
        raise RuntimeError(msg)


def _table_bounds(tableidx: torch.Tensor, nnz: int, num_tables: int) -> List[int]:
    if num_tables == 1:
        return [0, nnz]
    keys = torch.arange(num_tables + 1, device=tableidx.device, dtype=tableidx.dtype)
    return torch.searchsorted(tableidx[:nnz].contiguous(), keys).tolist()


def tt_forward(batch_count: int, num_tables: int, B: int, D: int, tt_p_shapes: Sequence[int],
               tt_q_shapes: Sequence[int], tt_ranks: Sequence[int], L: torch.Tensor, nnz: int,
               indices: torch.Tensor, rowidx: torch.Tensor, tableidx: torch.Tensor,
               tt_cores: Sequence[torch.Tensor]) -> torch.Tensor:
    """[num_tables, B, D] bag sums of the first ``nnz`` ids (tt_embeddings_cuda.cu:967-1081)."""
    dev = tt_cores[0].device
    out = torch.zeros((num_tables, B, D), dtype=torch.float32, device=dev)
    if nnz == 0:
        return out
    _need(batch_count > 0, "batch_count must be positive")
    _need(D > 0 and D % 4 == 0, "embedding_dim must be a positive multiple of 4")
    shape = _nat.make_shape(tt_p_shapes, tt_q_shapes, tt_ranks)
    bounds = _table_bounds(tableidx, nnz, num_tables)
    for k in range(num_tables):
        lo, hi = bounds[k], bounds[k + 1]
        if hi > lo:
            _nat.forward(shape, _nat.core_views(tt_cores, k), indices[lo:hi], rowidx[lo:hi], None, hi - lo,
                         None, B, out[k], _ws(dev))
    return out


def tt_dense_backward(batch_count: int, D: int, tt_p_shapes, tt_q_shapes, tt_ranks, L, nnz: int, indices,
                      rowidx, tableidx, d_output: torch.Tensor, tt_cores) -> List[torch.Tensor]:
    """Fresh ``d_core_t`` tensors shaped like the cores (tt_embeddings_cuda.cu:656-686)."""
    grads = [torch.zeros_like(c) for c in tt_cores]
    if nnz == 0:
        return grads
    shape = _nat.make_shape(tt_p_shapes, tt_q_shapes, tt_ranks)
    num_tables, B = tt_cores[0].shape[0], d_output.shape[-2]
    d_output = d_output.contiguous().view(num_tables, B, D)
    bounds = _table_bounds(tableidx, nnz, num_tables)
    for k in range(num_tables):
        lo, hi = bounds[k], bounds[k + 1]
        if hi > lo:
            _nat.backward_dense(shape, _nat.core_views(tt_cores, k), indices[lo:hi], rowidx[lo:hi], hi - lo,
                                None, B, d_output[k], _nat.core_views(grads, k), _ws(d_output.device))
    return grads


def tt_sgd_backward(batch_count: int, D: int, learning_rate: float, tt_p_shapes, tt_q_shapes, tt_ranks, L,
                    nnz: int, indices, rowidx, tableidx, d_output: torch.Tensor, tt_cores) -> None:
    """In-place fused SGD on the cores (tt_embeddings_cuda.cu:688-719)."""
    if nnz == 0:
        return
    shape = _nat.make_shape(tt_p_shapes, tt_q_shapes, tt_ranks)
    num_tables, B = tt_cores[0].shape[0], d_output.shape[-2]
    d_output = d_output.contiguous().view(num_tables, B, D)
    bounds = _table_bounds(tableidx, nnz, num_tables)
    for k in range(num_tables):
        lo, hi = bounds[k], bounds[k + 1]
        if hi > lo:
            _nat.backward_sgd(shape, _nat.core_views(tt_cores, k), indices[lo:hi], rowidx[lo:hi], hi - lo, None,
                              B, d_output[k], float(learning_rate), _ws(d_output.device))


def tt_adagrad_backward(batch_count: int, D: int, learning_rate: float, eps: float, tt_p_shapes, tt_q_shapes,
                        tt_ranks, L, nnz: int, indices, rowidx, tableidx, d_output: torch.Tensor,
                        optimizer_state, tt_cores) -> None:
    """In-place fused Adagrad on cores and state (tt_embeddings_cuda.cu:721-754)."""
    if nnz == 0:
        return
    shape = _nat.make_shape(tt_p_shapes, tt_q_shapes, tt_ranks)
    num_tables, B = tt_cores[0].shape[0], d_output.shape[-2]
    d_output = d_output.contiguous().view(num_tables, B, D)
    bounds = _table_bounds(tableidx, nnz, num_tables)
    for k in range(num_tables):
        lo, hi = bounds[k], bounds[k + 1]
        if hi > lo:
            _nat.backward_adagrad(shape, _nat.core_views(tt_cores, k), _nat.core_views(optimizer_state, k),
                                  indices[lo:hi], rowidx[lo:hi], hi - lo, None, B, d_output[k],
                                  float(learning_rate), float(eps), _ws(d_output.device))


def update_cache_state(indices: torch.Tensor, hashtbl: torch.Tensor, cache_freq: torch.Tensor) -> None:
    if indices.numel() == 0:
        return
    _need(hashtbl.numel() > 0 and hashtbl.numel() == cache_freq.numel(), "hashtbl / cache_freq size mismatch")
    _nat.cache_update(indices.contiguous(), hashtbl, cache_freq)


def cache_populate(num_embeddings: int, tt_p_shapes, tt_q_shapes, tt_ranks, tt_cores, L, hashtbl, cache_freq,
                   cache_state, cache_weight) -> None:
    _need(hashtbl.numel() > 0 and hashtbl.numel() == cache_freq.numel(), "hashtbl / cache_freq size mismatch")
    _need(hashtbl.numel() >= cache_weight.shape[0], "cache larger than hashtbl")
    shape = _nat.make_shape(tt_p_shapes, tt_q_shapes, tt_ranks)
    cw = cache_weight.data if isinstance(cache_weight, torch.nn.Parameter) else cache_weight
    _nat.cache_populate(shape, _nat.core_views(tt_cores), hashtbl, cache_freq, cache_state, cw,
                        _ws(hashtbl.device))


def preprocess_indices_sync(indices: torch.Tensor, offsets: torch.Tensor, num_tables: int, warmup: bool,
                            hashtbl: torch.Tensor, cache_state: torch.Tensor
                            ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, int, Optional[torch.Tensor]]:
    """(indices', rowidx, tableidx, num_tt_indices, cache_locations)
    (tt_embeddings_cuda.cu:1388-1507).  The host count costs one stream sync, as in
    the reference; the module class avoids it by keeping the count on the device."""
    dev = indices.device
    nnz = indices.numel()
    bag = torch.empty(nnz, dtype=torch.int64, device=dev)
    if nnz == 0:
        return indices, bag, torch.empty_like(bag), 0, None
    indices, offsets = indices.contiguous(), offsets.contiguous()
    total_bags = offsets.numel() - 1
    B = total_bags // num_tables
    if warmup or num_tables != 1:
        _nat.preprocess(indices, offsets, total_bags, True, None, None, None, bag, None, None, _ws(dev))
        return indices, bag % B, bag // B, nnz, None
    part = torch.empty_like(indices)
    loc = torch.empty(nnz, dtype=torch.int32, device=dev)
    count = torch.empty(1, dtype=torch.int32, device=dev)
    _nat.preprocess(indices, offsets, B, False, hashtbl, cache_state, part, bag, loc, count, _ws(dev))
    return part, bag, torch.zeros_like(bag), int(count.item()), loc


def cache_forward(B: int, nnz: int, cache_locations: torch.Tensor, rowidx: torch.Tensor,
                  cache_weight: torch.Tensor, output: torch.Tensor) -> None:
    _need(B > 0, "B must be positive")
    if nnz == 0:
        return
    cw = cache_weight.data if isinstance(cache_weight, torch.nn.Parameter) else cache_weight
    _nat.cache_forward(cache_locations.contiguous(), rowidx.contiguous(), 0, None, nnz, cw,
                       output.view(-1, cw.shape[1]))


def cache_backward_sgd(nnz: int, grad_output: torch.Tensor, cache_locations, rowidx, learning_rate: float,
                       cache_weight) -> None:
    if nnz == 0:
        return
    cw = cache_weight.data if isinstance(cache_weight, torch.nn.Parameter) else cache_weight
    _nat.cache_backward_sgd(cache_locations.contiguous(), rowidx.contiguous(), 0, None, nnz,
                            grad_output.contiguous().view(-1, cw.shape[1]), float(learning_rate), cw)


def cache_backward_dense(nnz: int, grad_output: torch.Tensor, cache_locations, rowidx, learning_rate: float,
                         cache_weight) -> torch.Tensor:
    cw = cache_weight.data if isinstance(cache_weight, torch.nn.Parameter) else cache_weight
    grad = torch.empty_like(cw)
    _nat.cache_backward_dense(cache_locations.contiguous(), rowidx.contiguous(), 0, None, nnz,
                              grad_output.contiguous().view(-1, cw.shape[1]), grad)
    return grad


def cache_backward_rowwise_adagrad_approx(nnz: int, grad_output: torch.Tensor, cache_locations, rowidx,
                                          learning_rate: float, eps: float, cache_optimizer_state,
                                          cache_weight) -> None:
    if nnz == 0:
        return
    cw = cache_weight.data if isinstance(cache_weight, torch.nn.Parameter) else cache_weight
    _nat.cache_backward_rowwise_adagrad(cache_locations.contiguous(), rowidx.contiguous(), 0, None, nnz,
                                        grad_output.contiguous().view(-1, cw.shape[1]), float(learning_rate),
                                        float(eps), cache_optimizer_state, cw)
