"""``Eff_TTEmbedding`` -- the reference's second TT-embedding API
(Efficient_TT/efficient_tt.py:214-307) on the same MI355X kernels as ``TTEmbeddingBag``.

The reference's "efficient" variant reuses the prefix product ``G0[i0] . G1[i1]`` across
lookups that share ``(i0, i1)`` and sums gradient rows over duplicate ids before the chain
(efficient_tt_cuda.cu:159-377, 970-1247).  Both ideas are what the native fast path does
for *every* caller (ids are grouped by ``(i0, i1)``; gradients of a group are reduced in
registers), so this class is a thin adapter:

* cores are 2-D ``[p_t, R_t q_t R_{t+1}]`` parameters (no table axis), same row layout as
  ``TTEmbeddingBag`` -- interchangeable after ``squeeze(0)``;
* ``forward(indices, offsets=None, unique=None, inverse=None) -> [len(indices), D]``: one row
  per id, no pooling; ``unique`` / ``inverse`` are accepted and unused (no host-side
  ``torch.unique`` is needed);
* backward applies the fused SGD step to the cores with the constructor's ``learning_rate``
  (the reference hard-codes 0.1 at efficient_tt.py:140,159) and hands autograd no gradients.

Unlike the reference there is no 136 900-group capacity limit, no float-division id decode
(ids above 2^24 are exact) and no process-global device buffers.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch

import ttemb_native as _nat
from FBTT.tt_embeddings_ops import suggested_tt_shapes

__all__ = ["Eff_TTEmbedding", "TT_core_function"]


class TT_core_function(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module: "Eff_TTEmbedding", indices: torch.Tensor, *tt_cores: torch.Tensor) -> torch.Tensor:
        n = indices.numel()
        dev = indices.device
        _, offsets = module._iota(n, dev)
        ctx.module = module
        ctx.save_for_backward(indices, offsets)
        out = torch.empty((n, module.embedding_dim), dtype=torch.float32, device=dev)
        ctx.plan = _nat.new_plan(module._shape, n, dev)
        _nat.forward(module._shape, _nat.core_views(tt_cores), indices, None, offsets, n, None, n, out, module._ws,
                     ctx.plan)
        return out

    @staticmethod
    def backward(ctx, grad_output: torch.Tensor):
        m = ctx.module
        indices, offsets = ctx.saved_tensors
        n = indices.numel()
        _nat.backward_sgd(m._shape, _nat.core_views(m.tt_cores), indices, None, n, None, n,
                          grad_output.contiguous().float(), float(m.learning_rate), m._ws, ctx.plan, offsets)
        return (None, None) + (None,) * len(m.tt_cores)


class Eff_TTEmbedding(torch.nn.Module):
    def __init__(self, num_embeddings: int, embedding_dim: int, tt_ranks: List[int],
                 tt_p_shapes: Optional[List[int]] = None, tt_q_shapes: Optional[List[int]] = None,
                 optimizer: str = "SGD", learning_rate: float = 0.1, weight_dist: str = "uniform", device=0,
                 batch_size: int = 4096) -> None:
        super().__init__()
        self.num_embeddings, self.embedding_dim = num_embeddings, embedding_dim
        self.num_tt_core = len(tt_ranks) + 1
        self.tt_ranks = [1] + [int(r) for r in tt_ranks] + [1]
        self.batch_size = batch_size  # kept for API compatibility; nothing is pre-sized by it
        self.tt_p_shapes: List[int] = (list(tt_p_shapes) if tt_p_shapes is not None
                                       else suggested_tt_shapes(num_embeddings, self.num_tt_core))
        self.tt_q_shapes: List[int] = (list(tt_q_shapes) if tt_q_shapes is not None
                                       else suggested_tt_shapes(embedding_dim, self.num_tt_core))
        assert int(np.prod(self.tt_p_shapes)) >= num_embeddings
        assert int(np.prod(self.tt_q_shapes)) == embedding_dim
        self.optimizer, self.learning_rate, self.weight_dist = optimizer, learning_rate, weight_dist
        if isinstance(device, int):
            device = torch.device("cuda", device) if torch.cuda.is_available() else torch.device("cpu")
        self.device = torch.device(device)
        self.tt_cores = torch.nn.ParameterList()
        for t in range(self.num_tt_core):
            self.tt_cores.append(torch.nn.Parameter(torch.empty(
                [self.tt_p_shapes[t], self.tt_ranks[t] * self.tt_q_shapes[t] * self.tt_ranks[t + 1]],
                device=self.device, dtype=torch.float32)))
        self.reset_parameters()
        self.tensor_p_shape = torch.tensor(self.tt_p_shapes, device=self.device)
        self.tensor_q_shape = torch.tensor(self.tt_q_shapes, device=self.device)
        self.tensor_tt_ranks = torch.tensor(self.tt_ranks, device=self.device)
        self._shape = _nat.make_shape(self.tt_p_shapes, self.tt_q_shapes, self.tt_ranks)
        self._ws = _nat.Workspace()
        self._iota_cache = None

    def reset_parameters(self) -> None:
        if self.weight_dist == "uniform":  # same formula as efficient_tt.py:274-283
            stddev = np.sqrt(2.0 / (self.num_embeddings + self.embedding_dim))
            rank_term = float(np.prod(np.array(self.tt_ranks, dtype=np.float64) ** (-1.0 / (2 * self.num_tt_core))))
            hi = stddev ** (1.0 / self.num_tt_core) * rank_term
            with torch.no_grad():
                for c in self.tt_cores:
                    c.uniform_(0.0, float(hi))

    def _iota(self, n: int, dev: torch.device):
        c = self._iota_cache
        if c is None or c[0].numel() < n + 1 or c[0].device != dev:
            c = (torch.arange(n + 1, dtype=torch.int64, device=dev),)
            self._iota_cache = c
        return c[0][:n], c[0][:n + 1]

    def forward(self, indices: torch.Tensor, offsets=None, unique=None, inverse=None) -> torch.Tensor:
        if not indices.is_cuda:
            raise RuntimeError("Eff_TTEmbedding.forward needs tensors on a ROCm device; there is no CPU fallback")
        indices = indices.long().contiguous()
        if not torch.is_grad_enabled():   # inference: no autograd node, no plan kept (ttemb_forward with plan == NULL)
            n = indices.numel()
            _, offs = self._iota(n, indices.device)
            out = torch.empty((n, self.embedding_dim), dtype=torch.float32, device=indices.device)
            _nat.forward(self._shape, _nat.core_views(self.tt_cores), indices, None, offs, n, None, n, out, self._ws, None)
            return out
        return TT_core_function.apply(self, indices, *self.tt_cores)
