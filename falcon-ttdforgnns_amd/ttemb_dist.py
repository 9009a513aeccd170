"""Data-parallel step for the TT layer: one collective per step over the flattened
core gradients (198 400 floats for ogbn-products r16), then the fused SGD epilogue.

The reference's multi-GPU path is a DDP stub that cannot train a TT model
(sage_dgl_partition.py:198-255), so this is designed fresh: the whole
table is replicated (it is < 10 MB), each rank looks up its own mini-batch in
dense-gradient mode, and the tiny gradients are summed with ONE all-reduce (RCCL over
xGMI with backend "nccl"; gloo on CPU in the tests) instead of one bucket per core.
The 1/world averaging is folded into the learning rate.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import torch
import torch.distributed as dist


class FlatGradBucket:
    """One contiguous buffer holding every core gradient (and the cache gradient)."""

    def __init__(self, params: Sequence[torch.Tensor]) -> None:
        self.params = list(params)
        self.sizes = [p.numel() for p in self.params]
        # keep every segment 16-byte aligned for the float4 optimiser kernel
        self.offsets, off = [], 0
        for n in self.sizes:
            self.offsets.append(off)
            off += (n + 3) & ~3
        dev = self.params[0].device
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.views = [self.flat[o:o + n].view_as(p) for o, n, p in zip(self.offsets, self.sizes, self.params)]

    def pack(self) -> None:
        for v, p in zip(self.views, self.params):
            if p.grad is None:
                v.zero_()
            else:
                v.copy_(p.grad)


def default_apply(weight: torch.Tensor, grad: torch.Tensor, lr: float) -> None:
    """Fused SGD epilogue on the device (libttemb_hip.so); no CPU fallback."""
    import ttemb_native as nat
    nat.sgd_step(weight.view(-1), grad.reshape(-1), lr)


class TTDataParallel:
    """Wrap a dense-mode (``sparse=False``) TT module for data-parallel SGD."""

    def __init__(self, module, process_group: Optional[dist.ProcessGroup] = None,
                 apply_fn: Callable[[torch.Tensor, torch.Tensor, float], None] = default_apply) -> None:
        assert not module.sparse, "data-parallel training needs dense gradients (sparse=False)"
        self.module = module
        self.group = process_group
        self.apply_fn = apply_fn
        params: List[torch.Tensor] = list(module.tt_cores)
        if getattr(module, "cache_weight", None) is not None:
            params.append(module.cache_weight)
        self.bucket = FlatGradBucket(params)
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1

    def broadcast_parameters(self, src: int = 0) -> None:
        if self.world > 1:
            for p in self.bucket.params:
                dist.broadcast(p.data, src, group=self.group)

    def step(self, lr: Optional[float] = None) -> None:
        """Call after ``loss.backward()``: all-reduce(sum) once, then w -= lr/world * g."""
        lr = float(self.module.learning_rate if lr is None else lr)
        self.bucket.pack()
        if self.world > 1:
            dist.all_reduce(self.bucket.flat, op=dist.ReduceOp.SUM, group=self.group)
        for p, g in zip(self.bucket.params, self.bucket.views):
            self.apply_fn(p.data, g, lr / self.world)
            p.grad = None
