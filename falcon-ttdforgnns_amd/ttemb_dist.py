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
        # behind the gradients: four floats of which the first is the FAULT COUNT of the step -- 1 on a rank whose gradient
        # came from a poisoned plan (a device-side wait of the grouping pass ran out: that gradient is NaN), else 0.  It is
        # summed by the same all-reduce, so every rank learns whether ANY rank faulted and all of them skip the update
        # together (TTDataParallel.flush): replicas stay identical, nobody trains on NaN.
        self.n_grad = off
        self.flat = torch.zeros(off + 4, dtype=torch.float32, device=dev)
        self.fault = self.flat[off:off + 1]
        self.views = [self.flat[o:o + n].view_as(p) for o, n, p in zip(self.offsets, self.sizes, self.params)]

    def pack(self) -> None:
        for v, p in zip(self.views, self.params):
            if p.grad is None:
                v.zero_()
            else:
                v.copy_(p.grad)


def default_apply(weight: torch.Tensor, grad: torch.Tensor, lr: float, skip: Optional[torch.Tensor] = None) -> None:
    """Fused SGD epilogue on the device (libttemb_hip.so); no CPU fallback.  ``skip``: a device float -- non-zero leaves the
    weights as they are (some rank's gradient of this step came from a poisoned plan)."""
    import ttemb_native as nat
    if skip is None:
        nat.sgd_step(weight.view(-1), grad.reshape(-1), lr)
    else:
        nat.sgd_step_guarded(weight.view(-1), grad.reshape(-1), lr, skip)


class TTDataParallel:
    """Wrap a dense-mode (``sparse=False``) TT module for data-parallel SGD.

    The core gradients are produced by the backward kernels directly inside one flat bucket
    (no packing copy), summed with ONE all-reduce, and applied by ONE fused SGD launch over a
    flat weight buffer the cores are views of.  If a caller re-points ``tt_cores[t].data``
    (the reference's initialisers do, gnn_model.py:142-178) the step falls back to per-core
    launches until ``adopt_parameters()`` is called again.
    """

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
        self.n_cores = len(module.tt_cores)
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.flat_weights: Optional[torch.Tensor] = None
        self.weight_views: List[torch.Tensor] = []
        self._pending = None
        self._skip: Optional[torch.Tensor] = None
        self._fault_dirty = True     # (the bucket's fault slot holds something other than 0)
        self._nat = None
        self._word_buf, self._word = None, None
        self.adopt_parameters()
        # the backward kernels write the core gradients straight into the bucket
        module._dense_grad_out = [v[0] if v.dim() == 3 and v.shape[0] == 1 else v
                                  for v in self.bucket.views[: self.n_cores]] if module.num_tables == 1 else None

    def adopt_parameters(self) -> None:
        """Move the parameters into one flat buffer (same layout as the gradient bucket)."""
        b = self.bucket
        self.flat_weights = torch.zeros(b.n_grad, dtype=torch.float32, device=b.flat.device)
        self.weight_views = [self.flat_weights[o:o + n].view_as(p) for o, n, p in zip(b.offsets, b.sizes, b.params)]
        with torch.no_grad():
            for v, p in zip(self.weight_views, b.params):
                v.copy_(p.data)
                p.data = v

    def _flat_ok(self) -> bool:
        return all(p.data.data_ptr() == v.data_ptr() for p, v in zip(self.bucket.params, self.weight_views))

    def broadcast_parameters(self, src: int = 0) -> None:
        if self.world > 1:
            if self._flat_ok():
                dist.broadcast(self.flat_weights, src, group=self.group)
            else:
                for p in self.bucket.params:
                    dist.broadcast(p.data, src, group=self.group)

    def step(self, lr: Optional[float] = None, overlap: bool = False) -> None:
        """Call after ``loss.backward()``: all-reduce(sum) once, then w -= lr/world * g.

        ``overlap=True`` only *starts* the all-reduce: waiting for it and the update are deferred to the moment the
        next ``forward`` needs the cores, i.e. after that forward's id-only work (grouping pass) has been enqueued --
        the collective then runs under ~45 us of kernels that do not depend on it.  ``flush()`` (called by the
        next forward, or by hand before reading the weights) finishes a deferred step.
        """
        if self._pending is not None:
            self.flush()
        lr = float(self.module.learning_rate if lr is None else lr)
        b = self.bucket
        if getattr(self.module, "_bucket_filled", False) and len(b.params) == self.n_cores:
            self.module._bucket_filled = False   # the backward wrote every core gradient straight into the bucket
        else:
            for v, p in zip(b.views, b.params):  # gradients that did not land in the bucket are packed
                if p.grad is None:
                    v.zero_()
                elif p.grad.data_ptr() != v.data_ptr():
                    v.copy_(p.grad)
            for p in b.params:
                p.grad = None
        # this rank's fault count rides in the bucket: the word the grouped backward's last kernel left in the workspace
        # header (1 = poisoned plan; nothing is synchronised here).  Backwards of the other kernel families have no
        # bounded waits and do not write the word.  A fault some EARLIER call reported raises here, before NaN gradients
        # are summed into every rank.
        word = self._poison_word()
        self._skip = b.fault
        if self.world == 1:
            self._skip = word      # one process: the guarded step reads the header word itself (no copy, nothing to sum)
        elif word is None:
            if self._fault_dirty:
                b.fault.zero_()
                self._fault_dirty = False
        else:
            b.fault.copy_(word)    # int32 -> float: summed by the all-reduce below
            self._fault_dirty = True
        work = None
        if self.world > 1:
            work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=overlap)
        self._pending = (work if overlap else None, lr)
        if overlap:
            self.module._before_weights = self.flush
        else:
            self.flush()

    def flush(self) -> None:
        """Finish a deferred step: wait for the all-reduce, apply the update."""
        pending, self._pending = getattr(self, "_pending", None), None
        self.module._before_weights = None
        if pending is None:
            return
        work, lr = pending
        if work is not None:
            work.wait()
        b = self.bucket
        guard = (self._skip,) if self.apply_fn is default_apply else ()   # (an injected epilogue -- the CPU tests' -- takes no guard)
        if self._flat_ok():
            self.apply_fn(self.flat_weights, b.flat[:b.n_grad], lr / self.world, *guard)
        else:
            for p, g in zip(b.params, b.views):
                self.apply_fn(p.data, g, lr / self.world, *guard)

    def _poison_word(self) -> Optional[torch.Tensor]:
        """int32[1] view of the word the module's last grouped backward left in its workspace header, or None (CPU, or a
        backward of a kernel family without device-side waits).  The view is cached per workspace buffer: this sits on the
        host path of every step (a data-parallel rank is host-bound on a slow host)."""
        m = self.module
        if not self.bucket.flat.is_cuda or not getattr(m, "_last_bwd_grouped", False):
            return None
        nat = self._nat
        if nat is None:
            import ttemb_native as nat
            self._nat = nat
        nat.status()   # an expired wait an earlier call reported: RuntimeError on this rank, before the collective
        buf = m._ws.buf
        if buf is not self._word_buf:
            self._word_buf, self._word = buf, nat.poison_word(m._ws)
        return self._word

