"""MI355X-native Tensor-Train embedding bag -- drop-in for the reference's
``FBTT.tt_embeddings_ops`` (same import path, class names, constructor arguments,
attributes and state_dict keys; reference: FBTT/tt_embeddings_ops.py:432-965).

What differs is everything underneath: the lookups, gradients, optimiser epilogues
and the LFU row cache run in ``libttemb_hip.so`` (hand-written HIP for gfx950,
C ABI in ``include/ttemb.h``) through ``ttemb_native``.  There is no
``batch_count`` chunking (the argument is accepted and ignored: partial products
never leave the chip), and no host synchronisation in ``forward`` -- the split
between TT-computed and cached ids stays on the device.

No CPU fallback exists: modules can be *constructed* without a GPU (so that shape
logic, initialisers and checkpoints can be exercised anywhere) but ``forward`` on
CPU tensors raises ``RuntimeError``.
"""
from __future__ import annotations

import logging
import math
from enum import Enum, unique
from typing import List, Optional, Sequence

import numpy as np
import torch
from torch import nn

import ttemb_native as _nat

__all__ = ["OptimType", "BufferList", "tt_matrix_to_full", "suggested_tt_shapes", "TTLookupFunction",
           "TableBatchedTTEmbeddingBag", "TTEmbeddingBag", "CapturedLookup"]

_LOG = logging.getLogger(__name__)


@unique
class OptimType(Enum):
    """Optimiser selector; value strings match the reference (tt_embeddings_ops.py:18-33).
    SGD / EXACT_SGD run the fused in-backward SGD; everything else runs fused Adagrad,
    exactly as the reference dispatches (:229-286)."""
    SGD = "sgd"
    EXACT_SGD = "exact_sgd"
    LAMB = "lamb"
    ADAM = "adam"
    EXACT_ADAGRAD = "exact_adagrad"
    EXACT_ROWWISE_ADAGRAD = "exact_row_wise_adagrad"
    LARS_SGD = "lars_sgd"
    PARTIAL_ROWWISE_ADAM = "partial_row_wise_adam"
    PARTIAL_ROWWISE_LAMB = "partial_row_wise_lamb"

    def __str__(self) -> str:
        return self.value


_SGD_LIKE = (OptimType.SGD, OptimType.EXACT_SGD)


class BufferList(nn.Module):
    """Ordered list of registered buffers named ``<name>0, <name>1, ...`` so that
    state_dict keys read ``optimizer_state.optimizer_state0`` like the reference's
    (tt_embeddings_ops.py:36-77)."""

    def __init__(self, name: str, buffers: Optional[Sequence[torch.Tensor]] = None) -> None:
        super().__init__()
        self._name = name
        self._length = 0
        for b in buffers or ():
            self.append(b)

    def append(self, buffer: torch.Tensor) -> "BufferList":
        self.register_buffer(f"{self._name}{self._length}", buffer)
        self._length += 1
        return self

    def extend(self, buffers: Sequence[torch.Tensor]) -> "BufferList":
        for b in buffers:
            self.append(b)
        return self

    def __len__(self) -> int:
        return self._length

    def __getitem__(self, index: int) -> torch.Tensor:
        if not -self._length <= index < self._length:
            raise IndexError(index)
        return getattr(self, f"{self._name}{index % self._length}")

    def __iter__(self):
        return (self[i] for i in range(self._length))


def _pad_ranks(tt_ranks: Sequence[int], T: int) -> List[int]:
    r = [int(x) for x in tt_ranks]
    return [1] + r + [1] if len(r) == T - 1 else r


def tt_matrix_to_full(tt_p_shapes: Sequence[int], tt_q_shapes: Sequence[int], tt_ranks: Sequence[int],
                      tt_cores: Sequence[torch.Tensor],
                      tt_permute: Optional[Sequence[int]] = None) -> torch.Tensor:
    """Dense ``[prod(p), prod(q)]`` table of a TT matrix (differentiable, any device).

    Same contract as the reference helper (tt_embeddings_ops.py:80-127): with
    ``tt_permute=[1, 0, 2, 3]`` core ``t`` is stored ``[p_t, R_t, q_t, R_{t+1}]``
    (the layer's layout); with ``None`` cores are already ``[R_t, p_t, q_t, R_{t+1}]``.
    """
    T = len(tt_p_shapes)
    R = _pad_ranks(tt_ranks, T)
    table = None
    for t, core in enumerate(tt_cores):
        dims = [R[t], int(tt_p_shapes[t]), int(tt_q_shapes[t]), R[t + 1]]
        if tt_permute is not None:
            stored = [dims[a] for a in tt_permute]
            g = core.reshape(stored).permute(*tt_permute)
        else:
            g = core.reshape(dims)
        # table: [P, Q, R_t] ; g: [R_t, p, q, R_{t+1}]  ->  [P, p, Q, q, R_{t+1}]
        if table is None:
            table = g.reshape(dims[1], dims[2], dims[3])
        else:
            P, Q = table.shape[0], table.shape[1]
            nxt = torch.tensordot(table, g, dims=([2], [0]))  # [P, Q, p, q, R]
            table = nxt.permute(0, 2, 1, 3, 4).reshape(P * dims[1], Q * dims[2], dims[3])
    return table.reshape(table.shape[0], table.shape[1]).float()


# --------------------------------------------------------------------------------------
# shape suggestion (reference: tt_embeddings_ops.py:369-429; sympy-free re-derivation)
# --------------------------------------------------------------------------------------
def _prime_factors(n: int) -> List[int]:
    out, f = [], 2
    while f * f <= n:
        while n % f == 0:
            out.append(f)
            n //= f
        f += 1 if f == 2 else 2
    if n > 1:
        out.append(n)
    return out


def _unordered_factorisations(n: int, d: int, lo: int = 2):
    """All non-decreasing d-tuples of integers >= lo with product n."""
    if d == 1:
        if n >= lo:
            yield (n,)
        return
    f = lo
    while f ** d <= n:
        if n % f == 0:
            for rest in _unordered_factorisations(n // f, d - 1, f):
                yield (f,) + rest
        f += 1


def _shape_entropy(factors: Sequence[int]) -> float:
    tot = float(sum(factors))
    return -sum((f / tot) * math.log(f / tot) for f in factors)


def _interleave(sorted_factors: Sequence[int]) -> List[int]:
    half = len(sorted_factors) // 2
    lo, hi = list(sorted_factors[:half]), list(sorted_factors[half:])
    out = []
    for i in range(len(hi)):
        if i < len(lo):
            out.append(lo[i])
        out.append(hi[i])
    return out


def _most_even_shape(n: int, d: int) -> List[int]:
    primes = _prime_factors(n)
    if len(primes) <= d:
        cands = [tuple(sorted(primes + [1] * (d - len(primes))))]
    else:
        cands = sorted(set(_unordered_factorisations(n, d)))
    best = max(cands, key=_shape_entropy)
    return _interleave(best)


def suggested_tt_shapes(n: int, d: int = 3, allow_round_up: bool = True) -> List[int]:
    """Factor ``n`` (optionally rounded up to a multiple of a power of ten) into ``d``
    factors that are as even as possible (maximum entropy of the normalised factors)."""
    n = int(n)
    if not allow_round_up:
        return _most_even_shape(n, d)
    best, best_h = None, -1.0
    for k in range(len(str(n))):
        step = 10 ** k
        shape = _most_even_shape(-(-n // step) * step, d)
        h = _shape_entropy(shape)
        if h > best_h + 1e-15:
            best, best_h = shape, h
    return best


# --------------------------------------------------------------------------------------
# autograd bridge
# --------------------------------------------------------------------------------------
class TTLookupFunction(torch.autograd.Function):
    """forward = TT rows (+ cached rows) bag-summed; backward = fused optimiser step
    (``sparse``) or dense core gradients.  Reference: tt_embeddings_ops.py:130-366."""

    @staticmethod
    def forward(ctx, module: "TableBatchedTTEmbeddingBag", table: int, B: int, indices: torch.Tensor,
                rowidx: torch.Tensor, offsets: Optional[torch.Tensor], nnz_dev: Optional[torch.Tensor],
                cache_loc: Optional[torch.Tensor], cache_weight: Optional[torch.Tensor],
                *tt_cores: torch.Tensor) -> torch.Tensor:
        ctx.module, ctx.table, ctx.B = module, table, B
        ctx.live_cache = cache_loc is not None
        # integer inputs, never differentiated: kept on ctx directly (save_for_backward costs version bookkeeping)
        ctx.inputs = (indices, rowidx, nnz_dev, cache_loc, offsets)
        cores = _nat.core_ptrs(tt_cores, table)
        nnz = indices.numel()
        out = torch.empty((B, module.embedding_dim), dtype=torch.float32, device=indices.device)
        # the forward's grouping of the ids is kept for the backward of this very call
        ctx.plan = _nat.new_plan(module._shape, nnz, indices.device)
        pending = getattr(module, "_before_weights", None)
        if pending is None:
            _nat.forward(module._shape, cores, indices, rowidx, offsets, nnz, nnz_dev, B, out, module._ws, ctx.plan)
        else:
            # something still has to write the cores (ttemb_dist.TTDataParallel: the previous step's all-reduce +
            # update): the id-only half of the forward is enqueued first so that it overlaps with it
            if ctx.plan is not None:
                _nat.forward(module._shape, cores, indices, rowidx, offsets, nnz, nnz_dev, B, out, module._ws, ctx.plan,
                             phase=1)
            pending()
            _nat.forward(module._shape, cores, indices, rowidx, offsets, nnz, nnz_dev, B, out, module._ws, ctx.plan,
                         phase=2 if ctx.plan is not None else 0)
        if ctx.live_cache and nnz > 0:
            _nat.cache_forward(cache_loc, rowidx, 0, nnz_dev, nnz, cache_weight.data, out, offsets)
        return out

    @staticmethod
    def backward(ctx, d_output: torch.Tensor):
        m, table, B = ctx.module, ctx.table, ctx.B
        indices, rowidx, nnz_dev, cache_loc, offsets = ctx.inputs
        nnz = indices.numel()
        if d_output.dtype != torch.float32 or not d_output.is_contiguous():
            d_output = d_output.contiguous().float()
        cores = _nat.core_ptrs(m.tt_cores, table)
        n_fixed = 9
        if m.sparse:
            if m.optimizer in _SGD_LIKE:
                _nat.backward_sgd(m._shape, cores, indices, rowidx, nnz, nnz_dev, B, d_output,
                                  float(m.learning_rate), m._ws, ctx.plan, offsets)
                if ctx.live_cache and nnz > 0:
                    _nat.cache_backward_sgd(cache_loc, rowidx, 0, nnz_dev, nnz, d_output,
                                            float(m.learning_rate), m.cache_weight.data,
                                            nnz_dev[1:] if nnz_dev.numel() > 1 else None)
            else:
                state = _nat.core_ptrs(list(m.optimizer_state), table)
                _nat.backward_adagrad(m._shape, cores, state, indices, rowidx, nnz, nnz_dev, B, d_output,
                                      float(m.learning_rate), float(m.eps), m._ws, ctx.plan, offsets)
                if ctx.live_cache and nnz > 0:
                    _nat.cache_backward_rowwise_adagrad(cache_loc, rowidx, 0, nnz_dev, nnz, d_output,
                                                        float(m.learning_rate), float(m.eps),
                                                        m.cache_optimizer_state, m.cache_weight.data)
            return (None,) * (n_fixed + len(m.tt_cores))
        # a data-parallel wrapper may have provided one flat bucket for the gradients (ttemb_dist): they are then
        # produced straight into it and nothing is handed back to autograd (no AccumulateGrad, no views)
        bucket = getattr(m, "_dense_grad_out", None)
        # (for ttemb_dist: did this gradient come from a grouped backward -- the family with bounded device-side waits,
        #  whose last kernel leaves its verdict in the workspace header?  Host-side rule, launches nothing.)
        fam_key = (nnz, B, rowidx is None, _nat.path_epoch)
        grouped = m._family_cache.get(fam_key)
        if grouped is None:
            if len(m._family_cache) > 256:
                m._family_cache.clear()
            grouped = m._family_cache[fam_key] = nnz > 0 and (_nat.kernel_family(m._shape, nnz, B, rowidx is None) & 7) in (
                _nat.FAMILY_GROUPED, _nat.FAMILY_GROUPED_WIDE)
        m._last_bwd_grouped = grouped
        if bucket is not None and not ctx.live_cache:
            if m._bucket_filled:
                # a second backward before dp.step() (micro-batches, two lookups through one module): the kernels
                # overwrite their destination, so this one goes to scratch and is added -- what AccumulateGrad does
                more = [torch.empty_like(b) for b in bucket]
                _nat.backward_dense(m._shape, cores, indices, rowidx, nnz, nnz_dev, B, d_output, more, m._ws,
                                    ctx.plan, offsets)
                torch._foreach_add_(bucket, more)
            else:
                _nat.backward_dense(m._shape, cores, indices, rowidx, nnz, nnz_dev, B, d_output, bucket, m._ws,
                                    ctx.plan, offsets)
                m._bucket_filled = True
            return (None,) * (n_fixed + len(m.tt_cores))
        grads = bucket or [torch.empty_like(c[table] if c.dim() == 3 else c) for c in m.tt_cores]
        _nat.backward_dense(m._shape, cores, indices, rowidx, nnz, nnz_dev, B, d_output, grads, m._ws, ctx.plan,
                            offsets)
        d_cache = None
        if ctx.live_cache:
            d_cache = torch.empty_like(m.cache_weight.data)
            _nat.cache_backward_dense(cache_loc, rowidx, 0, nnz_dev, nnz, d_output, d_cache,
                                      nnz_dev[1:] if nnz_dev.numel() > 1 else None)
        full = []
        for t, g in enumerate(grads):
            if m.num_tables == 1:
                full.append(g.unsqueeze(0))
            else:  # only this table's slice of the [num_tables, p, row] parameter gets gradient
                z = torch.zeros_like(m.tt_cores[t].data)
                z[table] = g
                full.append(z)
        return (None,) * (n_fixed - 1) + (d_cache,) + tuple(full)


class _TablesLookup(torch.autograd.Function):
    """``num_tables`` > 1 without a host synchronisation: every table is a *window* of the id list whose bounds the kernels
    read from ``offsets`` on the device (``ttemb_forward_window`` ...; the reference hands its kernels a per-id ``tableidx``
    instead, tt_embeddings_cuda.cu:1349-1365).  One node for the whole call: the output is the [num_tables, B, D] tensor the
    windows write their rows of; the backward runs table by table (fused step, or dense gradients of the [num_tables, p, row]
    parameters, each table's slice written by its own window)."""

    @staticmethod
    def forward(ctx, module: "TableBatchedTTEmbeddingBag", B: int, indices: torch.Tensor, offsets: torch.Tensor,
                *tt_cores: torch.Tensor) -> torch.Tensor:
        T = module.num_tables
        out = torch.empty((T, B, module.embedding_dim), dtype=torch.float32, device=indices.device)
        ctx.module, ctx.B, ctx.indices, ctx.offsets = module, B, indices, offsets
        if module._before_weights is not None:
            module._before_weights()   # a data-parallel update of the cores is pending: finish it first
        for k in range(T):
            _nat.forward_window(module._shape, _nat.core_ptrs(tt_cores, k), indices, offsets, k * B, B, out, module._ws)
        return out

    @staticmethod
    def backward(ctx, d_output: torch.Tensor):
        m, B, indices, offsets = ctx.module, ctx.B, ctx.indices, ctx.offsets
        if d_output.dtype != torch.float32 or not d_output.is_contiguous():
            d_output = d_output.contiguous().float()
        T = m.num_tables
        if m.sparse:
            for k in range(T):
                state = None if m.optimizer in _SGD_LIKE else _nat.core_ptrs(list(m.optimizer_state), k)
                _nat.backward_window(m._shape, _nat.core_ptrs(m.tt_cores, k), indices, offsets, k * B, B, d_output, m._ws,
                                     opt_state=state, lr=float(m.learning_rate), eps=float(m.eps))
            return (None,) * (4 + len(m.tt_cores))
        grads = [torch.empty_like(c) for c in m.tt_cores]
        for k in range(T):
            _nat.backward_window(m._shape, _nat.core_ptrs(m.tt_cores, k), indices, offsets, k * B, B, d_output, m._ws,
                                 d_cores=_nat.core_ptrs(grads, k))
        return (None,) * 4 + tuple(grads)


class _SparseLookup(torch.autograd.Function):
    """The common training call -- one table, ``sparse=True``, no live cache -- with as little Python around the two
    native calls as autograd allows: ONE tensor input (the first core, so that the node is recorded; every gradient is
    ``None`` because the update happens inside backward), bound native arguments (``ttemb_native.LeanCalls``).  Same
    kernels and results as ``TTLookupFunction``."""

    @staticmethod
    def forward(ctx, anchor: torch.Tensor, module: "TableBatchedTTEmbeddingBag", indices: torch.Tensor,
                offsets: torch.Tensor, B: int) -> torch.Tensor:
        nnz = indices.numel()
        out = torch.empty((B, module.embedding_dim), dtype=torch.float32, device=indices.device)
        ctx.module, ctx.indices, ctx.offsets, ctx.B = module, indices, offsets, B
        ctx.plan = module._lean.forward(module._cores(), indices, offsets, nnz, B, out)
        return out

    @staticmethod
    def backward(ctx, d_output: torch.Tensor):
        m = ctx.module
        if d_output.dtype != torch.float32 or not d_output.is_contiguous():
            d_output = d_output.contiguous().float()
        state = None if m.optimizer in _SGD_LIKE else m._states()
        m._lean.backward(m._cores(), state, ctx.indices, ctx.offsets, ctx.indices.numel(), ctx.B, d_output,
                         float(m.learning_rate), float(m.eps), ctx.plan)
        return None, None, None, None, None


class _BucketLookup(torch.autograd.Function):
    """The data-parallel step's call -- one table, ``sparse=False``, no live cache, a wrapper's flat gradient bucket attached
    (``ttemb_dist.TTDataParallel``) -- with the lean bridge of ``_SparseLookup``: one tensor input, bound native arguments.
    The forward is split around the wrapper's pending update; the backward writes the core gradients straight into the bucket
    and hands nothing back to autograd.  (Through ``TTLookupFunction`` the same step cost the host ~200 us on a slow host --
    as much as its GPU time at 409 600 ids.)  Same kernels and results."""

    @staticmethod
    def forward(ctx, anchor: torch.Tensor, module: "TableBatchedTTEmbeddingBag", indices: torch.Tensor,
                offsets: torch.Tensor, B: int) -> torch.Tensor:
        nnz = indices.numel()
        out = torch.empty((B, module.embedding_dim), dtype=torch.float32, device=indices.device)
        ctx.module, ctx.indices, ctx.offsets, ctx.B = module, indices, offsets, B
        pending = module._before_weights
        if pending is None:
            ctx.plan = module._lean.forward(module._cores(), indices, offsets, nnz, B, out)
        else:
            ctx.plan = module._lean.forward_split(module._cores(), indices, offsets, nnz, B, out, pending)
        return out

    @staticmethod
    def backward(ctx, d_output: torch.Tensor):
        m = ctx.module
        if d_output.dtype != torch.float32 or not d_output.is_contiguous():
            d_output = d_output.contiguous().float()
        bucket = m._dense_grad_out
        nnz = ctx.indices.numel()
        if bucket is None:   # the wrapper was detached between forward and backward: gradients through .grad
            grads = [torch.empty_like(c[0]) for c in m.tt_cores]
            m._last_bwd_grouped = m._lean.backward_dense(m._cores(), ctx.indices, ctx.offsets, nnz, ctx.B, d_output, grads, ctx.plan)
            for c, g in zip(m.tt_cores, grads):
                c.grad = g.unsqueeze(0) if c.grad is None else c.grad + g.unsqueeze(0)
        elif m._bucket_filled:   # a second backward before dp.step(): through scratch, added (what AccumulateGrad does)
            more = [torch.empty_like(b) for b in bucket]
            m._last_bwd_grouped = m._lean.backward_dense(m._cores(), ctx.indices, ctx.offsets, nnz, ctx.B, d_output, more, ctx.plan)
            torch._foreach_add_(bucket, more)
        else:
            m._last_bwd_grouped = m._lean.backward_dense(m._cores(), ctx.indices, ctx.offsets, nnz, ctx.B, d_output, bucket, ctx.plan)
            m._bucket_filled = True
        return None, None, None, None, None


class _ReplayLookup(torch.autograd.Function):
    """Autograd node of a captured lookup: forward and backward are one HIP-graph replay each."""

    @staticmethod
    def forward(ctx, anchor: torch.Tensor, cap: "CapturedLookup") -> torch.Tensor:
        ctx.cap = cap
        cap.fwd_graph.replay()
        return cap.output

    @staticmethod
    def backward(ctx, d_output: torch.Tensor):
        cap = ctx.cap
        cap.d_output.copy_(d_output)
        cap.bwd_graph.replay()
        return None, None


class CapturedLookup:
    """``emb.capture(nnz, B)``: a lookup of fixed size whose forward and whose backward (gradient + fused optimiser step)
    are each ONE HIP-graph replay -- for steps so small that Python and launch overhead are most of their time (the metric's
    literal "batch 2048": ~25 us of kernels under ~75-115 us of eager host work).  Call it like the module:
    ``out = cap(indices[, offsets])``; ``out`` is a static buffer that the next call overwrites (as with
    ``torch.cuda.make_graphed_callables``).  The workspace and plan the graphs were captured with are owned by this object,
    so other calls on the module cannot move them.  ``sparse=True`` modules with one table and no live cache."""

    def __init__(self, module: "TableBatchedTTEmbeddingBag", nnz: int, B: int, offsets: Optional[torch.Tensor] = None) -> None:
        assert module.sparse and module.num_tables == 1, "capture() covers the fused-optimiser mode of a single table"
        assert not (module.use_cache and not module.warmup), "capture() with a live row cache is not supported"
        self.module, self.nnz, self.B = module, int(nnz), int(B)
        dev = module.tt_cores[0].device
        self.indices = torch.zeros(self.nnz, dtype=torch.int64, device=dev)
        self.offsets = (torch.arange(self.B + 1, dtype=torch.int64, device=dev) if offsets is None
                        else offsets.to(dev, torch.int64).contiguous().clone())
        self.output = torch.empty((self.B, module.embedding_dim), dtype=torch.float32, device=dev)
        self.d_output = torch.zeros_like(self.output)
        self._lean = _nat.LeanCalls(module._shape, _nat.Workspace())   # private workspace: pinned for the graphs' lifetime
        cores = module._cores()
        state = None if module.optimizer in _SGD_LIKE else module._states()
        lr, eps = float(module.learning_rate), float(module.eps)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):   # warm-up outside capture: workspace allocation, LDS-size attributes, size queries
            plan = self._lean.forward(cores, self.indices, self.offsets, self.nnz, self.B, self.output)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        _nat.init()   # the pinned fault word exists before anything is captured (a capture must not allocate it)
        self.fwd_graph, self.bwd_graph = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.fwd_graph):
            self.plan = self._lean.forward(cores, self.indices, self.offsets, self.nnz, self.B, self.output)
        with torch.cuda.graph(self.bwd_graph):   # (a zero gradient: the captured update leaves the cores as they are)
            self._lean.backward(cores, state, self.indices, self.offsets, self.nnz, self.B, self.d_output, lr, eps, self.plan)
        self._lr, self._eps = lr, eps
        self._baked = self._pointers()

    def _pointers(self) -> tuple:
        """What the graphs hold by address: the cores and the optimiser state."""
        m = self.module
        ptrs = tuple(c.data_ptr() for c in m._cores())
        if m.optimizer not in _SGD_LIKE:
            ptrs += tuple(st.data_ptr() for st in m._states())
        return ptrs

    def __call__(self, indices: torch.Tensor, offsets: Optional[torch.Tensor] = None) -> torch.Tensor:
        m = self.module
        if m.use_cache and not m.warmup:
            # after cache_populate() the eager module serves and trains the hot ids in cache_weight; the captured graphs
            # read and update the TT cores only -- the two would diverge silently
            raise RuntimeError("the row cache went live after capture(): captured lookups do not cover a live cache")
        if float(m.learning_rate) != self._lr or float(m.eps) != self._eps:
            raise RuntimeError("learning rate / eps are part of the captured backward: capture() again after changing them")
        if self._pointers() != self._baked:
            raise RuntimeError("tt_cores / optimizer_state were re-allocated after capture() (.to(), .data = ..., "
                               "load_state_dict into new storage): capture() again")
        if m.use_cache:   # warm-up: the LFU statistics of a captured step count like those of an eager one
            m.update_cache(indices)
        self.indices.copy_(indices)
        if offsets is not None:
            self.offsets.copy_(offsets)
        return _ReplayLookup.apply(m._cores()[0], self)


# --------------------------------------------------------------------------------------
# the module
# --------------------------------------------------------------------------------------
class TableBatchedTTEmbeddingBag(nn.Module):
    """``num_tables`` TT tables with identical shapes looked up in one call.

    Constructor / attribute contract: reference tt_embeddings_ops.py:446-615.
    ``forward(indices, offsets)`` returns ``[num_tables, B, D]`` sum-pooled bags with
    ``include_last_offset`` semantics (``offsets`` has ``num_tables*B + 1`` entries).
    """

    __constants__ = ["num_tables", "num_embeddings", "embedding_dim", "tt_shape", "tt_rank"]

    def __init__(self, num_tables: int, num_embeddings: int, embedding_dim: int, tt_ranks: List[int],
                 tt_p_shapes: Optional[List[int]] = None, tt_q_shapes: Optional[List[int]] = None,
                 optimizer: OptimType = OptimType.SGD, learning_rate: float = 0.1, eps: float = 1.0e-10,
                 sparse: bool = True, use_cache: bool = False, cache_size: int = 0, hashtbl_size: int = 0,
                 weight_dist: str = "approx-normal", enforce_embedding_dim: bool = False,
                 batch_count: int = 1000) -> None:
        super().__init__()
        assert num_tables > 0 and num_embeddings > 0 and embedding_dim > 0
        assert num_tables == 1 or not use_cache, "cannot use cache when num_tables != 1"
        T = len(tt_ranks) + 1
        self.batch_count = batch_count  # accepted for compatibility; the kernels do not chunk
        self.tt_p_shapes: List[int] = (list(tt_p_shapes) if tt_p_shapes is not None
                                       else suggested_tt_shapes(num_embeddings, T))
        self.tt_q_shapes: List[int] = (list(tt_q_shapes) if tt_q_shapes is not None else
                                       suggested_tt_shapes(embedding_dim, T,
                                                           allow_round_up=not enforce_embedding_dim))
        assert 2 <= len(self.tt_p_shapes) <= 4
        assert len(self.tt_p_shapes) == T and len(self.tt_q_shapes) == T
        assert all(v > 0 for v in self.tt_p_shapes) and all(v > 0 for v in self.tt_q_shapes)
        assert all(v > 0 for v in tt_ranks)
        assert int(np.prod(np.array(self.tt_p_shapes, dtype=np.int64))) >= num_embeddings
        assert int(np.prod(np.array(self.tt_q_shapes, dtype=np.int64))) == embedding_dim
        self.num_tables, self.tt_ndim = num_tables, T
        self.num_embeddings, self.embedding_dim = num_embeddings, embedding_dim
        self.tt_ranks = [1] + [int(r) for r in tt_ranks] + [1]
        self.sparse, self.optimizer = sparse, optimizer
        self.learning_rate, self.eps = learning_rate, eps
        _LOG.info("TTEmbeddingBag p=%s q=%s R=%s sparse=%s optimizer=%s lr=%s eps=%s cache=%s/%s/%s",
                  self.tt_p_shapes, self.tt_q_shapes, self.tt_ranks, sparse, optimizer, learning_rate, eps,
                  use_cache, cache_size, hashtbl_size)
        dev = (torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available()
               else torch.device("cpu"))
        strides = [1] * T
        for t in range(T - 2, -1, -1):
            strides[t] = strides[t + 1] * self.tt_p_shapes[t + 1]
        self.register_buffer("L", torch.tensor(strides, dtype=torch.int64))
        self.tt_cores = nn.ParameterList()
        self.optimizer_state = BufferList("optimizer_state")
        for t in range(T):
            shape = [num_tables, self.tt_p_shapes[t],
                     self.tt_ranks[t] * self.tt_q_shapes[t] * self.tt_ranks[t + 1]]
            self.tt_cores.append(nn.Parameter(torch.empty(shape, device=dev, dtype=torch.float32)))
            st_shape = shape if optimizer not in _SGD_LIKE else 0
            self.optimizer_state.append(torch.zeros(st_shape, device=dev, dtype=torch.float32))
        self.reset_parameters(weight_dist)
        self.use_cache = use_cache
        if use_cache:
            if cache_size <= 0:
                cache_size = int(0.1 * num_embeddings)
            if hashtbl_size <= 0:
                hashtbl_size = num_embeddings
            assert hashtbl_size >= cache_size
            self.register_buffer("hashtbl", torch.full((hashtbl_size,), -1, device=dev, dtype=torch.int64))
            self.register_buffer("cache_freq", torch.zeros(hashtbl_size, device=dev, dtype=torch.int64))
            self.register_buffer("cache_state", torch.full((hashtbl_size,), -1, device=dev, dtype=torch.int32))
            self.cache_weight = nn.Parameter(torch.zeros((cache_size, embedding_dim), device=dev,
                                                         dtype=torch.float32))
            if sparse and optimizer not in _SGD_LIKE:
                st = (cache_size, embedding_dim) if optimizer == OptimType.EXACT_ADAGRAD else (cache_size,)
                self.register_buffer("cache_optimizer_state", torch.zeros(st, device=dev, dtype=torch.float32))
            else:
                self.cache_optimizer_state = None
        else:
            self.register_buffer("hashtbl", torch.empty(0, device=dev, dtype=torch.int64))
            self.register_buffer("cache_state", torch.empty(0, device=dev, dtype=torch.int32))
            self.cache_optimizer_state = None
            self.cache_weight = None
        self.warmup = True
        self._dense_grad_out = None
        self._before_weights = None   # set by ttemb_dist.TTDataParallel while an update of the cores is pending
        self._use_windows = True      # num_tables > 1: per-table windows read on the device (False: always split on the host)
        self._bucket_filled = False   # set by the backward when it wrote the core gradients into the wrapper's bucket
        self._family_cache: dict = {}  # (nnz, B, ...) -> "the backward of this size runs on the grouped kernels" (ttemb_dist)
        self._shape = _nat.make_shape(self.tt_p_shapes, self.tt_q_shapes, self.tt_ranks)
        self._ws = _nat.Workspace()
        self._lean = _nat.LeanCalls(self._shape, self._ws)
        self._use_lean = True   # (tools/host_breakdown.py switches it off to time the general autograd bridge)
        self._core_list: Optional[tuple] = None
        self._state_list: Optional[tuple] = None
        self.register_load_state_dict_post_hook(TableBatchedTTEmbeddingBag._after_load)

    @staticmethod
    def _after_load(module, incompatible_keys) -> None:
        """Checkpoint load path (the reference has none): `warmup` is not part of the
        state dict, so it is re-derived -- a populated cache (`cache_state` holds ranks) is live."""
        if module.use_cache and module.cache_state.numel():
            module.warmup = not bool((module.cache_state >= 0).any().item())

    # ---- weights ------------------------------------------------------------------
    def full_weight(self) -> torch.Tensor:
        assert self.num_tables == 1, "full_weight() only supported for num_tables == 1 for now"
        return tt_matrix_to_full(self.tt_p_shapes, self.tt_q_shapes, self.tt_ranks,
                                 list(self.tt_cores), [1, 0, 2, 3])

    def reset_parameters(self, weight_dist: str) -> None:
        """Initialisers of the reference (tt_embeddings_ops.py:629-808), vectorised."""
        from ttemb_init import init_cores
        init_cores(self, weight_dist)

    def set_learning_rate(self, lr: float) -> None:
        self.learning_rate = lr

    def get_params(self):
        params = self.tt_cores
        if self.use_cache:
            params.append(self.cache_weight)
        return params

    # ---- cache --------------------------------------------------------------------
    def reset_cache(self) -> None:
        """Forget every tracked id (the reference's version is dead code: typo at :811)."""
        if self.use_cache:
            self.hashtbl.fill_(-1)
            self.cache_freq.fill_(0)
            self.cache_state.fill_(-1)
            self.warmup = True

    def _fused_probe(self) -> bool:
        """Live cache, default insert: update_cache_state and preprocess_indices_sync of a forward are ONE probe pass
        (``ttemb_preprocess_update``).  The reference-exact one-sweep insert keeps its own launch."""
        return (self.use_cache and not self.warmup and self.num_tables == 1
                and not getattr(self, "lfu_one_sweep_insert", False))

    def update_cache(self, indices: torch.Tensor) -> None:
        if self.use_cache:   # `lfu_one_sweep_insert = True` on the module selects the reference's insert bit for bit
            _nat.cache_update(indices.long().contiguous(), self.hashtbl, self.cache_freq,
                              getattr(self, "lfu_one_sweep_insert", False))

    def cache_populate(self) -> None:
        """Freeze the LFU statistics: the ``cache_size`` hottest ids get their rows
        materialised in ``cache_weight``; ends the warm-up (reference :816-830)."""
        if self.use_cache:
            _nat.cache_populate(self._shape, _nat.core_views(self.tt_cores), self.hashtbl, self.cache_freq,
                                self.cache_state, self.cache_weight.data, self._ws)
            self.warmup = False

    def capture(self, nnz: int, B: int, offsets: Optional[torch.Tensor] = None) -> CapturedLookup:
        """Fixed-size lookup whose forward and backward replay captured HIP graphs (see ``CapturedLookup``)."""
        return CapturedLookup(self, nnz, B, offsets)

    # ---- lookup -------------------------------------------------------------------
    def _cores(self) -> tuple:
        """The core Parameters as a tuple (walking the ParameterList costs microseconds per call); rebuilt when ANY of the
        list's Parameter objects is no longer the cached one (2-4 identity tests)."""
        cl = self._core_list
        live = self.tt_cores._parameters
        if cl is None or len(cl) != len(live) or any(c is not live.get(str(t)) for t, c in enumerate(cl)):
            cl = self._core_list = tuple(self.tt_cores)
        return cl

    def _states(self) -> tuple:
        sl = self._state_list
        live = self.optimizer_state._buffers
        if sl is None or len(sl) != len(live) or any(b is not live.get(f"optimizer_state{t}") for t, b in enumerate(sl)):
            sl = self._state_list = tuple(self.optimizer_state)
        return sl

    def _lookup_one_table(self, table: int, B: int, indices: torch.Tensor, offsets: torch.Tensor) -> torch.Tensor:
        nnz = indices.numel()
        dev = indices.device
        live = self.use_cache and not self.warmup
        if not live:  # rows are derived from `offsets` inside the native calls: no separate launch, no tensor
            if self.num_tables == 1 and self._use_lean and not torch.is_grad_enabled():
                # inference (the drivers' evaluation passes run under no_grad): no autograd node and no plan kept -- a forward
                # that forms its prefix products in the chain kernel then stores none of them (1.08 GB at papers100M, 819 200 ids)
                if self._before_weights is not None:
                    self._before_weights()   # a data-parallel update of the cores is pending: finish it first
                out = torch.empty((B, self.embedding_dim), dtype=torch.float32, device=dev)
                self._lean.forward(self._cores(), indices, offsets, nnz, B, out, keep_plan=False)
                return out
            if self.sparse and self.num_tables == 1 and self._use_lean:
                return _SparseLookup.apply(self._cores()[0], self, indices, offsets, B)
            if not self.sparse and self.num_tables == 1 and self._use_lean and self._dense_grad_out is not None:
                return _BucketLookup.apply(self._cores()[0], self, indices, offsets, B)
            return TTLookupFunction.apply(self, table, B, indices, None, offsets, None, None, None,
                                          *self.tt_cores)
        rowidx = torch.empty(nnz, dtype=torch.int64, device=dev)
        part = torch.empty_like(indices)
        loc = torch.empty(nnz, dtype=torch.int32, device=dev)
        # [number of TT ids, "a cache row occurs twice in this batch"]: both stay on the device.  The second word lets
        # the cache backward update rows with one writer each without float atomics; it comes from per-row position
        # stamps (module scratch, any content, not part of the state dict).
        nnz_tt = torch.empty(2, dtype=torch.int32, device=dev)
        stamp = getattr(self, "_dup_stamp", None)
        if stamp is None or stamp.device != dev or stamp.numel() != self.cache_weight.shape[0]:
            stamp = self._dup_stamp = torch.empty(self.cache_weight.shape[0], dtype=torch.int32, device=dev)
        # the LFU update of this batch rides in the probe pass (forward() skipped update_cache for it)
        freq = self.cache_freq if self._fused_probe() else None
        _nat.preprocess(indices, offsets, B, False, self.hashtbl, self.cache_state, part, rowidx, loc, nnz_tt,
                        self._ws, stamp, 0, freq)
        return TTLookupFunction.apply(self, table, B, part, rowidx, offsets, nnz_tt, loc, self.cache_weight,
                                      *self.tt_cores)

    def forward(self, indices: torch.Tensor, offsets: torch.Tensor, warmup: bool = True) -> torch.Tensor:
        # `warmup` is accepted and ignored, like the reference (it reads self.warmup, :862)
        if not indices.is_cuda:
            raise RuntimeError("TTEmbeddingBag.forward needs tensors on a ROCm device; there is no CPU fallback")
        indices, offsets = indices.long().contiguous(), offsets.long().contiguous()
        assert (offsets.numel() - 1) % self.num_tables == 0
        B = (offsets.numel() - 1) // self.num_tables
        if not self._fused_probe():
            self.update_cache(indices)
        if self.num_tables == 1:
            return self._lookup_one_table(0, B, indices, offsets).unsqueeze(0)
        # every table is a window of the id list, its bounds read from `offsets` on the device (no host synchronisation) -- when the
        # grouped kernels serve the shape; else the id list is split on the host, one plain lookup per table
        nnz = indices.numel()
        if (not self.use_cache and self._use_windows and nnz > 0 and B > 0
                and _nat.window_workspace_bytes(self._shape, _nat.OP_BACKWARD, nnz, offsets.numel() - 1, B) >= 0):
            return _TablesLookup.apply(self, B, indices, offsets, *self.tt_cores)
        bounds = offsets[:: B].tolist()  # host sync: only this fallback of the multi-table path pays it
        outs = []
        for k in range(self.num_tables):
            lo, hi = int(bounds[k]), int(bounds[k + 1])
            offs_k = (offsets[k * B:(k + 1) * B + 1] - lo).contiguous()
            outs.append(self._lookup_one_table(k, B, indices[lo:hi].contiguous(), offs_k))
        return torch.stack(outs, 0)


class TTEmbeddingBag(TableBatchedTTEmbeddingBag):
    """TT embedding lookup for exactly one table; ``forward`` returns ``[B, D]``
    (reference: tt_embeddings_ops.py:918-965)."""

    def __init__(self, num_embeddings: int, embedding_dim: int, tt_ranks: List[int],
                 tt_p_shapes: Optional[List[int]] = None, tt_q_shapes: Optional[List[int]] = None,
                 optimizer: OptimType = OptimType.SGD, learning_rate: float = 0.1, eps: float = 1.0e-10,
                 sparse: bool = True, use_cache: bool = True, cache_size: int = 0, hashtbl_size: int = 0,
                 weight_dist: str = "approx-normal", enforce_embedding_dim: bool = False,
                 batch_count: int = 1000) -> None:
        super().__init__(1, num_embeddings, embedding_dim, tt_ranks, tt_p_shapes, tt_q_shapes, optimizer,
                         learning_rate, eps, sparse, use_cache, cache_size, hashtbl_size, weight_dist,
                         enforce_embedding_dim, batch_count)

    def forward(self, indices: torch.Tensor, offsets: torch.Tensor, warmup: bool = True) -> torch.Tensor:
        # same result as the reference's ``super().forward(...)[0]`` (:960-965) without the
        # [1, B, D] view: selecting table 0 would cost a zero-fill + copy of B*D floats in backward
        if not indices.is_cuda:
            raise RuntimeError("TTEmbeddingBag.forward needs tensors on a ROCm device; there is no CPU fallback")
        if indices.dtype != torch.int64 or not indices.is_contiguous():
            indices = indices.long().contiguous()
        if offsets.dtype != torch.int64 or not offsets.is_contiguous():
            offsets = offsets.long().contiguous()
        if self.use_cache and not self._fused_probe():
            self.update_cache(indices)
        return self._lookup_one_table(0, offsets.numel() - 1, indices, offsets)
