"""ctypes binding of ``libttemb_hip.so`` (the C ABI declared in ``include/ttemb.h``).

This is the only place Python touches the native library.  Everything here takes
``torch`` tensors living on a ROCm device, passes raw ``data_ptr()`` values and the
current HIP stream, and raises ``RuntimeError`` with the library's message on a
non-zero status -- the same error convention as the reference's pybind module
(``c10::Error`` -> ``RuntimeError``; FBTT/tt_embeddings.cpp:131-161).

There is no CPU fallback: if the shared library is missing the import of this
module fails, and CPU tensors are rejected.
"""
from __future__ import annotations

import ctypes
import os
from typing import List, Optional, Sequence

import torch  # noqa: F401  (loads torch's libamdhip64.so.7 first, so we share one HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TTEMB_LIB") or os.path.join(_HERE, "lib", "libttemb_hip.so")

MAX_CORES = 4
ABI_VERSION = 4
OP_FORWARD, OP_BACKWARD, OP_PREPROCESS, OP_CACHE_POPULATE = 0, 1, 2, 3
PATH_AUTO, PATH_GENERIC, PATH_FAST3, PATH_PER_BAG = 0, 1, 2, 3

# every symbol include/ttemb.h declares (tests check the library exports all of them)
EXPORTED_SYMBOLS = (
    "ttemb_abi_version", "ttemb_last_error", "ttemb_workspace_bytes", "ttemb_plan_bytes", "ttemb_set_path",
    "ttemb_profile_enable", "ttemb_profile_read", "ttemb_kernel_family", "ttemb_set_piece_limits", "ttemb_set_wide_slab_min_ids", "ttemb_init", "ttemb_status", "ttemb_set_spin_limit",
    "ttemb_forward", "ttemb_forward_group", "ttemb_forward_lookup", "ttemb_backward_dense", "ttemb_backward_sgd", "ttemb_backward_adagrad",
    "ttemb_window_workspace_bytes", "ttemb_forward_window", "ttemb_backward_dense_window", "ttemb_backward_sgd_window", "ttemb_backward_adagrad_window",
    "ttemb_sgd_step", "ttemb_sgd_step_guarded", "ttemb_adagrad_step", "ttemb_cache_update", "ttemb_cache_update_one_sweep", "ttemb_cache_populate",
    "ttemb_preprocess", "ttemb_preprocess_update", "ttemb_cache_forward", "ttemb_cache_backward_sgd",
    "ttemb_cache_backward_dense", "ttemb_cache_backward_rowwise_adagrad",
)


class Shape(ctypes.Structure):
    """Mirror of ``ttemb_shape_t``."""
    _fields_ = [("T", ctypes.c_int32), ("p", ctypes.c_int32 * MAX_CORES),
                ("q", ctypes.c_int32 * MAX_CORES), ("R", ctypes.c_int32 * (MAX_CORES + 1))]


def make_shape(p: Sequence[int], q: Sequence[int], ranks: Sequence[int]) -> Shape:
    """``ranks`` may be the inner ranks (T-1 values) or the padded list (T+1 values)."""
    T = len(p)
    r = [int(x) for x in ranks]
    if len(r) == T - 1:
        r = [1] + r + [1]
    if len(q) != T or len(r) != T + 1 or not (2 <= T <= MAX_CORES):
        raise RuntimeError(f"inconsistent TT shape: p={list(p)} q={list(q)} ranks={list(ranks)}")
    s = Shape()
    s.T = T
    for t in range(T):
        s.p[t], s.q[t] = int(p[t]), int(q[t])
    for t in range(T + 1):
        s.R[t] = r[t]
    return s


def _load() -> ctypes.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C falcon-ttdforgnns_amd/csrc`.  There is no CPU fallback for the TT embedding path.")
    lib = ctypes.CDLL(LIB_PATH)
    vp, i64, i32, f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_float
    shp = ctypes.POINTER(Shape)
    lib.ttemb_abi_version.restype = ctypes.c_int
    if lib.ttemb_abi_version() != ABI_VERSION:   # (before any other symbol is bound: a stale library fails HERE, with this message)
        raise ImportError(f"{LIB_PATH}: ABI version {lib.ttemb_abi_version()}, this binding needs {ABI_VERSION} -- rebuild the library "
                          "(`make -C falcon-ttdforgnns_amd/csrc`)")
    lib.ttemb_last_error.restype = ctypes.c_char_p
    lib.ttemb_workspace_bytes.restype = i64
    lib.ttemb_workspace_bytes.argtypes = [shp, i32, i64, i64]
    lib.ttemb_set_path.argtypes = [i32]
    lib.ttemb_profile_enable.argtypes = [i32]
    lib.ttemb_profile_read.argtypes = [i32, ctypes.POINTER(ctypes.c_float)]
    lib.ttemb_kernel_family.argtypes = [shp, i64, i64, i32]
    lib.ttemb_set_piece_limits.argtypes = [i64, i64]
    lib.ttemb_set_spin_limit.argtypes = [i64]
    lib.ttemb_set_wide_slab_min_ids.argtypes = [i64]
    lib.ttemb_status.argtypes = []
    lib.ttemb_init.argtypes = []
    lib.ttemb_plan_bytes.restype = i64
    lib.ttemb_plan_bytes.argtypes = [shp, i64]
    lib.ttemb_forward.argtypes = [shp, vp, vp, vp, vp, i64, vp, i64, vp, vp, i64, vp, i64, vp]
    lib.ttemb_forward_group.argtypes = lib.ttemb_forward.argtypes
    lib.ttemb_forward_lookup.argtypes = lib.ttemb_forward.argtypes
    lib.ttemb_backward_dense.argtypes = [shp, vp, vp, vp, vp, i64, vp, i64, vp, vp, vp, i64, vp, i64, vp]
    lib.ttemb_backward_sgd.argtypes = [shp, vp, vp, vp, vp, i64, vp, i64, vp, f32, vp, i64, vp, i64, vp]
    lib.ttemb_backward_adagrad.argtypes = [shp, vp, vp, vp, vp, vp, i64, vp, i64, vp, f32, f32, vp, i64, vp, i64, vp]
    lib.ttemb_window_workspace_bytes.restype = i64
    lib.ttemb_window_workspace_bytes.argtypes = [shp, i32, i64, i64, i64]
    lib.ttemb_forward_window.argtypes = [shp, vp, vp, vp, i64, i64, i64, i64, vp, vp, i64, vp]
    lib.ttemb_backward_dense_window.argtypes = [shp, vp, vp, vp, i64, i64, i64, i64, vp, vp, vp, i64, vp]
    lib.ttemb_backward_sgd_window.argtypes = [shp, vp, vp, vp, i64, i64, i64, i64, vp, f32, vp, i64, vp]
    lib.ttemb_backward_adagrad_window.argtypes = [shp, vp, vp, vp, vp, i64, i64, i64, i64, vp, f32, f32, vp, i64, vp]
    lib.ttemb_sgd_step.argtypes = [vp, vp, i64, f32, vp]
    lib.ttemb_sgd_step_guarded.argtypes = [vp, vp, i64, f32, vp, vp]
    lib.ttemb_adagrad_step.argtypes = [vp, vp, vp, i64, f32, f32, vp]
    lib.ttemb_cache_update.argtypes = [vp, i64, vp, vp, i64, vp]
    lib.ttemb_cache_update_one_sweep.argtypes = [vp, i64, vp, vp, i64, vp]
    lib.ttemb_cache_populate.argtypes = [shp, vp, vp, vp, vp, i64, vp, i64, vp, i64, vp]
    lib.ttemb_preprocess.argtypes = [vp, vp, i64, i64, i32, vp, vp, i64, vp, vp, vp, vp, vp, i32, vp, i64, vp]
    lib.ttemb_preprocess_update.argtypes = [vp, vp, i64, i64, vp, vp, vp, i64, vp, vp, vp, vp, vp, vp, i64, vp]
    lib.ttemb_cache_forward.argtypes = [vp, vp, vp, i64, vp, i64, vp, i64, vp, vp]
    lib.ttemb_cache_backward_sgd.argtypes = [vp, vp, i64, vp, i64, vp, i64, f32, vp, vp, vp]
    lib.ttemb_cache_backward_dense.argtypes = [vp, vp, i64, vp, i64, vp, i64, i64, vp, vp, vp]
    lib.ttemb_cache_backward_rowwise_adagrad.argtypes = [vp, vp, i64, vp, i64, vp, i64, f32, f32, vp, vp, vp]
    for name in EXPORTED_SYMBOLS:
        fn = getattr(lib, name)
        if name not in ("ttemb_last_error", "ttemb_workspace_bytes", "ttemb_plan_bytes", "ttemb_window_workspace_bytes"):
            fn.restype = ctypes.c_int
    return lib


LIB = _load()


def _check(rc: int) -> None:
    if rc != 0:
        raise RuntimeError(f"libttemb_hip: {LIB.ttemb_last_error().decode()} (status {rc})")


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("TT embedding kernels need tensors on a ROCm device; there is no CPU fallback")
    if not t.is_contiguous():
        raise RuntimeError("TT embedding kernels need contiguous tensors")
    return t.data_ptr() if t.numel() > 0 else None


def _ptr_array(ts):
    if isinstance(ts, ctypes.Array):   # already a pointer array (core_ptrs)
        return ts
    arr = (ctypes.c_void_p * MAX_CORES)()
    for i, t in enumerate(ts):
        if t.dtype != torch.float32:
            raise RuntimeError("TT cores must be float32")
        arr[i] = _ptr(t)
    return arr


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(ref: torch.Tensor) -> int:
    """Raw hipStream_t of torch's current stream on the tensor's device."""
    if _raw_stream is not None:
        return _raw_stream(ref.device.index)
    return torch.cuda.current_stream(ref.device).cuda_stream


class _on_device:
    """`with torch.cuda.device(dev)` only when dev is not already current (the common case costs ~0)."""

    def __init__(self, dev: torch.device) -> None:
        self.ctx = None if dev.index == torch.cuda.current_device() else torch.cuda.device(dev)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)


class Workspace:
    """Grow-only scratch buffer owned by the caller (the library never allocates).

    A buffer that was handed out while a HIP graph was being captured is baked into that graph: when a later, larger call
    makes the workspace grow, such a buffer is retired (kept alive for the life of this object) instead of freed, so a
    replay never writes into memory the allocator has given to someone else."""

    def __init__(self) -> None:
        self.buf: Optional[torch.Tensor] = None
        self._in_a_graph = False
        self._retired: List[torch.Tensor] = []

    def get(self, nbytes: int, device: torch.device) -> torch.Tensor:
        if not _initialised:
            init()
        nbytes = max(int(nbytes), 256)
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != device:
            if self.buf is not None and self._in_a_graph:
                self._retired.append(self.buf)
                self._in_a_graph = False
            self.buf = torch.empty(nbytes + nbytes // 4, dtype=torch.uint8, device=device)
        if not self._in_a_graph and torch.cuda.is_current_stream_capturing():
            self._in_a_graph = True
        return self.buf


_initialised = False


def init() -> None:
    """``ttemb_init()``: the pinned host word an expired device-side wait reports to -- the library's one allocation, made
    here and never inside a lookup.  Called by the first ``Workspace.get`` of the process that is not under a stream capture
    (every lookup through this module asks its workspace first) and by ``TTEmbeddingBag.capture`` before it captures; a
    failure is not an error of the caller's (such a process reports expired waits through NaN results only)."""
    global _initialised
    if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        return   # not now: no allocation during a capture; the next eager call tries again
    _initialised = True
    LIB.ttemb_init()


_size_cache: dict = {}   # (kind, shape bytes, op, nnz, B) -> bytes; emptied when the kernel family changes


path_epoch = 0   # bumped when the kernel family changes: callers that keep their own size caches compare it


def set_path(path: int) -> None:
    global path_epoch
    _check(LIB.ttemb_set_path(path))
    _size_cache.clear()
    path_epoch += 1


def profile_enable(on: bool) -> None:
    _check(LIB.ttemb_profile_enable(1 if on else 0))


def profile_read(which: int) -> float:
    """Milliseconds of the most recent forward (0) / backward (1) chain kernel."""
    ms = ctypes.c_float(0.0)
    _check(LIB.ttemb_profile_read(which, ctypes.byref(ms)))
    return float(ms.value)


def _shape_key(shape: Optional[Shape]):
    if shape is None:
        return None
    k = getattr(shape, "_key", None)
    if k is None:
        k = shape._key = bytes(shape)
    return k


def workspace_bytes(shape: Optional[Shape], op: int, nnz: int, B: int) -> int:
    key = ("w", _shape_key(shape), op, nnz, B)
    n = _size_cache.get(key)
    if n is None:
        n = LIB.ttemb_workspace_bytes(ctypes.byref(shape) if shape is not None else None, op, nnz, B)
        if n < 0:
            _check(int(n))
        n = _size_cache[key] = int(n)
        if len(_size_cache) > 4096:
            _size_cache.clear()
    return n


def plan_bytes(shape: Shape, nnz: int) -> int:
    key = ("p", _shape_key(shape), nnz)
    n = _size_cache.get(key)
    if n is None:
        n = LIB.ttemb_plan_bytes(ctypes.byref(shape), nnz)
        if n < 0:
            _check(int(n))
        n = _size_cache[key] = int(n)
    return n


FAMILY_SCALAR, FAMILY_PER_BAG, FAMILY_PER_BAG_RT, FAMILY_GROUPED, FAMILY_GROUPED_WIDE, FAMILY_MERGED, FAMILY_PADDED = 0, 1, 2, 3, 4, 16, 32
FAMILY_PREFIX_IN_CHAIN = 64   # | on FAMILY_GROUPED: a whole forward of this size forms the prefix products in its chain kernel
FAMILY_GROUP_PRODUCTS_IN_CHAIN = 128   # | on FAMILY_GROUPED: a backward of this size forms the per-group products in its chunk kernel
FAMILY_ROUTE_FLAGS = FAMILY_PREFIX_IN_CHAIN | FAMILY_GROUP_PRODUCTS_IN_CHAIN   # which kernels of the grouped family a call of this size takes


def kernel_family(shape: Shape, nnz: int, B: int, ids_with_offsets: bool = True) -> int:
    """Which kernels a lookup of this size would run (``FAMILY_*``, ``| FAMILY_MERGED`` for a 2- / 4-core table on a
    3-core view) under the current ``set_path``; launches nothing."""
    rc = LIB.ttemb_kernel_family(ctypes.byref(shape), nnz, B, 1 if ids_with_offsets else 0)
    if rc < 0:
        _check(rc)
    return rc


def set_spin_limit(tries: int = 0) -> None:
    """Diagnostic: tries of the grouping pass's bounded device-side waits (0 = default, negative = none: every wait expires)."""
    _check(LIB.ttemb_set_spin_limit(tries))


def status() -> None:
    """Raise ``RuntimeError`` when a device-side wait of an earlier grouped lookup ran out (its results are NaN); consumes
    the fault.  The caller synchronises first when it wants the answer for everything it has enqueued."""
    _check(LIB.ttemb_status())


def set_piece_limits(rows: int = 0, ids: int = 0) -> None:
    """Diagnostic: cut calls into pieces of at most ``rows`` bags / ``ids`` ids (0 = the hardware's limits)."""
    global path_epoch
    _check(LIB.ttemb_set_piece_limits(rows, ids))
    _size_cache.clear()
    path_epoch += 1


def set_wide_slab_min_ids(ids: int = 0) -> None:
    """Diagnostic: from how many ids on the wide-rank backward reduces dG2 in LDS (0 = the rule, 1 = always)."""
    global path_epoch
    _check(LIB.ttemb_set_wide_slab_min_ids(ids))
    _size_cache.clear()
    path_epoch += 1


def new_plan(shape: Shape, nnz: int, device: torch.device) -> Optional[torch.Tensor]:
    """Buffer in which forward leaves its id grouping for the matching backward (None if unused)."""
    n = plan_bytes(shape, nnz)
    return torch.empty(n, dtype=torch.uint8, device=device) if n > 0 else None


def _plan_args(plan: Optional[torch.Tensor]):
    return (None, 0) if plan is None else (plan.data_ptr(), plan.numel())


def forward(shape: Shape, cores: Sequence[torch.Tensor], indices: torch.Tensor, rowidx: torch.Tensor,
            offsets: Optional[torch.Tensor], nnz: int, nnz_dev: Optional[torch.Tensor], B: int,
            output: torch.Tensor, ws: Workspace, plan: Optional[torch.Tensor] = None, phase: int = 0) -> None:
    """phase 0 = the whole forward; 1 = the id-only half (grouping into `plan`); 2 = the lookup on that plan."""
    dev = output.device
    w = ws.get(workspace_bytes(shape, OP_FORWARD, nnz, B), dev)
    fn = (LIB.ttemb_forward, LIB.ttemb_forward_group, LIB.ttemb_forward_lookup)[phase]
    with _on_device(dev):
        _check(fn(ctypes.byref(shape), _ptr_array(cores), _ptr(indices), _ptr(rowidx),
                  _ptr(offsets), nnz, _ptr(nnz_dev), B, _ptr(output), _ptr(w), w.numel(),
                  *_plan_args(plan), _stream(output)))


def backward_dense(shape: Shape, cores: Sequence[torch.Tensor], indices, rowidx, nnz: int, nnz_dev, B: int,
                   d_output: torch.Tensor, d_cores: Sequence[torch.Tensor], ws: Workspace,
                   plan: Optional[torch.Tensor] = None, offsets: Optional[torch.Tensor] = None) -> None:
    dev = d_output.device
    w = ws.get(workspace_bytes(shape, OP_BACKWARD, nnz, B), dev)
    with _on_device(dev):
        _check(LIB.ttemb_backward_dense(ctypes.byref(shape), _ptr_array(cores), _ptr(indices), _ptr(rowidx),
                                        _ptr(offsets), nnz, _ptr(nnz_dev), B, _ptr(d_output), _ptr_array(d_cores), _ptr(w),
                                        w.numel(), *_plan_args(plan), _stream(d_output)))


def backward_sgd(shape: Shape, cores, indices, rowidx, nnz: int, nnz_dev, B: int, d_output, lr: float,
                 ws: Workspace, plan: Optional[torch.Tensor] = None, offsets: Optional[torch.Tensor] = None) -> None:
    dev = d_output.device
    w = ws.get(workspace_bytes(shape, OP_BACKWARD, nnz, B), dev)
    with _on_device(dev):
        _check(LIB.ttemb_backward_sgd(ctypes.byref(shape), _ptr_array(cores), _ptr(indices), _ptr(rowidx),
                                      _ptr(offsets), nnz,
                                      _ptr(nnz_dev), B, _ptr(d_output), lr, _ptr(w), w.numel(),
                                      *_plan_args(plan), _stream(d_output)))


def backward_adagrad(shape: Shape, cores, opt_state, indices, rowidx, nnz: int, nnz_dev, B: int, d_output,
                     lr: float, eps: float, ws: Workspace, plan: Optional[torch.Tensor] = None,
                     offsets: Optional[torch.Tensor] = None) -> None:
    dev = d_output.device
    w = ws.get(workspace_bytes(shape, OP_BACKWARD, nnz, B), dev)
    with _on_device(dev):
        _check(LIB.ttemb_backward_adagrad(ctypes.byref(shape), _ptr_array(cores), _ptr_array(opt_state),
                                          _ptr(indices), _ptr(rowidx), _ptr(offsets), nnz, _ptr(nnz_dev), B,
                                          _ptr(d_output),
                                          lr, eps, _ptr(w), w.numel(), *_plan_args(plan), _stream(d_output)))


def sgd_step(weights: torch.Tensor, grads: torch.Tensor, lr: float) -> None:
    with _on_device(weights.device):
        _check(LIB.ttemb_sgd_step(_ptr(weights), _ptr(grads), weights.numel(), lr, _stream(weights)))


HEADER_POISON_OFFSET = 32784   # TTEMB_HEADER_POISON_OFFSET


def sgd_step_guarded(weights: torch.Tensor, grads: torch.Tensor, lr: float, skip: torch.Tensor) -> None:
    """``weights -= lr * grads`` unless the device float ``skip[0]`` is non-zero (then nothing is written)."""
    with _on_device(weights.device):
        _check(LIB.ttemb_sgd_step_guarded(_ptr(weights), _ptr(grads), weights.numel(), lr, _ptr(skip), _stream(weights)))


def poison_word(ws: "Workspace") -> Optional[torch.Tensor]:
    """int32[1] view of the word the last GROUPED backward on this workspace left in its header (1 = poisoned plan and the
    host hears of it); None before the workspace exists.  Meaningful only right after a grouped backward."""
    if ws.buf is None or ws.buf.numel() < HEADER_POISON_OFFSET + 4:
        return None
    return ws.buf[HEADER_POISON_OFFSET:HEADER_POISON_OFFSET + 4].view(torch.int32)


def adagrad_step(weights, state, grads, lr: float, eps: float) -> None:
    with _on_device(weights.device):
        _check(LIB.ttemb_adagrad_step(_ptr(weights), _ptr(state), _ptr(grads), weights.numel(), lr, eps,
                                      _stream(weights)))


E_UNSUPPORTED = -3


def window_workspace_bytes(shape: Shape, op: int, nnz: int, bags_total: int, B: int) -> int:
    """Bytes a window call needs, or -1 when the grouped kernels do not serve such a window (the caller then splits the id
    list on the host: one plain call per table)."""
    key = ("win", _shape_key(shape), op, nnz, bags_total, B, path_epoch)
    n = _size_cache.get(key)
    if n is None:
        n = int(LIB.ttemb_window_workspace_bytes(ctypes.byref(shape), op, nnz, bags_total, B))
        if n < 0 and n != E_UNSUPPORTED:
            _check(n)
        n = _size_cache[key] = (-1 if n < 0 else n)
    return n


def forward_window(shape: Shape, cores: Sequence[torch.Tensor], indices: torch.Tensor, offsets: torch.Tensor, bag0: int, B: int,
                   output: torch.Tensor, ws: Workspace) -> None:
    """One table of a table-batched call: the bags [bag0, bag0 + B) of ``offsets`` (the whole call's) and their ids;
    ``output`` is the [bags_total, D] tensor of the whole call.  No host synchronisation."""
    nnz, bags = indices.numel(), offsets.numel() - 1
    dev = output.device
    w = ws.get(window_workspace_bytes(shape, OP_FORWARD, nnz, bags, B), dev)
    with _on_device(dev):
        _check(LIB.ttemb_forward_window(ctypes.byref(shape), _ptr_array(cores), _ptr(indices), _ptr(offsets), nnz, bags, bag0, B,
                                        _ptr(output), _ptr(w), w.numel(), _stream(output)))


def backward_window(shape: Shape, cores: Sequence[torch.Tensor], indices: torch.Tensor, offsets: torch.Tensor, bag0: int, B: int,
                    d_output: torch.Tensor, ws: Workspace, d_cores: Optional[Sequence[torch.Tensor]] = None,
                    opt_state: Optional[Sequence[torch.Tensor]] = None, lr: float = 0.0, eps: float = 0.0) -> None:
    """``d_cores``: dense gradients of the window's table; else the fused step (Adagrad when ``opt_state`` is given)."""
    nnz, bags = indices.numel(), offsets.numel() - 1
    dev = d_output.device
    w = ws.get(window_workspace_bytes(shape, OP_BACKWARD, nnz, bags, B), dev)
    head = (ctypes.byref(shape), _ptr_array(cores))
    ids = (_ptr(indices), _ptr(offsets), nnz, bags, bag0, B, _ptr(d_output))
    with _on_device(dev):
        if d_cores is not None:
            _check(LIB.ttemb_backward_dense_window(*head, *ids, _ptr_array(d_cores), _ptr(w), w.numel(), _stream(d_output)))
        elif opt_state is None:
            _check(LIB.ttemb_backward_sgd_window(*head, *ids, lr, _ptr(w), w.numel(), _stream(d_output)))
        else:
            _check(LIB.ttemb_backward_adagrad_window(*head, _ptr_array(opt_state), *ids, lr, eps, _ptr(w), w.numel(), _stream(d_output)))


def cache_update(indices: torch.Tensor, hashtbl: torch.Tensor, cache_freq: torch.Tensor, one_sweep: bool = False) -> None:
    """``one_sweep``: the reference's probe-and-insert in one pass (bit-for-bit its table, including the re-insertion of
    cached ids after evictions) instead of looking the key up in all of its probe slots first."""
    if indices.numel() == 0:
        return
    fn = LIB.ttemb_cache_update_one_sweep if one_sweep else LIB.ttemb_cache_update
    with _on_device(indices.device):
        _check(fn(_ptr(indices), indices.numel(), _ptr(hashtbl), _ptr(cache_freq), hashtbl.numel(), _stream(indices)))


def cache_populate(shape: Shape, cores, hashtbl, cache_freq, cache_state, cache_weight, ws: Workspace) -> None:
    dev = hashtbl.device
    H, C = hashtbl.numel(), cache_weight.shape[0]
    w = ws.get(workspace_bytes(shape, OP_CACHE_POPULATE, H, C), dev)
    with _on_device(dev):
        _check(LIB.ttemb_cache_populate(ctypes.byref(shape), _ptr_array(cores), _ptr(hashtbl), _ptr(cache_freq),
                                        _ptr(cache_state), H, _ptr(cache_weight), C, _ptr(w), w.numel(),
                                        _stream(hashtbl)))


def preprocess(indices, offsets, B: int, warmup: bool, hashtbl, cache_state, indices_out, rowidx_out,
               cache_loc_out, nnz_tt_dev, ws: Workspace, dup_stamp=None, epoch: int = 0, cache_freq=None) -> None:
    """``dup_stamp`` (int32[C] of scratch): ``nnz_tt_dev`` must hold two int32 and its second word tells the cache
    backward whether a cache row occurs twice in this call.  ``cache_freq``: the LFU update of the same ids rides in
    the probe pass (``ttemb_preprocess_update``; live cache only)."""
    dev = indices.device
    nnz = indices.numel()
    H = 0 if hashtbl is None else hashtbl.numel()
    need = 0 if (warmup or H == 0) else workspace_bytes(None, OP_PREPROCESS, nnz, B)
    w = ws.get(need, dev)
    if cache_freq is not None:
        if warmup or H == 0:
            raise RuntimeError("the fused probe pass serves a live cache only")
        with _on_device(dev):
            _check(LIB.ttemb_preprocess_update(_ptr(indices), _ptr(offsets), nnz, B, _ptr(hashtbl), _ptr(cache_freq),
                                               _ptr(cache_state), H, _ptr(indices_out), _ptr(rowidx_out),
                                               _ptr(cache_loc_out), _ptr(nnz_tt_dev), _ptr(dup_stamp), _ptr(w), w.numel(),
                                               _stream(indices)))
        return
    with _on_device(dev):
        _check(LIB.ttemb_preprocess(_ptr(indices), _ptr(offsets), nnz, B, 1 if warmup else 0, _ptr(hashtbl),
                                    _ptr(cache_state), H, _ptr(indices_out), _ptr(rowidx_out),
                                    _ptr(cache_loc_out), _ptr(nnz_tt_dev), _ptr(dup_stamp), epoch, _ptr(w), w.numel(),
                                    _stream(indices)))


def cache_forward(cache_loc, rowidx, start: int, start_dev, nnz: int, cache_weight, output,
                  offsets: Optional[torch.Tensor] = None) -> None:
    with _on_device(output.device):
        _check(LIB.ttemb_cache_forward(_ptr(cache_loc), _ptr(rowidx), _ptr(offsets), start, _ptr(start_dev), nnz,
                                       _ptr(cache_weight), cache_weight.shape[1], _ptr(output),
                                       _stream(output)))


def cache_backward_sgd(cache_loc, rowidx, start: int, start_dev, nnz: int, d_output, lr: float,
                       cache_weight, dup_dev=None) -> None:
    with _on_device(d_output.device):
        _check(LIB.ttemb_cache_backward_sgd(_ptr(cache_loc), _ptr(rowidx), start, _ptr(start_dev), nnz,
                                            _ptr(d_output), cache_weight.shape[1], lr, _ptr(cache_weight),
                                            _ptr(dup_dev), _stream(d_output)))


def cache_backward_dense(cache_loc, rowidx, start: int, start_dev, nnz: int, d_output,
                         d_cache_weight, dup_dev=None) -> None:
    with _on_device(d_output.device):
        _check(LIB.ttemb_cache_backward_dense(_ptr(cache_loc), _ptr(rowidx), start, _ptr(start_dev), nnz,
                                              _ptr(d_output), d_cache_weight.shape[1],
                                              d_cache_weight.shape[0], _ptr(d_cache_weight),
                                              _ptr(dup_dev), _stream(d_output)))


def cache_backward_rowwise_adagrad(cache_loc, rowidx, start: int, start_dev, nnz: int, d_output, lr: float,
                                   eps: float, state_sum, cache_weight) -> None:
    with _on_device(d_output.device):
        _check(LIB.ttemb_cache_backward_rowwise_adagrad(_ptr(cache_loc), _ptr(rowidx), start, _ptr(start_dev),
                                                        nnz, _ptr(d_output), cache_weight.shape[1], lr, eps,
                                                        _ptr(state_sum), _ptr(cache_weight),
                                                        _stream(d_output)))


def core_ptrs(tt_cores: Sequence[torch.Tensor], table: int = 0):
    """Pointer array of the per-table [p_t, row] slices of [num_tables, p_t, row] parameters: what core_views gives,
    without creating view tensors (this sits on the host path of every step)."""
    arr = (ctypes.c_void_p * MAX_CORES)()
    for i, c in enumerate(tt_cores):
        if c.dtype != torch.float32 or not c.is_cuda or not c.is_contiguous():
            raise RuntimeError("tt_cores must be contiguous float32 tensors on a ROCm device (no CPU fallback)")
        arr[i] = c.data_ptr() + (table * c.stride(0) * 4 if c.dim() == 3 and table else 0)
    return arr


def core_views(tt_cores: Sequence[torch.Tensor], table: int = 0) -> List[torch.Tensor]:
    """[num_tables, p_t, row] parameters -> contiguous per-table [p_t, row] views."""
    out = []
    for c in tt_cores:
        v = c.data if isinstance(c, torch.nn.Parameter) else c
        v = v[table] if v.dim() == 3 else v
        if not v.is_contiguous():
            raise RuntimeError("tt_cores must be contiguous")
        out.append(v)
    return out


class LeanCalls:
    """The two native calls of a training step (forward, fused-optimiser backward) of ONE single-table module with their
    invariant arguments bound once: shape reference, core / optimiser-state pointer arrays (refreshed only when a
    parameter's storage moved), workspace and plan sizes per (nnz, B).  At 2 048 ids the GPU work of a step is ~25 us, so
    every microsecond of Python between the two launches is step time."""

    def __init__(self, shape: Shape, ws: Workspace) -> None:
        self.shape, self.shape_ref, self.ws = shape, ctypes.byref(shape), ws
        self.sizes: dict = {}
        self.epoch = -1
        self.core_key, self.core_arr = None, None
        self.state_key, self.state_arr = None, None
        self.grad_key, self.grad_arr = None, None
        self.grouped: dict = {}

    def _entry(self, nnz: int, B: int):
        if self.epoch != path_epoch:
            self.sizes.clear()
            self.grouped.clear()
            self.epoch = path_epoch
        e = self.sizes.get((nnz, B))
        if e is None:
            if len(self.sizes) > 1024:
                self.sizes.clear()
                self.grouped.clear()
            e = self.sizes[(nnz, B)] = (workspace_bytes(self.shape, OP_FORWARD, nnz, B),
                                        workspace_bytes(self.shape, OP_BACKWARD, nnz, B), plan_bytes(self.shape, nnz))
        return e

    @staticmethod
    def _ptrs(tensors, key, arr):
        k = tuple(t.data_ptr() for t in tensors)
        if k != key:
            arr = core_ptrs(tensors)   # validates dtype / device / contiguity
            key = k
        return key, arr

    def forward(self, cores, indices, offsets, nnz: int, B: int, out, keep_plan: bool = True):
        """``keep_plan=False``: a forward whose backward never comes (inference): the plan lives and dies in the workspace."""
        fwd_ws, _, plan_n = self._entry(nnz, B)
        self.core_key, self.core_arr = self._ptrs(cores, self.core_key, self.core_arr)
        dev = out.device
        w = self.ws.get(fwd_ws, dev)
        plan = torch.empty(plan_n, dtype=torch.uint8, device=dev) if plan_n > 0 and keep_plan else None
        with _on_device(dev):
            rc = LIB.ttemb_forward(self.shape_ref, self.core_arr, indices.data_ptr() if nnz else None, None, offsets.data_ptr(),
                                   nnz, None, B, out.data_ptr() if B else None, w.data_ptr(), w.numel(),
                                   plan.data_ptr() if plan is not None else None, plan_n if plan is not None else 0, _stream(out))
        if rc:
            _check(rc)
        return plan

    def forward_split(self, cores, indices, offsets, nnz: int, B: int, out, pending):
        """The forward of a data-parallel step (ttemb_dist): the id-only half (grouping into the plan) is enqueued, THEN
        ``pending()`` finishes the previous step's all-reduce + update, then the half that reads the cores.  A call without a
        plan (per-bag / scalar kernels, a call in pieces) has no id-only half: ``pending()``, then the whole forward."""
        fwd_ws, _, plan_n = self._entry(nnz, B)
        self.core_key, self.core_arr = self._ptrs(cores, self.core_key, self.core_arr)
        dev = out.device
        w = self.ws.get(fwd_ws, dev)
        plan = torch.empty(plan_n, dtype=torch.uint8, device=dev) if plan_n > 0 else None
        args = (self.shape_ref, self.core_arr, indices.data_ptr() if nnz else None, None, offsets.data_ptr(), nnz, None, B,
                out.data_ptr() if B else None, w.data_ptr(), w.numel(), plan.data_ptr() if plan is not None else None, plan_n,
                _stream(out))
        with _on_device(dev):
            if plan is None:
                pending()
                rc = LIB.ttemb_forward(*args)
            else:
                rc = LIB.ttemb_forward_group(*args)
                if rc == 0:
                    pending()
                    rc = LIB.ttemb_forward_lookup(*args)
        if rc:
            _check(rc)
        return plan

    def backward_dense(self, cores, indices, offsets, nnz: int, B: int, d_output, grads, plan) -> bool:
        """Dense core gradients into ``grads`` (a fixed list of tensors: their pointer array is cached).  Returns whether
        the gradient came from the grouped kernels (the family whose last kernel leaves its verdict in the workspace header)."""
        _, bwd_ws, plan_n = self._entry(nnz, B)
        self.core_key, self.core_arr = self._ptrs(cores, self.core_key, self.core_arr)
        self.grad_key, self.grad_arr = self._ptrs(grads, self.grad_key, self.grad_arr)
        dev = d_output.device
        w = self.ws.get(bwd_ws, dev)
        pp, pn = (plan.data_ptr(), plan_n) if plan is not None else (None, 0)
        with _on_device(dev):
            rc = LIB.ttemb_backward_dense(self.shape_ref, self.core_arr, indices.data_ptr() if nnz else None, None, offsets.data_ptr(),
                                          nnz, None, B, d_output.data_ptr() if B else None, self.grad_arr, w.data_ptr(), w.numel(),
                                          pp, pn, _stream(d_output))
        if rc:
            _check(rc)
        g = self.grouped.get((nnz, B))
        if g is None:
            g = self.grouped[(nnz, B)] = nnz > 0 and (kernel_family(self.shape, nnz, B, True) & 7) in (FAMILY_GROUPED, FAMILY_GROUPED_WIDE)
        return g

    def backward(self, cores, state, indices, offsets, nnz: int, B: int, d_output, lr: float, eps: float, plan):
        _, bwd_ws, plan_n = self._entry(nnz, B)
        self.core_key, self.core_arr = self._ptrs(cores, self.core_key, self.core_arr)
        dev = d_output.device
        w = self.ws.get(bwd_ws, dev)
        pp, pn = (plan.data_ptr(), plan_n) if plan is not None else (None, 0)
        ids = indices.data_ptr() if nnz else None
        with _on_device(dev):
            if state is None:
                rc = LIB.ttemb_backward_sgd(self.shape_ref, self.core_arr, ids, None, offsets.data_ptr(), nnz, None, B,
                                            d_output.data_ptr() if B else None, lr, w.data_ptr(), w.numel(), pp, pn, _stream(d_output))
            else:
                self.state_key, self.state_arr = self._ptrs(state, self.state_key, self.state_arr)
                rc = LIB.ttemb_backward_adagrad(self.shape_ref, self.core_arr, self.state_arr, ids, None, offsets.data_ptr(), nnz, None,
                                                B, d_output.data_ptr() if B else None, lr, eps, w.data_ptr(), w.numel(), pp, pn,
                                                _stream(d_output))
        if rc:
            _check(rc)
