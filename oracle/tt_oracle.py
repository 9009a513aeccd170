"""CPU oracle for the TT-embedding hot path.  TEST INFRASTRUCTURE ONLY.

This file is a numpy restatement of what the reference's TT embedding layer
computes.  It exists to check the HIP path; the product never imports it.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import anything under ``oracle/``.

Parity pin: the forward/backward restatements below are checked, in
``tests/golden/make_golden.py`` (run in the build container, where /root/reference is
mounted), against the reference's own pure-PyTorch ``tt_matrix_to_full``
(FBTT/tt_embeddings_ops.py:80-127) and autograd through it.  The resulting
vectors are committed under ``tests/golden/``.  The reference ships no
executable test or golden vector for this path, and its CUDA
kernels cannot be built here (nvcc/cuBLAS/CUB absent), so those generated
vectors are the pin.

Conventions (reference file:line in brackets):
  * cores[t] is float32 [p_t, R_t*q_t*R_{t+1}] (the reference stores
    [num_tables=1, p_t, ...]; squeeze the table axis before calling).  A core
    row is a row-major [R_t][q_t][R_{t+1}] block
    [tt_embeddings_ops.py:528-545, permute [1,0,2,3] at :617-627].
  * idx -> (i_0..i_{T-1}) by repeated div/mod with L = [p1*p2, p2, 1]
    [tt_embeddings_ops.py:519-527; tt_embeddings_cuda.cu:796-802].
  * row(idx) = chain of (Q_{t-1} x R_t) @ (R_t x q_t R_{t+1}) products
    [tt_embeddings_cuda.cu:1002-1008, 1045-1061].
  * bags: output[b] = sum of rows of indices[offsets[b]:offsets[b+1]]
    [tt_embeddings_cuda.cu:923-965, 1349-1365].
"""
from __future__ import annotations

import numpy as np

MAX_PROBES = 3  # tt_embeddings_cuda.cu:31
UNUSED_KEY = -1  # hashtbl_cuda_utils.cuh:100


# --------------------------------------------------------------------------
# index arithmetic
# --------------------------------------------------------------------------
def strides_L(p_shapes):
    """L[t] = prod(p[t+1:])  [tt_embeddings_ops.py:519-527]."""
    L = [1] * len(p_shapes)
    for t in range(len(p_shapes) - 2, -1, -1):
        L[t] = L[t + 1] * int(p_shapes[t + 1])
    return L


def full_ranks(tt_ranks, T):
    """[r1..r_{T-1}] -> [1, r1, .., r_{T-1}, 1]  [tt_embeddings_ops.py:501]."""
    r = [int(x) for x in tt_ranks]
    if len(r) == T - 1:
        r = [1] + r + [1]
    assert len(r) == T + 1
    return r


def split_index(indices, p_shapes):
    """int64 ids -> list of T int64 arrays  [tt_embeddings_cuda.cu:796-802]."""
    rem = np.asarray(indices, dtype=np.int64).copy()
    out = []
    for Lt in strides_L(p_shapes):
        out.append(rem // Lt)
        rem = rem % Lt
    return out


def rowidx_from_offsets(offsets, nnz):
    """rowidx[l] = bag of position l  [tt_embeddings_cuda.cu:1349-1365]."""
    offsets = np.asarray(offsets, dtype=np.int64)
    B = offsets.shape[0] - 1
    lens = np.diff(offsets)
    row = np.repeat(np.arange(B, dtype=np.int64), lens)
    assert row.shape[0] == nnz
    return row


# --------------------------------------------------------------------------
# forward
# --------------------------------------------------------------------------
def tt_rows(indices, cores, p_shapes, q_shapes, ranks):
    """One embedding row per id: float32 [nnz, D].

    Batched form of the GEMM chain of tt_embeddings_cuda.cu:1045-1061 (fp32
    multiply-accumulate, like cublasGemmBatchedEx with CUDA_R_32F).
    """
    T = len(p_shapes)
    R = full_ranks(ranks, T)
    ii = split_index(indices, p_shapes)
    n = ii[0].shape[0]
    v = cores[0][ii[0]].reshape(n, int(q_shapes[0]), R[1]).astype(np.float32)
    for t in range(1, T):
        g = cores[t][ii[t]].reshape(n, R[t], int(q_shapes[t]) * R[t + 1])
        v = np.matmul(v, g)  # [n, Q, q_t R_{t+1}]
        v = v.reshape(n, -1, R[t + 1])
    return v.reshape(n, -1)


def tt_forward(indices, offsets, cores, p_shapes, q_shapes, ranks):
    """Bag-summed lookup: float32 [B, D]  [tt_embeddings_cuda.cu:967-1081]."""
    indices = np.asarray(indices, dtype=np.int64)
    offsets = np.asarray(offsets, dtype=np.int64)
    B = offsets.shape[0] - 1
    D = int(np.prod(q_shapes))
    out = np.zeros((B, D), dtype=np.float32)
    if indices.shape[0] == 0:
        return out
    rows = tt_rows(indices, cores, p_shapes, q_shapes, ranks)
    rowidx = rowidx_from_offsets(offsets, indices.shape[0])
    np.add.at(out, rowidx, rows)
    return out


def tt_full_table(cores, p_shapes, q_shapes, ranks):
    """Dense [prod(p), D] table; small shapes only."""
    n = int(np.prod(p_shapes))
    return tt_rows(np.arange(n, dtype=np.int64), cores, p_shapes, q_shapes, ranks)


# --------------------------------------------------------------------------
# backward (dense core gradients)
# --------------------------------------------------------------------------
def tt_dense_backward(indices, offsets, d_output, cores, p_shapes, q_shapes, ranks,
                      acc_dtype=np.float64):
    """d_core_t, same shapes as cores[t]  [tt_embeddings_cuda.cu:421-654].

    Per id n (bag row r): recompute the forward partials, then walk the chain
    backwards: dG_t[i_t] += v_{t-1}^T dV_t ; dV_{t-1} = dV_t G_t[i_t]^T
    [:531-609], scatter-add over ids [:364-379].  Accumulation is done in
    ``acc_dtype`` (float64 by default: the reference's atomicAdd order is
    undefined, so the oracle gives the exactly-rounded answer and the tests
    carry the tolerance).
    """
    T = len(p_shapes)
    R = full_ranks(ranks, T)
    q = [int(x) for x in q_shapes]
    indices = np.asarray(indices, dtype=np.int64)
    n = indices.shape[0]
    grads = [np.zeros(c.shape, dtype=acc_dtype) for c in cores]
    if n == 0:
        return [g.astype(np.float32) for g in grads]
    rowidx = rowidx_from_offsets(offsets, n)
    ii = split_index(indices, p_shapes)
    # forward partials v[t]: [n, Q_t, R_{t+1}]
    v = [cores[0][ii[0]].reshape(n, q[0], R[1]).astype(acc_dtype)]
    for t in range(1, T - 1):
        g = cores[t][ii[t]].reshape(n, R[t], q[t] * R[t + 1]).astype(acc_dtype)
        v.append(np.matmul(v[-1], g).reshape(n, -1, R[t + 1]))
    dv = np.asarray(d_output, dtype=acc_dtype)[rowidx]  # [n, D]
    for t in range(T - 1, 0, -1):
        a = v[t - 1]  # [n, Q_{t-1}, R_t]
        dc = dv.reshape(n, a.shape[1], q[t] * R[t + 1])
        dg = np.matmul(a.transpose(0, 2, 1), dc)  # [n, R_t, q_t R_{t+1}]
        np.add.at(grads[t], ii[t], dg.reshape(n, -1))
        g = cores[t][ii[t]].reshape(n, R[t], q[t] * R[t + 1]).astype(acc_dtype)
        dv = np.matmul(dc, g.transpose(0, 2, 1))  # [n, Q_{t-1}, R_t]
    np.add.at(grads[0], ii[0], dv.reshape(n, -1))
    return [g.astype(np.float32) for g in grads]


def sgd_step(cores, grads, lr):
    """core -= lr * g on EVERY row  [tt_embeddings_cuda.cu:381-397].

    (The reference's launch at :633-651 sizes the grid by the wrong dimension
    and skips trailing rows; that defect is not reproduced -- SURVEY.md §7-6.)
    """
    lr = np.float32(lr)
    return [(c - lr * g).astype(np.float32) for c, g in zip(cores, grads)]


def adagrad_step(cores, states, grads, lr, eps):
    """state += g^2; core -= lr*g/(sqrt(state)+eps)  [tt_embeddings_cuda.cu:399-419]."""
    lr = np.float32(lr)
    eps = np.float32(eps)
    new_c, new_s = [], []
    for c, s, g in zip(cores, states, grads):
        s2 = (s + g * g).astype(np.float32)
        new_s.append(s2)
        new_c.append((c - lr * g / (np.sqrt(s2) + eps)).astype(np.float32))
    return new_c, new_s


# --------------------------------------------------------------------------
# LFU hash-table cache
# --------------------------------------------------------------------------
_M32 = 0xFFFFFFFF


def _rotl32(x, r):
    return ((x << r) | (x >> (32 - r))) & _M32


def murmur_word(key, len_xor=2):
    """32-bit hash word before range reduction.  With len_xor=8 this is the
    standard MurmurHash3_x86_32(seed 0) of the key's 8 little-endian bytes; the
    reference xors 2 instead (hashtbl_cuda_utils.cuh:66)."""
    key &= 0xFFFFFFFFFFFFFFFF
    h = 0
    for k in (key & _M32, (key >> 32) & _M32):
        k = (k * 0xCC9E2D51) & _M32
        k = _rotl32(k, 15)
        k = (k * 0x1B873593) & _M32
        h ^= k
        h = _rotl32(h, 13)
        h = (h * 5 + 0xE6546B64) & _M32
    h ^= len_xor
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & _M32
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & _M32
    h ^= h >> 16
    return h


def murmur_slot(key, C):
    """hashtbl_cuda_utils.cuh:48-76 (int64 key): bit-exact slot in [0, C)
    via the multiply-shift range reduction of :74-75."""
    return (murmur_word(int(key)) * (C & _M32)) >> 32


def murmur_slots(keys, C):
    """Vectorised murmur_slot for an int64 array."""
    k64 = np.asarray(keys, dtype=np.int64).view(np.uint64)
    h = np.zeros(k64.shape, dtype=np.uint64)
    m = np.uint64(_M32)
    for part in (k64 & m, (k64 >> np.uint64(32)) & m):
        k = (part * np.uint64(0xCC9E2D51)) & m
        k = ((k << np.uint64(15)) | (k >> np.uint64(17))) & m
        k = (k * np.uint64(0x1B873593)) & m
        h = h ^ k
        h = ((h << np.uint64(13)) | (h >> np.uint64(19))) & m
        h = (h * np.uint64(5) + np.uint64(0xE6546B64)) & m
    h = h ^ np.uint64(2)
    h = h ^ (h >> np.uint64(16))
    h = (h * np.uint64(0x85EBCA6B)) & m
    h = h ^ (h >> np.uint64(13))
    h = (h * np.uint64(0xC2B2AE35)) & m
    h = h ^ (h >> np.uint64(16))
    return ((h * np.uint64(C)) >> np.uint64(32)).astype(np.int64)


def hashtbl_find(key, hashtbl):
    """hashtbl_cuda_utils.cuh:135-154: slot of key or -1 (never stops at empties)."""
    H = hashtbl.shape[0]
    s = murmur_slot(int(key), H)
    for _ in range(MAX_PROBES):
        if hashtbl[s] == key:
            return s
        if key == UNUSED_KEY:
            return -1
        s = (s + 1) % H
    return -1


def update_cache_state(indices, hashtbl, cache_freq, find_first=False):
    """Sequential hashtbl_insert<accumulate>(id, 1) over the ids
    [tt_embeddings_cuda.cu:1083-1095; hashtbl_cuda_utils.cuh:102-133].
    Mutates hashtbl / cache_freq in place; returns #failed inserts.
    (On the GPU the insertion order is undefined; the final table equals this
    one whenever no two distinct keys contend for a slot.)

    ``find_first=False`` is the reference: the probe stops at the first empty slot, so once
    ``cache_populate`` has evicted entries a key that sits *behind* such a hole is inserted a second
    time in front of itself -- and ``hashtbl_find`` then returns the new slot, whose cache_state is
    -1: the id silently drops out of the cache (which ids, depends on thread order on the GPU).
    ``find_first=True`` is what the HIP path does: look for the key in all probe slots before
    inserting.  The two are identical as long as nothing was evicted (the whole warm-up).
    """
    H = hashtbl.shape[0]
    failed = 0
    for key in np.asarray(indices, dtype=np.int64).tolist():
        s = murmur_slot(key, H)
        if find_first:
            t = hashtbl_find(key, hashtbl)
            if t >= 0:
                cache_freq[t] += 1
                continue
        for _ in range(MAX_PROBES):
            if hashtbl[s] == UNUSED_KEY:
                hashtbl[s] = key
            if hashtbl[s] == key:
                cache_freq[s] += 1
                break
            s = (s + 1) % H
        else:
            failed += 1
    return failed


def cache_populate(hashtbl, cache_freq, cache_state, cache_size):
    """Rank slots by frequency, keep the top ``cache_size``
    [tt_embeddings_cuda.cu:1270-1347, 1122-1149].

    Stable descending sort of (freq, key) over all H slots (CUB radix sort is
    stable: ties keep slot order).  rank < C -> cache_state[slot] = rank; other
    occupied slots are evicted.  Empty slots ranked < C stand for id 0 (the
    reference's "hack to use batch gemm", :1144-1147).  Returns the int64[C]
    array of ids whose rows fill cache_weight[0:C] (caller computes the rows).
    Mutates hashtbl / cache_freq / cache_state.
    """
    H = hashtbl.shape[0]
    order = np.argsort(-cache_freq, kind="stable")
    sorted_keys = hashtbl[order].copy()
    snapshot = hashtbl.copy()
    for rank in range(H):
        key = sorted_keys[rank]
        if key != UNUSED_KEY:
            slot = hashtbl_find(key, snapshot)
            if rank < cache_size:
                cache_state[slot] = rank
            else:
                hashtbl[slot] = UNUSED_KEY
                cache_freq[slot] = 0
        elif rank < cache_size:
            sorted_keys[rank] = 0
    return sorted_keys[:cache_size].copy()


def cache_lookup(indices, hashtbl, cache_state):
    """is_tt flags + cache rows  [tt_embeddings_cuda.cu:1367-1386]."""
    indices = np.asarray(indices, dtype=np.int64)
    is_tt = np.ones(indices.shape[0], dtype=bool)
    loc = np.full(indices.shape[0], -1, dtype=np.int32)
    for n, key in enumerate(indices.tolist()):
        s = hashtbl_find(key, hashtbl)
        if s != -1 and cache_state[s] != -1:
            is_tt[n] = False
            loc[n] = cache_state[s]
    return is_tt, loc


def partition_by_flag(arr, flags):
    """cub::DevicePartition::Flagged order: selected items first in input order,
    rejected items from the end backwards  [tt_embeddings_cuda.cu:1448-1490]."""
    arr = np.asarray(arr)
    return np.concatenate([arr[flags], arr[~flags][::-1]])


def preprocess_indices(indices, offsets, warmup, hashtbl, cache_state):
    """(indices', rowidx', nnz_tt, cache_locations' or None)
    [tt_embeddings_cuda.cu:1388-1507], single table."""
    indices = np.asarray(indices, dtype=np.int64)
    rowidx = rowidx_from_offsets(offsets, indices.shape[0])
    if warmup or indices.shape[0] == 0:
        return indices, rowidx, indices.shape[0], None
    is_tt, loc = cache_lookup(indices, hashtbl, cache_state)
    return (partition_by_flag(indices, is_tt), partition_by_flag(rowidx, is_tt),
            int(is_tt.sum()), partition_by_flag(loc, is_tt))


def cache_forward(output, cache_locations, rowidx, cache_weight):
    """output[row] += cache_weight[loc]  [tt_embeddings_cuda.cu:1509-1549]."""
    np.add.at(output, np.asarray(rowidx, dtype=np.int64),
              cache_weight[np.asarray(cache_locations, dtype=np.int64)])
    return output


def cache_backward_dense(grad_output, cache_locations, rowidx, cache_rows, D):
    """zeros[C,D] with grad rows scattered in  [tt_embeddings_cuda.cu:1670-1744]."""
    g = np.zeros((cache_rows, D), dtype=np.float64)
    np.add.at(g, np.asarray(cache_locations, dtype=np.int64),
              np.asarray(grad_output, dtype=np.float64)[np.asarray(rowidx, dtype=np.int64)])
    return g.astype(np.float32)


def cache_backward_sgd(grad_output, cache_locations, rowidx, lr, cache_weight):
    """cache_weight[loc] -= lr * grad[row]  [tt_embeddings_cuda.cu:1585-1668]."""
    g = cache_backward_dense(grad_output, cache_locations, rowidx,
                             cache_weight.shape[0], cache_weight.shape[1])
    return (cache_weight - np.float32(lr) * g).astype(np.float32)


def cache_backward_rowwise_adagrad(grad_output, cache_locations, rowidx, lr, eps,
                                   state, cache_weight):
    """Row-wise Adagrad on cached rows, one id at a time in the given order
    [tt_embeddings_cuda.cu:1746-1806].  (Duplicate cache rows in one call race
    on the GPU; the oracle applies them sequentially.)"""
    w = cache_weight.astype(np.float32).copy()
    st = state.astype(np.float32).copy()
    D = w.shape[1]
    g_all = np.asarray(grad_output, dtype=np.float32)
    for loc, row in zip(np.asarray(cache_locations).tolist(), np.asarray(rowidx).tolist()):
        g = g_all[row]
        gsq = np.float32(np.sum(g.astype(np.float64) ** 2) / D)
        old = st[loc]
        st[loc] = old + gsq
        mult = np.float32(lr) * np.float32(1.0 / (np.sqrt(old + gsq) + np.float32(eps)))
        w[loc] = w[loc] - g * mult
    return w, st
