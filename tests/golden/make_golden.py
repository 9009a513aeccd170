#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE itself.

Run in the build container only (needs /root/reference, read-only):

    cd /tmp && python /root/repo/tests/golden/make_golden.py

What is pinned:
  * forward  : reference ``tt_matrix_to_full`` (FBTT/tt_embeddings_ops.py:80-127)
               rows, bag-summed with torch.nn.functional.embedding_bag(sum,
               include_last_offset) -- the recipe of the reference's gutted
               test (sage_profiler.py:262-305).
  * backward : autograd through ``tt_matrix_to_full`` (sage_profiler.py:340-367).
  * sgd/adagrad: closed forms of sage_profiler.py:405-406, 466-477 on those grads.
  * murmur   : known-answer slots from a line-by-line run of
               hashtbl_cuda_utils.cuh:48-76 semantics (pure integer arithmetic).
The reference's compiled extension ``tt_embeddings`` is absent (CUDA only), so an
empty stub module is pre-seeded to let the pure-PyTorch helpers import.

Only data is written: inputs and expected outputs.  No reference source text.
"""
import hashlib
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))

sys.modules.setdefault("tt_embeddings", types.ModuleType("tt_embeddings"))
sys.path.insert(0, REF)
from FBTT.tt_embeddings_ops import tt_matrix_to_full  # noqa: E402  (reference)

from oracle import tt_oracle as orc  # noqa: E402  (ours, checked below)


def seeded_cores(p, q, R, seed, scale=None):
    """float32 cores [p_t, R_t q_t R_{t+1}] from numpy default_rng(seed)."""
    rng = np.random.default_rng(seed)
    cores = []
    for t in range(len(p)):
        c = rng.standard_normal((p[t], R[t] * q[t] * R[t + 1])).astype(np.float32)
        if scale is not None:
            c *= np.float32(scale)
        cores.append(c)
    return cores


def cores_sha256(cores):
    h = hashlib.sha256()
    for c in cores:
        h.update(np.ascontiguousarray(c).tobytes())
    return h.hexdigest()


def ref_full(p, q, R, cores_t):
    return tt_matrix_to_full(list(p), list(q), list(R), [c.unsqueeze(0) for c in cores_t],
                             [1, 0, 2, 3])


def ref_bag_forward(full, indices, offsets):
    return torch.nn.functional.embedding_bag(
        torch.as_tensor(indices), full, torch.as_tensor(offsets), mode="sum",
        include_last_offset=True)


def ragged_bags(rng, B, n_emb, mean_len, unique=False):
    """Variable-length bags incl. empty ones (cf. sage_profiler.py:71-100)."""
    lens = np.clip(np.round(rng.normal(mean_len, mean_len, B)), 0, None).astype(np.int64)
    lens[rng.integers(0, B)] = 0  # at least one empty bag
    nnz = int(lens.sum())
    idx = rng.choice(n_emb, size=nnz, replace=not unique).astype(np.int64)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    return idx, offsets


def tiny_case(name, p, q, ranks, seed):
    T = len(p)
    R = orc.full_ranks(ranks, T)
    n_emb = int(np.prod(p))
    D = int(np.prod(q))
    cores = seeded_cores(p, q, R, seed, scale=0.5)
    rng = np.random.default_rng(seed + 1)
    idx, offsets = ragged_bags(rng, 23, n_emb, 3.0)
    B = offsets.shape[0] - 1
    d_out = (rng.random((B, D)).astype(np.float32) * np.float32(0.1))

    cores_t = [torch.tensor(c, requires_grad=True) for c in cores]
    full = ref_full(p, q, R, cores_t)
    out = ref_bag_forward(full, idx, offsets)
    out.backward(torch.tensor(d_out))
    grads = [c.grad.numpy().copy() for c in cores_t]
    out_np = out.detach().numpy().copy()

    # our restatement must agree with the reference before anything is frozen
    o_out = orc.tt_forward(idx, offsets, cores, p, q, R)
    o_grads = orc.tt_dense_backward(idx, offsets, d_out, cores, p, q, R)
    assert np.abs(o_out - out_np).max() <= 1e-5 * max(1.0, np.abs(out_np).max()), name
    for a, b in zip(o_grads, grads):
        assert np.abs(a - b).max() <= 1e-5 * max(1.0, np.abs(b).max()), name
    o_full = orc.tt_full_table(cores, p, q, R)
    assert np.abs(o_full - full.detach().numpy()).max() <= 1e-5, name

    lr, eps = 0.05, 1.0e-10
    sgd = [c - np.float32(lr) * g for c, g in zip(cores, grads)]
    st = [g * g for g in grads]
    ada = [c - np.float32(lr) * g / (np.sqrt(s) + np.float32(eps))
           for c, g, s in zip(cores, grads, st)]
    save = dict(p=np.array(p), q=np.array(q), R=np.array(R), indices=idx, offsets=offsets,
                d_output=d_out, out=out_np, lr=np.float32(lr), eps=np.float32(eps))
    for t in range(T):
        save[f"core{t}"] = cores[t]
        save[f"grad{t}"] = grads[t]
        save[f"sgd{t}"] = sgd[t].astype(np.float32)
        save[f"ada_state{t}"] = st[t].astype(np.float32)
        save[f"ada{t}"] = ada[t].astype(np.float32)
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **save)
    print(f"{name}: B={B} nnz={idx.shape[0]} D={D} |out|max={np.abs(out_np).max():.4f}")


def sampled_case(name, p, q, ranks, seed, n_emb, sel_sizes, n_ids=64):
    """Big configs: rows of a sub-table computed by the reference's
    tt_matrix_to_full on cores restricted to selected p-slices (row (a,b,c) of
    the sub-table is id sel0[a]*L0 + sel1[b]*L1 + sel2[c] of the real table)."""
    T = len(p)
    R = orc.full_ranks(ranks, T)
    D = int(np.prod(q))
    sigma = 1.0 / np.sqrt(n_emb)  # the "normal" init SAGE uses (tt_embeddings_ops.py:651-657)
    cores = seeded_cores(p, q, R, seed, scale=None)
    # keep magnitudes O(1) so that 1e-4 abs is a meaningful bar on the big configs
    cores = [c * np.float32(0.5) for c in cores]
    del sigma
    rng = np.random.default_rng(seed + 7)
    L = orc.strides_L(p)
    sels = []
    for t in range(T):
        s = np.sort(rng.choice(p[t], size=min(sel_sizes[t], p[t]), replace=False))
        sels.append(s)
    # make sure the largest valid id's neighbourhood is covered
    last = n_emb - 1
    last_split = [int(x[0]) for x in orc.split_index(np.array([last]), p)]
    for t in range(T):
        if last_split[t] not in sels[t]:
            sels[t][-1] = last_split[t]
            sels[t] = np.sort(sels[t])
    sub_cores = [torch.tensor(cores[t][sels[t]]) for t in range(T)]
    sub_p = [len(s) for s in sels]
    sub_full = ref_full(sub_p, q, R, sub_cores).numpy()
    grid = np.stack(np.meshgrid(*sels, indexing="ij"), -1).reshape(-1, T)
    all_ids = (grid * np.array(L)[None, :]).sum(1).astype(np.int64)
    valid = np.nonzero(all_ids < n_emb)[0]
    pick = rng.choice(valid, size=n_ids - 1, replace=False)
    pick = np.concatenate([pick, [np.nonzero(all_ids == last)[0][0]]])
    ids = all_ids[pick]
    rows = sub_full[pick]
    o_rows = orc.tt_rows(ids, cores, p, q, R)
    assert np.abs(o_rows - rows).max() <= 1e-4 * max(1.0, np.abs(rows).max()), name
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), p=np.array(p), q=np.array(q),
                        R=np.array(R), seed=np.int64(seed), core_scale=np.float32(0.5),
                        num_embeddings=np.int64(n_emb), indices=ids, rows=rows,
                        cores_sha256=np.array(cores_sha256(cores)))
    print(f"{name}: ids={ids.shape[0]} max_id={ids.max()} D={D} |rows|max={np.abs(rows).max():.3f} "
          f"oracle-vs-ref max diff={np.abs(o_rows - rows).max():.2e}")


def murmur_vectors():
    keys = np.array([0, 1, 2, 3, 17, 255, 256, 65535, 65536, 2449028, 2**24, 2**24 + 1,
                     111059955, 2**31 - 1, 2**31, 2**32 - 1, 2**32, 2**40 + 12345,
                     2**62 + 99, -1, -2], dtype=np.int64)
    sizes = np.array([1, 7, 1000, 169343, 2449029, 111059956, 2**31 - 1], dtype=np.int64)
    table = np.zeros((sizes.shape[0], keys.shape[0]), dtype=np.int64)
    for a, C in enumerate(sizes.tolist()):
        for b, k in enumerate(keys.tolist()):
            table[a, b] = orc.murmur_slot(k, C)
        assert (orc.murmur_slots(keys, C) == table[a]).all()
    # independent anchor: with the length word 8 instead of the reference's 2 the hash
    # word is the standard MurmurHash3_x86_32(seed 0) of the 8 little-endian key bytes.
    from sklearn.utils import murmurhash3_32
    for k in keys.tolist():
        raw = int(np.int64(k).view(np.uint64)).to_bytes(8, "little")
        assert orc.murmur_word(k, len_xor=8) == murmurhash3_32(raw, seed=0, positive=True), k
    np.savez_compressed(os.path.join(HERE, "murmur_kat.npz"), keys=keys, sizes=sizes, slots=table)
    print("murmur_kat:", table.shape)


def suggested_shape_vectors():
    """Known answers of the reference's suggested_tt_shapes (tt_embeddings_ops.py:369-429)."""
    from FBTT.tt_embeddings_ops import suggested_tt_shapes
    ns = [100, 128, 169343, 2449029, 111059956, 1000, 4096, 99991, 360360, 7, 1]
    rows = []
    for n in ns:
        for d in (2, 3, 4):
            for up in (True, False):
                if n < 8 and not up and d > 2:
                    continue
                shape = [int(v) for v in suggested_tt_shapes(n, d, allow_round_up=up)]
                rows.append([n, d, int(up)] + shape + [0] * (4 - d))
    np.savez_compressed(os.path.join(HERE, "suggest_kat.npz"), table=np.array(rows, dtype=np.int64))
    print("suggest_kat:", len(rows), "rows")


def window_case(name, p, q, ranks, seed, n_emb, n_windows, width):
    """SURVEY §8d cfg-B3 / §8f-1: ids of a METIS-reordered frontier = windows of consecutive ids
    (seeds first).  Rows again come from the reference's tt_matrix_to_full on the sub-table of the
    p-slices the windows touch."""
    T = len(p)
    R = orc.full_ranks(ranks, T)
    cores = [c * np.float32(0.5) for c in seeded_cores(p, q, R, seed, scale=None)]
    rng = np.random.default_rng(seed + 7)
    L = orc.strides_L(p)
    starts = rng.integers(0, n_emb - width, size=n_windows)
    starts[0] = (starts[0] // p[-1]) * p[-1] + p[-1] - width // 2     # one window crosses an (i0,i1) boundary
    starts[-1] = n_emb - width                                        # one ends at the last valid id
    ids = (starts[:, None] + np.arange(width)[None, :]).reshape(-1).astype(np.int64)
    digits = orc.split_index(ids, p)
    sels = [np.unique(d) for d in digits]
    sub_full = ref_full([len(s) for s in sels], q, R, [torch.tensor(cores[t][sels[t]]) for t in range(T)]).numpy()
    pos = [np.searchsorted(sels[t], digits[t]) for t in range(T)]
    sub_id = np.zeros_like(ids)
    for t in range(T):
        sub_id = sub_id * len(sels[t]) + pos[t]
    rows = sub_full[sub_id]
    o_rows = orc.tt_rows(ids, cores, p, q, R)
    assert np.abs(o_rows - rows).max() <= 1e-4 * max(1.0, np.abs(rows).max()), name
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), p=np.array(p), q=np.array(q), R=np.array(R),
                        seed=np.int64(seed), core_scale=np.float32(0.5), num_embeddings=np.int64(n_emb),
                        indices=ids, rows=rows, cores_sha256=np.array(cores_sha256(cores)))
    print(f"{name}: ids={ids.shape[0]} prefixes={np.unique(ids // p[-1]).shape[0]} max_id={ids.max()} "
          f"|rows|max={np.abs(rows).max():.3f} oracle-vs-ref max diff={np.abs(o_rows - rows).max():.2e}")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "suggest":
        suggested_shape_vectors()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "windows":
        window_case("rows_products_b3", [125, 140, 140], [4, 5, 5], [16, 16], 22, 2449029, 6, 40)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "ranks":   # the rank sweep of run_script.sh
        rank_sweep_cases()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "scripts":   # the (q, rank) shapes of the reference's run scripts
        script_shape_cases()
        return
    torch.manual_seed(0)
    tiny_case("tt_tiny_T2", [6, 7], [4, 3], [5], seed=11)
    tiny_case("tt_tiny_T3", [3, 4, 5], [2, 3, 2], [4, 3], seed=12)
    tiny_case("tt_tiny_T4", [3, 2, 4, 3], [2, 2, 3, 2], [3, 4, 2], seed=13)
    # 3-core case with the products ranks/q (exercises the specialised fast path shapes)
    tiny_case("tt_small_prodshape", [5, 6, 7], [4, 5, 5], [16, 16], seed=14)
    tiny_case("tt_small_arxivshape", [4, 5, 3], [4, 4, 8], [8, 8], seed=15)
    tiny_case("tt_small_papershape", [3, 4, 3], [8, 4, 4], [32, 32], seed=16)
    sampled_case("rows_arxiv", [56, 60, 51], [4, 4, 8], [8, 8], 21, 169343, [8, 8, 8])
    sampled_case("rows_products", [125, 140, 140], [4, 5, 5], [16, 16], 22, 2449029, [8, 8, 8])
    sampled_case("rows_papers", [500, 560, 400], [8, 4, 4], [32, 32], 23, 111059956, [6, 6, 6])
    window_case("rows_products_b3", [125, 140, 140], [4, 5, 5], [16, 16], 22, 2449029, 6, 40)
    murmur_vectors()
    suggested_shape_vectors()
    script_shape_cases()
    rank_sweep_cases()


def rank_case(name, p, q, ranks, seed, scale):
    """A (q, rank) point of the rank sweep of run_script.sh:250-288.  The cores are NOT stored (the middle core of a rank-256
    table is megabytes): tests regenerate them from the seed (conftest.seeded_cores, same generator) and the file carries a
    digest of their bytes; of the middle core's gradient every 61st element is kept."""
    R = orc.full_ranks(ranks, 3)
    n_emb, D = int(np.prod(p)), int(np.prod(q))
    cores = seeded_cores(p, q, R, seed, scale=scale)
    rng = np.random.default_rng(seed + 1)
    idx, offsets = ragged_bags(rng, 19, n_emb, 3.0)
    B = offsets.shape[0] - 1
    d_out = (rng.random((B, D)).astype(np.float32) * np.float32(0.1))
    cores_t = [torch.tensor(c, requires_grad=True) for c in cores]
    out = ref_bag_forward(ref_full(p, q, R, cores_t), idx, offsets)
    out.backward(torch.tensor(d_out))
    grads = [c.grad.numpy().copy() for c in cores_t]
    out_np = out.detach().numpy().copy()
    o_out = orc.tt_forward(idx, offsets, cores, p, q, R)
    o_grads = orc.tt_dense_backward(idx, offsets, d_out, cores, p, q, R)
    assert np.abs(o_out - out_np).max() <= 1e-5 * max(1.0, np.abs(out_np).max()), name
    for a, b in zip(o_grads, grads):
        assert np.abs(a - b).max() <= 2e-5 * max(1.0, np.abs(b).max()), name
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), p=np.array(p), q=np.array(q), R=np.array(R), seed=np.int64(seed),
                        scale=np.float32(scale), cores_sha256=np.array(cores_sha256(cores)), indices=idx, offsets=offsets,
                        d_output=d_out, out=out_np, grad0=grads[0], grad2=grads[2], grad1_every61=grads[1].reshape(-1)[::61].copy(),
                        grad1_absmax=np.float32(np.abs(grads[1]).max()))
    print(f"{name}: B={B} nnz={idx.shape[0]} |out|max={np.abs(out_np).max():.4f} |g1|max={np.abs(grads[1]).max():.4f}")


def rank_sweep_cases():
    """run_script.sh:250-268 (q = 5,5,4, --tt-rank 8,8 ... 256,256) and :270-288 (q = 4,4,8); 4,5,5 at rank 8."""
    for r in (8, 32, 64, 128, 256):
        rank_case(f"tt_rank_q554r{r}", [3, 2, 4], [5, 5, 4], [r, r], seed=30 + r, scale=0.7 / np.sqrt(r))
    for r in (64, 128, 256):
        rank_case(f"tt_rank_q448r{r}", [2, 3, 3], [4, 4, 8], [r, r], seed=31 + r, scale=0.7 / np.sqrt(r))
    rank_case("tt_rank_q455r8", [4, 3, 5], [4, 5, 5], [8, 8], seed=29, scale=0.25)


def script_shape_cases():
    """q = 4,4,8 / 8,4,4 at rank 16 (run_ogbn-arxiv*.sh and friends) and the rank-32 variants of the fast path."""
    tiny_case("tt_small_q448r16", [5, 4, 6], [4, 4, 8], [16, 16], seed=17)
    tiny_case("tt_small_q844r16", [4, 5, 3], [8, 4, 4], [16, 16], seed=18)
    tiny_case("tt_small_q455r32", [3, 4, 5], [4, 5, 5], [32, 32], seed=19)
    tiny_case("tt_small_q448r32", [4, 3, 4], [4, 4, 8], [32, 32], seed=20)
    tiny_case("tt_small_q545r16", [7, 5, 4], [5, 4, 5], [16, 16], seed=21)   # q0 does not divide the MFMA tile height
    tiny_case("tt_small_q554r16", [4, 7, 5], [5, 5, 4], [16, 16], seed=22)   # q0 q1 = 25 is not a multiple of the MFMA K
    window_case("rows_q448r16_b3", [125, 140, 140], [4, 4, 8], [16, 16], 24, 2449029, 6, 40)


if __name__ == "__main__":
    main()
