"""GPU: the drop-in module (FBTT.tt_embeddings_ops.TTEmbeddingBag), the `tt_embeddings`
extension shim and the LFU cache, following the recipe of the reference's (gutted) unit
tests: forward == EmbeddingBag(sum) over full_weight(); dense grads == autograd through
tt_matrix_to_full; fused SGD/Adagrad == closed form (sage_profiler.py:262-500)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    import FBTT.tt_embeddings_ops as m
    return m


@pytest.fixture(scope="module")
def orc():
    from oracle import tt_oracle
    return tt_oracle


def ragged(rng, B, n_emb, mean_len):
    lens = np.clip(np.round(rng.normal(mean_len, mean_len, B)), 0, None).astype(np.int64)
    idx = rng.integers(0, n_emb, size=int(lens.sum()), dtype=np.int64)
    return torch.tensor(idx).cuda(), torch.tensor(np.concatenate([[0], np.cumsum(lens)])).cuda()


def reference_bag(full, idx, offs):
    return torch.nn.functional.embedding_bag(idx, full, offs, mode="sum", include_last_offset=True)


SHAPES = [
    ([7, 9, 11], [4, 5, 5], [16, 16]),
    ([7, 9, 11, 5], [2, 2, 5, 4], [5, 6, 3]),
    ([40, 50], [4, 8], [12]),
    ([10, 12, 9], [4, 4, 8], [8, 8]),
]


@pytest.mark.parametrize("p,q,r", SHAPES)
def test_forward_equals_embedding_bag_over_full_weight(ops, p, q, r):
    torch.manual_seed(1)
    n, D = int(np.prod(p)), int(np.prod(q))
    emb = ops.TTEmbeddingBag(n, D, r, p, q, sparse=False, use_cache=False, weight_dist="uniform")
    for c in emb.tt_cores:
        c.data.mul_(3.0)
    idx, offs = ragged(np.random.default_rng(0), 97, n, 5.0)
    out = emb(idx, offs)
    want = reference_bag(emb.full_weight(), idx, offs)
    assert out.shape == want.shape
    torch.testing.assert_close(out, want, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("sparse", [True, False])
@pytest.mark.parametrize("p,q,r,n_ids,prefix_in_chain", [
    ([50, 56, 40], [8, 4, 4], [32, 32], 9000, True),      # 3 ids per group: the forward chain kernel forms the prefix products
    ([40, 50, 60], [4, 4, 8], [16, 16], 6000, True),
    ([125, 140, 140], [4, 5, 5], [16, 16], 200000, False), # 11 ids per group: prefix launch, the chain kernel reads P whoever keeps the plan
    ([7, 9, 11], [4, 5, 5], [16, 16], 300, False),         # per-bag kernels: no plan at all
])
def test_forward_under_no_grad_equals_the_training_forward(ops, orc, p, q, r, n_ids, prefix_in_chain, sparse):
    """Inference (`torch.no_grad()`: the drivers' evaluation passes) takes no autograd node and keeps no plan -- a forward that
    forms its prefix products in the chain kernel stores none of them then.  Same rows, bit for bit, as the training forward;
    against the oracle; and a training step after an inference forward still trains (fused SGD against the oracle)."""
    import ttemb_native as nat
    torch.manual_seed(4)
    rng = np.random.default_rng(4)
    n, D, lr = int(np.prod(p)), int(np.prod(q)), 0.05
    emb = ops.TTEmbeddingBag(n, D, r, p, q, sparse=sparse, use_cache=False, weight_dist="normal", learning_rate=lr)
    for c in emb.tt_cores:
        c.data.mul_(2.0)
    fam = nat.kernel_family(nat.make_shape(p, q, r), n_ids, n_ids, True)
    assert bool(fam & nat.FAMILY_PREFIX_IN_CHAIN) == prefix_in_chain
    idx = rng.choice(n, size=n_ids, replace=False).astype(np.int64)
    ids, offs = torch.tensor(idx).cuda(), torch.arange(n_ids + 1).cuda()
    cores = [c.detach()[0].cpu().numpy().copy() for c in emb.tt_cores]
    R = [1] + r + [1]
    with torch.no_grad():
        out_inf = emb(ids, offs)
    assert not out_inf.requires_grad and out_inf.grad_fn is None
    out = emb(ids, offs)
    assert out.requires_grad
    assert torch.equal(out_inf, out.detach())
    want = orc.tt_forward(idx, np.arange(n_ids + 1), cores, p, q, R)
    np.testing.assert_allclose(out_inf.cpu().numpy(), want, rtol=1e-4, atol=1e-4 * max(1.0, float(np.abs(want).max())))
    d = torch.randn(n_ids, D, device="cuda")
    with torch.no_grad():   # an inference forward BETWEEN the training forward and its backward: the kept plan is the training forward's
        emb(ids[: n_ids // 2].contiguous(), offs[: n_ids // 2 + 1].contiguous())
    out.backward(d)
    grads = orc.tt_dense_backward(idx, np.arange(n_ids + 1), d.cpu().numpy(), cores, p, q, R)
    for t in range(3):
        if sparse:
            got, ref = emb.tt_cores[t].detach()[0].cpu().numpy(), cores[t] - np.float32(lr) * grads[t]
        else:
            got, ref = emb.tt_cores[t].grad[0].cpu().numpy(), grads[t]
        np.testing.assert_allclose(got, ref, rtol=2e-4, atol=2e-4 * max(1.0, float(np.abs(ref).max())))


@pytest.mark.parametrize("p,q,r", SHAPES)
def test_backward_dense_equals_autograd_through_full_weight(ops, p, q, r):
    torch.manual_seed(2)
    n, D = int(np.prod(p)), int(np.prod(q))
    emb = ops.TTEmbeddingBag(n, D, r, p, q, sparse=False, use_cache=False, weight_dist="uniform")
    for c in emb.tt_cores:
        c.data.mul_(3.0)
    idx, offs = ragged(np.random.default_rng(1), 97, n, 5.0)
    clones = [c.detach().clone().requires_grad_(True) for c in emb.tt_cores]
    full = ops.tt_matrix_to_full(p, q, r, clones, [1, 0, 2, 3])
    out = emb(idx, offs)
    d_out = torch.rand_like(out) * 0.1
    out.backward(d_out)
    reference_bag(full, idx, offs).backward(d_out)
    for a, b in zip(emb.tt_cores, clones):
        assert a.grad.shape == a.shape
        torch.testing.assert_close(a.grad, b.grad, rtol=1e-3, atol=1e-4 * float(b.grad.abs().max()))


@pytest.mark.parametrize("q,r", [([2, 4, 4, 4], [16, 16, 16]), ([5, 5, 2, 2], [16, 16, 16])])
def test_four_core_module_trains_on_the_grouped_path(ops, q, r):
    """The reference's 4-core run-script shapes through the class on a batch big enough for the grouped kernels:
    forward against the dense table, dense gradients and the in-backward Adagrad step against autograd through
    tt_matrix_to_full."""
    import ttemb_native as nat
    torch.manual_seed(3)
    p = [8, 9, 20, 25]
    n, D = int(np.prod(p)), int(np.prod(q))
    rng = np.random.default_rng(3)
    idx = torch.tensor(rng.integers(0, n, size=20000)).cuda()
    offs = torch.arange(idx.numel() + 1).cuda()
    assert nat.plan_bytes(nat.make_shape(p, q, r), idx.numel()) > 0   # AUTO takes the merged-pair mapping at this size
    emb = ops.TTEmbeddingBag(n, D, r, p, q, sparse=False, use_cache=False, weight_dist="uniform")
    for c in emb.tt_cores:
        c.data.mul_(3.0)
    clones = [c.detach().clone().requires_grad_(True) for c in emb.tt_cores]
    full = ops.tt_matrix_to_full(p, q, r, clones, [1, 0, 2, 3])
    out = emb(idx, offs)
    torch.testing.assert_close(out, full.detach()[idx], rtol=1e-4, atol=1e-4)
    d_out = torch.rand_like(out) * 0.1
    out.backward(d_out)
    full[idx].backward(d_out)
    for a, b in zip(emb.tt_cores, clones):
        torch.testing.assert_close(a.grad, b.grad, rtol=1e-3, atol=2e-4 * float(b.grad.abs().max()))
    # sparse mode: the optimiser step happens inside backward
    lr, eps = 0.05, 1e-8
    emb2 = ops.TTEmbeddingBag(n, D, r, p, q, sparse=True, use_cache=False, weight_dist="uniform", learning_rate=lr,
                              optimizer=ops.OptimType.EXACT_ADAGRAD, eps=eps)
    for c, src in zip(emb2.tt_cores, clones):
        c.data.copy_(src.detach())
    emb2(idx, offs).backward(d_out)
    for c, src in zip(emb2.tt_cores, clones):
        g = src.grad
        want = src.detach() - lr * g / (torch.sqrt(g * g) + eps)
        torch.testing.assert_close(c.detach(), want, rtol=1e-3, atol=2e-4)


@pytest.fixture(params=["e_table", "lds_slabs"])
def wide_backward_form(request):
    """Both backward forms of the wide-rank chain (see tests/test_gpu_parity.py): the module's few hundred ids would take the
    E table by the library's rule; the diagnostic sends them to the kernel that reduces dG2 in LDS."""
    import ttemb_native as nat
    nat.set_wide_slab_min_ids(1 if request.param == "lds_slabs" else 0)
    yield request.param
    nat.set_wide_slab_min_ids(0)


@pytest.mark.parametrize("q,r", [([5, 5, 4], [64, 64]), ([4, 4, 8], [128, 128])])
@pytest.mark.parametrize("mode", ["dense", "SGD", "EXACT_ADAGRAD"])
def test_wide_rank_module_on_the_grouped_chain(ops, q, r, mode, wide_backward_form):
    """Rank 64 / 128 tables through the module: batches of >= 256 ids take the wide-rank grouped chain (GEMM prefix,
    per-group backward, GEMM dG1 / dG0, optimiser step in the finalize kernel); checked against autograd through
    tt_matrix_to_full."""
    torch.manual_seed(11)
    p = [6, 7, 9]
    n, D = int(np.prod(p)), int(np.prod(q))
    lr, eps = 0.05, 1e-10
    kw = {} if mode == "dense" else {"optimizer": getattr(ops.OptimType, mode), "learning_rate": lr, "eps": eps}
    emb = ops.TTEmbeddingBag(n, D, r, p, q, sparse=(mode != "dense"), use_cache=False, weight_dist="uniform", **kw)
    before = [c.detach().clone() for c in emb.tt_cores]
    clones = [c.detach().clone().requires_grad_(True) for c in emb.tt_cores]
    idx, offs = ragged(np.random.default_rng(5), 400, n, 3.0)
    assert idx.numel() >= 256
    out = emb(idx, offs)
    want = reference_bag(ops.tt_matrix_to_full(p, q, r, clones, [1, 0, 2, 3]), idx, offs)
    torch.testing.assert_close(out, want, rtol=1e-4, atol=1e-4 * float(want.detach().abs().max()))
    d_out = torch.rand_like(out) * 0.1
    out.backward(d_out)
    want.backward(d_out)
    for t, (c, b, ref) in enumerate(zip(emb.tt_cores, before, clones)):
        tol = 2e-4 * float(ref.grad.abs().max())
        if mode == "dense":
            torch.testing.assert_close(c.grad, ref.grad, rtol=1e-3, atol=tol)
        elif mode == "SGD":
            assert c.grad is None
            torch.testing.assert_close(c.data, b - lr * ref.grad, rtol=0, atol=lr * tol + 1e-6)
        else:
            st = emb.optimizer_state[t]
            torch.testing.assert_close(st, ref.grad ** 2, rtol=2e-3, atol=1e-4 * float((ref.grad ** 2).max()))
            big = ref.grad.abs() > 1e-2 * ref.grad.abs().max()
            torch.testing.assert_close(c.data[big], (b - lr * ref.grad / (ref.grad.abs() + eps))[big], rtol=0, atol=1e-4)


@pytest.mark.parametrize("optimizer", ["SGD", "EXACT_ADAGRAD"])
def test_sparse_mode_updates_in_backward(ops, optimizer):
    torch.manual_seed(3)
    p, q, r = [7, 9, 11], [4, 5, 5], [16, 16]
    n, D = int(np.prod(p)), int(np.prod(q))
    lr, eps = 0.05, 1e-10
    emb = ops.TTEmbeddingBag(n, D, r, p, q, sparse=True, use_cache=False, weight_dist="uniform",
                             optimizer=getattr(ops.OptimType, optimizer), learning_rate=lr, eps=eps)
    for c in emb.tt_cores:
        c.data.mul_(3.0)
    before = [c.detach().clone() for c in emb.tt_cores]
    clones = [c.detach().clone().requires_grad_(True) for c in emb.tt_cores]
    idx, offs = ragged(np.random.default_rng(2), 64, n, 4.0)
    out = emb(idx, offs)
    d_out = torch.rand_like(out) * 0.1
    out.backward(d_out)
    reference_bag(ops.tt_matrix_to_full(p, q, r, clones, [1, 0, 2, 3]), idx, offs).backward(d_out)
    for t, (c, b, ref) in enumerate(zip(emb.tt_cores, before, clones)):
        assert c.grad is None  # autograd sees no gradient in sparse mode
        if optimizer == "SGD":
            torch.testing.assert_close(c.data, b - lr * ref.grad, rtol=0, atol=1e-5)
        else:
            st = emb.optimizer_state[t]
            torch.testing.assert_close(st, ref.grad ** 2, rtol=2e-3, atol=1e-4 * float((ref.grad ** 2).max()))
            big = ref.grad.abs() > 1e-3 * ref.grad.abs().max()
            want = b - lr * ref.grad / (ref.grad.abs() + eps)
            torch.testing.assert_close(c.data[big], want[big], rtol=0, atol=1e-5)
            assert torch.equal(c.data[ref.grad == 0], b[ref.grad == 0])


@pytest.mark.parametrize("windows", [True, False])
@pytest.mark.parametrize("sparse", [False, True])
def test_table_batched(ops, windows, sparse):
    """``num_tables`` = 3 in one call (reference: TableBatchedTTEmbeddingBag, tt_embeddings_ops.py:446-916).  windows: every
    table is a window of the id list whose bounds the kernels read from ``offsets`` on the device -- the call must not
    synchronise the host (checked with torch's sync-debug mode); not windows: the id list is split on the host, one plain lookup
    per table.  Dense gradients against autograd through the full weight; sparse: the fused SGD step against the same."""
    torch.manual_seed(4)
    p, q, r = [10, 12, 9], [4, 4, 8], [8, 8]
    n, D, Tn, B, lr = int(np.prod(p)), int(np.prod(q)), 3, 20, 0.05
    emb = ops.TableBatchedTTEmbeddingBag(Tn, n, D, r, p, q, sparse=sparse, use_cache=False, weight_dist="normal", learning_rate=lr)
    emb._use_windows = windows
    for c in emb.tt_cores:
        c.data.mul_(50.0)
    start = [c.detach().clone() for c in emb.tt_cores]
    idx, offs = ragged(np.random.default_rng(3), Tn * B, n, 3.0)
    d_out = torch.rand(Tn, B, D, device="cuda")
    emb(idx, offs)   # (first call: workspace allocation synchronises nothing, but keep it out of the checked region)
    torch.cuda.synchronize()
    if windows:
        torch.cuda.set_sync_debug_mode("error")
    try:
        out = emb(idx, offs)
        assert tuple(out.shape) == (Tn, B, D)
        out.backward(d_out)
    finally:
        torch.cuda.set_sync_debug_mode("default")
    bounds = offs[::B].tolist()
    for k in range(Tn):
        clones = [c[k:k + 1].clone().requires_grad_(True) for c in start]
        full = ops.tt_matrix_to_full(p, q, r, clones, [1, 0, 2, 3])
        o = reference_bag(full, idx[bounds[k]:bounds[k + 1]], offs[k * B:(k + 1) * B + 1] - bounds[k])
        torch.testing.assert_close(out[k], o, rtol=1e-4, atol=1e-5)
        o.backward(d_out[k])
        for a, b, c0 in zip(emb.tt_cores, clones, start):
            if sparse:
                torch.testing.assert_close(a.data[k:k + 1], c0[k:k + 1] - lr * b.grad, rtol=1e-3, atol=1e-4 * float(b.grad.abs().max()))
            else:
                torch.testing.assert_close(a.grad[k:k + 1], b.grad, rtol=1e-3, atol=1e-4 * float(b.grad.abs().max()))


def test_extension_shim_signatures(ops, orc):
    import tt_embeddings as ext
    torch.manual_seed(5)
    p, q, r = [7, 9, 11], [4, 5, 5], [1, 16, 16, 1]
    n, D = int(np.prod(p)), 100
    cores = [torch.randn(1, p[t], r[t] * q[t] * r[t + 1], device="cuda") * 0.3 for t in range(3)]
    L = torch.tensor([99, 11, 1], device="cuda")
    idx, offs = ragged(np.random.default_rng(4), 50, n, 3.0)
    B = offs.numel() - 1
    hashtbl = torch.empty(0, dtype=torch.int64, device="cuda")
    state = torch.empty(0, dtype=torch.int32, device="cuda")
    i2, rowidx, tableidx, ntt, loc = ext.preprocess_indices_sync(idx, offs, 1, True, hashtbl, state)
    assert ntt == idx.numel() and loc is None and (tableidx == 0).all()
    out = ext.tt_forward(1000, 1, B, D, p, q, r, L, ntt, i2, rowidx, tableidx, cores)
    np_cores = [c[0].cpu().numpy() for c in cores]
    want = orc.tt_forward(idx.cpu().numpy(), offs.cpu().numpy(), np_cores, p, q, r)
    np.testing.assert_allclose(out[0].cpu().numpy(), want, rtol=1e-5, atol=1e-4)
    d_out = torch.rand(1, B, D, device="cuda") * 0.1
    grads = ext.tt_dense_backward(1000, D, p, q, r, L, ntt, i2, rowidx, tableidx, d_out, cores)
    want_g = orc.tt_dense_backward(idx.cpu().numpy(), offs.cpu().numpy(), d_out[0].cpu().numpy(), np_cores, p, q, r)
    for a, b in zip(grads, want_g):
        assert tuple(a.shape) == (1,) + b.shape
        np.testing.assert_allclose(a[0].cpu().numpy(), b, rtol=1e-3, atol=1e-4 * np.abs(b).max())
    before = [c.clone() for c in cores]
    ext.tt_sgd_backward(1000, D, 0.1, p, q, r, L, ntt, i2, rowidx, tableidx, d_out, cores)
    for c, b, g in zip(cores, before, grads):
        torch.testing.assert_close(c, b - 0.1 * g, rtol=0, atol=1e-5)
    with pytest.raises(RuntimeError):
        ext.tt_forward(0, 1, B, D, p, q, r, L, ntt, i2, rowidx, tableidx, cores)  # batch_count <= 0
    with pytest.raises(RuntimeError):
        ext.tt_forward(10, 1, B, 98, p, [2, 7, 7], r, L, ntt, i2, rowidx, tableidx, cores)  # D % 4 != 0


# ---------------------------------------------------------------------------------------
# LFU cache
# ---------------------------------------------------------------------------------------
def test_hash_table_matches_oracle_bit_exactly(ops, orc):
    import ttemb_native as nat
    rng = np.random.default_rng(6)
    H = 4099
    ids = rng.choice(10 ** 7, size=600, replace=False).astype(np.int64)
    ids = np.concatenate([ids, np.array([2 ** 40 + 5, 2 ** 31, 2 ** 32 + 1], dtype=np.int64)])
    # keep only ids whose probe windows are disjoint, so the final table is order-independent
    used, keep = set(), []
    for k in ids.tolist():
        s = orc.murmur_slot(k, H)
        win = {(s + j) % H for j in range(-3, 4)}
        if not (win & used):
            used |= win
            keep.append(k)
    keep = np.array(keep, dtype=np.int64)
    stream = np.repeat(keep, rng.integers(1, 6, size=keep.shape[0]))
    rng.shuffle(stream)
    keys = torch.full((H,), -1, dtype=torch.int64, device="cuda")
    freq = torch.zeros(H, dtype=torch.int64, device="cuda")
    nat.cache_update(torch.tensor(stream).cuda(), keys, freq)
    o_keys, o_freq = np.full(H, -1, dtype=np.int64), np.zeros(H, dtype=np.int64)
    assert orc.update_cache_state(stream, o_keys, o_freq) == 0
    assert np.array_equal(keys.cpu().numpy(), o_keys) and np.array_equal(freq.cpu().numpy(), o_freq)
    # populate + lookup + partition, all integer: bit-exact against the oracle
    p, q, R = [125, 140, 140], [4, 5, 5], [1, 16, 16, 1]
    cores_np = [(rng.standard_normal((p[t], R[t] * q[t] * R[t + 1])) * 0.3).astype(np.float32) for t in range(3)]
    cores = [torch.tensor(c).cuda() for c in cores_np]
    C = 64
    state = torch.full((H,), -1, dtype=torch.int32, device="cuda")
    weight = torch.zeros(C, 100, device="cuda")
    ws = nat.Workspace()
    nat.cache_populate(nat.make_shape(p, q, R), cores, keys, freq, state, weight, ws)
    o_state = np.full(H, -1, dtype=np.int32)
    kept = orc.cache_populate(o_keys, o_freq, o_state, C)
    assert np.array_equal(keys.cpu().numpy(), o_keys)
    assert np.array_equal(freq.cpu().numpy(), o_freq)
    assert np.array_equal(state.cpu().numpy(), o_state)
    valid = kept < 125 * 140 * 140  # ids past prod(p) are outside the table (the kernels clamp them)
    np.testing.assert_allclose(weight.cpu().numpy()[valid], orc.tt_rows(kept[valid], cores_np, p, q, R),
                               rtol=1e-5, atol=1e-4)
    probe = np.concatenate([kept[:20], keep[-20:], np.array([1, 2, 3], dtype=np.int64)])
    rng.shuffle(probe)
    lens = rng.integers(0, 4, size=30)
    lens[-1] += probe.shape[0] - lens.sum() if lens.sum() <= probe.shape[0] else 0
    if lens.sum() != probe.shape[0]:
        lens = np.ones(probe.shape[0], dtype=np.int64)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    B = offs.shape[0] - 1
    out_idx = torch.empty(probe.shape[0], dtype=torch.int64, device="cuda")
    out_row = torch.empty_like(out_idx)
    out_loc = torch.empty(probe.shape[0], dtype=torch.int32, device="cuda")
    count = torch.zeros(1, dtype=torch.int32, device="cuda")
    nat.preprocess(torch.tensor(probe).cuda(), torch.tensor(offs).cuda(), B, False, keys, state, out_idx,
                   out_row, out_loc, count, ws)
    pi, pr, ntt, ploc = orc.preprocess_indices(probe, offs, False, o_keys, o_state)
    assert int(count.item()) == ntt
    assert np.array_equal(out_idx.cpu().numpy(), pi) and np.array_equal(out_row.cpu().numpy(), pr)
    assert np.array_equal(out_loc.cpu().numpy()[ntt:], ploc[ntt:])


@pytest.mark.parametrize("n_ids,bags", [(70000, "single"), (50003, "ragged"), (257, "ragged"), (256, "single")])
def test_partition_of_many_blocks_is_bit_exact(ops, orc, n_ids, bags):
    """ttemb_preprocess on batches that span hundreds of 256-id blocks (per-block counts, prefix over the blocks
    before, ballot ranks inside a block): partitioned ids / bags / cache rows in exactly the order the oracle's
    restatement of cub::DevicePartition::Flagged gives, and the duplicate flag."""
    import ttemb_native as nat
    rng = np.random.default_rng(n_ids)
    H, C = 40000, 6000
    keys = np.full(H, -1, dtype=np.int64)
    freq = np.zeros(H, dtype=np.int64)
    universe = rng.choice(10 ** 7, size=12000, replace=False).astype(np.int64)
    orc.update_cache_state(np.repeat(universe, rng.integers(1, 4, size=universe.shape[0])), keys, freq)
    state = np.full(H, -1, dtype=np.int32)
    orc.cache_populate(keys, freq, state, C)
    cached_ids = keys[state >= 0]
    ids = np.where(rng.random(n_ids) < 0.4, rng.choice(cached_ids, size=n_ids), rng.integers(0, 10 ** 7, size=n_ids))
    unique_cached = len(set(ids[np.isin(ids, cached_ids)].tolist())) == int(np.isin(ids, cached_ids).sum())
    if bags == "single":
        offs = np.arange(n_ids + 1, dtype=np.int64)
    else:
        lens = rng.integers(0, 5, size=n_ids)
        lens = lens[np.cumsum(lens) <= n_ids]
        lens[-1] += n_ids - lens.sum()
        offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    B = offs.shape[0] - 1
    t_ids, t_offs = torch.tensor(ids).cuda(), torch.tensor(offs).cuda()
    out_idx, out_row = torch.empty_like(t_ids), torch.empty_like(t_ids)
    out_loc = torch.empty(n_ids, dtype=torch.int32, device="cuda")
    count = torch.full((2,), -5, dtype=torch.int32, device="cuda")
    stamp = torch.zeros(C, dtype=torch.int32, device="cuda")
    nat.preprocess(t_ids, t_offs, B, False, torch.tensor(keys).cuda(), torch.tensor(state).cuda(), out_idx, out_row,
                   out_loc, count, nat.Workspace(), stamp, 9)
    pi, pr, ntt, ploc = orc.preprocess_indices(ids, offs, False, keys, state)
    assert count.cpu().tolist() == [ntt, 0 if unique_cached else 1]
    assert 0 < ntt < n_ids
    assert np.array_equal(out_idx.cpu().numpy(), pi)
    assert np.array_equal(out_row.cpu().numpy(), pr)
    assert np.array_equal(out_loc.cpu().numpy()[ntt:], ploc[ntt:])


@pytest.mark.parametrize("sparse", [False, True])
def test_cache_lifecycle_keeps_forward_and_trains(ops, orc, sparse):
    torch.manual_seed(7)
    rng = np.random.default_rng(7)
    p, q, r = [20, 25, 30], [4, 5, 5], [16, 16]
    n, D = int(np.prod(p)), 100
    lr = 0.1
    emb = ops.TTEmbeddingBag(n, D, r, p, q, sparse=sparse, use_cache=True, cache_size=200, hashtbl_size=n,
                             weight_dist="normal", learning_rate=lr)
    for c in emb.tt_cores:
        c.data.mul_(40.0)
    hot = rng.choice(n, size=150, replace=False)
    for _ in range(5):  # warm-up epoch: everything goes the TT way, frequencies accumulate
        batch = np.concatenate([rng.choice(hot, size=300), rng.integers(0, n, size=100)])
        idx = torch.tensor(batch).cuda()
        out = emb(idx, torch.arange(idx.numel() + 1).cuda())
        assert emb.warmup
    emb.cache_populate()
    assert not emb.warmup
    cached = emb.hashtbl[emb.cache_state >= 0]
    assert 0 < cached.numel() <= 200
    seen_hot = set(hot.tolist())
    # the hottest ids are the cached ones (up to hash-insert failures)
    assert len(seen_hot & set(cached.tolist())) >= 140
    # forward is unchanged by switching the cache on (cache rows == TT rows at populate time)
    batch = np.concatenate([rng.choice(hot, size=200), rng.integers(0, n, size=200)])
    lens = rng.integers(0, 4, size=250)
    lens = lens[np.cumsum(lens) <= batch.shape[0]]
    batch = batch[: int(lens.sum())]
    idx = torch.tensor(batch).cuda()
    offs = torch.tensor(np.concatenate([[0], np.cumsum(lens)])).cuda()
    cores_np = [c.detach()[0].cpu().numpy() for c in emb.tt_cores]
    want = orc.tt_forward(batch, offs.cpu().numpy(), cores_np, p, q, [1] + r + [1])
    cache_before = emb.cache_weight.detach().clone()
    cores_before = [c.detach().clone() for c in emb.tt_cores]
    out = emb(idx, offs)
    np.testing.assert_allclose(out.detach().cpu().numpy(), want, rtol=1e-5, atol=1e-4)
    # backward: cached ids train cache_weight, the rest train the cores
    d_out = (torch.rand_like(out) - 0.5) * 0.1
    out.backward(d_out)
    is_tt, loc = orc.cache_lookup(batch, emb.hashtbl.cpu().numpy(), emb.cache_state.cpu().numpy())
    assert (~is_tt).sum() > 50 and is_tt.sum() > 50
    rowidx = orc.rowidx_from_offsets(offs.cpu().numpy(), batch.shape[0])
    g_cache = orc.cache_backward_dense(d_out.cpu().numpy(), loc[~is_tt], rowidx[~is_tt], 200, D)
    # core grads come only from the TT part: rebuild that sub-batch for the oracle
    tt_ids, tt_rows_ = batch[is_tt], rowidx[is_tt]
    sub_offs = np.concatenate([[0], np.cumsum(np.bincount(tt_rows_, minlength=offs.numel() - 1))])
    g_cores = orc.tt_dense_backward(tt_ids, sub_offs, d_out.cpu().numpy(), cores_np, p, q, [1] + r + [1])
    if sparse:
        np.testing.assert_allclose(emb.cache_weight.detach().cpu().numpy(),
                                   cache_before.cpu().numpy() - lr * g_cache, rtol=0, atol=1e-5)
        for c, b, g in zip(emb.tt_cores, cores_before, g_cores):
            np.testing.assert_allclose(c.detach()[0].cpu().numpy(), b[0].cpu().numpy() - lr * g, rtol=0, atol=2e-5)
    else:
        np.testing.assert_allclose(emb.cache_weight.grad.cpu().numpy(), g_cache, rtol=1e-4, atol=1e-6)
        for c, g in zip(emb.tt_cores, g_cores):
            np.testing.assert_allclose(c.grad[0].cpu().numpy(), g, rtol=1e-3, atol=1e-4 * np.abs(g).max())


@pytest.mark.parametrize("sparse", [False, True])
def test_cache_backward_with_unique_cache_rows(ops, orc, sparse):
    """A batch in which no cached id repeats: ttemb_preprocess reports it (second word of the device count) and the
    cache backward updates every row with a plain read-modify-write instead of float atomics; a batch with a repeat
    flips the flag and takes the atomic path.  Both against the oracle."""
    import ttemb_native as nat
    torch.manual_seed(11)
    rng = np.random.default_rng(11)
    p, q, r = [20, 25, 30], [4, 5, 5], [16, 16]
    n, D, lr = int(np.prod(p)), 100, 0.1
    emb = ops.TTEmbeddingBag(n, D, r, p, q, sparse=sparse, use_cache=True, cache_size=300, hashtbl_size=n,
                             weight_dist="normal", learning_rate=lr)
    for c in emb.tt_cores:
        c.data.mul_(40.0)
    hot = rng.choice(n, size=250, replace=False)
    for _ in range(4):
        idx = torch.tensor(np.concatenate([rng.choice(hot, size=400), rng.integers(0, n, size=100)])).cuda()
        emb(idx, torch.arange(idx.numel() + 1).cuda())
    emb.cache_populate()
    keys, state = emb.hashtbl.cpu().numpy(), emb.cache_state.cpu().numpy()
    for repeat in (False, True):
        batch = np.concatenate([rng.choice(hot, size=180, replace=False),
                                rng.choice(np.setdiff1d(np.arange(n), hot), size=220, replace=False)])
        if repeat:   # one cached id a second time, in the place of an uncached one
            is_tt0, _ = orc.cache_lookup(batch, keys, state)
            batch[np.flatnonzero(is_tt0)[0]] = batch[np.flatnonzero(~is_tt0)[0]]
        rng.shuffle(batch)
        idx = torch.tensor(batch).cuda()
        offs = torch.arange(batch.shape[0] + 1).cuda()
        # the flag itself, through the C ABI
        nnz_tt = torch.full((2,), -7, dtype=torch.int32).cuda()
        stamp = torch.zeros(300, dtype=torch.int32).cuda()
        nat.preprocess(idx, offs, batch.shape[0], False, emb.hashtbl, emb.cache_state, torch.empty_like(idx),
                       torch.empty_like(idx), torch.empty(batch.shape[0], dtype=torch.int32).cuda(), nnz_tt,
                       nat.Workspace(), stamp, 3)
        is_tt, loc = orc.cache_lookup(batch, keys, state)
        assert nnz_tt.cpu().tolist() == [int(is_tt.sum()), 1 if repeat else 0]
        assert (~is_tt).sum() > 100
        # one training step through the module
        cache_before = emb.cache_weight.detach().clone().cpu().numpy()
        if emb.cache_weight.grad is not None:
            emb.cache_weight.grad = None
        out = emb(idx, offs)
        d_out = (torch.rand_like(out) - 0.5) * 0.1
        out.backward(d_out)
        g_cache = orc.cache_backward_dense(d_out.cpu().numpy(), loc[~is_tt], np.arange(batch.shape[0])[~is_tt], 300, D)
        if sparse:
            np.testing.assert_allclose(emb.cache_weight.detach().cpu().numpy(), cache_before - lr * g_cache,
                                       rtol=0, atol=1e-5)
        else:
            np.testing.assert_allclose(emb.cache_weight.grad.cpu().numpy(), g_cache, rtol=1e-4, atol=1e-6)


def test_lfu_update_after_populate_keeps_cached_ids(ops, orc):
    """After cache_populate's evictions the update must find a displaced cached key instead of
    inserting it again in front of itself (see oracle.update_cache_state, find_first=True)."""
    import ttemb_native as nat
    rng = np.random.default_rng(21)
    H, C = 4096, 600
    warm = rng.integers(0, 50000, size=6000)
    warm = np.concatenate([warm, rng.choice(warm[:1500], size=6000)])        # some ids are hot
    tbl, freq, state = np.full(H, -1, np.int64), np.zeros(H, np.int64), np.full(H, -1, np.int32)
    orc.update_cache_state(warm, tbl, freq)
    orc.cache_populate(tbl, freq, state, C)
    d_tbl, d_freq = torch.tensor(tbl).cuda(), torch.tensor(freq).cuda()
    cached = tbl[state >= 0]
    # only cached ids + ids that cannot be inserted anywhere near them would be too easy: mix in new ids
    later = np.concatenate([rng.choice(cached, size=5000), rng.integers(50000, 60000, size=300)])
    nat.cache_update(torch.tensor(later).cuda(), d_tbl, d_freq)
    o_tbl, o_freq = tbl.copy(), freq.copy()
    orc.update_cache_state(later, o_tbl, o_freq, find_first=True)
    g_tbl, g_freq = d_tbl.cpu().numpy(), d_freq.cpu().numpy()
    live = state >= 0
    assert np.array_equal(g_tbl[live], tbl[live])                   # cached entries stay where they were
    assert np.array_equal(g_freq[live], o_freq[live])               # and all their hits were counted there
    tracked = g_tbl[g_tbl >= 0]
    assert np.unique(tracked).size == tracked.size                  # no key in two slots
    is_tt, _ = orc.cache_lookup(cached, g_tbl, state)
    assert not is_tt.any()                                          # every cached id still resolves to its row
    # the reference's one-sweep insert loses some of them on the same input
    r_tbl, r_freq = tbl.copy(), freq.copy()
    orc.update_cache_state(later, r_tbl, r_freq)
    assert orc.cache_lookup(cached, r_tbl, state)[0].any()


def test_cache_rowwise_adagrad(ops, orc):
    import ttemb_native as nat
    rng = np.random.default_rng(8)
    C, D, B = 50, 100, 40
    w = rng.standard_normal((C, D)).astype(np.float32)
    st = rng.random(C).astype(np.float32)
    g = ((rng.random((B, D)) - 0.5) * 0.2).astype(np.float32)
    loc = rng.choice(C, size=B, replace=False).astype(np.int32)  # distinct rows: the update is race-free
    rowidx = np.arange(B, dtype=np.int64)
    wt, stt = torch.tensor(w).cuda(), torch.tensor(st).cuda()
    nat.cache_backward_rowwise_adagrad(torch.tensor(loc).cuda(), torch.tensor(rowidx).cuda(), 0, None, B,
                                       torch.tensor(g).cuda(), 0.05, 1e-10, stt, wt)
    w2, st2 = orc.cache_backward_rowwise_adagrad(g, loc, rowidx, 0.05, 1e-10, st, w)
    np.testing.assert_allclose(stt.cpu().numpy(), st2, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(wt.cpu().numpy(), w2, rtol=1e-5, atol=1e-6)


def test_eff_tt_embedding_api(ops, orc):
    """Second API of the reference (Efficient_TT/efficient_tt.py:214-307): 2-D cores, one row per id,
    fused SGD in backward with the constructor's learning rate."""
    from Efficient_TT.efficient_tt import Eff_TTEmbedding
    torch.manual_seed(9)
    p, q, r = [125, 140, 140], [4, 5, 5], [16, 16]
    emb = Eff_TTEmbedding(2449029, 100, r, p, q, learning_rate=0.05, device=0)
    assert [tuple(c.shape) for c in emb.tt_cores] == [(125, 64), (140, 1280), (140, 80)]
    for c in emb.tt_cores:
        c.data.mul_(8.0)
    cores = [c.detach().cpu().numpy().copy() for c in emb.tt_cores]
    rng = np.random.default_rng(9)
    ids = rng.integers(0, 2449029, size=50000, dtype=np.int64)  # duplicates included, fast path engaged
    ids[:4] = [0, 2449028, 7, 7]
    with torch.no_grad():   # inference: same rows, no autograd node (and no plan kept)
        out_inf = emb(torch.tensor(ids).cuda(), None, None, None)
    out = emb(torch.tensor(ids).cuda(), None, None, None)
    assert out_inf.grad_fn is None and out.grad_fn is not None and torch.equal(out_inf, out.detach())
    R = [1] + r + [1]
    pick = rng.choice(ids.shape[0], size=400, replace=False)
    np.testing.assert_allclose(out.detach().cpu().numpy()[pick], orc.tt_rows(ids[pick], cores, p, q, R), rtol=1e-5,
                               atol=1e-4)
    d_out = ((torch.rand_like(out) - 0.5) * 0.01)
    out.backward(d_out)
    assert all(c.grad is None for c in emb.tt_cores)
    sub = np.arange(0, ids.shape[0])  # full oracle backward on 50k ids is fine (vectorised)
    g = orc.tt_dense_backward(ids[sub], np.arange(sub.shape[0] + 1), d_out.cpu().numpy(), cores, p, q, R)
    for c, c0, gr in zip(emb.tt_cores, cores, g):
        np.testing.assert_allclose(c.detach().cpu().numpy(), c0 - np.float32(0.05) * gr, rtol=0,
                                   atol=1e-5 + 1e-4 * float(np.abs(0.05 * gr).max()))


def test_cache_live_on_the_fast_path(ops, orc):
    """Cache switched on with a batch large enough for the grouped MFMA path: the TT/cached split count
    stays on the device (nnz_dev) and both kernel families must honour it, forward and backward."""
    torch.manual_seed(11)
    rng = np.random.default_rng(11)
    p, q, r = [125, 140, 140], [4, 5, 5], [16, 16]
    n, D, lr = 2449029, 100, 0.05
    emb = ops.TTEmbeddingBag(n, D, r, p, q, sparse=True, use_cache=True, cache_size=20000, hashtbl_size=n,
                             weight_dist="normal", learning_rate=lr)
    for c in emb.tt_cores:
        c.data.mul_(300.0)
    hot = rng.choice(n, size=15000, replace=False)
    for _ in range(3):
        batch = np.concatenate([rng.choice(hot, size=30000), rng.integers(0, n, size=30000)])
        emb(torch.tensor(batch).cuda(), torch.arange(batch.shape[0] + 1).cuda())
    emb.cache_populate()
    batch = np.concatenate([rng.choice(hot, size=30000), rng.integers(0, n, size=70000)])
    rng.shuffle(batch)
    idx = torch.tensor(batch).cuda()
    offs = torch.arange(batch.shape[0] + 1).cuda()
    R = [1] + r + [1]
    cores_np = [c.detach()[0].cpu().numpy().copy() for c in emb.tt_cores]
    cache_before = emb.cache_weight.detach().cpu().numpy().copy()
    out = emb(idx, offs)
    pick = rng.choice(batch.shape[0], size=2000, replace=False)
    np.testing.assert_allclose(out.detach().cpu().numpy()[pick], orc.tt_rows(batch[pick], cores_np, p, q, R),
                               rtol=1e-5, atol=1e-4)
    d_out = (torch.rand_like(out) - 0.5) * 0.02
    out.backward(d_out)
    is_tt, loc = orc.cache_lookup(batch, emb.hashtbl.cpu().numpy(), emb.cache_state.cpu().numpy())
    assert 10000 < (~is_tt).sum() < 40000
    rows = np.arange(batch.shape[0])
    g_cache = orc.cache_backward_dense(d_out.cpu().numpy(), loc[~is_tt], rows[~is_tt], 20000, D)
    np.testing.assert_allclose(emb.cache_weight.detach().cpu().numpy(), cache_before - lr * g_cache, rtol=0, atol=1e-5)
    tt_ids = batch[is_tt]
    g = orc.tt_dense_backward(tt_ids, np.arange(tt_ids.shape[0] + 1), d_out.cpu().numpy()[is_tt], cores_np, p, q, R)
    for c, c0, gr in zip(emb.tt_cores, cores_np, g):
        np.testing.assert_allclose(c.detach()[0].cpu().numpy(), c0 - np.float32(lr) * gr, rtol=0,
                                   atol=1e-5 + 2e-4 * float(np.abs(lr * gr).max()))


def _dp_gpu_worker(rank, world, port, out_dir, overlap=False, poison_rank=-1):
    import os
    import sys
    from conftest import PKG, ROOT
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)  # both ranks share the one GPU of the box
    from FBTT.tt_embeddings_ops import TTEmbeddingBag
    from ttemb_dist import TTDataParallel
    torch.cuda.set_device(0)
    torch.manual_seed(50 + rank)
    p, q, r = [125, 140, 140], [4, 5, 5], [16, 16]
    emb = TTEmbeddingBag(2449029, 100, r, p, q, sparse=False, use_cache=False, weight_dist="normal", learning_rate=0.2)
    for c in emb.tt_cores:
        c.data.mul_(300.0)
    dp = TTDataParallel(emb)
    dp.broadcast_parameters(0)
    start = [c.detach().cpu().clone() for c in emb.tt_cores]
    g = torch.Generator().manual_seed(7 + rank)
    ids = torch.randperm(2449029, generator=g)[:60000]
    d_out = (torch.rand(60000, 100, generator=g) - 0.5) * 0.02
    if rank == poison_rank:   # every bounded wait of this rank's grouping pass expires: NaN rows, NaN gradients
        import ttemb_native as nat
        nat.set_spin_limit(-1)
    out = emb(ids.cuda(), torch.arange(60001).cuda())
    if poison_rank >= 0:
        import ttemb_native as nat
        torch.cuda.synchronize()
        nat.set_spin_limit(0)
        if rank == poison_rank:
            assert bool(torch.isnan(out).all())
            with pytest.raises(RuntimeError, match="gave up waiting"):   # consumed here: the backward and step() below must reach the collective
                nat.status()
        out.backward(d_out.cuda())   # on the forward's (poisoned) plan: NaN gradients, the header's poison word set
        dp.step(overlap=overlap)
        dp.flush()
        torch.cuda.synchronize()
        # the fault count travelled with the gradients: BOTH ranks skipped the update
        assert float(dp.bucket.fault.item()) == 1.0
        torch.save({"start": start, "end": [c.detach().cpu().clone() for c in emb.tt_cores]}, os.path.join(out_dir, f"rank{rank}.pt"))
        dist.barrier()
        dist.destroy_process_group()
        return
    out.backward(d_out.cuda())
    dp.step(overlap=overlap)
    if overlap:  # the update is still pending; the next forward groups its ids, finishes the update, then looks up
        assert emb._before_weights is not None
    ids2 = ids[:50000].cuda()
    with torch.no_grad():
        out2 = emb(ids2, torch.arange(50001).cuda())
    assert emb._before_weights is None
    torch.cuda.synchronize()
    torch.save({"start": start, "ids": ids, "d_out": d_out, "end": [c.detach().cpu().clone() for c in emb.tt_cores],
                "out2": out2.cpu()}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [False, True])
def test_data_parallel_step_two_ranks_on_gpu(orc, tmp_path, overlap):
    """TTDataParallel end to end with the real kernels: two ranks (gloo carries the GPU tensors; RCCL needs
    one GPU per rank), dense backward -> ONE all-reduce of the flattened core gradients -> fused SGD."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_dp_gpu_worker, args=(2, port, str(tmp_path), overlap), nprocs=2, join=True)
    r = [torch.load(tmp_path / f"rank{k}.pt") for k in range(2)]
    p, q, R = [125, 140, 140], [4, 5, 5], [1, 16, 16, 1]
    cores = [c[0].numpy() for c in r[0]["start"]]
    grads = [orc.tt_dense_backward(r[k]["ids"].numpy(), np.arange(60001), r[k]["d_out"].numpy(), cores, p, q, R)
             for k in range(2)]
    for t in range(3):
        assert torch.equal(r[0]["start"][t], r[1]["start"][t])
        want = cores[t] - np.float32(0.2) * (grads[0][t] + grads[1][t]) / 2
        for k in range(2):
            np.testing.assert_allclose(r[k]["end"][t][0].numpy(), want, rtol=0,
                                       atol=1e-5 + 2e-4 * float(np.abs(0.1 * (grads[0][t] + grads[1][t])).max()))
        assert torch.equal(r[0]["end"][t], r[1]["end"][t])  # replicas stay bit-identical
    # the forward that followed the step used the UPDATED cores (with overlap the update ran between its two halves)
    for k in range(2):
        new_cores = [c[0].numpy() for c in r[k]["end"]]
        ids2 = r[k]["ids"][:50000].numpy()
        want_rows = orc.tt_rows(ids2, new_cores, p, q, R)
        np.testing.assert_allclose(r[k]["out2"].numpy(), want_rows, rtol=1e-4, atol=1e-4 * float(np.abs(want_rows).max()))


def test_a_poisoned_rank_makes_every_rank_skip_the_data_parallel_update(tmp_path):
    """One rank's grouping pass gives up (ttemb_set_spin_limit(-1)): its gradient is NaN.  The count of such ranks rides in the
    all-reduced bucket and the guarded SGD step (ttemb_sgd_step_guarded) skips the update on EVERY rank: replicas stay
    identical and untouched instead of all-NaN."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_dp_gpu_worker, args=(2, port, str(tmp_path), False, 1), nprocs=2, join=True)
    r = [torch.load(tmp_path / f"rank{k}.pt") for k in range(2)]
    for t in range(3):
        for k in range(2):
            assert torch.equal(r[k]["end"][t], r[0]["start"][t]) and bool(torch.isfinite(r[k]["end"][t]).all())


def test_single_process_data_parallel_step_skips_a_poisoned_gradient(ops):
    """World size 1 (no collective): the guarded step reads the poison word of the module's workspace header itself --
    a step whose grouping pass gave up leaves the weights alone, the next healthy step updates them."""
    import ttemb_native as nat
    from ttemb_dist import TTDataParallel
    torch.manual_seed(4)
    emb = ops.TTEmbeddingBag(2449029, 100, [16, 16], [125, 140, 140], [4, 5, 5], sparse=False, use_cache=False,
                             weight_dist="normal", learning_rate=0.1)
    for c in emb.tt_cores:
        c.data.mul_(300.0)
    dp = TTDataParallel(emb)
    ids = torch.randperm(2449029)[:30000].cuda()
    offs = torch.arange(30001).cuda()
    d_out = ((torch.rand(30000, 100) - 0.5) * 0.02).cuda()
    before = [c.detach().clone() for c in emb.tt_cores]
    nat.set_spin_limit(-1)
    try:
        out = emb(ids, offs)
        torch.cuda.synchronize()
        with pytest.raises(RuntimeError, match="gave up waiting"):
            nat.status()
    finally:
        nat.set_spin_limit(0)
    assert bool(torch.isnan(out).all())
    out.backward(d_out)
    dp.step()
    torch.cuda.synchronize()
    assert all(torch.equal(c.detach(), b) for c, b in zip(emb.tt_cores, before)), "a poisoned gradient must not be applied"
    emb(ids, offs).backward(d_out)
    dp.step()
    torch.cuda.synchronize()
    assert not any(torch.equal(c.detach(), b) for c, b in zip(emb.tt_cores, before))
    assert all(bool(torch.isfinite(c).all()) for c in emb.tt_cores)


def test_bucket_accumulates_over_two_backwards_before_the_step(ops, orc):
    """Gradient accumulation with the data-parallel wrapper attached: two forward/backward passes before dp.step()
    must sum into the flat bucket (the kernels overwrite their destination; the second pass goes through scratch),
    as autograd accumulates into .grad."""
    from ttemb_dist import TTDataParallel
    torch.manual_seed(3)
    p, q, r = [125, 140, 140], [4, 5, 5], [16, 16]
    lr = 0.2
    emb = ops.TTEmbeddingBag(2449029, 100, r, p, q, sparse=False, use_cache=False, weight_dist="normal", learning_rate=lr)
    for c in emb.tt_cores:
        c.data.mul_(300.0)
    dp = TTDataParallel(emb)   # no process group: world size 1
    cores = [c.detach()[0].cpu().numpy().copy() for c in emb.tt_cores]
    rng = np.random.default_rng(5)
    want = [np.zeros_like(c) for c in cores]
    for n in (9000, 2000):   # grouped kernels, then the wave-per-id kernels
        ids = rng.choice(2449029, size=n, replace=False).astype(np.int64)
        d_out = ((rng.random((n, 100)) - 0.5) * 0.05).astype(np.float32)
        emb(torch.tensor(ids).cuda(), torch.arange(n + 1).cuda()).backward(torch.tensor(d_out).cuda())
        for w, g in zip(want, orc.tt_dense_backward(ids, np.arange(n + 1), d_out, cores, p, q, [1] + r + [1])):
            w += g
    dp.step()
    torch.cuda.synchronize()
    for c, c0, g in zip(emb.tt_cores, cores, want):
        np.testing.assert_allclose(c.detach()[0].cpu().numpy(), c0 - np.float32(lr) * g, rtol=0,
                                   atol=1e-6 + 2e-4 * float(np.abs(lr * g).max()))


@pytest.mark.parametrize("optimizer", ["SGD", "EXACT_ADAGRAD"])
def test_captured_lookup_trains_like_the_eager_module(ops, orc, optimizer):
    """emb.capture(nnz, B): forward and backward as HIP-graph replays give the rows and the in-backward update of the eager
    module on the same inputs, step after step (two steps with different ids through the same graphs)."""
    torch.manual_seed(11)
    p, q, r = [125, 140, 140], [4, 5, 5], [16, 16]
    mk = lambda: ops.TTEmbeddingBag(2449029, 100, r, p, q, optimizer=getattr(ops.OptimType, optimizer), sparse=True, use_cache=False,
                                    weight_dist="normal", learning_rate=0.1)
    a, b = mk(), mk()
    for ca, cb in zip(a.tt_cores, b.tt_cores):
        ca.data.mul_(300.0)
        cb.data.copy_(ca.data)
    n = 2048
    cap = b.capture(n, n)
    rng = np.random.default_rng(2)
    for _ in range(2):
        ids = torch.tensor(rng.choice(2449029, size=n, replace=False).astype(np.int64)).cuda()
        d = torch.tensor(((rng.random((n, 100)) - 0.5) * 0.05).astype(np.float32)).cuda()
        out_a = a(ids, torch.arange(n + 1).cuda())
        out_b = cap(ids)
        # (the second step looks up weights that already differ by the first step's summation order: Adagrad's bound below)
        torch.testing.assert_close(out_b, out_a, rtol=1e-5, atol=1e-6 if optimizer == "SGD" else 1e-5)
        out_a.backward(d)
        out_b.backward(d)
        torch.cuda.synchronize()
        for ca, cb in zip(a.tt_cores, b.tt_cores):   # float atomics: the two runs differ by summation order only (Adagrad's
            # g / sqrt(g^2) turns a last-bit difference of a tiny gradient into a visible one: absolute bound 2e-5 there)
            torch.testing.assert_close(cb.data, ca.data, rtol=1e-4, atol=1e-6 if optimizer == "SGD" else 2e-5)


def test_captured_lookup_on_the_wide_rank_chain(ops):
    """emb.capture() on a rank-64 table: the wide-rank chain (GEMM prefix, row lists, per-group backward, column-sliced
    reduce, GEMMs, finalize) replays from HIP graphs and trains like the eager module."""
    torch.manual_seed(5)
    p, q, r = [20, 15, 30], [5, 5, 4], [64, 64]
    n_emb = int(np.prod(p))
    mk = lambda: ops.TTEmbeddingBag(n_emb, 100, r, p, q, optimizer=ops.OptimType.SGD, sparse=True, use_cache=False,
                                    weight_dist="uniform", learning_rate=0.05)
    a, b = mk(), mk()
    for ca, cb in zip(a.tt_cores, b.tt_cores):
        cb.data.copy_(ca.data)
    n = 1024
    cap = b.capture(n, n)
    rng = np.random.default_rng(6)
    for _ in range(2):
        ids = torch.tensor(rng.choice(n_emb, size=n, replace=False).astype(np.int64)).cuda()
        d = torch.tensor(((rng.random((n, 100)) - 0.5) * 0.05).astype(np.float32)).cuda()
        out_a = a(ids, torch.arange(n + 1).cuda())
        out_b = cap(ids)
        torch.testing.assert_close(out_b, out_a, rtol=1e-5, atol=1e-5 * float(out_a.detach().abs().max()))
        out_a.backward(d)
        out_b.backward(d)
        torch.cuda.synchronize()
        for ca, cb in zip(a.tt_cores, b.tt_cores):   # summation order (row atomics of the shared slab, split-K slabs) differs only
            torch.testing.assert_close(cb.data, ca.data, rtol=1e-4, atol=1e-5 * float(ca.data.abs().max()))


@pytest.mark.parametrize("n", [20000, 200000])
def test_captured_lookup_on_the_grouped_chain(ops, orc, n):
    """emb.capture() on the narrow grouped chain (products shape): 20 000 ids take the forward that forms its prefix products
    in the chain kernel (1.1 ids per group), 200 000 the prefix launch (11 per group); grouping pass with its call counter on
    the device, chain kernels and the fused backward replay from HIP graphs -- three steps with different ids through the same
    graphs -- give the oracle's rows and train like the eager module; nothing is pending on the fault word afterwards."""
    import ttemb_native as nat
    torch.manual_seed(13)
    p, q, r = [125, 140, 140], [4, 5, 5], [16, 16]
    n_emb = 2449029
    mk = lambda: ops.TTEmbeddingBag(n_emb, 100, r, p, q, optimizer=ops.OptimType.SGD, sparse=True, use_cache=False,
                                    weight_dist="normal", learning_rate=0.05)
    a, b = mk(), mk()
    for ca, cb in zip(a.tt_cores, b.tt_cores):
        ca.data.mul_(300.0)
        cb.data.copy_(ca.data)
    fam = nat.kernel_family(nat.make_shape(p, q, [1] + r + [1]), n, n, True)
    assert fam & ~nat.FAMILY_GROUP_PRODUCTS_IN_CHAIN == nat.FAMILY_GROUPED | (nat.FAMILY_PREFIX_IN_CHAIN if n < 8 * p[0] * p[1] else 0)
    cap = b.capture(n, n)
    rng = np.random.default_rng(14)
    offs = torch.arange(n + 1).cuda()
    for step in range(3):
        ids_np = rng.choice(n_emb, size=n, replace=False).astype(np.int64)
        ids = torch.tensor(ids_np).cuda()
        d = torch.tensor(((rng.random((n, 100)) - 0.5) * 0.05).astype(np.float32)).cuda()
        if step == 0:
            cores_np = [c.detach()[0].cpu().numpy() for c in b.tt_cores]
            want = orc.tt_rows(ids_np[:2000], cores_np, p, q, [1] + r + [1])
        out_a = a(ids, offs)
        out_b = cap(ids)
        if step == 0:
            np.testing.assert_allclose(out_b.detach()[:2000].cpu().numpy(), want, rtol=1e-5, atol=1e-4)
        torch.testing.assert_close(out_b, out_a, rtol=1e-5, atol=2e-6)
        out_a.backward(d)
        out_b.backward(d)
        torch.cuda.synchronize()
        for ca, cb in zip(a.tt_cores, b.tt_cores):   # (summation order inside a group comes from LDS atomics: rounding only)
            torch.testing.assert_close(cb.data, ca.data, rtol=1e-4, atol=1e-6)
    nat.status()


@pytest.mark.parametrize("one_sweep", [False, True])
def test_lfu_update_on_a_colliding_stream(orc, one_sweep):
    """cache_update on a table that is far too small for its stream (H = 96, hundreds of distinct ids): whatever the
    thread order, (1) a key sits in at most one slot, (2) a tracked key's count is exactly its number of occurrences (an
    occurrence either finds the key or inserts it), (3) an untracked id has its three probe slots taken by other keys --
    for both insert forms, which only differ after evictions."""
    import ttemb_native as nat
    rng = np.random.default_rng(3)
    H = 96
    ids = rng.integers(0, 400, size=5000).astype(np.int64)
    keys = torch.full((H,), -1, dtype=torch.int64, device="cuda")
    freq = torch.zeros(H, dtype=torch.int64, device="cuda")
    nat.cache_update(torch.tensor(ids).cuda(), keys, freq, one_sweep)
    torch.cuda.synchronize()
    k, f = keys.cpu().numpy(), freq.cpu().numpy()
    held = k[k >= 0]
    assert len(set(held.tolist())) == held.shape[0]
    counts = {int(v): int(c) for v, c in zip(*np.unique(ids, return_counts=True))}
    for slot in np.nonzero(k >= 0)[0]:
        assert f[slot] == counts[int(k[slot])]
    assert (f[k < 0] == 0).all()
    slots = orc.murmur_slots(np.array(sorted(counts), dtype=np.int64), H)
    for key, s0 in zip(sorted(counts), slots.tolist()):
        if key not in set(held.tolist()):
            assert all(k[(s0 + d) % H] >= 0 and k[(s0 + d) % H] != key for d in range(3))
    assert int(f.sum()) == sum(counts[int(v)] for v in held)


def test_one_sweep_insert_reproduces_the_reference_after_an_eviction(orc):
    """After cache_populate has evicted the key in FRONT of a tracked key's slot, the reference's one-sweep insert
    (hashtbl_cuda_utils.cuh:102-133) puts the tracked key into the hole a second time; the default update finds it where it
    is.  One key, so the outcome does not depend on thread order."""
    import ttemb_native as nat
    H = 64
    # two keys with the same first probe slot: a takes it, b the next one
    cand = np.arange(0, 20000, dtype=np.int64)
    slots = orc.murmur_slots(cand, H)
    s0 = int(slots[0])
    same = cand[slots == s0]
    a, b = int(same[0]), int(same[1])
    for one_sweep, want_slots in ((False, 1), (True, 2)):
        keys = torch.full((H,), -1, dtype=torch.int64, device="cuda")
        freq = torch.zeros(H, dtype=torch.int64, device="cuda")
        nat.cache_update(torch.tensor([a]).cuda(), keys, freq)
        nat.cache_update(torch.tensor([b]).cuda(), keys, freq)
        torch.cuda.synchronize()
        assert int(keys[s0]) == a and int(keys[(s0 + 1) % H]) == b
        keys[s0], freq[s0] = -1, 0          # what populate does to a key that did not make the cache
        nat.cache_update(torch.tensor([b, b, b]).cuda(), keys, freq, one_sweep)
        torch.cuda.synchronize()
        assert int((keys == b).sum()) == want_slots
        assert int(freq[keys == b].sum()) == 4


def check_lfu_update_by_key(orc, k0, f0, k1, f1, batch, state1=None):
    """The LFU update of one batch (every probe slot searched before an insert; tt_embeddings_cuda.cu:1083-1095 with
    hashtbl_cuda_utils.cuh:102-154) judged BY KEY, which is deterministic whatever order the threads ran in -- a slot-by-slot
    comparison of two tables is not: which of two colliding new keys gets a contested slot is a compare-and-swap race.
      1. a key tracked before stays in its slot and its counter grows by its occurrences in the batch;
      2. a new key sits in one of its three probe slots, once, with exactly its occurrences (and is not cached);
      3. an id of the batch that is NOT tracked afterwards finds all three of its probe slots taken by other keys (slots only
         fill during an update, so an empty one would have taken it) -- these are the failed inserts, counted exactly.
    Returns the number of failed inserts."""
    H = k0.shape[0]
    uniq, cnt = np.unique(batch, return_counts=True)

    def occurrences(keys):
        at = np.searchsorted(uniq, keys)
        at = np.minimum(at, uniq.shape[0] - 1)
        return np.where(uniq[at] == keys, cnt[at], 0)

    t0 = k0 != -1
    np.testing.assert_array_equal(k1[t0], k0[t0])
    np.testing.assert_array_equal(f1[t0], f0[t0] + occurrences(k0[t0]))
    new = ~t0 & (k1 != -1)
    nk = k1[new]
    assert not np.isin(nk, k0[t0]).any(), "a tracked key was inserted a second time"
    assert np.unique(k1[k1 != -1]).shape[0] == int((k1 != -1).sum()), "a key sits in two slots"
    np.testing.assert_array_equal(f1[new], occurrences(nk))
    assert (f1[new] > 0).all()
    assert (((np.flatnonzero(new) - orc.murmur_slots(nk, H)) % H) < 3).all(), "a new key outside its probe window"
    if state1 is not None:
        assert (state1[new] == -1).all()
    np.testing.assert_array_equal(f1[k1 == -1], 0)
    missing = uniq[~np.isin(uniq, k1[k1 != -1])]
    s = orc.murmur_slots(missing, H)
    for j in range(3):
        assert (k1[(s + j) % H] != -1).all(), "an id was dropped although one of its probe slots is empty"
    return int(missing.shape[0])


@pytest.mark.parametrize("n_ids", [300, 40000])
def test_fused_probe_pass_equals_update_then_preprocess(orc, n_ids):
    """ttemb_preprocess_update == ttemb_cache_update followed by ttemb_preprocess(warmup = 0), bit for bit: the table, the
    counters, the partitioned ids / rows / locations, the TT count and the duplicate flag -- on a populated table whose
    batch holds cached, tracked-but-evicted, new and repeated ids; and both equal the oracle."""
    import ttemb_native as nat
    rng = np.random.default_rng(21)
    H, C, n_emb = 4096, 600, 100000
    keys = torch.full((H,), -1, dtype=torch.int64).cuda()
    freq = torch.zeros(H, dtype=torch.int64).cuda()
    state = torch.full((H,), -1, dtype=torch.int32).cuda()
    hot = rng.choice(n_emb, size=1500, replace=False)
    for _ in range(6):
        nat.cache_update(torch.tensor(rng.choice(hot, size=2000)).cuda(), keys, freq)
    # populate by hand (ranks of the C most frequent slots), evicting the rest like mark_popular does
    f = freq.cpu().numpy()
    order = np.argsort(-f, kind="stable")
    st = np.full(H, -1, np.int32)
    k_np = keys.cpu().numpy().copy()
    live = order[:C][f[order[:C]] > 0]
    st[live] = np.arange(live.shape[0], dtype=np.int32)
    gone = np.setdiff1d(np.flatnonzero(k_np != -1), live)
    k_np[gone] = -1
    f[gone] = 0
    keys, freq, state = torch.tensor(k_np).cuda(), torch.tensor(f).cuda(), torch.tensor(st).cuda()
    batch = np.concatenate([rng.choice(hot, size=n_ids // 2), rng.integers(0, n_emb, size=n_ids - n_ids // 2)])
    rng.shuffle(batch)
    cuts = np.sort(rng.integers(0, n_ids + 1, size=n_ids // 2))   # ragged bags, some of them empty
    offs = np.concatenate([[0], cuts, [n_ids]]).astype(np.int64)
    B = offs.shape[0] - 1
    idx, t_offs = torch.tensor(batch).cuda(), torch.tensor(offs).cuda()

    def run(fused):
        k, fq = keys.clone(), freq.clone()
        outs = [torch.empty_like(idx), torch.empty_like(idx), torch.empty(n_ids, dtype=torch.int32).cuda(),
                torch.full((2,), -7, dtype=torch.int32).cuda()]
        stamp = torch.empty(C, dtype=torch.int32).cuda().fill_(12345)   # any content
        if not fused:
            nat.cache_update(idx, k, fq)
        nat.preprocess(idx, t_offs, B, False, k, state, outs[0], outs[1], outs[2], outs[3], nat.Workspace(), stamp, 0,
                       fq if fused else None)
        torch.cuda.synchronize()
        return [k.cpu().numpy(), fq.cpu().numpy()] + [o.cpu().numpy() for o in outs]

    two, one = run(False), run(True)
    ntt = int(two[5][0])
    assert 0 < ntt < n_ids
    # the outputs exactly; the tables BY KEY (which of two colliding new keys gets a contested slot is a race in both forms):
    # each form satisfies the update's exact invariants, and a key both tables track has the same counter in both
    for a, b in zip(two[2:], one[2:]):
        np.testing.assert_array_equal(a, b)
    f_np = f   # (counters before the batch, after the hand-made populate)
    failed = [check_lfu_update_by_key(orc, k_np, f_np, t[0], t[1], batch, st) for t in (two, one)]
    tracked = lambda k, fq: dict(zip(k[k != -1].tolist(), fq[k != -1].tolist()))
    da, db = tracked(two[0], two[1]), tracked(one[0], one[1])
    common = da.keys() & db.keys()
    assert all(da[k] == db[k] for k in common)
    # a key only one table tracks lost every slot of its window in the other: no more of them than that table's failed inserts
    assert len(da.keys() - db.keys()) <= failed[1] and len(db.keys() - da.keys()) <= failed[0]
    if n_ids <= 300:
        assert da == db and failed[0] == failed[1]   # (no contested slot at this size: the same ids are dropped by both)
    # against the oracle's lookup on the table as it was before the batch (the update never moves a tracked key)
    is_tt, loc = orc.cache_lookup(batch, k_np, st)
    assert ntt == int(is_tt.sum())
    np.testing.assert_array_equal(one[4][ntt:][::-1], loc[~is_tt])
    assert int(one[5][1]) == int(np.unique(loc[~is_tt]).shape[0] != (~is_tt).sum())


def test_cache_live_step_with_and_without_the_fused_probe_pass(ops, orc):
    """The live-cache training step through the class with update_cache_state + preprocess_indices_sync as ONE probe pass
    (default) and as the reference's two calls (`lfu_one_sweep_insert`-free module forced onto the two-call route): same
    rows, same cache rows, same cores, same LFU counters.  Both modules start from one state (a copy of the first one's
    state dict: two tables filled independently differ by insertion races)."""
    torch.manual_seed(5)
    rng = np.random.default_rng(5)
    p, q, r = [125, 140, 140], [4, 5, 5], [16, 16]
    n = 2449029
    mk = lambda: ops.TTEmbeddingBag(n, 100, r, p, q, sparse=True, use_cache=True, cache_size=20000, hashtbl_size=n,
                                    weight_dist="normal", learning_rate=0.05)
    a = mk()
    for ca in a.tt_cores:
        ca.data.mul_(300.0)
    hot = rng.choice(n, size=15000, replace=False)
    for _ in range(3):
        w = np.concatenate([rng.choice(hot, size=30000), rng.integers(0, n, size=30000)])
        a(torch.tensor(w).cuda(), torch.arange(w.shape[0] + 1).cuda())
    a.cache_populate()
    b = mk()
    b.load_state_dict(a.state_dict())
    assert not b.warmup
    b._fused_probe = lambda: False   # the two-call route
    failed_total = [0, 0]
    for step in range(2):
        batch = np.concatenate([rng.choice(hot, size=12000, replace=False), rng.integers(0, n, size=60000)])
        if step == 0:
            offs = np.arange(batch.shape[0] + 1)
        else:   # ragged bags: multi-id bags take the atomic path in both the TT and the cache kernels
            cuts = np.sort(rng.choice(np.arange(1, batch.shape[0]), size=50000, replace=False))
            offs = np.concatenate([[0], cuts, [batch.shape[0]]])
        idx, t_offs = torch.tensor(batch).cuda(), torch.tensor(offs).cuda()
        before = [(m.hashtbl.cpu().numpy().copy(), m.cache_freq.cpu().numpy().copy()) for m in (a, b)]
        oa, ob = a(idx, t_offs), b(idx, t_offs)
        torch.testing.assert_close(oa, ob, rtol=1e-5, atol=1e-6)
        d = (torch.rand_like(oa) - 0.5) * 0.02
        oa.backward(d)
        ob.backward(d)
        torch.cuda.synchronize()
        torch.testing.assert_close(a.cache_weight.data, b.cache_weight.data, rtol=1e-5, atol=1e-6)
        for ca, cb in zip(a.tt_cores, b.tt_cores):
            torch.testing.assert_close(ca.data, cb.data, rtol=1e-4, atol=1e-6)
        # The LFU tables, BY KEY.  Ids met for the first time race for contested slots in either form (~60 000 new keys per
        # step over 2.4 M slots with three-slot probe windows), so WHICH slot a new key gets is a compare-and-swap race and
        # a slot-by-slot comparison of two tables is the wrong invariant; what every order must give is exact: tracked keys
        # stay put and count up by their occurrences, new keys sit once in their probe window with their occurrences, and a
        # dropped id found its whole window taken (check_lfu_update_by_key).  Then: same counter for every key both track.
        after = [(m.hashtbl.cpu().numpy(), m.cache_freq.cpu().numpy(), m.cache_state.cpu().numpy()) for m in (a, b)]
        failed = [check_lfu_update_by_key(orc, before[i][0], before[i][1], after[i][0], after[i][1], batch, after[i][2]) for i in (0, 1)]
        da, db = (dict(zip(k[k != -1].tolist(), f[k != -1].tolist())) for k, f, _ in after)
        assert all(da[k] == db[k] for k in da.keys() & db.keys())
        only_a, only_b = len(da.keys() - db.keys()), len(db.keys() - da.keys())
        lost_a, lost_b = failed_total[0] + failed[0], failed_total[1] + failed[1]   # (a key one table lost in an earlier step may be new to it now)
        assert only_a <= lost_b and only_b <= lost_a, f"step {step}: {only_a} / {only_b} keys in one table only, failed inserts {failed}"
        failed_total = [lost_a, lost_b]
        np.testing.assert_array_equal(after[0][2], after[1][2])   # (a step caches and evicts nothing: new keys carry -1 like empty slots)


@pytest.mark.parametrize("D", [4, 12, 16, 36, 100, 128, 256])
@pytest.mark.parametrize("unique", [True, False])
def test_cache_gather_and_update_kernels_over_row_widths(orc, D, unique):
    """ttemb_cache_forward / _backward_sgd / _backward_dense at the C ABI against the oracle, for row widths that take every
    instantiation of the streamed kernels (1-8 sixteen-byte pieces per lane; D = 256: the general forward, the update in
    two passes): a cached range that starts inside the list and is no multiple of a 16-id step, bags of one id (plain
    stores) and of several (accumulating), cache rows that repeat (atomic update) or not (read-modify-write)."""
    import ttemb_native as nat
    rng = np.random.default_rng(100 + D + (1 if unique else 0))
    C, B = 5000, 3001
    lens = rng.choice([1, 1, 1, 2, 3], size=B)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    nnz = int(offs[-1])
    rowidx = np.repeat(np.arange(B), lens).astype(np.int64)
    start = nnz // 3 + 5                                  # the cached ids are the tail of the partitioned list
    start += 1 if (nnz - start) % 16 == 0 else 0          # (never a whole number of 16-id steps)
    n_c = nnz - start
    loc = np.full(nnz, -1, dtype=np.int32)
    loc[start:] = rng.choice(C, size=n_c, replace=not unique).astype(np.int32)
    # the partition keeps the TT ids in front: their rows are whatever, the cached tail's rows are a shuffled subset of the bags
    rows_tail = rng.permutation(rowidx)[:n_c] if not unique else rng.permutation(rowidx)[:n_c]
    rowidx_p = rowidx.copy()
    rowidx_p[start:] = np.sort(rows_tail)[::-1]           # (CUB order: the rejected items fill the tail backwards)
    w = rng.standard_normal((C, D)).astype(np.float32)
    out0 = rng.standard_normal((B, D)).astype(np.float32)
    dev = lambda a: torch.tensor(a).cuda()
    t_loc, t_row, t_off, t_w = dev(loc), dev(rowidx_p), dev(offs), dev(w)
    # forward: rows of one-id bags are overwritten (the caller's offsets vouch nothing else writes them), the others accumulate
    single = lens[rowidx_p[start:]] == 1
    want = out0.copy()
    seen_single = np.zeros(B, dtype=bool)
    seen_single[rowidx_p[start:][single]] = True
    want[seen_single] = 0.0
    orc.cache_forward(want, loc[start:], rowidx_p[start:], w)
    t_out = dev(out0)
    nat.cache_forward(t_loc, t_row, start, None, nnz, t_w, t_out, offsets=t_off)
    np.testing.assert_allclose(t_out.cpu().numpy(), want, rtol=1e-5, atol=1e-5)
    # the same with the start of the cached range on the device (the module's form)
    t_out2 = dev(out0)
    nat.cache_forward(t_loc, t_row, 0, torch.tensor([start], dtype=torch.int32).cuda(), nnz, t_w, t_out2, offsets=t_off)
    np.testing.assert_allclose(t_out2.cpu().numpy(), want, rtol=1e-5, atol=1e-5)   # (atomic sums: not bit-equal to the first call)
    # backward: dense gradient and the SGD step, with the "no cache row repeats" word as the preprocess pass would leave it
    d_out = rng.standard_normal((B, D)).astype(np.float32)
    dup = torch.tensor([n_c, 0 if unique else 1], dtype=torch.int32).cuda()
    g = torch.empty(C, D, dtype=torch.float32, device="cuda")
    nat.cache_backward_dense(t_loc, t_row, start, None, nnz, dev(d_out), g, dup_dev=dup[1:])
    np.testing.assert_allclose(g.cpu().numpy(), orc.cache_backward_dense(d_out, loc[start:], rowidx_p[start:], C, D), rtol=1e-5, atol=1e-5)
    t_w2 = dev(w)
    nat.cache_backward_sgd(t_loc, t_row, start, None, nnz, dev(d_out), 0.05, t_w2, dup_dev=dup[1:])
    np.testing.assert_allclose(t_w2.cpu().numpy(), orc.cache_backward_sgd(d_out, loc[start:], rowidx_p[start:], 0.05, w), rtol=1e-5, atol=1e-5)


def test_capture_guards(ops):
    """A captured lookup refuses to run once what it baked in no longer holds: a cache gone live, a changed eps, re-allocated
    cores; during warm-up it keeps the LFU statistics going."""
    p, q, r = [125, 140, 140], [4, 5, 5], [16, 16]
    emb = ops.TTEmbeddingBag(2449029, 100, r, p, q, sparse=True, use_cache=True, cache_size=1000, hashtbl_size=50000,
                             weight_dist="normal", learning_rate=0.1)
    n = 512
    cap = emb.capture(n, n)
    ids = torch.arange(n, dtype=torch.int64).cuda() * 7
    cap(ids)
    torch.cuda.synchronize()
    assert int(emb.cache_freq.sum().item()) == n   # warm-up statistics counted through the captured call
    emb.eps = 1e-3
    with pytest.raises(RuntimeError):
        cap(ids)
    emb.eps = 1.0e-10
    cap(ids)
    emb.tt_cores[1] = torch.nn.Parameter(emb.tt_cores[1].data.clone())
    assert emb._cores()[1] is emb.tt_cores[1]   # the cached tuple follows a swap of ANY core, not only the first
    with pytest.raises(RuntimeError):
        cap(ids)
    cap2 = emb.capture(n, n)
    cap2(ids)
    emb.cache_populate()
    with pytest.raises(RuntimeError):
        cap2(ids)
    with pytest.raises(AssertionError):
        emb.capture(n, n)


def test_the_reference_papers100M_invocation_with_its_five_percent_cache(ops, orc):
    """run_script.sh:408-431 (final-papers): p = 400,500,600, q = 4,4,8, ranks 16,16, --sparse --use-cached --cache-size 5
    (gnn_model.py:98-100: 5 % of the 111 M nodes cached, a hash table of num_nodes slots).  p2 = 600 takes the unfused
    E-table backward of the grouped chain, together with the live cache.  30 000 ids against the oracle (rows, split count,
    cores and cache rows after the fused step), then properties at 819 200 ids: split count and CUB partition order on the
    device, rows equal to the cache-less lookup, the fused update equal to lr x the dense gradient of the TT share."""
    import tt_embeddings as b2
    import ttemb_native as nat
    torch.manual_seed(3)
    rng = np.random.default_rng(3)
    p, q, r = [400, 500, 600], [4, 4, 8], [16, 16]
    n, D, lr = 111059956, 128, 0.05
    R = [1] + r + [1]
    emb = ops.TTEmbeddingBag(n, D, r, p, q, sparse=True, use_cache=True, cache_size=int(0.05 * n), hashtbl_size=n,
                             weight_dist="normal", learning_rate=lr, batch_count=14000)
    assert emb.cache_weight.shape == (int(0.05 * n), D) and emb.hashtbl.numel() == n
    for c in emb.tt_cores:
        c.data.mul_(300.0)
    hot = rng.choice(n, size=400000, replace=False)
    for _ in range(4):   # the counting epoch
        w = np.concatenate([rng.choice(hot, size=300000), rng.integers(0, n, size=100000)])
        emb(torch.tensor(w).cuda(), torch.arange(w.shape[0] + 1).cuda())
    emb.cache_populate()
    assert not emb.warmup
    keys_h, state_h = emb.hashtbl.cpu().numpy(), emb.cache_state.cpu().numpy()
    cores_np = [c.detach()[0].cpu().numpy() for c in emb.tt_cores]

    # ---- 30 000 ids, ragged bags, against the oracle ----
    batch = np.concatenate([rng.choice(hot, size=15000), rng.integers(0, n, size=15000)])
    rng.shuffle(batch)
    cuts = np.sort(rng.choice(np.arange(1, batch.shape[0]), size=19999, replace=False))
    offs = np.concatenate([[0], cuts, [batch.shape[0]]]).astype(np.int64)
    B = offs.shape[0] - 1
    idx, t_offs = torch.tensor(batch).cuda(), torch.tensor(offs).cuda()
    is_tt, loc = orc.cache_lookup(batch, keys_h, state_h)
    assert 5000 < int(is_tt.sum()) < 25000
    got = b2.preprocess_indices_sync(idx, t_offs, 1, False, emb.hashtbl, emb.cache_state)
    assert int(got[3]) == int(is_tt.sum())   # the split count, as the B2 function returns it
    np.testing.assert_array_equal(got[0].cpu().numpy(), orc.partition_by_flag(batch, is_tt))
    np.testing.assert_array_equal(got[4].cpu().numpy(), orc.partition_by_flag(loc, is_tt))
    want = orc.tt_forward(batch, offs, cores_np, p, q, R)
    cores_before = [c.detach().clone() for c in emb.tt_cores]
    touched = np.unique(loc[~is_tt])
    rows_before = emb.cache_weight.detach()[torch.tensor(touched).cuda().long()].cpu().numpy()
    out = emb(idx, t_offs)
    np.testing.assert_allclose(out.detach().cpu().numpy(), want, rtol=1e-5, atol=1e-4)
    d_out = (torch.rand_like(out) - 0.5) * 0.1
    out.backward(d_out)
    torch.cuda.synchronize()
    d_np = d_out.cpu().numpy()
    rowidx = orc.rowidx_from_offsets(offs, batch.shape[0])
    g_rows = np.zeros((touched.shape[0], D), dtype=np.float64)
    np.add.at(g_rows, np.searchsorted(touched, loc[~is_tt]), d_np[rowidx[~is_tt]].astype(np.float64))
    rows_after = emb.cache_weight.detach()[torch.tensor(touched).cuda().long()].cpu().numpy()
    np.testing.assert_allclose(rows_after, rows_before - lr * g_rows, rtol=0, atol=1e-5)
    tt_ids, tt_rows_ = batch[is_tt], rowidx[is_tt]
    sub_offs = np.concatenate([[0], np.cumsum(np.bincount(tt_rows_, minlength=B))])
    g_cores = orc.tt_dense_backward(tt_ids, sub_offs, d_np, cores_np, p, q, R)
    for c, b, g in zip(emb.tt_cores, cores_before, g_cores):
        np.testing.assert_allclose(c.detach()[0].cpu().numpy(), b[0].cpu().numpy() - lr * g, rtol=0, atol=1e-5 + 2e-5 * np.abs(lr * g).max())

    # ---- 819 200 ids: properties ----
    N = 819200
    big = np.concatenate([hot, rng.choice(n, size=N - hot.shape[0], replace=False)])
    big = np.unique(big)   # bags of one id: rows are comparable one by one
    rng.shuffle(big)
    N = big.shape[0]
    idx, t_offs = torch.tensor(big).cuda(), torch.arange(N + 1).cuda()
    cached_keys = emb.hashtbl[emb.cache_state >= 0]
    hit = torch.isin(idx, cached_keys)
    got = b2.preprocess_indices_sync(idx, t_offs, 1, False, emb.hashtbl, emb.cache_state)
    ntt = int(got[3])
    assert ntt == int((~hit).sum()) and 0 < ntt < N
    assert torch.equal(got[0][:ntt], idx[~hit]) and torch.equal(got[0][ntt:].flip(0), idx[hit])   # cub::DevicePartition::Flagged order
    assert torch.equal(got[1][:ntt], torch.arange(N, device="cuda")[~hit])
    shape = nat.make_shape(p, q, R)
    # the TT share rides on the grouped chain (unfused E table: p2 = 600); ~2 ids per group: the forward forms P in its chain kernel
    # (p2 = 600: the dG2 reduction is not fused into the chunk kernel, and at ~4 ids per group the backward forms its group products there)
    assert nat.kernel_family(shape, ntt, N) == nat.FAMILY_GROUPED | nat.FAMILY_PREFIX_IN_CHAIN | nat.FAMILY_GROUP_PRODUCTS_IN_CHAIN
    plain = ops.TTEmbeddingBag(n, D, r, p, q, sparse=False, use_cache=False, weight_dist="normal")
    for a, b in zip(plain.tt_cores, emb.tt_cores):
        a.data.copy_(b.data)
    cores_before = [c.detach().clone() for c in emb.tt_cores]
    cache_before = emb.cache_weight.detach().clone()
    out = emb(idx, t_offs)
    ref_rows = plain(idx, t_offs)
    # a cached row is the TT row of populate time unless the 30 000-id step above trained it or the cores under it
    fresh = ~hit
    torch.testing.assert_close(out.detach()[fresh], ref_rows.detach()[fresh], rtol=1e-5, atol=1e-5)
    d_out = (torch.rand_like(out) - 0.5) * 0.1
    out.backward(d_out)
    # the TT share through the cache-less module: its dense gradient is what the fused step applied
    tt_idx = idx[~hit]
    sub = plain(tt_idx, torch.arange(tt_idx.numel() + 1).cuda())
    sub.backward(d_out[~hit])
    torch.cuda.synchronize()
    for c, b, pc in zip(emb.tt_cores, cores_before, plain.tt_cores):
        step = lr * pc.grad
        torch.testing.assert_close(c.detach(), b - step, rtol=0, atol=1e-6 + 1e-4 * float(step.abs().max()))
    # every cached id's row moved by lr x its gradient row (ids are unique: no cache row repeats)
    locs = got[4][ntt:].flip(0).long()
    torch.testing.assert_close(emb.cache_weight.detach()[locs], cache_before[locs] - lr * d_out[hit], rtol=0, atol=1e-6)
    nat.status()
