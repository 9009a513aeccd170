"""CPU: host-side logic of the drop-in layer and the C-ABI library surface
(no kernel is launched here; GPU parity lives in test_gpu_*.py)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import PKG, ROOT, load_golden

import ttemb_native as nat
from FBTT.tt_embeddings_ops import (BufferList, OptimType, TableBatchedTTEmbeddingBag, TTEmbeddingBag,
                                    suggested_tt_shapes, tt_matrix_to_full)
from oracle import tt_oracle as orc


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "ttemb.h")).read()
    declared = set(re.findall(r"\b(ttemb_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"ttemb_workspace_bytes"} - set(nat.EXPORTED_SYMBOLS)
    assert declared == set(nat.EXPORTED_SYMBOLS)
    lib = ctypes.CDLL(nat.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ttemb_abi_version() == 4


def test_no_shipped_kernel_spills():
    """Every ttemb:: kernel of the shipped library fits its registers: private_segment_fixed_size == 0 in the code objects'
    metadata (tools/kres.py reads the notes of the gfx950 code objects bundled into the .so).  A spilling instance is a
    routing or launch-bounds mistake here -- round 4 shipped one (fast3_forward_pfuse_kernel<4,5,5,32,32>: 20 B/lane)."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kres.py"), nat.LIB_PATH, "--fail-on-scratch"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.count("fast3_") > 100   # (the check did see the kernels)


def test_a_stale_library_fails_with_the_abi_message(tmp_path):
    """The binding checks ttemb_abi_version() before it binds any other symbol, so a library of another ABI version is
    refused with the version message (round 4: AttributeError on the first symbol the old library lacked)."""
    import subprocess
    import sys
    src = tmp_path / "stale.c"
    src.write_text("int ttemb_abi_version(void) { return 2; }\n")
    lib = tmp_path / "libstale.so"
    subprocess.run(["gcc", "-shared", "-fPIC", str(src), "-o", str(lib)], check=True)
    code = f"import sys; sys.path.insert(0, {PKG!r}); import ttemb_native"
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, TTEMB_LIB=str(lib)), capture_output=True, text=True)
    assert r.returncode != 0 and "ABI version 2" in r.stderr and "AttributeError" not in r.stderr


def test_abi_argument_validation_without_gpu():
    lib = nat.LIB
    s = nat.make_shape([3, 4, 5], [2, 3, 2], [4, 3])
    s.T = 7
    assert lib.ttemb_workspace_bytes(ctypes.byref(s), nat.OP_FORWARD, 10, 10) == -1
    assert b"2..4" in lib.ttemb_last_error()
    s = nat.make_shape([3, 4, 5], [2, 3, 3], [4, 3])  # D = 18, not a multiple of 4
    assert lib.ttemb_workspace_bytes(ctypes.byref(s), nat.OP_BACKWARD, 10, 10) == -1
    assert b"multiple of 4" in lib.ttemb_last_error()
    s = nat.make_shape([125, 140, 140], [4, 5, 5], [16, 16])
    assert lib.ttemb_workspace_bytes(ctypes.byref(s), nat.OP_BACKWARD, 2048, 2048) >= 198400 * 4
    assert lib.ttemb_set_path(9) == -1
    with pytest.raises(RuntimeError):
        nat.make_shape([3, 4], [2, 2, 2], [4])


def test_workspace_and_plan_sizes_follow_the_kernel_family():
    """Host-side sizing only (no kernel runs): which family a (shape, batch) gets shows in what it asks for."""
    lib = nat.LIB
    G, n = 125 * 140, 409600
    wide = nat.make_shape([125, 140, 140], [5, 5, 4], [256, 256])
    narrow = nat.make_shape([125, 140, 140], [5, 5, 4], [32, 32])
    ws = lambda shp, op, nnz: lib.ttemb_workspace_bytes(ctypes.byref(shp), op, nnz, nnz)
    # wide-rank chain: the backward keeps a dP table of G * q0 q1 r2 floats; a SMALL call also an E table of nnz * r2 q2 floats,
    # a large one reduces dG2 in LDS and keeps partial slabs instead (one per share of the chunk table: p2 * r2 q2 floats each)
    small = 16384
    assert ws(wide, nat.OP_BACKWARD, small) >= small * 256 * 4 * 4 + G * 25 * 256 * 4
    assert G * 25 * 256 * 4 + 16 * 140 * 1024 * 4 <= ws(wide, nat.OP_BACKWARD, n) < n * 256 * 4 * 4   # (no 1.7 GB E table)
    assert ws(wide, nat.OP_BACKWARD, n) < 6 * 2**30
    assert ws(wide, nat.OP_FORWARD, n) < ws(wide, nat.OP_BACKWARD, n)
    # the plan carries the prefix products of every group
    assert lib.ttemb_plan_bytes(ctypes.byref(wide), n) >= G * 25 * 256 * 4
    assert lib.ttemb_plan_bytes(ctypes.byref(narrow), n) >= G * 25 * 32 * 4
    # rank 256 takes the grouped chain at every batch size, rank 64 only past its crossover (per-bag kernels: no plan)
    assert lib.ttemb_plan_bytes(ctypes.byref(wide), 8) > 0
    r64 = nat.make_shape([125, 140, 140], [5, 5, 4], [64, 64])
    assert lib.ttemb_plan_bytes(ctypes.byref(r64), 256) == 0
    assert lib.ttemb_plan_bytes(ctypes.byref(r64), 8192) > 0
    # the per-bag family can be forced for crossover measurements, and only known families are accepted
    assert lib.ttemb_set_path(nat.PATH_PER_BAG) == 0
    assert lib.ttemb_plan_bytes(ctypes.byref(wide), n) == 0
    assert lib.ttemb_set_path(nat.PATH_AUTO) == 0
    assert lib.ttemb_set_path(4) == -1


def test_a_large_merged_first_pair_needs_a_batch_that_amortises_its_rebuild():
    """4-core table, per-bag route (below the grouped crossover): the virtual first core G0.G1 is rebuilt per call, so a table
    whose first pair is large takes the merged per-bag view only for batches of at least one id per 16 rows of it; small
    virtual cores (the run scripts' 3 000 rows) always do.  Host-side decision only: ttemb_kernel_family launches nothing."""
    small = nat.make_shape([50, 60, 60, 60], [2, 4, 4, 4], [16, 16, 16])        # V: 3 000 rows x 128 floats = 1.5 MB
    large = nat.make_shape([700, 800, 60, 60], [2, 4, 4, 4], [16, 16, 16])      # V: 560 000 rows x 128 floats = 287 MB > 256 MB: never
    medium = nat.make_shape([300, 400, 60, 60], [2, 4, 4, 4], [16, 16, 16])     # V: 120 000 rows = 61 MB
    merged = lambda shp, n: bool(nat.kernel_family(shp, n, n, True) & nat.FAMILY_MERGED)
    assert merged(small, 8) and merged(small, 2048)
    assert not merged(large, 2048)
    assert not merged(medium, 256)           # 256 ids against 120 000 rows of V: the scalar kernels
    assert merged(medium, 120000 // 16)      # one id per 16 rows: the merged per-bag view


def test_size_queries_over_every_shape_and_size():
    """ttemb_workspace_bytes / ttemb_plan_bytes / ttemb_kernel_family are host arithmetic over (shape, nnz, B): swept over the
    (q, rank) list x small and large p2 x empty to large calls (a p2 whose dG2 slice does not fit a CU's LDS once divided by zero
    in the wide-rank sizing -- at rank 16, where that form is never taken)."""
    qs = ([4, 5, 5], [4, 4, 8], [8, 4, 4], [5, 4, 5], [5, 5, 4], [4, 4, 4], [2, 2, 4], [16, 4, 2])
    for q in qs:
        for r in (4, 8, 16, 32, 64, 128, 256, 24):
            for p in ([125, 140, 140], [400, 500, 600], [3, 2, 5000], [1, 1, 1], [7, 300, 900]):
                shape = nat.make_shape(p, q, [r, r])
                for nnz, B in ((0, 0), (0, 7), (1, 1), (5000, 1200), (409600, 409600), (819200, 819200), (6000000, 100)):
                    for op in (nat.OP_FORWARD, nat.OP_BACKWARD):
                        assert nat.workspace_bytes(shape, op, nnz, B) >= 0
                    assert nat.plan_bytes(shape, nnz) >= 0
                    assert nat.kernel_family(shape, nnz, B, True) >= 0 and nat.kernel_family(shape, nnz, B, False) >= 0


def test_window_size_query_is_host_arithmetic():
    """ttemb_window_workspace_bytes: what a window call (one table of a table-batched call) needs, -1 through the binding when
    the grouped kernels do not serve the window -- the caller then splits the id list on the host."""
    ok = nat.make_shape([10, 12, 9], [4, 4, 8], [8, 8])
    f = nat.window_workspace_bytes(ok, nat.OP_FORWARD, 1000, 80, 20)
    b = nat.window_workspace_bytes(ok, nat.OP_BACKWARD, 1000, 80, 20)
    assert 40960 < f <= b
    assert nat.window_workspace_bytes(ok, nat.OP_FORWARD, 0, 80, 20) == 40960      # nothing but the header
    assert nat.window_workspace_bytes(nat.make_shape([125, 140, 140], [5, 5, 4], [256, 256]), nat.OP_BACKWARD, 100000, 3000, 1000) > 0
    assert nat.window_workspace_bytes(nat.make_shape([7, 9, 11, 5], [2, 2, 5, 4], [5, 6, 3]), nat.OP_FORWARD, 1000, 80, 20) == -1
    assert nat.window_workspace_bytes(nat.make_shape([10, 12, 9], [4, 4, 8], [12, 12]), nat.OP_FORWARD, 1000, 80, 20) == -1
    with pytest.raises(RuntimeError):
        nat.window_workspace_bytes(ok, nat.OP_FORWARD, 1000, 10, 20)                # more bags in the window than in the call


def test_kernel_family_reports_the_routes_of_the_grouped_path():
    """Host-side rules only (ttemb_kernel_family launches nothing): which frontiers form their prefix products in the forward
    chain kernel and their group products in the backward chunk kernel."""
    G, PF, GP = nat.FAMILY_GROUPED, nat.FAMILY_PREFIX_IN_CHAIN, nat.FAMILY_GROUP_PRODUCTS_IN_CHAIN
    fam = lambda p, q, r, n: nat.kernel_family(nat.make_shape(p, q, r), n, n, True)
    # BASELINE configs[1]: 23 ids per group, dG2 fused into the chunk kernel: neither route
    assert fam([125, 140, 140], [4, 5, 5], [16, 16], 409600) == G
    # BASELINE configs[4]'s table, 819 200 ids (3 per group): both
    assert fam([500, 560, 400], [8, 4, 4], [32, 32], 819200) == G | PF | GP
    # ... 8.6 ids per group: still both (q0 = 8: up to 16); 20 per group: the backward's route only (rank 32: every density)
    assert fam([500, 560, 400], [8, 4, 4], [32, 32], 2400000) == G | PF | GP
    assert fam([100, 112, 400], [8, 4, 4], [32, 32], 224000) == G | GP
    assert fam([500, 560, 400], [8, 4, 4], [32, 32], 5600000) == G            # past one piece: each piece is routed by itself
    # the products table at rank 32 (unfused dG2): the backward's route at 23 ids per group, the forward's not (and never at rank 32
    # for q0 = 4: that instance spilled)
    assert fam([125, 140, 140], [4, 5, 5], [32, 32], 409600) == G | GP
    assert fam([125, 140, 140], [4, 5, 5], [32, 32], 98000) == G | GP
    # the reference's own papers100M invocation (run_script.sh:408-431): p2 = 600 is past the fused dG2 form at rank 16
    assert fam([400, 500, 600], [4, 4, 8], [16, 16], 819200) == G | PF | GP
    assert fam([100, 125, 600], [4, 4, 8], [16, 16], 250000) == G             # 20 ids per group at rank 16: neither
    # wide ranks: their own family, no route flags
    assert fam([125, 140, 140], [5, 5, 4], [256, 256], 409600) == nat.FAMILY_GROUPED_WIDE


def test_status_and_spin_limit_are_host_side_calls():
    """ttemb_status() reads a pinned host word that ttemb_init() (or the first status call) creates: without a GPU there is
    no such memory, init says so once and status reports nothing; the spin limit is a process-wide diagnostic value."""
    assert nat.LIB.ttemb_init() in (0, -4)   # (-4 = TTEMB_E_HIP on a box without a device: "reported through NaN results only")
    nat.init()                                # the binding's wrapper never raises
    nat.status()
    nat.set_spin_limit(-1)
    nat.set_spin_limit(123)
    nat.set_spin_limit(0)
    nat.status()


def test_suggested_shapes_match_reference_answers():
    t = load_golden("suggest_kat")["table"]
    for row in t.tolist():
        n, d, up = row[:3]
        assert suggested_tt_shapes(n, d, allow_round_up=bool(up)) == row[3:3 + d]
    assert suggested_tt_shapes(2449029, 3) == [125, 140, 140]
    assert suggested_tt_shapes(100, 3) == [4, 5, 5]


def test_tt_matrix_to_full_matches_oracle_and_is_differentiable():
    g = load_golden("tt_tiny_T3")
    cores = [torch.tensor(g[f"core{t}"]).unsqueeze(0).requires_grad_(True) for t in range(3)]
    full = tt_matrix_to_full(g["p"].tolist(), g["q"].tolist(), g["R"].tolist(), cores, [1, 0, 2, 3])
    ref = orc.tt_full_table([g[f"core{t}"] for t in range(3)], g["p"], g["q"], g["R"])
    np.testing.assert_allclose(full.detach().numpy(), ref, rtol=1e-5, atol=1e-5)
    out = torch.nn.functional.embedding_bag(torch.tensor(g["indices"]), full, torch.tensor(g["offsets"]),
                                            mode="sum", include_last_offset=True)
    out.backward(torch.tensor(g["d_output"]))
    for t in range(3):
        np.testing.assert_allclose(cores[t].grad[0].numpy(), g[f"grad{t}"], rtol=1e-4, atol=1e-5)
    # un-permuted layout ([R, p, q, R'])
    c2 = [torch.tensor(g[f"core{t}"]).reshape(g["p"][t], g["R"][t], g["q"][t], g["R"][t + 1]).permute(1, 0, 2, 3)
          .contiguous() for t in range(3)]
    full2 = tt_matrix_to_full(g["p"].tolist(), g["q"].tolist(), g["R"].tolist()[1:-1], c2)
    np.testing.assert_allclose(full2.numpy(), ref, rtol=1e-5, atol=1e-5)


def test_module_state_layout_matches_reference_contract():
    m = TTEmbeddingBag(169343, 128, [8, 8], [56, 60, 51], [4, 4, 8], sparse=False, use_cache=True,
                       cache_size=1000, hashtbl_size=5000, weight_dist="normal")
    sd = m.state_dict()
    assert sd["L"].tolist() == [60 * 51, 51, 1] and sd["L"].dtype == torch.int64
    assert tuple(sd["tt_cores.0"].shape) == (1, 56, 32)
    assert tuple(sd["tt_cores.1"].shape) == (1, 60, 8 * 4 * 8)
    assert tuple(sd["tt_cores.2"].shape) == (1, 51, 64)
    assert tuple(sd["optimizer_state.optimizer_state0"].shape) == (0,)
    assert sd["hashtbl"].dtype == torch.int64 and sd["hashtbl"].numel() == 5000 and (sd["hashtbl"] == -1).all()
    assert sd["cache_freq"].dtype == torch.int64 and (sd["cache_freq"] == 0).all()
    assert sd["cache_state"].dtype == torch.int32 and (sd["cache_state"] == -1).all()
    assert tuple(sd["cache_weight"].shape) == (1000, 128)
    assert m.warmup is True and m.tt_ranks == [1, 8, 8, 1] and m.tt_ndim == 3
    assert isinstance(m.tt_cores, torch.nn.ParameterList) and isinstance(m.optimizer_state, BufferList)
    m2 = TTEmbeddingBag(1000, 16, [4, 4], optimizer=OptimType.EXACT_ADAGRAD, use_cache=True, weight_dist="uniform")
    assert m2.tt_p_shapes == [10, 10, 10] and np.prod(m2.tt_q_shapes) == 16
    assert tuple(m2.optimizer_state[1].shape) == tuple(m2.tt_cores[1].shape)
    assert m2.cache_weight.shape[0] == 100 and m2.hashtbl.numel() == 1000  # defaults: 10 % / num_embeddings
    assert tuple(m2.cache_optimizer_state.shape) == (100, 16)
    m3 = TTEmbeddingBag(1000, 16, [4, 4], optimizer=OptimType.EXACT_ROWWISE_ADAGRAD, use_cache=True,
                        weight_dist="uniform")
    assert tuple(m3.cache_optimizer_state.shape) == (100,)
    m4 = TableBatchedTTEmbeddingBag(3, 1000, 16, [4], [25, 40], [4, 4], weight_dist="naive-uniform")
    assert tuple(m4.tt_cores[0].shape) == (3, 25, 16) and m4.cache_weight is None and m4.hashtbl.numel() == 0
    # round trip
    m5 = TTEmbeddingBag(169343, 128, [8, 8], [56, 60, 51], [4, 4, 8], sparse=False, use_cache=True,
                        cache_size=1000, hashtbl_size=5000, weight_dist="normal")
    m5.load_state_dict(sd)
    assert torch.equal(m5.tt_cores[1].data, m.tt_cores[1].data)


def test_constructor_asserts_like_reference():
    with pytest.raises(AssertionError):
        TTEmbeddingBag(1000, 16, [4, 4], [5, 5, 5], [2, 2, 4], weight_dist="normal")  # prod(p) < n
    with pytest.raises(AssertionError):
        TTEmbeddingBag(1000, 16, [4, 4], [10, 10, 10], [2, 2, 5], weight_dist="normal")  # prod(q) != D
    with pytest.raises(AssertionError):
        TTEmbeddingBag(1000, 16, [4], [10, 10, 10], [2, 2, 4], weight_dist="normal")  # rank count
    with pytest.raises(AssertionError):
        TableBatchedTTEmbeddingBag(2, 1000, 16, [4], [25, 40], [4, 4], use_cache=True, weight_dist="normal")
    with pytest.raises(AssertionError):
        TTEmbeddingBag(1000, 16, [4, 4], [10, 10, 10], [2, 2, 4], weight_dist="bogus")


@pytest.mark.parametrize("dist", ["uniform", "naive-uniform", "normal", "approx-normal", "approx-uniform"])
def test_initialisers(dist):
    np.random.seed(0)
    torch.manual_seed(0)
    n = 27000
    m = TTEmbeddingBag(n, 64, [8, 8], [30, 30, 30], [4, 4, 4], use_cache=False, weight_dist=dist)
    cores = [c.detach().numpy() for c in m.tt_cores]
    assert all(np.isfinite(c).all() for c in cores)
    if dist == "normal":
        for c in cores:
            assert abs(c.std() - 1 / np.sqrt(n)) < 0.1 / np.sqrt(n) and abs(c.mean()) < 3e-4
    elif dist == "naive-uniform":
        for c in cores:
            assert c.min() >= 0 and c.max() <= 1 / np.sqrt(n) and abs(c.mean() - 0.5 / np.sqrt(n)) < 1e-4
    elif dist == "uniform":
        hi = np.sqrt(2.0 / (n + 64)) ** (1 / 3) * np.prod(np.array([1, 8, 8, 1.0]) ** (-1 / 6))
        for c in cores:
            assert c.min() >= 0 and c.max() <= hi and c.max() > 0.95 * hi
    elif dist == "approx-normal":
        s = (1 / np.sqrt(3 * n)) ** (1 / 3)
        for c in cores:
            assert np.abs(c).min() >= 2 * s * (1 - 1e-6)
    else:
        # the product table should look roughly uniform on (-1, 1) / sqrt(n): mean |x| ~ 0.5 / sqrt(n)
        full = m.full_weight().detach().numpy()
        assert np.isfinite(full).all() and 0.3 < np.abs(full).mean() * np.sqrt(n) < 0.7


def test_no_cpu_fallback():
    m = TTEmbeddingBag(1000, 16, [4, 4], [10, 10, 10], [2, 2, 4], use_cache=False, weight_dist="normal")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.arange(4), torch.arange(5))


def test_optimtype_values():
    assert str(OptimType.SGD) == "sgd" and OptimType("exact_row_wise_adagrad") is OptimType.EXACT_ROWWISE_ADAGRAD
    assert len(OptimType) == 9


# ---- initialisers applied by the drivers (tt_utils.py:117-201) ---------------------------------

def _full(cores, p, q, r):
    from FBTT.tt_embeddings_ops import tt_matrix_to_full
    return tt_matrix_to_full(p, q, [1] + list(r) + [1], [c for c in cores], [1, 0, 2, 3])


@pytest.mark.parametrize("p,q,r", [([6, 5, 4], [2, 3, 2], [4, 3]), ([7, 6], [4, 2], [5]), ([3, 4, 3, 2], [2, 2, 2, 2], [3, 4, 2])])
def test_tt_svd_recovers_a_tt_matrix(p, q, r):
    import ttemb_init
    g = torch.Generator().manual_seed(7)
    T = len(p)
    R = [1] + r + [1]
    truth = [torch.randn(1, p[t], R[t] * q[t] * R[t + 1], generator=g) for t in range(T)]
    table = _full(truth, p, q, r)
    cores, ranks = ttemb_init.tt_svd_cores(table, r, p, q)
    assert ranks == R
    for t in range(T):
        assert cores[t].shape == (1, p[t], R[t] * q[t] * R[t + 1]) and cores[t].dtype == torch.float32
    again = _full(cores, p, q, r)
    assert (again - table).abs().max().item() < 1e-4 * table.abs().max().item()
    # the oracle reads the same cores the same way
    ids = np.arange(int(np.prod(p)))
    rows = orc.tt_rows(ids, [c.numpy()[0] for c in cores], p, q, R)
    np.testing.assert_allclose(rows, table.numpy(), atol=1e-4 * float(table.abs().max()))


def test_tt_svd_truncation_is_monotone_and_shim_matches():
    import ttemb_init
    import ttemb_tt_utils as tt_utils
    p, q = [8, 6, 5], [2, 2, 3]
    table = torch.randn(240, 12, generator=torch.Generator().manual_seed(3))
    errs = []
    for rank in (1, 2, 4, 8, 64):
        cores, ranks = ttemb_init.tt_svd_cores(table, [rank, rank], p, q)
        errs.append((_full(cores, p, q, ranks[1:-1]) - table).norm().item())
    assert all(a >= b - 1e-5 for a, b in zip(errs, errs[1:])) and errs[-1] < 1e-3
    cores, ranks = tt_utils.tt_matrix_decomp(table.numpy(), [1, 4, 4, 1], p, q)   # reference call shape
    assert ranks == [1, 4, 4, 1] and all(torch.is_tensor(c) and c.device.type == "cpu" for c in cores)
    assert abs((_full(cores, p, q, [4, 4]) - table).norm().item() - errs[2]) < 1e-3


def test_ortho_cores_are_orthonormal_frames():
    import ttemb_init
    import ttemb_tt_utils as tt_utils
    p, q, r = [12, 14, 48], [4, 5, 5], [8, 8]
    R = [1] + r + [1]
    cores = ttemb_init.ortho_cores(r, p, q, generator=torch.Generator().manual_seed(1))
    for t, c in enumerate(cores):
        assert c.shape == (1, p[t], R[t] * q[t] * R[t + 1])
        core4 = c.reshape(p[t], R[t], q[t], R[t + 1]).permute(1, 2, 0, 3).reshape(R[t] * q[t], p[t] * R[t + 1])
        gram = core4 @ core4.t()
        assert (gram - torch.eye(gram.shape[0])).abs().max().item() < 1e-5
    got = tt_utils.get_ortho([1, 8, 8, 1], p, q)
    assert all(isinstance(c, np.ndarray) and c.dtype == np.float32 for c in got)
    with pytest.raises(ValueError):
        ttemb_init.ortho_cores([64, 64], [2, 2, 2], [4, 5, 5])


def test_prefix_locality_metric():
    import ttemb_init
    p = [125, 140, 140]
    g = torch.Generator().manual_seed(0)
    uniform = torch.randperm(2449029, generator=g)[:40960]
    starts = torch.randint(0, 2449029 - 200, (205,), generator=g)
    windows = (starts[:, None] + torch.arange(200)[None, :]).reshape(-1)
    u, w = ttemb_init.prefix_locality(uniform, p), ttemb_init.prefix_locality(windows, p)
    assert u["ids"] == 40960 and w["ids"] == 41000
    assert w["distinct_prefixes"] <= 3 * 205 and w["ids_per_prefix"] > 20 * u["ids_per_prefix"]
    assert ttemb_init.prefix_locality(torch.empty(0, dtype=torch.long), p)["distinct_prefixes"] == 0


def test_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under the package (Python or HIP sources) may import, call or even
    name it; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do."""
    import pathlib
    root = pathlib.Path(__file__).resolve().parents[1]
    offenders = []
    for path in (root / "falcon-ttdforgnns_amd").rglob("*"):
        if path.suffix in (".py", ".hip", ".h", ".cpp") and "oracle" in path.read_text(errors="ignore"):
            offenders.append(str(path.relative_to(root)))
    assert offenders == []
    bench = (root / "bench.py").read_text()
    assert bench.count("from oracle import") == 1 and "cpu_einsum" in bench   # the cpu_baseline leg only


def test_import_resolution_next_to_a_reference_shaped_tree(tmp_path):
    """the package directory goes in FRONT of the reference tree on sys.path.  The reference keeps
    `FBTT/` as a namespace directory (no __init__.py) and has a top-level `tt_utils.py` that every driver star-imports
    for its argument parser.  With a fake tree of that shape: `FBTT.tt_embeddings_ops` must resolve to this package,
    `tt_utils` to the reference's file (this package ships its initialisers as `ttemb_tt_utils`)."""
    import subprocess
    import sys
    ref = tmp_path / "reference"
    (ref / "FBTT").mkdir(parents=True)
    (ref / "FBTT" / "tt_embeddings_ops.py").write_text("WHO = 'reference'\n")
    (ref / "tt_utils.py").write_text("WHO = 'reference'\ndef parse_args():\n    return 'reference parser'\n")
    (ref / "driver.py").write_text(
        "from tt_utils import *\n"
        "import tt_utils, FBTT.tt_embeddings_ops as ops, ttemb_tt_utils\n"
        "print(tt_utils.WHO, parse_args(), ops.__file__, hasattr(ops, 'TTEmbeddingBag'), ttemb_tt_utils.__file__)\n")
    for how in ("pythonpath", "insert"):
        env = dict(os.environ)
        if how == "pythonpath":
            env["PYTHONPATH"] = PKG + os.pathsep + env.get("PYTHONPATH", "")
            cmd = [sys.executable, "driver.py"]
        else:
            cmd = [sys.executable, "-c", f"import sys; sys.path.insert(0, {str(ref)!r}); sys.path.insert(0, {PKG!r}); "
                                         "exec(open('driver.py').read())"]
        out = subprocess.run(cmd, cwd=ref, env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr
        who, parser, _, ops_file, has_class, shim = out.stdout.split()   # "reference parser" prints as two words
        assert who == "reference" and parser == "reference"
        assert ops_file.startswith(PKG) and has_class == "True" and shim.startswith(PKG)


def test_bench_starts_its_own_ranks_when_started_plainly(monkeypatch):
    """`python bench.py --gpus N` without a launcher: the same command line goes under torch.distributed.run as a child
    process (one rank per GPU, rendezvous on 127.0.0.1) before the parent has touched the GPU."""
    import importlib.util
    import subprocess
    import sys
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return subprocess.CompletedProcess(cmd, 0)

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert bench._launch_ranks(4) == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "7"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
