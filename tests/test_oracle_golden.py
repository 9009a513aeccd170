"""CPU: the oracle (oracle/tt_oracle.py) against the golden vectors generated from the
reference's own tt_matrix_to_full / autograd (tests/golden/make_golden.py)."""
import hashlib

import numpy as np
import pytest

from conftest import RANK_CASES, ROW_CASES, TINY_CASES, golden_cores, load_golden, rank_case_cores, seeded_cores
from oracle import tt_oracle as orc


@pytest.mark.parametrize("name", TINY_CASES)
def test_forward_matches_reference(name):
    g = load_golden(name)
    out = orc.tt_forward(g["indices"], g["offsets"], golden_cores(g), g["p"], g["q"], g["R"])
    np.testing.assert_allclose(out, g["out"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("name", TINY_CASES)
def test_dense_backward_matches_reference_autograd(name):
    g = load_golden(name)
    grads = orc.tt_dense_backward(g["indices"], g["offsets"], g["d_output"], golden_cores(g), g["p"], g["q"],
                                  g["R"])
    for t, gr in enumerate(grads):
        np.testing.assert_allclose(gr, g[f"grad{t}"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("name", TINY_CASES)
def test_fused_updates_match_closed_form(name):
    g = load_golden(name)
    cores = golden_cores(g)
    grads = [g[f"grad{t}"] for t in range(len(cores))]
    sgd = orc.sgd_step(cores, grads, g["lr"])
    ada, st = orc.adagrad_step(cores, [np.zeros_like(c) for c in cores], grads, g["lr"], g["eps"])
    for t in range(len(cores)):
        np.testing.assert_allclose(sgd[t], g[f"sgd{t}"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(st[t], g[f"ada_state{t}"], rtol=1e-6, atol=0)
        np.testing.assert_allclose(ada[t], g[f"ada{t}"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name", RANK_CASES)
def test_rank_sweep_points_match_reference(name):
    """(q, rank) points of run_script.sh:250-288: bag sums and autograd gradients through the reference's tt_matrix_to_full."""
    g = load_golden(name)
    cores = rank_case_cores(g)
    out = orc.tt_forward(g["indices"], g["offsets"], cores, g["p"], g["q"], g["R"])
    np.testing.assert_allclose(out, g["out"], rtol=1e-5, atol=1e-5)
    grads = orc.tt_dense_backward(g["indices"], g["offsets"], g["d_output"], cores, g["p"], g["q"], g["R"])
    np.testing.assert_allclose(grads[0], g["grad0"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(grads[2], g["grad2"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(grads[1].reshape(-1)[::61], g["grad1_every61"], rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("name", ROW_CASES)
def test_big_config_rows(name):
    g = load_golden(name)
    cores = seeded_cores(g["p"], g["q"], g["R"], g["seed"], g["core_scale"])
    h = hashlib.sha256()
    for c in cores:
        h.update(np.ascontiguousarray(c).tobytes())
    assert h.hexdigest() == str(g["cores_sha256"]), "RNG drift: regenerate the golden vectors"
    rows = orc.tt_rows(g["indices"], cores, g["p"], g["q"], g["R"])
    np.testing.assert_allclose(rows, g["rows"], rtol=1e-5, atol=1e-4)
    assert g["indices"].max() == g["num_embeddings"] - 1  # the last valid id is covered
    if name == "rows_papers":
        assert g["indices"].max() > 2 ** 24  # exercises ids a float32 cannot represent


def test_murmur_known_answers():
    g = load_golden("murmur_kat")
    for a, C in enumerate(g["sizes"].tolist()):
        assert (orc.murmur_slots(g["keys"], C) == g["slots"][a]).all()
        for b, k in enumerate(g["keys"].tolist()):
            assert orc.murmur_slot(k, C) == g["slots"][a, b]
            assert 0 <= g["slots"][a, b] < C


def test_murmur_body_is_standard_murmur3():
    from sklearn.utils import murmurhash3_32
    for k in [0, 1, 12345, 2449028, 2 ** 40 + 7]:
        raw = int(k).to_bytes(8, "little")
        assert orc.murmur_word(k, len_xor=8) == murmurhash3_32(raw, seed=0, positive=True)


def test_index_split_and_rowidx():
    p = [125, 140, 140]
    ids = np.array([0, 139, 140, 19599, 19600, 2449028], dtype=np.int64)
    i0, i1, i2 = orc.split_index(ids, p)
    assert (i0 * 19600 + i1 * 140 + i2 == ids).all()
    assert i0.tolist() == [0, 0, 0, 0, 1, 124]
    assert orc.strides_L(p) == [19600, 140, 1]
    row = orc.rowidx_from_offsets([0, 2, 2, 5], 5)
    assert row.tolist() == [0, 0, 2, 2, 2]


def test_partition_order_is_cub_flagged():
    arr = np.arange(8)
    flags = np.array([1, 0, 1, 1, 0, 0, 1, 0], dtype=bool)
    assert orc.partition_by_flag(arr, flags).tolist() == [0, 2, 3, 6, 7, 5, 4, 1]


def test_cache_lifecycle_oracle():
    rng = np.random.default_rng(3)
    H, C = 257, 16
    hot = rng.choice(5000, size=8, replace=False)
    stream = np.concatenate([np.repeat(hot, 9), rng.choice(5000, size=60, replace=False)]).astype(np.int64)
    rng.shuffle(stream)
    keys = np.full(H, -1, dtype=np.int64)
    freq = np.zeros(H, dtype=np.int64)
    state = np.full(H, -1, dtype=np.int32)
    failed = orc.update_cache_state(stream, keys, freq)
    assert freq.sum() == stream.shape[0] - failed
    kept = orc.cache_populate(keys, freq, state, C)
    assert kept.shape[0] == C
    # the hottest ids that made it into the table are cached, and nothing else survives
    assert set(keys[keys >= 0].tolist()) == set(kept.tolist()) - ({0} if 0 not in stream else set())
    for k in hot:
        if k in kept:
            s = orc.hashtbl_find(k, keys)
            assert state[s] >= 0
    idx = np.concatenate([hot[:4], np.array([4999, 4998])]).astype(np.int64)
    offs = np.arange(idx.shape[0] + 1)
    pi, pr, ntt, loc = orc.preprocess_indices(idx, offs, False, keys, state)
    assert ntt + (loc[ntt:] >= 0).sum() == idx.shape[0]
    assert sorted(pi.tolist()) == sorted(idx.tolist())
    assert (np.diff(pr[:ntt]) >= 0).all() and (np.diff(pr[ntt:]) <= 0).all()


def test_update_after_eviction_reference_vs_find_first():
    """hashtbl_insert probes and inserts in one sweep (hashtbl_cuda_utils.cuh:102-133): once
    cache_populate has evicted the entry in front of a displaced cached key, the reference inserts that
    key a second time and the lookup then lands on the copy with cache_state -1.  The HIP path looks for
    the key first (oracle: find_first=True); before any eviction the two agree."""
    H = 64
    keys = [k for k in range(20000) if orc.murmur_slot(k, H) == 5][:3]
    a, b, c = keys
    for find_first in (False, True):
        tbl, freq = np.full(H, -1, np.int64), np.zeros(H, np.int64)
        state = np.full(H, -1, np.int32)
        orc.update_cache_state([a, b, b, b, c, c], tbl, freq, find_first=find_first)
        assert tbl[5] == a and tbl[6] == b and tbl[7] == c          # same table during warm-up
        assert freq[5] == 1 and freq[6] == 3 and freq[7] == 2
        orc.cache_populate(tbl, freq, state, 2)                      # keeps b and c, evicts a
        assert tbl[5] == -1 and state[6] == 0 and state[7] == 1
        orc.update_cache_state([c], tbl, freq, find_first=find_first)
        is_tt, loc = orc.cache_lookup(np.array([c]), tbl, state)
        if find_first:
            assert tbl[5] == -1 and freq[7] == 3 and not is_tt[0] and loc[0] == 1
        else:
            assert tbl[5] == c and is_tt[0]                          # the cached id fell out of the cache
