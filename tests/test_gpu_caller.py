"""GPU: the caller contract either side of the layer.

* `SAGE.forward` / `SAGE.inference` (gnn_model.py:193-253): `offsets = arange(N+1)`, frontier ids with the
  seeds first, `inference` looks the whole table up in one call;
* a 20-step training loss curve through the MI355X layer against the same model with the embedding done
  on the CPU by autograd through the dense table;
* `state_dict()` files interchange and a populated cache survives a save/load.
"""
import io
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import ROOT

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available()
    import FBTT.tt_embeddings_ops as m
    return m


P, Q, R = [20, 25, 20], [4, 5, 5], [16, 16]
N_NODES, D = 10000, 100


def test_inference_lookup_of_the_whole_table(ops):
    torch.manual_seed(3)
    emb = ops.TTEmbeddingBag(N_NODES, D, R, P, Q, sparse=True, use_cache=False, weight_dist="normal")
    for c in emb.tt_cores:
        c.data.mul_(30.0)
    ids = torch.arange(N_NODES, device="cuda")
    with torch.no_grad():
        out = emb(ids, torch.arange(N_NODES + 1, device="cuda"))
    full = emb.full_weight()[:N_NODES]
    assert out.shape == (N_NODES, D)
    assert (out - full).abs().max().item() < 1e-4


def test_twenty_step_loss_curve_matches_cpu_table_autograd(ops):
    import sage_epoch as harness
    from FBTT.tt_embeddings_ops import tt_matrix_to_full
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(5)
    indptr, indices = harness.build_graph(N_NODES, 10, 0.5, dev, gen)
    labels = torch.randint(0, 7, (N_NODES,), device=dev, generator=gen)
    lut = torch.full((N_NODES,), -1, dtype=torch.int64, device=dev)
    lr_emb, steps, batch = 0.05, 20, 128
    torch.manual_seed(11)
    emb = ops.TTEmbeddingBag(N_NODES, D, R, P, Q, sparse=True, use_cache=False, weight_dist="normal",
                             learning_rate=lr_emb)
    for c in emb.tt_cores:
        c.data.mul_(30.0)
    layers = torch.nn.ModuleList([harness.MeanSAGE(D, 32), harness.MeanSAGE(32, 7)]).to(dev)
    # the CPU replica: same weights, embedding rows by autograd through the dense table
    cpu_cores = [c.detach().cpu().clone().requires_grad_(True) for c in emb.tt_cores]
    cpu_layers = torch.nn.ModuleList([harness.MeanSAGE(D, 32), harness.MeanSAGE(32, 7)])
    cpu_layers.load_state_dict({k: v.cpu() for k, v in layers.state_dict().items()})
    opt = torch.optim.Adam(layers.parameters(), lr=0.01)         # tt_utils.py:23
    cpu_opt = torch.optim.Adam(cpu_layers.parameters(), lr=0.01)
    curve, cpu_curve = [], []
    for s in range(steps):
        seeds = torch.randint(0, N_NODES, (batch,), device=dev, generator=gen).unique()
        input_nodes, blocks = harness.sample_blocks(indptr, indices, seeds, [5, 5], gen, lut)
        assert torch.equal(input_nodes[: seeds.numel()], seeds)        # seeds first (gnn_model.py:211)
        assert input_nodes.unique().numel() == input_nodes.numel()

        h = emb(input_nodes, torch.arange(input_nodes.numel() + 1, device=dev))
        x = h
        for li, (layer, block) in enumerate(zip(layers, blocks)):
            x = layer(x, block)
            x = F.relu(x) if li == 0 else x
        loss = F.cross_entropy(x, labels[seeds])
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        curve.append(loss.item())

        table = tt_matrix_to_full(P, Q, [1] + R + [1], cpu_cores, [1, 0, 2, 3])
        x = table[input_nodes.cpu()]
        for li, (layer, block) in enumerate(zip(cpu_layers, blocks)):
            x = layer(x, tuple(b.cpu() if torch.is_tensor(b) else b for b in block))
            x = F.relu(x) if li == 0 else x
        cpu_loss = F.cross_entropy(x, labels[seeds].cpu())
        cpu_opt.zero_grad(set_to_none=True)
        grads = torch.autograd.grad(cpu_loss, cpu_cores + list(cpu_layers.parameters()))
        with torch.no_grad():
            for c, g in zip(cpu_cores, grads[: len(cpu_cores)]):
                c -= lr_emb * g                                     # the layer's fused SGD
        for prm, g in zip(cpu_layers.parameters(), grads[len(cpu_cores):]):
            prm.grad = g
        cpu_opt.step()
        cpu_curve.append(cpu_loss.item())
    curve, cpu_curve = np.array(curve), np.array(cpu_curve)
    assert np.isfinite(curve).all()
    np.testing.assert_allclose(curve, cpu_curve, rtol=2e-3, atol=2e-3)
    for c, ref in zip(emb.tt_cores, cpu_cores):
        assert (c.detach().cpu() - ref.detach()).abs().max().item() < 2e-3 * ref.detach().abs().max().item()


def test_state_dict_interchange_and_cache_survives_reload(ops):
    torch.manual_seed(9)
    rng = np.random.default_rng(9)
    kw = dict(sparse=True, use_cache=True, cache_size=300, hashtbl_size=N_NODES, weight_dist="normal",
              learning_rate=0.05)
    emb = ops.TTEmbeddingBag(N_NODES, D, R, P, Q, **kw)
    for c in emb.tt_cores:
        c.data.mul_(30.0)
    hot = rng.choice(N_NODES, size=250, replace=False)
    batches = [np.concatenate([rng.choice(hot, size=1500), rng.integers(0, N_NODES, size=1500)]) for _ in range(6)]
    for b in batches[:4]:
        ids = torch.tensor(b).cuda()
        o = emb(ids, torch.arange(ids.numel() + 1).cuda())
        o.backward((torch.rand_like(o) - 0.5) * 0.01)
    emb.cache_populate()
    sd = emb.state_dict()
    # key names / shapes / dtypes of the reference's module (tt_embeddings_ops.py:519-614)
    want = {"L": torch.int64, "tt_cores.0": torch.float32, "tt_cores.1": torch.float32, "tt_cores.2": torch.float32,
            "hashtbl": torch.int64, "cache_freq": torch.int64, "cache_state": torch.int32,
            "cache_weight": torch.float32}
    for k, dt in want.items():
        assert k in sd and sd[k].dtype == dt, k
    assert sd["tt_cores.1"].shape == (1, P[1], R[0] * Q[1] * R[1]) and sd["cache_weight"].shape == (300, D)
    blob = io.BytesIO()
    torch.save(sd, blob)
    blob.seek(0)
    emb2 = ops.TTEmbeddingBag(N_NODES, D, R, P, Q, **kw)
    assert emb2.warmup
    emb2.load_state_dict(torch.load(blob))
    assert not emb2.warmup                      # a populated cache is live after the load
    for it, b in enumerate(batches[4:]):        # both copies now train identically, cache and cores
        ids = torch.tensor(b).cuda()
        offs = torch.arange(ids.numel() + 1).cuda()
        o1, o2 = emb(ids, offs), emb2(ids, offs)
        assert torch.isfinite(o1).all()
        bad = ((o1 - o2).abs().max(1).values > 1e-5).nonzero().flatten()
        assert bad.numel() == 0, (it, bad.numel(), bad[:8].tolist(), ids[bad[:8]].tolist(),
                                  (o1 - o2).abs().max().item())
        g = (torch.rand_like(o1) - 0.5) * 0.01
        o1.backward(g)
        o2.backward(g)
    sd1, sd2 = emb.state_dict(), emb2.state_dict()
    for k in sd1:
        a, b = sd1[k], sd2[k]
        if a.dtype.is_floating_point and a.numel():
            assert (a - b).abs().max().item() < 1e-4 * max(1.0, a.abs().max().item()), k
        elif k in ("L", "cache_state"):
            assert torch.equal(a, b), k
    # ids first seen after the populate race for free slots (CAS order), so only the cached entries are
    # pinned slot by slot; the tracked id sets still agree up to probe-limit insert failures.  No key
    # sits in two slots (the reference's insert would duplicate cached ids into eviction holes).
    tracked = sd1["hashtbl"][sd1["hashtbl"] >= 0]
    assert tracked.unique().numel() == tracked.numel()
    live = sd1["cache_state"] >= 0
    assert torch.equal(sd1["hashtbl"][live], sd2["hashtbl"][live])
    assert torch.equal(sd1["cache_freq"][live], sd2["cache_freq"][live])
    k1 = set(sd1["hashtbl"][sd1["hashtbl"] >= 0].tolist())
    k2 = set(sd2["hashtbl"][sd2["hashtbl"] >= 0].tolist())
    assert len(k1 ^ k2) <= 0.01 * len(k1)
    # a warm-up-phase checkpoint stays in warm-up
    emb3 = ops.TTEmbeddingBag(N_NODES, D, R, P, Q, **kw)
    emb4 = ops.TTEmbeddingBag(N_NODES, D, R, P, Q, **kw)
    emb4.load_state_dict(emb3.state_dict())
    assert emb4.warmup


@pytest.mark.parametrize("n", [2048, 16384])
def test_training_step_replays_from_a_hip_graph(ops, n):
    """The library only enqueues on the caller's stream (no sync, no hidden allocation, no memset node), so a whole
    step -- forward, backward, fused SGD -- can be captured once and replayed; 2 048 ids take the wave-per-id kernels,
    16 384 the grouped MFMA path.  Replays must leave the cores where the same number of eager steps leaves them."""
    torch.manual_seed(4)
    rng = np.random.default_rng(4)
    emb = ops.TTEmbeddingBag(2449029, 100, [16, 16], [125, 140, 140], [4, 5, 5], sparse=True, use_cache=False,
                             weight_dist="normal", learning_rate=0.05)
    for c in emb.tt_cores:
        c.data.mul_(300.0)
    ids = torch.from_numpy(rng.choice(2449029, size=n, replace=False).astype(np.int64)).cuda()
    offs = torch.arange(n + 1, device="cuda")
    d_out = (torch.rand(n, 100, device="cuda") - 0.5) * 0.02

    def step():
        emb(ids, offs).backward(d_out)

    start = [c.detach().clone() for c in emb.tt_cores]
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    want = [c.detach().clone() for c in emb.tt_cores]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):      # warm-up on a side stream, as torch.cuda.graphs asks for
        step()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    for c, s0 in zip(emb.tt_cores, start):
        c.data.copy_(s0)
    for _ in range(4):
        graph.replay()
    torch.cuda.synchronize()
    for c, w in zip(emb.tt_cores, want):
        assert torch.isfinite(c).all()
        assert (c.detach() - w).abs().max().item() <= 1e-5 + 1e-4 * w.abs().max().item()
    for _ in range(100):               # back-to-back replays (this is what exposed the memset-node race)
        graph.replay()
    torch.cuda.synchronize()
    assert all(torch.isfinite(c).all() for c in emb.tt_cores)


def test_roctx_ranges_do_not_disturb_a_call(tmp_path):
    """TTEMB_ROCTX=1: the entry points bracket themselves with roctx ranges (marker library looked up at run time); a
    forward + backward in a fresh process gives the same rows as without the variable."""
    import os
    import subprocess
    import sys
    code = (
        "import sys, numpy as np, torch\n"
        f"sys.path[:0] = [{repr(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))}, "
        f"{repr(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'falcon-ttdforgnns_amd'))}]\n"
        "from FBTT.tt_embeddings_ops import TTEmbeddingBag\n"
        "torch.manual_seed(0)\n"
        "emb = TTEmbeddingBag(2449029, 100, [16, 16], [125, 140, 140], [4, 5, 5], sparse=True, use_cache=False, weight_dist='normal')\n"
        "ids = torch.arange(0, 20000, dtype=torch.int64, device='cuda') * 97\n"
        "out = emb(ids, torch.arange(20001, device='cuda'))\n"
        "out.backward(torch.ones_like(out) * 1e-3)\n"
        "torch.cuda.synchronize()\n"
        "print(float(out.double().sum().item()))\n")
    sums = []
    for flag in ("0", "1"):
        env = dict(os.environ, TTEMB_ROCTX=flag)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        sums.append(float(r.stdout.strip().splitlines()[-1]))
    assert sums[0] == sums[1]


@pytest.mark.parametrize("sparse", [True, False])
def test_module_call_in_pieces_matches_the_uncut_call(sparse):
    """The drop-in class on a call that is cut into pieces (diagnostic limits standing in for 2^24 rows / 2 GiB): same
    rows, and the same cores after the in-backward update (sparse) / the same .grad (dense), as the uncut call."""
    import ttemb_native as nat
    from FBTT.tt_embeddings_ops import TTEmbeddingBag
    torch.manual_seed(9)
    p, q, r = [125, 140, 140], [4, 5, 5], [16, 16]
    mk = lambda: TTEmbeddingBag(2449029, 100, r, p, q, sparse=sparse, use_cache=False, weight_dist="normal", learning_rate=0.1)
    a, b = mk(), mk()
    for ca, cb in zip(a.tt_cores, b.tt_cores):
        ca.data.mul_(300.0)
        cb.data.copy_(ca.data)
    rng = np.random.default_rng(9)
    n = 30000
    ids = torch.tensor(rng.integers(0, 2449029, size=n)).cuda()
    cuts = np.sort(rng.integers(0, n + 1, size=n - 1))          # ragged bags, some empty
    offs = torch.tensor(np.concatenate([[0], cuts, [n]])).cuda()
    d = (torch.rand(n, 100, device="cuda") - 0.5) * 0.05
    out_a = a(ids, offs)
    out_a.backward(d)
    nat.set_piece_limits(5000, 7000)
    try:
        out_b = b(ids, offs)
        out_b.backward(d)
        torch.cuda.synchronize()
    finally:
        nat.set_piece_limits(0, 0)
    torch.testing.assert_close(out_b, out_a, rtol=1e-5, atol=1e-6)
    for ca, cb in zip(a.tt_cores, b.tt_cores):
        if sparse:
            torch.testing.assert_close(cb.data, ca.data, rtol=1e-4, atol=1e-6)
        else:
            torch.testing.assert_close(cb.grad, ca.grad, rtol=1e-4, atol=1e-4 * float(ca.grad.abs().max()))


def test_bench_two_ranks_as_a_fresh_child_process():
    """`python bench.py --gpus 2` the way the driver's scaling run starts it -- a fresh process that launches its own ranks
    through torch.distributed.run, rendezvous on 127.0.0.1, TTDataParallel.step(overlap=True) every step -- rehearsed on the
    one GPU of this box (TTEMB_BENCH_REHEARSAL=1: both ranks on GPU 0, gloo carries the all-reduce; RCCL needs a GPU per
    rank).  Asserts the JSON line: two ranks, a per-rank time each, the weak-scaling value."""
    import json
    import subprocess
    env = dict(os.environ, TTEMB_BENCH_REHEARSAL="1", OMP_NUM_THREADS="4")
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--no-extras", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["dist"]["world_size"] == 2 and res["dist"]["backend"] == "gloo"
    assert res["dist"]["rehearsal_on_one_gpu"] is True and len(res["dist"]["ms_per_step_by_rank"]) == 2
    assert res["scaling"] == "weak" and res["config"]["parallelism"] == "dp2" and res["steps"] == 3
    assert res["value"] > 0 and res["ms_per_step_gpu_events"] > 0 and res["roofline"]["kernel"].startswith("fast3_")
