#!/usr/bin/env python3
"""Randomised stress of the grouped MFMA path against the CPU oracle: every instantiated (q, rank) shape, random
table factorisations, batch sizes from 1 to 60 000, random bags (empty / long / duplicates), forward + dense
backward + fused SGD, plan reuse as the module does it.  Prints one line per failure and a summary."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "falcon-ttdforgnns_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
import ttemb_native as nat
from oracle import tt_oracle as orc

SHAPES = [(4, 5, 5, 16, 16), (4, 4, 8, 8, 8), (8, 4, 4, 32, 32), (4, 4, 8, 16, 16), (8, 4, 4, 16, 16), (4, 5, 5, 32, 32),
          (4, 4, 8, 32, 32), (5, 4, 5, 16, 16), (5, 5, 4, 16, 16)]
FOUR = [([2, 4, 4, 4], [1, 16, 16, 16, 1]), ([4, 2, 4, 4], [1, 16, 16, 16, 1]), ([2, 2, 5, 5], [1, 8, 16, 16, 1]),
        ([5, 5, 2, 2], [1, 16, 16, 16, 1]), ([5, 5, 2, 2], [1, 16, 16, 4, 1])]
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
nat.set_path(nat.PATH_FAST3)
shared_ws = nat.Workspace()   # half of the cases share ONE workspace: tables of every shape take turns on its header words
t_end = time.time() + seconds
t_note = time.time() + 30.0
n_cases = n_bad = 0
case = seed0
while time.time() < t_end:
    case += 1
    rng = np.random.default_rng(case)
    sh = SHAPES[int(rng.integers(0, len(SHAPES)))]
    q, R = list(sh[:3]), [1, sh[3], sh[4], 1]
    p = [int(rng.integers(1, 80)), int(rng.integers(1, 80)), int(rng.integers(1, 400))]
    if rng.random() < 0.25:   # a 4-core table that maps onto the grouped path through a merged pair of cores
        q, R = FOUR[int(rng.integers(0, len(FOUR)))]
        p = [int(rng.integers(1, 30)), int(rng.integers(1, 30)), int(rng.integers(1, 40)), int(rng.integers(1, 40))]
        sh = tuple(q)
    n_emb = int(np.prod(p))
    nnz_target = int(rng.choice([1, 7, 200, 3000, 20000, 60000]))
    mode = int(rng.integers(0, 3))
    if mode == 0:
        lens = np.ones(nnz_target, dtype=np.int64)
    elif mode == 1:
        lens = rng.integers(0, 4, size=nnz_target)
    else:
        lens = np.concatenate([[0, min(nnz_target, 5000), 0], rng.integers(0, 3, size=nnz_target)])
    lens = lens[np.cumsum(lens) <= nnz_target]
    nnz = int(lens.sum())
    if nnz == 0:
        continue
    idx = rng.integers(0, n_emb, size=nnz).astype(np.int64)
    if rng.random() < 0.5 and nnz > 10:   # windows of consecutive ids (dense groups) and duplicates
        w = int(rng.integers(2, 300))
        st = rng.integers(0, max(1, n_emb - w), size=nnz // w + 1)
        idx = (st[:, None] + np.arange(w)[None, :]).reshape(-1)[:nnz].astype(np.int64)
        idx = np.minimum(idx, n_emb - 1)
        idx[: nnz // 10] = idx[nnz // 10: 2 * (nnz // 10)]
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    B, D = offsets.shape[0] - 1, int(np.prod(q))
    cores = [(rng.standard_normal((p[t], R[t] * q[t] * R[t + 1])) * 0.3).astype(np.float32) for t in range(len(p))]
    d_out = ((rng.random((B, D)) - 0.5) * 0.2).astype(np.float32)
    shape = nat.make_shape(p, q, R)
    ws = shared_ws if rng.random() < 0.5 else nat.Workspace()
    # a third of the cases are cut into pieces (by rows, by ids or both), as a call past one row window would be
    cut = int(rng.integers(0, 3)) == 0
    nat.set_piece_limits(int(rng.choice([0, max(1, B // 7), max(1, B // 40)])) if cut else 0,
                         int(rng.choice([0, max(1, nnz // 9), max(1, nnz // 30)])) if cut else 0)   # at most ~70 pieces
    c = [torch.from_numpy(x).cuda() for x in cores]
    ti, to = torch.from_numpy(idx).cuda(), torch.from_numpy(offsets).cuda()
    out = torch.full((B, D), float("nan"), device="cuda")
    plan = nat.new_plan(shape, nnz, ti.device)   # (None for a call in pieces)
    nat.forward(shape, c, ti, None, to, nnz, None, B, out, ws, plan)
    grads = [torch.full_like(x, float("nan")) for x in c]
    nat.backward_dense(shape, c, ti, None, nnz, None, B, torch.from_numpy(d_out).cuda(), grads, ws, plan, to)
    c2 = [x.clone() for x in c]
    nat.backward_sgd(shape, c2, ti, None, nnz, None, B, torch.from_numpy(d_out).cuda(), 0.05, ws, plan, to)
    torch.cuda.synchronize()
    want = orc.tt_forward(idx, offsets, cores, p, q, R)
    wg = orc.tt_dense_backward(idx, offsets, d_out, cores, p, q, R)
    ok = np.allclose(out.cpu().numpy(), want, rtol=1e-4, atol=1e-4 * max(1.0, float(np.abs(want).max())))
    for t in range(len(p)):
        sc = max(float(np.abs(wg[t]).max()), 1e-6)
        ok &= float(np.abs(grads[t].cpu().numpy() - wg[t]).max()) <= 3e-4 * sc + 1e-6
        ok &= float(np.abs(c2[t].cpu().numpy() - (cores[t] - np.float32(0.05) * wg[t])).max()) <= 0.05 * (3e-4 * sc) + 2e-6
    n_cases += 1
    if time.time() > t_note:   # a line every half minute: a silent GPU job looks hung to the runner
        print(f"... {n_cases} cases, {n_bad} failures so far", flush=True)
        t_note = time.time() + 30.0
    if not ok:
        n_bad += 1
        print(f"FAIL case {case}: shape {sh} p {p} nnz {nnz} B {B} mode {mode} cut {cut}", flush=True)
print(f"{n_cases} cases, {n_bad} failures")
sys.exit(1 if n_bad else 0)
