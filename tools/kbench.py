#!/usr/bin/env python3
"""Kernel micro-benchmark: times the forward / backward chain kernels alone (HIP events
inside the library) on the ogbn-products shape for a few frontier sizes."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "falcon-ttdforgnns_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
import ttemb_native as nat

CFG = {"products": ([125, 140, 140], [4, 5, 5], [1, 16, 16, 1], 2449029),
       "arxiv": ([56, 60, 51], [4, 4, 8], [1, 8, 8, 1], 169343),
       "papers": ([500, 560, 400], [8, 4, 4], [1, 32, 32, 1], 111059956),
       # the reference's run scripts: ogbn-arxiv / D = 128 at rank 16 on the products factorisation
       "arxiv_r16": ([125, 140, 140], [4, 4, 8], [1, 16, 16, 1], 2449029),
       "q844_r16": ([125, 140, 140], [8, 4, 4], [1, 16, 16, 1], 2449029),
       "products_r32": ([125, 140, 140], [4, 5, 5], [1, 32, 32, 1], 2449029),
       # 4-core shapes of the run scripts (generic kernels only)
       "arxiv_4core": ([50, 60, 60, 60], [2, 4, 4, 4], [1, 16, 16, 16, 1], 10800000),
       "products_4core": ([50, 60, 60, 60], [5, 5, 2, 2], [1, 16, 16, 16, 1], 10800000),
       "q2255_4core": ([50, 60, 60, 60], [2, 2, 5, 5], [1, 16, 16, 16, 1], 10800000)}
# the rank sweep of run_script.sh:250-288 on the products factorisation (use small --n for the ranks the generic kernels take)
for _r in (8, 32, 64, 128, 256):
    CFG[f"q554_r{_r}"] = ([125, 140, 140], [5, 5, 4], [1, _r, _r, 1], 2449029)
for _r in (64, 128, 256):
    CFG[f"q448_r{_r}"] = ([125, 140, 140], [4, 4, 8], [1, _r, _r, 1], 2449029)
CFG["q455_r8"] = ([125, 140, 140], [4, 5, 5], [1, 8, 8, 1], 2449029)
# shapes without a template: the run-time-shape per-bag kernels (ttemb_rt3.inc) against the scalar ones (--path generic)
CFG["products_r12"] = ([125, 140, 140], [4, 5, 5], [1, 12, 12, 1], 2449029)
CFG["products_r24"] = ([125, 140, 140], [4, 5, 5], [1, 24, 24, 1], 2449029)
CFG["q2510_r16"] = ([125, 140, 140], [2, 5, 10], [1, 16, 16, 1], 2449029)
CFG["arxiv_4core_r12"] = ([50, 60, 60, 60], [2, 4, 4, 4], [1, 12, 12, 12, 1], 10800000)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", default="products")
    ap.add_argument("--n", type=int, nargs="+", default=[409600])
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--dist", default="uniform", choices=["uniform", "windows", "arange", "grouped"])
    ap.add_argument("--path", default="auto", choices=["auto", "generic", "fast3", "per_bag"])
    ap.add_argument("--what", default="both", choices=["both", "fwd", "bwd"])
    ap.add_argument("--no-rowidx", action="store_true", help="ids + offsets only, as the module passes them (the per-bag kernels need this form)")
    ap.add_argument("--plan", action="store_true", help="the forward keeps a plan and the backward runs on it, as the module's training step does (default: neither call gets one -- the forward is then the inference form, the backward groups the ids itself)")
    ap.add_argument("--split", action="store_true", help="two-phase forward (ttemb_forward_group, then ttemb_forward_lookup): the grouping steps and the prefix products as launches of their own")
    a = ap.parse_args()
    p, q, R, n_emb = CFG[a.cfg]
    D = int(np.prod(q))
    nat.set_path({"auto": 0, "generic": 1, "fast3": 2, "per_bag": 3}[a.path])
    shape = nat.make_shape(p, q, R)
    rng = np.random.default_rng(0)
    cores = [torch.tensor((rng.standard_normal((p[t], R[t] * q[t] * R[t + 1])) * 0.3).astype(np.float32)).cuda()
             for t in range(len(p))]
    ws = nat.Workspace()
    nat.profile_enable(True)
    for N in a.n:
        if a.dist == "uniform":
            ids = rng.choice(n_emb, size=N, replace=False)
        elif a.dist == "arange":
            ids = np.arange(N) % n_emb
        elif a.dist == "grouped":   # uniform ids handed over in the order the grouping pass gives them: rows of the
            ids = rng.choice(n_emb, size=N, replace=False)   # [N, D] tensors are then visited (almost) sequentially
            i0, rem = ids // (p[1] * p[2]), ids % (p[1] * p[2])
            ids = ids[np.argsort((rem // p[2]) * p[0] + i0, kind="stable")]
        else:  # METIS-like: windows of 200 consecutive ids
            starts = rng.choice(n_emb // 200 - 1, size=(N + 199) // 200, replace=False) * 200
            ids = (starts[:, None] + np.arange(200)[None, :]).reshape(-1)[:N]
        idx = torch.tensor(ids.astype(np.int64)).cuda()
        offs = torch.arange(N + 1, device="cuda")
        rowidx = torch.empty(N, dtype=torch.int64, device="cuda")
        nat.preprocess(idx, offs, N, True, None, None, None, rowidx, None, None, ws)
        if a.no_rowidx:
            rowidx = None
        out = torch.empty(N, D, device="cuda")
        d_out = (torch.rand(N, D, device="cuda") - 0.5) * 0.1
        grads = [torch.empty_like(c) for c in cores]
        f, b, c, g, ep, fi = [], [], [], [], [], []
        for i in range(a.iters + 3):
            if a.what in ("both", "fwd"):
                if a.split:
                    plan = nat.new_plan(shape, N, idx.device)
                    nat.forward(shape, cores, idx, rowidx, offs, N, None, N, out, ws, plan, phase=1)
                    nat.forward(shape, cores, idx, rowidx, offs, N, None, N, out, ws, plan, phase=2)
                elif a.plan:
                    plan = nat.new_plan(shape, N, idx.device)
                    nat.forward(shape, cores, idx, rowidx, offs, N, None, N, out, ws, plan)
                else:
                    nat.forward(shape, cores, idx, rowidx, offs, N, None, N, out, ws)
                if i >= 3:
                    f.append(nat.profile_read(0))
                    try:
                        g.append(nat.profile_read(3))
                    except RuntimeError:   # no grouping pass on this path (per-bag / scalar kernels)
                        g.append(0.0)
            if a.what in ("both", "bwd"):
                nat.backward_dense(shape, cores, idx, rowidx, N, None, N, d_out, grads, ws, plan if (a.plan and a.what == "both") else None, offs)
                if i >= 3:
                    b.append(nat.profile_read(1))
                    c.append(nat.profile_read(2))
                    try:
                        ep.append(nat.profile_read(8))
                        fi.append(nat.profile_read(9))
                    except RuntimeError:   # (kernel families without these two kernels)
                        pass
        torch.cuda.synchronize()
        msg = f"{a.cfg} {a.dist} N={N}:"
        if f:
            msg += f" grouping {np.mean(g)*1e3:6.1f} us fwd {np.mean(f)*1e3:8.1f} us ({N/np.mean(f)/1e6:8.2f} G lookups/s)"
        if b:
            msg += f" bwd {np.mean(b)*1e3:8.1f} us ({N/np.mean(b)/1e6:8.2f} G lookups/s) [chunk kernel {np.mean(c)*1e3:7.1f} us]"
            if ep:
                msg += f" [epilogue {np.mean(ep)*1e3:5.1f} finalize {np.mean(fi)*1e3:5.1f} us]"
        print(msg, flush=True)


if __name__ == "__main__":
    main()
