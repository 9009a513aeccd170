#!/usr/bin/env python3
"""Where does the grouped MFMA path start to pay?  Whole forward + dense backward (shared plan) per batch size, for the
kernel families a shape has: usage  crossover.py [kbench cfg ...] [--n N ...] [--paths generic per_bag fast3]."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "falcon-ttdforgnns_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import numpy as np, torch
import ttemb_native as nat
from kbench import CFG

ap = argparse.ArgumentParser()
ap.add_argument("cfg", nargs="*", default=["products"])
ap.add_argument("--n", type=int, nargs="+", default=[4096, 8192, 12288, 16384, 24576, 35000, 50000])
ap.add_argument("--paths", nargs="+", default=["generic", "fast3"])
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
PATH = {"generic": nat.PATH_GENERIC, "fast3": nat.PATH_FAST3, "per_bag": nat.PATH_PER_BAG, "auto": nat.PATH_AUTO}
for cfg in a.cfg:
    p, q, R, n_emb = CFG[cfg]
    D = int(np.prod(q))
    shape = nat.make_shape(p, q, R)
    rng = np.random.default_rng(0)
    cores = [torch.tensor((rng.standard_normal((p[t], R[t] * q[t] * R[t + 1])) * 0.3).astype(np.float32)).cuda() for t in range(len(p))]
    ws = nat.Workspace()
    for N in a.n:
        idx = torch.tensor(rng.choice(n_emb, size=N, replace=False).astype(np.int64)).cuda()
        offs = torch.arange(N + 1, device="cuda")
        out = torch.empty(N, D, device="cuda")
        d_out = (torch.rand(N, D, device="cuda") - 0.5) * 0.1
        grads = [torch.empty_like(c) for c in cores]
        res = {}
        for name in a.paths:
            nat.set_path(PATH[name])
            def step():
                plan = nat.new_plan(shape, N, idx.device)
                nat.forward(shape, cores, idx, None, offs, N, None, N, out, ws, plan)
                nat.backward_dense(shape, cores, idx, None, N, None, N, d_out, grads, ws, plan, offs)
            for _ in range(3): step()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters): step()
            e1.record(); torch.cuda.synchronize()
            res[name] = e0.elapsed_time(e1) / a.iters * 1e3
        print(f"{cfg} N={N:6d}: " + "   ".join(f"{k} {v:9.1f} us" for k, v in res.items()), flush=True)
nat.set_path(nat.PATH_AUTO)
