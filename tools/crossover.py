#!/usr/bin/env python3
"""Where does the grouped MFMA path start to pay?  fwd+bwd kernel time, generic vs fast3, per batch size."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "falcon-ttdforgnns_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
import ttemb_native as nat
p, q, R, n_emb = [125, 140, 140], [4, 5, 5], [1, 16, 16, 1], 2449029
shape = nat.make_shape(p, q, R)
rng = np.random.default_rng(0)
cores = [torch.tensor((rng.standard_normal((p[t], R[t] * q[t] * R[t + 1])) * 0.3).astype(np.float32)).cuda() for t in range(3)]
ws = nat.Workspace()
for N in (4096, 8192, 12288, 16384, 24576, 35000, 50000):
    idx = torch.tensor(rng.choice(n_emb, size=N, replace=False).astype(np.int64)).cuda()
    offs = torch.arange(N + 1, device="cuda")
    out = torch.empty(N, 100, device="cuda")
    d_out = (torch.rand(N, 100, device="cuda") - 0.5) * 0.1
    grads = [torch.empty_like(c) for c in cores]
    res = {}
    for name, path in (("generic", 1), ("fast3", 2)):
        nat.set_path(path)
        def step():
            plan = nat.new_plan(shape, N, idx.device)
            nat.forward(shape, cores, idx, None, offs, N, None, N, out, ws, plan)
            nat.backward_dense(shape, cores, idx, None, N, None, N, d_out, grads, ws, plan, offs)
        for _ in range(5): step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): step()
        e1.record(); torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 20 * 1e3
    print(f"N={N:6d}: generic {res['generic']:7.1f} us   fast3 {res['fast3']:7.1f} us", flush=True)
