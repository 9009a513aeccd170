#!/usr/bin/env python3
"""Host cost of one small step, layer by layer (2 048 ids: the GPU work is negligible next to the host's)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "falcon-ttdforgnns_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
import ttemb_native as nat
from FBTT.tt_embeddings_ops import TTEmbeddingBag
from ttemb_dist import TTDataParallel

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
rng = np.random.default_rng(0)
ids = torch.from_numpy(rng.choice(2449029, size=n, replace=False).astype(np.int64)).cuda()
offs = torch.arange(n + 1, device="cuda")
d = (torch.rand(n, 100, device="cuda") - 0.5) * 0.1


def timeit(fn, iters=300):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


emb = TTEmbeddingBag(2449029, 100, [16, 16], [125, 140, 140], [4, 5, 5], sparse=True, use_cache=False, weight_dist="normal",
                     learning_rate=0.01)
cores = nat.core_views(emb.tt_cores, 0)
out = torch.empty(n, 100, device="cuda")
ws = nat.Workspace()


def native_only():
    plan = nat.new_plan(emb._shape, n, ids.device)
    nat.forward(emb._shape, cores, ids, None, offs, n, None, n, out, ws, plan)
    nat.backward_sgd(emb._shape, cores, ids, None, n, None, n, d, 0.01, ws, plan, offs)


print(f"{n} ids")
print(f"  native calls only (ctypes)        : {timeit(native_only):6.1f} us/step")
print(f"  module + autograd, sparse         : {timeit(lambda: emb(ids, offs).backward(d)):6.1f} us/step")
emb._use_lean = False
print(f"  ... through the general bridge    : {timeit(lambda: emb(ids, offs).backward(d)):6.1f} us/step")
emb._use_lean = True
print(f"  module + autograd, sparse (again) : {timeit(lambda: emb(ids, offs).backward(d)):6.1f} us/step")
g = torch.cuda.CUDAGraph()
s_ids, s_d = ids.clone(), d.clone()
with torch.cuda.graph(g):
    emb(s_ids, offs).backward(s_d)
print(f"  the same step captured (graph replay): {timeit(g.replay):6.1f} us/step")
cap = emb.capture(n, n)
print(f"  emb.capture(): cap(ids).backward(d) : {timeit(lambda: cap(ids).backward(d)):6.1f} us/step")
emb2 = TTEmbeddingBag(2449029, 100, [16, 16], [125, 140, 140], [4, 5, 5], sparse=False, use_cache=False, weight_dist="normal",
                      learning_rate=0.01)
def dense():
    emb2(ids, offs).backward(d)
    for c in emb2.tt_cores:
        c.grad = None
print(f"  module + autograd, dense grads    : {timeit(dense):6.1f} us/step")
dp = TTDataParallel(emb2)
def dps(overlap):
    emb2(ids, offs).backward(d)
    dp.step(overlap=overlap)
print(f"  + TTDataParallel.step()           : {timeit(lambda: dps(False)):6.1f} us/step")
print(f"  + TTDataParallel.step(overlap)    : {timeit(lambda: dps(True)):6.1f} us/step")
dp.flush()
