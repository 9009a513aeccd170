#!/usr/bin/env python3
"""Small-batch latency: one TTEmbeddingBag step (forward + backward + fused SGD) on 2 048 ids, launched eagerly and
replayed from a HIP graph (torch.cuda.CUDAGraph).  The library enqueues everything on the caller's stream and never
synchronises, so a step can be captured as is; with static input buffers the replay removes the host launch cost."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "falcon-ttdforgnns_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
from FBTT.tt_embeddings_ops import TTEmbeddingBag


def main(n=2048, iters=200):
    torch.manual_seed(0)
    emb = TTEmbeddingBag(2449029, 100, [16, 16], [125, 140, 140], [4, 5, 5], sparse=True, use_cache=False,
                         weight_dist="normal", learning_rate=0.01)
    rng = np.random.default_rng(0)
    ids = torch.from_numpy(rng.choice(2449029, size=n, replace=False).astype(np.int64)).cuda()
    offs = torch.arange(n + 1, device="cuda")
    d_out = (torch.rand(n, 100, device="cuda") - 0.5) * 0.1

    def step():
        emb(ids, offs).backward(d_out)

    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / iters * 1e6
    print(f"eager ok {eager:.1f} us", flush=True)

    ref = [c.detach().clone() for c in emb.tt_cores]
    step()
    torch.cuda.synchronize()
    want = [c.detach().clone() for c in emb.tt_cores]          # cores after one more eager step
    for c, r in zip(emb.tt_cores, ref):
        c.data.copy_(r)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    print("side-stream warm-up ok", flush=True)
    for c, r in zip(emb.tt_cores, ref):
        c.data.copy_(r)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    for c, r in zip(emb.tt_cores, ref):
        c.data.copy_(r)
    print("captured", flush=True)
    g.replay()
    torch.cuda.synchronize()
    print("first replay ok", flush=True)
    err = max(float((c.detach() - w).abs().max()) for c, w in zip(emb.tt_cores, want))
    t0 = time.perf_counter()
    for _ in range(iters):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / iters * 1e6
    print(f"{n} ids: eager {eager:.1f} us/step, graph replay {graph:.1f} us/step, replay vs eager step max diff {err:.2e}")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 2048)
