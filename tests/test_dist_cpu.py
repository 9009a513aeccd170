"""CPU, world_size 2 over gloo: the data-parallel step's collective logic
(ttemb_dist.TTDataParallel).  The optimiser epilogue is injected (plain torch) because
the product's epilogue is a HIP kernel and there is no GPU here."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    import sys
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from FBTT.tt_embeddings_ops import TTEmbeddingBag
    from ttemb_dist import TTDataParallel
    torch.manual_seed(100 + rank)  # deliberately different replicas before the broadcast
    m = TTEmbeddingBag(1000, 16, [4, 4], [10, 10, 10], [2, 2, 4], sparse=False, use_cache=False,
                       weight_dist="normal", learning_rate=0.5)
    dp = TTDataParallel(m, apply_fn=lambda w, g, lr: w.sub_(lr * g.view_as(w)))
    assert dp._flat_ok() and all(c.data.data_ptr() == v.data_ptr() for c, v in zip(m.tt_cores, dp.weight_views))
    dp.broadcast_parameters(0)
    start = [c.detach().clone() for c in m.tt_cores]
    g = torch.Generator().manual_seed(7 + rank)
    grads = [torch.randn(c.shape, generator=g) for c in m.tt_cores]
    for c, gr in zip(m.tt_cores, grads):
        c.grad = gr.clone()
    dp.step()
    assert all(c.grad is None for c in m.tt_cores)
    end = [c.detach().clone() for c in m.tt_cores]
    # the same step again, deferred: nothing changes until flush() (which the next forward calls before it reads
    # the cores), then the result is the same update
    for c, gr in zip(m.tt_cores, grads):
        c.grad = gr.clone()
    dp.step(overlap=True)
    assert m._before_weights is not None and all(c.grad is None for c in m.tt_cores)
    assert all(torch.equal(c.detach(), e) for c, e in zip(m.tt_cores, end))
    m._before_weights()
    assert m._before_weights is None
    end2 = [c.detach().clone() for c in m.tt_cores]
    dp.flush()   # idempotent
    assert all(torch.equal(c.detach(), e) for c, e in zip(m.tt_cores, end2))
    torch.save({"start": start, "grads": grads, "end": end, "end2": end2}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("world", [2, 8])   # 8: the size of the node the scaling bench runs on (BASELINE configs[3] / [4])
def test_dp_step(tmp_path, world):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(tmp_path, f"rank{k}.pt")) for k in range(world)]
    for t in range(3):
        for k in range(1, world):
            assert torch.equal(r[0]["start"][t], r[k]["start"][t])  # broadcast made replicas identical
        mean_g = sum(r[k]["grads"][t] for k in range(world)) / world
        want = r[0]["start"][t] - 0.5 * mean_g
        for k in range(world):
            np.testing.assert_allclose(r[k]["end"][t].numpy(), want.numpy(), rtol=1e-5, atol=1e-6)
            assert torch.equal(r[0]["end"][t], r[k]["end"][t])  # replicas stay bit-identical
        want2 = r[0]["end"][t] - 0.5 * mean_g                # the deferred (overlap=True) second step
        for k in range(world):
            np.testing.assert_allclose(r[k]["end2"][t].numpy(), want2.numpy(), rtol=1e-5, atol=1e-6)
            assert torch.equal(r[0]["end2"][t], r[k]["end2"][t])


def test_flat_bucket_layout():
    import sys
    sys.path.insert(0, PKG)
    from ttemb_dist import FlatGradBucket
    ps = [torch.nn.Parameter(torch.zeros(1, 5, 7)), torch.nn.Parameter(torch.zeros(1, 3, 2))]
    b = FlatGradBucket(ps)
    assert b.offsets == [0, 36] and b.n_grad == 36 + 8 and b.flat.numel() == 36 + 8 + 4   # gradients, then the fault count
    assert b.fault.data_ptr() == b.flat[44:].data_ptr() and float(b.fault) == 0.0
    ps[0].grad = torch.ones_like(ps[0])
    b.pack()
    assert b.flat[:35].eq(1).all() and b.flat[35:].eq(0).all()
    assert b.views[1].shape == ps[1].shape
