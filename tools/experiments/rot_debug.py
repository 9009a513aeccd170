import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for p in (ROOT, os.path.join(ROOT, "falcon-ttdforgnns_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import ttemb_native as nat
nat.set_path(nat.PATH_FAST3)
for (p, q, R) in (([30, 20, 40], [4, 5, 5], [1, 32, 32, 1]), ([30, 20, 40], [4, 5, 5], [1, 16, 16, 1]), ([25, 30, 35], [4, 4, 8], [1, 32, 32, 1])):
    rng = np.random.default_rng(1)
    n = 30000
    shape = nat.make_shape(p, q, R)
    cores = [torch.tensor((rng.standard_normal((p[t], R[t] * q[t] * R[t + 1])) * 0.3).astype(np.float32)).cuda() for t in range(3)]
    ids = torch.tensor(rng.integers(0, int(np.prod(p)), size=n).astype(np.int64)).cuda()
    offs = torch.arange(n + 1, device="cuda")
    out = torch.empty(n, int(np.prod(q)), device="cuda")
    ws = nat.Workspace()
    for it in range(3):
        nat.forward(shape, cores, ids, None, offs, n, None, n, out, ws)
        torch.cuda.synchronize()
        msg = "ok"
        try:
            nat.status()
        except RuntimeError as e:
            msg = str(e)[:140]
        print(p, q, R[1], "iter", it, "nan" if bool(torch.isnan(out).any()) else "finite", msg, flush=True)
