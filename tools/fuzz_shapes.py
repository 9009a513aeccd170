#!/usr/bin/env python3
"""More seeds of the random-table parity tests (every grouped-path shape, narrow and wide ranks) than the suite runs:
usage (GPU box, repo root)  python tools/fuzz_shapes.py [first_seed last_seed]."""
import sys, os, types
sys.path.insert(0, "tests"); sys.path.insert(0, "."); sys.path.insert(0, "falcon-ttdforgnns_amd")
import numpy as np, torch
import conftest  # noqa
import test_gpu_parity as T
import ttemb_native as nat
from oracle import tt_oracle as orc
bad = 0
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2, 6)
for seed in range(lo, hi + 1):
    for shape in sorted(T.FAST3_SHAPES):
        try:
            T.test_random_tables_fast_path(nat, orc, shape, seed)
        except BaseException as e:
            if type(e).__name__ == "Skipped": continue
            bad += 1; print("FAIL fast", shape, seed, repr(e)[:200], flush=True)
    for shape in T.WIDE3_SHAPES:
        for form in ("e_table", "lds_slabs"):   # both backward forms of the wide-rank chain (the suite's fixture does the same)
            nat.set_wide_slab_min_ids(1 if form == "lds_slabs" else 1 << 40)
            try:
                T.test_random_tables_wide_rank_chain(nat, orc, shape, seed, form)
            except BaseException as e:
                bad += 1; print("FAIL wide", shape, seed, form, repr(e)[:200], flush=True)
            nat.set_wide_slab_min_ids(0)
    nat.set_path(nat.PATH_AUTO)
    print("seed", seed, "done", flush=True)
print("failures:", bad)
