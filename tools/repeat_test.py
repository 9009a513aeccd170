#!/usr/bin/env python3
"""usage (GPU box): tools/repeat_test.py <test function in tests/test_gpu_module.py> [times]
Calls one module-level GPU test repeatedly in ONE process and prints every failure (to chase an intermittent assertion)."""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "falcon-ttdforgnns_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import FBTT.tt_embeddings_ops as ops   # noqa: E402
import test_gpu_module as tm           # noqa: E402

name, times = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 10
bad = 0
for i in range(times):
    try:
        getattr(tm, name)(ops)
    except Exception:
        bad += 1
        print(f"--- run {i} failed")
        traceback.print_exc(limit=3)
        sys.stdout.flush()
print(f"{name}: {bad} of {times} runs failed")
