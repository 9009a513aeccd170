#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel, per counter average."""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if "ttemb" not in name:
            continue
        short = name.split("(")[0].replace("void ttemb::", "").replace("ttemb::", "")[:48]
        acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    print(k)
    for c, v in sorted(cs.items()):
        print(f"    {c:32s} n={len(v):3d} avg={sum(v)/len(v):16.1f}")
