#!/bin/bash
# usage (on the GPU box): [TRAFFIC_KBENCH_ARGS="--cfg q554_r256 --no-rowidx"] tools/traffic.sh <out.json>
# Two counter-only rocprofv3 passes (FETCH_SIZE, WRITE_SIZE) over tools/kbench.py --iters 5 and a JSON of HBM
# bytes per launch per kernel, corrected as MI355X_MICROARCH.md §HBM prescribes for gfx950
# (bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024).
out=$(cd $GRAFT_REPO_ROOT && realpath -m "$1")
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/tools/kbench.py --iters 5 $TRAFFIC_KBENCH_ARGS > $R/gpurun_out/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/tools/kbench.py --iters 5 $TRAFFIC_KBENCH_ARGS > $R/gpurun_out/pmc_write.log 2>&1 || exit 1
python3 - "$R" "$out" "$TRAFFIC_KBENCH_ARGS" <<'PY'
import csv, glob, json, sys
from collections import defaultdict
R, out = sys.argv[1], sys.argv[2]
def collect(d, counter):
    acc = defaultdict(list)
    for path in glob.glob(f"{R}/gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter and "ttemb" in r["Kernel_Name"]:
                name = r["Kernel_Name"].split("(")[0].replace("void ttemb::", "").replace("ttemb::", "")
                name = name if "gemm" in name else name.split("<")[0]   # the three GEMM instances stay apart
                acc[name].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}
f, w = collect("pmc_fetch", "FETCH_SIZE"), collect("pmc_write", "WRITE_SIZE")
N = 409600
alg = {"fast3_forward_kernel": N * 408, "fast3_bwd_chunk_kernel": N * 408} if not (len(sys.argv) > 3 and sys.argv[3]) else {}
res = {"_how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, counters only) around tools/kbench.py "
               "--iters 5 " + (sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] else "(products r16)") + "; KB per launch averaged over launches (409600 unique uniform ids unless --n says otherwise). "
               "bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for "
               "gfx950 (it counts 128-B requests as 64 B); WRITE_SIZE is taken as is. The doubling is calibrated for "
               "16-B/lane streams only, so read-side figures of dword gathers are upper bounds.",
       "kernels": {}}
for k in sorted(set(f) | set(w)):
    e = {"fetch_kb": round(f.get(k, 0.0), 1), "write_kb": round(w.get(k, 0.0), 1),
         "hbm_bytes": int((2 * f.get(k, 0.0) + w.get(k, 0.0)) * 1024)}
    if k in alg:
        e["algorithmic_bytes"] = alg[k]
    res["kernels"][k] = e
json.dump(res, open(out, "w"), indent=1)
for k, e in res["kernels"].items():
    print(f"{k:34s} fetch {e['fetch_kb']:10.1f} KB  write {e['write_kb']:10.1f} KB  -> {e['hbm_bytes']/1e6:8.1f} MB")
PY
