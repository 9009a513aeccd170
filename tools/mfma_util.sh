#!/bin/bash
# usage (on the GPU box): [TRAFFIC_KBENCH_ARGS="--cfg papers --n 819200 --no-rowidx"] tools/mfma_util.sh <out.json>
# One counter-only rocprofv3 pass (SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE, SQ_WAVES) over
# tools/kbench.py --iters 5, plus a kernel-trace pass for the durations; MFMA-pipe utilisation of a kernel =
# MFMA busy cycles summed over the 1024 SIMDs / (duration x 2.4 GHz x 1024).
out=$(cd $GRAFT_REPO_ROOT && realpath -m "$1")
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_mfma $R/gpurun_out/pmc_mfma_t
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $R/gpurun_out/pmc_mfma -- python3 $R/tools/kbench.py --iters 5 $TRAFFIC_KBENCH_ARGS > $R/gpurun_out/pmc_mfma.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pmc_mfma_t -- python3 $R/tools/kbench.py --iters 5 $TRAFFIC_KBENCH_ARGS > $R/gpurun_out/pmc_mfma_t.log 2>&1 || exit 1
python3 - "$R" "$out" "$TRAFFIC_KBENCH_ARGS" <<'PY'
import csv, glob, json, sys
from collections import defaultdict
R, out = sys.argv[1], sys.argv[2]
what = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] else "(products r16, 409600 unique uniform ids)"
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(f"{R}/gpurun_out/pmc_mfma/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "ttemb" in r["Kernel_Name"]:
            name = r["Kernel_Name"].split("(")[0].replace("void ttemb::", "").replace("ttemb::", "").split("<")[0]
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = {}
for path in glob.glob(f"{R}/gpurun_out/pmc_mfma_t/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "ttemb" in r["Name"]:
            dur[r["Name"].split("(")[0].replace("void ttemb::", "").replace("ttemb::", "").split("<")[0]] = float(r["AverageNs"]) / 1e3
res = {"_how": "tools/mfma_util.sh: rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES (counters only) and a "
               "separate --kernel-trace --stats pass over tools/kbench.py --iters 5 " + what + ". "
               "mfma_util = MFMA busy cycles (summed over SIMDs) / (duration x 2.4 GHz x 1024 SIMDs).", "kernels": {}}
for k, cs in sorted(acc.items()):
    e = {c: round(sum(v) / len(v), 1) for c, v in cs.items()}
    if k in dur:
        e["avg_us"] = round(dur[k], 1)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e:
            e["mfma_util"] = round(e["SQ_VALU_MFMA_BUSY_CYCLES"] / (dur[k] * 1e-6 * 2.4e9 * 1024), 4)
    res["kernels"][k] = e
json.dump(res, open(out, "w"), indent=1)
for k, e in res["kernels"].items():
    if "mfma_util" in e and e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) > 0:
        print(f"{k:32s} {e['avg_us']:7.1f} us  MFMA busy {e['SQ_VALU_MFMA_BUSY_CYCLES']:.3g}  util {e['mfma_util']:.3f}  GUI_ACTIVE {e.get('GRBM_GUI_ACTIVE')}")
PY
