#!/bin/bash
# usage (on the GPU box): tools/prof_kernels.sh <tag> [kbench args...]  -> prints per-kernel averages
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag -- python3 $R/tools/kbench.py "$@" > $R/gpurun_out/$tag.log 2>&1
python3 - <<PY
import csv,glob
f = glob.glob('$R/gpurun_out/$tag/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'ttemb' in r['Name'] or 'rocprim' in r['Name']:
        print(f"{r['Name'][:64]:64s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:9.1f}")
PY
