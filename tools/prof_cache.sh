#!/bin/bash
# usage (on the GPU box): tools/prof_cache.sh <tag>  -> per-kernel averages of tools/cache_bench.py
tag=${1:-prof_cache}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag -- python3 $R/tools/cache_bench.py > $R/gpurun_out/$tag.log 2>&1
cat $R/gpurun_out/$tag.log | grep -v amdgpu.ids
python3 - <<PY
import csv,glob
f = glob.glob('$R/gpurun_out/$tag/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if float(r['Percentage']) > 0.3:
        print(f"{r['Name'][:80]:80s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={r['Percentage']}")
PY
