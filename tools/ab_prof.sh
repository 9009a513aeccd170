#!/bin/bash
# usage (on the GPU box): tools/ab_prof.sh <kernel-name-regex> [kbench args]  -- per-kernel averages per lib/libttemb_*.so
pat=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lib in $R/falcon-ttdforgnns_amd/lib/libttemb_*.so; do
  tag=abp_$(basename $lib .so)
  rm -rf $R/gpurun_out/$tag
  TTEMB_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag -- python3 $R/tools/kbench.py "$@" > $R/gpurun_out/$tag.log 2>&1
  echo "== $(basename $lib)"
  python3 - <<PY
import csv,glob,re
f = sorted(glob.glob('$R/gpurun_out/$tag/*/*kernel_stats.csv'))[-1]
for r in csv.DictReader(open(f)):
    if re.search(r'$pat', r['Name']):
        print(f"   {r['Name'][:60]:60s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:9.1f}")
PY
done
