#!/bin/bash
# usage (GPU box): tools/prof_cfg.sh <kbench cfg>...   -> gpurun_out/prof_<cfg>/ + a per-kernel summary on stdout
# rocprofv3 --kernel-trace --stats over tools/kbench.py for each configuration (ids + offsets only, 5 timed iterations).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for c in "$@"; do
  rm -rf $R/gpurun_out/prof_$c
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$c -- python3 $R/tools/kbench.py --cfg $c --iters 5 --no-rowidx > $R/gpurun_out/prof_$c.log 2>&1 || exit 1
  f=$(find $R/gpurun_out/prof_$c -name "*kernel_stats.csv" | head -1)
  echo "== $c"
  grep -v amdgpu.ids $R/gpurun_out/prof_$c.log
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:9.1f} us  {float(r["Percentage"]):5.1f} %')
PY
done
