#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for n in 102400 204800 409600 819200; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/nsw_$n -- python3 $R/tools/kbench.py --iters 10 --n $n --what bwd > $R/gpurun_out/nsw_$n.log 2>&1
  python3 - <<PY
import csv,glob
f = sorted(glob.glob('$R/gpurun_out/nsw_$n/*/*kernel_stats.csv'))[-1]
d = {r['Name'].split('(')[0].replace('void ttemb::','').replace('ttemb::','').split('<')[0]: float(r['AverageNs'])/1e3 for r in csv.DictReader(open(f))}
print("N=$n", " ".join(f"{k.replace('fast3_','')}={v:.1f}" for k,v in d.items() if k.startswith('fast3') and ('chunk' in k or 'reduce' in k or 'epilogue' in k or 'finalize' in k)), f" reduce ns/row={d['fast3_dg2_reduce_kernel']*1e3/$n:.3f} chunk ns/row={d['fast3_bwd_chunk_kernel']*1e3/$n:.3f}")
PY
done
