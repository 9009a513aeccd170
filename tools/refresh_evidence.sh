#!/bin/bash
# usage (on the GPU box): tools/refresh_evidence.sh <tag>
# Regenerates the evidence bench.py and profiles/README.md cite: PMC traffic + MFMA utilisation (counter-only passes),
# the rocprofv3 --kernel-trace --stats summary of a short bench.py run, and the default bench.py line.
# Everything lands in gpurun_out/<tag>/ (copy what should be judged into profiles/).
tag=${1:-evidence}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
bash $R/tools/traffic.sh profiles/r05_traffic.json > $O/traffic.log 2>&1 || exit 1
cp $R/profiles/r05_traffic.json $O/
bash $R/tools/mfma_util.sh profiles/r05_mfma_util.json > $O/mfma.log 2>&1 || exit 1
cp $R/profiles/r05_mfma_util.json $O/
# the papers100M shape on one GPU (BASELINE configs[4]'s table): the same two counter passes + a kernel-stats pass
# (--plan: the forward keeps its plan -- the training form, P table stored -- and the backward runs on it, as the module and bench.py do)
TRAFFIC_KBENCH_ARGS="--cfg papers --n 819200 --no-rowidx --plan" bash $R/tools/traffic.sh profiles/r05_papers_traffic.json > $O/papers_traffic.log 2>&1 || exit 1
TRAFFIC_KBENCH_ARGS="--cfg papers --n 819200 --no-rowidx --plan" bash $R/tools/mfma_util.sh profiles/r05_papers_mfma_util.json > $O/papers_mfma.log 2>&1 || exit 1
cp $R/profiles/r05_papers_traffic.json $R/profiles/r05_papers_mfma_util.json $O/
cp $(find $R/gpurun_out/pmc_mfma_t -name "*kernel_stats.csv" | head -1) $O/papers_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_profiled.json 2> $O/bench_profiled.err || exit 1
cp $O/prof/*/*kernel_stats.csv $O/bench_kernel_stats.csv
cd $R && python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
# per-kernel summaries of the wide-rank chain (ranks 64 / 256 of the rank sweep), uniform and METIS-like ids
bash $R/tools/prof_cfg.sh q554_r64 q554_r256 q448_r256 < /dev/null 2>&1 | grep -E "^==|uniform|calls" > $O/wide_kernels.txt || exit 1
cd $R && for c in q554_r64 q554_r256; do python3 tools/kbench.py --cfg $c --iters 5 --no-rowidx --dist windows 2>/dev/null | tail -1 >> $O/wide_kernels.txt; done
# RCCL under this code at the one world size a one-GPU lease allows
cd $R && python3 tools/rccl_selfcheck.py > $O/rccl_selfcheck.txt 2>&1 || exit 1
# the wide-rank backward's traffic at rank 256 (no E table since round 5)
TRAFFIC_KBENCH_ARGS="--cfg q554_r256 --no-rowidx --what bwd" bash $R/tools/traffic.sh profiles/r05_wide_traffic_q554_r256.json > $O/wide_traffic.log 2>&1 || exit 1
cp $R/profiles/r05_wide_traffic_q554_r256.json $O/
cat $O/bench_default.json
