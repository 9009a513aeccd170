#!/bin/bash
# usage (on the GPU box): tools/pmc_sq.sh <tag> [kbench args...]
# Counter-only rocprofv3 passes with SQ wave-state / instruction-mix counters over tools/kbench.py --iters 5;
# prints per-kernel averages (tools/pmc_summary.py) into gpurun_out/<tag>.txt
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/$tag
rm -rf $O; mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1
pass() { n=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $O/p$n -- python3 $R/tools/kbench.py --iters 5 $EXTRA > $O/p$n.log 2>&1 || echo "pass $n failed"; }
EXTRA="$*"
pass 1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
pass 2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS
pass 3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVES
pass 4 GRBM_GUI_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_COEXEC_CYCLES SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_WAIT_INST_ANY
python3 $R/tools/pmc_summary.py $O > $O.txt 2>&1
grep -c . $O.txt
