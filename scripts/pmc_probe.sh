#!/bin/bash
# PMC passes over a probe script (separate runs per counter group; kernel-trace only).  usage: scripts/pmc_probe.sh <outdir-under-gpurun_out> <script> [args]
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
export PYTHONPATH=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pass$i -- python3 $GRAFT_REPO_ROOT/"$@" > $OUT.pass$i.log 2>&1
done
ls $OUT
