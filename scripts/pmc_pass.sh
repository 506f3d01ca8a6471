#!/bin/bash
# PMC passes for the bench step (separate runs per counter group; never combined with other trace domains).
# usage: scripts/pmc_pass.sh <outdir-under-gpurun_out> [bench args]      (the passes run `bench.py --headline-only`: one scoring leg)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_MFMA" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pass$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --headline-only "$@" > $OUT.pass$i.log 2>&1
done
ls $OUT
