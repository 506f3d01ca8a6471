#!/bin/bash
# LDS bank-conflict / MFMA counters of the bf16x3 kernels (KC images read with ds_read_b128, [k][row] images read with
# ds_read_b64_tr_b16) on one training step.  One counter group per pass, no other trace domains.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pass$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --mode train --precision bf16x3 > $OUT.pass$i.log 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "gemm_f32_kernel" not in name: continue
        agg[name.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    c = {n: sum(v) / len(v) for n, v in agg[k].items()}
    conf = c.get("SQ_LDS_BANK_CONFLICT", 0.0); act = c.get("SQ_LDS_IDX_ACTIVE", 1.0)
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0 / max(1.0, c.get("GRBM_GUI_ACTIVE", 1.0) / 8.0)
    print(f"{k:80s} launches {len(next(iter(agg[k].values())))}  LDS conflict/active {conf/act:.4f}  MFMA busy {busy:.3f}")
PY
