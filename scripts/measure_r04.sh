#!/bin/bash
# Round-4 measurement pass on the GPU box: the default bench line, every mode's JSON line, rocprofv3 kernel stats of the modes the
# round touched, the single-video (one video per call) kernel timeline and the BPTT phase stamps.
# usage (through gpurun): bash scripts/measure_r04.sh <tag>     -> gpurun_out/<tag>/
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-r04_meas}; OUT=gpurun_out/$TAG; mkdir -p $OUT
run() { name=$1; shift; timeout 300 python3 bench.py "$@" 2>/dev/null | tail -1 > $OUT/$name.json; }
timeout 500 python3 bench.py 2>$OUT/headline.err | tail -1 > $OUT/headline.json
timeout 500 python3 bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/headline_driver_flags.json      # what the driver runs
run vasnet_train --no-cpu-baseline --headline-only --mode train
run vasnet_train_bf16 --no-cpu-baseline --headline-only --mode train --precision bf16
run vasnet_score_bf16x6 --no-cpu-baseline --headline-only --precision bf16x6
run vasnet_score_bf16x3 --no-cpu-baseline --headline-only --precision bf16x3
run dsn_score --no-cpu-baseline --model dsn
run dsn_train --no-cpu-baseline --model dsn --mode train
run dsn_reinforce --no-cpu-baseline --model dsn --mode reinforce
run slstm_score --no-cpu-baseline --model slstm --steps 20 --warmup 5
run slstm_train --no-cpu-baseline --model slstm --mode train --steps 10 --warmup 3
run transformer_score --no-cpu-baseline --model transformer --steps 50 --warmup 10
run stress --no-cpu-baseline --workload stress --steps 10 --warmup 3
run vasnet_stream --no-cpu-baseline --mode stream --steps 100 --warmup 10
for m in "vasnet_score" "vasnet_train --mode train" "vasnet_train_bf16 --mode train --precision bf16" "dsn_score --model dsn" "dsn_train --model dsn --mode train" "dsn_reinforce --model dsn --mode reinforce"; do
  set -- $m; name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -o p -- python3 bench.py --no-cpu-baseline --headline-only --steps 50 --warmup 10 "$@" > $OUT/prof_$name.log 2>&1
  grep '^{' $OUT/prof_$name.log | tail -1 > $OUT/${name}_profiled_bench_line.json
  cp $OUT/prof_$name/*/p_kernel_stats.csv $OUT/${name}_kernel_stats.csv 2>/dev/null || cp $OUT/prof_$name/p_kernel_stats.csv $OUT/${name}_kernel_stats.csv
  rm -rf $OUT/prof_$name
done
# one video per call: kernel trace of the probe (score x 220, train step x 110; VASNet then DSN) -> per-kernel stats + one step's timeline
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_single -o p -- python3 scripts/single_video_probe.py > $OUT/single_video_probe.log 2>&1
T=$(ls $OUT/prof_single/*/p_kernel_trace.csv $OUT/prof_single/p_kernel_trace.csv 2>/dev/null | head -1)
python3 - "$T" > $OUT/single_video_timeline.txt 2>&1 <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"^void ", "", n); n = re.sub(r"\(.*", "", n)
    return n.replace("sumk::", "")[:70]
names = [short(r["Kernel_Name"]) for r in rows]
# a scoring call = [QKV lean NT (grid 240 blocks)] ... [layernorm_kernel<true, 4, true>]; a training step ends with adam_kernel
def grid(i): return int(rows[i]["Grid_Size_X"]) // int(rows[i]["Workgroup_Size_X"])
starts = [i for i, n in enumerate(names) if n.startswith("gemm_lean_kernel<true, true, true>") and grid(i) == 240]
def show(a, b, title):
    t0 = int(rows[a]["Start_Timestamp"])
    print(title)
    for i in range(a, b):
        s, e = int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])
        print(f"{(s-t0)/1e3:8.1f} +{(e-s)/1e3:7.1f}  {names[i]}  grid={grid(i)}x{rows[i]['Workgroup_Size_X']}")
    print("period us", (int(rows[b]["Start_Timestamp"]) - t0) / 1e3, "\n")
show(starts[100], starts[101], "== VASNet, one video per call (T = 300, D = 1024): one scoring call")
tr = [i for i in starts if any(names[j].startswith("adam_kernel") for j in range(i, min(i + 60, len(names))))]
show(tr[50], tr[51], "== one eager training step (zero_grad + forward + MSE + backward + Adam)")
PY
cp $(dirname $T)/p_kernel_stats.csv $OUT/single_video_kernel_stats.csv
rm -rf $OUT/prof_single
python3 scripts/single_video_probe.py > $OUT/single_video_probe_unprofiled.log 2>&1
# BPTT / forward recurrence phase stamps (diagnostic build)
SUMK_LIB_PATH=$PWD/summarizer_amd/libsumk_diag.so SUMK_LSTM_STAMPS=1 python3 bench.py --model dsn --mode reinforce --no-cpu-baseline --headline-only --steps 3 --warmup 1 2>&1 | grep "stamps\]" | sort | uniq > $OUT/lstm_phase_stamps.txt
for f in $OUT/*.json; do echo "$(basename $f .json): $(python3 -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('frac'))" 2>/dev/null)"; done
