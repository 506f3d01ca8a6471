#!/bin/bash
# Full measurement pass on the GPU box: every bench mode's JSON line + rocprofv3 kernel stats of the headline run.
# usage (through gpurun): bash scripts/measure_all.sh <tag>     -> gpurun_out/<tag>/
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-meas}; OUT=gpurun_out/$TAG; mkdir -p $OUT
run() { name=$1; shift; timeout 300 python3 bench.py "$@" 2>/dev/null | tail -1 > $OUT/$name.json; }
timeout 400 python3 bench.py 2>/dev/null | tail -1 > $OUT/headline.json
run vasnet_score_bf16x3 --no-cpu-baseline --precision bf16x3
run vasnet_train --no-cpu-baseline --mode train
run vasnet_train_bf16x3 --no-cpu-baseline --mode train --precision bf16x3
run vasnet_train_bf16 --no-cpu-baseline --mode train --precision bf16
run vasnet_score_bf16x6 --no-cpu-baseline --precision bf16x6
run slstm_train --no-cpu-baseline --model slstm --mode train --steps 10 --warmup 3
run dsn_score --no-cpu-baseline --model dsn
run dsn_score_bf16x3 --no-cpu-baseline --model dsn --precision bf16x3
run dsn_train --no-cpu-baseline --model dsn --mode train
run dsn_reinforce --no-cpu-baseline --model dsn --mode reinforce
run slstm_score --no-cpu-baseline --model slstm --steps 20 --warmup 5
run slstm_score_bf16x3 --no-cpu-baseline --model slstm --precision bf16x3 --steps 20 --warmup 5
run transformer_score --no-cpu-baseline --model transformer --steps 50 --warmup 10
run transformer_score_bf16x3 --no-cpu-baseline --model transformer --precision bf16x3 --steps 50 --warmup 10
run transformer_train --no-cpu-baseline --model transformer --mode train --steps 20 --warmup 5
run stress --no-cpu-baseline --workload stress --steps 10 --warmup 3
run stress_bf16x3 --no-cpu-baseline --workload stress --precision bf16x3 --steps 10 --warmup 3
run vasnet_stream --no-cpu-baseline --mode stream --steps 100 --warmup 10
run dsn_stream --no-cpu-baseline --mode stream --model dsn --steps 100 --warmup 10
run sumgan_train --model sumgan --mode train --steps 5 --warmup 1
for m in "vasnet_score" "dsn_score --model dsn" "slstm_score --model slstm --steps 10 --warmup 3" "vasnet_train --mode train" "vasnet_train_bf16 --mode train --precision bf16" "dsn_train --model dsn --mode train" "dsn_reinforce --model dsn --mode reinforce" "slstm_train --model slstm --mode train --steps 10 --warmup 3" "vasnet_score_bf16x6 --precision bf16x6"; do
  set -- $m; name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -o p -- python3 bench.py --no-cpu-baseline --headline-only --steps 50 --warmup 10 "$@" > $OUT/prof_$name.log 2>&1
  rm -f $OUT/prof_$name/*kernel_trace.csv $OUT/prof_$name/*.db
done
for f in $OUT/*.json; do echo "$(basename $f .json): $(python3 -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('frac'))")"; done
