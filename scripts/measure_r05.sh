# Round-5 measurement pass (run on the GPU box from the repo root): the default bench line and the driver's flags, rocprofv3 kernel stats
# of every scoring / training mode, every mode's own bench line.  Outputs under gpurun_out/r05_final (copied into profiles/ by hand).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05_final; mkdir -p $OUT
timeout 600 python3 bench.py 2>$OUT/default.err | tail -1 > $OUT/bench_default_line.json
timeout 600 python3 bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_driver_flags_line.json
prof() {   # name, bench args...
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -o p -- python3 bench.py --no-cpu-baseline --headline-only --steps 30 --warmup 5 "$@" > $OUT/prof_$name.log 2>&1
  cp $OUT/prof_$name/*/p_kernel_stats.csv $OUT/${name}_kernel_stats.csv 2>/dev/null || cp $OUT/prof_$name/p_kernel_stats.csv $OUT/${name}_kernel_stats.csv
  tail -1 $OUT/prof_$name.log | grep '^{' >> $OUT/bench_lines_profiled.jsonl
  rm -rf $OUT/prof_$name
}
prof vasnet_score
prof vasnet_score_bf16x6 --precision bf16x6
prof vasnet_score_bf16x3 --precision bf16x3
prof vasnet_score_folded_bf16x6 --precision bf16x6 --fold-vo
prof vasnet_score_folded_bf16x3 --precision bf16x3 --fold-vo
prof transformer_score --model transformer
prof transformer_score_bf16x6 --model transformer --precision bf16x6
prof transformer_score_bf16x3 --model transformer --precision bf16x3
prof vasnet_train --mode train
prof vasnet_train_bf16 --mode train --precision bf16
prof dsn_score --model dsn
prof dsn_score_bf16x6 --model dsn --precision bf16x6
prof dsn_train --model dsn --mode train
prof dsn_reinforce --model dsn --mode reinforce
prof slstm_score --model slstm
prof slstm_score_bf16x6 --model slstm --precision bf16x6
for args in "" "--precision bf16x6" "--precision bf16x3" "--mode train" "--mode train --precision bf16" "--model dsn" "--model dsn --precision bf16x6" "--model dsn --mode train" "--model dsn --mode reinforce" "--model slstm" "--model slstm --precision bf16x6" "--model transformer" "--model transformer --precision bf16x6" "--model transformer --precision bf16x3" "--precision bf16x6 --fold-vo" "--precision bf16x3 --fold-vo" "--workload stress" "--workload stress --precision bf16x6" "--mode stream"; do
  timeout 300 python3 bench.py --no-cpu-baseline --headline-only --steps 30 --warmup 5 $args 2>/dev/null | tail -1 | grep '^{' >> $OUT/bench_lines.jsonl
done
ls $OUT
