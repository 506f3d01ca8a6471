# Round-6 second measurement pass (after the attention re-cut): the default bench line (compact and --notes), the driver's flags, rocprofv3 kernel stats of
# the legs the re-cut touches, those legs' own bench lines.  Outputs under gpurun_out/r06b (copied into profiles/ by hand as r06b_*).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06b; mkdir -p $OUT
if [ -z "$ONLY_PROF" ]; then
timeout 900 python3 bench.py 2>$OUT/default.err | tail -1 > $OUT/bench_default_line.json
timeout 900 python3 bench.py --notes 2>/dev/null | tail -1 > $OUT/bench_default_line_notes.json
timeout 600 python3 bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_driver_flags_line.json
fi
prof() {   # name, bench args...
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -o p -- python3 bench.py --no-cpu-baseline --headline-only --no-probe --steps 30 --warmup 5 "$@" > $OUT/prof_$name.log 2>&1
  cp $(ls $OUT/prof_$name/*/p_kernel_stats.csv $OUT/prof_$name/p_kernel_stats.csv 2>/dev/null | head -1) $OUT/${name}_kernel_stats.csv
  rm -rf $OUT/prof_$name
}
prof vasnet_score_bf16x6 --precision bf16x6
prof vasnet_score_bf16x3 --precision bf16x3
prof vasnet_score_folded_bf16x6 --precision bf16x6 --fold-vo
prof vasnet_train_bf16 --mode train --precision bf16
prof transformer_score_bf16x6 --model transformer --precision bf16x6
prof transformer_score_bf16x3 --model transformer --precision bf16x3
[ -n "$ONLY_PROF" ] || for args in "" "--precision bf16x6" "--precision bf16x3" "--precision bf16x6 --fold-vo" "--precision bf16x3 --fold-vo" "--mode train --precision bf16" "--mode train --precision bf16x6" "--model dsn --mode reinforce" "--model dsn --mode reinforce --precision bf16x6" "--model transformer --precision bf16x6" "--model transformer --precision bf16x3" "--workload stress --precision bf16x6 --steps 3 --warmup 1"; do
  timeout 300 python3 bench.py --no-cpu-baseline --headline-only --steps 30 --warmup 5 $args 2>/dev/null | tail -1 | grep '^{' >> $OUT/bench_lines.jsonl
done
wc -c $OUT/bench_default_line.json; ls $OUT
