# Round-6 measurement pass (run on the GPU box from the repo root): the default bench line (compact and --notes), the driver's flags, rocprofv3
# kernel stats of the modes this round touched + the headline, every mode's own bench line, the recurrence probes.  Outputs under
# gpurun_out/r06_final (copied into profiles/ by hand).  PMC passes: scripts/pmc_pass.sh (separate call).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=${OUT:-gpurun_out/r06_final}; mkdir -p $OUT
timeout 900 python3 bench.py 2>$OUT/default.err | tail -1 > $OUT/bench_default_line.json
timeout 900 python3 bench.py --notes 2>/dev/null | tail -1 > $OUT/bench_default_line_notes.json
timeout 600 python3 bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_driver_flags_line.json
prof() {   # name, bench args...
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -o p -- python3 bench.py --no-cpu-baseline --headline-only --no-probe --steps 30 --warmup 5 "$@" > $OUT/prof_$name.log 2>&1
  cp $OUT/prof_$name/*/p_kernel_stats.csv $OUT/${name}_kernel_stats.csv 2>/dev/null || cp $OUT/prof_$name/p_kernel_stats.csv $OUT/${name}_kernel_stats.csv
  tail -1 $OUT/prof_$name.log | grep '^{' >> $OUT/bench_lines_profiled.jsonl
  rm -rf $OUT/prof_$name
}
prof vasnet_score
prof vasnet_score_bf16x6 --precision bf16x6
prof vasnet_score_bf16x3 --precision bf16x3
prof vasnet_train_bf16 --mode train --precision bf16
prof dsn_score --model dsn
prof dsn_score_bf16x6 --model dsn --precision bf16x6
prof dsn_score_bf16x3 --model dsn --precision bf16x3
prof dsn_train --model dsn --mode train
prof dsn_reinforce --model dsn --mode reinforce
prof slstm_score --model slstm
prof slstm_score_bf16x6 --model slstm --precision bf16x6
prof slstm_score_bf16x3 --model slstm --precision bf16x3
prof slstm_train --model slstm --mode train --steps 5 --warmup 2
prof stress_bf16x6 --workload stress --precision bf16x6 --steps 3 --warmup 1
prof transformer_score_bf16x6 --model transformer --precision bf16x6
prof transformer_score_bf16x3 --model transformer --precision bf16x3
prof vasnet_score_folded_bf16x6 --precision bf16x6 --fold-vo
for args in "" "--precision bf16x6" "--precision bf16x3" "--mode train" "--mode train --precision bf16" "--model dsn" "--model dsn --precision bf16x6" "--model dsn --precision bf16x3" "--model dsn --mode train" "--model dsn --mode reinforce" "--model dsn --mode reinforce --precision bf16x6" "--model dsn --mode train --precision bf16x6" "--mode train --precision bf16x6" "--model slstm" "--model slstm --precision bf16x6" "--model slstm --precision bf16x3" "--model slstm --mode train --steps 5 --warmup 2" "--model transformer" "--model transformer --precision bf16x6" "--model transformer --precision bf16x3" "--precision bf16x6 --fold-vo" "--precision bf16x3 --fold-vo" "--workload stress --steps 3 --warmup 1" "--workload stress --precision bf16x6 --steps 3 --warmup 1" "--workload stress --precision bf16x3 --steps 3 --warmup 1" "--mode stream"; do
  timeout 300 python3 bench.py --no-cpu-baseline --headline-only --steps 30 --warmup 5 $args 2>/dev/null | tail -1 | grep '^{' >> $OUT/bench_lines.jsonl
done
python3 scripts/probes/wide2_probe.py > $OUT/wide2_probe.txt 2>&1
python3 scripts/probes/dsn_probe.py > $OUT/dsn_probe.txt 2>&1
python3 scripts/probes/dsn_train_probe.py > $OUT/dsn_train_probe.txt 2>&1
python3 scripts/probes/slstm_train_probe.py > $OUT/slstm_train_probe.txt 2>&1
SUMK_LIB_PATH=$PWD/summarizer_amd/libsumk_diag.so SUMK_LSTM_STAMPS=1 python3 scripts/probes/wide2_probe.py > $OUT/wide2_stamps.txt 2>&1
ls $OUT
