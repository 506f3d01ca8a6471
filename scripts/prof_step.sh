#!/bin/bash
# rocprofv3 kernel stats of the headline scoring step under a set of environment variants.
# usage (through gpurun): bash scripts/prof_step.sh <tag> "<name>:<ENV=VAL ...>[:<bench.py args>]" ...
#   -> gpurun_out/<tag>/<name>_kernel_stats.csv (+ <name>_bench_line.json: the JSON line of the profiled run)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=$1; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT
for spec in "$@"; do
  IFS=: read -r name envs bargs <<< "$spec"
  ( export $envs; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -o p -- python3 bench.py --no-cpu-baseline --headline-only --steps 50 --warmup 10 $bargs > $OUT/prof_$name.log 2>&1 )
  grep '^{' $OUT/prof_$name.log | tail -1 > $OUT/${name}_bench_line.json
  cp $OUT/prof_$name/*/p_kernel_stats.csv $OUT/${name}_kernel_stats.csv 2>/dev/null || cp $OUT/prof_$name/p_kernel_stats.csv $OUT/${name}_kernel_stats.csv
  rm -rf $OUT/prof_$name
  echo "== $name"; python3 - "$OUT/${name}_kernel_stats.csv" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    n = re.sub(r"\(.*", "", r["Name"]).replace("void sumk::", "").replace("sumk::", "")[:70]
    print(f"{n:72s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:9.1f} us  {float(r['Percentage']):5.1f}%")
PY
done
