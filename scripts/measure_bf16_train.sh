# bf16 training step (BASELINE config 2): bench line + rocprofv3 kernel stats; $1 = output tag under gpurun_out/
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1; mkdir -p $OUT
timeout 500 python3 bench.py --no-cpu-baseline --headline-only --mode train --precision bf16 2>$OUT/line.err | tail -1 > $OUT/line.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python3 bench.py --no-cpu-baseline --headline-only --mode train --precision bf16 --steps 50 --warmup 10 > $OUT/prof.log 2>&1
cp $OUT/prof/*/p_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null || cp $OUT/prof/p_kernel_stats.csv $OUT/kernel_stats.csv
rm -rf $OUT/prof
python3 -c "
import json, csv
d = json.load(open('$OUT/line.json')); print(d['value'], d['ms_per_step'])
for r in list(csv.DictReader(open('$OUT/kernel_stats.csv')))[:24]:
    print(f\"{r['Name'][:100]:100s} {r['Calls']:>5s} {float(r['AverageNs'])/1e3:8.1f} us {r['Percentage']:>6s}%\")
"
