cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04_final; mkdir -p $OUT
timeout 500 python3 bench.py 2>$OUT/headline.err | tail -1 > $OUT/headline.json
timeout 500 python3 bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/headline_driver_flags.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_vs -o p -- python3 bench.py --no-cpu-baseline --headline-only --steps 50 --warmup 10 > $OUT/prof_vs.log 2>&1
cp $OUT/prof_vs/*/p_kernel_stats.csv $OUT/vasnet_score_kernel_stats.csv 2>/dev/null || cp $OUT/prof_vs/p_kernel_stats.csv $OUT/vasnet_score_kernel_stats.csv
rm -rf $OUT/prof_vs
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_single -o p -- python3 scripts/single_video_probe.py > $OUT/single_video_probe.log 2>&1
T=$(ls $OUT/prof_single/*/p_kernel_trace.csv $OUT/prof_single/p_kernel_trace.csv 2>/dev/null | head -1)
python3 scripts/trace_timeline.py $T "gemm_lean_kernel<true, true, true>" 500 > $OUT/single_video_timeline_raw.txt 2>&1
cp $(dirname $T)/p_kernel_stats.csv $OUT/single_video_kernel_stats.csv
python3 - "$T" > $OUT/single_video_timeline.txt <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"^void ", "", n); n = re.sub(r"\(.*", "", n)
    return n.replace("sumk::", "")[:70]
names = [short(r["Kernel_Name"]) for r in rows]
# a scoring call = [QKV lean NT (grid 240 blocks)] ... [layernorm_kernel<true, 4>]; a training step ends with adam_kernel
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
rm -rf $OUT/prof_single
python3 scripts/single_video_probe.py > $OUT/single_video_probe_unprofiled.log 2>&1
cat $OUT/single_video_probe_unprofiled.log; head -14 $OUT/single_video_timeline.txt
python3 -c "
import json; d=json.load(open('$OUT/headline.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['parity_max_abs_diff_vs_port'], d['cpu_baseline']['value']); print(json.dumps(d['single_video_mode']['vasnet'])); print(d['trainer_test_mode']['ms_per_call'], d['stream_mode']['frames_per_s'])
d=json.load(open('$OUT/headline_driver_flags.json')); print('driver flags', d['value'], d['ms_per_step'], d['roofline']['frac'])"
