# A/B of the non-temporal K / Q loads of the logits launch (attn_pw.hip): default rule vs SUMK_ATTN_NT=0 / 1
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_planes.py tests/test_gpu_fuzz_planes.py tests/test_gpu_transformer.py tests/test_gpu_poison.py -x -q 2>&1 | tail -2
run() { timeout 200 python3 bench.py --no-cpu-baseline --headline-only --steps 60 --warmup 10 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
for a in "--precision bf16x6" "--precision bf16x3" "--precision bf16x6 --fold-vo" "--model transformer --precision bf16x6" "--model transformer --precision bf16x3"; do
  for nt in default 0 1; do echo "== $a nt=$nt"; if [ $nt = default ]; then run $a; else SUMK_ATTN_NT=$nt run $a; fi; done
done
