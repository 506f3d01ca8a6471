# A/B of the 128-query context blocks (attn_pw.hip, QT = 4): product library vs variant libraries / SUMK_ATTN_WIDE=0, plus kernel stats of the split legs
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ctxw; O=gpurun_out/ctxw
timeout 900 python -m pytest tests/test_gpu_planes.py tests/test_gpu_fuzz_planes.py tests/test_gpu_poison.py -x -q 2>&1 | tail -3
run() { timeout 200 python3 bench.py --no-cpu-baseline --headline-only --steps 60 --warmup 10 "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
for p in bf16x6 bf16x3; do
  echo "== $p strips64"; SUMK_ATTN_WIDE=0 run --precision $p
  for lib in $(ls summarizer_amd/libsumk_*.so | grep -v diag); do echo "== $p $lib"; SUMK_LIB_PATH=$PWD/$lib run --precision $p; done
  echo "== $p product"; run --precision $p
  echo "== $p product folded"; run --precision $p --fold-vo
done
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for p in bf16x6 bf16x3; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python3 bench.py --no-cpu-baseline --headline-only --steps 30 --warmup 5 --precision $p > $O/prof.log 2>&1
cp $O/prof/*/p_kernel_stats.csv $O/${p}_kernel_stats.csv 2>/dev/null || cp $O/prof/p_kernel_stats.csv $O/${p}_kernel_stats.csv; rm -rf $O/prof
head -6 $O/${p}_kernel_stats.csv | cut -d, -f1-4 | cut -c1-140
done
