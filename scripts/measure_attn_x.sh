cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ax
for x in "$@"; do
  export SUMK_ATTN_X=$x
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ax/p$x -o p -- python3 bench.py --no-cpu-baseline --headline-only --mode train --precision bf16 --steps 30 --warmup 5 > gpurun_out/ax/log$x 2>&1
  f=$(ls gpurun_out/ax/p$x/*/p_kernel_stats.csv gpurun_out/ax/p$x/p_kernel_stats.csv 2>/dev/null | head -1)
  echo "== SUMK_ATTN_X=$x"; grep "attn_strip" $f | awk -F, '{print $1, $4}'
  rm -rf gpurun_out/ax/p$x
done
