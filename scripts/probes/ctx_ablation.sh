# Timing ablations of attn_pw_context_kernel (QT = 4): variant libraries summarizer_amd/libsumk_abl<bits>.so built with -DSUMK_CTX_ABL=<bits>
# (WRONG results by construction; never the product library).  Prints the context kernel's average duration per variant.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ctxabl; O=gpurun_out/ctxabl
for p in ${PRECS:-bf16x6 bf16x3}; do
for lib in product $(ls summarizer_amd/libsumk_abl*.so 2>/dev/null); do
  if [ $lib = product ]; then unset SUMK_LIB_PATH; else export SUMK_LIB_PATH=$PWD/$lib; fi
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python3 bench.py --no-cpu-baseline --headline-only --steps 20 --warmup 5 --precision $p > $O/prof.log 2>&1
  f=$(ls $O/prof/*/p_kernel_stats.csv $O/prof/p_kernel_stats.csv 2>/dev/null | head -1)
  echo "$p $lib: $(python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'attn_pw' in r['Name']: print(r['Name'].split('attn_pw_')[1][:28], '%.1f us |' % (float(r['AverageNs'])/1e3), end=' ')
")"
  rm -rf $O/prof
done; done
