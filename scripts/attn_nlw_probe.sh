#!/bin/bash
# Probe: how many waves of a block issue the LDS-DMA pieces of the attention-on-planes kernels.  The product library has all 8 waves issue
# (SUMK_ATTN_NLW = 8); summarizer_amd/libsumk_diag.so is built with another count (make DIAG=1 EXTRA=-DSUMK_ATTN_NLW=4).  Timed by the
# library's event pairs inside bench.py (gemm_kernels.qkt = logits + softmax launch, gemm_kernels.alpha_v = context launch), alternating.
for rep in 1 2; do
for lib in "" "$PWD/summarizer_amd/libsumk_diag.so"; do
  for p in bf16x6 bf16x3; do
    SUMK_LIB_PATH=$lib python bench.py --precision $p --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['gemm_kernels']; print('${lib:+probe lib}${lib:-product } $p  step', d['ms_per_step'], 'ms  logits', k['qkt']['avg_launch_us'], 'us  context', k['alpha_v']['avg_launch_us'], 'us  oproj', k['out_proj']['avg_launch_us'])"
  done
done
done
