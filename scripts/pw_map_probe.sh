#!/bin/bash
# Probe (diagnostic build): XCD tile maps of the plane GEMM (SUMK_PW_XCD_MAP = 1: 2 x 4 rectangles, 2: row bands) inside the scoring step.
export SUMK_LIB_PATH=$PWD/summarizer_amd/libsumk_diag.so
for p in bf16x6 bf16x3; do for m in 1 2 1 2; do
  SUMK_PW_XCD_MAP=$m python bench.py --precision $p --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['gemm_kernels']; print('$p map=$m  step', d['ms_per_step'], 'ms  qkv', d['roofline']['avg_launch_us'], ' oproj', k['out_proj']['avg_launch_us'], ' k1', k['k1']['avg_launch_us'], ' logits', k['qkt']['avg_launch_us'], ' context', k['alpha_v']['avg_launch_us'])"
done; done
