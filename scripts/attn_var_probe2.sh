#!/bin/bash
# Probe (diagnostic build): the logits kernel with one half of its work removed -- variant 3 = DMA + fragment reads + barriers only (no
# MFMAs), variant 4 = MFMAs + reads + barriers only (no DMA after the prologue), variant 5 = the DMA stream alone (no reads, MFMAs, barriers)
# -- against the whole kernel (variant 0).
export SUMK_LIB_PATH=$PWD/summarizer_amd/libsumk_diag.so
for va in 0 3 4 5; do
  SUMK_ATTN_VAR_A=$va python bench.py --precision bf16x6 --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['gemm_kernels']; print('bf16x6 A=$va  logits', k['qkt']['avg_launch_us'], 'us  context', k['alpha_v']['avg_launch_us'])"
done
