#!/bin/bash
# Probe (diagnostic build): the DMA placement variants of the attention-on-planes kernels, timed by the library's event pairs inside
# `bench.py --precision P` (gemm_kernels.qkt = logits + softmax launch, gemm_kernels.alpha_v = context launch).
export SUMK_LIB_PATH=$PWD/summarizer_amd/libsumk_diag.so
for p in bf16x6 bf16x3; do
  for va in 0 1 2; do for vb in 0 2; do
    [ $va != 0 ] && [ $vb != 0 ] && [ $va != $vb ] && continue
    SUMK_ATTN_VAR_A=$va SUMK_ATTN_VAR_B=$vb python bench.py --precision $p --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['gemm_kernels']; print('$p A=$va B=$vb  step', d['ms_per_step'], 'ms  logits', k['qkt']['avg_launch_us'], 'us  context', k['alpha_v']['avg_launch_us'], 'us  oproj', k['out_proj']['avg_launch_us'], ' k1', k['k1']['avg_launch_us'])"
  done; done
done
