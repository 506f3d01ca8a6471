#!/bin/bash
# Probe (diagnostic build): the bf16x3 scoring step with its plane GEMMs on the 16x16x32 MFMA shape (default) against the 32x32x16 kernel
# (SUMK_PW16=0), alternating on ONE box.
export SUMK_LIB_PATH=$PWD/summarizer_amd/libsumk_diag.so
for i in 1 2 3; do for v in 1 0; do
  SUMK_PW16=$v python bench.py --precision bf16x3 --no-cpu-baseline --headline-only --steps 100 --warmup 20 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('bf16x3 step, plane GEMMs on', '16x16x32' if $v else '32x32x16', ':', d['ms_per_step'], 'ms  (QKV launch', d['roofline']['avg_launch_us'], 'us)')"
done; done
