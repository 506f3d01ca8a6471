#!/bin/bash
# Audit (CPU, no GPU needed): compile every kernel source to gfx950 assembly and count WATERFALL loops -- the v_readfirstlane / v_cmp_eq / s_and_saveexec
# loop the compiler wraps around a buffer instruction whose descriptor or scalar offset it could not prove wave-uniform.  Round 6 found them around every
# LDS-DMA instruction of the re-cut context launch (block coordinates out of shuffles), around the MC-operand loads of the wide bf16 GEMM (k-tile strides
# carried through the tile loop) and around the 32 exchange loads per step of lstm_wide2_kernel (wave id left in a VGPR): expected output is 0 everywhere.
cd "$(dirname "$0")/../summarizer_amd/csrc" || exit 1
rc=0
for f in *.hip; do
  n=$(/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -S --cuda-device-only "$f" -o - 2>/dev/null | grep -c 's_and_saveexec_b64 vcc, vcc')
  echo "$f: $n"
  [ "$n" = 0 ] || rc=1
done
exit $rc
