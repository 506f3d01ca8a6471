cd $GRAFT_REPO_ROOT
python3 -c "
from summarizer_amd import kernels as k
for rnd in (False, True):
  for kind in ('bf16','bf16_16','f32'):
    for it in (4000, 16000):
        print(kind, 'random' if rnd else 'const', it, k.mfma_sustained_rate(kind, it, rnd))
"
