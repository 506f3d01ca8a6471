for e in "$@"; do
  echo "== AT_EXP=$e"
  SUMK_LIB_PATH=$GRAFT_REPO_ROOT/summarizer_amd/libsumk_exp$e.so timeout 300 python3 scripts/probes/attn_stamps_tmp.py 2>&1 | grep "us:"
done
