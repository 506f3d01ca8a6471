"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatements (numpy, plus one stock-PyTorch functional port) of the reference's
frame-importance scoring hot path (sylvainma/Summarizer, summarizer/models/{vasnet,dsn,sumgan}.py,
summarizer/utils/{eval,knapsack}.py).  Every function cites the reference file:line it follows.

Who may import this package: tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg --
and there only as the CHECKER (or as the reported CPU baseline), never as the thing shipped or measured.
Nothing under summarizer_amd/ imports it; the product path fails loudly when the HIP library is absent.

Pinning status (see DESIGN.md section "Oracle"):
  * vasnet_np / lstm_np / reward_np / eval_np / torch_port : PINNED against outputs of the real reference,
    imported read-only in the build container (tests/golden/make_golden.py -> tests/golden/*.npz).
  * knapsack_np : PARITY UNPINNED.  The reference delegates to ortools==7.5.7466 (not vendored, not
    installable here) and no reference test holds a knapsack vector.  The DP here restates the published
    OR-tools KnapsackDynamicProgrammingSolver from memory; only the optimal VALUE is checked (brute force).
"""
