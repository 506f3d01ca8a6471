"""0/1 knapsack used for key-shot selection.  TEST INFRASTRUCTURE ONLY.  *** PARITY PINNED EXCEPT TIES ***

Reference call site: summarizer/utils/knapsack.py:5-23, reached from summarizer/utils/eval.py:99.
The arithmetic lives in a third-party dependency that is NOT under /root/reference:
  ortools==7.5.7466 (summarizer/requirements.txt:11), KnapsackSolver(KNAPSACK_DYNAMIC_PROGRAMMING_SOLVER).
It is not installed and cannot be installed here (no network), and no reference test or fixture holds a
knapsack input/output pair.  What follows restates, FROM MEMORY of the published source
(ortools/algorithms/knapsack_solver.cc, KnapsackDynamicProgrammingSolver), the solver's DP and its
solution reconstruction:
  SolveSubProblem(cap, n): profits[0..cap]=0, sel[0..cap]=0; for item in 0..n-1: for c = cap down to w[item]:
        if profits[c-w]+v > profits[c] (STRICT): profits[c]=..., sel[c]=item;  return sel[cap]
  Solve(): rem=capacity, n=num_items; while rem>0 and n>0: s=SolveSubProblem(rem,n); rem-=w[s]; n=s;
        if rem>=0: best[s]=True
What IS pinned (tests/test_host_eval.py):
  * the optimal VALUE against brute force;
  * the selected SET on every instance whose optimum is unique (exhaustive check, S <= 18): any exact solver, OR-tools'
    included, must return that set;
  * the whole reference pipeline around the solver -- segment means, trunc(1000 * mean) values, capacity, expansion to
    frames, F-scores -- against tests/golden/knapsack_e2e.npz, produced by the REAL `generate_summary(method="knapsack")`
    + `evaluate_summary` with only the solver swapped for an exhaustive one (make_golden_knapsack.py), unique-optimum
    videos only.
What remains UNVERIFIED: which of several equally valuable subsets OR-tools returns (tie-breaking; also its optional
problem-reduction pass).  A tie needs two feasible subsets with the same sum of trunc(1000 * mean) values -- e.g. equal
segment scores, or zero-valued segments that may be included or not.
The value/weight conversion IS the reference's: values = trunc(score*1000) as int, weights = int(nfps)
(knapsack.py:11-15).
"""
import itertools
import numpy as np


def to_int_problem(values, weights):
    """knapsack.py:11-15: float64 array * 1000, truncated toward zero (np.int astype)."""
    v = (np.array(values, dtype=np.float64) * 1000).astype(np.int64)
    w = np.array(weights).astype(np.int64)
    return v, w


def _sub(v, w, cap, n):
    prof = np.zeros(cap + 1, dtype=np.int64)
    sel = np.zeros(cap + 1, dtype=np.int64)
    for it in range(n):
        wi, vi = int(w[it]), int(v[it])
        if wi > cap:
            continue
        if wi <= 0:
            # zero-weight item: every capacity can take it
            better = prof + vi > prof
            prof = np.where(better, prof + vi, prof); sel = np.where(better, it, sel)
            continue
        cand = prof[: cap + 1 - wi] + vi          # uses the PREVIOUS row (descending-c in-place update)
        better = cand > prof[wi:]
        prof[wi:] = np.where(better, cand, prof[wi:])
        sel[wi:] = np.where(better, it, sel[wi:])
    return int(sel[cap]), int(prof[cap])


def knapsack_dp(values, weights, items, capacity):
    """Returns sorted list of selected item indices (same contract as knapsack.py:19-23)."""
    v, w = to_int_problem(values, weights)
    n = int(items)
    best = [False] * n
    rem = int(capacity)
    while rem > 0 and n > 0:
        s, _ = _sub(v, w, rem, n)
        rem -= int(w[s])
        n = s
        if rem >= 0:
            best[s] = True
    return [i for i, b in enumerate(best) if b]


def knapsack_value(values, weights, picks):
    v, w = to_int_problem(values, weights)
    return int(sum(v[i] for i in picks)), int(sum(w[i] for i in picks))


def knapsack_bruteforce_value(values, weights, capacity):
    v, w = to_int_problem(values, weights)
    n = len(v)
    best = 0
    for mask in itertools.product((0, 1), repeat=n):
        ww = sum(int(w[i]) for i in range(n) if mask[i])
        if ww <= capacity:
            best = max(best, sum(int(v[i]) for i in range(n) if mask[i]))
    return best
