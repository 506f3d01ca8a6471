"""Key-shot knapsack -- mirror of `summarizer/utils/knapsack.py` (same function name and argument meaning) over
the native DP in libsumk.so (`sumk_knapsack_dp`, csrc/knapsack.hip) instead of Google OR-tools."""
import ctypes as C
import numpy as np

from .. import _lib


def knapsack_ortools(values, weights, items, capacity):
    """0-1 Knapsack problem solver (name kept so `eval.generate_summary` reads like the reference's).

    values: float segment scores; weights: frames per segment; items: number of segments; capacity: frame budget.
    Returns the list of packed item indices (knapsack.py:19-23)."""
    lib = _lib.load()
    scale = 1000                                                       # knapsack.py:10
    v = (np.array(values, dtype=np.float64) * scale).astype(np.int64)  # knapsack.py:13 (np.int truncation)
    w = np.array(weights).astype(np.int64)                             # knapsack.py:14
    n = int(items)
    sel = np.zeros(max(n, 1), dtype=np.uint8)
    rc = lib.sumk_knapsack_dp(v.ctypes.data_as(C.POINTER(C.c_int64)), w.ctypes.data_as(C.POINTER(C.c_int64)), n,
                              int(capacity), sel.ctypes.data_as(C.POINTER(C.c_uint8)))
    _lib.check(rc, "sumk_knapsack_dp")
    return [x for x in range(0, len(w)) if sel[x]]
