// Host-side 0/1 knapsack for key-shot selection (pure C++, no device code).
// Replaces knapsack_ortools (summarizer/utils/knapsack.py:5-23): OR-tools' KNAPSACK_DYNAMIC_PROGRAMMING_SOLVER
// (ortools==7.5.7466, third party, not vendored).  Restated from the published solver: a 1-D DP over capacity
// keeping, per capacity, the best profit and the LAST item that improved it (strict '>'), and a reconstruction that
// re-solves the shrinking sub-problem (items < last selected, remaining capacity).  Parity is pinned wherever the
// optimum is unique (selected set == the unique optimal set; the reference pipeline's summaries and F-scores around an
// exhaustive solver, tests/golden/knapsack_e2e.npz); tie-breaking versus OR-tools stays unverified (oracle/knapsack_np.py).
#include "sumk_internal.h"
#include <vector>

extern "C" int sumk_knapsack_dp(const int64_t* values, const int64_t* weights, int32_t n_items, int64_t capacity,
                                uint8_t* selected) {
  using namespace sumk;
  SUMK_ARG(n_items >= 0 && (n_items == 0 || (values && weights && selected)), "knapsack: null pointer");
  SUMK_ARG(capacity < ((int64_t)1 << 31), "knapsack: capacity too large");
  for (int i = 0; i < n_items; ++i) {
    selected[i] = 0;
    SUMK_ARG(weights[i] >= 0, "knapsack: negative weight at %d", i);
  }
  if (capacity <= 0 || n_items == 0) return SUMK_OK;
  std::vector<int64_t> profit((size_t)capacity + 1);
  std::vector<int32_t> last((size_t)capacity + 1);
  int64_t rem = capacity;
  int n = n_items;
  while (rem > 0 && n > 0) {
    std::fill(profit.begin(), profit.begin() + rem + 1, (int64_t)0);
    std::fill(last.begin(), last.begin() + rem + 1, 0);
    for (int it = 0; it < n; ++it) {
      const int64_t w = weights[it], v = values[it];
      for (int64_t c = rem; c >= w; --c) {
        const int64_t cand = profit[c - w] + v;
        if (cand > profit[c]) { profit[c] = cand; last[c] = it; }
      }
    }
    const int s = last[rem];
    rem -= weights[s];
    n = s;
    if (rem >= 0) selected[s] = 1;
  }
  return SUMK_OK;
}
