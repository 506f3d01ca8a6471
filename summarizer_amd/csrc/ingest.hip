// Host-side feature ingest (pure C++, no device code): pack the (T_i, D) fp32 feature matrices of a batch of videos back to
// back into ONE staging buffer -- the pinned buffer a single H2D copy then ships -- with a pool of memcpy threads.
// Replaces the per-video `torch.from_numpy(features).unsqueeze(1).cuda()` of the reference's inference / training loops
// (summarizer/models/__init__.py:47-51, vasnet.py:194-205, dsn.py:98-110): SURVEY.md section 8f rank 3.  One thread moves
// ~10 GB/s; the GPU scores 36 GB/s of features and PCIe Gen5 x16 carries ~50, so the pack has to be parallel.
#include "sumk_internal.h"
#include <algorithm>
#include <cstring>
#include <thread>
#include <vector>

extern "C" int sumk_pack_rows(float* dst, const float* const* srcs, const int32_t* n_rows, int32_t n_videos, int32_t D,
                              int32_t n_threads) {
  using namespace sumk;
  SUMK_ARG(n_videos >= 0 && D > 0, "pack_rows: bad shape (n_videos=%d, D=%d)", n_videos, D);
  if (n_videos == 0) return SUMK_OK;
  SUMK_ARG(dst && srcs && n_rows, "pack_rows: null pointer");
  std::vector<int64_t> off((size_t)n_videos + 1, 0);
  for (int i = 0; i < n_videos; ++i) {
    SUMK_ARG(n_rows[i] > 0 && srcs[i] != nullptr, "pack_rows: video %d is empty", i);
    off[i + 1] = off[i] + (int64_t)n_rows[i] * D;
  }
  const int64_t total = off[n_videos];
  int nt = n_threads > 0 ? n_threads : (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
  nt = (int)std::max<int64_t>(1, std::min<int64_t>(nt, total * 4 / (1 << 20)));   // at least 1 MB per thread
  // thread t copies the float range [t*chunk, (t+1)*chunk) of the packed buffer, whatever videos it spans
  const int64_t chunk = ((total + nt - 1) / nt + 15) & ~(int64_t)15;
  auto work = [&](int t) {
    int64_t lo = (int64_t)t * chunk, hi = std::min(total, lo + chunk);
    if (lo >= hi) return;
    int v = (int)(std::upper_bound(off.begin(), off.end(), lo) - off.begin()) - 1;
    while (lo < hi) {
      const int64_t end = std::min(hi, off[v + 1]);
      std::memcpy(dst + lo, srcs[v] + (lo - off[v]), (size_t)(end - lo) * sizeof(float));
      lo = end; ++v;
    }
  };
  if (nt == 1) { work(0); return SUMK_OK; }
  std::vector<std::thread> pool;
  for (int t = 0; t < nt; ++t) pool.emplace_back(work, t);
  for (auto& th : pool) th.join();
  return SUMK_OK;
}
