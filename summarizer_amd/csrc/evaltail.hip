// Host-side evaluation tail for a batch of videos, natively threaded (pure C++, no device code).
// Replaces the per-video Python/numpy loops of the reference's `Trainer.test` tail:
//   upsample            summarizer/utils/eval.py:15-35
//   generate_summary    eval.py:74-123   (segment means -> knapsack / rank -> binary frame vector)
//   evaluate_summary    eval.py:125-165  (precision / recall / F per annotator, mean and max)
//   evaluate_scores     eval.py:49-72    (Spearman = Pearson of average ranks)
// Bit-exactness contract (tests/test_host_eval.py): machine summaries and F-scores equal the numpy implementation BIT FOR
// BIT -- which requires reproducing numpy's float32 PAIRWISE summation (8 accumulators, blocks of 128, recursive halves)
// for the segment means that feed `int(mean * 1000)` and for the mean over annotators; the rank correlation is float64 and
// agrees to ~1e-15 (summation order of the dot products differs from BLAS).  Compiled with -ffp-contract=off semantics
// (no FMA contraction in this file: every operation rounds like numpy's).
#include "sumk_internal.h"
#include <algorithm>
#include <cmath>
#include <numeric>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <atomic>
#include <functional>
#include <unistd.h>
#include <vector>

#pragma clang fp contract(off)

namespace {

// numpy's pairwise summation (numpy/core/src/umath/loops_utils.h.src), for T = float or double
template <typename T>
T pairwise_sum(const T* a, int64_t n) {
  if (n < 8) {
    T r = (T)0;   // numpy starts from -0.0 for floats; +0 vs -0 only differs for an all -0.0 input
    for (int64_t i = 0; i < n; ++i) r += a[i];
    return r;
  }
  if (n <= 128) {
    T r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int64_t i = 8;
    for (; i < n - (n % 8); i += 8)
      for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    T res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
  }
  int64_t n2 = n / 2;
  n2 -= n2 % 8;
  return pairwise_sum(a, n2) + pairwise_sum(a + n2, n - n2);
}

struct Scratch {
  std::vector<float> frame_scores, summary;
  std::vector<double> rank, dsum;
  std::vector<int32_t> order;
  std::vector<int64_t> values, weights;
  std::vector<uint8_t> sel;
  std::vector<float> f32;
  std::vector<double> f64;
};

// average ranks (1-based) of key[i] = -x[i], ties averaged -- scipy.stats.rankdata(-x)
void rank_desc(const float* x, int n, Scratch& S) {
  S.order.resize(n); S.rank.resize(n);
  std::iota(S.order.begin(), S.order.end(), 0);
  std::stable_sort(S.order.begin(), S.order.end(), [&](int a, int b) { return -x[a] < -x[b]; });
  int i = 0;
  while (i < n) {
    int j = i;
    while (j + 1 < n && x[S.order[j + 1]] == x[S.order[i]]) ++j;
    const double r = 0.5 * ((double)(i + 1) + (double)(j + 1));
    for (int k = i; k <= j; ++k) S.rank[S.order[k]] = r;
    i = j + 1;
  }
}

template <typename T>
void fscores(const T* m, const float* user_summary, int n_users, int n_frames, T eps, double* f_avg, double* f_max, std::vector<T>& f) {
  T m_sum = pairwise_sum(m, (int64_t)n_frames);
  f.resize(n_users);
  std::vector<T> prod(n_frames), u(n_frames);
  for (int k = 0; k < n_users; ++k) {
    const float* us = user_summary + (int64_t)k * n_frames;
    for (int i = 0; i < n_frames; ++i) { u[i] = us[i] > 0.f ? (T)1 : (T)0; prod[i] = m[i] * u[i]; }
    const T overlap = pairwise_sum(prod.data(), (int64_t)n_frames);
    const T precision = overlap / (T)(m_sum + eps);
    // the annotator's sum stays float32 in numpy (a float32 array summed, + a python float under NEP 50) even when the
    // machine summary was promoted to float64 by its zero padding
    float gsum = 0.f;
    for (int i = 0; i < n_frames; ++i) gsum += us[i] > 0.f ? 1.f : 0.f;      // exact: a count below 2^24
    const T recall = overlap / (T)(float)(gsum + 1e-8f);
    f[k] = (precision == (T)0 && recall == (T)0) ? (T)0 : (((T)2 * precision) * recall) / (precision + recall);
  }
  *f_avg = (double)(T)(pairwise_sum(f.data(), (int64_t)n_users) / (T)n_users);
  *f_max = (double)*std::max_element(f.begin(), f.end());
}

int eval_one(sumk_eval_video& v, double proportion, int method, Scratch& S) {
  const int n_frames = v.n_frames;
  const bool from_device = v.seg_means != nullptr;     // upsampling, segment means and correlation were done by sumk_eval_device
  if (!from_device) {
  // ---- upsample (eval.py:24-34): literal interval assignment, sentinel n_frames appended when the last pick differs
  S.frame_scores.assign((size_t)n_frames, 0.f);
  const int np_ = v.n_picks;
  const bool sentinel = np_ == 0 || v.picks[np_ - 1] != n_frames;
  const int n_int = np_ - 1 + (sentinel ? 1 : 0);
  if (n_int > v.n_steps + 1) return -1;
  for (int i = 0; i < n_int; ++i) {
    const int lo = std::max(0, v.picks[i]);
    const int hi = std::min(n_frames, i + 1 < np_ ? v.picks[i + 1] : n_frames);
    const float val = i < v.n_steps ? v.scores[i] : 0.f;
    for (int f = lo; f < hi; ++f) S.frame_scores[f] = val;
  }
  // ---- rank correlation with the annotators (eval.py:49-72)
  v.corr = std::nan("");
  if (v.user_ranks != nullptr && v.n_users > 0) {
    rank_desc(S.frame_scores.data(), n_frames, S);
    double mean = pairwise_sum(S.rank.data(), (int64_t)n_frames) / (double)n_frames;
    S.dsum.resize(n_frames);
    double smm = 0.0;
    for (int i = 0; i < n_frames; ++i) { S.dsum[i] = S.rank[i] - mean; smm += S.dsum[i] * S.dsum[i]; }
    double acc = 0.0;
    for (int k = 0; k < v.n_users; ++k) {
      const double* ru = v.user_ranks + (int64_t)k * n_frames;
      const double mu = pairwise_sum(ru, (int64_t)n_frames) / (double)n_frames;
      double sxy = 0.0, suu = 0.0;
      for (int i = 0; i < n_frames; ++i) { const double d = ru[i] - mu; sxy += d * S.dsum[i]; suu += d * d; }
      acc += sxy / std::sqrt(suu * smm);
    }
    v.corr = acc / (double)v.n_users;
  }
  }
  // ---- key-shot summary (eval.py:74-123)
  v.f_avg = v.f_max = std::nan("");
  if (v.cps == nullptr || v.n_segs <= 0) return 0;
  const int S_ = v.n_segs;
  S.values.resize(S_); S.weights.resize(S_); S.sel.assign(S_, 0);
  std::vector<double> seg(S_);
  for (int s = 0; s < S_; ++s) {
    const int lo = std::max(0, std::min(n_frames, v.cps[2 * s])), hi = std::max(lo, std::min(n_frames, v.cps[2 * s + 1] + 1));
    const float mean = from_device ? v.seg_means[s]
                     : hi > lo ? pairwise_sum(S.frame_scores.data() + lo, (int64_t)(hi - lo)) / (float)(hi - lo) : 0.f;   // float32 mean (numpy gives NaN for an empty segment)
    seg[s] = (double)mean;
    S.values[s] = (int64_t)(seg[s] * 1000.0);   // np.int truncation, knapsack.py:13
    S.weights[s] = v.nfps[s];
  }
  const int64_t limits = (int64_t)std::floor((double)n_frames * proportion);
  if (method == 0) {
    if (sumk_knapsack_dp(S.values.data(), S.weights.data(), S_, limits, S.sel.data()) != SUMK_OK) return -2;
  } else {
    S.order.resize(S_);
    std::iota(S.order.begin(), S.order.end(), 0);
    std::stable_sort(S.order.begin(), S.order.end(), [&](int a, int b) { return seg[a] < seg[b]; });
    int64_t total = 0;
    for (int q = S_ - 1; q >= 0; --q) {          // descending score; strict '<' of eval.py:105
      const int i = S.order[q];
      if (total + v.nfps[i] < limits) { S.sel[i] = 1; total += v.nfps[i]; }
    }
  }
  int64_t len = 0;
  for (int s = 0; s < S_; ++s) len += v.nfps[s];
  S.summary.assign((size_t)len, 0.f);
  {
    int64_t at = 0;
    for (int s = 0; s < S_; ++s) { if (S.sel[s]) std::fill(S.summary.begin() + at, S.summary.begin() + at + v.nfps[s], 1.f); at += v.nfps[s]; }
  }
  v.summary_len = (int32_t)len;
  if (v.machine_summary != nullptr) std::copy(S.summary.begin(), S.summary.end(), v.machine_summary);
  // ---- F-scores (eval.py:125-165): float32 arithmetic, or float64 once the summary had to be zero-padded
  if (v.user_summary != nullptr && v.n_users > 0) {
    if (len >= n_frames) {
      fscores<float>(S.summary.data(), v.user_summary, v.n_users, n_frames, 1e-8f, &v.f_avg, &v.f_max, S.f32);
    } else {
      std::vector<double> md((size_t)n_frames, 0.0);
      for (int64_t i = 0; i < len; ++i) md[i] = S.summary[i];
      fscores<double>(md.data(), v.user_summary, v.n_users, n_frames, 1e-8, &v.f_avg, &v.f_max, S.f64);
    }
  }
  return 0;
}

// Worker threads kept between calls (the default n_threads <= 0 path): a call's per-video work is ~50 us x 50 videos, and starting 16
// threads per call cost more than they saved (0.62 ms with 16 fresh threads, 0.55 with 8, 2.6 single-threaded, for 50 TVSum videos).
// The pool is created on first use, never destroyed (detached workers parked on a condition variable) and rebuilt in a child process
// after fork() -- threads do not survive it.
struct EvalPool {
  std::mutex m;
  std::condition_variable cv_go, cv_done;
  uint64_t gen = 0;
  int active = 0, n_workers = 0, n_items = 0;
  std::atomic<int> next{0};
  const std::function<void(int, Scratch&)>* fn = nullptr;
  pid_t pid = 0;
};
EvalPool* g_eval_pool = nullptr;
std::mutex g_eval_pool_mutex;

void eval_pool_worker(EvalPool* P) {
  uint64_t seen = 0;
  Scratch S;
  for (;;) {
    std::unique_lock<std::mutex> lk(P->m);
    P->cv_go.wait(lk, [&] { return P->gen != seen; });
    seen = P->gen;
    const std::function<void(int, Scratch&)>* f = P->fn;
    const int n = P->n_items;
    lk.unlock();
    for (int i; (i = P->next.fetch_add(1, std::memory_order_relaxed)) < n;) (*f)(i, S);
    lk.lock();
    if (--P->active == 0) P->cv_done.notify_one();
  }
}

void eval_pool_run(int n_items, const std::function<void(int, Scratch&)>& f) {
  std::lock_guard<std::mutex> one_call_at_a_time(g_eval_pool_mutex);
  if (g_eval_pool == nullptr || g_eval_pool->pid != getpid()) {
    EvalPool* P = new EvalPool;          // (a pool inherited through fork() is abandoned, not freed: its workers do not exist here)
    P->pid = getpid();
    P->n_workers = (int)std::min(15u, std::max(1u, std::thread::hardware_concurrency()) - 1u);
    for (int t = 0; t < P->n_workers; ++t) std::thread(eval_pool_worker, P).detach();
    g_eval_pool = P;
  }
  EvalPool* P = g_eval_pool;
  {
    std::lock_guard<std::mutex> lk(P->m);
    P->fn = &f; P->n_items = n_items; P->next.store(0, std::memory_order_relaxed); P->active = P->n_workers; ++P->gen;
  }
  P->cv_go.notify_all();
  Scratch S;
  for (int i; (i = P->next.fetch_add(1, std::memory_order_relaxed)) < n_items;) f(i, S);
  std::unique_lock<std::mutex> lk(P->m);
  P->cv_done.wait(lk, [&] { return P->active == 0; });
}

}  // namespace

extern "C" int sumk_eval_videos(sumk_eval_video* vids, int32_t n_videos, double proportion, int32_t method, int32_t n_threads) {
  using namespace sumk;
  SUMK_ARG(n_videos >= 0 && (n_videos == 0 || vids != nullptr), "eval_videos: null batch");
  SUMK_ARG(method == 0 || method == 1, "eval_videos: method must be 0 (knapsack) or 1 (rank)");
  for (int i = 0; i < n_videos; ++i) {
    SUMK_ARG(vids[i].seg_means || (vids[i].scores && vids[i].picks && vids[i].n_steps > 0), "eval_videos: video %d is incomplete", i);
    SUMK_ARG(vids[i].n_frames > 0, "eval_videos: video %d has no frames", i);
    SUMK_ARG(vids[i].n_segs == 0 || (vids[i].cps && vids[i].nfps), "eval_videos: video %d has segments but no change points", i);
  }
  int nt = n_threads > 0 ? n_threads : (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
  nt = std::max(1, std::min(nt, n_videos));
  std::vector<int> status((size_t)std::max(1, n_videos), 0);
  auto work = [&](int t) {
    Scratch S;
    for (int i = t; i < n_videos; i += nt) { try { status[i] = eval_one(vids[i], proportion, method, S); } catch (...) { status[i] = -3; } }
  };
  if (n_threads <= 0 && n_videos > 1) {
    // (an exception on a detached worker -- std::bad_alloc in a scratch vector -- would be std::terminate: it becomes the item's status)
    eval_pool_run(n_videos, [&](int i, Scratch& S) {
      try { status[i] = eval_one(vids[i], proportion, method, S); } catch (...) { status[i] = -3; }
    });
  } else if (nt == 1) {
    work(0);
  } else {
    std::vector<std::thread> pool;
    for (int t = 0; t < nt; ++t) pool.emplace_back(work, t);
    for (auto& th : pool) th.join();
  }
  for (int i = 0; i < n_videos; ++i)
    if (status[i] != 0) { set_error("eval_videos: video %d failed (%s)", i, status[i] == -1 ? "more pick intervals than scores + 1" : status[i] == -3 ? "out of memory / exception in the evaluation worker" : "knapsack"); return SUMK_ERR_ARG; }
  return SUMK_OK;
}
