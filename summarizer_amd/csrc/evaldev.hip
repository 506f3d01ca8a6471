// Device-side evaluation tail (SURVEY.md section 8f rank 1 as worded: "upsample + segment-mean + Spearman ranks on GPU, knapsack DP in
// C++ on host threads"): the scores of a test batch stay in HBM; ONE launch -- a block per video -- expands them to frame scores
// (summarizer/utils/eval.py:15-35), takes the float32 segment means that feed `int(mean * 1000)` of the key-shot selection
// (eval.py:91-94; numpy's pairwise summation reproduced operation for operation, so the host knapsack sees the same integers) and
// the mean Spearman correlation with the annotators (eval.py:49-72).  One small D2H (segment means + one double per video) then
// feeds the host side (sumk_eval_videos with seg_means given: knapsack / rank selection, summary expansion, F-scores).
#include "sumk_internal.h"
#include <math.h>

#pragma clang fp contract(off)

namespace sumk {

constexpr int ED_MAX_INT = 4096;     // pick intervals per video the block keeps in LDS (T <= 4095 steps)
constexpr int ED_MAX_USERS = 32;

// numpy's pairwise summation (numpy/core/src/umath/loops_utils.h.src), float32 -- same tree as csrc/evaltail.hip
__device__ float ed_pairwise_sum(const float* a, int n) {
  if (n < 8) {
    float r = 0.f;
    for (int i = 0; i < n; ++i) r += a[i];
    return r;
  }
  if (n <= 128) {
    float r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i = 8;
    for (; i < n - (n % 8); i += 8)
      for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
  }
  int n2 = n / 2;
  n2 -= n2 % 8;
  return ed_pairwise_sum(a, n2) + ed_pairwise_sum(a + n2, n - n2);
}

__global__ __launch_bounds__(256) void eval_device_kernel(const float* __restrict__ scores, const sumk_eval_dev_video* __restrict__ vids,
                                                          float* __restrict__ frame_scratch, float* __restrict__ seg_means,
                                                          double* __restrict__ corr) {
  __shared__ float s_val[ED_MAX_INT + 1];      // value of pick interval i; slot n_int = "frames no interval covers" (value 0)
  __shared__ int s_lo[ED_MAX_INT + 1], s_hi[ED_MAX_INT + 1];
  __shared__ double s_rank[ED_MAX_INT + 1];
  __shared__ double s_red[4][ED_MAX_USERS + 1];
  const sumk_eval_dev_video v = vids[blockIdx.x];
  const int tid = threadIdx.x, n_frames = v.n_frames, np_ = v.n_picks;
  float* fs = frame_scratch + v.frame0;
  const bool sentinel = np_ == 0 || v.picks[np_ - 1] != n_frames;
  const int n_int = np_ - 1 + (sentinel ? 1 : 0);
  // the LDS tables are fixed-size: a descriptor past them (the host wrapper refuses such videos; the descriptors live in device memory,
  // so the entry point cannot) must not write out of bounds -- its results are NaN instead
  if (n_int > ED_MAX_INT || v.n_users > ED_MAX_USERS || n_int < 0) {
    for (int s = tid; s < v.n_segs; s += 256) seg_means[v.seg0 + s] = nanf("");
    if (tid == 0) corr[blockIdx.x] = nan("");
    return;
  }
  // ---- upsample (eval.py:24-34): frames default to 0, interval i = [picks[i], picks[i + 1]) takes score i (0 past the scores)
  for (int f = tid; f < n_frames; f += 256) fs[f] = 0.f;
  for (int i = tid; i < n_int; i += 256) {
    s_lo[i] = max(0, v.picks[i]);
    s_hi[i] = min(n_frames, i + 1 < np_ ? v.picks[i + 1] : n_frames);
    s_val[i] = i < v.n_steps ? scores[v.row0 + i] : 0.f;
  }
  __syncthreads();
  for (int i = 0; i < n_int; ++i) {            // intervals in order (a later interval overwrites an earlier one, like the reference's loop)
    const float val = s_val[i];
    for (int f = s_lo[i] + tid; f < s_hi[i]; f += 256) fs[f] = val;
  }
  __syncthreads();
  // ---- float32 segment means (eval.py:91-94), one thread per segment
  for (int s = tid; s < v.n_segs; s += 256) {
    const int lo = max(0, min(n_frames, v.cps[2 * s])), hi = max(lo, min(n_frames, v.cps[2 * s + 1] + 1));
    seg_means[v.seg0 + s] = hi > lo ? ed_pairwise_sum(fs + lo, hi - lo) / (float)(hi - lo) : 0.f;
  }
  // ---- Spearman (eval.py:49-72): average ranks of -frame_scores.  Frames of one interval share their value, so ranks are taken per
  // interval with the interval's frame count as multiplicity (picks are ascending: checked by the host wrapper).
  if (v.user_ranks == nullptr || v.n_users <= 0) { if (tid == 0) corr[blockIdx.x] = nan(""); return; }
  int covered = 0;
  for (int i = 0; i < n_int; ++i) covered += max(0, s_hi[i] - s_lo[i]);      // (uniform: every thread computes it)
  const int uncovered = n_frames - covered;                                 // frames before the first pick keep the value 0
  if (tid == 0) { s_val[n_int] = 0.f; s_lo[n_int] = 0; s_hi[n_int] = uncovered; }
  __syncthreads();
  const int n_grp = n_int + 1;
  for (int i = tid; i < n_grp; i += 256) {
    const float x = s_val[i];
    long long greater = 0, equal = 0;
    for (int j = 0; j < n_grp; ++j) {
      const int c = max(0, s_hi[j] - s_lo[j]);
      greater += s_val[j] > x ? c : 0;
      equal += s_val[j] == x ? c : 0;
    }
    s_rank[i] = (double)greater + 0.5 * ((double)equal + 1.0);              // average of the 1-based positions greater + 1 .. greater + equal
  }
  __syncthreads();
  const double mean = 0.5 * ((double)n_frames + 1.0);
  double acc[ED_MAX_USERS];
#pragma unroll
  for (int u = 0; u < ED_MAX_USERS; ++u) acc[u] = 0.0;
  double smm = 0.0;
  int it = 0;                                                               // interval of frame f (frames ascend per thread: resume the walk)
  for (int f = tid; f < n_frames; f += 256) {
    while (it < n_int && !(f >= s_lo[it] && f < s_hi[it])) ++it;
    double r;
    if (it < n_int) r = s_rank[it];
    else { r = s_rank[n_int]; it = 0; }                                     // not covered: value 0 (only in front of the first pick)
    const double dm = r - mean;
    smm += dm * dm;
#pragma unroll
    for (int u = 0; u < ED_MAX_USERS; ++u)
      if (u < v.n_users) acc[u] += dm * (v.user_ranks[(long long)u * n_frames + f] - v.user_mean[u]);
  }
  // block reduction (fixed order: lanes by xor butterflies, then the four waves in order)
  const int lane = tid & 63, wave = tid >> 6;
  for (int u = 0; u <= v.n_users; ++u) {
    double x = u < v.n_users ? acc[u < ED_MAX_USERS ? u : 0] : smm;
    if (u < v.n_users) {
      // (acc is indexed with a compile-time-unrollable select to stay in registers)
      x = 0.0;
#pragma unroll
      for (int q = 0; q < ED_MAX_USERS; ++q) x = q == u ? acc[q] : x;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
    if (lane == 0) s_red[wave][u] = x;
  }
  __syncthreads();
  if (tid == 0) {
    const double smm_t = (s_red[0][v.n_users] + s_red[1][v.n_users]) + (s_red[2][v.n_users] + s_red[3][v.n_users]);
    double c = 0.0;
    for (int u = 0; u < v.n_users; ++u) {
      const double sxy = (s_red[0][u] + s_red[1][u]) + (s_red[2][u] + s_red[3][u]);
      c += sxy / sqrt(v.user_ssq[u] * smm_t);
    }
    corr[blockIdx.x] = c / (double)v.n_users;
  }
}

// ---- The same tail as two launches (round 4): the segment means -- all the host's key-shot selection waits for -- leave the device
// after a short first kernel and the knapsack / F-score threads run UNDER the correlation, which is itself spread over ED_CHUNKS blocks
// per video instead of one (the single-block form reads n_users x n_frames doubles with 256 threads: 0.44 ms for 50 videos, all of it
// in front of the host tail).  Partial sums per (video, chunk) are added in chunk order by a second small kernel: deterministic; the
// float64 sums are re-associated with respect to the one-block form (~1e-16 relative).
constexpr int ED_CHUNKS = 8;

// the interval tables of a video (s_lo / s_hi / s_val of eval_device_kernel), shared by both kernels
__device__ __forceinline__ void ed_intervals(const sumk_eval_dev_video& v, const float* __restrict__ scores, int n_int, int* s_lo, int* s_hi,
                                             float* s_val) {
  const int np_ = v.n_picks, n_frames = v.n_frames;
  for (int i = threadIdx.x; i < n_int; i += 256) {
    s_lo[i] = max(0, v.picks[i]);
    s_hi[i] = min(n_frames, i + 1 < np_ ? v.picks[i + 1] : n_frames);
    s_val[i] = i < v.n_steps ? scores[v.row0 + i] : 0.f;
  }
}

__global__ __launch_bounds__(256) void eval_segments_kernel(const float* __restrict__ scores, const sumk_eval_dev_video* __restrict__ vids,
                                                            float* __restrict__ frame_scratch, float* __restrict__ seg_means) {
  __shared__ float s_val[ED_MAX_INT + 1];
  __shared__ int s_lo[ED_MAX_INT + 1], s_hi[ED_MAX_INT + 1];
  const sumk_eval_dev_video v = vids[blockIdx.x];
  const int tid = threadIdx.x, n_frames = v.n_frames, np_ = v.n_picks;
  float* fs = frame_scratch + v.frame0;
  const bool sentinel = np_ == 0 || v.picks[np_ - 1] != n_frames;
  const int n_int = np_ - 1 + (sentinel ? 1 : 0);
  if (n_int > ED_MAX_INT || n_int < 0) {       // (see eval_device_kernel)
    for (int s = tid; s < v.n_segs; s += 256) seg_means[v.seg0 + s] = nanf("");
    return;
  }
  for (int f = tid; f < n_frames; f += 256) fs[f] = 0.f;
  ed_intervals(v, scores, n_int, s_lo, s_hi, s_val);
  __syncthreads();
  for (int i = 0; i < n_int; ++i) {
    const float val = s_val[i];
    for (int f = s_lo[i] + tid; f < s_hi[i]; f += 256) fs[f] = val;
  }
  __syncthreads();
  for (int s = tid; s < v.n_segs; s += 256) {
    const int lo = max(0, min(n_frames, v.cps[2 * s])), hi = max(lo, min(n_frames, v.cps[2 * s + 1] + 1));
    seg_means[v.seg0 + s] = hi > lo ? ed_pairwise_sum(fs + lo, hi - lo) / (float)(hi - lo) : 0.f;
  }
}

// grid (ED_CHUNKS, n_videos): frames [c L, (c + 1) L) of the video, L = ceil(n_frames / ED_CHUNKS); part[(video * ED_CHUNKS + c) * (ED_MAX_USERS + 1) + u]
__global__ __launch_bounds__(256) void eval_spearman_part_kernel(const float* __restrict__ scores, const sumk_eval_dev_video* __restrict__ vids,
                                                                 double* __restrict__ part) {
  __shared__ float s_val[ED_MAX_INT + 1];
  __shared__ int s_lo[ED_MAX_INT + 1], s_hi[ED_MAX_INT + 1];
  __shared__ double s_rank[ED_MAX_INT + 1];
  __shared__ double s_red[4][ED_MAX_USERS + 1];
  const sumk_eval_dev_video v = vids[blockIdx.y];
  const int tid = threadIdx.x, n_frames = v.n_frames, np_ = v.n_picks, chunk = blockIdx.x;
  double* const out = part + ((size_t)blockIdx.y * ED_CHUNKS + chunk) * (ED_MAX_USERS + 1);
  const bool sentinel = np_ == 0 || v.picks[np_ - 1] != n_frames;
  const int n_int = np_ - 1 + (sentinel ? 1 : 0);
  if (n_int > ED_MAX_INT || v.n_users > ED_MAX_USERS || n_int < 0 || v.user_ranks == nullptr || v.n_users <= 0) {
    for (int u = tid; u <= ED_MAX_USERS; u += 256) out[u] = nan("");
    return;
  }
  ed_intervals(v, scores, n_int, s_lo, s_hi, s_val);
  __syncthreads();
  int covered = 0;
  for (int i = 0; i < n_int; ++i) covered += max(0, s_hi[i] - s_lo[i]);
  const int uncovered = n_frames - covered;
  if (tid == 0) { s_val[n_int] = 0.f; s_lo[n_int] = 0; s_hi[n_int] = uncovered; }
  __syncthreads();
  const int n_grp = n_int + 1;
  for (int i = tid; i < n_grp; i += 256) {
    const float x = s_val[i];
    long long greater = 0, equal = 0;
    for (int j = 0; j < n_grp; ++j) {
      const int c = max(0, s_hi[j] - s_lo[j]);
      greater += s_val[j] > x ? c : 0;
      equal += s_val[j] == x ? c : 0;
    }
    s_rank[i] = (double)greater + 0.5 * ((double)equal + 1.0);
  }
  __syncthreads();
  const double mean = 0.5 * ((double)n_frames + 1.0);
  double acc[ED_MAX_USERS];
#pragma unroll
  for (int u = 0; u < ED_MAX_USERS; ++u) acc[u] = 0.0;
  double smm = 0.0;
  const int L = (n_frames + ED_CHUNKS - 1) / ED_CHUNKS, f_end = min(n_frames, (chunk + 1) * L);
  int it = 0;
  for (int f = chunk * L + tid; f < f_end; f += 256) {
    while (it < n_int && !(f >= s_lo[it] && f < s_hi[it])) ++it;
    double r;
    if (it < n_int) r = s_rank[it];
    else { r = s_rank[n_int]; it = 0; }
    const double dm = r - mean;
    smm += dm * dm;
#pragma unroll
    for (int u = 0; u < ED_MAX_USERS; ++u)
      if (u < v.n_users) acc[u] += dm * (v.user_ranks[(long long)u * n_frames + f] - v.user_mean[u]);
  }
  const int lane = tid & 63, wave = tid >> 6;
  for (int u = 0; u <= v.n_users; ++u) {
    double x = smm;
    if (u < v.n_users) {
      x = 0.0;
#pragma unroll
      for (int q = 0; q < ED_MAX_USERS; ++q) x = q == u ? acc[q] : x;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
    if (lane == 0) s_red[wave][u] = x;
  }
  __syncthreads();
  // slot u < n_users: sum_f dm (rank_u - mean_u); slot ED_MAX_USERS: sum_f dm^2
  for (int u = tid; u <= v.n_users; u += 256)
    out[u < v.n_users ? u : ED_MAX_USERS] = (s_red[0][u] + s_red[1][u]) + (s_red[2][u] + s_red[3][u]);
}

__global__ __launch_bounds__(64) void eval_spearman_final_kernel(const sumk_eval_dev_video* __restrict__ vids, const double* __restrict__ part,
                                                                 double* __restrict__ corr) {
  __shared__ double s_sum[ED_MAX_USERS + 1];
  const sumk_eval_dev_video v = vids[blockIdx.x];
  const double* p = part + (size_t)blockIdx.x * ED_CHUNKS * (ED_MAX_USERS + 1);
  const int u = threadIdx.x;
  if (u <= ED_MAX_USERS) {
    double a = 0.0;
    for (int c = 0; c < ED_CHUNKS; ++c) a += p[c * (ED_MAX_USERS + 1) + u];      // chunk order
    s_sum[u] = a;
  }
  __syncthreads();
  if (u == 0) {
    if (v.user_ranks == nullptr || v.n_users <= 0 || v.n_users > ED_MAX_USERS || s_sum[ED_MAX_USERS] != s_sum[ED_MAX_USERS]) { corr[blockIdx.x] = nan(""); return; }
    double c = 0.0;
    for (int q = 0; q < v.n_users; ++q) c += s_sum[q] / sqrt(v.user_ssq[q] * s_sum[ED_MAX_USERS]);
    corr[blockIdx.x] = c / (double)v.n_users;
  }
}

}  // namespace sumk

extern "C" int sumk_eval_device_segments(const float* scores_dev, const sumk_eval_dev_video* videos_dev, int32_t n_videos,
                                         float* frame_scratch_dev, float* seg_means_dev, void* stream) {
  using namespace sumk;
  SUMK_ARG(n_videos >= 0, "eval_device_segments: n_videos=%d", n_videos);
  if (n_videos == 0) return SUMK_OK;
  SUMK_ARG(scores_dev && videos_dev && frame_scratch_dev && seg_means_dev, "eval_device_segments: null pointer");
  hipLaunchKernelGGL(eval_segments_kernel, dim3(n_videos), dim3(256), 0, (hipStream_t)stream, scores_dev, videos_dev, frame_scratch_dev, seg_means_dev);
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}
extern "C" size_t sumk_eval_device_spearman_scratch_bytes(int32_t n_videos) {
  return (size_t)(n_videos > 0 ? n_videos : 0) * sumk::ED_CHUNKS * (sumk::ED_MAX_USERS + 1) * sizeof(double);
}
extern "C" int sumk_eval_device_spearman(const float* scores_dev, const sumk_eval_dev_video* videos_dev, int32_t n_videos, double* scratch_dev,
                                         double* corr_dev, void* stream) {
  using namespace sumk;
  SUMK_ARG(n_videos >= 0, "eval_device_spearman: n_videos=%d", n_videos);
  if (n_videos == 0) return SUMK_OK;
  SUMK_ARG(scores_dev && videos_dev && scratch_dev && corr_dev, "eval_device_spearman: null pointer");
  hipLaunchKernelGGL(eval_spearman_part_kernel, dim3(ED_CHUNKS, n_videos), dim3(256), 0, (hipStream_t)stream, scores_dev, videos_dev, scratch_dev);
  hipLaunchKernelGGL(eval_spearman_final_kernel, dim3(n_videos), dim3(64), 0, (hipStream_t)stream, videos_dev, scratch_dev, corr_dev);
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}

extern "C" int sumk_eval_device(const float* scores_dev, const sumk_eval_dev_video* videos_dev, int32_t n_videos, float* frame_scratch_dev,
                                float* seg_means_dev, double* corr_dev, void* stream) {
  using namespace sumk;
  SUMK_ARG(n_videos >= 0, "eval_device: n_videos=%d", n_videos);
  if (n_videos == 0) return SUMK_OK;
  SUMK_ARG(scores_dev && videos_dev && frame_scratch_dev && seg_means_dev && corr_dev, "eval_device: null pointer");
  hipLaunchKernelGGL(eval_device_kernel, dim3(n_videos), dim3(256), 0, (hipStream_t)stream, scores_dev, videos_dev, frame_scratch_dev,
                     seg_means_dev, corr_dev);
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}
