// DSN diversity-representativeness reward (reference: DSNTrainer.compute_reward, summarizer/models/dsn.py:185-236)
// for E episodes of a packed batch of videos.
//
// The reference rebuilds two (T,D)x(D,T) products per episode (cosine Gram, dsn.py:217-218; squared distances,
// dsn.py:228-230).  Both are functions of the features only, so here ONE Gram matrix G = X X^T per video is built
// with the MFMA GEMM and every episode is a cheap masked reduction over it:
//   cos(a,b)   = G[a][b] / (|x_a| |x_b|),  |x_a|^2 = G[a][a]
//   d2(t,p)    = G[t][t] + G[p][p] - 2 G[t][p]
//   R_div      = sum_{a,b in picks} (|a-b| > thre && !far_sim ? 1 : 1 - cos(a,b)) / (n (n-1))      dsn.py:219-225
//   R_rep      = exp(- mean_t min_{p in picks} d2(t,p))                                           dsn.py:231-233
//   reward     = (R_div + R_rep) / 2 ; 0 when nothing is picked (dsn.py:199-203); R_div = 0 for one pick (dsn.py:211-214;
//                the reference then crashes on a 0-dim index, dsn.py:229-230 -- here R_rep is simply evaluated).
#include "sumk_internal.h"
#include <math.h>

namespace sumk {

struct RSeq { int64_t goff; int32_t row0, T, ldG, pad_; };

struct RewardWs { size_t gram, seq, prob, rowres, total; int64_t g_elems; int32_t n_rows; };

static int reward_carve(int D, int n_seq, const int32_t* off, int n_ep, RewardWs* w) {
  SUMK_ARG(D > 0 && D % 4 == 0, "dsn_reward: D=%d must be a positive multiple of 4", D);
  SUMK_ARG(n_seq > 0 && off && off[0] == 0 && n_ep > 0, "dsn_reward: empty batch");
  int64_t e = 0;
  for (int s = 0; s < n_seq; ++s) {
    int T = off[s + 1] - off[s];
    SUMK_ARG(T > 0, "dsn_reward: video %d has %d frames", s, T);
    e += (int64_t)T * ((T + 3) & ~3);
  }
  size_t p = 0;
  auto take = [&](size_t bytes) { size_t at = p; p += align_up(bytes, 256); return at; };
  w->n_rows = off[n_seq]; w->g_elems = e;
  w->gram = take((size_t)e * 4);
  w->seq = take((size_t)n_seq * sizeof(RSeq));
  w->prob = take((size_t)n_seq * sizeof(GemmProb));
  w->rowres = take((size_t)n_ep * w->n_rows * 2 * 4);
  w->total = p;
  return SUMK_OK;
}

__global__ void reward_setup_kernel(const int32_t* off, int n_seq, int D, RSeq* seq, GemmProb* prob) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_seq) return;
  int64_t goff = 0; int ts = 0;
  for (int q = 0; q < s; ++q) {
    int T = off[q + 1] - off[q], tm = (T + 63) / 64;
    goff += (int64_t)T * ((T + 3) & ~3); ts += tm * tm;
  }
  const int row0 = off[s], T = off[s + 1] - row0, ld = (T + 3) & ~3, tm = (T + 63) / 64;
  RSeq si; si.goff = goff; si.row0 = row0; si.T = T; si.ldG = ld; si.pad_ = 0;
  seq[s] = si;
  GemmProb q;
  q.a_off = (int64_t)row0 * D; q.b_off = (int64_t)row0 * D; q.c_off = goff; q.r_off = 0;
  q.M = T; q.N = T; q.K = D; q.lda = D; q.ldb = D; q.ldc = ld; q.ldr = 0; q.tile_start = ts; q.tiles_n = tm;
  for (int i = 0; i < 7; ++i) q.pad_[i] = 0;
  prob[s] = q;
}

// one wave per (episode, frame): min over picks of d2, and (if the frame is picked) its row of the diversity sum
__global__ __launch_bounds__(256) void reward_rows_kernel(const float* __restrict__ G, const RSeq* __restrict__ seq,
                                                          const int32_t* __restrict__ off, int n_seq, int n_rows,
                                                          const float* __restrict__ actions, int n_ep, int far_sim,
                                                          int thre, float* __restrict__ rowres) {
  const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= (int64_t)n_ep * n_rows) return;
  const int lane = threadIdx.x & 63;
  const int ep = (int)(wid / n_rows), row = (int)(wid % n_rows);
  int lo = 0, hi = n_seq - 1;
  while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (off[mid] <= row) lo = mid; else hi = mid - 1; }
  const RSeq si = seq[lo];
  const int t = row - si.row0, T = si.T;
  const float* g = G + si.goff;
  const float* act = actions + (int64_t)ep * n_rows + si.row0;
  const float gtt = g[(int64_t)t * si.ldG + t];
  const bool picked = act[t] != 0.f;
  const float inv_nt = 1.0f / sqrtf(gtt);
  float mn = INFINITY, div = 0.f;
  for (int p = lane; p < T; p += 64) {
    if (act[p] == 0.f) continue;
    const float gtp = g[(int64_t)t * si.ldG + p], gpp = g[(int64_t)p * si.ldG + p];
    mn = fminf(mn, (gtt + gpp) - 2.f * gtp);
    if (picked) {
      int dist = t > p ? t - p : p - t;
      div += (!far_sim && dist > thre) ? 1.f : 1.f - gtp * inv_nt * (1.0f / sqrtf(gpp));
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o)); div += __shfl_xor(div, o); }
  if (lane == 0) { rowres[2 * wid] = mn; rowres[2 * wid + 1] = div; }
}

// one wave per (episode, video): fixed-order sums over the video's frames
__global__ __launch_bounds__(64) void reward_final_kernel(const float* __restrict__ rowres, const RSeq* __restrict__ seq,
                                                          const float* __restrict__ actions, int n_seq, int n_rows,
                                                          float* __restrict__ reward) {
  const int ep = blockIdx.x / n_seq, s = blockIdx.x % n_seq, lane = threadIdx.x;
  const RSeq si = seq[s];
  float smin = 0.f, sdiv = 0.f, cnt = 0.f;
  for (int t = lane; t < si.T; t += 64) {
    const int64_t wid = (int64_t)ep * n_rows + si.row0 + t;
    smin += rowres[2 * wid]; sdiv += rowres[2 * wid + 1];
    cnt += actions[(int64_t)ep * n_rows + si.row0 + t] != 0.f ? 1.f : 0.f;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { smin += __shfl_xor(smin, o); sdiv += __shfl_xor(sdiv, o); cnt += __shfl_xor(cnt, o); }
  if (lane == 0) {
    float r = 0.f;
    if (cnt > 0.f) {
      float rdiv = cnt > 1.f ? sdiv / (cnt * (cnt - 1.f)) : 0.f;
      float rrep = expf(-(smin / (float)si.T));
      r = (rdiv + rrep) * 0.5f;
    }
    reward[(int64_t)ep * n_seq + s] = r;
  }
}

}  // namespace sumk

using namespace sumk;

extern "C" size_t sumk_dsn_reward_workspace_bytes(int32_t D, int32_t n_seq, const int32_t* seq_off_host, int32_t n_episodes) {
  RewardWs w;
  if (reward_carve(D, n_seq, seq_off_host, n_episodes, &w) != SUMK_OK) return 0;
  return w.total;
}

extern "C" int sumk_dsn_reward(const float* x, int32_t D, int32_t n_seq, const int32_t* seq_off_host,
                               const int32_t* seq_off_dev, const float* actions, int32_t n_episodes, int32_t far_sim,
                               int32_t temp_dist_thre, float* reward, void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SUMK_ARG(x && seq_off_dev && actions && reward && workspace, "dsn_reward: null pointer");
  RewardWs L;
  SUMK_TRY(reward_carve(D, n_seq, seq_off_host, n_episodes, &L));
  if (workspace_bytes < L.total) { set_error("dsn_reward: workspace %zu < required %zu", workspace_bytes, L.total); return SUMK_ERR_WORKSPACE; }
  char* ws = (char*)workspace;
  float* G = (float*)(ws + L.gram);
  RSeq* seq = (RSeq*)(ws + L.seq);
  GemmProb* prob = (GemmProb*)(ws + L.prob);
  float* rowres = (float*)(ws + L.rowres);
  hipLaunchKernelGGL(reward_setup_kernel, dim3((n_seq + 63) / 64), dim3(64), 0, stream, seq_off_dev, n_seq, D, seq, prob);
  int tiles = 0;
  for (int s = 0; s < n_seq; ++s) { int tm = (seq_off_host[s + 1] - seq_off_host[s] + 63) / 64; tiles += tm * tm; }
  GemmLaunch g;
  g.A = x; g.B[0] = x; g.C = G; g.probs = prob; g.nprob = n_seq; g.small_tile = 1; g.total_tiles = tiles;
  SUMK_TRY(launch_gemm(GEMM_NT, EPI_NONE, g, stream));
  const int64_t nw = (int64_t)n_episodes * L.n_rows;
  hipLaunchKernelGGL(reward_rows_kernel, dim3((unsigned)((nw + 3) / 4)), dim3(256), 0, stream, G, seq, seq_off_dev, n_seq,
                     L.n_rows, actions, n_episodes, far_sim, temp_dist_thre, rowres);
  hipLaunchKernelGGL(reward_final_kernel, dim3(n_episodes * n_seq), dim3(64), 0, stream, rowres, seq, actions, n_seq, L.n_rows,
                     reward);
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}

// ------------------------------------------------------------------------------------------- REINFORCE policy loss
// The loss glue of DSNTrainer.train (dsn.py:113-140) for a packed batch, one block per video:
//   l_v = [ beta * (mean_t p_t - eps_t)^2  -  sum_e (r[e,v] - b[v]) * mean_t log P(a[e,t] | p_t) ] / E
// with log P = a log(pc) + (1 - a) log1p(-pc), pc = clamp(p, 2^-23 .., 1 - ..) -- torch.distributions.Bernoulli.log_prob
// (probs_to_logits + binary_cross_entropy_with_logits) written out -- and its gradient w.r.t. p
//   dl_v/dp_t = [ 2 beta (mean p - eps_t) - sum_e (r - b)[e,v] (a[e,t] - pc) / (pc (1 - pc)) * [p inside the clamp] ] / (E T_v).
// As ~35 torch element-wise / reduction launches and their autograd twins this was 0.33 ms of a 4.5 ms REINFORCE step.
namespace sumk {
constexpr int PL_MAX_E = 16;
constexpr float PL_EPS = 1.1920928955078125e-07f;   // torch.finfo(float32).eps: clamp_probs

__global__ __launch_bounds__(256) void policy_loss_fwd_kernel(const float* __restrict__ probs, const float* __restrict__ actions,
                                                              const float* __restrict__ rewards, const float* __restrict__ base,
                                                              const int32_t* __restrict__ off, int n_seq, int n_rows, int E,
                                                              float beta, float eps_t, float* __restrict__ lv,
                                                              float* __restrict__ meanp) {
  __shared__ float red[4][PL_MAX_E + 1];
  const int v = blockIdx.x, r0 = off[v], T = off[v + 1] - r0;
  const float invT = 1.f / (float)T;
  float l = 0.f, mp = 0.f;                  // thread 0 only
  // episodes in chunks of PL_MAX_E (the reference takes any num_episodes, dsn.py:53): the per-episode sums of a chunk live in
  // registers; for E <= PL_MAX_E this is one pass, with the summation order it always had
  for (int e0 = 0; e0 < E; e0 += PL_MAX_E) {
    float s[PL_MAX_E + 1];
#pragma unroll
    for (int e = 0; e <= PL_MAX_E; ++e) s[e] = 0.f;
    for (int t = threadIdx.x; t < T; t += 256) {
      const float p = probs[r0 + t];
      const float pc = fminf(fmaxf(p, PL_EPS), 1.f - PL_EPS);
      const float lp = logf(pc), lq = log1pf(-pc);
      s[PL_MAX_E] += p;
#pragma unroll
      for (int e = 0; e < PL_MAX_E; ++e)
        if (e0 + e < E) { const float a = actions[(int64_t)(e0 + e) * n_rows + r0 + t]; s[e] += a * lp + (1.f - a) * lq; }
    }
#pragma unroll
    for (int e = 0; e <= PL_MAX_E; ++e) {
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) s[e] += __shfl_xor(s[e], m, 64);
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][e] = s[e];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      if (e0 == 0) {
        mp = ((red[0][PL_MAX_E] + red[1][PL_MAX_E]) + (red[2][PL_MAX_E] + red[3][PL_MAX_E])) * invT;
        l = beta * (mp - eps_t) * (mp - eps_t);
      }
      for (int e = 0; e < PL_MAX_E && e0 + e < E; ++e) {
        const float lpm = ((red[0][e] + red[1][e]) + (red[2][e] + red[3][e])) * invT;
        l -= lpm * (rewards[(int64_t)(e0 + e) * n_seq + v] - base[v]);
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    lv[v] = l / (float)E;
    meanp[v] = mp;
  }
}

__global__ __launch_bounds__(256) void policy_loss_bwd_kernel(const float* __restrict__ probs, const float* __restrict__ actions,
                                                              const float* __restrict__ rewards, const float* __restrict__ base,
                                                              const float* __restrict__ meanp, const float* __restrict__ dlv,
                                                              const int32_t* __restrict__ off, int n_seq, int n_rows, int E,
                                                              float beta, float eps_t, float* __restrict__ dprobs) {
  extern __shared__ float adv[];            // E advantages (dynamic: any number of episodes)
  const int v = blockIdx.x, r0 = off[v], T = off[v + 1] - r0;
  for (int e = threadIdx.x; e < E; e += 256) adv[e] = rewards[(int64_t)e * n_seq + v] - base[v];
  __syncthreads();
  const float scale = dlv[v] / ((float)E * (float)T);
  const float g0 = 2.f * beta * (meanp[v] - eps_t);
  for (int t = threadIdx.x; t < T; t += 256) {
    const float p = probs[r0 + t];
    const float pc = fminf(fmaxf(p, PL_EPS), 1.f - PL_EPS);
    const bool inside = p >= PL_EPS && p <= 1.f - PL_EPS;     // torch.clamp passes the gradient on its closed range
    float g = g0;
    if (inside) {
      const float inv = 1.f / (pc * (1.f - pc));
      float acc = 0.f;
      for (int e = 0; e < E; ++e) acc += adv[e] * (actions[(int64_t)e * n_rows + r0 + t] - pc);
      g -= acc * inv;
    }
    dprobs[r0 + t] = g * scale;
  }
}
}  // namespace sumk

extern "C" int sumk_dsn_policy_loss_forward(const float* probs, const float* actions, const float* rewards, const float* base,
                                            int32_t n_seq, int32_t n_rows, const int32_t* seq_off_dev, int32_t n_episodes,
                                            float beta, float eps_target, float* loss_per_video, float* mean_probs,
                                            void* stream) {
  using namespace sumk;
  SUMK_ARG(probs && actions && rewards && base && seq_off_dev && loss_per_video && mean_probs, "policy_loss_forward: null pointer");
  SUMK_ARG(n_seq > 0 && n_rows > 0 && n_episodes > 0, "policy_loss_forward: n_seq=%d n_rows=%d episodes=%d", n_seq, n_rows, n_episodes);
  hipLaunchKernelGGL(policy_loss_fwd_kernel, dim3(n_seq), dim3(256), 0, (hipStream_t)stream, probs, actions, rewards, base, seq_off_dev,
                     n_seq, n_rows, n_episodes, beta, eps_target, loss_per_video, mean_probs);
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}

extern "C" int sumk_dsn_policy_loss_backward(const float* probs, const float* actions, const float* rewards, const float* base,
                                             const float* mean_probs, const float* dloss_per_video, int32_t n_seq, int32_t n_rows,
                                             const int32_t* seq_off_dev, int32_t n_episodes, float beta, float eps_target,
                                             float* dprobs, void* stream) {
  using namespace sumk;
  SUMK_ARG(probs && actions && rewards && base && mean_probs && dloss_per_video && seq_off_dev && dprobs, "policy_loss_backward: null pointer");
  SUMK_ARG(n_seq > 0 && n_rows > 0 && n_episodes > 0 && n_episodes <= 8192, "policy_loss_backward: n_seq=%d n_rows=%d episodes=%d (max 8192)",
           n_seq, n_rows, n_episodes);
  hipLaunchKernelGGL(policy_loss_bwd_kernel, dim3(n_seq), dim3(256), (size_t)n_episodes * sizeof(float), (hipStream_t)stream, probs, actions, rewards, base, mean_probs,
                     dloss_per_video, seq_off_dev, n_seq, n_rows, n_episodes, beta, eps_target, dprobs);
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}

// ------------------------------------------------------------------------------------------- per-video MSE
// nn.MSELoss of every video of a packed batch (vasnet.py:209, transformer.py:161): mse[v] = mean_t (s_t - y_t)^2 and
// ds_t = 2 (s_t - y_t) / T_v * dmse[v].  One block per video; replaces sub / pow / index_add / div and their autograd twins.
namespace sumk {
__global__ __launch_bounds__(256) void segment_mse_fwd_kernel(const float* __restrict__ s, const float* __restrict__ y,
                                                              const int32_t* __restrict__ off, float* __restrict__ mse) {
  __shared__ float red[4];
  const int v = blockIdx.x, r0 = off[v], T = off[v + 1] - r0;
  float acc = 0.f;
  for (int t = threadIdx.x; t < T; t += 256) { const float d = s[r0 + t] - y[r0 + t]; acc += d * d; }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) mse[v] = ((red[0] + red[1]) + (red[2] + red[3])) / (float)T;
}
__global__ __launch_bounds__(256) void segment_mse_bwd_kernel(const float* __restrict__ s, const float* __restrict__ y,
                                                              const float* __restrict__ dmse, const int32_t* __restrict__ off,
                                                              float* __restrict__ ds) {
  const int v = blockIdx.x, r0 = off[v], T = off[v + 1] - r0;
  const float c = 2.f * dmse[v] / (float)T;
  for (int t = threadIdx.x; t < T; t += 256) ds[r0 + t] = c * (s[r0 + t] - y[r0 + t]);
}
// The trainers' loss in one launch: loss = scale * sum_v mse[v] (scale = 1 / n_videos: the mean over the videos of a step).  A block per
// video (as segment_mse_fwd_kernel); the block that draws the last ticket adds the per-video values in video order (deterministic) and
// leaves the ticket word zero for the next launch.  Replaces the per-video kernel + torch's mean and, in the backward pass, the expand /
// fill kernels of its autograd twin: five ~5 us launches per optimiser step become two.  (First version: ONE block, a wave per video in
// turn -- 50 videos were four dependent rounds of three round trips, 12.8-15 us.)
__global__ __launch_bounds__(256) void segment_mse_mean_fwd_kernel(const float* __restrict__ s, const float* __restrict__ y,
                                                                   const int32_t* __restrict__ off, int n_seq, float scale,
                                                                   float* __restrict__ mse, float* __restrict__ loss, unsigned* __restrict__ ticket) {
  __shared__ float red[4];
  __shared__ int last;
  const int v = blockIdx.x, r0 = off[v], T = off[v + 1] - r0;
  float acc = 0.f;
  for (int t = threadIdx.x; t < T; t += 256) { const float d = s[r0 + t] - y[r0 + t]; acc += d * d; }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_store(mse + v, ((red[0] + red[1]) + (red[2] + red[3])) / (float)T, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();                                           // the value is out before the ticket is drawn
    last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(n_seq - 1);
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  // the last arriver: every video's value is in memory; 256 loads in flight at a time (through LDS), added in video order by one thread
  __shared__ float vals[256];
  float part = 0.f;
  for (int q0 = 0; q0 < n_seq; q0 += 256) {
    const int q = q0 + threadIdx.x;
    vals[threadIdx.x] = q < n_seq ? __hip_atomic_load(mse + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
    __syncthreads();
    if (threadIdx.x == 0) {
      const int n = min(256, n_seq - q0);
      for (int i = 0; i < n; ++i) part += vals[i];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    loss[0] = part * scale;
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__global__ __launch_bounds__(256) void segment_mse_mean_bwd_kernel(const float* __restrict__ s, const float* __restrict__ y,
                                                                   const float* __restrict__ dloss, float scale,
                                                                   const int32_t* __restrict__ off, float* __restrict__ ds) {
  const int v = blockIdx.x, r0 = off[v], T = off[v + 1] - r0;
  const float c = 2.f * (dloss[0] * scale) / (float)T;
  for (int t = threadIdx.x; t < T; t += 256) ds[r0 + t] = c * (s[r0 + t] - y[r0 + t]);
}
}  // namespace sumk

extern "C" int sumk_segment_mse_mean_forward(const float* scores, const float* target, int32_t n_seq, const int32_t* seq_off_dev,
                                             float scale, float* mse_per_video, float* loss, uint32_t* ticket, void* stream) {
  SUMK_ARG(scores && target && seq_off_dev && mse_per_video && loss && ticket && n_seq > 0, "segment_mse_mean_forward: bad argument");
  hipLaunchKernelGGL(sumk::segment_mse_mean_fwd_kernel, dim3(n_seq), dim3(256), 0, (hipStream_t)stream, scores, target, seq_off_dev, n_seq, scale,
                     mse_per_video, loss, ticket);
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}
extern "C" int sumk_segment_mse_mean_backward(const float* scores, const float* target, const float* dloss, float scale, int32_t n_seq,
                                              const int32_t* seq_off_dev, float* dscores, void* stream) {
  SUMK_ARG(scores && target && dloss && seq_off_dev && dscores && n_seq > 0, "segment_mse_mean_backward: bad argument");
  hipLaunchKernelGGL(sumk::segment_mse_mean_bwd_kernel, dim3(n_seq), dim3(256), 0, (hipStream_t)stream, scores, target, dloss, scale,
                     seq_off_dev, dscores);
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}

extern "C" int sumk_segment_mse_forward(const float* scores, const float* target, int32_t n_seq, const int32_t* seq_off_dev,
                                        float* mse_per_video, void* stream) {
  SUMK_ARG(scores && target && seq_off_dev && mse_per_video && n_seq > 0, "segment_mse_forward: bad argument");
  hipLaunchKernelGGL(sumk::segment_mse_fwd_kernel, dim3(n_seq), dim3(256), 0, (hipStream_t)stream, scores, target, seq_off_dev, mse_per_video);
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}
extern "C" int sumk_segment_mse_backward(const float* scores, const float* target, const float* dmse_per_video, int32_t n_seq,
                                         const int32_t* seq_off_dev, float* dscores, void* stream) {
  SUMK_ARG(scores && target && dmse_per_video && seq_off_dev && dscores && n_seq > 0, "segment_mse_backward: bad argument");
  hipLaunchKernelGGL(sumk::segment_mse_bwd_kernel, dim3(n_seq), dim3(256), 0, (hipStream_t)stream, scores, target, dmse_per_video,
                     seq_off_dev, dscores);
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}
