// VASNet scorer for a packed batch of videos on gfx950 (reference: summarizer/models/vasnet.py:92-148).
//
// Pipeline (fp32 throughout; every dense contraction is the MFMA GEMM of gemm_f32.hip):
//   0 pos-embed add (in place, optional)                         vasnet.py:106-112
//   1 [Q|K|V] = X . [Wq;Wk;Wv]^T           one grouped-B NT GEMM   vasnet.py:114-116
//   2 E_s = Q_s . K_s^T  per video (ragged, 64x64 tiles)          vasnet.py:118
//   3 alpha = softmax(mask(E*scale))  one wave per row, shuffles  vasnet.py:119-129
//   4 C_s = alpha_s . V_s  per video                              vasnet.py:131
//   5 Y0 = C . Wo^T + X     (residual fused in the epilogue)      vasnet.py:132-135
//   6 Y1 = LayerNorm(Y0)    one wave per row                      vasnet.py:137
//   7 Z  = relu(Y1 . W1^T + b1)  (bias+ReLU fused in epilogue)    vasnet.py:140-141
//   8 s  = sigmoid(LayerNorm(Z) . w2 + b2)  SAME LayerNorm, fused with the head   vasnet.py:143-145
// Logits are materialised: with a single head of width D the products have D/6 >= 170 FLOP per byte of E
// traffic, far above the fp32 ridge (~20 FLOP/B), so a flash-style kernel would buy nothing.
#include "sumk_internal.h"
#include <math.h>

namespace sumk {

struct SeqInfo {
  int64_t eoff;  // element offset of this video's (T x ldE) logits block in E
  int32_t row0, T, ldE, pad_;
};

// Workspace carve-up, computed identically by the size query and by forward/backward.
struct VasnetWs {
  size_t qkv, e, ctx, y0, y1, z, seq, prob_row, prob_s, prob_pv, stats, total;
  // training-only buffers
  size_t dz, dy1, dy0, dctx, dqkv, de, rowtmp;
  int64_t e_elems;
  int32_t n_rows;
};

static inline int round4(int v) { return (v + 3) & ~3; }

static int carve(int D, int n_seq, const int32_t* off, int training, VasnetWs* w) {
  SUMK_ARG(D > 0 && D % 4 == 0, "vasnet: D=%d must be a positive multiple of 4", D);
  SUMK_ARG(n_seq > 0 && off != nullptr, "vasnet: empty batch");
  SUMK_ARG(off[0] == 0, "vasnet: seq_off[0] must be 0");
  int64_t e = 0;
  for (int s = 0; s < n_seq; ++s) {
    int T = off[s + 1] - off[s];
    SUMK_ARG(T > 0, "vasnet: video %d has %d frames", s, T);
    e += (int64_t)T * round4(T);
  }
  const size_t R = (size_t)off[n_seq];
  size_t p = 0;
  auto take = [&](size_t bytes) { size_t at = p; p += align_up(bytes, 256); return at; };
  w->n_rows = (int32_t)R; w->e_elems = e;
  w->qkv = take(R * 3 * D * 4);
  w->e = take((size_t)e * 4);
  w->ctx = take(R * D * 4);
  w->y0 = take(R * D * 4);
  w->y1 = take(R * D * 4);
  w->z = take(R * D * 4);
  w->seq = take((size_t)n_seq * sizeof(SeqInfo));
  w->prob_row = take(8 * sizeof(GemmProb));
  w->prob_s = take((size_t)n_seq * sizeof(GemmProb));
  w->prob_pv = take((size_t)n_seq * sizeof(GemmProb));
  w->stats = take(R * 4 * 4);  // mean/rstd of both LayerNorm applications (training)
  w->dz = w->dy1 = w->dy0 = w->dctx = w->dqkv = w->de = w->rowtmp = 0;
  if (training) {
    w->dz = take(R * D * 4);
    w->dy1 = take(R * D * 4);
    w->dy0 = take(R * D * 4);
    w->dctx = take(R * D * 4);
    w->dqkv = take(R * 3 * D * 4);
    w->de = take((size_t)e * 4);
    w->rowtmp = take(R * 4 * 4);
  }
  w->total = p;
  return SUMK_OK;
}

// ------------------------------------------------------------------------------------------- setup tables
__global__ void vasnet_setup_kernel(const int32_t* off, int n_seq, int D, SeqInfo* seq, GemmProb* ps, GemmProb* ppv,
                                    int bt) {
  // serial prefix sums: n_seq is a few tens to a few thousands, this runs once per call in ~microseconds
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int64_t eoff = 0;
  int ts = 0, tpv = 0;
  for (int s = 0; s < n_seq; ++s) {
    int row0 = off[s], T = off[s + 1] - off[s], ldE = (T + 3) & ~3;
    SeqInfo si; si.eoff = eoff; si.row0 = row0; si.T = T; si.ldE = ldE; si.pad_ = 0;
    seq[s] = si;
    int tm = (T + bt - 1) / bt;
    GemmProb a;  // E_s = Q_s K_s^T
    a.a_off = (int64_t)row0 * 3 * D; a.b_off = (int64_t)row0 * 3 * D + D; a.c_off = eoff; a.r_off = 0;
    a.M = T; a.N = T; a.K = D; a.lda = 3 * D; a.ldb = 3 * D; a.ldc = ldE; a.ldr = 0;
    a.tile_start = ts; a.tiles_n = tm;
    for (int i = 0; i < 7; ++i) a.pad_[i] = 0;
    ps[s] = a; ts += tm * tm;
    GemmProb b;  // C_s = alpha_s V_s
    int tn = (D + bt - 1) / bt;
    b.a_off = eoff; b.b_off = (int64_t)row0 * 3 * D + 2 * D; b.c_off = (int64_t)row0 * D; b.r_off = 0;
    b.M = T; b.N = D; b.K = T; b.lda = ldE; b.ldb = 3 * D; b.ldc = D; b.ldr = 0;
    b.tile_start = tpv; b.tiles_n = tn;
    for (int i = 0; i < 7; ++i) b.pad_[i] = 0;
    ppv[s] = b; tpv += tm * tn;
    eoff += (int64_t)T * ldE;
  }
}

// ------------------------------------------------------------------------------------------- wave helpers
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

__device__ __forceinline__ int find_seq(const int32_t* off, int n_seq, int row) {
  int lo = 0, hi = n_seq - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (off[mid] <= row) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// Masked, scaled logit exactly as vasnet.py:119-127 produces it.
__device__ __forceinline__ float masked_logit(float raw, float scale, int i, int j, int ignore_self, int aperture) {
  float e = raw * scale;
  if (ignore_self && i == j) e = -INFINITY;
  if (aperture >= 0) {
    // scope = tril(e, w) * triu(e, -w);  e[scope == 0] = -inf   (also masks in-band logits whose square underflows)
    float lo = (j - i <= aperture) ? e : 0.f;
    float up = (j - i >= -aperture) ? e : 0.f;
    if (lo * up == 0.f) e = -INFINITY;
  }
  return e;
}

// ------------------------------------------------------------------------------------------- softmax rows
// One wave per query row.  Reads raw Q.K^T, writes alpha in place and zeroes the [T, ldE) pad so the
// alpha.V product can stream K in float4 units.
__global__ __launch_bounds__(256) void vasnet_softmax_kernel(float* E, const SeqInfo* seq, const int32_t* off,
                                                             int n_seq, int n_rows, float scale, int ignore_self,
                                                             int aperture) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const int lane = threadIdx.x & 63;
  const int s = find_seq(off, n_seq, row);
  const SeqInfo si = seq[s];
  const int i = row - si.row0, T = si.T;
  float* e = E + si.eoff + (int64_t)i * si.ldE;
  float m = -INFINITY;
  for (int j = lane; j < T; j += 64) m = fmaxf(m, masked_logit(e[j], scale, i, j, ignore_self, aperture));
  m = wave_max(m);
  float sum = 0.f;
  for (int j = lane; j < T; j += 64) sum += expf(masked_logit(e[j], scale, i, j, ignore_self, aperture) - m);
  sum = wave_sum(sum);
  for (int j = lane; j < si.ldE; j += 64) {
    float v = 0.f;
    if (j < T) v = expf(masked_logit(e[j], scale, i, j, ignore_self, aperture) - m) / sum;
    e[j] = v;
  }
}

// ------------------------------------------------------------------------------------------- LayerNorm rows
// y = (x - mean) * rstd * g + b over D, biased variance, one wave per row (torch.nn.LayerNorm, vasnet.py:54).
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ X, float* __restrict__ Y,
                                                        const float* __restrict__ g, const float* __restrict__ b,
                                                        int n_rows, int D, float eps, float* __restrict__ stats) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const int lane = threadIdx.x & 63;
  const float4* x4 = reinterpret_cast<const float4*>(X + (int64_t)row * D);
  const int D4 = D >> 2;
  float s = 0.f;
  for (int c = lane; c < D4; c += 64) { float4 v = x4[c]; s += (v.x + v.y) + (v.z + v.w); }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
  for (int c = lane; c < D4; c += 64) {
    float4 v = x4[c];
    float a = v.x - mean, bb = v.y - mean, cc = v.z - mean, d = v.w - mean;
    q += (a * a + bb * bb) + (cc * cc + d * d);
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
  float4* y4 = reinterpret_cast<float4*>(Y + (int64_t)row * D);
  const float4* g4 = reinterpret_cast<const float4*>(g);
  const float4* b4 = reinterpret_cast<const float4*>(b);
  for (int c = lane; c < D4; c += 64) {
    float4 v = x4[c], gg = g4[c], bv = b4[c], o;
    o.x = (v.x - mean) * rstd * gg.x + bv.x; o.y = (v.y - mean) * rstd * gg.y + bv.y;
    o.z = (v.z - mean) * rstd * gg.z + bv.z; o.w = (v.w - mean) * rstd * gg.w + bv.w;
    y4[c] = o;
  }
  if (stats != nullptr && lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
}

// scores[r] = sigmoid( LayerNorm(Z[r]) . w2 + b2 )        vasnet.py:143-145
__global__ __launch_bounds__(256) void ln_head_kernel(const float* __restrict__ Z, const float* __restrict__ g,
                                                      const float* __restrict__ b, const float* __restrict__ w2,
                                                      const float* __restrict__ b2, float* __restrict__ scores,
                                                      int n_rows, int D, float eps, float* __restrict__ stats) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const int lane = threadIdx.x & 63;
  const float4* x4 = reinterpret_cast<const float4*>(Z + (int64_t)row * D);
  const int D4 = D >> 2;
  float s = 0.f;
  for (int c = lane; c < D4; c += 64) { float4 v = x4[c]; s += (v.x + v.y) + (v.z + v.w); }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
  for (int c = lane; c < D4; c += 64) {
    float4 v = x4[c];
    float a = v.x - mean, bb = v.y - mean, cc = v.z - mean, d = v.w - mean;
    q += (a * a + bb * bb) + (cc * cc + d * d);
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
  const float4* g4 = reinterpret_cast<const float4*>(g);
  const float4* b4 = reinterpret_cast<const float4*>(b);
  const float4* w4 = reinterpret_cast<const float4*>(w2);
  float dot = 0.f;
  for (int c = lane; c < D4; c += 64) {
    float4 v = x4[c], gg = g4[c], bv = b4[c], ww = w4[c];
    dot += ((v.x - mean) * rstd * gg.x + bv.x) * ww.x + ((v.y - mean) * rstd * gg.y + bv.y) * ww.y +
           ((v.z - mean) * rstd * gg.z + bv.z) * ww.z + ((v.w - mean) * rstd * gg.w + bv.w) * ww.w;
  }
  dot = wave_sum(dot);
  if (lane == 0) {
    scores[row] = 1.0f / (1.0f + expf(-(dot + b2[0])));
    if (stats != nullptr) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
  }
}

// x[r,:] += table[pos_rows[r],:]   (in place, like vasnet.py:109/111)
__global__ void add_pos_kernel(float* x, const float* table, const int32_t* pos_rows, int n_rows, int D) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t n4 = (int64_t)n_rows * (D >> 2);
  if (idx >= n4) return;
  int r = (int)(idx / (D >> 2)), c = (int)(idx % (D >> 2));
  float4 a = reinterpret_cast<float4*>(x)[idx];
  float4 t = reinterpret_cast<const float4*>(table + (int64_t)pos_rows[r] * D)[c];
  a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
  reinterpret_cast<float4*>(x)[idx] = a;
}

static int rowwise_small_tile(int M, int N) { return gemm_tiles(M, N, 0) >= 512 ? 0 : 1; }

}  // namespace sumk

using namespace sumk;

extern "C" size_t sumk_vasnet_workspace_bytes(int32_t D, int32_t n_seq, const int32_t* seq_off_host, int32_t training) {
  VasnetWs w;
  if (carve(D, n_seq, seq_off_host, training, &w) != SUMK_OK) return 0;
  return w.total;
}

extern "C" int sumk_vasnet_forward(float* x, int32_t D, int32_t n_seq, const int32_t* seq_off_host,
                                   const int32_t* seq_off_dev, const sumk_vasnet_weights* w,
                                   const sumk_vasnet_opts* opts, const float* pos_table, const int32_t* pos_rows,
                                   float* scores, void* workspace, size_t workspace_bytes, int32_t training,
                                   void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SUMK_ARG(x && seq_off_dev && w && opts && scores && workspace, "vasnet_forward: null pointer");
  SUMK_ARG(w->Wk && w->Wq && w->Wv && w->Wo && w->W1 && w->b1 && w->w2 && w->b2 && w->ln_w && w->ln_b,
           "vasnet_forward: null weight");
  SUMK_ARG((pos_table == nullptr) == (pos_rows == nullptr), "vasnet_forward: pos_table and pos_rows go together");
  SUMK_ARG(opts->dropout_p == 0.f || training, "vasnet_forward: dropout needs training mode");
  VasnetWs L;
  SUMK_TRY(carve(D, n_seq, seq_off_host, training, &L));
  if (workspace_bytes < L.total) {
    set_error("vasnet_forward: workspace %zu < required %zu", workspace_bytes, L.total);
    return SUMK_ERR_WORKSPACE;
  }
  char* ws = (char*)workspace;
  const int R = L.n_rows;
  float* QKV = (float*)(ws + L.qkv);
  float* E = (float*)(ws + L.e);
  float* CTX = (float*)(ws + L.ctx);
  float* Y0 = (float*)(ws + L.y0);
  float* Y1 = (float*)(ws + L.y1);
  float* Z = (float*)(ws + L.z);
  SeqInfo* seq = (SeqInfo*)(ws + L.seq);
  GemmProb* prow = (GemmProb*)(ws + L.prob_row);
  GemmProb* ps = (GemmProb*)(ws + L.prob_s);
  GemmProb* ppv = (GemmProb*)(ws + L.prob_pv);
  float* stats = training ? (float*)(ws + L.stats) : nullptr;

  if (pos_table) {
    int64_t n4 = (int64_t)R * (D >> 2);
    hipLaunchKernelGGL(add_pos_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, x, pos_table, pos_rows, R, D);
  }
  hipLaunchKernelGGL(vasnet_setup_kernel, dim3(1), dim3(64), 0, stream, seq_off_dev, n_seq, D, seq, ps, ppv, 64);
  int tiles_s = 0, tiles_pv = 0;
  for (int s = 0; s < n_seq; ++s) {
    int T = seq_off_host[s + 1] - seq_off_host[s], tm = (T + 63) / 64;
    tiles_s += tm * tm; tiles_pv += tm * ((D + 63) / 64);
  }

  // 1: QKV projection
  const int st_qkv = rowwise_small_tile(R, 3 * D), st_d = rowwise_small_tile(R, D);
  SUMK_TRY(fill_single_prob(prow + 0, R, 3 * D, D, D, D, 3 * D, 0, st_qkv, stream));
  SUMK_TRY(fill_single_prob(prow + 1, R, D, D, D, D, D, D, st_d, stream));
  {
    GemmLaunch g;
    g.A = x; g.B[0] = w->Wq; g.B[1] = w->Wk; g.B[2] = w->Wv; g.n_group = D; g.C = QKV; g.probs = prow + 0;
    g.small_tile = st_qkv; g.total_tiles = gemm_tiles(R, 3 * D, st_qkv); g.prof_tag = SUMK_PROF_GEMM_QKV;
    SUMK_TRY(launch_gemm(GEMM_NT, EPI_NONE, g, stream));
  }
  // 2: logits per video
  {
    GemmLaunch g;
    g.A = QKV; g.B[0] = QKV; g.C = E; g.probs = ps; g.nprob = n_seq; g.small_tile = 1; g.total_tiles = tiles_s;
    SUMK_TRY(launch_gemm(GEMM_NT, EPI_NONE, g, stream));
  }
  // 3: softmax
  hipLaunchKernelGGL(vasnet_softmax_kernel, dim3((R + 3) / 4), dim3(256), 0, stream, E, seq, seq_off_dev, n_seq, R,
                     opts->scale, opts->ignore_self, opts->aperture);
  // 4: context
  {
    GemmLaunch g;
    g.A = E; g.B[0] = QKV; g.C = CTX; g.probs = ppv; g.nprob = n_seq; g.small_tile = 1; g.total_tiles = tiles_pv;
    SUMK_TRY(launch_gemm(GEMM_NN, EPI_NONE, g, stream));
  }
  // 5: output projection + residual
  {
    GemmLaunch g;
    g.A = CTX; g.B[0] = w->Wo; g.C = Y0; g.R = x; g.probs = prow + 1; g.small_tile = st_d;
    g.total_tiles = gemm_tiles(R, D, st_d);
    SUMK_TRY(launch_gemm(GEMM_NT, EPI_RESIDUAL, g, stream));
  }
  // 6: LayerNorm
  hipLaunchKernelGGL(layernorm_kernel, dim3((R + 3) / 4), dim3(256), 0, stream, Y0, Y1, w->ln_w, w->ln_b, R, D,
                     opts->eps, stats);
  // 7: k1 + bias + ReLU
  {
    GemmLaunch g;
    g.A = Y1; g.B[0] = w->W1; g.bias0[0] = w->b1; g.C = Z; g.probs = prow + 1; g.small_tile = st_d;
    g.total_tiles = gemm_tiles(R, D, st_d);
    SUMK_TRY(launch_gemm(GEMM_NT, EPI_BIAS_RELU, g, stream));
  }
  // 8: LayerNorm (same weights) + k2 + sigmoid
  hipLaunchKernelGGL(ln_head_kernel, dim3((R + 3) / 4), dim3(256), 0, stream, Z, w->ln_w, w->ln_b, w->w2, w->b2,
                     scores, R, D, opts->eps, stats ? stats + 2 * (size_t)R : nullptr);
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}
