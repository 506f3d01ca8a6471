// Transformer-encoder scorer (inference) for a packed batch of videos on gfx950.
// Reference: summarizer/models/transformer.py:74-103 -- stock nn.TransformerEncoder (post-norm layers: multi-head
// self-attention, residual, LayerNorm, ReLU feed-forward, residual, LayerNorm), final norm = the SHARED layer_norm,
// optional extra residual, then k1 + ReLU + the same layer_norm + k2 + sigmoid.
//
// Built from the kernels of the VASNet path: every projection is the fp32 MFMA GEMM (bias / bias+ReLU / bias+residual fused
// in its epilogue); per-(video, head) logits and context are grouped GEMMs over a device table with one sub-problem per
// (video, head) that address Q/K/V head slices in place (ld = 3D) and write the context straight into the concatenated
// (frames, D) layout; softmax is one wave per (query row, head) with shuffle reductions.
#include "sumk_internal.h"
#include <math.h>
#include <algorithm>

namespace sumk {

struct TfSeq { int64_t eoff; int32_t row0, T, ldE, pad_; };   // eoff: offset of this video's [heads][T][ldE] logits block

struct TfWs { size_t qkv, e, ctx, h0, h1, h2, t1, ff, seq, prob_row, prob_s, prob_pv, total; int64_t e_elems; int32_t n_rows; };

static int tf_carve(int D, int F, int heads, int n_seq, const int32_t* off, TfWs* w) {
  SUMK_ARG(D > 0 && D % 4 == 0 && F > 0 && F % 4 == 0, "transformer: D=%d / F=%d must be positive multiples of 4", D, F);
  SUMK_ARG(heads > 0 && D % heads == 0 && (D / heads) % 4 == 0, "transformer: head dim must be a multiple of 4 (D=%d heads=%d)", D, heads);
  SUMK_ARG(n_seq > 0 && off && off[0] == 0, "transformer: empty batch / seq_off[0] != 0");
  int64_t e = 0;
  for (int s = 0; s < n_seq; ++s) {
    int T = off[s + 1] - off[s];
    SUMK_ARG(T > 0, "transformer: video %d has %d frames", s, T);
    e += (int64_t)heads * T * ((T + 3) & ~3);
  }
  const size_t R = (size_t)off[n_seq];
  size_t p = 0;
  auto take = [&](size_t bytes) { size_t at = p; p += align_up(bytes, 256); return at; };
  w->n_rows = (int32_t)R; w->e_elems = e;
  w->qkv = take(R * 3 * D * 4); w->e = take((size_t)e * 4); w->ctx = take(R * D * 4);
  w->h0 = take(R * D * 4); w->h1 = take(R * D * 4); w->h2 = take(R * D * 4); w->t1 = take(R * D * 4); w->ff = take(R * (size_t)F * 4);
  w->seq = take((size_t)n_seq * sizeof(TfSeq));
  w->prob_row = take(8 * sizeof(GemmProb));
  w->prob_s = take((size_t)n_seq * heads * sizeof(GemmProb));
  w->prob_pv = take((size_t)n_seq * heads * sizeof(GemmProb));
  w->total = p;
  return SUMK_OK;
}

__global__ void tf_setup_kernel(const int32_t* off, int n_seq, int D, int heads, TfSeq* seq, GemmProb* ps, GemmProb* ppv) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_seq) return;
  const int dh = D / heads, bt = 64, tnv = (dh + bt - 1) / bt;
  int64_t eoff = 0; int ts = 0, tpv = 0;
  for (int q = 0; q < s; ++q) {
    int T = off[q + 1] - off[q], tm = (T + bt - 1) / bt;
    eoff += (int64_t)heads * T * ((T + 3) & ~3); ts += heads * tm * tm; tpv += heads * tm * tnv;
  }
  const int row0 = off[s], T = off[s + 1] - row0, ldE = (T + 3) & ~3, tm = (T + bt - 1) / bt;
  TfSeq si; si.eoff = eoff; si.row0 = row0; si.T = T; si.ldE = ldE; si.pad_ = 0;
  seq[s] = si;
  for (int h = 0; h < heads; ++h) {
    const int64_t q0 = (int64_t)row0 * 3 * D + h * dh, eb = eoff + (int64_t)h * T * ldE;
    GemmProb a;   // E = Q_h K_h^T   (NT, K = dh)
    a.a_off = q0; a.b_off = q0 + D; a.c_off = eb; a.r_off = 0;
    a.M = T; a.N = T; a.K = dh; a.lda = 3 * D; a.ldb = 3 * D; a.ldc = ldE; a.ldr = 0;
    a.tile_start = ts + h * tm * tm; a.tiles_n = tm;
    for (int i = 0; i < 7; ++i) a.pad_[i] = 0;
    ps[s * heads + h] = a;
    GemmProb b;   // C_h = alpha_h V_h  (NN, K = T) written into columns [h*dh, (h+1)*dh) of the (R, D) context
    b.a_off = eb; b.b_off = q0 + 2 * D; b.c_off = (int64_t)row0 * D + h * dh; b.r_off = 0;
    b.M = T; b.N = dh; b.K = T; b.lda = ldE; b.ldb = 3 * D; b.ldc = D; b.ldr = 0;
    b.tile_start = tpv + h * tm * tnv; b.tiles_n = tnv;
    for (int i = 0; i < 7; ++i) b.pad_[i] = 0;
    ppv[s * heads + h] = b;
  }
}

// one wave per (query row, head): alpha = softmax(scale * logits); pad columns zeroed
__global__ __launch_bounds__(256) void tf_softmax_kernel(float* E, const TfSeq* seq, const int32_t* off, int n_seq, int n_rows,
                                                         int heads, float scale) {
  const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= (int64_t)n_rows * heads) return;
  const int lane = threadIdx.x & 63;
  const int row = (int)(wid / heads), h = (int)(wid % heads);
  int lo = 0, hi = n_seq - 1;
  while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (off[mid] <= row) lo = mid; else hi = mid - 1; }
  const TfSeq si = seq[lo];
  const int i = row - si.row0, T = si.T;
  float* e = E + si.eoff + ((int64_t)h * T + i) * si.ldE;
  float m = -INFINITY;
  for (int j = lane; j < T; j += 64) m = fmaxf(m, e[j] * scale);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  float sum = 0.f;
  for (int j = lane; j < T; j += 64) sum += expf(e[j] * scale - m);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  for (int j = lane; j < si.ldE; j += 64) e[j] = j < T ? expf(e[j] * scale - m) / sum : 0.f;
}

__global__ void add_rows_kernel(float* __restrict__ y, const float* __restrict__ x, int64_t n4) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  float4 a = reinterpret_cast<float4*>(y)[i], b = reinterpret_cast<const float4*>(x)[i];
  a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  reinterpret_cast<float4*>(y)[i] = a;
}

}  // namespace sumk

using namespace sumk;

extern "C" size_t sumk_transformer_workspace_bytes(int32_t D, int32_t F, int32_t n_heads, int32_t n_seq,
                                                   const int32_t* seq_off_host) {
  TfWs w;
  if (tf_carve(D, F, n_heads, n_seq, seq_off_host, &w) != SUMK_OK) return 0;
  return w.total;
}

extern "C" int sumk_transformer_forward(float* x, int32_t D, int32_t F, int32_t n_heads, int32_t n_layers, int32_t n_seq,
                                        const int32_t* seq_off_host, const int32_t* seq_off_dev,
                                        const sumk_tf_layer_weights* layers, const sumk_tf_head_weights* head,
                                        float layer_eps, float final_eps, int32_t more_residuals, const float* pos_table,
                                        const int32_t* pos_rows, float* scores, void* workspace, size_t workspace_bytes,
                                        void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SUMK_ARG(x && seq_off_dev && layers && head && scores && workspace, "transformer_forward: null pointer");
  SUMK_ARG(n_layers > 0, "transformer_forward: n_layers=%d", n_layers);
  SUMK_ARG((pos_table == nullptr) == (pos_rows == nullptr), "transformer_forward: pos_table and pos_rows go together");
  SUMK_ARG(head->ln_w && head->ln_b && head->k1_w && head->k1_b && head->k2_w && head->k2_b, "transformer_forward: null head weight");
  TfWs L;
  SUMK_TRY(tf_carve(D, F, n_heads, n_seq, seq_off_host, &L));
  if (workspace_bytes < L.total) { set_error("transformer_forward: workspace %zu < required %zu", workspace_bytes, L.total); return SUMK_ERR_WORKSPACE; }
  char* ws = (char*)workspace;
  const int R = L.n_rows, dh = D / n_heads;
  float* QKV = (float*)(ws + L.qkv); float* E = (float*)(ws + L.e); float* CTX = (float*)(ws + L.ctx);
  float* Hb[3] = {(float*)(ws + L.h0), (float*)(ws + L.h1), (float*)(ws + L.h2)};
  float* T1 = (float*)(ws + L.t1); float* FF = (float*)(ws + L.ff);
  TfSeq* seq = (TfSeq*)(ws + L.seq);
  GemmProb* prow = (GemmProb*)(ws + L.prob_row);
  GemmProb* ps = (GemmProb*)(ws + L.prob_s); GemmProb* ppv = (GemmProb*)(ws + L.prob_pv);

  if (pos_table) SUMK_TRY(launch_add_pos(x, pos_table, pos_rows, R, D, stream));
  hipLaunchKernelGGL(tf_setup_kernel, dim3((n_seq + 63) / 64), dim3(64), 0, stream, seq_off_dev, n_seq, D, n_heads, seq, ps, ppv);
  int tiles_s = 0, tiles_pv = 0;
  for (int s = 0; s < n_seq; ++s) {
    int T = seq_off_host[s + 1] - seq_off_host[s], tm = (T + 63) / 64;
    tiles_s += n_heads * tm * tm; tiles_pv += n_heads * tm * ((dh + 63) / 64);
  }
  auto cfg = [&](int M, int N) { return gemm_tiles(M, N, 0) >= 512 ? 0 : 1; };
  const int c_qkv = cfg(R, 3 * D), c_dd = cfg(R, D), c_df = cfg(R, F);
  enum { P_QKV = 0, P_DD = 1, P_DF = 2, P_FD = 3 };
  SUMK_TRY(fill_single_prob(prow + P_QKV, R, 3 * D, D, D, D, 3 * D, 0, c_qkv, stream));
  SUMK_TRY(fill_single_prob(prow + P_DD, R, D, D, D, D, D, D, c_dd, stream));      // (R,D) <- (R,D) x (D,D)^T (+ residual ld D)
  SUMK_TRY(fill_single_prob(prow + P_DF, R, F, D, D, D, F, 0, c_df, stream));      // (R,F) <- (R,D) x (F,D)^T
  SUMK_TRY(fill_single_prob(prow + P_FD, R, D, F, F, F, D, D, c_dd, stream));      // (R,D) <- (R,F) x (D,F)^T (+ residual)

  const float* hin = x;
  for (int l = 0; l < n_layers; ++l) {
    const sumk_tf_layer_weights& W = layers[l];
    SUMK_ARG(W.in_proj_w && W.in_proj_b && W.out_proj_w && W.out_proj_b && W.lin1_w && W.lin1_b && W.lin2_w && W.lin2_b &&
             W.norm1_w && W.norm1_b && W.norm2_w && W.norm2_b, "transformer_forward: null weight in layer %d", l);
    float* hmid = Hb[0]; float* hout = Hb[1 + (l & 1)];   // never aliases this layer's input (x or the previous hout)
    {  // packed in-projection  [Q|K|V] = h Win^T + bin
      GemmLaunch g;
      g.A = hin; g.B[0] = W.in_proj_w; g.bias0[0] = W.in_proj_b; g.C = QKV; g.probs = prow + P_QKV; g.small_tile = c_qkv;
      g.total_tiles = gemm_tiles(R, 3 * D, c_qkv); g.xcd_M = R; g.xcd_N = 3 * D;
      SUMK_TRY(launch_gemm(GEMM_NT, EPI_BIAS2, g, stream));
    }
    {  // logits per (video, head)
      GemmLaunch g;
      g.A = QKV; g.B[0] = QKV; g.C = E; g.probs = ps; g.nprob = n_seq * n_heads; g.small_tile = 1; g.total_tiles = tiles_s;
      SUMK_TRY(launch_gemm(GEMM_NT, EPI_NONE, g, stream));
    }
    hipLaunchKernelGGL(tf_softmax_kernel, dim3((unsigned)(((int64_t)R * n_heads + 3) / 4)), dim3(256), 0, stream, E, seq,
                       seq_off_dev, n_seq, R, n_heads, 1.0f / sqrtf((float)dh));
    {  // context, heads written side by side
      GemmLaunch g;
      g.A = E; g.B[0] = QKV; g.C = CTX; g.probs = ppv; g.nprob = n_seq * n_heads; g.small_tile = 1; g.total_tiles = tiles_pv;
      SUMK_TRY(launch_gemm(GEMM_NN, EPI_NONE, g, stream));
    }
    {  // out-projection + bias + residual
      GemmLaunch g;
      g.A = CTX; g.B[0] = W.out_proj_w; g.bias0[0] = W.out_proj_b; g.R = hin; g.C = T1; g.probs = prow + P_DD; g.small_tile = c_dd;
      g.total_tiles = gemm_tiles(R, D, c_dd); g.xcd_M = R; g.xcd_N = D;
      SUMK_TRY(launch_gemm(GEMM_NT, EPI_BIAS_RESIDUAL, g, stream));
    }
    SUMK_TRY(launch_layernorm(T1, hmid, W.norm1_w, W.norm1_b, R, D, layer_eps, nullptr, stream));
    {  // feed-forward 1: bias + ReLU
      GemmLaunch g;
      g.A = hmid; g.B[0] = W.lin1_w; g.bias0[0] = W.lin1_b; g.C = FF; g.probs = prow + P_DF; g.small_tile = c_df;
      g.total_tiles = gemm_tiles(R, F, c_df); g.xcd_M = R; g.xcd_N = F;
      SUMK_TRY(launch_gemm(GEMM_NT, EPI_BIAS_RELU, g, stream));
    }
    {  // feed-forward 2: bias + residual
      GemmLaunch g;
      g.A = FF; g.B[0] = W.lin2_w; g.bias0[0] = W.lin2_b; g.R = hmid; g.C = T1; g.probs = prow + P_FD; g.small_tile = c_dd;
      g.total_tiles = gemm_tiles(R, D, c_dd); g.xcd_M = R; g.xcd_N = D;
      SUMK_TRY(launch_gemm(GEMM_NT, EPI_BIAS_RESIDUAL, g, stream));
    }
    SUMK_TRY(launch_layernorm(T1, hout, W.norm2_w, W.norm2_b, R, D, layer_eps, nullptr, stream));
    hin = hout;
  }
  // final (shared) LayerNorm of the encoder, optional extra residual, scoring head
  float* hfin = Hb[0];
  SUMK_TRY(launch_layernorm(hin, hfin, head->ln_w, head->ln_b, R, D, final_eps, nullptr, stream));
  if (more_residuals) {
    int64_t n4 = (int64_t)R * (D >> 2);
    hipLaunchKernelGGL(add_rows_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, hfin, x, n4);
  }
  {
    GemmLaunch g;
    g.A = hfin; g.B[0] = head->k1_w; g.bias0[0] = head->k1_b; g.C = T1; g.probs = prow + P_DD; g.small_tile = c_dd;
    g.total_tiles = gemm_tiles(R, D, c_dd); g.xcd_M = R; g.xcd_N = D;
    SUMK_TRY(launch_gemm(GEMM_NT, EPI_BIAS_RELU, g, stream));
  }
  SUMK_TRY(launch_ln_head(T1, head->ln_w, head->ln_b, head->k2_w, head->k2_b, scores, R, D, final_eps, stream));
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}
