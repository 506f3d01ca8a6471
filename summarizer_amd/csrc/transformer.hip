// Transformer-encoder scorer (inference) for a packed batch of videos on gfx950.
// Reference: summarizer/models/transformer.py:74-103 -- stock nn.TransformerEncoder (post-norm layers: multi-head
// self-attention, residual, LayerNorm, ReLU feed-forward, residual, LayerNorm), final norm = the SHARED layer_norm,
// optional extra residual, then k1 + ReLU + the same layer_norm + k2 + sigmoid.
//
// Built from the kernels of the VASNet path: every projection is the fp32 MFMA GEMM (bias / bias+ReLU / bias+residual fused
// in its epilogue); per-(video, head) logits and context are grouped GEMMs over a device table with one sub-problem per
// (video, head) that address Q/K/V head slices in place (ld = 3D) and write the context straight into the concatenated
// (frames, D) layout; softmax is one wave per (query row, head) with shuffle reductions.
// Inference in bf16x6 / bf16x3 with the weights' planes given (round 5): every projection on the plane GEMM (gemm_pw.hip), and -- heads of 128
// columns, videos of at most 320 frames -- the per-head attention on the multi-head form of attn_pw.hip; see "plane path" below.
#include "sumk_internal.h"
#include <math.h>
#include <algorithm>

namespace sumk {

struct TfSeq { int64_t eoff; int32_t row0, T, ldE, pad_; };   // eoff: offset of this video's [heads][T][ldE] logits block

struct TfWs {
  size_t qkv, e, ctx, h0, h1, h2, t1, ff, seq, prob_row, prob_tabs, total;
  // training: per-layer saves (offset of layer 0 + l * lay_stride) and backward scratch
  size_t lay0, lay_stride, l_qkv, l_p, l_pd, l_ctx, l_t1a, l_hmid, l_ff, l_t1b, l_hout, l_stats;
  size_t hfin, z, stats_fin, scores, g0, g1, g2, dqkv, dff, lnpart, colpart, slab, prob_sk;
  size_t slab_elems;
  int64_t e_elems; int32_t n_rows;
};
enum { TT_S = 0, TT_PV = 1, TT_DV = 2, TT_DP = 3, TT_DQ = 4, TT_DK = 5, TT_COUNT = 6 };

// Plane path (inference in bf16x6 / bf16x3; gemm_pw.hip): every projection of the stack reads bf16 planes of both operands.  The
// weights' planes are built once per weight change (sumk_transformer_wplanes_build: per layer [Win | Wo | W1 | W2], then k1), the
// activations' by split_planes (layer input, context, LayerNorm output) or by the producing epilogue (ReLU(lin1): PW_PLANES with bias +
// ReLU, never stored as fp32).  Two plane buffers behind the regular carve-up (sumk_transformer_workspace_bytes_for).
struct TfWPlanes { size_t win, wo, w1, w2, lay_stride, k1, total; };
static bool tf_wplanes_ok(int D, int F, int np) {
  return (np == 2 || np == 3) && D >= 256 && D % 256 == 0 && F >= 256 && F % 256 == 0 && pw_ok(256, 3 * (int64_t)D, D, 256, 3 * (int64_t)D, np) &&
         pw_ok(256, F, D, 256, F, np) && pw_ok(256, D, F, 256, D, np);
}
static TfWPlanes tf_wplanes_layout(int D, int F, int n_layers, int np) {
  TfWPlanes l; size_t p = 0;
  auto take = [&](size_t bytes) { size_t at = p; p += align_up(bytes, 256); return at; };
  l.win = take(pw_planes_bytes(3 * (int64_t)D, D, np)); l.wo = take(pw_planes_bytes(D, D, np));
  l.w1 = take(pw_planes_bytes(F, D, np)); l.w2 = take(pw_planes_bytes(D, F, np));
  l.lay_stride = p; l.k1 = p * (size_t)n_layers;
  l.total = l.k1 + align_up(pw_planes_bytes(D, D, np), 256);
  return l;
}
// ... and, when the batch is eligible (T <= 320, heads of 128 columns: attn_pw.hip's multi-head form), the attention on planes too: [Q | K | V] planes
// written by the in-projection's epilogue, one set of alpha planes per head, the SeqInfo table those kernels read.
struct TfPwExtra { size_t pa, pb, qp, ap, seqinfo, total; bool attn; };
static TfPwExtra tf_pw_extra(int D, int F, int64_t R, int np, size_t base, int heads, int n_seq, int t_max) {
  TfPwExtra e; size_t p = align_up(base, 256);
  auto take = [&](size_t bytes) { size_t at = p; p += align_up(bytes, 256); return at; };
  e.pa = take(pw_planes_bytes(R, std::max(D, F), np)); e.pb = take(pw_planes_bytes(R, std::max(D, F), np));
  e.attn = attn_pw_heads_ok(t_max, D, heads, R, np);
  e.qp = e.ap = e.seqinfo = 0;
  if (e.attn) {
    e.qp = take(pw_planes_bytes(R, 3 * D, np)); e.ap = take((size_t)heads * align_up(pw_alpha_bytes(R, t_max, np), 256));
    e.seqinfo = take((size_t)n_seq * sizeof(SeqInfo));
  }
  e.total = p;
  return e;
}
static int tf_t_max(int n_seq, const int32_t* off) {
  int t = 0;
  for (int s = 0; s < n_seq; ++s) t = std::max(t, off[s + 1] - off[s]);
  return t;
}
__global__ void tf_seqinfo_kernel(const int32_t* off, int n_seq, SeqInfo* out) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_seq) return;
  SeqInfo si; si.eoff = 0; si.row0 = off[s]; si.T = off[s + 1] - off[s]; si.ldE = (si.T + 3) & ~3; si.pad_ = 0; si.e16off = 0;
  out[s] = si;
}
static int tf_np(int precision) { return precision == SUMK_PRECISION_BF16X6 ? 3 : precision == SUMK_PRECISION_BF16X3 ? 2 : 0; }
static bool tf_pw_rows_ok(int D, int F, int64_t R, int np) {
  return R >= 128 && pw_ok(R, 3 * (int64_t)D, D, R, 3 * (int64_t)D, np) && pw_ok(R, F, D, R, F, np) && pw_ok(R, D, F, R, D, np);
}
constexpr int TF_SPLITK_PROBS = 64, TF_COLSUM_CHUNKS = 128;

static int tf_carve(int D, int F, int heads, int n_layers, int n_seq, const int32_t* off, int training, TfWs* w) {
  SUMK_ARG(D > 0 && D % 4 == 0 && F > 0 && F % 4 == 0, "transformer: D=%d / F=%d must be positive multiples of 4", D, F);
  SUMK_ARG(heads > 0 && D % heads == 0 && (D / heads) % 4 == 0, "transformer: head dim must be a multiple of 4 (D=%d heads=%d)", D, heads);
  SUMK_ARG(n_seq > 0 && off && off[0] == 0, "transformer: empty batch / seq_off[0] != 0");
  int64_t e = 0;
  for (int s = 0; s < n_seq; ++s) {
    int T = off[s + 1] - off[s];
    SUMK_ARG(T > 0, "transformer: video %d has %d frames", s, T);
    SUMK_ARG(T < (1 << 20), "transformer: video %d has %d frames (limit 2^20)", s, T);
    e += (int64_t)heads * T * ((T + 3) & ~3);
  }
  const size_t R = (size_t)off[n_seq];
  size_t p = 0;
  auto take = [&](size_t bytes) { size_t at = p; p += align_up(bytes, 256); return at; };
  w->n_rows = (int32_t)R; w->e_elems = e;
  w->seq = take((size_t)n_seq * sizeof(TfSeq));
  w->prob_row = take(8 * sizeof(GemmProb));
  w->prob_tabs = take((size_t)TT_COUNT * n_seq * heads * sizeof(GemmProb));
  w->t1 = take(R * D * 4);
  if (!training) {
    w->qkv = take(R * 3 * D * 4); w->e = take((size_t)e * 4); w->ctx = take(R * D * 4);
    w->h0 = take(R * D * 4); w->h1 = take(R * D * 4); w->h2 = take(R * D * 4); w->ff = take(R * (size_t)F * 4);
  } else {
    size_t q = 0;
    auto lt = [&](size_t bytes) { size_t at = q; q += align_up(bytes, 256); return at; };
    w->l_qkv = lt(R * 3 * D * 4); w->l_p = lt((size_t)e * 4); w->l_pd = lt((size_t)e * 4); w->l_ctx = lt(R * D * 4);
    w->l_t1a = lt(R * D * 4); w->l_hmid = lt(R * D * 4); w->l_ff = lt(R * (size_t)F * 4); w->l_t1b = lt(R * D * 4);
    w->l_hout = lt(R * D * 4); w->l_stats = lt(R * 4 * 4);
    w->lay_stride = q;
    w->lay0 = take(q * (size_t)n_layers);
    w->hfin = take(R * D * 4); w->z = take(R * D * 4); w->stats_fin = take(R * 4 * 4); w->scores = take(R * 4);
    w->g0 = take(R * D * 4); w->g1 = take(R * D * 4); w->g2 = take(R * D * 4);
    w->dqkv = take(R * 3 * D * 4); w->dff = take(R * (size_t)F * 4);
    w->lnpart = take((size_t)(LNB_MAX_WAVES / 4) * ln_slot_floats(D) * 4);
    w->colpart = take((size_t)TF_COLSUM_CHUNKS * 3 * D * 4 + (size_t)TF_COLSUM_CHUNKS * F * 4);
    w->slab_elems = (size_t)24 * D * std::max(D, F);
    w->slab = take(w->slab_elems * 4);
    w->prob_sk = take(TF_SPLITK_PROBS * sizeof(GemmProb));
  }
  w->total = p;
  return SUMK_OK;
}

__global__ void tf_setup_kernel(const int32_t* off, int n_seq, int D, int heads, TfSeq* seq, GemmProb* tabs) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_seq) return;
  const int dh = D / heads, bt = 64, tnv = (dh + bt - 1) / bt, np = n_seq * heads;
  int64_t eoff = 0; int ts = 0, tpv = 0;
  for (int q = 0; q < s; ++q) {
    int T = off[q + 1] - off[q], tm = (T + bt - 1) / bt;
    eoff += (int64_t)heads * T * ((T + 3) & ~3); ts += heads * tm * tm; tpv += heads * tm * tnv;
  }
  const int row0 = off[s], T = off[s + 1] - row0, ldE = (T + 3) & ~3, tm = (T + bt - 1) / bt;
  TfSeq si; si.eoff = eoff; si.row0 = row0; si.T = T; si.ldE = ldE; si.pad_ = 0;
  seq[s] = si;
  auto put = [&](int tab, int h, int64_t a_off, int64_t b_off, int64_t c_off, int M, int N, int K, int lda, int ldb, int ldc,
                 int tile_start, int tiles_n) {
    GemmProb q;
    q.a_off = a_off; q.b_off = b_off; q.c_off = c_off; q.r_off = 0;
    q.M = M; q.N = N; q.K = K; q.lda = lda; q.ldb = ldb; q.ldc = ldc; q.ldr = 0; q.tile_start = tile_start; q.tiles_n = tiles_n;
    for (int i = 0; i < 7; ++i) q.pad_[i] = 0;
    tabs[(size_t)tab * np + s * heads + h] = q;
  };
  for (int h = 0; h < heads; ++h) {
    const int64_t q0 = (int64_t)row0 * 3 * D + h * dh, eb = eoff + (int64_t)h * T * ldE, c0 = (int64_t)row0 * D + h * dh;
    const int t_s = ts + h * tm * tm, t_pv = tpv + h * tm * tnv;
    put(TT_S, h, q0, q0 + D, eb, T, T, dh, 3 * D, 3 * D, ldE, t_s, tm);              // E = Q_h K_h^T          (NT, K = dh)
    put(TT_PV, h, eb, q0 + 2 * D, c0, T, dh, T, ldE, 3 * D, D, t_pv, tnv);           // C_h = alpha_h V_h      (NN, K = T)
    put(TT_DV, h, eb, c0, q0 + 2 * D, T, dh, T, ldE, D, 3 * D, t_pv, tnv);           // dV_h = alpha_h^T dC_h  (TN)
    put(TT_DP, h, c0, q0 + 2 * D, eb, T, T, dh, D, 3 * D, ldE, t_s, tm);             // dAlpha = dC_h V_h^T    (NT)
    put(TT_DQ, h, eb, q0 + D, q0, T, dh, T, ldE, 3 * D, 3 * D, t_pv, tnv);           // dQ_h = dS K_h          (NN)
    put(TT_DK, h, eb, q0, q0 + D, T, dh, T, ldE, 3 * D, 3 * D, t_pv, tnv);           // dK_h = dS^T Q_h        (TN)
  }
}

// one wave per (query row, head): alpha = softmax(scale * logits); pad columns zeroed; training with dropout also writes
// dropout(alpha) to E2 (mask index = ((row*heads + h) << 20) | j)
__global__ __launch_bounds__(256) void tf_softmax_kernel(float* E, float* E2, const TfSeq* seq, const int32_t* off, int n_seq,
                                                         int n_rows, int heads, float scale, Drop drop, uint32_t site) {
  const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= (int64_t)n_rows * heads) return;
  const int lane = threadIdx.x & 63;
  const int row = (int)(wid / heads), h = (int)(wid % heads);
  int lo = 0, hi = n_seq - 1;
  while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (off[mid] <= row) lo = mid; else hi = mid - 1; }
  const TfSeq si = seq[lo];
  const int i = row - si.row0, T = si.T;
  const int64_t ro = si.eoff + ((int64_t)h * T + i) * si.ldE;
  float* e = E + ro;
  if (T <= 512) {        // the row lives in registers: one read, one exp per element (same operations in the same order as the loops below)
    float v[8];
    float m = -INFINITY;
#pragma unroll
    for (int q = 0; q < 8; ++q) { const int j = lane + 64 * q; v[q] = j < T ? e[j] : -INFINITY; }
#pragma unroll
    for (int q = 0; q < 8; ++q) m = fmaxf(m, v[q] * scale);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (lane + 64 * q < T) { v[q] = expf(__builtin_fmaf(v[q], scale, -m)); sum += v[q]; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int j = lane + 64 * q;
      if (j < si.ldE) {
        const float a = j < T ? v[q] / sum : 0.f;
        e[j] = a;
        if (E2) E2[ro + j] = drop.thr ? drop_apply(drop, site, ((uint64_t)wid << 20) | (uint64_t)j, a) : a;
      }
    }
    return;
  }
  float m = -INFINITY;
  for (int j = lane; j < T; j += 64) m = fmaxf(m, e[j] * scale);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  float sum = 0.f;
  for (int j = lane; j < T; j += 64) sum += expf(__builtin_fmaf(e[j], scale, -m));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  for (int j = lane; j < si.ldE; j += 64) {
    const float v = j < T ? expf(__builtin_fmaf(e[j], scale, -m)) / sum : 0.f;
    e[j] = v;
    if (E2) E2[ro + j] = drop.thr ? drop_apply(drop, site, ((uint64_t)wid << 20) | (uint64_t)j, v) : v;
  }
}

// dLogits = scale * alpha * (dAlpha - sum_j dAlpha_j alpha_j), dAlpha = dropout'(dAlphaDropped); in place on E2
__global__ __launch_bounds__(256) void tf_softmax_bwd_kernel(const float* E, float* E2, const TfSeq* seq, const int32_t* off,
                                                             int n_seq, int n_rows, int heads, float scale, Drop drop,
                                                             uint32_t site) {
  const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= (int64_t)n_rows * heads) return;
  const int lane = threadIdx.x & 63;
  const int row = (int)(wid / heads), h = (int)(wid % heads);
  int lo = 0, hi = n_seq - 1;
  while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (off[mid] <= row) lo = mid; else hi = mid - 1; }
  const TfSeq si = seq[lo];
  const int i = row - si.row0, T = si.T;
  const int64_t ro = si.eoff + ((int64_t)h * T + i) * si.ldE;
  const float* p = E + ro; float* g = E2 + ro;
  float dot = 0.f;
  for (int j = lane; j < T; j += 64) {
    float d = g[j];
    if (drop.thr) d = drop_apply(drop, site, ((uint64_t)wid << 20) | (uint64_t)j, d);
    dot += d * p[j];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
  for (int j = lane; j < si.ldE; j += 64) {
    float v = 0.f;
    if (j < T) {
      float d = g[j];
      if (drop.thr) d = drop_apply(drop, site, ((uint64_t)wid << 20) | (uint64_t)j, d);
      v = p[j] * (d - dot) * scale;
    }
    g[j] = v;
  }
}

__global__ void add_rows_kernel(float* __restrict__ y, const float* __restrict__ x, int64_t n4) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  float4 a = reinterpret_cast<float4*>(y)[i], b = reinterpret_cast<const float4*>(x)[i];
  a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  reinterpret_cast<float4*>(y)[i] = a;
}
// dst[r,c] = src[r,c] * (keep(site, r*N + c) ? scale : 0)      (backward of an epilogue dropout)
__global__ void mask_scale_kernel(float* __restrict__ dst, const float* __restrict__ src, int64_t n, Drop drop, uint32_t site) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = drop_apply(drop, site, (uint64_t)i, src[i]);
}
// g[i] = act[i] > 0 ? g[i] * scale : 0   -- backward of dropout(relu(.)) given the stored post-dropout activation
__global__ void relu_drop_bwd_kernel(float* __restrict__ g, const float* __restrict__ act, int64_t n, float scale) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) g[i] = act[i] > 0.f ? g[i] * scale : 0.f;
}

struct TfGeom { TfWs L; int R, dh, tiles_s, tiles_pv, c_qkv, c_dd, c_df; };
enum { P_QKV = 0, P_DD = 1, P_DF = 2, P_FD = 3, P_DQKV = 4, P_NN_DF = 5, P_NN_FD = 6 };

static int tf_geometry(int D, int F, int heads, int n_layers, int n_seq, const int32_t* off, int training, TfGeom* G) {
  SUMK_TRY(tf_carve(D, F, heads, n_layers, n_seq, off, training, &G->L));
  G->R = G->L.n_rows; G->dh = D / heads; G->tiles_s = G->tiles_pv = 0;
  for (int s = 0; s < n_seq; ++s) {
    int T = off[s + 1] - off[s], tm = (T + 63) / 64;
    G->tiles_s += heads * tm * tm; G->tiles_pv += heads * tm * ((G->dh + 63) / 64);
  }
  auto cfg = [&](int M, int N) { return gemm_tiles(M, N, 0) >= 512 ? 0 : 1; };
  G->c_qkv = cfg(G->R, 3 * D); G->c_dd = cfg(G->R, D); G->c_df = cfg(G->R, F);
  return SUMK_OK;
}

static int tf_tables(const TfGeom& G, int D, int F, int heads, int n_seq, const int32_t* off_dev, char* ws, hipStream_t stream) {
  const int R = G.R;
  GemmProb* prow = (GemmProb*)(ws + G.L.prob_row);
  hipLaunchKernelGGL(tf_setup_kernel, dim3((n_seq + 63) / 64), dim3(64), 0, stream, off_dev, n_seq, D, heads,
                     (TfSeq*)(ws + G.L.seq), (GemmProb*)(ws + G.L.prob_tabs));
  SUMK_TRY(fill_single_prob(prow + P_QKV, R, 3 * D, D, D, D, 3 * D, 0, G.c_qkv, stream));   // (R,3D) <- (R,D) x (3D,D)^T
  SUMK_TRY(fill_single_prob(prow + P_DD, R, D, D, D, D, D, D, G.c_dd, stream));             // (R,D)  <- (R,D) x (D,D)   [+R ld D]
  SUMK_TRY(fill_single_prob(prow + P_DF, R, F, D, D, D, F, 0, G.c_df, stream));             // (R,F)  <- (R,D) x (F,D)^T ; also NN with B (D,F)
  SUMK_TRY(fill_single_prob(prow + P_FD, R, D, F, F, F, D, D, G.c_dd, stream));             // (R,D)  <- (R,F) x (D,F)^T ; NN with B (F,D): ldb = D set below
  SUMK_TRY(fill_single_prob(prow + P_DQKV, R, D, 3 * D, 3 * D, D, D, D, G.c_dd, stream));   // (R,D)  <- (R,3D) x (3D,D)  (NN)
  SUMK_TRY(fill_single_prob(prow + P_NN_DF, R, F, D, D, F, F, 0, G.c_df, stream));          // (R,F)  <- (R,D) x (D,F)    (NN)
  SUMK_TRY(fill_single_prob(prow + P_NN_FD, R, D, F, F, D, D, D, G.c_dd, stream));          // (R,D)  <- (R,F) x (F,D)    (NN)
  return SUMK_OK;
}

}  // namespace sumk

using namespace sumk;

extern "C" size_t sumk_transformer_workspace_bytes(int32_t D, int32_t F, int32_t n_heads, int32_t n_layers, int32_t n_seq,
                                                   const int32_t* seq_off_host, int32_t training) {
  TfWs w;
  if (tf_carve(D, F, n_heads, n_layers, n_seq, seq_off_host, training, &w) != SUMK_OK) return 0;
  return w.total;
}

extern "C" size_t sumk_transformer_workspace_bytes_for(int32_t D, int32_t F, int32_t n_heads, int32_t n_layers, int32_t n_seq,
                                                       const int32_t* seq_off_host, int32_t training, int32_t precision) {
  TfWs w;
  if (tf_carve(D, F, n_heads, n_layers, n_seq, seq_off_host, training, &w) != SUMK_OK) return 0;
  const int np = tf_np(precision);
  if (!training && np && tf_wplanes_ok(D, F, np) && tf_pw_rows_ok(D, F, w.n_rows, np))
    return tf_pw_extra(D, F, w.n_rows, np, w.total, n_heads, n_seq, tf_t_max(n_seq, seq_off_host)).total;
  return w.total;
}

extern "C" size_t sumk_transformer_wplanes_bytes(int32_t D, int32_t F, int32_t n_layers, int32_t n_planes) {
  if (n_layers < 1 || D < 1 || F < 1 || !tf_wplanes_ok(D, F, n_planes)) return 0;
  return tf_wplanes_layout(D, F, n_layers, n_planes).total;
}

extern "C" int sumk_transformer_wplanes_build(int32_t D, int32_t F, int32_t n_layers, const sumk_tf_layer_weights* layers,
                                              const sumk_tf_head_weights* head, int32_t n_planes, void* out, size_t out_bytes,
                                              void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SUMK_ARG(layers && head && out && n_layers > 0, "transformer_wplanes_build: null pointer");
  SUMK_ARG(tf_wplanes_ok(D, F, n_planes), "transformer_wplanes_build: D=%d F=%d planes=%d is not eligible (D, F multiples of 256; 2 or 3 planes)", D, F, n_planes);
  const TfWPlanes l = tf_wplanes_layout(D, F, n_layers, n_planes);
  SUMK_ARG(out_bytes >= l.total && ((uintptr_t)out & 255) == 0, "transformer_wplanes_build: %zu bytes (256-byte aligned) needed, got %zu", l.total, out_bytes);
  char* o = (char*)out;
  for (int i = 0; i < n_layers; ++i) {
    const sumk_tf_layer_weights& W = layers[i];
    SUMK_ARG(W.in_proj_w && W.out_proj_w && W.lin1_w && W.lin2_w, "transformer_wplanes_build: null weight in layer %d", i);
    char* lb = o + (size_t)i * l.lay_stride;
    SUMK_TRY(split_planes(W.in_proj_w, 3 * (int64_t)D, D, D, n_planes, lb + l.win, stream));
    SUMK_TRY(split_planes(W.out_proj_w, D, D, D, n_planes, lb + l.wo, stream));
    SUMK_TRY(split_planes(W.lin1_w, F, D, D, n_planes, lb + l.w1, stream));
    SUMK_TRY(split_planes(W.lin2_w, D, F, F, n_planes, lb + l.w2, stream));
  }
  SUMK_ARG(head->k1_w, "transformer_wplanes_build: null k1 weight");
  SUMK_TRY(split_planes(head->k1_w, D, D, D, n_planes, o + l.k1, stream));
  return SUMK_OK;
}

extern "C" int sumk_transformer_forward(float* x, int32_t D, int32_t F, int32_t n_heads, int32_t n_layers, int32_t n_seq,
                                        const int32_t* seq_off_host, const int32_t* seq_off_dev,
                                        const sumk_tf_layer_weights* layers, const sumk_tf_head_weights* head,
                                        const sumk_tf_opts* opts, const float* pos_table, const int32_t* pos_rows,
                                        float* scores, void* workspace, size_t workspace_bytes, int32_t training,
                                        void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SUMK_ARG(x && seq_off_dev && layers && head && opts && scores && workspace, "transformer_forward: null pointer");
  SUMK_ARG(n_layers > 0, "transformer_forward: n_layers=%d", n_layers);
  SUMK_ARG((pos_table == nullptr) == (pos_rows == nullptr), "transformer_forward: pos_table and pos_rows go together");
  SUMK_ARG(head->ln_w && head->ln_b && head->k1_w && head->k1_b && head->k2_w && head->k2_b, "transformer_forward: null head weight");
  SUMK_ARG((opts->layer_dropout_p == 0.f && opts->head_dropout_p == 0.f) || training, "transformer_forward: dropout needs training mode");
  TfGeom G;
  SUMK_TRY(tf_geometry(D, F, n_heads, n_layers, n_seq, seq_off_host, training, &G));
  const TfWs& L = G.L;
  if (workspace_bytes < L.total) { set_error("transformer_forward: workspace %zu < required %zu", workspace_bytes, L.total); return SUMK_ERR_WORKSPACE; }
  char* ws = (char*)workspace;
  const int R = G.R, dh = G.dh, np = n_seq * n_heads;
  // the plane path: inference in a split-bf16 arithmetic with the weights' planes at hand
  const int npl = tf_np(opts->precision);
  const bool pw = !training && npl && opts->wplanes && tf_wplanes_ok(D, F, npl) && tf_pw_rows_ok(D, F, R, npl);
  TfPwExtra PX{}; TfWPlanes WL{};
  const int t_max = tf_t_max(n_seq, seq_off_host);
  if (pw) {
    PX = tf_pw_extra(D, F, R, npl, L.total, n_heads, n_seq, t_max); WL = tf_wplanes_layout(D, F, n_layers, npl);
    SUMK_ARG(((uintptr_t)opts->wplanes & 255) == 0, "transformer_forward: wplanes must be 256-byte aligned");
    if (workspace_bytes < PX.total) { set_error("transformer_forward: workspace %zu < %zu the plane path needs (sumk_transformer_workspace_bytes_for)", workspace_bytes, PX.total); return SUMK_ERR_WORKSPACE; }
  }
  char* const PA = ws + PX.pa; char* const PB = ws + PX.pb;
  const bool pw_attn = pw && PX.attn;
  SeqInfo* const seqinfo = (SeqInfo*)(ws + PX.seqinfo);
  if (pw_attn) {
    hipLaunchKernelGGL(tf_seqinfo_kernel, dim3((n_seq + 63) / 64), dim3(64), 0, stream, seq_off_dev, n_seq, seqinfo);
    // (the slack a V read past the LAST sub-array's pitch lands in: cleared once per call, no kernel writes it -- 0 x stale NaN bits would be NaN)
    SUMK_HIP(hipMemsetAsync(ws + PX.qp + pw_planes_bytes(R, 3 * D, npl) - 8192, 0, 8192, stream));
  }
  const char* const wpl = (const char*)opts->wplanes;
  // C (R, N) = A planes (R, K) x weight planes (N, K)^T + bias [+ Rs] [ReLU]; O: the result as planes instead
  auto pw_linear = [&](const void* Ap, const char* Wp, int N, int K, const float* bias, const float* Rs, int relu, float* C_, void* O_) {
    PwLaunch g; g.A = Ap; g.B = Wp; g.a_rows = R; g.b_rows = N; g.M = R; g.N = N; g.K = K; g.np = npl;
    g.bias = bias; g.R = Rs; g.ldr = N; g.relu = relu; g.C = C_; g.ldc = N; g.O = O_; g.o_rows = R;
    return launch_gemm_pw(O_ ? PW_PLANES : Rs ? PW_RES_F32 : PW_F32, g, stream);
  };
  TfSeq* seq = (TfSeq*)(ws + L.seq);
  GemmProb* prow = (GemmProb*)(ws + L.prob_row);
  GemmProb* tabs = (GemmProb*)(ws + L.prob_tabs);
  float* T1 = (float*)(ws + L.t1);
  const Drop dl = make_drop(opts->layer_dropout_p, opts->seed), dhd = make_drop(opts->head_dropout_p, opts->seed);
  const float att_scale = 1.0f / sqrtf((float)dh);

  if (pos_table) SUMK_TRY(launch_add_pos(x, pos_table, pos_rows, R, D, stream));
  SUMK_TRY(tf_tables(G, D, F, n_heads, n_seq, seq_off_dev, ws, stream));

  const float* hin = x;
  for (int l = 0; l < n_layers; ++l) {
    const sumk_tf_layer_weights& W = layers[l];
    SUMK_ARG(W.in_proj_w && W.in_proj_b && W.out_proj_w && W.out_proj_b && W.lin1_w && W.lin1_b && W.lin2_w && W.lin2_b &&
             W.norm1_w && W.norm1_b && W.norm2_w && W.norm2_b, "transformer_forward: null weight in layer %d", l);
    float *QKV, *E, *E2 = nullptr, *CTX, *T1a, *hmid, *FF, *T1b, *hout, *stats = nullptr;
    if (training) {
      char* lb = ws + L.lay0 + (size_t)l * L.lay_stride;
      QKV = (float*)(lb + L.l_qkv); E = (float*)(lb + L.l_p); E2 = dl.thr ? (float*)(lb + L.l_pd) : nullptr;
      CTX = (float*)(lb + L.l_ctx); T1a = (float*)(lb + L.l_t1a); hmid = (float*)(lb + L.l_hmid); FF = (float*)(lb + L.l_ff);
      T1b = (float*)(lb + L.l_t1b); hout = (float*)(lb + L.l_hout); stats = (float*)(lb + L.l_stats);
    } else {
      float* Hb[3] = {(float*)(ws + L.h0), (float*)(ws + L.h1), (float*)(ws + L.h2)};
      QKV = (float*)(ws + L.qkv); E = (float*)(ws + L.e); CTX = (float*)(ws + L.ctx); FF = (float*)(ws + L.ff);
      T1a = T1b = T1; hmid = Hb[0]; hout = Hb[1 + (l & 1)];   // never aliases this layer's input (x or the previous hout)
    }
    const uint32_t site = 10u * (uint32_t)l;
    const char* const wl = pw ? wpl + (size_t)l * WL.lay_stride : nullptr;
    if (pw_attn) {   // in-projection -> planes of [Q | K | V] (+ bias); per (video, head): logits + softmax -> alpha planes, alpha . V -> context planes
      SUMK_TRY(split_planes(hin, R, D, D, npl, PA, stream));
      {
        PwLaunch g; g.A = PA; g.B = wl + WL.win; g.a_rows = R; g.b_rows = 3 * (int64_t)D; g.M = R; g.N = 3 * D; g.K = D; g.np = npl;
        g.bias = W.in_proj_b; g.O = ws + PX.qp; g.o_rows = R; g.o_store_rows = pw_rows_pitch(R);      // (pad rows = the bias: finite -- the context kernel multiplies V rows up to 31 past the last video by alpha = 0)
        SUMK_TRY(launch_gemm_pw(PW_PLANES, g, stream));
      }
      SUMK_TRY(launch_attn_pw_logits(npl, ws + PX.qp, R, D, nullptr, ws + PX.ap, seqinfo, n_seq, t_max, att_scale, 0, -1, stream, n_heads));
      SUMK_TRY(launch_attn_pw_context(npl, ws + PX.qp, R, D, ws + PX.ap, PA, seqinfo, n_seq, t_max, stream, n_heads));      // PA: the planes of hin are spent
      SUMK_TRY(pw_linear(PA, wl + WL.wo, D, D, W.out_proj_b, hin, 0, T1a, nullptr));
    } else {   // the per-(video, head) products stay on the in-loop kernels (fp32 Q / K / V, logits, context), between plane GEMMs when the weights' planes are given
      if (pw) {
        SUMK_TRY(split_planes(hin, R, D, D, npl, PA, stream));
        SUMK_TRY(pw_linear(PA, wl + WL.win, 3 * D, D, W.in_proj_b, nullptr, 0, QKV, nullptr));
      } else {  // packed in-projection  [Q|K|V] = h Win^T + bin
        GemmLaunch g; g.precision = opts->precision;
        g.A = hin; g.B[0] = W.in_proj_w; g.bias0[0] = W.in_proj_b; g.C = QKV; g.probs = prow + P_QKV; g.small_tile = G.c_qkv;
        g.total_tiles = gemm_tiles(R, 3 * D, G.c_qkv); g.xcd_M = R; g.xcd_N = 3 * D; g.lean = gemm_lean_ok(R, 3 * D, D, D, D);
        SUMK_TRY(launch_gemm(GEMM_NT, EPI_BIAS2, g, stream));
      }
      {  // logits per (video, head)
        GemmLaunch g; g.precision = opts->precision;
        g.A = QKV; g.B[0] = QKV; g.C = E; g.probs = tabs + (size_t)TT_S * np; g.nprob = np; g.small_tile = 1; g.total_tiles = G.tiles_s;
        SUMK_TRY(launch_gemm(GEMM_NT, EPI_NONE, g, stream));
      }
      hipLaunchKernelGGL(tf_softmax_kernel, dim3((unsigned)(((int64_t)R * n_heads + 3) / 4)), dim3(256), 0, stream, E, E2, seq,
                         seq_off_dev, n_seq, R, n_heads, att_scale, dl, site + 0);
      {  // context, heads written side by side
        GemmLaunch g; g.precision = opts->precision;
        g.A = E2 ? E2 : E; g.B[0] = QKV; g.C = CTX; g.probs = tabs + (size_t)TT_PV * np; g.nprob = np; g.small_tile = 1;
        g.total_tiles = G.tiles_pv;
        SUMK_TRY(launch_gemm(GEMM_NN, EPI_NONE, g, stream));
      }
      if (pw) {
        SUMK_TRY(split_planes(CTX, R, D, D, npl, PA, stream));
        SUMK_TRY(pw_linear(PA, wl + WL.wo, D, D, W.out_proj_b, hin, 0, T1a, nullptr));
      } else {  // out-projection + bias (+dropout1) + residual
        GemmLaunch g; g.precision = opts->precision;
        g.A = CTX; g.B[0] = W.out_proj_w; g.bias0[0] = W.out_proj_b; g.R = hin; g.C = T1a; g.probs = prow + P_DD; g.small_tile = G.c_dd;
        g.total_tiles = gemm_tiles(R, D, G.c_dd); g.xcd_M = R; g.xcd_N = D; g.lean = gemm_lean_ok(R, D, D, D, D);
        g.drop_seed = dl.seed; g.drop_thr = dl.thr; g.drop_scale = dl.scale; g.drop_site = site + 1;
        SUMK_TRY(launch_gemm(GEMM_NT, EPI_BIAS_RESIDUAL, g, stream));
      }
    }
    SUMK_TRY(launch_layernorm(T1a, hmid, W.norm1_w, W.norm1_b, R, D, opts->layer_eps, stats, stream));
    if (pw) {  // both feed-forward layers; ReLU(lin1) exists as planes only
      SUMK_TRY(split_planes(hmid, R, D, D, npl, PA, stream));
      SUMK_TRY(pw_linear(PA, wl + WL.w1, F, D, W.lin1_b, nullptr, 1, nullptr, PB));
      SUMK_TRY(pw_linear(PB, wl + WL.w2, D, F, W.lin2_b, hmid, 0, T1b, nullptr));
    } else {
      {  // feed-forward 1: bias + ReLU (+dropout)
        GemmLaunch g; g.precision = opts->precision;
        g.A = hmid; g.B[0] = W.lin1_w; g.bias0[0] = W.lin1_b; g.C = FF; g.probs = prow + P_DF; g.small_tile = G.c_df;
        g.total_tiles = gemm_tiles(R, F, G.c_df); g.xcd_M = R; g.xcd_N = F; g.lean = gemm_lean_ok(R, F, D, D, D);
        g.drop_seed = dl.seed; g.drop_thr = dl.thr; g.drop_scale = dl.scale; g.drop_site = site + 2;
        SUMK_TRY(launch_gemm(GEMM_NT, EPI_BIAS_RELU, g, stream));
      }
      {  // feed-forward 2: bias (+dropout2) + residual
        GemmLaunch g; g.precision = opts->precision;
        g.A = FF; g.B[0] = W.lin2_w; g.bias0[0] = W.lin2_b; g.R = hmid; g.C = T1b; g.probs = prow + P_FD; g.small_tile = G.c_dd;
        g.total_tiles = gemm_tiles(R, D, G.c_dd); g.xcd_M = R; g.xcd_N = D; g.lean = gemm_lean_ok(R, D, F, F, F);
        g.drop_seed = dl.seed; g.drop_thr = dl.thr; g.drop_scale = dl.scale; g.drop_site = site + 3;
        SUMK_TRY(launch_gemm(GEMM_NT, EPI_BIAS_RESIDUAL, g, stream));
      }
    }
    SUMK_TRY(launch_layernorm(T1b, hout, W.norm2_w, W.norm2_b, R, D, opts->layer_eps, stats ? stats + 2 * (size_t)R : nullptr, stream));
    hin = hout;
  }
  // final (shared) LayerNorm of the encoder, optional extra residual, scoring head
  float* hfin = training ? (float*)(ws + L.hfin) : (float*)(ws + L.h0);
  float* Z = training ? (float*)(ws + L.z) : T1;
  float* sfin = training ? (float*)(ws + L.stats_fin) : nullptr;
  SUMK_TRY(launch_layernorm(hin, hfin, head->ln_w, head->ln_b, R, D, opts->final_eps, sfin, stream));
  if (opts->more_residuals) {
    int64_t n4 = (int64_t)R * (D >> 2);
    hipLaunchKernelGGL(add_rows_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, hfin, x, n4);
  }
  if (pw) {
    SUMK_TRY(split_planes(hfin, R, D, D, npl, PA, stream));
    SUMK_TRY(pw_linear(PA, wpl + WL.k1, D, D, head->k1_b, nullptr, 1, Z, nullptr));
  } else {
    GemmLaunch g; g.precision = opts->precision;
    g.A = hfin; g.B[0] = head->k1_w; g.bias0[0] = head->k1_b; g.C = Z; g.probs = prow + P_DD; g.small_tile = G.c_dd;
    g.total_tiles = gemm_tiles(R, D, G.c_dd); g.xcd_M = R; g.xcd_N = D; g.lean = gemm_lean_ok(R, D, D, D, D);
    SUMK_TRY(launch_gemm(GEMM_NT, EPI_BIAS_RELU, g, stream));
  }
  SUMK_TRY(launch_ln_head_drop(Z, head->ln_w, head->ln_b, head->k2_w, head->k2_b, scores, R, D, opts->final_eps,
                               sfin ? sfin + 2 * (size_t)R : nullptr, dhd, 1000u, stream));
  if (training) SUMK_HIP(hipMemcpyAsync(ws + L.scores, scores, (size_t)R * 4, hipMemcpyDeviceToDevice, stream));
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}

extern "C" int sumk_transformer_backward(const float* x, int32_t D, int32_t F, int32_t n_heads, int32_t n_layers,
                                         int32_t n_seq, const int32_t* seq_off_host, const int32_t* seq_off_dev,
                                         const sumk_tf_layer_weights* layers, const sumk_tf_head_weights* head,
                                         const sumk_tf_opts* opts, const float* dscores, const sumk_tf_layer_grads* lgr,
                                         const sumk_tf_head_grads* hgr, float* dx, void* workspace, size_t workspace_bytes,
                                         void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SUMK_ARG(x && seq_off_dev && layers && head && opts && dscores && lgr && hgr && workspace, "transformer_backward: null pointer");
  TfGeom G;
  SUMK_TRY(tf_geometry(D, F, n_heads, n_layers, n_seq, seq_off_host, 1, &G));
  const TfWs& L = G.L;
  if (workspace_bytes < L.total) { set_error("transformer_backward: workspace %zu < required %zu (needs the training-mode forward's workspace)", workspace_bytes, L.total); return SUMK_ERR_WORKSPACE; }
  char* ws = (char*)workspace;
  const int R = G.R, dh = G.dh, np = n_seq * n_heads;
  const int64_t nRD = (int64_t)R * D, nRF = (int64_t)R * F;
  TfSeq* seq = (TfSeq*)(ws + L.seq);
  GemmProb* prow = (GemmProb*)(ws + L.prob_row);
  GemmProb* tabs = (GemmProb*)(ws + L.prob_tabs);
  GemmProb* psk = (GemmProb*)(ws + L.prob_sk);
  float* g0 = (float*)(ws + L.g0); float* g1 = (float*)(ws + L.g1); float* g2 = (float*)(ws + L.g2);
  float* dQKV = (float*)(ws + L.dqkv); float* dFF = (float*)(ws + L.dff);
  float* lnpart = (float*)(ws + L.lnpart); float* colpart = (float*)(ws + L.colpart); float* slab = (float*)(ws + L.slab);
  const float* hfin = (const float*)(ws + L.hfin); const float* Z = (const float*)(ws + L.z);
  const float* sfin = (const float*)(ws + L.stats_fin); const float* scores = (const float*)(ws + L.scores);
  const Drop dl = make_drop(opts->layer_dropout_p, opts->seed), dhd = make_drop(opts->head_dropout_p, opts->seed);
  const Drop none = make_drop(0.f, 0);
  const float att_scale = 1.0f / sqrtf((float)dh);
  int nw = 0;
  auto blocks = [](int64_t n) { return dim3((unsigned)((n + 255) / 256)); };
  auto nn = [&](const float* A, const float* B, float* C, int prob, GemmEpi epi, int N) -> int {   // C (op)= A . B, B stored (K, N)
    GemmLaunch g; g.precision = opts->precision;
    g.A = A; g.B[0] = B; g.C = C; g.probs = prow + prob; g.small_tile = (N == F ? G.c_df : G.c_dd);
    g.total_tiles = gemm_tiles(R, N, g.small_tile);
    return launch_gemm(GEMM_NN, epi, g, stream);
  };
  auto wgrad = [&](const float* dY, int ldy, int M, const float* Xin, int ldx, int N, float* out0, float* out1, float* out2,
                   int rows_per_out) -> int {   // out[M,N] += dY^T Xin
    float* out[4] = {out0, out1, out2, nullptr};
    return gemm_tn_splitk_accum(dY, ldy, Xin, ldx, M, N, R, slab, L.slab_elems, psk, TF_SPLITK_PROBS, out, rows_per_out, N, 1.f, stream, opts->precision);
  };

  // ---- scoring head: dZ (ReLU + head dropout masks applied), dk2, db2, shared-LN grads
  SUMK_TRY(launch_ln_head_bwd(D, R, Z, sfin + 2 * (size_t)R, head->ln_w, head->ln_b, head->k2_w, scores, dscores, g0, lnpart, dhd,
                              1000u, &nw, stream));
  SUMK_TRY(ln_bwd_reduce(lnpart, nw, D, hgr->ln_w, hgr->ln_b, hgr->k2_w, hgr->k2_b, hgr->k1_b, stream));   // dk1_b = column sums of g0
  SUMK_TRY(wgrad(g0, D, D, hfin, D, D, hgr->k1_w, nullptr, nullptr, D));
  SUMK_TRY(nn(g0, head->k1_w, g1, P_DD, EPI_NONE, D));                       // dHfin' -> g1
  if (opts->more_residuals && dx) SUMK_HIP(hipMemcpyAsync(dx, g1, (size_t)nRD * 4, hipMemcpyDeviceToDevice, stream));
  // encoder's final (shared) LayerNorm
  const float* hlast = (const float*)(ws + L.lay0 + (size_t)(n_layers - 1) * L.lay_stride + L.l_hout);
  SUMK_TRY(launch_ln_bwd_rows(D, R, hlast, sfin, head->ln_w, head->ln_b, g1, g2, lnpart, none, 0u, &nw, stream));
  SUMK_TRY(ln_bwd_reduce(lnpart, nw, D, hgr->ln_w, hgr->ln_b, nullptr, nullptr, nullptr, stream));
  float* dH = g2;            // gradient w.r.t. the current layer's output
  float* fa = g0; float* fb = g1;   // the two free (R,D) buffers

  for (int l = n_layers - 1; l >= 0; --l) {
    const sumk_tf_layer_weights& W = layers[l];
    const sumk_tf_layer_grads& Gd = lgr[l];
    char* lb = ws + L.lay0 + (size_t)l * L.lay_stride;
    const float* QKV = (const float*)(lb + L.l_qkv); const float* P = (const float*)(lb + L.l_p);
    float* E2 = (float*)(lb + L.l_pd);
    const float* CTX = (const float*)(lb + L.l_ctx); const float* T1a = (const float*)(lb + L.l_t1a);
    const float* hmid = (const float*)(lb + L.l_hmid); const float* FFa = (const float*)(lb + L.l_ff);
    const float* T1b = (const float*)(lb + L.l_t1b); const float* stats = (const float*)(lb + L.l_stats);
    const float* hin = l == 0 ? x : (const float*)(ws + L.lay0 + (size_t)(l - 1) * L.lay_stride + L.l_hout);
    const uint32_t site = 10u * (uint32_t)l;
    // norm2
    SUMK_TRY(launch_ln_bwd_rows(D, R, T1b, stats + 2 * (size_t)R, W.norm2_w, W.norm2_b, dH, fa, lnpart, none, 0u, &nw, stream));
    SUMK_TRY(ln_bwd_reduce(lnpart, nw, D, Gd.norm2_w, Gd.norm2_b, nullptr, nullptr, nullptr, stream));
    // fa = dT1b (also the residual gradient into hmid).  dropout2 mask -> fb = d(linear2 output)
    const float* dL2 = fa;
    if (dl.thr) { hipLaunchKernelGGL(mask_scale_kernel, blocks(nRD), dim3(256), 0, stream, fb, fa, nRD, dl, site + 3); dL2 = fb; }
    SUMK_TRY(colsum_accum(dL2, D, R, D, colpart, TF_COLSUM_CHUNKS, Gd.lin2_b, stream));
    SUMK_TRY(wgrad(dL2, D, D, FFa, F, F, Gd.lin2_w, nullptr, nullptr, D));
    SUMK_TRY(nn(dL2, W.lin2_w, dFF, P_NN_DF, EPI_NONE, F));                     // dFF = dL2 . W2   (W2 stored (D,F) = (K,N))
    hipLaunchKernelGGL(relu_drop_bwd_kernel, blocks(nRF), dim3(256), 0, stream, dFF, FFa, nRF, dl.scale);
    SUMK_TRY(colsum_accum(dFF, F, R, F, colpart, TF_COLSUM_CHUNKS, Gd.lin1_b, stream));
    SUMK_TRY(wgrad(dFF, F, F, hmid, D, D, Gd.lin1_w, nullptr, nullptr, F));
    SUMK_TRY(nn(dFF, W.lin1_w, fa, P_NN_FD, EPI_ACCUM, D));                     // fa = dT1b + dFF . W1 = dHmid   (W1 stored (F,D) = (K,N))
    // norm1
    SUMK_TRY(launch_ln_bwd_rows(D, R, T1a, stats, W.norm1_w, W.norm1_b, fa, dH, lnpart, none, 0u, &nw, stream));   // dH buffer reused: dT1a
    SUMK_TRY(ln_bwd_reduce(lnpart, nw, D, Gd.norm1_w, Gd.norm1_b, nullptr, nullptr, nullptr, stream));
    float* dT1a = dH;
    const float* dAO = dT1a;
    if (dl.thr) { hipLaunchKernelGGL(mask_scale_kernel, blocks(nRD), dim3(256), 0, stream, fb, dT1a, nRD, dl, site + 1); dAO = fb; }
    SUMK_TRY(colsum_accum(dAO, D, R, D, colpart, TF_COLSUM_CHUNKS, Gd.out_proj_b, stream));
    SUMK_TRY(wgrad(dAO, D, D, CTX, D, D, Gd.out_proj_w, nullptr, nullptr, D));
    SUMK_TRY(nn(dAO, W.out_proj_w, fa, P_DD, EPI_NONE, D));                  // fa = dCTX
    // multi-head attention backward, per (video, head)
    const float* Pd = dl.thr ? (const float*)E2 : P;
    {
      GemmLaunch g; g.precision = opts->precision;
      g.A = Pd; g.B[0] = fa; g.C = dQKV; g.probs = tabs + (size_t)TT_DV * np; g.nprob = np; g.small_tile = 1; g.total_tiles = G.tiles_pv;
      SUMK_TRY(launch_gemm(GEMM_TN, EPI_NONE, g, stream));
    }
    {
      GemmLaunch g; g.precision = opts->precision;
      g.A = fa; g.B[0] = QKV; g.C = E2; g.probs = tabs + (size_t)TT_DP * np; g.nprob = np; g.small_tile = 1; g.total_tiles = G.tiles_s;
      SUMK_TRY(launch_gemm(GEMM_NT, EPI_NONE, g, stream));
    }
    hipLaunchKernelGGL(tf_softmax_bwd_kernel, dim3((unsigned)(((int64_t)R * n_heads + 3) / 4)), dim3(256), 0, stream, P, E2, seq,
                       seq_off_dev, n_seq, R, n_heads, att_scale, dl, site + 0);
    {
      GemmLaunch g; g.precision = opts->precision;
      g.A = E2; g.B[0] = QKV; g.C = dQKV; g.probs = tabs + (size_t)TT_DQ * np; g.nprob = np; g.small_tile = 1; g.total_tiles = G.tiles_pv;
      SUMK_TRY(launch_gemm(GEMM_NN, EPI_NONE, g, stream));
    }
    {
      GemmLaunch g; g.precision = opts->precision;
      g.A = E2; g.B[0] = QKV; g.C = dQKV; g.probs = tabs + (size_t)TT_DK * np; g.nprob = np; g.small_tile = 1; g.total_tiles = G.tiles_pv;
      SUMK_TRY(launch_gemm(GEMM_TN, EPI_NONE, g, stream));
    }
    SUMK_TRY(colsum_accum(dQKV, 3 * D, R, 3 * D, colpart, TF_COLSUM_CHUNKS, Gd.in_proj_b, stream));
    SUMK_TRY(wgrad(dQKV, 3 * D, 3 * D, hin, D, D, Gd.in_proj_w, nullptr, nullptr, 3 * D));
    SUMK_TRY(nn(dQKV, W.in_proj_w, dT1a, P_DQKV, EPI_ACCUM, D));             // dHin = dT1a + dQKV . Win   (Win stored (3D,D) = (K,N))
    // dH (= dT1a buffer) now holds the gradient w.r.t. this layer's input = the previous layer's output
  }
  if (dx) {
    if (opts->more_residuals) {
      int64_t n4 = nRD >> 2;
      hipLaunchKernelGGL(add_rows_kernel, blocks(n4), dim3(256), 0, stream, dx, dH, n4);
    } else {
      SUMK_HIP(hipMemcpyAsync(dx, dH, (size_t)nRD * 4, hipMemcpyDeviceToDevice, stream));
    }
  }
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}
