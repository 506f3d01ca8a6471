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
#include "gemm_device.h"
#include <math.h>
#include <algorithm>
#include <cstdlib>

namespace sumk {

// per-video problem tables built on device by vasnet_setup_kernel (index = table id)
enum { TB_S = 0, TB_PV = 1, TB_DV = 2, TB_DP = 3, TB_DQ = 4, TB_DK = 5, TB_COUNT = 6 };
constexpr int ROW_PROBS = 8;
constexpr int SPLITK_PROBS = 64;
constexpr int COLSUM_CHUNKS = 128;

// Workspace carve-up, computed identically by the size query and by forward/backward.
struct VasnetWs {
  size_t qkv, e, ctx, y0, y1, z, seq, prob_row, prob_seq, prob_dvk, prob_skw, stats, scores, row_seq, total;
  size_t total_core;     // without the bf16 shadows at the end (all a step needs unless it runs on the bf16-source kernels)
  // training-only buffers
  size_t e2, dz, dy1, dy0, dctx, dqkv, lnpart, colpart, slab, prob_sk;
  // bf16 shadows of the operands of the row-wise GEMMs (training; used when the step runs on the bf16-source kernels)
  size_t x16, w16, ctx16, y116, dz16, dy016, dqkv16, qkv16, dctx16, p16, s16;
  size_t slab_elems;
  int64_t e_elems;
  int32_t n_rows;
  // small-batch path (SK launches, gemm_lean.hip): partial tiles, tickets, sliced problem tables -- present when sk_rows_ok(n_rows)
  size_t sk_part, sk_cnt, sk_tabs, sk_part_bytes;
  // seq, row_seq, prob_row, prob_seq, sk_cnt, sk_tabs form ONE block at the FRONT of the workspace (tab_bytes): everything the setup
  // kernels write and nothing else.  A caller may keep that block in a buffer of its own (sumk_vasnet_opts::tables, built once per
  // batch geometry by sumk_vasnet_build_tables): the offsets are the same, the base pointer differs, and no setup kernel runs per call.
  size_t tab_bytes;
};

// ---- small-batch path: every GEMM of the step as an in-launch split-K launch on 64x64 tiles (gemm_lean.hip, SK instances) ----
// Taken for batches of at most SK_MAX_ROWS frames (one to three TVSum-sized videos: the reference's one-video-per-call pattern),
// exact fp32 only.  Summation order differs from the large-batch kernels (K slices added in slice order), so scores agree with
// them to fp32 rounding (~1e-7), not bit for bit; each path is deterministic and independent of what else is in the batch.
constexpr int SK_MAX_ROWS = 1024;
constexpr int SK_TICKETS = 4096;
// row-wise SK tables (entries: groups x slices each) and where they start in the table region, in entries
enum { SR_QKV = 0, SR_OPROJ, SR_K1, SR_DY1, SR_DCTX, SR_DWO1, SR_DWQKV, SR_COUNT };
constexpr int SK_ROW_ENTRIES = 3 * SK_MAX_SLICES;
static inline bool sk_rows_ok(int64_t R) { return R <= SK_MAX_ROWS; }

static inline int round4(int v) { return (v + 3) & ~3; }

static int carve(int D, int n_seq, const int32_t* off, int training, VasnetWs* w) {
  SUMK_ARG(D > 0 && D % 4 == 0, "vasnet: D=%d must be a positive multiple of 4", D);
  SUMK_ARG(n_seq > 0 && off != nullptr, "vasnet: empty batch");
  SUMK_ARG(off[0] == 0, "vasnet: seq_off[0] must be 0");
  int64_t e = 0, e16 = 0;
  for (int s = 0; s < n_seq; ++s) {
    int T = off[s + 1] - off[s];
    SUMK_ARG(T > 0, "vasnet: video %d has %d frames", s, T);
    SUMK_ARG(T < (1 << 20), "vasnet: video %d has %d frames (limit 2^20)", s, T);
    e += (int64_t)T * round4(T);
    e16 += (int64_t)T * ((T + 63) & ~63);
  }
  const size_t R = (size_t)off[n_seq];
  size_t p = 0;
  auto take = [&](size_t bytes) { size_t at = p; p += align_up(bytes, 256); return at; };
  w->n_rows = (int32_t)R; w->e_elems = e;
  // the table block (see VasnetWs::tab_bytes)
  w->seq = take((size_t)n_seq * sizeof(SeqInfo));
  w->prob_row = take(ROW_PROBS * sizeof(GemmProb));
  w->prob_seq = take((size_t)TB_COUNT * n_seq * sizeof(GemmProb));
  w->prob_dvk = take((size_t)2 * n_seq * sizeof(GemmProb));      // dV and dK of the fused-attention bf16 step as ONE table (launch_setup)
  w->row_seq = take(R * 4);          // video of every packed row (vasnet_setup_kernel): one load instead of a binary search per row
  w->prob_skw = take((size_t)2 * SPLITK_PROBS * sizeof(GemmProb));   // K-slice tables of the two weight-gradient launches (dWo + dW1, dWqkv): built with the rest
  w->sk_cnt = w->sk_tabs = 0;
  if (sk_rows_ok((int64_t)R)) {
    w->sk_cnt = take((size_t)SK_TICKETS * 4);
    w->sk_tabs = take(((size_t)SR_COUNT * SK_ROW_ENTRIES + (size_t)TB_COUNT * n_seq * SK_MAX_SLICES) * sizeof(GemmProb));
  }
  w->tab_bytes = p;
  w->qkv = take(R * 3 * D * 4);
  w->e = take((size_t)e * 4);
  w->ctx = take(R * D * 4);
  w->y0 = take(R * D * 4);
  w->y1 = take(R * D * 4);
  w->z = take(R * D * 4);
  w->stats = take(R * 4 * 4);  // mean/rstd of both LayerNorm applications (training)
  w->scores = take(R * 4);
  w->e2 = w->dz = w->dy1 = w->dy0 = w->dctx = w->dqkv = w->lnpart = w->colpart = w->slab = w->prob_sk = 0;
  w->slab_elems = 0;
  w->x16 = w->w16 = w->ctx16 = w->y116 = w->dz16 = w->dy016 = w->dqkv16 = w->qkv16 = w->dctx16 = w->p16 = w->s16 = 0;
  w->sk_part = w->sk_part_bytes = 0;
  auto take_sk = [&]() {     // small-batch path: inside the core size (every arithmetic's workspace holds it)
    if (!sk_rows_ok((int64_t)R)) return;
    // scratch of the SK launches: partial tiles of the slices that meet inside a launch, or the K-slice slabs a row kernel adds
    // (partial tiles: 4 slices of the QKV projection or 8 of a (R, D) one -- sk_plan's caps; slabs: 8 of (R, D) or of the E layout)
    const size_t t64 = (R + 63) / 64, part_tiles = std::max(4 * t64 * ((3 * (size_t)D + 63) / 64), 8 * t64 * (((size_t)D + 63) / 64)) + 64;
    w->sk_part_bytes = std::max(part_tiles * 64 * 64, (size_t)SK_MAX_SLICES * std::max(R * (size_t)D, (size_t)e)) * 4;
    w->sk_part = take(w->sk_part_bytes);
  };
  if (training) {
    w->e2 = take((size_t)e * 4);     // dropped-out alpha in forward, then dAlpha / dLogits in backward
    w->dz = take(R * D * 4);
    w->dy1 = take(R * D * 4);
    w->dy0 = take(R * D * 4);
    w->dctx = take(R * D * 4);
    w->dqkv = take(R * 3 * D * 4);
    w->lnpart = take((size_t)(LNB_MAX_WAVES / 4) * ln_slot_floats(D) * 4);
    w->colpart = take((size_t)COLSUM_CHUNKS * D * 4);
    w->slab_elems = (size_t)32 * D * D;
    w->slab = take(w->slab_elems * 4);
    w->prob_sk = take(SPLITK_PROBS * sizeof(GemmProb));
    take_sk();
    w->total_core = p;
    w->x16 = take(R * D * 2);
    w->w16 = take((size_t)5 * D * D * 2);      // [Wq; Wk; Wv] stacked (one 3D x D operand), Wo, W1
    w->ctx16 = take(R * D * 2);
    w->y116 = take(R * D * 2);
    w->dz16 = take(R * D * 2);
    w->dy016 = take(R * D * 2);
    w->dqkv16 = take(R * 3 * D * 2);
    w->qkv16 = take(R * 3 * D * 2);
    w->dctx16 = take(R * D * 2);
    w->p16 = take((size_t)e16 * 2);            // per video (T x ld16): alpha (dropped-out alpha) in the forward, dLogits in the backward
    w->s16 = take((size_t)e16 * 2);            // fused attention strips: bf16(dLogits) beside P16 (dV and dK then run as ONE launch after the backward strip)
  }
  if (!training) take_sk();
  w->total = p;
  if (!training) w->total_core = p;
  return SUMK_OK;
}

// ------------------------------------------------------------------------------------------- setup tables
// One launch builds every table of the call: per-video SeqInfo + the six per-video GEMM tables (threads 0..n_seq-1,
// each computing its own prefix sums -- O(n_seq^2) reads of the tiny offset array, but fully parallel), and the
// row-wise single-problem entries (threads of block 1).
struct RowProbSpec { int32_t M, N, K, lda, ldb, ldc, ldr, small; };
struct SetupArgs {
  int32_t* row_seq;      // [n_rows] video index of every packed row
  int32_t p16;           // 1: the attention operand of the PV / dV / dQ / dK tables is the bf16 block (e16off, ld16) instead of (eoff, ldE)
  int32_t fake_seq0;     // diagnostic builds only: every video READS video 0's Q / K / V rows (operands stay L2-resident)
  const int32_t* off; int32_t n_seq, D;
  SeqInfo* seq; GemmProb* tabs;   // tabs[TB_COUNT][n_seq]
  int32_t s_tm, s_tn, pv_tm, pv_tn;  // block-tile dims of the (T x T) and (T x D) per-video products
  GemmProb* prow; RowProbSpec rows[ROW_PROBS]; int32_t n_rowprobs;
  // fused-attention bf16 step: dV = P^T dC and dK = dS^T Q as ONE table of 2 n_seq problems for one launch (A base = P16, B base = dCTX16,
  // C base = dQKV16); the dK entries reach dS (kept beside P, not over it) and Q through these element offsets between the buffers
  GemmProb* dvk; int64_t dk_a_delta, dk_b_delta;
};

__device__ inline void put_prob(GemmProb* p, int64_t a_off, int64_t b_off, int64_t c_off, int M, int N, int K, int lda,
                                int ldb, int ldc, int tile_start, int tiles_n) {
  GemmProb q;
  q.a_off = a_off; q.b_off = b_off; q.c_off = c_off; q.r_off = 0;
  q.M = M; q.N = N; q.K = K; q.lda = lda; q.ldb = ldb; q.ldc = ldc; q.ldr = 0;
  q.tile_start = tile_start; q.tiles_n = tiles_n;
  for (int i = 0; i < 7; ++i) q.pad_[i] = 0;
  *p = q;
}

__global__ void vasnet_setup_kernel(SetupArgs a) {
  if (blockIdx.y >= 2) {   // row -> video table: every row by binary search, rows dealt over the extra blocks (one thread per video writing
    // its T rows one after the other made this the longest part of the launch)
    const int n_rows = a.off[a.n_seq];
    for (int r = (blockIdx.y - 2) * blockDim.x + threadIdx.x; r < n_rows; r += (gridDim.y - 2) * blockDim.x) {
      if (blockIdx.x != 0) break;
      int lo = 0, hi = a.n_seq - 1;
      while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (a.off[mid] <= r) lo = mid; else hi = mid - 1; }
      a.row_seq[r] = lo;
    }
    return;
  }
  if (blockIdx.y == 1) {
    int i = threadIdx.x;
    if (blockIdx.x == 0 && i < a.n_rowprobs) {
      const RowProbSpec r = a.rows[i];
      int t = r.small == 3 ? 256 : r.small ? 64 : 128;     // (3: the wide bf16-source tiles)
      GemmProb q;
      q.a_off = q.b_off = q.c_off = q.r_off = 0;
      q.M = r.M; q.N = r.N; q.K = r.K; q.lda = r.lda; q.ldb = r.ldb; q.ldc = r.ldc; q.ldr = r.ldr;
      q.tile_start = 0; q.tiles_n = (r.N + t - 1) / t;
      for (int k = 0; k < 7; ++k) q.pad_[k] = 0;
      a.prow[i] = q;
    }
    return;
  }
  // Every thread needs three prefix sums over the videos before its own (logit elements, tiles of the two per-video products).
  // The per-video terms -- with their integer divisions -- are computed ONCE per block into LDS, one video per thread and
  // pass; each thread then just adds up the terms before its own video.  (Recomputing the terms inside an O(n_seq) loop per
  // thread, from global memory, made this tiny kernel take 12 us per call.)
  constexpr int STAGE_MAX = 2048;
  __shared__ int32_t sT[STAGE_MAX];      // frames of video q
  __shared__ int32_t sTs[STAGE_MAX];     // tiles of its (T x T) product
  __shared__ int32_t sTpv[STAGE_MAX];    // tiles of its (T x D) product
  const int D = a.D, tn = (D + a.pv_tn - 1) / a.pv_tn;
  const bool staged = a.n_seq <= STAGE_MAX;
  if (staged) {
    for (int q = threadIdx.x; q < a.n_seq; q += blockDim.x) {
      const int T = a.off[q + 1] - a.off[q];
      sT[q] = T;
      sTs[q] = ((T + a.s_tm - 1) / a.s_tm) * ((T + a.s_tn - 1) / a.s_tn);
      sTpv[q] = ((T + a.pv_tm - 1) / a.pv_tm) * tn;
    }
    __syncthreads();
  }
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= a.n_seq) return;
  int64_t eoff = 0, e16off = 0;
  int ts = 0, tpv = 0;
  if (staged) {
    for (int q = 0; q < s; ++q) { const int T = sT[q]; eoff += (int64_t)T * ((T + 3) & ~3); e16off += (int64_t)T * ((T + 63) & ~63); ts += sTs[q]; tpv += sTpv[q]; }
  } else {
    for (int q = 0; q < s; ++q) {
      const int T = a.off[q + 1] - a.off[q];
      eoff += (int64_t)T * ((T + 3) & ~3);
      e16off += (int64_t)T * ((T + 63) & ~63);
      ts += ((T + a.s_tm - 1) / a.s_tm) * ((T + a.s_tn - 1) / a.s_tn);
      tpv += ((T + a.pv_tm - 1) / a.pv_tm) * tn;
    }
  }
  const int row0 = a.off[s], T = a.off[s + 1] - a.off[s], ldE = (T + 3) & ~3;
  const int tm = (T + a.s_tn - 1) / a.s_tn;   // tiles along N of the (T x T) products
  SeqInfo si; si.eoff = eoff; si.row0 = row0; si.T = T; si.ldE = ldE; si.pad_ = 0; si.e16off = e16off;
  const int64_t po = a.p16 ? e16off : eoff;           // the attention matrix as a GEMM operand
  const int ldp = a.p16 ? ((T + 63) & ~63) : ldE;
  a.seq[s] = si;
  const int64_t q0 = (int64_t)row0 * 3 * D, c0 = (int64_t)row0 * D;
  const int n = a.n_seq;
  const int64_t qr = a.fake_seq0 ? 0 : q0;     // where Q / K / V are READ (q0 except in the diagnostic aliasing experiment)
  // forward
  put_prob(a.tabs + TB_S * n + s, qr, qr + D, eoff, T, T, D, 3 * D, 3 * D, ldE, ts, tm);            // E = Q K^T        (NT)
  put_prob(a.tabs + TB_PV * n + s, po, qr + 2 * D, c0, T, D, T, ldp, 3 * D, D, tpv, tn);             // C = alpha V      (NN)
  a.tabs[TB_PV * n + s].r_off = c0; a.tabs[TB_PV * n + s].ldr = D;   // folded inference path: + X in the epilogue (same rows as C)
  // backward
  put_prob(a.tabs + TB_DV * n + s, po, c0, q0 + 2 * D, T, D, T, ldp, D, 3 * D, tpv, tn);             // dV = alpha^T dC  (TN)
  put_prob(a.tabs + TB_DP * n + s, c0, q0 + 2 * D, eoff, T, T, D, D, 3 * D, ldE, ts, tm);            // dAlpha = dC V^T  (NT)
  put_prob(a.tabs + TB_DQ * n + s, po, q0 + D, q0, T, D, T, ldp, 3 * D, 3 * D, tpv, tn);             // dQ = dS K        (NN)
  put_prob(a.tabs + TB_DK * n + s, po, q0, q0 + D, T, D, T, ldp, 3 * D, 3 * D, tpv, tn);             // dK = dS^T Q      (TN)
  if (a.dvk != nullptr) {
    int tpv_all = 0;                                    // tiles of the whole dV table: where the dK entries' tiles start
    if (staged) { for (int q = 0; q < n; ++q) tpv_all += sTpv[q]; }
    else { for (int q = 0; q < n; ++q) { const int Tq = a.off[q + 1] - a.off[q]; tpv_all += ((Tq + a.pv_tm - 1) / a.pv_tm) * tn; } }
    put_prob(a.dvk + s, po, c0, q0 + 2 * D, T, D, T, ldp, D, 3 * D, tpv, tn);
    put_prob(a.dvk + n + s, po + a.dk_a_delta, q0 + a.dk_b_delta, q0 + D, T, D, T, ldp, 3 * D, 3 * D, tpv_all + tpv, tn);
  }
}


__device__ __forceinline__ int find_seq(const int32_t* off, int n_seq, int row) {
  int lo = 0, hi = n_seq - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (off[mid] <= row) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// ------------------------------------------------------------------------------------------- small-batch (SK) tables
// K slices of ONE problem contracting over K when its launch asked for S_req: slices of whole 32-wide k-tiles (host: sk_slices)
__host__ __device__ inline void sk_slice(int K, int S_req, int* kc, int* S) {
  const int c = ((K + S_req - 1) / S_req + 31) / 32 * 32;
  *kc = c; *S = (K + c - 1) / c;
}
// ... of one VIDEO's sub-problem in per-video table t: at most S_req, never slices shorter than four k-tiles
__host__ __device__ inline void sk_slice_seq(int K, int S_req, int* kc, int* S) {
  int r = K / 128; r = r < 1 ? 1 : r; r = r < S_req ? r : S_req;
  sk_slice(K, r, kc, S);
}
// (M, N, K) of video T's sub-problem in per-video table t
__host__ __device__ inline void sk_seq_dims(int t, int T, int D, int* M, int* N, int* K) {
  if (t == TB_S || t == TB_DP) { *M = T; *N = T; *K = D; } else { *M = T; *N = D; *K = T; }
}
struct SkTab { int entries, blocks, tiles, S_req, S; };    // table entries, (tile, slice) blocks, tiles (= tickets), requested slices, slices of a K = S_req-defining problem
// per group g: a_off = g a_goff, c_off = g c_goff, B / C pointer g when bsel / csel.  slab != 0: slice s stores its own (M x N) matrix
// at c_off + s slab with a plain epilogue (no tickets: the consuming ROW kernel adds the slabs, SlabIn) instead of meeting the other
// slices in the launch.
struct SkRowSpec { int32_t M, N, K, lda, ldb, ldc, ldr, layout, groups, S_req, a_goff, c_goff, bsel, csel; int64_t slab, a_goff64, b_goff64; };   // a_goff64 / b_goff64: group offsets of A / B beyond 31 bits (two operands of one allocation)
struct SkSetupArgs {
  const int32_t* off; int32_t n_seq, D;
  SeqInfo* seq; int32_t* row_seq; unsigned* cnt;
  GemmProb* tabs;                    // [SR_COUNT][SK_ROW_ENTRIES], then [TB_COUNT][n_seq * SK_MAX_SLICES]
  SkRowSpec rows[SR_COUNT];
  int32_t S_seq[TB_COUNT];
  int64_t slab_seq[TB_COUNT];        // per-video tables: slab stride of the sliced output (0: the slices meet in the launch)
  int32_t tile;                      // tile edge the tables count in: 64 (gemm_lean.hip SK instances) or 32 (gemm_direct.hip)
};
static inline GemmProb* sk_row_tab(GemmProb* tabs, int r) { return tabs + (size_t)r * SK_ROW_ENTRIES; }
static inline GemmProb* sk_seq_tab(GemmProb* tabs, int t, int n_seq) { return tabs + (size_t)SR_COUNT * SK_ROW_ENTRIES + (size_t)t * n_seq * SK_MAX_SLICES; }

__device__ inline void put_sk(GemmProb* tab, int ent0, int blk0, int tile0, int layout, int64_t a_off, int64_t b_off, int64_t c_off,
                              int M, int N, int K, int lda, int ldb, int ldc, int ldr, int S_req, int bsel, int csel, int te, int64_t slab,
                              bool per_video = true) {
  const int tn = (N + te - 1) / te, tiles = ((M + te - 1) / te) * tn;
  int kc, S;
  if (per_video) sk_slice_seq(K, S_req, &kc, &S); else sk_slice(K, S_req, &kc, &S);
  for (int sl = 0; sl < S; ++sl) {
    const int k0 = sl * kc;
    GemmProb q;
    q.a_off = a_off + (layout == GEMM_TN ? (int64_t)k0 * lda : (int64_t)k0);
    q.b_off = b_off + (layout == GEMM_NT ? (int64_t)k0 : (int64_t)k0 * ldb);
    q.c_off = c_off + sl * slab; q.r_off = c_off;         // (the residual, where there is one, has the layout of C)
    q.M = M; q.N = N; q.K = min(kc, K - k0); q.lda = lda; q.ldb = ldb; q.ldc = ldc; q.ldr = ldr;
    q.tile_start = blk0 + sl * tiles; q.tiles_n = tn;
    for (int i = 0; i < 7; ++i) q.pad_[i] = 0;
    q.pad_[SK_N] = (S > 1 && slab == 0) ? S : 0; q.pad_[SK_IDX] = sl; q.pad_[SK_TILE0] = tile0; q.pad_[SK_BSEL] = bsel; q.pad_[SK_CSEL] = csel; q.pad_[SK_PART0] = blk0;
    tab[ent0 + sl] = q;
  }
}

// One launch per call: SeqInfo, the row -> video table, zeroed tickets and every sliced problem table of the step.
__global__ void vasnet_sk_setup_kernel(SkSetupArgs a) {
  const int n_rows = a.off[a.n_seq];
  if (blockIdx.y >= 2) {   // row -> video table, tickets
    const int i0 = (blockIdx.y - 2) * blockDim.x + threadIdx.x, stride = (gridDim.y - 2) * blockDim.x;
    for (int r = i0; r < n_rows; r += stride) a.row_seq[r] = find_seq(a.off, a.n_seq, r);
    for (int i = i0; i < SK_TICKETS; i += stride) a.cnt[i] = 0u;
    return;
  }
  if (blockIdx.y == 1) {   // row-wise tables: one thread per table
    const int r = threadIdx.x;
    if (r >= SR_COUNT) return;
    const SkRowSpec sp = a.rows[r];
    if (sp.groups <= 0) return;
    GemmProb* tab = a.tabs + (size_t)r * SK_ROW_ENTRIES;
    const int te = a.tile;
    const int tiles = ((sp.M + te - 1) / te) * ((sp.N + te - 1) / te);
    int kc, S;
    sk_slice(sp.K, sp.S_req, &kc, &S);
    for (int g = 0; g < sp.groups; ++g)
      put_sk(tab, g * S, g * S * tiles, g * tiles, sp.layout, (int64_t)g * (sp.a_goff + sp.a_goff64), (int64_t)g * sp.b_goff64, (int64_t)g * sp.c_goff, sp.M, sp.N, sp.K, sp.lda, sp.ldb,
             sp.ldc, sp.ldr, sp.S_req, sp.bsel ? g : 0, sp.csel ? g : 0, te, sp.slab, false);
    return;
  }
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= a.n_seq) return;
  const int D = a.D;
  int64_t eoff = 0, e16off = 0;
  int ent[TB_COUNT], blk[TB_COUNT], til[TB_COUNT];
  for (int t = 0; t < TB_COUNT; ++t) ent[t] = blk[t] = til[t] = 0;
  for (int q = 0; q < s; ++q) {
    const int T = a.off[q + 1] - a.off[q];
    eoff += (int64_t)T * ((T + 3) & ~3);
    e16off += (int64_t)T * ((T + 63) & ~63);
    for (int t = 0; t < TB_COUNT; ++t) {
      int M, N, K, kc, S;
      sk_seq_dims(t, T, D, &M, &N, &K);
      sk_slice_seq(K, a.S_seq[t], &kc, &S);
      const int tiles = ((M + a.tile - 1) / a.tile) * ((N + a.tile - 1) / a.tile);
      ent[t] += S; blk[t] += S * tiles; til[t] += tiles;
    }
  }
  const int row0 = a.off[s], T = a.off[s + 1] - a.off[s], ldE = (T + 3) & ~3;
  SeqInfo si; si.eoff = eoff; si.row0 = row0; si.T = T; si.ldE = ldE; si.pad_ = 0; si.e16off = e16off;
  a.seq[s] = si;
  const int64_t q0 = (int64_t)row0 * 3 * D, c0 = (int64_t)row0 * D;
  GemmProb* base = a.tabs + (size_t)SR_COUNT * SK_ROW_ENTRIES;
  const size_t cap = (size_t)a.n_seq * SK_MAX_SLICES;
  // forward (operands as in vasnet_setup_kernel)
  put_sk(base + TB_S * cap, ent[TB_S], blk[TB_S], til[TB_S], GEMM_NT, q0, q0 + D, eoff, T, T, D, 3 * D, 3 * D, ldE, 0, a.S_seq[TB_S], 0, 0, a.tile, a.slab_seq[TB_S]);           // E = Q K^T
  put_sk(base + TB_PV * cap, ent[TB_PV], blk[TB_PV], til[TB_PV], GEMM_NN, eoff, q0 + 2 * D, c0, T, D, T, ldE, 3 * D, D, D, a.S_seq[TB_PV], 0, 0, a.tile, a.slab_seq[TB_PV]);     // C = alpha V
  // backward
  put_sk(base + TB_DV * cap, ent[TB_DV], blk[TB_DV], til[TB_DV], GEMM_TN, eoff, c0, q0 + 2 * D, T, D, T, ldE, D, 3 * D, 0, a.S_seq[TB_DV], 0, 0, a.tile, a.slab_seq[TB_DV]);     // dV = alpha^T dC
  put_sk(base + TB_DP * cap, ent[TB_DP], blk[TB_DP], til[TB_DP], GEMM_NT, c0, q0 + 2 * D, eoff, T, T, D, D, 3 * D, ldE, 0, a.S_seq[TB_DP], 0, 0, a.tile, a.slab_seq[TB_DP]);     // dAlpha = dC V^T
  put_sk(base + TB_DQ * cap, ent[TB_DQ], blk[TB_DQ], til[TB_DQ], GEMM_NN, eoff, q0 + D, q0, T, D, T, ldE, 3 * D, 3 * D, 0, a.S_seq[TB_DQ], 0, 0, a.tile, a.slab_seq[TB_DQ]);     // dQ = dS K
  put_sk(base + TB_DK * cap, ent[TB_DK], blk[TB_DK], til[TB_DK], GEMM_TN, eoff, q0, q0 + D, T, D, T, ldE, 3 * D, 3 * D, 0, a.S_seq[TB_DK], 0, 0, a.tile, a.slab_seq[TB_DK]);     // dK = dS^T Q
}

// ------------------------------------------------------------------------------------------- wave helpers
// four consecutive bf16 (round to nearest even, v_cvt_pk_bf16_f32 -- what the plane GEMM kernels do to the same fp32 values)
__device__ __forceinline__ void store_bf16x4(unsigned short* p, float4 v) {
  typedef float f32x4_t __attribute__((ext_vector_type(4)));
  typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
  const f32x4_t f = {v.x, v.y, v.z, v.w};
  *reinterpret_cast<bf16x4_t*>(p) = __builtin_convertvector(f, bf16x4_t);
}
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

// Sum of n K-slice slabs (SlabIn), slab order, with all n loads in flight: n is kernel-uniform and one of 1, 2, 4, 8 (the host
// only builds such slab sets: sk_plan), so each case is a fully unrolled run of independent loads behind ONE scalar branch.  (A
// run-time trip count made these loads -- and the row kernels -- a chain of n dependent round trips: softmax 5.8 -> 13.5 us.)
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float add4(float a, float b) { return a + b; }
template <int NS, typename V>
__device__ __forceinline__ V slab_sum_n(const V* p, int64_t stride) {
  V t[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) t[s] = p[s * stride];
  V v = t[0];
#pragma unroll
  for (int s = 1; s < NS; ++s) v = add4(v, t[s]);
  return v;
}
template <typename V>
__device__ __forceinline__ V slab_sum(const V* p, int n, int64_t stride) {    // stride in units of V
  if (n <= 1) return p[0];
  if (n == 2) return slab_sum_n<2, V>(p, stride);
  if (n == 4) return slab_sum_n<4, V>(p, stride);
  return slab_sum_n<8, V>(p, stride);
}

// Small-batch path, register-resident rows: every slab / residual / bias load of the row's NQ column chunks is issued before the
// first add, so the row costs one memory round trip instead of one per chunk.  Columns past D4 are clamped (loaded, never used).
template <int NQ, int NS, bool ADD, bool BIAS, typename V>
__device__ __forceinline__ void ln_gather(V (&xr)[NQ], const V* x4, const V* add4p, const V* bias4, int lane, int D4, int64_t stride4) {
  V t[NQ][NS], ta[NQ], tb[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int c = min(lane + 64 * q, D4 - 1);
#pragma unroll
    for (int s = 0; s < NS; ++s) t[q][s] = x4[s * stride4 + c];
    if constexpr (ADD) ta[q] = add4p[c];
    if constexpr (BIAS) tb[q] = bias4[c];
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    V v = t[q][0];
#pragma unroll
    for (int s = 1; s < NS; ++s) v = add4(v, t[q][s]);      // slab order, as slab_sum
    if constexpr (ADD) v = add4(v, ta[q]);
    if constexpr (BIAS) v = add4(v, tb[q]);
    xr[q] = v;
  }
}
template <int NQ, int NS>
__device__ __forceinline__ void ln_gather_n(float4 (&xr)[NQ], const float4* x4, const float4* add4p, const float4* bias4, int lane,
                                            int D4, int64_t stride4) {
  if (add4p && bias4) ln_gather<NQ, NS, true, true>(xr, x4, add4p, bias4, lane, D4, stride4);
  else if (add4p) ln_gather<NQ, NS, true, false>(xr, x4, add4p, bias4, lane, D4, stride4);
  else if (bias4) ln_gather<NQ, NS, false, true>(xr, x4, add4p, bias4, lane, D4, stride4);
  else ln_gather<NQ, NS, false, false>(xr, x4, add4p, bias4, lane, D4, stride4);
}

// the slabs alone, set size chosen at run time (kernel-uniform)
template <int NQ, typename V>
__device__ __forceinline__ void row_gather(V (&xr)[NQ], const V* p, int lane, int n, int n_slab, int64_t stride) {
  if (n_slab <= 1) ln_gather<NQ, 1, false, false, V>(xr, p, nullptr, nullptr, lane, n, stride);
  else if (n_slab == 2) ln_gather<NQ, 2, false, false, V>(xr, p, nullptr, nullptr, lane, n, stride);
  else if (n_slab == 4) ln_gather<NQ, 4, false, false, V>(xr, p, nullptr, nullptr, lane, n, stride);
  else ln_gather<NQ, 8, false, false, V>(xr, p, nullptr, nullptr, lane, n, stride);
}

// ------------------------------------------------------------------------------------------- softmax rows
// One wave per query row.  Reads raw Q.K^T, writes alpha in place and zeroes the [T, ldE) pad so the alpha.V product
// can stream K in float4 units.  Training with dropout (vasnet.py:130) also writes dropout(alpha) to E2.
// NR > 0: the row (T <= 64 * NR keys) is held in registers -- one read of the logits, one exp per element, one write.  NR == 0:
// any length, three passes over the (L2-resident) row.  Same operations per element either way, so the results are identical.
// Eraw / n_slab / slab_stride: where the raw logits are read -- E itself (n_slab <= 1), or n_slab K-slice slabs of the E layout
// (small-batch path, SlabIn) that are added in slab order on load.
template <int NR, bool PRE = false>   // PRE (few rows: latency-bound): the row's slab loads are all issued before the first add
__global__ __launch_bounds__(256) void vasnet_softmax_kernel(float* E, float* E2, const SeqInfo* seq, const int32_t* off,
                                                             int n_seq, int n_rows, float scale, int ignore_self,
                                                             int aperture, Drop drop_in, unsigned short* P16,
                                                             const float* Eraw, int n_slab, int64_t slab_stride) {
  const Drop drop = drop_resolve(drop_in);
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const int lane = threadIdx.x & 63;
  const int s = n_seq == 1 ? 0 : off[row];        // `off` = the row -> video table of the setup kernel (VasnetWs::row_seq)
  const SeqInfo si = seq[s];
  const int i = row - si.row0, T = si.T;
  float* e = E + si.eoff + (int64_t)i * si.ldE;
  float* e2 = E2 ? E2 + si.eoff + (int64_t)i * si.ldE : nullptr;
  const float* er = Eraw + si.eoff + (int64_t)i * si.ldE;
  auto raw = [&](int j) { return slab_sum(er + j, n_slab, slab_stride); };
  // P16: the matrix the alpha.V / alpha^T.dC products read (dropout(alpha) when there is dropout), as bf16 with the row zero-padded
  // to ld16 = a whole number of 64-wide k-tiles
  const int ld16 = (T + 63) & ~63;
  unsigned short* p16 = P16 ? P16 + si.e16off + (int64_t)i * ld16 : nullptr;
  if constexpr (NR > 0) {
    float v[NR];
    float m = -INFINITY;
    if constexpr (PRE) row_gather<NR>(v, er, lane, T, n_slab, slab_stride);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int j = lane + 64 * r;
      if constexpr (PRE) v[r] = j < T ? masked_logit(v[r], scale, i, j, ignore_self, aperture) : -INFINITY;
      else v[r] = j < T ? masked_logit(raw(j), scale, i, j, ignore_self, aperture) : -INFINITY;
      m = fmaxf(m, v[r]);
    }
    m = wave_max(m);
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int j = lane + 64 * r;
      v[r] = j < T ? expf(v[r] - m) : 0.f;
      sum += v[r];
    }
    sum = wave_sum(sum);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int j = lane + 64 * r;
      const float a = j < T ? v[r] / sum : 0.f;
      float ad = a;
      if (e2 && drop.thr && j < T) ad = drop_apply(drop, 0, ((uint64_t)row << 20) | (uint64_t)j, a);
      if (j < si.ldE) {
        e[j] = a;
        if (e2) e2[j] = ad;
      }
      if (p16 && j < ld16) p16[j] = __builtin_bit_cast(unsigned short, (__bf16)ad);
    }
    return;
  }
  float m = -INFINITY;
  for (int j = lane; j < T; j += 64) m = fmaxf(m, masked_logit(raw(j), scale, i, j, ignore_self, aperture));
  m = wave_max(m);
  float sum = 0.f;
  for (int j = lane; j < T; j += 64) sum += expf(masked_logit(raw(j), scale, i, j, ignore_self, aperture) - m);
  sum = wave_sum(sum);
  for (int j = lane; j < (p16 ? ld16 : si.ldE); j += 64) {
    float v = 0.f;
    if (j < T) v = expf(masked_logit(raw(j), scale, i, j, ignore_self, aperture) - m) / sum;
    float vd = v;
    if (e2 && drop.thr && j < T) vd = drop_apply(drop, 0, ((uint64_t)row << 20) | (uint64_t)j, v);
    if (j < si.ldE) {
      e[j] = v;
      if (e2) e2[j] = vd;
    }
    if (p16) p16[j] = __builtin_bit_cast(unsigned short, (__bf16)vd);
  }
}

// dLogits(raw) = scale * alpha * (dAlpha - sum_j dAlpha_j alpha_j), dAlpha = dropout'(dAlphaDropped).  In place on E2.
// NR > 0 (few rows, T <= 64 NR): the row's alpha and dAlpha are loaded up front and held in registers; same per-lane order of
// operations as the streaming form (NR == 0), so the results are identical.
template <int NR>
__global__ __launch_bounds__(256) void vasnet_softmax_bwd_kernel(const float* E, float* E2, const SeqInfo* seq,
                                                                 const int32_t* off, int n_seq, int n_rows, float scale,
                                                                 Drop drop_in, unsigned short* S16,  // S16: bf16(dLogits), rows zero-padded to ld16
                                                                 const float* Graw, int n_slab, int64_t slab_stride) {   // the incoming dAlphaD: E2 itself, or K-slice slabs (SlabIn)
  const Drop drop = drop_resolve(drop_in);
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const int lane = threadIdx.x & 63;
  const int s = n_seq == 1 ? 0 : off[row];        // `off` = the row -> video table of the setup kernel (VasnetWs::row_seq)
  const SeqInfo si = seq[s];
  const int i = row - si.row0, T = si.T;
  const float* p = E + si.eoff + (int64_t)i * si.ldE;
  float* g = E2 + si.eoff + (int64_t)i * si.ldE;
  const float* gr = Graw + si.eoff + (int64_t)i * si.ldE;
  auto gin = [&](int j) { return slab_sum(gr + j, n_slab, slab_stride); };
  float dot = 0.f;
  if constexpr (NR > 0) {
    float d[NR], pv[NR];
    row_gather<NR>(d, gr, lane, T, n_slab, slab_stride);
#pragma unroll
    for (int r = 0; r < NR; ++r) pv[r] = p[min(lane + 64 * r, T - 1)];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int j = lane + 64 * r;
      if (j < T) {
        if (drop.thr) d[r] = drop_apply(drop, 0, ((uint64_t)row << 20) | (uint64_t)j, d[r]);
        dot += d[r] * pv[r];
      }
    }
    dot = wave_sum(dot);
    const int ld16 = (T + 63) & ~63;
    unsigned short* s16 = S16 ? S16 + si.e16off + (int64_t)i * ld16 : nullptr;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int j = lane + 64 * r;
      const float v = j < T ? pv[r] * (d[r] - dot) * scale : 0.f;
      if (j < si.ldE) g[j] = v;
      if (s16 && j < ld16) s16[j] = __builtin_bit_cast(unsigned short, (__bf16)v);
    }
    return;
  }
  for (int j = lane; j < T; j += 64) {
    float d = gin(j);
    if (drop.thr) d = drop_apply(drop, 0, ((uint64_t)row << 20) | (uint64_t)j, d);
    dot += d * p[j];
  }
  dot = wave_sum(dot);
  const int ld16 = (T + 63) & ~63;
  unsigned short* s16 = S16 ? S16 + si.e16off + (int64_t)i * ld16 : nullptr;
  for (int j = lane; j < (s16 ? ld16 : si.ldE); j += 64) {
    float v = 0.f;
    if (j < T) {
      float d = gin(j);
      if (drop.thr) d = drop_apply(drop, 0, ((uint64_t)row << 20) | (uint64_t)j, d);
      v = p[j] * (d - dot) * scale;
    }
    if (j < si.ldE) g[j] = v;
    if (s16) s16[j] = __builtin_bit_cast(unsigned short, (__bf16)v);
  }
}

// ------------------------------------------------------------------------------------------- LayerNorm rows
// y = (x - mean) * rstd * g + b over D, biased variance, one wave per row (torch.nn.LayerNorm, vasnet.py:54).
// `site` dropout (vasnet.py:136 / :142) is applied to x on load.  HEAD: no y is written; instead
// scores[r] = sigmoid(y . w2 + b2)  (vasnet.py:144-145).
// NQ > 0: the row (D <= 256 * NQ) is held in registers (NQ float4 per lane): ONE read of x instead of three.  NQ == 0: any D,
// three passes.  The per-lane accumulation order is the same, so both forms give identical results.
template <bool HEAD, int NQ, bool SLAB>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ X, float* __restrict__ Y,
                                                        const float* __restrict__ g, const float* __restrict__ b,
                                                        const float* __restrict__ w2, const float* __restrict__ b2,
                                                        float* __restrict__ scores, int n_rows, int D, float eps,
                                                        float* __restrict__ stats, Drop drop_in, uint32_t site,
                                                        unsigned short* __restrict__ Y16,     // !HEAD: bf16(y) too (Y may then be null)
                                                        SlabIn sl) {                          // small-batch path: X = K-slice slabs (+ residual / bias, ReLU)
  const Drop drop = drop_resolve(drop_in);
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const int lane = threadIdx.x & 63;
  const float4* x4 = reinterpret_cast<const float4*>(X + (int64_t)row * D);
  const int D4 = D >> 2;
  auto ld = [&](int c) {
    float4 v = slab_sum(x4 + c, sl.n, sl.stride >> 2);
    if (sl.n > 1 || sl.add || sl.bias) {     // kernel-uniform
      if (sl.add) { const float4 t = reinterpret_cast<const float4*>(sl.add + (int64_t)row * D)[c]; v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
      if (sl.bias) { const float4 t = reinterpret_cast<const float4*>(sl.bias)[c]; v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
      if (sl.relu) { v.x = v.x < 0.f ? 0.f : v.x; v.y = v.y < 0.f ? 0.f : v.y; v.z = v.z < 0.f ? 0.f : v.z; v.w = v.w < 0.f ? 0.f : v.w; }   // NaN-propagating, like torch.relu
      if (sl.store) reinterpret_cast<float4*>(sl.store + (int64_t)row * D)[c] = v;
    }
    if (drop.thr) {
      uint64_t base = (uint64_t)row * D + 4 * c;
      v.x = drop_apply(drop, site, base, v.x); v.y = drop_apply(drop, site, base + 1, v.y);
      v.z = drop_apply(drop, site, base + 2, v.z); v.w = drop_apply(drop, site, base + 3, v.w);
    }
    return v;
  };
  constexpr int NR = NQ > 0 ? NQ : 1;
  float4 xr[NR];
  const float4* g4 = reinterpret_cast<const float4*>(g);
  const float4* b4 = reinterpret_cast<const float4*>(b);
  const float4* w4 = reinterpret_cast<const float4*>(w2);
  float4 gq[NR], bq[NR], wq[NR];             // NQ > 0: the affine (and k2) rows, in flight with the row itself
  if constexpr (NQ > 0 && SLAB) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int c = min(lane + 64 * q, D4 - 1);
      gq[q] = g4[c]; bq[q] = b4[c];
      if constexpr (HEAD) wq[q] = w4[c];
    }
  }
  if constexpr (NQ > 0) {
    if constexpr (SLAB) {                    // SLAB == (sl.n > 1 || sl.add || sl.bias), chosen by the launcher
      const float4* a4 = sl.add ? reinterpret_cast<const float4*>(sl.add + (int64_t)row * D) : nullptr;
      const float4* bi4 = reinterpret_cast<const float4*>(sl.bias);
      const int64_t st4 = sl.stride >> 2;
      if (sl.n <= 1) ln_gather_n<NQ, 1>(xr, x4, a4, bi4, lane, D4, st4);
      else if (sl.n == 2) ln_gather_n<NQ, 2>(xr, x4, a4, bi4, lane, D4, st4);
      else if (sl.n == 4) ln_gather_n<NQ, 4>(xr, x4, a4, bi4, lane, D4, st4);
      else ln_gather_n<NQ, 8>(xr, x4, a4, bi4, lane, D4, st4);
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int c = lane + 64 * q;
        float4 v = xr[q];
        if (sl.relu) { v.x = v.x < 0.f ? 0.f : v.x; v.y = v.y < 0.f ? 0.f : v.y; v.z = v.z < 0.f ? 0.f : v.z; v.w = v.w < 0.f ? 0.f : v.w; }   // NaN-propagating, like torch.relu
        if (sl.store && c < D4) reinterpret_cast<float4*>(sl.store + (int64_t)row * D)[c] = v;
        if (drop.thr) {
          uint64_t base = (uint64_t)row * D + 4 * c;
          v.x = drop_apply(drop, site, base, v.x); v.y = drop_apply(drop, site, base + 1, v.y);
          v.z = drop_apply(drop, site, base + 2, v.z); v.w = drop_apply(drop, site, base + 3, v.w);
        }
        xr[q] = c < D4 ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else {
#pragma unroll
      for (int q = 0; q < NQ; ++q) { const int c = lane + 64 * q; xr[q] = c < D4 ? ld(c) : make_float4(0.f, 0.f, 0.f, 0.f); }
    }
  }
  float s = 0.f;
  if constexpr (NQ > 0) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) { const int c = lane + 64 * q; if (c < D4) { float4 v = xr[q]; s += (v.x + v.y) + (v.z + v.w); } }
  } else {
    for (int c = lane; c < D4; c += 64) { float4 v = ld(c); s += (v.x + v.y) + (v.z + v.w); }
  }
  const float mean = wave_sum(s) / (float)D;
  float q2 = 0.f;
  auto sq = [&](float4 v) {
    float a = v.x - mean, bb = v.y - mean, cc = v.z - mean, d = v.w - mean;
    q2 += (a * a + bb * bb) + (cc * cc + d * d);
  };
  if constexpr (NQ > 0) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) { const int c = lane + 64 * q; if (c < D4) sq(xr[q]); }
  } else {
    for (int c = lane; c < D4; c += 64) sq(ld(c));
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q2) / (float)D + eps);
  if constexpr (!HEAD) {
    float4* y4 = reinterpret_cast<float4*>(Y + (int64_t)row * D);
    auto emit = [&](int c, float4 v, float4 gg, float4 bv) {
      float4 o;
      o.x = (v.x - mean) * rstd * gg.x + bv.x; o.y = (v.y - mean) * rstd * gg.y + bv.y;
      o.z = (v.z - mean) * rstd * gg.z + bv.z; o.w = (v.w - mean) * rstd * gg.w + bv.w;
      if (Y != nullptr) y4[c] = o;
      if (Y16 != nullptr) store_bf16x4(Y16 + (int64_t)row * D + 4 * c, o);
    };
    if constexpr (NQ > 0) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) { const int c = lane + 64 * q; if (c < D4) { if constexpr (SLAB) emit(c, xr[q], gq[q], bq[q]); else emit(c, xr[q], g4[c], b4[c]); } }
    } else {
      for (int c = lane; c < D4; c += 64) emit(c, ld(c), g4[c], b4[c]);
    }
  } else {
    float dot = 0.f;
    auto acc = [&](float4 v, float4 gg, float4 bv, float4 ww) {
      dot += ((v.x - mean) * rstd * gg.x + bv.x) * ww.x + ((v.y - mean) * rstd * gg.y + bv.y) * ww.y +
             ((v.z - mean) * rstd * gg.z + bv.z) * ww.z + ((v.w - mean) * rstd * gg.w + bv.w) * ww.w;
    };
    if constexpr (NQ > 0) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) { const int c = lane + 64 * q; if (c < D4) { if constexpr (SLAB) acc(xr[q], gq[q], bq[q], wq[q]); else acc(xr[q], g4[c], b4[c], w4[c]); } }
    } else {
      for (int c = lane; c < D4; c += 64) acc(ld(c), g4[c], b4[c], w4[c]);
    }
    dot = wave_sum(dot);
    if (lane == 0) {
      const float sc = 1.0f / (1.0f + expf(-(dot + b2[0])));
      scores[row] = sc;
      if (Y != nullptr) Y[row] = sc;        // (HEAD: Y = a second destination for the scores -- the copy the backward reads from the workspace)
    }
  }
  if (stats != nullptr && lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
}

// Tail of the fused inference path (k1 GEMM with EPI_BIAS_RELU_HEAD): per row the N / 32 slot moments {sum z, sum z^2, sum z g w2}
// are added (double) and  score = sigmoid( rstd (S3 - mean GW) + BW + b2 ),  mean = S1 / D, rstd = 1 / sqrt(S2 / D - mean^2 + eps),
// GW = sum g w2, BW = sum b w2  -- algebraically the LayerNorm + k2 of layernorm_kernel<true>, without ever storing z.
__global__ __launch_bounds__(256) void head_finalize_kernel(const float4* __restrict__ part, int slots, int n_rows, int D,
                                                            const float* __restrict__ g, const float* __restrict__ b,
                                                            const float* __restrict__ w2, const float* __restrict__ b2, float eps,
                                                            float* __restrict__ scores) {
  __shared__ double red[2][4];
  double gw = 0.0, bw = 0.0;
  for (int c = threadIdx.x; c < D; c += 256) { const double w = (double)w2[c]; gw += (double)g[c] * w; bw += (double)b[c] * w; }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { gw += __shfl_xor(gw, m, 64); bw += __shfl_xor(bw, m, 64); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = gw; red[1][threadIdx.x >> 6] = bw; }
  __syncthreads();
  gw = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
  bw = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  // 8 lanes per row (32 rows per block): each adds every 8th slot, then a 3-step butterfly inside the 8-lane group
  const int row = blockIdx.x * 32 + (threadIdx.x >> 3), sub = threadIdx.x & 7;
  double s1 = 0.0, s2 = 0.0, s3 = 0.0;
  if (row < n_rows)
    for (int k = sub; k < slots; k += 8) { const float4 p = part[(int64_t)row * slots + k]; s1 += p.x; s2 += p.y; s3 += p.z; }
#pragma unroll
  for (int m = 1; m <= 4; m <<= 1) { s1 += __shfl_xor(s1, m, 64); s2 += __shfl_xor(s2, m, 64); s3 += __shfl_xor(s3, m, 64); }
  if (row >= n_rows || sub != 0) return;
  const double mean = s1 / D, var = s2 / D - mean * mean;
  const double rstd = 1.0 / sqrt((var > 0.0 ? var : 0.0) + (double)eps);
  const double pre = rstd * (s3 - mean * gw) + bw + (double)b2[0];
  scores[row] = (float)(1.0 / (1.0 + exp(-pre)));
}

// Prep launch of the fused inference tail: blocks [0, D) fold the first LayerNorm's gain into the k1 weights,
//   W1g[n][k] = W1[n][k] gamma[k],  c1[n] = sum_k gamma[k] W1[n][k],  c2[n] = sum_k beta[k] W1[n][k]     (one block per output row n),
// so that  LN(y) . W1[n]  =  rstd (y . W1g[n] - mean c1[n]) + c2[n]  and the k1 GEMM can read the raw Y0; the remaining blocks turn
// the out-projection's per-row slot moments into {mean, rstd} (8 lanes per row, double).  (The fold depends on weights only; it is
// redone per call -- 8 MB of traffic inside a launch that is needed anyway -- rather than cached across calls.)
__global__ __launch_bounds__(256) void ln_fold_stats_kernel(const float* __restrict__ W1, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int D, float* __restrict__ W1g,
                                                            float* __restrict__ c1, float* __restrict__ c2,
                                                            const float2* __restrict__ moments, int slots, int n_rows, float eps,
                                                            float2* __restrict__ stats) {
  if ((int)blockIdx.x < D) {
    __shared__ double red[2][4];
    const int n = blockIdx.x;
    double a1 = 0.0, a2 = 0.0;
    for (int k = 4 * threadIdx.x; k < D; k += 1024) {
      const float4 w = *reinterpret_cast<const float4*>(W1 + (int64_t)n * D + k);
      const float4 g = *reinterpret_cast<const float4*>(gamma + k), b = *reinterpret_cast<const float4*>(beta + k);
      float4 o; o.x = w.x * g.x; o.y = w.y * g.y; o.z = w.z * g.z; o.w = w.w * g.w;
      *reinterpret_cast<float4*>(W1g + (int64_t)n * D + k) = o;
      a1 += ((double)o.x + (double)o.y) + ((double)o.z + (double)o.w);
      a2 += ((double)w.x * b.x + (double)w.y * b.y) + ((double)w.z * b.z + (double)w.w * b.w);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { a1 += __shfl_xor(a1, m, 64); a2 += __shfl_xor(a2, m, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a1; red[1][threadIdx.x >> 6] = a2; }
    __syncthreads();
    if (threadIdx.x == 0) {
      c1[n] = (float)((red[0][0] + red[0][1]) + (red[0][2] + red[0][3]));
      c2[n] = (float)((red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
    }
    return;
  }
  const int row = ((int)blockIdx.x - D) * 32 + (threadIdx.x >> 3), sub = threadIdx.x & 7;
  double s1 = 0.0, s2 = 0.0;
  if (row < n_rows)
    for (int k = sub; k < slots; k += 8) { const float2 p = moments[(int64_t)row * slots + k]; s1 += p.x; s2 += p.y; }
#pragma unroll
  for (int m = 1; m <= 4; m <<= 1) { s1 += __shfl_xor(s1, m, 64); s2 += __shfl_xor(s2, m, 64); }
  if (row >= n_rows || sub != 0) return;
  const double mean = s1 / D, var = s2 / D - mean * mean;
  stats[row] = make_float2((float)mean, (float)(1.0 / sqrt((var > 0.0 ? var : 0.0) + (double)eps)));
}

// {mean, rstd} per row from the slot moments of a GEMM epilogue (the second half of ln_fold_stats_kernel by itself: the plane path keeps
// the weight fold in its per-weight-change block, sumk_vasnet_wplanes_build)
__global__ __launch_bounds__(256) void ln_row_stats_kernel(const float2* __restrict__ moments, int slots, int n_rows, int D, float eps,
                                                           float2* __restrict__ stats) {
  const int row = (int)blockIdx.x * 32 + (threadIdx.x >> 3), sub = threadIdx.x & 7;
  double s1 = 0.0, s2 = 0.0;
  if (row < n_rows)
    for (int k = sub; k < slots; k += 8) { const float2 p = moments[(int64_t)row * slots + k]; s1 += p.x; s2 += p.y; }
#pragma unroll
  for (int m = 1; m <= 4; m <<= 1) { s1 += __shfl_xor(s1, m, 64); s2 += __shfl_xor(s2, m, 64); }
  if (row >= n_rows || sub != 0) return;
  const double mean = s1 / D, var = s2 / D - mean * mean;
  stats[row] = make_float2((float)mean, (float)(1.0 / sqrt((var > 0.0 ? var : 0.0) + (double)eps)));
}
// biasc = b1 + c2 (k1's bias + the LayerNorm shift seen through W1), gw = ln_w * w2: the per-column vectors of the PW_HEAD epilogue
__global__ void head_vectors_kernel(const float* __restrict__ b1, const float* __restrict__ c2, const float* __restrict__ ln_w,
                                    const float* __restrict__ w2, int D, float* __restrict__ biasc, float* __restrict__ gw) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n < D) { biasc[n] = b1[n] + c2[n]; gw[n] = ln_w[n] * w2[n]; }
}

// Weight-plane block of the plane path (sumk_vasnet_opts::wplanes): everything that depends on the weights only.
struct WPlanes { size_t wqkv, wo, w1g, c1, biasc, gw, tmp, total; };
static WPlanes wplanes_layout(int D, int np) {
  WPlanes l; size_t p = 0;
  auto take = [&](size_t bytes) { size_t at = p; p += align_up(bytes, 256); return at; };
  l.wqkv = take(pw_planes_bytes(3 * (int64_t)D, D, np)); l.wo = take(pw_planes_bytes(D, D, np)); l.w1g = take(pw_planes_bytes(D, D, np));
  l.c1 = take((size_t)D * 4); l.biasc = take((size_t)D * 4); l.gw = take((size_t)D * 4);
  l.tmp = take((size_t)D * D * 4 + (size_t)D * 4);        // build scratch: fp32 W1 diag(ln_w), c2
  l.total = p;
  return l;
}
// Extra workspace of the plane path when the per-video attention runs on planes too (attn_pw.hip): [Q | K | V] planes (later re-used for
// Y0's planes), alpha planes, context planes -- appended behind the regular carve-up (sumk_vasnet_workspace_bytes_for adds it for
// inference in bf16x6 / bf16x3 when the batch is eligible).
struct PwExtra { size_t qkv, ap, ctx, total; };
static PwExtra pw_extra(int D, int64_t R, int t_max, int np, size_t base) {
  PwExtra e; size_t p = align_up(base, 256);
  auto take = [&](size_t bytes) { size_t at = p; p += align_up(bytes, 256); return at; };
  e.qkv = take(pw_planes_bytes(R, 3 * D, np)); e.ap = take(pw_alpha_bytes(R, t_max, np)); e.ctx = take(pw_planes_bytes(R, D, np));
  e.total = p;
  return e;
}
// Long videos on planes (round 6; BASELINE config 5): one video at a time -- raw logits E_s (T x Tn fp32, Tn = T rounded up to the plane GEMM's
// 256-column tile), planes of Q_s, K_s (T x D), of V_s^T (D x Kp) and of alpha_s (T x Kp), Kp = T rounded up to 32 -- behind the regular carve-up.
constexpr int PW_LONG_TMIN = 1536;       // below this a (T x T) product does not fill the chip with 192 x 256 tiles: the in-loop grouped kernels keep it
struct PwLong { size_t e, qp, kp, vt, ap, st, qk, total; int tn, kpad, qk_direct; };
static PwLong pw_long_layout(int D, int64_t R, int t_max, int np, size_t base) {
  PwLong e; size_t p = align_up(base, 256);
  auto take = [&](size_t bytes) { size_t at = p; p += align_up(bytes, 256); return at; };
  e.tn = (t_max + 255) / 256 * 256; e.kpad = (t_max + 31) / 32 * 32;
  e.e = take((size_t)t_max * e.tn * 4);
  // the planes of [Q | K] of the WHOLE batch straight from the projection's epilogue when that array stays below 2 GiB (every video's Q / K are row ranges of
  // it); otherwise the projection leaves as fp32 and each video's Q and K are split on their own (qp / kp)
  e.qk_direct = (R >= e.tn && pw_ok(R, 2 * (int64_t)D, D, R, 3 * (int64_t)D, np) && (int64_t)pw_planes_bytes(R, 2 * D, np) < (((int64_t)1 << 31) - 65536)) ? 1 : 0;      // (R >= Tn: a lone video's logits tiles read key rows up to Tn)
  e.qk = take(e.qk_direct ? pw_planes_bytes(R, 2 * D, np) : 0);
  e.qp = take(e.qk_direct ? 0 : pw_planes_bytes(e.tn, D, np)); e.kp = take(e.qk_direct ? 0 : pw_planes_bytes(e.tn, D, np));          // (rows up to Tn: the logits tile reads K rows up to there)
  e.vt = take(pw_planes_bytes(D, e.kpad, np)); e.ap = take(pw_planes_bytes(t_max, e.kpad, np));
  e.st = take((size_t)t_max * 8);                   // {max, sum} per query row
  e.total = p;
  return e;
}
static bool pw_long_ok(int D, int t_min, int t_max, int np) {
  if (t_min < PW_LONG_TMIN || D % 256 != 0) return false;
  const int tn = (t_max + 255) / 256 * 256, kpad = (t_max + 31) / 32 * 32;
  return pw_ok(t_max, tn, D, tn, tn, np) && pw_ok(t_max, D, kpad, t_max, D, np);
}
static bool wplanes_ok(int D, int np) { return D >= 256 && D % 256 == 0 && (np == 2 || np == 3) && pw_ok(256, 3 * (int64_t)D, D, 256, 3 * (int64_t)D, np); }

// one launcher for every LayerNorm call site: picks the register-resident form when the row fits (D <= 2048)
template <bool HEAD>
static void launch_ln_rows(const float* X, float* Y, const float* g, const float* b, const float* w2, const float* b2, float* scores,
                           int n_rows, int D, float eps, float* stats, Drop drop, uint32_t site, hipStream_t stream,
                           unsigned short* y16 = nullptr, SlabIn sl = SlabIn()) {
  const dim3 grid((n_rows + 3) / 4), block(256);
  const int D4 = D >> 2;
  const bool slab = sl.n > 1 || sl.add || sl.bias;
#define SUMK_LN_(NQ, SL) hipLaunchKernelGGL((layernorm_kernel<HEAD, NQ, SL>), grid, block, 0, stream, X, Y, g, b, w2, b2, scores, n_rows, D, eps, stats, drop, site, y16, sl)
#define SUMK_LN(NQ) do { if (slab) SUMK_LN_(NQ, true); else SUMK_LN_(NQ, false); } while (0)
  if (D4 <= 64) SUMK_LN(1); else if (D4 <= 128) SUMK_LN(2); else if (D4 <= 256) SUMK_LN(4); else if (D4 <= 512) SUMK_LN(8); else SUMK_LN(0);
#undef SUMK_LN
#undef SUMK_LN_
}

// Backward of  y = LN(drop(x)) * g + b  for a strided set of rows per wave.
//   HEAD : upstream is the scalar du = dscore * s(1-s) through y . w2 + b2; x = Z (post-ReLU): dX also takes the ReLU mask.
//   !HEAD: upstream is the matrix dY.
// Each wave keeps per-column partial sums (dgamma, dbeta [, dw2, column sums of the dX it writes = the bias gradient of the layer
// below]) in registers over its rows; the block's 4 waves are added through LDS in a fixed order and the block writes ONE slot
// of `part` ([n_blocks][4*D + 4], sumk_internal.h: ln_slot_floats); ln_bwd_reduce adds the slots into the gradients in one
// launch (deterministic).  Round 1 wrote one slot per wave and reduced each vector with its own launch over ~1000 slots, and the
// bias gradient took a separate pass over dX: 7 launches x 18 us + 45 us per training step.
// (all of a row's loads -- x, the affine vectors, dY or its K-slice slabs -- are issued before the first use: ln_gather)
template <int NQ, bool HEAD>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ X, const float* __restrict__ stats,
                                                            const float* __restrict__ g, const float* __restrict__ b,
                                                            const float* __restrict__ dY, const float* __restrict__ w2,
                                                            const float* __restrict__ scores,
                                                            const float* __restrict__ dscores, float* __restrict__ dX,
                                                            float* __restrict__ part, int n_rows, int D, Drop drop_in,
                                                            uint32_t site, unsigned short* __restrict__ dX16,     // bf16(dX) too (dX may then be null)
                                                            int n_slab, int64_t slab_stride) {                    // !HEAD: dY = n_slab K-slice slabs added on load (SlabIn)
  const Drop drop = drop_resolve(drop_in);
  const int lane = threadIdx.x & 63;
  const int wave_id = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int n_waves = gridDim.x * 4;
  const int D4 = D >> 2;
  __shared__ float4 red[3][NQ * 64];
  float4 ag[NQ], ab[NQ], aw[NQ], ax[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) ag[q] = ab[q] = aw[q] = ax[q] = make_float4(0.f, 0.f, 0.f, 0.f);
  float ab2 = 0.f;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  const float4* b4 = reinterpret_cast<const float4*>(b);
  const float4* w4 = reinterpret_cast<const float4*>(w2);
  for (int row = wave_id; row < n_rows; row += n_waves) {
    const float mean = stats[2 * row], rstd = stats[2 * row + 1];
    const float4* x4 = reinterpret_cast<const float4*>(X + (int64_t)row * D);
    float du = 0.f;
    if constexpr (HEAD) { float sc = scores[row]; du = dscores[row] * sc * (1.f - sc); ab2 += du; }
    float4 xh[NQ], dxh[NQ], keep[NQ];
    float4 pxv[NQ], pgv[NQ], pwv[NQ], pbv[NQ], pdy[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int c = min(lane + 64 * q, D4 - 1);
      pxv[q] = x4[c]; pgv[q] = g4[c];
      if constexpr (HEAD) { pwv[q] = w4[c]; pbv[q] = b4[c]; }
    }
    if constexpr (!HEAD) {
      const float4* dy4 = reinterpret_cast<const float4*>(dY + (int64_t)row * D);
      const int64_t st4 = slab_stride >> 2;
      row_gather<NQ>(pdy, dy4, lane, D4, n_slab, st4);
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int c = lane + 64 * q;
      xh[q] = dxh[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      keep[q] = make_float4(1.f, 1.f, 1.f, 1.f);
      if (c < D4) {
        float4 v = pxv[q];
        if constexpr (HEAD) {  // ReLU mask on the stored post-ReLU value (vasnet.py:141)
          keep[q].x = v.x > 0.f ? 1.f : 0.f; keep[q].y = v.y > 0.f ? 1.f : 0.f;
          keep[q].z = v.z > 0.f ? 1.f : 0.f; keep[q].w = v.w > 0.f ? 1.f : 0.f;
        }
        if (drop.thr) {
          uint64_t base = (uint64_t)row * D + 4 * c;
          float k0 = dropout_keep(drop.seed, site, base, drop.thr) ? drop.scale : 0.f;
          float k1 = dropout_keep(drop.seed, site, base + 1, drop.thr) ? drop.scale : 0.f;
          float k2 = dropout_keep(drop.seed, site, base + 2, drop.thr) ? drop.scale : 0.f;
          float k3 = dropout_keep(drop.seed, site, base + 3, drop.thr) ? drop.scale : 0.f;
          v.x *= k0; v.y *= k1; v.z *= k2; v.w *= k3;
          keep[q].x *= k0; keep[q].y *= k1; keep[q].z *= k2; keep[q].w *= k3;
        }
        float4 h;
        h.x = (v.x - mean) * rstd; h.y = (v.y - mean) * rstd; h.z = (v.z - mean) * rstd; h.w = (v.w - mean) * rstd;
        xh[q] = h;
        const float4 gg = pgv[q];
        float4 dy;
        if constexpr (HEAD) {
          const float4 ww = pwv[q], bv = pbv[q];
          dy.x = du * ww.x; dy.y = du * ww.y; dy.z = du * ww.z; dy.w = du * ww.w;
          aw[q].x += du * (h.x * gg.x + bv.x); aw[q].y += du * (h.y * gg.y + bv.y);
          aw[q].z += du * (h.z * gg.z + bv.z); aw[q].w += du * (h.w * gg.w + bv.w);
        } else {
          dy = pdy[q];
        }
        ag[q].x += dy.x * h.x; ag[q].y += dy.y * h.y; ag[q].z += dy.z * h.z; ag[q].w += dy.w * h.w;
        ab[q].x += dy.x; ab[q].y += dy.y; ab[q].z += dy.z; ab[q].w += dy.w;
        float4 dh;
        dh.x = dy.x * gg.x; dh.y = dy.y * gg.y; dh.z = dy.z * gg.z; dh.w = dy.w * gg.w;
        dxh[q] = dh;
        s1 += (dh.x + dh.y) + (dh.z + dh.w);
        s2 += (dh.x * h.x + dh.y * h.y) + (dh.z * h.z + dh.w * h.w);
      }
    }
    s1 = wave_sum(s1) / (float)D;
    s2 = wave_sum(s2) / (float)D;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int c = lane + 64 * q;
      if (c < D4) {
        float4 o;
        o.x = rstd * (dxh[q].x - s1 - xh[q].x * s2) * keep[q].x;
        o.y = rstd * (dxh[q].y - s1 - xh[q].y * s2) * keep[q].y;
        o.z = rstd * (dxh[q].z - s1 - xh[q].z * s2) * keep[q].z;
        o.w = rstd * (dxh[q].w - s1 - xh[q].w * s2) * keep[q].w;
        if (dX != nullptr) reinterpret_cast<float4*>(dX + (int64_t)row * D)[c] = o;
        if (dX16 != nullptr) store_bf16x4(dX16 + (int64_t)row * D + 4 * c, o);
        if constexpr (HEAD) { ax[q].x += o.x; ax[q].y += o.y; ax[q].z += o.z; ax[q].w += o.w; }
      }
    }
  }
  // block combine, one vector at a time: waves 1..3 park their registers in LDS, wave 0 adds them in wave order and stores
  const int wv = threadIdx.x >> 6;
  float* slot = part + (int64_t)blockIdx.x * (4 * D + 4);
  auto combine = [&](float4 (&acc)[NQ], float* dst) {
    __syncthreads();
    if (wv > 0) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) red[wv - 1][lane + 64 * q] = acc[q];
    }
    __syncthreads();
    if (wv == 0) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int c = lane + 64 * q;
        if (c < D4) {
          float4 t = acc[q];
#pragma unroll
          for (int w3 = 0; w3 < 3; ++w3) { const float4 o = red[w3][c]; t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w; }
          reinterpret_cast<float4*>(dst)[c] = t;
        }
      }
    }
  };
  combine(ag, slot);
  combine(ab, slot + D);
  if constexpr (HEAD) {
    combine(aw, slot + 2 * D);
    combine(ax, slot + 3 * D);
    __shared__ float red_b2[4];
    if (lane == 0) red_b2[wv] = ab2;
    __syncthreads();
    if (threadIdx.x == 0) slot[4 * D] = (red_b2[0] + red_b2[1]) + (red_b2[2] + red_b2[3]);
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

static int rowwise_small_tile(int M, int N) {
  static const char* env = SUMK_TUNE_ENV("SUMK_ROW_CFG");   // tuning override: 0 = 128x128, 1 = 64x64, 2 = 128x64
  if (env && env[0] >= '0' && env[0] <= '2') return env[0] - '0';
  return gemm_tiles(M, N, 0) >= 512 ? 0 : 1;
}

struct SkPlan { SkTab row[SR_COUNT], seq[TB_COUNT]; SkRowSpec spec[SR_COUNT]; int64_t slab_seq[TB_COUNT]; };
struct Geometry {  // what both forward and backward derive from the batch
  VasnetWs L;
  int R, st_qkv, st_d, tiles_s, tiles_pv, cfg_s, cfg_pv, t_max;
  bool b16;          // the mixed-precision training step on the bf16-source kernels (use_b16)
  bool attn_fused;   // ... with the per-video attention as fused strips (attn_b16.hip): t_max <= 320
  int sk;            // 1: the small-batch path (use_sk) -- every GEMM an in-launch split-K launch of gemm_lean.hip; P = its tables
  SkPlan P;
};
static bool use_b16(const Geometry& G, int D, int precision, int training);

// Small-batch path (SK launches): a function of the batch geometry and the arithmetic only -- forward and backward must agree.
// SUMK_SK=0 keeps the large-batch kernels for every batch (the A/B switch of tests/test_gpu_vasnet.py::test_small_batch_path_...).
static int use_sk(int R, int D, int precision) {
  static const bool on = !(getenv("SUMK_SK") && getenv("SUMK_SK")[0] == '0');
  // (D <= 2048: the row kernels that add K-slice slabs hold a row in registers)
  return on && precision == SUMK_PRECISION_FP32 && sk_rows_ok(R) && D % 4 == 0 && D <= 2048 ? 1 : 0;
}
// Every table of the step: requested slices, entries, (tile, slice) blocks and tickets -- the host mirror of vasnet_sk_setup_kernel.
static void sk_plan(int R, int D, int n_seq, const int32_t* off, const VasnetWs& L, SkPlan* P) {
  const int64_t e_elems = L.e_elems;
  const size_t ws_off_dz = L.dz, ws_off_dy0 = L.dy0, ws_off_y1 = L.y1, ws_off_ctx = L.ctx;
  constexpr int te = 64;
  // slab: the output's consumer is a row kernel that adds K-slice slabs on load (SlabIn) -- the slices then need no in-launch meeting
  // Slice counts are a function of the contraction length alone (smax > 0: min(smax, K / 128) slices of at least four k-tiles) wherever K
  // is a property of the model or of one video -- a video's scores and the gradients it contributes then do not depend on what else
  // is in the batch, exactly as on the large-batch path.  Only the weight gradients, which contract over ALL rows of the batch, are
  // sliced by launch size (smax = 0).
  auto row = [&](int r, int layout, int M, int N, int K, int lda, int ldb, int ldc, int ldr, int groups, int a_goff, int c_goff, int bsel, int csel, int smax, bool slab = false) {
    const int tiles = ((M + te - 1) / te) * ((N + te - 1) / te);
    int kc, S;
    int S_req = smax > 0 ? std::max(1, std::min(smax, K / 128)) : sk_slices(tiles * groups, K, &kc);
    // (R, D) projections consumed by the LayerNorm kernels: 4 slabs measured 2.4 us per video better than 8 -- the row kernel's extra loads cost
    // what the shorter GEMM chains gain -- and equal to 2; Q.K^T / dAlpha stay at 8 (scripts/probes/sk_smax_sweep.py, third / fourth digit)
    int slab_cap = 4;
    if (const char* e = SUMK_TUNE_ENV("SUMK_SK_SMAX")) if (e[0] && e[1] && (e[2] == '1' || e[2] == '2' || e[2] == '4' || e[2] == '8')) slab_cap = e[2] - '0';
    if (slab) for (S_req = slab_cap; S_req > 1; S_req >>= 1) {        // slab sets: exactly 1, 2, 4 or 8 slices (slab_sum), each at least four k-tiles
      sk_slice(K, S_req, &kc, &S);
      if (S == S_req && kc >= 128) break;
    }
    sk_slice(K, S_req, &kc, &S);
    P->spec[r] = SkRowSpec{M, N, K, lda, ldb, ldc, ldr, layout, groups, S_req, a_goff, c_goff, bsel, csel, slab ? (int64_t)M * ldc : 0, 0, 0};
    P->row[r] = SkTab{groups * S, groups * S * tiles, groups * tiles, S_req, S};
  };
  // (diagnostic build: SUMK_SK_SMAX="<qkv><dctx>" overrides the two slice caps of the launches whose slices meet in the launch)
  // QKV: 240 tiles at T = 300 already give every CU a block; 4 slices + the in-launch meeting measured the same 26 us as one chain per
  // tile (scripts/probes/sk_smax_sweep.py: 112.9 vs 112.7 us per video), so it stays unsliced -- the same k order as the large-batch kernel
  int smax_qkv = 1, smax_dctx = 8;
  if (const char* e = SUMK_TUNE_ENV("SUMK_SK_SMAX")) if (e[0] >= '1' && e[0] <= '8' && e[1] >= '1' && e[1] <= '8') { smax_qkv = e[0] - '0'; smax_dctx = e[1] - '0'; }
  row(SR_QKV, GEMM_NT, R, D, D, D, D, 3 * D, 0, 3, 0, D, 1, 0, smax_qkv);   // [Q|K|V] = X [Wq;Wk;Wv]^T: group g = B pointer g, columns g D..
  row(SR_OPROJ, GEMM_NT, R, D, D, D, D, D, D, 1, 0, 0, 0, 0, 8, true);      // Y0 = CTX Wo^T + X        (slabs -> LayerNorm kernel, which adds X)
  row(SR_K1, GEMM_NT, R, D, D, D, D, D, 0, 1, 0, 0, 0, 0, 8, true);         // Z = relu(Y1 W1^T + b1)   (slabs -> LayerNorm + head kernel: + b1, ReLU)
  row(SR_DY1, GEMM_NN, R, D, D, D, D, D, 0, 1, 0, 0, 0, 0, 8, true);        // dY1 = dZ W1              (slabs -> LayerNorm backward kernel)
  row(SR_DCTX, GEMM_NN, R, D, D, D, D, D, 0, 1, 0, 0, 0, 0, smax_dctx);     // dCTX = dY0 Wo
  // dWo += dY0^T CTX and dW1 += dZ^T Y1 in ONE launch: group 1's operands are addressed relative to group 0's (dZ - dY0, Y1 - CTX: all four
  // are regions of the workspace), its output is C pointer 1
  row(SR_DWO1, GEMM_TN, D, D, R, D, D, D, 0, 2, 0, 0, 0, 1, 0);
  P->spec[SR_DWO1].a_goff64 = ((int64_t)ws_off_dz - (int64_t)ws_off_dy0) / 4;
  P->spec[SR_DWO1].b_goff64 = ((int64_t)ws_off_y1 - (int64_t)ws_off_ctx) / 4;
  row(SR_DWQKV, GEMM_TN, D, D, R, 3 * D, D, D, 0, 3, D, 0, 0, 1, 0);        // d[Wq;Wk;Wv] += dQKV^T X: group g = columns g D.. of dQKV, output g
  for (int t = 0; t < TB_COUNT; ++t) {
    int kc;
    // (per-video products: the request is the cap; vasnet_sk_setup_kernel / sk_slice_seq give a video with K = T_s frames min(4, T_s / 128)
    //  slices -- a function of that video alone)
    int S_req = 4;
    int slab_cap = 8;
    if (const char* e = SUMK_TUNE_ENV("SUMK_SK_SMAX")) if (e[0] && e[1] && e[2] && (e[3] == '1' || e[3] == '2' || e[3] == '4' || e[3] == '8')) slab_cap = e[3] - '0';   // (fourth digit: Q.K^T / dAlpha)
    if (t == TB_S || t == TB_DP) for (S_req = slab_cap; S_req > 1; S_req >>= 1) {     // slab sets (K = D for every video): 1, 2, 4 or 8 slices
      int S; sk_slice_seq(D, S_req, &kc, &S);          // (the rule vasnet_sk_setup_kernel applies to each video)
      if (S == S_req) break;
    }
    SkTab tb{0, 0, 0, S_req, 1};
    // Q.K^T and dAlpha = dC V^T (K = D for every video: the same slice count) are consumed by the softmax kernels: slabs
    P->slab_seq[t] = (t == TB_S || t == TB_DP) ? e_elems : 0;
    for (int q = 0; q < n_seq; ++q) {
      int M, N, K, S;
      sk_seq_dims(t, off[q + 1] - off[q], D, &M, &N, &K);
      sk_slice_seq(K, S_req, &kc, &S);
      const int tl = ((M + te - 1) / te) * ((N + te - 1) / te);
      tb.entries += S; tb.blocks += S * tl; tb.tiles += tl; tb.S = std::max(tb.S, S);
    }
    P->seq[t] = tb;
  }
}

static int geometry(int D, int n_seq, const int32_t* off, int training, int precision, size_t workspace_bytes, Geometry* G) {
  SUMK_TRY(carve(D, n_seq, off, training, &G->L));
  G->R = G->L.n_rows;
  G->st_qkv = rowwise_small_tile(G->R, 3 * D);
  G->st_d = rowwise_small_tile(G->R, D);
  // Tile shape of the ragged per-video products.  64x64 is the measured best on S-TVSum (T~235: 8.66 M frames/s vs 8.57
  // with 128x128 for the (T x D) products and 8.28 with 128x128 for both -- bigger tiles waste more on ragged T and
  // under-fill the resident slots); long videos (mean T >= 1024, e.g. BASELINE config 5) take the 128x128 tile, whose
  // 2x2 register blocking halves the LDS traffic per MFMA.
  static const char* env = SUMK_TUNE_ENV("SUMK_ATTN_CFG");   // tuning override "<cfg_s><cfg_pv>", e.g. "20"
  const int cfg_auto = (G->R / n_seq >= 1024) ? 0 : 1;
  G->cfg_s = cfg_auto; G->cfg_pv = cfg_auto;
  if (env && env[0] >= '0' && env[0] <= '2' && env[1] >= '0' && env[1] <= '2') { G->cfg_s = env[0] - '0'; G->cfg_pv = env[1] - '0'; }
  G->t_max = 0;
  for (int s = 0; s < n_seq; ++s) G->t_max = std::max(G->t_max, off[s + 1] - off[s]);
  // A function of the batch and the options ONLY (never of workspace_bytes: forward and backward must take the same path -- the
  // forward leaves the bf16 shadows and table layout the backward reads); a workspace without the shadows is refused below.
  (void)workspace_bytes;
  G->b16 = use_b16(*G, D, precision, training);
  if (G->b16) G->cfg_s = G->cfg_pv = 0;       // the bf16-source kernel has 128x128 tiles
  {  // SUMK_ATTN_FUSED=0 keeps the separate launches (GEMM -> softmax kernel -> GEMM): the A/B switch of tests/test_gpu_train_full.py
    static const bool fused_on = !(getenv("SUMK_ATTN_FUSED") && getenv("SUMK_ATTN_FUSED")[0] == '0');
    G->attn_fused = G->b16 && fused_on && attn_strip_ok(G->t_max, D, G->R, 3 * D);
  }
  G->tiles_s = G->tiles_pv = 0;
  for (int s = 0; s < n_seq; ++s) {
    int T = off[s + 1] - off[s];
    G->tiles_s += gemm_tiles(T, T, G->cfg_s); G->tiles_pv += gemm_tiles(T, D, G->cfg_pv);
  }
  G->sk = G->b16 ? 0 : use_sk(G->R, D, precision);
  if (G->sk) sk_plan(G->R, D, n_seq, off, G->L, &G->P);
  return SUMK_OK;
}

// Does the training step run its row-wise GEMMs on the bf16-source kernels (gemm_b16.hip)?  A function of the batch geometry and the
// options only: forward and backward must agree (the forward leaves the bf16 shadows the backward reads).  SUMK_BF16_SRC=0 keeps
// the plane kernels (fp32 operands converted per k-tile) -- the A/B switch.
static bool use_b16(const Geometry& G, int D, int precision, int training) {
  static const bool on = !(getenv("SUMK_BF16_SRC") && getenv("SUMK_BF16_SRC")[0] == '0');
  const int R = G.R;
  // (per-video attention blocks: T x ld16 bf16 with ld16 = T rounded up to 64 -- their byte offsets must fit 31 bits too)
  const int64_t ld16 = ((int64_t)G.t_max + 63) & ~(int64_t)63;
  if (((int64_t)G.t_max + 128) * ld16 * 2 >= ((int64_t)1 << 31)) return false;
  return on && training && precision == SUMK_PRECISION_BF16 && G.st_qkv == 0 && G.st_d == 0 &&
         gemm_b16_ok(R, 3 * D, D, D, D, true, true) && gemm_b16_ok(R, D, D, D, D, true, false) &&
         gemm_b16_ok(R, D, D, 3 * D, D, true, false) && gemm_b16_ok(3 * D, D, R, 3 * D, D, false, false);
}

// put a row-wise launch on the bf16-source kernels: bf16 operands and, where it fills the chip, the wide tile with its problem entry
static void to_b16(GemmLaunch& g, const void* A16, const void* B16, int M, int N, GemmProb* prow, int slot_wide) {
  g.A = (const float*)A16; g.B[0] = (const float*)B16; g.B[1] = g.B[2] = g.B[3] = nullptr; g.n_group = 0; g.src16 = 1;
  static const int force = getenv("SUMK_B16_WIDE") ? atoi(getenv("SUMK_B16_WIDE")) : -1;     // A/B switch: 0 = 128x128 tiles everywhere
  const int wide = force >= 0 ? force : gemm_b16_wide_bm(M, N);
  if (wide == 192 || wide == 256) { g.wide16 = wide; g.probs = prow + slot_wide; g.total_tiles = gemm_tiles_wide(M, N, wide); }
}

// row-wise problem slots (index into prob_row)
enum { RP_QKV = 0, RP_DD = 1, RP_DX = 2, RP_QKV_W = 3, RP_DD_W = 4, RP_DX_W = 5 };   // _W: the same problems on 256-column tiles (gemm_b16.hip)

static void launch_setup(const Geometry& G, int D, int n_seq, const int32_t* off_dev, char* ws, hipStream_t stream) {     // ws: the TABLE base (workspace front, or the caller's table buffer)
  SetupArgs a;
  a.fake_seq0 = 0;
#ifdef SUMK_DIAG
  if (getenv("SUMK_FAKE_SEQ0")) a.fake_seq0 = 1;    // wrong results by design: timing experiment (are the per-video GEMMs bound by where their operands come from?)
#endif
  a.p16 = G.b16 ? 1 : 0;
  a.off = off_dev; a.n_seq = n_seq; a.D = D;
  a.row_seq = (int32_t*)(ws + G.L.row_seq);
  a.seq = (SeqInfo*)(ws + G.L.seq); a.tabs = (GemmProb*)(ws + G.L.prob_seq); a.prow = (GemmProb*)(ws + G.L.prob_row);
  const int R = G.R;
  a.rows[RP_QKV] = RowProbSpec{R, 3 * D, D, D, D, 3 * D, 0, G.st_qkv};   // [Q|K|V] = X W^T
  a.rows[RP_DD] = RowProbSpec{R, D, D, D, D, D, D, G.st_d};              // (R,D) = (R,D) x (D,D), any layout
  a.rows[RP_DX] = RowProbSpec{R, D, D, 3 * D, D, D, D, G.st_d};          // dX += dQKV[:, part] W  (A has lda 3D)
  a.rows[RP_QKV_W] = a.rows[RP_QKV]; a.rows[RP_QKV_W].small = 3;
  a.rows[RP_DD_W] = a.rows[RP_DD]; a.rows[RP_DD_W].small = 3;
  a.rows[RP_DX_W] = a.rows[RP_DX]; a.rows[RP_DX_W].small = 3;
  a.n_rowprobs = 6;
  a.s_tm = gemm_tile_m(G.cfg_s); a.s_tn = gemm_tile_n(G.cfg_s); a.pv_tm = gemm_tile_m(G.cfg_pv); a.pv_tn = gemm_tile_n(G.cfg_pv);
  a.dvk = nullptr; a.dk_a_delta = a.dk_b_delta = 0;
  if (G.attn_fused) {   // (offsets between buffers of ONE workspace carve: the same for every workspace of this geometry)
    a.dvk = (GemmProb*)(ws + G.L.prob_dvk);
    a.dk_a_delta = ((int64_t)G.L.s16 - (int64_t)G.L.p16) / 2;
    a.dk_b_delta = ((int64_t)G.L.qkv16 - (int64_t)G.L.dctx16) / 2;
  }
  hipLaunchKernelGGL(vasnet_setup_kernel, dim3((n_seq + 63) / 64, 2 + 32), dim3(64), 0, stream, a);   // y: 0 per-video tables, 1 row problems, 2.. row -> video table
}

static Drop make_drop(const sumk_vasnet_opts* o) { Drop d = make_drop(o->dropout_p, o->seed); d.seed_dev = o->seed_dev; return d; }

static int launch_sk_setup(const Geometry& G, int D, int n_seq, const int32_t* off_dev, char* ws, hipStream_t stream) {
  SkSetupArgs a;
  a.off = off_dev; a.n_seq = n_seq; a.D = D;
  a.seq = (SeqInfo*)(ws + G.L.seq); a.row_seq = (int32_t*)(ws + G.L.row_seq); a.cnt = (unsigned*)(ws + G.L.sk_cnt);
  a.tabs = (GemmProb*)(ws + G.L.sk_tabs);
  a.tile = 64;
  const bool tickets = true;
  for (int r = 0; r < SR_COUNT; ++r) a.rows[r] = G.P.spec[r];
  for (int t = 0; t < TB_COUNT; ++t) {
    a.S_seq[t] = G.P.seq[t].S_req; a.slab_seq[t] = G.P.slab_seq[t];
    SUMK_ARG((!tickets || G.P.seq[t].tiles <= SK_TICKETS) && G.P.seq[t].entries <= n_seq * SK_MAX_SLICES, "vasnet: small-batch table %d out of range", t);
  }
  for (int r = 0; r < SR_COUNT; ++r)
    SUMK_ARG((!tickets || G.P.row[r].tiles <= SK_TICKETS) && G.P.row[r].entries <= SK_ROW_ENTRIES, "vasnet: small-batch row table %d out of range", r);
  hipLaunchKernelGGL(vasnet_sk_setup_kernel, dim3((n_seq + 63) / 64, 2 + 8), dim3(64), 0, stream, a);   // y: 0 per-video tables, 1 row tables, 2.. row -> video table + tickets
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}

static size_t sk_part_bytes(const Geometry& G, int n_seq) { return G.L.sk_part_bytes; }
// one SK launch over table `tb` (row-wise table r >= 0, or per-video table t)
struct SkCall { const float* A; const float* B[4]; float* C[4]; const float* R; const float* bias; int prof_tag; };
static int launch_sk(const Geometry& G, int n_seq, char* ws, char* tbase, GemmLayout layout, GemmEpi epi, int row_tab, int seq_tab, const SkCall& c, hipStream_t stream) {
  const SkTab& tb = row_tab >= 0 ? G.P.row[row_tab] : G.P.seq[seq_tab];
  GemmProb* tabs = (GemmProb*)(tbase + G.L.sk_tabs);
  GemmLaunch g;
  g.A = c.A;
  for (int i = 0; i < 4; ++i) { g.B[i] = c.B[i]; g.Csel[i] = c.C[i]; }
  g.C = c.C[0]; g.R = c.R; g.bias0[0] = c.bias;
  g.probs = row_tab >= 0 ? sk_row_tab(tabs, row_tab) : sk_seq_tab(tabs, seq_tab, n_seq);
  g.nprob = tb.entries; g.total_tiles = tb.blocks; g.small_tile = 1; g.prof_tag = c.prof_tag;
  g.sk = 1; g.sk_part = (float*)(ws + G.L.sk_part); g.sk_cnt = (unsigned*)(tbase + G.L.sk_cnt);
  SUMK_ARG(tb.blocks == tb.tiles || (size_t)tb.blocks * 64 * 64 * 4 <= sk_part_bytes(G, n_seq), "vasnet: small-batch launch needs %d partial tiles", tb.blocks);   // (blocks == tiles: nothing is sliced)
  return launch_gemm(layout, epi, g, stream);
}

int launch_layernorm(const float* X, float* Y, const float* g, const float* b, int n_rows, int D, float eps, float* stats,
                     hipStream_t stream) {
  const Drop none = make_drop(0.f, 0);
  launch_ln_rows<false>(X, Y, g, b, nullptr, nullptr, nullptr, n_rows, D, eps, stats, none, 0u, stream);
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}
int launch_ln_head(const float* Z, const float* g, const float* b, const float* w2, const float* b2, float* scores,
                   int n_rows, int D, float eps, hipStream_t stream) {
  const Drop none = make_drop(0.f, 0);
  launch_ln_rows<true>(Z, nullptr, g, b, w2, b2, scores, n_rows, D, eps, nullptr, none, 0u, stream);
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}
int launch_layernorm_drop(const float* X, float* Y, const float* g, const float* b, int n_rows, int D, float eps, float* stats,
                          Drop drop, uint32_t site, hipStream_t stream) {
  launch_ln_rows<false>(X, Y, g, b, nullptr, nullptr, nullptr, n_rows, D, eps, stats, drop, site, stream);
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}
int launch_ln_head_drop(const float* Z, const float* g, const float* b, const float* w2, const float* b2, float* scores,
                        int n_rows, int D, float eps, float* stats, Drop drop, uint32_t site, hipStream_t stream) {
  launch_ln_rows<true>(Z, nullptr, g, b, w2, b2, scores, n_rows, D, eps, stats, drop, site, stream);
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}
int launch_add_pos(float* x, const float* table, const int32_t* pos_rows, int n_rows, int D, hipStream_t stream) {
  int64_t n4 = (int64_t)n_rows * (D >> 2);
  hipLaunchKernelGGL(add_pos_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, x, table, pos_rows, n_rows, D);
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}

}  // namespace sumk

using namespace sumk;

// ---- the SK launch by itself (tests / probes): ONE problem cut into S K slices that meet inside the launch
namespace sumk {
struct SkOneArgs { GemmProb* tab; unsigned* cnt; int32_t tiles, layout, M, N, K, lda, ldb, ldc, ldr, S; };
__global__ void sk_one_setup_kernel(SkOneArgs a) {
  for (int i = threadIdx.x; i < a.tiles; i += blockDim.x) a.cnt[i] = 0u;
  if (threadIdx.x == 0) put_sk(a.tab, 0, 0, 0, a.layout, 0, 0, 0, a.M, a.N, a.K, a.lda, a.ldb, a.ldc, a.ldr, a.S, 0, 0, 64, 0, false);
}
}  // namespace sumk
extern "C" size_t sumk_gemm_splitk_workspace_bytes(int32_t M, int32_t N, int32_t slices) {
  if (M <= 0 || N <= 0 || slices < 1 || slices > SK_MAX_SLICES) return 0;
  const size_t tiles = (size_t)((M + 63) / 64) * ((N + 63) / 64);
  return align_up(SK_MAX_SLICES * sizeof(GemmProb), 256) + align_up(tiles * 4, 256) + tiles * slices * 64 * 64 * 4;
}
extern "C" int sumk_gemm_splitk(int32_t layout, const float* A, const float* B, float* C, int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldb,
                                int32_t ldc, int32_t slices, int32_t epilogue, const float* R, int32_t ldr, const float* bias, float alpha,
                                void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SUMK_ARG(layout >= 0 && layout <= 2 && A && B && C && workspace, "gemm_splitk: bad arguments");
  SUMK_ARG(M > 0 && N > 0 && K > 0 && lda % 4 == 0 && ldb % 4 == 0, "gemm_splitk: M=%d N=%d K=%d lda=%d ldb=%d (leading dimensions in multiples of 4)", M, N, K, lda, ldb);
  SUMK_ARG(epilogue == EPI_NONE || (epilogue == EPI_RESIDUAL && R) || (epilogue == EPI_BIAS_RELU && bias) || epilogue == EPI_ACCUM, "gemm_splitk: epilogue %d", epilogue);
  const size_t need = sumk_gemm_splitk_workspace_bytes(M, N, slices);
  SUMK_ARG(need != 0 && ((uintptr_t)workspace & 255) == 0, "gemm_splitk: 1 .. %d slices, 256-byte aligned workspace", SK_MAX_SLICES);
  if (workspace_bytes < need) { set_error("gemm_splitk: workspace %zu < required %zu", workspace_bytes, need); return SUMK_ERR_WORKSPACE; }
  const int tiles = ((M + 63) / 64) * ((N + 63) / 64);
  char* ws = (char*)workspace;
  int kc, S;
  sk_slice(K, slices, &kc, &S);
  SkOneArgs a{(GemmProb*)ws, (unsigned*)(ws + align_up(SK_MAX_SLICES * sizeof(GemmProb), 256)), tiles, layout, M, N, K, lda, ldb, ldc, ldr, slices};
  hipLaunchKernelGGL(sk_one_setup_kernel, dim3(1), dim3(256), 0, stream, a);
  GemmLaunch g;
  g.A = A; g.B[0] = B; g.C = C; g.R = R; g.bias0[0] = bias; g.alpha = alpha;
  g.probs = a.tab; g.nprob = S; g.total_tiles = S * tiles; g.small_tile = 1;
  g.sk = 1; g.sk_cnt = a.cnt; g.sk_part = (float*)(ws + align_up(SK_MAX_SLICES * sizeof(GemmProb), 256) + align_up((size_t)tiles * 4, 256));
  return launch_gemm((GemmLayout)layout, (GemmEpi)epilogue, g, stream);
}

// ---- the table block by itself: built once per batch geometry, handed to every call through sumk_vasnet_opts::tables
extern "C" size_t sumk_vasnet_tables_bytes(int32_t D, int32_t n_seq, const int32_t* seq_off_host) {
  VasnetWs w;
  if (carve(D, n_seq, seq_off_host, 0, &w) != SUMK_OK) return 0;
  return w.tab_bytes;
}
// The K-slice tables of the backward pass's two weight-gradient launches (d[Wo, W1] in one launch, d[Wq; Wk; Wv] in another), as
// sumk_vasnet_backward would build them per call: they depend on the shapes, the slab / table capacities and the DISTANCES between the
// operands inside the workspace -- all functions of the geometry -- so they are built once with the other tables.
static int build_skw_tables(const Geometry& G, int D, GemmProb* pskw, int precision, hipStream_t stream) {
  const VasnetWs& L = G.L;
  const int R = G.R;
  const bool b16 = G.b16;
  const char* const base = reinterpret_cast<const char*>((uintptr_t)1 << 40);       // never dereferenced: only differences are taken
  const float* const As[2] = {(const float*)(base + (b16 ? L.dy016 : L.dy0)), (const float*)(base + (b16 ? L.dz16 : L.dz))};
  const float* const Bs[2] = {(const float*)(base + (b16 ? L.ctx16 : L.ctx)), (const float*)(base + (b16 ? L.y116 : L.y1))};
  float* none[4] = {nullptr, nullptr, nullptr, nullptr};
  SUMK_TRY(gemm_tn_splitk_accum_multi(2, As, Bs, D, D, D, D, R, nullptr, L.slab_elems, pskw, SPLITK_PROBS, none, D, D, 1.f, stream, precision, b16, SPLITK_TABLE_ONLY));
  SUMK_TRY(gemm_tn_splitk_accum((const float*)base, 3 * D, (const float*)base, D, 3 * D, D, R, nullptr, L.slab_elems, pskw + SPLITK_PROBS, SPLITK_PROBS, none, D, D, 1.f,
                                stream, precision, b16, SPLITK_TABLE_ONLY));
  return SUMK_OK;
}

extern "C" int sumk_vasnet_build_tables(int32_t D, int32_t n_seq, const int32_t* seq_off_host, const int32_t* seq_off_dev, int32_t training,
                                        int32_t precision, void* tables, size_t tables_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SUMK_ARG(seq_off_dev && tables && ((uintptr_t)tables & 255) == 0, "vasnet_build_tables: null or misaligned (256 B) pointer");
  SUMK_ARG(precision >= SUMK_PRECISION_FP32 && precision <= SUMK_PRECISION_MAX, "vasnet_build_tables: unknown precision %d", precision);
  Geometry G;
  SUMK_TRY(geometry(D, n_seq, seq_off_host, training, precision, 0, &G));
  if (tables_bytes < G.L.tab_bytes) { set_error("vasnet_build_tables: %zu bytes < required %zu", tables_bytes, G.L.tab_bytes); return SUMK_ERR_WORKSPACE; }
  launch_setup(G, D, n_seq, seq_off_dev, (char*)tables, stream);                 // the large-batch tables (also what a dX-producing backward uses)
  if (G.sk) SUMK_TRY(launch_sk_setup(G, D, n_seq, seq_off_dev, (char*)tables, stream));   // the small-batch tables + zeroed tickets
  if (training) SUMK_TRY(build_skw_tables(G, D, (GemmProb*)((char*)tables + G.L.prob_skw), precision, stream));
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}

extern "C" size_t sumk_vasnet_wplanes_bytes(int32_t D, int32_t n_planes) {
  return wplanes_ok(D, n_planes) ? wplanes_layout(D, n_planes).total : 0;
}

extern "C" int sumk_vasnet_wplanes_build(int32_t D, const sumk_vasnet_weights* w, const float* Wvo, int32_t n_planes, void* out, size_t out_bytes,
                                         void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SUMK_ARG(wplanes_ok(D, n_planes), "vasnet_wplanes_build: D=%d planes=%d is not eligible (D %% 256, 2 or 3 planes)", D, n_planes);
  SUMK_ARG(w && w->Wk && w->Wq && w->Wv && w->Wo && w->W1 && w->b1 && w->w2 && w->ln_w && w->ln_b && out, "vasnet_wplanes_build: null pointer");
  const WPlanes l = wplanes_layout(D, n_planes);
  SUMK_ARG(out_bytes >= l.total && ((uintptr_t)out & 255) == 0, "vasnet_wplanes_build: buffer of %zu bytes (256-byte aligned) needed, got %zu", l.total, out_bytes);
  char* o = (char*)out;
  SUMK_HIP(hipMemsetAsync(o, 0, l.c1, stream));           // (the slack behind each plane array is read by row tiles: keep it finite)
  const float* parts[3] = {w->Wq, w->Wk, Wvo ? Wvo : w->Wv};
  for (int i = 0; i < 3; ++i) SUMK_TRY(split_planes_at(parts[i], D, D, D, n_planes, o + l.wqkv, (int64_t)i * D, 3 * (int64_t)D, stream));
  SUMK_TRY(split_planes(w->Wo, D, D, D, n_planes, o + l.wo, stream));
  float* W1g = (float*)(o + l.tmp); float* c2 = W1g + (size_t)D * D;
  hipLaunchKernelGGL(ln_fold_stats_kernel, dim3(D), dim3(256), 0, stream, w->W1, w->ln_w, w->ln_b, D, W1g, (float*)(o + l.c1), c2,
                     (const float2*)nullptr, 0, 0, 0.f, (float2*)nullptr);
  SUMK_HIP(hipGetLastError());
  SUMK_TRY(split_planes(W1g, D, D, D, n_planes, o + l.w1g, stream));
  hipLaunchKernelGGL(head_vectors_kernel, dim3((D + 255) / 256), dim3(256), 0, stream, w->b1, c2, w->ln_w, w->w2, D, (float*)(o + l.biasc), (float*)(o + l.gw));
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}

extern "C" size_t sumk_vasnet_workspace_bytes(int32_t D, int32_t n_seq, const int32_t* seq_off_host, int32_t training) {
  VasnetWs w;
  if (carve(D, n_seq, seq_off_host, training, &w) != SUMK_OK) return 0;
  return w.total;
}
extern "C" size_t sumk_vasnet_workspace_bytes_for(int32_t D, int32_t n_seq, const int32_t* seq_off_host, int32_t training, int32_t precision) {
  VasnetWs w;
  if (carve(D, n_seq, seq_off_host, training, &w) != SUMK_OK) return 0;
  const int np = precision == SUMK_PRECISION_BF16X6 ? 3 : precision == SUMK_PRECISION_BF16X3 ? 2 : 0;
  if (!training && np && wplanes_ok(D, np)) {
    int t_max = 0;
    for (int s = 0; s < n_seq; ++s) t_max = std::max(t_max, seq_off_host[s + 1] - seq_off_host[s]);
    if (w.n_rows >= 256 && attn_pw_ok(t_max, D, w.n_rows, np)) return pw_extra(D, w.n_rows, t_max, np, w.total_core).total;
    int t_min = t_max;
    for (int s = 0; s < n_seq; ++s) t_min = std::min(t_min, seq_off_host[s + 1] - seq_off_host[s]);
    if (w.n_rows >= 256 && pw_long_ok(D, t_min, t_max, np)) return pw_long_layout(D, w.n_rows, t_max, np, w.total_core).total;
  }
  return (training && precision == SUMK_PRECISION_BF16) ? w.total : w.total_core;
}

// The small-batch forward: the same nine stages as vasnet_forward_impl, every GEMM an SK launch (gemm_lean.hip) over the sliced tables
// of vasnet_sk_setup_kernel, the row kernels unchanged.  Inference and training (the intermediates the backward needs are the same).
static int vasnet_forward_sk(const Geometry& G, float* x, int32_t D, int32_t n_seq, const int32_t* seq_off_host, const int32_t* seq_off_dev,
                             const sumk_vasnet_weights* w, const sumk_vasnet_opts* opts, const float* pos_table, const int32_t* pos_rows,
                             float* scores, char* ws, char* tb, int32_t training, hipStream_t stream) {     // tb: table base (ws, or opts->tables: then already built)
  const VasnetWs& L = G.L;
  const int R = G.R;
  float* QKV = (float*)(ws + L.qkv);
  float* E = (float*)(ws + L.e);
  float* E2 = training ? (float*)(ws + L.e2) : nullptr;
  float* CTX = (float*)(ws + L.ctx);
  float* Y0 = (float*)(ws + L.y0);
  float* Y1 = (float*)(ws + L.y1);
  float* Z = (float*)(ws + L.z);
  SeqInfo* seq = (SeqInfo*)(tb + L.seq);
  float* stats = training ? (float*)(ws + L.stats) : nullptr;
  const Drop drop = make_drop(opts);
  const bool use_e2 = training && drop.thr != 0;
  if (pos_table) {
    int64_t n4 = (int64_t)R * (D >> 2);
    hipLaunchKernelGGL(add_pos_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, x, pos_table, pos_rows, R, D);
  }
  for (int S : {G.P.seq[TB_S].S, G.P.seq[TB_DP].S, G.P.row[SR_OPROJ].S, G.P.row[SR_K1].S, G.P.row[SR_DY1].S})     // slab sets the row kernels add: slab_sum's cases
    SUMK_ARG(S == 1 || S == 2 || S == 4 || S == 8, "vasnet: slab set of %d slices (internal: sk_plan builds 1, 2, 4 or 8)", S);
  if (tb == ws) SUMK_TRY(launch_sk_setup(G, D, n_seq, seq_off_dev, tb, stream));
  {  // 1: QKV projection (three B pointers, one launch)
    const SkCall c{x, {w->Wq, w->Wk, w->Wv, nullptr}, {QKV, nullptr, nullptr, nullptr}, nullptr, nullptr, SUMK_PROF_GEMM_QKV};
    SUMK_TRY(launch_sk(G, n_seq, ws, tb, GEMM_NT, EPI_NONE, SR_QKV, -1, c, stream));
  }
  // K-slice slabs (SlabIn): where a row kernel consumes a GEMM's output, the slices store their own slab in the scratch and the row
  // kernel adds them -- S slabs of the E layout / of (R, D)
  float* scratch = (float*)(ws + L.sk_part);
  {  // 2: logits per video
    const SkCall c{QKV, {QKV, nullptr, nullptr, nullptr}, {scratch, nullptr, nullptr, nullptr}, nullptr, nullptr, SUMK_PROF_GEMM_QKT};
    SUMK_TRY(launch_sk(G, n_seq, ws, tb, GEMM_NT, EPI_NONE, -1, TB_S, c, stream));
  }
  {  // 3: softmax (+ dropout of alpha into E2 when training with p > 0)
    const dim3 sg((R + 3) / 4), sb(256);
    float* e2p = use_e2 ? E2 : nullptr;
    const float* eraw = scratch;
    const int n_slab = G.P.seq[TB_S].S;
#define SUMK_SOFTMAX(NR) hipLaunchKernelGGL((vasnet_softmax_kernel<NR, true>), sg, sb, 0, stream, E, e2p, seq, (const int32_t*)(tb + L.row_seq), n_seq, R, opts->scale, opts->ignore_self, opts->aperture, drop, (unsigned short*)nullptr, eraw, n_slab, (int64_t)L.e_elems)
    if (G.t_max <= 256) SUMK_SOFTMAX(4); else if (G.t_max <= 512) SUMK_SOFTMAX(8); else if (G.t_max <= 1024) SUMK_SOFTMAX(16); else SUMK_SOFTMAX(0);
#undef SUMK_SOFTMAX
  }
  {  // 4: context
    const SkCall c{use_e2 ? E2 : E, {QKV, nullptr, nullptr, nullptr}, {CTX, nullptr, nullptr, nullptr}, nullptr, nullptr, SUMK_PROF_GEMM_PV};
    SUMK_TRY(launch_sk(G, n_seq, ws, tb, GEMM_NN, EPI_NONE, -1, TB_PV, c, stream));
  }
  {
    // 5 + 6: output projection as K-slice slabs; the LayerNorm kernel adds them and the residual (Y0 itself is kept for the backward pass)
    const SkCall c{CTX, {w->Wo, nullptr, nullptr, nullptr}, {scratch, nullptr, nullptr, nullptr}, nullptr, nullptr, SUMK_PROF_GEMM_OPROJ};
    SUMK_TRY(launch_sk(G, n_seq, ws, tb, GEMM_NT, EPI_NONE, SR_OPROJ, -1, c, stream));
    SlabIn sl; sl.n = G.P.row[SR_OPROJ].S; sl.stride = (int64_t)R * D; sl.add = x; sl.store = training ? Y0 : nullptr;
    launch_ln_rows<false>(scratch, Y1, w->ln_w, w->ln_b, nullptr, nullptr, nullptr, R, D, opts->eps, stats, drop, 1u, stream, nullptr, sl);
    // 7 + 8: k1 as slabs; the LayerNorm + head kernel adds them, the bias and the ReLU (Z is kept for the backward pass)
    const SkCall c2{Y1, {w->W1, nullptr, nullptr, nullptr}, {scratch, nullptr, nullptr, nullptr}, nullptr, nullptr, SUMK_PROF_GEMM_K1};
    SUMK_TRY(launch_sk(G, n_seq, ws, tb, GEMM_NT, EPI_NONE, SR_K1, -1, c2, stream));
    SlabIn s2; s2.n = G.P.row[SR_K1].S; s2.stride = (int64_t)R * D; s2.bias = w->b1; s2.relu = 1; s2.store = training ? Z : nullptr;
    launch_ln_rows<true>(scratch, training ? (float*)(ws + L.scores) : nullptr, w->ln_w, w->ln_b, w->w2, w->b2, scores, R, D, opts->eps, stats ? stats + 2 * (size_t)R : nullptr, drop, 2u, stream, nullptr, s2);      // (the backward reads the scores from the workspace: written here, not copied)
  }
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}

// Wvo != nullptr: inference with the value and output projections FOLDED (Wvo = Wo . Wv, computed once per weight change by the
// caller): (alpha V) Wo^T = alpha (X Wv^T Wo^T) = alpha (X Wvo^T), so the third slice of the packed projection is U = X Wvo^T,
// the per-video product alpha U lands directly in Y0 with the residual added in its epilogue, and the R x D x D output
// projection (18 % of the step's FLOPs) disappears.  Same mathematics, a different association of the fp32 products.
static int vasnet_forward_impl(float* x, int32_t D, int32_t n_seq, const int32_t* seq_off_host,
                               const int32_t* seq_off_dev, const sumk_vasnet_weights* w,
                               const sumk_vasnet_opts* opts, const float* pos_table, const int32_t* pos_rows,
                               float* scores, void* workspace, size_t workspace_bytes, int32_t training,
                               void* stream_, const float* Wvo) {
  hipStream_t stream = (hipStream_t)stream_;
  SUMK_ARG(Wvo == nullptr || !training, "vasnet_forward: the folded projection is an inference-only path");
  SUMK_ARG(x && seq_off_dev && w && opts && scores && workspace, "vasnet_forward: null pointer");
  SUMK_ARG(w->Wk && w->Wq && w->Wv && w->Wo && w->W1 && w->b1 && w->w2 && w->b2 && w->ln_w && w->ln_b,
           "vasnet_forward: null weight");
  SUMK_ARG((pos_table == nullptr) == (pos_rows == nullptr), "vasnet_forward: pos_table and pos_rows go together");
  SUMK_ARG(opts->dropout_p >= 0.f && opts->dropout_p < 1.f, "vasnet_forward: dropout_p=%f out of [0,1)", opts->dropout_p);
  SUMK_ARG(opts->dropout_p == 0.f || training, "vasnet_forward: dropout needs training mode");
  SUMK_ARG(opts->precision >= SUMK_PRECISION_FP32 && opts->precision <= SUMK_PRECISION_MAX, "vasnet_forward: unknown precision %d", opts->precision);
  Geometry G;
  SUMK_TRY(geometry(D, n_seq, seq_off_host, training, opts->precision, workspace_bytes, &G));
  const VasnetWs& L = G.L;
  if (workspace_bytes < (G.b16 ? L.total : L.total_core)) {
    set_error("vasnet_forward: workspace %zu < required %zu", workspace_bytes, G.b16 ? L.total : L.total_core);
    return SUMK_ERR_WORKSPACE;
  }
  char* ws = (char*)workspace;
  char* tb = opts->tables ? (char*)opts->tables : ws;      // table base: the caller's prebuilt block, or the front of the workspace (built per call)
  if (G.sk && !Wvo)
    return vasnet_forward_sk(G, x, D, n_seq, seq_off_host, seq_off_dev, w, opts, pos_table, pos_rows, scores, ws, tb, training, stream);
  const int R = G.R;
  float* QKV = (float*)(ws + L.qkv);
  float* E = (float*)(ws + L.e);
  float* E2 = training ? (float*)(ws + L.e2) : nullptr;
  float* CTX = (float*)(ws + L.ctx);
  float* Y0 = (float*)(ws + L.y0);
  float* Y1 = (float*)(ws + L.y1);
  float* Z = (float*)(ws + L.z);
  SeqInfo* seq = (SeqInfo*)(tb + L.seq);
  GemmProb* prow = (GemmProb*)(tb + L.prob_row);
  GemmProb* tabs = (GemmProb*)(tb + L.prob_seq);
  float* stats = training ? (float*)(ws + L.stats) : nullptr;
  const Drop drop = make_drop(opts);
  const bool use_e2 = training && drop.thr != 0;

  if (pos_table) {
    int64_t n4 = (int64_t)R * (D >> 2);
    hipLaunchKernelGGL(add_pos_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, x, pos_table, pos_rows, R, D);
  }
  if (tb == ws) launch_setup(G, D, n_seq, seq_off_dev, tb, stream);
  // mixed-precision training: bf16 shadows of x and of the five weight matrices, then every row-wise GEMM reads bf16 from HBM
  const bool b16 = G.b16;
  SUMK_ARG(!(b16 && opts->x16 && pos_table), "vasnet_forward: opts->x16 (a shadow of x) cannot be combined with pos_table (x changes in place)");
  const unsigned short* x16 = (b16 && opts->x16) ? (const unsigned short*)opts->x16 : (const unsigned short*)(ws + L.x16);
  unsigned short* Wqkv16 = (unsigned short*)(ws + L.w16);
  unsigned short* Wo16 = Wqkv16 + (size_t)3 * D * D;
  unsigned short* W116 = Wo16 + (size_t)D * D;
  if (b16) {   // one launch: the five matrices ([Wq; Wk; Wv] land stacked: one 3D x D operand) and, unless the caller keeps its shadow, x
    const float* const src[6] = {w->Wq, w->Wk, w->Wv, w->Wo, w->W1, x};
    void* const dst[6] = {Wqkv16, Wqkv16 + (size_t)D * D, Wqkv16 + (size_t)2 * D * D, Wo16, W116, ws + L.x16};
    const int64_t dd = (int64_t)D * D;
    const int64_t ne[6] = {dd, dd, dd, dd, dd, (int64_t)R * D};
    SUMK_TRY(cast_flat_b16(opts->x16 ? 5 : 6, src, dst, ne, stream));
  }

  // row-wise NT GEMMs with K = D: eligible for the buffer-load instances when D is a whole number of k-tiles and byte offsets fit 31 bits
  const int lean_rows = (D % 32 == 0 && (int64_t)R * 3 * D * 4 < ((int64_t)1 << 31)) ? 1 : 0;
  // Plane path (inference in bf16x6 / bf16x3 with the caller's x planes and weight-plane block): the three row-wise GEMMs on gemm_pw.hip.
  // (A/B: a call without the plane pointers runs the in-loop split kernels; tests/test_gpu_planes.py compares the two.)
  const int np = opts->precision == SUMK_PRECISION_BF16X6 ? 3 : opts->precision == SUMK_PRECISION_BF16X3 ? 2 : 0;
  const bool pw = np && !training && opts->xplanes && opts->wplanes && !pos_table && wplanes_ok(D, np) && R >= 256 && G.st_qkv == 0 &&
                  G.st_d == 0 && pw_ok(R, 3 * (int64_t)D, D, R, 3 * (int64_t)D, np) && (Wvo == nullptr || G.cfg_pv == 1);
  const WPlanes wl = pw ? wplanes_layout(D, np) : WPlanes();
  const char* const wp = (const char*)opts->wplanes;
  // ... and the per-video attention on planes as well (attn_pw.hip): T <= 320, the extra workspace present (folded path: the context strips add the residual and emit the moments)
  const PwExtra px = pw ? pw_extra(D, R, G.t_max, np, L.total_core) : PwExtra();
  const bool pw_attn = pw && attn_pw_ok(G.t_max, D, R, np) && workspace_bytes >= px.total;
  // Long videos on planes (round 6): per video, logits and context on the plane GEMM itself; the context leaves as planes (rows of the video inside
  // the batch's CTX planes), so the output projection below reads them as on the strip path.  SUMK_PW_LONG=0: the in-loop grouped kernels (A/B).
  int t_min_h = G.t_max;
  for (int q = 0; q < n_seq; ++q) t_min_h = std::min(t_min_h, seq_off_host[q + 1] - seq_off_host[q]);
  static const bool pw_long_on = !(getenv("SUMK_PW_LONG") && getenv("SUMK_PW_LONG")[0] == '0');
  const PwLong pl = (pw && !pw_attn) ? pw_long_layout(D, R, G.t_max, np, L.total_core) : PwLong();
  const bool pw_long = pw && !pw_attn && !Wvo && pw_long_on && pw_long_ok(D, t_min_h, G.t_max, np) && workspace_bytes >= pl.total;
  if (pw_attn) {  // 1-4: projection -> planes of [Q | K | V]; logits + softmax -> alpha planes; alpha . V -> context planes
    PwLaunch g; g.A = opts->xplanes; g.a_rows = R; g.B = wp + wl.wqkv; g.b_rows = 3 * (int64_t)D; g.M = R; g.N = 3 * D; g.K = D; g.np = np;
    // the context kernel multiplies V rows up to 31 past the last video's end by alpha = 0: every row up to the pitch is stored (zeros: x's pad rows are
    // zero) and the slack a read past the LAST sub-array lands in is cleared -- 0 x (stale NaN bits) would be NaN
    g.O = ws + px.qkv; g.o_rows = R; g.o_store_rows = pw_rows_pitch(R); g.prof_tag = SUMK_PROF_GEMM_QKV;
    SUMK_HIP(hipMemsetAsync(ws + px.qkv + pw_planes_bytes(R, 3 * D, np) - 8192, 0, 8192, stream));
    SUMK_TRY(launch_gemm_pw(PW_PLANES, g, stream));
    prof_begin(SUMK_PROF_GEMM_QKT, stream);
    SUMK_TRY(launch_attn_pw_logits(np, ws + px.qkv, R, D, nullptr, ws + px.ap, seq, n_seq, G.t_max, opts->scale, opts->ignore_self, opts->aperture, stream));
    prof_end(SUMK_PROF_GEMM_QKT, stream);
    prof_begin(SUMK_PROF_GEMM_PV, stream);
    if (Wvo) {   // folded: "V" = x Wvo^T, so alpha . V + x IS Y0 -- it leaves as planes with its LayerNorm moments (D / 32 slots per row; the small arrays live in Z)
      SUMK_TRY(launch_attn_pw_context(np, ws + px.qkv, R, D, ws + px.ap, ws + px.ctx, seq, n_seq, G.t_max, stream, 1, x, D, (float*)(ws + L.z)));
    } else {
      SUMK_TRY(launch_attn_pw_context(np, ws + px.qkv, R, D, ws + px.ap, ws + px.ctx, seq, n_seq, G.t_max, stream));
    }
    prof_end(SUMK_PROF_GEMM_PV, stream);
  } else
  if (pw_long && pl.qk_direct) {  // 1: [Q | K] straight to planes (every video's Q and K are row ranges of them), V as fp32 (R x D: its transpose is split per video)
    PwLaunch g; g.A = opts->xplanes; g.a_rows = R; g.B = wp + wl.wqkv; g.b_rows = 3 * (int64_t)D; g.M = R; g.N = 2 * D; g.K = D; g.np = np;
    g.O = ws + pl.qk; g.o_rows = R; g.o_store_rows = pw_rows_pitch(R); g.prof_tag = SUMK_PROF_GEMM_QKV;
    SUMK_HIP(hipMemsetAsync(ws + pl.qk + pw_planes_bytes(R, 2 * D, np) - 8192, 0, 8192, stream));      // the last video's logits tiles read key rows past the array's last sub-array
    SUMK_TRY(launch_gemm_pw(PW_PLANES, g, stream));
    PwLaunch gv; gv.A = opts->xplanes; gv.a_rows = R; gv.B = wp + wl.wqkv + (size_t)2 * D * 16; gv.b_rows = 3 * (int64_t)D; gv.M = R; gv.N = D; gv.K = D; gv.np = np;
    gv.C = QKV; gv.ldc = D; gv.prof_tag = SUMK_PROF_GEMM_QKV;
    SUMK_TRY(launch_gemm_pw(PW_F32, gv, stream));
  } else
  if (pw) {  // 1: QKV projection from planes (fp32 output: the per-video products below read it)
    PwLaunch g; g.A = opts->xplanes; g.a_rows = R; g.B = wp + wl.wqkv; g.b_rows = 3 * (int64_t)D; g.M = R; g.N = 3 * D; g.K = D; g.np = np;
    g.C = QKV; g.ldc = 3 * D; g.prof_tag = SUMK_PROF_GEMM_QKV;
    SUMK_TRY(launch_gemm_pw(PW_F32, g, stream));
  } else
  {  // 1: QKV projection
    GemmLaunch g; g.precision = opts->precision;
    g.A = x; g.B[0] = w->Wq; g.B[1] = w->Wk; g.B[2] = Wvo ? Wvo : w->Wv; g.n_group = D; g.C = QKV; g.probs = prow + RP_QKV;
    g.small_tile = G.st_qkv; g.total_tiles = gemm_tiles(R, 3 * D, G.st_qkv); g.prof_tag = SUMK_PROF_GEMM_QKV;
    g.xcd_M = R; g.xcd_N = 3 * D;
    g.lean = lean_rows;
    // (b16: Q / K / V are read by bf16-source GEMMs only -- the fp32 form, 147 MB of stores on S-TVSum, is not written at all)
    if (b16) { to_b16(g, x16, Wqkv16, R, 3 * D, prow, RP_QKV_W); g.C16 = ws + L.qkv16; g.C = nullptr; }
    SUMK_TRY(launch_gemm(GEMM_NT, EPI_NONE, g, stream));
  }
  if (pw_long) {
    char* const ctxp = ws + L.y0;                       // = pw_ctxp below: planes of the (R x D) context
    float* const Es = (float*)(ws + pl.e);
    for (int q = 0; q < n_seq; ++q) {
      const int r0 = seq_off_host[q], T = seq_off_host[q + 1] - r0;
      const int Tn = (T + 255) / 256 * 256, Kp = (T + 31) / 32 * 32;
      // planes of this video's Q and K: row ranges of the batch's [Q | K] planes, or split here from the fp32 projection (the K array then has the pitch of
      // Tn rows).  Either way the logits tiles read key rows up to Tn -- whatever those hold lands in columns [T, Tn) of E, which the softmax never reads as keys
      const char* qpl; const char* kpl; int64_t qrows, krows;
      if (pl.qk_direct) {
        qpl = ws + pl.qk + (size_t)r0 * 16; kpl = ws + pl.qk + (size_t)(D >> 4) * np * 2 * (pw_rows_pitch(R) * 16) + (size_t)r0 * 16; qrows = krows = R;
        SUMK_TRY(split_planes_t(QKV + (size_t)r0 * D, T, D, D, np, ws + pl.vt, Kp, stream));
      } else {
        SUMK_TRY(split_planes(QKV + (size_t)r0 * 3 * D, T, D, 3 * D, np, ws + pl.qp, stream));
        SUMK_TRY(split_planes_pitched(QKV + (size_t)r0 * 3 * D + D, T, D, 3 * D, np, ws + pl.kp, Tn, stream));
        SUMK_TRY(split_planes_t(QKV + (size_t)r0 * 3 * D + 2 * D, T, D, 3 * D, np, ws + pl.vt, Kp, stream));
        qpl = ws + pl.qp; kpl = ws + pl.kp; qrows = T; krows = Tn;
      }
      {  // 2: raw logits  E_s = Q_s K_s^T  (T x Tn, fp32)
        PwLaunch g; g.A = qpl; g.a_rows = qrows; g.B = kpl; g.b_rows = krows; g.M = T; g.N = Tn; g.K = D; g.np = np;
        g.C = Es; g.ldc = Tn; g.prof_tag = SUMK_PROF_GEMM_QKT;
        SUMK_TRY(launch_gemm_pw(PW_F32, g, stream));
      }
      // 3: scale, masks, softmax -> planes of alpha_s
      SUMK_TRY(softmax_planes(Es, T, Tn, np, ws + pl.ap, Kp, opts->scale, opts->ignore_self, opts->aperture, (float*)(ws + pl.st), stream));
      {  // 4: context rows of this video = alpha_s V_s, straight into the batch's CTX planes
        PwLaunch g; g.A = ws + pl.ap; g.a_rows = T; g.B = ws + pl.vt; g.b_rows = D; g.M = T; g.N = D; g.K = Kp; g.np = np;
        g.O = ctxp + (size_t)r0 * 16; g.o_rows = R; g.prof_tag = SUMK_PROF_GEMM_PV;
        SUMK_TRY(launch_gemm_pw(PW_PLANES, g, stream));
      }
    }
  } else
  if (pw_attn) {
  } else
  if (G.attn_fused) {  // 2-4 in one launch per (video, 64-row strip): logits, softmax (+ dropout), context
    AttnStripArgs at;
    const unsigned short* qkv16 = (const unsigned short*)(ws + L.qkv16);
    at.A16 = qkv16; at.lda = 3 * D; at.B16 = qkv16 + D; at.ldb = 3 * D; at.C16 = qkv16 + 2 * D; at.ldc = 3 * D;
    at.O16 = (unsigned short*)(ws + L.ctx16); at.ldo = D; at.E = E; at.P16 = (unsigned short*)(ws + L.p16);
    at.seq = seq; at.n_seq = n_seq; at.strips = (G.t_max + 63) / 64; at.D = D;
    at.scale = opts->scale; at.ignore_self = opts->ignore_self; at.aperture = opts->aperture; at.drop = drop;
    SUMK_TRY(launch_attn_strip(false, at, stream));
  } else {
  {  // 2: logits per video
    GemmLaunch g; g.precision = opts->precision;
    g.A = QKV; g.B[0] = QKV; g.C = E; g.probs = tabs + TB_S * n_seq; g.nprob = n_seq; g.small_tile = G.cfg_s;
    g.total_tiles = G.tiles_s; g.prof_tag = SUMK_PROF_GEMM_QKT;
    if (b16) { g.A = g.B[0] = (const float*)(ws + L.qkv16); g.src16 = 1; }
    SUMK_TRY(launch_gemm(GEMM_NT, EPI_NONE, g, stream));
  }
  // 3: softmax (+ dropout of alpha into E2 when training with p > 0)
  {
    int t_max = 0;
    for (int q = 0; q < n_seq; ++q) t_max = std::max(t_max, seq_off_host[q + 1] - seq_off_host[q]);
    const dim3 sg((R + 3) / 4), sb(256);
    float* e2p = use_e2 ? E2 : nullptr;
    unsigned short* p16 = b16 ? (unsigned short*)(ws + L.p16) : nullptr;
#define SUMK_SOFTMAX(NR) hipLaunchKernelGGL(vasnet_softmax_kernel<NR>, sg, sb, 0, stream, E, e2p, seq, (const int32_t*)(tb + L.row_seq), n_seq, R, opts->scale, opts->ignore_self, opts->aperture, drop, p16, (const float*)E, 0, (int64_t)0)
    if (t_max <= 256) SUMK_SOFTMAX(4); else if (t_max <= 512) SUMK_SOFTMAX(8); else if (t_max <= 1024) SUMK_SOFTMAX(16); else SUMK_SOFTMAX(0);
#undef SUMK_SOFTMAX
  }
  }
  // Fused inference tail (128x128 tiles, no dropout; SUMK_FUSED_HEAD=0 / SUMK_FUSED_LN=0 are the A/B switches, training always
  // runs the separate kernels):
  //  (b) fused_tail: the k1 epilogue reduces relu(.) to the LayerNorm + k2 moments -- Z never exists;
  //  (a) fused_ln:   the epilogue of the GEMM that produces Y0 also emits its per-row moments, one prep launch folds the first
  //      LayerNorm's gain into W1 and turns the moments into {mean, rstd}, and the k1 GEMM reads the RAW Y0: the LayerNorm is
  //      applied to the product in the k1 epilogue -- no LayerNorm kernel, Y1 never exists.  Its scratch (moments, folded
  //      weights, c1 / c2, stats) lives in the unused Y1 region, which must be large enough (R >~ 1.1 D).
  static const bool fused_head_on = !(getenv("SUMK_FUSED_HEAD") && getenv("SUMK_FUSED_HEAD")[0] == '0');
  static const bool fused_ln_on = !(getenv("SUMK_FUSED_LN") && getenv("SUMK_FUSED_LN")[0] == '0');
  const bool fused_tail = fused_head_on && !training && drop.thr == 0 && G.st_d == 0 && D % 64 == 0;
  // producer of Y0: the output projection (128x128 tiles, D / 32 moment slots per row) or, on the folded path, the per-video
  // alpha.(X Wvo) product with the residual (64x64 tiles, D / 16 slots)
  const int ln_slots = Wvo ? D / 16 : D / 32;
  const size_t ln_mom_f = align_up((size_t)R * ln_slots * 2, 64), ln_w_f = (size_t)D * D, ln_c_f = align_up((size_t)2 * D, 64);
  const bool fused_ln = pw || (fused_tail && fused_ln_on && (!Wvo || G.cfg_pv == 1) && ln_mom_f + ln_w_f + ln_c_f + (size_t)2 * R <= (size_t)R * D);
  // plane path: CTX planes span the Y0 / Y1 regions, Y0's planes take the (by then dead) fp32 Q/K/V region, the small per-row arrays live in Z
  char* const pw_ctxp = pw_attn ? ws + px.ctx : ws + L.y0;
  char* const pw_y0p = (pw_attn && Wvo) ? ws + px.ctx : pw_attn ? ws + px.qkv : ws + L.qkv;          // ([Q | K | V]'s planes are dead once the context exists; folded: the context strips wrote Y0's planes)
  float* const pw_mom = Z;                                                        // float2[R][slots]
  float* const pw_stats = Z + align_up((size_t)R * (D / 16) * 2, 64);             // float2[R]
  float* const pw_part = pw_stats + align_up((size_t)R * 2, 64);                  // float4[R][D / 64]
  float* ln_moments = pw ? pw_mom : Y1;
  float* ln_W1g = Y1 + ln_mom_f;
  float* ln_c1 = ln_W1g + ln_w_f;
  float* ln_stats = ln_c1 + ln_c_f;
  if (!G.attn_fused && !pw_attn && !pw_long) {  // 4: context
    GemmLaunch g; g.precision = opts->precision;
    g.A = use_e2 ? E2 : E; g.B[0] = QKV; g.C = Wvo ? Y0 : CTX; g.R = x; g.probs = tabs + TB_PV * n_seq; g.nprob = n_seq; g.small_tile = G.cfg_pv;
    g.total_tiles = G.tiles_pv; g.prof_tag = SUMK_PROF_GEMM_PV;
    if (Wvo && fused_ln) g.moments = ln_moments;
    if (b16) { g.A = (const float*)(ws + L.p16); g.B[0] = (const float*)(ws + L.qkv16); g.src16 = 1; g.C16 = ws + L.ctx16; g.C = nullptr; }
    SUMK_TRY(launch_gemm(GEMM_NN, Wvo ? (fused_ln ? EPI_RESIDUAL_MOMENTS : EPI_RESIDUAL) : EPI_NONE, g, stream));
    if (Wvo && pw) {        // folded plane path: Y0 (fp32, with its moments) -> {mean, rstd}, and its planes for k1
      hipLaunchKernelGGL(ln_row_stats_kernel, dim3((R + 31) / 32), dim3(256), 0, stream, (const float2*)pw_mom, ln_slots, R, D, opts->eps, (float2*)pw_stats);
      SUMK_HIP(hipGetLastError());
      SUMK_TRY(split_planes(Y0, R, D, D, np, pw_y0p, stream));
    } else
    if (Wvo && fused_ln) {
      hipLaunchKernelGGL(ln_fold_stats_kernel, dim3(D + (R + 31) / 32), dim3(256), 0, stream, w->W1, w->ln_w, w->ln_b, D, ln_W1g, ln_c1,
                         ln_c1 + D, (const float2*)ln_moments, ln_slots, R, opts->eps, (float2*)ln_stats);
      SUMK_HIP(hipGetLastError());
    }
  }
  if (pw_attn && Wvo) {   // (4 + 5 happened in the context strips) moments -> {mean, rstd}
    hipLaunchKernelGGL(ln_row_stats_kernel, dim3((R + 31) / 32), dim3(256), 0, stream, (const float2*)pw_mom, D / 32, R, D, opts->eps, (float2*)pw_stats);
    SUMK_HIP(hipGetLastError());
  } else
  if (pw && !Wvo) {  // 5: output projection + residual from planes: CTX is split once, Y0 leaves as planes + per-row moments only
    if (!pw_attn && !pw_long) SUMK_TRY(split_planes(CTX, R, D, D, np, pw_ctxp, stream));
    PwLaunch g; g.A = pw_ctxp; g.a_rows = R; g.B = wp + wl.wo; g.b_rows = D; g.M = R; g.N = D; g.K = D; g.np = np;
    g.R = x; g.ldr = D; g.moments = pw_mom; g.O = pw_y0p; g.o_rows = R; g.prof_tag = SUMK_PROF_GEMM_OPROJ;
    SUMK_TRY(launch_gemm_pw(PW_RES_MOM_PLANES, g, stream));
    hipLaunchKernelGGL(ln_row_stats_kernel, dim3((R + 31) / 32), dim3(256), 0, stream, (const float2*)pw_mom, D / 64, R, D, opts->eps, (float2*)pw_stats);
    SUMK_HIP(hipGetLastError());
  } else
  if (!Wvo) {  // 5: output projection + residual
    GemmLaunch g; g.precision = opts->precision;
    g.A = CTX; g.B[0] = w->Wo; g.C = Y0; g.R = x; g.probs = prow + RP_DD; g.small_tile = G.st_d;
    g.total_tiles = gemm_tiles(R, D, G.st_d); g.xcd_M = R; g.xcd_N = D; g.prof_tag = SUMK_PROF_GEMM_OPROJ;
    g.lean = lean_rows;
    if (fused_ln) {
      g.moments = ln_moments;
      SUMK_TRY(launch_gemm(GEMM_NT, EPI_RESIDUAL_MOMENTS, g, stream));
      hipLaunchKernelGGL(ln_fold_stats_kernel, dim3(D + (R + 31) / 32), dim3(256), 0, stream, w->W1, w->ln_w, w->ln_b, D, ln_W1g, ln_c1,
                         ln_c1 + D, (const float2*)ln_moments, ln_slots, R, opts->eps, (float2*)ln_stats);
      SUMK_HIP(hipGetLastError());
    } else {
      if (b16) to_b16(g, ws + L.ctx16, Wo16, R, D, prow, RP_DD_W);
      SUMK_TRY(launch_gemm(GEMM_NT, EPI_RESIDUAL, g, stream));
    }
  }
  // 6: dropout + LayerNorm
  // (b16: the LayerNorm output is consumed by bf16-source GEMMs only -- k1 here, dW1 in the backward -- so only its bf16 form is written)
  if (!fused_ln) launch_ln_rows<false>(Y0, b16 ? nullptr : Y1, w->ln_w, w->ln_b, nullptr, nullptr, nullptr, R, D, opts->eps, stats, drop, 1u, stream,
                                       b16 ? (unsigned short*)(ws + L.y116) : nullptr);
  // 7 + 8 fused (inference, 128x128 tiles): k1 + bias + ReLU with the LayerNorm + k2 moments taken in the GEMM epilogue -- the
  // (R, D) activation matrix is neither written (49 MB in the lock-stepped store burst of this single-round launch) nor read
  // back by a LayerNorm kernel.
  if (pw) {   // 7 + 8: k1 on the RAW Y0 planes with the LayerNorm applied to the product, moments of relu(.) per 64-column slot, then the head
    PwLaunch g; g.A = pw_y0p; g.a_rows = R; g.B = wp + wl.w1g; g.b_rows = D; g.M = R; g.N = D; g.K = D; g.np = np;
    g.bias = (const float*)(wp + wl.biasc); g.gw = (const float*)(wp + wl.gw); g.ln_c1 = (const float*)(wp + wl.c1); g.ln_stats = pw_stats;
    g.head_part = pw_part; g.prof_tag = SUMK_PROF_GEMM_K1;
    SUMK_TRY(launch_gemm_pw(PW_HEAD, g, stream));
    hipLaunchKernelGGL(head_finalize_kernel, dim3((R + 31) / 32), dim3(256), 0, stream, (const float4*)pw_part, D / 64, R, D, w->ln_w,
                       w->ln_b, w->w2, w->b2, opts->eps, scores);
    SUMK_HIP(hipGetLastError());
    return SUMK_OK;
  }
  if (fused_tail) {
    GemmLaunch g; g.precision = opts->precision;
    g.A = Y1; g.B[0] = w->W1; g.bias0[0] = w->b1; g.bias1[0] = w->ln_w; g.bias1[1] = w->w2;
    if (fused_ln) { g.A = Y0; g.B[0] = ln_W1g; g.ln_stats = ln_stats; g.ln_c1 = ln_c1; g.ln_c2 = ln_c1 + D; }
    g.C = Z;                       // reused as float4[R][D / 32] moments (R * D / 8 floats of the R * D region)
    g.probs = prow + RP_DD; g.small_tile = 0;
    g.total_tiles = gemm_tiles(R, D, 0); g.xcd_M = R; g.xcd_N = D; g.prof_tag = SUMK_PROF_GEMM_K1;
    g.lean = lean_rows;
    SUMK_TRY(launch_gemm(GEMM_NT, EPI_BIAS_RELU_HEAD, g, stream));
    hipLaunchKernelGGL(head_finalize_kernel, dim3((R + 31) / 32), dim3(256), 0, stream, (const float4*)Z, D / 32, R, D, w->ln_w,
                       w->ln_b, w->w2, w->b2, opts->eps, scores);
    SUMK_HIP(hipGetLastError());
    return SUMK_OK;
  }
  {  // 7: k1 + bias + ReLU
    GemmLaunch g; g.precision = opts->precision;
    g.A = Y1; g.B[0] = w->W1; g.bias0[0] = w->b1; g.C = Z; g.probs = prow + RP_DD; g.small_tile = G.st_d;
    g.total_tiles = gemm_tiles(R, D, G.st_d); g.xcd_M = R; g.xcd_N = D; g.prof_tag = SUMK_PROF_GEMM_K1;
    g.lean = lean_rows;
    if (b16) to_b16(g, ws + L.y116, W116, R, D, prow, RP_DD_W);
    SUMK_TRY(launch_gemm(GEMM_NT, EPI_BIAS_RELU, g, stream));
  }
  // 8: dropout + LayerNorm (same weights) + k2 + sigmoid
  launch_ln_rows<true>(Z, training ? (float*)(ws + L.scores) : nullptr, w->ln_w, w->ln_b, w->w2, w->b2, scores, R, D, opts->eps, stats ? stats + 2 * (size_t)R : nullptr, drop, 2u, stream);
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}

extern "C" int sumk_vasnet_forward(float* x, int32_t D, int32_t n_seq, const int32_t* seq_off_host,
                                   const int32_t* seq_off_dev, const sumk_vasnet_weights* w,
                                   const sumk_vasnet_opts* opts, const float* pos_table, const int32_t* pos_rows,
                                   float* scores, void* workspace, size_t workspace_bytes, int32_t training,
                                   void* stream) {
  return vasnet_forward_impl(x, D, n_seq, seq_off_host, seq_off_dev, w, opts, pos_table, pos_rows, scores, workspace,
                             workspace_bytes, training, stream, nullptr);
}

extern "C" int sumk_vasnet_forward_folded(float* x, int32_t D, int32_t n_seq, const int32_t* seq_off_host,
                                          const int32_t* seq_off_dev, const sumk_vasnet_weights* w, const float* Wvo,
                                          const sumk_vasnet_opts* opts, const float* pos_table, const int32_t* pos_rows,
                                          float* scores, void* workspace, size_t workspace_bytes, void* stream) {
  SUMK_ARG(Wvo != nullptr, "vasnet_forward_folded: null folded weight");
  return vasnet_forward_impl(x, D, n_seq, seq_off_host, seq_off_dev, w, opts, pos_table, pos_rows, scores, workspace,
                             workspace_bytes, 0, stream, Wvo);
}

template <bool HEAD>
static int launch_ln_bwd(int D, int R, const float* X, const float* stats, const float* g, const float* b,
                         const float* dY, const float* w2, const float* scores, const float* dscores, float* dX,
                         float* part, Drop drop, uint32_t site, int* n_waves_out, hipStream_t stream, unsigned short* dX16 = nullptr,
                         int n_slab = 0, int64_t slab_stride = 0) {
  const int D4 = D >> 2;
  const int nq = (D4 + 63) / 64;
  // 512 blocks of four waves: with every load of a row issued before its first use (PRE) a wave holds 200-240 VGPRs, two waves per SIMD =
  // 2,048 resident, each walking ~6 rows of the 50-video batch.  Measured against the per-chunk form on 768 blocks (three waves per SIMD):
  // 29.8 / 29.3 -> 27.9 / 27.7 us for the two LayerNorm backward launches of the step, and 256 fewer slots for the slot reduce (8.6 -> 6.7 us).
  int blocks = std::min((R + 3) / 4, std::min(512, LNB_MAX_WAVES / 4));
  blocks = std::max(blocks, 1);
  *n_waves_out = blocks;              // slots written: one per block
  dim3 grid(blocks), block(256);
#define LNB(NQ) hipLaunchKernelGGL((layernorm_bwd_kernel<NQ, HEAD>), grid, block, 0, stream, X, stats, g, b, dY, w2, scores, dscores, dX, part, R, D, drop, site, dX16, n_slab, slab_stride)
  if (nq <= 1) LNB(1); else if (nq <= 2) LNB(2); else if (nq <= 4) LNB(4); else if (nq <= 8) LNB(8);
  else { set_error("vasnet_backward: D=%d > 2048 is not supported by the LayerNorm backward kernel", D); return SUMK_ERR_ARG; }
#undef LNB
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}

namespace sumk {
int ln_bwd_reduce(const float* part, int n_slots, int D, float* dgamma, float* dbeta, float* dw2, float* db2, float* dcol,
                  hipStream_t stream) {
  const ReduceSeg segs[5] = {{0, D, dgamma}, {D, D, dbeta}, {2 * D, D, dw2}, {3 * D, D, dcol}, {4 * D, 1, db2}};
  return partial_reduce_multi(part, n_slots, ln_slot_floats(D), segs, 5, stream);
}
int launch_ln_bwd_rows(int D, int R, const float* X, const float* stats, const float* g, const float* b, const float* dY,
                       float* dX, float* part, Drop drop, uint32_t site, int* n_waves, hipStream_t stream) {
  return launch_ln_bwd<false>(D, R, X, stats, g, b, dY, nullptr, nullptr, nullptr, dX, part, drop, site, n_waves, stream);
}
int launch_ln_head_bwd(int D, int R, const float* Z, const float* stats, const float* g, const float* b, const float* w2,
                       const float* scores, const float* dscores, float* dZ, float* part, Drop drop, uint32_t site,
                       int* n_waves, hipStream_t stream) {
  return launch_ln_bwd<true>(D, R, Z, stats, g, b, nullptr, w2, scores, dscores, dZ, part, drop, site, n_waves, stream);
}
}  // namespace sumk

extern "C" int sumk_vasnet_backward(const float* x, int32_t D, int32_t n_seq, const int32_t* seq_off_host,
                                    const int32_t* seq_off_dev, const sumk_vasnet_weights* w,
                                    const sumk_vasnet_opts* opts, const float* dscores, const sumk_vasnet_grads* gr,
                                    float* dx, void* workspace, size_t workspace_bytes, void* stream_,
                                    void* tail_grads_ready_event) {
  hipStream_t stream = (hipStream_t)stream_;
  SUMK_ARG(x && seq_off_dev && w && opts && dscores && gr && workspace, "vasnet_backward: null pointer");
  SUMK_ARG(gr->Wk && gr->Wq && gr->Wv && gr->Wo && gr->W1 && gr->b1 && gr->w2 && gr->b2 && gr->ln_w && gr->ln_b,
           "vasnet_backward: null gradient target");
  Geometry G;
  SUMK_TRY(geometry(D, n_seq, seq_off_host, 1, opts->precision, workspace_bytes, &G));
  const VasnetWs& L = G.L;
  if (workspace_bytes < (G.b16 ? L.total : L.total_core)) {
    set_error("vasnet_backward: workspace %zu < required %zu (needs the training-mode forward's workspace)", workspace_bytes, G.b16 ? L.total : L.total_core);
    return SUMK_ERR_WORKSPACE;
  }
  char* ws = (char*)workspace;
  const int R = G.R;
  float* QKV = (float*)(ws + L.qkv);
  float* E = (float*)(ws + L.e);
  float* E2 = (float*)(ws + L.e2);
  float* CTX = (float*)(ws + L.ctx);
  float* Y0 = (float*)(ws + L.y0);
  float* Y1 = (float*)(ws + L.y1);
  float* Z = (float*)(ws + L.z);
  float* dZ = (float*)(ws + L.dz);
  float* dY1 = (float*)(ws + L.dy1);
  float* dY0 = (float*)(ws + L.dy0);
  float* dCTX = (float*)(ws + L.dctx);
  float* dQKV = (float*)(ws + L.dqkv);
  float* lnpart = (float*)(ws + L.lnpart);
  float* slab = (float*)(ws + L.slab);
  float* stats = (float*)(ws + L.stats);
  const float* scores = (const float*)(ws + L.scores);
  char* tb = opts->tables ? (char*)opts->tables : ws;      // as in the forward
  SeqInfo* seq = (SeqInfo*)(tb + L.seq);
  GemmProb* prow = (GemmProb*)(tb + L.prob_row);
  GemmProb* tabs = (GemmProb*)(tb + L.prob_seq);
  // K-slice tables of the two weight-gradient launches: in the table block; prebuilt when the caller keeps the block (opts->tables)
  GemmProb* pskw = (GemmProb*)(tb + L.prob_skw);
  const bool skw_ready = tb != ws;
  const Drop drop = make_drop(opts);
  const bool use_e2 = drop.thr != 0;
  int nw = 0;
  // bf16-source row-wise GEMMs (see the forward): dZ and dY0 are produced in bf16 by the LayerNorm backward kernels, dQKV by the
  // per-video GEMM epilogues; fp32 dZ is never needed, fp32 dY0 only for the residual branch of dx
  const bool b16 = G.b16;
  const float* x16 = (G.b16 && opts->x16) ? (const float*)opts->x16 : (const float*)(ws + L.x16);
  const unsigned short* Wqkv16 = (const unsigned short*)(ws + L.w16);
  const float* Wo16 = (const float*)(Wqkv16 + (size_t)3 * D * D);
  const float* W116 = (const float*)(Wqkv16 + (size_t)4 * D * D);
  const float* CTX16 = (const float*)(ws + L.ctx16);
  const float* Y116 = (const float*)(ws + L.y116);
  unsigned short* dZ16 = (unsigned short*)(ws + L.dz16);
  unsigned short* dY016 = (unsigned short*)(ws + L.dy016);
  unsigned short* dQKV16 = (unsigned short*)(ws + L.dqkv16);

  if (G.sk && dx && tb == ws) launch_setup(G, D, n_seq, seq_off_dev, tb, stream);   // (dx asked for: the large-batch kernels below; their tables were not built by the SK forward)
  if (G.sk && !dx) {   // the small-batch backward: the same stages below, every GEMM an SK launch over the tables the forward built
    const SkCall none{};
    (void)none;
    SUMK_TRY(launch_ln_bwd<true>(D, R, Z, stats + 2 * (size_t)R, w->ln_w, w->ln_b, nullptr, w->w2, scores, dscores, dZ, lnpart, drop, 2u, &nw, stream));
    SUMK_TRY(ln_bwd_reduce(lnpart, nw, D, gr->ln_w, gr->ln_b, gr->w2, gr->b2, gr->b1, stream));
    float* scratch = (float*)(ws + L.sk_part);
    {
      const SkCall c{dZ, {w->W1, nullptr, nullptr, nullptr}, {scratch, nullptr, nullptr, nullptr}, nullptr, nullptr, -1};       // dY1 = dZ W1 as K-slice slabs: the LayerNorm backward kernel adds them
      SUMK_TRY(launch_sk(G, n_seq, ws, tb, GEMM_NN, EPI_NONE, SR_DY1, -1, c, stream));
    }
    SUMK_TRY(launch_ln_bwd<false>(D, R, Y0, stats, w->ln_w, w->ln_b, scratch, nullptr, nullptr, nullptr, dY0, lnpart, drop, 1u, &nw, stream, nullptr,
                                  G.P.row[SR_DY1].S, (int64_t)R * D));
    SUMK_TRY(ln_bwd_reduce(lnpart, nw, D, gr->ln_w, gr->ln_b, nullptr, nullptr, nullptr, stream));
    {
      const SkCall c{dY0, {CTX, nullptr, nullptr, nullptr}, {gr->Wo, gr->W1, nullptr, nullptr}, nullptr, nullptr, -1};      // dWo += dY0^T CTX and dW1 += dZ^T Y1 (group 1)
      SUMK_TRY(launch_sk(G, n_seq, ws, tb, GEMM_TN, EPI_ACCUM, SR_DWO1, -1, c, stream));
    }
    {
      const SkCall c{dY0, {w->Wo, nullptr, nullptr, nullptr}, {dCTX, nullptr, nullptr, nullptr}, nullptr, nullptr, -1};     // dCTX = dY0 Wo
      SUMK_TRY(launch_sk(G, n_seq, ws, tb, GEMM_NN, EPI_NONE, SR_DCTX, -1, c, stream));
    }
    if (tail_grads_ready_event) SUMK_HIP(hipEventRecord((hipEvent_t)tail_grads_ready_event, stream));
    const float* Pd = use_e2 ? E2 : E;
    {
      const SkCall c{Pd, {dCTX, nullptr, nullptr, nullptr}, {dQKV, nullptr, nullptr, nullptr}, nullptr, nullptr, -1};       // dV = alphaD^T dC
      SUMK_TRY(launch_sk(G, n_seq, ws, tb, GEMM_TN, EPI_NONE, -1, TB_DV, c, stream));
    }
    {
      const SkCall c{dCTX, {QKV, nullptr, nullptr, nullptr}, {scratch, nullptr, nullptr, nullptr}, nullptr, nullptr, -1};        // dAlphaD = dC V^T as slabs: the softmax backward kernel adds them
      SUMK_TRY(launch_sk(G, n_seq, ws, tb, GEMM_NT, EPI_NONE, -1, TB_DP, c, stream));
    }
#define SUMK_SOFTMAX_BWD(NR) hipLaunchKernelGGL(vasnet_softmax_bwd_kernel<NR>, dim3((R + 3) / 4), dim3(256), 0, stream, E, E2, seq, (const int32_t*)(tb + L.row_seq), n_seq, R, \
                       opts->scale, drop, (unsigned short*)nullptr, (const float*)scratch, G.P.seq[TB_DP].S, (int64_t)L.e_elems)
    if (G.t_max <= 256) SUMK_SOFTMAX_BWD(4); else if (G.t_max <= 512) SUMK_SOFTMAX_BWD(8); else if (G.t_max <= 1024) SUMK_SOFTMAX_BWD(16); else SUMK_SOFTMAX_BWD(0);
#undef SUMK_SOFTMAX_BWD
    {
      const SkCall c{E2, {QKV, nullptr, nullptr, nullptr}, {dQKV, nullptr, nullptr, nullptr}, nullptr, nullptr, -1};        // dQ = dS K
      SUMK_TRY(launch_sk(G, n_seq, ws, tb, GEMM_NN, EPI_NONE, -1, TB_DQ, c, stream));
    }
    {
      const SkCall c{E2, {QKV, nullptr, nullptr, nullptr}, {dQKV, nullptr, nullptr, nullptr}, nullptr, nullptr, -1};        // dK = dS^T Q
      SUMK_TRY(launch_sk(G, n_seq, ws, tb, GEMM_TN, EPI_NONE, -1, TB_DK, c, stream));
    }
    {
      const SkCall c{dQKV, {x, nullptr, nullptr, nullptr}, {gr->Wq, gr->Wk, gr->Wv, nullptr}, nullptr, nullptr, -1};        // d[Wq;Wk;Wv] += dQKV^T X
      SUMK_TRY(launch_sk(G, n_seq, ws, tb, GEMM_TN, EPI_ACCUM, SR_DWQKV, -1, c, stream));
    }
    SUMK_HIP(hipGetLastError());
    return SUMK_OK;
  }

  // 8': head + second LayerNorm + dropout + ReLU  ->  dZ (w.r.t. the k1 pre-activation), dw2, db2, dgamma, dbeta
  SUMK_TRY(launch_ln_bwd<true>(D, R, Z, stats + 2 * (size_t)R, w->ln_w, w->ln_b, nullptr, w->w2, scores, dscores, b16 ? nullptr : dZ, lnpart,
                               drop, 2u, &nw, stream, b16 ? dZ16 : nullptr));
  SUMK_TRY(ln_bwd_reduce(lnpart, nw, D, gr->ln_w, gr->ln_b, gr->w2, gr->b2, gr->b1, stream));   // db1 = column sums of dZ, from the same slots
  // 7': k1 -- dY1 = dZ . W1 (its weight gradient is taken together with the output projection's below: one split-K launch)
  {
    GemmLaunch g; g.precision = opts->precision;
    g.A = dZ; g.B[0] = w->W1; g.C = dY1; g.probs = prow + RP_DD; g.small_tile = G.st_d; g.total_tiles = gemm_tiles(R, D, G.st_d); g.xcd_M = R; g.xcd_N = D;
    if (b16) to_b16(g, dZ16, W116, R, D, prow, RP_DD_W);
    SUMK_TRY(launch_gemm(GEMM_NN, EPI_NONE, g, stream));
  }
  // 6': first LayerNorm + dropout -> dY0 (gradient of the residual sum)
  SUMK_TRY(launch_ln_bwd<false>(D, R, Y0, stats, w->ln_w, w->ln_b, dY1, nullptr, nullptr, nullptr, (b16 && !dx) ? nullptr : dY0, lnpart, drop, 1u, &nw, stream,
                                b16 ? dY016 : nullptr));
  SUMK_TRY(ln_bwd_reduce(lnpart, nw, D, gr->ln_w, gr->ln_b, nullptr, nullptr, nullptr, stream));
  // 5' + 7': dWo += dY0^T CTX and dW1 += dZ^T Y1 -- two (D x D) products over all R rows in ONE split-K launch (twice as long K
  // slices per block, one slab reduce) -- then dCTX = dY0 . Wo (+ residual branch into dx)
  {
    const float* const As[2] = {b16 ? (const float*)dY016 : dY0, b16 ? (const float*)dZ16 : dZ};
    const float* const Bs[2] = {b16 ? CTX16 : CTX, b16 ? Y116 : Y1};
    float* out[4] = {gr->Wo, gr->W1, nullptr, nullptr};
    SUMK_TRY(gemm_tn_splitk_accum_multi(2, As, Bs, D, D, D, D, R, slab, L.slab_elems, pskw, SPLITK_PROBS, out, D, D, 1.f, stream, opts->precision, b16,
                                        skw_ready ? SPLITK_TABLE_READY : SPLITK_BUILD_AND_RUN));
    GemmLaunch g; g.precision = opts->precision;  // dCTX = dY0 . Wo
    g.A = dY0; g.B[0] = w->Wo; g.C = dCTX; g.probs = prow + RP_DD; g.small_tile = G.st_d; g.total_tiles = gemm_tiles(R, D, G.st_d); g.xcd_M = R; g.xcd_N = D;
    if (b16) { to_b16(g, dY016, Wo16, R, D, prow, RP_DD_W); g.C16 = ws + L.dctx16; g.C = nullptr; }
    SUMK_TRY(launch_gemm(GEMM_NN, EPI_NONE, g, stream));
  }
  // Wo, W1, b1, w2, b2 (the tail of the parameter order) are final from here on: a data-parallel caller starts their
  // all-reduce on a side stream now, under the attention backward and the QKV weight gradients (60 % of the bucket).
  if (tail_grads_ready_event) SUMK_HIP(hipEventRecord((hipEvent_t)tail_grads_ready_event, stream));
  // 4': dV = alphaD^T dC ; dAlphaD = dC V^T
  const float* Pd = use_e2 ? E2 : E;
  if (!G.attn_fused) {
    GemmLaunch g; g.precision = opts->precision;
    g.A = Pd; g.B[0] = dCTX; g.C = dQKV; g.probs = tabs + TB_DV * n_seq; g.nprob = n_seq; g.small_tile = G.cfg_pv; g.total_tiles = G.tiles_pv;
    if (b16) { g.A = (const float*)(ws + L.p16); g.B[0] = (const float*)(ws + L.dctx16); g.src16 = 1; g.C16 = dQKV16; g.C = nullptr; }
    SUMK_TRY(launch_gemm(GEMM_TN, EPI_NONE, g, stream));
  }
  if (G.attn_fused) {  // dAlphaD, softmax backward and dQ in one launch per (video, 64-row strip); bf16(dLogits) replaces bf16(alphaD) in place
    AttnStripArgs at;
    const unsigned short* qkv16 = (const unsigned short*)(ws + L.qkv16);
    at.A16 = (const unsigned short*)(ws + L.dctx16); at.lda = D; at.B16 = qkv16 + 2 * D; at.ldb = 3 * D; at.C16 = qkv16 + D; at.ldc = 3 * D;
    at.O16 = (unsigned short*)dQKV16; at.ldo = 3 * D; at.E = E; at.P16 = (unsigned short*)(ws + L.s16);      // bf16(dLogits): beside P16, not over it
    at.seq = seq; at.n_seq = n_seq; at.strips = (G.t_max + 63) / 64; at.D = D;
    at.scale = opts->scale; at.ignore_self = opts->ignore_self; at.aperture = opts->aperture; at.drop = drop;
    SUMK_TRY(launch_attn_strip(true, at, stream));
    // dV = P^T dC and dK = dS^T Q: the two products that contract over the query rows of ALL strips, as ONE launch of 2 n_seq problems
    // (separately each was 800-960 tiles on 768 resident blocks -- one and a quarter rounds of latency-bound tiles, 25 us per launch)
    GemmLaunch g; g.precision = opts->precision;
    g.A = (const float*)(ws + L.p16); g.B[0] = (const float*)(ws + L.dctx16); g.src16 = 1; g.C16 = dQKV16; g.C = nullptr;
    g.probs = (GemmProb*)(tb + L.prob_dvk); g.nprob = 2 * n_seq; g.small_tile = G.cfg_pv; g.total_tiles = 2 * G.tiles_pv;
    SUMK_TRY(launch_gemm(GEMM_TN, EPI_NONE, g, stream));
  } else {
  {
    GemmLaunch g; g.precision = opts->precision;
    g.A = dCTX; g.B[0] = QKV; g.C = E2; g.probs = tabs + TB_DP * n_seq; g.nprob = n_seq; g.small_tile = G.cfg_s; g.total_tiles = G.tiles_s;
    if (b16) { g.A = (const float*)(ws + L.dctx16); g.B[0] = (const float*)(ws + L.qkv16); g.src16 = 1; }
    SUMK_TRY(launch_gemm(GEMM_NT, EPI_NONE, g, stream));
  }
  // 3': softmax (+dropout, +scale) backward, in place on E2
  // (b16: bf16(dLogits) goes where bf16(alpha) was -- its last reader, the dV product, is queued before this kernel)
  hipLaunchKernelGGL(vasnet_softmax_bwd_kernel<0>, dim3((R + 3) / 4), dim3(256), 0, stream, E, E2, seq, (const int32_t*)(tb + L.row_seq), n_seq, R,
                     opts->scale, drop, b16 ? (unsigned short*)(ws + L.p16) : nullptr, (const float*)E2, 0, (int64_t)0);
  // 2': dQ = dS K ; dK = dS^T Q
  {
    GemmLaunch g; g.precision = opts->precision;
    g.A = E2; g.B[0] = QKV; g.C = dQKV; g.probs = tabs + TB_DQ * n_seq; g.nprob = n_seq; g.small_tile = G.cfg_pv; g.total_tiles = G.tiles_pv;
    if (b16) { g.A = (const float*)(ws + L.p16); g.B[0] = (const float*)(ws + L.qkv16); g.src16 = 1; g.C16 = dQKV16; g.C = nullptr; }
    SUMK_TRY(launch_gemm(GEMM_NN, EPI_NONE, g, stream));
  }
  }
  if (!G.attn_fused) {
    GemmLaunch g; g.precision = opts->precision;
    g.A = E2; g.B[0] = QKV; g.C = dQKV; g.probs = tabs + TB_DK * n_seq; g.nprob = n_seq; g.small_tile = G.cfg_pv; g.total_tiles = G.tiles_pv;
    if (b16) { g.A = (const float*)(ws + L.p16); g.B[0] = (const float*)(ws + L.qkv16); g.src16 = 1; g.C16 = dQKV16; g.C = nullptr; }
    SUMK_TRY(launch_gemm(GEMM_TN, EPI_NONE, g, stream));
  }
  // 1': projection weights  d[Wq;Wk;Wv] += dQKV^T X
  {
    float* out[4] = {gr->Wq, gr->Wk, gr->Wv, nullptr};
    SUMK_TRY(gemm_tn_splitk_accum(b16 ? (const float*)dQKV16 : dQKV, 3 * D, b16 ? x16 : x, D, 3 * D, D, R, slab, L.slab_elems, pskw + SPLITK_PROBS, SPLITK_PROBS, out, D, D, 1.f,
                                  stream, opts->precision, b16, skw_ready ? SPLITK_TABLE_READY : SPLITK_BUILD_AND_RUN));
  }
  if (dx) {  // dX = dY0 (residual) + dQ Wq + dK Wk + dV Wv
    SUMK_HIP(hipMemcpyAsync(dx, dY0, (size_t)R * D * 4, hipMemcpyDeviceToDevice, stream));
    const float* Ws[3] = {w->Wq, w->Wk, w->Wv};
    for (int part = 0; part < 3; ++part) {
      GemmLaunch g; g.precision = opts->precision;
      g.A = dQKV + (size_t)part * D; g.B[0] = Ws[part]; g.C = dx; g.probs = prow + RP_DX; g.small_tile = G.st_d;
      g.total_tiles = gemm_tiles(R, D, G.st_d); g.xcd_M = R; g.xcd_N = D;
      if (b16) to_b16(g, dQKV16 + (size_t)part * D, Wqkv16 + (size_t)part * D * D, R, D, prow, RP_DX_W);
      SUMK_TRY(launch_gemm(GEMM_NN, EPI_ACCUM, g, stream));
    }
  }
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}
