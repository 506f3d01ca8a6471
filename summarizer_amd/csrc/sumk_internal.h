// Internal declarations shared by the .hip translation units of libsumk.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/sumk.h"

namespace sumk {

void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

#define SUMK_HIP(call)                                             \
  do {                                                             \
    hipError_t _e = (call);                                        \
    if (_e != hipSuccess) return ::sumk::hip_fail(_e, #call);      \
  } while (0)
#define SUMK_ARG(cond, ...)                                        \
  do {                                                             \
    if (!(cond)) { ::sumk::set_error(__VA_ARGS__); return SUMK_ERR_ARG; } \
  } while (0)
#define SUMK_TRY(call)                                             \
  do { int _r = (call); if (_r != SUMK_OK) return _r; } while (0)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Tile-shape / schedule switches left from the tuning work (none changes a result beyond the summation order of another tile shape):
// read ONLY by the diagnostic build (make DIAG=1 -> libsumk_diag.so, loaded through SUMK_LIB_PATH by scripts/probes); the product
// library takes the measured-best setting unconditionally.  What the product library does read are the A/B switches that have a
// test comparing both settings: SUMK_SK, SUMK_LEAN, SUMK_FUSED_HEAD, SUMK_FUSED_LN, SUMK_BF16_SRC, SUMK_B16_WIDE, SUMK_LSTM_PERSIST,
// SUMK_LSTM_LL, SUMK_LSTM_LL_BWD, SUMK_LSTM_M16.
#ifdef SUMK_DIAG
#define SUMK_TUNE_ENV(name) getenv(name)
#else
#define SUMK_TUNE_ENV(name) ((const char*)nullptr)
#endif

// ------------------------------------------------------------------------------------------- GEMM
// One sub-problem of a grouped launch.  Offsets are in elements from the launch's base pointers.
struct GemmProb {
  int64_t a_off, b_off, c_off, r_off;
  int32_t M, N, K;
  int32_t lda, ldb, ldc, ldr;
  int32_t tile_start;  // first tile id of this problem in the launch
  int32_t tiles_n;     // tiles along N
  int32_t pad_[7];     // zero, except in the tables of the small-batch (in-launch split-K) launches: indices SK_* below
};
static_assert(sizeof(GemmProb) == 96, "GemmProb layout");
// GemmProb::pad_ indices read by the SK instances of gemm_lean_kernel (gemm_lean.hip).  An SK table holds one entry per
// (problem, K slice): a_off / b_off / K describe the slice, M / N / c_off / r_off / ld* the whole problem, and
//   SK_N     number of K slices of this problem (0 or 1: not split),   SK_IDX  this entry's slice,
//   SK_TILE0 index of the problem's first tile in the launch's ticket / partial-tile arrays (the same for all its slices),
//   SK_BSEL  which of GemmLaunch::B[0..3] is this entry's B operand,   SK_CSEL  which of C / Csel[1..3] its output,
//   SK_PART0 index of the problem's first partial tile: tile j's slice s is partial SK_PART0 + j SK_N + s (problems of one launch may
//            have different slice counts, so this is not SK_TILE0 x SK_N).
enum { SK_N = 0, SK_IDX = 1, SK_TILE0 = 2, SK_BSEL = 3, SK_CSEL = 4, SK_PART0 = 5 };

enum GemmLayout { GEMM_NT = 0, GEMM_NN = 1, GEMM_TN = 2 };
enum GemmEpi {
  EPI_NONE = 0,       // C = alpha*acc
  EPI_RESIDUAL = 1,   // C = acc + R
  EPI_BIAS_RELU = 2,  // C = relu(acc + bias0[col])
  EPI_BIAS2 = 3,      // C = acc + bias0[col] + bias1[col]
  EPI_ACCUM = 4,      // C += alpha*acc   (gradient accumulation)
  EPI_BIAS_RESIDUAL = 5,  // C = acc + bias0[col] + R
  // v = relu(acc + bias0[col]) is NOT stored: per row and slot (N / 32 slots) the kernel writes {sum v, sum v^2, sum v * bias1[0][col] *
  // bias1[1][col], 0} to C viewed as float4[M][N / 32] -- the moments a LayerNorm + dot-product head over the row needs (VASNet
  // inference tail: k1 + ReLU + LayerNorm + k2 in one pass, vasnet.hip head_finalize_kernel).  128x128 NT tiles, N % 64 == 0.
  EPI_BIAS_RELU_HEAD = 6,
  // C = acc + R as EPI_RESIDUAL, AND per row and slot (N / 32 slots) {sum C, sum C^2} to `moments` viewed as float2[M][N / 32]: what
  // a LayerNorm over the rows of C needs, taken while the tile is still in registers.  128x128 NT tiles.
  EPI_RESIDUAL_MOMENTS = 7,
};

struct GemmLaunch {
  const float* A = nullptr;
  const float* B[4] = {nullptr, nullptr, nullptr, nullptr};  // B is split in groups of n_group columns of C
  float* C = nullptr;
  const float* R = nullptr;
  const float* bias0[4] = {nullptr, nullptr, nullptr, nullptr};
  const float* bias1[4] = {nullptr, nullptr, nullptr, nullptr};
  const GemmProb* probs = nullptr;  // device table, nprob entries
  int32_t nprob = 1;
  int32_t n_group = 0;  // columns of C per B pointer (0 -> single group)
  int32_t total_tiles = 0;
  float alpha = 1.f;
  int32_t small_tile = 0;  // tile config: 0 -> 128x128, 1 -> 64x64 (ragged per-video problems), 2 -> 128x64
  int32_t prof_tag = -1;
  int32_t precision = 0;   // 0: exact fp32 MFMA; 1: bf16x3 split on bf16 MFMA (NT layout only; other layouts stay fp32)
  int32_t xcd_M = 0, xcd_N = 0;  // single-problem launches: (M,N) so the kernel may use the XCD-aware tile map
  int32_t no_dma = 0;            // 1: keep the register-staged kernel (a K-contiguous operand whose K tail is not zero-padded)
  // training-mode dropout fused in the epilogue (EPI_BIAS_RELU: after the ReLU; EPI_BIAS_RESIDUAL: on acc+bias, before +R);
  // drop_thr == 0 disables it.  Element index of the mask = row * N + col.
  uint64_t drop_seed = 0; uint32_t drop_thr = 0, drop_site = 0; float drop_scale = 1.f;
  float* moments = nullptr;            // EPI_RESIDUAL_MOMENTS output
  // 1: the caller asserts a single problem with K % 32 == 0 whose operand byte offsets (row * ld * 4) stay below 2^31 -- NT launches on
  // 128x128 tiles then run the instances whose k-loop fetches with buffer loads + a scalar k offset (no VALU address arithmetic)
  int32_t lean = 0;
  // 1: A and B[0] point at bf16 arrays (offsets / leading dimensions of the problem table count bf16 elements); 128x128 tiles, fp32
  // accumulate and output (gemm_b16.hip).  The caller has checked gemm_b16_ok for every sub-problem.
  int32_t src16 = 0;
  // src16 only: 0 = 128x128 tiles; 192 / 256 = BM of the wide (BM x 256) tile -- `probs` (tiles_n, tile_start) and total_tiles must have
  // been built for that tile (gemm_tiles_wide; problem-table tile code 3 = 256 columns)
  int32_t wide16 = 0;
  void* C16 = nullptr;                 // EPI_NONE: also store bf16(C) here, same offsets / ldc (the operand of a later src16 launch); C may then be null
  int32_t group_remap = 0;             // grouped launch: deal tile ids so that one XCD walks a contiguous range (gemm_device.h decode_tile)
  // EPI_BIAS_RELU_HEAD on an A operand that is the INPUT of a LayerNorm whose gain was folded into B (B' = B diag(gamma)):
  // v = rstd_r (acc - mean_r c1[n]) + c2[n] + bias0[n] with ln_stats = float2[M] {mean, rstd}, c1[n] = sum_k gamma_k B[n][k],
  // c2[n] = sum_k beta_k B[n][k] -- the LayerNorm is applied to the PRODUCT, the normalised matrix never exists.
  const float* ln_stats = nullptr; const float* ln_c1 = nullptr; const float* ln_c2 = nullptr;
  // Small-batch launches (gemm_lean.hip, SK instances; 64x64 tiles, exact fp32, NT / NN / TN, epilogues NONE / RESIDUAL /
  // BIAS_RELU / ACCUM): sk = 1 selects them; `probs` is an SK table (above), total_tiles counts (tile, slice) blocks.  sk_part holds
  // 4096 floats per (tile, slice), sk_cnt one zeroed word per tile (the reducer leaves it zero again); Csel[1..3] are the outputs
  // entries with SK_CSEL = 1..3 write (C is output 0).
  int32_t sk = 0; float* sk_part = nullptr; unsigned* sk_cnt = nullptr; float* Csel[4] = {nullptr, nullptr, nullptr, nullptr};
};
// K slices for an SK launch of `tiles` 64x64 tiles contracting over K: enough (tile, slice) blocks for ~3-4 per CU, slices of whole
// 32-wide k-tiles, at least four k-tiles each, at most SK_MAX_SLICES.  Returns S and the slice length in *kchunk.
constexpr int SK_MAX_SLICES = 8;
inline int sk_slices(int tiles, int K, int* kchunk) {
  int S = tiles > 0 ? 1024 / tiles : 1;
  S = S < 1 ? 1 : S > SK_MAX_SLICES ? SK_MAX_SLICES : S;
  if (S > K / 128) S = K / 128 < 1 ? 1 : K / 128;
  int kc = ((K + S - 1) / S + 31) / 32 * 32;
  S = (K + kc - 1) / kc;
  *kchunk = kc;
  return S;
}

// number of tiles an (M,N) problem takes with the chosen tile size
inline int gemm_tile_m(int cfg) { return cfg == 1 ? 64 : 128; }
inline int gemm_tile_n(int cfg) { return cfg == 0 ? 128 : cfg == 3 ? 256 : 64; }   // (3: the wide bf16-source tiles, gemm_b16.hip)
inline int gemm_tile_dim(int cfg) { return gemm_tile_n(cfg); }   // tiles_n divisor of a config
inline int gemm_tiles(int M, int N, int cfg) {
  int tm = gemm_tile_m(cfg), tn = gemm_tile_n(cfg);
  return ((M + tm - 1) / tm) * ((N + tn - 1) / tn);
}
int launch_gemm(GemmLayout layout, GemmEpi epi, const GemmLaunch& g, hipStream_t stream);
// GemmLaunch::src16 eligibility of one problem: kc_a / kc_b = is that operand K-contiguous (A of NT / NN, B of NT).  A K-contiguous
// operand needs K % 64 == 0, an M/N-contiguous one rows of whole 16-byte chunks; every byte offset must stay below 2^31.
inline int gemm_b16_ok(int64_t M, int64_t N, int64_t K, int lda, int ldb, bool kc_a, bool kc_b) {
  const int64_t lim = (int64_t)1 << 31;
  if (lda % 8 || ldb % 8) return 0;
  if (kc_a ? (K % 64 != 0 || (M + 128) * lda * 2 >= lim) : (M % 8 != 0 || (K + 64) * lda * 2 >= lim)) return 0;
  if (kc_b ? (K % 64 != 0 || (N + 128) * ldb * 2 >= lim) : (N % 8 != 0 || (K + 64) * ldb * 2 >= lim)) return 0;
  return 1;
}
inline int gemm_tiles_wide(int M, int N, int bm) { return ((M + bm - 1) / bm) * ((N + 255) / 256); }
// BM of the wide tile for one (M, N) problem on 256 CUs (one block per CU): the BM whose rounds x rows-per-tile is smaller; 0 = the
// problem is too small to fill the chip with wide tiles (128x128 tiles instead)
inline int gemm_b16_wide_bm(int M, int N) {
  const int t192 = gemm_tiles_wide(M, N, 192), t256 = gemm_tiles_wide(M, N, 256);
  if (t256 < 96 || N % 256 != 0) return 0;
  return ((t192 + 255) / 256) * 192 <= ((t256 + 255) / 256) * 256 ? 192 : 256;
}
// dst[k] (bf16) = src[k] (fp32) for n <= 6 dense arrays of n_elems[k] (% 4 == 0) elements, one launch (gemm_b16.hip)
int cast_flat_b16(int n, const float* const src[], void* const dst[], const int64_t n_elems[], hipStream_t stream);
// GemmLaunch::lean for a single NT problem C(M,N) = A(M,K) B(N,K)^T: K a whole number of 32-wide k-tiles and every operand row
// within 2^31 bytes of its base (the buffer-load instances address with 32-bit offsets)
inline int gemm_lean_ok(int64_t M, int64_t N, int K, int lda, int ldb) {
  return (K % 32 == 0 && M * lda * 4 < ((int64_t)1 << 31) && N * ldb * 4 < ((int64_t)1 << 31)) ? 1 : 0;
}
// Fills ONE GemmProb (device) for a plain single problem; returns SUMK_OK.
int fill_single_prob(GemmProb* dev_prob, int M, int N, int K, int lda, int ldb, int ldc, int ldr, int small_tile,
                     hipStream_t stream);

// C(M,N) = A(K,M)^T B(K,N) contracted over a long K with deterministic split-K; out[g][rl*ldo+col] += alpha*C[row][col]
// for row = g*rows_per_out + rl.  probs_dev must hold probs_cap entries; slab holds slab_elems floats.
enum { SPLITK_BUILD_AND_RUN = 0, SPLITK_TABLE_READY = 1, SPLITK_TABLE_ONLY = 2 };   // `mode` of the two entries below (csrc/gemm_f32.hip)
int gemm_tn_splitk_accum(const float* A, int lda, const float* B, int ldb, int M, int N, int K, float* slab,
                         size_t slab_elems, GemmProb* probs_dev, int probs_cap, float* const out[4], int rows_per_out,
                         int ldo, float alpha, hipStream_t stream, int precision = 0, int src16 = 0, int mode = 0);   // src16: A and B are bf16 arrays (GemmLaunch::src16)
// np (<= 4) same-shaped products C_p = A_p^T B_p in ONE split-K launch + ONE reduce: out[p] += alpha * C_p (rows_per_out == M when np > 1).
// Every A_p (B_p) must lie in the same allocation as A_0 (B_0): the kernel addresses them as offsets from it.
int gemm_tn_splitk_accum_multi(int np, const float* const A[], const float* const B[], int lda, int ldb, int M, int N, int K, float* slab,
                               size_t slab_elems, GemmProb* probs_dev, int probs_cap, float* const out[4], int rows_per_out,
                               int ldo, float alpha, hipStream_t stream, int precision = 0, int src16 = 0, int mode = 0);
// out[c] += sum_r X[r*ld + c]; partial must hold max_chunks*N floats.
int colsum_accum(const float* X, int ld, int R, int N, float* partial, int max_chunks, float* out, hipStream_t stream);
// out[c] += sum_p partial[p*stride + c]
int partial_reduce_accum(const float* partial, int n_part, int stride, int N, float* out, hipStream_t stream);
// ONE launch for several column ranges of the same slots: segs[k].out[c] += sum_p partial[p*stride + segs[k].off + c], c < segs[k].n.
// Every off must be a multiple of 4 (float4 columns); nseg <= 6; null outputs are skipped.
struct ReduceSeg { int32_t off, n; float* out; };
int partial_reduce_multi(const float* partial, int n_part, int stride, const ReduceSeg* segs, int nseg, hipStream_t stream);
// segs[k].out[c] += sum_r X[r*ld + segs[k].off + c] for c < segs[k].n: one pass over X(R, N), any number of (overlapping) column
// ranges; partial must hold max_chunks*N floats.  N % 4 == 0 is required (partial_reduce_multi's stride).
int colsum_multi(const float* X, int ld, int R, int N, float* partial, int max_chunks, const ReduceSeg* segs, int nseg,
                 hipStream_t stream);

// Dropout keep-mask: a pure function of (seed, site, element index) so backward regenerates it and the numpy oracle
// can reproduce it bit for bit (tests/golden/recipes.py: dropout_keep).  splitmix64 finaliser.
__host__ __device__ inline bool dropout_keep(uint64_t seed, uint32_t site, uint64_t idx, uint32_t thr) {
  uint64_t z = (seed ^ ((uint64_t)(site + 1) * 0x9E3779B97F4A7C15ull)) + idx * 0xD1342543DE82EF95ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (uint32_t)(z >> 32) >= thr;
}
inline uint32_t dropout_threshold(float p) {
  double t = (double)p * 4294967296.0;
  return t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
}
struct Drop {  // dropout descriptor of one call; thr == 0 means "no dropout"
  uint64_t seed; uint32_t thr; float scale;
  // non-null: the kernels add this device word to `seed` when they run (drop_resolve) -- a step captured into a HIP graph then draws
  // fresh masks on every replay, the caller bumping the word between replays (sumk_vasnet_opts::seed_dev)
  const uint64_t* seed_dev;
};
__device__ __forceinline__ Drop drop_resolve(Drop d) {
  if (d.thr && d.seed_dev) d.seed += *d.seed_dev;
  d.seed_dev = nullptr;
  return d;
}
inline Drop make_drop(float p, uint64_t seed) {
  Drop d; d.seed = seed; d.thr = 0; d.scale = 1.f; d.seed_dev = nullptr;
  if (p > 0.f) { d.thr = dropout_threshold(p); d.scale = 1.0f / (1.0f - p); }
  return d;
}
__host__ __device__ inline float drop_apply(const Drop& d, uint32_t site, uint64_t idx, float v) {
  return dropout_keep(d.seed, site, idx, d.thr) ? v * d.scale : 0.f;
}

// One video of a packed batch as the attention kernels see it (built on device by vasnet_setup_kernel, csrc/vasnet.hip).
struct SeqInfo {
  int64_t eoff;  // element offset of this video's (T x ldE) logits block in E
  int32_t row0, T, ldE, pad_;
  int64_t e16off;  // element offset of its (T x ld16) bf16 attention block, ld16 = T rounded up to 64 (bf16-source training step)
};

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

// Fused attention strips of the bf16-source training step (attn_b16.hip): one workgroup per (video, 64 query rows), T <= 320 keys.
//   forward : S = Q K^T (bf16 operands, fp32 accumulate) -> alpha = softmax(mask(S * scale)) (fp32, written to E) ->
//             P = bf16(dropout(alpha)) (written to P16: the dV product reads it) -> CTX = P V (bf16, written to O16)
//   backward: dP = dCTX V^T -> dS = scale * alpha * (dropout'(dP) - sum_j dropout'(dP)_j alpha_j) (alpha re-read from E) ->
//             bf16(dS) (written over P16: the dK product reads it) -> dQ = dS K (bf16, written to O16)
// The same rounding points as the separate launches (GEMM -> softmax kernel -> GEMM) they replace.
struct AttnStripArgs {
  const unsigned short* A16; int32_t lda;    // query-side rows of GEMM-1: Q (forward) / dCTX (backward)
  const unsigned short* B16; int32_t ldb;    // key-side rows of GEMM-1: K (forward) / V (backward)
  const unsigned short* C16; int32_t ldc;    // [key][column] operand of GEMM-2: V (forward) / K (backward)
  unsigned short* O16; int32_t ldo;          // output rows: CTX (forward) / dQ (backward)
  float* E;                                  // alpha, per-video (T x ldE) blocks (SeqInfo::eoff)
  unsigned short* P16;                       // per-video (T x ld16) bf16 blocks (SeqInfo::e16off)
  const SeqInfo* seq; int32_t n_seq, strips, D;
  float scale; int32_t ignore_self, aperture; Drop drop;
};
bool attn_strip_ok(int t_max, int D, int64_t rows, int ld_max);
int launch_attn_strip(bool backward, const AttnStripArgs& a, hipStream_t stream);

// Optional pre-reduction of a row kernel's input (small-batch path: the producing GEMM was cut into K slices that each stored their
// own slab with plain stores -- no in-launch reduction, no tickets -- and the row kernel that consumes the matrix anyway adds them):
//   value(r, c) = sum_{s < n} X[s * stride + r * ld + c]  (slab order)  [+ add[r * ld + c]]  [+ bias[c]]  [relu]
// n <= 1 and no add / bias: X is read as is.  `store` (same shape as one slab) receives the reduced value -- what a later kernel
// (the backward pass) reads as the matrix itself.
struct SlabIn {
  int32_t n = 0; int32_t relu = 0;
  int64_t stride = 0;
  const float* add = nullptr; const float* bias = nullptr;
  float* store = nullptr;
};

// ------------------------------------------------------------------------------------------- plane-aware wide GEMM (gemm_pw.hip)
// "KB planes": an fp32 matrix X (rows x K, K % 16 == 0) held as NP = 2 (hi + lo: bf16x3) or 3 (x1 + x2 + x3 = x EXACTLY: bf16x6)
// bf16 planes, k-blocked so that the GEMM's LDS-DMA reads and the producers' stores are whole contiguous runs:
//   byte offset of element (row, k) of plane p = (((k / 16) * NP + p) * 2 + (k / 8) % 2) * rp16 + row * 16 + (k % 8) * 2,
//   rp16 = 16 * rows_pitch (rows rounded up to 64): [k16 block][plane][k half][row][8 bf16].
// One (k16 block, plane, half) sub-array is `rows` 16-byte chunks -- exactly the lane fragment of v_mfma_f32_32x32x16_bf16 -- so 64
// consecutive rows of it are one 1-KiB LDS-DMA instruction, and the LDS image [plane][half][row] is read by ds_read_b128 without bank
// conflicts and without padding.  Planes are written ONCE: x per dataset, weights per weight change, activations by the epilogue of
// the kernel that produced them -- no fp32 -> bf16 split inside any k-loop (the in-loop split of gemm_regstage.h kept the matrix
// pipe at 0.3-0.5 busy).  pw_planes_bytes includes the slack a row tile may read past the last sub-array.
inline int64_t pw_rows_pitch(int64_t rows) { return (rows + 63) / 64 * 64; }
inline size_t pw_planes_bytes(int64_t rows, int K, int np) { return (size_t)pw_rows_pitch(rows) * K * np * 2 + 8192; }
int split_planes(const float* src, int64_t rows, int K, int ld, int np, void* planes, hipStream_t stream);
// the same into rows [row0, row0 + rows) of a plane array built for `total_rows` rows (row0 % 64 == 0; pad rows are written only behind the
// LAST row of the array): stacking several matrices into one operand ([Wq; Wk; Wv])
int split_planes_at(const float* src, int64_t rows, int K, int ld, int np, void* planes, int64_t row0, int64_t total_rows, hipStream_t stream);
// round 6 (long videos): planes of the TRANSPOSE of an fp32 (T x Dm) matrix (rows = its columns, k = t, zeros for t in [T, Kp)); row softmax
// of one video's raw logits (T x T valid, masks and scale of vasnet.py:118-128) straight to the planes of alpha (k = key, zeros in [T, Kp))
int split_planes_pitched(const float* src, int64_t rows, int K, int ld, int np, void* planes, int64_t pitch_rows, hipStream_t stream);
int split_planes_t(const float* src, int T, int Dm, int ld, int np, void* planes, int Kp, hipStream_t stream);
int softmax_planes(const float* E, int T, int64_t ldE, int np, void* planes, int Kp, float scale, int ignore_self, int aperture, float* stats /* 2 T floats */, hipStream_t stream);
enum PwEpi {
  PW_F32 = 0,             // C (fp32, row-major, ldc) = product [+ bias[n]] [+ R[m][n]] [relu]
  PW_PLANES = 1,          // O (KB planes of the (M, N) result: the K-contiguous operand of a later product over N) = product [+ bias[n]] [relu]
  PW_RES_MOM_PLANES = 2,  // v = product + R; moments float2[M][N / 64] {sum v, sum v^2} per 64-column slot; O = planes of v
  PW_HEAD = 3,            // v = relu(rstd_m (product - mean_m c1[n]) + bias[n]) is NOT stored: head_part float4[M][N / 64] {sum v, sum v^2, sum v gw[n], 0}
  PW_RES_F32 = 4,         // C (fp32) = product + R [+ bias[n]] in the transposed orientation (lane = row, 16-byte pieces): R is preloaded into the
                          // accumulators, the epilogue is float4 stores alone (the residual projections of the Transformer scorer)
};
struct PwLaunch {
  const void* A = nullptr; const void* B = nullptr;    // KB planes of A (M x K) and B (N x K): C = A B^T
  int64_t a_rows = 0, b_rows = 0;                      // rows the plane arrays were built for (their pitch = pw_rows_pitch)
  int32_t M = 0, N = 0, K = 0, np = 3;
  float* C = nullptr; int32_t ldc = 0;
  void* O = nullptr; int64_t o_rows = 0;
  int64_t o_store_rows = 0;                             // PW_PLANES: rows [M, o_store_rows) are stored too (0: none) -- zeros when A's pad rows are zero: readers
                                                        // that run a few rows past the last video (attn_pw.hip) then multiply finite data
  const float* R = nullptr; int32_t ldr = 0;
  float* moments = nullptr;
  const float* bias = nullptr; const float* gw = nullptr; const float* ln_c1 = nullptr; const float* ln_stats = nullptr; float* head_part = nullptr;
  int32_t relu = 0;                                     // PW_F32 / PW_PLANES: ReLU after the bias (the feed-forward layers of the Transformer scorer)
  int32_t prof_tag = -1;
  int32_t variant = 0;                                  // schedule variant (probes; 0 = the product's; 32 = two planes on the 32x32x16 kernel instead of gemm_pw16.hip)
};
// M >= 1, N % 256 == 0, K % 32 == 0, K >= 128, plane arrays below 2^31 bytes
inline int pw_ok(int64_t M, int64_t N, int64_t K, int64_t a_rows, int64_t b_rows, int np) {
  const int64_t lim = ((int64_t)1 << 31) - 65536;
  return (M >= 1 && N % 256 == 0 && K % 32 == 0 && K >= 128 && (np == 2 || np == 3) &&
          (int64_t)pw_planes_bytes(a_rows, (int)K, np) < lim && (int64_t)pw_planes_bytes(b_rows, (int)K, np) < lim) ? 1 : 0;
}
int launch_gemm_pw(PwEpi epi, const PwLaunch& g, hipStream_t stream);

// Per-video attention on planes (attn_pw.hip; T <= 320): logits + softmax -> alpha planes (rows = packed frames, k = key index inside the
// video, written up to T rounded up to 32) [+ fp32 alpha in E], then context = alpha . V -> planes of the (rows x D) context matrix.
// alpha / context planes use the row pitch of the [Q | K | V] planes (pw_rows_pitch(rows)).
bool attn_pw_ok(int t_max, int D, int64_t rows, int np);
inline size_t pw_alpha_bytes(int64_t rows, int t_max, int np) { return (size_t)pw_rows_pitch(rows) * ((t_max + 31) / 32 * 32) * np * 2 + 8192; }
// heads > 1 (Transformer scorer; heads of 128 columns, E = nullptr): head h contracts columns [128 h, 128 h + 128) of Q / K, its alpha planes start at
// alpha_planes + h * align_up(pw_alpha_bytes(rows, t_max, np), 256), its context lands in the same columns of the context planes
bool attn_pw_heads_ok(int t_max, int D, int heads, int64_t rows, int np);
int launch_attn_pw_logits(int np, const void* qkv_planes, int64_t rows, int D, float* E, void* alpha_planes, const SeqInfo* seq, int n_seq,
                          int t_max, float scale, int ignore_self, int aperture, hipStream_t stream, int heads = 1);
// R (single head; the folded VASNet path where V = x Wvo^T): context + R[row][column] leaves as planes, with {sum, sum of squares} per (row, 32-column slot) in moments
int launch_attn_pw_context(int np, const void* qkv_planes, int64_t rows, int D, const void* alpha_planes, void* ctx_planes, const SeqInfo* seq,
                           int n_seq, int t_max, hipStream_t stream, int heads = 1, const float* R = nullptr, int ldr = 0, float* moments = nullptr);

// ------------------------------------------------------------------------------------------- shared row kernels (vasnet.hip)
// Y = LayerNorm(X) * g + b over D (one wave per row); optional (mean, rstd) per row into stats.
int launch_layernorm(const float* X, float* Y, const float* g, const float* b, int n_rows, int D, float eps, float* stats,
                     hipStream_t stream);
// scores[r] = sigmoid(LayerNorm(Z[r]) . w2 + b2)
int launch_ln_head(const float* Z, const float* g, const float* b, const float* w2, const float* b2, float* scores,
                   int n_rows, int D, float eps, hipStream_t stream);
// training-mode variants: dropout `site` applied to X on load; stats (mean, rstd per row) saved for the backward
int launch_layernorm_drop(const float* X, float* Y, const float* g, const float* b, int n_rows, int D, float eps, float* stats,
                          Drop drop, uint32_t site, hipStream_t stream);
int launch_ln_head_drop(const float* Z, const float* g, const float* b, const float* w2, const float* b2, float* scores,
                        int n_rows, int D, float eps, float* stats, Drop drop, uint32_t site, hipStream_t stream);
// Backward of Y = LN(drop(X)) * g + b: dX from dY; per-BLOCK partial sums (the block's 4 waves are combined through LDS in a
// fixed order) go to `part`: [n_slots][ln_slot_floats(D)] floats = [dgamma D][dbeta D][dw2 D][column sums of dX D][db2, pad x3],
// n_slots <= LNB_MAX_WAVES / 4 returned in *n_waves; ln_bwd_reduce adds the slots into the gradients in ONE launch.
constexpr int LNB_MAX_WAVES = 3072;   // 768 blocks = 3 per CU (the kernels hold 124-164 VGPRs: 3-4 waves per SIMD fit)
inline int ln_slot_floats(int D) { return 4 * D + 4; }
// dgamma / dbeta always; dw2, db2, dcol (= column sums of the dX the kernel wrote: the bias gradient of the layer below) when
// non-null (head variant only).
int ln_bwd_reduce(const float* part, int n_slots, int D, float* dgamma, float* dbeta, float* dw2, float* db2, float* dcol,
                  hipStream_t stream);
int launch_ln_bwd_rows(int D, int R, const float* X, const float* stats, const float* g, const float* b, const float* dY,
                       float* dX, float* part, Drop drop, uint32_t site, int* n_waves, hipStream_t stream);
// Backward of scores = sigmoid(LN(drop(Z)) . w2 + b2) with Z = post-ReLU activations: dZ (ReLU + dropout masks applied),
// partials of (dgamma, dbeta, dw2), the column sums of dZ and db2 (slot layout above).
int launch_ln_head_bwd(int D, int R, const float* Z, const float* stats, const float* g, const float* b, const float* w2,
                       const float* scores, const float* dscores, float* dZ, float* part, Drop drop, uint32_t site,
                       int* n_waves, hipStream_t stream);
// x[r,:] += table[pos_rows[r],:]  in place
int launch_add_pos(float* x, const float* table, const int32_t* pos_rows, int n_rows, int D, hipStream_t stream);

// ------------------------------------------------------------------------------------------- profiling
void prof_begin(int tag, hipStream_t s);
void prof_end(int tag, hipStream_t s);

}  // namespace sumk
