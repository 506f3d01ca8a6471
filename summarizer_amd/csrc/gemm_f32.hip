// fp32 MFMA GEMM for gfx950 (v_mfma_f32_32x32x2_f32: exact f32 in / f32 accumulate, 64 FLOP/clk/SIMD).
//
// One kernel template serves every dense contraction on the scoring path:
//   NT  C = A(M,K) . B(N,K)^T   QKV / out-proj / k1 projections (vasnet.py:114-116,132,140), Q.K^T (vasnet.py:118),
//                               LSTM input projection (dsn.py:45), dgrad of the attention products
//   NN  C = A(M,K) . B(K,N)     alpha.V (vasnet.py:131), dX = dY.W
//   TN  C = A(K,M)^T . B(K,N)   weight gradients, dV = P^T dC, dK = dS^T Q
// A launch is GROUPED: a device table of GemmProb sub-problems (one per video for the ragged per-sequence
// products, a single entry for the packed row-wise projections); blockIdx.x -> (problem, m-tile, n-tile).
//
// Tiling: 256 threads = 4 waves (2x2); block tile BTxBT (128 or 64), BK = 32; each wave owns a (BT/2)^2 tile =
// TMxTN MFMA tiles of 32x32 (16 accumulator VGPRs each).  Operands are staged global -> registers (16-B loads,
// issued one k-tile ahead so HBM/L2 latency hides under the 64-cycle MFMAs) -> LDS.
//   K-contiguous operand ("KC": A of NT/NN, B of NT): LDS image [row][BK+4]; the +4 pad makes the wave's
//     ds_read_b128 (16 rows x 16 B per lane group) conflict free.  A lane reads 4 consecutive k at once.
//   M/N-contiguous operand ("MC": B of NN, A and B of TN): LDS image [k][BT]; a lane reads [k][i] with
//     ds_read_b32, consecutive lanes consecutive i.
// K ORDER: the 32x32x2 MFMA takes k from the lane half h (A[i][k=h], B[k=h][j]).  Within an 8-wide k chunk we
//   feed step j (0..3) with k = 4h + j for BOTH operands, so a KC lane's float4 supplies four MFMA steps.
//   The sum over k is therefore re-associated relative to a sequential loop (fp32, ~1e-7 relative).
#include "gemm_regstage.h"

namespace sumk {

// one 64 KB stamp buffer per process, allocated on first use under SUMK_GEMM_DBG=2
static unsigned long long* gemm_stamp_buffer() {
  static unsigned long long* buf = nullptr;
  if (!buf && hipMalloc(&buf, 2048 * 12 * sizeof(unsigned long long)) != hipSuccess) buf = nullptr;
  return buf;
}

int launch_gemm(GemmLayout layout, GemmEpi epi, const GemmLaunch& g, hipStream_t stream) {
  SUMK_ARG(g.A && g.B[0] && (g.C || (g.C16 && epi == EPI_NONE)) && g.probs, "gemm: null operand");
  if (g.total_tiles <= 0) return SUMK_OK;
  GemmKArgs ka;
  ka.A = g.A;
  for (int i = 0; i < 4; ++i) { ka.B[i] = g.B[i]; ka.bias0[i] = g.bias0[i]; ka.bias1[i] = g.bias1[i]; }
  ka.C = g.C; ka.R = g.R; ka.probs = g.probs; ka.nprob = g.nprob; ka.n_group = g.n_group; ka.alpha = g.alpha;
  ka.total_tiles = g.total_tiles; ka.xcd_tiles_m = 0;
  // SUMK_GROUP_REMAP: 0 never; 1 (default) split-K slices in every arithmetic + the per-video products of the bf16-plane modes, which
  // are bound by operand bytes (bf16 training step 1.35 -> 1.21 ms, bf16x3 / bf16x6 ~1 %); the exact-fp32 per-video products did
  // not gain (alpha.V 85 -> 89 us) and keep the plain order; 2 every grouped launch
  static const int group_remap = SUMK_TUNE_ENV("SUMK_GROUP_REMAP") ? atoi(SUMK_TUNE_ENV("SUMK_GROUP_REMAP")) : 1;
  // (per-video products only while a video is a handful of tiles: at T = 10 000 a sub-problem is thousands of tiles, a contiguous
  //  range is a band of one video and the plain order was 6 % faster)
  const bool small_groups = (int64_t)g.total_tiles <= (int64_t)256 * g.nprob;
  ka.group_remap = (g.nprob > 1 && (group_remap == 2 || (group_remap == 1 && (g.group_remap || (g.precision != SUMK_PRECISION_FP32 && small_groups))))) ? 1 : 0;
#ifdef SUMK_DIAG   // `make DIAG=1` only: SUMK_GEMM_DBG=1 skips the epilogue stores (wrong results by design), =2 in-kernel cycle stamps
  static const int dbg = getenv("SUMK_GEMM_DBG") ? atoi(getenv("SUMK_GEMM_DBG")) : 0;
#else
  constexpr int dbg = 0;
#endif
  ka.dbg = dbg; ka.dbg_buf = nullptr;
  if (dbg & 2) {   // diagnostic only (never on a product path); SUMK_STAMP_TAG=<prof tag>: stamp only that GEMM of a forward pass
    static const int only_tag = getenv("SUMK_STAMP_TAG") ? atoi(getenv("SUMK_STAMP_TAG")) : -1;
    if (only_tag < 0 || only_tag == g.prof_tag) ka.dbg_buf = gemm_stamp_buffer(); else ka.dbg &= ~2;
  }
  ka.drop.seed = g.drop_seed; ka.drop.thr = g.drop_thr; ka.drop.scale = g.drop_scale; ka.drop.seed_dev = nullptr; ka.drop_site = g.drop_site;
  static const bool lean128_on = !(SUMK_TUNE_ENV("SUMK_LEAN128") && SUMK_TUNE_ENV("SUMK_LEAN128")[0] == '0');
  ka.lean = (lean128_on && g.lean && g.nprob == 1 && layout == GEMM_NT && g.small_tile == 0 && g.precision == SUMK_PRECISION_FP32 &&
             (g.n_group == 0 || g.n_group % 128 == 0)) ? 1 : 0;
  ka.C16 = (unsigned short*)g.C16;
  SUMK_ARG(!g.C16 || epi == EPI_NONE, "gemm: the bf16 copy of C goes with the plain epilogue");
  SUMK_ARG(!g.src16 || (g.small_tile == 0 && g.n_group == 0), "gemm: bf16-source launches use 128x128 tiles and one B operand");
  ka.moments = g.moments; ka.ln_stats = g.ln_stats; ka.ln_c1 = g.ln_c1; ka.ln_c2 = g.ln_c2;
  SUMK_ARG(epi != EPI_RESIDUAL_MOMENTS || g.moments, "gemm: the moments epilogue needs an output buffer");
  SUMK_ARG(!g.ln_stats || (epi == EPI_BIAS_RELU_HEAD && g.ln_c1 && g.ln_c2), "gemm: ln_stats goes with the head epilogue and c1 / c2");
  if (g.prof_tag >= 0) prof_begin(g.prof_tag, stream);
  prof_begin(SUMK_PROF_GEMM_ALL, stream);
  static const bool xcd_map = !(SUMK_TUNE_ENV("SUMK_XCD_MAP") && SUMK_TUNE_ENV("SUMK_XCD_MAP")[0] == '0');
  if (xcd_map && g.nprob == 1 && g.xcd_M > 0) {
    const int bm = (g.src16 && g.wide16) ? g.wide16 : gemm_tile_m(g.small_tile), bn = (g.src16 && g.wide16) ? 256 : gemm_tile_n(g.small_tile);
    const int tm = (g.xcd_M + bm - 1) / bm, tn = (g.xcd_N + bn - 1) / bn;
    if (tn % 4 == 0 && tm >= 16) { ka.xcd_tiles_m = tm; ka.total_tiles = 8 * ((tm + 1) / 2) * (tn / 4); }
  }
  int rc;
  ka.sk_epi = (int32_t)epi; ka.sk_part = g.sk_part; ka.sk_cnt = g.sk_cnt;
  for (int i = 0; i < 4; ++i) ka.Csel[i] = g.Csel[i];
  if (g.sk) {   // small-batch launch: in-launch split-K on 64x64 exact-fp32 tiles (gemm_lean.hip, SK instances)
    SUMK_ARG(g.precision == SUMK_PRECISION_FP32 && g.small_tile == 1 && g.n_group == 0 && !g.src16 && !g.C16 && g.C,
             "gemm: an SK launch is exact fp32 on 64x64 tiles with one B group per table entry");
    SUMK_ARG(epi == EPI_NONE || epi == EPI_RESIDUAL || epi == EPI_BIAS_RELU || epi == EPI_ACCUM, "gemm: epilogue %d has no SK form", (int)epi);
    SUMK_ARG(g.drop_thr == 0, "gemm: SK launches have no epilogue dropout");
    ka.group_remap = 0; ka.xcd_tiles_m = 0;
    rc = launch_gemm_lean(layout, ka, ka.total_tiles, stream, 1);
    prof_end(SUMK_PROF_GEMM_ALL, stream);
    if (g.prof_tag >= 0) prof_end(g.prof_tag, stream);
    if (rc != SUMK_OK) return rc;
    SUMK_HIP(hipGetLastError());
    return SUMK_OK;
  }
  // 64x64 tiles, plain epilogue, K-contiguous A, exact fp32: the lean kernel (gemm_lean.hip; SUMK_LEAN=0 keeps the generic one)
  static const bool lean_on = !(getenv("SUMK_LEAN") && getenv("SUMK_LEAN")[0] == '0');
  if (lean_on && g.precision == SUMK_PRECISION_FP32 && g.small_tile == 1 && epi == EPI_NONE && (layout == GEMM_NT || layout == GEMM_NN) &&
      g.n_group == 0 && ka.xcd_tiles_m == 0 && !g.C16 && !g.src16) {
    rc = launch_gemm_lean(layout, ka, ka.total_tiles, stream);
    prof_end(SUMK_PROF_GEMM_ALL, stream);
    if (g.prof_tag >= 0) prof_end(g.prof_tag, stream);
    if (rc != SUMK_OK) return rc;
    SUMK_HIP(hipGetLastError());
    return SUMK_OK;
  }
  // BK = 64 for the 64x64 tile measured no better than BK = 32 on S-TVSum (8.64 vs 8.68 M frames/s): kept selectable
  static const bool bk64 = SUMK_TUNE_ENV("SUMK_BK64") && SUMK_TUNE_ENV("SUMK_BK64")[0] == '1';
  if (g.src16) {                              // bf16 operands in HBM: gemm_b16.hip
    SUMK_ARG(g.wide16 == 0 || g.wide16 == 192 || g.wide16 == 256, "gemm: wide16 must be 0, 192 or 256");
    rc = launch_gemm_b16(layout, epi, ka, ka.total_tiles, g.wide16, stream);
  } else
  if (g.precision != SUMK_PRECISION_FP32) {   // bf16-plane arithmetics: instantiated in gemm_split.hip
    rc = launch_gemm_split(g.precision, layout, epi, ka, ka.total_tiles, g.small_tile, stream);
  } else
  if (g.small_tile == 1) rc = bk64 ? launch_layout<64, 64, 64>(layout, epi, ka, ka.total_tiles, stream)
                                   : launch_layout<64, 64, 32>(layout, epi, ka, ka.total_tiles, stream);
  else if (g.small_tile == 2) rc = launch_layout<128, 64, 32>(layout, epi, ka, ka.total_tiles, stream);
  else rc = launch_layout<128, 128, 32>(layout, epi, ka, ka.total_tiles, stream);
  prof_end(SUMK_PROF_GEMM_ALL, stream);
  if (g.prof_tag >= 0) prof_end(g.prof_tag, stream);
  if (rc != SUMK_OK) return rc;
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}

__global__ void fill_single_prob_kernel(GemmProb* p, int M, int N, int K, int lda, int ldb, int ldc, int ldr, int bt) {
  GemmProb q;
  q.a_off = q.b_off = q.c_off = q.r_off = 0;
  q.M = M; q.N = N; q.K = K; q.lda = lda; q.ldb = ldb; q.ldc = ldc; q.ldr = ldr;
  q.tile_start = 0; q.tiles_n = (N + bt - 1) / bt;
  for (int i = 0; i < 7; ++i) q.pad_[i] = 0;
  *p = q;
}

int fill_single_prob(GemmProb* dev_prob, int M, int N, int K, int lda, int ldb, int ldc, int ldr, int small_tile,
                     hipStream_t stream) {
  hipLaunchKernelGGL(fill_single_prob_kernel, dim3(1), dim3(1), 0, stream, dev_prob, M, N, K, lda, ldb, ldc, ldr,
                     gemm_tile_dim(small_tile));
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}

// ------------------------------------------------------------------------------------------- split-K (TN)
// Weight gradients contract over ALL frames (K = n_rows ~ 1e4) into a small (M,N): one tile grid would leave most
// CUs idle, so K is cut into S slices, each slice writes its own fp32 slab, and a second kernel sums the slabs in a
// fixed order (deterministic; no float atomics) and ACCUMULATES alpha*sum into up to four row-group outputs.
// np same-shaped products share one launch: entry p * S + s = K slice s of product p, whose operands start rel_a[p] / rel_b[p] elements
// after the launch's A / B pointers; slab layout [S][np * M][N] (slice-major), so the reduce kernel sees ONE (np * M) x N problem
struct SplitKSetup { GemmProb* p; int64_t rel_a[4], rel_b[4]; int32_t np, S, M, N, K, kchunk, lda, ldb, bt; };
__global__ void splitk_setup_kernel(SplitKSetup a) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= a.np * a.S) return;
  const int p = e / a.S, s = e - p * a.S;
  GemmProb q;
  const int k0 = s * a.kchunk;
  const int64_t ra = p == 0 ? a.rel_a[0] : p == 1 ? a.rel_a[1] : p == 2 ? a.rel_a[2] : a.rel_a[3];
  const int64_t rb = p == 0 ? a.rel_b[0] : p == 1 ? a.rel_b[1] : p == 2 ? a.rel_b[2] : a.rel_b[3];
  q.a_off = ra + (int64_t)k0 * a.lda; q.b_off = rb + (int64_t)k0 * a.ldb;
  q.c_off = ((int64_t)s * a.np + p) * a.M * a.N; q.r_off = 0;
  q.M = a.M; q.N = a.N; q.K = min(a.kchunk, a.K - k0); q.lda = a.lda; q.ldb = a.ldb; q.ldc = a.N; q.ldr = 0;
  const int tn = (a.N + a.bt - 1) / a.bt, tm = (a.M + a.bt - 1) / a.bt;
  q.tile_start = e * tm * tn; q.tiles_n = tn;
  for (int i = 0; i < 7; ++i) q.pad_[i] = 0;
  a.p[e] = q;
}

struct SlabReduceArgs { const float* slab; float* out[4]; int32_t S, M, N, rows_per_out, ldo; float alpha; };
__global__ void slab_reduce_kernel(SlabReduceArgs a) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t mn = (int64_t)a.M * a.N;
  if (idx >= mn) return;
  float v = 0.f;
  for (int s = 0; s < a.S; ++s) v += a.slab[(int64_t)s * mn + idx];
  int row = (int)(idx / a.N), col = (int)(idx % a.N);
  int g = row / a.rows_per_out, rl = row - g * a.rows_per_out;
  float* o = g == 0 ? a.out[0] : g == 1 ? a.out[1] : g == 2 ? a.out[2] : a.out[3];
  o[(int64_t)rl * a.ldo + col] += a.alpha * v;
}
// N % 4 == 0 and ldo % 4 == 0 (every weight gradient of the models): four columns per thread, the S slab loads of a thread in
// flight together -- the scalar form moved 24 MB in 19.5 us.  Same summation order per element (slab 0, 1, ...): same bits.
__global__ __launch_bounds__(256) void slab_reduce4_kernel(SlabReduceArgs a) {
  const int64_t i4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t mn = (int64_t)a.M * a.N;
  const int64_t idx = 4 * i4;
  if (idx >= mn) return;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  int s = 0;
  for (; s + 4 <= a.S; s += 4) {
    float4 t[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) t[u] = *reinterpret_cast<const float4*>(a.slab + (int64_t)(s + u) * mn + idx);
#pragma unroll
    for (int u = 0; u < 4; ++u) { v.x += t[u].x; v.y += t[u].y; v.z += t[u].z; v.w += t[u].w; }
  }
  for (; s < a.S; ++s) {
    const float4 t = *reinterpret_cast<const float4*>(a.slab + (int64_t)s * mn + idx);
    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
  }
  const int row = (int)(idx / a.N), col = (int)(idx % a.N);
  const int g = row / a.rows_per_out, rl = row - g * a.rows_per_out;
  float* o = (g == 0 ? a.out[0] : g == 1 ? a.out[1] : g == 2 ? a.out[2] : a.out[3]) + (int64_t)rl * a.ldo + col;
  float4 c = *reinterpret_cast<float4*>(o);
  c.x += a.alpha * v.x; c.y += a.alpha * v.y; c.z += a.alpha * v.z; c.w += a.alpha * v.w;
  *reinterpret_cast<float4*>(o) = c;
}

// mode 0 (default): build the K-slice table, multiply, reduce.  SPLITK_TABLE_READY: the table at probs_dev was built before for the same
// shapes, capacities and operand distances (it depends on nothing else) -- no setup launch.  SPLITK_TABLE_ONLY: build the table, nothing else
// (A / B are used for their DISTANCES only and may be offsets dressed as pointers).
int gemm_tn_splitk_accum_multi(int np, const float* const A[], const float* const B[], int lda, int ldb, int M, int N, int K, float* slab,
                               size_t slab_elems, GemmProb* probs_dev, int probs_cap, float* const out[4], int rows_per_out,
                               int ldo, float alpha, hipStream_t stream, int precision, int src16, int mode) {
  SUMK_ARG(np >= 1 && np <= 4 && M > 0 && N > 0 && K > 0, "splitk: bad shape");
  SUMK_ARG(np == 1 || rows_per_out == M, "splitk: several products take one output each");
  SUMK_ARG(slab_elems >= (size_t)np * M * N, "splitk: slab too small");
  SUMK_ARG(!src16 || gemm_b16_ok(M, N, K, lda, ldb, false, false), "splitk: operands not eligible for the bf16-source kernel");
  const int small = (src16 || gemm_tiles(M, N, 0) >= 64) ? 0 : 1;     // (src16: A and B are bf16 arrays)
  const int wide = (src16 && M >= 512 && N >= 512 && M % 256 == 0 && N % 256 == 0) ? 256 : 0;    // 256x256 tiles, one block per CU
  const int tiles = np * (wide ? gemm_tiles_wide(M, N, 256) : gemm_tiles(M, N, small));
  // K slices so that S x tiles fills the resident slots of the persistent grid ONCE (768 blocks of the 128x128 kernel, 2048 of the
  // 64x64 one, 256 of the wide one): every block then walks exactly one (long) tile.  The first version aimed at >= 1024 tiles: 1152
  // for the QKV weight gradient = one and a half rounds, the second half-empty (744 -> 6xx us), 1024 for the D x D ones.
  const int slots = wide ? 256 : small ? 2048 : 768;
  int S = std::max(1, slots / tiles);
  S = std::min(S, (K + 63) / 64);
  S = std::min(S, (int)std::min<size_t>(slab_elems / ((size_t)np * M * N), (size_t)(probs_cap / np)));
  S = std::max(S, 1);
  const int kq = src16 ? 64 : 32;              // whole k-tiles per slice
  int kchunk = ((K + S - 1) / S + kq - 1) / kq * kq;
  S = (K + kchunk - 1) / kchunk;
  SplitKSetup su;
  su.p = probs_dev; su.np = np; su.S = S; su.M = M; su.N = N; su.K = K; su.kchunk = kchunk; su.lda = lda; su.ldb = ldb;
  su.bt = wide ? 256 : gemm_tile_dim(small);
  const int64_t esz = src16 ? 2 : 4;
  for (int p = 0; p < 4; ++p) {
    su.rel_a[p] = p < np ? ((const char*)A[p] - (const char*)A[0]) / esz : 0;
    su.rel_b[p] = p < np ? ((const char*)B[p] - (const char*)B[0]) / esz : 0;
  }
  if (mode != SPLITK_TABLE_READY) hipLaunchKernelGGL(splitk_setup_kernel, dim3((np * S + 63) / 64), dim3(64), 0, stream, su);
  if (mode == SPLITK_TABLE_ONLY) { SUMK_HIP(hipGetLastError()); return SUMK_OK; }
  GemmLaunch g;
  g.A = A[0]; g.B[0] = B[0]; g.C = slab; g.probs = probs_dev; g.nprob = np * S; g.small_tile = small; g.total_tiles = S * tiles;
  g.precision = precision; g.src16 = src16; g.wide16 = wide;
  // the tiles of one K slice share their operand rows: keep a slice on one XCD (measured: the bf16 training step 1.35 -> 1.21 ms,
  // fp32 unchanged; the per-video attention products did not gain and are left in plain order)
  g.group_remap = 1;
  SUMK_TRY(launch_gemm(GEMM_TN, EPI_NONE, g, stream));
  SlabReduceArgs r;
  r.slab = slab; for (int i = 0; i < 4; ++i) r.out[i] = out[i];
  r.S = S; r.M = np * M; r.N = N; r.rows_per_out = rows_per_out; r.ldo = ldo; r.alpha = alpha;
  int64_t mn = (int64_t)np * M * N;
  bool vec4 = (N % 4 == 0) && (ldo % 4 == 0);
  for (int i = 0; i < 4; ++i) vec4 = vec4 && (out[i] == nullptr || ((uintptr_t)out[i] & 15) == 0);
  if (vec4) hipLaunchKernelGGL(slab_reduce4_kernel, dim3((unsigned)((mn / 4 + 255) / 256)), dim3(256), 0, stream, r);
  else hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((mn + 255) / 256)), dim3(256), 0, stream, r);
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}

int gemm_tn_splitk_accum(const float* A, int lda, const float* B, int ldb, int M, int N, int K, float* slab,
                         size_t slab_elems, GemmProb* probs_dev, int probs_cap, float* const out[4], int rows_per_out,
                         int ldo, float alpha, hipStream_t stream, int precision, int src16, int mode) {
  const float* const As[1] = {A};
  const float* const Bs[1] = {B};
  return gemm_tn_splitk_accum_multi(1, As, Bs, lda, ldb, M, N, K, slab, slab_elems, probs_dev, probs_cap, out, rows_per_out, ldo, alpha,
                                    stream, precision, src16, mode);
}

// ------------------------------------------------------------------------------------------- column sums
// out[c] += sum_r X[r, c]  in two deterministic stages: (chunk, column) partials, then a fixed-order sum.
__global__ void colsum_partial_kernel(const float* __restrict__ X, int ld, int R, int N, int rows_per_chunk,
                                      float* __restrict__ partial) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= N) return;
  int r0 = blockIdx.y * rows_per_chunk, r1 = min(R, r0 + rows_per_chunk);
  float s = 0.f;
  for (int r = r0; r < r1; ++r) s += X[(int64_t)r * ld + c];
  partial[(int64_t)blockIdx.y * N + c] = s;
}
// block = 16 columns x 16 partial-groups; each thread strides over the partials of its group, then a fixed-order LDS
// combine (deterministic).  n_part can be ~1000 (per-wave partials of the LayerNorm backward), so the partial axis must
// be spread over threads: a one-thread-per-column loop took 190 us per call.
__global__ __launch_bounds__(256) void partial_reduce_kernel(const float* __restrict__ partial, int n_part, int stride, int N,
                                                             float* __restrict__ out) {
  __shared__ float red[16][17];
  const int cl = threadIdx.x & 15, pg = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  float s = 0.f;
  if (c < N)
    for (int p = pg; p < n_part; p += 16) s += partial[(int64_t)p * stride + c];
  red[pg][cl] = s;
  __syncthreads();
  if (pg == 0 && c < N) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q][cl];
    out[c] += t;
  }
}
// four columns per thread, four rows in flight per thread (N % 4 == 0, ld % 4 == 0, X 16-byte aligned)
__global__ __launch_bounds__(256) void colsum_partial4_kernel(const float* __restrict__ X, int ld, int R, int N, int rows_per_chunk,
                                                              float* __restrict__ partial) {
  const int c = 4 * (blockIdx.x * blockDim.x + threadIdx.x);
  if (c >= N) return;
  const int r0 = blockIdx.y * rows_per_chunk, r1 = min(R, r0 + rows_per_chunk);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  int r = r0;
  for (; r + 4 <= r1; r += 4) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(X + (int64_t)(r + u) * ld + c);
#pragma unroll
    for (int u = 0; u < 4; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
  }
  for (; r < r1; ++r) {
    const float4 v = *reinterpret_cast<const float4*>(X + (int64_t)r * ld + c);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  *reinterpret_cast<float4*>(partial + (int64_t)blockIdx.y * N + c) = s;
}
// segs[k].out[c] += sum_r X[r*ld + segs[k].off + c]: ONE pass over X for every output (an LSTM layer's b_ih and b_hh receive the
// same column sums of dG, both directions side by side: one pass instead of four) and one fixed-order reduce launch.
int colsum_multi(const float* X, int ld, int R, int N, float* partial, int max_chunks, const ReduceSeg* segs, int nseg,
                 hipStream_t stream) {
  int chunks = std::max(1, std::min(max_chunks, (R + 31) / 32));
  int rpc = (R + chunks - 1) / chunks;
  chunks = (R + rpc - 1) / rpc;
  if (N % 4 == 0 && ld % 4 == 0 && ((uintptr_t)X & 15) == 0)
    hipLaunchKernelGGL(colsum_partial4_kernel, dim3((N / 4 + 255) / 256, chunks), dim3(256), 0, stream, X, ld, R, N, rpc, partial);
  else
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((N + 255) / 256, chunks), dim3(256), 0, stream, X, ld, R, N, rpc, partial);
  SUMK_HIP(hipGetLastError());
  return partial_reduce_multi(partial, chunks, N, segs, nseg, stream);
}
int colsum_accum(const float* X, int ld, int R, int N, float* partial, int max_chunks, float* out, hipStream_t stream) {
  if (N % 4 == 0) {     // the float4 pass + the one-launch reduce
    const ReduceSeg seg = {0, N, out};
    return colsum_multi(X, ld, R, N, partial, max_chunks, &seg, 1, stream);
  }
  int chunks = std::max(1, std::min(max_chunks, (R + 63) / 64));
  int rpc = (R + chunks - 1) / chunks;
  chunks = (R + rpc - 1) / rpc;
  hipLaunchKernelGGL(colsum_partial_kernel, dim3((N + 255) / 256, chunks), dim3(256), 0, stream, X, ld, R, N, rpc, partial);
  hipLaunchKernelGGL(partial_reduce_kernel, dim3((N + 15) / 16), dim3(256), 0, stream, partial, chunks, N, N, out);
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}
// block = 64 slot columns (16 lanes x float4: 256 contiguous bytes per partial row) x 16 partial groups; every thread has its
// n_part / 16 loads in flight at once, then a fixed-order LDS combine (deterministic).
struct ReduceMultiArgs { const float* partial; int32_t n_part, stride, nseg; int32_t off[6], n[6]; float* out[6]; };
__global__ __launch_bounds__(256) void partial_reduce_multi_kernel(ReduceMultiArgs a) {
  __shared__ float4 red[16][17];
  const int cl = threadIdx.x & 15, pg = threadIdx.x >> 4;
  // column blocks are laid over the segments one after the other (each segment padded to 64 columns)
  int blk = blockIdx.x, seg = 0;
  for (; seg < a.nseg; ++seg) { const int nb = (a.n[seg] + 63) >> 6; if (blk < nb) break; blk -= nb; }
  if (seg >= a.nseg) return;
  const int c = blk * 64 + 4 * cl;                     // column inside the segment
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < a.n[seg]) {
    const float* base = a.partial + a.off[seg] + c;
    for (int p0 = pg; p0 < a.n_part; p0 += 64) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int p = p0 + 16 * u;
        v[u] = p < a.n_part ? *reinterpret_cast<const float4*>(base + (int64_t)p * a.stride) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
  }
  red[pg][cl] = s;
  __syncthreads();
  if (pg == 0 && c < a.n[seg]) {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int q = 0; q < 16; ++q) { const float4 o = red[q][cl]; t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w; }
    float* o = a.out[seg] + c;
    const int left = a.n[seg] - c;
    o[0] += t.x;
    if (left > 1) o[1] += t.y;
    if (left > 2) o[2] += t.z;
    if (left > 3) o[3] += t.w;
  }
}
int partial_reduce_multi(const float* partial, int n_part, int stride, const ReduceSeg* segs, int nseg, hipStream_t stream) {
  SUMK_ARG(nseg >= 0 && nseg <= 6 && stride % 4 == 0, "partial_reduce_multi: bad segment list");
  ReduceMultiArgs a;
  a.partial = partial; a.n_part = n_part; a.stride = stride; a.nseg = 0;
  int blocks = 0;
  for (int k = 0; k < nseg; ++k) {
    if (segs[k].out == nullptr || segs[k].n <= 0) continue;
    SUMK_ARG(segs[k].off % 4 == 0, "partial_reduce_multi: segment offset %d is not a multiple of 4", segs[k].off);
    a.off[a.nseg] = segs[k].off; a.n[a.nseg] = segs[k].n; a.out[a.nseg] = segs[k].out; ++a.nseg;
    blocks += (segs[k].n + 63) >> 6;
  }
  if (blocks == 0) return SUMK_OK;
  hipLaunchKernelGGL(partial_reduce_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a);
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}
int partial_reduce_accum(const float* partial, int n_part, int stride, int N, float* out, hipStream_t stream) {
  hipLaunchKernelGGL(partial_reduce_kernel, dim3((N + 15) / 16), dim3(256), 0, stream, partial, n_part, stride, N, out);
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}

}  // namespace sumk

// ------------------------------------------------------------------------------------------------ C ABI
// The plain GEMM entry points keep their one-entry problem table in a small per-thread device scratch that is
// allocated once (first call) -- the only allocation the library ever makes, and only on this test/bench path.
namespace {
sumk::GemmProb* scratch_prob() {
  static thread_local sumk::GemmProb* p = nullptr;
  if (!p) { if (hipMalloc(&p, sizeof(sumk::GemmProb)) != hipSuccess) p = nullptr; }
  return p;
}
int plain_gemm(sumk::GemmLayout layout, const float* A, const float* B, float* C, int M, int N, int K, int lda,
               int ldb, void* stream, int precision = SUMK_PRECISION_FP32) {
  using namespace sumk;
  SUMK_ARG(A && B && C, "gemm: null pointer");
  SUMK_ARG(M > 0 && N > 0 && K > 0, "gemm: non-positive size M=%d N=%d K=%d", M, N, K);
  SUMK_ARG(lda % 4 == 0 && ldb % 4 == 0, "gemm: leading dimensions must be multiples of 4 (lda=%d ldb=%d)", lda, ldb);
  GemmProb* p = scratch_prob();
  SUMK_ARG(p != nullptr, "gemm: cannot allocate problem scratch");
  hipStream_t s = (hipStream_t)stream;
  int small = (M <= 64 || N <= 64) ? 1 : 0;
  if (const char* env = SUMK_TUNE_ENV("SUMK_ROW_CFG")) if (env[0] >= '0' && env[0] <= '2') small = env[0] - '0';
#ifdef SUMK_DIAG
  if (getenv("SUMK_FAKE_LD")) { lda = 0; ldb = 0; }   // `make DIAG=1` only: every row aliases row 0 (tiny footprint, all cache hits; wrong results by design)
#endif
  SUMK_TRY(fill_single_prob(p, M, N, K, lda, ldb, N, 0, small, s));
  GemmLaunch g;
  g.A = A; g.B[0] = B; g.C = C; g.probs = p; g.nprob = 1; g.small_tile = small;
  g.total_tiles = gemm_tiles(M, N, small); g.xcd_M = M; g.xcd_N = N; g.precision = precision;
  // a K-contiguous operand is fetched in 16-byte chunks: with K % 4 != 0 the last chunk of a row runs into the next row, which
  // only the register-staged kernel masks element by element (internal callers pad such rows with zeros instead)
  g.no_dma = (layout != GEMM_TN && K % 4 != 0) ? 1 : 0;
  return launch_gemm(layout, EPI_NONE, g, s);
}
}  // namespace

extern "C" int sumk_prof_gemm_stamps(uint64_t* out, int32_t n_blocks) {
  using namespace sumk;
#ifdef SUMK_DIAG
  static const bool on = getenv("SUMK_GEMM_DBG") && (atoi(getenv("SUMK_GEMM_DBG")) & 2);
#else
  constexpr bool on = false;
#endif
  SUMK_ARG(on && out && n_blocks != 0 && n_blocks <= 2048 && n_blocks >= -2048, "gemm stamps: needs a diagnostic build (make -C summarizer_amd/csrc DIAG=1) started with SUMK_GEMM_DBG=2 (n_blocks <= 2048)");
  SUMK_HIP(hipDeviceSynchronize());
  // n_blocks < 0: the fine records of SUMK_GEMM_DBG & 4 (8 values per block, -n_blocks of them)
  if (n_blocks < 0) { SUMK_HIP(hipMemcpy(out, gemm_stamp_buffer() + 2048 * 4, (size_t)(-n_blocks) * 8 * sizeof(uint64_t), hipMemcpyDeviceToHost)); return SUMK_OK; }
  SUMK_HIP(hipMemcpy(out, gemm_stamp_buffer(), (size_t)n_blocks * 4 * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return SUMK_OK;
}

extern "C" int sumk_gemm_nt(const float* A, const float* B, float* C, int32_t M, int32_t N, int32_t K, void* stream) {
  return plain_gemm(sumk::GEMM_NT, A, B, C, M, N, K, K, K, stream);
}
extern "C" int sumk_gemm_prec(int32_t layout, const float* A, const float* B, float* C, int32_t M, int32_t N, int32_t K,
                              int32_t precision, void* stream) {
  SUMK_ARG(precision >= SUMK_PRECISION_FP32 && precision <= SUMK_PRECISION_MAX, "gemm: unknown precision %d", precision);
  SUMK_ARG(layout >= 0 && layout <= 2, "gemm: layout must be 0 (NT), 1 (NN) or 2 (TN), got %d", layout);
  if (layout == 0) return plain_gemm(sumk::GEMM_NT, A, B, C, M, N, K, K, K, stream, precision);
  if (layout == 1) return plain_gemm(sumk::GEMM_NN, A, B, C, M, N, K, K, N, stream, precision);
  return plain_gemm(sumk::GEMM_TN, A, B, C, M, N, K, M, N, stream, precision);
}
// bf16 operands in HBM (gemm_b16.hip).  workspace == nullptr: C = product.  workspace given (TN only): deterministic split-K,
// C += product (the weight-gradient form; workspace = [64 problem entries | fp32 slabs], at least 8192 + 4 M N bytes).
extern "C" int sumk_gemm_bf16src(int32_t layout, const void* A16, const void* B16, float* C, int32_t M, int32_t N, int32_t K,
                                 void* workspace, size_t workspace_bytes, void* stream) {
  using namespace sumk;
  SUMK_ARG(layout >= 0 && layout <= 2, "gemm: layout must be 0 (NT), 1 (NN) or 2 (TN), got %d", layout);
  SUMK_ARG(A16 && B16 && C && M > 0 && N > 0 && K > 0, "gemm_bf16src: bad arguments");
  const int lda = layout == 2 ? M : K, ldb = layout == 0 ? K : N;
  SUMK_ARG(gemm_b16_ok(M, N, K, lda, ldb, layout != 2, layout == 0),
           "gemm_bf16src: M=%d N=%d K=%d is not eligible (K %% 64 for K-contiguous operands, rows %% 8 otherwise, offsets < 2^31)", M, N, K);
  hipStream_t s = (hipStream_t)stream;
  if (workspace) {
    SUMK_ARG(layout == 2, "gemm_bf16src: split-K is the TN form");
    SUMK_ARG(workspace_bytes >= 8192 + (size_t)M * N * 4 && ((uintptr_t)workspace & 255) == 0, "gemm_bf16src: workspace too small or misaligned");
    static_assert(sizeof(GemmProb) * 64 <= 8192, "problem table");
    float* const out[4] = {C, nullptr, nullptr, nullptr};
    return gemm_tn_splitk_accum((const float*)A16, lda, (const float*)B16, ldb, M, N, K, (float*)((char*)workspace + 8192),
                                (workspace_bytes - 8192) / 4, (GemmProb*)workspace, 64, out, M, N, 1.f, s, SUMK_PRECISION_BF16, 1);
  }
  GemmProb* p = scratch_prob();
  SUMK_ARG(p != nullptr, "gemm: cannot allocate problem scratch");
  int wide = gemm_b16_wide_bm(M, N);
  if (const char* env = getenv("SUMK_B16_WIDE")) wide = atoi(env);     // probe override: 0 / 192 / 256
  SUMK_TRY(fill_single_prob(p, M, N, K, lda, ldb, N, 0, wide ? 3 : 0, s));
  GemmLaunch g;
  g.A = (const float*)A16; g.B[0] = (const float*)B16; g.C = C; g.probs = p; g.nprob = 1; g.small_tile = 0;
  g.total_tiles = wide ? gemm_tiles_wide(M, N, wide) : gemm_tiles(M, N, 0); g.xcd_M = M; g.xcd_N = N; g.precision = SUMK_PRECISION_BF16;
  g.src16 = 1; g.wide16 = wide;
  return launch_gemm((GemmLayout)layout, EPI_NONE, g, s);
}
extern "C" int sumk_gemm_nn(const float* A, const float* B, float* C, int32_t M, int32_t N, int32_t K, void* stream) {
  return plain_gemm(sumk::GEMM_NN, A, B, C, M, N, K, K, N, stream);
}
extern "C" int sumk_gemm_tn(const float* A, const float* B, float* C, int32_t M, int32_t N, int32_t K, void* stream) {
  return plain_gemm(sumk::GEMM_TN, A, B, C, M, N, K, M, N, stream);
}
