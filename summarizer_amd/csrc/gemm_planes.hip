// GEMM on PRE-SPLIT bf16 planes, staged by LDS-DMA: the row-wise projections of the split-bf16 inference modes.
//
// gemm_split.hip splits every fp32 operand into bf16 planes on its way into LDS, inside the k-loop, for every tile: at 2-3
// MFMAs per product that conversion work (VALU + staging registers: 2 waves / SIMD) is what bounds the kernel (bf16x6: 43 % of
// the 417 TFLOP/s six MFMAs allow).  Here the planes exist in HBM before the GEMM starts --
//   x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2)          stored as  [plane][row][K]  bf16
// (weights: split once per call, 5 D x D matrices = 12 us; activations: by the streaming kernel below, or by the producing
// kernel) -- so the k-loop is pure copy + matrix work: every operand byte goes global -> LDS by `global_load_lds_dwordx4` into a
// double-buffered ring, ONE barrier per k-tile, no VALU, no staging registers.
//   C(M,N) = sum over plane pairs (i, j), i + j <= NP - 1, of A_i(M,K) . B_j(N,K)^T      NT only, K % 32 == 0, fp32 accumulate
// with the same products and the same smallest-terms-first order as the in-loop kernel (NP = 3: six MFMAs per 16 k, fp32-grade;
// NP = 2: three; NP = 1: plain bf16).
// Block = 8 waves (2 x 4), tile 128 x 128 x 32, an S-stage LDS ring with the DMAs S - 1 k-tiles ahead; LDS image per operand and plane: [row][32] bf16 = 64-B rows, the 16-B chunk c of
// row r holds global chunk c ^ ((r >> 2) & 3): a `ds_read_b128` lane group (16 rows, one logical chunk) then covers 16 distinct
// 16-B slots of the 256-B bank row (rows with equal r & 3 share a 64-B quarter; the XOR separates them) -- conflict free.
// Cost model (MI355X_MICROARCH.md; profiles/r02_probe_mfma_f32_ceiling.txt): a 1-KiB DMA piece costs its SIMD ~62 cycles, a
// 32x32x16 bf16 MFMA 32: per wave and k-tile 2 NP pieces against 4 x (pairs) MFMAs -> NP = 3: 372 vs 768 cycles (ceiling 0.67 of
// 417 TFLOP/s), NP = 2: 248 vs 384 (0.61 of 833), NP = 1: 124 vs 128 (0.51 of 2500).
#include "gemm_device.h"
#include <algorithm>
#include <cstdlib>

namespace sumk {

typedef __bf16 pl_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 pl_bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* pl_lds_void_p;

// One LDS-DMA piece (see gemm_dma.hip: inline asm so that hipcc does not wait for it in front of the next ds_read).
__device__ __forceinline__ void pl_dma16(const void* src, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(lds_byte_addr) : "memory");
}

struct PlanesKArgs {
  GemmKArgs g;            // A / B[] carry the plane-0 base pointers (bf16, reinterpret); C, R, bias, probs, ... as everywhere
  int64_t a_plane;        // elements between consecutive planes of A
  int64_t b_plane;        // ... of every B group
};

// S ring stages; the DMAs run S - 1 k-tiles ahead of the MFMAs.  At 24 (NP = 3) .. 4 (NP = 1) bf16 MFMAs per wave and k-tile an MFMA
// phase lasts only 0.1-0.6 us -- less than one L2 round trip -- so a double buffer (one k-tile ahead) leaves the load latency in
// the open; two to three k-tiles in flight cover it.  The wait in front of a k-tile is COUNTED (`s_waitcnt vmcnt((S-2) * NL)`: all
// but the pieces of the S - 2 younger k-tiles have landed); only the drain at the end of the block's walk waits for zero.
template <int NP> struct PlanesCfg { static constexpr int S = NP == 3 ? 3 : 4; };

template <int NP, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_planes_kernel(PlanesKArgs pa) {
  const GemmKArgs& ka = pa.g;
  constexpr int BM = 128, BN = 128, BK = 32, WN = 4, NW = 8, S = PlanesCfg<NP>::S;
  constexpr int WTM = BM / 2, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;         // 64 x 32 per wave: 2 x 1 MFMA tiles
  constexpr int PLANE_BYTES = BM * BK * 2;                                          // 8 KB: one operand, one plane
  constexpr int STAGE_BYTES = 2 * NP * PLANE_BYTES;                                 // A planes, then B planes
  constexpr int NPIECE = STAGE_BYTES / 1024, NL = NPIECE / NW;                     // 16 NP pieces, 2 NP per wave
  __shared__ __attribute__((aligned(16))) char lds[S * STAGE_BYTES];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;

  // ---- fetch cursor: (ftile, fk0) = the next k-tile to request; src[q] = this lane's source of piece q at k = 0 of that tile.
  // piece id p = q * NW + wave; operand = p / (8 NP), plane = (p / 8) % NP, 16-row block = p % 8
  const __bf16* src[NL];
  int ftile = blockIdx.x, fk0 = 0, fK = 0;
  bool fvalid = ftile < ka.total_tiles;
  auto fsetup = [&]() -> bool {
    GemmProb P; TileCtx c;
    if (!decode_tile<BM, BN>(ka, ftile, c, P)) return false;
    fK = P.K;
#pragma unroll
    for (int q = 0; q < NL; ++q) {
      const int p = q * NW + wave;
      const int is_b = p / (8 * NP), plane = (p / 8) % NP, rb = p % 8;
      const int row = rb * 16 + (lane >> 2);
      const int kc = 8 * ((lane & 3) ^ ((row >> 2) & 3));
      if (!is_b) {
        src[q] = reinterpret_cast<const __bf16*>(ka.A) + P.a_off + plane * pa.a_plane + (int64_t)min(c.m0 + row, P.M - 1) * P.lda + kc;
      } else {
        const int n = min(c.n0 + row, P.N - 1);
        int g = 0, nl = n;
        if (ka.n_group > 0) { g = n / ka.n_group; nl = n - g * ka.n_group; }
        const float* bg = g == 0 ? ka.B[0] : g == 1 ? ka.B[1] : g == 2 ? ka.B[2] : ka.B[3];
        src[q] = reinterpret_cast<const __bf16*>(bg) + P.b_off + plane * pa.b_plane + (int64_t)nl * P.ldb + kc;
      }
    }
    return true;
  };
  const unsigned lds0 = (unsigned)(uintptr_t)(pl_lds_void_p)lds;
  unsigned ldst[NL];
#pragma unroll
  for (int q = 0; q < NL; ++q) ldst[q] = __builtin_amdgcn_readfirstlane(lds0 + 1024u * (unsigned)(q * NW + wave));
  int inflight = 0;                 // k-tiles requested and not yet multiplied (wave-uniform)
  auto fetch = [&](int stage) {     // request the cursor's k-tile into ring stage `stage` and move the cursor on
    if (!fvalid) return;
    if (!(ka.dbg & 8)) {
#pragma unroll
      for (int q = 0; q < NL; ++q) pl_dma16(src[q] + fk0, ldst[q] + (unsigned)(stage * STAGE_BYTES));
    }
    ++inflight;
    fk0 += BK;
    if (fk0 >= fK) { ftile += gridDim.x; fk0 = 0; fvalid = ftile < ka.total_tiles && fsetup(); }
  };

  // ---- fragment addresses: row (wave origin + t*32 + li), logical chunk 2*ks + lh -> physical chunk ^ ((li >> 2) & 3)
  const int sw = (li >> 2) & 3;
  int ch[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) ch[ks] = 16 * ((2 * ks + lh) ^ sw);
  const int a_row = (wm * WTM + li) * 64, b_row = NP * PLANE_BYTES + (wn * WTN + li) * 64;

  if (!fvalid) return;
  fvalid = fsetup();
  if (!fvalid) return;   // (remapped walk: a rectangle's tiles are exhausted in increasing order)
  int ctile = blockIdx.x;
  TileCtx cur;
  { GemmProb P; decode_tile<BM, BN>(ka, ctile, cur, P); }
#pragma unroll
  for (int s = 0; s < S - 1; ++s) fetch(s);         // prologue: the first S - 1 k-tiles of this block's walk
  int gk = 0, k0 = 0;                               // stage of the k-tile being multiplied = gk % S
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  while (true) {
    // the pieces of the k-tile about to be multiplied have landed: every older operation of this wave is complete once at most
    // the pieces of the S - 2 younger k-tiles are outstanding (fewer are in flight only while the walk drains: wait for all)
    if (inflight >= S - 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((S - 2) * NL) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();              // ... every wave's have; and every wave is done reading the stage requested next
    --inflight;
    fetch((gk + S - 1) % S);
    const char* sS = lds + (gk % S) * STAGE_BYTES;
    if (!(ka.dbg & 4))
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      pl_bf16x8 af[NP][TM], bf[NP][TN];
#pragma unroll
      for (int q = 0; q < NP; ++q) {
#pragma unroll
        for (int t = 0; t < TM; ++t) af[q][t] = *reinterpret_cast<const pl_bf16x8*>(sS + q * PLANE_BYTES + a_row + t * 32 * 64 + ch[ks]);
#pragma unroll
        for (int t = 0; t < TN; ++t) bf[q][t] = *reinterpret_cast<const pl_bf16x8*>(sS + q * PLANE_BYTES + b_row + t * 32 * 64 + ch[ks]);
      }
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {   // smallest terms first: plane pairs (i, j) by descending i + j
#pragma unroll
          for (int sum = NP - 1; sum >= 0; --sum)
#pragma unroll
            for (int i = NP - 1; i >= 0; --i) {
              const int j = sum - i;
              if (j < 0 || j >= NP) continue;
              acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][tm], bf[j][tn], acc[tm][tn], 0, 0, 0);
            }
        }
    }
    ++gk;
    k0 += BK;
    if (k0 >= cur.K) {            // tile finished (wave-uniform)
      epilogue_store<EPI, TM, TN>(ka, cur, acc, cur.m0 + wm * WTM, cur.n0 + wn * WTN, li, lh);
      ctile += gridDim.x;
      GemmProb P;
      if (ctile >= ka.total_tiles || !decode_tile<BM, BN>(ka, ctile, cur, P)) break;
      k0 = 0;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
  }
}

// ------------------------------------------------------------------------------------------- split kernel
// dst[p][i] = plane p of src[i], p < NP: x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2) (every subtraction exact).
template <int NP>
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, int64_t n,
                                                           int64_t plane_stride) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  const int64_t n4 = n >> 2, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    v4f r = reinterpret_cast<const v4f*>(src)[i];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const pl_bf16x4 b = __builtin_convertvector(r, pl_bf16x4);
      reinterpret_cast<pl_bf16x4*>(dst + p * plane_stride)[i] = b;
      if (p + 1 < NP) r = r - __builtin_convertvector(b, v4f);
    }
  }
}

int planes_of_precision(int precision) {
  return precision == SUMK_PRECISION_BF16X6 ? 3 : precision == SUMK_PRECISION_BF16X3 ? 2 : precision == SUMK_PRECISION_BF16 ? 1 : 0;
}

// Opt-in (SUMK_PLANES=1).  Measured on the S-TVSum batch ("Pre-split planes"): results are bit-identical to the in-loop
// split kernels, but only the one-plane mode gains (QKV 168 -> 113 us); with two / three planes the k-loop is bound by what a CU
// can pull from L2 (~30 B/clk: 102 us of pure DMA for 16 KB k-tiles) plus the DMA issue slots (62 SIMD cycles per KiB), and the
// three extra streaming passes that split x, CTX and Y1 cost more than the projections gain (bf16x6 step 0.95 -> 1.07 ms).
bool gemm_planes_enabled() {
  static const bool on = getenv("SUMK_PLANES") && getenv("SUMK_PLANES")[0] == '1';
  return on;
}

int launch_split_planes(const float* src, void* dst_bf16, int64_t n, int64_t plane_stride, int precision, hipStream_t stream) {
  SUMK_ARG(src && dst_bf16 && n > 0 && (n & 3) == 0, "split_planes: bad argument (n must be a multiple of 4)");
  const int np = planes_of_precision(precision);
  SUMK_ARG(np > 0, "split_planes: precision %d has no bf16 planes", precision);
  const int blocks = (int)std::min<int64_t>((n / 4 + 255) / 256, 4096);
  __bf16* dst = (__bf16*)dst_bf16;
  if (np == 3) hipLaunchKernelGGL(split_planes_kernel<3>, dim3(blocks), dim3(256), 0, stream, src, dst, n, plane_stride);
  else if (np == 2) hipLaunchKernelGGL(split_planes_kernel<2>, dim3(blocks), dim3(256), 0, stream, src, dst, n, plane_stride);
  else hipLaunchKernelGGL(split_planes_kernel<1>, dim3(blocks), dim3(256), 0, stream, src, dst, n, plane_stride);
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}

template <int NP>
static int launch_planes_np(GemmEpi epi, const PlanesKArgs& pa, int tiles, hipStream_t s) {
  constexpr int occ = NP == 1 ? 2 : 1;     // LDS: 3 x 48 = 144 KB (NP = 3), 4 x 32 = 128 KB (NP = 2), 4 x 16 = 64 KB (NP = 1) per block
  dim3 grid(std::min(tiles, 256 * occ)), block(512);
  switch (epi) {
    case EPI_NONE: hipLaunchKernelGGL((gemm_planes_kernel<NP, EPI_NONE>), grid, block, 0, s, pa); break;
    case EPI_RESIDUAL: hipLaunchKernelGGL((gemm_planes_kernel<NP, EPI_RESIDUAL>), grid, block, 0, s, pa); break;
    case EPI_BIAS_RELU: hipLaunchKernelGGL((gemm_planes_kernel<NP, EPI_BIAS_RELU>), grid, block, 0, s, pa); break;
    default: set_error("gemm_planes: epilogue %d is not instantiated", (int)epi); return SUMK_ERR_ARG;
  }
  return SUMK_OK;
}

// Row-wise NT GEMM on planes: single problem (g.probs[0]), A planes (M, K) with plane stride a_plane, up to four B groups.
int launch_gemm_planes(GemmEpi epi, const GemmLaunch& g, int64_t a_plane, int64_t b_plane, hipStream_t stream) {
  SUMK_ARG(g.A && g.B[0] && g.C && g.probs && g.nprob == 1 && g.small_tile == 0, "gemm_planes: single 128x128-tiled problem expected");
  const int np = planes_of_precision(g.precision);
  SUMK_ARG(np > 0, "gemm_planes: precision %d has no bf16 planes", g.precision);
  PlanesKArgs pa;   // (callers guarantee K % 32 == 0 and K >= 128: the fetch cursor runs up to 3 k-tiles ahead, within one tile of the MFMAs)
  GemmKArgs& ka = pa.g;
  ka.lean = 0;
  ka.A = g.A;
  for (int i = 0; i < 4; ++i) { ka.B[i] = g.B[i]; ka.bias0[i] = g.bias0[i]; ka.bias1[i] = g.bias1[i]; }
  ka.C = g.C; ka.R = g.R; ka.probs = g.probs; ka.nprob = 1; ka.n_group = g.n_group; ka.alpha = g.alpha;
  static const int dbg = getenv("SUMK_GEMM_DBG") ? atoi(getenv("SUMK_GEMM_DBG")) : 0;   // 4: no MFMAs, 8: no DMAs (timing diagnostics)
  ka.total_tiles = g.total_tiles; ka.xcd_tiles_m = 0; ka.group_remap = 0; ka.dbg = dbg; ka.dbg_buf = nullptr;
  ka.drop.seed = 0; ka.drop.thr = 0; ka.drop.scale = 1.f; ka.drop_site = 0;
  ka.moments = nullptr; ka.ln_stats = ka.ln_c1 = ka.ln_c2 = nullptr;
  pa.a_plane = a_plane; pa.b_plane = b_plane;
  if (g.xcd_M > 0) {
    const int tm = (g.xcd_M + 127) / 128, tn = (g.xcd_N + 127) / 128;
    if (tn % 4 == 0 && tm >= 16) { ka.xcd_tiles_m = tm; ka.total_tiles = 8 * ((tm + 1) / 2) * (tn / 4); }
  }
  if (g.prof_tag >= 0) prof_begin(g.prof_tag, stream);
  int rc = np == 3 ? launch_planes_np<3>(epi, pa, ka.total_tiles, stream)
         : np == 2 ? launch_planes_np<2>(epi, pa, ka.total_tiles, stream) : launch_planes_np<1>(epi, pa, ka.total_tiles, stream);
  if (g.prof_tag >= 0) prof_end(g.prof_tag, stream);
  if (rc != SUMK_OK) return rc;
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}

}  // namespace sumk
