// Exact-fp32 MFMA GEMM staged by LDS-DMA (gfx950 `global_load_lds_dwordx4`): the alternative to gemm_f32.hip's register staging
// (opt-in: SUMK_GEMM_DMA=1).
//
// Operand tiles go global -> LDS directly, asynchronously, into a DOUBLE-BUFFERED LDS ring: no VGPR staging, no ds_write pass,
// ONE workgroup barrier per k-tile; the DMA of k-tile t+1 (or of the NEXT output tile's first k-tile: the tile walk is
// persistent) is in flight while the MFMAs of k-tile t run, its eight 1-KiB pieces issued one per MFMA group.
// What was measured (MI355X, scripts/probes/mfma_f32_ceiling.hip, scripts/gemm_stamp_probe.py; DESIGN.md has the table):
//   * the bare loop (fragment reads + 64 MFMAs + one barrier) sustains 152-155 TFLOP/s; every 1-KiB DMA piece costs the SIMD one
//     MFMA slot (~62 cycles), so 8 pieces per 64 MFMAs cap this tiling at 136 TFLOP/s;
//   * __builtin_amdgcn_global_load_lds is an LDS store to hipcc, which then waits `vmcnt(0)` before the next ds_read: the DMA
//     latency lands in front of the SAME k-tile's fragment reads (88 TFLOP/s in the probe) -- hence the inline-asm form below;
//   * in the full GEMM this kernel's k-loop runs at 93 % of the matrix-pipe floor (2 blocks / CU), the register-staged one at
//     97 % (3 blocks / CU); both end at 120-130 TFLOP/s because what is left is outside the loop: lock-stepped epilogue store
//     bursts (5-8 % of a block's cycles), the tail of the tile walk, and a ~2.25 GHz clock under this load.
// LDS images (an LDS-DMA instruction writes 64 lanes x 16 B = 1 KiB LINEARLY at a wave-uniform base, so padding is not
// available; bank conflicts are avoided by permuting what each lane FETCHES):
//   K-contiguous operand ("KC": A of NT/NN, B of NT): [row][BK] floats, 128-B rows, the 16-B chunk c of row r holds global
//     chunk c ^ ((r >> 1) & 7).  A `ds_read_b128` lane group covers 16 rows at one logical chunk: with that XOR the 8 even and
//     the 8 odd rows land on 16 distinct 16-B slots of the 256-B bank row -- conflict free (MI355X_MICROARCH.md, LDS table).
//   M/N-contiguous operand ("MC": B of NN, A and B of TN): [k][BT] floats in natural order, read with ds_read_b32
//     (consecutive lanes, consecutive columns).
// Rows / columns past M or N are clamped (they only feed outputs that are never stored); k past K must read zeros: those
// lanes fetch from a 16-byte zero page in the code object instead.
// Arithmetic, k order, epilogues and therefore RESULTS are bit-identical to gemm_f32.hip's exact-fp32 path
// (tests/test_gpu_vasnet.py::test_gemm_dma_equals_register_staged_kernel).
#include "gemm_device.h"
#include <algorithm>
#include <cstdlib>

namespace sumk {

__device__ __attribute__((aligned(16))) float g_gemm_zero_page[4] = {0.f, 0.f, 0.f, 0.f};

typedef __attribute__((address_space(3))) void* lds_void_p;
typedef const __attribute__((address_space(1))) void* glb_void_p;

// One LDS-DMA piece: 64 lanes x 16 B from per-lane global addresses to 1 KiB of LDS at a wave-uniform byte address.
// Inline asm, not __builtin_amdgcn_global_load_lds: to hipcc the builtin is an LDS store that may alias every later ds_read, so
// it put `s_waitcnt vmcnt(0)` between the DMA issue and the fragment reads of the SAME k-tile -- the whole load latency exposed
// (88 vs 136 TFLOP/s in scripts/probes/mfma_f32_ceiling.hip).  An asm statement is not counted; this kernel orders its DMAs
// itself: `s_waitcnt vmcnt(0)` + workgroup barrier before the first read of a stage (cdna_hip_programming.md 5.7).
// M0 (the LDS destination base) is compiler-reserved: saved, set and restored inside the one statement.
__device__ __forceinline__ void dma16(const float* src, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(lds_byte_addr) : "memory");
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// WM x WN waves of 64 lanes; block tile BM x BN; BK = 32.  OCC = blocks per CU the launch bounds are sized for.
template <int BM, int BN, int WM, int WN, bool A_KC, bool B_KC, int EPI, int OCC>
__global__ __launch_bounds__(64 * WM * WN, OCC * WM * WN / 4) void gemm_dma_kernel(GemmKArgs ka) {
  constexpr int BK = 32;
  constexpr int NW = WM * WN;
  constexpr int WTM = BM / WM, WTN = BN / WN;          // wave tile
  constexpr int TM = WTM / 32, TN = WTN / 32;          // 32x32 MFMA tiles per wave
  static_assert(WTM % 32 == 0 && WTN % 32 == 0, "wave tile must be a multiple of the 32x32 MFMA tile");
  constexpr int A_FLOATS = BM * BK, B_FLOATS = BN * BK, STAGE = A_FLOATS + B_FLOATS;
  constexpr int NIA = A_FLOATS / 256, NIB = B_FLOATS / 256;      // 1-KiB DMA pieces per operand tile
  static_assert(NIA % NW == 0 && NIB % NW == 0, "DMA pieces must divide over the waves");
  constexpr int NLA = NIA / NW, NLB = NIB / NW;                  // pieces per wave
  __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;

  // ---- per-lane DMA source description of the tile whose operands are being loaded
  const float* pa[NLA];
  const float* pb[NLB];
  int ka_off[NLA], kb_off[NLB];    // k index (KC: first k of the lane's chunk; MC: the lane's k-row) inside a k-tile

  auto setup = [&](int tile, TileCtx& c) -> bool {
    GemmProb P;
    if (!decode_tile<BM, BN>(ka, tile, c, P)) return false;
#pragma unroll
    for (int p = 0; p < NLA; ++p) {
      const int piece = p * NW + wave;
      if constexpr (A_KC) {                      // 8 rows x 128 B per piece
        const int row = piece * 8 + (lane >> 3);
        const int kc = 4 * ((lane & 7) ^ ((row >> 1) & 7));
        pa[p] = ka.A + P.a_off + (int64_t)min(c.m0 + row, P.M - 1) * P.lda + kc;
        ka_off[p] = kc;
      } else {                                   // (1024 / (4 BM)) k-rows x BM columns per piece
        constexpr int LPR = BM / 4;              // lanes per k-row
        const int krow = piece * (64 / LPR) + lane / LPR;
        int col = c.m0 + (lane % LPR) * 4;
        col = col < P.M ? col : 0;
        pa[p] = ka.A + P.a_off + col + (int64_t)krow * P.lda;
        ka_off[p] = krow;
      }
    }
#pragma unroll
    for (int p = 0; p < NLB; ++p) {
      const int piece = p * NW + wave;
      if constexpr (B_KC) {
        const int row = piece * 8 + (lane >> 3);
        const int kc = 4 * ((lane & 7) ^ ((row >> 1) & 7));
        const int n = min(c.n0 + row, P.N - 1);
        int g = 0, nl = n;
        if (ka.n_group > 0) { g = n / ka.n_group; nl = n - g * ka.n_group; }
        const float* bg = g == 0 ? ka.B[0] : g == 1 ? ka.B[1] : g == 2 ? ka.B[2] : ka.B[3];
        pb[p] = bg + P.b_off + (int64_t)nl * P.ldb + kc;
        kb_off[p] = kc;
      } else {
        constexpr int LPR = BN / 4;
        const int krow = piece * (64 / LPR) + lane / LPR;
        int col = c.n0 + (lane % LPR) * 4;
        col = col < P.N ? col : 0;
        pb[p] = ka.B[0] + P.b_off + col + (int64_t)krow * P.ldb;
        kb_off[p] = krow;
      }
    }
    return true;
  };

  // DMA pieces of one k-tile, this wave's share: pieces 0 .. NLA-1 belong to A, NLA .. NP-1 to B.  `src_of` = the lane's source
  // address of piece q for k-tile k0 of tile c (zero page past K); `put` issues it into ring stage `stg`.
  constexpr int NP = NLA + NLB;
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_p)lds;
  auto src_of = [&](const TileCtx& c, int k0, int q) -> const float* {
    const float* src;
    int koff;
    if (q < NLA) { src = A_KC ? pa[q] + k0 : pa[q] + (int64_t)k0 * c.lda; koff = ka_off[q]; }
    else { src = B_KC ? pb[q - NLA] + k0 : pb[q - NLA] + (int64_t)k0 * c.ldb; koff = kb_off[q - NLA]; }
    return (k0 + koff >= c.K) ? (const float*)g_gemm_zero_page : src;
  };
  auto put = [&](const float* src, int stg, int q) {
    const int fl = stg * STAGE + (q < NLA ? (q * NW + wave) * 256 : A_FLOATS + ((q - NLA) * NW + wave) * 256);
    dma16(src, __builtin_amdgcn_readfirstlane(lds0 + 4u * (unsigned)fl));
  };
  auto issue = [&](const TileCtx& c, int k0, int stg) {
#pragma unroll
    for (int q = 0; q < NP; ++q) put(src_of(c, k0, q), stg, q);
  };

  // ---- per-lane fragment addresses (floats, inside a stage)
  // KC: row (wave origin + t*32 + li), logical chunk 2*kk + lh  ->  physical chunk (2*kk) ^ (lh ^ ((li >> 1) & 7)):
  //     the wave origin and t*32 are multiples of 32, so the row's swizzle bits come from li alone.
  const int e = lh ^ ((li >> 1) & 7);
  int kc_off[BK / 8];
#pragma unroll
  for (int kk = 0; kk < BK / 8; ++kk) kc_off[kk] = 4 * ((2 * kk) ^ e);
  const int a_base = A_KC ? (wm * WTM + li) * BK : (4 * lh) * BM + wm * WTM + li;
  const int b_base = A_FLOATS + (B_KC ? (wn * WTN + li) * BK : (4 * lh) * BN + wn * WTN + li);

  int tile = blockIdx.x;
  if (tile >= ka.total_tiles) return;
  TileCtx cur, nxt;
  if (!setup(tile, cur)) return;
  int st = 0;
  issue(cur, 0, st);          // prologue: the first k-tile of the first tile
  unsigned long long t_begin = 0, t_k = 0, t_e = 0, n_t = 0;
  if (ka.dbg & 2) t_begin = __builtin_amdgcn_s_memtime();

  while (true) {
    unsigned long long ta = 0;
    if (ka.dbg & 2) ta = __builtin_amdgcn_s_memtime();
    f32x16 acc[TM][TN];
    if constexpr (EPI == EPI_RESIDUAL) {   // as the register-staged kernel: the accumulators start at R (bit-identical results)
      residual_init<TM, TN>(ka, cur, acc, cur.m0 + wm * WTM, cur.n0 + wn * WTN, li, lh);
    } else {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }

    const int next_tile = tile + gridDim.x;
    bool has_next = next_tile < ka.total_tiles;
    const int K = cur.K;
    for (int k0 = 0; k0 < K; k0 += BK) {
      // this wave's DMAs of this k-tile have landed; after the barrier every wave's have, and every wave has finished reading the
      // other stage, which the next DMAs overwrite
      dma_wait_all();
      __syncthreads();
      // what to fetch next: this tile's next k-tile, or (last k-tile) the NEXT tile's first operands, which then land under this
      // tile's last MFMAs and its epilogue
      bool fetch = true;
      const float* srcs[NP];
      if (k0 + BK < K) {
#pragma unroll
        for (int q = 0; q < NP; ++q) srcs[q] = src_of(cur, k0 + BK, q);
      } else if (has_next && (has_next = setup(next_tile, nxt))) {
#pragma unroll
        for (int q = 0; q < NP; ++q) srcs[q] = src_of(nxt, 0, q);
      } else {
        fetch = false;
#pragma unroll
        for (int q = 0; q < NP; ++q) srcs[q] = g_gemm_zero_page;
      }
      const float* sS = lds + st * STAGE;
#pragma unroll
      for (int kk = 0; kk < BK / 8; ++kk) {
        float av[TM][4], bv[TN][4];
#pragma unroll
        for (int t = 0; t < TM; ++t) {
          if constexpr (A_KC) {
            const float4 v = *reinterpret_cast<const float4*>(&sS[a_base + t * 32 * BK + kc_off[kk]]);
            av[t][0] = v.x; av[t][1] = v.y; av[t][2] = v.z; av[t][3] = v.w;
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) av[t][j] = sS[a_base + (kk * 8 + j) * BM + t * 32];
          }
        }
#pragma unroll
        for (int t = 0; t < TN; ++t) {
          if constexpr (B_KC) {
            const float4 v = *reinterpret_cast<const float4*>(&sS[b_base + t * 32 * BK + kc_off[kk]]);
            bv[t][0] = v.x; bv[t][1] = v.y; bv[t][2] = v.z; bv[t][3] = v.w;
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[t][j] = sS[b_base + (kk * 8 + j) * BN + t * 32];
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
              acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tm][j], bv[tn][j], acc[tm][tn], 0, 0, 0);
          // the NP pieces of the next k-tile are spread over the 16 MFMA groups of this one: a piece costs ~60 issue cycles,
          // which hide under the 64-cycle MFMAs only one at a time
          if (fetch) {
#pragma unroll
            for (int q = 0; q < NP; ++q)
              if ((q * 16) / NP == kk * 4 + j) put(srcs[q], st ^ 1, q);
          }
        }
      }
      st ^= 1;
    }

    unsigned long long tb = 0;
    if (ka.dbg & 2) tb = __builtin_amdgcn_s_memtime();
    if (!(ka.dbg & 1) || acc[0][0][0] == 12345.f)
    epilogue_store<EPI, TM, TN, true>(ka, cur, acc, cur.m0 + wm * WTM, cur.n0 + wn * WTN, li, lh);
    if (ka.dbg & 2) { const unsigned long long tc = __builtin_amdgcn_s_memtime(); t_k += tb - ta; t_e += tc - tb; n_t += 1; }
    if (!has_next) break;
    tile = next_tile;
    cur = nxt;
  }
  if ((ka.dbg & 2) && ka.dbg_buf && tid == 0 && blockIdx.x < 2048) {
    unsigned long long* o = ka.dbg_buf + (size_t)blockIdx.x * 4;
    o[0] = __builtin_amdgcn_s_memtime() - t_begin; o[1] = t_k; o[2] = t_e; o[3] = n_t;
  }
}

// Opt-in (SUMK_GEMM_DMA=1): measured equal to the register-staged kernel on the bench shapes (DESIGN.md "GEMM: where the
// cycles go"), so the default stays the kernel with the longer track record.
bool gemm_dma_enabled() {
  static const bool on = getenv("SUMK_GEMM_DMA") && getenv("SUMK_GEMM_DMA")[0] == '1';
  return on;
}

template <int BM, int BN, int WM, int WN, bool A_KC, bool B_KC, int OCC>
static int launch_dma_epi(GemmEpi epi, const GemmKArgs& ka, int tiles, hipStream_t s) {
  static const bool persist = !(getenv("SUMK_PERSIST") && getenv("SUMK_PERSIST")[0] == '0');
  dim3 grid(persist ? std::min(tiles, 256 * OCC) : tiles), block(64 * WM * WN);
#define SUMK_DMA_CASE(E) case E: hipLaunchKernelGGL((gemm_dma_kernel<BM, BN, WM, WN, A_KC, B_KC, E, OCC>), grid, block, 0, s, ka); break;
  switch (epi) {
    SUMK_DMA_CASE(EPI_NONE) SUMK_DMA_CASE(EPI_RESIDUAL) SUMK_DMA_CASE(EPI_BIAS_RELU) SUMK_DMA_CASE(EPI_BIAS2) SUMK_DMA_CASE(EPI_ACCUM)
    SUMK_DMA_CASE(EPI_BIAS_RESIDUAL)
    default: set_error("gemm: bad epilogue %d", (int)epi); return SUMK_ERR_ARG;
  }
#undef SUMK_DMA_CASE
  return SUMK_OK;
}

template <int BM, int BN, int WM, int WN, int OCC>
static int launch_dma_layout(GemmLayout layout, GemmEpi epi, const GemmKArgs& ka, int tiles, hipStream_t s) {
  if (layout == GEMM_NT) return launch_dma_epi<BM, BN, WM, WN, true, true, OCC>(epi, ka, tiles, s);
  if (layout == GEMM_NN) return launch_dma_epi<BM, BN, WM, WN, true, false, OCC>(epi, ka, tiles, s);
  return launch_dma_epi<BM, BN, WM, WN, false, false, OCC>(epi, ka, tiles, s);
}

// LDS per block = 2 stages x (BM + BN) x 32 floats: 64 KB (128x128) -> 2 blocks / CU, 48 KB (128x64) -> 3, 32 KB (64x64) -> 4.
int launch_gemm_dma(GemmLayout layout, GemmEpi epi, const GemmKArgs& ka, int tiles, int cfg, hipStream_t s) {
  static const int waves8 = getenv("SUMK_DMA_WAVES") ? atoi(getenv("SUMK_DMA_WAVES")) : 4;
  if (cfg == 1) return launch_dma_layout<64, 64, 2, 2, 4>(layout, epi, ka, tiles, s);
  if (cfg == 2) return launch_dma_layout<128, 64, 2, 2, 3>(layout, epi, ka, tiles, s);
  if (waves8 == 8) return launch_dma_layout<128, 128, 2, 4, 2>(layout, epi, ka, tiles, s);
  return launch_dma_layout<128, 128, 2, 2, 2>(layout, epi, ka, tiles, s);
}

}  // namespace sumk
