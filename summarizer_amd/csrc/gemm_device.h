// Device-side pieces shared by the GEMM kernels of libsumk.so (gemm_f32.hip / gemm_split.hip: register-staged): kernel argument block, scalar reads of the problem table, tile decode and the
// fused epilogues.
#pragma once
#include "sumk_internal.h"

namespace sumk {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct GemmKArgs {
  const float* A;
  const float* B[4];
  float* C;
  const float* R;
  const float* bias0[4];
  const float* bias1[4];
  const GemmProb* probs;
  int32_t nprob;
  int32_t n_group;
  int32_t total_tiles;   // loop bound of the persistent tile walk (virtual tiles when xcd_tiles_m > 0)
  Drop drop; uint32_t drop_site;   // epilogue dropout (drop.thr == 0: none); mask index = row * N + col
  int32_t xcd_tiles_m;   // > 0: single-problem launch with the XCD-aware tile map below; value = tiles along M
  int32_t group_remap;   // 1: grouped (per-video) launch, tile ids dealt so that one XCD walks a CONTIGUOUS range of tiles (decode_tile)
  float alpha;
  int32_t dbg;           // diagnostic switches (SUMK_GEMM_DBG): 1 = skip the epilogue stores, 2 = in-kernel cycle stamps
  unsigned long long* dbg_buf;   // dbg & 2: per block {total, k-loop, epilogue, tiles} shader cycles (scripts/gemm_stamp_probe.py)
  int32_t lean;                  // host-side only: 1 = NT 128x128 launch eligible for the buffer-load (VALU-free k-loop) instances
  float* moments;                // EPI_RESIDUAL_MOMENTS: float2[M][N / 32]
  unsigned short* C16;           // EPI_NONE: when set, bf16(C) is stored too, same offsets / leading dimension (operand of a later bf16-source GEMM); C may then be null
  const float* ln_stats; const float* ln_c1; const float* ln_c2;   // EPI_BIAS_RELU_HEAD with the LayerNorm of A applied to the product
  // SK instances of gemm_lean_kernel (GemmLaunch::sk): run-time epilogue (a GemmEpi), partial tiles, tickets, outputs 1..3
  int32_t sk_epi; float* sk_part; unsigned* sk_cnt; float* Csel[4];
};

// Scalar reads of the problem table (CONSTANT address space + wave-uniform index -> s_load, lgkmcnt).  As vector loads they
// left VM events pending on registers the k-loop reuses, and the compiler's waitcnt pass then put an s_waitcnt vmcnt(0) at
// the join in front of the fragment reads -- every k-tile waited for the NEXT k-tile's global loads before its own MFMAs.
typedef const __attribute__((address_space(4))) int64_t* cptr64;
typedef const __attribute__((address_space(4))) int32_t* cptr32;
__device__ __forceinline__ int prob_tile_start(const GemmProb* p, int i) { return ((cptr32)(uintptr_t)(p + i))[15]; }
__device__ __forceinline__ GemmProb load_prob(const GemmProb* p, int i) {
  const cptr64 q = (cptr64)(uintptr_t)(p + i);
  const cptr32 r = (cptr32)(uintptr_t)(p + i);
  GemmProb P;
  P.a_off = q[0]; P.b_off = q[1]; P.c_off = q[2]; P.r_off = q[3];
  P.M = r[8]; P.N = r[9]; P.K = r[10]; P.lda = r[11]; P.ldb = r[12]; P.ldc = r[13]; P.ldr = r[14];
  P.tile_start = r[15]; P.tiles_n = r[16];
  return P;
}
static_assert(offsetof(GemmProb, M) == 32 && offsetof(GemmProb, tile_start) == 60 && offsetof(GemmProb, tiles_n) == 64, "GemmProb layout");

// Per-tile scalars (wave-uniform, live in SGPRs).
struct TileCtx {
  int64_t c_off, r_off;
  int32_t M, N, K, lda, ldb, ldc, ldr, m0, n0, klast;
  const float* ab; const float* bb;   // LEAN kernels: operand bases of the problem (buffer descriptors; per-thread byte offsets beside them)
};

// Which sub-problem / tile does persistent tile id `tile` name?  Wave-uniform scalar work.  Returns false when a remapped
// (XCD-aware) walk has run past its rectangle.
template <int BM, int BN>
__device__ __forceinline__ bool decode_tile(const GemmKArgs& ka, int tile, TileCtx& c, GemmProb& P, int* prob_idx = nullptr) {
  if (ka.group_remap) {
    // Grouped launches number their tiles sub-problem by sub-problem, and block b runs on XCD b % 8 (round-robin dispatch; speed
    // only): dealt as they are, the tiles of one sub-problem land on all eight XCDs and each of the eight private L2s pulls that
    // sub-problem's operand rows from the Infinity Cache for itself.  Here XCD x (tile ids 8 j + x) walks the contiguous range
    // [start_x, start_x + n_x): whole sub-problems per XCD, operand rows fetched into ONE L2 and shared by their tiles.  Used by
    // the split-K weight-gradient GEMMs (a K slice = a sub-problem).  A bijection on [0, total_tiles).
    const int total = ka.total_tiles, x = tile & 7, j = tile >> 3, base = total >> 3, r = total & 7;
    tile = x * base + min(x, r) + j;
  }
  int lo = 0, hi = ka.nprob - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (prob_tile_start(ka.probs, mid) <= tile) lo = mid; else hi = mid - 1;
  }
  P = load_prob(ka.probs, lo);
  if (prob_idx) *prob_idx = lo;
  int mt, nt;
  if (ka.xcd_tiles_m > 0) {
    // XCD-aware map (speed only; correctness never depends on placement).  Blocks b and b+8 share an XCD and its 4 MB
    // L2 (round-robin dispatch), and the persistent stride is a multiple of 8, so tile%8 labels the XCD for the whole
    // walk.  The tile grid is cut into 2 (M) x 4 (N) rectangles, one per XCD: a weight quarter (3 MB at D=1024) stays
    // L2-resident and each A panel is fetched by 4 XCDs instead of 8 -- HBM/Infinity-Cache reads drop ~2.4x vs the
    // plain round-robin order (PMC FETCH_SIZE, profiles/).
    const int x = tile & 7, j = tile >> 3;
    const int sm = (ka.xcd_tiles_m + 1) >> 1, sn = P.tiles_n >> 2;
    mt = (x >> 2) * sm + j / sn;
    nt = (x & 3) * sn + j % sn;
    if (mt >= ka.xcd_tiles_m) return false;
  } else {
    const int local = tile - P.tile_start;
    mt = local / P.tiles_n; nt = local % P.tiles_n;
  }
  c.m0 = mt * BM; c.n0 = nt * BN;
  c.M = P.M; c.N = P.N; c.K = P.K; c.lda = P.lda; c.ldb = P.ldb; c.ldc = P.ldc; c.ldr = P.ldr;
  c.c_off = P.c_off; c.r_off = P.r_off;
  c.klast = P.K > 4 ? ((P.K + 3) & ~3) - 4 : 0;   // last legal float4 start along a K-contiguous row
  return true;
}

// Epilogue of one wave's (TM x TN) 32x32 accumulator tiles.  C/D map of the 32x32 MFMA: col = lane&31,
// row = (r&3) + 8*(r>>2) + 4*(lane>>5).  (row0, col0) = the wave's origin inside the matrix.
// RES_DONE: the residual of EPI_RESIDUAL is already in the accumulators (residual_init below).
template <int EPI, int TM, int TN, bool RES_DONE = false>
__device__ __forceinline__ void epilogue_store(const GemmKArgs& ka, const TileCtx& cur, const f32x16 (&acc)[TM][TN], int row0,
                                               int col0, int li, int lh) {
  if constexpr (EPI == EPI_NONE) {
    // bf16-only output (C == nullptr): neighbouring lanes hold neighbouring columns of the same rows, so the lane pair (c, c + 1)
    // swaps one value per row pair over DPP and each lane stores ONE packed bf16x2 -- the even lane rows r, the odd lane rows r + 1 --
    // half the store instructions of the 2-byte form (kernel-uniform branch; needs even N / ldc / c_off for the 4-byte alignment)
    if (ka.C == nullptr && ka.C16 != nullptr && ((cur.N | cur.ldc) & 1) == 0 && (cur.c_off & 1) == 0) {
      typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
      typedef float f32x2_t __attribute__((ext_vector_type(2)));
      const bool odd = li & 1;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const int col = col0 + tn * 32 + (li & ~1);          // first column of the pair
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const float a0 = acc[tm][tn][r] * ka.alpha, a1 = acc[tm][tn][r + 1] * ka.alpha;
            const float n0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a0), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
            const float n1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a1), 0xB1, 0xF, 0xF, true));
            const f32x2_t v = odd ? f32x2_t{n1, a1} : f32x2_t{a0, n0};
            const int row = row0 + tm * 32 + ((r + (odd ? 1 : 0)) & 3) + 8 * (r >> 2) + 4 * lh;
            if (row < cur.M && col < cur.N)
              *reinterpret_cast<bf16x2_t*>(ka.C16 + cur.c_off + (int64_t)row * cur.ldc + col) = __builtin_convertvector(v, bf16x2_t);
          }
        }
      }
      return;
    }
  }
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int col = col0 + tn * 32 + li;
    if (col >= cur.N) continue;
    float bsum = 0.f;
    if constexpr (EPI == EPI_BIAS_RELU || EPI == EPI_BIAS2 || EPI == EPI_BIAS_RESIDUAL) {
      int g = 0, nl = col;
      if (ka.n_group > 0) { g = col / ka.n_group; nl = col - g * ka.n_group; }
      const float* b0 = g == 0 ? ka.bias0[0] : g == 1 ? ka.bias0[1] : g == 2 ? ka.bias0[2] : ka.bias0[3];
      bsum = b0[nl];
      if constexpr (EPI == EPI_BIAS2) {
        const float* b1 = g == 0 ? ka.bias1[0] : g == 1 ? ka.bias1[1] : g == 2 ? ka.bias1[2] : ka.bias1[3];
        if (b1 != nullptr) bsum += b1[nl];   // single-bias projections pass bias1 = nullptr
      }
    }
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row >= cur.M) continue;
        float v = acc[tm][tn][r];
        float* cp = ka.C + cur.c_off + (int64_t)row * cur.ldc + col;
        if constexpr (EPI == EPI_NONE) v *= ka.alpha;
        if constexpr ((EPI == EPI_RESIDUAL || EPI == EPI_RESIDUAL_MOMENTS) && !RES_DONE) v += ka.R[cur.r_off + (int64_t)row * cur.ldr + col];
        if constexpr (EPI == EPI_BIAS_RELU) {
          v += bsum; v = (v < 0.f) ? 0.f : v;   // NaN-propagating, like torch.relu
          if (ka.drop.thr) v = drop_apply(ka.drop, ka.drop_site, (uint64_t)row * (uint64_t)cur.N + (uint64_t)col, v);
        }
        if constexpr (EPI == EPI_BIAS2) v += bsum;
        if constexpr (EPI == EPI_BIAS_RESIDUAL) {
          v += bsum;
          if (ka.drop.thr) v = drop_apply(ka.drop, ka.drop_site, (uint64_t)row * (uint64_t)cur.N + (uint64_t)col, v);
          v += ka.R[cur.r_off + (int64_t)row * cur.ldr + col];
        }
        if constexpr (EPI == EPI_ACCUM) v = *cp + ka.alpha * v;
        if constexpr (EPI == EPI_NONE) {     // C may be null when only the bf16 form has a reader (kernel-uniform branches)
          if (ka.C) *cp = v;
          if (ka.C16) ka.C16[cur.c_off + (int64_t)row * cur.ldc + col] = __builtin_bit_cast(unsigned short, (__bf16)v);
        } else {
          *cp = v;
        }
      }
    }
  }
}

// Sum over the 16 lanes of a DPP row, left in every lane of the row: four VALU adds with DPP operands (quad xor 1, quad xor 2,
// mirror within 8, mirror within 16) -- no LDS crossbar traffic (a __shfl_xor butterfly is five ds_bpermute round trips).
__device__ __forceinline__ float dpp_row_sum16(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
  return v;
}

// EPI_BIAS_RELU_HEAD: the wave's 64-column slice of relu(acc + bias) is reduced per row to the three moments of the fused
// LayerNorm + head tail and written as ONE float4 per (row, 16-column slot) -- slot = (32-column MFMA tile, 16-lane DPP row): the
// activations themselves never leave registers.  (First version: xor-shuffle butterflies over the 32-lane half, 480 ds_bpermute
// per wave and tile -- the epilogue cost more than storing the tile did.)
// lds_stats: block-shared scratch of >= 128 float2 (the operand staging area, free between two k-loops); m0: the tile's first row.
template <int TM, int TN>
__device__ __forceinline__ void epilogue_head_moments(const GemmKArgs& ka, const TileCtx& cur, const f32x16 (&acc)[TM][TN], int row0,
                                                      int col0, int li, int lh, float2* lds_stats, int m0, int tid) {
  static_assert(TN == 2, "head epilogue: a wave covers one 64-column slot");
  float bias[TN], gw[TN], c1[TN];
  bool colok[TN];
  const bool ln = ka.ln_stats != nullptr;      // kernel-uniform
  if (ln) {   // {mean, rstd} of the tile's 128 rows: one coalesced load into LDS instead of 32 dependent broadcast loads per lane
    __syncthreads();                           // every wave is done with the last k-tile's operands
    if (tid < 128) lds_stats[tid] = reinterpret_cast<const float2*>(ka.ln_stats)[min(m0 + tid, cur.M - 1)];
    __syncthreads();
  }
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int col = col0 + tn * 32 + li;
    colok[tn] = col < cur.N;
    const int cc = colok[tn] ? col : 0;
    bias[tn] = ka.bias0[0][cc];
    gw[tn] = ka.bias1[0][cc] * ka.bias1[1][cc];
    c1[tn] = 0.f;
    if (ln) { c1[tn] = ka.ln_c1[cc]; bias[tn] += ka.ln_c2[cc]; }
  }
  // slots per row: N / 32 (the wave's two 32-column tiles are added in-lane first) x 2 DPP rows of 16 lanes
  const int slots = cur.N >> 5, slot = (col0 >> 5) + (li >> 4);
  float4* part = reinterpret_cast<float4*>(ka.C);
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float s1 = 0.f, s2 = 0.f, s3 = 0.f;
      float mean = 0.f, rstd = 1.f;
      if (ln) {
        const float2 st = lds_stats[row0 - m0 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
        mean = st.x; rstd = st.y;
      }
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        float v = rstd * (acc[tm][tn][r] - mean * c1[tn]) + bias[tn];
        v = (v < 0.f) ? 0.f : v;          // NaN-propagating like torch.relu
        if (!colok[tn]) v = 0.f;
        s1 += v; s2 += v * v; s3 += v * gw[tn];
      }
      s1 = dpp_row_sum16(s1); s2 = dpp_row_sum16(s2); s3 = dpp_row_sum16(s3);
      const int row = row0 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if ((li & 15) == 0 && row < cur.M) part[(int64_t)row * slots + slot] = make_float4(s1, s2, s3, 0.f);
    }
  }
}

// EPI_RESIDUAL_MOMENTS: per row {sum v, sum v^2} of the wave's 64 columns, one float2 per (row, 16-lane DPP row), next to the
// ordinary store of the tile (acc already holds acc + R: residual_init).
template <int TM, int TN>
__device__ __forceinline__ void epilogue_row_moments(const GemmKArgs& ka, const TileCtx& cur, const f32x16 (&acc)[TM][TN], int row0,
                                                     int col0, int li, int lh) {
  static_assert(TN == 1 || TN == 2, "moments epilogue: a wave covers 32 or 64 columns");
  bool colok[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) colok[tn] = col0 + tn * 32 + li < cur.N;
  // TN = 2 (128x128 tiles): the two 32-column tiles are added in-lane, N / 32 slots per row; TN = 1 (64x64 tiles): N / 16 slots.
  // Grouped (per-video) launches: `row` is relative to the sub-problem, the moments are indexed by the row of the packed matrix.
  const int slots = TN == 2 ? cur.N >> 5 : cur.N >> 4, slot = (TN == 2 ? col0 >> 5 : col0 >> 4) + (li >> 4);
  float2* part = reinterpret_cast<float2*>(ka.moments) + (cur.c_off / cur.ldc) * slots;
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const float v = colok[tn] ? acc[tm][tn][r] : 0.f;
        s1 += v; s2 += v * v;
      }
      s1 = dpp_row_sum16(s1); s2 = dpp_row_sum16(s2);
      const int row = row0 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if ((li & 15) == 0 && row < cur.M) part[(int64_t)row * slots + slot] = make_float2(s1, s2);
    }
  }
}

// EPI_RESIDUAL with the residual read at the START of a tile: the accumulators begin at R instead of zero, so the 64 KB read of
// a 128 x 128 tile overlaps the first operand loads instead of joining the store burst at the end (where every co-resident
// block of a single-round launch reads and writes at once).  Same C/D map as epilogue_store; rows / columns outside the problem
// start at zero.
template <int TM, int TN>
__device__ __forceinline__ void residual_init(const GemmKArgs& ka, const TileCtx& cur, f32x16 (&acc)[TM][TN], int row0, int col0,
                                              int li, int lh) {
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int col = col0 + tn * 32 + li;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        acc[tm][tn][r] = (col < cur.N && row < cur.M) ? ka.R[cur.r_off + (int64_t)row * cur.ldr + col] : 0.f;
      }
    }
  }
}

// gemm_lean.hip: 64x64 exact-fp32 tiles with a VALU-free main loop (NT / NN, plain epilogue) for the per-video products
int launch_gemm_lean(GemmLayout layout, const GemmKArgs& ka, int tiles, hipStream_t stream, int sk = 0);   // sk: the small-batch (in-launch split-K) instances
// gemm_b16.hip: bf16 operands in HBM (A and B[0] point at bf16 data), fp32 accumulate / output; 128x128 tiles or (192 | 256) x 256
int launch_gemm_b16(GemmLayout layout, GemmEpi epi, const GemmKArgs& ka, int tiles, int wide, hipStream_t stream);
// gemm_split.hip: the register-staged kernel with fp32 operands split into bf16 planes on their way into LDS
// (SUMK_PRECISION_BF16 / BF16X3 / BF16X6).
int launch_gemm_split(int precision, GemmLayout layout, GemmEpi epi, const GemmKArgs& ka, int tiles, int cfg, hipStream_t stream);

}  // namespace sumk
