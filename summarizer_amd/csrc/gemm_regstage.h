// Register-staged MFMA GEMM kernel template of libsumk.so (see gemm_f32.hip for the design notes).  Header so that the
// exact-fp32 instances (gemm_f32.hip) and the bf16-plane instances (gemm_split.hip) compile as separate translation units.
#pragma once
#include "gemm_device.h"
#include <algorithm>
#include <cstdlib>

namespace sumk {

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// one plane's fragment (8 consecutive k of this lane's row) of a [k][row] bf16 image: 2 transposing reads
__device__ __forceinline__ bf16x8 tr_frag(const char* p, int pitch) {
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
  const s16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p));
  const s16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p + 4 * pitch));
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
}

// __launch_bounds__(256, 3): three blocks per CU (152 VGPRs, 36 KB LDS for the 128x128 tile).
// PERSISTENT tile loop: the grid is at most (resident slots) blocks and block b walks tiles b, b+grid, ...  While the
// LAST k-tile of a tile is being multiplied, the block already decodes its next tile and issues that tile's first global
// loads, so the epilogue stores of tile i and the prologue latency of tile i+1 overlap instead of leaving the MFMA pipe
// idle (measured before: ~19k idle cycles per tile per SIMD at K=1024, because co-resident blocks run in lockstep).
// X3: "bf16x3" arithmetic.  Each fp32 operand is split on its way into LDS into hi = bf16(x) and
// lo = bf16(x - hi); the product is accumulated in fp32 as  lo.hi + hi.lo + hi.hi  with v_mfma_f32_32x32x16_bf16 (16x the
// fp32 MFMA rate, 3 MFMAs per 16-deep k step instead of 8 fp32 ones): the dropped lo.lo term is ~2^-16 relative, which keeps
// VASNet scores within ~1e-5 of the fp32 path (tests/probes/bf16x3_emulation.py; the 1e-4 gate holds, plain bf16 misses it by
// 20-100x).  The LDS row becomes [hi: BK bf16 | lo: BK bf16] -- the same BK*4 bytes and the same +16 B pad, so the
// conflict-free ds_read_b128 argument is unchanged; a lane's 16-B fragment is 8 consecutive k of one plane.
// An operand whose rows are NOT K-contiguous in memory (the "MC" side of NN / TN) keeps its natural [k][row] order in LDS,
// as two bf16 planes with a (2*BT + 64)-byte pitch, and is read with ds_read_b64_tr_b16 -- gfx950's transposing LDS read
// hands lane (i, h) the 4 consecutive k of column i, two reads per 8-k fragment (4 k-rows x 64 B per 32-lane half land on
// 4 distinct 16-bank groups with that pitch).  The kernel has no divergent lanes in its main loop (EXEC all ones, as the
// instruction requires).
// NS = number of bf16 planes an fp32 operand is split into: 0 = none (exact fp32 MFMA), 1 = plain "bf16" (x1 = bf16(x) only:
// one MFMA per product, the mixed-precision training arithmetic), 2 = "bf16x3" (above), 3 = "bf16x6":
// x = x1 + x2 + x3 EXACTLY (3 x 8 significand bits = fp32's 24), products x_i y_j for i + j <= 4 -- six bf16 MFMAs per 16 k
// instead of eight fp32 ones at 1/16 of their cost; the three dropped terms are <= 3 * 2^-24 relative, i.e. the result is
// fp32-grade (the fp32 tolerances of tests/test_gpu_vasnet.py::test_gemm_layouts_vs_float64 hold).  The KC image row
// becomes [x1 | x2 | x3] = 6 BK bytes + 16 B pad (pitch 52 dwords at BK = 32: 16 rows still hit 16 distinct 16-byte slots).
// LEAN (NT, exact fp32, K % BK == 0, one B group per tile): operands are fetched with buffer loads -- per-thread byte offset fixed per
// tile, the k advance a scalar soffset -- so the k-loop carries no address arithmetic.  v_mfma_f32_32x32x2_f32 shares its SIMD's
// vector ALU (gemm_lean.hip header): the ~28 64-bit pointer instructions per k-tile of the plain form cost 3-4 % of the matrix rate.
template <int BM, int BN, int BK, bool A_KC, bool B_KC, int EPI, int NS = 0, bool LEAN = false>
__global__ __launch_bounds__(256, 3) void gemm_f32_kernel(GemmKArgs ka) {
  static_assert(!LEAN || (A_KC && B_KC && NS == 0), "LEAN: NT fp32 instances only");
  constexpr bool X3 = NS > 0;
  static_assert(NS >= 0 && NS <= 3, "operand split: 0 (fp32), 1 (plain bf16), 2 or 3 bf16 planes");
  constexpr int KC_PITCH = (NS == 0 ? BK : NS * BK / 2) + 4;   // +4 floats: conflict-free ds_read_b128 (pitch 20 / 36 / 52 dwords at BK = 32: 16 rows hit 16 distinct slots)
  constexpr int TPK = BK / 4, RPP = 256 / TPK;  // KC image: threads per row, rows covered per pass
  constexpr int WTM = BM / 2, WTN = BN / 2;   // wave tile
  constexpr int TM = WTM / 32, TN = WTN / 32; // MFMA tiles per wave along M / N
  constexpr int NLDA = BM * BK / 1024, NLDB = BN * BK / 1024;  // float4 loads per thread per operand per k-tile
  constexpr int MCP_A = 2 * BM + 64, MCP_B = 2 * BN + 64;   // X3: byte pitch of one k-row of a [k][row] bf16 plane
  constexpr int A_ELEMS = A_KC ? BM * KC_PITCH : (X3 ? NS * BK * MCP_A / 4 : BK * BM);
  constexpr int B_ELEMS = B_KC ? BN * KC_PITCH : (X3 ? NS * BK * MCP_B / 4 : BK * BN);
  constexpr int STAGE = A_ELEMS + B_ELEMS;
  __shared__ __attribute__((aligned(16))) float lds[STAGE];
  float* sA = lds;
  float* sB = lds + A_ELEMS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  constexpr int TPRA = BM / 4, TPRB = BN / 4;              // MC image: threads per k-row
  constexpr int KROWSA = 256 / TPRA, KROWSB = 256 / TPRB;  // MC image: k-rows covered per pass
  const int kq4 = (tid % TPK) * 4;     // KC image: this thread's k offset inside a k-tile

  // Per-thread global source descriptors of the tile whose operands are being LOADED.  Loads are UNCONDITIONAL:
  // addresses are clamped into the operand so the compiler can issue all of a k-tile's 16-B loads back to back and wait
  // for them only where they are written to LDS.  Rows/columns past M or N are clamped, not zeroed -- they only feed
  // output rows/columns the epilogue never stores.  Only the K tail must contribute zeros: masked before the LDS write.
  const float* pa[NLDA];
  const float* pb[NLDB];
  int voa[NLDA], vob[NLDB];      // LEAN: byte offsets of this thread's rows (+ its k chunk) from TileCtx::ab / bb

  // ---- tile decode (wave-uniform scalar work, gemm_device.h) + this thread's row pointers
  auto setup = [&](int tile, TileCtx& c) -> bool {
    GemmProb P;
    if (!decode_tile<BM, BN>(ka, tile, c, P)) return false;
    if constexpr (LEAN) {
      c.ab = ka.A + P.a_off;
      int g = 0, ng = 0;                              // the tile's rows of B lie in ONE group (host: n_group % BN == 0)
      if (ka.n_group > 0) { g = c.n0 / ka.n_group; ng = g * ka.n_group; }
      c.bb = (g == 0 ? ka.B[0] : g == 1 ? ka.B[1] : g == 2 ? ka.B[2] : ka.B[3]) + P.b_off;
#pragma unroll
      for (int p = 0; p < NLDA; ++p) voa[p] = (min(c.m0 + tid / TPK + RPP * p, P.M - 1) * P.lda + kq4) * 4;
#pragma unroll
      for (int p = 0; p < NLDB; ++p) vob[p] = ((min(c.n0 + tid / TPK + RPP * p, P.N - 1) - ng) * P.ldb + kq4) * 4;
      return true;
    }
    if constexpr (A_KC) {
#pragma unroll
      for (int p = 0; p < NLDA; ++p) {
        int r = min(c.m0 + tid / TPK + RPP * p, P.M - 1);
        pa[p] = ka.A + P.a_off + (int64_t)r * P.lda;
      }
    } else {
      int col = c.m0 + (tid % TPRA) * 4;
      col = col < P.M ? col : 0;
#pragma unroll
      for (int p = 0; p < NLDA; ++p) pa[p] = ka.A + P.a_off + col;
    }
    if constexpr (B_KC) {
#pragma unroll
      for (int p = 0; p < NLDB; ++p) {
        int n = min(c.n0 + tid / TPK + RPP * p, P.N - 1);
        int g = 0, nl = n;
        if (ka.n_group > 0) { g = n / ka.n_group; nl = n - g * ka.n_group; }
        const float* bg = g == 0 ? ka.B[0] : g == 1 ? ka.B[1] : g == 2 ? ka.B[2] : ka.B[3];
        pb[p] = bg + P.b_off + (int64_t)nl * P.ldb;
      }
    } else {
      int col = c.n0 + (tid % TPRB) * 4;
      col = col < P.N ? col : 0;
#pragma unroll
      for (int p = 0; p < NLDB; ++p) pb[p] = ka.B[0] + P.b_off + col;
    }
    return true;
  };

  float4 ra[NLDA], rb[NLDB];

  auto gload = [&](const TileCtx& c, int k0) {
    if constexpr (LEAN) {
      typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
      const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(c.ab), (short)0, 0x7FFFFFFF, 0x00020000);
      const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(c.bb), (short)0, 0x7FFFFFFF, 0x00020000);
#pragma unroll
      for (int p = 0; p < NLDA; ++p) ra[p] = __builtin_bit_cast(float4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(rA, voa[p], k0 * 4, 0));
#pragma unroll
      for (int p = 0; p < NLDB; ++p) rb[p] = __builtin_bit_cast(float4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(rB, vob[p], k0 * 4, 0));
      return;
    }
#pragma unroll
    for (int p = 0; p < NLDA; ++p) {
      if constexpr (A_KC) ra[p] = ldg4(pa[p] + min(k0 + kq4, c.klast));
      else ra[p] = ldg4(pa[p] + (int64_t)min(k0 + tid / TPRA + KROWSA * p, c.K - 1) * c.lda);
    }
#pragma unroll
    for (int p = 0; p < NLDB; ++p) {
      if constexpr (B_KC) rb[p] = ldg4(pb[p] + min(k0 + kq4, c.klast));
      else rb[p] = ldg4(pb[p] + (int64_t)min(k0 + tid / TPRB + KROWSB * p, c.K - 1) * c.ldb);
    }
  };
  // zero what lies past K (wave-uniform branch: only the last k-tile of a ragged K pays for it)
  auto ktail = [&](int K, int k0) {
    if (k0 + BK <= K) return;
#pragma unroll
    for (int p = 0; p < NLDA; ++p) {
      if constexpr (A_KC) {
        int k = k0 + kq4;
        if (k >= K) ra[p].x = 0.f;
        if (k + 1 >= K) ra[p].y = 0.f;
        if (k + 2 >= K) ra[p].z = 0.f;
        if (k + 3 >= K) ra[p].w = 0.f;
      } else {
        if (k0 + tid / TPRA + KROWSA * p >= K) ra[p] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int p = 0; p < NLDB; ++p) {
      if constexpr (B_KC) {
        int k = k0 + kq4;
        if (k >= K) rb[p].x = 0.f;
        if (k + 1 >= K) rb[p].y = 0.f;
        if (k + 2 >= K) rb[p].z = 0.f;
        if (k + 3 >= K) rb[p].w = 0.f;
      } else {
        if (k0 + tid / TPRB + KROWSB * p >= K) rb[p] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  // planes p = 0 .. NS-1 of this thread's 4 values: x_p = bf16(remainder), remainder -= x_p  (each subtraction is exact)
  auto split_planes = [&](float4 v, bf16x4 (&pl)[3]) {
    f32x4 r = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < (NS > 0 ? NS : 1); ++q) {
      pl[q] = __builtin_convertvector(r, bf16x4);
      if (q + 1 < NS) r = r - __builtin_convertvector(pl[q], f32x4);
    }
  };
  auto split_store = [&](float* row, float4 v) {   // row: LDS row start; this thread's 4 k values -> planes [x1 | x2 (| x3)]
    bf16x4 pl[3];
    split_planes(v, pl);
    char* r8 = reinterpret_cast<char*>(row) + 2 * kq4;
#pragma unroll
    for (int q = 0; q < (NS > 0 ? NS : 1); ++q) *reinterpret_cast<bf16x4*>(r8 + q * 2 * BK) = pl[q];
  };
  auto split_store_mc = [&](float* img, int krow, int col4, int pitch, float4 v) {   // 4 consecutive rows at one k
    bf16x4 pl[3];
    split_planes(v, pl);
    char* r8 = reinterpret_cast<char*>(img) + krow * pitch + 2 * col4;
#pragma unroll
    for (int q = 0; q < (NS > 0 ? NS : 1); ++q) *reinterpret_cast<bf16x4*>(r8 + q * BK * pitch) = pl[q];
  };
  auto swrite = [&](float* sA, float* sB) {
    if constexpr (X3) {
#pragma unroll
      for (int p = 0; p < NLDA; ++p) {
        if constexpr (A_KC) split_store(&sA[(tid / TPK + RPP * p) * KC_PITCH], ra[p]);
        else split_store_mc(sA, tid / TPRA + KROWSA * p, (tid % TPRA) * 4, MCP_A, ra[p]);
      }
#pragma unroll
      for (int p = 0; p < NLDB; ++p) {
        if constexpr (B_KC) split_store(&sB[(tid / TPK + RPP * p) * KC_PITCH], rb[p]);
        else split_store_mc(sB, tid / TPRB + KROWSB * p, (tid % TPRB) * 4, MCP_B, rb[p]);
      }
      return;
    }
#pragma unroll
    for (int p = 0; p < NLDA; ++p) {
      if constexpr (A_KC) *reinterpret_cast<float4*>(&sA[(tid / TPK + RPP * p) * KC_PITCH + kq4]) = ra[p];
      else *reinterpret_cast<float4*>(&sA[(tid / TPRA + KROWSA * p) * BM + (tid % TPRA) * 4]) = ra[p];
    }
#pragma unroll
    for (int p = 0; p < NLDB; ++p) {
      if constexpr (B_KC) *reinterpret_cast<float4*>(&sB[(tid / TPK + RPP * p) * KC_PITCH + kq4]) = rb[p];
      else *reinterpret_cast<float4*>(&sB[(tid / TPRB + KROWSB * p) * BN + (tid % TPRB) * 4]) = rb[p];
    }
  };

  // fragment reads + MFMAs of ONE k-tile whose operand images start at sA / sB
  // (half = 0 / 1: only the first / second half of the k-tile; -1: all of it)
  auto compute = [&](const float* sA, const float* sB, f32x16 (&acc)[TM][TN], int half = -1) {
      if constexpr (X3) {
        constexpr int NP = NS > 0 ? NS : 1;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
          if (half >= 0 && (ks >= BK / 32) != (half == 1)) continue;
          bf16x8 af[NP][TM], bf[NP][TN];      // [plane][tile]: 8 consecutive k of this lane's row
#pragma unroll
          for (int t = 0; t < TM; ++t) {
            if constexpr (A_KC) {
              const char* rp = reinterpret_cast<const char*>(&sA[(wm * WTM + t * 32 + li) * KC_PITCH]) + 32 * ks + 16 * lh;
#pragma unroll
              for (int q = 0; q < NP; ++q) af[q][t] = *reinterpret_cast<const bf16x8*>(rp + q * 2 * BK);
            } else {
              const char* rp = reinterpret_cast<const char*>(sA) + (16 * ks + 8 * lh + ((lane & 15) >> 2)) * MCP_A +
                               2 * (wm * WTM + t * 32 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3));
#pragma unroll
              for (int q = 0; q < NP; ++q) af[q][t] = tr_frag(rp + q * BK * MCP_A, MCP_A);
            }
          }
#pragma unroll
          for (int t = 0; t < TN; ++t) {
            if constexpr (B_KC) {
              const char* rp = reinterpret_cast<const char*>(&sB[(wn * WTN + t * 32 + li) * KC_PITCH]) + 32 * ks + 16 * lh;
#pragma unroll
              for (int q = 0; q < NP; ++q) bf[q][t] = *reinterpret_cast<const bf16x8*>(rp + q * 2 * BK);
            } else {
              const char* rp = reinterpret_cast<const char*>(sB) + (16 * ks + 8 * lh + ((lane & 15) >> 2)) * MCP_B +
                               2 * (wn * WTN + t * 32 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3));
#pragma unroll
              for (int q = 0; q < NP; ++q) bf[q][t] = tr_frag(rp + q * BK * MCP_B, MCP_B);
            }
          }
#pragma unroll
          for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {   // smallest terms first: planes (i, j) with i + j descending
#pragma unroll
              for (int sum = NP; sum >= 0; --sum)          // i + j = sum, 0-based planes; keep i + j <= NP - 1 ... plus (NS = 2) nothing more
#pragma unroll
                for (int i = NP - 1; i >= 0; --i) {
                  const int j = sum - i;
                  if (j < 0 || j >= NP || i + j > NP - 1) continue;
                  acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][tm], bf[j][tn], acc[tm][tn], 0, 0, 0);
                }
            }
        }
      } else {
#pragma unroll
      for (int kk = 0; kk < BK / 8; ++kk) {
        if (half >= 0 && (kk >= BK / 16) != (half == 1)) continue;
        float av[TM][4], bv[TN][4];
#pragma unroll
        for (int t = 0; t < TM; ++t) {
          if constexpr (A_KC) {
            float4 v = *reinterpret_cast<const float4*>(&sA[(wm * WTM + t * 32 + li) * KC_PITCH + kk * 8 + 4 * lh]);
            av[t][0] = v.x; av[t][1] = v.y; av[t][2] = v.z; av[t][3] = v.w;
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) av[t][j] = sA[(kk * 8 + 4 * lh + j) * BM + wm * WTM + t * 32 + li];
          }
        }
#pragma unroll
        for (int t = 0; t < TN; ++t) {
          if constexpr (B_KC) {
            float4 v = *reinterpret_cast<const float4*>(&sB[(wn * WTN + t * 32 + li) * KC_PITCH + kk * 8 + 4 * lh]);
            bv[t][0] = v.x; bv[t][1] = v.y; bv[t][2] = v.z; bv[t][3] = v.w;
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[t][j] = sB[(kk * 8 + 4 * lh + j) * BN + wn * WTN + t * 32 + li];
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
              acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tm][j], bv[tn][j], acc[tm][tn], 0, 0, 0);
      }
      }
  };

  int tile = blockIdx.x;
  if (tile >= ka.total_tiles) return;
  TileCtx cur, nxt;
  if (!setup(tile, cur)) return;   // (remapped walk: a rectangle's tiles are exhausted in increasing order)
  gload(cur, 0);
  unsigned long long t_begin = 0, t_k = 0, t_e = 0, n_t = 0;
#ifdef SUMK_DIAG
  const unsigned long long rt_begin = __builtin_amdgcn_s_memrealtime();
#endif
  if (ka.dbg & 2) t_begin = __builtin_amdgcn_s_memtime();

  while (true) {
    unsigned long long ta = 0;
    if (ka.dbg & 2) ta = __builtin_amdgcn_s_memtime();
    f32x16 acc[TM][TN];
    if constexpr (EPI == EPI_RESIDUAL || EPI == EPI_RESIDUAL_MOMENTS) {
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
    bool has_next = false;
    const int K = cur.K;
    {
      // all k-tiles but the last: nothing but "barrier, LDS write, barrier, next loads, MFMAs"; the last one is peeled -- it carries
      // the K-tail masks and fetches the NEXT tile's first operands under this tile's last MFMAs (its decode would otherwise sit in
      // the loop as a merge of two register sets, copied every k-tile)
      int k0 = 0;
      for (; k0 + BK < K; k0 += BK) {
        __syncthreads();
        swrite(sA, sB);
        __syncthreads();
        gload(cur, k0 + BK);
        compute(sA, sB, acc);
      }
      __syncthreads();
      ktail(K, k0);
      swrite(sA, sB);
      __syncthreads();
      has_next = next_tile < ka.total_tiles;
      if (has_next) has_next = setup(next_tile, nxt);
      if (has_next) gload(nxt, 0);
      compute(sA, sB, acc);
    }

    // ---- epilogue of `cur` (gemm_device.h)
    unsigned long long tb = 0;
    if (ka.dbg & 2) tb = __builtin_amdgcn_s_memtime();
    if constexpr (EPI == EPI_BIAS_RELU_HEAD) {
      epilogue_head_moments<TM, TN>(ka, cur, acc, cur.m0 + wm * WTM, cur.n0 + wn * WTN, li, lh, reinterpret_cast<float2*>(lds), cur.m0, tid);
    } else if constexpr (EPI == EPI_RESIDUAL_MOMENTS) {
      epilogue_row_moments<TM, TN>(ka, cur, acc, cur.m0 + wm * WTM, cur.n0 + wn * WTN, li, lh);
      epilogue_store<EPI, TM, TN, true>(ka, cur, acc, cur.m0 + wm * WTM, cur.n0 + wn * WTN, li, lh);
    } else
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
#ifdef SUMK_DIAG
    if (ka.dbg & 4) {   // second record, behind the 2048 first ones: k-loop shares, wall-clock window (100 MHz), placement
      unsigned long long* q = ka.dbg_buf + (size_t)2048 * 4 + (size_t)blockIdx.x * 8;
      q[0] = 0; q[1] = 0; q[2] = 0; q[3] = t_k; q[4] = rt_begin; q[5] = __builtin_amdgcn_s_memrealtime();
      q[6] = __builtin_amdgcn_s_getreg((31 << 11) | 4); q[7] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
#endif
  }
}

template <int BM, int BN, int BK, bool A_KC, bool B_KC, int X3 = 0>
static int launch_epi(GemmEpi epi, const GemmKArgs& ka, int tiles, hipStream_t s) {
  // persistent grid: no more blocks than can be resident (256 CUs x blocks/CU for this tile's LDS/VGPR footprint);
  // every block then loops over tiles  b, b+grid, ...
  constexpr int occ = (BM == 128 && BN == 128) ? 3 : (BM == 128 ? 4 : (BK == 64 ? 4 : 8));
  static const bool persist = !(SUMK_TUNE_ENV("SUMK_PERSIST") && SUMK_TUNE_ENV("SUMK_PERSIST")[0] == '0');
  dim3 grid(persist ? std::min(tiles, 256 * occ) : tiles), block(256);
  if constexpr (BM == 128 && BN == 128 && BK == 32 && A_KC && B_KC && X3 == 0) {
    if (ka.lean) {   // buffer-load instances (launch_gemm checked: one problem, NT, fp32, K % 32 == 0, one B group per tile)
#define SUMK_LEAN_CASE(E) case E: hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BK, true, true, E, 0, true>), grid, block, 0, s, ka); return SUMK_OK;
      switch (epi) {
        SUMK_LEAN_CASE(EPI_NONE) SUMK_LEAN_CASE(EPI_RESIDUAL) SUMK_LEAN_CASE(EPI_BIAS_RELU) SUMK_LEAN_CASE(EPI_BIAS2)
        SUMK_LEAN_CASE(EPI_BIAS_RESIDUAL) SUMK_LEAN_CASE(EPI_BIAS_RELU_HEAD) SUMK_LEAN_CASE(EPI_RESIDUAL_MOMENTS)
        default: break;      // (EPI_ACCUM: the plain instance below)
      }
#undef SUMK_LEAN_CASE
    }
  }
  switch (epi) {
    case EPI_NONE: hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BK, A_KC, B_KC, EPI_NONE, X3>), grid, block, 0, s, ka); break;
    case EPI_RESIDUAL: hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BK, A_KC, B_KC, EPI_RESIDUAL, X3>), grid, block, 0, s, ka); break;
    case EPI_BIAS_RELU: hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BK, A_KC, B_KC, EPI_BIAS_RELU, X3>), grid, block, 0, s, ka); break;
    case EPI_BIAS2: hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BK, A_KC, B_KC, EPI_BIAS2, X3>), grid, block, 0, s, ka); break;
    case EPI_ACCUM: hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BK, A_KC, B_KC, EPI_ACCUM, X3>), grid, block, 0, s, ka); break;
    case EPI_BIAS_RESIDUAL: hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BK, A_KC, B_KC, EPI_BIAS_RESIDUAL, X3>), grid, block, 0, s, ka); break;
    case EPI_BIAS_RELU_HEAD:
      if constexpr (BM == 128 && BN == 128 && BK == 32 && A_KC && B_KC) {
        hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BK, A_KC, B_KC, EPI_BIAS_RELU_HEAD, X3>), grid, block, 0, s, ka); break;
      } else { set_error("gemm: the head epilogue exists for 128x128 NT tiles only"); return SUMK_ERR_ARG; }
    case EPI_RESIDUAL_MOMENTS:
      if constexpr ((BM == 128 && BN == 128 && BK == 32 && A_KC && B_KC) || (BM == 64 && BN == 64 && BK == 32 && A_KC && !B_KC)) {
        hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BK, A_KC, B_KC, EPI_RESIDUAL_MOMENTS, X3>), grid, block, 0, s, ka); break;
      } else { set_error("gemm: the moments epilogue exists for 128x128 NT and 64x64 NN tiles only"); return SUMK_ERR_ARG; }
    default: set_error("gemm: bad epilogue %d", (int)epi); return SUMK_ERR_ARG;
  }
  return SUMK_OK;
}

template <int BM, int BN, int BK, int X3 = 0>
static int launch_layout(GemmLayout layout, GemmEpi epi, const GemmKArgs& ka, int tiles, hipStream_t s) {
  if (layout == GEMM_NT) return launch_epi<BM, BN, BK, true, true, X3>(epi, ka, tiles, s);
  if (layout == GEMM_NN) return launch_epi<BM, BN, BK, true, false, X3>(epi, ka, tiles, s);
  return launch_epi<BM, BN, BK, false, false, X3>(epi, ka, tiles, s);
}


}  // namespace sumk
