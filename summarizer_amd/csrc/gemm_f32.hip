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
// NS = number of bf16 planes an fp32 operand is split into: 0 = none (exact fp32 MFMA), 2 = "bf16x3" (above), 3 = "bf16x6":
// x = x1 + x2 + x3 EXACTLY (3 x 8 significand bits = fp32's 24), products x_i y_j for i + j <= 4 -- six bf16 MFMAs per 16 k
// instead of eight fp32 ones at 1/16 of their cost; the three dropped terms are <= 3 * 2^-24 relative, i.e. the result is
// fp32-grade (the fp32 tolerances of tests/test_gpu_vasnet.py::test_gemm_layouts_vs_float64 hold).  The KC image row
// becomes [x1 | x2 | x3] = 6 BK bytes + 16 B pad (pitch 52 dwords at BK = 32: 16 rows still hit 16 distinct 16-byte slots).
template <int BM, int BN, int BK, bool A_KC, bool B_KC, int EPI, int NS = 0>
__global__ __launch_bounds__(256, 3) void gemm_f32_kernel(GemmKArgs ka) {
  constexpr bool X3 = NS > 0;
  static_assert(NS == 0 || NS == 2 || NS == 3, "operand split: 0, 2 or 3 bf16 planes");
  constexpr int KC_PITCH = (NS == 3 ? 3 * BK / 2 : BK) + 4;   // +4 floats: conflict-free ds_read_b128 (pitch 36 / 52 / 68 dwords: 16 rows hit 16 distinct slots)
  constexpr int TPK = BK / 4, RPP = 256 / TPK;  // KC image: threads per row, rows covered per pass
  constexpr int WTM = BM / 2, WTN = BN / 2;   // wave tile
  constexpr int TM = WTM / 32, TN = WTN / 32; // MFMA tiles per wave along M / N
  constexpr int NLDA = BM * BK / 1024, NLDB = BN * BK / 1024;  // float4 loads per thread per operand per k-tile
  constexpr int MCP_A = 2 * BM + 64, MCP_B = 2 * BN + 64;   // X3: byte pitch of one k-row of a [k][row] bf16 plane
  constexpr int A_ELEMS = A_KC ? BM * KC_PITCH : (X3 ? NS * BK * MCP_A / 4 : BK * BM);
  constexpr int B_ELEMS = B_KC ? BN * KC_PITCH : (X3 ? NS * BK * MCP_B / 4 : BK * BN);
  __shared__ __attribute__((aligned(16))) float lds[A_ELEMS + B_ELEMS];
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

  // ---- tile decode (wave-uniform scalar work, gemm_device.h) + this thread's row pointers
  auto setup = [&](int tile, TileCtx& c) -> bool {
    GemmProb P;
    if (!decode_tile<BM, BN>(ka, tile, c, P)) return false;
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
  auto swrite = [&]() {
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

  int tile = blockIdx.x;
  if (tile >= ka.total_tiles) return;
  TileCtx cur, nxt;
  if (!setup(tile, cur)) return;   // (remapped walk: a rectangle's tiles are exhausted in increasing order)
  gload(cur, 0);
  unsigned long long t_begin = 0, t_k = 0, t_e = 0, n_t = 0;
  if (ka.dbg & 2) t_begin = __builtin_amdgcn_s_memtime();

  while (true) {
    unsigned long long ta = 0;
    if (ka.dbg & 2) ta = __builtin_amdgcn_s_memtime();
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int next_tile = tile + gridDim.x;
    bool has_next = next_tile < ka.total_tiles;
    const int K = cur.K;
    for (int k0 = 0; k0 < K; k0 += BK) {
      __syncthreads();
      ktail(K, k0);
      swrite();
      __syncthreads();
      if (k0 + BK < K) {
        gload(cur, k0 + BK);
      } else if (has_next) {   // last k-tile: fetch the NEXT tile's first operands under this tile's last 64 MFMAs
        has_next = setup(next_tile, nxt);
        if (has_next) gload(nxt, 0);
      }
      if constexpr (X3) {
        constexpr int NP = NS > 0 ? NS : 1;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
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
    }

    // ---- epilogue of `cur` (gemm_device.h)
    unsigned long long tb = 0;
    if (ka.dbg & 2) tb = __builtin_amdgcn_s_memtime();
    if (!(ka.dbg & 1) || acc[0][0][0] == 12345.f)
    epilogue_store<EPI, TM, TN>(ka, cur, acc, cur.m0 + wm * WTM, cur.n0 + wn * WTN, li, lh);
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

template <int BM, int BN, int BK, bool A_KC, bool B_KC, int X3 = 0>
static int launch_epi(GemmEpi epi, const GemmKArgs& ka, int tiles, hipStream_t s) {
  // persistent grid: no more blocks than can be resident (256 CUs x blocks/CU for this tile's LDS/VGPR footprint);
  // every block then loops over tiles  b, b+grid, ...
  constexpr int occ = (BM == 128 && BN == 128) ? 3 : (BM == 128 ? 4 : (BK == 64 ? 4 : 8));
  static const bool persist = !(getenv("SUMK_PERSIST") && getenv("SUMK_PERSIST")[0] == '0');
  dim3 grid(persist ? std::min(tiles, 256 * occ) : tiles), block(256);
  switch (epi) {
    case EPI_NONE: hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BK, A_KC, B_KC, EPI_NONE, X3>), grid, block, 0, s, ka); break;
    case EPI_RESIDUAL: hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BK, A_KC, B_KC, EPI_RESIDUAL, X3>), grid, block, 0, s, ka); break;
    case EPI_BIAS_RELU: hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BK, A_KC, B_KC, EPI_BIAS_RELU, X3>), grid, block, 0, s, ka); break;
    case EPI_BIAS2: hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BK, A_KC, B_KC, EPI_BIAS2, X3>), grid, block, 0, s, ka); break;
    case EPI_ACCUM: hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BK, A_KC, B_KC, EPI_ACCUM, X3>), grid, block, 0, s, ka); break;
    case EPI_BIAS_RESIDUAL: hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BK, A_KC, B_KC, EPI_BIAS_RESIDUAL, X3>), grid, block, 0, s, ka); break;
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

// one 64 KB stamp buffer per process, allocated on first use under SUMK_GEMM_DBG=2
static unsigned long long* gemm_stamp_buffer() {
  static unsigned long long* buf = nullptr;
  if (!buf && hipMalloc(&buf, 2048 * 4 * sizeof(unsigned long long)) != hipSuccess) buf = nullptr;
  return buf;
}

int launch_gemm(GemmLayout layout, GemmEpi epi, const GemmLaunch& g, hipStream_t stream) {
  SUMK_ARG(g.A && g.B[0] && g.C && g.probs, "gemm: null operand");
  if (g.total_tiles <= 0) return SUMK_OK;
  GemmKArgs ka;
  ka.A = g.A;
  for (int i = 0; i < 4; ++i) { ka.B[i] = g.B[i]; ka.bias0[i] = g.bias0[i]; ka.bias1[i] = g.bias1[i]; }
  ka.C = g.C; ka.R = g.R; ka.probs = g.probs; ka.nprob = g.nprob; ka.n_group = g.n_group; ka.alpha = g.alpha;
  ka.total_tiles = g.total_tiles; ka.xcd_tiles_m = 0;
  static const int dbg = getenv("SUMK_GEMM_DBG") ? atoi(getenv("SUMK_GEMM_DBG")) : 0;
  ka.dbg = dbg; ka.dbg_buf = nullptr;
  if (dbg & 2) ka.dbg_buf = gemm_stamp_buffer();   // diagnostic only (never on a product path)
  ka.drop.seed = g.drop_seed; ka.drop.thr = g.drop_thr; ka.drop.scale = g.drop_scale; ka.drop_site = g.drop_site;
  if (g.prof_tag >= 0) prof_begin(g.prof_tag, stream);
  prof_begin(SUMK_PROF_GEMM_ALL, stream);
  static const bool xcd_map = !(getenv("SUMK_XCD_MAP") && getenv("SUMK_XCD_MAP")[0] == '0');
  if (xcd_map && g.nprob == 1 && g.xcd_M > 0) {
    const int tm = (g.xcd_M + gemm_tile_m(g.small_tile) - 1) / gemm_tile_m(g.small_tile);
    const int tn = (g.xcd_N + gemm_tile_n(g.small_tile) - 1) / gemm_tile_n(g.small_tile);
    if (tn % 4 == 0 && tm >= 16) { ka.xcd_tiles_m = tm; ka.total_tiles = 8 * ((tm + 1) / 2) * (tn / 4); }
  }
  int rc;
  if (g.precision == SUMK_PRECISION_FP32 && !g.no_dma && gemm_dma_enabled()) {   // opt-in (SUMK_GEMM_DMA=1): LDS-DMA staging (gemm_dma.hip)
    rc = launch_gemm_dma(layout, epi, ka, ka.total_tiles, g.small_tile, stream);
    prof_end(SUMK_PROF_GEMM_ALL, stream);
    if (g.prof_tag >= 0) prof_end(g.prof_tag, stream);
    if (rc != SUMK_OK) return rc;
    SUMK_HIP(hipGetLastError());
    return SUMK_OK;
  }
  // BK = 64 for the 64x64 tile measured no better than BK = 32 on S-TVSum (8.64 vs 8.68 M frames/s): kept selectable
  static const bool bk64 = getenv("SUMK_BK64") && getenv("SUMK_BK64")[0] == '1';
  if (g.precision == SUMK_PRECISION_BF16X3) {   // bf16x3 arithmetic (same tiles, same k order per tile shape)
    if (g.small_tile == 1) rc = launch_layout<64, 64, 32, 2>(layout, epi, ka, ka.total_tiles, stream);
    else if (g.small_tile == 2) rc = launch_layout<128, 64, 32, 2>(layout, epi, ka, ka.total_tiles, stream);
    else rc = launch_layout<128, 128, 32, 2>(layout, epi, ka, ka.total_tiles, stream);
  } else if (g.precision == SUMK_PRECISION_BF16X6) {   // fp32-grade emulation on the bf16 MFMA
    if (g.small_tile == 1) rc = launch_layout<64, 64, 32, 3>(layout, epi, ka, ka.total_tiles, stream);
    else if (g.small_tile == 2) rc = launch_layout<128, 64, 32, 3>(layout, epi, ka, ka.total_tiles, stream);
    else rc = launch_layout<128, 128, 32, 3>(layout, epi, ka, ka.total_tiles, stream);
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
__global__ void splitk_setup_kernel(GemmProb* p, int S, int M, int N, int K, int kchunk, int lda, int ldb, int bt) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  GemmProb q;
  int k0 = s * kchunk;
  q.a_off = (int64_t)k0 * lda; q.b_off = (int64_t)k0 * ldb; q.c_off = (int64_t)s * M * N; q.r_off = 0;
  q.M = M; q.N = N; q.K = min(kchunk, K - k0); q.lda = lda; q.ldb = ldb; q.ldc = N; q.ldr = 0;
  int tn = (N + bt - 1) / bt, tm = (M + bt - 1) / bt;
  q.tile_start = s * tm * tn; q.tiles_n = tn;
  for (int i = 0; i < 7; ++i) q.pad_[i] = 0;
  p[s] = q;
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

int gemm_tn_splitk_accum(const float* A, int lda, const float* B, int ldb, int M, int N, int K, float* slab,
                         size_t slab_elems, GemmProb* probs_dev, int probs_cap, float* const out[4], int rows_per_out,
                         int ldo, float alpha, hipStream_t stream, int precision) {
  SUMK_ARG(M > 0 && N > 0 && K > 0, "splitk: bad shape");
  SUMK_ARG(slab_elems >= (size_t)M * N, "splitk: slab too small");
  const int small = gemm_tiles(M, N, 0) >= 64 ? 0 : 1;
  const int tiles = gemm_tiles(M, N, small);
  int S = (1024 + tiles - 1) / tiles;
  S = std::min(S, (K + 63) / 64);
  S = std::min(S, (int)std::min<size_t>(slab_elems / ((size_t)M * N), (size_t)probs_cap));
  S = std::max(S, 1);
  int kchunk = ((K + S - 1) / S + 31) / 32 * 32;
  S = (K + kchunk - 1) / kchunk;
  hipLaunchKernelGGL(splitk_setup_kernel, dim3((S + 63) / 64), dim3(64), 0, stream, probs_dev, S, M, N, K, kchunk, lda, ldb,
                     gemm_tile_dim(small));
  GemmLaunch g;
  g.A = A; g.B[0] = B; g.C = slab; g.probs = probs_dev; g.nprob = S; g.small_tile = small; g.total_tiles = S * tiles;
  g.precision = precision;
  SUMK_TRY(launch_gemm(GEMM_TN, EPI_NONE, g, stream));
  SlabReduceArgs r;
  r.slab = slab; for (int i = 0; i < 4; ++i) r.out[i] = out[i];
  r.S = S; r.M = M; r.N = N; r.rows_per_out = rows_per_out; r.ldo = ldo; r.alpha = alpha;
  int64_t mn = (int64_t)M * N;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((mn + 255) / 256)), dim3(256), 0, stream, r);
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
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
int colsum_accum(const float* X, int ld, int R, int N, float* partial, int max_chunks, float* out, hipStream_t stream) {
  int chunks = std::max(1, std::min(max_chunks, (R + 63) / 64));
  int rpc = (R + chunks - 1) / chunks;
  chunks = (R + rpc - 1) / rpc;
  hipLaunchKernelGGL(colsum_partial_kernel, dim3((N + 255) / 256, chunks), dim3(256), 0, stream, X, ld, R, N, rpc, partial);
  hipLaunchKernelGGL(partial_reduce_kernel, dim3((N + 15) / 16), dim3(256), 0, stream, partial, chunks, N, N, out);
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
  if (const char* env = getenv("SUMK_ROW_CFG")) if (env[0] >= '0' && env[0] <= '2') small = env[0] - '0';
  if (getenv("SUMK_FAKE_LD")) { lda = 0; ldb = 0; }   // diagnostic: every row aliases row 0 (tiny footprint, all cache hits)
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
  static const bool on = getenv("SUMK_GEMM_DBG") && (atoi(getenv("SUMK_GEMM_DBG")) & 2);
  SUMK_ARG(on && out && n_blocks > 0 && n_blocks <= 2048, "gemm stamps: start the process with SUMK_GEMM_DBG=2 (n_blocks <= 2048)");
  SUMK_HIP(hipDeviceSynchronize());
  SUMK_HIP(hipMemcpy(out, gemm_stamp_buffer(), (size_t)n_blocks * 4 * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return SUMK_OK;
}

extern "C" int sumk_gemm_nt(const float* A, const float* B, float* C, int32_t M, int32_t N, int32_t K, void* stream) {
  return plain_gemm(sumk::GEMM_NT, A, B, C, M, N, K, K, K, stream);
}
extern "C" int sumk_gemm_prec(int32_t layout, const float* A, const float* B, float* C, int32_t M, int32_t N, int32_t K,
                              int32_t precision, void* stream) {
  SUMK_ARG(precision >= SUMK_PRECISION_FP32 && precision <= SUMK_PRECISION_BF16X6, "gemm: unknown precision %d", precision);
  SUMK_ARG(layout >= 0 && layout <= 2, "gemm: layout must be 0 (NT), 1 (NN) or 2 (TN), got %d", layout);
  if (layout == 0) return plain_gemm(sumk::GEMM_NT, A, B, C, M, N, K, K, K, stream, precision);
  if (layout == 1) return plain_gemm(sumk::GEMM_NN, A, B, C, M, N, K, K, N, stream, precision);
  return plain_gemm(sumk::GEMM_TN, A, B, C, M, N, K, M, N, stream, precision);
}
extern "C" int sumk_gemm_nn(const float* A, const float* B, float* C, int32_t M, int32_t N, int32_t K, void* stream) {
  return plain_gemm(sumk::GEMM_NN, A, B, C, M, N, K, K, N, stream);
}
extern "C" int sumk_gemm_tn(const float* A, const float* B, float* C, int32_t M, int32_t N, int32_t K, void* stream) {
  return plain_gemm(sumk::GEMM_TN, A, B, C, M, N, K, M, N, stream);
}
