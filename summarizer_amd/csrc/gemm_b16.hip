// bf16-SOURCE GEMM for the mixed-precision training step (SUMK_PRECISION_BF16; BASELINE config 2): operands are bf16 arrays in HBM
// (shadows written once by the kernels that produce the fp32 activations, csrc/vasnet.hip), accumulation and output are fp32.
//
// Why a second kernel family: the plane kernels of gemm_split.hip read fp32 operands and convert every k-tile on the vector ALU --
// at one v_mfma_f32_32x32x16_bf16 per product that conversion (and twice the operand bytes) is what the step pays for (QKV forward
// 0.19 of the bf16 matrix peak, weight gradients 0.12: profiles/r02, r03 *_train_bf16_*).  Here a k-tile is 16-byte buffer loads
// straight into the LDS image: no conversion, half the operand bytes, and no address arithmetic in the k-loop (per-thread byte
// offset fixed per tile, the k advance a scalar soffset -- the same form as the fp32 lean kernels).
//
// 128x128x64 tiles, 256 threads = 2x2 waves of 64x64 (2x2 MFMA tiles of 32x32), three blocks per CU.
//   KC operand (rows K-contiguous: A of NT / NN, B of NT): LDS image [row][64 bf16 + 16 B pad] (pitch 36 dwords: the 16 rows of a
//     ds_read_b128 lane group land on 16 distinct 16-byte slots); a lane's fragment is one 16-byte read.
//   MC operand ([k][row] in memory: B of NN, A and B of TN): LDS image [k][128 bf16 + 64 B pad], read with ds_read_b64_tr_b16
//     (two per 8-k fragment), exactly the plane image of gemm_regstage.h.
// Bounds come from the BUFFER DESCRIPTOR: num_records = the bytes from the (sub-)problem's first element to its last, so rows past M / N (KC) and
// k-rows past K (MC; a split-K slice ends where its descriptor ends) read as zero -- no clamps, no tail masks.  A K-contiguous
// operand needs K % 64 == 0 OR rows the caller zero-padded to a whole number of 64-wide k-tiles (the per-video attention matrices of
// csrc/vasnet.hip: ld16 = T rounded up to 64); an MC operand needs its row length % 8 == 0 or the same kind of padding (16-byte
// chunks); byte offsets must fit 31 bits.  The host checks the row-wise problems (gemm_b16_ok); the callers fall back to the plane
// kernels otherwise.
#include "gemm_regstage.h"
#include <type_traits>
#include <atomic>

namespace sumk {

namespace {

constexpr int BT = 128, BKH = 64;                 // block tile, k-tile (bf16 elements)
constexpr int KCP = 2 * BKH + 16;                 // KC image: bytes per row (144)
constexpr int MCP = 2 * BT + 64;                  // MC image: bytes per k-row (320)
constexpr int KC_BYTES = BT * KCP, MC_BYTES = BKH * MCP;

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <bool A_KC, bool B_KC, int EPI>
__global__ __launch_bounds__(256, 3) void gemm_b16_kernel(GemmKArgs ka) {
  constexpr int A_BYTES = A_KC ? KC_BYTES : MC_BYTES, B_BYTES = B_KC ? KC_BYTES : MC_BYTES;
  __shared__ __attribute__((aligned(16))) char lds[A_BYTES + B_BYTES];
  char* const sA = lds;
  char* const sB = lds + A_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
  const unsigned short* const A16 = reinterpret_cast<const unsigned short*>(ka.A);
  const unsigned short* const B16 = reinterpret_cast<const unsigned short*>(ka.B[0]);

  // this thread's four 16-byte chunks per operand and k-tile: KC = rows tid/8 + 32 p, k chunk tid%8; MC = k-rows tid/16 + 16 p, column chunk tid%16
  const int lds_a = A_KC ? (tid >> 3) * KCP + (tid & 7) * 16 : (tid >> 4) * MCP + (tid & 15) * 16;
  const int lds_b = B_KC ? (tid >> 3) * KCP + (tid & 7) * 16 : (tid >> 4) * MCP + (tid & 15) * 16;
  constexpr int LSTEP_A = A_KC ? 32 * KCP : 16 * MCP, LSTEP_B = B_KC ? 32 * KCP : 16 * MCP;

  struct Src { const unsigned short* a; const unsigned short* b; int na, nb, sa, sb; };   // operand bases, descriptor sizes (bytes), k-tile strides (bytes)
  int voa[4], vob[4];
  auto setup = [&](int tile, TileCtx& c, Src& s) -> bool {
    GemmProb P;
    if (!decode_tile<BT, BT>(ka, tile, c, P)) return false;
    s.a = A16 + P.a_off; s.b = B16 + P.b_off;
    if constexpr (A_KC) {
      s.na = ((P.M - 1) * P.lda + ((P.K + 63) & ~63)) * 2; s.sa = BKH * 2;
#pragma unroll
      for (int p = 0; p < 4; ++p) voa[p] = ((c.m0 + (tid >> 3) + 32 * p) * P.lda + (tid & 7) * 8) * 2;
    } else {
      s.na = ((P.K - 1) * P.lda + ((P.M + 7) & ~7)) * 2; s.sa = BKH * P.lda * 2;
#pragma unroll
      for (int p = 0; p < 4; ++p) voa[p] = (((tid >> 4) + 16 * p) * P.lda + c.m0 + (tid & 15) * 8) * 2;
    }
    if constexpr (B_KC) {
      s.nb = ((P.N - 1) * P.ldb + ((P.K + 63) & ~63)) * 2; s.sb = BKH * 2;
#pragma unroll
      for (int p = 0; p < 4; ++p) vob[p] = ((c.n0 + (tid >> 3) + 32 * p) * P.ldb + (tid & 7) * 8) * 2;
    } else {
      s.nb = ((P.K - 1) * P.ldb + ((P.N + 7) & ~7)) * 2; s.sb = BKH * P.ldb * 2;
#pragma unroll
      for (int p = 0; p < 4; ++p) vob[p] = (((tid >> 4) + 16 * p) * P.ldb + c.n0 + (tid & 15) * 8) * 2;
    }
    return true;
  };

  u32x4 ra[4], rb[4];
  auto gload = [&](const Src& s, int kt) {
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(s.a), (short)0, s.na, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(s.b), (short)0, s.nb, 0x00020000);
#pragma unroll
    for (int p = 0; p < 4; ++p) ra[p] = (u32x4)__builtin_amdgcn_raw_buffer_load_b128(rA, voa[p], kt * __builtin_amdgcn_readfirstlane(s.sa), 0);     // (uniform strides: see the wide kernel's gload_slice)
#pragma unroll
    for (int p = 0; p < 4; ++p) rb[p] = (u32x4)__builtin_amdgcn_raw_buffer_load_b128(rB, vob[p], kt * __builtin_amdgcn_readfirstlane(s.sb), 0);
  };
  auto swrite = [&]() {
#pragma unroll
    for (int p = 0; p < 4; ++p) *reinterpret_cast<u32x4*>(sA + lds_a + p * LSTEP_A) = ra[p];
#pragma unroll
    for (int p = 0; p < 4; ++p) *reinterpret_cast<u32x4*>(sB + lds_b + p * LSTEP_B) = rb[p];
  };
  // fragment bases of this lane (k step ks adds 32 B along a KC row, 16 k-rows of an MC image)
  const char* const fa = A_KC ? sA + (wm * 64 + li) * KCP + 16 * lh
                              : sA + (8 * lh + ((lane & 15) >> 2)) * MCP + 2 * (wm * 64 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3));
  const char* const fb = B_KC ? sB + (wn * 64 + li) * KCP + 16 * lh
                              : sB + (8 * lh + ((lane & 15) >> 2)) * MCP + 2 * (wn * 64 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3));
  // fragments of step ks + 1 are requested before the MFMAs of step ks (pinned with sched_barrier: see the wide kernel below)
  struct Frags { bf16x8 a[2], b[2]; };
  auto read_frags = [&](int ks, Frags& f) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if constexpr (A_KC) f.a[t] = *reinterpret_cast<const bf16x8*>(fa + t * 32 * KCP + 32 * ks);
      else f.a[t] = tr_frag(fa + 16 * ks * MCP + 64 * t, MCP);
      if constexpr (B_KC) f.b[t] = *reinterpret_cast<const bf16x8*>(fb + t * 32 * KCP + 32 * ks);
      else f.b[t] = tr_frag(fb + 16 * ks * MCP + 64 * t, MCP);
    }
  };
  auto compute = [&](f32x16 (&acc)[2][2]) {
    Frags f[2];
    read_frags(0, f[0]);
#pragma unroll
    for (int ks = 0; ks < BKH / 16; ++ks) {
      if (ks + 1 < BKH / 16) read_frags(ks + 1, f[(ks + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[ks & 1].a[tm], f[ks & 1].b[tn], acc[tm][tn], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  int tile = blockIdx.x;
  if (tile >= ka.total_tiles) return;
  TileCtx cur, nxt;
  Src scur, snxt;
  if (!setup(tile, cur, scur)) return;
  gload(scur, 0);

  while (true) {
    f32x16 acc[2][2];
    if constexpr (EPI == EPI_RESIDUAL) {
      residual_init<2, 2>(ka, cur, acc, cur.m0 + wm * 64, cur.n0 + wn * 64, li, lh);
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
    const int nk = (cur.K + BKH - 1) / BKH;
    const int next_tile = tile + gridDim.x;
    bool has_next = false;
    for (int kt = 0; kt + 1 < nk; ++kt) {
      __syncthreads();
      swrite();
      __syncthreads();
      gload(scur, kt + 1);
      compute(acc);
    }
    // last k-tile: the next tile's decode and first loads go out under its MFMAs
    __syncthreads();
    swrite();
    __syncthreads();
    has_next = next_tile < ka.total_tiles;
    if (has_next) has_next = setup(next_tile, nxt, snxt);
    if (has_next) gload(snxt, 0);
    compute(acc);

    epilogue_store<EPI, 2, 2, true>(ka, cur, acc, cur.m0 + wm * 64, cur.n0 + wn * 64, li, lh);
    if (!has_next) break;
    tile = next_tile; cur = nxt; scur = snxt;
  }
}

template <bool A_KC, bool B_KC>
int launch_epi_b16(GemmEpi epi, const GemmKArgs& ka, int tiles, hipStream_t s) {
  const dim3 grid(std::min(tiles, 256 * 3)), block(256);
  switch (epi) {
    case EPI_NONE: hipLaunchKernelGGL((gemm_b16_kernel<A_KC, B_KC, EPI_NONE>), grid, block, 0, s, ka); break;
    case EPI_RESIDUAL: hipLaunchKernelGGL((gemm_b16_kernel<A_KC, B_KC, EPI_RESIDUAL>), grid, block, 0, s, ka); break;
    case EPI_BIAS_RELU: hipLaunchKernelGGL((gemm_b16_kernel<A_KC, B_KC, EPI_BIAS_RELU>), grid, block, 0, s, ka); break;
    case EPI_ACCUM: hipLaunchKernelGGL((gemm_b16_kernel<A_KC, B_KC, EPI_ACCUM>), grid, block, 0, s, ka); break;
    default: set_error("gemm (bf16 sources): epilogue %d is not instantiated", (int)epi); return SUMK_ERR_ARG;
  }
  return SUMK_OK;
}


// ------------------------------------------------------------------------------------------- wide tiles
// (192 | 256) x 256 x 64 tiles, 512 threads = 2 x 4 waves of (BM / 2) x 64, one block per CU, DOUBLE-BUFFERED LDS (one barrier per
// k-tile).  Why: at one bf16 MFMA per product the 128x128 tile asks the L2 for 32 KB per 512 matrix-pipe cycles and CU -- 38 TB/s
// chip-wide at full rate, twice what the L2s deliver (MI355X_MICROARCH.md: ~17-19 TB/s): measured 9.7 TB/s and 0.25 of the bf16
// peak.  A 256-wide tile halves the operand bytes per FLOP; BM = 192 exists because 12 003 rows are 63 x 192 (one full round of 252
// blocks on 256 CUs for an N = 1024 output, three full rounds at N = 3072) but 47 x 256 (0.73 of a round).
template <int BM, bool A_KC, bool B_KC, int EPI>
__global__ __launch_bounds__(512) void gemm_b16_wide_kernel(GemmKArgs ka) {
  constexpr int BN = 256, TM = BM / 64, TN = 2, WTM = BM / 2;
  constexpr int MCP_A = 2 * BM + 64, MCP_B = 2 * BN + 64;            // MC images: bytes per k-row (pitch = 64 mod 256: the 4 k-rows of a transposing read hit 4 bank groups)
  constexpr int A_BYTES = A_KC ? BM * KCP : BKH * MCP_A, B_BYTES = B_KC ? BN * KCP : BKH * MCP_B, STAGE = A_BYTES + B_BYTES;
  constexpr int NLA = BM / 64, NLB = BN / 64;                        // 16-byte chunks per thread, operand and k-tile
  constexpr int CPR_A = BM / 8, CPR_B = BN / 8;                      // MC: chunks per k-row
  extern __shared__ __attribute__((aligned(16))) char lds_w[];       // 2 stages

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3, li = lane & 31, lh = lane >> 5;
  const unsigned short* const A16 = reinterpret_cast<const unsigned short*>(ka.A);
  const unsigned short* const B16 = reinterpret_cast<const unsigned short*>(ka.B[0]);

  struct Src { __amdgpu_buffer_rsrc_t ra, rb; int sa, sb; };   // operand descriptors of the tile's (sub-)problem, k-tile strides (bytes)
  // (every field is forced into SGPRs: carried around the tile loop the descriptors were classed as divergent and each buffer load
  //  became a waterfall loop of v_readfirstlane + compare)
  auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
  auto uni_ptr = [&](const unsigned short* p) {
    const uint64_t u = (uint64_t)p;
    return (unsigned short*)(((uint64_t)(uint32_t)uni((int)(u >> 32)) << 32) | (uint32_t)uni((int)(uint32_t)u));
  };
  int voa[NLA], vob[NLB], lda_[NLA], ldb_[NLB];     // global byte offsets / LDS byte offsets (inside a stage) of this thread's chunks
#pragma unroll
  for (int p = 0; p < NLA; ++p) {
    const int idx = tid + 512 * p;
    lda_[p] = A_KC ? (idx >> 3) * KCP + (idx & 7) * 16 : (idx / CPR_A) * MCP_A + (idx % CPR_A) * 16;
  }
#pragma unroll
  for (int p = 0; p < NLB; ++p) {
    const int idx = tid + 512 * p;
    ldb_[p] = A_BYTES + (B_KC ? (idx >> 3) * KCP + (idx & 7) * 16 : (idx / CPR_B) * MCP_B + (idx % CPR_B) * 16);
  }
  auto setup = [&](int tile, TileCtx& c, Src& s) -> bool {
    GemmProb P;
    if (!decode_tile<BM, BN>(ka, tile, c, P)) return false;
    int na, nb;
#pragma unroll
    for (int p = 0; p < NLA; ++p) {
      const int idx = tid + 512 * p;
      voa[p] = A_KC ? ((c.m0 + (idx >> 3)) * P.lda + (idx & 7) * 8) * 2 : ((idx / CPR_A) * P.lda + c.m0 + (idx % CPR_A) * 8) * 2;
    }
#pragma unroll
    for (int p = 0; p < NLB; ++p) {
      const int idx = tid + 512 * p;
      vob[p] = B_KC ? ((c.n0 + (idx >> 3)) * P.ldb + (idx & 7) * 8) * 2 : ((idx / CPR_B) * P.ldb + c.n0 + (idx % CPR_B) * 8) * 2;
    }
    if constexpr (A_KC) { na = ((P.M - 1) * P.lda + ((P.K + 63) & ~63)) * 2; s.sa = BKH * 2; }
    else { na = ((P.K - 1) * P.lda + ((P.M + 7) & ~7)) * 2; s.sa = uni(BKH * P.lda * 2); }
    if constexpr (B_KC) { nb = ((P.N - 1) * P.ldb + ((P.K + 63) & ~63)) * 2; s.sb = BKH * 2; }
    else { nb = ((P.K - 1) * P.ldb + ((P.N + 7) & ~7)) * 2; s.sb = uni(BKH * P.ldb * 2); }
    s.ra = __builtin_amdgcn_make_buffer_rsrc(uni_ptr(A16 + P.a_off), (short)0, uni(na), 0x00020000);
    s.rb = __builtin_amdgcn_make_buffer_rsrc(uni_ptr(B16 + P.b_off), (short)0, uni(nb), 0x00020000);
    return true;
  };

  u32x4 ra[NLA], rb[NLB];
  // the staging work of one k-tile is cut into four SLICES, one per 16-deep MFMA step: chunk p of an operand belongs to slice p & 3
  // (BM = 192: A has chunks 0..2).  Slice q of k-tile kt + 1 is written to the other LDS stage, then the same registers are re-loaded
  // with k-tile kt + 2, behind the MFMAs of step q of k-tile kt -- the LDS store burst and the load issue are spread over the k-tile
  // instead of standing between the barrier and the first MFMA.
  // (the k-tile strides are wave-uniform but travel through the tile loop's Src copies, where the compiler re-classes them as divergent -- each MC-operand
  //  load then sat in a waterfall loop, sixteen per k-tile in the TN kernels: readfirstlane at the use)
  auto gload_slice = [&](const Src& s, int kt, int q) {
    if (q < NLA) ra[q] = (u32x4)__builtin_amdgcn_raw_buffer_load_b128(s.ra, voa[q], kt * uni(s.sa), 0);
    if (q < NLB) rb[q] = (u32x4)__builtin_amdgcn_raw_buffer_load_b128(s.rb, vob[q], kt * uni(s.sb), 0);
  };
  auto gload = [&](const Src& s, int kt) {
#pragma unroll
    for (int q = 0; q < 4; ++q) gload_slice(s, kt, q);
  };
  auto swrite_slice = [&](char* st, int q) {
    if (q < NLA) *reinterpret_cast<u32x4*>(st + lda_[q]) = ra[q];
    if (q < NLB) *reinterpret_cast<u32x4*>(st + ldb_[q]) = rb[q];
  };
  const int fa = A_KC ? (wm * WTM + li) * KCP + 16 * lh
                      : (8 * lh + ((lane & 15) >> 2)) * MCP_A + 2 * (wm * WTM + 16 * ((lane >> 4) & 1) + 4 * (lane & 3));
  const int fb = A_BYTES + (B_KC ? (wn * 64 + li) * KCP + 16 * lh
                                 : (8 * lh + ((lane & 15) >> 2)) * MCP_B + 2 * (wn * 64 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3)));
  struct Frags { bf16x8 a[TM], b[TN]; };
  auto read_frags = [&](const char* st, int ks, Frags& f) {
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      if constexpr (A_KC) f.a[t] = *reinterpret_cast<const bf16x8*>(st + fa + t * 32 * KCP + 32 * ks);
      else f.a[t] = tr_frag(st + fa + 16 * ks * MCP_A + 64 * t, MCP_A);
    }
#pragma unroll
    for (int t = 0; t < TN; ++t) {
      if constexpr (B_KC) f.b[t] = *reinterpret_cast<const bf16x8*>(st + fb + t * 32 * KCP + 32 * ks);
      else f.b[t] = tr_frag(st + fb + 16 * ks * MCP_B + 64 * t, MCP_B);
    }
  };
  // One k-tile = four 16-deep steps.  The fragments of step ks + 1 are requested BEFORE the MFMAs of step ks (left to itself the
  // scheduler sinks every ds_read next to its first use and waits lgkmcnt(0) in front of each pair of MFMAs: 0.28 of the matrix rate
  // with loads, LDS writes and stores switched off); sched_barrier pins that order.  The workgroup barrier sits between steps 2 and 3:
  // by then this wave has requested its last fragments of the current stage and written all its slices of the next one, so after the
  // barrier step 0 of the NEXT k-tile can be requested and lands under the MFMAs of step 3 -- no LDS latency is exposed at the k-tile
  // boundary, where all eight waves would otherwise wait for their first fragments at once.
  // MODE 2: write k-tile kt + 1 to the other stage and load k-tile kt + 2; 1: write only; 0: neither (last k-tile of the tile).
  // f[0] holds step 0 of `st` on entry and step 0 of `nst` on exit (MODE >= 1).
  struct Frags2 { Frags f[2]; };
  auto mfma_step = [&](const Frags& f, f32x16 (&acc)[TM][TN]) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[tm], f.b[tn], acc[tm][tn], 0, 0, 0);
  };
  auto ktile = [&](const char* st, char* nst, const Src& s, int kt, auto mode, Frags2& F, f32x16 (&acc)[TM][TN]) {
    constexpr int MODE = decltype(mode)::value;
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      read_frags(st, ks + 1, F.f[(ks + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      mfma_step(F.f[ks & 1], acc);
      // staging slices: 0 behind step 0, 1 behind step 1, 2 and 3 behind step 2
      if constexpr (MODE >= 1) { swrite_slice(nst, ks); if (ks == 2) swrite_slice(nst, 3); }
      if constexpr (MODE >= 2) { gload_slice(s, kt + 2, ks); if (ks == 2) gload_slice(s, kt + 2, 3); }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    if constexpr (MODE >= 1) read_frags(nst, 0, F.f[0]);
    __builtin_amdgcn_sched_barrier(0);
    mfma_step(F.f[1], acc);
    __builtin_amdgcn_sched_barrier(0);
  };

  int tile = blockIdx.x;
  if (tile >= ka.total_tiles) return;
  TileCtx cur, nxt;
  Src scur, snxt;
  if (!setup(tile, cur, scur)) return;
  gload(scur, 0);

  while (true) {
    f32x16 acc[TM][TN];
    if constexpr (EPI == EPI_RESIDUAL) {
      residual_init<TM, TN>(ka, cur, acc, cur.m0 + wm * WTM, cur.n0 + wn * 64, li, lh);
    } else {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
    const int nk = (cur.K + BKH - 1) / BKH;
    const int next_tile = tile + gridDim.x;
    // stage 0 <- k-tile 0 (every wave is past the previous tile's last barrier and holds no fragment of it: both stages are free)
#pragma unroll
    for (int q = 0; q < 4; ++q) swrite_slice(lds_w, q);
    if (nk > 1) gload(scur, 1);
    __syncthreads();
    Frags2 F;
    read_frags(lds_w, 0, F.f[0]);
    int kt = 0;
    for (; kt + 2 < nk; ++kt)
      ktile(lds_w + (kt & 1) * STAGE, lds_w + ((kt + 1) & 1) * STAGE, scur, kt, std::integral_constant<int, 2>{}, F, acc);
    if (kt + 1 < nk) {
      ktile(lds_w + (kt & 1) * STAGE, lds_w + ((kt + 1) & 1) * STAGE, scur, kt, std::integral_constant<int, 1>{}, F, acc);
      ++kt;
    }
    // last k-tile: the next tile's decode and first loads go out under its MFMAs
    bool has_next = next_tile < ka.total_tiles;
    if (has_next) has_next = setup(next_tile, nxt, snxt);
    if (has_next) gload(snxt, 0);
    ktile(lds_w + (kt & 1) * STAGE, nullptr, scur, kt, std::integral_constant<int, 0>{}, F, acc);
    epilogue_store<EPI, TM, TN, true>(ka, cur, acc, cur.m0 + wm * WTM, cur.n0 + wn * 64, li, lh);
    if (!has_next) break;
    tile = next_tile; cur = nxt; scur = snxt;
  }
}

template <int BM, bool A_KC, bool B_KC, int EPI>
int launch_wide_one(const GemmKArgs& ka, int tiles, hipStream_t s) {
  constexpr int A_BYTES = A_KC ? BM * KCP : BKH * (2 * BM + 64), B_BYTES = B_KC ? 256 * KCP : BKH * (2 * 256 + 64);
  constexpr int LDS = 2 * (A_BYTES + B_BYTES);
  static_assert(LDS <= 160 * 1024, "two stages must fit the CU's LDS");
  const void* fn = (const void*)gemm_b16_wide_kernel<BM, A_KC, B_KC, EPI>;
  static std::atomic<uint64_t> attr_done{0};          // one bit per device (the opt-in is a per-device attribute)
  int dev = 0;
  SUMK_HIP(hipGetDevice(&dev));
  const uint64_t bit = 1ull << (dev & 63);
  if (!(attr_done.load(std::memory_order_acquire) & bit)) {
    SUMK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_done.fetch_or(bit, std::memory_order_release);
  }
  hipLaunchKernelGGL((gemm_b16_wide_kernel<BM, A_KC, B_KC, EPI>), dim3(std::min(tiles, 256)), dim3(512), LDS, s, ka);
  return SUMK_OK;
}
template <int BM, bool A_KC, bool B_KC>
int launch_epi_wide(GemmEpi epi, const GemmKArgs& ka, int tiles, hipStream_t s) {
  switch (epi) {
    case EPI_NONE: return launch_wide_one<BM, A_KC, B_KC, EPI_NONE>(ka, tiles, s);
    case EPI_RESIDUAL: return launch_wide_one<BM, A_KC, B_KC, EPI_RESIDUAL>(ka, tiles, s);
    case EPI_BIAS_RELU: return launch_wide_one<BM, A_KC, B_KC, EPI_BIAS_RELU>(ka, tiles, s);
    case EPI_ACCUM: return launch_wide_one<BM, A_KC, B_KC, EPI_ACCUM>(ka, tiles, s);
    default: set_error("gemm (bf16 sources): epilogue %d is not instantiated", (int)epi); return SUMK_ERR_ARG;
  }
}

}  // namespace

// wide = 0: 128x128 tiles; 192 / 256: BM of the (BM x 256) tile (the problem table's tiles_n / tile_start and `tiles` were built for it)
int launch_gemm_b16(GemmLayout layout, GemmEpi epi, const GemmKArgs& ka, int tiles, int wide, hipStream_t stream) {
  if (wide == 192) {
    if (layout == GEMM_NT) return launch_epi_wide<192, true, true>(epi, ka, tiles, stream);
    if (layout == GEMM_NN) return launch_epi_wide<192, true, false>(epi, ka, tiles, stream);
    return launch_epi_wide<192, false, false>(epi, ka, tiles, stream);
  }
  if (wide == 256) {
    if (layout == GEMM_NT) return launch_epi_wide<256, true, true>(epi, ka, tiles, stream);
    if (layout == GEMM_NN) return launch_epi_wide<256, true, false>(epi, ka, tiles, stream);
    return launch_epi_wide<256, false, false>(epi, ka, tiles, stream);
  }
  if (layout == GEMM_NT) return launch_epi_b16<true, true>(epi, ka, tiles, stream);
  if (layout == GEMM_NN) return launch_epi_b16<true, false>(epi, ka, tiles, stream);
  return launch_epi_b16<false, false>(epi, ka, tiles, stream);
}

// ------------------------------------------------------------------------------------------- fp32 -> bf16 shadows
// dst_k = bf16(src_k) for up to six dense fp32 arrays in ONE launch (round to nearest even, like the plane kernels' v_cvt_pk_bf16_f32):
// the features and the five weight matrices of a training step.  Element counts are multiples of 4.
struct CastArgs { const float* src[6]; unsigned short* dst[6]; int64_t end4[6]; int32_t n; };   // end4: running total of float4 groups
__global__ __launch_bounds__(256) void cast_flat_b16_kernel(CastArgs a) {
  const int64_t i4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i4 >= a.end4[a.n - 1]) return;
  int k = 0;
#pragma unroll
  for (int q = 0; q < 5; ++q) k += (q + 1 < a.n && i4 >= a.end4[q]) ? 1 : 0;
  const int64_t local = i4 - (k == 0 ? 0 : a.end4[k - 1]);
  const float4 v = reinterpret_cast<const float4*>(a.src[k])[local];
  const f32x4 f = {v.x, v.y, v.z, v.w};
  reinterpret_cast<bf16x4*>(a.dst[k])[local] = __builtin_convertvector(f, bf16x4);
}

int cast_flat_b16(int n, const float* const src[], void* const dst[], const int64_t n_elems[], hipStream_t stream) {
  SUMK_ARG(n >= 1 && n <= 6, "cast_flat_b16: 1..6 arrays");
  CastArgs a;
  int64_t run = 0;
  for (int k = 0; k < 6; ++k) {
    if (k < n) { SUMK_ARG(src[k] && dst[k] && n_elems[k] % 4 == 0, "cast_flat_b16: bad array %d", k); run += n_elems[k] / 4; }
    a.src[k] = k < n ? src[k] : nullptr; a.dst[k] = k < n ? (unsigned short*)dst[k] : nullptr; a.end4[k] = run;
  }
  a.n = n;
  if (run == 0) return SUMK_OK;
  hipLaunchKernelGGL(cast_flat_b16_kernel, dim3((unsigned)((run + 255) / 256)), dim3(256), 0, stream, a);
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}

}  // namespace sumk
