// Plane-aware wide GEMM for the split-bf16 arithmetics (SUMK_PRECISION_BF16X6 / BF16X3) of the packed scoring path
// (reference: summarizer/models/vasnet.py:114-145 -- the K/Q/V projections, the output projection and k1).
//
// C(M, N) = A(M, K) . B(N, K)^T with BOTH operands already split into bf16 planes in HBM ("KB planes", sumk_internal.h): x once per
// dataset, weights once per weight change, activations by the epilogue of the kernel that produced them.  The in-loop split kernels
// (gemm_regstage.h, NS = 2 / 3) convert every k-tile of both operands on the vector ALU of every block that touches it and read
// 0.8 LDS fragments per MFMA: matrix pipe busy 0.31-0.53 (profiles/r03_pmc_clock_mfma_by_mode.json).  Here
//   * a (192 x 256) block tile, 512 threads = 2 x 4 waves of 96 x 64 (3 x 2 MFMA tiles of 32 x 32), one block per CU;
//   * one k16 step = NP (A planes) x 3 + NP (B planes) x 2 fragment reads (ds_read_b128, conflict-free, no padding) feeding
//     6 x NT MFMAs (NT = 6 products for three planes, 3 for two): 15 reads per 36 MFMAs at bf16x6, 10 per 18 at bf16x3;
//   * operands reach LDS by LDS-DMA (buffer_load_dwordx4 ... lds): 1 KiB = 64 rows x 16 B per wave instruction, contiguous in HBM AND
//     in LDS by construction of the plane format -- no VGPR staging, no ds_write, no VALU, no address arithmetic in the loop (the
//     per-lane offset is lane * 16 for every piece; everything else is a scalar offset);
//   * a ring of NS k16 stages (bf16x6: 3 x 42 KB; bf16x3: 4 x 28 KB), ONE barrier per k16 step placed after two thirds of the step's
//     MFMAs: the fragments of step s + 1 are requested right behind the barrier and land under the last third, so neither the
//     barrier nor the LDS latency is exposed; the two waves of a SIMD (w, w + 4) issue their DMA pieces at different points of the
//     step (a piece holds its wave's instruction issue for ~100 cycles; the partner's MFMAs fill the pipe meanwhile).
// Term order per accumulator and k16 step is that of the in-loop kernels (smallest products first), the planes are the same roundings,
// so on the same operand values the results are bit-identical to precision = bf16x6 / bf16x3 of gemm_regstage.h (tested).
//
// Epilogues (PwEpi).  PW_F32 keeps the usual orientation (lane = output column: 128-byte row segments of fp32).  The others run the
// MFMA TRANSPOSED (B rows on the M axis): a lane then owns ONE output row and 4 consecutive columns per register quad, which is
// exactly an 8-byte piece of a KB-plane chunk -- the result is split into planes in registers and leaves as 512-byte contiguous runs.
#include "pw_common.h"
#include <atomic>
#include <cstdlib>
#include <type_traits>

namespace sumk {

namespace {



template <int NP, int BM, int EPI, int NS, int VAR>
__global__ __launch_bounds__(512) void gemm_pw_kernel(PwArgs a) {
  constexpr int BN = 256, WTM = BM / 2, TM = WTM / 32, TN = 2, NSUB = 2 * NP;
  constexpr bool SWAP = EPI != PW_F32;
  constexpr int A_BYTES = NSUB * BM * 16, B_BYTES = NSUB * BN * 16, STAGE = A_BYTES + B_BYTES;
  constexpr int PA = BM / 64, PB = BN / 64, PPS = PA + PB;
  static_assert(NS * STAGE + PW_CONST_BYTES <= 160 * 1024, "the stage ring must fit the CU's LDS");
  extern __shared__ __attribute__((aligned(16))) char lds[];

  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  // Schedule variants (probes; the product runs VAR = 2).  When does a wave issue the DMA pieces that refill the slot a barrier freed?
  //   0: waves 0-3 right behind the barrier, waves 4-7 after the step's last MFMAs;  1: every wave right behind the barrier;
  //   2: every wave after the step's last MFMAs;  3: as 2, with cycle stamps (diagnostic build only).
  // Measured (profiles/r05_pw_gemm_variants.txt, QKV shape): 2 is the fastest at two planes and ties the rest at three; also tried and
  // within +-2 % of it: the barrier after one third of the step's MFMAs instead of two thirds, s_setprio(1) around the MFMA groups,
  // both, and the refill spread between the MFMA groups of the next step -- the loop runs at 83 % of its MFMA floor whatever the
  // placement, and the chip holds ~1.7 GHz under it (stamps: profiles/r05_pw_stamps.txt).
  const bool early = VAR == 1 ? true : (VAR == 0 ? wave < 4 : false);
  constexpr int P1 = TM - 1;                                             // MFMA row tiles in front of the barrier

  // DMA pieces: a stage is NSUB sub-arrays x (PA + PB) blocks of 64 rows; wave w < PA + PB owns row block w of EVERY sub-array (waves
  // 0 .. PA - 1: blocks of A, the next PB: blocks of B; wave 7 issues none) -- one descriptor, one row offset and one LDS offset per
  // wave, the sub-array a compile-time multiple: no per-piece tables (a first version kept three scalar arrays per wave and spilled
  // 150 SGPRs into the k-loop).
  static_assert(PPS <= 8, "one 64-row block per wave");
  const bool dma_wave = wave < PPS, dma_a = wave < PA;
  const int blk = dma_a ? wave : wave - PA;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(dma_a ? a.A : a.B), (short)0, 0x80000000u, 0x00020000);
  const int rp16 = (int)(dma_a ? a.a_rp16 : a.b_rp16);
  const int lds_blk = dma_a ? blk * 1024 : A_BYTES + blk * 1024;
  constexpr int LSUB_A = BM * 16, LSUB_B = BN * 16;
  const int lsub = dma_a ? LSUB_A : LSUB_B;
  const int vlane = lane * 16;
  const int k_step = NSUB * rp16;                                       // bytes per k16 block
  auto dma = [&](int m0, int n0, int kb, int slot) {
    if (!dma_wave) return;
    char* const st = lds + slot * STAGE + lds_blk;
    const int g0 = kb * k_step + ((dma_a ? m0 : n0) + blk * 64) * 16;
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_vptr)(st + sub * lsub), 16, vlane, g0 + sub * rp16, 0, 0);
  };
  constexpr int IN_FLIGHT = (NS - 2) * NSUB;                            // DMA instructions of this wave that may stay in flight across a step's barrier

  // fragments: plane p, MFMA tile t of this wave -> 16 bytes per lane
  const int fa = (lh * BM + wm * WTM + li) * 16, fb = A_BYTES + (lh * BN + wn * 64 + li) * 16;
  struct Frags { bf16x8 a[NP][TM], b[NP][TN]; };
  auto read_frags = [&](int slot, Frags& f) {
    const char* const st = lds + slot * STAGE;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
#pragma unroll
      for (int t = 0; t < TN; ++t) f.b[p][t] = *reinterpret_cast<const bf16x8*>(st + fb + p * 2 * BN * 16 + t * 512);
#pragma unroll
      for (int t = 0; t < TM; ++t) f.a[p][t] = *reinterpret_cast<const bf16x8*>(st + fa + p * 2 * BM * 16 + t * 512);
    }
  };
  auto mfma_rows = [&](const Frags& f, f32x16 (&acc)[TM][TN], int tm_lo, int tm_hi) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      if (tm < tm_lo || tm >= tm_hi) continue;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
#pragma unroll
        for (int sum = NP - 1; sum >= 0; --sum)          // planes (i, j) with i + j descending: smallest products first
#pragma unroll
          for (int i = NP - 1; i >= 0; --i) {
            const int j = sum - i;
            if (j < 0 || j >= NP) continue;
            if constexpr (SWAP) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.b[j][tn], f.a[i][tm], acc[tm][tn], 0, 0, 0);
            else acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][tm], f.b[j][tn], acc[tm][tn], 0, 0, 0);
          }
      }
    }
  };

  // Tile walk (speed only; block b runs on XCD b % 8 under round-robin dispatch and the persistent stride keeps tile % 8 = b % 8).
  // xcd_map 2: XCD x owns a BAND of ceil(tiles_m / 8) row tiles and every column tile; inside the band the tiles are dealt in groups of 4
  // column tiles, rows fastest across groups of 4 columns -- the 32 blocks an XCD runs at a time are 8 row tiles x 4 column tiles: each A
  // panel is fetched into ONE L2 and hit by the other three blocks that need it, each B panel by the other seven.  (The 2 x 4 rectangles
  // of gemm_device.h, xcd_map 1, give an XCD with N = 1024 a single column tile: every A panel was fetched by four L2s and never re-used
  // inside one -- PMC: L2 hit 0.52, 343 MB fetched for 74 MB of A; profiles/r05_pmc_pw_x6_tail.json.)
  auto decode = [&](int t, int& m0, int& n0) -> bool {
    int mt, nt;
    if (a.xcd_map == 2) {
      const int x = t & 7, j = t >> 3, rx = (a.tiles_m + 7) >> 3, per = rx * 4;
      const int grp = j / per, jj = j - grp * per, r = jj >> 2;
      mt = x * rx + r; nt = grp * 4 + (jj & 3);
      if (grp >= (a.tiles_n >> 2) || mt >= a.tiles_m) return false;
    } else if (a.xcd_map == 1) {
      const int x = t & 7, j = t >> 3, sm = (a.tiles_m + 1) >> 1, sn = a.tiles_n >> 2;
      const int jm = j / sn;
      mt = (x >> 2) * sm + jm; nt = (x & 3) * sn + (j - jm * sn);
      if (jm >= sm || mt >= a.tiles_m) return false;
    } else {
      if (t >= a.total_tiles) return false;
      mt = t / a.tiles_n; nt = t - mt * a.tiles_n;
    }
    m0 = mt * BM; n0 = nt * BN;
    return true;
  };

  const int nk = a.K >> 4;
#ifdef SUMK_DIAG
  unsigned long long t_start = 0, t_loop = 0, t_epi = 0, rt_start = 0, n_tiles = 0;
  if constexpr (VAR == 3) { t_start = __builtin_amdgcn_s_memtime(); rt_start = __builtin_amdgcn_s_memrealtime(); }
#endif
  // (a virtual tile id of an XCD map may name no tile: walk on to this block's next one)
  auto next_valid = [&](int t, int& m, int& n) -> int {
    for (; t < a.total_tiles; t += gridDim.x)
      if (decode(t, m, n)) return t;
    return -1;
  };
  int m0 = 0, n0 = 0;
  int tile = next_valid(blockIdx.x, m0, n0);
  if (tile < 0) return;
#pragma unroll
  for (int s = 0; s < NS; ++s) dma(m0, n0, s, s);

  while (true) {
    f32x16 acc[TM][TN];
    if constexpr (EPI == PW_RES_MOM_PLANES || EPI == PW_RES_F32) {
      // lane = row m, registers 4 g .. 4 g + 3 = columns n0w + 32 tn + 8 g + 4 lh + (0..3)
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const int m = m0 + wm * WTM + tm * 32 + li;
        const float* const rp = a.R + (int64_t)min(m, a.M - 1) * a.ldr + n0 + wn * 64 + 4 * lh;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float4 v = *reinterpret_cast<const float4*>(rp + tn * 32 + 8 * g);
            if constexpr (EPI == PW_RES_F32) {
              if (a.bias) { const float4 b = *reinterpret_cast<const float4*>(a.bias + n0 + wn * 64 + 4 * lh + tn * 32 + 8 * g); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
            }
            acc[tm][tn][4 * g] = v.x; acc[tm][tn][4 * g + 1] = v.y; acc[tm][tn][4 * g + 2] = v.z; acc[tm][tn][4 * g + 3] = v.w;
          }
      }
    } else {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }

    // ---- k loop: stage s lives in slot s % NS; Frags[s & 1]
    wait_vm<0>();                      // (start of a tile: the previous epilogue's stores share the counter: drain; stages 1.. are mostly there by now)
    __builtin_amdgcn_s_barrier();
#ifdef SUMK_DIAG
    unsigned long long t0 = 0;
    if constexpr (VAR == 3) t0 = __builtin_amdgcn_s_memtime();
#endif
    Frags F0, F1;
    read_frags(0, F0);
    int slot = 0;
    // MAIN: steady state (stage s + NS exists: refill the slot of stage s; stages s + 2 .. s + NS - 1 stay in flight across the barrier).
    // !MAIN: the last steps of a tile (run-time conditions, full drain before the barrier).
    auto kstep = [&](auto main_, const Frags& cur, Frags& nxt, int s) {
      constexpr bool MAIN = decltype(main_)::value;
      __builtin_amdgcn_sched_barrier(0);
      mfma_rows(cur, acc, 0, P1);
      __builtin_amdgcn_sched_barrier(0);
      const bool more = MAIN || s + 1 < nk, fill = MAIN || s + NS < nk;
      const int nslot = slot + 1 == NS ? 0 : slot + 1;
      if (more) {
        if constexpr (MAIN) wait_vm<IN_FLIGHT>(); else wait_vm<0>();
        __builtin_amdgcn_s_waitcnt(0xC07F);                   // lgkmcnt(0): this wave HOLDS every fragment of stage s (the ones its second MFMA group uses included)
        __builtin_amdgcn_s_barrier();                         // stage s + 1 has landed; every wave is past its reads of stage s: its slot may be refilled
        read_frags(nslot, nxt);
        __builtin_amdgcn_sched_barrier(0);
        if (fill && early) dma(m0, n0, s + NS, slot);
      }
      __builtin_amdgcn_sched_barrier(0);
      mfma_rows(cur, acc, P1, TM);
      __builtin_amdgcn_sched_barrier(0);
      if (more && fill && !early) dma(m0, n0, s + NS, slot);
      slot = nslot;
    };
    int s = 0;
    for (; s + NS + 1 < nk; s += 2) {
      kstep(std::true_type{}, F0, F1, s);
      kstep(std::true_type{}, F1, F0, s + 1);
    }
    for (; s < nk; s += 2) {
      kstep(std::false_type{}, F0, F1, s);
      kstep(std::false_type{}, F1, F0, s + 1);
    }

    // ---- next tile's first stages go out under this tile's epilogue
    lds_barrier();                     // every wave is past its last fragment read
#ifdef SUMK_DIAG
    unsigned long long t1 = 0;
    if constexpr (VAR == 3) t1 = __builtin_amdgcn_s_memtime();
#endif
    int m1 = 0, n1 = 0;
    const int next_tile = next_valid(tile + gridDim.x, m1, n1);
    const bool has_next = next_tile >= 0;
    if (has_next) {
#pragma unroll
      for (int s = 0; s < NS; ++s) dma(m1, n1, s, s);
    }

    // ---- epilogue
    const int row_w = m0 + wm * WTM, col_w = n0 + wn * 64;
    if constexpr (EPI == PW_F32) {
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const int col = col_w + tn * 32 + li;
        const float bv = a.bias ? a.bias[col] : 0.f;            // (kernel-uniform: the input projection of the LSTM scorers adds its summed biases)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = row_w + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (row < a.M) {
              float v = acc[tm][tn][r] + bv;
              if (a.R) v += a.R[(int64_t)row * a.ldr + col];
              if (a.relu) v = (v < 0.f) ? 0.f : v;            // NaN-propagating like torch.relu
              a.C[(int64_t)row * a.ldc + col] = v;
            }
          }
      }
    } else if constexpr (EPI == PW_PLANES || EPI == PW_RES_MOM_PLANES) {
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const int m = row_w + tm * 32 + li;
        if constexpr (EPI == PW_RES_MOM_PLANES) {
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float v = acc[tm][tn][r]; s1 += v; s2 += v * v; }
          s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
          if (lh == 0 && m < a.M) reinterpret_cast<float2*>(a.moments)[(int64_t)m * (a.N >> 6) + (col_w >> 6)] = make_float2(s1, s2);
        }
        if (m < (EPI == PW_PLANES ? a.o_store_rows : a.M)) {
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              u32x2 pl[NP];
              f32x4 v = f32x4{acc[tm][tn][4 * g], acc[tm][tn][4 * g + 1], acc[tm][tn][4 * g + 2], acc[tm][tn][4 * g + 3]};
              if constexpr (EPI == PW_PLANES) {
                if (a.bias) { const float4 b = *reinterpret_cast<const float4*>(a.bias + col_w + tn * 32 + 8 * g + 4 * lh); v += f32x4{b.x, b.y, b.z, b.w}; }
                if (a.relu) {
#pragma unroll
                  for (int c = 0; c < 4; ++c) v[c] = (v[c] < 0.f) ? 0.f : v[c];
                }
              }
              split4<NP>(v, pl);
              const int kb = (col_w >> 4) + tn * 2 + (g >> 1), h = g & 1;
              char* const op = a.O + ((int64_t)(kb * NP) * 2 + h) * a.o_rp16 + (int64_t)m * 16 + 8 * lh;
#pragma unroll
              for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x2*>(op + (int64_t)p * 2 * a.o_rp16) = pl[p];      // (non-temporal stores here: measured, no gain -- profiles/r06_pw_nt_stores_probe.txt)
            }
        }
      }
    } else if constexpr (EPI == PW_RES_F32) {
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const int m = row_w + tm * 32 + li;
        if (m < a.M) {
          float* const cp = a.C + (int64_t)m * a.ldc + col_w + 4 * lh;
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int g = 0; g < 4; ++g)
              *reinterpret_cast<float4*>(cp + tn * 32 + 8 * g) = make_float4(acc[tm][tn][4 * g], acc[tm][tn][4 * g + 1], acc[tm][tn][4 * g + 2], acc[tm][tn][4 * g + 3]);
        }
      }
    } else {   // PW_HEAD
      float* const cst = reinterpret_cast<float*>(lds + NS * STAGE);     // [c1 | bias | gw] x 256 columns of this tile
      if (tid < 256) {
        cst[tid] = a.ln_c1[n0 + tid]; cst[256 + tid] = a.bias[n0 + tid]; cst[512 + tid] = a.gw[n0 + tid];
      }
      lds_barrier();
      // one 32-row tile at a time (its accumulators die with it); the column constants are re-read from LDS per tile
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const int m = row_w + tm * 32 + li;
        const float2 st = reinterpret_cast<const float2*>(a.ln_stats)[min(m, a.M - 1)];
        const float mean = st.x, rstd = st.y;
        float s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int cl = wn * 64 + tn * 32 + 8 * g + 4 * lh;
            const float4 c1 = *reinterpret_cast<const float4*>(cst + cl), bi = *reinterpret_cast<const float4*>(cst + 256 + cl),
                         gw = *reinterpret_cast<const float4*>(cst + 512 + cl);
            const float c1v[4] = {c1.x, c1.y, c1.z, c1.w}, biv[4] = {bi.x, bi.y, bi.z, bi.w}, gwv[4] = {gw.x, gw.y, gw.z, gw.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              float v = rstd * (acc[tm][tn][4 * g + c] - mean * c1v[c]) + biv[c];
              v = (v < 0.f) ? 0.f : v;            // NaN-propagating like torch.relu
              s1 += v; s2 += v * v; s3 += v * gwv[c];
            }
          }
        s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32); s3 += __shfl_xor(s3, 32);
        if (lh == 0 && m < a.M) reinterpret_cast<float4*>(a.head_part)[(int64_t)m * (a.N >> 6) + (col_w >> 6)] = make_float4(s1, s2, s3, 0.f);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#ifdef SUMK_DIAG
    if constexpr (VAR == 3) { const unsigned long long t2 = __builtin_amdgcn_s_memtime(); t_loop += t1 - t0; t_epi += t2 - t1; n_tiles += 1; }
#endif
    if (!has_next) break;
    tile = next_tile; m0 = m1; n0 = n1;
  }
#ifdef SUMK_DIAG
  if constexpr (VAR == 3 && EPI == PW_F32) {     // diagnostic build only: the stamps OVERWRITE the first floats of C (scripts/pw_bench.py reads them)
    __syncthreads();
    if (tid == 0) {
      unsigned long long* o = reinterpret_cast<unsigned long long*>(a.C) + (size_t)blockIdx.x * 8;
      o[0] = __builtin_amdgcn_s_memtime() - t_start; o[1] = t_loop; o[2] = t_epi; o[3] = n_tiles;
      o[4] = __builtin_amdgcn_s_memrealtime() - rt_start; o[5] = __builtin_amdgcn_s_getreg((31 << 11) | 20 /* XCC_ID */);
    }
  }
#endif
}

std::atomic<uint64_t> g_attr_done[64];      // per kernel instance (index below): bit d = device d has the dynamic-LDS opt-in

template <int NP, int BM, int EPI, int NS, int VAR>
int launch_one(const PwArgs& a, int inst, hipStream_t s) {
  constexpr int LDS = NS * (2 * NP * (BM + 256) * 16) + PW_CONST_BYTES;
  const void* fn = (const void*)gemm_pw_kernel<NP, BM, EPI, NS, VAR>;
  int dev = 0;
  SUMK_HIP(hipGetDevice(&dev));
  const uint64_t bit = 1ull << (dev & 63);
  if (!(g_attr_done[inst].load(std::memory_order_acquire) & bit)) {
    SUMK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    g_attr_done[inst].fetch_or(bit, std::memory_order_release);
  }
  const int grid = a.xcd_map ? 256 : std::min(a.total_tiles, 256);
  hipLaunchKernelGGL((gemm_pw_kernel<NP, BM, EPI, NS, VAR>), dim3(grid), dim3(512), LDS, s, a);
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}

template <int NP, int NS>
int launch_np(PwEpi epi, const PwArgs& a, int variant, hipStream_t s) {
  constexpr int base = NP == 3 ? 0 : 16;
  switch (epi) {
    case PW_F32:
      switch (variant) {
        case 10: return launch_one<NP, 192, PW_F32, NS, 0>(a, base + 4, s);      // (probe: waves 0-3 issue their DMA early)
        case 1: return launch_one<NP, 192, PW_F32, NS, 1>(a, base + 5, s);
        case 3: return launch_one<NP, 192, PW_F32, NS, 3>(a, base + 6, s);
        default: return launch_one<NP, 192, PW_F32, NS, 2>(a, base + 0, s);
      }
    case PW_PLANES: return launch_one<NP, 192, PW_PLANES, NS, 2>(a, base + 1, s);
    case PW_RES_MOM_PLANES: return launch_one<NP, 192, PW_RES_MOM_PLANES, NS, 2>(a, base + 2, s);
    case PW_HEAD: return launch_one<NP, 192, PW_HEAD, NS, 2>(a, base + 3, s);
    case PW_RES_F32: return launch_one<NP, 192, PW_RES_F32, NS, 2>(a, base + 7, s);
  }
  set_error("gemm_pw: bad epilogue %d", (int)epi);
  return SUMK_ERR_ARG;
}

// ------------------------------------------------------------------------------------------- fp32 -> KB planes
// One block: 64 rows x 128 columns.  Rows are read coalesced (512-byte row segments) into LDS, then lane = row writes, per 8-column chunk
// and plane, the 16-byte chunk of its row: 64 lanes = 1 KiB contiguous of one sub-array.  Rows in [rows, pitch) are written as zeros.
template <int NP>
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ src, int64_t rows, int K, int ld, char* __restrict__ dst, int64_t rp16) {
  __shared__ float tile[64][132];
  const int tid = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * 64;
  const int c0 = blockIdx.y * 128;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int i = tid + 256 * j, r = i >> 5, c4 = (i & 31) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + r < rows && c0 + c4 < K) v = *reinterpret_cast<const float4*>(src + (r0 + r) * ld + c0 + c4);
    *reinterpret_cast<float4*>(&tile[r][c4]) = v;
  }
  __syncthreads();
  const int r = tid & 63, cg = tid >> 6;            // 4 chunk groups x 4 chunks of 8 columns
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int ch = cg * 4 + q, k = c0 + ch * 8;
    if (k >= K) continue;
    const float4 lo = *reinterpret_cast<const float4*>(&tile[r][ch * 8]), hi = *reinterpret_cast<const float4*>(&tile[r][ch * 8 + 4]);
    u32x2 pa[NP], pb[NP];
    split4<NP>(f32x4{lo.x, lo.y, lo.z, lo.w}, pa);
    split4<NP>(f32x4{hi.x, hi.y, hi.z, hi.w}, pb);
    char* const op = dst + ((int64_t)((k >> 4) * NP) * 2 + ((k >> 3) & 1)) * rp16 + (r0 + r) * 16;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
      *reinterpret_cast<u32x4*>(op + (int64_t)p * 2 * rp16) = u32x4{pa[p].x, pa[p].y, pb[p].x, pb[p].y};
    }
  }
}

// ------------------------------------------------------------------------------------------- long sequences (round 6): V^T planes, softmax -> alpha planes
// The per-video products of a LONG video (T in the thousands: BASELINE config 5) are big enough to run on gemm_pw_kernel itself, one launch
// per (video, product): logits = Q K^T from the planes of Q and K (rows of the video, k = D); context = alpha V from the planes of alpha
// (rows = queries, k = key index) and of V^T (rows = the D columns, k = key index).  Two producers of planes are new:

// fp32 (T x Dm, leading dimension ld) -> KB planes of its TRANSPOSE (rows = the Dm columns, k = t, zero for t in [T, Kp)).
// One block: 64 output rows (columns d0 ..) x 128 k (rows t0 ..) through an LDS tile; lane = output row writes 16-byte chunks, as split_planes_kernel.
template <int NP>
__global__ __launch_bounds__(256) void split_planes_t_kernel(const float* __restrict__ src, int T, int Dm, int ld, int Kp, char* __restrict__ dst, int64_t rp16) {
  __shared__ float tile[128][68];
  const int tid = threadIdx.x;
  const int d0 = blockIdx.x * 64, t0 = blockIdx.y * 128;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int i = tid + 256 * j, tt = i >> 4, c4 = (i & 15) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t0 + tt < T && d0 + c4 < Dm) v = *reinterpret_cast<const float4*>(src + (int64_t)(t0 + tt) * ld + d0 + c4);
    *reinterpret_cast<float4*>(&tile[tt][c4]) = v;
  }
  __syncthreads();
  const int r = tid & 63, cg = tid >> 6;
  if (d0 + r >= (int)((Dm + 63) / 64 * 64)) return;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int ch = cg * 4 + q, k = t0 + ch * 8;
    if (k >= Kp) continue;
    f32x4 lo, hi;
#pragma unroll
    for (int c = 0; c < 4; ++c) { lo[c] = tile[ch * 8 + c][r]; hi[c] = tile[ch * 8 + 4 + c][r]; }
    u32x2 pa[NP], pb[NP];
    split4<NP>(lo, pa);
    split4<NP>(hi, pb);
    char* const op = dst + ((int64_t)((k >> 4) * NP) * 2 + ((k >> 3) & 1)) * rp16 + (int64_t)(d0 + r) * 16;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
      *reinterpret_cast<u32x4*>(op + (int64_t)p * 2 * rp16) = u32x4{pa[p].x, pa[p].y, pb[p].x, pb[p].y};
    }
  }
}

// Row softmax of one video's RAW logits E (T x T valid, leading dimension ldE; scale and the masks of vasnet.py:118-128 applied here, as
// vasnet_softmax_kernel does) -> KB planes of alpha (rows = queries, k = key, zero for keys in [T, Kp)); alpha never exists in fp32.
// (1) softmax_stats_kernel: one WAVE per row finds {max, sum of exp} in one pass (running maximum with rescaling) -- T / 4 blocks;
// (2) softmax_split_kernel: one block per (64 rows, 128 keys) tile normalises it into LDS and lane = row writes the 16-byte chunks, as
//     split_planes_kernel -- (T / 64) x (Kp / 128) blocks.  (A first version did both in one block per 64 rows: 157 blocks of 4 waves for
//     T = 10 000 -- 2.0 ms per video, a sixth of the step; profiles/r06_stress_x6_first_kernel_stats.csv.)
__global__ __launch_bounds__(256) void softmax_stats_kernel(const float* __restrict__ E, int T, int64_t ldE, float scale, int ignore_self, int aperture, float2* __restrict__ stats) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= T) return;
  const float* e = E + (int64_t)i * ldE;
  float m = -INFINITY, sum = 0.f;
  // sixteen loads in flight per lane (a row of T = 10 000 is 40 KB: one load at a time ran at 2 TB/s), then a block maximum first so that the
  // running sum is rescaled once per 16 values instead of at every new maximum
  for (int j0 = 0; j0 < T; j0 += 64 * 16) {
    float x[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int j = j0 + 64 * u + lane;
      x[u] = j < T ? e[j] : 0.f;
    }
    float bm = -INFINITY;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int j = j0 + 64 * u + lane;
      x[u] = j < T ? masked_logit(x[u], scale, i, j, ignore_self, aperture) : -INFINITY;
      bm = fmaxf(bm, x[u]);
    }
    if (bm > -INFINITY) {
      if (bm > m) { sum *= expf(m - bm); m = bm; }            // (m = -inf: sum is 0 and expf(-inf) = 0)
#pragma unroll
      for (int u = 0; u < 16; ++u) sum += expf(x[u] - m);     // (masked values: expf(-inf) = 0)
    }
  }
  float M = m;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) M = fmaxf(M, __shfl_xor(M, o, 64));
  float part = (m > -INFINITY) ? sum * expf(m - M) : 0.f;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
  if (lane == 0) stats[i] = make_float2(M, part);
}

template <int NP>
__global__ __launch_bounds__(256) void softmax_split_kernel(const float* __restrict__ E, int T, int64_t ldE, int Kp, float scale, int ignore_self, int aperture,
                                                            const float2* __restrict__ stats, char* __restrict__ dst, int64_t rp16) {
  __shared__ float tile[64][132];
  const int tid = threadIdx.x;
  const int i0 = blockIdx.x * 64, c0 = blockIdx.y * 128;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int idx = tid + 256 * j, rr = idx >> 5, c4 = (idx & 31) * 4;
    const int i = i0 + rr;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (i < T && c0 + c4 < T) {
      const float4 x = *reinterpret_cast<const float4*>(E + (int64_t)i * ldE + c0 + c4);      // (ldE % 4 == 0; what columns [T, ldE) hold is masked below)
      const float xv[4] = {x.x, x.y, x.z, x.w};
      const float2 st = stats[i];
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (c0 + c4 + c < T) v[c] = expf(masked_logit(xv[c], scale, i, c0 + c4 + c, ignore_self, aperture) - st.x) / st.y;     // (an all-masked row: NaN, like torch's softmax)
    }
    *reinterpret_cast<float4*>(&tile[rr][c4]) = make_float4(v[0], v[1], v[2], v[3]);
  }
  __syncthreads();
  const int r = tid & 63, cg = tid >> 6;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int ch = cg * 4 + q, k = c0 + ch * 8;
    if (k >= Kp) continue;
    const float4 lo = *reinterpret_cast<const float4*>(&tile[r][ch * 8]), hi = *reinterpret_cast<const float4*>(&tile[r][ch * 8 + 4]);
    u32x2 pa[NP], pb[NP];
    split4<NP>(f32x4{lo.x, lo.y, lo.z, lo.w}, pa);
    split4<NP>(f32x4{hi.x, hi.y, hi.z, hi.w}, pb);
    char* const op = dst + ((int64_t)((k >> 4) * NP) * 2 + ((k >> 3) & 1)) * rp16 + (int64_t)(i0 + r) * 16;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
      *reinterpret_cast<u32x4*>(op + (int64_t)p * 2 * rp16) = u32x4{pa[p].x, pa[p].y, pb[p].x, pb[p].y};
    }
  }
}

}  // namespace

int split_planes_t(const float* src, int T, int Dm, int ld, int np, void* planes, int Kp, hipStream_t stream) {
  SUMK_ARG(src && planes && T >= 1 && Dm >= 4 && Dm % 4 == 0 && ld >= Dm && ld % 4 == 0 && Kp >= T && Kp % 16 == 0 && (np == 2 || np == 3), "split_planes_t: bad arguments");
  SUMK_ARG(((uintptr_t)planes & 15) == 0 && ((uintptr_t)src & 15) == 0, "split_planes_t: 16-byte aligned buffers");
  const int64_t rp = pw_rows_pitch(Dm);
  const dim3 grid((unsigned)(rp / 64), (unsigned)((Kp + 127) / 128));
  if (np == 3) hipLaunchKernelGGL(split_planes_t_kernel<3>, grid, dim3(256), 0, stream, src, T, Dm, ld, Kp, (char*)planes, rp * 16);
  else hipLaunchKernelGGL(split_planes_t_kernel<2>, grid, dim3(256), 0, stream, src, T, Dm, ld, Kp, (char*)planes, rp * 16);
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}

int softmax_planes(const float* E, int T, int64_t ldE, int np, void* planes, int Kp, float scale, int ignore_self, int aperture, float* stats, hipStream_t stream) {
  SUMK_ARG(E && planes && stats && T >= 1 && ldE >= T && ldE % 4 == 0 && Kp >= T && Kp % 16 == 0 && (np == 2 || np == 3), "softmax_planes: bad arguments");
  SUMK_ARG(((uintptr_t)planes & 15) == 0 && ((uintptr_t)E & 15) == 0 && ((uintptr_t)stats & 7) == 0, "softmax_planes: aligned buffers");
  const int64_t rp = pw_rows_pitch(T);
  hipLaunchKernelGGL(softmax_stats_kernel, dim3((unsigned)((T + 3) / 4)), dim3(256), 0, stream, E, T, ldE, scale, ignore_self, aperture, (float2*)stats);
  const dim3 grid((unsigned)(rp / 64), (unsigned)((Kp + 127) / 128));
  if (np == 3) hipLaunchKernelGGL(softmax_split_kernel<3>, grid, dim3(256), 0, stream, E, T, ldE, Kp, scale, ignore_self, aperture, (const float2*)stats, (char*)planes, rp * 16);
  else hipLaunchKernelGGL(softmax_split_kernel<2>, grid, dim3(256), 0, stream, E, T, ldE, Kp, scale, ignore_self, aperture, (const float2*)stats, (char*)planes, rp * 16);
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}

int split_planes_at(const float* src, int64_t rows, int K, int ld, int np, void* planes, int64_t row0, int64_t total_rows, hipStream_t stream) {
  SUMK_ARG(src && planes && rows >= 1 && K >= 16 && K % 16 == 0 && ld >= K && ld % 4 == 0 && (np == 2 || np == 3), "split_planes: bad arguments (K %% 16, ld %% 4, 2 or 3 planes)");
  SUMK_ARG(((uintptr_t)planes & 15) == 0 && ((uintptr_t)src & 15) == 0, "split_planes: 16-byte aligned buffers");
  SUMK_ARG(row0 >= 0 && row0 % 64 == 0 && row0 + rows <= total_rows && (row0 + rows == total_rows || rows % 64 == 0), "split_planes: a stacked part starts and ends on a 64-row boundary");
  const int64_t rp = pw_rows_pitch(total_rows);
  const dim3 grid((unsigned)((rows + 63) / 64), (unsigned)((K + 127) / 128));
  char* const dst = (char*)planes + row0 * 16;
  if (np == 3) hipLaunchKernelGGL(split_planes_kernel<3>, grid, dim3(256), 0, stream, src, rows, K, ld, dst, rp * 16);
  else hipLaunchKernelGGL(split_planes_kernel<2>, grid, dim3(256), 0, stream, src, rows, K, ld, dst, rp * 16);
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}
int split_planes(const float* src, int64_t rows, int K, int ld, int np, void* planes, hipStream_t stream) {
  return split_planes_at(src, rows, K, ld, np, planes, 0, rows, stream);
}
// `rows` rows into a plane array whose pitch was chosen for `pitch_rows` >= rows rows (an operand whose tiles read past its last row):
// rows up to the next multiple of 64 are written as zeros, the rest of the pitch is left as it is
int split_planes_pitched(const float* src, int64_t rows, int K, int ld, int np, void* planes, int64_t pitch_rows, hipStream_t stream) {
  SUMK_ARG(src && planes && rows >= 1 && pitch_rows >= rows && K >= 16 && K % 16 == 0 && ld >= K && ld % 4 == 0 && (np == 2 || np == 3), "split_planes_pitched: bad arguments");
  SUMK_ARG(((uintptr_t)planes & 15) == 0 && ((uintptr_t)src & 15) == 0, "split_planes_pitched: 16-byte aligned buffers");
  const int64_t rp = pw_rows_pitch(pitch_rows);
  const dim3 grid((unsigned)((rows + 63) / 64), (unsigned)((K + 127) / 128));
  if (np == 3) hipLaunchKernelGGL(split_planes_kernel<3>, grid, dim3(256), 0, stream, src, rows, K, ld, (char*)planes, rp * 16);
  else hipLaunchKernelGGL(split_planes_kernel<2>, grid, dim3(256), 0, stream, src, rows, K, ld, (char*)planes, rp * 16);
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}

int launch_gemm_pw(PwEpi epi, const PwLaunch& g, hipStream_t stream) {
  SUMK_ARG(g.A && g.B && pw_ok(g.M, g.N, g.K, g.a_rows, g.b_rows, g.np), "gemm_pw: M=%d N=%d K=%d planes=%d is not eligible (N %% 256, K %% 32, K >= 128, plane arrays < 2 GiB)", g.M, g.N, g.K, g.np);
  SUMK_ARG(g.a_rows >= g.M && g.b_rows >= g.N, "gemm_pw: operand planes have fewer rows than the product reads");
  PwArgs a;
  a.A = (const char*)g.A; a.B = (const char*)g.B;
  a.a_rp16 = (uint32_t)(pw_rows_pitch(g.a_rows) * 16); a.b_rp16 = (uint32_t)(pw_rows_pitch(g.b_rows) * 16);
  a.M = g.M; a.N = g.N; a.K = g.K;
  a.tiles_m = (g.M + 191) / 192; a.tiles_n = g.N / 256;
  static const int map_env = SUMK_TUNE_ENV("SUMK_PW_XCD_MAP") ? atoi(SUMK_TUNE_ENV("SUMK_PW_XCD_MAP")) : 2;      // (diagnostic build only)
  a.xcd_map = (a.tiles_n % 4 == 0 && a.tiles_m >= 16) ? map_env : 0;
  a.total_tiles = a.xcd_map == 2 ? 8 * ((a.tiles_m + 7) / 8) * a.tiles_n : a.xcd_map == 1 ? 8 * ((a.tiles_m + 1) / 2) * (a.tiles_n / 4) : a.tiles_m * a.tiles_n;
  a.C = g.C; a.ldc = g.ldc; a.O = (char*)g.O; a.o_rp16 = pw_rows_pitch(g.o_rows) * 16;
  a.o_store_rows = (int32_t)std::max<int64_t>(g.M, std::min<int64_t>(g.o_store_rows, std::min<int64_t>(pw_rows_pitch(g.o_rows), pw_rows_pitch(g.a_rows)))); a.R = g.R; a.ldr = g.ldr; a.moments = g.moments;
  a.bias = g.bias; a.gw = g.gw; a.ln_c1 = g.ln_c1; a.ln_stats = g.ln_stats; a.head_part = g.head_part; a.relu = g.relu;
  switch (epi) {
    case PW_F32: SUMK_ARG(g.C && g.ldc >= g.N && (!g.R || g.ldr >= g.N), "gemm_pw: fp32 output missing / residual pitch"); break;
    case PW_PLANES: SUMK_ARG(g.O && g.o_rows >= g.M, "gemm_pw: plane output missing"); break;
    case PW_RES_MOM_PLANES: SUMK_ARG(g.O && g.o_rows >= g.M && g.R && g.ldr >= g.N && g.ldr % 4 == 0 && g.moments, "gemm_pw: residual / moments / plane output missing"); break;
    case PW_HEAD: SUMK_ARG(g.bias && g.gw && g.ln_c1 && g.ln_stats && g.head_part, "gemm_pw: head epilogue operands missing"); break;
    case PW_RES_F32: SUMK_ARG(g.C && g.ldc >= g.N && g.ldc % 4 == 0 && g.R && g.ldr >= g.N && g.ldr % 4 == 0 && ((uintptr_t)g.C & 15) == 0 && ((uintptr_t)g.R & 15) == 0 && !g.relu,
                              "gemm_pw: residual epilogue needs 16-byte aligned C and R with pitches %% 4 (and no ReLU)"); break;
  }
  SUMK_ARG(!(g.bias && (epi == PW_PLANES || epi == PW_RES_F32)) || ((uintptr_t)g.bias & 15) == 0, "gemm_pw: the bias of the transposed epilogues is read as float4: 16-byte aligned");
  if (g.prof_tag >= 0) prof_begin(g.prof_tag, stream);
  prof_begin(SUMK_PROF_GEMM_ALL, stream);
  // two planes run on the 16x16x32 MFMA shape (gemm_pw16.hip: -13 % on the same operands); three planes do not fit its five-stage ring
  int rc;
  static const bool pw16_on = !(SUMK_TUNE_ENV("SUMK_PW16") && SUMK_TUNE_ENV("SUMK_PW16")[0] == '0');      // (diagnostic build only: scripts/pw16_step_probe.sh)
  if (g.np == 2 && g.K >= 160 && g.variant != 32 && pw16_on) {
    if (a.xcd_map == 1) { a.xcd_map = 2; a.total_tiles = 8 * ((a.tiles_m + 7) / 8) * a.tiles_n; }
    rc = launch_gemm_pw16((int)epi, a, stream);
  } else {
    rc = g.np == 3 ? launch_np<3, 3>(epi, a, g.variant, stream) : launch_np<2, 4>(epi, a, g.variant, stream);
  }
  prof_end(SUMK_PROF_GEMM_ALL, stream);
  if (g.prof_tag >= 0) prof_end(g.prof_tag, stream);
  return rc;
}

}  // namespace sumk

// ------------------------------------------------------------------------------------------- C ABI (tests, probes, host-side caches)
extern "C" size_t sumk_planes_bytes(int64_t rows, int32_t K, int32_t n_planes) {
  if (rows < 1 || K < 16 || K % 16 != 0 || (n_planes != 2 && n_planes != 3)) return 0;
  return sumk::pw_planes_bytes(rows, K, n_planes);
}

extern "C" int sumk_split_planes(const float* src, int64_t rows, int32_t K, int32_t ld, int32_t n_planes, void* planes, void* stream) {
  return sumk::split_planes(src, rows, K, ld, n_planes, planes, (hipStream_t)stream);
}

extern "C" int sumk_gemm_planes(const void* A_planes, int64_t a_rows, const void* B_planes, int64_t b_rows, float* C, int32_t M, int32_t N,
                                int32_t K, int32_t n_planes, int32_t variant, void* stream) {
  using namespace sumk;
  PwLaunch g;
  g.A = A_planes; g.B = B_planes; g.a_rows = a_rows; g.b_rows = b_rows; g.M = M; g.N = N; g.K = K; g.np = n_planes; g.C = C; g.ldc = N;
  g.variant = variant;
  return launch_gemm_pw(PW_F32, g, (hipStream_t)stream);
}
