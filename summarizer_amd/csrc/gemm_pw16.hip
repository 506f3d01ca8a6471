// The plane-aware wide GEMM of gemm_pw.hip on the OTHER bf16 MFMA shape, v_mfma_f32_16x16x32_bf16, for TWO planes (bf16x3 scoring).
// Why: under bf16 MFMA load the chip holds ~1.7 GHz, and the 16x16x32 shape sustains more FLOP/s at equal cycles per FLOP (MI355X_MICROARCH.md,
// DVFS give-back item 7).  Measured on one box, same operands (profiles/r05_pw16_mfma_shape_probe.txt): QKV projection 213-215 -> 185-202 us,
// (12 003 x 1024) x (1024 x 1024)^T 74.6 -> 64.6 us (-13 %).
// Same operand format, same 192 x 256 tile, same eight waves of 96 x 64 (6 x 4 tiles of 16 x 16), same 1-KiB LDS-DMA pieces -- but an MFMA spans TWO k16
// stages (lane quarters 0, 1 read the k halves of stage 2t, quarters 2, 3 of stage 2t + 1), so the loop advances by stage PAIRS over a ring of FIVE k16
// stages: the pair being multiplied, the next pair (landed) and one stage in flight; the two slots a pair frees are refilled with stages 2t + 5, 2t + 6.
// Fragments: B (8) double-buffered, A streamed in two halves of three row tiles: the barrier sits between the halves, when every wave holds all of its
// pair's fragments in registers.  Three planes do not fit (five 42-KB stages), so bf16x6 stays on the 32x32x16 kernel.
// Accumulation order: a term is summed over 32 k before the next term starts -- results agree with the 32x32x16 / in-loop kernels to rounding (tested
// against float64 and against them), not bit for bit.  Epilogues as gemm_pw.hip (PwEpi); C/D map of the shape: col = lane & 15, row = 4 (lane >> 4) + reg.
#include "pw_common.h"
#include <atomic>

namespace sumk {

namespace {

typedef float f32x4acc __attribute__((ext_vector_type(4)));

template <int EPI>
__global__ __launch_bounds__(512) void gemm_pw16_kernel(PwArgs a) {
  constexpr int NP = 2, BM = 192, BN = 256, NSUB = 4, NS = 5;
  constexpr bool SWAP = EPI != PW_F32;
  constexpr int A_BYTES = NSUB * BM * 16, B_BYTES = NSUB * BN * 16, STAGE = A_BYTES + B_BYTES;
  constexpr int PA = BM / 64, PB = BN / 64, PPS = PA + PB;
  static_assert(NS * STAGE + PW_CONST_BYTES <= 160 * 1024, "ring");
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  const bool dma_wave = wave < PPS, dma_a = wave < PA;
  const int blk = dma_a ? wave : wave - PA;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(dma_a ? a.A : a.B), (short)0, 0x80000000u, 0x00020000);
  const int rp16 = (int)(dma_a ? a.a_rp16 : a.b_rp16);
  const int lds_blk = dma_a ? blk * 1024 : A_BYTES + blk * 1024;
  const int lsub = dma_a ? BM * 16 : BN * 16;
  const int vlane = lane * 16, k_step = NSUB * rp16;
  auto dma = [&](int m0, int n0, int kb, int slot) {
    if (!dma_wave) return;
    char* const st = lds + slot * STAGE + lds_blk;
    const int g0 = kb * k_step + ((dma_a ? m0 : n0) + blk * 64) * 16;
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_vptr)(st + sub * lsub), 16, vlane, g0 + sub * rp16, 0, 0);
  };
  // lane (row r16, k quarter kq): k half h = kq & 1 of stage (kq >> 1) of the pair
  const int fa = ((kq & 1) * BM + wm * 96 + r16) * 16, fb = A_BYTES + ((kq & 1) * BN + wn * 64 + r16) * 16;
  struct BFrags { bf16x8 b[NP][4]; };
  struct AFrags { bf16x8 a[NP][3]; };
  auto slot_base = [&](int s0, int s1) { return (kq >> 1) ? s1 * STAGE : s0 * STAGE; };
  auto read_b = [&](int sb, BFrags& f) {
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int t = 0; t < 4; ++t) f.b[p][t] = *reinterpret_cast<const bf16x8*>(lds + sb + fb + p * 2 * BN * 16 + t * 256);
  };
  auto read_a = [&](int sb, int half, AFrags& f) {
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int t = 0; t < 3; ++t) f.a[p][t] = *reinterpret_cast<const bf16x8*>(lds + sb + fa + p * 2 * BM * 16 + (half * 3 + t) * 256);
  };
  f32x4acc acc[6][4];
  auto mfma_half = [&](const AFrags& fa_, const BFrags& fb_, int half, int t_lo, int t_hi) {
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      if (t < t_lo || t >= t_hi) continue;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        f32x4acc c = acc[half * 3 + t][u];
        // (A plane, B plane) = (lo, hi), (hi, lo), (hi, hi): smallest products first, the term order of the other plane kernels
        if constexpr (SWAP) {
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb_.b[0][u], fa_.a[1][t], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb_.b[1][u], fa_.a[0][t], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb_.b[0][u], fa_.a[0][t], c, 0, 0, 0);
        } else {
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_.a[1][t], fb_.b[0][u], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_.a[0][t], fb_.b[1][u], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_.a[0][t], fb_.b[0][u], c, 0, 0, 0);
        }
        acc[half * 3 + t][u] = c;
      }
    }
  };

  // tile walk: the row-band XCD map of gemm_pw.hip (xcd_map 2) or the plain order
  auto decode = [&](int t, int& m0, int& n0) -> bool {
    int mt, nt;
    if (a.xcd_map == 2) {
      const int x = t & 7, j = t >> 3, rx = (a.tiles_m + 7) >> 3, per = rx * 4;
      const int grp = j / per, jj = j - grp * per, r = jj >> 2;
      mt = x * rx + r; nt = grp * 4 + (jj & 3);
      if (grp >= (a.tiles_n >> 2) || mt >= a.tiles_m) return false;
    } else {
      if (t >= a.total_tiles) return false;
      mt = t / a.tiles_n; nt = t - mt * a.tiles_n;
    }
    m0 = mt * BM; n0 = nt * BN;
    return true;
  };
  auto next_valid = [&](int t, int& m, int& n) -> int {
    for (; t < a.total_tiles; t += gridDim.x)
      if (decode(t, m, n)) return t;
    return -1;
  };
  const int nk = a.K >> 4, npair = nk >> 1;
  int m0 = 0, n0 = 0;
  int tile = next_valid(blockIdx.x, m0, n0);
  if (tile < 0) return;
#pragma unroll
  for (int s = 0; s < NS; ++s) dma(m0, n0, s, s);

  while (true) {
    const int row_w = m0 + wm * 96, col_w = n0 + wn * 64;
    if constexpr (EPI == PW_RES_MOM_PLANES || EPI == PW_RES_F32) {
      // transposed product: lane = row m (r16 of tile i), registers = columns 16 u + 4 kq + (0..3)
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int m = row_w + i * 16 + r16;
        const float* const rp = a.R + (int64_t)min(m, a.M - 1) * a.ldr + col_w + 4 * kq;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float4 v = *reinterpret_cast<const float4*>(rp + u * 16);
          if constexpr (EPI == PW_RES_F32) {
            if (a.bias) { const float4 b = *reinterpret_cast<const float4*>(a.bias + col_w + 4 * kq + u * 16); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
          }
          acc[i][u] = f32x4acc{v.x, v.y, v.z, v.w};
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4acc{0.f, 0.f, 0.f, 0.f};
    }
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    BFrags B0, B1;
    AFrags Alo, Ahi;
    int s0 = 0, s1 = 1;                      // slots of the pair being multiplied
    read_b(slot_base(s0, s1), B0);
    read_a(slot_base(s0, s1), 0, Alo);
    auto pair_step = [&](const BFrags& bc, BFrags& bn, int t) {
      const int sb = slot_base(s0, s1);
      // (the second half's A fragments are requested behind the first row tile's MFMAs, not in front of them: at the loop's back edge the
      //  compiler waits lgkmcnt(0) before the first MFMA, and reads issued ahead of it would be waited for there)
      __builtin_amdgcn_sched_barrier(0);
      mfma_half(Alo, bc, 0, 0, 1);
      __builtin_amdgcn_sched_barrier(0);
      read_a(sb, 1, Ahi);
      __builtin_amdgcn_sched_barrier(0);
      mfma_half(Alo, bc, 0, 1, 3);
      __builtin_amdgcn_sched_barrier(0);
      const bool more = t + 1 < npair;
      const int n0s = s0 + 2 >= NS ? s0 + 2 - NS : s0 + 2, n1s = s1 + 2 >= NS ? s1 + 2 - NS : s1 + 2;
      if (more) {
        // stages 2t + 2, 2t + 3 have landed when at most stage 2t + 4's pieces are outstanding (in issue order); near the end: drain
        if (2 * t + 4 < nk) wait_vm<NSUB>(); else wait_vm<0>();
        __builtin_amdgcn_s_waitcnt(0xC07F);                 // lgkmcnt(0): this wave holds all fragments of pair t
        __builtin_amdgcn_s_barrier();
        const int nb = (kq >> 1) ? n1s * STAGE : n0s * STAGE;
        read_b(nb, bn);
        read_a(nb, 0, Alo);
      }
      __builtin_amdgcn_sched_barrier(0);
      mfma_half(Ahi, bc, 1, 0, 3);
      __builtin_amdgcn_sched_barrier(0);
      if (more) {
        if (2 * t + 5 < nk) dma(m0, n0, 2 * t + 5, s0);
        if (2 * t + 6 < nk) dma(m0, n0, 2 * t + 6, s1);
      }
      s0 = n0s; s1 = n1s;
    };
    for (int t = 0; t < npair; t += 2) {
      pair_step(B0, B1, t);
      if (t + 1 < npair) pair_step(B1, B0, t + 1);
    }
    lds_barrier();
    int m1 = 0, n1 = 0;
    const int next_tile = next_valid(tile + gridDim.x, m1, n1);
    if (next_tile >= 0) {
#pragma unroll
      for (int s = 0; s < NS; ++s) dma(m1, n1, s, s);
    }

    // ---- epilogue
    if constexpr (EPI == PW_F32) {           // lane = column (r16 of tile j), registers = rows 4 kq + (0..3) of tile i
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = col_w + j * 16 + r16;
        const float bv = a.bias ? a.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = row_w + i * 16 + 4 * kq + r;
            if (row < a.M) {
              float v = acc[i][j][r] + bv;
              if (a.R) v += a.R[(int64_t)row * a.ldr + col];
              if (a.relu) v = (v < 0.f) ? 0.f : v;            // NaN-propagating like torch.relu
              a.C[(int64_t)row * a.ldc + col] = v;
            }
          }
      }
    } else if constexpr (EPI == PW_PLANES || EPI == PW_RES_MOM_PLANES) {
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int m = row_w + i * 16 + r16;
        if constexpr (EPI == PW_RES_MOM_PLANES) {
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float v = acc[i][u][r]; s1 += v; s2 += v * v; }
          s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
          s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
          if (kq == 0 && m < a.M) reinterpret_cast<float2*>(a.moments)[(int64_t)m * (a.N >> 6) + (col_w >> 6)] = make_float2(s1, s2);
        }
        if (m < (EPI == PW_PLANES ? a.o_store_rows : a.M)) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {         // columns col_w + 16 u + 4 kq + (0..3): chunk (row m, k-block (col_w >> 4) + u, half kq >> 1), byte 8 (kq & 1)
            u32x2 pl[NP];
            f32x4 v = f32x4{acc[i][u][0], acc[i][u][1], acc[i][u][2], acc[i][u][3]};
            if constexpr (EPI == PW_PLANES) {
              if (a.bias) { const float4 b = *reinterpret_cast<const float4*>(a.bias + col_w + 16 * u + 4 * kq); v += f32x4{b.x, b.y, b.z, b.w}; }
              if (a.relu) {
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = (v[c] < 0.f) ? 0.f : v[c];
              }
            }
            split4<NP>(v, pl);
            const int kb = (col_w >> 4) + u;
            char* const op = a.O + ((int64_t)(kb * NP) * 2 + (kq >> 1)) * a.o_rp16 + (int64_t)m * 16 + 8 * (kq & 1);
#pragma unroll
            for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x2*>(op + (int64_t)p * 2 * a.o_rp16) = pl[p];
          }
        }
      }
    } else if constexpr (EPI == PW_RES_F32) {
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int m = row_w + i * 16 + r16;
        if (m < a.M) {
          float* const cp = a.C + (int64_t)m * a.ldc + col_w + 4 * kq;
#pragma unroll
          for (int u = 0; u < 4; ++u) *reinterpret_cast<float4*>(cp + u * 16) = make_float4(acc[i][u][0], acc[i][u][1], acc[i][u][2], acc[i][u][3]);
        }
      }
    } else {   // PW_HEAD
      float* const cst = reinterpret_cast<float*>(lds + NS * STAGE);     // [c1 | bias | gw] x 256 columns of this tile
      if (tid < 256) {
        cst[tid] = a.ln_c1[n0 + tid]; cst[256 + tid] = a.bias[n0 + tid]; cst[512 + tid] = a.gw[n0 + tid];
      }
      lds_barrier();
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int m = row_w + i * 16 + r16;
        const float2 st = reinterpret_cast<const float2*>(a.ln_stats)[min(m, a.M - 1)];
        const float mean = st.x, rstd = st.y;
        float s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int cl = wn * 64 + u * 16 + 4 * kq;
          const float4 c1 = *reinterpret_cast<const float4*>(cst + cl), bi = *reinterpret_cast<const float4*>(cst + 256 + cl),
                       gw = *reinterpret_cast<const float4*>(cst + 512 + cl);
          const float c1v[4] = {c1.x, c1.y, c1.z, c1.w}, biv[4] = {bi.x, bi.y, bi.z, bi.w}, gwv[4] = {gw.x, gw.y, gw.z, gw.w};
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            float v = rstd * (acc[i][u][c] - mean * c1v[c]) + biv[c];
            v = (v < 0.f) ? 0.f : v;            // NaN-propagating like torch.relu
            s1 += v; s2 += v * v; s3 += v * gwv[c];
          }
        }
        s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16); s3 += __shfl_xor(s3, 16);
        s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32); s3 += __shfl_xor(s3, 32);
        if (kq == 0 && m < a.M) reinterpret_cast<float4*>(a.head_part)[(int64_t)m * (a.N >> 6) + (col_w >> 6)] = make_float4(s1, s2, s3, 0.f);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (next_tile < 0) break;
    tile = next_tile; m0 = m1; n0 = n1;
  }
}

std::atomic<uint64_t> g_attr16[5];

template <int EPI>
int launch16(const PwArgs& a, hipStream_t stream) {
  constexpr int LDS = 5 * (4 * (192 + 256) * 16) + PW_CONST_BYTES;
  int dev = 0;
  SUMK_HIP(hipGetDevice(&dev));
  const uint64_t bit = 1ull << (dev & 63);
  if (!(g_attr16[EPI].load(std::memory_order_acquire) & bit)) {
    SUMK_HIP(hipFuncSetAttribute((const void*)gemm_pw16_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    g_attr16[EPI].fetch_or(bit, std::memory_order_release);
  }
  hipLaunchKernelGGL(gemm_pw16_kernel<EPI>, dim3(a.xcd_map ? 256 : std::min(a.total_tiles, 256)), dim3(512), LDS, stream, a);
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}

}  // namespace

// two planes, K >= 160 (five k16 stages in the prologue); `a` as launch_gemm_pw built it (xcd_map 2 or 0)
int launch_gemm_pw16(int epi, const PwArgs& a, hipStream_t stream) {
  switch (epi) {
    case PW_F32: return launch16<PW_F32>(a, stream);
    case PW_PLANES: return launch16<PW_PLANES>(a, stream);
    case PW_RES_MOM_PLANES: return launch16<PW_RES_MOM_PLANES>(a, stream);
    case PW_HEAD: return launch16<PW_HEAD>(a, stream);
    case PW_RES_F32: return launch16<PW_RES_F32>(a, stream);
  }
  set_error("gemm_pw16: bad epilogue %d", epi);
  return SUMK_ERR_ARG;
}

}  // namespace sumk
