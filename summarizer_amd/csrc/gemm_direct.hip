// Small-batch exact-fp32 MFMA GEMM: ONE 32x32 output tile per workgroup, K split over the workgroup's waves, operands loaded
// straight into the MFMA fragment registers.
//
// What it is for.  The reference scores ONE video per forward and takes ONE optimiser step per video (vasnet.py:193-212,
// models/__init__.py:45-54): T ~ 300 rows.  At that size a projection is 80 tiles of 64x64 with a 32-k-tile dependent chain each
// (30 us per GEMM on the staged kernels, one wave per SIMD, every latency exposed) and Q.K^T is 25 tiles on 256 CUs.  What such a
// problem needs is waves, not tiles per wave:
//   * a tile is 32x32 = one v_mfma_f32_32x32x2_f32 accumulator (16 VGPRs) -- 320 tiles for a (300 x 1024) output, 100 for Q.K^T;
//   * the workgroup is G waves (1 <= G <= 8) and wave g multiplies k-tiles [g q, (g+1) q) of the SAME tile: the launch has
//     tiles x G waves (1 000 - 2 500 on the 1 024 SIMDs) and a wave's dependent chain is K / G long;
//   * no LDS staging and no barrier in the k-loop: lane (i, h) of the 32x32x2 MFMA consumes A[i][k], k = 8 kk + 4 h + j -- for a
//     K-contiguous operand that is ONE 16-byte load per kk, for a [k][row] operand four dword loads that 32 lanes coalesce into a
//     128-byte line.  fp32 MFMA runs at the vector-ALU rate (64 cycles per instruction), so a k-tile's 8-32 load instructions hide
//     behind its 16 MFMAs (1 024 cycles) as long as the loads are issued early: two register stages, one k-tile ahead, and 2-4
//     waves per SIMD;
//   * the G partial tiles meet in LDS: waves 1 .. G-1 park their accumulators, one barrier, wave 0 adds them in wave order
//     (deterministic) and runs the epilogue -- no global round trip, no tickets (an in-launch split-K over blocks was built first:
//     its store -> ticket -> reload chain cost 3-8 us per GEMM).
// Layouts NT / NN / TN, problem table as everywhere (GemmProb; tile_start / tiles_n count 32x32 tiles), per-entry choice among
// four B / C pointers (pad_[SK_BSEL] / pad_[SK_CSEL]: the three projection matrices, the three weight-gradient tensors of one
// launch), run-time epilogue: alpha acc | acc + R | relu(acc + bias) | C + alpha acc.
// Summation order: k ascending inside a wave (the staged kernels' order), then wave 0 + wave 1 + ... -- results differ from the
// single-chain kernels by fp32 re-association (~1e-7 relative); both are deterministic.
#include "gemm_device.h"
#include <algorithm>

namespace sumk {

namespace {
constexpr int DT = 32, DBK = 32;                   // tile edge, k-tile depth
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
}

// fragments of one k-tile: [kk][j] = operand value at k = 8 kk + 4 h + j for this lane's row / column
struct Frag { float a[4][4], b[4][4]; };

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(512, 4) void gemm_direct_kernel(GemmKArgs ka, int G) {
  static_assert(A_KC || !B_KC, "layouts: NT, NN, TN");
  __shared__ __attribute__((aligned(16))) float red[7 * 16 * 64];        // partial tiles of waves 1 .. G-1: [wave - 1][q][lane] float4
  const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile = blockIdx.x;

  // ---- which problem, which tile (wave-uniform scalar work)
  int lo = 0, hi = ka.nprob - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (prob_tile_start(ka.probs, mid) <= tile) lo = mid; else hi = mid - 1;
  }
  const GemmProb P = load_prob(ka.probs, lo);
  const cptr32 pq = (cptr32)(uintptr_t)(ka.probs + lo);
  const int bs = pq[17 + SK_BSEL], cs = pq[17 + SK_CSEL];
  const float* Bp = bs == 0 ? ka.B[0] : bs == 1 ? ka.B[1] : bs == 2 ? ka.B[2] : ka.B[3];
  float* Cp = cs == 0 ? ka.C : cs == 1 ? ka.Csel[1] : cs == 2 ? ka.Csel[2] : ka.Csel[3];
  const int local = tile - P.tile_start;
  const int m0 = (local / P.tiles_n) * DT, n0 = (local % P.tiles_n) * DT;
  const int M = P.M, N = P.N, K = P.K, lda = P.lda, ldb = P.ldb;

  // ---- this lane's operand pointers.  Rows / columns past M / N are clamped: they only feed outputs that are never stored.
  const int ar = min(m0 + li, M - 1), br = min(n0 + li, N - 1);
  const float* pa = A_KC ? ka.A + P.a_off + (int64_t)ar * lda + 4 * lh : ka.A + P.a_off + ar + (int64_t)(4 * lh) * lda;
  const float* pb = B_KC ? Bp + P.b_off + (int64_t)br * ldb + 4 * lh : Bp + P.b_off + br + (int64_t)(4 * lh) * ldb;

  // ---- this wave's k-tiles
  const int nk = (K + DBK - 1) / DBK;
  const int per = (nk + G - 1) / G;
  const int kt0 = min(wave * per, nk), kt1 = min(kt0 + per, nk);
  const bool tail = (K % DBK) != 0;                 // the LAST k-tile of the problem is ragged

  // the ragged last k-tile (k0 + 32 > K): element-wise, addresses clamped into the operand, k >= K contributes zeros
  auto load_tail = [&](Frag& f, int kt) {
    const int k0 = kt * DBK;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = k0 + 8 * kk + 4 * lh + j, kc = min(k, K - 1);
        const float av = A_KC ? pa[kc - 4 * lh] : pa[(int64_t)(kc - 4 * lh) * lda];
        const float bv = B_KC ? pb[kc - 4 * lh] : pb[(int64_t)(kc - 4 * lh) * ldb];
        f.a[kk][j] = k < K ? av : 0.f;
        f.b[kk][j] = k < K ? bv : 0.f;
      }
  };
  // a full k-tile: unconditional loads, nothing that waits
  auto load = [&](Frag& f, int kt) {
    const int k0 = kt * DBK;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if constexpr (A_KC) {
        const float4 v = *reinterpret_cast<const float4*>(pa + k0 + 8 * kk);
        f.a[kk][0] = v.x; f.a[kk][1] = v.y; f.a[kk][2] = v.z; f.a[kk][3] = v.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) f.a[kk][j] = pa[(int64_t)(k0 + 8 * kk + j) * lda];
      }
      if constexpr (B_KC) {
        const float4 v = *reinterpret_cast<const float4*>(pb + k0 + 8 * kk);
        f.b[kk][0] = v.x; f.b[kk][1] = v.y; f.b[kk][2] = v.z; f.b[kk][3] = v.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) f.b[kk][j] = pb[(int64_t)(k0 + 8 * kk + j) * ldb];
      }
    }
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  auto mfma = [&](const Frag& f) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[kk][j], f.b[kk][j], acc, 0, 0, 0);
  };

  // ---- k-loop: two register stages, a k-tile's loads are issued a whole k-tile (16 MFMAs = 1 024 cycles, times the waves sharing the
  // SIMD) ahead of its MFMAs.  The steady loop is straight-line (two k-tiles per trip, static stage registers, no branch between a
  // load and its use -- with the ragged-tile branch inside, hipcc's wait counts at the joins made every k-tile wait for the NEXT
  // k-tile's loads); the ragged last k-tile runs after it.  (Three stages -- 96 fragment registers -- spill at the 128-VGPR budget
  // that keeps four waves per SIMD.)
  {
    const bool my_tail = tail && kt1 == nk && kt0 < kt1;     // this wave owns the ragged k-tile
    const int kf = my_tail ? kt1 - 1 : kt1;                   // its full k-tiles: [kt0, kf)
    Frag f0, f1;
    int kt = kt0;
    if (kt < kf) {
      // (sched_barrier: left alone, hipcc sinks each stage's loads next to their MFMAs and waits for them there)
      load(f0, kt);
      for (; kt + 2 < kf; kt += 2) {
        load(f1, kt + 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma(f0);
        __builtin_amdgcn_sched_barrier(0);
        load(f0, kt + 2);
        __builtin_amdgcn_sched_barrier(0);
        mfma(f1);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (kt + 1 < kf) { load(f1, kt + 1); __builtin_amdgcn_sched_barrier(0); mfma(f0); mfma(f1); }
      else mfma(f0);
    }
    if (my_tail) { load_tail(f0, nk - 1); mfma(f0); }
  }

  // ---- the G partial tiles: waves 1 .. G-1 -> LDS, wave 0 adds them in wave order
  if (G > 1) {
    if (wave > 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(red + (((wave - 1) * 4 + q) * 64 + lane) * 4) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
    }
    __syncthreads();
    if (wave > 0) return;
    for (int g = 1; g < G; ++g) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(red + (((g - 1) * 4 + q) * 64 + lane) * 4);
        acc[4 * q] += v.x; acc[4 * q + 1] += v.y; acc[4 * q + 2] += v.z; acc[4 * q + 3] += v.w;
      }
    }
  }

  // ---- epilogue (wave 0): C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  const int col = n0 + li;
  if (col >= N) return;
  const int rows_left = M - m0 - 4 * lh;
  const int epi = ka.sk_epi;
  const float bias = epi == EPI_BIAS_RELU ? ka.bias0[0][col] : 0.f;
  float* cp = Cp + P.c_off + (int64_t)(m0 + 4 * lh) * P.ldc + col;
  const float* rp = ka.R + P.r_off + (int64_t)(m0 + 4 * lh) * P.ldr + col;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int ro = (r & 3) + 8 * (r >> 2);
    if (ro < rows_left) {
      float v = acc[r] * ka.alpha;
      if (epi == EPI_RESIDUAL) v = acc[r] + rp[(int64_t)ro * P.ldr];
      else if (epi == EPI_BIAS_RELU) { v = acc[r] + bias; v = (v < 0.f) ? 0.f : v; }      // NaN-propagating, like torch.relu
      else if (epi == EPI_ACCUM) v = cp[(int64_t)ro * P.ldc] + ka.alpha * acc[r];
      cp[(int64_t)ro * P.ldc] = v;
    }
  }
}

// G for a launch of `tiles` 32x32 tiles contracting over K: about two waves per SIMD chip-wide, a wave keeps at least two k-tiles
int gemm_direct_waves(int tiles, int K) {
  const int nk = (K + DBK - 1) / DBK;
  int G = tiles > 0 ? (2048 + tiles - 1) / tiles : 1;
  G = std::min(G, std::max(1, nk / 2));
  return std::max(1, std::min(G, 8));
}

int launch_gemm_direct(GemmLayout layout, const GemmKArgs& ka, int tiles, int G, hipStream_t s) {
  SUMK_ARG(G >= 1 && G <= 8, "gemm_direct: %d waves per tile", G);
  SUMK_ARG(ka.sk_epi == EPI_NONE || ka.sk_epi == EPI_RESIDUAL || ka.sk_epi == EPI_BIAS_RELU || ka.sk_epi == EPI_ACCUM, "gemm_direct: epilogue %d", ka.sk_epi);
  const dim3 grid(tiles), block(64 * G);
  if (layout == GEMM_NT) hipLaunchKernelGGL((gemm_direct_kernel<true, true>), grid, block, 0, s, ka, G);
  else if (layout == GEMM_NN) hipLaunchKernelGGL((gemm_direct_kernel<true, false>), grid, block, 0, s, ka, G);
  else hipLaunchKernelGGL((gemm_direct_kernel<false, false>), grid, block, 0, s, ka, G);
  return SUMK_OK;
}

}  // namespace sumk
