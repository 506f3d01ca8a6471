// Device helpers shared by the kernels that work on KB planes (gemm_pw.hip, attn_pw.hip): counted waits for LDS-DMA, the plane split.
#pragma once
#include "gemm_regstage.h"

namespace sumk {

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_vptr;

// s_waitcnt vmcnt(N) alone (gfx9 encoding: vmcnt in bits 3:0 and 15:14, expcnt 6:4, lgkmcnt 11:8)
template <int N> __device__ __forceinline__ void wait_vm() { __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14)); }
// workgroup barrier for LDS traffic alone: unlike __syncthreads() it does not drain the LDS-DMA requests in flight (vmcnt)
__device__ __forceinline__ void lds_barrier() { __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_s_barrier(); }

// x -> NP bf16 planes of 4 values (each subtraction exact): the roundings of gemm_regstage.h's split_planes
template <int NP>
__device__ __forceinline__ void split4(f32x4 r, u32x2 (&pl)[NP]) {
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    const bf16x4 b = __builtin_convertvector(r, bf16x4);
    pl[q] = __builtin_bit_cast(u32x2, b);
    if (q + 1 < NP) r = r - __builtin_convertvector(b, f32x4);
  }
}

// Which (video, sub-block) a workgroup owns, from the FLAT list of real blocks (videos in order, blocks_of(T) entries each): the hardware deals workgroup
// ids round-robin over the 8 XCDs, so XCD x = id % 8 takes the contiguous range [x G, (x + 1) G) of the list, G = ceil(total / 8) -- every XCD gets the
// same number of blocks (+- 1) whatever the videos' lengths, and a video's blocks still share an L2.  (Dealing whole VIDEOS round-robin, id % 8 = video % 8,
// gave one XCD 34 blocks for its 32 CUs on the S-TVSum batch at 128-query blocks: a second round on that XCD, the launch twice as long.)
// Every wave computes the same answer: 64 videos per step, one per lane, an inclusive scan by shuffles.
template <typename F>
__device__ __forceinline__ bool locate_block(const SeqInfo* seq, int n_seq, F blocks_of, int& sv, int& sub) {
  const int lane = threadIdx.x & 63;
  auto scan = [&](int v0, int& cnt) {
    const int v = v0 + lane;
    cnt = v < n_seq ? blocks_of(seq[v].T) : 0;
    int inc = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(inc, d); if (lane >= d) inc += t; }
    return inc;
  };
  int total = 0, cnt;
  // (every shuffled value below is wave-uniform by construction; readfirstlane says so to the compiler, which otherwise treats the loop exits -- and with
  //  them every address the caller derives from sv / sub -- as divergent: waterfall loops around each LDS-DMA instruction)
  for (int v0 = 0; v0 < n_seq; v0 += 64) total += __builtin_amdgcn_readfirstlane(__shfl(scan(v0, cnt), 63));
  const int G = (total + 7) >> 3, slot = blockIdx.x >> 3;
  const int L = (blockIdx.x & 7) * G + slot;
  if (slot >= G || L >= total) return false;
  int base = 0;
  for (int v0 = 0; v0 < n_seq; v0 += 64) {
    const int inc = scan(v0, cnt), tot = __builtin_amdgcn_readfirstlane(__shfl(inc, 63));
    if (L < base + tot) {
      const int first = __ffsll((unsigned long long)__ballot(base + inc > L)) - 1;
      sv = __builtin_amdgcn_readfirstlane(v0 + first);
      sub = __builtin_amdgcn_readfirstlane(L - base - __shfl(inc - cnt, first));
      return true;
    }
    base += tot;
  }
  return false;
}

// kernel arguments of the plane GEMMs (gemm_pw.hip: 32x32x16 MFMA; gemm_pw16.hip: 16x16x32 MFMA, two planes)
struct PwArgs {
  const char* A; const char* B;
  uint32_t a_rp16, b_rp16;             // bytes of one (k16 block, plane, half) sub-array
  int32_t M, N, K;
  int32_t tiles_m, tiles_n, total_tiles, xcd_map;
  float* C; int32_t ldc;
  char* O; int64_t o_rp16; int32_t o_store_rows;
  const float* R; int32_t ldr;
  float* moments;
  const float* bias; const float* gw; const float* ln_c1; const float* ln_stats; float* head_part;
  int32_t relu;                        // PW_F32 / PW_PLANES: max(v, 0) after the bias (and before nothing else: a residual and ReLU never meet)
};

// gemm_pw16.hip: the same product on v_mfma_f32_16x16x32_bf16 (two planes; K >= 160); `a` as launch_gemm_pw built it
int launch_gemm_pw16(int epi, const PwArgs& a, hipStream_t stream);

constexpr int PW_CONST_BYTES = 4096;   // PW_HEAD: the tile's 256 columns of c1 / bias / gw

}  // namespace sumk
