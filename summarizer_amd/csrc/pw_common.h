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
// Every wave computes the same answer: 64 videos per step, one per lane -- ONE 32-byte load per lane (the whole SeqInfo, so that the chosen video's entry is
// read out of a lane's registers instead of by a second, dependent load), an inclusive scan of the block counts over DPP (six VALU adds; by ds_bpermute
// shuffles the scan alone cost a block ~0.5 us), a ballot.  Every result is wave-uniform by construction and comes out of readlane / readfirstlane: left as
// VGPR values the compiler treats every address derived from them as divergent and wraps each LDS-DMA instruction in a waterfall loop.
__device__ __forceinline__ int wave_inclusive_scan(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);      // row_shr:1 (lanes without a source keep 0)
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);      // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);      // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);      // row_shr:8   -> scanned inside each row of 16
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);      // row_bcast:15 into rows 1 and 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);      // row_bcast:31 into rows 2 and 3
  return v;
}

template <typename F>
__device__ __forceinline__ bool locate_block(const SeqInfo* seq, int n_seq, F blocks_of, SeqInfo& si, int& sv, int& sub) {
  static_assert(sizeof(SeqInfo) == 32, "SeqInfo is read as two 16-byte halves");
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63;
  i32x4 lo, hi;          // this lane's video: {eoff (2 dwords), row0, T}, {ldE, pad_, e16off (2 dwords)}
  int cnt;
  auto chunk = [&](int v0) {                                // loads the lane's video of this 64-video step; returns the inclusive scan of the block counts
    const i32x4* p = reinterpret_cast<const i32x4*>(seq + min(v0 + lane, n_seq - 1));
    lo = p[0]; hi = p[1];
    cnt = v0 + lane < n_seq ? blocks_of(lo[3]) : 0;
    return wave_inclusive_scan(cnt);
  };
  int inc = chunk(0);
  int total = __builtin_amdgcn_readlane(inc, 63);
  for (int v0 = 64; v0 < n_seq; v0 += 64) total += __builtin_amdgcn_readlane(chunk(v0), 63);
  const int G = (total + 7) >> 3, slot = blockIdx.x >> 3;
  const int L = (blockIdx.x & 7) * G + slot;
  if (slot >= G || L >= total) return false;
  int base = 0;
  for (int v0 = 0; v0 < n_seq; v0 += 64) {
    if (n_seq > 64) inc = chunk(v0);                         // (one step: the scan of the first pass is still in registers)
    const int tot = __builtin_amdgcn_readlane(inc, 63);
    if (L < base + tot) {
      const int first = __ffsll((unsigned long long)__ballot(base + inc > L)) - 1;
      sv = v0 + first;
      sub = L - base - __builtin_amdgcn_readlane(inc - cnt, first);
      auto rl = [&](int x) { return (uint32_t)__builtin_amdgcn_readlane(x, first); };
      si.eoff = (int64_t)(((uint64_t)rl(lo[1]) << 32) | rl(lo[0])); si.row0 = (int32_t)rl(lo[2]); si.T = (int32_t)rl(lo[3]);
      si.ldE = (int32_t)rl(hi[0]); si.pad_ = (int32_t)rl(hi[1]); si.e16off = (int64_t)(((uint64_t)rl(hi[3]) << 32) | rl(hi[2]));
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
