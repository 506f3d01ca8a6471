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
