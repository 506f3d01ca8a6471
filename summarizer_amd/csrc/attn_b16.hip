// Fused attention strips for the bf16-source training step (SUMK_PRECISION_BF16; reference: summarizer/models/vasnet.py:118-131 and its
// autograd backward).  One workgroup owns 64 query rows of ONE video (T <= 320 keys) and runs, without leaving the CU,
//   GEMM-1   S^T[key][query] = B16[key][:] . A16[query][:]      K = D, both operands K-contiguous (Q, K of the forward; dCTX, V of the backward)
//   row op   forward : alpha = softmax(mask(S * scale)), P = bf16(dropout(alpha))
//            backward: dS = scale * alpha * (dropout'(dP) - sum_j dropout'(dP)_j alpha_j)
//   GEMM-2   O^T[col][query] = C16[key][col]^T . P[query][key]   K = T, C in its natural [key][col] order (V forward, K backward)
// The logits never leave the accumulators: the separate launches this replaces (Q.K^T GEMM -> softmax kernel -> alpha.V GEMM, and dC.V^T GEMM
// -> softmax backward kernel -> dS.K GEMM) wrote and re-read the (T x T) fp32 block of every video three times and ran their MFMAs at 0.11 of
// the bf16 peak (profiles/r03_pmc_bf16_train_gemms.json).  What still reaches HBM: alpha (fp32, the forward's second output and what the
// backward re-reads), bf16(dropout(alpha)) for the dV product, bf16(dS) for the dK product -- those two contract over the QUERY rows of all
// strips and stay separate GEMM launches (csrc/vasnet.hip).  Rounding points are those of the separate launches: bf16 operands, fp32
// accumulation, fp32 row arithmetic, bf16 P / dS / CTX / dQ.
//
// Both products are computed TRANSPOSED (keys resp. output columns along the MFMA M axis, queries along N): a lane of the 32x32x16 result
// then holds 4 consecutive keys (columns) of ONE query, so the row statistics reduce in registers (+ one cross-half shuffle + one LDS
// exchange between the two waves that share a query tile), P goes to LDS and HBM in 8-byte pieces, and the outputs are 8-byte stores.
//
// LDS (96 KB, one workgroup per CU; 250 strips on the S-TVSum batch = one per CU):
//   [0, 46080)      key-side k-tile of GEMM-1: 320 rows x (64 bf16 + 16 B)          | phase 2: the [64 keys][256 cols + 64 B] tile of C
//   [46080, 55296)  query-side k-tile: 64 rows x (64 bf16 + 16 B)
//   [55296, 97280)  P: 64 queries x (320 bf16 + 16 B)   (pitch 41 x 16 B: conflict-free ds_read_b128)
//   [97280, 98304)  row-statistic exchange
// Workgroup ids are dealt round-robin over the 8 XCDs, so id = xcd + 8 k: the strips of one video are given ids of the same residue and
// its K / V rows (1.3 MB bf16) are fetched into ONE L2 instead of eight.
#include "gemm_regstage.h"
#include <math.h>
#include <cstdlib>

namespace sumk {

namespace {

#ifndef AT_EXP
#define AT_EXP 0
#endif
constexpr int AT_ROWS = 64, AT_TMAX = 320, AT_BK = 64;
constexpr int AT_KCP = 2 * AT_BK + 16;         // K-contiguous image: bytes per row (144)
constexpr int AT_PP = 2 * AT_TMAX + 16;        // P image: bytes per query row (656)
constexpr int AT_NCH = 256;                    // GEMM-2: output columns per pass
constexpr int AT_MCP = 2 * AT_NCH + 64;        // [key][col] image: bytes per key row (576)
constexpr int AT_STAGE1 = (AT_TMAX + AT_ROWS) * AT_KCP;   // one GEMM-1 k-tile: key rows then query rows (55296)
constexpr int AT_SP = 0;                                   // P
constexpr int AT_ST = AT_ROWS * AT_PP;                     // two staging buffers (GEMM-1 k-tiles; GEMM-2 tiles of C)
constexpr int AT_RED = AT_ST + 2 * AT_STAGE1;
constexpr int AT_LDS = AT_RED + 4 * 64 * 4;
constexpr int AT_R1 = 4, AT_R2 = 4;                        // register rings: k-tiles of GEMM-1 / tiles of C in flight
static_assert(AT_BK * AT_MCP <= AT_STAGE1, "a C tile of phase 2 fits a GEMM-1 staging buffer");
static_assert(AT_LDS <= 160 * 1024, "LDS");

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x2 pack_bf16x4(float a, float b, float c, float d) {
  return __builtin_bit_cast(u32x2, __builtin_convertvector(f32x4v{a, b, c, d}, bf16x4v));
}

// NJ = T64 / 64: key tiles of 32 per wave in GEMM-1 (a wave owns tiles kg, kg + 2, ...) and 64-key tiles of C in GEMM-2.  A template
// parameter, not a run-time bound: with `if (j < nj)` around every fragment read and MFMA the compiler put a branch and a full
// lgkmcnt(0) wait in front of each MFMA (74 us per launch); the workgroup picks its instance with one switch.
__device__ unsigned long long attn_stamps[2][512][8];
#define STAMP(k) do { if (threadIdx.x == 0) attn_stamps[BWD ? 1 : 0][blockIdx.x & 511][k] = __builtin_amdgcn_s_memtime(); } while (0)
template <bool BWD, int NJ>
__device__ __forceinline__ void attn_strip_body(const AttnStripArgs& a, const SeqInfo& si, const int strip, char* const lds) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int T = si.T, i0 = strip * AT_ROWS;
  const Drop drop = drop_resolve(a.drop);
  constexpr int T64 = NJ * 64;
  const int D = a.D;
  const int rows = min(AT_ROWS, T - i0);

  // ---------------------------------------------------------------------------------------------- GEMM-1
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned short*>(a.A16 + (int64_t)(si.row0 + i0) * a.lda), (short)0, ((rows - 1) * a.lda + D) * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned short*>(a.B16 + (int64_t)si.row0 * a.ldb), (short)0, ((T - 1) * a.ldb + D) * 2, 0x00020000);
  const int lr = tid >> 3, lc = tid & 7;      // this thread's 16-byte chunks: rows lr + 32 p, k chunk lc (rows past the video read as zero)
  const int voa = (lr * a.lda + lc * 8) * 2, vob = (lr * a.ldb + lc * 8) * 2;
  const int sta = 32 * a.lda * 2, stb = 32 * a.ldb * 2;
  const int KT1 = D / AT_BK;                  // a multiple of AT_R1 (D % 256 == 0)
  // Register ring, AT_R1 k-tiles deep: with one workgroup per CU nothing else hides the L2 / HBM latency (one tile in flight: 100 us per
  // launch), so a tile's loads are issued four tiles before its LDS write.
  u32x4 rq[AT_R1][2], rk[AT_R1][2 * NJ];
  // (requests past the last tile are ISSUED too, with a vector offset outside the descriptor -- they return zero without touching memory.
  //  Under an `if` the compiler's wait counts assume the branch not taken: vmcnt(7..0) in front of a tile's LDS write with 32 newer
  //  loads behind it, i.e. the whole ring drained at every tile.)
  constexpr int AT_OOB = 0x40000000;
  auto gload1 = [&](int kt, u32x4 (&q)[2], u32x4 (&k)[2 * NJ]) {
    const int oob = kt < KT1 ? 0 : AT_OOB;
#pragma unroll
    for (int p = 0; p < 2; ++p) q[p] = (u32x4)__builtin_amdgcn_raw_buffer_load_b128(rA, voa + p * sta + oob, kt * (AT_BK * 2), 0);
#pragma unroll
    for (int p = 0; p < 2 * NJ; ++p) k[p] = (u32x4)__builtin_amdgcn_raw_buffer_load_b128(rB, vob + p * stb + oob, kt * (AT_BK * 2), 0);
  };
  // LDS write of a staged k-tile, in four parts (chunks c = part mod 4) so that a tile's writes sit between the MFMA groups of the tile before it
  auto swrite1_part = [&](char* buf, const u32x4 (&q)[2], const u32x4 (&k)[2 * NJ], int part) {
#pragma unroll
    for (int c = 0; c < 2 + 2 * NJ; ++c) {
      if ((c & 3) != part) continue;
      if (c < 2) *reinterpret_cast<u32x4*>(buf + (AT_TMAX + lr + 32 * c) * AT_KCP + lc * 16) = q[c];
      else *reinterpret_cast<u32x4*>(buf + (lr + 32 * (c - 2)) * AT_KCP + lc * 16) = k[c - 2];
    }
  };
  const int qt = wave & 1, kg = wave >> 1;    // this wave: query tile qt, key tiles kg + 2 j
  const int fq = (AT_TMAX + qt * 32 + li) * AT_KCP + 16 * lh;     // fragment offsets inside a staging buffer
  const int fk = (kg * 32 + li) * AT_KCP + 16 * lh;
  struct Frag1 { bf16x8 bq, ak[NJ]; };
  auto read1 = [&](const char* buf, int ks, Frag1& f) {
    f.bq = *reinterpret_cast<const bf16x8*>(buf + fq + 32 * ks);
#pragma unroll
    for (int j = 0; j < NJ; ++j) f.ak[j] = *reinterpret_cast<const bf16x8*>(buf + fk + j * 64 * AT_KCP + 32 * ks);
  };
  f32x16 acc[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // One wave per SIMD: nothing but the instruction order hides LDS latency, so the order is pinned (sched_barrier).  Iteration kt: the
  // fragments of step ks + 1 and a quarter of tile kt + 1's LDS write are issued, THEN the MFMAs of step ks; tile kt + 5's loads follow
  // the last step, one workgroup barrier closes the iteration (two staging buffers: tile kt + 1 is written while tile kt is read).
  STAMP(0);
#pragma unroll
  for (int st = 0; st < AT_R1; ++st) gload1(st, rq[st], rk[st]);
#pragma unroll
  for (int part = 0; part < 4; ++part) swrite1_part(lds + AT_ST, rq[0], rk[0], part);
  gload1(AT_R1, rq[0], rk[0]);
  __syncthreads();
  for (int kt0 = 0; kt0 < KT1; kt0 += AT_R1) {
#pragma unroll
    for (int st = 0; st < AT_R1; ++st) {
      const char* const cur = lds + AT_ST + (st & 1) * AT_STAGE1;
      char* const nxt = lds + AT_ST + ((st + 1) & 1) * AT_STAGE1;
      constexpr int R1M = AT_R1 - 1;
      const int ns = (st + 1) & R1M;
      Frag1 f[2];
      if (!(AT_EXP & 2) || kt0 + st == 0) read1(cur, 0, f[0]);
#pragma unroll
      for (int ks = 0; ks < AT_BK / 16; ++ks) {
        if (!(AT_EXP & 2)) if (ks + 1 < AT_BK / 16) read1(cur, ks + 1, f[(ks + 1) & 1]);
        if (!(AT_EXP & 4)) swrite1_part(nxt, rq[ns], rk[ns], ks);
        __builtin_amdgcn_sched_barrier(0);
        if (!(AT_EXP & 1)) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[ks & 1].ak[j], f[ks & 1].bq, acc[j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (!(AT_EXP & 8)) gload1(kt0 + st + 1 + AT_R1, rq[ns], rk[ns]);
      if (!(AT_EXP & 16)) __syncthreads();
    }
  }

  STAMP(1);
  // ---------------------------------------------------------------------------------------------- GEMM-2 operands: the first column pass in flight under the row op
  const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned short*>(a.C16 + (int64_t)si.row0 * a.ldc), (short)0, ((T - 1) * a.ldc + D) * 2, 0x00020000);
  const int cr = tid >> 5, cc = tid & 31;     // 16-byte chunks of the [64 keys][256 cols] tile: key rows cr + 8 p, column chunk cc
  const int NC = D / AT_NCH;
  u32x4 rc[NJ][8];                            // ring slot = key tile: tile (nc + 1, kt) is requested when tile (nc, kt) has gone to LDS
  auto gload2 = [&](int nc, int kt, u32x4 (&c)[8]) {  // key rows in the VECTOR offset: rows past T are out of the descriptor's range and read as zero
    const int oob = nc < NC ? 0 : AT_OOB;
#pragma unroll
    for (int p = 0; p < 8; ++p)
      c[p] = (u32x4)__builtin_amdgcn_raw_buffer_load_b128(rC, ((kt * 64 + cr + 8 * p) * a.ldc + nc * AT_NCH + cc * 8) * 2 + oob, 0, 0);
  };
#pragma unroll
  for (int kt = 0; kt < NJ; ++kt) gload2(0, kt, rc[kt]);

  // ---------------------------------------------------------------------------------------------- row op
  // acc[j][r]: key = (kg + 2 j) * 32 + 8 (r >> 2) + 4 lh + (r & 3), query = qt * 32 + li
  const int qi = qt * 32 + li, i = i0 + qi;
  const bool row_ok = i < T;
  float* const red = reinterpret_cast<float*>(lds + AT_RED);
  const uint64_t drow = (uint64_t)(si.row0 + i) << 20;
  float* const erow = a.E + si.eoff + (int64_t)i * si.ldE;
  unsigned short* const prow = a.P16 + si.e16off + (int64_t)i * T64;
  char* const lrow = lds + AT_SP + qi * AT_PP;
  if constexpr (!BWD) {
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = (kg + 2 * j) * 32 + 8 * (r >> 2) + 4 * lh + (r & 3);
        const float e = key < T ? masked_logit(acc[j][r], a.scale, i, key, a.ignore_self, a.aperture) : -INFINITY;
        acc[j][r] = e;
        m = fmaxf(m, e);
      }
    m = fmaxf(m, __shfl_xor(m, 32));
    if (lh == 0) red[kg * 64 + qi] = m;
    __syncthreads();
    m = fmaxf(red[qi], red[64 + qi]);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = (kg + 2 * j) * 32 + 8 * (r >> 2) + 4 * lh + (r & 3);
        const float p = key < T ? expf(acc[j][r] - m) : 0.f;
        acc[j][r] = p;
        sum += p;
      }
    sum += __shfl_xor(sum, 32);
    if (lh == 0) red[128 + kg * 64 + qi] = sum;
    __syncthreads();
    sum = red[128 + qi] + red[192 + qi];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int k0 = (kg + 2 * j) * 32 + 8 * g + 4 * lh;
        float al[4], ad[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          al[c] = k0 + c < T ? acc[j][4 * g + c] / sum : 0.f;
          ad[c] = (drop.thr && k0 + c < T) ? drop_apply(drop, 0, drow | (uint64_t)(k0 + c), al[c]) : al[c];
        }
        const u32x2 pk = pack_bf16x4(ad[0], ad[1], ad[2], ad[3]);
        if (row_ok) {
          if (k0 < si.ldE) *reinterpret_cast<float4*>(erow + k0) = make_float4(al[0], al[1], al[2], al[3]);
          *reinterpret_cast<u32x2*>(prow + k0) = pk;
        }
        *reinterpret_cast<u32x2*>(lrow + k0 * 2) = row_ok ? pk : u32x2{0u, 0u};
      }
  } else {
    float dot = 0.f;
    float al[NJ][16];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int k0 = (kg + 2 * j) * 32 + 8 * g + 4 * lh;
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row_ok && k0 < si.ldE) t = *reinterpret_cast<const float4*>(erow + k0);
        al[j][4 * g] = t.x; al[j][4 * g + 1] = t.y; al[j][4 * g + 2] = t.z; al[j][4 * g + 3] = t.w;
      }
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = (kg + 2 * j) * 32 + 8 * (r >> 2) + 4 * lh + (r & 3);
        float d = acc[j][r];
        if (drop.thr) d = drop_apply(drop, 0, drow | (uint64_t)key, d);
        acc[j][r] = d;
        dot += d * al[j][r];
      }
    dot += __shfl_xor(dot, 32);
    if (lh == 0) red[kg * 64 + qi] = dot;
    __syncthreads();
    dot = red[qi] + red[64 + qi];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int k0 = (kg + 2 * j) * 32 + 8 * g + 4 * lh;
        float v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = al[j][4 * g + c] * (acc[j][4 * g + c] - dot) * a.scale;
        const u32x2 pk = row_ok ? pack_bf16x4(v[0], v[1], v[2], v[3]) : u32x2{0u, 0u};
        if (row_ok) *reinterpret_cast<u32x2*>(prow + k0) = pk;
        *reinterpret_cast<u32x2*>(lrow + k0 * 2) = pk;
      }
  }

  // ---------------------------------------------------------------------------------------------- GEMM-2
  // wave w: output columns nc * 256 + w * 64 + [0, 64) (two M tiles), both query tiles
  const int fc = (8 * lh + ((lane & 15) >> 2)) * AT_MCP + 2 * (wave * 64 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3));
  const char* const fp = lds + AT_SP + li * AT_PP + 16 * lh;
  STAMP(2);
  __syncthreads();                            // P complete
  STAMP(3);
  struct Frag2 { bf16x8 fa[2], fb[2]; };
  auto read2 = [&](const char* buf, int kt, int ks, Frag2& f) {
#pragma unroll
    for (int t = 0; t < 2; ++t) f.fa[t] = tr_frag(buf + fc + 16 * ks * AT_MCP + 64 * t, AT_MCP);
#pragma unroll
    for (int u = 0; u < 2; ++u) f.fb[u] = *reinterpret_cast<const bf16x8*>(fp + u * 32 * AT_PP + (kt * 64 + ks * 16) * 2);
  };
  auto swrite2_part = [&](char* buf, const u32x4 (&c)[8], int part) {
#pragma unroll
    for (int p = 2 * part; p < 2 * part + 2; ++p) *reinterpret_cast<u32x4*>(buf + (cr + 8 * p) * AT_MCP + cc * 16) = c[p];
  };
  // same schedule as GEMM-1: tile (nc, kt) is read from one staging buffer while the next tile of the (nc, kt) sequence goes into the other
#pragma unroll
  for (int part = 0; part < 4; ++part) swrite2_part(lds + AT_ST, rc[0], part);
  gload2(1, 0, rc[0]);
  __syncthreads();
  int par = 0;
  for (int nc = 0; nc < NC; ++nc) {
    f32x16 o[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][u][r] = 0.f;
#pragma unroll
    for (int kt = 0; kt < NJ; ++kt) {
      const char* const cur = lds + AT_ST + par * AT_STAGE1;
      char* const nxt = lds + AT_ST + (par ^ 1) * AT_STAGE1;
      par ^= 1;
      constexpr int NJ1 = NJ;
      const int nk = (kt + 1) % NJ1;          // ring slot (= key tile) of the next tile; its column pass: nc + (kt + 1 == NJ)
      Frag2 f[2];
      read2(cur, kt, 0, f[0]);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (ks + 1 < 4) read2(cur, kt, ks + 1, f[(ks + 1) & 1]);
        swrite2_part(nxt, rc[nk], ks);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int u = 0; u < 2; ++u) o[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[ks & 1].fa[t], f[ks & 1].fb[u], o[t][u], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      gload2(nc + (kt + 1 == NJ1 ? 2 : 1), nk, rc[nk]);
      __syncthreads();
    }
    // o[t][u][r]: column = nc * 256 + wave * 64 + t * 32 + 8 (r >> 2) + 4 lh + (r & 3), query = u * 32 + li
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int q = u * 32 + li;
      unsigned short* const orow = a.O16 + (int64_t)(si.row0 + i0 + q) * a.ldo + nc * AT_NCH + wave * 64 + 4 * lh;
      if (q < rows) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *reinterpret_cast<u32x2*>(orow + t * 32 + 8 * g) = pack_bf16x4(o[t][u][4 * g], o[t][u][4 * g + 1], o[t][u][4 * g + 2], o[t][u][4 * g + 3]);
      }
    }
    if (nc == 0) STAMP(4);
  }
  STAMP(5);
  if (threadIdx.x == 0) attn_stamps[BWD ? 1 : 0][blockIdx.x & 511][6] = NJ;
}

template <bool BWD>
__global__ __launch_bounds__(256) void attn_strip_kernel(AttnStripArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int s = (slot / a.strips) * 8 + xcd, strip = slot - (slot / a.strips) * a.strips;
  if (s >= a.n_seq) return;
  const SeqInfo si = a.seq[s];
  if (strip * AT_ROWS >= si.T) return;
  switch ((si.T + 63) >> 6) {
    case 1: attn_strip_body<BWD, 1>(a, si, strip, lds); break;
    case 2: attn_strip_body<BWD, 2>(a, si, strip, lds); break;
    case 3: attn_strip_body<BWD, 3>(a, si, strip, lds); break;
    case 4: attn_strip_body<BWD, 4>(a, si, strip, lds); break;
    default: attn_strip_body<BWD, 5>(a, si, strip, lds); break;
  }
}

}  // namespace

// T <= 320 keys per video, D a whole number of 256-column passes, byte offsets inside 31 bits
bool attn_strip_ok(int t_max, int D, int64_t rows, int ld_max) {
  (void)rows;
  return t_max >= 1 && t_max <= AT_TMAX && D % AT_NCH == 0 && D >= AT_NCH && ((int64_t)(t_max + 64) * ld_max + D) * 2 < ((int64_t)1 << 31);
}

int launch_attn_strip(bool backward, const AttnStripArgs& a_in, hipStream_t stream) {
  const AttnStripArgs& a = a_in;
  const void* fn = backward ? (const void*)attn_strip_kernel<true> : (const void*)attn_strip_kernel<false>;
  static bool attr_set[2] = {false, false};
  if (!attr_set[backward ? 1 : 0]) {
    SUMK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, AT_LDS));
    attr_set[backward ? 1 : 0] = true;
  }
  const unsigned grid = (unsigned)(8 * ((a.n_seq + 7) / 8) * a.strips);
  if (backward) hipLaunchKernelGGL(attn_strip_kernel<true>, dim3(grid), dim3(256), AT_LDS, stream, a);
  else hipLaunchKernelGGL(attn_strip_kernel<false>, dim3(grid), dim3(256), AT_LDS, stream, a);
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}

extern "C" int sumk_attn_stamps_tmp(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(attn_stamps), sizeof(unsigned long long) * 2 * 512 * 8);
}
}  // namespace sumk
