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
// LDS (148 KB, one workgroup of eight waves per CU; 250 strips on the S-TVSum batch = one per CU):
//   [0, 147456)       GEMM-1: three stages of (320 key rows + 64 query rows) x 128 B (one 64-deep k-tile), filled by LDS-DMA
//   [0, 41984)        after GEMM-1: P, 64 queries x (320 bf16 + 16 B)   (pitch 41 x 16 B: conflict-free ds_read_b128)
//   [43008, 141312)   GEMM-2: three stages of [64 keys][256 cols] x 2 B
//   [147456, 149504)  row-statistic exchange
// Workgroup ids are dealt round-robin over the 8 XCDs, so id = xcd + 8 k: XCD x takes a contiguous eighth of the flat strip list (locate_block, pw_common.h) --
// the strips of one video run on one XCD and its K / V rows (1.3 MB bf16) are fetched into ONE L2 instead of eight, and every XCD gets the same number of strips.
//
// Where the time goes (T = 320, in-kernel stamps, shader cycles per strip): GEMM-1 33 k, row op 14-18 k, GEMM-2 44 k against 12 k + 10 k of
// MFMA: an iteration of either product costs ~2,000 cycles whatever the number of tiles in flight (one or two: equal), whichever of the
// SIMD's two waves requests first, with or without an L2 prefetch of the tile four steps ahead by a ninth wave (tried, no effect, removed),
// with or without the dropout hash in the loop (-4 us).  The LDS-DMA stream alone runs at 40 B/clk/CU (1,225 cycles per 48 KB k-tile),
// the fragment reads + MFMAs alone at ~1,200, and the two add instead of overlapping: at 64 query rows per strip a k-tile carries
// 54 FLOP per staged byte against the ~100 a CU needs to be matrix-bound.  Not tried: 2 x 3 register blocking per wave (-40 % fragment
// reads), 128-row strips on half the CUs.
#include "pw_common.h"
#include <math.h>
#include <cstdlib>
#include <atomic>

namespace sumk {

namespace {

constexpr int AT_ROWS = 64, AT_TMAX = 320, AT_BK = 64;
constexpr int AT_PP = 2 * AT_TMAX + 16;        // P image: bytes per query row (656)
constexpr int AT_NCH = 256;                    // GEMM-2: output columns per pass
constexpr int AT_ROWB = 2 * AT_BK;             // GEMM-1 staged row: 64 bf16 = 128 B, no pad (LDS-DMA writes 1 KB runs): 16-byte chunks XOR-swizzled
constexpr int AT_CROW = 2 * AT_NCH;            // GEMM-2 staged key row: 256 bf16 = 512 B, 64-byte pieces XOR-swizzled
constexpr int AT_NST = 3;                      // LDS stages per GEMM: two tiles in flight behind the one being multiplied
constexpr int AT_STAGE1 = (AT_TMAX + AT_ROWS) * AT_ROWB;      // 49152
constexpr int AT_STAGE2 = AT_BK * AT_CROW;                    // 32768
constexpr int AT_SP = 0;                                      // P (written after GEMM-1: overlays its first stage)
constexpr int AT_ST2 = 43008;                                 // GEMM-2 stages, behind P
constexpr int AT_RED = 3 * AT_STAGE1;                    // row-statistic exchange
constexpr int AT_LDS = AT_RED + 8 * 64 * 4;
static_assert(AT_ROWS * AT_PP <= AT_ST2 && AT_ST2 + 3 * AT_STAGE2 <= AT_RED && AT_LDS <= 160 * 1024, "LDS map");

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x2 pack_bf16x4(float a, float b, float c, float d) {
  return __builtin_bit_cast(u32x2, __builtin_convertvector(f32x4v{a, b, c, d}, bf16x4v));
}
// (wait_vm<N>, lds_barrier, u32x2, lds_vptr: pw_common.h)


// NJ = T64 / 64 (1 ... 5) is a template parameter, not a run-time bound: with `if (j < nj)` around every fragment read and MFMA the compiler
// put a branch and a full lgkmcnt(0) wait in front of each MFMA; the workgroup picks its instance with one switch.
//
// EIGHT waves (two per SIMD).  GEMM-1: wave = (query tile qt, key group kg of 4), key tiles kg + 4 j, j < NTW = ceil(2 NJ / 4) -- tile
// slots past 2 NJ are dummies (they multiply whatever rows follow in the stage; their logits are masked as key >= T and never stored).
// GEMM-2: wave w owns output columns nc * 256 + 32 w + [0, 32) for both query tiles.
//
// Operands reach LDS by LDS-DMA (buffer_load_dwordx4 ... lds: 1 KB per wave instruction, no VGPRs, no ds_write), three stages per
// product.  How this shape was arrived at (stamps + builds with one piece removed, 4 waves, T = 320):
//   register staging, one tile in flight: 100 us per launch; four tiles in flight 74; branches out of the MFMA stream (NJ template) 68;
//   of a k-tile's 2,300 cycles 1,200 were its 12 ds_write_b128 per wave (13 cycles of issue each, the waves' stores serialised on the
//   store path, nothing hidden behind the 640 cycles of MFMA by a single wave per SIMD);
//   LDS-DMA instead: no stores, but a DMA instruction holds its wave's issue ~100 cycles -- 1,200 cycles per wave and k-tile again,
//   and the dropout hash moved into the k-loop simply added its cycles: with ONE wave per SIMD every stall of that wave idles the
//   matrix pipe.  Hence two waves per SIMD that each issue half the DMAs and half the MFMAs.
// A DMA writes lane-linear runs, so the images carry no pad; bank conflicts are avoided by an XOR swizzle applied on the SOURCE address
// (which 16-byte chunk a lane fetches) and again by the fragment reads.  Rows past the video are CLAMPED to its last row (finite
// data; their logits are masked / their P columns are zero), never out of range.
template <bool BWD, int NJ>
__device__ __forceinline__ void attn_strip_body(const AttnStripArgs& a, const SeqInfo& si, const int strip, char* const lds) {
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T = si.T, i0 = strip * AT_ROWS;
  const Drop drop = drop_resolve(a.drop);
  constexpr int T64 = NJ * 64;
  constexpr int NTW = (2 * NJ + 3) / 4;       // key-tile slots per wave
  const int D = a.D;
  const int rows = min(AT_ROWS, T - i0);

  // ---------------------------------------------------------------------------------------------- GEMM-1
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned short*>(a.A16 + (int64_t)(si.row0 + i0) * a.lda), (short)0, ((rows - 1) * a.lda + D) * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned short*>(a.B16 + (int64_t)si.row0 * a.ldb), (short)0, ((T - 1) * a.ldb + D) * 2, 0x00020000);
  // DMA instruction p of this wave fills rows (8 p + wave) * 8 + [0, 8) of the stage, 8 chunks each: lane -> (row lane / 8, slot lane % 8);
  // the slot holds chunk  slot ^ ((row >> 1) & 7)  of the row  (= slot ^ ((lane >> 4) + 4 (wave & 1)) & 7: the same for every p)
  const int d_row = wave * 8 + (lane >> 3);
  const int d_chunk = (lane & 7) ^ (((lane >> 4) + 4 * (wave & 1)) & 7);
  int vk[NJ];
#pragma unroll
  for (int p = 0; p < NJ; ++p) vk[p] = (min(d_row + 64 * p, T - 1) * a.ldb + d_chunk * 8) * 2;
  const int vq = (min(d_row, rows - 1) * a.lda + d_chunk * 8) * 2;
  constexpr int NI1 = NJ + 1;                 // DMA instructions per wave and k-tile
  auto dma1 = [&](int kt, int stage) {
    char* const dst = lds + stage * AT_STAGE1 + wave * 1024;
#pragma unroll
    for (int p = 0; p < NJ; ++p)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_vptr)(dst + p * 8192), 16, vk[p], kt * (AT_BK * 2), 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_vptr)(dst + T64 * AT_ROWB), 16, vq, kt * (AT_BK * 2), 0, 0);
  };
  const int qt = wave & 1, kg = wave >> 1;    // this wave: query tile qt, key tiles kg + 4 j
  const int sw = (li >> 1) & 7;               // fragment reads: chunk c of row r sits in slot c ^ ((r >> 1) & 7)
  const int fq = (T64 + qt * 32 + li) * AT_ROWB, fk = (kg * 32 + li) * AT_ROWB;
  struct Frag1 { bf16x8 bq, ak[NTW]; };
  auto read1 = [&](const char* buf, int ks, Frag1& f) {
    const int so = ((2 * ks + lh) ^ sw) * 16;
    f.bq = *reinterpret_cast<const bf16x8*>(buf + fq + so);
#pragma unroll
    for (int j = 0; j < NTW; ++j) f.ak[j] = *reinterpret_cast<const bf16x8*>(buf + fk + j * 128 * AT_ROWB + so);
  };
  f32x16 acc[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // Dropout keep bits of this lane's 16 NTW elements (a pure function of seed, row and key: sumk_internal.h dropout_keep), one accumulator
  // register index r per k-tile, computed inside the k-loop where the partner wave's MFMAs cover the three 64-bit multiplies per element
  const int qi = qt * 32 + li, i = i0 + qi;
  const uint64_t drow = (uint64_t)(si.row0 + i) << 20;
  uint32_t keepm[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j) keepm[j] = 0u;
  auto keep_step = [&](int r) {
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      const int key = (kg + 4 * j) * 32 + 8 * (r >> 2) + 4 * lh + (r & 3);
      keepm[j] |= (dropout_keep(drop.seed, 0, drow | (uint64_t)key, drop.thr) ? 1u : 0u) << r;
    }
  };
  const int KT1 = D / AT_BK;                  // >= 4
  dma1(0, 0);
  if (AT_NST > 2) dma1(1, 1);
  int stage = 0;
  for (int kt = 0; kt < KT1; ++kt) {
    // tile kt has landed (this wave's share: all but the NI1 younger requests; the barrier adds everyone else's), and every wave is
    // past its reads of tile kt - 1, whose stage the requests for tile kt + 2 overwrite
    if (AT_NST > 2 && kt + 1 < KT1) wait_vm<NI1>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    const int st2 = stage == 0 ? AT_NST - 1 : stage - 1;
    const char* const cur = lds + stage * AT_STAGE1;
    auto compute1 = [&]() {
      if (drop.thr && kt < 16) keep_step(kt);
      Frag1 f[2];
      read1(cur, 0, f[0]);
#pragma unroll
      for (int ks = 0; ks < AT_BK / 16; ++ks) {
        if (ks + 1 < AT_BK / 16) read1(cur, ks + 1, f[(ks + 1) & 1]);
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[ks & 1].ak[j], f[ks & 1].bq, acc[j], 0, 0, 0);
      }
    };
    // The two waves of a SIMD (w and w + 4) take the iteration in opposite orders: a DMA instruction holds its wave's issue for ~100-200
    // cycles, and when both waves requested first and multiplied second the matrix pipe idled through every request phase (k-tile 2,000
    // cycles = requests + MFMAs in series, the same with one or two tiles in flight).
    if (wave < 4) {
      if (kt + AT_NST - 1 < KT1) dma1(kt + AT_NST - 1, st2);
      compute1();
    } else {
      compute1();
      if (kt + AT_NST - 1 < KT1) dma1(kt + AT_NST - 1, st2);
    }
    stage = stage == AT_NST - 1 ? 0 : stage + 1;
  }
  if (drop.thr)
    for (int r = KT1; r < 16; ++r) keep_step(r);     // D < 1024: fewer k-tiles than accumulator registers
  __syncthreads();                            // every wave is past its last GEMM-1 fragment read: the stages may be overwritten

  // ---------------------------------------------------------------------------------------------- GEMM-2 operands: two tiles in flight under the row op
  const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned short*>(a.C16 + (int64_t)si.row0 * a.ldc), (short)0, ((T - 1) * a.ldc + D) * 2, 0x00020000);
  // DMA instruction p of this wave fills key rows (8 p + wave) * 2 + [0, 2) of the [64 keys][256 cols] tile, 32 chunks of 16 B each:
  // lane -> (row lane / 32, chunk lane % 32); the 64-byte piece q = chunk / 4 of a row is stored at piece position q ^ (row & 3)
  const int c_row = wave * 2 + (lane >> 5);
  const int c_chunk = ((((lane & 31) >> 2) ^ (c_row & 3)) << 2) | (lane & 3);
  const int NC = D / AT_NCH, n_it = NC * NJ;
  constexpr int NI2 = 4;
  auto dma2 = [&](int nc, int kt, int stage) {
    char* const dst = lds + AT_ST2 + stage * AT_STAGE2 + wave * 1024;
#pragma unroll
    for (int p = 0; p < 4; ++p)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rC, (lds_vptr)(dst + p * 8192), 16,
                                               (min(kt * 64 + c_row + 16 * p, T - 1) * a.ldc + nc * AT_NCH + c_chunk * 8) * 2, 0, 0, 0);
  };
  dma2(0, 0, 0);
  if (n_it > 1) dma2(NJ > 1 ? 0 : 1, NJ > 1 ? 1 : 0, 1);

  // ---------------------------------------------------------------------------------------------- row op
  // acc[j][r]: key = (kg + 4 j) * 32 + 8 (r >> 2) + 4 lh + (r & 3), query = qt * 32 + li
  const bool row_ok = i < T;
  float* const red = reinterpret_cast<float*>(lds + AT_RED);     // [4 key groups][64 queries], twice
  float* const erow = a.E + si.eoff + (int64_t)i * si.ldE;
  unsigned short* const prow = a.P16 + si.e16off + (int64_t)i * T64;
  char* const lrow = lds + AT_SP + qi * AT_PP;
  if constexpr (!BWD) {
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = (kg + 4 * j) * 32 + 8 * (r >> 2) + 4 * lh + (r & 3);
        const float e = key < T ? masked_logit(acc[j][r], a.scale, i, key, a.ignore_self, a.aperture) : -INFINITY;
        acc[j][r] = e;
        m = fmaxf(m, e);
      }
    m = fmaxf(m, __shfl_xor(m, 32));
    if (lh == 0) red[kg * 64 + qi] = m;
    lds_barrier();
    m = fmaxf(fmaxf(red[qi], red[64 + qi]), fmaxf(red[128 + qi], red[192 + qi]));
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = (kg + 4 * j) * 32 + 8 * (r >> 2) + 4 * lh + (r & 3);
        const float p = key < T ? __expf(acc[j][r] - m) : 0.f;
        acc[j][r] = p;
        sum += p;
      }
    sum += __shfl_xor(sum, 32);
    if (lh == 0) red[256 + kg * 64 + qi] = sum;
    lds_barrier();
    sum = (red[256 + qi] + red[320 + qi]) + (red[384 + qi] + red[448 + qi]);
    const float rsum = 1.0f / sum;
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int k0 = (kg + 4 * j) * 32 + 8 * g + 4 * lh;
        if (k0 >= T64) continue;              // dummy tile slot
        float al[4], ad[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          al[c] = k0 + c < T ? acc[j][4 * g + c] * rsum : 0.f;
          ad[c] = drop.thr ? (((keepm[j] >> (4 * g + c)) & 1u) ? al[c] * drop.scale : 0.f) : al[c];
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
    float al[NTW][16];
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int k0 = (kg + 4 * j) * 32 + 8 * g + 4 * lh;
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row_ok && k0 < si.ldE) t = *reinterpret_cast<const float4*>(erow + k0);
        al[j][4 * g] = t.x; al[j][4 * g + 1] = t.y; al[j][4 * g + 2] = t.z; al[j][4 * g + 3] = t.w;
      }
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float d = acc[j][r];
        if (drop.thr) d = ((keepm[j] >> r) & 1u) ? d * drop.scale : 0.f;
        acc[j][r] = d;
        dot += d * al[j][r];
      }
    dot += __shfl_xor(dot, 32);
    if (lh == 0) red[kg * 64 + qi] = dot;
    lds_barrier();
    dot = (red[qi] + red[64 + qi]) + (red[128 + qi] + red[192 + qi]);
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int k0 = (kg + 4 * j) * 32 + 8 * g + 4 * lh;
        if (k0 >= T64) continue;              // dummy tile slot
        float v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = al[j][4 * g + c] * (acc[j][4 * g + c] - dot) * a.scale;
        const u32x2 pk = row_ok ? pack_bf16x4(v[0], v[1], v[2], v[3]) : u32x2{0u, 0u};
        if (row_ok) *reinterpret_cast<u32x2*>(prow + k0) = pk;
        *reinterpret_cast<u32x2*>(lrow + k0 * 2) = pk;
      }
  }

  // ---------------------------------------------------------------------------------------------- GEMM-2
  // wave w: output columns nc * 256 + 32 w + [0, 32) (one M tile), both query tiles
  const int rr = (lane >> 2) & 3;             // (key row & 3) of this lane's transposing reads
  const int fct = (8 * lh + ((lane & 15) >> 2)) * AT_CROW + 32 * ((lane >> 4) & 1) + 8 * (lane & 3) + ((wave ^ rr) << 6);
  const char* const fp = lds + AT_SP + li * AT_PP + 16 * lh;
  struct Frag2 { bf16x8 fa, fb[2]; };
  auto read2 = [&](const char* buf, int kt, int ks, Frag2& f) {
    f.fa = tr_frag(buf + fct + 16 * ks * AT_CROW, AT_CROW);
#pragma unroll
    for (int u = 0; u < 2; ++u) f.fb[u] = *reinterpret_cast<const bf16x8*>(fp + u * 32 * AT_PP + (kt * 64 + ks * 16) * 2);
  };
  int stage2 = 0, it = 0;
  int nc2 = NJ > 2 ? 0 : (NJ == 2 ? 1 : 2), kt2 = NJ > 2 ? 2 : 0;     // (nc, kt) of tile it + 2
  for (int nc = 0; nc < NC; ++nc) {
    f32x16 o[2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[u][r] = 0.f;
#pragma unroll
    for (int kt = 0; kt < NJ; ++kt) {
      // (the pass epilogue's global stores share the counter and complete out of order with loads: the first tile after them drains it)
      if (it + 1 < n_it && !(kt == 0 && nc > 0)) wait_vm<NI2>(); else wait_vm<0>();
      __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): (first pass) this wave's P writes are in LDS before the barrier publishes them
      __builtin_amdgcn_s_barrier();
      const int st2 = stage2 == 0 ? 2 : stage2 - 1;
      const char* const cur = lds + AT_ST2 + stage2 * AT_STAGE2;
      auto compute2 = [&]() {
        Frag2 f[2];
        read2(cur, kt, 0, f[0]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          if (ks + 1 < 4) read2(cur, kt, ks + 1, f[(ks + 1) & 1]);
#pragma unroll
          for (int u = 0; u < 2; ++u) o[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[ks & 1].fa, f[ks & 1].fb[u], o[u], 0, 0, 0);
        }
      };
      if (wave < 4) {
        if (it + 2 < n_it) dma2(nc2, kt2, st2);
        compute2();
      } else {
        compute2();
        if (it + 2 < n_it) dma2(nc2, kt2, st2);
      }
      if (++kt2 == NJ) { kt2 = 0; ++nc2; }
      stage2 = stage2 == 2 ? 0 : stage2 + 1;
      ++it;
    }
    // o[u][r]: column = nc * 256 + wave * 32 + 8 (r >> 2) + 4 lh + (r & 3), query = u * 32 + li
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int q = u * 32 + li;
      unsigned short* const orow = a.O16 + (int64_t)(si.row0 + i0 + q) * a.ldo + nc * AT_NCH + wave * 32 + 4 * lh;
      if (q < rows) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<u32x2*>(orow + 8 * g) = pack_bf16x4(o[u][4 * g], o[u][4 * g + 1], o[u][4 * g + 2], o[u][4 * g + 3]);
      }
    }
  }
}

template <bool BWD>
__global__ __launch_bounds__(512) void attn_strip_kernel(AttnStripArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  int s, strip;
  SeqInfo si;
  if (!locate_block(a.seq, a.n_seq, [](int t) { return (t + AT_ROWS - 1) / AT_ROWS; }, si, s, strip)) return;      // pw_common.h: every XCD the same number of strips
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
  // (the opt-in is per DEVICE: one bit per device, set atomically -- a process that drives a second GPU must not skip it there)
  static std::atomic<uint64_t> attr_done[2];
  int dev = 0;
  SUMK_HIP(hipGetDevice(&dev));
  const uint64_t bit = 1ull << (dev & 63);
  if (!(attr_done[backward ? 1 : 0].load(std::memory_order_acquire) & bit)) {
    SUMK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, AT_LDS));
    attr_done[backward ? 1 : 0].fetch_or(bit, std::memory_order_release);
  }
  const unsigned grid = (unsigned)(8 * ((a.n_seq + 7) / 8) * a.strips);
  if (backward) hipLaunchKernelGGL(attn_strip_kernel<true>, dim3(grid), dim3(512), AT_LDS, stream, a);
  else hipLaunchKernelGGL(attn_strip_kernel<false>, dim3(grid), dim3(512), AT_LDS, stream, a);
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}

}  // namespace sumk
