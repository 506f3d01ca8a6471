// Per-video attention of the split-bf16 scoring path on operand planes (SUMK_PRECISION_BF16X6 / BF16X3 inference, T <= 320 keys):
// reference summarizer/models/vasnet.py:118-131 (logits = Q K^T * scale, masks, softmax over keys, context = alpha V).
//
// Two launches -- A one workgroup per (video, 64 query rows), B one per (video, 128 query rows, half of the columns) since round 6 -- both reading the K / Q / V PLANES the projection GEMM's epilogue wrote
// (gemm_pw.hip, PW_PLANES: "KB planes" of the (R x 3D) matrix [Q | K | V]) -- nothing is split on the vector ALU inside a k-loop:
//   A  attn_pw_logits_kernel : S^T[key][query] = K . Q^T over D (NT, both operands K-contiguous planes), masked softmax in registers (the
//      transposed product leaves 4 consecutive keys of ONE query per lane register quad), alpha split into planes in registers and
//      stored as 512-byte runs (alpha planes: rows = packed query rows, k = key index inside the video); optionally alpha in fp32 (E).
//   B  attn_pw_context_kernel: CTX^T[col][query] = V^T . alpha^T over the keys; alpha planes are K-contiguous, V is not ([key][col]
//      chunks of 8 columns): its fragments come out of LDS with ds_read_b64_tr_b16, from an image whose DMA pieces interleave the four
//      sub-arrays a transposing read touches (conflict-free without padding).  CTX leaves as planes: the operand of the output projection.
// The in-loop split kernels these replace (64 x 64 tiles, two barriers per k-tile, both operands split per block) ran the two products at
// 0.2 of the split-bf16 ceiling (69 + 13 + 65 us at bf16x6 on S-TVSum) and CTX needed a separate fp32 -> planes pass (30 us).
#include "pw_common.h"
#include <math.h>
#include <atomic>
#include <vector>
#include <algorithm>
#include <cstdlib>

namespace sumk {

namespace {

constexpr int AP_ROWS = 64, AP_TMAX = 320;

#ifndef SUMK_CTX_ABL
#define SUMK_CTX_ABL 0               // kernel B timing probes (WRONG results; variant libraries only, scripts/probes/ctx_wide_ab.sh): 1 no MFMAs, 2 no DMA behind the prologue,
#endif                               // 4 no fragment reads in the loop, 8 no barriers, 16 no pass epilogue
#ifndef SUMK_ATTN_NLW
#define SUMK_ATTN_NLW 8              // waves of a block that issue the LDS-DMA pieces of a stage (probe: scripts/attn_nlw_probe.sh)
#endif

struct AttnPwArgs {
  const char* QKV; uint32_t rp16;          // KB planes of [Q | K | V] (rows = packed frames, k = 3 D columns)
  int32_t D;
  float* E;                                // null, or alpha in fp32: per-video (T x ldE) blocks (SeqInfo::eoff)
  char* AP; uint32_t ap_rp16;              // alpha planes: rows = packed frames, k = key index inside the video (< T rounded up to 16)
  char* CP; uint32_t cp_rp16;              // context planes (kernel B): rows = packed frames, k = D columns
  const SeqInfo* seq; int32_t n_seq, strips;
  float scale; int32_t ignore_self, aperture;
  int32_t heads, dh;                       // multi-head form (Transformer scorer): head h contracts columns [h dh, (h + 1) dh) of Q / K and owns alpha planes
  int64_t ap_head_bytes;                   // AP + h * ap_head_bytes; its context lands in columns [h dh, (h + 1) dh).  heads = 1, dh = D: VASNet
  const float* R; int32_t ldr;             // kernel B, folded VASNet path (V = x Wvo^T): context + R is what leaves as planes, with its LayerNorm moments
  float* moments;                          // float2[rows][D / 32] {sum v, sum v^2} per 32-column slot
  int32_t csplit;                          // kernel B, QT = 4 form: a block owns 128 query rows and 1 / csplit of the 256-column passes
  unsigned long long* stamps;              // diagnostic build: per block {T, prologue, k-loop, row op, total} shader cycles + realtime
};

// ------------------------------------------------------------------------------------------------ A: logits + softmax
// EIGHT waves (two per SIMD): wave = (query tile qt of 32, key group kg of 4); key tiles kg + 4 j, j < NTW = ceil(2 NJ / 4), NJ = ceil(T / 64)
// (tile slots past 2 NJ multiply whatever rows follow in the stage -- the query rows, finite -- and are masked as key >= T).
// One k16 step of D per stage: NSUB sub-arrays (plane, k half) x (64 NJ key rows + 64 query rows) x 16 B, filled by LDS-DMA pieces of 64
// rows; NS stages in a ring, ONE barrier per step behind all but the last key tile's MFMAs (gemm_pw.hip's loop).
// VAR (when a wave issues the DMA pieces that refill the slot a barrier freed): 0 = after the step's last MFMAs, 1 = right behind the
// barrier, 2 = spread: a share behind every key tile's MFMAs, from the step's last tile through the next step's tiles in front of the barrier.
// NT: the K / Q loads carry the non-temporal policy.  Chosen by the host when the [Q | K | V] planes are larger than ~3/4 of the Infinity Cache: the logits
// launch then streams its 2/3 of them without displacing V, which launch B reads next (three planes, S-TVSum: B 73 -> 63 us, A unchanged; two planes, where
// all the planes fit the cache anyway: A 43.5 -> 52 us, B unchanged -- hence not unconditional; nt on B's own loads: slower in both.  profiles/r06_attn_ctx_ablation.txt)
template <int NP, int NJ, int VAR, bool MH = false, bool NT = false>
__device__ __forceinline__ void attn_logits_body(const AttnPwArgs& a, const SeqInfo& si, const int strip, const int head, char* const lds) {
  constexpr int NSUB = 2 * NP, T64 = NJ * 64, ROWS = T64 + 64, STAGE = NSUB * ROWS * 16;
  // (deeper rings, 4 / 6 stages, were measured: no change -- the stream is not latency-bound).  MH (multi-head form: dh / 16 = 8 steps per block, the
  // prologue and the row op are most of a block's time): one stage less, so that TWO blocks fit a CU and overlap each other's phases
  constexpr int NS = (NP == 3 ? 3 : 4) - (MH ? 1 : 0);
  constexpr int NTW = (2 * NJ + 3) / 4;
  constexpr int NLW = SUMK_ATTN_NLW;                                   // waves that issue the DMA pieces (the others only wait at the barrier)
  constexpr int NPIECE = NSUB * (NJ + 1), MAXP = (NPIECE + NLW - 1) / NLW;
  static_assert(NS * STAGE + 4096 <= 160 * 1024, "LDS map");
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T = si.T, i0 = strip * AP_ROWS, D = a.D;
  const int qt = wave & 1, kg = wave >> 1;

  // this wave's DMA pieces of a stage: piece idx = wave + 8 i -> (sub-array, 64-row block: NJ key blocks, then the query block)
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.QKV), (short)0, 0x80000000u, 0x00020000);
  int pg[MAXP], pl[MAXP];
#pragma unroll
  for (int i = 0; i < MAXP; ++i) {
    int idx = wave + NLW * i;
    idx = idx < NPIECE ? idx : NPIECE - 1;
    const int sub = idx / (NJ + 1), blk = idx - sub * (NJ + 1);
    const bool key = blk < NJ;
    // K columns are k-blocks D / 16 ... 2 D / 16 - 1 of the [Q | K | V] planes, Q columns the first D / 16
    pg[i] = (((key ? (D >> 4) : 0) + head * (a.dh >> 4)) * NSUB + sub) * (int)a.rp16 + (si.row0 + (key ? blk * 64 : i0)) * 16;
    pl[i] = (sub * ROWS + (key ? blk * 64 : T64)) * 16;
  }
  const bool full = (NPIECE % NLW == 0) || wave < NPIECE % NLW;          // this wave issues MAXP pieces (else MAXP - 1)
  const bool loader = wave < NLW;
  const int vlane = lane * 16, k_step = NSUB * (int)a.rp16;
  auto dma_share = [&](int kb, int slot, int share, int n_shares) {      // pieces i with i % n_shares == share
    if (!loader) return;
#ifdef SUMK_DIAG
    if (VAR == 4 && kb >= NS) return;                                     // timing probe: MFMAs + reads + barriers only (stale stages)
#endif
    char* const st = lds + slot * STAGE;
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
      if (i % n_shares != share) continue;
      if (i == MAXP - 1 && !full) break;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_vptr)(st + pl[i]), 16, vlane, kb * k_step + pg[i], 0, NT ? 2 : 0);
    }
  };
  auto dma = [&](int kb, int slot) { dma_share(kb, slot, 0, 1); };
  const int fk = (lh * ROWS + kg * 32 + li) * 16, fq = (lh * ROWS + T64 + qt * 32 + li) * 16;
  struct Frags { bf16x8 k[NP][NTW], q[NP]; };
  auto read_frags = [&](int slot, Frags& f) {
    const char* const st = lds + slot * STAGE;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      f.q[p] = *reinterpret_cast<const bf16x8*>(st + fq + p * 2 * ROWS * 16);
#pragma unroll
      for (int j = 0; j < NTW; ++j) f.k[p][j] = *reinterpret_cast<const bf16x8*>(st + fk + p * 2 * ROWS * 16 + j * 128 * 16);
    }
  };
  f32x16 acc[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  // (key tiles interleaved per plane product: consecutive MFMAs write different accumulators -- see kernel B's mfma_part)
  auto mfma_tiles = [&](const Frags& f, int j_lo, int j_hi) {
#pragma unroll
    for (int sum = NP - 1; sum >= 0; --sum)              // (Q plane i, K plane j2), smallest products first: the order of the NT GEMM Q . K^T
#pragma unroll
      for (int i = NP - 1; i >= 0; --i) {
        const int j2 = sum - i;
        if (j2 < 0 || j2 >= NP) continue;
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
          if (j < j_lo || j >= j_hi) continue;
#ifdef SUMK_DIAG
          if constexpr (VAR == 3) { asm volatile("" :: "v"(f.k[j2][j]), "v"(f.q[i])); continue; }      // timing probe: DMA + reads + barriers only
#endif
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.k[j2][j], f.q[i], acc[j], 0, 0, 0);
        }
      }
  };
  // the step's MFMAs as two groups of PLANE PRODUCTS (every key tile in each): products [p_lo, p_hi) of the NP (NP + 1) / 2, in the order above
  constexpr int NPROD = NP * (NP + 1) / 2;
  auto mfma_prods = [&](const Frags& f, int p_lo, int p_hi) {
    int pi = 0;
#pragma unroll
    for (int sum = NP - 1; sum >= 0; --sum)
#pragma unroll
      for (int i = NP - 1; i >= 0; --i) {
        const int j2 = sum - i;
        if (j2 < 0 || j2 >= NP) continue;
        const bool on = pi >= p_lo && pi < p_hi;
        ++pi;
        if (!on) continue;
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
#ifdef SUMK_DIAG
          if constexpr (VAR == 3) { asm volatile("" :: "v"(f.k[j2][j]), "v"(f.q[i])); continue; }
#endif
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.k[j2][j], f.q[i], acc[j], 0, 0, 0);
        }
      }
  };

  const int nk = a.dh >> 4;
#ifdef SUMK_DIAG
  const unsigned long long st0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll
  for (int s = 0; s < NS; ++s) dma(s, s);
  wait_vm<0>();
  __builtin_amdgcn_s_barrier();
#ifdef SUMK_DIAG
  const unsigned long long st1 = __builtin_amdgcn_s_memtime();
#endif
  Frags F0, F1;
  read_frags(0, F0);
  int slot = 0;
  // MH: ONE fragment set (two blocks per CU need <= 128 VGPRs): all of a step's MFMAs run in front of its barrier, the next fragments are read behind it into
  // the same registers -- the other block's waves cover the read latency
  constexpr int P1 = MH ? NTW : (NTW > 1 ? NTW - 1 : 0);
  constexpr int PA = MH ? NPROD : (2 * NPROD + 2) / 3;        // products in front of the barrier (two thirds), the rest behind it
  int pend_kb = -1, pend_slot = 0;               // VAR 2: the refill in progress (stage, slot)
  auto kstep = [&](const Frags& cur, Frags& nxt, int s) {
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (VAR == 2 && NTW > 1) {
#pragma unroll
      for (int j = 0; j < P1; ++j) {
        mfma_tiles(cur, j, j + 1);
        __builtin_amdgcn_sched_barrier(0);
        if (pend_kb >= 0) dma_share(pend_kb, pend_slot, j, NTW);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      mfma_prods(cur, 0, PA);
    }
    __builtin_amdgcn_sched_barrier(0);
    const bool more = s + 1 < nk, fill = s + NS < nk;
    const int nslot = slot + 1 == NS ? 0 : slot + 1;
    if (more) {
      if (s + NS - 1 < nk) { if (full) wait_vm<(NS - 2) * MAXP>(); else wait_vm<(NS - 2) * (MAXP - 1)>(); }
      else wait_vm<0>();
      __builtin_amdgcn_s_waitcnt(0xC07F);                   // lgkmcnt(0): this wave holds every fragment of the stage whose slot is refilled behind the barrier
      __builtin_amdgcn_s_barrier();
      read_frags(nslot, nxt);
      __builtin_amdgcn_sched_barrier(0);
      if (VAR == 1 && fill) dma(s + NS, slot);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (VAR == 2 && NTW > 1) mfma_tiles(cur, P1, NTW); else mfma_prods(cur, PA, NPROD);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (VAR == 2 && NTW > 1) {
      pend_kb = (more && fill) ? s + NS : -1; pend_slot = slot;
      if (pend_kb >= 0) dma_share(pend_kb, pend_slot, NTW - 1, NTW);
    } else if (VAR != 1) {
      if (more && fill) dma(s + NS, slot);
    }
    slot = nslot;
  };
#ifdef SUMK_DIAG
  if constexpr (VAR == 5) {      // timing probe: the DMA stream alone -- no fragment reads, no MFMAs, no barriers; NS - 1 stages kept in flight
    for (int s = 0; s < nk; ++s) {
      if (full) wait_vm<(NS - 1) * MAXP>(); else wait_vm<(NS - 1) * (MAXP - 1)>();
      if (s + NS < nk) dma(s + NS, s % NS);
    }
    wait_vm<0>();
  } else
#endif
  for (int s = 0; s < nk; s += 2) {
    if constexpr (MH) {
      kstep(F0, F0, s);
      kstep(F0, F0, s + 1);
    } else {
      kstep(F0, F1, s);
      kstep(F1, F0, s + 1);
    }
  }

  // ---- row op: acc[j][r]: key = (kg + 4 j) * 32 + 8 (r >> 2) + 4 lh + (r & 3), query = qt * 32 + li
  lds_barrier();                                  // every wave is past its last fragment read: the stages are free
#ifdef SUMK_DIAG
  const unsigned long long st2 = __builtin_amdgcn_s_memtime();
#endif
  float* const red = reinterpret_cast<float*>(lds);        // [4 key groups][64 queries], twice
  const int qi = qt * 32 + li, i = i0 + qi;
  const bool row_ok = i < T;
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
  const int T32 = (T + 31) & ~31;                 // alpha planes are written (zeros past T) up to the k32 step the context kernel ends on
  float* const erow = a.E ? a.E + si.eoff + (int64_t)i * si.ldE : nullptr;
  char* const arow = a.AP + head * a.ap_head_bytes + (int64_t)(si.row0 + i) * 16 + 8 * lh;
#pragma unroll
  for (int j = 0; j < NTW; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int k0 = (kg + 4 * j) * 32 + 8 * g + 4 * lh;
      if (k0 >= T32 || !row_ok) continue;
      float al[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) al[c] = k0 + c < T ? acc[j][4 * g + c] * rsum : 0.f;
      if (erow && k0 < si.ldE) *reinterpret_cast<float4*>(erow + k0) = make_float4(al[0], al[1], al[2], al[3]);
      u32x2 pl2[NP];
      split4<NP>(f32x4{al[0], al[1], al[2], al[3]}, pl2);
      char* const op = arow + (int64_t)(((k0 >> 4) * NP) * 2 + ((k0 >> 3) & 1)) * a.ap_rp16;
#pragma unroll
      for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x2*>(op + (int64_t)p * 2 * a.ap_rp16) = pl2[p];
    }
#ifdef SUMK_DIAG
  if (a.stamps && tid == 0) {
    wait_vm<0>();
    const unsigned long long st3 = __builtin_amdgcn_s_memtime();
    unsigned long long* o = a.stamps + (size_t)blockIdx.x * 8;
    o[0] = T; o[1] = st1 - st0; o[2] = st2 - st1; o[3] = st3 - st2; o[4] = st3 - st0; o[5] = __builtin_amdgcn_s_memrealtime() - rt0; o[6] = rt0;
    o[7] = __builtin_amdgcn_s_getreg((31 << 11) | 20) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 32);
  }
#endif
}

template <int NP, bool NT = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void attn_pw_logits_mh_kernel(AttnPwArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  int s, sub;
  const int nh = a.heads;
  SeqInfo si;
  if (!locate_block(a.seq, a.n_seq, [nh](int t) { return ((t + AP_ROWS - 1) / AP_ROWS) * nh; }, si, s, sub)) return;
  const int strip = sub / nh, head = sub - strip * nh;      // (the heads of one strip are neighbours in the list: they share the strip's query rows and the video's keys in L2)
  switch ((si.T + 63) >> 6) {
    case 1: attn_logits_body<NP, 1, 0, true, NT>(a, si, strip, head, lds); break;
    case 2: attn_logits_body<NP, 2, 0, true, NT>(a, si, strip, head, lds); break;
    case 3: attn_logits_body<NP, 3, 0, true, NT>(a, si, strip, head, lds); break;
    case 4: attn_logits_body<NP, 4, 0, true, NT>(a, si, strip, head, lds); break;
    default: attn_logits_body<NP, 5, 0, true, NT>(a, si, strip, head, lds); break;
  }
}

template <int NP, int VAR, bool NT = false>
__global__ __launch_bounds__(512) void attn_pw_logits_kernel(AttnPwArgs a) {       // single head (the multi-head form: attn_pw_logits_mh_kernel)
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int head = 0;
  int s, strip;
  SeqInfo si;
  if (!locate_block(a.seq, a.n_seq, [](int t) { return (t + AP_ROWS - 1) / AP_ROWS; }, si, s, strip)) return;      // every XCD the same number of strips
  switch ((si.T + 63) >> 6) {
    case 1: attn_logits_body<NP, 1, VAR, false, NT>(a, si, strip, head, lds); break;
    case 2: attn_logits_body<NP, 2, VAR, false, NT>(a, si, strip, head, lds); break;
    case 3: attn_logits_body<NP, 3, VAR, false, NT>(a, si, strip, head, lds); break;
    case 4: attn_logits_body<NP, 4, VAR, false, NT>(a, si, strip, head, lds); break;
    default: attn_logits_body<NP, 5, VAR, false, NT>(a, si, strip, head, lds); break;
  }
}

// ------------------------------------------------------------------------------------------------ B: context = alpha . V
// EIGHT waves; per pass of 256 output columns wave w owns columns 32 w + [0, 32) for every 32-query tile of the block (QT = 2 or 4 tiles).  A stage holds KH k16 blocks of keys
// (three planes: one block, a ring of four 30-KB stages; two planes: two blocks, a ring of three 40-KB stages -- with 60-KB stages only two
// fit and every step waited for its own refill: 107 us instead of ~70 at bf16x6):
//   alpha: KH k16 blocks x NSUB sub-arrays x 64 query rows x 16 B (plain 1-KiB pieces);
//   V    : per (32-column pair of k-blocks fbp, plane, k16 block) ONE 1-KiB piece = 16 keys x the 4 sub-arrays {fb, fb + 1} x {h 0, 1} a
//          transposing read touches, ordered [key / 4][sub-array c][key % 4][16 B] by the per-lane SOURCE offset of the DMA -- the 32 lanes
//          of one ds_read_b64_tr_b16 half then cover 256 contiguous bytes: no bank conflict, no padding.
// MFMA A operand = V^T (columns on the M axis), B operand = alpha (queries on the N axis): a lane ends with ONE query row and 4 consecutive
// columns per register quad -- the 8-byte piece of a context-plane chunk.
// HP2 (multi-head form, dh = 128): a 256-column pass holds TWO heads -- waves 0-3 multiply the alpha of head 2 nc, waves 4-7 of head 2 nc + 1; a stage
// carries both alpha sets (V pieces unchanged: the pass's 256 V columns are those two heads').
// RM (folded VASNet path): the accumulators start from the residual R[query][column] instead of zero, and a pass also leaves {sum v, sum v^2} of every
// (query, 32-column slot) -- what the output projection's PW_RES_MOM_PLANES epilogue does on the unfolded path.
// QT = 4 (round 6): a block owns 128 query rows (four 32-query tiles per wave) and HALF of the 256-column passes (a.csplit = 2) -- the V rows a video's
// blocks stream through their CUs halve (T / 128 x D instead of T / 64 x D per video) while the block count stays at about one per CU; the stream is what
// bounds these launches (~35-45 GB/s per CU, MI355X_MICROARCH.md "Indexed rows": the Infinity-Cache rate).  One k16 block per stage in both plane counts.
template <int NP, bool HP2, bool RM = false, int QT = 2>
__device__ __forceinline__ void attn_context_body(const AttnPwArgs& a, char* const lds) {
  constexpr int KH = (NP == 3 || QT == 4) ? 1 : 2, NS = QT == 4 ? (NP == 3 ? (HP2 ? 3 : 4) : (HP2 ? 4 : 6)) : (NP == 3 ? 4 : 3);           // (rings of 5 / 4 stages: measured, no change)
  constexpr int NSETS = HP2 ? 2 : 1;
  constexpr int QR = 32 * QT, QH = QT / 2;                             // query rows per block; 64-row alpha pieces per sub-array
  constexpr int NSUB = 2 * NP, A_SET = KH * NSUB * QR * 16, A_BYTES = NSETS * A_SET, V_BYTES = KH * 8 * NP * 1024, STAGE = A_BYTES + V_BYTES;
  constexpr int NLW = SUMK_ATTN_NLW;
  constexpr int NA = NSETS * KH * NSUB * QH, NV = KH * 8 * NP, NPIECE = NA + NV, MAXP = (NPIECE + NLW - 1) / NLW;
  static_assert(NS * STAGE <= 160 * 1024, "LDS map");
  int sv, bsub;
  const int cs = a.csplit;
  SeqInfo si;
  if (!locate_block(a.seq, a.n_seq, [cs](int t) { return ((t + QR - 1) / QR) * cs; }, si, sv, bsub)) return;       // pw_common.h: every XCD the same number of blocks
  const int strip = bsub / cs, chalf = bsub - strip * cs;
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T = si.T, i0 = strip * QR, D = a.D;
  const int NC = (D >> 8) / a.csplit, nc0 = chalf * NC;                              // this block's 256-column passes: nc0 ... nc0 + NC - 1
  const int nks = (T + 16 * KH - 1) / (16 * KH), n_it = NC * nks;       // (alpha planes hold zeros from T up to T rounded up to 32)

  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(a.AP, (short)0, 0x80000000u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.QKV), (short)0, 0x80000000u, 0x00020000);
  // pieces of a stage: idx < NA: alpha (k16 block kbh = idx / NSUB of the stage, sub-array idx % NSUB); else V (fbp, plane, kbh)
  bool pa[MAXP]; int pg[MAXP], pl[MAXP], pset[MAXP];
#pragma unroll
  for (int i = 0; i < MAXP; ++i) {
    int idx = wave + NLW * i;
    idx = idx < NPIECE ? idx : NPIECE - 1;
    pa[i] = idx < NA;
    pset[i] = 0;
    if (pa[i]) {
      pset[i] = idx / (KH * NSUB * QH);
      const int rem = idx - pset[i] * KH * NSUB * QH, sa = rem / QH, qh = rem - sa * QH;               // sub-array (k16 block of the stage, plane, k half), 64-row piece
      pg[i] = sa * (int)a.ap_rp16 + (si.row0 + i0 + 64 * qh) * 16;                                      // + ks * KH NSUB ap_rp16 (+ the head's alpha planes)
      pl[i] = pset[i] * A_SET + (sa * QR + 64 * qh) * 16;
    } else {
      const int v = idx - NA, fbp = v / (KH * NP), rem = v - fbp * KH * NP, p = rem / KH, kbh = rem - p * KH;
      pg[i] = (((2 * D) >> 4) + 2 * fbp) * NSUB * (int)a.rp16 + p * 2 * (int)a.rp16 + (si.row0 + 16 * kbh) * 16;      // + nc * 16 NSUB rp16 + ks * 16 KH rows
      pl[i] = A_BYTES + ((kbh * 8 + fbp) * NP + p) * 1024;
    }
  }
  const bool full = (NPIECE % NLW == 0) || wave < NPIECE % NLW;
  const bool loader = wave < NLW;
  const int vlane = lane * 16;
  const int vperm = ((lane >> 3) & 1) * NSUB * (int)a.rp16 + ((lane >> 2) & 1) * (int)a.rp16 + (4 * (lane >> 4) + (lane & 3)) * 16;
  auto dma = [&](int it, int slot) {
    if (!loader) return;
    if ((SUMK_CTX_ABL & 2) && it >= NS) return;
    const int ncl = it / nks, ks = it - ncl * nks, nc = nc0 + ncl;
    char* const st = lds + slot * STAGE;
    const int ga = ks * KH * NSUB * (int)a.ap_rp16, gv = nc * 16 * NSUB * (int)a.rp16 + ks * KH * 256;
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
      if (i == MAXP - 1 && !full) break;
      if (pa[i]) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_vptr)(st + pl[i]), 16, vlane, ga + pg[i] + (HP2 ? (2 * nc + pset[i]) * (int)a.ap_head_bytes : 0), 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rV, (lds_vptr)(st + pl[i]), 16, vperm, gv + pg[i], 0, 0);
    }
  };
  // fragments of the k16 block kbh of a stage: V plane p (tr reads), alpha plane p of query tile u
  const int fv = A_BYTES + wave * NP * 1024 + (2 * lh) * 256 + ((((lane >> 4) & 1) << 1) | ((lane & 3) >> 1)) * 64 + ((lane & 15) >> 2) * 16 + (lane & 1) * 8;
  const int fa = (lh * QR + li) * 16 + (HP2 ? (wave >> 2) * A_SET : 0);
  struct Frags { bf16x8 v[KH][NP], al[KH][NP][QT]; };
  auto read_frags = [&](int slot, Frags& f) {
    const char* const st = lds + slot * STAGE;
#pragma unroll
    for (int kbh = 0; kbh < KH; ++kbh)
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        f.v[kbh][p] = tr_frag(st + fv + (kbh * 8 * NP + p) * 1024, 64);
#pragma unroll
        for (int u = 0; u < QT; ++u) f.al[kbh][p][u] = *reinterpret_cast<const bf16x8*>(st + fa + ((kbh * NSUB + p * 2) * QR + u * 32) * 16);
      }
  };
  f32x16 o[QT];
  // the step's MFMAs in two parts around its barrier: part 0 = the first k16 block (KH = 2) or the first half of the plane products (KH = 1), part 1 = the rest
  // Issue order: inside a part the query tiles are interleaved per plane product -- consecutive MFMAs write different accumulators (measured against tile by
  // tile, each tile's products in a row: no difference; per accumulator the order of the products is the same either way: same bits).  What the loop's MFMAs cost
  // by themselves: builds with everything else removed (SUMK_CTX_ABL) run the launch's 1 920 MFMAs per SIMD in ~46 us = 24 ns each, the rate the plane GEMM
  // sustains as well (1.3 PFLOP/s over the chip) -- profiles/r06_attn_ctx_ablation.txt.
  constexpr int NPROD = NP * (NP + 1) / 2, PA = (NPROD + 1) / 2;     // KH = 1: part 0 = the first PA plane products of every query tile, part 1 = the rest
  auto mfma_part = [&](const Frags& f, int part) {
#pragma unroll
    for (int kbh = 0; kbh < KH; ++kbh) {
      int pi = 0;
#pragma unroll
      for (int sum = NP - 1; sum >= 0; --sum)              // (alpha plane i, V plane j), smallest products first: the order of the NN GEMM alpha . V
#pragma unroll
        for (int i = NP - 1; i >= 0; --i) {
          const int j = sum - i;
          if (j < 0 || j >= NP) continue;
          const int ppart = pi < PA ? 0 : 1;
          ++pi;
          if ((KH == 2 ? kbh : ppart) != part) continue;
#pragma unroll
          for (int u = 0; u < QT; ++u) {
            if constexpr ((SUMK_CTX_ABL & 1) != 0) { asm volatile("" :: "v"(f.v[kbh][j]), "v"(f.al[kbh][i][u])); continue; }
            o[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.v[kbh][j], f.al[kbh][i][u], o[u], 0, 0, 0);
          }
        }
    }
  };

#pragma unroll
  for (int s = 0; s < NS; ++s) if (s < n_it) dma(s, s);
  wait_vm<0>();
  __builtin_amdgcn_s_barrier();
  Frags F0, F1;
  read_frags(0, F0);
  int slot = 0, ks = 0, nc = nc0;
  auto init_o = [&](int pass) {
#pragma unroll
    for (int u = 0; u < QT; ++u) {
      if constexpr (RM) {
        if (pass < nc0 + NC) {
          const float* const rp = a.R + (int64_t)(si.row0 + min(i0 + u * 32 + li, T - 1)) * a.ldr + pass * 256 + 32 * wave + 4 * lh;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 v = *reinterpret_cast<const float4*>(rp + 8 * g);
            o[u][4 * g] = v.x; o[u][4 * g + 1] = v.y; o[u][4 * g + 2] = v.z; o[u][4 * g + 3] = v.w;
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) o[u][r] = 0.f;
      }
    }
  };
  init_o(nc0);
  // (Tried for the QT = 4 form: the barrier of step `it` publishing stage it + 2, so that the fragments of stage it + 1 are requested at the START of the step,
  //  under all of its MFMAs, and the slot refilled behind the barrier is stage it + 1's -- same ring, same stages in flight: 97 us against 77 at three planes,
  //  52 against 45 at two.  profiles/r06_attn_ctx_ablation.txt.)
  auto step = [&](const Frags& cur, Frags& nxt, int it) {
    const bool more = it + 1 < n_it, fill = it + NS < n_it;
    const int nslot = slot + 1 == NS ? 0 : slot + 1;
    __builtin_amdgcn_sched_barrier(0);
    mfma_part(cur, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (more) {
      // (the tail, and the step right behind a pass epilogue -- its stores share the counter and retire out of order with the loads --
      //  drain fully)
      if (it + NS - 1 < n_it && !(ks == 0 && nc > nc0)) { if (full) wait_vm<(NS - 2) * MAXP>(); else wait_vm<(NS - 2) * (MAXP - 1)>(); }
      else wait_vm<0>();
      __builtin_amdgcn_s_waitcnt(0xC07F);                   // lgkmcnt(0): this wave holds every fragment of the stage whose slot is refilled behind the barrier
      if constexpr ((SUMK_CTX_ABL & 8) == 0) __builtin_amdgcn_s_barrier();
      if constexpr ((SUMK_CTX_ABL & 4) == 0) read_frags(nslot, nxt);
      if constexpr ((SUMK_CTX_ABL & 4) != 0) nxt = cur;
    }
    __builtin_amdgcn_sched_barrier(0);
    mfma_part(cur, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (more && fill) dma(it + NS, slot);
    slot = nslot;
    if ((SUMK_CTX_ABL & 16) ? it + 1 == n_it : ++ks == nks) {                        // end of a 256-column pass: o[u][4 g + c] = CTX[query u * 32 + li][nc * 256 + 32 wave + 8 g + 4 lh + c]
#pragma unroll
      for (int u = 0; u < QT; ++u) {
        const int q = i0 + u * 32 + li;
        if constexpr (RM) {
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) { const float v = o[u][r]; s1 += v; s2 += v * v; }
          s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
          if (lh == 0 && q < T) reinterpret_cast<float2*>(a.moments)[(int64_t)(si.row0 + q) * (D >> 5) + nc * 8 + wave] = make_float2(s1, s2);
        }
        if (q < T) {
          char* const orow = a.CP + (int64_t)(si.row0 + q) * 16 + 8 * lh;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            u32x2 pl2[NP];
            split4<NP>(f32x4{o[u][4 * g], o[u][4 * g + 1], o[u][4 * g + 2], o[u][4 * g + 3]}, pl2);
            const int kb = nc * 16 + wave * 2 + (g >> 1), h = g & 1;
            char* const op = orow + (int64_t)((kb * NP) * 2 + h) * a.cp_rp16;
#pragma unroll
            for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x2*>(op + (int64_t)p * 2 * a.cp_rp16) = pl2[p];
          }
        }
      }
      ks = 0; ++nc;
      init_o(nc);
    }
  };
  for (int it = 0; it < n_it; it += 2) {
    step(F0, F1, it);
    if (it + 1 < n_it) step(F1, F0, it + 1);
  }
}

template <int NP, bool HP2, bool RM = false, int QT = 2>
__global__ __launch_bounds__(512) void attn_pw_context_kernel(AttnPwArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  attn_context_body<NP, HP2, RM, QT>(a, lds);
}

std::atomic<uint64_t> g_attr[28];
constexpr int ATTN_VAR_A = 0;      // the product's schedule variant of the logits kernel (measured: profiles/r05_attn_pw_dma_variants.txt)

template <typename K>
int set_lds_once(K kernel, int inst, int bytes) {
  int dev = 0;
  SUMK_HIP(hipGetDevice(&dev));
  const uint64_t bit = 1ull << (dev & 63);
  if (!(g_attr[inst].load(std::memory_order_acquire) & bit)) {
    SUMK_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    g_attr[inst].fetch_or(bit, std::memory_order_release);
  }
  return SUMK_OK;
}

}  // namespace

// T <= 320 keys per video, D a multiple of 256, every plane offset inside 31 bits
bool attn_pw_ok(int t_max, int D, int64_t rows, int np) {
  const int64_t lim = ((int64_t)1 << 31) - 65536;
  return t_max >= 1 && t_max <= AP_TMAX && D % 256 == 0 && D >= 256 && (np == 2 || np == 3) && (int64_t)pw_planes_bytes(rows, 3 * D, np) < lim;
}

// the multi-head form: heads of 128 columns (a context pass of 256 columns = two heads), every head's alpha planes inside 31 bits of offset
bool attn_pw_heads_ok(int t_max, int D, int heads, int64_t rows, int np) {
  if (heads == 1) return attn_pw_ok(t_max, D, rows, np);
  const int64_t lim = ((int64_t)1 << 31) - 65536;
  return heads >= 2 && D == heads * 128 && attn_pw_ok(t_max, D, rows, np) && (int64_t)heads * (int64_t)align_up(pw_alpha_bytes(rows, t_max, np), 256) < lim;
}

int launch_attn_pw_logits(int np, const void* qkv_planes, int64_t rows, int D, float* E, void* alpha_planes, const SeqInfo* seq, int n_seq,
                          int t_max, float scale, int ignore_self, int aperture, hipStream_t stream, int heads) {
  SUMK_ARG(qkv_planes && alpha_planes && seq && attn_pw_heads_ok(t_max, D, heads, rows, np) && (heads == 1 || !E), "attn_pw: not eligible (T <= 320, D %% 256, 2 or 3 planes; heads of 128 columns, no fp32 alpha)");
  AttnPwArgs a;
  a.QKV = (const char*)qkv_planes; a.rp16 = (uint32_t)(pw_rows_pitch(rows) * 16); a.D = D; a.E = E;
  a.AP = (char*)alpha_planes; a.ap_rp16 = a.rp16; a.CP = nullptr; a.cp_rp16 = 0;
  a.seq = seq; a.n_seq = n_seq; a.strips = (t_max + 63) / 64; a.scale = scale; a.ignore_self = ignore_self; a.aperture = aperture;
  a.heads = heads; a.dh = D / heads; a.ap_head_bytes = heads == 1 ? 0 : (int64_t)align_up(pw_alpha_bytes(rows, t_max, np), 256);
  a.R = nullptr; a.ldr = 0; a.moments = nullptr; a.csplit = 1;
  const unsigned grid = (unsigned)(8 * ((n_seq + 7) / 8) * a.strips * heads);
  a.stamps = nullptr;
#ifdef SUMK_DIAG
  static unsigned long long* stamp_buf = nullptr;
  static int stamp_calls = 0;
  if (getenv("SUMK_ATTN_STAMPS")) {
    if (!stamp_buf) SUMK_HIP(hipMalloc(&stamp_buf, 4096 * 8 * sizeof(unsigned long long)));
    SUMK_HIP(hipMemsetAsync(stamp_buf, 0, 4096 * 8 * sizeof(unsigned long long), stream));
    a.stamps = grid <= 4096 ? stamp_buf : nullptr;
  }
#endif
  [[maybe_unused]] static const int var = SUMK_TUNE_ENV("SUMK_ATTN_VAR_A") ? atoi(SUMK_TUNE_ENV("SUMK_ATTN_VAR_A")) : ATTN_VAR_A;      // (diagnostic build only)
  constexpr int LDS3 = 3 * (6 * 384 * 16) + 4096, LDS2 = 4 * (4 * 384 * 16) + 4096;
  // non-temporal K / Q loads when the planes do not fit the Infinity Cache beside what the next launch reads (SUMK_ATTN_NT=0 / 1 forces it: A/B)
  static const int nt_env = getenv("SUMK_ATTN_NT") ? atoi(getenv("SUMK_ATTN_NT")) : -1;
  // (single head only: on the Transformer's multi-head launches the rule measured 1 % slower at three planes)
  const bool nt = nt_env >= 0 ? nt_env != 0 : (heads == 1 && pw_planes_bytes(rows, 3 * D, np) > ((size_t)192 << 20));
#define SUMK_A_CASE(NP_, V_, LDS_, I_) { SUMK_TRY(set_lds_once(attn_pw_logits_kernel<NP_, V_>, I_, LDS_)); hipLaunchKernelGGL((attn_pw_logits_kernel<NP_, V_>), dim3(grid), dim3(512), LDS_, stream, a); }
#ifdef SUMK_DIAG
  if (np == 3 && var == 3) { SUMK_A_CASE(3, 3, LDS3, 10) } else if (np == 3 && var == 4) { SUMK_A_CASE(3, 4, LDS3, 11) } else if (np == 3 && var == 5) { SUMK_A_CASE(3, 5, LDS3, 9) } else
  if (np == 3 && (var == 1 || var == 2)) { if (var == 1) SUMK_A_CASE(3, 1, LDS3, 0) else SUMK_A_CASE(3, 2, LDS3, 1) } else
  if (np == 2 && (var == 1 || var == 2)) { if (var == 1) SUMK_A_CASE(2, 1, LDS2, 3) else SUMK_A_CASE(2, 2, LDS2, 4) } else
#endif
  if (heads > 1) {
    constexpr int LDSH3 = 2 * (6 * 384 * 16) + 4096, LDSH2 = 3 * (4 * 384 * 16) + 4096;      // 76 KB each: two blocks per CU
#define SUMK_MH_CASE(NP_, NT_, LDS_, I_) { SUMK_TRY(set_lds_once(attn_pw_logits_mh_kernel<NP_, NT_>, I_, LDS_)); hipLaunchKernelGGL((attn_pw_logits_mh_kernel<NP_, NT_>), dim3(grid), dim3(512), LDS_, stream, a); }
    if (np == 3) { if (nt) SUMK_MH_CASE(3, true, LDSH3, 24) else SUMK_MH_CASE(3, false, LDSH3, 8) }
    else { if (nt) SUMK_MH_CASE(2, true, LDSH2, 25) else SUMK_MH_CASE(2, false, LDSH2, 14) }
#undef SUMK_MH_CASE
  } else
  if (nt) {
    if (np == 3) { SUMK_TRY(set_lds_once(attn_pw_logits_kernel<3, 0, true>, 26, LDS3)); hipLaunchKernelGGL((attn_pw_logits_kernel<3, 0, true>), dim3(grid), dim3(512), LDS3, stream, a); }
    else { SUMK_TRY(set_lds_once(attn_pw_logits_kernel<2, 0, true>, 27, LDS2)); hipLaunchKernelGGL((attn_pw_logits_kernel<2, 0, true>), dim3(grid), dim3(512), LDS2, stream, a); }
  } else
  if (np == 3) SUMK_A_CASE(3, 0, LDS3, 2) else SUMK_A_CASE(2, 0, LDS2, 5)
#undef SUMK_A_CASE
#ifdef SUMK_DIAG
  if (a.stamps && ++stamp_calls == 40) {       // one report, from a warm call
    std::vector<unsigned long long> h((size_t)grid * 8);
    SUMK_HIP(hipStreamSynchronize(stream));
    SUMK_HIP(hipMemcpy(h.data(), stamp_buf, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    unsigned long long rmin = ~0ull, rmax = 0;
    for (unsigned b = 0; b < grid; ++b) if (h[b * 8]) { rmin = std::min(rmin, h[b * 8 + 6]); rmax = std::max(rmax, h[b * 8 + 6] + h[b * 8 + 5]); }
    fprintf(stderr, "[attn logits stamps] grid %u, kernel window %.1f us (first block start to last block end)\n", grid, (rmax - rmin) / 100.0);
    for (unsigned b = 0; b < grid; ++b) {
      const unsigned long long* e = &h[b * 8];
      if (!e[0] || (b % 9 != 0 && e[0] < 300)) continue;
      fprintf(stderr, "  block %3u T %3llu start +%.1f us: prologue %6llu  k-loop %7llu (%.0f / k16 step)  row op + stores %6llu  total %7llu cycles = %.1f us, clock %.0f MHz, xcc %llu cu-id %llx\n",
              b, e[0], (e[6] - rmin) / 100.0, e[1], e[2], e[2] / (double)(a.dh / 16), e[3], e[4], e[5] / 100.0, e[4] / (e[5] / 100.0), e[7] & 0xf, e[7] >> 32);
    }
  }
#endif
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}

int launch_attn_pw_context(int np, const void* qkv_planes, int64_t rows, int D, const void* alpha_planes, void* ctx_planes, const SeqInfo* seq,
                           int n_seq, int t_max, hipStream_t stream, int heads, const float* R, int ldr, float* moments) {
  SUMK_ARG(qkv_planes && alpha_planes && ctx_planes && seq && attn_pw_heads_ok(t_max, D, heads, rows, np), "attn_pw: not eligible (T <= 320, D %% 256, 2 or 3 planes; heads of 128 columns)");
  SUMK_ARG(!R || (heads == 1 && moments && ldr >= D && ldr % 4 == 0 && ((uintptr_t)R & 15) == 0), "attn_pw: the residual form is single-head and needs moments, a pitch %% 4 and a 16-byte aligned residual");
  AttnPwArgs a;
  a.QKV = (const char*)qkv_planes; a.rp16 = (uint32_t)(pw_rows_pitch(rows) * 16); a.D = D; a.E = nullptr;
  a.AP = (char*)const_cast<void*>(alpha_planes); a.ap_rp16 = a.rp16; a.CP = (char*)ctx_planes; a.cp_rp16 = a.rp16;
  a.seq = seq; a.n_seq = n_seq; a.strips = (t_max + 63) / 64; a.scale = 0.f; a.ignore_self = 0; a.aperture = -1;
  a.heads = heads; a.dh = D / heads; a.ap_head_bytes = heads == 1 ? 0 : (int64_t)align_up(pw_alpha_bytes(rows, t_max, np), 256);
  a.R = R; a.ldr = ldr; a.moments = moments; a.csplit = 1;
  constexpr int LDS3 = 4 * (6 + 24) * 1024, LDS2 = 3 * (8 + 32) * 1024, LDS3H = 4 * (12 + 24) * 1024, LDS2H = 3 * (16 + 32) * 1024;
  // single head, D a whole number of 512-column halves: 128-query blocks x half the columns (SUMK_ATTN_WIDE=0: the 64-query strips, A/B)
  static const bool wide_on = !(getenv("SUMK_ATTN_WIDE") && getenv("SUMK_ATTN_WIDE")[0] == '0');
  if (wide_on && D % 512 == 0) {
    a.strips = (t_max + 127) / 128; a.csplit = 2;
    const unsigned gridw = (unsigned)(8 * ((n_seq + 7) / 8) * a.strips * a.csplit);
    constexpr int LDS3W = 4 * (12 + 24) * 1024, LDS2W = 6 * (8 + 16) * 1024, LDS3WH = 3 * (24 + 24) * 1024, LDS2WH = 4 * (16 + 16) * 1024;
#define SUMK_B_WIDE(NP_, HP_, RM_, LDS_, I_) { SUMK_TRY(set_lds_once(attn_pw_context_kernel<NP_, HP_, RM_, 4>, I_, LDS_)); hipLaunchKernelGGL((attn_pw_context_kernel<NP_, HP_, RM_, 4>), dim3(gridw), dim3(512), LDS_, stream, a); }
    if (heads > 1) { if (np == 3) SUMK_B_WIDE(3, true, false, LDS3WH, 22) else SUMK_B_WIDE(2, true, false, LDS2WH, 23) }
    else if (R) { if (np == 3) SUMK_B_WIDE(3, false, true, LDS3W, 18) else SUMK_B_WIDE(2, false, true, LDS2W, 19) }
    else { if (np == 3) SUMK_B_WIDE(3, false, false, LDS3W, 20) else SUMK_B_WIDE(2, false, false, LDS2W, 21) }
#undef SUMK_B_WIDE
    SUMK_HIP(hipGetLastError());
    return SUMK_OK;
  }
  const unsigned grid = (unsigned)(8 * ((n_seq + 7) / 8) * a.strips);
  if (heads > 1) {
    if (np == 3) { SUMK_TRY(set_lds_once(attn_pw_context_kernel<3, true>, 12, LDS3H)); hipLaunchKernelGGL((attn_pw_context_kernel<3, true>), dim3(grid), dim3(512), LDS3H, stream, a); }
    else { SUMK_TRY(set_lds_once(attn_pw_context_kernel<2, true>, 13, LDS2H)); hipLaunchKernelGGL((attn_pw_context_kernel<2, true>), dim3(grid), dim3(512), LDS2H, stream, a); }
  } else if (R) {
    if (np == 3) { SUMK_TRY(set_lds_once(attn_pw_context_kernel<3, false, true>, 16, LDS3)); hipLaunchKernelGGL((attn_pw_context_kernel<3, false, true>), dim3(grid), dim3(512), LDS3, stream, a); }
    else { SUMK_TRY(set_lds_once(attn_pw_context_kernel<2, false, true>, 17, LDS2)); hipLaunchKernelGGL((attn_pw_context_kernel<2, false, true>), dim3(grid), dim3(512), LDS2, stream, a); }
  } else if (np == 3) { SUMK_TRY(set_lds_once(attn_pw_context_kernel<3, false>, 6, LDS3)); hipLaunchKernelGGL((attn_pw_context_kernel<3, false>), dim3(grid), dim3(512), LDS3, stream, a); }
  else { SUMK_TRY(set_lds_once(attn_pw_context_kernel<2, false>, 7, LDS2)); hipLaunchKernelGGL((attn_pw_context_kernel<2, false>), dim3(grid), dim3(512), LDS2, stream, a); }
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}

}  // namespace sumk

// ------------------------------------------------------------------------------------------------ C ABI (tests / probes)
extern "C" size_t sumk_attn_planes_alpha_bytes(int64_t rows, int32_t t_max, int32_t n_planes) {
  if (rows < 1 || t_max < 1 || (n_planes != 2 && n_planes != 3)) return 0;
  return sumk::pw_alpha_bytes(rows, t_max, n_planes);
}

extern "C" int sumk_attn_planes(const void* qkv_planes, int64_t rows, int32_t D, int32_t n_planes, int32_t n_seq, const int32_t* seq_off_host,
                                float scale, int32_t ignore_self, int32_t aperture, float* E, void* alpha_planes, void* ctx_planes, void* stream_) {
  using namespace sumk;
  hipStream_t stream = (hipStream_t)stream_;
  SUMK_ARG(qkv_planes && seq_off_host && alpha_planes && n_seq >= 1 && seq_off_host[0] == 0 && seq_off_host[n_seq] == rows, "attn_planes: bad arguments");
  std::vector<SeqInfo> h(n_seq);
  int64_t eoff = 0; int t_max = 0;
  for (int s = 0; s < n_seq; ++s) {
    const int T = seq_off_host[s + 1] - seq_off_host[s];
    SUMK_ARG(T >= 1, "attn_planes: empty video %d", s);
    h[s].eoff = eoff; h[s].row0 = seq_off_host[s]; h[s].T = T; h[s].ldE = (T + 3) & ~3; h[s].pad_ = 0; h[s].e16off = 0;
    eoff += (int64_t)T * h[s].ldE; t_max = std::max(t_max, T);
  }
  SUMK_ARG(attn_pw_ok(t_max, D, rows, n_planes), "attn_planes: not eligible (T <= 320, D %% 256, 2 or 3 planes)");
  SeqInfo* d = nullptr;
  SUMK_HIP(hipMalloc(&d, sizeof(SeqInfo) * n_seq));
  SUMK_HIP(hipMemcpyAsync(d, h.data(), sizeof(SeqInfo) * n_seq, hipMemcpyHostToDevice, stream));
  int rc = launch_attn_pw_logits(n_planes, qkv_planes, rows, D, E, alpha_planes, d, n_seq, t_max, scale, ignore_self, aperture, stream);
  if (rc == SUMK_OK && ctx_planes) rc = launch_attn_pw_context(n_planes, qkv_planes, rows, D, alpha_planes, ctx_planes, d, n_seq, t_max, stream);
  (void)hipStreamSynchronize(stream);    // (test entry: the table is freed here)
  (void)hipFree(d);
  return rc;
}
