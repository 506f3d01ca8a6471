// Lean 64x64 exact-fp32 MFMA GEMM for the ragged per-video products (Q.K^T: NT, alpha.V: NN) -- a main loop with (almost) no VALU
// instructions.
//
// Why a second kernel.  v_mfma_f32_32x32x2_f32 runs at the f32 VECTOR rate and does not overlap VALU work on its SIMD the way the
// bf16 matrix cores do (scripts/probes/mfma_chain.hip: eight v_fma per MFMA cut the matrix rate to 59 % with one wave per SIMD and
// to 72-75 % with two to four -- every VALU cycle is a lost MFMA cycle).  A 64x64 tile has only 16 MFMAs per wave and k-tile
// (1024 cycles), so the ~50 address / clamp / select instructions the generic register-staged loop spends per k-tile
// (gemm_regstage.h) cost it a fifth of the matrix pipe whatever the occupancy: 73-81 % utilisation in in-kernel stamps with every
// operand L2-resident, the same with one barrier per k-tile, with distinct wave priorities and with capped residency.  Here:
//   * operands are fetched with buffer loads whose per-thread byte offset is computed ONCE per tile; the k advance is a scalar
//     (soffset for K-contiguous rows, a scalar pointer bump for the [k][n] operand): no VALU per load;
//   * two LDS images per operand, one barrier per k-tile, image addresses fixed per thread (the loop is unrolled by two, so the
//     image select is an immediate);
//   * the K tail (K % 32 != 0: alpha.V has K = T) and the last k-tiles are peeled out of the steady-state loop;
//   * a wave whose 32x32 sub-tile lies entirely outside the problem (ragged T) skips its fragment reads and MFMAs.
// Same arithmetic and summation order as gemm_f32_kernel<64, 64, 32, ...>: k ascending, one fmaf chain per element -> results are
// bit-identical to the generic kernel's (tests/test_gpu_vasnet.py::test_lean_gemm_equals_generic_kernel).
//
// SK instances (round 4): the SMALL-BATCH form -- what ONE T ~ 300 video per call needs, i.e. the reference's own calling pattern
// (vasnet.py:193-212 one optimiser step per video, models/__init__.py:45-54 one video per forward).  There a projection is 80-240
// tiles with a 32-k-tile dependent chain each: one block per CU, one wave per SIMD, every ds_read / global-load latency exposed
// (30 us per K = 1024 GEMM = 0.15 of the fp32 peak), and Q.K^T is 25 tiles on 256 CUs.  The SK instances cut K into S slices per tile
// INSIDE the launch -- the problem table carries one entry per (problem, slice), GemmProb::pad_[SK_*] -- so a launch has 3-4 blocks
// per CU: every slice block stores its 64 x 64 partial write-through, the block that draws the last ticket of its tile adds the S
// partials in slice order (deterministic, no float atomics) and runs the epilogue.  Hand-off = cdna_hip_programming.md "In-launch
// split-K reduction", sc1 form: sc1 partial stores -> every storing wave drains vmcnt(0) -> workgroup barrier -> ONE relaxed
// agent-scope ticket add; the reducer reads the partials with sc1 loads ONLY (never a plain load of those bytes), so no fence is
// needed on either side and nothing depends on which XCD a slice ran on.  The instances also add the TN layout (both operands
// [k][row]: dV, dK, weight gradients), a B / C pointer choice per table entry (the three projection matrices, the three gradient
// tensors of ONE launch) and a run-time epilogue (plain / + residual / + bias, ReLU / accumulate).
#include "gemm_device.h"
#include <algorithm>
#include <cstdlib>

namespace sumk {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int LT = 64, LBK = 32;
constexpr int LPITCH = LBK + 4;                  // K-contiguous image: [row][LBK + 4] floats (conflict-free ds_read_b128)
constexpr int LKC = LT * LPITCH;                 // 2304 floats
constexpr int LMC = LBK * LT;                    // [k][n] image: 2048 floats
}

// What one tile needs: wave-uniform scalars + this thread's load offsets (computed once per tile, ~30 VALU instructions).
struct LeanTile {
  const float* Abase; const float* Bbase; float* Cbase;
  int M, N, K, lda, ldb, ldc, m0, n0, nk, has_tail;
  int voA[2], voB[2];
  // SK instances: K slices of the tile (1 = not split), this block's slice, the tile's index in the ticket / partial arrays, residual
  int sk_n, sk_idx, sk_tile, sk_part, ldr; const float* Rbase;
};

template <bool B_KC, bool A_KC = true, bool SK = false>
__global__ __launch_bounds__(256, 4) void gemm_lean_kernel(GemmKArgs ka) {
  static_assert(A_KC || !B_KC, "layouts: NT (A, B K-contiguous), NN (A K-contiguous, B [k][n]), TN (A [k][m], B [k][n])");
  static_assert(A_KC || SK, "the TN form is built as an SK instance only");
  constexpr int AIMG = A_KC ? LKC : LMC;           // floats of one A image
  constexpr int STAGE = AIMG + (B_KC ? LKC : LMC);
  // ONE __shared__ object (a second one beside the staging images can cost a vmcnt(0) in front of every k-tile's fragment reads:
  // cdna_hip_programming.md, "Three .s-level traps"); the SK ticket broadcast uses the four extra floats behind the images
  __shared__ __attribute__((aligned(16))) float lds[2 * STAGE + (SK ? 4 : 0)];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
  const int kq = (tid & 7) * 4;                  // K-contiguous images: this thread's k offset inside a k-tile
#ifdef SUMK_DIAG
  unsigned long long st0 = 0, st_loop = 0, st_epi = 0, rt0 = 0, n_tiles = 0;
  if (ka.dbg_buf) { st0 = __builtin_amdgcn_s_memtime(); rt0 = __builtin_amdgcn_s_memrealtime(); }
#endif

  auto setup = [&](int tile, LeanTile& t) -> bool {
    TileCtx c; GemmProb P;
    int pi = 0;
    if (!decode_tile<LT, LT>(ka, tile, c, P, SK ? &pi : nullptr)) return false;
    t.M = c.M; t.N = c.N; t.K = c.K; t.lda = c.lda; t.ldb = c.ldb; t.ldc = c.ldc; t.m0 = c.m0; t.n0 = c.n0;
    t.nk = (c.K + LBK - 1) / LBK; t.has_tail = (c.K % LBK) != 0;
    // bases = first row (A: m0; K-contiguous B: n0) / first column (B [k][n]: n0) of this tile: offsets stay small
    const float* Bp = ka.B[0];
    float* Cp = ka.C;
    t.sk_n = 1; t.sk_idx = 0; t.sk_tile = 0; t.sk_part = 0; t.ldr = 0; t.Rbase = nullptr;
    if constexpr (SK) {   // the SK fields of the table entry (scalar loads, like load_prob)
      const cptr32 q = (cptr32)(uintptr_t)(ka.probs + pi);
      t.sk_n = max(q[17 + SK_N], 1); t.sk_idx = q[17 + SK_IDX];
      const int tl = (c.m0 / LT) * P.tiles_n + c.n0 / LT;            // this tile inside its problem
      t.sk_tile = q[17 + SK_TILE0] + tl;                               // its ticket
      t.sk_part = q[17 + SK_PART0] + tl * t.sk_n;                      // its first partial tile (slice s: + s)
      const int bs = q[17 + SK_BSEL], cs = q[17 + SK_CSEL];
      Bp = bs == 0 ? ka.B[0] : bs == 1 ? ka.B[1] : bs == 2 ? ka.B[2] : ka.B[3];
      Cp = cs == 0 ? ka.C : cs == 1 ? ka.Csel[1] : cs == 2 ? ka.Csel[2] : ka.Csel[3];
      t.Rbase = ka.R + P.r_off; t.ldr = c.ldr;
    }
    t.Abase = A_KC ? ka.A + P.a_off + (int64_t)c.m0 * c.lda : ka.A + P.a_off + c.m0;
    t.Bbase = B_KC ? Bp + P.b_off + (int64_t)c.n0 * c.ldb : Bp + P.b_off + c.n0;
    t.Cbase = Cp + P.c_off;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      if constexpr (A_KC) {
        const int r = min((tid >> 3) + 32 * p, c.M - 1 - c.m0);       // rows past M are clamped: they only feed rows never stored
        t.voA[p] = (r * c.lda + kq) * 4;
      } else {                                                          // [k][m]: columns past M likewise
        const int col = (tid & 15) * 4;
        t.voA[p] = (((tid >> 4) + 16 * p) * c.lda + (c.m0 + col < c.M ? col : 0)) * 4;
      }
      if constexpr (B_KC) {
        const int n = min((tid >> 3) + 32 * p, c.N - 1 - c.n0);
        t.voB[p] = (n * c.ldb + kq) * 4;
      } else {
        const int col = (tid & 15) * 4;
        t.voB[p] = (((tid >> 4) + 16 * p) * c.ldb + (c.n0 + col < c.N ? col : 0)) * 4;
      }
    }
    return true;
  };

  float4 ra[2], rb[2];
  auto as4 = [](u32x4 v) { return __builtin_bit_cast(float4, v); };
  auto rsrc = [](const float* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), (short)0, 0x7FFFFFFF, 0x00020000); };
  // full k-tile kt (no masking): the k advance is scalar -- soffset for K-contiguous rows, a pointer bump for the [k][n] operand
  auto gload = [&](const LeanTile& t, int kt) {
    if constexpr (A_KC) {
      const __amdgpu_buffer_rsrc_t rA = rsrc(t.Abase);
#pragma unroll
      for (int p = 0; p < 2; ++p) ra[p] = as4(__builtin_amdgcn_raw_buffer_load_b128(rA, t.voA[p], kt * (LBK * 4), 0));
    } else {
      const __amdgpu_buffer_rsrc_t rA = rsrc(t.Abase + (int64_t)kt * LBK * t.lda);
#pragma unroll
      for (int p = 0; p < 2; ++p) ra[p] = as4(__builtin_amdgcn_raw_buffer_load_b128(rA, t.voA[p], 0, 0));
    }
    if constexpr (B_KC) {
      const __amdgpu_buffer_rsrc_t rB = rsrc(t.Bbase);
#pragma unroll
      for (int p = 0; p < 2; ++p) rb[p] = as4(__builtin_amdgcn_raw_buffer_load_b128(rB, t.voB[p], kt * (LBK * 4), 0));
    } else {
      const __amdgpu_buffer_rsrc_t rB = rsrc(t.Bbase + (int64_t)kt * LBK * t.ldb);
#pragma unroll
      for (int p = 0; p < 2; ++p) rb[p] = as4(__builtin_amdgcn_raw_buffer_load_b128(rB, t.voB[p], 0, 0));
    }
  };
  // the K-tail tile (k0 + 32 > K): addresses clamped into the operand, everything at k >= K zeroed (VALU; once per tile at most)
  auto gload_tail = [&](const LeanTile& t, int kt) {
    const int k0 = kt * LBK, K = t.K;
    const int klast = K > 4 ? ((K + 3) & ~3) - 4 : 0;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int k = k0 + kq;
      if constexpr (A_KC) {
        const int r = min((tid >> 3) + 32 * p, t.M - 1 - t.m0);
        ra[p] = *reinterpret_cast<const float4*>(t.Abase + (int64_t)r * t.lda + min(k0 + kq, klast));
        if (k >= K) ra[p].x = 0.f;
        if (k + 1 >= K) ra[p].y = 0.f;
        if (k + 2 >= K) ra[p].z = 0.f;
        if (k + 3 >= K) ra[p].w = 0.f;
      } else {
        const int kr = k0 + (tid >> 4) + 16 * p;
        const int col = (tid & 15) * 4;
        ra[p] = *reinterpret_cast<const float4*>(t.Abase + (int64_t)min(kr, K - 1) * t.lda + (t.m0 + col < t.M ? col : 0));
        if (kr >= K) ra[p] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      if constexpr (B_KC) {
        const int n = min((tid >> 3) + 32 * p, t.N - 1 - t.n0);
        rb[p] = *reinterpret_cast<const float4*>(t.Bbase + (int64_t)n * t.ldb + min(k0 + kq, klast));
        if (k >= K) rb[p].x = 0.f;
        if (k + 1 >= K) rb[p].y = 0.f;
        if (k + 2 >= K) rb[p].z = 0.f;
        if (k + 3 >= K) rb[p].w = 0.f;
      } else {
        const int kr = k0 + (tid >> 4) + 16 * p;
        const int col = (tid & 15) * 4;
        rb[p] = *reinterpret_cast<const float4*>(t.Bbase + (int64_t)min(kr, K - 1) * t.ldb + (t.n0 + col < t.N ? col : 0));
        if (kr >= K) rb[p] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  auto gload_any = [&](const LeanTile& t, int kt) { if (t.has_tail && kt == t.nk - 1) gload_tail(t, kt); else gload(t, kt); };

  // ---- LDS addresses (floats), fixed per thread
  const int wA = A_KC ? (tid >> 3) * LPITCH + kq : (tid >> 4) * LT + (tid & 15) * 4;     // + 32 * LPITCH / + 16 * LT for p = 1
  const int wB = B_KC ? AIMG + (tid >> 3) * LPITCH + kq : AIMG + (tid >> 4) * LT + (tid & 15) * 4;
  const int fA = A_KC ? (wm * 32 + li) * LPITCH + 4 * lh : (4 * lh) * LT + wm * 32 + li;    // + 8 kk / + 8 kk * LT
  const int fB = B_KC ? AIMG + (wn * 32 + li) * LPITCH + 4 * lh : AIMG + (4 * lh) * LT + wn * 32 + li;
  auto swrite = [&](float* img) {
    *reinterpret_cast<float4*>(img + wA) = ra[0];
    *reinterpret_cast<float4*>(img + wA + (A_KC ? 32 * LPITCH : 16 * LT)) = ra[1];
    *reinterpret_cast<float4*>(img + wB) = rb[0];
    *reinterpret_cast<float4*>(img + wB + (B_KC ? 32 * LPITCH : 16 * LT)) = rb[1];
  };
  f32x16 acc;
  // fragments of one k-tile: 4 (kk) x float4 of A and of B per lane, read at the top of the k-tile; the MFMAs then run in two halves
  // with the LDS write of the next k-tile between them (nothing the second half waits for)
  float fa[4][4];
  float fb[4][4];
  auto read_frags = [&](const float* img) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if constexpr (A_KC) {
        const float4 a = *reinterpret_cast<const float4*>(img + fA + 8 * kk);
        fa[kk][0] = a.x; fa[kk][1] = a.y; fa[kk][2] = a.z; fa[kk][3] = a.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) fa[kk][j] = img[fA + (8 * kk + j) * LT];
      }
      if constexpr (B_KC) {
        const float4 b = *reinterpret_cast<const float4*>(img + fB + 8 * kk);
        fb[kk][0] = b.x; fb[kk][1] = b.y; fb[kk][2] = b.z; fb[kk][3] = b.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[kk][j] = img[fB + (8 * kk + j) * LT];
      }
    }
  };
  auto mfma_half = [&](int h) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int kk = 2 * h + q;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk][0], fb[kk][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk][1], fb[kk][1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk][2], fb[kk][2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk][3], fb[kk][3], acc, 0, 0, 0);
    }
  };

  // ---- persistent tile walk b, b + grid, ...: the NEXT tile's decode and first loads ride under the current tile's last k-tile(s)
  float* im0 = lds;
  float* im1 = lds + STAGE;
  int tile = blockIdx.x;
  if (tile >= ka.total_tiles) return;
  LeanTile cur, nxt;
  if (!setup(tile, cur)) return;
  gload_any(cur, 0);
  while (true) {
    bool has_next = false;
    const int next_tile = tile + gridDim.x;
    // what goes into the staging registers after k-tile `k_load - 1` has been written: the tile's k-tile k_load, or -- once, when the
    // tile has no more k-tiles to fetch -- the next tile's first one
    auto next_load = [&](int k_load) {
      if (k_load < cur.nk) { gload_any(cur, k_load); return; }
      if (k_load == cur.nk && next_tile < ka.total_tiles) {
        has_next = setup(next_tile, nxt);
        if (has_next) gload_any(nxt, 0);
      }
    };
    // a wave whose whole 32 x 32 sub-tile is padding (rows >= M or columns >= N) has nothing to multiply
    const bool live = (cur.m0 + wm * 32 < cur.M) && (cur.n0 + wn * 32 < cur.N);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int nk = cur.nk;
    swrite(im0);
    next_load(1);
    __syncthreads();
#ifdef SUMK_DIAG
    unsigned long long sa = 0;
    if (ka.dbg_buf) sa = __builtin_amdgcn_s_memtime();
#endif
    // steady state, unrolled by two (image addresses are immediates): k-tiles kt whose kt + 2 is a FULL k-tile -- no branch, no mask,
    // no VALU instruction
    const int n_steady = cur.has_tail ? nk - 3 : nk - 2;
    int kt = 0;
    auto step = [&](float* cur_img, float* oth_img, int kt_) {
      if (live) { read_frags(cur_img); mfma_half(0); }
      swrite(oth_img);                 // k-tile kt + 1 (its loads were issued a whole k-tile ago)
      if (live) mfma_half(1);
      gload(cur, kt_ + 2);
      __syncthreads();
    };
    for (; kt + 1 < n_steady; kt += 2) {
      step(im0, im1, kt);
      step(im1, im0, kt + 1);
    }
    // the remaining k-tiles (at most four), with the general load
    float* cur_img = im0; float* oth_img = im1;
    for (; kt < nk; ++kt) {
      if (live) { read_frags(cur_img); mfma_half(0); }
      if (kt + 1 < nk) swrite(oth_img);
      if (live) mfma_half(1);
      next_load(kt + 2);
      __syncthreads();
      float* t = cur_img; cur_img = oth_img; oth_img = t;
    }
#ifdef SUMK_DIAG
    unsigned long long sb = 0;
    if (ka.dbg_buf) { asm volatile("" :: "v"(acc[0])); sb = __builtin_amdgcn_s_memtime(); }
#endif

    // ---- SK: a tile cut into K slices.  Partial -> global (sc1), ticket; the last arriver adds the partials in slice order.
    bool finish = true;                       // does this block run the tile's epilogue?  (wave-uniform)
    if constexpr (SK) {
      if (cur.sk_n > 1) {
        // partial-tile image: [wave][q][lane] float4 = acc[4q .. 4q+3] -- 1 KB contiguous per store instruction; the reducer has
        // the same thread <-> element map, so the image needs no other meaning.  num_records ends at the tile's last slice and the
        // slice offset travels in the VECTOR offset (the range check does not see the scalar one): the reducer's loads of slices
        // >= sk_n (it always fetches two at a time) return zeros.
        float* part = ka.sk_part + (int64_t)cur.sk_part * (LT * LT);
        const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc(part, (short)0, cur.sk_n * (LT * LT * 4), 0x00020000);
        const int po = (wave * 4) * 1024 + lane * 16;
        if (live) {
          // (bit-cast the WHOLE accumulator: __builtin_bit_cast of a vector ELEMENT makes hipcc store elements 0..3 four times --
          //  the trap DESIGN.md records for the LSTM packets; checked in the ISA: four distinct register quads)
          typedef unsigned int u32x16 __attribute__((ext_vector_type(16)));
          const u32x16 au = __builtin_bit_cast(u32x16, acc);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const u32x4 v = {au[4 * q], au[4 * q + 1], au[4 * q + 2], au[4 * q + 3]};
            __builtin_amdgcn_raw_buffer_store_b128(v, rP, po + q * 1024 + cur.sk_idx * (LT * LT * 4), 0, 16 /* sc1 */);
          }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // EVERY storing wave drains its write-through stores
        __syncthreads();
        int* ticket = reinterpret_cast<int*>(lds + 2 * STAGE);
        if (tid == 0) *ticket = (int)__hip_atomic_fetch_add(ka.sk_cnt + cur.sk_tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        finish = *ticket == cur.sk_n - 1;
        if (finish) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // no instruction: keeps the compiler from hoisting the loads
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = 0.f;
          if (live) {
            for (int s0 = 0; s0 < cur.sk_n; s0 += 2) {            // slices s0, s0 + 1 in flight together, added in slice order
              u32x4 pv[2][4];
#pragma unroll
              for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                  pv[u][q] = __builtin_amdgcn_raw_buffer_load_b128(rP, po + q * 1024 + (s0 + u) * (LT * LT * 4), 0, 16 /* sc1 */);
#pragma unroll
              for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  const float4 f = as4(pv[u][q]);
                  acc[4 * q] += f.x; acc[4 * q + 1] += f.y; acc[4 * q + 2] += f.z; acc[4 * q + 3] += f.w;
                }
            }
          }
          // the word is zero again for the next launch that uses this ticket array (stream order separates the launches)
          if (tid == 0) __hip_atomic_store(ka.sk_cnt + cur.sk_tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    // ---- epilogue: C/D map of the 32x32 MFMA (col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)); row offsets are scalar
    const int col = cur.n0 + wn * 32 + li;
    if (finish && live && col < cur.N) {
      const int rows_left = cur.M - (cur.m0 + wm * 32) - 4 * lh;      // rows of this lane's half still inside the problem
      const __amdgpu_buffer_rsrc_t rC = rsrc(cur.Cbase + (int64_t)(cur.m0 + wm * 32) * cur.ldc + cur.n0 + wn * 32);
      const int vo = (4 * lh * cur.ldc + li) * 4;
      if constexpr (!SK) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ro = (r & 3) + 8 * (r >> 2);
          if (ro < rows_left) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, acc[r] * ka.alpha), rC, vo, ro * cur.ldc * 4, 0);
        }
      } else {
        // run-time epilogue (kernel-uniform): EPI_NONE alpha acc | EPI_RESIDUAL acc + R | EPI_BIAS_RELU relu(acc + bias0[col]) |
        // EPI_ACCUM C + alpha acc
        const int epi = ka.sk_epi;
        const float bias = epi == EPI_BIAS_RELU ? ka.bias0[0][col] : 0.f;
        const float* rp = cur.Rbase + (int64_t)(cur.m0 + wm * 32 + 4 * lh) * cur.ldr + col;
        const float* cp = cur.Cbase + (int64_t)(cur.m0 + wm * 32 + 4 * lh) * cur.ldc + col;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ro = (r & 3) + 8 * (r >> 2);
          if (ro < rows_left) {
            float v = acc[r] * ka.alpha;
            if (epi == EPI_RESIDUAL) v = acc[r] + rp[(int64_t)ro * cur.ldr];
            else if (epi == EPI_BIAS_RELU) { v = acc[r] + bias; v = (v < 0.f) ? 0.f : v; }      // NaN-propagating, like torch.relu
            else if (epi == EPI_ACCUM) v = cp[(int64_t)ro * cur.ldc] + ka.alpha * acc[r];
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), rC, vo, ro * cur.ldc * 4, 0);
          }
        }
      }
    }
#ifdef SUMK_DIAG
    if (ka.dbg_buf) { const unsigned long long sc = __builtin_amdgcn_s_memtime(); st_loop += sb - sa; st_epi += sc - sb; n_tiles += 1; }
#endif
    if (!has_next) break;
    tile = next_tile;
    cur = nxt;
  }
#ifdef SUMK_DIAG
  if (ka.dbg_buf && tid == 0 && blockIdx.x < 2048) {   // {total, k-loops, epilogues, tiles | -, -, -, k-loops, start, end, hw, xcc}
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long st4 = __builtin_amdgcn_s_memtime();
    unsigned long long* o = ka.dbg_buf + (size_t)blockIdx.x * 4;
    o[0] = st4 - st0; o[1] = st_loop; o[2] = st_epi; o[3] = n_tiles;
    unsigned long long* q = ka.dbg_buf + (size_t)2048 * 4 + (size_t)blockIdx.x * 8;
    q[0] = 0; q[1] = 0; q[2] = 0; q[3] = st_loop; q[4] = rt0; q[5] = __builtin_amdgcn_s_memrealtime();
    q[6] = __builtin_amdgcn_s_getreg((31 << 11) | 4); q[7] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
  }
#endif
}

// Takes a grouped or single-problem launch the generic dispatcher prepared (GemmKArgs: problem table, group_remap, alpha).
int launch_gemm_lean(GemmLayout layout, const GemmKArgs& ka, int tiles, hipStream_t s, int sk) {
  if (sk) {   // small-batch instances: `tiles` counts (tile, K slice) blocks; at most 4 x 256 resident, a longer table is walked
    SUMK_ARG(ka.sk_part && ka.sk_cnt, "gemm_lean: SK launch without partial / ticket buffers");
    SUMK_ARG(ka.sk_epi == EPI_NONE || ka.sk_epi == EPI_RESIDUAL || ka.sk_epi == EPI_BIAS_RELU || ka.sk_epi == EPI_ACCUM, "gemm_lean: SK epilogue %d", ka.sk_epi);
    const dim3 grid(std::min(tiles, 1024)), block(256);
    if (layout == GEMM_NT) hipLaunchKernelGGL((gemm_lean_kernel<true, true, true>), grid, block, 0, s, ka);
    else if (layout == GEMM_NN) hipLaunchKernelGGL((gemm_lean_kernel<false, true, true>), grid, block, 0, s, ka);
    else hipLaunchKernelGGL((gemm_lean_kernel<false, false, true>), grid, block, 0, s, ka);
    return SUMK_OK;
  }
  SUMK_ARG(layout == GEMM_NT || layout == GEMM_NN, "gemm_lean: NT and NN layouts only");
  // persistent: at most 4 blocks per CU (what the two LDS images admit) are resident; block b walks tiles b, b + grid, ...
  static const int lean_grid = SUMK_TUNE_ENV("SUMK_LEAN_GRID") ? atoi(SUMK_TUNE_ENV("SUMK_LEAN_GRID")) : 1024;
  const dim3 grid(std::min(tiles, lean_grid)), block(256);
  if (layout == GEMM_NT) hipLaunchKernelGGL(gemm_lean_kernel<true>, grid, block, 0, s, ka);
  else hipLaunchKernelGGL(gemm_lean_kernel<false>, grid, block, 0, s, ka);
  return SUMK_OK;
}

}  // namespace sumk
