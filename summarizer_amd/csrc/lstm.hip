// Bidirectional LSTM layer for a packed batch of videos on gfx950.
// Reference: nn.LSTM inside DSN (summarizer/models/dsn.py:23-27,45) and sLSTM (summarizer/models/sumgan.py:27-32,43);
// gate order i,f,g,o, two bias vectors, h0 = c0 = 0 (torch.nn.LSTM).
//
// 1. Input projection hoisted out of the recurrence: G = X . [W_ih_fwd ; W_ih_rev]^T + b_ih + b_hh  -> (n_rows, 8H),
//    ONE grouped-B MFMA GEMM (gemm_f32.hip) with both biases fused in the epilogue.
// 2. Recurrence: one launch per time step t covering EVERY (video, direction) still running at t.
//    A dependent kernel boundary costs ~1.5 us on MI355X, less than a grid-wide barrier (4-5 us), so the
//    step loop is a chain of small launches rather than a persistent kernel (MI355X_MICROARCH.md price list).
//    Step kernel = a skinny MFMA GEMM  pre[video, gate-col] = h_prev[video,:] . W_hh[gate-col,:]^T :
//      block = (32 videos) x (8 hidden units x 4 gates) x direction; its 4 waves split K = H (chunks of 8),
//      fragments are loaded straight from L2 (h_prev rows / W_hh rows, 16 B per lane), partial tiles are summed
//      through LDS, then 256 threads = 32 videos x 8 units apply  +G, sigmoid/tanh, c/h update  in registers.
//    W_hh (1 MB for DSN) stays L2-resident across steps; h_prev is read from the output rows written one step
//    earlier (t-1 for the forward direction, t+1 for the reverse one), so no separate state buffer exists.
#include "sumk_internal.h"
#include <cstring>
#include <cstdio>
#include <math.h>
#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace sumk {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int PSTATE_WORDS = 16 + 16384;  // word 0: error flag; words 16..: step counters per work item (one word; lstm_wide2_kernel: four shards on
                                          // lines of their own, 128 words per item); the last 512 words: phase stamps of the diagnostic build

struct LstmWs {
  size_t g, cstate, prob, pstate, gates, call, hprev, dg, slab, dhrec, dcstate, prob_sk, colpart, pstate_b, xchg, partial, total;
  size_t xchg_bytes;
  size_t ll, ll_bytes;     // forward recurrence hand-off buffer (H <= 256): directly behind pstate, zeroed with it
  size_t llb, llb_bytes;   // backward recurrence hand-off buffer (H <= 256): directly behind pstate_b, zeroed with it
  size_t wx, wx_bytes;     // forward recurrence exchange of lstm_wide2_kernel (256 < H <= 1024): [group][step parity 2][direction 2][k/4 256][row 64][4 floats]
  size_t slab_elems;
  int32_t n_rows, t_max;
};

constexpr int GV_MAXB = 8;      // "a few sequences": the small-batch mat-vec step kernels below take over up to this many
// geometry of the transposed mat-vec: strips of 256 columns x k splits sized for ~2048 blocks
struct TGemvGeom { int strips, n_ksplit, rows_per_split; };
static TGemvGeom tgemv_geom(int H) {
  TGemvGeom g;
  g.strips = (H + 255) / 256;
  const int H4 = 4 * H;
  int want = std::max(1, 2048 / g.strips);                // ~8 blocks = 32 waves per CU: the weight stream needs the loads in flight
  want = std::min(want, std::max(1, H4 / 32));            // at least 32 rows per split (8 per wave)
  want = std::min(want, 128);
  g.rows_per_split = ((H4 + want - 1) / want + 3) / 4 * 4;
  g.n_ksplit = (H4 + g.rows_per_split - 1) / g.rows_per_split;
  return g;
}
static size_t tgemv_partial_bytes(int H, int n_seq, int nd = 1) {
  return (size_t)tgemv_geom(H).n_ksplit * (size_t)std::min(n_seq, GV_MAXB) * nd * H * 4;
}

static int lstm_carve(int In, int H, int n_seq, const int32_t* off, int training, LstmWs* w) {
  SUMK_ARG(In > 0 && In % 4 == 0, "bilstm: input size %d must be a positive multiple of 4", In);
  SUMK_ARG(H > 0 && H % 4 == 0, "bilstm: hidden size %d must be a positive multiple of 4", H);
  SUMK_ARG(n_seq > 0 && off != nullptr && off[0] == 0, "bilstm: empty batch / seq_off[0] != 0");
  int tmax = 0;
  for (int s = 0; s < n_seq; ++s) {
    int T = off[s + 1] - off[s];
    SUMK_ARG(T > 0, "bilstm: video %d has %d frames", s, T);
    tmax = T > tmax ? T : tmax;
  }
  const size_t R = (size_t)off[n_seq];
  size_t p = 0;
  auto take = [&](size_t bytes) { size_t at = p; p += align_up(bytes, 256); return at; };
  w->n_rows = (int32_t)R; w->t_max = tmax;
  w->g = take(R * 8 * H * 4);                       // pre-activations from the input projection
  w->cstate = take((size_t)n_seq * 2 * H * 4);      // running cell state (inference)
  w->prob = take(8 * sizeof(GemmProb));
  w->pstate = take(PSTATE_WORDS * 4);               // persistent-kernel error flag + per-item step counters
  // flag-in-data hand-off of the H <= 256 forward recurrence: [step parity 2][direction 2][video][H] x {float h, uint32 step tag}
  w->ll_bytes = H <= 256 ? (size_t)4 * n_seq * H * 8 : 0;
  w->ll = take(w->ll_bytes);
  w->wx_bytes = (H > 256 && H <= 1024) ? (size_t)((n_seq + 63) / 64) * 4 * 256 * 64 * 16 : 0;
  w->wx = take(w->wx_bytes);
  w->gates = w->call = w->hprev = w->dg = w->slab = w->dhrec = w->dcstate = w->prob_sk = w->colpart = 0;
  w->pstate_b = w->xchg = 0; w->xchg_bytes = 0; w->llb = 0; w->llb_bytes = 0;
  w->slab_elems = 0;
  if (training) {
    w->gates = take(R * 8 * H * 4);                 // post-nonlinearity i,f,g,o per row and direction
    w->call = take(R * 2 * H * 4);                  // cell state per row and direction
    w->hprev = take(R * 2 * H * 4);                 // h_{t-1} per row and direction (0 at a sequence start)
    w->dg = take(R * 8 * H * 4);                    // gradient w.r.t. gate pre-activations
    w->dcstate = take((size_t)n_seq * 2 * H * 4);
    size_t big = (size_t)(8 * H) * (size_t)(In > H ? In : H);
    w->slab_elems = (size_t)8 * big;                // split-K partial slabs for the weight gradients
    w->slab = take(w->slab_elems * 4);
    w->prob_sk = take(64 * sizeof(GemmProb));
    w->colpart = take((size_t)128 * 8 * H * 4);
    w->pstate_b = take(PSTATE_WORDS * 4);
    {  // flag-in-data exchange of the H <= 256 BPTT: [step parity 2][item][member 32][video < gsize][H] x {float partial, uint32 step tag}
      const int gsz = std::min(32, std::max(1, (2 * n_seq + 7) / 8));
      const int items = 2 * ((n_seq + gsz - 1) / gsz);
      static const bool ll_bwd = getenv("SUMK_LSTM_LL_BWD") && getenv("SUMK_LSTM_LL_BWD")[0] == '1';   // (opt-in variant: no buffer otherwise)
      w->llb_bytes = (ll_bwd && H <= 256) ? (size_t)2 * items * 32 * gsz * H * 8 : 0;
      w->llb = take(w->llb_bytes);
    }
    {  // persistent BPTT exchange: [parity 2][item][member 32][video 32][H] partial sums of dh (H <= 256 only)
      int gsize = std::min(32, std::max(1, (2 * n_seq + 7) / 8));
      int items = 2 * ((n_seq + gsize - 1) / gsize);
      w->xchg_bytes = H <= 256 ? (size_t)2 * items * 32 * 32 * H * 4 : 0;
      // wide persistent BPTT (256 < H <= 1024, H % 128 == 0): [dir 2][group parity 2][step parity 2][KG = H/32][64 videos][H]
      if (H > 256 && H <= 1024 && H % 128 == 0) w->xchg_bytes = (size_t)8 * (H / 32) * 64 * H * 4;
      w->xchg = take(w->xchg_bytes);
    }
    w->partial = take(n_seq <= GV_MAXB ? tgemv_partial_bytes(H, n_seq, 2) : 0);   // small-batch transposed mat-vec (H > 256 at a few videos)
  }
  w->total = p;
  return SUMK_OK;
}

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }
// Gate math of the persistent H <= 256 recurrence, whose per-step latency chain contains it: v_exp_f32 / v_rcp_f32 forms (a few ulp:
// ~3e-7 relative for the sigmoid, ~1e-7 ABSOLUTE for tanh) instead of the library expf / IEEE division / tanhf (~170 instructions per
// step and thread against ~30).  tanh(x) = 1 - 2 / (1 + e^{2x}) saturates correctly at both ends (e^{2x} -> inf gives 1, -> 0 gives -1).
// (round 6: __frcp_rn is the CORRECTLY ROUNDED reciprocal -- hipcc expands it to the v_div_scale / v_div_fmas / v_div_fixup sequence, 5 x ~10
//  instructions in every cell update; __builtin_amdgcn_rcpf is the bare v_rcp_f32 these forms were written for: 1 ulp)
__device__ __forceinline__ float fast_sigmoid(float v) { return __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
__device__ __forceinline__ float fast_tanh(float v) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * v)); }

// x = hi + lo with hi = bf16(x), lo = bf16(x - hi): the operand split of the bf16x3 arithmetic (gemm_f32.hip)
__device__ __forceinline__ void split8(f32x4 x0, f32x4 x1, bf16x8& hi, bf16x8& lo) {
  const f32x8 x = __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7);
  hi = __builtin_convertvector(x, bf16x8);
  lo = __builtin_convertvector(x - __builtin_convertvector(hi, f32x8), bf16x8);
}
// x = p1 + p2 + p3 EXACTLY (three bf16 planes: 3 x 8 significand bits), the operand split of the bf16x6 arithmetic
__device__ __forceinline__ void split8x3(f32x4 x0, f32x4 x1, bf16x8& p1, bf16x8& p2, bf16x8& p3) {
  f32x8 x = __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7);
  p1 = __builtin_convertvector(x, bf16x8);
  x = x - __builtin_convertvector(p1, f32x8);
  p2 = __builtin_convertvector(x, bf16x8);
  x = x - __builtin_convertvector(p2, f32x8);
  p3 = __builtin_convertvector(x, bf16x8);
}

// ------------------------------------------------------------------------------------------- small-batch (mat-vec) steps
// With a handful of sequences (SumGAN trains one video at a time) a recurrence step is a matrix-VECTOR product: nothing to
// feed an MFMA tile with, and the step time is the time to stream the weights (up to 134 MB per decoder step) once.  The
// MFMA step kernels above read them in 32-byte (forward) or 128-byte-strided (backward, transposed) pieces from at most
// 64 blocks; these kernels are shaped for bandwidth instead (n_seq <= GV_MAXB):
//   forward:  one WAVE per hidden unit streams its 4 gate rows of W_ih / W_hh, lanes along k (1 KB contiguous per load),
//             dot products on the VALU, butterfly reduction across the wave, lane b finishes the cell of sequence b;
//   backward: the transposed product dh = dG W is cut into (256-column strip) x (k split) blocks, lanes along the columns
//             (1 KB contiguous per row), partial sums per split; a second tiny kernel sums the splits and does the cell backward.

struct GemvStepArgs {
  const float* a1; int32_t a1_shift;     // first operand rows (R, H): row + shift (-1: previous step, 0: same step); nullptr = none
  const float* w1;                       // (4H, H) or nullptr
  const float* G;                        // (R, 4H) hoisted input projection incl. both biases, or nullptr
  const float* b1; const float* b2;      // (4H) biases when G == nullptr
  const float* w2;                       // (4H, H) recurrent weights
  const float* h0; const float* c0;      // (n_seq, H) or nullptr
  float* hseq;                           // (R, H)
  float* c_all; float* cstate;           // (R, H) per-row cell states, or (n_seq, H) running state when c_all == nullptr
  float* gates; float* hprev; float* xin;// saves for the backward pass (nullptr in inference)
  const int32_t* off;
  int32_t n_seq, H, t;
};

__global__ __launch_bounds__(256) void lstm_gemv_step_kernel(GemvStepArgs a) {
  const int H = a.H, t = a.t, nb = a.n_seq;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = blockIdx.x * 4 + wave;
  if (j >= H) return;                    // whole wave
  int64_t row[GV_MAXB]; bool act[GV_MAXB];
#pragma unroll
  for (int b = 0; b < GV_MAXB; ++b) {
    act[b] = false; row[b] = 0;
    if (b < nb) { const int r0 = a.off[b], T = a.off[b + 1] - r0; if (t < T) { act[b] = true; row[b] = r0 + t; } }
  }
  float acc[4][GV_MAXB];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int b = 0; b < GV_MAXB; ++b) acc[q][b] = 0.f;
  for (int pair = 0; pair < 2; ++pair) {
    const float* w = pair == 0 ? a.w1 : a.w2;
    if (w == nullptr) continue;
    const float* src[GV_MAXB];
#pragma unroll
    for (int b = 0; b < GV_MAXB; ++b) {
      src[b] = nullptr;
      if (b < nb && act[b]) {
        if (pair == 0) { if (a.a1 && (a.a1_shift == 0 || t > 0)) src[b] = a.a1 + (row[b] + a.a1_shift) * H; }
        else src[b] = t > 0 ? a.hseq + (row[b] - 1) * H : (a.h0 ? a.h0 + (int64_t)b * H : nullptr);
      }
    }
    bool any = false;
#pragma unroll
    for (int b = 0; b < GV_MAXB; ++b) any |= src[b] != nullptr;
    if (!any) continue;                  // uniform: e.g. layer 0 of the decoder at t = 0, or no initial state
    // 4 k-chunks = 16 weight loads (256 B) in flight per lane before the first FMA.  (Measured neutral: 20.6 us per step on
    // the SumGAN mix either way, as was one block per unit with K split over its waves, and rotating the start column per
    // unit -- the 64-128 MB weight stream of an H = 2048 step already runs at 3-4 TB/s.)
    for (int k0 = 4 * lane; k0 < H; k0 += 1024) {
      float4 wq[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = k0 + 256 * u;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          wq[u][q] = k < H ? *reinterpret_cast<const float4*>(w + (int64_t)(q * H + j) * H + k) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int b = 0; b < GV_MAXB; ++b) {
        if (b < nb && src[b] != nullptr) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int k = k0 + 256 * u;
            const float4 x = k < H ? *reinterpret_cast<const float4*>(src[b] + k) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q][b] += (wq[u][q].x * x.x + wq[u][q].y * x.y) + (wq[u][q].z * x.z + wq[u][q].w * x.w);
          }
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int b = 0; b < GV_MAXB; ++b)
      if (b < nb) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) acc[q][b] += __shfl_xor(acc[q][b], m, 64);
      }
  // lane b finishes sequence b
#pragma unroll
  for (int b = 0; b < GV_MAXB; ++b) {
    if (lane != b || b >= nb || !act[b]) continue;
    const int64_t r = row[b];
    float pre[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      pre[q] = acc[q][b] + (a.G ? a.G[r * 4 * H + q * H + j] : a.b1[q * H + j] + a.b2[q * H + j]);
    float cprev;
    if (t > 0) cprev = a.c_all ? a.c_all[(r - 1) * H + j] : a.cstate[(int64_t)b * H + j];
    else cprev = a.c0 ? a.c0[(int64_t)b * H + j] : 0.f;
    const float ig = sigmoidf_(pre[0]), fg = sigmoidf_(pre[1]), gg = tanhf(pre[2]), og = sigmoidf_(pre[3]);
    const float c = fg * cprev + ig * gg;
    const float h = og * tanhf(c);
    if (a.hprev) a.hprev[r * H + j] = t > 0 ? a.hseq[(r - 1) * H + j] : (a.h0 ? a.h0[(int64_t)b * H + j] : 0.f);
    if (a.xin) a.xin[r * H + j] = (a.a1 && (a.a1_shift == 0 || t > 0)) ? a.a1[(r + a.a1_shift) * H + j] : 0.f;
    a.hseq[r * H + j] = h;
    if (a.c_all) a.c_all[r * H + j] = c; else a.cstate[(int64_t)b * H + j] = c;
    if (a.gates) { float* gs = a.gates + r * 4 * H; gs[j] = ig; gs[H + j] = fg; gs[2 * H + j] = gg; gs[3 * H + j] = og; }
  }
}

// partial[split][b][j] = sum over this split's k of  dGa[row_b + 1][k] Wa[k][j]  (+ dGb[row_b + b_shift][k] Wb[k][j])
struct TGemvArgs {
  const float* dGa; const float* wa;                    // own dG (R, nd*4H), W_hh (4H, H) of direction 0
  const float* dGb; const float* wb; int32_t b_shift;   // consumer's dG and W_ih, or nullptr (nd = 1 only)
  float* partial;                                       // (n_ksplit, n_seq, nd, H)
  const int32_t* off;
  int32_t n_seq, H, t, rows_per_split;
  int32_t nd = 1;                                       // 2: bidirectional layer, blockIdx.z = direction (1 runs backwards in time)
  const float* wa1 = nullptr;                           // W_hh of direction 1
};

__global__ __launch_bounds__(256) void lstm_tgemv_partial_kernel(TGemvArgs a) {
  __shared__ float4 red[3][GV_MAXB][64];
  const int H = a.H, H4 = 4 * H, t = a.t, nb = a.n_seq, nd = a.nd, d = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int jq = blockIdx.x * 256 + 4 * lane;            // this lane's 4 columns
  const bool jok = jq < H;
  const int kbeg = blockIdx.y * a.rows_per_split, kend = min(H4, kbeg + a.rows_per_split);
  float4 acc[GV_MAXB];
#pragma unroll
  for (int b = 0; b < GV_MAXB; ++b) acc[b] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int pair = 0; pair < 2; ++pair) {
    const float* w = pair == 0 ? (d == 0 ? a.wa : a.wa1) : a.wb;
    const float* dG = pair == 0 ? a.dGa : a.dGb;
    if (w == nullptr || dG == nullptr) continue;
    const float* g[GV_MAXB];
    bool any = false;
#pragma unroll
    for (int b = 0; b < GV_MAXB; ++b) {
      g[b] = nullptr;
      if (b < nb) {
        const int r0 = a.off[b], T = a.off[b + 1] - r0;
        const int sh = pair == 0 ? 1 : a.b_shift;
        if ((pair == 0 || t >= 0) && t + sh < T && t + sh >= 0) {
          const int64_t row = d == 0 ? r0 + t + sh : r0 + T - 1 - (t + sh);     // the row of step t + sh in this direction
          g[b] = dG + row * (nd * H4) + d * H4; any = true;
        }
      }
    }
    if (!any) continue;
    for (int k0 = kbeg + wave; k0 < kend; k0 += 16) {   // 4 rows in flight per lane before the first FMA
      float4 w4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = k0 + 4 * u;
        w4[u] = (jok && k < kend) ? *reinterpret_cast<const float4*>(w + (int64_t)k * H + jq) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int b = 0; b < GV_MAXB; ++b) {
        if (b < nb && g[b] != nullptr) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int k = k0 + 4 * u;
            const float gv = k < kend ? g[b][k] : 0.f;
            acc[b].x += gv * w4[u].x; acc[b].y += gv * w4[u].y; acc[b].z += gv * w4[u].z; acc[b].w += gv * w4[u].w;
          }
        }
      }
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int b = 0; b < GV_MAXB; ++b) if (b < nb) red[wave - 1][b][lane] = acc[b];
  }
  __syncthreads();
  if (wave == 0 && jok) {
#pragma unroll
    for (int b = 0; b < GV_MAXB; ++b) {
      if (b >= nb) continue;
      float4 v = acc[b];
#pragma unroll
      for (int w3 = 0; w3 < 3; ++w3) { const float4 o = red[w3][b][lane]; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
      *reinterpret_cast<float4*>(a.partial + (((int64_t)blockIdx.y * nb + b) * nd + d) * H + jq) = v;
    }
  }
}

// cell backward of one (layer, step) for n_seq <= GV_MAXB: dh = dext + dh_last (last step) + sum of the mat-vec partials
struct CellBwdArgs {
  const float* dext;                     // (R, H) or nullptr
  const float* dh_last; const float* dc_last;   // (n_seq, H) or nullptr
  const float* partial; int32_t n_ksplit;
  const float* gates; const float* c_all; const float* c0;
  float* dG; float* dcstate; float* dh0; float* dc0;
  const int32_t* off;
  int32_t n_seq, H, t;
  int32_t nd = 1;                        // 2: bidirectional layer (blockIdx.z = direction; state tensors are (n_seq, nd, H))
};

// block = 64 columns x 4 split groups: the n_ksplit partial sums of a column are added by 4 threads (independent,
// unrolled loads) and combined through LDS -- a single thread walking 64-128 dependent loads made this kernel 22 us.
__global__ __launch_bounds__(256) void lstm_cellbwd_kernel(CellBwdArgs a) {
  __shared__ float red[4][64];
  const int H = a.H, H4 = 4 * H, t = a.t, nd = a.nd, d = blockIdx.z, S1 = nd * H, S4 = nd * H4;
  const int b = blockIdx.y, jl = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + jl;
  const int r0 = a.off[b], T = a.off[b + 1] - r0;
  if (t >= T) return;                                   // whole block
  float rec = 0.f;
  if (j < H) {
    const float* p = a.partial + ((int64_t)b * nd + d) * H + j;
    const int64_t stride = (int64_t)a.n_seq * nd * H;
    int s = grp;
    for (; s + 12 < a.n_ksplit; s += 16) {
      const float v0 = p[s * stride], v1 = p[(s + 4) * stride], v2 = p[(s + 8) * stride], v3 = p[(s + 12) * stride];
      rec += (v0 + v1) + (v2 + v3);
    }
    for (; s < a.n_ksplit; s += 4) rec += p[s * stride];
  }
  red[grp][jl] = rec;
  __syncthreads();
  if (grp != 0 || j >= H) return;
  rec = (red[0][jl] + red[1][jl]) + (red[2][jl] + red[3][jl]);
  const int64_t idx = ((int64_t)b * nd + d) * H + j;
  float* dcs = a.dcstate + idx;
  if (t < 0) {
    if (a.dh0) a.dh0[idx] = rec;
    if (a.dc0) a.dc0[idx] = *dcs;
    return;
  }
  const int64_t row = d == 0 ? r0 + t : r0 + T - 1 - t;
  float dh = (a.dext ? a.dext[row * S1 + d * H + j] : 0.f) + rec;   // rec is zero by construction when no later step feeds this one
  if (t + 1 == T && a.dh_last) dh += a.dh_last[idx];
  const float* gs = a.gates + row * S4 + d * H4;
  const float ig = gs[j], fg = gs[H + j], gg = gs[2 * H + j], og = gs[3 * H + j];
  const float c = a.c_all[row * S1 + d * H + j];
  const float cprev = t > 0 ? a.c_all[(d == 0 ? row - 1 : row + 1) * S1 + d * H + j] : (a.c0 ? a.c0[idx] : 0.f);
  const float tc = tanhf(c);
  const float dc = (t + 1 < T ? *dcs : (a.dc_last ? a.dc_last[idx] : 0.f)) + dh * og * (1.f - tc * tc);
  float* dg = a.dG + row * S4 + d * H4;
  dg[j] = dc * gg * ig * (1.f - ig);
  dg[H + j] = dc * cprev * fg * (1.f - fg);
  dg[2 * H + j] = dc * ig * (1.f - gg * gg);
  dg[3 * H + j] = dh * tc * og * (1.f - og);
  *dcs = dc * fg;
}


// ------------------------------------------------------------------------------------------- step kernel
// grid.x = n_mtiles * n_ublk * 2 ; block = 256
struct StepArgs {
  const float* G;        // (R, 8H)
  const float* whh[2];   // (4H, H)
  float* Hout;           // (R, 2H)
  float* cstate;         // (n_seq, 2, H) running cell state, or nullptr when c_all is used
  float* gates;          // (R, 8H) or nullptr
  float* c_all;          // (R, 2H) or nullptr
  float* hprev;          // (R, 2H) or nullptr
  const int32_t* off;
  int32_t n_seq, H, t, n_ublk;
  // nd = 2: bidirectional layer (row layouts (R, 2H) / (R, 8H), dir 1 runs backwards in time); nd = 1: one forward-running
  // direction (row layouts (R, H) / (R, 4H)).  h0 / c0: optional initial state (n_seq, nd, H); nullptr = zeros.
  int32_t nd = 2;
  const float* h0 = nullptr;
  const float* c0 = nullptr;
};

__global__ __launch_bounds__(256) void lstm_step_kernel(StepArgs a) {
  __shared__ float part[4][32][33];
  const int H = a.H, nd = a.nd, S1 = nd * H, S4 = nd * 4 * H;
  const int d = blockIdx.x % nd;
  const int ublk = (blockIdx.x / nd) % a.n_ublk;
  const int mtile = (blockIdx.x / nd) / a.n_ublk;
  const bool rec = a.t > 0 || a.h0 != nullptr;      // a recurrent term exists at this step
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int j0 = ublk * 8;
  const int t = a.t;

  // ---- epilogue operands first: their loads (G row slice, previous cell state) do not depend on the mat-vec, so they
  // are issued before it and their latency hides under the fragment loads / MFMAs.
  const int ei = tid >> 3, eu = tid & 7;
  const int esv = mtile * 32 + ei, j = j0 + eu;
  bool eact = false;
  int64_t row = 0, prow = 0;
  float pre[4] = {0.f, 0.f, 0.f, 0.f};
  float cprev = 0.f;
  if (esv < a.n_seq && j < H) {
    const int r0 = a.off[esv], T = a.off[esv + 1] - r0;
    if (t < T) {
      eact = true;
      row = d == 0 ? r0 + t : r0 + T - 1 - t;
      prow = d == 0 ? row - 1 : row + 1;
      const float* g = a.G + row * S4 + d * 4 * H;
#pragma unroll
      for (int q = 0; q < 4; ++q) pre[q] = g[q * H + j];
      if (t > 0) cprev = a.c_all ? a.c_all[prow * S1 + d * H + j] : a.cstate[((int64_t)esv * nd + d) * H + j];
      else if (a.c0) cprev = a.c0[((int64_t)esv * nd + d) * H + j];
    }
  }

  if (rec) {
    // A operand: h_prev of video (mtile*32 + li); B operand: W_hh row of gate-column li = g*8 + u.
    // Fragments come straight from memory (L2 does not survive the kernel boundary: Infinity Cache / HBM).  Measured
    // alternatives: staging the 32x128 panels through LDS in 128-B-coalesced segments was SLOWER (12.3 vs 10.5 us per
    // step at 50 videos, H=256: two extra barriers on a latency-bound kernel); see DESIGN.md "LSTM step latency".
    const int sv = mtile * 32 + li;
    bool act = false;
    const float* hp = a.Hout;  // clamped to a legal row when inactive
    if (sv < a.n_seq) {
      int r0 = a.off[sv], T = a.off[sv + 1] - r0;
      if (t < T) {
        act = true;
        hp = t > 0 ? a.Hout + (int64_t)(d == 0 ? r0 + t - 1 : r0 + T - t) * S1 + d * H : a.h0 + ((int64_t)sv * nd + d) * H;
      }
    }
    const int gcol = li >> 3, u = li & 7;
    const int unit = min(j0 + u, H - 1);
    const float* wp = a.whh[d] + (int64_t)(gcol * H + unit) * H;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int nchunk = (H + 7) >> 3;
    // chunks kk = wave, wave+4, ...; handled 8 at a time with ALL 16 fragment loads in flight before the first MFMA
    for (int kb = wave; kb < nchunk; kb += 32) {
      float4 av[8], bv[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int k = (kb + 4 * q) * 8 + 4 * lh, kc = min(k, H - 4);
        bv[q] = *reinterpret_cast<const float4*>(wp + kc);
        av[q] = *reinterpret_cast<const float4*>(hp + kc);
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int k = (kb + 4 * q) * 8 + 4 * lh;
        if (!act || k >= H) av[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k >= H) bv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].x, bv[q].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].y, bv[q].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].z, bv[q].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].w, bv[q].w, acc, 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) part[wave][(r & 3) + 8 * (r >> 2) + 4 * lh][li] = acc[r];
    __syncthreads();
  }

  // ---- cell update: thread = (video ei, unit eu)
  if (!eact) return;
  const int i = ei, u = eu;
  if (rec) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      pre[q] += (part[0][i][q * 8 + u] + part[1][i][q * 8 + u]) + (part[2][i][q * 8 + u] + part[3][i][q * 8 + u]);
  }
  const float ig = sigmoidf_(pre[0]), fg = sigmoidf_(pre[1]), gg = tanhf(pre[2]), og = sigmoidf_(pre[3]);
  const int sv = esv;
  const float c = fg * cprev + ig * gg;
  const float h = og * tanhf(c);
  a.Hout[row * S1 + d * H + j] = h;
  if (a.c_all) a.c_all[row * S1 + d * H + j] = c; else a.cstate[((int64_t)sv * nd + d) * H + j] = c;
  if (a.gates) {
    float* gs = a.gates + row * S4 + d * 4 * H;
    gs[j] = ig; gs[H + j] = fg; gs[2 * H + j] = gg; gs[3 * H + j] = og;
  }
  if (a.hprev) a.hprev[row * S1 + d * H + j] = t > 0 ? a.Hout[prow * S1 + d * H + j] : (a.h0 ? a.h0[((int64_t)sv * nd + d) * H + j] : 0.f);
}

// scores[r] = sigmoid(h[r,:] . w + b)       dsn.py:34-36,46 / sumgan.py:33-34,44-45
__global__ __launch_bounds__(256) void frame_head_kernel(const float* __restrict__ h, const float* __restrict__ w,
                                                         const float* __restrict__ b, float* __restrict__ scores,
                                                         int n_rows, int F) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const int lane = threadIdx.x & 63;
  const float4* x4 = reinterpret_cast<const float4*>(h + (int64_t)row * F);
  const float4* w4 = reinterpret_cast<const float4*>(w);
  float dot = 0.f;
  for (int c = lane; c < (F >> 2); c += 64) {
    float4 v = x4[c], ww = w4[c];
    dot += (v.x * ww.x + v.y * ww.y) + (v.z * ww.z + v.w * ww.w);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
  if (lane == 0) scores[row] = sigmoidf_(dot + b[0]);
}


// ------------------------------------------------------------------------------------------- persistent recurrence
// One launch runs ALL time steps (H <= 256).  256 blocks (one per CU, cooperative launch so all are resident) form 8 teams
// of 32; team = blockIdx % 8, which under the observed round-robin dispatch puts a team on one XCD -- used for SPEED only:
// every cross-block hand-off follows the placement-independent protocol of cdna_hip_programming.md Guideline 16 (R1):
//   producer: payload stored write-through (sc1: agent-scope relaxed atomic stores), every storing wave drains vmcnt(0),
//             workgroup barrier, ONE lane adds to the item's step counter (agent-scope relaxed atomic);
//   consumer: ONE lane polls the counter (relaxed, agent scope, s_sleep), workgroup barrier, then EVERY load of handed-off
//             bytes is an sc1 load (agent-scope relaxed atomic load) -- no plain load ever touches them.
// A work item = (video group <= 32 videos, direction); items are dealt to teams round-robin.  Team member m owns hidden
// units [m*upm, (m+1)*upm) of every video of the item: its 4*upm rows of W_hh stay in LDS for the whole item, its cell
// state stays in registers, and per step it needs only h_{t-1} of the group (<= 32 KB, published by the 32 members).
// Versus the launch-per-step chain this removes the kernel boundary, the per-step re-read of W_hh (L2 does not survive a
// boundary) and three dependent global-load hops: measured step time in DESIGN.md.
struct PersistArgs {
  const float* G; const float* whh[2]; float* Hout;
  float* gates; float* c_all; float* hprev;   // training-mode saves (nullptr in inference)
  const int32_t* off; unsigned* state;
  int32_t n_seq, H, gsize, n_groups, upm, n_active, hout_bytes, n_teams;
  unsigned long long* ll; int32_t ll_bytes;   // LL instances: the {h, step tag} hand-off buffer [parity 2][direction 2][video][H]
};

// The hand-off counters of a persistent launch are zeroed by a KERNEL, not hipMemsetAsync: as a graph memset node the fill can
// bypass the L2 lines the previous replay's atomics left behind, and the next launch's first poll then reads the OLD counts (seen
// as replay-to-replay drift of the scores under hipGraphLaunch; scripts/probes/graph_capture_debug.py).  A kernel's stores are
// ordered with the following kernel's loads at the launch boundary.
__global__ void zero_words_kernel(unsigned* p, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = 0u;
}

// Sticky per-device health word: every persistent kernel ORs 1 into it when a bounded hand-off wait times out, in addition to
// the per-call word in the workspace (which the next call on a shared inference workspace clears).  sumk_health_check reads
// and resets it; the Python host calls that at the synchronisation points it already has (score D2H, per-epoch loss).
__device__ unsigned g_sumk_health = 0u;

constexpr int PK_THREADS = 512;
constexpr int PK_TEAMS = 8;
constexpr unsigned PK_SPIN_LIMIT = 1u << 20;   // ~1 s of polling; after one timeout the block stops waiting altogether
// The flag-in-data hand-offs poll with their OWN data loads (a turn = four to thirty-two sc1 loads, ~1 us), so a turn count is a poor
// clock: they share the counter protocol's ~1 s budget in WALL time -- s_memrealtime, the 100 MHz constant counter -- read every 256
// turns, the first time at turn 256 (a healthy wait ends long before).  (A count of PK_SPIN_LIMIT / 16 turns was tens of ms: resume skew after a CWSR preemption, or several ranks time-slicing
// one GPU, could have tripped it on a healthy run.)
constexpr unsigned long long PK_WAIT_TICKS = 100000000ull;
__device__ __forceinline__ bool pk_ll_timed_out(unsigned& spins, unsigned long long& t0) {     // (the clock is first read at turn 256: nothing on the fast path)
  if ((++spins & 255u) != 0u) return false;
  const unsigned long long now = __builtin_amdgcn_s_memrealtime();
  if (spins == 256u) { t0 = now; return false; }
  return now - t0 > PK_WAIT_TICKS;
}

__device__ __forceinline__ float ld_sc1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// CPW = 8-wide k chunks per wave: the block's 8 waves cover K = 64 * CPW >= H (4 for DSN's H = 256, 16 for sLSTM's 1024).
// The member's W_hh fragments (32 gate rows x its wave's k range) live in REGISTERS for the whole work item -- exactly the
// MFMA B operands every step needs -- so LDS holds only the staged h_{t-1} panel and the split-K partial tiles.
// DIRECT: the A fragments (h_{t-1}[video li][k..k+3]) go straight from the sc1 buffer loads into the MFMAs, as in
// lstm_wide_kernel below -- no LDS panel, one workgroup barrier less per step.
// LL ("flag in the data", the low-latency protocol of collective libraries): h_t is published as ONE 8-byte {value, step tag} store per
// element into a double-buffered exchange array, and the consumers' OWN loads of h_{t-1} are the poll -- reloaded until every tag
// they need says t.  An aligned 8-byte store is single-copy atomic, so a matching tag proves its value; no other ordering is needed.
// Against the counter protocol above this removes, per step, the producers' vmcnt(0) drain + barrier + atomic add and the consumer's
// counter poll + barrier in front of the loads: two of the ~five dependent L2 round trips of a step (3.94 -> measured in DESIGN.md).
// Parity = step & 1: a member can publish step t + 1 only after it has read every member's step t, i.e. after all of them finished
// reading step t - 1 -- the slot it overwrites.  The output matrix Hout is written with plain stores (nobody reads it in this launch).
// M16 (LL only, groups of <= 16 videos -- DSN's 50-video batch makes 8 items of 13): the step's product runs on v_mfma_f32_16x16x4_f32
// (two 16-column tiles x 8 MFMAs of 32 cycles per wave) instead of one 32-row tile of v_mfma_f32_32x32x2_f32 (16 x 64 cycles) whose
// upper half would multiply zeros: half the matrix-pipe time of a step.  Lane group g = lane / 16 takes the CONTIGUOUS k range
// [32 wave + 8 g, + 8) of both operands (the sum over k does not care which lane group carries which k), so a lane's eight h values are
// four 16-byte loads.
template <int CPW, bool DIRECT, bool LL = false, bool M16 = false>
__global__ __launch_bounds__(PK_THREADS) void lstm_persist_kernel(PersistArgs a) {
  static_assert(!LL || DIRECT, "the flag-in-data hand-off feeds the MFMAs straight from the loads");
  static_assert(!M16 || (LL && CPW == 4), "the 16-row form exists for the flag-in-data kernel with 32 k per wave");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int H = a.H, P = H + 4;
  float* sH = smem;                       // [gsize][P]  h_{t-1} of the group's videos
  // [8 waves][32 rows][PP] split-K partial tiles; PP = 32: the epilogue's float4 read (a 16-lane group = 2 videos x 8 units x 16 B)
  // covers the 64 banks exactly once -- the pitch 33 + scalar reads of the first version were 8-way conflicted (ei + eu collides
  // along anti-diagonals); the MFMA-side scalar writes are 2-way at most, which a ds_write_b32 absorbs
  constexpr int PP = 32;
  float* part = sH + a.gsize * P;
  int* sR0 = reinterpret_cast<int*>(part + 8 * 32 * PP);   // [32] first row of each video
  int* sT = sR0 + 32;                     // [32] length of each video
  int* sTg = sT + 32;                     // [1]  longest video of the group

  const int team = blockIdx.x % a.n_teams, slot = blockIdx.x / a.n_teams;
  if (slot >= a.n_active) return;
  // buffer descriptor over the output/exchange matrix (wave-uniform: built from kernel arguments only)
  const __amdgpu_buffer_rsrc_t hrsrc = __builtin_amdgcn_make_buffer_rsrc(a.Hout, (short)0, a.hout_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t lrsrc = __builtin_amdgcn_make_buffer_rsrc(a.ll, (short)0, LL ? a.ll_bytes : 0, 0x00020000);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int u0 = slot * a.upm, nu = min(a.upm, H - u0);
  const int n_items = 2 * a.n_groups;
  int loaded_dir = -1;
  bool dead = false;   // (thread 0 only; LL: every wave) a wait timed out: results are invalid, state[0] says so

  float4 wreg[CPW];
#ifdef SUMK_DIAG   // `make DIAG=1`: per-phase shader cycles of the flag-in-data 16-row path, waves 0 and 7 of member 0 of team 0 -> state words 600..
  unsigned long long dg_wait = 0, dg_mfma = 0, dg_bar1 = 0, dg_epi = 0, dg_bar2 = 0, dg_spins = 0;
  const unsigned long long dg_t0 = __builtin_amdgcn_s_memtime();
#endif
  for (int item = team; item < n_items; item += a.n_teams) {
    const int g = item >> 1, d = item & 1;
    const int v0 = g * a.gsize, nv = min(a.gsize, a.n_seq - v0);
    unsigned* bar = a.state + 16 + item;
    __syncthreads();   // previous item fully done with LDS
    if (loaded_dir != d) {   // this lane's W_hh fragments -> registers (plain loads: weights are never written in this launch)
      // Column order of the member's 32 gate columns: n = 4 unit + gate -- the four gates of a unit side by side, so that the cell
      // update reads ONE float4 per wave partial (8 LDS reads per thread and step instead of 32).
      if constexpr (M16) {   // column n = 16 tile + lane % 16; k = 32 wave + 8 (lane / 16) + 0..7
#pragma unroll
        for (int tile = 0; tile < 2; ++tile) {
          const int n = 16 * tile + (lane & 15);
          const float* wrow = a.whh[d] + (int64_t)((n & 3) * H + min(u0 + (n >> 2), H - 1)) * H;
          const int k = wave * 32 + 8 * (lane >> 4);
          wreg[2 * tile] = k < H ? *reinterpret_cast<const float4*>(wrow + k) : make_float4(0.f, 0.f, 0.f, 0.f);
          wreg[2 * tile + 1] = k + 4 < H ? *reinterpret_cast<const float4*>(wrow + k + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      } else {
      const float* wrow = a.whh[d] + (int64_t)((li & 3) * H + min(u0 + (li >> 2), H - 1)) * H;
#pragma unroll
      for (int c = 0; c < CPW; ++c) {
        const int k = (wave * CPW + c) * 8 + 4 * lh;
        wreg[c] = k < H ? *reinterpret_cast<const float4*>(wrow + k) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      }
      loaded_dir = d;
    }
    if (tid == 0) *sTg = 0;
    __syncthreads();
    if (tid < 32) {
      int r0 = 0, T = 0;
      if (tid < nv) { r0 = a.off[v0 + tid]; T = a.off[v0 + tid + 1] - r0; atomicMax(sTg, T); }
      sR0[tid] = r0; sT[tid] = T;
    }
    __syncthreads();
    const int Tg = *sTg;

    // epilogue role: thread (video i, unit u) for tid < 256; cell state lives in a register for the whole item
    const int ei = tid >> 3, eu = tid & 7;
    const bool erole = tid < 256 && ei < nv && eu < nu;
    const int er0 = erole ? sR0[ei] : 0, eT = erole ? sT[ei] : 0;
    const int j = u0 + eu;
    float c = 0.f, hlast = 0.f;
    float gcur[4] = {0.f, 0.f, 0.f, 0.f};
    if (erole && eT > 0) {
      const int64_t row = d == 0 ? er0 : er0 + eT - 1;
      const float* gp = a.G + row * (8 * H) + d * 4 * H;
#pragma unroll
      for (int q = 0; q < 4; ++q) gcur[q] = gp[q * H + j];
    }
    const int vl = M16 ? (lane & 15) : li;     // the video (row of the group) this lane feeds to the MFMAs (DIRECT)
    const int r0l = sR0[vl], Tl = sT[vl];

    for (int t = 0; t < Tg; ++t) {
      if (t > 0) {
        if (!LL && tid == 0 && !dead) {   // wait until every member published step t-1
          const unsigned want = (unsigned)t * (unsigned)a.n_active;
          unsigned spins = 0;
          while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > PK_SPIN_LIMIT || ((spins & 1023) == 0 &&
                 __hip_atomic_load(a.state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
              atomicOr(a.state, 1u); atomicOr(&g_sumk_health, 1u); dead = true; break;   // never hang the GPU: flag the failure and stop waiting
            }
          }
        }
        if constexpr (!LL) __syncthreads();
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if constexpr (LL && M16) {
#ifdef SUMK_DIAG
          const unsigned long long st0 = __builtin_amdgcn_s_memtime();
#endif
          // h_{t-1}[video lane % 16][k .. k+7], k = 32 wave + 8 (lane / 16): four 16-byte loads of {value, tag} pairs, repeated until
          // every tag this lane needs says t
          const bool need_row = t < Tl;
          const unsigned want = (unsigned)t;
          const int k0 = wave * 32 + 8 * (lane >> 4);
          const unsigned rowb = (unsigned)((((unsigned)((t - 1) & 1) * 2u + (unsigned)d) * (unsigned)a.n_seq + (unsigned)(v0 + vl)) * (unsigned)H) * 8u;
          u32x4 va[4];
          unsigned spins = 0;
          unsigned long long ll_t0 = 0;
          while (true) {
            asm volatile("" ::: "memory");   // the loads below are a poll: they must be re-issued every turn
#pragma unroll
            for (int p = 0; p < 4; ++p) {
              const unsigned o = (need_row && k0 + 2 * p < H) ? rowb + 8u * (unsigned)(k0 + 2 * p) : 0x7ffffff0u;     // beyond num_records: zeros
              va[p] = __builtin_amdgcn_raw_buffer_load_b128(lrsrc, o, 0, 16 /* sc1 */);
            }
            // Branch-free tag checks.  With `if (needed) ok = ok && tag == want && ...` the compiler built a ladder of exec-mask branches, one per tag, around
            // counted vmcnt waits: 0.45 us of a 2.55-us step (816 -> 665 us per DSN recurrence launch, same results).
            bool ok = true;
#pragma unroll
            for (int p = 0; p < 4; ++p)
              ok = ok & (((va[p][1] == want) & (va[p][3] == want)) | !(need_row && k0 + 2 * p < H));
            if (__all(ok) || dead) break;
            if (pk_ll_timed_out(spins, ll_t0)) {     // never hang the GPU: flag the failure and stop waiting
              if (lane == 0) { atomicOr(a.state, 1u); atomicOr(&g_sumk_health, 1u); }
              dead = true;
            }
          }
#ifdef SUMK_DIAG
          const unsigned long long st1 = __builtin_amdgcn_s_memtime();
          dg_wait += st1 - st0; dg_spins += spins;
#endif
          f32x4 hv[4];
#pragma unroll
          for (int p = 0; p < 4; ++p) hv[p] = __builtin_bit_cast(f32x4, va[p]);     // {h_k, tag, h_k+1, tag}
          f32x4 acc16[2];
#pragma unroll
          for (int tile = 0; tile < 2; ++tile) {
            acc16[tile] = f32x4{0.f, 0.f, 0.f, 0.f};
            const float4 w0 = wreg[2 * tile], w1 = wreg[2 * tile + 1];
            acc16[tile] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[0][0], w0.x, acc16[tile], 0, 0, 0);
            acc16[tile] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[0][2], w0.y, acc16[tile], 0, 0, 0);
            acc16[tile] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[1][0], w0.z, acc16[tile], 0, 0, 0);
            acc16[tile] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[1][2], w0.w, acc16[tile], 0, 0, 0);
            acc16[tile] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[2][0], w1.x, acc16[tile], 0, 0, 0);
            acc16[tile] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[2][2], w1.y, acc16[tile], 0, 0, 0);
            acc16[tile] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[3][0], w1.z, acc16[tile], 0, 0, 0);
            acc16[tile] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[3][2], w1.w, acc16[tile], 0, 0, 0);
          }
          // C/D map of the 16x16 MFMA: row = 4 (lane / 16) + r, column = lane % 16; same [wave][video][33] partial tiles as the 32-row form
#pragma unroll
          for (int tile = 0; tile < 2; ++tile)
#pragma unroll
            for (int r = 0; r < 4; ++r) part[(wave * 32 + 4 * (lane >> 4) + r) * PP + 16 * tile + (lane & 15)] = acc16[tile][r];
          (void)acc;
#ifdef SUMK_DIAG
          dg_mfma += __builtin_amdgcn_s_memtime() - st1;
#endif
        } else
        if constexpr (LL) {
          // h_{t-1}[video li][k .. k+3] = two 16-byte loads of {value, tag} pairs, repeated until every tag this lane needs says t
          const bool need_row = t < Tl;
          const unsigned want = (unsigned)t;
          const unsigned rowb = (unsigned)((((unsigned)((t - 1) & 1) * 2u + (unsigned)d) * (unsigned)a.n_seq + (unsigned)(v0 + li)) * (unsigned)H) * 8u;
          u32x4 va[2 * CPW];
          unsigned spins = 0;
          unsigned long long ll_t0 = 0;
          while (true) {
            asm volatile("" ::: "memory");   // the loads below are a poll: they must be re-issued every turn (a read-only buffer load is otherwise loop invariant)
#pragma unroll
            for (int c = 0; c < CPW; ++c) {
              const int k = (wave * CPW + c) * 8 + 4 * lh;
              const unsigned o = (need_row && k < H) ? rowb + 8u * (unsigned)k : 0x7ffffff0u;     // beyond num_records: zeros
              va[2 * c] = __builtin_amdgcn_raw_buffer_load_b128(lrsrc, o, 0, 16 /* sc1 */);
              va[2 * c + 1] = __builtin_amdgcn_raw_buffer_load_b128(lrsrc, o, 16, 16 /* sc1 */);
            }
            bool ok = true;
#pragma unroll
            for (int c = 0; c < CPW; ++c) {
              const int k = (wave * CPW + c) * 8 + 4 * lh;
              ok = ok & (((va[2 * c][1] == want) & (va[2 * c][3] == want) & (va[2 * c + 1][1] == want) & (va[2 * c + 1][3] == want)) | !(need_row && k < H));      // (branch-free: see the 16-row form above)
            }
            if (__all(ok) || dead) break;
            if (pk_ll_timed_out(spins, ll_t0)) {     // never hang the GPU: flag the failure and stop waiting
              if (lane == 0) { atomicOr(a.state, 1u); atomicOr(&g_sumk_health, 1u); }
              dead = true;
            }
          }
#pragma unroll
          for (int c = 0; c < CPW; ++c) {
            const float4 bv = wreg[c];
            const f32x4 p0 = __builtin_bit_cast(f32x4, va[2 * c]), p1 = __builtin_bit_cast(f32x4, va[2 * c + 1]);   // {h_k, tag, h_k+1, tag}, {h_k+2, tag, h_k+3, tag}
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(p0[0], bv.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(p0[2], bv.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(p1[0], bv.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(p1[2], bv.w, acc, 0, 0, 0);
          }
        } else
        if constexpr (DIRECT) {
          constexpr unsigned OOB = 0x7ffffff0u;   // beyond num_records: the buffer load returns zeros
          const unsigned basel = t < Tl ? (unsigned)(((int64_t)(d == 0 ? r0l + t - 1 : r0l + Tl - t) * (2 * H) + d * H) * 4) : OOB;
          u32x4 va[CPW];
#pragma unroll
          for (int c = 0; c < CPW; ++c) {
            const int k = (wave * CPW + c) * 8 + 4 * lh;
            va[c] = __builtin_amdgcn_raw_buffer_load_b128(hrsrc, k < H ? basel + 4u * k : OOB, 0, 16 /* sc1 */);
          }
#pragma unroll
          for (int c = 0; c < CPW; ++c) {
            const float4 bv = wreg[c];
            const f32x4 av = __builtin_bit_cast(f32x4, va[c]);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], bv.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], bv.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[2], bv.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[3], bv.w, acc, 0, 0, 0);
          }
        } else {
        // h_{t-1} of every video of the group -> LDS.  sc1 (write-through / L1-bypassing) 16-B buffer loads ONLY, four in
        // flight per thread before the first LDS write.
        {
          const int H4 = H >> 2, n4 = nv * H4;
          for (int base = 0; base < n4; base += 4 * PK_THREADS) {
            u32x4 v[4];
            int dst[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
              const int idx = base + tid + p * PK_THREADS;
              dst[p] = -1; v[p] = u32x4{0u, 0u, 0u, 0u};
              if (idx < n4) {
                const int i = idx / H4, k = (idx - i * H4) * 4;
                const int r0 = sR0[i], T = sT[i];
                dst[p] = i * P + k;
                if (t < T) {
                  const unsigned boff = (unsigned)(((int64_t)(d == 0 ? r0 + t - 1 : r0 + T - t) * (2 * H) + d * H + k) * 4);
                  v[p] = __builtin_amdgcn_raw_buffer_load_b128(hrsrc, boff, 0, 16 /* sc1 */);
                }
              }
            }
#pragma unroll
            for (int p = 0; p < 4; ++p)
              if (dst[p] >= 0) *reinterpret_cast<u32x4*>(&sH[dst[p]]) = v[p];
          }
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
          const int k = (wave * CPW + c) * 8 + 4 * lh;
          float4 av = make_float4(0.f, 0.f, 0.f, 0.f);
          const float4 bv = wreg[c];
          if (k < H && li < nv) av = *reinterpret_cast<const float4*>(&sH[li * P + k]);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
        }
        }
        if constexpr (!M16) {
#pragma unroll
          for (int r = 0; r < 16; ++r) part[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * PP + li] = acc[r];
        }
#ifdef SUMK_DIAG
        const unsigned long long sb0 = __builtin_amdgcn_s_memtime();
#endif
        __syncthreads();
#ifdef SUMK_DIAG
        dg_bar1 += __builtin_amdgcn_s_memtime() - sb0;
#endif
      }
#ifdef SUMK_DIAG
      const unsigned long long se0 = __builtin_amdgcn_s_memtime();
#endif
      if (erole && t < eT) {
        const int64_t row = d == 0 ? er0 + t : er0 + eT - 1 - t;
        float pre[4] = {gcur[0], gcur[1], gcur[2], gcur[3]};
        if (t > 0) {
          float4 ps = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int w8 = 0; w8 < 8; ++w8) {     // fixed order: wave 0 .. 7
            const float4 pw = *reinterpret_cast<const float4*>(&part[(w8 * 32 + ei) * PP + 4 * eu]);
            ps.x += pw.x; ps.y += pw.y; ps.z += pw.z; ps.w += pw.w;
          }
          pre[0] += ps.x; pre[1] += ps.y; pre[2] += ps.z; pre[3] += ps.w;
        }
        const float ig = fast_sigmoid(pre[0]), fg = fast_sigmoid(pre[1]), gg = fast_tanh(pre[2]), og = fast_sigmoid(pre[3]);
        c = fg * c + ig * gg;
        const float h = og * fast_tanh(c);
        if constexpr (LL) {
          const unsigned long long pkt = ((unsigned long long)(unsigned)(t + 1) << 32) | (unsigned long long)__builtin_bit_cast(unsigned, h);
          __hip_atomic_store(a.ll + ((int64_t)((t & 1) * 2 + d) * a.n_seq + (v0 + ei)) * H + j, pkt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          a.Hout[row * (2 * H) + d * H + j] = h;
        } else {
          st_sc1(a.Hout + row * (2 * H) + d * H + j, h);
        }
        if (a.gates) {
          float* gs = a.gates + row * (8 * H) + d * 4 * H;
          gs[j] = ig; gs[H + j] = fg; gs[2 * H + j] = gg; gs[3 * H + j] = og;
          a.c_all[row * (2 * H) + d * H + j] = c;
          a.hprev[row * (2 * H) + d * H + j] = hlast;
        }
        hlast = h;
        if (t + 1 < eT) {   // next step's input-projection slice, one step ahead
          const int64_t nrow = d == 0 ? row + 1 : row - 1;
          const float* gp = a.G + nrow * (8 * H) + d * 4 * H;
#pragma unroll
          for (int q = 0; q < 4; ++q) gcur[q] = gp[q * H + j];
        }
      }
      if constexpr (LL) {
#ifdef SUMK_DIAG
        const unsigned long long se1 = __builtin_amdgcn_s_memtime();
        dg_epi += se1 - se0;
#endif
        __syncthreads();   // the split-K partial tiles in LDS are free again (the published h needs no further signal)
#ifdef SUMK_DIAG
        dg_bar2 += __builtin_amdgcn_s_memtime() - se1;
#endif
      } else {
        // publish step t: every storing wave drains its stores, then one lane signals
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
#ifdef SUMK_DIAG
  if (LL && M16 && blockIdx.x < 12 && lane == 0 && (wave == 0 || wave == 7)) {
    unsigned long long* q = reinterpret_cast<unsigned long long*>(a.state + 600) + (blockIdx.x * 2 + (wave == 7)) * 8;
    q[0] = __builtin_amdgcn_s_memtime() - dg_t0; q[1] = dg_wait; q[2] = dg_mfma; q[3] = dg_bar1; q[4] = dg_epi; q[5] = dg_bar2; q[6] = dg_spins;
  }
#endif
}

// ------------------------------------------------------------------------------------------- persistent recurrence with the input projection inside (round 6)
// Inference in the split-bf16 modes: the hoisted projection GEMM G = X W_ih^T + b (215 us in front of an 836-us recurrence on the S-TVSum
// batch) disappears INTO the recurrence.  A member's slice of it -- the group's <= 16 videos x its 32 gate columns x In = 1024 -- does not
// depend on h, and a step of the flag-in-data kernel spends two thirds of its ~4 700 cycles waiting (hand-off latency, arrival skew).
// Each wave multiplies its k range (128) of the NEXT step's x rows with its W_ih fragments (registers / LDS for the whole item: 2 column
// tiles x 4 k32 steps x NP planes of 16-byte chunks -- exactly the lane fragment of v_mfma_f32_16x16x32_bf16 in the "KB planes" format of
// the weight-plane block) while it would otherwise wait: the waves without a cell-update role right behind the step's barrier, the others
// behind their publish.  The products START the next step's accumulators; the recurrent MFMAs add on top and the existing cross-wave sum
// delivers  x W_ih^T + h W_hh^T  in one piece; the epilogue adds the two biases.  G (R x 8H fp32, 98 MB here) is never written or read.
// x comes from the fp32 feature matrix itself: a lane's k range of a k32 step is 32 contiguous bytes, the four lanes of a row cover one
// 128-byte line (the x PLANES would be a gather here: one 16-byte chunk per 128-byte line, eight frames of a video to a line -- measured
// 8.4 us per step, 786 KB per member and step through L2); it is split into the planes in registers (the roundings of split_planes).
// The partial tiles are double-buffered by step parity, so the step has ONE workgroup barrier.  Same hand-off (flag in the data), same
// 16-row MFMAs, same gate arithmetic as lstm_persist_kernel<4, true, true, true>; term order of the split products as gemm_regstage.h.
#ifndef SUMK_PROJ_KA
#define SUMK_PROJ_KA 4
#endif
constexpr int PROJ_KA = SUMK_PROJ_KA;
constexpr int PROJ_RF = 12;     // W_ih fragments a lane keeps in registers; the rest of its 8 / 16 / 24 are read from LDS every step
struct PersistProjArgs {
  const float* x; const char* wpl; const float* bias;     // x (n_rows x In, fp32); KB planes of [w_ih[0]; w_ih[1]] (8H x In); b_ih + b_hh (8H)
  const float* whh[2]; float* Hout;
  const int32_t* off; unsigned* state;
  unsigned long long* ll; int32_t ll_bytes;
  int32_t n_seq, H, In, gsize, n_groups, upm, n_active, n_teams;
  uint32_t w_rp16;                                         // bytes of one (k16 block, plane, half) sub-array of the weight planes
};

// KA: k32 steps of the projection per wave 0..3 (the waves with the cell update); waves 4..7 take KB = 8 - KA each (4 KA + 4 KB = 32 steps
// = In = 1024): the cell-update waves have the step's critical chain and only the hand-off latency behind their publish to work in, the
// others idle from the step's barrier on.  (Measured: the symmetric 4 : 4 is the fastest -- 3 : 5, 2 : 6 and 5 : 3 all lose 0.1-0.5 us per step,
// profiles/r06_dsn_fused_projection_probe.txt -- so PROJ_KA = 4; the parameter stays as the record of that sweep.)  A wave holds its first RF = 12 W_ih fragments (in (k32 step, tile, plane) order) in registers,
// the rest in LDS.
template <int NP, int KA>
__global__ __launch_bounds__(PK_THREADS) void lstm_persist_proj_kernel(PersistProjArgs a) {
  constexpr int KB = 8 - KA, KMAX = KB > KA ? KB : KA;
  constexpr int RF = PROJ_RF;       // W_ih fragments per lane in registers (4 VGPRs each)
  constexpr int FA = KA * 2 * NP;                                            // fragments per wave of role A (role B: KB * 2 * NP)
  constexpr int LA = FA > RF ? FA - RF : 0;                                  // of them in LDS (role A's come first)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int H = a.H;
  constexpr int PP = 32;
  float* part = smem;               // [step parity 2][8 waves][16 rows][PP] split-K partial tiles
  int* sR0 = reinterpret_cast<int*>(part + 2 * 8 * 16 * PP);
  int* sT = sR0 + 32;
  int* sTg = sT + 32;
  bf16x8* wlds = reinterpret_cast<bf16x8*>(sTg + 32);      // [(LA | LB) fragments][256 threads of the role] x 16 B: waves 0..3 first

  const int team = blockIdx.x % a.n_teams, slot = blockIdx.x / a.n_teams;
  if (slot >= a.n_active) return;
  const __amdgpu_buffer_rsrc_t lrsrc = __builtin_amdgcn_make_buffer_rsrc(a.ll, (short)0, a.ll_bytes, 0x00020000);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int u0 = slot * a.upm, nu = min(a.upm, H - u0);
  const int n_items = 2 * a.n_groups;
  const int vl = lane & 15, kq = lane >> 4;     // this lane's MFMA row (video of the group) / column, and its k quarter
  int loaded_dir = -1;
  bool dead = false;
  if (wave < 4) __builtin_amdgcn_s_setprio(1);  // the waves with the cell update (the step's critical chain) win the SIMD's issue arbitration

  float4 wreg[4];                   // W_hh fragments (fp32 16x16x4 form): column n = 16 tile + lane % 16; k = 32 wave + 8 kq + 0..7
  bf16x8 wih[RF];                   // W_ih fragments f = (s 2 + tile) NP + plane < RF: column n = 16 tile + lane % 16, k = 32 (ks0 + s) + 8 kq + 0..7
  const bool roleA = wave < 4;
  const int ks0 = roleA ? KA * wave : 4 * KA + KB * (wave - 4);               // the wave's first k32 step
  bf16x8* wl = wlds + (roleA ? 0 : LA * 256) + (tid & 255);                  // this thread's LDS fragments: wl[f' * 256]
  for (int item = team; item < n_items; item += a.n_teams) {
    const int g = item >> 1, d = item & 1;
    const int v0 = g * a.gsize, nv = min(a.gsize, a.n_seq - v0);
    __syncthreads();   // previous item fully done with LDS
    if (loaded_dir != d) {
#pragma unroll
      for (int tile = 0; tile < 2; ++tile) {
        const int n = 16 * tile + vl;
        const int wr = (n & 3) * H + min(u0 + (n >> 2), H - 1);
        const float* wrow = a.whh[d] + (int64_t)wr * H;
        const int k = wave * 32 + 8 * kq;
        wreg[2 * tile] = k < H ? *reinterpret_cast<const float4*>(wrow + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        wreg[2 * tile + 1] = k + 4 < H ? *reinterpret_cast<const float4*>(wrow + k + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      auto load_w = [&](auto nk_) {
        constexpr int NK = decltype(nk_)::value;
#pragma unroll
        for (int sl = 0; sl < NK; ++sl)
#pragma unroll
          for (int tile = 0; tile < 2; ++tile) {
            const int n = 16 * tile + vl;
            const char* wp = a.wpl + (size_t)(d * 4 * H + (n & 3) * H + min(u0 + (n >> 2), H - 1)) * 16;
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
              const unsigned sub = (unsigned)((((2 * (ks0 + sl) + (kq >> 1)) * NP + pl) * 2) + (kq & 1));
              const bf16x8 wv = *reinterpret_cast<const bf16x8*>(wp + (size_t)sub * a.w_rp16);
              const int f = (sl * 2 + tile) * NP + pl;
              if (f < RF) wih[f < RF ? f : 0] = wv; else wl[(f - RF) * 256] = wv;
            }
          }
      };
      if (roleA) load_w(std::integral_constant<int, KA>{}); else load_w(std::integral_constant<int, KB>{});
      loaded_dir = d;
    }
    if (tid == 0) *sTg = 0;
    __syncthreads();
    if (tid < 32) {
      int r0 = 0, T = 0;
      if (tid < nv) { r0 = a.off[v0 + tid]; T = a.off[v0 + tid + 1] - r0; atomicMax(sTg, T); }
      sR0[tid] = r0; sT[tid] = T;
    }
    __syncthreads();
    const int Tg = *sTg;

    // epilogue role: thread (video i, unit u) for tid < 256; cell state lives in a register for the whole item
    const int ei = tid >> 3, eu = tid & 7;
    const bool erole = tid < 256 && ei < nv && eu < nu;
    const int er0 = erole ? sR0[ei] : 0, eT = erole ? sT[ei] : 0;
    const int j = u0 + eu;
    float c = 0.f;
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    if (erole) {
#pragma unroll
      for (int q = 0; q < 4; ++q) bsum[q] = a.bias[d * 4 * H + q * H + j];
    }
    const int r0l = sR0[vl], Tl = sT[vl];
    // x of a step: row (r0 + t | r0 + T - 1 - t), the wave's k32 steps, 8 floats per lane and step (32 contiguous bytes)
    f32x4 xr[KMAX][2], xn[KMAX][2];     // two steps of x rows in flight / in use; the step loop alternates their roles
    auto load_x = [&](int t, f32x4 (&xr)[KMAX][2]) {
      const bool on = t < Tl;
      const float* xp = a.x + (size_t)(d == 0 ? r0l + t : r0l + Tl - 1 - t) * a.In + 32 * ks0 + 8 * kq;
#pragma unroll
      for (int sl = 0; sl < KMAX; ++sl) {
        f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
        if (on && (sl < KA || !roleA)) { v0 = *reinterpret_cast<const f32x4*>(xp + 32 * sl); v1 = *reinterpret_cast<const f32x4*>(xp + 32 * sl + 4); }
        xr[sl][0] = v0; xr[sl][1] = v1;
      }
    };
    // the slice of the input projection of one step -> accn (i + j descending, as gemm_regstage.h)
    f32x4 accn[2];
    auto project_n = [&](auto nk_, f32x4 (&xr)[KMAX][2]) {
      constexpr int NK = decltype(nk_)::value;
#pragma unroll
      for (int sl = 0; sl < NK; ++sl) {
        bf16x8 xp[3];
        if constexpr (NP == 3) split8x3(xr[sl][0], xr[sl][1], xp[0], xp[1], xp[2]);
        else { split8(xr[sl][0], xr[sl][1], xp[0], xp[1]); xp[2] = xp[1]; }
#pragma unroll
        for (int tile = 0; tile < 2; ++tile)
#pragma unroll
          for (int sum = NP - 1; sum >= 0; --sum)
#pragma unroll
            for (int i = NP - 1; i >= 0; --i) {
              const int jj = sum - i;
              if (jj < 0 || jj >= NP) continue;
              const int f = (sl * 2 + tile) * NP + jj;
              const bf16x8 wv = f < RF ? wih[f < RF ? f : 0] : wl[(f - RF) * 256];
              accn[tile] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xp[i], wv, accn[tile], 0, 0, 0);
            }
      }
    };
    auto project = [&](f32x4 (&xb)[KMAX][2]) {
      accn[0] = f32x4{0.f, 0.f, 0.f, 0.f}; accn[1] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (roleA) project_n(std::integral_constant<int, KA>{}, xb); else project_n(std::integral_constant<int, KB>{}, xb);
    };
    load_x(0, xn);
    load_x(1, xr);
    project(xn);

    // one step: xa holds the x rows of step t + 1 (projected behind this step's publish), xb receives those of step t + 2
    auto step = [&](const int t, f32x4 (&xa)[KMAX][2], f32x4 (&xb)[KMAX][2]) {
      const bool need_row = t < Tl;
      const unsigned want = (unsigned)t;
      const int k0 = wave * 32 + 8 * kq;
      const unsigned rowb = (unsigned)((((unsigned)((t - 1) & 1) * 2u + (unsigned)d) * (unsigned)a.n_seq + (unsigned)(v0 + vl)) * (unsigned)H) * 8u;
      f32x4 acc16[2] = {accn[0], accn[1]};
      if (t > 0) {
        u32x4 va[4];
        unsigned spins = 0;
        unsigned long long ll_t0 = 0;
        while (true) {
          asm volatile("" ::: "memory");   // the loads below are a poll: they must be re-issued every turn
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            const unsigned o = (need_row && k0 + 2 * p < H) ? rowb + 8u * (unsigned)(k0 + 2 * p) : 0x7ffffff0u;     // beyond num_records: zeros
            va[p] = __builtin_amdgcn_raw_buffer_load_b128(lrsrc, o, 0, 16 /* sc1 */);
          }
          bool ok = true;
#pragma unroll
          for (int p = 0; p < 4; ++p)
            ok = ok & (((va[p][1] == want) & (va[p][3] == want)) | !(need_row && k0 + 2 * p < H));      // (branch-free tag checks: lstm_persist_kernel)
          if (__all(ok) || dead) break;
          if (pk_ll_timed_out(spins, ll_t0)) {     // never hang the GPU: flag the failure and stop waiting
            if (lane == 0) { atomicOr(a.state, 1u); atomicOr(&g_sumk_health, 1u); }
            dead = true;
          }
        }
        f32x4 hv[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) hv[p] = __builtin_bit_cast(f32x4, va[p]);     // {h_k, tag, h_k+1, tag}
#pragma unroll
        for (int tile = 0; tile < 2; ++tile) {
          const float4 w0 = wreg[2 * tile], w1 = wreg[2 * tile + 1];
          acc16[tile] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[0][0], w0.x, acc16[tile], 0, 0, 0);
          acc16[tile] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[0][2], w0.y, acc16[tile], 0, 0, 0);
          acc16[tile] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[1][0], w0.z, acc16[tile], 0, 0, 0);
          acc16[tile] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[1][2], w0.w, acc16[tile], 0, 0, 0);
          acc16[tile] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[2][0], w1.x, acc16[tile], 0, 0, 0);
          acc16[tile] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[2][2], w1.y, acc16[tile], 0, 0, 0);
          acc16[tile] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[3][0], w1.z, acc16[tile], 0, 0, 0);
          acc16[tile] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[3][2], w1.w, acc16[tile], 0, 0, 0);
        }
      }
      if (t + 2 < Tg) load_x(t + 2, xb);     // the x rows of the step after next: a whole step in flight (8 waves x 8 KB per step queue for ~1 000 cycles in the CU's load path)
      // C/D map of the 16x16 MFMAs: row = 4 (lane / 16) + r, column = lane % 16
      float* pt = part + (t & 1) * (8 * 16 * PP);
#pragma unroll
      for (int tile = 0; tile < 2; ++tile)
#pragma unroll
        for (int r = 0; r < 4; ++r) pt[(wave * 16 + 4 * kq + r) * PP + 16 * tile + vl] = acc16[tile][r];
      __syncthreads();
      if (erole && t < eT) {
        const int64_t row = d == 0 ? er0 + t : er0 + eT - 1 - t;
        float4 ps = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) {     // fixed order: wave 0 .. 7
          const float4 pw = *reinterpret_cast<const float4*>(&pt[(w8 * 16 + ei) * PP + 4 * eu]);
          ps.x += pw.x; ps.y += pw.y; ps.z += pw.z; ps.w += pw.w;
        }
        const float ig = fast_sigmoid(ps.x + bsum[0]), fg = fast_sigmoid(ps.y + bsum[1]), gg = fast_tanh(ps.z + bsum[2]), og = fast_sigmoid(ps.w + bsum[3]);
        c = fg * c + ig * gg;
        const float h = og * fast_tanh(c);
        const unsigned long long pkt = ((unsigned long long)(unsigned)(t + 1) << 32) | (unsigned long long)__builtin_bit_cast(unsigned, h);
        __hip_atomic_store(a.ll + ((int64_t)((t & 1) * 2 + d) * a.n_seq + (v0 + ei)) * H + j, pkt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.Hout[row * (2 * H) + d * H + j] = h;
      }
      // (no second barrier: the next step writes the OTHER parity of the partial tiles, and the step after that comes behind the next barrier)
      if (t + 1 < Tg) project(xa);        // the next step's projection: under the hand-off latency of the publish above
    };
    for (int t = 0; t < Tg; t += 2) {
      step(t, xr, xn);
      if (t + 1 < Tg) step(t + 1, xn, xr);
    }
  }
}

// ------------------------------------------------------------------------------------------- wide persistent recurrence
// The same one-launch recurrence for 256 < H <= 1024 (sLSTM: H = 1024, W_hh = 16 MB per direction -- the launch chain
// re-reads it from the Infinity Cache every step, 28 us per step).  Here W_hh never moves: the 256 CUs split into TWO teams
// of 128 (team = direction; XCDs 0-3 / 4-7 under the observed round-robin dispatch -- speed only, as above), member m keeps
// the 32 gate rows of its 8 hidden units as MFMA B fragments in registers (64 VGPRs per lane: 32 rows x the wave's 128 k),
// and a work item is a group of up to 64 videos = two 32-row MFMA tiles.  A 64 x 1024 fp32 panel of h_{t-1} (256 KB) does
// not fit in LDS, so the A fragments are loaded straight into registers (sc1 16-B buffer loads, 32 in flight per lane; a
// lane pair covers 32 contiguous bytes of one video's row, L2-served after the first CU of the XCD has pulled the line).
// Hand-off protocol, bounded spins and error word are exactly those of lstm_persist_kernel (Guideline 16 R1).
// Per step and CU: 2 x 16 x 4 fp32 MFMAs per wave = 6.8 us of MFMA time at 2 waves per SIMD -- the fp32 floor of this
// shape -- plus the panel load and one hand-off.
struct WideArgs {
  const float* G; const float* whh[2]; float* Hout;
  float* gates; float* c_all; float* hprev;   // training-mode saves (nullptr in inference)
  const int32_t* off; unsigned* state;
  int32_t n_seq, H, n_groups, upm, n_active, hout_bytes;
};
constexpr int WK_GROUP = 64;    // videos per work item
constexpr int WK_CPW = 16;      // 8-wide k chunks per wave: 8 waves x 16 x 8 = 1024 >= H


// X3: the recurrent product in bf16x3 arithmetic (3 v_mfma_f32_32x32x16_bf16 per 16 k instead of 8 fp32 MFMAs): W_hh is
// split once into hi/lo bf16 fragments (the same 64 VGPRs), h_{t-1} is split in registers after the load.
template <bool X3>
__global__ __launch_bounds__(PK_THREADS) void lstm_wide_kernel(WideArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int H = a.H;
  // [8 waves][64 videos][PP] split-K partial tiles; gate columns ordered n = 4 unit + gate and PP = 32, so that the cell update reads
  // ONE conflict-free float4 per wave partial (as in lstm_persist_kernel: the pitch-33 scalar reads of the first version were 8-way)
  constexpr int PP = 32;
  float* part = smem;
  int* sR0 = reinterpret_cast<int*>(part + 8 * 64 * PP);  // [64] first row of each video
  int* sT = sR0 + 64;                                     // [64] length of each video (0 past the group)
  int* sTg = sT + 64;                                     // [1]  longest video of the group

  const int d = (blockIdx.x & 7) >> 2;                    // team = direction
  const int slot = (blockIdx.x >> 3) * 4 + (blockIdx.x & 3);
  if (slot >= a.n_active) return;
  const __amdgpu_buffer_rsrc_t hrsrc = __builtin_amdgcn_make_buffer_rsrc(a.Hout, (short)0, a.hout_bytes, 0x00020000);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int u0 = slot * a.upm, nu = min(a.upm, H - u0);
  bool dead = false;   // (thread 0 only) a wait timed out: results are invalid, state[0] says so

  // this lane's W_hh fragments -> registers for the whole launch (plain loads: weights are never written here)
  // fp32: chunk c of 8 k, this lane holds k = 8c + 4 lh .. +3.  X3: k16-step s, this lane holds k = 16s + 8 lh .. +7.
  constexpr int NS = WK_CPW / 2;    // k16 steps per wave
  float4 wreg[X3 ? 1 : WK_CPW];
  bf16x8 whi[X3 ? NS : 1], wlo[X3 ? NS : 1];
  {
    const float* wrow = a.whh[d] + (int64_t)((li & 3) * H + min(u0 + (li >> 2), H - 1)) * H;     // column li = 4 unit + gate
    if constexpr (X3) {
#pragma unroll
      for (int sidx = 0; sidx < NS; ++sidx) {
        const int k = (wave * NS + sidx) * 16 + 8 * lh;
        const float4 w0 = k < H ? *reinterpret_cast<const float4*>(wrow + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 w1 = k + 4 < H ? *reinterpret_cast<const float4*>(wrow + k + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        split8(f32x4{w0.x, w0.y, w0.z, w0.w}, f32x4{w1.x, w1.y, w1.z, w1.w}, whi[sidx], wlo[sidx]);
      }
    } else {
#pragma unroll
      for (int c = 0; c < WK_CPW; ++c) {
        const int k = (wave * WK_CPW + c) * 8 + 4 * lh;
        wreg[c] = k < H ? *reinterpret_cast<const float4*>(wrow + k) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }
  constexpr unsigned OOB = 0x7ffffff0u;   // beyond num_records: the buffer load returns zeros

  for (int g = 0; g < a.n_groups; ++g) {
    const int item = 2 * g + d;
    const int v0 = g * WK_GROUP, nv = min(WK_GROUP, a.n_seq - v0);
    unsigned* bar = a.state + 16 + item;
    __syncthreads();   // previous item fully done with LDS
    if (tid == 0) *sTg = 0;
    __syncthreads();
    if (tid < 64) {
      int r0 = 0, T = 0;
      if (tid < nv) { r0 = a.off[v0 + tid]; T = a.off[v0 + tid + 1] - r0; atomicMax(sTg, T); }
      sR0[tid] = r0; sT[tid] = T;
    }
    __syncthreads();
    const int Tg = *sTg;
    const int r0a = sR0[li], Ta = sT[li], r0b = sR0[32 + li], Tb = sT[32 + li];   // the two videos this lane feeds to the MFMAs

    // epilogue role: thread (video ei, unit eu); cell state and last h live in registers for the whole item
    const int ei = tid >> 3, eu = tid & 7;
    const bool erole = ei < nv && eu < nu;
    const int er0 = erole ? sR0[ei] : 0, eT = erole ? sT[ei] : 0;
    const int j = u0 + eu;
    float c = 0.f, hlast = 0.f;
    float gcur[4] = {0.f, 0.f, 0.f, 0.f};
    if (erole && eT > 0) {
      const int64_t row = d == 0 ? er0 : er0 + eT - 1;
      const float* gp = a.G + row * (8 * H) + d * 4 * H;
#pragma unroll
      for (int q = 0; q < 4; ++q) gcur[q] = gp[q * H + j];
    }

    for (int t = 0; t < Tg; ++t) {
      if (t > 0) {
        if (tid == 0 && !dead) {   // wait until every member published step t-1
          const unsigned want = (unsigned)t * (unsigned)a.n_active;
          unsigned spins = 0;
          while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > PK_SPIN_LIMIT || ((spins & 1023) == 0 &&
                 __hip_atomic_load(a.state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
              atomicOr(a.state, 1u); atomicOr(&g_sumk_health, 1u); dead = true; break;   // never hang the GPU: flag the failure and stop waiting
            }
          }
        }
        __syncthreads();
        // A fragments: h_{t-1}[video][k..k+3], sc1 (L1-bypassing) 16-B buffer loads ONLY; finished videos read zeros
        const unsigned basea = t < Ta ? (unsigned)(((int64_t)(d == 0 ? r0a + t - 1 : r0a + Ta - t) * (2 * H) + d * H) * 4) : OOB;
        const unsigned baseb = t < Tb ? (unsigned)(((int64_t)(d == 0 ? r0b + t - 1 : r0b + Tb - t) * (2 * H) + d * H) * 4) : OOB;
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
        u32x4 va[WK_CPW], vb[WK_CPW];
        if constexpr (X3) {
#pragma unroll
          for (int cc = 0; cc < WK_CPW; ++cc) {     // [2 sidx + half]: 32 contiguous bytes per lane, 64 per lane pair
            const int k = (wave * NS + (cc >> 1)) * 16 + 8 * lh + 4 * (cc & 1);
            va[cc] = __builtin_amdgcn_raw_buffer_load_b128(hrsrc, k < H ? basea + 4u * k : OOB, 0, 16 /* sc1 */);
          }
#pragma unroll
          for (int cc = 0; cc < WK_CPW; ++cc) {
            const int k = (wave * NS + (cc >> 1)) * 16 + 8 * lh + 4 * (cc & 1);
            vb[cc] = __builtin_amdgcn_raw_buffer_load_b128(hrsrc, k < H ? baseb + 4u * k : OOB, 0, 16 /* sc1 */);
          }
#pragma unroll
          for (int sidx = 0; sidx < NS; ++sidx) {
            bf16x8 ah, al;
            split8(__builtin_bit_cast(f32x4, va[2 * sidx]), __builtin_bit_cast(f32x4, va[2 * sidx + 1]), ah, al);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, whi[sidx], acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wlo[sidx], acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, whi[sidx], acc0, 0, 0, 0);
          }
          if (nv > 32)      // second MFMA tile only when the group has more than 32 videos (uniform)
#pragma unroll
          for (int sidx = 0; sidx < NS; ++sidx) {
            bf16x8 bh, bl;
            split8(__builtin_bit_cast(f32x4, vb[2 * sidx]), __builtin_bit_cast(f32x4, vb[2 * sidx + 1]), bh, bl);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, whi[sidx], acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, wlo[sidx], acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, whi[sidx], acc1, 0, 0, 0);
          }
        } else {
#pragma unroll
          for (int cc = 0; cc < WK_CPW; ++cc) {
            const int k = (wave * WK_CPW + cc) * 8 + 4 * lh;
            va[cc] = __builtin_amdgcn_raw_buffer_load_b128(hrsrc, k < H ? basea + 4u * k : OOB, 0, 16 /* sc1 */);
          }
#pragma unroll
          for (int cc = 0; cc < WK_CPW; ++cc) {
            const int k = (wave * WK_CPW + cc) * 8 + 4 * lh;
            vb[cc] = __builtin_amdgcn_raw_buffer_load_b128(hrsrc, k < H ? baseb + 4u * k : OOB, 0, 16 /* sc1 */);
          }
          // (the whole vector is bit-cast before its elements are taken: bit-casting va[cc][j] element-wise made hipcc narrow
          //  the load to one dword and feed element 0 to all four MFMAs)
#pragma unroll
          for (int cc = 0; cc < WK_CPW; ++cc) {
            const float4 bv = wreg[cc];
            const f32x4 av = __builtin_bit_cast(f32x4, va[cc]);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], bv.x, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], bv.y, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[2], bv.z, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[3], bv.w, acc0, 0, 0, 0);
          }
          if (nv > 32)      // second MFMA tile only when the group has more than 32 videos (uniform)
#pragma unroll
          for (int cc = 0; cc < WK_CPW; ++cc) {
            const float4 bv = wreg[cc];
            const f32x4 av = __builtin_bit_cast(f32x4, vb[cc]);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], bv.x, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], bv.y, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[2], bv.z, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[3], bv.w, acc1, 0, 0, 0);
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
          part[(wave * 64 + row) * PP + li] = acc0[r];
          part[(wave * 64 + 32 + row) * PP + li] = acc1[r];
        }
        __syncthreads();
      }
      if (erole && t < eT) {
        const int64_t row = d == 0 ? er0 + t : er0 + eT - 1 - t;
        float pre[4] = {gcur[0], gcur[1], gcur[2], gcur[3]};
        if (t > 0) {
          float4 ps = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int w8 = 0; w8 < 8; ++w8) {     // fixed order: wave 0 .. 7
            const float4 pw = *reinterpret_cast<const float4*>(&part[(w8 * 64 + ei) * PP + 4 * eu]);
            ps.x += pw.x; ps.y += pw.y; ps.z += pw.z; ps.w += pw.w;
          }
          pre[0] += ps.x; pre[1] += ps.y; pre[2] += ps.z; pre[3] += ps.w;
        }
        const float ig = fast_sigmoid(pre[0]), fg = fast_sigmoid(pre[1]), gg = fast_tanh(pre[2]), og = fast_sigmoid(pre[3]);
        c = fg * c + ig * gg;
        const float h = og * fast_tanh(c);
        st_sc1(a.Hout + row * (2 * H) + d * H + j, h);
        if (a.gates) {
          float* gs = a.gates + row * (8 * H) + d * 4 * H;
          gs[j] = ig; gs[H + j] = fg; gs[2 * H + j] = gg; gs[3 * H + j] = og;
          a.c_all[row * (2 * H) + d * H + j] = c;
          a.hprev[row * (2 * H) + d * H + j] = hlast;
        }
        hlast = h;
        if (t + 1 < eT) {   // next step's input-projection slice, one step ahead
          const int64_t nrow = d == 0 ? row + 1 : row - 1;
          const float* gp = a.G + nrow * (8 * H) + d * 4 * H;
#pragma unroll
          for (int q = 0; q < 4; ++q) gcur[q] = gp[q * H + j];
        }
      }
      // publish step t: every storing wave drains its stores, then one lane signals
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ------------------------------------------------------------------------------------------- wide persistent recurrence, round 6
// The same team structure as lstm_wide_kernel (two teams of 128, W_hh slices in registers, counter hand-off R1), re-partitioned around
// what that kernel's step consisted of ("Round 6: the wide recurrence"):
//  * h_{t-1} travels through an EXCHANGE buffer laid out for the consumers, [step parity][direction][k / 4][video row 64][4 floats],
//    not through the output matrix: one A-fragment load instruction of a wave is two contiguous 512-byte runs (the old form touched 32
//    rows x 32 bytes -- 32 half-used 64-byte requests per instruction, twice the bytes through L2), and a member's publish is two
//    contiguous 1-KB runs of 16-byte sc1 stores (was 512 dword stores, one fabric write each).  Parity by step: a member can publish
//    step t + 1 only after every member published step t, i.e. after all of them finished reading step t - 1 -- the slot it overwrites.
//    Each group of 64 videos has its own exchange region (a short group could otherwise be overrun by the next one's step 0).
//  * the group's videos are SORTED by length (descending, ties by index) onto the 64 MFMA rows, so the second 32-row tile -- its
//    loads, its MFMAs, its partial tiles -- stops as soon as fewer than 33 videos are still running (t >= the 33rd longest length).
//    A row's result does not depend on which row it is (one fixed k order per output element): batch composition still cannot
//    change a video's scores.
//  * the step counter is SHARDED four ways (member % 4 = its XCD under round-robin dispatch; each shard on a line of its own) and
//    polled by four lanes of one wave: 128 atomics on one word serialise at ~12 ns each (MI355X_MICROARCH.md, fanin).
//  * the next step's input-projection slice is requested at the TOP of a step, and the output / training-save stores are issued
//    AFTER the signal: the producer's vmcnt(0) drain in front of the signal then waits for the exchange stores only (it used to
//    wait for four strided G loads issued just before it, and for up to eight plain stores).
//  * MODE 2 ("bf16x6", fp32-grade): x = x1 + x2 + x3 exactly, the six products with i + j <= 4 on v_mfma_f32_32x32x16_bf16 in the
//    term order of gemm_regstage.h (smallest first): 96 matrix instructions of 32 cycles per wave and step instead of 128 of 64.
//    W_hh planes 1, 2 in registers (64 VGPRs), plane 3 as fragments in LDS (64 KB, one ds_read_b128 per k16 step); h is split in
//    registers after the load; the second tile's loads re-use the first tile's registers as they are consumed.
struct Wide2Args {
  const float* G; const float* whh[2]; float* Hout;
  float* gates; float* c_all; float* hprev;   // training-mode saves (nullptr in inference)
  const int32_t* off; unsigned* state; float* xchg;
  int32_t n_seq, H, n_groups, upm, n_active, pack16;
};
constexpr int W2_SLICE_FLOATS = 256 * 64 * 4;   // one (parity, direction) slice of the exchange: [k4 256][row 64][4 floats] = 256 KB
constexpr int W2_ITEM_WORDS = 128;              // state words per work item: four counter shards, 32 words (128 B) apart
[[maybe_unused]] constexpr int W2_STAMP_WORD = PSTATE_WORDS - 512;   // diagnostic build: phase stamps at the end of the state block


template <int MODE>   // 0: exact fp32 MFMA, 1: bf16x3 (hi / lo), 2: bf16x6 (three planes)
__global__ __launch_bounds__(PK_THREADS) void lstm_wide2_kernel(Wide2Args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int H = a.H;
  constexpr int PP = 32;
  float* part = smem;                                     // [8 waves][64 rows][PP] split-K partial tiles (64 KB)
  int* sR0 = reinterpret_cast<int*>(part + 8 * 64 * PP);  // [64] first frame row of the video on MFMA row r (sorted by length)
  int* sT = sR0 + 64;                                     // [64] its length (0 past the group)
  int* sR0u = sT + 64;                                    // [64] the same two in batch order (input of the sort)
  int* sTu = sR0u + 64;
  bf16x8* w3s = reinterpret_cast<bf16x8*>(sTu + 64);      // MODE 2: [8 k16 steps][512 threads] third W_hh plane (64 KB)

  const int d = (blockIdx.x & 7) >> 2;                    // team = direction
  const int slot = (blockIdx.x >> 3) * 4 + (blockIdx.x & 3);
  if (slot >= a.n_active) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (uniform, and said so: as a VGPR value it put each of a step's 32 exchange loads -- their scalar offset is wave * ... -- into a waterfall loop)
  const int li = lane & 31, lh = lane >> 5;
  const int u0 = slot * a.upm, nu = min(a.upm, H - u0);
  bool dead = false;   // (wave 0) a wait timed out: results are invalid, state[0] says so

  constexpr int NS = WK_CPW / 2;    // k16 steps per wave
  float4 wreg[MODE == 0 ? WK_CPW : 1];
  bf16x8 wA[MODE != 0 ? NS : 1], wB[MODE != 0 ? NS : 1];
  {
    const float* wrow = a.whh[d] + (int64_t)((li & 3) * H + min(u0 + (li >> 2), H - 1)) * H;     // column li = 4 unit + gate
    if constexpr (MODE == 0) {
#pragma unroll
      for (int c = 0; c < WK_CPW; ++c) {
        const int k = (wave * WK_CPW + c) * 8 + 4 * lh;
        wreg[c] = k < H ? *reinterpret_cast<const float4*>(wrow + k) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else {
#pragma unroll
      for (int sidx = 0; sidx < NS; ++sidx) {
        const int k = (wave * NS + sidx) * 16 + 8 * lh;
        const float4 w0 = k < H ? *reinterpret_cast<const float4*>(wrow + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 w1 = k + 4 < H ? *reinterpret_cast<const float4*>(wrow + k + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (MODE == 1) {
          split8(f32x4{w0.x, w0.y, w0.z, w0.w}, f32x4{w1.x, w1.y, w1.z, w1.w}, wA[sidx], wB[sidx]);
        } else {
          bf16x8 p3;
          split8x3(f32x4{w0.x, w0.y, w0.z, w0.w}, f32x4{w1.x, w1.y, w1.z, w1.w}, wA[sidx], wB[sidx], p3);
          w3s[sidx * PK_THREADS + tid] = p3;
        }
      }
    }
  }
  constexpr unsigned OOB = 0x7ffffff0u;   // beyond num_records: the buffer load returns zeros
#ifdef SUMK_DIAG
  unsigned long long dg_poll = 0, dg_mfma = 0, dg_bar1 = 0, dg_epi = 0, dg_pub = 0, dg_steps = 0, dg_two = 0;
  const unsigned long long dg_t0 = __builtin_amdgcn_s_memtime();
#endif

  for (int g = 0; g < a.n_groups; ++g) {
    const int item = 2 * g + d;
    const int v0 = g * WK_GROUP, nv = min(WK_GROUP, a.n_seq - v0);
    unsigned* bar = a.state + 16 + item * W2_ITEM_WORDS;
    float* xg = a.xchg + (size_t)g * 4 * W2_SLICE_FLOATS;
    // one descriptor per step parity over THIS direction's slice, ending behind its last written k4 row (k < H): rows of k >= H -- never
    // written -- and the OOB offsets of finished videos read as zeros, so no load needs a compare
    const __amdgpu_buffer_rsrc_t xr0 = __builtin_amdgcn_make_buffer_rsrc(xg + (0 * 2 + d) * W2_SLICE_FLOATS, (short)0, (H >> 2) * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t xr1 = __builtin_amdgcn_make_buffer_rsrc(xg + (1 * 2 + d) * W2_SLICE_FLOATS, (short)0, (H >> 2) * 1024, 0x00020000);
    __syncthreads();   // previous item fully done with LDS
    if (tid < 64) {
      int r0 = 0, T = 0;
      if (tid < nv) { r0 = a.off[v0 + tid]; T = a.off[v0 + tid + 1] - r0; }
      sR0u[tid] = r0; sTu[tid] = (T << 6) | (63 - tid);
    }
    __syncthreads();
    if (tid < 64) {     // MFMA row = rank by length, descending; ties by batch index
      const int key = sTu[tid];     // (T << 6) | (63 - batch index): distinct keys
      int rank = 0;
#pragma unroll 8
      for (int q = 0; q < 64; ++q) rank += sTu[q] > key ? 1 : 0;
      sT[rank] = key >> 6; sR0[rank] = sR0u[tid];
    }
    __syncthreads();
    const int Tg = sT[0], T32 = sT[32];       // the second tile runs while t < T32
    const int Ta = sT[li], Tb = sT[32 + li];  // the two rows this lane feeds to the MFMAs

    // epilogue role: thread (row ei, unit eu); cell state and last h live in registers for the whole item
    const int ei = tid >> 3, eu = tid & 7;
    const bool erole = ei < nv && eu < nu;
    const int er0 = sR0[ei], eT = erole ? sT[ei] : 0;
    const int j = u0 + eu;
    float c = 0.f, hlast = 0.f;
    float gcur[4] = {0.f, 0.f, 0.f, 0.f};
    if (eT > 0) {
      const int64_t row = d == 0 ? er0 : er0 + eT - 1;
      const float* gp = a.G + row * (8 * H) + d * 4 * H;
#pragma unroll
      for (int q = 0; q < 4; ++q) gcur[q] = gp[q * H + j];
    }
    const unsigned xrow_w = (unsigned)(((j >> 2) * 64 + ei) * 4 + (j & 3));   // this thread's float inside a slice

    for (int t = 0; t < Tg; ++t) {
#ifdef SUMK_DIAG
      const unsigned long long s0 = __builtin_amdgcn_s_memtime();
#endif
      float gnext[4] = {0.f, 0.f, 0.f, 0.f};
      if (t + 1 < eT) {   // next step's input-projection slice: requested a whole step before it is used
        const int64_t nrow = d == 0 ? er0 + t + 1 : er0 + eT - 2 - t;
        const float* gp = a.G + nrow * (8 * H) + d * 4 * H;
#pragma unroll
        for (int q = 0; q < 4; ++q) gnext[q] = gp[q * H + j];
      }
      const bool two = t < T32;
      if (t > 0) {
        if (wave == 0 && !dead) {   // wait until every member published step t-1: lanes 0..3 poll the four shards
          const unsigned cnt = (lane < 4 && lane < a.n_active) ? (unsigned)((a.n_active - lane + 3) >> 2) : 0u;
          const unsigned want = (unsigned)t * cnt;
          unsigned spins = 0;
          while (true) {
            const unsigned v = lane < 4 ? __hip_atomic_load(bar + 32 * lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffffffu;
            if (__all(v >= want)) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > PK_SPIN_LIMIT || ((spins & 1023) == 0 &&
                 __hip_atomic_load(a.state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
              if (lane == 0) { atomicOr(a.state, 1u); atomicOr(&g_sumk_health, 1u); }
              dead = true; break;   // never hang the GPU: flag the failure and stop waiting
            }
          }
        }
        __syncthreads();
#ifdef SUMK_DIAG
        const unsigned long long s1 = __builtin_amdgcn_s_memtime();
        dg_poll += s1 - s0;
#endif
        // A fragments: h_{t-1}[row][k..], sc1 (L1-bypassing) 16-B buffer loads of the exchange slice ONLY; finished rows read zeros.
        // Sixteen loads (the first tile) are in flight; the second tile's loads take each register pair as it is consumed.
        const __amdgpu_buffer_rsrc_t xr = ((t - 1) & 1) ? xr1 : xr0;
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
        u32x4 va[WK_CPW];
        if constexpr (MODE == 0) {
          // chunk c8 = 16 wave + cc: k = 8 c8 + 4 lh .. + 3  ->  k4 row 2 c8 + lh (a 512-byte run per lane half)
          const unsigned voa = (t < Ta ? (unsigned)li * 16u : OOB) + (unsigned)lh * 1024u;
          const unsigned vob = (t < Tb ? (unsigned)(32 + li) * 16u : OOB) + (unsigned)lh * 1024u;
#pragma unroll
          for (int cc = 0; cc < WK_CPW; ++cc) va[cc] = __builtin_amdgcn_raw_buffer_load_b128(xr, voa, (wave * WK_CPW + cc) * 2048, 16 /* sc1 */);
          // (the whole vector is bit-cast before its elements are taken: see lstm_wide_kernel)
#pragma unroll
          for (int cc = 0; cc < WK_CPW; ++cc) {
            const float4 bv = wreg[cc];
            const f32x4 av = __builtin_bit_cast(f32x4, va[cc]);
            if (two) va[cc] = __builtin_amdgcn_raw_buffer_load_b128(xr, vob, (wave * WK_CPW + cc) * 2048, 16 /* sc1 */);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], bv.x, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], bv.y, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[2], bv.z, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[3], bv.w, acc0, 0, 0, 0);
          }
          if (two)
#pragma unroll
          for (int cc = 0; cc < WK_CPW; ++cc) {
            const float4 bv = wreg[cc];
            const f32x4 av = __builtin_bit_cast(f32x4, va[cc]);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], bv.x, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], bv.y, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[2], bv.z, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[3], bv.w, acc1, 0, 0, 0);
          }
        } else {
          // k16 step s of this wave: k = 16 (8 wave + s) + 8 lh .. + 7  ->  k4 rows 4 (8 wave + s) + 2 lh, + 1: two loads, each a 512-byte run per lane half
          const unsigned voa = (t < Ta ? (unsigned)li * 16u : OOB) + (unsigned)lh * 2048u;
          const unsigned vob = (t < Tb ? (unsigned)(32 + li) * 16u : OOB) + (unsigned)lh * 2048u;
#pragma unroll
          for (int cc = 0; cc < WK_CPW; ++cc) va[cc] = __builtin_amdgcn_raw_buffer_load_b128(xr, voa, ((wave * NS + (cc >> 1)) * 4 + (cc & 1)) * 1024, 16 /* sc1 */);
#pragma unroll
          for (int sidx = 0; sidx < NS; ++sidx) {
            const f32x4 x0 = __builtin_bit_cast(f32x4, va[2 * sidx]), x1 = __builtin_bit_cast(f32x4, va[2 * sidx + 1]);
            if (two) {
              va[2 * sidx] = __builtin_amdgcn_raw_buffer_load_b128(xr, vob, ((wave * NS + sidx) * 4) * 1024, 16 /* sc1 */);
              va[2 * sidx + 1] = __builtin_amdgcn_raw_buffer_load_b128(xr, vob, ((wave * NS + sidx) * 4 + 1) * 1024, 16 /* sc1 */);
            }
            if constexpr (MODE == 1) {
              bf16x8 ah, al;
              split8(x0, x1, ah, al);
              acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, wA[sidx], acc0, 0, 0, 0);
              acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wB[sidx], acc0, 0, 0, 0);
              acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wA[sidx], acc0, 0, 0, 0);
            } else {
              bf16x8 h1, h2, h3;
              split8x3(x0, x1, h1, h2, h3);
              const bf16x8 w3 = w3s[sidx * PK_THREADS + tid];
              acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h3, wA[sidx], acc0, 0, 0, 0);   // (i, j) = (2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0)
              acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h2, wB[sidx], acc0, 0, 0, 0);
              acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h1, w3, acc0, 0, 0, 0);
              acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h2, wA[sidx], acc0, 0, 0, 0);
              acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h1, wB[sidx], acc0, 0, 0, 0);
              acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h1, wA[sidx], acc0, 0, 0, 0);
            }
          }
          if (two)
#pragma unroll
          for (int sidx = 0; sidx < NS; ++sidx) {
            const f32x4 x0 = __builtin_bit_cast(f32x4, va[2 * sidx]), x1 = __builtin_bit_cast(f32x4, va[2 * sidx + 1]);
            if constexpr (MODE == 1) {
              bf16x8 bh, bl;
              split8(x0, x1, bh, bl);
              acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, wA[sidx], acc1, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, wB[sidx], acc1, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, wA[sidx], acc1, 0, 0, 0);
            } else {
              bf16x8 h1, h2, h3;
              split8x3(x0, x1, h1, h2, h3);
              const bf16x8 w3 = w3s[sidx * PK_THREADS + tid];
              acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h3, wA[sidx], acc1, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h2, wB[sidx], acc1, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h1, w3, acc1, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h2, wA[sidx], acc1, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h1, wB[sidx], acc1, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(h1, wA[sidx], acc1, 0, 0, 0);
            }
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) part[(wave * 64 + (r & 3) + 8 * (r >> 2) + 4 * lh) * PP + li] = acc0[r];
        if (two)
#pragma unroll
          for (int r = 0; r < 16; ++r) part[(wave * 64 + 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * PP + li] = acc1[r];
#ifdef SUMK_DIAG
        const unsigned long long s2 = __builtin_amdgcn_s_memtime();
        dg_mfma += s2 - s1;
#endif
        __syncthreads();
#ifdef SUMK_DIAG
        dg_bar1 += __builtin_amdgcn_s_memtime() - s2; dg_steps += 1; dg_two += two ? 1 : 0;
#endif
      }
#ifdef SUMK_DIAG
      const unsigned long long s3 = __builtin_amdgcn_s_memtime();
#endif
      const bool live = t < eT;     // (eT = 0 for threads without a role; rows >= 32 are never live when the second tile was skipped)
      const int64_t row = d == 0 ? er0 + t : er0 + eT - 1 - t;
      float ig = 0.f, fg = 0.f, gg = 0.f, og = 0.f, h = 0.f;
      if (live) {
        float pre[4] = {gcur[0], gcur[1], gcur[2], gcur[3]};
        if (t > 0) {
          float4 ps = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int w8 = 0; w8 < 8; ++w8) {     // fixed order: wave 0 .. 7
            const float4 pw = *reinterpret_cast<const float4*>(&part[(w8 * 64 + ei) * PP + 4 * eu]);
            ps.x += pw.x; ps.y += pw.y; ps.z += pw.z; ps.w += pw.w;
          }
          pre[0] += ps.x; pre[1] += ps.y; pre[2] += ps.z; pre[3] += ps.w;
        }
        ig = fast_sigmoid(pre[0]); fg = fast_sigmoid(pre[1]); gg = fast_tanh(pre[2]); og = fast_sigmoid(pre[3]);
        c = fg * c + ig * gg;
        h = og * fast_tanh(c);
      }
      // publish step t into the exchange slice of parity t & 1
      const unsigned wslice = (unsigned)(((t & 1) * 2 + d) * W2_SLICE_FLOATS);
      if (a.pack16) {     // every member owns 8 aligned units: lanes eu = 0, 4 store four units of their row as ONE 16-byte piece
        const float h1 = __shfl_down(h, 1), h2 = __shfl_down(h, 2), h3 = __shfl_down(h, 3);
        if (live && (eu & 3) == 0) {
          const f32x4 hv = {h, h1, h2, h3};
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hv), (t & 1) ? xr1 : xr0, xrow_w * 4u, 0, 16 /* sc1 */);
        }
      } else if (live) {
        st_sc1(xg + wslice + xrow_w, h);
      }
#ifdef SUMK_DIAG
      const unsigned long long s4 = __builtin_amdgcn_s_memtime();
      dg_epi += s4 - s3;
#endif
      // every storing wave drains its stores (and nothing younger than the G loads of the step's top), then one lane signals
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(bar + 32 * (slot & 3), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef SUMK_DIAG
      dg_pub += __builtin_amdgcn_s_memtime() - s4;
#endif
      if (live) {   // the layer's output and the training saves: nobody reads them in this launch, they leave behind the signal
        a.Hout[row * (2 * H) + d * H + j] = h;
        if constexpr (MODE != 2) {     // (the bf16x6 recurrence is an inference mode: the host never selects it with training saves)
          if (a.gates) {
            float* gs = a.gates + row * (8 * H) + d * 4 * H;
            gs[j] = ig; gs[H + j] = fg; gs[2 * H + j] = gg; gs[3 * H + j] = og;
            a.c_all[row * (2 * H) + d * H + j] = c;
            a.hprev[row * (2 * H) + d * H + j] = hlast;
          }
          hlast = h;
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) gcur[q] = gnext[q];
    }
  }
#ifdef SUMK_DIAG
  if (blockIdx.x < 12 && lane == 0 && (wave == 0 || wave == 7)) {
    unsigned long long* q = reinterpret_cast<unsigned long long*>(a.state + W2_STAMP_WORD) + (blockIdx.x * 2 + (wave == 7)) * 8;
    q[0] = __builtin_amdgcn_s_memtime() - dg_t0; q[1] = dg_poll; q[2] = dg_mfma; q[3] = dg_bar1; q[4] = dg_epi; q[5] = dg_pub; q[6] = dg_steps; q[7] = dg_two;
  }
#endif
}

// ------------------------------------------------------------------------------------------- BPTT step kernel
// Step t (run for t = t_max-1 .. 0).  Block = (32 videos) x (32 hidden units) x direction, 512 threads = 8 waves that
// split the K = 4H contraction  dh_rec[video, j] = sum_k dG[next row][k] * W_hh[k][j]  (MFMA, fragments straight from
// L2), partial tiles summed through LDS; then each thread owns two (video, unit) pairs of the cell backward.
struct BwdStepArgs {
  const float* whh[2];   // (4H, H)
  const float* dHout;    // (R, 2H) upstream gradient of the layer output
  const float* gates;    // (R, 8H) i,f,g,o (post-nonlinearity) saved by the forward
  const float* c_all;    // (R, 2H)
  float* dG;             // (R, 8H) gradient w.r.t. gate pre-activations (output)
  float* dcstate;        // (n_seq, 2, H) running dc
  const int32_t* off;
  int32_t n_seq, H, t, n_jblk;
  // nd as in StepArgs.  Optional (n_seq, nd, H) tensors: c0 = initial cell state of the forward pass; dh_last / dc_last =
  // upstream gradients of the final (h, c); dh0 / dc0 = outputs written by the extra pass t = -1.  dHout may be nullptr.
  int32_t nd = 2;
  const float* c0 = nullptr;
  const float* dh_last = nullptr;
  const float* dc_last = nullptr;
  float* dh0 = nullptr;
  float* dc0 = nullptr;
};

__global__ __launch_bounds__(512) void lstm_bwd_step_kernel(BwdStepArgs a) {
  __shared__ float part[8][32][33];
  const int H = a.H, H4 = 4 * H, nd = a.nd, S1 = nd * H, S4 = nd * H4;
  const int d = blockIdx.x % nd;
  const int jblk = (blockIdx.x / nd) % a.n_jblk;
  const int mtile = (blockIdx.x / nd) / a.n_jblk;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int j0 = jblk * 32;
  const int t = a.t;
  {
    // A[i][k] = dG[next row of video i][d*4H + k] (zero when video i has no step t+1); B[k][j] = W_hh[d][k][j0 + j]
    const int sv = mtile * 32 + li;
    bool act = false;
    const float* gp = a.dG;
    if (sv < a.n_seq) {
      int r0 = a.off[sv], T = a.off[sv + 1] - r0;
      if (t + 1 < T) { act = true; gp = a.dG + (int64_t)(d == 0 ? r0 + t + 1 : r0 + T - 2 - t) * S4 + d * H4; }
    }
    const int jc = min(j0 + li, H - 1);
    const float* wp = a.whh[d] + jc;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int nchunk = H4 >> 3;   // 4H is a multiple of 16
    // chunks kk = wave, wave+8, ...; 8 at a time with all 40 loads in flight before the first MFMA (see lstm_step_kernel)
    for (int kb = wave; kb < nchunk; kb += 64) {
      float4 av[8];
      float bq[8][4];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int k = min((kb + 8 * q) * 8 + 4 * lh, H4 - 4);
        av[q] = *reinterpret_cast<const float4*>(gp + k);
#pragma unroll
        for (int e = 0; e < 4; ++e) bq[q][e] = wp[(int64_t)(k + e) * H];
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const bool live = act && (kb + 8 * q) < nchunk;
        if (!live) av[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].x, bq[q][0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].y, bq[q][1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].z, bq[q][2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].w, bq[q][3], acc, 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) part[wave][(r & 3) + 8 * (r >> 2) + 4 * lh][li] = acc[r];
    __syncthreads();
  }
  // ---- cell backward: thread handles (video i, unit u) for i = tid>>5 and i + 16
#pragma unroll
  for (int rep = 0; rep < 2; ++rep) {
    const int i = (tid >> 5) + 16 * rep, u = tid & 31;
    const int sv = mtile * 32 + i, j = j0 + u;
    if (sv >= a.n_seq || j >= H) continue;
    const int r0 = a.off[sv], T = a.off[sv + 1] - r0;
    if (t >= T) continue;
    const int64_t sidx = ((int64_t)sv * nd + d) * H + j;
    float rec = 0.f;
#pragma unroll
    for (int w8 = 0; w8 < 8; ++w8) rec += part[w8][i][u];
    float* dcs = a.dcstate + sidx;
    if (t < 0) {   // extra pass: gradient of the initial state
      if (a.dh0) a.dh0[sidx] = rec;
      if (a.dc0) a.dc0[sidx] = *dcs;
      continue;
    }
    const int64_t row = d == 0 ? r0 + t : r0 + T - 1 - t;
    float dh = a.dHout ? a.dHout[row * S1 + d * H + j] : 0.f;
    if (t + 1 < T) dh += rec;
    else if (a.dh_last) dh += a.dh_last[sidx];
    const float* gs = a.gates + row * S4 + d * H4;
    const float ig = gs[j], fg = gs[H + j], gg = gs[2 * H + j], og = gs[3 * H + j];
    const float c = a.c_all[row * S1 + d * H + j];
    const float cprev = t > 0 ? a.c_all[(d == 0 ? row - 1 : row + 1) * S1 + d * H + j] : (a.c0 ? a.c0[sidx] : 0.f);
    const float tc = tanhf(c);
    float dc = (t + 1 < T ? *dcs : (a.dc_last ? a.dc_last[sidx] : 0.f)) + dh * og * (1.f - tc * tc);
    float* dg = a.dG + row * S4 + d * H4;
    dg[j] = dc * gg * ig * (1.f - ig);
    dg[H + j] = dc * cprev * fg * (1.f - fg);
    dg[2 * H + j] = dc * ig * (1.f - gg * gg);
    dg[3 * H + j] = dh * tc * og * (1.f - og);
    *dcs = dc * fg;
  }
}

// ------------------------------------------------------------------------------------------- persistent BPTT
// Same teams / work items / hand-off protocol as lstm_persist_kernel.  Member m owns hidden units U_m = [m*upm, ...):
//   step t:  dh[i][j in U_m] = dHout[row][j] + sum over the 32 members m' of  partial_{m'}(t+1)[i][j]      (sc1 loads)
//            cell backward -> dG_t[i][4 gates x U_m]  (kept in LDS as the MFMA A operand, and stored for the weight-gradient GEMMs)
//            partial_m(t)[i][all j] = dG_t[i][cols of U_m] . W_hh[rows of U_m][all j]   (MFMA, K = 32, one N-tile per wave)
//            published (sc1 stores) to the exchange buffer of parity t&1, then the step counter is bumped.
// So a member publishes H floats per video per step and reads 32 x |U_m| -- the same bytes as the forward pass, with the
// sum over members taken in a FIXED order (deterministic).  dc lives in a register; W_hh rows of U_m stay in LDS.
struct PersistBwdArgs {
  const float* whh[2]; const float* dHout; const float* gates; const float* c_all;
  float* dG; float* xchg; const int32_t* off; unsigned* state;
  int32_t n_seq, H, gsize, n_groups, upm, n_active, n_items;
  // round 6 (counter hand-off): item_words / n_shards -- the step counter of an item as n_shards words 32 words (128 B) apart, member m adds to
  // shard m % n_shards, n_shards lanes poll (32 atomics on one word serialise at ~12 ns each); coal -- the exchange laid out for its READERS
  // (members own 8 aligned units): [parity][item][reader c = column / 8][producer m][video < gsize][8 columns], so the 32 partials a cell-update
  // thread sums are 32 loads of ONE contiguous 13-KB block per member instead of 32 B out of every producer's 1-KB rows
  int32_t item_words, n_shards, coal;
  unsigned long long* ll;     // LL instances: {partial, step tag} packets [parity 2][item][member 32][video < gsize][H]
};

// LL / M16: as in lstm_persist_kernel -- the partial products travel as 8-byte {value, step tag} packets and the consumers' loads
// are the poll (no counter, no vmcnt(0) drain, no atomic); groups of <= 16 videos multiply on v_mfma_f32_16x16x4_f32 with the W_hh
// fragments of the wave's 32 output columns held in REGISTERS for the whole item (the 32-row form re-reads them from LDS every step).
// Tag of the partials of step s (s = Tg-1 .. 1): Tg - s >= 1; the buffer is zeroed before the launch (0 matches no step).
template <bool LL = false, bool M16 = false>
__global__ __launch_bounds__(PK_THREADS) void lstm_persist_bwd_kernel(PersistBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int H = a.H, P = H + 4;
  float* sW = smem;                       // [32][P]  W_hh rows (gate*8+unit) of this member, all H columns
  float* sA = sW + 32 * P;                // [32][36] dG_t of the group's videos, this member's 32 gate columns
  int* sR0 = reinterpret_cast<int*>(sA + 32 * 36);
  int* sT = sR0 + 32;
  int* sTg = sT + 32;

  const int team = blockIdx.x % PK_TEAMS, slot = blockIdx.x / PK_TEAMS;
  if (slot >= a.n_active) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int u0 = slot * a.upm, nu = min(a.upm, H - u0);
  const int H4 = 4 * H;
  int loaded_dir = -1;
  bool dead = false;
  float wB[2][8];     // M16: this lane's W_hh fragments
#ifdef SUMK_DIAG   // `make DIAG=1`: per-phase shader cycles of a step, wave 0 (counter poll + epilogue role) and wave 7 of the first members -> state words 600..
  unsigned long long bg_wait = 0, bg_cell = 0, bg_bar = 0, bg_mfma = 0, bg_drain = 0, bg_steps = 0;
  const unsigned long long bg_t0 = __builtin_amdgcn_s_memtime();
#endif

  for (int item = team; item < a.n_items; item += PK_TEAMS) {
    const int g = item >> 1, d = item & 1;
    const int v0 = g * a.gsize, nv = min(a.gsize, a.n_seq - v0);
    unsigned* bar = a.state + 16 + item * a.item_words;
    __syncthreads();
    if (M16 && loaded_dir != d) {   // B fragments: W_hh[row k = 8 (lane / 16) + m of this member's 32][column 32 wave + 16 tile + lane % 16]
#pragma unroll
      for (int tile = 0; tile < 2; ++tile) {
        const int n = wave * 32 + 16 * tile + (lane & 15);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          const int r = 8 * (lane >> 4) + m;
          wB[tile][m] = n < H ? a.whh[d][(int64_t)((r >> 3) * H + min(u0 + (r & 7), H - 1)) * H + n] : 0.f;
        }
      }
      loaded_dir = d;
    }
    if (loaded_dir != d) {
      for (int idx = tid; idx < 32 * (H >> 2); idx += PK_THREADS) {
        const int r = idx / (H >> 2), k4 = (idx % (H >> 2)) * 4;
        const int unit = min(u0 + (r & 7), H - 1);
        *reinterpret_cast<float4*>(&sW[r * P + k4]) =
            *reinterpret_cast<const float4*>(a.whh[d] + (int64_t)((r >> 3) * H + unit) * H + k4);
      }
      loaded_dir = d;
    }
    if (tid == 0) *sTg = 0;
    __syncthreads();
    if (tid < 32) {
      int r0 = 0, T = 0;
      if (tid < nv) { r0 = a.off[v0 + tid]; T = a.off[v0 + tid + 1] - r0; atomicMax(sTg, T); }
      sR0[tid] = r0; sT[tid] = T;
    }
    __syncthreads();
    const int Tg = *sTg;

    const int ei = tid >> 3, eu = tid & 7;
    const bool erole = tid < 256 && ei < nv && eu < nu;
    const int er0 = erole ? sR0[ei] : 0, eT = erole ? sT[ei] : 0;
    const int j = u0 + eu;
    float dcarry = 0.f;
    // saved activations of the step about to be processed (prefetched one step ahead)
    float sv_i = 0.f, sv_f = 0.f, sv_g = 0.f, sv_o = 0.f, sv_c = 0.f, sv_cp = 0.f, sv_dh = 0.f;
    float nx_i = 0.f, nx_f = 0.f, nx_g = 0.f, nx_o = 0.f, nx_c = 0.f, nx_cp = 0.f, nx_dh = 0.f;     // the step after: requested at the TOP of a step (round 6)
    auto fetch = [&](int t) {
      const int64_t row = d == 0 ? er0 + t : er0 + eT - 1 - t;
      const float* gs = a.gates + row * (8 * H) + d * H4;
      nx_i = gs[j]; nx_f = gs[H + j]; nx_g = gs[2 * H + j]; nx_o = gs[3 * H + j];
      nx_c = a.c_all[row * (2 * H) + d * H + j];
      nx_cp = t > 0 ? a.c_all[(d == 0 ? row - 1 : row + 1) * (2 * H) + d * H + j] : 0.f;
      nx_dh = a.dHout[row * (2 * H) + d * H + j];
    };
    if (erole && eT > 0 && eT - 1 == Tg - 1) {
      fetch(Tg - 1);
      sv_i = nx_i; sv_f = nx_f; sv_g = nx_g; sv_o = nx_o; sv_c = nx_c; sv_cp = nx_cp; sv_dh = nx_dh;
    }

    for (int t = Tg - 1; t >= 0; --t) {
#ifdef SUMK_DIAG
      const unsigned long long bs0 = __builtin_amdgcn_s_memtime();
#endif
      // next step's saved activations: seven strided loads per thread, requested here so that they have the whole step -- the producer's
      // vmcnt(0) drain in front of the signal used to wait for them (they were issued behind the cell update)
      const bool have_next = erole && t - 1 >= 0 && t - 1 < eT;
      if (have_next) fetch(t - 1);
      // zero this step's A tile (rows of inactive videos and columns of absent units must contribute nothing)
      for (int idx = tid; idx < 32 * 36; idx += PK_THREADS) sA[idx] = 0.f;
      if (!LL && t < Tg - 1) {
        if (wave == 0 && !dead) {   // wait until every member published step t+1: lanes 0 .. n_shards-1 poll one shard each
          const unsigned cnt = (lane < a.n_shards && lane < a.n_active) ? (unsigned)((a.n_active - lane + a.n_shards - 1) / a.n_shards) : 0u;
          const unsigned want = (unsigned)(Tg - 1 - t) * cnt;
          unsigned spins = 0;
          while (true) {
            const unsigned v = lane < a.n_shards ? __hip_atomic_load(bar + 32 * lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffffffu;
            if (__all(v >= want)) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > PK_SPIN_LIMIT || ((spins & 1023) == 0 &&
                 __hip_atomic_load(a.state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
              if (lane == 0) { atomicOr(a.state, 1u); atomicOr(&g_sumk_health, 1u); }
              dead = true; break;
            }
          }
        }
      }
      __syncthreads();
#ifdef SUMK_DIAG
      const unsigned long long bs1 = __builtin_amdgcn_s_memtime();
#endif
      float rec_ll = 0.f;
      if constexpr (LL) {   // the 32 members' partials of step t+1: the loads are the poll (every wave of the epilogue role spins for itself)
        if (tid < 256) {
          const bool need = erole && t + 1 < eT;
          const unsigned want = (unsigned)(Tg - 1 - t);
          const unsigned long long* xp = a.ll + ((((int64_t)((t + 1) & 1) * a.n_items + item) * 32) * a.gsize + (need ? ei : 0)) * H + (need ? j : 0);
          unsigned spins = 0;
          unsigned long long ll_t0 = 0;
          const unsigned long long* xb64 = a.ll + (((int64_t)((t + 1) & 1) * a.n_items + item) * 32) * a.gsize * H;
          while (true) {
            unsigned long long pv[32];
            bool ok = true;
            float rec = 0.f;
            if (a.n_active == 32) {     // all members present: 32-bit buffer offsets walked in a VGPR, no per-member predicate (see the counter form below)
              typedef unsigned int u32x2_ __attribute__((ext_vector_type(2)));
              const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned long long*>(xb64), (short)0, 0x7FFFFFFF, 0x00020000);
              unsigned vo = (unsigned)(xp - xb64) * 8u;
              const unsigned st = (unsigned)(a.gsize * H) * 8u;
#pragma unroll
              for (int m = 0; m < 32; ++m) { pv[m] = __builtin_bit_cast(unsigned long long, (u32x2_)__builtin_amdgcn_raw_buffer_load_b64(xr, vo, 0, 16 /* sc1 */)); vo += st; }
#pragma unroll
              for (int m = 0; m < 32; ++m) {
                ok = ok & ((unsigned)(pv[m] >> 32) == want);
                rec += __builtin_bit_cast(float, (unsigned)pv[m]);
              }
              ok = ok | !need;
              if (!need) rec = 0.f;
            } else {
#pragma unroll
            for (int m = 0; m < 32; ++m)
              pv[m] = (need && m < a.n_active) ? __hip_atomic_load(xp + (int64_t)m * a.gsize * H, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
#pragma unroll
            for (int m = 0; m < 32; ++m) {
              if (need && m < a.n_active) ok = ok && (unsigned)(pv[m] >> 32) == want;
              rec += __builtin_bit_cast(float, (unsigned)pv[m]);       // fixed order m = 0 .. 31 (absent members add +0)
            }
            }
            rec_ll = rec;
            if (__all(ok) || dead) break;
            if (pk_ll_timed_out(spins, ll_t0)) {
              if (lane == 0) { atomicOr(a.state, 1u); atomicOr(&g_sumk_health, 1u); }
              dead = true;
            }
          }
        }
      }
      if (erole && t < eT) {
        float dh = sv_dh;
        if (LL) {
          if (t + 1 < eT) dh += rec_ll;
        } else
        if (t + 1 < eT) {   // recurrent part: fixed-order sum of the 32 members' partials of step t+1 (sc1 loads)
          const float* xb = a.xchg + (((int64_t)((t + 1) & 1) * a.n_items + item) * 32) * 32 * H;
          const float* xp = a.coal ? xb + ((int64_t)slot * 32 * a.gsize + ei) * 8 + eu : xb + (int64_t)ei * H + j;
          const int64_t xs = a.coal ? (int64_t)a.gsize * 8 : (int64_t)32 * H;        // producer to producer
          float pv[32];
          if (a.n_active == 32) {
            // All 32 members (H = 256 at 8 units each, the DSN shape): no per-member predicate, and the member stride walks a 32-bit VGPR offset of a buffer
            // load.  The generic form below kept 32 predicate masks and 32 64-bit offsets alive across the step loop -- 145 spilled SGPRs, four v_readlane
            // and a branch in front of every one of these loads, ~1 000 of a step's 7 400 cycles, all behind the poll.
            const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), (short)0, 0x7FFFFFFF, 0x00020000);
            unsigned vo = (unsigned)(xp - xb) * 4u;
            const unsigned st = (unsigned)xs * 4u;
#pragma unroll
            for (int m = 0; m < 32; ++m) { pv[m] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, vo, 0, 16 /* sc1 */)); vo += st; }
          } else {
#pragma unroll
            for (int m = 0; m < 32; ++m) pv[m] = m < a.n_active ? ld_sc1(xp + (int64_t)m * xs) : 0.f;
          }
          float rec = 0.f;
#pragma unroll
          for (int m = 0; m < 32; ++m) rec += pv[m];
          dh += rec;
        }
        const float tc = fast_tanh(sv_c);     // (round 6: the forward kernels' v_exp / v_rcp form, ~1e-7 absolute, instead of the library tanhf's ~40 instructions)
        const float dc = dcarry + dh * sv_o * (1.f - tc * tc);
        const float d_i = dc * sv_g * sv_i * (1.f - sv_i);
        const float d_f = dc * sv_cp * sv_f * (1.f - sv_f);
        const float d_g = dc * sv_i * (1.f - sv_g * sv_g);
        const float d_o = dh * tc * sv_o * (1.f - sv_o);
        dcarry = dc * sv_f;
        const int64_t row = d == 0 ? er0 + t : er0 + eT - 1 - t;
        float* dg = a.dG + row * (8 * H) + d * H4;
        dg[j] = d_i; dg[H + j] = d_f; dg[2 * H + j] = d_g; dg[3 * H + j] = d_o;
        sA[ei * 36 + eu] = d_i; sA[ei * 36 + 8 + eu] = d_f; sA[ei * 36 + 16 + eu] = d_g; sA[ei * 36 + 24 + eu] = d_o;
      }
      if (have_next) { sv_i = nx_i; sv_f = nx_f; sv_g = nx_g; sv_o = nx_o; sv_c = nx_c; sv_cp = nx_cp; sv_dh = nx_dh; }
#ifdef SUMK_DIAG
      const unsigned long long bs2 = __builtin_amdgcn_s_memtime();
#endif
      __syncthreads();
#ifdef SUMK_DIAG
      const unsigned long long bs3 = __builtin_amdgcn_s_memtime();
#endif
      if (t > 0) {   // partial_m(t) is only ever read by step t-1
        if constexpr (M16) {
          // 16 videos x (this wave's 32 columns, two 16-column tiles) x K = 32: lane group g = lane / 16 carries k = 8 g .. 8 g + 7
          const int i16 = lane & 15, g4 = lane >> 4;
          const float4 a0 = *reinterpret_cast<const float4*>(&sA[i16 * 36 + 8 * g4]);
          const float4 a1 = *reinterpret_cast<const float4*>(&sA[i16 * 36 + 8 * g4 + 4]);
          const unsigned tag = (unsigned)(Tg - t);
          unsigned long long* xo = a.ll + ((((int64_t)(t & 1) * a.n_items + item) * 32 + slot) * a.gsize) * H;
          float* xf = a.xchg + ((((int64_t)(t & 1) * a.n_items + item) * 32 + slot) * 32) * H;        // counter hand-off: fp32 partials
          if constexpr (!LL) {
            // Counter hand-off: the operands SWAPPED (W as "A", dG as "B") -- the tile comes out transposed, D^T[n][i], so a lane holds
            // FOUR CONSECUTIVE COLUMNS n = 4 (lane / 16) + r of video i = lane % 16 and the publish is ONE 16-byte sc1 store per tile
            // instead of four dword ones (dword sc1 stores cost ~6x per byte: the stamps of profiles/r04_lstm_bptt_phase_stamps.txt
            // had 2 700 of a step's 11 250 cycles in this block, 512 of them MFMA).  Same products, same k order: the same bits.
#pragma unroll
            for (int tile = 0; tile < 2; ++tile) {
              f32x4 acc = {0.f, 0.f, 0.f, 0.f};
              acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wB[tile][0], a0.x, acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wB[tile][1], a0.y, acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wB[tile][2], a0.z, acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wB[tile][3], a0.w, acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wB[tile][4], a1.x, acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wB[tile][5], a1.y, acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wB[tile][6], a1.z, acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wB[tile][7], a1.w, acc, 0, 0, 0);
              const int n4 = wave * 32 + 16 * tile + 4 * g4;           // first of this lane's four columns (H % 4 == 0: all four inside or outside)
              if (n4 < H && i16 < nv && t < sT[i16]) {
                typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
                // coal: the reader of columns n4 .. n4 + 3 is member n4 / 8; its block of THIS producer is [video < gsize][8 columns]
                float* xw = a.coal ? a.xchg + (((int64_t)(t & 1) * a.n_items + item) * 32) * 32 * H : xf;
                const unsigned xo4 = a.coal ? (unsigned)(((((n4 >> 3) * 32 + slot) * a.gsize + i16) * 8 + (n4 & 7)) * 4) : (unsigned)((i16 * H + n4) * 4);
                const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(xw, (short)0, 0x7FFFFFFF, 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_, acc), xr, xo4, 0, 16 /* sc1 */);
              }
            }
          } else
#pragma unroll
          for (int tile = 0; tile < 2; ++tile) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, wB[tile][0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, wB[tile][1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, wB[tile][2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, wB[tile][3], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, wB[tile][4], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, wB[tile][5], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, wB[tile][6], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, wB[tile][7], acc, 0, 0, 0);
            const int n = wave * 32 + 16 * tile + i16;
            if (n < H) {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int i = 4 * g4 + r;      // C/D map of the 16x16 MFMA: row = 4 (lane / 16) + r, column = lane % 16
                const float pv_ = acc[r];     // (a scalar copy: __builtin_bit_cast applied to a vector ELEMENT read element 0 with this compiler)
                if (i < nv && t < sT[i]) {
                  if constexpr (LL)
                    __hip_atomic_store(xo + (int64_t)i * H + n, ((unsigned long long)tag << 32) | (unsigned long long)__builtin_bit_cast(unsigned, pv_),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  else
                    st_sc1(a.coal ? a.xchg + (((int64_t)(t & 1) * a.n_items + item) * 32) * 32 * H + ((((int64_t)(n >> 3) * 32 + slot) * a.gsize + i) * 8 + (n & 7)) : xf + (int64_t)i * H + n, pv_);
                }
              }
            }
          }
        } else {
        float* xo = a.xchg + ((((int64_t)(t & 1) * a.n_items + item) * 32 + slot) * 32) * H;
        unsigned long long* xl = a.ll + ((((int64_t)(t & 1) * a.n_items + item) * 32 + slot) * a.gsize) * H;
        const unsigned tag = (unsigned)(Tg - t);
        const int ntile = (H + 31) >> 5;
        for (int nt = wave; nt < ntile; nt += 8) {
          const int n = nt * 32 + li, nc = min(n, H - 1);
          f32x16 acc;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const float4 av = *reinterpret_cast<const float4*>(&sA[li * 36 + kk * 8 + 4 * lh]);
            const float b0 = sW[(kk * 8 + 4 * lh + 0) * P + nc], b1 = sW[(kk * 8 + 4 * lh + 1) * P + nc];
            const float b2 = sW[(kk * 8 + 4 * lh + 2) * P + nc], b3 = sW[(kk * 8 + 4 * lh + 3) * P + nc];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b2, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b3, acc, 0, 0, 0);
          }
          if (n < H) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;
              if (i < nv && t < sT[i]) {
                if constexpr (LL) {
                  const float pv_ = acc[r];   // (scalar copy: see the 16-row form)
                  __hip_atomic_store(xl + (int64_t)i * H + n, ((unsigned long long)tag << 32) | (unsigned long long)__builtin_bit_cast(unsigned, pv_),
                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else
                  st_sc1(a.coal ? a.xchg + (((int64_t)(t & 1) * a.n_items + item) * 32) * 32 * H + ((((int64_t)(n >> 3) * 32 + slot) * a.gsize + i) * 8 + (n & 7)) : xo + (int64_t)i * H + n, acc[r]);
              }
            }
          }
        }
        }
      }
#ifdef SUMK_DIAG
      const unsigned long long bs4 = __builtin_amdgcn_s_memtime();
#endif
      if constexpr (LL) {
        __syncthreads();   // the A tile in LDS is free again (the published partials need no further signal)
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(bar + 32 * (slot % a.n_shards), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#ifdef SUMK_DIAG
      bg_wait += bs1 - bs0; bg_cell += bs2 - bs1; bg_bar += bs3 - bs2; bg_mfma += bs4 - bs3; bg_drain += __builtin_amdgcn_s_memtime() - bs4; bg_steps += 1;
#endif
    }
  }
#ifdef SUMK_DIAG
  if (!LL && M16 && blockIdx.x < 96 && (blockIdx.x % PK_TEAMS) == 0 && lane == 0 && (wave == 0 || wave == 7)) {      // members 0 .. 11 of team 0
    unsigned long long* q = reinterpret_cast<unsigned long long*>(a.state + 600) + ((blockIdx.x / PK_TEAMS) * 2 + (wave == 7)) * 8;
    q[0] = __builtin_amdgcn_s_memtime() - bg_t0; q[1] = bg_wait; q[2] = bg_cell; q[3] = bg_bar; q[4] = bg_mfma; q[5] = bg_drain; q[6] = bg_steps;
  }
#endif
}

// ------------------------------------------------------------------------------------------- wide persistent BPTT
// One launch runs the whole backward recurrence for 256 < H <= 1024 (sLSTM: H = 1024; the launch chain above re-reads the
// 16 MB W_hh per direction from the Infinity Cache every step: 74 us per step measured, 60 % of an sLSTM training step).
// Two teams of up to 128 CUs, team = direction (as lstm_wide_kernel).  The contraction of step t,
//     dh_rec[i][j] = sum_k dG_t[i][k] W_hh[k][j]        (i < 64 videos, k < 4H, j < H),
// is tiled K x N over the team: member (kg, ng) keeps W_hh[4 gates x units 32kg..32kg+31][columns 256ng..256ng+255] -- a
// 128 x 256 slice, 128 KB -- as MFMA B fragments in REGISTERS for the whole launch (64 VGPRs per lane; wave w owns the 32
// columns 256ng + 32w..), so W_hh never moves again.  Per step a member
//   1. sums, in a FIXED order, the KG partial products of step t+1 for ITS 32 units (sc1 16-B loads: 256 KB per member and
//      step, the same bytes as the forward kernel's h panel), adds dHout and runs the cell backward for 64 videos x 32 units
//      (dc stays in registers) -- the NG members that share kg do this redundantly and bit-identically, only ng = 0 stores dG;
//   2. puts dG_t[64][its 128 gate columns] in LDS as the MFMA A operand (34 KB), multiplies by its register-resident slice
//      (2 x 64 fp32 MFMAs per wave, no split-K inside the CU) and publishes partial_kg(t)[64][256 columns] with sc1 stores
//      into the exchange buffer of parity t & 1, then bumps the item's step counter.
// Hand-off protocol, bounded spins and the error word are those of lstm_persist_kernel (Guideline 16 R1); exchange buffers
// are double-buffered by step parity AND by group parity (a member may start group g+1 while a slow one still reads group g).
struct WideBwdArgs {
  const float* whh[2]; const float* dHout; const float* gates; const float* c_all;
  float* dG; float* xchg; const int32_t* off; unsigned* state;
  int32_t n_seq, H, n_groups, KG, NG, n_active, xchg_dir_bytes;
  int32_t item_words, n_shards;      // round 6: the step counter of an item as n_shards words 32 words apart (lstm_persist_bwd_kernel)
};
constexpr int WB_P = 136;   // LDS row pitch of the A tile (floats): 8 lanes x 16 B cover the 32 banks

__global__ __launch_bounds__(PK_THREADS) void lstm_wide_bwd_kernel(WideBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int H = a.H, H4 = 4 * H;
  float* sA = smem;                                          // [64 videos][WB_P]  dG_t, this member's 128 gate columns
  int* sR0 = reinterpret_cast<int*>(sA + 64 * WB_P);         // [64] first frame row of the video on MFMA row r (rows sorted by length, as lstm_wide2_kernel)
  int* sT = sR0 + 64;                                        // [64] its length
  int* sR0u = sT + 64;                                       // [64] the same in batch order (input of the sort)
  int* sTu = sR0u + 64;

  const int d = (blockIdx.x & 7) >> 2;                       // team = direction
  const int slot = (blockIdx.x >> 3) * 4 + (blockIdx.x & 3);
  if (slot >= a.n_active) return;
  // (kg, ng): the NG members that share kg read the same partials -- kept on one XCD (speed only)
  const int rest = slot >> 2, ng = rest % a.NG, kg = (rest / a.NG) * 4 + (slot & 3);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  float* xbase = a.xchg + (size_t)d * (a.xchg_dir_bytes >> 2);
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(xbase, (short)0, a.xchg_dir_bytes, 0x00020000);
  constexpr unsigned OOB = 0x7ffffff0u;
  bool dead = false;

  // this lane's W_hh fragments: chunk c = local k 8c..8c+7 (gate c>>2, units 8(c&3)..), lane half lh holds k = 8c+4lh .. +3
  const int ncol = ng * 256 + wave * 32 + li;
  float4 wreg[16];
  {
    const float* wp = a.whh[d] + min(ncol, H - 1);
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const int64_t krow = (int64_t)(c >> 2) * H + 32 * kg + 8 * (c & 3) + 4 * lh;
      float4 w4;
      w4.x = wp[(krow + 0) * H]; w4.y = wp[(krow + 1) * H]; w4.z = wp[(krow + 2) * H]; w4.w = wp[(krow + 3) * H];
      wreg[c] = ncol < H ? w4 : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  const size_t slab = (size_t)a.KG * 64 * H;                 // floats per (group parity, step parity) exchange buffer

  for (int g = 0; g < a.n_groups; ++g) {
    const int item = 2 * g + d;
    const int v0 = g * WK_GROUP, nv = min(WK_GROUP, a.n_seq - v0);
    unsigned* bar = a.state + 16 + item * a.item_words;
    __syncthreads();
    if (tid < 64) {
      int r0 = 0, T = 0;
      if (tid < nv) { r0 = a.off[v0 + tid]; T = a.off[v0 + tid + 1] - r0; }
      sR0u[tid] = r0; sTu[tid] = (T << 6) | (63 - tid);
    }
    __syncthreads();
    if (tid < 64) {     // MFMA row = rank by length, descending; ties by batch index (round 6: the second 32-row tile stops once fewer than 33 videos run)
      const int key = sTu[tid];
      int rank = 0;
#pragma unroll 8
      for (int q = 0; q < 64; ++q) rank += sTu[q] > key ? 1 : 0;
      sT[rank] = key >> 6; sR0[rank] = sR0u[tid];
    }
    __syncthreads();
    const int Tg = sT[0], T32 = sT[32];

    // cell role: thread = (video ei, unit quad uq): units j..j+3 of this member's 32
    const int ei = tid >> 3, uq = tid & 7;
    const bool erole = ei < nv;
    const int er0 = erole ? sR0[ei] : 0, eT = erole ? sT[ei] : 0;
    const int j = 32 * kg + 4 * uq;
    float dcarry[4] = {0.f, 0.f, 0.f, 0.f};
    float4 sv_i, sv_f, sv_g, sv_o, sv_c, sv_cp, sv_dh;
    sv_i = sv_f = sv_g = sv_o = sv_c = sv_cp = sv_dh = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 nx_i = sv_i, nx_f = sv_i, nx_g = sv_i, nx_o = sv_i, nx_c = sv_i, nx_cp = sv_i, nx_dh = sv_i;      // the step after: requested at the TOP of a step (round 6)
    auto fetch = [&](int t) {
      const int64_t row = d == 0 ? er0 + t : er0 + eT - 1 - t;
      const float* gs = a.gates + row * (8 * H) + d * H4 + j;
      nx_i = *reinterpret_cast<const float4*>(gs); nx_f = *reinterpret_cast<const float4*>(gs + H);
      nx_g = *reinterpret_cast<const float4*>(gs + 2 * H); nx_o = *reinterpret_cast<const float4*>(gs + 3 * H);
      nx_c = *reinterpret_cast<const float4*>(a.c_all + row * (2 * H) + d * H + j);
      nx_cp = t > 0 ? *reinterpret_cast<const float4*>(a.c_all + (d == 0 ? row - 1 : row + 1) * (2 * H) + d * H + j)
                    : make_float4(0.f, 0.f, 0.f, 0.f);
      nx_dh = *reinterpret_cast<const float4*>(a.dHout + row * (2 * H) + d * H + j);
    };
    if (erole && eT == Tg) {
      fetch(Tg - 1);
      sv_i = nx_i; sv_f = nx_f; sv_g = nx_g; sv_o = nx_o; sv_c = nx_c; sv_cp = nx_cp; sv_dh = nx_dh;
    }
    const size_t gbase = (size_t)(g & 1) * 2 * slab;

    for (int t = Tg - 1; t >= 0; --t) {
      const bool have_next = erole && t - 1 >= 0 && t - 1 < eT;
      if (have_next) fetch(t - 1);      // next step's saved activations: the whole step in flight (they used to sit in front of the publish drain)
      if (t < Tg - 1) {
        if (wave == 0 && !dead) {   // wait until every member published step t+1: lanes 0 .. n_shards-1 poll one shard each
          const unsigned cnt = (lane < a.n_shards && lane < a.n_active) ? (unsigned)((a.n_active - lane + a.n_shards - 1) / a.n_shards) : 0u;
          const unsigned want = (unsigned)(Tg - 1 - t) * cnt;
          unsigned spins = 0;
          while (true) {
            const unsigned v = lane < a.n_shards ? __hip_atomic_load(bar + 32 * lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffffffu;
            if (__all(v >= want)) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > PK_SPIN_LIMIT || ((spins & 1023) == 0 &&
                 __hip_atomic_load(a.state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
              if (lane == 0) { atomicOr(a.state, 1u); atomicOr(&g_sumk_health, 1u); }
              dead = true; break;
            }
          }
        }
      }
      __syncthreads();   // hand-off seen by every wave; previous step's MFMA reads of sA are over
      float4 o_i, o_f, o_g, o_o;
      o_i = o_f = o_g = o_o = make_float4(0.f, 0.f, 0.f, 0.f);
      if (erole && t < eT) {
        float dh[4] = {sv_dh.x, sv_dh.y, sv_dh.z, sv_dh.w};
        if (t + 1 < eT) {   // recurrent part: fixed-order sum over kg' of partial_{kg'}(t+1)[ei][j..j+3]   (sc1 loads ONLY)
          const unsigned b0 = (unsigned)((gbase + (size_t)((t + 1) & 1) * slab + (size_t)ei * H + j) * 4);
#pragma unroll 1
          for (int mb = 0; mb < 32; mb += 16) {   // 16 loads in flight per lane (32 would spill next to the 64 weight VGPRs)
            u32x4 pv[16];
#pragma unroll
            for (int m = 0; m < 16; ++m)
              pv[m] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, mb + m < a.KG ? b0 + (unsigned)(mb + m) * (unsigned)(64 * H * 4) : OOB,
                                                            0, 16 /* sc1 */);
#pragma unroll
            for (int m = 0; m < 16; ++m) {
              const f32x4 p = __builtin_bit_cast(f32x4, pv[m]);
              dh[0] += p[0]; dh[1] += p[1]; dh[2] += p[2]; dh[3] += p[3];
            }
          }
        }
        const float si[4] = {sv_i.x, sv_i.y, sv_i.z, sv_i.w}, sf[4] = {sv_f.x, sv_f.y, sv_f.z, sv_f.w};
        const float sg[4] = {sv_g.x, sv_g.y, sv_g.z, sv_g.w}, so[4] = {sv_o.x, sv_o.y, sv_o.z, sv_o.w};
        const float sc[4] = {sv_c.x, sv_c.y, sv_c.z, sv_c.w}, scp[4] = {sv_cp.x, sv_cp.y, sv_cp.z, sv_cp.w};
        float di[4], df[4], dg_[4], do_[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float tc = fast_tanh(sc[e]);      // (round 6: the forward kernels' v_exp / v_rcp form instead of the library tanhf)
          const float dc = dcarry[e] + dh[e] * so[e] * (1.f - tc * tc);
          di[e] = dc * sg[e] * si[e] * (1.f - si[e]);
          df[e] = dc * scp[e] * sf[e] * (1.f - sf[e]);
          dg_[e] = dc * si[e] * (1.f - sg[e] * sg[e]);
          do_[e] = dh[e] * tc * so[e] * (1.f - so[e]);
          dcarry[e] = dc * sf[e];
        }
        o_i = make_float4(di[0], di[1], di[2], di[3]); o_f = make_float4(df[0], df[1], df[2], df[3]);
        o_g = make_float4(dg_[0], dg_[1], dg_[2], dg_[3]); o_o = make_float4(do_[0], do_[1], do_[2], do_[3]);
        if (ng == 0) {
          const int64_t row = d == 0 ? er0 + t : er0 + eT - 1 - t;
          float* dg = a.dG + row * (8 * H) + d * H4 + j;
          *reinterpret_cast<float4*>(dg) = o_i; *reinterpret_cast<float4*>(dg + H) = o_f;
          *reinterpret_cast<float4*>(dg + 2 * H) = o_g; *reinterpret_cast<float4*>(dg + 3 * H) = o_o;
        }
      }
      {   // every thread owns its 16 entries of the A tile: finished / absent videos contribute zeros
        float* ap = sA + ei * WB_P + 4 * uq;
        *reinterpret_cast<float4*>(ap) = o_i; *reinterpret_cast<float4*>(ap + 32) = o_f;
        *reinterpret_cast<float4*>(ap + 64) = o_g; *reinterpret_cast<float4*>(ap + 96) = o_o;
      }
      if (have_next) { sv_i = nx_i; sv_f = nx_f; sv_g = nx_g; sv_o = nx_o; sv_c = nx_c; sv_cp = nx_cp; sv_dh = nx_dh; }
      __syncthreads();
      if (t > 0) {   // partial_kg(t) is only ever read by step t-1
        float* xo = xbase + gbase + (size_t)(t & 1) * slab + (size_t)kg * 64 * H;
        const int ntile = t < T32 ? 2 : 1;      // rows 32.. hold videos of length <= T32: past that step they publish nothing and nobody reads them
        for (int tile = 0; tile < ntile; ++tile) {
          f32x16 acc;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = 0.f;
          const float* ap = sA + (tile * 32 + li) * WB_P + 4 * lh;
#pragma unroll
          for (int c = 0; c < 16; ++c) {
            const float4 av = *reinterpret_cast<const float4*>(ap + 8 * c);
            const float4 bv = wreg[c];
            // operands swapped (W as the MFMA "A", dG as "B"): the tile comes out transposed, lane = video, and each lane holds
            // 4 x 4 CONSECUTIVE columns -> 16-byte sc1 stores (dword sc1 stores cost ~6x per byte: MI355X_MICROARCH.md)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bv.x, av.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bv.y, av.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bv.z, av.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bv.w, av.w, acc, 0, 0, 0);
          }
          const int i = tile * 32 + li;
          if (t < sT[i]) {
            const unsigned rowoff = (unsigned)((xo - xbase + (size_t)i * H + ng * 256 + wave * 32 + 4 * lh) * 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const u32x4 v = __builtin_bit_cast(u32x4, f32x4{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]});
              if (ng * 256 + wave * 32 + 8 * q + 4 * lh < H)
                __builtin_amdgcn_raw_buffer_store_b128(v, xrsrc, rowoff + 32u * q, 0, 16 /* sc1 */);
            }
          }
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(bar + 32 * (slot % a.n_shards), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// Frame head backward: du = ds*s*(1-s); dh[r,:] = du*w; per-wave partial sums of du*h[r,:] and du (deterministic reduce).
__global__ __launch_bounds__(256) void frame_head_bwd_kernel(const float* __restrict__ h, const float* __restrict__ scores,
                                                             const float* __restrict__ dscores, const float* __restrict__ w,
                                                             float* __restrict__ dh, float* __restrict__ part, int n_rows,
                                                             int F) {
  const int lane = threadIdx.x & 63;
  const int wave_id = blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = gridDim.x * 4;
  const int F4 = F >> 2;
  float* slot = part + (int64_t)wave_id * (F + 4);
  float db = 0.f;
  for (int c = lane; c < F4; c += 64) reinterpret_cast<float4*>(slot)[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int row = wave_id; row < n_rows; row += n_waves) {
    const float sc = scores[row], du = dscores[row] * sc * (1.f - sc);
    db += du;
    for (int c = lane; c < F4; c += 64) {
      float4 hv = reinterpret_cast<const float4*>(h + (int64_t)row * F)[c], ww = reinterpret_cast<const float4*>(w)[c];
      float4 acc = reinterpret_cast<float4*>(slot)[c];
      acc.x += du * hv.x; acc.y += du * hv.y; acc.z += du * hv.z; acc.w += du * hv.w;
      reinterpret_cast<float4*>(slot)[c] = acc;   // each lane re-reads only what it wrote: no cross-lane hazard
      reinterpret_cast<float4*>(dh + (int64_t)row * F)[c] = make_float4(du * ww.x, du * ww.y, du * ww.z, du * ww.w);
    }
  }
  if (lane == 0) slot[F] = db;
}

}  // namespace sumk

using namespace sumk;

// The persistent recurrences are written for the full chip: 256 co-resident blocks, one per CU.  On a partitioned device
// (CPX / fewer CUs) or with SUMK_LSTM_PERSIST=0 the launch-per-step kernels run instead -- same arithmetic.
static bool persistent_kernels_usable() {
  static const bool ok = [] {
    if (getenv("SUMK_LSTM_PERSIST") && getenv("SUMK_LSTM_PERSIST")[0] == '0') return false;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
    return cus >= 256;
  }();
  return ok;
}

extern "C" size_t sumk_bilstm_workspace_bytes(int32_t In, int32_t H, int32_t n_seq, const int32_t* seq_off_host,
                                              int32_t training) {
  LstmWs w;
  if (lstm_carve(In, H, n_seq, seq_off_host, training, &w) != SUMK_OK) return 0;
  return w.total;
}

// weight-plane block of the input projection (sumk_lstm_layer_weights::w_planes): planes of [w_ih[0]; w_ih[1]] (8H x In), then b_ih + b_hh (8H)
static bool bilstm_wplanes_ok(int In, int H, int np) {
  return H > 0 && (8 * H) % 256 == 0 && In % 32 == 0 && In >= 128 && (np == 2 || np == 3) && pw_ok(1024, 8 * (int64_t)H, In, 1024, 8 * (int64_t)H, np);
}
__global__ void bilstm_bias_sum_kernel(const float* b0, const float* b1, const float* b2, const float* b3, int H4, float* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < H4) { out[i] = b0[i] + b1[i]; out[H4 + i] = b2[i] + b3[i]; }
}
extern "C" size_t sumk_bilstm_wplanes_bytes(int32_t In, int32_t H, int32_t n_planes) {
  if (!bilstm_wplanes_ok(In, H, n_planes)) return 0;
  return align_up(pw_planes_bytes(8 * (int64_t)H, In, n_planes), 256) + (size_t)8 * H * 4;
}
extern "C" int sumk_bilstm_wplanes_build(int32_t In, int32_t H, const sumk_lstm_layer_weights* w, int32_t n_planes, void* out, size_t out_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SUMK_ARG(bilstm_wplanes_ok(In, H, n_planes), "bilstm_wplanes_build: In=%d H=%d planes=%d is not eligible (8H %% 256, In %% 32, In >= 128)", In, H, n_planes);
  SUMK_ARG(w && out && w->w_ih[0] && w->w_ih[1] && w->b_ih[0] && w->b_ih[1] && w->b_hh[0] && w->b_hh[1], "bilstm_wplanes_build: null pointer");
  const size_t pb = align_up(pw_planes_bytes(8 * (int64_t)H, In, n_planes), 256);
  SUMK_ARG(out_bytes >= pb + (size_t)8 * H * 4 && ((uintptr_t)out & 255) == 0, "bilstm_wplanes_build: buffer too small or misaligned");
  SUMK_HIP(hipMemsetAsync(out, 0, pb, stream));
  for (int d = 0; d < 2; ++d) SUMK_TRY(split_planes_at(w->w_ih[d], 4 * H, In, In, n_planes, out, (int64_t)d * 4 * H, 8 * (int64_t)H, stream));
  hipLaunchKernelGGL(bilstm_bias_sum_kernel, dim3((4 * H + 255) / 256), dim3(256), 0, stream, w->b_ih[0], w->b_hh[0], w->b_ih[1], w->b_hh[1], 4 * H,
                     (float*)((char*)out + pb));
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}

extern "C" int sumk_bilstm_layer_forward(const float* x, int32_t In, int32_t H, int32_t n_seq,
                                         const int32_t* seq_off_host, const int32_t* seq_off_dev,
                                         const sumk_lstm_layer_weights* w, float* h_out, void* workspace,
                                         size_t workspace_bytes, int32_t training, int32_t precision, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SUMK_ARG(x && seq_off_dev && w && h_out && workspace, "bilstm_forward: null pointer");
  SUMK_ARG(precision >= SUMK_PRECISION_FP32 && precision <= SUMK_PRECISION_MAX, "bilstm_forward: unknown precision %d", precision);
  for (int d = 0; d < 2; ++d)
    SUMK_ARG(w->w_ih[d] && w->w_hh[d] && w->b_ih[d] && w->b_hh[d], "bilstm_forward: null weight (dir %d)", d);
  LstmWs L;
  SUMK_TRY(lstm_carve(In, H, n_seq, seq_off_host, training, &L));
  if (workspace_bytes < L.total) {
    set_error("bilstm_forward: workspace %zu < required %zu", workspace_bytes, L.total);
    return SUMK_ERR_WORKSPACE;
  }
  char* ws = (char*)workspace;
  const int R = L.n_rows;
  float* G = (float*)(ws + L.g);
  GemmProb* prob = (GemmProb*)(ws + L.prob);

  // 1: input projection for both directions, biases fused
  const int small = gemm_tiles(R, 8 * H, 0) >= 512 ? 0 : 1;
  const int np_in = precision == SUMK_PRECISION_BF16X6 ? 3 : precision == SUMK_PRECISION_BF16X3 ? 2 : 0;
  static const bool persist_ok = persistent_kernels_usable();
  const bool planes_in = !training && np_in && w->x_planes && w->w_planes && bilstm_wplanes_ok(In, H, np_in) && R >= 1024 && pw_ok(R, 8 * (int64_t)H, In, R, 8 * (int64_t)H, np_in);
  // round 6: with operand planes at hand and the H <= 256 persistent recurrence on 16-row MFMAs, the projection CAN run inside the recurrence
  // (lstm_persist_proj_kernel): no G, no GEMM launch.  It was the default while the plain recurrence cost 2.55 us per step (1.065 against 1.09 ms at three
  // planes); with the branch-free tag checks the plain step is 2.05 us and the plane GEMM in front of it wins (DSN scoring bf16x6 0.94-0.96 against 1.04-1.05
  // ms, bf16x3 0.84-0.88 against 0.88-0.89: the projection costs 1.07 us per step inside, 0.23 ms as a launch) -- SUMK_LSTM_PROJ=1 selects the fused form
  // (A/B: tests/test_gpu_lstm.py)
  {
    static const bool proj_on = getenv("SUMK_LSTM_PROJ") && getenv("SUMK_LSTM_PROJ")[0] == '1';
    static const bool ll_on = !(getenv("SUMK_LSTM_LL") && getenv("SUMK_LSTM_LL")[0] == '0');
    static const bool m16_on = !(getenv("SUMK_LSTM_M16") && getenv("SUMK_LSTM_M16")[0] == '0');
    const int gsize = std::min(32, std::max(1, (2 * n_seq + PK_TEAMS - 1) / PK_TEAMS));
    const int n_groups = (n_seq + gsize - 1) / gsize;
    if (planes_in && proj_on && ll_on && m16_on && persist_ok && H <= 256 && In == 1024 && gsize <= 16 && (size_t)R * 2 * H * 4 < 0x7fffffff &&
        L.ll_bytes > 0 && L.ll_bytes < 0x7fffffe0 && 2 * n_groups <= PSTATE_WORDS - 16 && 16 * (uint64_t)pw_rows_pitch(R) < 0xffffffffull) {
      const size_t words = (L.ll + L.ll_bytes - L.pstate) / 4;
      hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)std::min<size_t>(64, (words + 255) / 256)), dim3(256), 0, stream, (unsigned*)(ws + L.pstate), (int)words);
      PersistProjArgs pa;
      pa.x = x; pa.In = In; pa.wpl = (const char*)w->w_planes;
      pa.bias = (const float*)((const char*)w->w_planes + align_up(pw_planes_bytes(8 * (int64_t)H, In, np_in), 256));
      pa.whh[0] = w->w_hh[0]; pa.whh[1] = w->w_hh[1]; pa.Hout = h_out;
      pa.off = seq_off_dev; pa.state = (unsigned*)(ws + L.pstate);
      pa.ll = (unsigned long long*)(ws + L.ll); pa.ll_bytes = (int32_t)L.ll_bytes;
      pa.n_seq = n_seq; pa.H = H; pa.gsize = gsize; pa.n_groups = n_groups; pa.n_teams = PK_TEAMS;
      pa.upm = std::min(8, (H + 31) / 32); pa.n_active = (H + pa.upm - 1) / pa.upm;
      pa.w_rp16 = (uint32_t)(16 * pw_rows_pitch(8 * (int64_t)H));
      const void* fn = np_in == 3 ? (const void*)lstm_persist_proj_kernel<3, PROJ_KA> : (const void*)lstm_persist_proj_kernel<2, PROJ_KA>;
      static bool pp_attr_set[2] = {false, false};
      if (!pp_attr_set[np_in - 2]) {
        SUMK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        pp_attr_set[np_in - 2] = true;
      }
      void* kargs[] = {&pa};
      prof_begin(SUMK_PROF_LSTM_REC, stream);
      // LDS: 33 KB of partial tiles and video tables + the W_ih fragments beyond the PROJ_RF a lane keeps in registers; > 80 KB keeps one block per CU
      const int frag_a = PROJ_KA * 2 * np_in, frag_b = (8 - PROJ_KA) * 2 * np_in;
      const size_t shmem = std::max<size_t>(96 * 1024, 34 * 1024 + (size_t)(std::max(frag_a - PROJ_RF, 0) + std::max(frag_b - PROJ_RF, 0)) * 256 * 16);
      SUMK_HIP(hipLaunchCooperativeKernel(fn, dim3(256), dim3(PK_THREADS), kargs, (unsigned)shmem, stream));
      prof_end(SUMK_PROF_LSTM_REC, stream);
      return SUMK_OK;
    }
  }
  if (planes_in) {
    // operand planes (x per dataset / per call, weights per weight change): the plane-aware wide GEMM, no split in the k-loop
    PwLaunch g; g.A = w->x_planes; g.a_rows = R; g.B = w->w_planes; g.b_rows = 8 * (int64_t)H; g.M = R; g.N = 8 * H; g.K = In; g.np = np_in;
    g.C = G; g.ldc = 8 * H; g.bias = (const float*)((const char*)w->w_planes + align_up(pw_planes_bytes(8 * (int64_t)H, In, np_in), 256));
    SUMK_TRY(launch_gemm_pw(PW_F32, g, stream));
  } else {
  SUMK_TRY(fill_single_prob(prob, R, 8 * H, In, In, In, 8 * H, 0, small, stream));
  {
    GemmLaunch g;
    g.A = x; g.B[0] = w->w_ih[0]; g.B[1] = w->w_ih[1]; g.n_group = 4 * H;
    g.bias0[0] = w->b_ih[0]; g.bias0[1] = w->b_ih[1]; g.bias1[0] = w->b_hh[0]; g.bias1[1] = w->b_hh[1];
    g.C = G; g.probs = prob; g.small_tile = small; g.total_tiles = gemm_tiles(R, 8 * H, small);
    g.precision = precision; g.lean = gemm_lean_ok(R, 4 * H, In, In, In);
    SUMK_TRY(launch_gemm(GEMM_NT, EPI_BIAS2, g, stream));
  }
  }
  // 2: recurrence.  The health word is cleared on BOTH paths so sumk_bilstm_check never reads stale workspace bytes.
  // (the flag-in-data hand-off buffer lies directly behind the state words: one launch clears both -- a tag of 0 matches no step)
  {
    const size_t words = (L.ll + L.ll_bytes - L.pstate) / 4;
    hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)std::min<size_t>(64, (words + 255) / 256)), dim3(256), 0, stream, (unsigned*)(ws + L.pstate), (int)words);
  }
  // H <= 256: 8 XCD teams with an LDS panel; 256 < H <= 1024: the two-team register-resident kernel; otherwise the launch chain
  if (persist_ok && H <= 256 && (size_t)R * 2 * H * 4 < 0x7fffffff) {
    PersistArgs pa;
    pa.G = G; pa.whh[0] = w->w_hh[0]; pa.whh[1] = w->w_hh[1]; pa.Hout = h_out;
    pa.gates = training ? (float*)(ws + L.gates) : nullptr;
    pa.c_all = training ? (float*)(ws + L.call) : nullptr;
    pa.hprev = training ? (float*)(ws + L.hprev) : nullptr;
    pa.off = seq_off_dev; pa.state = (unsigned*)(ws + L.pstate);
    pa.n_seq = n_seq; pa.H = H; pa.hout_bytes = (int32_t)std::min<size_t>((size_t)R * 2 * H * 4, 0x7fffffff);
    pa.ll = (unsigned long long*)(ws + L.ll); pa.ll_bytes = (int32_t)std::min<size_t>(L.ll_bytes, 0x7fffffe0);
    // H <= 256: 8 teams of 32 CUs (one XCD each); larger H: 2 teams of 128 CUs so every member still owns only 8 units
    pa.n_teams = H <= 256 ? PK_TEAMS : 2;
    const int team_size = 256 / pa.n_teams;
    pa.upm = std::min(8, (H + team_size - 1) / team_size); pa.n_active = (H + pa.upm - 1) / pa.upm;
    const int gmax = H <= 256 ? 32 : 24;       // videos per work item, bounded by the LDS panel gsize x (H+4) floats
    int gsize = std::min(gmax, std::max(1, (2 * n_seq + pa.n_teams - 1) / pa.n_teams));
    pa.gsize = gsize; pa.n_groups = (n_seq + gsize - 1) / gsize;
    if (2 * pa.n_groups <= PSTATE_WORDS - 16 && pa.n_active <= team_size) {
      const size_t shmem = std::max<size_t>(((size_t)gsize * (H + 4) + 8 * 32 * 32 + 96) * sizeof(float), 96 * 1024);  // >80 KB: one block per CU
      static const bool direct = !(SUMK_TUNE_ENV("SUMK_LSTM_PANEL") && SUMK_TUNE_ENV("SUMK_LSTM_PANEL")[0] == '1');   // 1: stage h through LDS
      // SUMK_LSTM_LL=0: the counter hand-off (A/B switch); the flag-in-data one needs the exchange buffer's byte offsets in 31 bits
      static const bool ll_on = !(getenv("SUMK_LSTM_LL") && getenv("SUMK_LSTM_LL")[0] == '0');
      const bool ll = direct && ll_on && L.ll_bytes > 0 && L.ll_bytes < 0x7fffffe0;
      static const bool m16_on = !(getenv("SUMK_LSTM_M16") && getenv("SUMK_LSTM_M16")[0] == '0');
      const bool m16 = ll && m16_on && gsize <= 16;
      const int which = m16 ? 3 : ll ? 2 : direct ? 1 : 0;
      const void* fn = m16 ? (const void*)lstm_persist_kernel<4, true, true, true>
                     : ll ? (const void*)lstm_persist_kernel<4, true, true>
#ifdef SUMK_DIAG
                          : direct ? (const void*)lstm_persist_kernel<4, true> : (const void*)lstm_persist_kernel<4, false>;   // (LDS-panel staging: measured slower, diagnostic build only)
#else
                          : (const void*)lstm_persist_kernel<4, true>;
#endif
      static bool attr_set[4] = {false, false, false, false};
      if (!attr_set[which]) {
        SUMK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[which] = true;
      }
      void* kargs[] = {&pa};
      prof_begin(SUMK_PROF_LSTM_REC, stream);
      SUMK_HIP(hipLaunchCooperativeKernel(fn, dim3(256), dim3(PK_THREADS), kargs, (unsigned)shmem, stream));
      prof_end(SUMK_PROF_LSTM_REC, stream);
#ifdef SUMK_DIAG
      if (getenv("SUMK_LSTM_STAMPS") && m16) {
        unsigned long long q[12 * 2 * 8];
        SUMK_HIP(hipStreamSynchronize(stream));
        SUMK_HIP(hipMemcpy(q, (const char*)(ws + L.pstate) + 600 * 4, sizeof(q), hipMemcpyDeviceToHost));
        for (int b = 0; b < 12; b += 5)
          for (int wv = 0; wv < 2; ++wv) {
            const unsigned long long* e = q + (b * 2 + wv) * 8;
            fprintf(stderr, "[lstm stamps] block %d wave %d: total %llu  wait %llu (spins %llu)  mfma+part %llu  barrier1 %llu  epilogue %llu  barrier2 %llu\n",
                    b, wv ? 7 : 0, e[0], e[1], e[6], e[2], e[3], e[4], e[5]);
          }
      }
#endif
      return SUMK_OK;
    }
  }
  if (persist_ok && H > 256 && H <= 64 * WK_CPW && (size_t)R * 2 * H * 4 < 0x7fffffe0 &&
      2 * ((n_seq + WK_GROUP - 1) / WK_GROUP) <= PSTATE_WORDS - 16) {
    WideArgs wa;
    wa.G = G; wa.whh[0] = w->w_hh[0]; wa.whh[1] = w->w_hh[1]; wa.Hout = h_out;
    wa.gates = training ? (float*)(ws + L.gates) : nullptr;
    wa.c_all = training ? (float*)(ws + L.call) : nullptr;
    wa.hprev = training ? (float*)(ws + L.hprev) : nullptr;
    wa.off = seq_off_dev; wa.state = (unsigned*)(ws + L.pstate);
    wa.n_seq = n_seq; wa.H = H; wa.hout_bytes = (int32_t)((size_t)R * 2 * H * 4);
    wa.n_groups = (n_seq + WK_GROUP - 1) / WK_GROUP;
    wa.upm = (H + 127) / 128; wa.n_active = (H + wa.upm - 1) / wa.upm;     // <= 8 units per member, <= 128 members per team
    // round 6: the re-partitioned form (exchange buffer laid out for the consumers, rows sorted by length, sharded counter, bf16x6 mode);
    // SUMK_LSTM_WIDE2=0 keeps lstm_wide_kernel (A/B switch: tests/test_gpu_lstm.py::test_wide_recurrence_forms_agree)
    static const bool wide2_on = !(getenv("SUMK_LSTM_WIDE2") && getenv("SUMK_LSTM_WIDE2")[0] == '0');
    if (wide2_on && L.wx_bytes > 0 && 2 * wa.n_groups * W2_ITEM_WORDS <= PSTATE_WORDS - 16 - 512) {
      Wide2Args w2;
      w2.G = wa.G; w2.whh[0] = wa.whh[0]; w2.whh[1] = wa.whh[1]; w2.Hout = h_out;
      w2.gates = wa.gates; w2.c_all = wa.c_all; w2.hprev = wa.hprev; w2.off = wa.off; w2.state = wa.state;
      w2.xchg = (float*)(ws + L.wx);
      w2.n_seq = n_seq; w2.H = H; w2.n_groups = wa.n_groups; w2.upm = wa.upm; w2.n_active = wa.n_active;
      w2.pack16 = (wa.upm == 8 && wa.n_active * 8 == H) ? 1 : 0;
      // the recurrent product follows the layer's arithmetic: exact fp32 MFMA, bf16x3 (hi / lo) or the fp32-grade bf16x6 (three planes);
      // training keeps exact fp32 in the bf16x6 mode (the BPTT multiplies in fp32)
      const int mode = precision == SUMK_PRECISION_BF16X3 ? 1 : (precision == SUMK_PRECISION_BF16X6 && !training) ? 2 : 0;
      const void* fn2 = mode == 2 ? (const void*)lstm_wide2_kernel<2> : mode == 1 ? (const void*)lstm_wide2_kernel<1> : (const void*)lstm_wide2_kernel<0>;
      static bool w2_attr_set[3] = {false, false, false};
      if (!w2_attr_set[mode]) {
        SUMK_HIP(hipFuncSetAttribute(fn2, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        w2_attr_set[mode] = true;
      }
      const size_t shmem2 = mode == 2 ? (size_t)(64 + 1 + 64) * 1024 : (size_t)96 * 1024;   // 65 KB (+ 64 KB of W_hh plane 3); > 80 KB keeps one block per CU
      void* kargs2[] = {&w2};
      prof_begin(SUMK_PROF_LSTM_REC, stream);
      SUMK_HIP(hipLaunchCooperativeKernel(fn2, dim3(256), dim3(PK_THREADS), kargs2, (unsigned)shmem2, stream));
      prof_end(SUMK_PROF_LSTM_REC, stream);
#ifdef SUMK_DIAG
      if (getenv("SUMK_LSTM_STAMPS")) {
        unsigned long long q[12 * 2 * 8];
        SUMK_HIP(hipStreamSynchronize(stream));
        SUMK_HIP(hipMemcpy(q, (const char*)(ws + L.pstate) + W2_STAMP_WORD * 4, sizeof(q), hipMemcpyDeviceToHost));
        for (int b = 0; b < 12; b += 5)
          for (int wv = 0; wv < 2; ++wv) {
            const unsigned long long* e = q + (b * 2 + wv) * 8;
            const double n = e[6] ? (double)e[6] : 1.0;
            fprintf(stderr, "[wide2 stamps] mode %d block %d wave %d: steps %llu (two tiles in %llu)  cycles/step: total %.0f = poll + barrier %.0f + loads, MFMAs, partial tiles %.0f + "
                            "barrier %.0f + cell update, exchange stores %.0f + drain, barrier, signal %.0f\n",
                    mode, b, wv ? 7 : 0, e[6], e[7], (double)e[0] / n, e[1] / n, e[2] / n, e[3] / n, e[4] / n, e[5] / n);
          }
      }
#endif
      return SUMK_OK;
    }
    const size_t shmem = 96 * 1024;   // 68 KB used; > 80 KB keeps one block per CU
    const bool x3 = precision == SUMK_PRECISION_BF16X3;
    const void* fn = x3 ? (const void*)lstm_wide_kernel<true> : (const void*)lstm_wide_kernel<false>;
    static bool wide_attr_set[2] = {false, false};
    if (!wide_attr_set[x3]) {
      SUMK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      wide_attr_set[x3] = true;
    }
    void* kargs[] = {&wa};
    prof_begin(SUMK_PROF_LSTM_REC, stream);
    SUMK_HIP(hipLaunchCooperativeKernel(fn, dim3(256), dim3(PK_THREADS), kargs, (unsigned)shmem, stream));
    prof_end(SUMK_PROF_LSTM_REC, stream);
    return SUMK_OK;
  }
  StepArgs a;
  a.G = G; a.whh[0] = w->w_hh[0]; a.whh[1] = w->w_hh[1]; a.Hout = h_out;
  a.cstate = training ? nullptr : (float*)(ws + L.cstate);
  a.gates = training ? (float*)(ws + L.gates) : nullptr;
  a.c_all = training ? (float*)(ws + L.call) : nullptr;
  a.hprev = training ? (float*)(ws + L.hprev) : nullptr;
  a.off = seq_off_dev; a.n_seq = n_seq; a.H = H; a.n_ublk = (H + 7) / 8;
  const int n_mtiles = (n_seq + 31) / 32;
  const dim3 grid((unsigned)(n_mtiles * a.n_ublk * 2)), block(256);
  prof_begin(SUMK_PROF_LSTM_REC, stream);
  for (int t = 0; t < L.t_max; ++t) {
    a.t = t;
    hipLaunchKernelGGL(lstm_step_kernel, grid, block, 0, stream, a);
  }
  prof_end(SUMK_PROF_LSTM_REC, stream);
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}

extern "C" int sumk_frame_head_forward(const float* h, int32_t n_rows, int32_t F, const float* w, const float* b,
                                       float* scores, void* stream) {
  SUMK_ARG(h && w && b && scores, "frame_head_forward: null pointer");
  SUMK_ARG(n_rows > 0 && F > 0 && F % 4 == 0, "frame_head_forward: bad shape n_rows=%d F=%d", n_rows, F);
  hipLaunchKernelGGL(frame_head_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, h, w, b, scores,
                     n_rows, F);
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}

extern "C" int sumk_bilstm_layer_backward(const float* x, const float* h_out, const float* dh_out, int32_t In, int32_t H,
                                          int32_t n_seq, const int32_t* seq_off_host, const int32_t* seq_off_dev,
                                          const sumk_lstm_layer_weights* w, const sumk_lstm_layer_grads* gr, float* dx,
                                          void* workspace, size_t workspace_bytes, int32_t precision, void* stream_) {
  SUMK_ARG(precision >= SUMK_PRECISION_FP32 && precision <= SUMK_PRECISION_MAX, "bilstm_backward: unknown precision %d", precision);
  hipStream_t stream = (hipStream_t)stream_;
  SUMK_ARG(x && h_out && dh_out && seq_off_dev && w && gr && workspace, "bilstm_backward: null pointer");
  for (int d = 0; d < 2; ++d)
    SUMK_ARG(w->w_ih[d] && w->w_hh[d] && gr->w_ih[d] && gr->w_hh[d] && gr->b_ih[d] && gr->b_hh[d],
             "bilstm_backward: null weight/grad (dir %d)", d);
  LstmWs L;
  SUMK_TRY(lstm_carve(In, H, n_seq, seq_off_host, 1, &L));
  if (workspace_bytes < L.total) {
    set_error("bilstm_backward: workspace %zu < required %zu (needs the training-mode forward's workspace)", workspace_bytes, L.total);
    return SUMK_ERR_WORKSPACE;
  }
  char* ws = (char*)workspace;
  const int R = L.n_rows;
  float* dG = (float*)(ws + L.dg);
  float* hprev = (float*)(ws + L.hprev);
  float* slab = (float*)(ws + L.slab);
  float* colpart = (float*)(ws + L.colpart);
  GemmProb* prob = (GemmProb*)(ws + L.prob);
  GemmProb* psk = (GemmProb*)(ws + L.prob_sk);

  static const bool persist_ok = persistent_kernels_usable();
  {   // state words + (flag-in-data hand-off only: directly behind them) the exchange buffer: a tag of 0 matches no step
    static const bool ll_bwd = getenv("SUMK_LSTM_LL_BWD") && getenv("SUMK_LSTM_LL_BWD")[0] == '1';
    const size_t words = ll_bwd ? (L.llb + L.llb_bytes - L.pstate_b) / 4 : (size_t)PSTATE_WORDS;
    hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)std::min<size_t>(256, (words + 255) / 256)), dim3(256), 0, stream, (unsigned*)(ws + L.pstate_b), (int)words);
  }
  bool done = false;
  if (persist_ok && H <= 256 && L.xchg_bytes > 0) {
    PersistBwdArgs pa;
    pa.whh[0] = w->w_hh[0]; pa.whh[1] = w->w_hh[1]; pa.dHout = dh_out; pa.gates = (const float*)(ws + L.gates);
    pa.c_all = (const float*)(ws + L.call); pa.dG = dG; pa.xchg = (float*)(ws + L.xchg);
    pa.off = seq_off_dev; pa.state = (unsigned*)(ws + L.pstate_b);
    pa.n_seq = n_seq; pa.H = H;
    pa.upm = std::min(8, (H + 31) / 32); pa.n_active = (H + pa.upm - 1) / pa.upm;
    int gsize = std::min(32, std::max(1, (2 * n_seq + PK_TEAMS - 1) / PK_TEAMS));
    pa.gsize = gsize; pa.n_groups = (n_seq + gsize - 1) / gsize; pa.n_items = 2 * pa.n_groups;
    // round 6: sharded step counter (while the items fit the state block; the last 512 words hold the diagnostic stamps) and the reader-shaped
    // exchange (members own 8 aligned units); SUMK_LSTM_BWD_R6=0 keeps the round-5 forms (A/B: tests/test_gpu_lstm.py)
    static const bool r6_on = !(getenv("SUMK_LSTM_BWD_R6") && getenv("SUMK_LSTM_BWD_R6")[0] == '0');
    const bool shard = r6_on && pa.n_items * 128 <= PSTATE_WORDS - 16 - 512;
    pa.item_words = shard ? 128 : 1; pa.n_shards = shard ? 4 : 1;
    pa.coal = (r6_on && pa.upm == 8 && pa.n_active * 8 == H) ? 1 : 0;
    if (pa.n_items <= PSTATE_WORDS - 16) {
      const size_t shmem = std::max<size_t>(((size_t)32 * (H + 4) + 32 * 36 + 96) * sizeof(float), 96 * 1024);  // >80 KB: one block per CU
      pa.ll = (unsigned long long*)(ws + L.llb);
      // The BPTT keeps the COUNTER hand-off by default: its exchange is a reduce-scatter of 32 partials per element (each epilogue thread
      // polls 32 packets, the exchange bytes double), and the flag-in-data form measured SLOWER here (DSN training step 3.93 vs 3.76 ms);
      // SUMK_LSTM_LL_BWD=1 selects it.  The 16-row MFMA form (groups of <= 16 videos) is independent of the hand-off.
      static const bool ll_on = getenv("SUMK_LSTM_LL_BWD") && getenv("SUMK_LSTM_LL_BWD")[0] == '1';
      static const bool m16_on = !(getenv("SUMK_LSTM_M16") && getenv("SUMK_LSTM_M16")[0] == '0');
      const bool ll = ll_on && L.llb_bytes > 0, m16 = m16_on && gsize <= 16;
      const int which = (ll ? 2 : 0) + (m16 ? 1 : 0);
      const void* fn = ll ? (m16 ? (const void*)lstm_persist_bwd_kernel<true, true> : (const void*)lstm_persist_bwd_kernel<true, false>)
                          : (m16 ? (const void*)lstm_persist_bwd_kernel<false, true> : (const void*)lstm_persist_bwd_kernel<false, false>);
      static bool attr_set[4] = {false, false, false, false};
      if (!attr_set[which]) {
        SUMK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[which] = true;
      }
      void* kargs[] = {&pa};
      SUMK_HIP(hipLaunchCooperativeKernel(fn, dim3(PK_TEAMS * 32), dim3(PK_THREADS), kargs, (unsigned)shmem, stream));
      done = true;
#ifdef SUMK_DIAG
      if (getenv("SUMK_LSTM_STAMPS") && !ll && m16) {      // phase table of the BPTT step (scripts/probes: profiles/r04_lstm_bptt_phase_stamps.txt)
        unsigned long long q[12 * 2 * 8];
        SUMK_HIP(hipStreamSynchronize(stream));
        SUMK_HIP(hipMemcpy(q, (const char*)pa.state + 600 * 4, sizeof(q), hipMemcpyDeviceToHost));
        for (int b = 0; b < 12; b += 5)
          for (int wv = 0; wv < 2; ++wv) {
            const unsigned long long* e = q + (b * 2 + wv) * 8;
            const double n = e[6] ? (double)e[6] : 1.0;
            fprintf(stderr, "[lstm bptt stamps] member %d wave %d: steps %llu  cycles/step: total %.0f = wait for the counter %.0f + partial sums, cell backward, dG stores %.0f + "
                            "barrier %.0f + MFMAs, partial stores %.0f + drain, barrier, signal %.0f\n",
                    b, wv ? 7 : 0, e[6], (double)(e[1] + e[2] + e[3] + e[4] + e[5]) / n, e[1] / n, e[2] / n, e[3] / n, e[4] / n, e[5] / n);
          }
      }
#endif
    }
  }
  static const bool wide_bwd = !(SUMK_TUNE_ENV("SUMK_LSTM_WIDE_BWD") && SUMK_TUNE_ENV("SUMK_LSTM_WIDE_BWD")[0] == '0');
  if (!done && persist_ok && wide_bwd && n_seq > GV_MAXB && H > 256 && H <= 1024 && H % 128 == 0 && L.xchg_bytes > 0 &&
      2 * ((n_seq + WK_GROUP - 1) / WK_GROUP) <= PSTATE_WORDS - 16) {
    WideBwdArgs wa;
    wa.whh[0] = w->w_hh[0]; wa.whh[1] = w->w_hh[1]; wa.dHout = dh_out; wa.gates = (const float*)(ws + L.gates);
    wa.c_all = (const float*)(ws + L.call); wa.dG = dG; wa.xchg = (float*)(ws + L.xchg);
    wa.off = seq_off_dev; wa.state = (unsigned*)(ws + L.pstate_b);
    wa.n_seq = n_seq; wa.H = H; wa.n_groups = (n_seq + WK_GROUP - 1) / WK_GROUP;
    wa.KG = H / 32; wa.NG = (H + 255) / 256; wa.n_active = wa.KG * wa.NG;
    wa.xchg_dir_bytes = (int32_t)(L.xchg_bytes / 2);
    {   // round 6: sharded step counter while the items fit the state block (SUMK_LSTM_BWD_R6=0: one word per item, as round 5)
      static const bool r6_on = !(getenv("SUMK_LSTM_BWD_R6") && getenv("SUMK_LSTM_BWD_R6")[0] == '0');
      const bool shard = r6_on && 2 * wa.n_groups * 128 <= PSTATE_WORDS - 16 - 512;
      wa.item_words = shard ? 128 : 1; wa.n_shards = shard ? 4 : 1;
    }
    static bool wb_attr_set = false;
    if (!wb_attr_set) {
      SUMK_HIP(hipFuncSetAttribute((const void*)lstm_wide_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      wb_attr_set = true;
    }
    void* kargs[] = {&wa};
    SUMK_HIP(hipLaunchCooperativeKernel((const void*)lstm_wide_bwd_kernel, dim3(256), dim3(PK_THREADS), kargs,
                                        (unsigned)(96 * 1024), stream));   // 35 KB used; > 80 KB keeps one block per CU
    done = true;
  }
  if (!done && n_seq <= GV_MAXB) {   // a few videos and no persistent BPTT (H > 256): bandwidth-shaped transposed mat-vec per step
    const TGemvGeom tg = tgemv_geom(H);
    TGemvArgs ta;
    ta.dGa = dG; ta.wa = w->w_hh[0]; ta.wa1 = w->w_hh[1]; ta.dGb = nullptr; ta.wb = nullptr; ta.b_shift = 0; ta.nd = 2;
    ta.partial = (float*)(ws + L.partial); ta.off = seq_off_dev; ta.n_seq = n_seq; ta.H = H; ta.rows_per_split = tg.rows_per_split;
    CellBwdArgs ca;
    ca.dext = dh_out; ca.dh_last = ca.dc_last = nullptr; ca.partial = ta.partial; ca.n_ksplit = tg.n_ksplit; ca.nd = 2;
    ca.gates = (const float*)(ws + L.gates); ca.c_all = (const float*)(ws + L.call); ca.c0 = nullptr;
    ca.dG = dG; ca.dcstate = (float*)(ws + L.dcstate); ca.dh0 = ca.dc0 = nullptr;
    ca.off = seq_off_dev; ca.n_seq = n_seq; ca.H = H;
    for (int t = L.t_max - 1; t >= 0; --t) {
      ta.t = ca.t = t;
      hipLaunchKernelGGL(lstm_tgemv_partial_kernel, dim3((unsigned)tg.strips, (unsigned)tg.n_ksplit, 2), dim3(256), 0, stream, ta);
      hipLaunchKernelGGL(lstm_cellbwd_kernel, dim3((unsigned)((H + 63) / 64), (unsigned)n_seq, 2), dim3(256), 0, stream, ca);
    }
    SUMK_HIP(hipGetLastError());
    done = true;
  }
  if (!done) {
    BwdStepArgs a;
    a.whh[0] = w->w_hh[0]; a.whh[1] = w->w_hh[1]; a.dHout = dh_out; a.gates = (const float*)(ws + L.gates);
    a.c_all = (const float*)(ws + L.call); a.dG = dG; a.dcstate = (float*)(ws + L.dcstate);
    a.off = seq_off_dev; a.n_seq = n_seq; a.H = H; a.n_jblk = (H + 31) / 32;
    const int n_mtiles = (n_seq + 31) / 32;
    const dim3 grid((unsigned)(n_mtiles * a.n_jblk * 2)), block(512);
    for (int t = L.t_max - 1; t >= 0; --t) {
      a.t = t;
      hipLaunchKernelGGL(lstm_bwd_step_kernel, grid, block, 0, stream, a);
    }
    SUMK_HIP(hipGetLastError());
  }
  // weight gradients: dW_ih[d] += dG_d^T X (both directions in one split-K launch), dW_hh[d] += dG_d^T h_prev_d
  auto dw_hh = [&](int d) -> int {
    float* out[4] = {gr->w_hh[d], nullptr, nullptr, nullptr};
    return gemm_tn_splitk_accum(dG + (size_t)d * 4 * H, 8 * H, hprev + (size_t)d * H, 2 * H, 4 * H, H, R, slab, L.slab_elems,
                                psk, 64, out, 4 * H, H, 1.f, stream, precision);
  };
  auto bias_grads = [&]() -> int {   // b_ih and b_hh of a direction both receive the column sums of its half of dG: one pass over dG for all four
    const ReduceSeg segs[4] = {{0, 4 * H, gr->b_ih[0]}, {0, 4 * H, gr->b_hh[0]}, {4 * H, 4 * H, gr->b_ih[1]}, {4 * H, 4 * H, gr->b_hh[1]}};
    return colsum_multi(dG, 8 * H, R, 8 * H, colpart, 128, segs, 4, stream);
  };
  if (gr->tail_ready_event) {
    // data-parallel overlap: the biases and the REVERSE direction first, then the event -- the caller's all-reduce of [reverse | head] runs
    // on a side stream under the forward direction's GEMMs (sumk.h: sumk_lstm_layer_grads::tail_ready_event)
    SUMK_TRY(bias_grads());
    for (int d = 1; d >= 0; --d) {
      float* out[4] = {gr->w_ih[d], nullptr, nullptr, nullptr};
      SUMK_TRY(gemm_tn_splitk_accum(dG + (size_t)d * 4 * H, 8 * H, x, In, 4 * H, In, R, slab, L.slab_elems, psk, 64, out, 4 * H, In, 1.f, stream, precision));
      SUMK_TRY(dw_hh(d));
      if (d == 1) SUMK_HIP(hipEventRecord((hipEvent_t)gr->tail_ready_event, stream));
    }
  } else {
    {
      float* out[4] = {gr->w_ih[0], gr->w_ih[1], nullptr, nullptr};
      SUMK_TRY(gemm_tn_splitk_accum(dG, 8 * H, x, In, 8 * H, In, R, slab, L.slab_elems, psk, 64, out, 4 * H, In, 1.f, stream, precision));
    }
    for (int d = 0; d < 2; ++d) SUMK_TRY(dw_hh(d));
    SUMK_TRY(bias_grads());
  }
  if (dx) {  // dX = dG_fwd W_ih_fwd + dG_rev W_ih_rev
    const int small = gemm_tiles(R, In, 0) >= 512 ? 0 : 1;
    SUMK_TRY(fill_single_prob(prob + 1, R, In, 4 * H, 8 * H, In, In, 0, small, stream));
    for (int d = 0; d < 2; ++d) {
      GemmLaunch g;
      g.A = dG + (size_t)d * 4 * H; g.B[0] = w->w_ih[d]; g.C = dx; g.probs = prob + 1; g.small_tile = small;
      g.total_tiles = gemm_tiles(R, In, small); g.precision = precision;
      SUMK_TRY(launch_gemm(GEMM_NN, d == 0 ? EPI_NONE : EPI_ACCUM, g, stream));
    }
  }
  return SUMK_OK;
}

// ------------------------------------------------------------------------------------------- unidirectional layer
// One forward-running LSTM direction with an optional initial state and the final state as an output -- the layers of
// SumGAN's eLSTM / dLSTM / cLSTM (summarizer/models/sumgan.py:48-115,185-210: nn.LSTM(bidirectional=False), `(h_0, c_0)`
// inputs, `(h_n, c_n)` outputs).  Runs the launch-per-step kernels with nd = 1 (row layouts (R, H) / (R, 4H)).
struct Lstm1Ws {
  size_t g, cstate, prob, gates, call, hprev, dg, dcstate, slab, prob_sk, colpart, partial, total;
  size_t slab_elems;
  int32_t n_rows, t_max;
};

static int lstm1_carve(int In, int H, int n_seq, const int32_t* off, int training, Lstm1Ws* w) {
  SUMK_ARG(In > 0 && In % 4 == 0, "lstm: input size %d must be a positive multiple of 4", In);
  SUMK_ARG(H > 0 && H % 4 == 0, "lstm: hidden size %d must be a positive multiple of 4", H);
  SUMK_ARG(n_seq > 0 && off != nullptr && off[0] == 0, "lstm: empty batch / seq_off[0] != 0");
  int tmax = 0;
  for (int s = 0; s < n_seq; ++s) {
    int T = off[s + 1] - off[s];
    SUMK_ARG(T > 0, "lstm: sequence %d has %d steps", s, T);
    tmax = T > tmax ? T : tmax;
  }
  const size_t R = (size_t)off[n_seq];
  size_t p = 0;
  auto take = [&](size_t bytes) { size_t at = p; p += align_up(bytes, 256); return at; };
  w->n_rows = (int32_t)R; w->t_max = tmax;
  w->g = take(R * 4 * H * 4);
  w->cstate = take((size_t)n_seq * H * 4);
  w->prob = take(8 * sizeof(GemmProb));
  w->gates = w->call = w->hprev = w->dg = w->dcstate = w->slab = w->prob_sk = w->colpart = w->partial = 0;
  w->slab_elems = 0;
  if (training) {
    w->gates = take(R * 4 * H * 4);
    w->call = take(R * H * 4);
    w->hprev = take(R * H * 4);
    w->dg = take(R * 4 * H * 4);
    w->dcstate = take((size_t)n_seq * H * 4);
    w->slab_elems = (size_t)8 * (4 * H) * (size_t)(In > H ? In : H);
    w->slab = take(w->slab_elems * 4);
    w->prob_sk = take(64 * sizeof(GemmProb));
    w->colpart = take((size_t)128 * 4 * H * 4);
    w->partial = take(tgemv_partial_bytes(H, n_seq));     // split partial sums of the small-batch transposed mat-vec
  }
  w->total = p;
  return SUMK_OK;
}

// h_last[s][j] = h_out[last row of s][j]; c_last from the per-row cell states (training) or the running state (inference)
__global__ void lstm_last_state_kernel(const float* __restrict__ h_out, const float* __restrict__ c_all,
                                       const float* __restrict__ cstate, const int32_t* __restrict__ off, int n_seq, int H,
                                       float* __restrict__ h_last, float* __restrict__ c_last) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)n_seq * H) return;
  const int s = (int)(idx / H), j = (int)(idx % H);
  const int64_t row = off[s + 1] - 1;
  if (h_last) h_last[idx] = h_out[row * H + j];
  if (c_last) c_last[idx] = c_all ? c_all[row * H + j] : cstate[idx];
}

extern "C" size_t sumk_lstm_workspace_bytes(int32_t In, int32_t H, int32_t n_seq, const int32_t* seq_off_host, int32_t training) {
  Lstm1Ws w;
  if (lstm1_carve(In, H, n_seq, seq_off_host, training, &w) != SUMK_OK) return 0;
  return w.total;
}

extern "C" int sumk_lstm_layer_forward(const float* x, int32_t In, int32_t H, int32_t n_seq, const int32_t* seq_off_host,
                                       const int32_t* seq_off_dev, const sumk_lstm_dir_weights* w, const float* h0,
                                       const float* c0, float* h_out, float* h_last, float* c_last, void* workspace,
                                       size_t workspace_bytes, int32_t training, int32_t precision, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SUMK_ARG(x && seq_off_dev && w && h_out && workspace, "lstm_forward: null pointer");
  SUMK_ARG(w->w_ih && w->w_hh && w->b_ih && w->b_hh, "lstm_forward: null weight");
  SUMK_ARG(precision >= SUMK_PRECISION_FP32 && precision <= SUMK_PRECISION_MAX, "lstm_forward: unknown precision %d", precision);
  Lstm1Ws L;
  SUMK_TRY(lstm1_carve(In, H, n_seq, seq_off_host, training, &L));
  if (workspace_bytes < L.total) {
    set_error("lstm_forward: workspace %zu < required %zu", workspace_bytes, L.total);
    return SUMK_ERR_WORKSPACE;
  }
  char* ws = (char*)workspace;
  const int R = L.n_rows;
  float* G = (float*)(ws + L.g);
  GemmProb* prob = (GemmProb*)(ws + L.prob);
  const int small = gemm_tiles(R, 4 * H, 0) >= 512 ? 0 : 1;
  SUMK_TRY(fill_single_prob(prob, R, 4 * H, In, In, In, 4 * H, 0, small, stream));
  {
    GemmLaunch g;   // G = X W_ih^T + b_ih + b_hh
    g.A = x; g.B[0] = w->w_ih; g.bias0[0] = w->b_ih; g.bias1[0] = w->b_hh;
    g.C = G; g.probs = prob; g.small_tile = small; g.total_tiles = gemm_tiles(R, 4 * H, small); g.precision = precision;
    g.lean = gemm_lean_ok(R, 4 * H, In, In, In);
    SUMK_TRY(launch_gemm(GEMM_NT, EPI_BIAS2, g, stream));
  }
  if (n_seq <= GV_MAXB) {      // a few sequences: bandwidth-shaped mat-vec steps
    GemvStepArgs ga;
    ga.a1 = nullptr; ga.a1_shift = 0; ga.w1 = nullptr; ga.G = G; ga.b1 = ga.b2 = nullptr; ga.w2 = w->w_hh; ga.h0 = h0; ga.c0 = c0;
    ga.hseq = h_out; ga.c_all = training ? (float*)(ws + L.call) : nullptr; ga.cstate = training ? nullptr : (float*)(ws + L.cstate);
    ga.gates = training ? (float*)(ws + L.gates) : nullptr; ga.hprev = training ? (float*)(ws + L.hprev) : nullptr; ga.xin = nullptr;
    ga.off = seq_off_dev; ga.n_seq = n_seq; ga.H = H;
    for (int t = 0; t < L.t_max; ++t) {
      ga.t = t;
      hipLaunchKernelGGL(lstm_gemv_step_kernel, dim3((unsigned)((H + 3) / 4)), dim3(256), 0, stream, ga);
    }
    if (h_last || c_last) {
      const int64_t n = (int64_t)n_seq * H;
      hipLaunchKernelGGL(lstm_last_state_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, h_out,
                         (const float*)ga.c_all, (const float*)ga.cstate, seq_off_dev, n_seq, H, h_last, c_last);
    }
    SUMK_HIP(hipGetLastError());
    return SUMK_OK;
  }
  StepArgs a;
  a.G = G; a.whh[0] = w->w_hh; a.whh[1] = nullptr; a.Hout = h_out;
  a.cstate = training ? nullptr : (float*)(ws + L.cstate);
  a.gates = training ? (float*)(ws + L.gates) : nullptr;
  a.c_all = training ? (float*)(ws + L.call) : nullptr;
  a.hprev = training ? (float*)(ws + L.hprev) : nullptr;
  a.off = seq_off_dev; a.n_seq = n_seq; a.H = H; a.n_ublk = (H + 7) / 8;
  a.nd = 1; a.h0 = h0; a.c0 = c0;
  const int n_mtiles = (n_seq + 31) / 32;
  const dim3 grid((unsigned)(n_mtiles * a.n_ublk)), block(256);
  for (int t = 0; t < L.t_max; ++t) {
    a.t = t;
    hipLaunchKernelGGL(lstm_step_kernel, grid, block, 0, stream, a);
  }
  if (h_last || c_last) {
    const int64_t n = (int64_t)n_seq * H;
    hipLaunchKernelGGL(lstm_last_state_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, h_out,
                       (const float*)a.c_all, (const float*)a.cstate, seq_off_dev, n_seq, H, h_last, c_last);
  }
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}

extern "C" int sumk_lstm_layer_backward(const float* x, const float* h_out, const float* dh_out, const float* dh_last,
                                        const float* dc_last, int32_t In, int32_t H, int32_t n_seq,
                                        const int32_t* seq_off_host, const int32_t* seq_off_dev,
                                        const sumk_lstm_dir_weights* w, const float* c0, const sumk_lstm_dir_grads* gr,
                                        float* dx, float* dh0, float* dc0, void* workspace, size_t workspace_bytes,
                                        int32_t precision, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SUMK_ARG(x && h_out && seq_off_dev && w && gr && workspace, "lstm_backward: null pointer");
  SUMK_ARG(w->w_ih && w->w_hh && gr->w_ih && gr->w_hh && gr->b_ih && gr->b_hh, "lstm_backward: null weight/grad");
  SUMK_ARG(precision >= SUMK_PRECISION_FP32 && precision <= SUMK_PRECISION_MAX, "lstm_backward: unknown precision %d", precision);
  Lstm1Ws L;
  SUMK_TRY(lstm1_carve(In, H, n_seq, seq_off_host, 1, &L));
  if (workspace_bytes < L.total) {
    set_error("lstm_backward: workspace %zu < required %zu (needs the training-mode forward's workspace)", workspace_bytes, L.total);
    return SUMK_ERR_WORKSPACE;
  }
  char* ws = (char*)workspace;
  const int R = L.n_rows;
  float* dG = (float*)(ws + L.dg);
  float* hprev = (float*)(ws + L.hprev);
  float* slab = (float*)(ws + L.slab);
  float* colpart = (float*)(ws + L.colpart);
  GemmProb* prob = (GemmProb*)(ws + L.prob);
  GemmProb* psk = (GemmProb*)(ws + L.prob_sk);
  if (n_seq <= GV_MAXB) {      // a few sequences: bandwidth-shaped transposed mat-vec + cell kernel per step
    const TGemvGeom tg = tgemv_geom(H);
    TGemvArgs ta;
    ta.dGa = dG; ta.wa = w->w_hh; ta.dGb = nullptr; ta.wb = nullptr; ta.b_shift = 0; ta.partial = (float*)(ws + L.partial);
    ta.off = seq_off_dev; ta.n_seq = n_seq; ta.H = H; ta.rows_per_split = tg.rows_per_split;
    CellBwdArgs ca;
    ca.dext = dh_out; ca.dh_last = dh_last; ca.dc_last = dc_last; ca.partial = ta.partial; ca.n_ksplit = tg.n_ksplit;
    ca.gates = (const float*)(ws + L.gates); ca.c_all = (const float*)(ws + L.call); ca.c0 = c0;
    ca.dG = dG; ca.dcstate = (float*)(ws + L.dcstate); ca.dh0 = dh0; ca.dc0 = dc0;
    ca.off = seq_off_dev; ca.n_seq = n_seq; ca.H = H;
    for (int t = L.t_max - 1; t >= ((dh0 || dc0) ? -1 : 0); --t) {
      ta.t = ca.t = t;
      hipLaunchKernelGGL(lstm_tgemv_partial_kernel, dim3((unsigned)tg.strips, (unsigned)tg.n_ksplit), dim3(256), 0, stream, ta);
      hipLaunchKernelGGL(lstm_cellbwd_kernel, dim3((unsigned)((H + 63) / 64), (unsigned)n_seq), dim3(256), 0, stream, ca);
    }
  } else {
  BwdStepArgs a;
  a.whh[0] = w->w_hh; a.whh[1] = nullptr; a.dHout = dh_out; a.gates = (const float*)(ws + L.gates);
  a.c_all = (const float*)(ws + L.call); a.dG = dG; a.dcstate = (float*)(ws + L.dcstate);
  a.off = seq_off_dev; a.n_seq = n_seq; a.H = H; a.n_jblk = (H + 31) / 32;
  a.nd = 1; a.c0 = c0; a.dh_last = dh_last; a.dc_last = dc_last; a.dh0 = dh0; a.dc0 = dc0;
  const int n_mtiles = (n_seq + 31) / 32;
  const dim3 grid((unsigned)(n_mtiles * a.n_jblk)), block(512);
  for (int t = L.t_max - 1; t >= ((dh0 || dc0) ? -1 : 0); --t) {
    a.t = t;
    hipLaunchKernelGGL(lstm_bwd_step_kernel, grid, block, 0, stream, a);
  }
  }
  SUMK_HIP(hipGetLastError());
  {
    float* out[4] = {gr->w_ih, nullptr, nullptr, nullptr};
    SUMK_TRY(gemm_tn_splitk_accum(dG, 4 * H, x, In, 4 * H, In, R, slab, L.slab_elems, psk, 64, out, 4 * H, In, 1.f, stream, precision));
  }
  {
    float* out[4] = {gr->w_hh, nullptr, nullptr, nullptr};
    SUMK_TRY(gemm_tn_splitk_accum(dG, 4 * H, hprev, H, 4 * H, H, R, slab, L.slab_elems, psk, 64, out, 4 * H, H, 1.f, stream, precision));
    const ReduceSeg segs[2] = {{0, 4 * H, gr->b_ih}, {0, 4 * H, gr->b_hh}};
    SUMK_TRY(colsum_multi(dG, 4 * H, R, 4 * H, colpart, 128, segs, 2, stream));
  }
  if (dx) {
    const int small = gemm_tiles(R, In, 0) >= 512 ? 0 : 1;
    SUMK_TRY(fill_single_prob(prob + 1, R, In, 4 * H, 4 * H, In, In, 0, small, stream));
    GemmLaunch g;
    g.A = dG; g.B[0] = w->w_ih; g.C = dx; g.probs = prob + 1; g.small_tile = small;
    g.total_tiles = gemm_tiles(R, In, small); g.precision = precision;
    SUMK_TRY(launch_gemm(GEMM_NN, EPI_NONE, g, stream));
  }
  return SUMK_OK;
}

// ------------------------------------------------------------------------------------------- step-wise decoder
// SumGAN's dLSTM (summarizer/models/sumgan.py:74-115): an L-layer forward-running LSTM driven one step at a time, whose
// input at step t is its own top-layer output of step t-1 (zeros at t = 0) -- so no input projection can be hoisted and
// every (layer, step) is a cell with TWO mat-vecs:  pre = W_ih a1 + W_hh a2 + b_ih + b_hh  with
//   a1 = h_{top}[t-1] (layer 0)  or  h_{l-1}[t] (layer l > 0),   a2 = h_l[t-1]  (h0_l at t = 0).
// One launch per (layer, step); block = (32 sequences) x (8 hidden units x 4 gates), 4 waves split K, fragments straight
// from memory, exactly like lstm_step_kernel.  Everything the backward pass needs is kept per row.
struct DecStepArgs {
  const float* a1;       // (R, H) source rows of the first operand, or nullptr (layer 0 at t = 0: zeros)
  int32_t a1_shift;      // row offset of a1 relative to this step's row: -1 (previous step) or 0 (same step)
  const float* w1;       // (4H, H)
  const float* w2;       // (4H, H)
  const float* b1; const float* b2;   // (4H)
  const float* h0; const float* c0;   // (n_seq, H) or nullptr
  float* hseq;           // (R, H) this layer's outputs (a2 = previous row)
  float* c_all;          // (R, H)
  float* gates;          // (R, 4H)
  float* hprev;          // (R, H) a2 as used (for dW_hh)
  float* xin;            // (R, H) a1 as used (for dW_ih), or nullptr when a1 rows are addressable directly
  const int32_t* off;
  int32_t n_seq, H, t, n_ublk;
};

__global__ __launch_bounds__(256) void lstm_dec_step_kernel(DecStepArgs a) {
  __shared__ float part[4][32][33];
  const int H = a.H, t = a.t;
  const int ublk = blockIdx.x % a.n_ublk, mtile = blockIdx.x / a.n_ublk;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int j0 = ublk * 8;
  {
    const int sv = mtile * 32 + li;
    bool act = false;
    int64_t row = 0;
    if (sv < a.n_seq) { const int r0 = a.off[sv], T = a.off[sv + 1] - r0; if (t < T) { act = true; row = r0 + t; } }
    const float* p1 = (act && a.a1 && (a.a1_shift == 0 || t > 0)) ? a.a1 + (row + a.a1_shift) * H : nullptr;
    const float* p2 = act ? (t > 0 ? a.hseq + (row - 1) * H : (a.h0 ? a.h0 + (int64_t)sv * H : nullptr)) : nullptr;
    const int gcol = li >> 3, unit = min(j0 + (li & 7), H - 1);
    const float* wp1 = a.w1 + (int64_t)(gcol * H + unit) * H;
    const float* wp2 = a.w2 + (int64_t)(gcol * H + unit) * H;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int nchunk = (H + 7) >> 3;
    for (int pair = 0; pair < 2; ++pair) {
      const float* ap = pair == 0 ? p1 : p2;
      const float* wp = pair == 0 ? wp1 : wp2;
      const float* asafe = ap ? ap : wp;           // a legal address for the unconditional loads; masked below
      for (int kb = wave; kb < nchunk; kb += 16) {
        float4 av[4], bv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int kc = min((kb + 4 * q) * 8 + 4 * lh, H - 4);
          bv[q] = *reinterpret_cast<const float4*>(wp + kc);
          av[q] = *reinterpret_cast<const float4*>(asafe + kc);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int k = (kb + 4 * q) * 8 + 4 * lh;
          if (ap == nullptr || k >= H) av[q] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (k >= H) bv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].x, bv[q].x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].y, bv[q].y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].z, bv[q].z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].w, bv[q].w, acc, 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) part[wave][(r & 3) + 8 * (r >> 2) + 4 * lh][li] = acc[r];
    __syncthreads();
  }
  const int i = tid >> 3, u = tid & 7;
  const int sv = mtile * 32 + i, j = j0 + u;
  if (sv >= a.n_seq || j >= H) return;
  const int r0 = a.off[sv], T = a.off[sv + 1] - r0;
  if (t >= T) return;
  const int64_t row = r0 + t;
  float pre[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
    pre[q] = a.b1[q * H + j] + a.b2[q * H + j] +
             ((part[0][i][q * 8 + u] + part[1][i][q * 8 + u]) + (part[2][i][q * 8 + u] + part[3][i][q * 8 + u]));
  const float cprev = t > 0 ? a.c_all[(row - 1) * H + j] : (a.c0 ? a.c0[(int64_t)sv * H + j] : 0.f);
  const float ig = sigmoidf_(pre[0]), fg = sigmoidf_(pre[1]), gg = tanhf(pre[2]), og = sigmoidf_(pre[3]);
  const float c = fg * cprev + ig * gg;
  const float h = og * tanhf(c);
  a.hprev[row * H + j] = t > 0 ? a.hseq[(row - 1) * H + j] : (a.h0 ? a.h0[(int64_t)sv * H + j] : 0.f);
  if (a.xin) a.xin[row * H + j] = (a.a1 && (a.a1_shift == 0 || t > 0)) ? a.a1[(row + a.a1_shift) * H + j] : 0.f;
  a.hseq[row * H + j] = h;
  a.c_all[row * H + j] = c;
  float* gs = a.gates + row * 4 * H;
  gs[j] = ig; gs[H + j] = fg; gs[2 * H + j] = gg; gs[3 * H + j] = og;
}

// Backward of one (layer, step), run for t = T-1 .. 0 (and t = -1 for the initial-state gradients).
//   dh[i][j] = dext[row][j] + sum_k dGa[row + 1][k] Wa[k][j]  (pair A: own recurrence, needs step t+1)
//                           + sum_k dGb[row + sb][k] Wb[k][j]  (pair B: the consumer of this output: the layer above at the same
//                                                               step (sb = 0) or layer 0 at the next step (sb = 1))
struct DecBwdArgs {
  const float* dext;     // (R, H) external gradient of this layer's outputs, or nullptr
  const float* dGa; const float* wa;                     // own dG (R, 4H) and W_hh (4H, H)
  const float* dGb; const float* wb; int32_t b_shift;    // consumer's dG and W_ih; nullptr = none
  const float* gates; const float* c_all; const float* c0;
  float* dG;             // (R, 4H) output
  float* dcstate;        // (n_seq, H)
  float* dh0; float* dc0;// (n_seq, H) outputs of the pass t = -1
  const int32_t* off;
  int32_t n_seq, H, t, n_jblk;
};

__global__ __launch_bounds__(512) void lstm_dec_bwd_step_kernel(DecBwdArgs a) {
  __shared__ float part[8][32][33];
  const int H = a.H, H4 = 4 * H, t = a.t;
  const int jblk = blockIdx.x % a.n_jblk, mtile = blockIdx.x / a.n_jblk;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int j0 = jblk * 32;
  {
    const int sv = mtile * 32 + li;
    const float* pa = nullptr; const float* pb = nullptr;
    if (sv < a.n_seq) {
      const int r0 = a.off[sv], T = a.off[sv + 1] - r0;
      if (t + 1 < T) pa = a.dGa + (int64_t)(r0 + t + 1) * H4;
      if (a.dGb && t >= 0 && t + a.b_shift < T) pb = a.dGb + (int64_t)(r0 + t + a.b_shift) * H4;
    }
    const int jc = min(j0 + li, H - 1);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int nchunk = H4 >> 3;
    for (int pair = 0; pair < 2; ++pair) {
      const float* gp = pair == 0 ? pa : pb;
      const float* wsrc = pair == 0 ? a.wa : a.wb;
      if (wsrc == nullptr) continue;                 // (uniform: kernel argument)
      const float* wp = wsrc + jc;
      const float* gsafe = gp ? gp : a.dGa;
      for (int kb = wave; kb < nchunk; kb += 32) {
        float4 av[4];
        float bq[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int k = min((kb + 8 * q) * 8 + 4 * lh, H4 - 4);
          av[q] = *reinterpret_cast<const float4*>(gsafe + k);
#pragma unroll
          for (int e = 0; e < 4; ++e) bq[q][e] = wp[(int64_t)(k + e) * H];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool live = gp != nullptr && (kb + 8 * q) < nchunk;
          if (!live) av[q] = make_float4(0.f, 0.f, 0.f, 0.f);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].x, bq[q][0], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].y, bq[q][1], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].z, bq[q][2], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].w, bq[q][3], acc, 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) part[wave][(r & 3) + 8 * (r >> 2) + 4 * lh][li] = acc[r];
    __syncthreads();
  }
#pragma unroll
  for (int rep = 0; rep < 2; ++rep) {
    const int i = (tid >> 5) + 16 * rep, u = tid & 31;
    const int sv = mtile * 32 + i, j = j0 + u;
    if (sv >= a.n_seq || j >= H) continue;
    const int r0 = a.off[sv], T = a.off[sv + 1] - r0;
    if (t >= T) continue;
    float rec = 0.f;
#pragma unroll
    for (int w8 = 0; w8 < 8; ++w8) rec += part[w8][i][u];
    float* dcs = a.dcstate + (int64_t)sv * H + j;
    if (t < 0) {
      if (a.dh0) a.dh0[(int64_t)sv * H + j] = rec;
      if (a.dc0) a.dc0[(int64_t)sv * H + j] = *dcs;
      continue;
    }
    const int64_t row = r0 + t;
    const float dh = (a.dext ? a.dext[row * H + j] : 0.f) + rec;
    const float* gs = a.gates + row * H4;
    const float ig = gs[j], fg = gs[H + j], gg = gs[2 * H + j], og = gs[3 * H + j];
    const float c = a.c_all[row * H + j];
    const float cprev = t > 0 ? a.c_all[(row - 1) * H + j] : (a.c0 ? a.c0[(int64_t)sv * H + j] : 0.f);
    const float tc = tanhf(c);
    const float dc = (t + 1 < T ? *dcs : 0.f) + dh * og * (1.f - tc * tc);
    float* dg = a.dG + row * H4;
    dg[j] = dc * gg * ig * (1.f - ig);
    dg[H + j] = dc * cprev * fg * (1.f - fg);
    dg[2 * H + j] = dc * ig * (1.f - gg * gg);
    dg[3 * H + j] = dh * tc * og * (1.f - og);
    *dcs = dc * fg;
  }
}

struct DecWs {
  size_t hseq, call, gates, hprev, dg, dcstate;   // per layer: offset of layer 0, layer l at + l * stride
  size_t s_hseq, s_gates, s_dcstate;
  size_t xin0, slab, prob_sk, colpart, partial, total, slab_elems;
  int32_t n_rows, t_max;
};
static int dec_carve(int H, int Lr, int n_seq, const int32_t* off, DecWs* w) {
  SUMK_ARG(H > 0 && H % 4 == 0, "lstm_decoder: hidden size %d must be a positive multiple of 4", H);
  SUMK_ARG(Lr >= 1 && Lr <= 8, "lstm_decoder: %d layers (supported: 1..8)", Lr);
  SUMK_ARG(n_seq > 0 && off != nullptr && off[0] == 0, "lstm_decoder: empty batch / seq_off[0] != 0");
  int tmax = 0;
  for (int s = 0; s < n_seq; ++s) { int T = off[s + 1] - off[s]; SUMK_ARG(T > 0, "lstm_decoder: sequence %d has %d steps", s, T); tmax = T > tmax ? T : tmax; }
  const size_t R = (size_t)off[n_seq];
  size_t p = 0;
  auto take = [&](size_t bytes) { size_t at = p; p += align_up(bytes, 256); return at; };
  w->n_rows = (int32_t)R; w->t_max = tmax;
  w->s_hseq = align_up(R * H * 4, 256); w->s_gates = align_up(R * 4 * H * 4, 256); w->s_dcstate = align_up((size_t)n_seq * H * 4, 256);
  w->hseq = take(w->s_hseq * Lr); w->call = take(w->s_hseq * Lr); w->hprev = take(w->s_hseq * Lr);
  w->gates = take(w->s_gates * Lr); w->dg = take(w->s_gates * Lr); w->dcstate = take(w->s_dcstate * Lr);
  w->xin0 = take(R * H * 4);
  w->slab_elems = (size_t)8 * (4 * H) * H;
  w->slab = take(w->slab_elems * 4);
  w->prob_sk = take(64 * sizeof(GemmProb));
  w->colpart = take((size_t)128 * 4 * H * 4);
  w->partial = take(tgemv_partial_bytes(H, n_seq));
  w->total = p;
  return SUMK_OK;
}

extern "C" size_t sumk_lstm_decoder_workspace_bytes(int32_t H, int32_t n_layers, int32_t n_seq, const int32_t* seq_off_host) {
  DecWs w;
  if (dec_carve(H, n_layers, n_seq, seq_off_host, &w) != SUMK_OK) return 0;
  return w.total;
}

extern "C" int sumk_lstm_decoder_forward(int32_t H, int32_t n_layers, int32_t n_seq, const int32_t* seq_off_host,
                                         const int32_t* seq_off_dev, const sumk_lstm_dir_weights* w, const float* h0,
                                         const float* c0, float* out, void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SUMK_ARG(seq_off_dev && w && out && workspace, "lstm_decoder_forward: null pointer");
  DecWs L;
  SUMK_TRY(dec_carve(H, n_layers, n_seq, seq_off_host, &L));
  for (int l = 0; l < n_layers; ++l) SUMK_ARG(w[l].w_ih && w[l].w_hh && w[l].b_ih && w[l].b_hh, "lstm_decoder_forward: null weight (layer %d)", l);
  if (workspace_bytes < L.total) { set_error("lstm_decoder_forward: workspace %zu < required %zu", workspace_bytes, L.total); return SUMK_ERR_WORKSPACE; }
  char* ws = (char*)workspace;
  const int top = n_layers - 1;
  auto hseq = [&](int l) { return l == top ? out : (float*)(ws + L.hseq + L.s_hseq * l); };   // the top layer writes the output itself
  if (n_seq <= GV_MAXB) {      // a few sequences: bandwidth-shaped mat-vec steps
    GemvStepArgs ga;
    ga.G = nullptr; ga.cstate = nullptr; ga.off = seq_off_dev; ga.n_seq = n_seq; ga.H = H;
    for (int t = 0; t < L.t_max; ++t) {
      ga.t = t;
      for (int l = 0; l < n_layers; ++l) {
        ga.a1 = l == 0 ? hseq(top) : hseq(l - 1); ga.a1_shift = l == 0 ? -1 : 0;
        ga.w1 = w[l].w_ih; ga.w2 = w[l].w_hh; ga.b1 = w[l].b_ih; ga.b2 = w[l].b_hh;
        ga.h0 = h0 ? h0 + (size_t)l * n_seq * H : nullptr; ga.c0 = c0 ? c0 + (size_t)l * n_seq * H : nullptr;
        ga.hseq = hseq(l); ga.c_all = (float*)(ws + L.call + L.s_hseq * l); ga.gates = (float*)(ws + L.gates + L.s_gates * l);
        ga.hprev = (float*)(ws + L.hprev + L.s_hseq * l); ga.xin = l == 0 ? (float*)(ws + L.xin0) : nullptr;
        hipLaunchKernelGGL(lstm_gemv_step_kernel, dim3((unsigned)((H + 3) / 4)), dim3(256), 0, stream, ga);
      }
    }
    SUMK_HIP(hipGetLastError());
    return SUMK_OK;
  }
  DecStepArgs a;
  a.off = seq_off_dev; a.n_seq = n_seq; a.H = H; a.n_ublk = (H + 7) / 8;
  const dim3 grid((unsigned)(((n_seq + 31) / 32) * a.n_ublk)), block(256);
  for (int t = 0; t < L.t_max; ++t) {
    a.t = t;
    for (int l = 0; l < n_layers; ++l) {
      a.a1 = l == 0 ? hseq(top) : hseq(l - 1); a.a1_shift = l == 0 ? -1 : 0;
      a.w1 = w[l].w_ih; a.w2 = w[l].w_hh; a.b1 = w[l].b_ih; a.b2 = w[l].b_hh;
      a.h0 = h0 ? h0 + (size_t)l * n_seq * H : nullptr; a.c0 = c0 ? c0 + (size_t)l * n_seq * H : nullptr;
      a.hseq = hseq(l); a.c_all = (float*)(ws + L.call + L.s_hseq * l); a.gates = (float*)(ws + L.gates + L.s_gates * l);
      a.hprev = (float*)(ws + L.hprev + L.s_hseq * l); a.xin = l == 0 ? (float*)(ws + L.xin0) : nullptr;
      hipLaunchKernelGGL(lstm_dec_step_kernel, grid, block, 0, stream, a);
    }
  }
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}

extern "C" int sumk_lstm_decoder_backward(int32_t H, int32_t n_layers, int32_t n_seq, const int32_t* seq_off_host,
                                          const int32_t* seq_off_dev, const sumk_lstm_dir_weights* w, const float* c0,
                                          const float* out, const float* dout, const sumk_lstm_dir_grads* gr, float* dh0,
                                          float* dc0, void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SUMK_ARG(seq_off_dev && w && out && dout && gr && workspace, "lstm_decoder_backward: null pointer");
  DecWs L;
  SUMK_TRY(dec_carve(H, n_layers, n_seq, seq_off_host, &L));
  if (workspace_bytes < L.total) { set_error("lstm_decoder_backward: workspace %zu < required %zu", workspace_bytes, L.total); return SUMK_ERR_WORKSPACE; }
  char* ws = (char*)workspace;
  const int R = L.n_rows, top = n_layers - 1;
  auto dG = [&](int l) { return (float*)(ws + L.dg + L.s_gates * l); };
  auto hseq = [&](int l) { return l == top ? out : (const float*)(ws + L.hseq + L.s_hseq * l); };
  const bool want0 = dh0 || dc0;
  if (n_seq <= GV_MAXB) {      // a few sequences: transposed mat-vec partials + cell kernel per (layer, step)
    const TGemvGeom tg = tgemv_geom(H);
    TGemvArgs ta;
    ta.partial = (float*)(ws + L.partial); ta.off = seq_off_dev; ta.n_seq = n_seq; ta.H = H; ta.rows_per_split = tg.rows_per_split;
    CellBwdArgs ca;
    ca.dh_last = ca.dc_last = nullptr; ca.partial = ta.partial; ca.n_ksplit = tg.n_ksplit; ca.off = seq_off_dev; ca.n_seq = n_seq; ca.H = H;
    for (int t = L.t_max - 1; t >= (want0 ? -1 : 0); --t) {
      ta.t = ca.t = t;
      for (int l = top; l >= 0; --l) {
        ta.dGa = dG(l); ta.wa = w[l].w_hh;
        if (l == top) { ta.dGb = dG(0); ta.wb = w[0].w_ih; ta.b_shift = 1; }
        else { ta.dGb = dG(l + 1); ta.wb = w[l + 1].w_ih; ta.b_shift = 0; }
        ca.dext = l == top ? dout : nullptr;
        ca.gates = (const float*)(ws + L.gates + L.s_gates * l); ca.c_all = (const float*)(ws + L.call + L.s_hseq * l);
        ca.c0 = c0 ? c0 + (size_t)l * n_seq * H : nullptr;
        ca.dG = dG(l); ca.dcstate = (float*)(ws + L.dcstate + L.s_dcstate * l);
        ca.dh0 = dh0 ? dh0 + (size_t)l * n_seq * H : nullptr; ca.dc0 = dc0 ? dc0 + (size_t)l * n_seq * H : nullptr;
        hipLaunchKernelGGL(lstm_tgemv_partial_kernel, dim3((unsigned)tg.strips, (unsigned)tg.n_ksplit), dim3(256), 0, stream, ta);
        hipLaunchKernelGGL(lstm_cellbwd_kernel, dim3((unsigned)((H + 63) / 64), (unsigned)n_seq), dim3(256), 0, stream, ca);
      }
    }
  } else {
  DecBwdArgs a;
  a.off = seq_off_dev; a.n_seq = n_seq; a.H = H; a.n_jblk = (H + 31) / 32;
  const dim3 grid((unsigned)(((n_seq + 31) / 32) * a.n_jblk)), block(512);
  for (int t = L.t_max - 1; t >= (want0 ? -1 : 0); --t) {
    a.t = t;
    for (int l = top; l >= 0; --l) {
      a.dext = l == top ? dout : nullptr;
      a.dGa = dG(l); a.wa = w[l].w_hh;
      if (l == top) { a.dGb = dG(0); a.wb = w[0].w_ih; a.b_shift = 1; }          // x_{t+1} = h_top[t] feeds layer 0 at the next step
      else { a.dGb = dG(l + 1); a.wb = w[l + 1].w_ih; a.b_shift = 0; }            // feeds the layer above at the same step
      a.gates = (const float*)(ws + L.gates + L.s_gates * l); a.c_all = (const float*)(ws + L.call + L.s_hseq * l);
      a.c0 = c0 ? c0 + (size_t)l * n_seq * H : nullptr;
      a.dG = dG(l); a.dcstate = (float*)(ws + L.dcstate + L.s_dcstate * l);
      a.dh0 = dh0 ? dh0 + (size_t)l * n_seq * H : nullptr; a.dc0 = dc0 ? dc0 + (size_t)l * n_seq * H : nullptr;
      hipLaunchKernelGGL(lstm_dec_bwd_step_kernel, grid, block, 0, stream, a);
    }
  }
  }
  SUMK_HIP(hipGetLastError());
  float* slab = (float*)(ws + L.slab);
  GemmProb* psk = (GemmProb*)(ws + L.prob_sk);
  float* colpart = (float*)(ws + L.colpart);
  for (int l = 0; l < n_layers; ++l) {
    SUMK_ARG(gr[l].w_ih && gr[l].w_hh && gr[l].b_ih && gr[l].b_hh, "lstm_decoder_backward: null grad (layer %d)", l);
    const float* xin = l == 0 ? (const float*)(ws + L.xin0) : hseq(l - 1);
    { float* o[4] = {gr[l].w_ih, nullptr, nullptr, nullptr};
      SUMK_TRY(gemm_tn_splitk_accum(dG(l), 4 * H, xin, H, 4 * H, H, R, slab, L.slab_elems, psk, 64, o, 4 * H, H, 1.f, stream)); }
    { float* o[4] = {gr[l].w_hh, nullptr, nullptr, nullptr};
      SUMK_TRY(gemm_tn_splitk_accum(dG(l), 4 * H, (const float*)(ws + L.hprev + L.s_hseq * l), H, 4 * H, H, R, slab, L.slab_elems, psk, 64, o,
                                    4 * H, H, 1.f, stream)); }
    const ReduceSeg segs[2] = {{0, 4 * H, gr[l].b_ih}, {0, 4 * H, gr[l].b_hh}};
    SUMK_TRY(colsum_multi(dG(l), 4 * H, R, 4 * H, colpart, 128, segs, 2, stream));
  }
  return SUMK_OK;
}

// ------------------------------------------------------------------------------------------- dense layer
// y = x W^T + b and its backward, for the small Linear layers around the LSTM stacks (eLSTM mu / logvar, dLSTM recons:
// sumgan.py:58-59,84) -- the MFMA GEMM with the bias in its epilogue, split-K weight gradient, column-sum bias gradient.
struct LinWs { size_t prob, prob_sk, slab, colpart, total, slab_elems; };
static void linear_carve(int N, int K, LinWs* w) {
  size_t p = 0;
  auto take = [&](size_t bytes) { size_t at = p; p += align_up(bytes, 256); return at; };
  w->prob = take(4 * sizeof(GemmProb));
  w->prob_sk = take(64 * sizeof(GemmProb));
  w->slab_elems = (size_t)8 * N * K;
  w->slab = take(w->slab_elems * 4);
  w->colpart = take((size_t)128 * N * 4);
  w->total = p;
}
extern "C" size_t sumk_linear_workspace_bytes(int32_t N, int32_t K) {
  if (N <= 0 || K <= 0) return 0;
  LinWs w; linear_carve(N, K, &w); return w.total;
}
extern "C" int sumk_linear_forward(const float* x, const float* w, const float* b, float* y, int32_t M, int32_t N, int32_t K,
                                   void* workspace, size_t workspace_bytes, int32_t precision, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SUMK_ARG(x && w && y && workspace, "linear_forward: null pointer");
  SUMK_ARG(M > 0 && N > 0 && K > 0 && K % 4 == 0, "linear_forward: bad shape M=%d N=%d K=%d (K must be a multiple of 4)", M, N, K);
  SUMK_ARG(precision >= SUMK_PRECISION_FP32 && precision <= SUMK_PRECISION_MAX, "linear_forward: unknown precision %d", precision);
  LinWs L; linear_carve(N, K, &L);
  if (workspace_bytes < L.total) { set_error("linear_forward: workspace %zu < required %zu", workspace_bytes, L.total); return SUMK_ERR_WORKSPACE; }
  GemmProb* prob = (GemmProb*)((char*)workspace + L.prob);
  const int small = gemm_tiles(M, N, 0) >= 512 ? 0 : 1;
  SUMK_TRY(fill_single_prob(prob, M, N, K, K, K, N, 0, small, stream));
  GemmLaunch g;
  g.A = x; g.B[0] = w; g.bias0[0] = b; g.C = y; g.probs = prob; g.small_tile = small; g.total_tiles = gemm_tiles(M, N, small);
  g.precision = precision; g.lean = gemm_lean_ok(M, N, K, K, K);
  return launch_gemm(GEMM_NT, b ? EPI_BIAS2 : EPI_NONE, g, stream);
}
extern "C" int sumk_linear_backward(const float* x, const float* w, const float* dy, int32_t M, int32_t N, int32_t K, float* dx,
                                    float* dw, float* db, void* workspace, size_t workspace_bytes, int32_t precision,
                                    void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SUMK_ARG(x && w && dy && workspace, "linear_backward: null pointer");
  SUMK_ARG(M > 0 && N > 0 && K > 0 && K % 4 == 0 && N % 4 == 0, "linear_backward: bad shape M=%d N=%d K=%d (N, K multiples of 4)", M, N, K);
  SUMK_ARG(precision >= SUMK_PRECISION_FP32 && precision <= SUMK_PRECISION_MAX, "linear_backward: unknown precision %d", precision);
  LinWs L; linear_carve(N, K, &L);
  if (workspace_bytes < L.total) { set_error("linear_backward: workspace %zu < required %zu", workspace_bytes, L.total); return SUMK_ERR_WORKSPACE; }
  char* ws = (char*)workspace;
  GemmProb* prob = (GemmProb*)(ws + L.prob);
  if (dw) {   // dW (N, K) += dY^T X
    float* out[4] = {dw, nullptr, nullptr, nullptr};
    SUMK_TRY(gemm_tn_splitk_accum(dy, N, x, K, N, K, M, (float*)(ws + L.slab), L.slab_elems, (GemmProb*)(ws + L.prob_sk), 64, out, N, K,
                                  1.f, stream, precision));
  }
  if (db) SUMK_TRY(colsum_accum(dy, N, M, N, (float*)(ws + L.colpart), 128, db, stream));
  if (dx) {   // dX (M, K) = dY W
    const int small = gemm_tiles(M, K, 0) >= 512 ? 0 : 1;
    SUMK_TRY(fill_single_prob(prob + 1, M, K, N, N, K, K, 0, small, stream));
    GemmLaunch g;
    g.A = dy; g.B[0] = w; g.C = dx; g.probs = prob + 1; g.small_tile = small; g.total_tiles = gemm_tiles(M, K, small);
    g.precision = precision;
    SUMK_TRY(launch_gemm(GEMM_NN, EPI_NONE, g, stream));
  }
  return SUMK_OK;
}

extern "C" size_t sumk_frame_head_workspace_bytes(int32_t F) { return (size_t)1024 * ((size_t)F + 4) * 4; }

extern "C" int sumk_frame_head_backward(const float* h, const float* scores, const float* dscores, int32_t n_rows,
                                        int32_t F, const float* w, float* dh, float* dw, float* db, void* workspace,
                                        size_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SUMK_ARG(h && scores && dscores && w && dh && dw && db && workspace, "frame_head_backward: null pointer");
  SUMK_ARG(n_rows > 0 && F > 0 && F % 4 == 0, "frame_head_backward: bad shape n_rows=%d F=%d", n_rows, F);
  SUMK_ARG(workspace_bytes >= sumk_frame_head_workspace_bytes(F), "frame_head_backward: workspace too small");
  int blocks = std::max(1, std::min((n_rows + 3) / 4, 256));
  float* part = (float*)workspace;
  hipLaunchKernelGGL(frame_head_bwd_kernel, dim3(blocks), dim3(256), 0, stream, h, scores, dscores, w, dh, part, n_rows, F);
  SUMK_TRY(partial_reduce_accum(part, blocks * 4, F + 4, F, dw, stream));
  SUMK_TRY(partial_reduce_accum(part + F, blocks * 4, F + 4, 1, db, stream));
  return SUMK_OK;
}

// Synchronous health check of the persistent recurrence kernels that last ran on this workspace: their bounded waits set
// an error word instead of hanging; a set word means some hand-off timed out (e.g. the 256 co-resident blocks the
// cooperative launch asked for were not all running) and the layer's outputs are INVALID.
extern "C" int sumk_bilstm_check(const void* workspace, int32_t In, int32_t H, int32_t n_seq, const int32_t* seq_off_host,
                                 int32_t training, int32_t after_backward, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SUMK_ARG(workspace, "bilstm_check: null workspace");
  LstmWs L;
  SUMK_TRY(lstm_carve(In, H, n_seq, seq_off_host, training, &L));
  unsigned flags[2] = {0u, 0u};
  SUMK_HIP(hipMemcpyAsync(&flags[0], (const char*)workspace + L.pstate, 4, hipMemcpyDeviceToHost, stream));
  if (training && after_backward) SUMK_HIP(hipMemcpyAsync(&flags[1], (const char*)workspace + L.pstate_b, 4, hipMemcpyDeviceToHost, stream));
  SUMK_HIP(hipStreamSynchronize(stream));
  if (flags[0] != 0u || flags[1] != 0u) {
    set_error("bilstm: persistent recurrence kernel timed out waiting for a team member (forward flag %u, backward flag %u); "
              "outputs are invalid -- rerun with SUMK_LSTM_PERSIST=0 to use the launch-per-step kernels", flags[0], flags[1]);
    return SUMK_ERR_HIP;
  }
  return SUMK_OK;
}

// Sticky health check (no workspace needed): synchronises `stream`, then fails if ANY persistent recurrence kernel on this
// device has timed out since the last check -- every result produced in between is suspect.  Resets the word.
extern "C" int sumk_health_check(void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  unsigned flag = 0u;
  SUMK_HIP(hipMemcpyFromSymbolAsync(&flag, HIP_SYMBOL(g_sumk_health), sizeof(flag), 0, hipMemcpyDeviceToHost, stream));
  SUMK_HIP(hipStreamSynchronize(stream));
  if (flag != 0u) {
    const unsigned zero = 0u;
    SUMK_HIP(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_sumk_health), &zero, sizeof(zero), 0, hipMemcpyHostToDevice, stream));
    SUMK_HIP(hipStreamSynchronize(stream));
    set_error("persistent LSTM recurrence kernel timed out waiting for a team member since the last health check (flag %u): "
              "scores / gradients produced in between are INVALID -- rerun with SUMK_LSTM_PERSIST=0 (launch-per-step kernels)", flag);
    return SUMK_ERR_HIP;
  }
  return SUMK_OK;
}
