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
#include <math.h>
#include <algorithm>

namespace sumk {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct LstmWs {
  size_t g, cstate, prob, gates, call, hprev, dg, slab, dhrec, dcstate, prob_sk, colpart, total;
  size_t slab_elems;
  int32_t n_rows, t_max;
};

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
  w->gates = w->call = w->hprev = w->dg = w->slab = w->dhrec = w->dcstate = w->prob_sk = w->colpart = 0;
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
  }
  w->total = p;
  return SUMK_OK;
}

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

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
};

__global__ __launch_bounds__(256) void lstm_step_kernel(StepArgs a) {
  __shared__ float part[4][32][33];
  const int H = a.H;
  const int d = blockIdx.x & 1;
  const int ublk = (blockIdx.x >> 1) % a.n_ublk;
  const int mtile = (blockIdx.x >> 1) / a.n_ublk;
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
      const float* g = a.G + row * (8 * H) + d * 4 * H;
#pragma unroll
      for (int q = 0; q < 4; ++q) pre[q] = g[q * H + j];
      if (t > 0) cprev = a.c_all ? a.c_all[prow * (2 * H) + d * H + j] : a.cstate[((int64_t)esv * 2 + d) * H + j];
    }
  }

  if (t > 0) {
    // A operand: h_prev of video (mtile*32 + li); B operand: W_hh row of gate-column li = g*8 + u.
    // Fragments come straight from memory (L2 does not survive the kernel boundary: Infinity Cache / HBM).  Measured
    // alternatives: staging the 32x128 panels through LDS in 128-B-coalesced segments was SLOWER (12.3 vs 10.5 us per
    // step at 50 videos, H=256: two extra barriers on a latency-bound kernel); see DESIGN.md "LSTM step latency".
    const int sv = mtile * 32 + li;
    bool act = false;
    const float* hp = a.Hout;  // clamped to a legal row when inactive
    if (sv < a.n_seq) {
      int r0 = a.off[sv], T = a.off[sv + 1] - r0;
      if (t < T) { act = true; hp = a.Hout + (int64_t)(d == 0 ? r0 + t - 1 : r0 + T - t) * (2 * H) + d * H; }
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
  if (t > 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      pre[q] += (part[0][i][q * 8 + u] + part[1][i][q * 8 + u]) + (part[2][i][q * 8 + u] + part[3][i][q * 8 + u]);
  }
  const float ig = sigmoidf_(pre[0]), fg = sigmoidf_(pre[1]), gg = tanhf(pre[2]), og = sigmoidf_(pre[3]);
  const int sv = esv;
  const float c = fg * cprev + ig * gg;
  const float h = og * tanhf(c);
  a.Hout[row * (2 * H) + d * H + j] = h;
  if (a.c_all) a.c_all[row * (2 * H) + d * H + j] = c; else a.cstate[((int64_t)sv * 2 + d) * H + j] = c;
  if (a.gates) {
    float* gs = a.gates + row * (8 * H) + d * 4 * H;
    gs[j] = ig; gs[H + j] = fg; gs[2 * H + j] = gg; gs[3 * H + j] = og;
  }
  if (a.hprev) a.hprev[row * (2 * H) + d * H + j] = t > 0 ? a.Hout[prow * (2 * H) + d * H + j] : 0.f;
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
};

__global__ __launch_bounds__(512) void lstm_bwd_step_kernel(BwdStepArgs a) {
  __shared__ float part[8][32][33];
  const int H = a.H, H4 = 4 * H;
  const int d = blockIdx.x & 1;
  const int jblk = (blockIdx.x >> 1) % a.n_jblk;
  const int mtile = (blockIdx.x >> 1) / a.n_jblk;
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
      if (t + 1 < T) { act = true; gp = a.dG + (int64_t)(d == 0 ? r0 + t + 1 : r0 + T - 2 - t) * (8 * H) + d * H4; }
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
    const int64_t row = d == 0 ? r0 + t : r0 + T - 1 - t;
    float dh = a.dHout[row * (2 * H) + d * H + j];
    float rec = 0.f;
#pragma unroll
    for (int w8 = 0; w8 < 8; ++w8) rec += part[w8][i][u];
    if (t + 1 < T) dh += rec;
    const float* gs = a.gates + row * (8 * H) + d * H4;
    const float ig = gs[j], fg = gs[H + j], gg = gs[2 * H + j], og = gs[3 * H + j];
    const float c = a.c_all[row * (2 * H) + d * H + j];
    const float cprev = t > 0 ? a.c_all[(d == 0 ? row - 1 : row + 1) * (2 * H) + d * H + j] : 0.f;
    const float tc = tanhf(c);
    float* dcs = a.dcstate + ((int64_t)sv * 2 + d) * H + j;
    float dc = (t + 1 < T ? *dcs : 0.f) + dh * og * (1.f - tc * tc);
    float* dg = a.dG + row * (8 * H) + d * H4;
    dg[j] = dc * gg * ig * (1.f - ig);
    dg[H + j] = dc * cprev * fg * (1.f - fg);
    dg[2 * H + j] = dc * ig * (1.f - gg * gg);
    dg[3 * H + j] = dh * tc * og * (1.f - og);
    *dcs = dc * fg;
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

extern "C" size_t sumk_bilstm_workspace_bytes(int32_t In, int32_t H, int32_t n_seq, const int32_t* seq_off_host,
                                              int32_t training) {
  LstmWs w;
  if (lstm_carve(In, H, n_seq, seq_off_host, training, &w) != SUMK_OK) return 0;
  return w.total;
}

extern "C" int sumk_bilstm_layer_forward(const float* x, int32_t In, int32_t H, int32_t n_seq,
                                         const int32_t* seq_off_host, const int32_t* seq_off_dev,
                                         const sumk_lstm_layer_weights* w, float* h_out, void* workspace,
                                         size_t workspace_bytes, int32_t training, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SUMK_ARG(x && seq_off_dev && w && h_out && workspace, "bilstm_forward: null pointer");
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
  SUMK_TRY(fill_single_prob(prob, R, 8 * H, In, In, In, 8 * H, 0, small, stream));
  {
    GemmLaunch g;
    g.A = x; g.B[0] = w->w_ih[0]; g.B[1] = w->w_ih[1]; g.n_group = 4 * H;
    g.bias0[0] = w->b_ih[0]; g.bias0[1] = w->b_ih[1]; g.bias1[0] = w->b_hh[0]; g.bias1[1] = w->b_hh[1];
    g.C = G; g.probs = prob; g.small_tile = small; g.total_tiles = gemm_tiles(R, 8 * H, small);
    SUMK_TRY(launch_gemm(GEMM_NT, EPI_BIAS2, g, stream));
  }
  // 2: recurrence
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
                                          void* workspace, size_t workspace_bytes, void* stream_) {
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
  // weight gradients: dW_ih[d] += dG_d^T X (both directions in one split-K launch), dW_hh[d] += dG_d^T h_prev_d
  {
    float* out[4] = {gr->w_ih[0], gr->w_ih[1], nullptr, nullptr};
    SUMK_TRY(gemm_tn_splitk_accum(dG, 8 * H, x, In, 8 * H, In, R, slab, L.slab_elems, psk, 64, out, 4 * H, In, 1.f, stream));
  }
  for (int d = 0; d < 2; ++d) {
    float* out[4] = {gr->w_hh[d], nullptr, nullptr, nullptr};
    SUMK_TRY(gemm_tn_splitk_accum(dG + (size_t)d * 4 * H, 8 * H, hprev + (size_t)d * H, 2 * H, 4 * H, H, R, slab, L.slab_elems,
                                  psk, 64, out, 4 * H, H, 1.f, stream));
    SUMK_TRY(colsum_accum(dG + (size_t)d * 4 * H, 8 * H, R, 4 * H, colpart, 128, gr->b_ih[d], stream));
    SUMK_TRY(colsum_accum(dG + (size_t)d * 4 * H, 8 * H, R, 4 * H, colpart, 128, gr->b_hh[d], stream));
  }
  if (dx) {  // dX = dG_fwd W_ih_fwd + dG_rev W_ih_rev
    const int small = gemm_tiles(R, In, 0) >= 512 ? 0 : 1;
    SUMK_TRY(fill_single_prob(prob + 1, R, In, 4 * H, 8 * H, In, In, 0, small, stream));
    for (int d = 0; d < 2; ++d) {
      GemmLaunch g;
      g.A = dG + (size_t)d * 4 * H; g.B[0] = w->w_ih[d]; g.C = dx; g.probs = prob + 1; g.small_tile = small;
      g.total_tiles = gemm_tiles(R, In, small);
      SUMK_TRY(launch_gemm(GEMM_NN, d == 0 ? EPI_NONE : EPI_ACCUM, g, stream));
    }
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
