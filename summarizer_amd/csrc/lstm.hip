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

namespace sumk {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct LstmWs {
  size_t g, cstate, prob, gates, call, hprev, dg, slab, dhrec, dcstate, total;
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
  w->gates = w->call = w->hprev = w->dg = w->slab = w->dhrec = w->dcstate = 0;
  if (training) {
    w->gates = take(R * 8 * H * 4);                 // post-nonlinearity i,f,g,o per row and direction
    w->call = take(R * 2 * H * 4);                  // cell state per row and direction
    w->hprev = take(R * 2 * H * 4);                 // h_{t-1} per row and direction (0 at a sequence start)
    w->dg = take(R * 8 * H * 4);                    // gradient w.r.t. gate pre-activations
    w->dcstate = take((size_t)n_seq * 2 * H * 4);
    size_t big = (size_t)(4 * H) * (size_t)(In > H ? In : H);
    w->slab = take((size_t)32 * big * 4);           // split-K partial slabs for the weight gradients
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

  if (t > 0) {
    // A operand: h_prev of video (mtile*32 + li); B operand: W_hh row of gate-column li = g*8 + u
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
    for (int kk = wave; kk < nchunk; kk += 4) {
      const int k = kk * 8 + 4 * lh, kc = min(k, H - 4);
      // unconditional 16-B loads (legal clamped addresses), zeroed afterwards: keeps the loads back to back
      float4 bv = *reinterpret_cast<const float4*>(wp + kc);
      float4 av = *reinterpret_cast<const float4*>(hp + kc);
      if (!act || k >= H) av = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k >= H) bv = make_float4(0.f, 0.f, 0.f, 0.f);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) part[wave][(r & 3) + 8 * (r >> 2) + 4 * lh][li] = acc[r];
    __syncthreads();
  }

  // ---- cell update: thread = (video i, unit u)
  const int i = tid >> 3, u = tid & 7;
  const int sv = mtile * 32 + i, j = j0 + u;
  if (sv >= a.n_seq || j >= H) return;
  const int r0 = a.off[sv], T = a.off[sv + 1] - r0;
  if (t >= T) return;
  const int64_t row = d == 0 ? r0 + t : r0 + T - 1 - t;
  const float* g = a.G + row * (8 * H) + d * 4 * H;
  float pre[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float v = g[q * H + j];
    if (t > 0) v += (part[0][i][q * 8 + u] + part[1][i][q * 8 + u]) + (part[2][i][q * 8 + u] + part[3][i][q * 8 + u]);
    pre[q] = v;
  }
  const float ig = sigmoidf_(pre[0]), fg = sigmoidf_(pre[1]), gg = tanhf(pre[2]), og = sigmoidf_(pre[3]);
  float cprev = 0.f;
  const int64_t prow = d == 0 ? row - 1 : row + 1;
  if (t > 0) cprev = a.c_all ? a.c_all[prow * (2 * H) + d * H + j] : a.cstate[((int64_t)sv * 2 + d) * H + j];
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
