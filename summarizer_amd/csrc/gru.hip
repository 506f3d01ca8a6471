// GRU cell of the reference's optional `DSN(cell="gru")` (summarizer/models/dsn.py:28-33: nn.GRU, bidirectional).  DSNTrainer never
// builds it, so this is the functional path, not a tuned one: the host (summarizer_amd/models/_bilstm.py) walks the time steps
// and per step calls the MFMA GEMM for Gh = h_{t-1} W_hh^T + b_hh and ONE fused element-wise kernel below; every arithmetic result
// still comes from this library.  torch.nn.GRU semantics (gate order r, z, n):
//   r = sigmoid(Gx_r + Gh_r)   z = sigmoid(Gx_z + Gh_z)   n = tanh(Gx_n + r * Gh_n)   h = (1 - z) * n + z * h_prev
// with Gx = x W_ih^T + b_ih and Gh = h_prev W_hh^T + b_hh.  A row whose mask is 0 (a video that has ended in a time-major batch)
// passes h_prev through and takes no gradient.
#include "sumk_internal.h"
#include <math.h>
#include <algorithm>

namespace sumk {

__device__ __forceinline__ float gru_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(256) void gru_cell_fwd_kernel(const float* __restrict__ gx, const float* __restrict__ gh,
                                                           const float* __restrict__ h_prev, const float* __restrict__ mask,
                                                           float* __restrict__ h_out, float* __restrict__ rzn, int B, int H) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * H) return;
  const int b = (int)(i / H), j = (int)(i % H);
  const float hp = h_prev[i];
  const float* x3 = gx + (int64_t)b * 3 * H;
  const float* h3 = gh + (int64_t)b * 3 * H;
  const float r = gru_sigmoid(x3[j] + h3[j]);
  const float z = gru_sigmoid(x3[H + j] + h3[H + j]);
  const float n = tanhf(x3[2 * H + j] + r * h3[2 * H + j]);
  const bool on = mask == nullptr || mask[b] != 0.f;
  h_out[i] = on ? (1.f - z) * n + z * hp : hp;
  if (rzn) { float* s = rzn + (int64_t)b * 3 * H; s[j] = r; s[H + j] = z; s[2 * H + j] = n; }
}

// dh = gradient w.r.t. this step's h (the output's gradient plus what flows back from step t+1)
__global__ __launch_bounds__(256) void gru_cell_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ rzn,
                                                           const float* __restrict__ gh, const float* __restrict__ h_prev,
                                                           const float* __restrict__ mask, float* __restrict__ dgx,
                                                           float* __restrict__ dgh, float* __restrict__ dh_prev, int B, int H) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * H) return;
  const int b = (int)(i / H), j = (int)(i % H);
  float* dx3 = dgx + (int64_t)b * 3 * H;
  float* dh3 = dgh + (int64_t)b * 3 * H;
  const float g = dh[i];
  if (mask != nullptr && mask[b] == 0.f) {
    dx3[j] = dx3[H + j] = dx3[2 * H + j] = 0.f;
    dh3[j] = dh3[H + j] = dh3[2 * H + j] = 0.f;
    dh_prev[i] = g;
    return;
  }
  const float* s = rzn + (int64_t)b * 3 * H;
  const float r = s[j], z = s[H + j], n = s[2 * H + j];
  const float ghn = gh[(int64_t)b * 3 * H + 2 * H + j];
  const float dn_pre = g * (1.f - z) * (1.f - n * n);
  const float dz_pre = g * (h_prev[i] - n) * z * (1.f - z);
  const float dr_pre = dn_pre * ghn * r * (1.f - r);
  dx3[j] = dr_pre; dx3[H + j] = dz_pre; dx3[2 * H + j] = dn_pre;
  dh3[j] = dr_pre; dh3[H + j] = dz_pre; dh3[2 * H + j] = dn_pre * r;
  dh_prev[i] = g * z;       // the direct path; the caller adds dgh . W_hh
}

}  // namespace sumk

using namespace sumk;

extern "C" int sumk_gru_cell_forward(const float* gx, const float* gh, const float* h_prev, const float* mask, float* h_out,
                                     float* rzn, int32_t B, int32_t H, void* stream) {
  SUMK_ARG(gx && gh && h_prev && h_out, "gru_cell_forward: null pointer");
  SUMK_ARG(B > 0 && H > 0, "gru_cell_forward: B=%d H=%d", B, H);
  const int64_t n = (int64_t)B * H;
  hipLaunchKernelGGL(gru_cell_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gx, gh, h_prev, mask,
                     h_out, rzn, B, H);
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}

extern "C" int sumk_gru_cell_backward(const float* dh, const float* rzn, const float* gh, const float* h_prev, const float* mask,
                                      float* dgx, float* dgh, float* dh_prev, int32_t B, int32_t H, void* stream) {
  SUMK_ARG(dh && rzn && gh && h_prev && dgx && dgh && dh_prev, "gru_cell_backward: null pointer");
  SUMK_ARG(B > 0 && H > 0, "gru_cell_backward: B=%d H=%d", B, H);
  const int64_t n = (int64_t)B * H;
  hipLaunchKernelGGL(gru_cell_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dh, rzn, gh, h_prev,
                     mask, dgx, dgh, dh_prev, B, H);
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}
