// Flat-buffer optimiser kernels: Adam exactly as torch.optim.Adam(lr, weight_decay) computes it
// (the trainers' optimiser: summarizer/models/vasnet.py:181, dsn.py:70-73) and the sum of squares behind
// clip_grad_norm_ (dsn.py:145).  HBM-bound streaming kernels: 16 B per lane, grid-stride.
#include "sumk_internal.h"
#include <math.h>
#include <algorithm>

namespace sumk {

// torch (non-amsgrad, maximize=False):  g = grad*grad_scale + wd*p ; m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g
//   p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// dyn != nullptr: step size, bias correction and gradient scale come from the device block adam_prep_kernel wrote (the
// sync-free / graph-capturable form: the step counter and the clip coefficient never visit the host).
// zero_grad: the gradient is set to zero once it has been read -- the next step's zero_grad() folded into this pass (21 MB written
// here instead of a 21 MB fill kernel plus its launch boundary ahead of every step; the reference's order zero_grad -> backward -> step,
// vasnet.py:210-212, leaves the same state).
template <bool ZERO>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps,
                                                   float wd, float step_size, float inv_sqrt_bc2, float grad_scale,
                                                   const float* __restrict__ dyn) {
  if (dyn != nullptr) { step_size = dyn[1]; inv_sqrt_bc2 = dyn[2]; grad_scale = dyn[3]; }
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 pp = reinterpret_cast<float4*>(p)[i], gg = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
#define ADAM1(c)                                                        \
    {                                                                   \
      float gr = gg.c * grad_scale + wd * pp.c;                         \
      mm.c = b1 * mm.c + (1.f - b1) * gr;                               \
      vv.c = b2 * vv.c + (1.f - b2) * gr * gr;                          \
      pp.c -= step_size * (mm.c / (sqrtf(vv.c) * inv_sqrt_bc2 + eps));  \
    }
    ADAM1(x) ADAM1(y) ADAM1(z) ADAM1(w)
    reinterpret_cast<float4*>(p)[i] = pp; reinterpret_cast<float4*>(m)[i] = mm; reinterpret_cast<float4*>(v)[i] = vv;
    if constexpr (ZERO) reinterpret_cast<float4*>(g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // tail (n not a multiple of 4)
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float gr = g[i] * grad_scale + wd * p[i];
    float mi = b1 * m[i] + (1.f - b1) * gr, vi = b2 * v[i] + (1.f - b2) * gr * gr;
    m[i] = mi; v[i] = vi;
    p[i] -= step_size * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
    if constexpr (ZERO) g[i] = 0.f;
  }
}

// state[0] (int32): optimiser steps taken so far, incremented here; state[1..3] (float): lr / (1 - b1^t), 1 / sqrt(1 - b2^t) and
// the effective gradient scale = grad_scale * min(1, max_norm / (sqrt(sumsq) * grad_scale + 1e-6)) -- torch's clip_grad_norm_ --
// all in double like the host path of sumk_adam_step.
__global__ void adam_prep_kernel(int32_t* state, float lr, float b1, float b2, float grad_scale, const float* sumsq, float max_norm) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int step = state[0] + 1;
  state[0] = step;
  const double bc1 = 1.0 - pow((double)b1, (double)step);
  const double bc2 = 1.0 - pow((double)b2, (double)step);
  float* f = reinterpret_cast<float*>(state);
  f[1] = (float)((double)lr / bc1);
  f[2] = (float)(1.0 / sqrt(bc2));
  double gs = (double)grad_scale;
  if (sumsq != nullptr) {
    const double norm = sqrt((double)sumsq[0]) * (double)grad_scale;
    const double coef = fmin(1.0, (double)max_norm / (norm + 1e-6));
    gs = (double)grad_scale * coef;
  }
  f[3] = (float)gs;
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ partial) {
  __shared__ float red[4];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) { float v = x[i]; s += v * v; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void sumsq_final_kernel(const float* __restrict__ partial, int n, float* __restrict__ out) {
  // one wave, fixed order -> deterministic
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) s += partial[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (threadIdx.x == 0) out[0] += s;
}

// fp32 <-> bf16 of a flat buffer (round to nearest even; hipcc emits v_cvt_pk_bf16_f32, NaN stays NaN): the gradient bucket
// travels through the data-parallel all-reduce as bf16 in the mixed-precision training mode (half the bytes over xGMI).
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, int64_t n) {
  const int64_t n4 = n >> 2, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
    reinterpret_cast<bf16x4_t*>(dst)[i] = __builtin_convertvector(reinterpret_cast<const f32x4_t*>(src)[i], bf16x4_t);
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = (__bf16)src[i];
}
__global__ __launch_bounds__(256) void bf16_to_f32_kernel(const __bf16* __restrict__ src, float* __restrict__ dst, int64_t n) {
  const int64_t n4 = n >> 2, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
    reinterpret_cast<f32x4_t*>(dst)[i] = __builtin_convertvector(reinterpret_cast<const bf16x4_t*>(src)[i], f32x4_t);
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = (float)src[i];
}

}  // namespace sumk

using namespace sumk;

extern "C" int sumk_cast_f32_bf16(const float* src, void* dst_bf16, int64_t n, void* stream) {
  SUMK_ARG(src && dst_bf16 && n > 0, "cast_f32_bf16: bad argument");
  SUMK_ARG(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst_bf16 & 7) == 0, "cast_f32_bf16: buffers must be 16- / 8-byte aligned");
  int blocks = (int)std::min<int64_t>((n / 4 + 255) / 256 + 1, 2048);
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, (__bf16*)dst_bf16, n);
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}
extern "C" int sumk_cast_bf16_f32(const void* src_bf16, float* dst, int64_t n, void* stream) {
  SUMK_ARG(src_bf16 && dst && n > 0, "cast_bf16_f32: bad argument");
  SUMK_ARG(((uintptr_t)dst & 15) == 0 && ((uintptr_t)src_bf16 & 7) == 0, "cast_bf16_f32: buffers must be 16- / 8-byte aligned");
  int blocks = (int)std::min<int64_t>((n / 4 + 255) / 256 + 1, 2048);
  hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const __bf16*)src_bf16, dst, n);
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}

extern "C" int sumk_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                              float beta1, float beta2, float eps, float weight_decay, int32_t step, float grad_scale,
                              void* stream) {
  SUMK_ARG(param && grad && exp_avg && exp_avg_sq, "adam: null pointer");
  SUMK_ARG(n > 0 && step >= 1, "adam: n=%lld step=%d", (long long)n, step);
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  int blocks = (int)std::min<int64_t>((n / 4 + 255) / 256 + 1, 2048);
  hipLaunchKernelGGL(adam_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, param, const_cast<float*>(grad), exp_avg, exp_avg_sq, n, lr,
                     beta1, beta2, eps, weight_decay, step_size, inv_sqrt_bc2, grad_scale, (const float*)nullptr);
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}

static int adam_step_dev_impl(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                             float beta1, float beta2, float eps, float weight_decay, int32_t* state, float grad_scale,
                             const float* sumsq, float max_norm, void* stream, bool zero_grad) {
  SUMK_ARG(param && grad && exp_avg && exp_avg_sq && state, "adam_dev: null pointer");
  SUMK_ARG(n > 0, "adam_dev: n=%lld", (long long)n);
  SUMK_ARG(sumsq == nullptr || max_norm > 0.f, "adam_dev: max_norm=%g with a norm given", (double)max_norm);
  hipLaunchKernelGGL(adam_prep_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, lr, beta1, beta2, grad_scale, sumsq, max_norm);
  int blocks = (int)std::min<int64_t>((n / 4 + 255) / 256 + 1, 2048);
  if (zero_grad) hipLaunchKernelGGL(adam_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n, lr,
                                    beta1, beta2, eps, weight_decay, 0.f, 0.f, 0.f, (const float*)state);
  else hipLaunchKernelGGL(adam_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n, lr,
                          beta1, beta2, eps, weight_decay, 0.f, 0.f, 0.f, (const float*)state);
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}
extern "C" int sumk_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                                  float beta1, float beta2, float eps, float weight_decay, int32_t* state, float grad_scale,
                                  const float* sumsq, float max_norm, void* stream) {
  return adam_step_dev_impl(param, const_cast<float*>(grad), exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, state, grad_scale, sumsq,
                            max_norm, stream, false);
}
extern "C" int sumk_adam_step_dev_zero_grad(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                                            float beta1, float beta2, float eps, float weight_decay, int32_t* state, float grad_scale,
                                            const float* sumsq, float max_norm, void* stream) {
  return adam_step_dev_impl(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, state, grad_scale, sumsq, max_norm, stream, true);
}

extern "C" size_t sumk_sumsq_workspace_bytes(void) { return 1024 * sizeof(float); }

extern "C" int sumk_sumsq(const float* v, int64_t n, float* out, void* workspace, void* stream) {
  SUMK_ARG(v && out && workspace, "sumsq: null pointer");
  SUMK_ARG(n > 0, "sumsq: n=%lld", (long long)n);
  int blocks = (int)std::min<int64_t>((n + 1023) / 1024, 1024);
  hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, v, n, (float*)workspace);
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const float*)workspace, blocks, out);
  SUMK_HIP(hipGetLastError());
  return SUMK_OK;
}
