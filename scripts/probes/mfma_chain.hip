// Matrix-pipe probe: W waves per SIMD, each issuing v_mfma_f32_32x32x2_f32 on C independent accumulator chains, no memory traffic.
// Does a single dependent chain per wave (the 64x64 GEMM tile: one 32x32 accumulator per wave) reach the pipe's rate when 3-4 such
// waves share a SIMD?   hipcc --offload-arch=gfx950 -O3 scripts/probes/mfma_chain.hip -o /tmp/mfma_chain && /tmp/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int C, int GAP>
__global__ __launch_bounds__(256) void chain_kernel(float* out, int iters, float a0, float b0) {
  f32x16 acc[C];
#pragma unroll
  for (int c = 0; c < C; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
  float junk = a;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      acc[j % C] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j % C], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < GAP; ++g) junk = junk * 1.0001f + 0.25f;     // VALU filler between MFMAs
    }
  }
  float s = junk;
#pragma unroll
  for (int c = 0; c < C; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[c][r];
  if (s == 12345.678f) out[0] = s;
}

template <int C, int GAP>
void run(int waves_per_simd, float* out) {
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((chain_kernel<C, GAP>), dim3(256 * waves_per_simd), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double flops = 256.0 * waves_per_simd * 4 * (double)iters * 16 * 32 * 32 * 2 * 2;
  printf("chains/wave %d, VALU fillers per MFMA %d, waves/SIMD %d: %.2f ms, %.1f TFLOP/s (%.0f %% of 157.3)\n", C, GAP, waves_per_simd, best,
         flops / best / 1e9, flops / best / 1e9 / 157.3 * 100);
}

int main() {
  float* out; hipMalloc(&out, 4096);
  for (int w = 1; w <= 4; ++w) run<1, 0>(w, out);
  for (int w = 1; w <= 4; ++w) run<2, 0>(w, out);
  for (int w = 1; w <= 3; ++w) run<4, 0>(w, out);
  for (int w = 1; w <= 4; ++w) run<1, 8>(w, out);
  for (int w = 1; w <= 4; ++w) run<2, 8>(w, out);
  return 0;
}
