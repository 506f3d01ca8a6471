// Probe: what does the fp32 MFMA pipe sustain on this box, and what does each ingredient of the GEMM loop cost?
//   mode 0: 64 MFMAs per iteration from registers only (4 accumulators)            -> pipe ceiling + clock under MFMA load
//   mode 1: + the GEMM's 16 ds_read_b128 fragment reads per iteration (conflict-free image)
//   mode 2: + one workgroup barrier per iteration
//   mode 3: + 8 LDS-DMA pieces per wave per iteration streaming a buffer from L2/HBM (the GEMM's operand traffic)
// Prints TFLOP/s (wall), the in-kernel clock (s_memtime / s_memrealtime) and cycles per MFMA per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 -o mfma_f32_ceiling mfma_f32_ceiling.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_void_p;
typedef const __attribute__((address_space(1))) void* glb_void_p;

template <int MODE>
__global__ __launch_bounds__(256, 2) void probe(const float* __restrict__ src, size_t src_floats, float* out, unsigned long long* stamps,
                                                int iters) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 8192];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 2 * 8192; i += 256) lds[i] = 1e-3f * (float)((i * 7 + blockIdx.x) % 13);
  __syncthreads();
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float a[2][4] = {{1.f + lane, 2.f, 3.f, 4.f}, {0.5f, 0.25f * lane, 2.f, 1.f}}, b[2][4] = {{1.f, 0.5f, 2.f, 3.f}, {0.1f, 0.2f, 0.3f + lane, 1.f}};
  const int li = lane & 31, lh = lane >> 5;
  const int e = lh ^ ((li >> 1) & 7);
  size_t goff = ((size_t)blockIdx.x * 256 + tid) * 4;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  int st = 0;
  for (int it = 0; it < iters; ++it) {
    if (MODE >= 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own DMAs of the previous iteration have landed
    if (MODE >= 2) __syncthreads();
    if (MODE >= 3) {
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const float* s = src + (goff & (src_floats - 1));   // src_floats is a power of two
        goff += (size_t)gridDim.x * 1024;
        // asm form: hipcc does not count it, so it puts no vmcnt(0) in front of the fragment reads below (the builtin does:
        // an LDS-DMA is a may-alias LDS store to the compiler)
        const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_void_p)(lds + (st ^ 1) * 8192 + (p * 4 + wave) * 256));
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(s), "s"(dst) : "memory");
      }
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if (MODE >= 1) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const float4 va = *reinterpret_cast<const float4*>(&lds[st * 8192 + (wave / 2 * 64 + t * 32 + li) * 32 + 4 * ((2 * kk) ^ e)]);
          const float4 vb = *reinterpret_cast<const float4*>(&lds[st * 8192 + 4096 + (wave % 2 * 64 + t * 32 + li) * 32 + 4 * ((2 * kk) ^ e)]);
          a[t][0] = va.x; a[t][1] = va.y; a[t][2] = va.z; a[t][3] = va.w;
          b[t][0] = vb.x; b[t][1] = vb.y; b[t][2] = vb.z; b[t][3] = vb.w;
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
          for (int tn = 0; tn < 2; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][j], b[tn][j], acc[tm][tn], 0, 0, 0);
    }
    if (MODE >= 3) st ^= 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[(size_t)blockIdx.x * 256 + tid] = s;
  if (lane == 0) { stamps[((size_t)blockIdx.x * 4 + wave) * 2] = t1 - t0; stamps[((size_t)blockIdx.x * 4 + wave) * 2 + 1] = r1 - r0; }
}

template <int MODE>
static void run(const float* src, size_t n, float* out, unsigned long long* st, int blocks, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, src, n, out, st, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int reps = 5;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, src, n, out, st, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  std::vector<unsigned long long> h((size_t)blocks * 8);
  hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> clk, cyc;
  for (int i = 0; i < blocks * 4; ++i) { clk.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 100.0); cyc.push_back((double)h[2 * i]); }
  std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
  const double flops = (double)blocks * 4 * iters * 64 * 4096.0;
  const int wps = blocks * 4 / 1024 > 0 ? blocks * 4 / 1024 : 1;   // waves per SIMD
  printf("mode %d  blocks %d (%d waves/SIMD) iters %d: %.3f ms  %.1f TFLOP/s  clock median %.0f MHz  wave cycles median %.0f -> %.1f cycles per MFMA per SIMD\n",
         MODE, blocks, wps, iters, ms, flops / ms / 1e9, clk[clk.size() / 2], cyc[cyc.size() / 2], cyc[cyc.size() / 2] / ((double)iters * 64 * wps));
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  const size_t n_alloc = (size_t)64 << 20;   // 256 MB source
  float *src, *out; unsigned long long* st;
  hipMalloc(&src, n_alloc * 4);
  {   // RANDOM operand data: zero / trivial operands let the chip hold a higher clock than real data does (MI355X_MICROARCH.md, DVFS)
    std::vector<float> h(n_alloc);
    unsigned long long x = 88172645463325252ull;
    for (size_t i = 0; i < n_alloc; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = (float)((double)(x >> 11) / 9007199254740992.0 * 2.0 - 1.0); }
    if (argc > 2 && atoi(argv[2]) == 0) std::fill(h.begin(), h.end(), 0.f);
    hipMemcpy(src, h.data(), n_alloc * 4, hipMemcpyHostToDevice);
  } hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&st, 1024 * 8 * 8);
  for (int blocks : {512}) {
    const size_t n = n_alloc;
    run<0>(src, n, out, st, blocks, iters); run<1>(src, n, out, st, blocks, iters);
    run<2>(src, n, out, st, blocks, iters); run<3>(src, n, out, st, blocks, iters);
  }
  // mode 3 again with the streamed region shrunk: 256 MB (HBM), 64 MB (Infinity Cache), 16 MB (all L2s together), 2 MB (one L2)
  for (size_t mb : {256, 64, 16, 2}) {
    printf("-- streamed region %zu MB\n", mb);
    run<3>(src, mb << 18, out, st, 512, iters);
  }
  return 0;
}
