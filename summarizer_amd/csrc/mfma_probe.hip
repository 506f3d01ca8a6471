// Measurement utility (no reference counterpart): the matrix-pipe rate this GPU SUSTAINS with nothing but MFMAs in flight -- one workgroup of eight
// waves per CU (two per SIMD, the occupancy of the GEMM / attention kernels of this library), four independent accumulators per wave, operands in
// registers, no memory traffic.  bench.py prints it beside the guide's peak (`mfma_sustained`, `roofline.frac_of_sustained`).  Measured on the MI355X boxes
// of this pool (profiles/r06_mfma_sustained_probe.txt): with ONE constant operand pair 2.0-2.4 PFLOP/s bf16 (32x32x16) and 157 TFLOP/s f32 = the guide's
// peaks; with pseudo-random operands, a different register pair per MFMA -- what real data looks like to the operand buses -- 1.7-1.8 PFLOP/s (32x32x16),
// 1.9-2.1 (16x16x32), f32 unchanged: the chip lowers its clock under the bf16 load, by how much depends on the data.
#include "sumk_internal.h"

namespace sumk {
namespace {

typedef __bf16 pb16x8 __attribute__((ext_vector_type(8)));
typedef float pf32x16 __attribute__((ext_vector_type(16)));

// kind 0: v_mfma_f32_32x32x16_bf16 (32 768 FLOP), 1: v_mfma_f32_32x32x2_f32 (4 096 FLOP), 2: v_mfma_f32_16x16x32_bf16 (16 384 FLOP)
// RANDOM: every MFMA of an iteration reads its own operand registers, filled with pseudo-random values (the operand buses toggle as under real data);
// else one constant operand pair feeds all of them
template <int KIND, bool RANDOM, bool BOTH = false>
__global__ __launch_bounds__(512) void mfma_rate_kernel(int iters, float seed, float* sink) {
  pf32x16 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const float v = seed + (float)(threadIdx.x & 7) * 0.125f;
  pb16x8 a16v[4], b16v[4];
  float af[4], bf[4];
  {
    unsigned h = (unsigned)(blockIdx.x * 512 + threadIdx.x) * 2654435761u + 12345u;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        h = h * 1664525u + 1013904223u; const float ra = RANDOM ? (float)((int)(h >> 8) - (1 << 23)) * (1.0f / (1 << 23)) : v + r;
        h = h * 1664525u + 1013904223u; const float rb = RANDOM ? (float)((int)(h >> 8) - (1 << 23)) * (1.0f / (1 << 23)) : v - r;
        a16v[q][r] = (__bf16)ra; b16v[q][r] = (__bf16)rb;
      }
      af[q] = RANDOM ? (float)a16v[q][0] * 1.37f + (float)a16v[q][1] : v; bf[q] = RANDOM ? (float)b16v[q][0] * 0.73f - (float)b16v[q][1] : v + 1.f;
    }
  }
  typedef float pf32x4 __attribute__((ext_vector_type(4)));
  pf32x4 c4[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u)          // 16 MFMAs per iteration, consecutive ones on different accumulators
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int qa = RANDOM ? (BOTH ? ((t + u) & 3) : u) : 0, qb = RANDOM ? t : 0;      // BOTH: the A operand changes with every MFMA too (else every fourth)
        if constexpr (KIND == 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a16v[qa], b16v[qb], acc[t], 0, 0, 0);
        else if constexpr (KIND == 1) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[qa], bf[qb], acc[t], 0, 0, 0);
        else c4[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a16v[qa], b16v[qb], c4[t], 0, 0, 0);
      }
  }
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[t][r];
#pragma unroll
    for (int r = 0; r < 4; ++r) s += c4[t][r];
  }
  if (s == 12345.678f) sink[0] = s;      // (never true in practice: keeps the accumulators alive)
}

}  // namespace
}  // namespace sumk

extern "C" int sumk_probe_mfma_rate(int32_t kind_in, int32_t iters, double* tflops, double* seconds, void* stream_) {
  const int kind = kind_in & 3, random = (kind_in >> 2) & 1, both = (kind_in >> 3) & 1;      // bit 2: pseudo-random operands, a different register pair per MFMA
  using namespace sumk;
  hipStream_t stream = (hipStream_t)stream_;
  SUMK_ARG(kind_in >= 0 && kind_in <= 14 && kind <= 2 && (!both || random) && iters >= 1 && tflops, "probe_mfma_rate: kind 0..2 (+ 4: random operands), iters >= 1");
  int dev = 0, cus = 0;
  SUMK_HIP(hipGetDevice(&dev));
  SUMK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  // (resources released on every path: the HIP calls below report through rc instead of returning early)
  float* sink = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  float ms = 0.f;
  hipError_t err = hipMalloc(&sink, 64);
  if (err == hipSuccess) err = hipEventCreate(&e0);
  if (err == hipSuccess) err = hipEventCreate(&e1);
  auto launch = [&]() {
#define SUMK_PROBE(K_, R_) hipLaunchKernelGGL((mfma_rate_kernel<K_, R_>), dim3(cus), dim3(512), 0, stream, iters, 0.5f, sink)
    if (kind == 0 && both) hipLaunchKernelGGL((mfma_rate_kernel<0, true, true>), dim3(cus), dim3(512), 0, stream, iters, 0.5f, sink);
    else if (kind == 2 && both) hipLaunchKernelGGL((mfma_rate_kernel<2, true, true>), dim3(cus), dim3(512), 0, stream, iters, 0.5f, sink);
    else if (kind == 0) { if (random) SUMK_PROBE(0, true); else SUMK_PROBE(0, false); }
    else if (kind == 1) { if (random) SUMK_PROBE(1, true); else SUMK_PROBE(1, false); }
    else { if (random) SUMK_PROBE(2, true); else SUMK_PROBE(2, false); }
#undef SUMK_PROBE
  };
  if (err == hipSuccess) {
    launch();                                  // warm-up: code object, clocks
    err = hipEventRecord(e0, stream);
    if (err == hipSuccess) { launch(); err = hipEventRecord(e1, stream); }
    if (err == hipSuccess) err = hipEventSynchronize(e1);
    if (err == hipSuccess) err = hipEventElapsedTime(&ms, e0, e1);
    if (err == hipSuccess) err = hipGetLastError();
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (sink) (void)hipFree(sink);
  if (err != hipSuccess) return hip_fail(err, "probe_mfma_rate");
  const double flop_per = kind == 0 ? 32768.0 : kind == 1 ? 4096.0 : 16384.0;
  const double flops = (double)cus * 8.0 * (double)iters * 16.0 * flop_per;
  *tflops = flops / ((double)ms * 1e-3) / 1e12;
  if (seconds) *seconds = (double)ms * 1e-3;
  return SUMK_OK;
}
