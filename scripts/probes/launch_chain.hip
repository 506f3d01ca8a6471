// Probe: cost of a chain of dependent tiny launches on one stream (eager and hipGraph), and of small L2/MALL reads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
__global__ void empty_k(float* p) { if (p == nullptr) return; }
__global__ __launch_bounds__(256) void read_k(const float4* __restrict__ src, float* out, int per_block_f4) {
  float4 acc = make_float4(0, 0, 0, 0);
  const float4* s = src + (size_t)blockIdx.x * per_block_f4;
  for (int i = threadIdx.x; i < per_block_f4; i += 256) { float4 v = s[i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[blockIdx.x] = acc.x;
}
struct Big { float* p; const float* a[8]; int v[16]; };
__global__ void empty_big(Big b) { if (b.p == nullptr) return; }
int main() {
  float* d; CK(hipMalloc(&d, 64 << 20));
  CK(hipMemset(d, 0, 64 << 20));
  hipStream_t s; CK(hipStreamCreate(&s));
  const int N = 2000;
  auto time_loop = [&](auto launch, const char* name) {
    for (int i = 0; i < 50; ++i) launch();
    hipStreamSynchronize(s);
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int i = 0; i < N; ++i) launch();
    hipStreamSynchronize(s);
    double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / N;
    printf("%-44s %7.2f us/launch\n", name, us);
  };
  time_loop([&] { hipLaunchKernelGGL(empty_k, dim3(1), dim3(64), 0, s, d); }, "empty 1 block");
  time_loop([&] { hipLaunchKernelGGL(empty_k, dim3(128), dim3(256), 0, s, d); }, "empty 128 blocks x 256");
  Big b; b.p = d; time_loop([&] { hipLaunchKernelGGL(empty_big, dim3(128), dim3(256), 0, s, b); }, "empty 128 blocks, 136-B kernarg");
  for (int kb : {4, 32, 64, 256}) {
    char nm[64]; snprintf(nm, 64, "read %d KB/block x128 blocks", kb);
    time_loop([&] { hipLaunchKernelGGL(read_k, dim3(128), dim3(256), 0, s, (const float4*)d, d + (48 << 18), kb * 64); }, nm);
  }
  // graph of 320 empty launches
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int i = 0; i < 320; ++i) hipLaunchKernelGGL(empty_k, dim3(128), dim3(256), 0, s, d);
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int i = 0; i < 3; ++i) hipGraphLaunch(ge, s);
  hipStreamSynchronize(s);
  auto t0 = std::chrono::high_resolution_clock::now();
  for (int i = 0; i < 10; ++i) hipGraphLaunch(ge, s);
  hipStreamSynchronize(s);
  double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / 10 / 320;
  printf("%-44s %7.2f us/launch\n", "graph of 320 empty (128x256)", us);
  return 0;
}
