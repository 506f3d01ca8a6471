// Census probe: how does the dispatcher place the blocks of a launch over the CUs, and does a dynamic-LDS pad cap the residency?
// Each block records {XCC_ID, HW_ID, start, end} (s_memrealtime, 100 MHz) and spins for `spin_us`.
//   hipcc --offload-arch=gfx950 -O2 scripts/probes/census.hip -o /tmp/census && /tmp/census <grid> <static-like LDS pad bytes> <spin_us>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(256) void census_kernel(unsigned long long* out, int spin_ticks) {
  __shared__ float lds[4608];            // 18 KB static, like the 64x64 GEMM tile
  lds[threadIdx.x] = (float)threadIdx.x;
  __syncthreads();
  const unsigned hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_REG_HW_ID
  const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);     // HW_REG_XCC_ID
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long t1 = t0;
  float acc = lds[(threadIdx.x * 7) & 255];
  while ((long long)(t1 - t0) < spin_ticks) {
    for (int i = 0; i < 64; ++i) acc = acc * 1.0001f + 0.5f;
    t1 = __builtin_amdgcn_s_memrealtime();
  }
  if (threadIdx.x == 0) {
    out[blockIdx.x * 4 + 0] = xcc; out[blockIdx.x * 4 + 1] = hwid; out[blockIdx.x * 4 + 2] = t0; out[blockIdx.x * 4 + 3] = t1;
  }
  if (acc == 12345.678f) out[0] = 1;
}

int main(int argc, char** argv) {
  const int grid = argc > 1 ? atoi(argv[1]) : 919;
  const int pad = argc > 2 ? atoi(argv[2]) : 0;
  const int spin_us = argc > 3 ? atoi(argv[3]) : 50;
  unsigned long long* d;
  hipMalloc(&d, (size_t)grid * 32);
  std::vector<unsigned long long> h((size_t)grid * 4);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(census_kernel, dim3(grid), dim3(256), pad, 0, d, spin_us * 100);
    hipDeviceSynchronize();
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { printf("launch error: %s\n", hipGetErrorString(e)); return 1; }
  hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  std::map<unsigned long long, int> per_cu, per_cu_first;
  unsigned long long tmin = ~0ull, tmax = 0;
  for (int b = 0; b < grid; ++b) tmin = std::min(tmin, h[b * 4 + 2]), tmax = std::max(tmax, h[b * 4 + 3]);
  for (int b = 0; b < grid; ++b) {
    const unsigned xcc = (unsigned)h[b * 4] & 0xF, hw = (unsigned)h[b * 4 + 1];
    const unsigned cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
    const unsigned long long key = ((unsigned long long)xcc << 16) | (se << 8) | (sh << 4) | cu;
    per_cu[key]++;
    if (h[b * 4 + 2] - tmin < (unsigned long long)spin_us * 50) per_cu_first[key]++;   // started within the first half spin: co-resident
  }
  std::map<int, int> hist, hist_first;
  for (auto& kv : per_cu) hist[kv.second]++;
  for (auto& kv : per_cu_first) hist_first[kv.second]++;
  printf("grid %d pad %d: %zu distinct CUs; wall %.1f us\n  blocks per CU over the launch:", grid, pad, per_cu.size(), (tmax - tmin) / 100.0);
  for (auto& kv : hist) printf("  %d:%d", kv.first, kv.second);
  printf("\n  co-resident at start:");
  for (auto& kv : hist_first) printf("  %d:%d", kv.first, kv.second);
  {   // which block ids share a CU?  (first 6 CUs in key order)
    std::map<unsigned long long, std::vector<int>> ids;
    for (int b = 0; b < grid; ++b) {
      const unsigned xcc = (unsigned)h[b * 4] & 0xF, hw = (unsigned)h[b * 4 + 1];
      ids[((unsigned long long)xcc << 16) | (((hw >> 13) & 0x7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF)].push_back(b);
    }
    int shown = 0;
    printf("\n  block ids per CU:");
    for (auto& kv : ids) { if (shown++ >= 6) break; printf(" [xcc%llu:", kv.first >> 16); for (int b : kv.second) printf(" %d", b); printf("]"); }
    // is {b, b + 256, b + 512, ...} one CU?
    int same = 0, tot = 0;
    std::map<int, unsigned long long> cu_of;
    for (auto& kv : ids) for (int b : kv.second) cu_of[b] = kv.first;
    for (int b = 0; b + 256 < grid; ++b) { tot++; same += cu_of[b] == cu_of[b + 256]; }
    printf("\n  blocks b and b + 256 on the same CU: %d of %d", same, tot);
  }
  printf("\n  first blocks: ");
  for (int b = 0; b < 12 && b < grid; ++b) printf("[b%d xcc%llu hw%08llx] ", b, h[b * 4] & 0xF, h[b * 4 + 1]);
  printf("\n");
  return 0;
}
