// Load-path probe: the operand fetch pattern of the 64x64 per-video GEMM tiles WITHOUT the arithmetic.  Each block walks `iters`
// k-tiles of a (128 rows x 32 floats) slab: thread t loads float4 at row (t / 8 + 32 p), floats (t % 8) * 4 .. +3 of the k-tile (8 rows x
// 128 B per wave-instruction), rows `ld` floats apart.  Reports GB/s per CU for several row strides / footprints.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/rowload.hip -o /tmp/rowload && /tmp/rowload
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int MODE>   // 0: wait for each k-tile's loads before issuing the next (one k-tile in flight); 1: two k-tiles in flight
__global__ __launch_bounds__(256) void rowload_kernel(const float* __restrict__ A, float* out, int ld, int n_row_blocks, int k_tiles, int tiles_per_block) {
  __shared__ float lds[4608];
  const int tid = threadIdx.x;
  float acc = 0.f;
  for (int t = 0; t < tiles_per_block; ++t) {
    const int tile = blockIdx.x + t * gridDim.x;
    const int rb = (tile * 7) % n_row_blocks;                 // a "video row block" of 128 rows
    const float* base = A + (size_t)rb * 128 * ld + (size_t)(tid / 8) * ld + (tid % 8) * 4;
    float4 r[4], q[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) r[p] = *reinterpret_cast<const float4*>(base + (size_t)32 * p * ld);
    for (int kt = 1; kt <= k_tiles; ++kt) {
      const int k0 = (kt < k_tiles ? kt : 0) * 32;
#pragma unroll
      for (int p = 0; p < 4; ++p) q[p] = *reinterpret_cast<const float4*>(base + (size_t)32 * p * ld + k0);
      if (MODE == 0) {
#pragma unroll
        for (int p = 0; p < 4; ++p) { lds[(tid * 4 + p * 1024) % 4608] = r[p].x + r[p].y + r[p].z + r[p].w; }
        __syncthreads();
        acc += lds[(tid * 13) % 4608];
        __syncthreads();
      } else {
#pragma unroll
        for (int p = 0; p < 4; ++p) acc += r[p].x + r[p].y + r[p].z + r[p].w;
      }
#pragma unroll
      for (int p = 0; p < 4; ++p) r[p] = q[p];
    }
  }
  if (acc == 12345.678f) out[0] = acc;
}

int main() {
  const int n_row_blocks = 96;            // 96 x 128 rows = 12288 rows, like the S-TVSum batch
  const int k_tiles = 32;
  float* out; hipMalloc(&out, 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int ld : {1024, 1056, 3072, 3104, 3136}) {
    float* A; size_t bytes = (size_t)n_row_blocks * 128 * ld * 4;
    hipMalloc(&A, bytes); hipMemset(A, 0, bytes);
    for (int grid : {768, 1024}) {
      for (int mode = 0; mode < 2; ++mode) {
        const int tiles_per_block = 4;
        for (int rep = 0; rep < 3; ++rep) {
          hipEventRecord(e0);
          if (mode == 0) hipLaunchKernelGGL(rowload_kernel<0>, dim3(grid), dim3(256), 22528 - (grid == 768 ? -12288 : 0), 0, A, out, ld, n_row_blocks, k_tiles, tiles_per_block);
          else hipLaunchKernelGGL(rowload_kernel<1>, dim3(grid), dim3(256), 22528 - (grid == 768 ? -12288 : 0), 0, A, out, ld, n_row_blocks, k_tiles, tiles_per_block);
          hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double gb = (double)grid * tiles_per_block * k_tiles * 16384.0 / 1e9;
        printf("ld %5d (row stride %6d B, footprint %4zu MB) grid %4d mode %d: %.1f us, %.2f TB/s, %.1f GB/s per CU, %.1f B/clk/CU @2.25GHz\n", ld, ld * 4,
               bytes >> 20, grid, mode, ms * 1e3, gb / ms, gb / ms * 1e3 / 256, gb / ms * 1e3 / 256 / 2.25);
      }
    }
    hipFree(A);
  }
  return 0;
}
