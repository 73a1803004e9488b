// achievable HBM read (+write) rate of the shared-MLP access pattern: 256 persistent workgroups,
// each walking a contiguous column range of a (B, ROWS, R) fp32 tensor, reading RUN bytes of every
// row per step (and writing WROWS rows of the same width)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
template <int RUNF4>  // float4 per row per step (RUN = 16 * RUNF4 bytes)
__global__ void __launch_bounds__(256) k(const float4 *x, float4 *y, int rows, int wrows, size_t r4, int steps_total, int steps_per_cloud) {
  const int lanes_per_row = RUNF4, rows_per_pass = 256 / lanes_per_row;
  const int row0 = threadIdx.x / lanes_per_row, c = threadIdx.x % lanes_per_row;
  const int per = (steps_total + gridDim.x - 1) / gridDim.x;
  const int lo = blockIdx.x * per, hi = min(lo + per, steps_total);
  float4 acc = make_float4(0, 0, 0, 0);
  for (int s = lo; s < hi; ++s) {
    const int b = s / steps_per_cloud, col = (s % steps_per_cloud) * RUNF4;
    const float4 *xb = x + (size_t)b * rows * r4 + col + c;
    float4 v[8];
    for (int p0 = 0; p0 < rows; p0 += rows_per_pass * 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int row = p0 + u * rows_per_pass + row0; v[u] = row < rows ? xb[(size_t)row * r4] : make_float4(0,0,0,0); }
#pragma unroll
      for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    float4 *yb = y + (size_t)b * wrows * r4 + col + c;
    for (int p0 = 0; p0 < wrows; p0 += rows_per_pass) { const int row = p0 + row0; if (row < wrows) __builtin_nontemporal_store(acc.x, &yb[(size_t)row * r4].x), yb[(size_t)row * r4] = acc; }
  }
}
int main() {
  const int B = 8, ROWS = 384, WROWS = 128; const size_t R = 32768;
  float4 *x, *y; hipMalloc(&x, (size_t)B * ROWS * R * 4); hipMalloc(&y, (size_t)B * WROWS * R * 4);
  hipMemset(x, 0, (size_t)B * ROWS * R * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wr = 0; wr < 2; ++wr)
  for (int run = 0; run < 4; ++run) {
    const int f4 = 8 << run;  // 128, 256, 512, 1024 bytes
    const int spc = (int)(R / 4 / f4), total = B * spc;
    for (int it = 0; it < 3; ++it) {
      hipEventRecord(e0);
      const int wrows = wr ? WROWS : 0;
      if (run == 0) k<8><<<256, 256>>>(x, y, ROWS, wrows, R / 4, total, spc);
      if (run == 1) k<16><<<256, 256>>>(x, y, ROWS, wrows, R / 4, total, spc);
      if (run == 2) k<32><<<256, 256>>>(x, y, ROWS, wrows, R / 4, total, spc);
      if (run == 3) k<64><<<256, 256>>>(x, y, ROWS, wrows, R / 4, total, spc);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double bytes = (double)B * (ROWS + wrows) * R * 4;
      if (it == 2) printf("run %4d B  writes %d: %.1f us  %.2f TB/s\n", f4 * 16, wr, ms * 1e3, bytes / ms / 1e9);
    }
  }
}
