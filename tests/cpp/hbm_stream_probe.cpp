// measurement probe (not product): what a pure streaming kernel reaches on this device -- read-only sum of a 1.7 GB buffer
// (the traffic shape of an SpMV: almost all reads) and a copy (half reads, half writes) -- to put the SpMV's GB/s next to the
// practical HBM ceiling rather than only next to the 8 TB/s datasheet figure.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void k_read(const double2* __restrict__ a, size_t n2, double* __restrict__ out) {
  double s = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) {
    const double2 v = a[i];
    s += v.x + v.y;
  }
  if (s == 12345.678) out[0] = s;   // never true: keeps the loads
}
__global__ __launch_bounds__(256) void k_copy(const double2* __restrict__ a, double2* __restrict__ b, size_t n2) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) b[i] = a[i];
}

int main() {
  const size_t bytes = 1700000000ull & ~15ull, n2 = bytes / 16;
  double2 *a, *b;
  double* out;
  hipMalloc(&a, bytes);
  hipMalloc(&b, bytes);
  hipMalloc(&out, 8);
  hipMemset(a, 0, bytes);
  hipMemset(b, 0, bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms;
  printf("{");
  for (int blocks_per_cu : {4, 8, 16, 32}) {
    const int grid = 256 * blocks_per_cu;
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, n2, out);
    hipEventRecord(e0);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, n2, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("\"read_%dblk_per_cu_GBps\": %.0f, ", blocks_per_cu, bytes / (ms / 20) / 1e6);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n2);
    hipEventRecord(e0);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n2);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("\"copy_%dblk_per_cu_GBps\": %.0f%s", blocks_per_cu, 2.0 * bytes / (ms / 20) / 1e6, blocks_per_cu == 32 ? "}\n" : ", ");
  }
  return 0;
}
