// Probe: does a producer -> consumer pair through a 1.5 GB buffer get cheaper when run in chunks that fit the 256 MB Infinity Cache?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k_write(double2* p, size_t n, double v) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = make_double2(v + i, v);
}
__global__ __launch_bounds__(256) void k_read(const double2* p, size_t n, double* out) {
  double s = 0;
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { double2 a = p[i]; s += a.x + a.y; }
  if (s == 1.2345) out[0] = s;
}
int main() {
  const size_t bytes = 1536ull << 20, n = bytes / 16;
  double2* buf; double* out;
  hipMalloc(&buf, bytes); hipMalloc(&out, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int chunks : {1, 4, 8, 16, 32, 64}) {
    const size_t cn = n / chunks;
    float best = 1e9;
    for (int rep = 0; rep < 4; rep++) {
      hipEventRecord(e0);
      for (int c = 0; c < chunks; c++) {
        k_write<<<2048, 256>>>(buf + c * cn, cn, 1.0 + rep);
        k_read<<<2048, 256>>>(buf + c * cn, cn, out);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("chunks %2d (%4zu MB each): write+read of 1.5 GB: %.3f ms  (%.2f TB/s over 3 GB)\n", chunks, (bytes / chunks) >> 20, best, 2.0 * bytes / best * 1e-9);
  }
  return 0;
}
