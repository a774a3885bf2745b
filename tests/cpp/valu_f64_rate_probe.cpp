// Probe: issue rate of FP64 vector instructions on gfx950 (cycles per wave64 instruction per SIMD) at 1, 2, 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ __launch_bounds__(256) void k(double* out, int iters) {
  double v[16];
  for (int i = 0; i < 16; i++) v[i] = threadIdx.x + i;
  const double a = 0.999 + threadIdx.x * 1e-9, b = 1e-3;
  for (int it = 0; it < iters; it++)
#pragma unroll
    for (int i = 0; i < 16; i++) {
      if (KIND == 0) v[i] = fma(v[i], a, b);
      else if (KIND == 1) v[i] = v[i] * a;
      else v[i] = v[i] + b;
    }
  double r = 0;
  for (int i = 0; i < 16; i++) r += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int KIND>
void run(const char* name, double* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int wg = 1; wg <= 4; wg *= 2) {
    k<KIND><<<256 * wg, 256>>>(out, 10);
    hipEventRecord(e0); k<KIND><<<256 * wg, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ninstr = (double)iters * 16 * wg;   // per SIMD
    printf("%-8s %d waves/SIMD: %.3f ms  %.2f ns per wave-instruction per SIMD (%.1f cycles at 2.4 GHz)  %.1f T%s/s\n", name, wg, ms, ms * 1e6 / ninstr,
           ms * 1e-3 * 2.4e9 / ninstr, 256.0 * wg * 256 * iters * 16 * (KIND == 0 ? 2 : 1) / ms * 1e-9, KIND == 0 ? "FLOP" : "OP");
  }
}
int main() {
  double* out; hipMalloc(&out, 8 * 1024 * 256);
  run<0>("fma f64", out); run<1>("mul f64", out); run<2>("add f64", out);
  return 0;
}
