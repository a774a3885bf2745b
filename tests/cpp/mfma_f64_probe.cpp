// Probe: rate and fragment layout of v_mfma_f64_16x16x4_f64 on gfx950 (what the element-matrix kernel is built on).
//   hipcc --offload-arch=gfx950 -O3 tests/cpp/mfma_f64_probe.cpp -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_rate(double* out, int iters, int nacc) {
  d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  for (int it = 0; it < iters; it++) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    if (nacc > 1) c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    if (nacc > 2) c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, c2, 0, 0, 0);
    if (nacc > 3) c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, b, c3, 0, 0, 0);
  }
  d4 s = c0 + c1 + c2 + c3;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

// MFMA (3 accumulators) with `nv` independent FP64 FMAs per MFMA triple on the VALU of the same wave
__global__ __launch_bounds__(256) void k_mix(double* out, int iters, int nv) {
  d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  double v0 = a, v1 = b, v2 = a + b, v3 = a - b;
  for (int it = 0; it < iters; it++) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, c2, 0, 0, 0);
    for (int k = 0; k < nv; k += 4) {
      v0 = fma(v0, 0.999, 1e-3); v1 = fma(v1, 0.999, 1e-3); v2 = fma(v2, 0.999, 1e-3); v3 = fma(v3, 0.999, 1e-3);
    }
  }
  d4 s = c0 + c1 + c2;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + v0 + v1 + v2 + v3;
}

__global__ void k_layout(const double* A, const double* B, double* C) {   // C(16x16) = A(16x4) B(4x16), row-major
  const int l = threadIdx.x;
  d4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], c, 0, 0, 0);
  for (int r = 0; r < 4; r++) C[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];
}

int main() {
  double *dA, *dB, *dC, *out;
  std::vector<double> A(64), B(64), C(256);
  for (int i = 0; i < 64; i++) { A[i] = sin(i + 1.0); B[i] = cos(2.0 * i); }
  hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dC, 2048); hipMalloc(&out, 8 * 256 * 4096);
  hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
  k_layout<<<1, 64>>>(dA, dB, dC);
  hipMemcpy(C.data(), dC, 2048, hipMemcpyDeviceToHost);
  double err = 0;
  for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) {
    double s = 0; for (int k = 0; k < 4; k++) s += A[i * 4 + k] * B[k * 16 + j];
    err = fmax(err, fabs(s - C[i * 16 + j]));
  }
  printf("layout max err %.3e\n", err);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int wg = 1; wg <= 4; wg *= 2)
    for (int nacc = 1; nacc <= 4; nacc++) {
      const int grid = 256 * wg;
      k_rate<<<grid, 256>>>(out, 100, nacc);
      hipEventRecord(e0); k_rate<<<grid, 256>>>(out, iters, nacc); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double fl = (double)grid * 4 * iters * nacc * 2048.0;
      printf("rate: %d WG/CU(4 waves) nacc %d: %.3f ms  %.1f TFLOP/s  (%.1f cycles/MFMA/SIMD at 2.4 GHz)\n", wg, nacc, ms, fl / ms * 1e-9,
             ms * 1e-3 * 2.4e9 / ((double)iters * nacc * wg));
    }
  for (int wg = 1; wg <= 2; wg++)
    for (int nv = 0; nv <= 48; nv += 8) {
      const int grid = 256 * wg;
      k_mix<<<grid, 256>>>(out, 100, nv);
      hipEventRecord(e0); k_mix<<<grid, 256>>>(out, iters, nv); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("mix: %d WG/CU, 3 MFMA + %2d FMA: %.3f ms  mfma %.1f TF + valu %.1f TF\n", wg, nv, ms, (double)grid * 4 * iters * 3 * 2048.0 / ms * 1e-9,
             (double)grid * 256 * iters * nv * 2.0 / ms * 1e-9);
    }
  return 0;
}
