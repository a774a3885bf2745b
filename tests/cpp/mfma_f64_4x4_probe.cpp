// Probe: rate and layout of v_mfma_f64_4x4x4_4b_f64 (4 independent 4x4x4 blocks per instruction) on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
__global__ __launch_bounds__(256) void k_rate(double* out, int iters) {
  double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  for (int it = 0; it < iters; it++) {
    c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, a, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, b, c3, 0, 0, 0);
    c4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, a, c4, 0, 0, 0);
    c5 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c5, 0, 0, 0);
    c6 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, a, c6, 0, 0, 0);
    c7 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, b, c7, 0, 0, 0);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
}
__global__ void k_layout(const double* A, const double* B, double* C) {
  const int l = threadIdx.x;
  C[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], 0.0, 0, 0, 0);
}
int main() {
  double *dA, *dB, *dC, *out;
  std::vector<double> A(64), B(64), C(64);
  for (int i = 0; i < 64; i++) { A[i] = sin(i + 1.0); B[i] = cos(2.0 * i); }
  hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dC, 512); hipMalloc(&out, 8 * 256 * 4096);
  hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
  k_layout<<<1, 64>>>(dA, dB, dC);
  hipMemcpy(C.data(), dC, 512, hipMemcpyDeviceToHost);
  // hypothesis: lane l: block = l>>4 ; A[i = l&3][k = (l>>2)&3] ; B[k = (l>>2)&3][j = l&3] ; D[i = (l>>2)&3][j = l&3]
  for (int hyp = 0; hyp < 4; hyp++) {
    double err = 0;
    for (int l = 0; l < 64; l++) {
      const int b = l >> 4, lo = l & 3, hi = (l >> 2) & 3;
      double s = 0;
      for (int k = 0; k < 4; k++) {
        // A value of (block b, row i, k) sits in lane b*16 + k*4 + i ; B value of (k, col j) in lane b*16 + k*4 + j
        const int i = (hyp & 1) ? lo : hi, j = (hyp & 1) ? hi : lo;
        const double av = (hyp & 2) ? A[b * 16 + i * 4 + k] : A[b * 16 + k * 4 + i];
        const double bv = (hyp & 2) ? B[b * 16 + j * 4 + k] : B[b * 16 + k * 4 + j];
        s += av * bv;
      }
      err = fmax(err, fabs(s - C[l]));
    }
    printf("layout hypothesis %d: max err %.3e\n", hyp, err);
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int wg = 1; wg <= 4; wg *= 2) {
    const int grid = 256 * wg;
    k_rate<<<grid, 256>>>(out, 100);
    hipEventRecord(e0); k_rate<<<grid, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)grid * 4 * iters * 8 * 512.0;
    printf("rate: %d WG/CU: %.3f ms  %.1f TFLOP/s  (%.1f cycles/MFMA/SIMD at 2.4 GHz)\n", wg, ms, fl / ms * 1e-9, ms * 1e-3 * 2.4e9 / ((double)iters * 8 * wg));
  }
  return 0;
}
