// Probe: does v_mfma_f64_16x16x4_f64 overlap with vector work of OTHER waves on the same SIMD?
// Workgroup of 512 threads = 8 waves = 2 per SIMD: even waves issue only MFMAs (4 independent accumulators), odd waves only
// vector instructions of the chosen kind.  Times: MFMA waves alone, vector waves alone, both.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int KIND>   // 0: f64 fma, 1: f32 fma, 2: int mad, 3: LDS reads
__global__ __launch_bounds__(512) void k(double* out, int iters_m, int iters_v) {
  __shared__ double lds[4096];
  const int wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = i;
  __syncthreads();
  double r = 0;
  if ((wave & 4) == 0) {       // waves 0-3: one per SIMD
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters_m; it++) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, b, c3, 0, 0, 0);
    }
    d4 s = c0 + c1 + c2 + c3;
    r = s[0] + s[1] + s[2] + s[3];
  } else {
    if (KIND == 0) {
      double v[16];
      for (int k = 0; k < 16; k++) v[k] = threadIdx.x + k;
      for (int it = 0; it < iters_v; it++)
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = fma(v[k], 0.999, 1e-3);
      for (int k = 0; k < 16; k++) r += v[k];
    } else if (KIND == 1) {
      float v[16];
      for (int k = 0; k < 16; k++) v[k] = threadIdx.x + k;
      for (int it = 0; it < iters_v; it++)
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = fmaf(v[k], 0.999f, 1e-3f);
      for (int k = 0; k < 16; k++) r += v[k];
    } else if (KIND == 2) {
      int v[16];
      for (int k = 0; k < 16; k++) v[k] = threadIdx.x + k;
      for (int it = 0; it < iters_v; it++)
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = v[k] * 3 + it;
      for (int k = 0; k < 16; k++) r += v[k];
    } else {
      int idx = threadIdx.x & 63;
      for (int it = 0; it < iters_v; it++) {
#pragma unroll
        for (int k = 0; k < 16; k++) r += lds[(idx + k * 64 + it) & 4095];
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int KIND>
void run(const char* name, double* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int im = 10000, iv = 10000;
  float ms[3];
  const int cfg[3][2] = {{im, 0}, {0, iv}, {im, iv}};
  for (int c = 0; c < 3; c++) {
    k<KIND><<<256, 512>>>(out, 10, 10);
    hipEventRecord(e0); k<KIND><<<256, 512>>>(out, cfg[c][0], cfg[c][1]); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms[c], e0, e1);
  }
  printf("%-10s mfma alone %.3f ms (%.1f TF), vector alone %.3f ms, both %.3f ms  -> %s\n", name, ms[0], 256.0 * 4 * im * 4 * 2048 / ms[0] * 1e-9, ms[1], ms[2],
         ms[2] < 0.6 * (ms[0] + ms[1]) + 0.4 * (ms[0] > ms[1] ? ms[0] : ms[1]) ? "overlap" : "serialised");
}

int main() {
  double* out; hipMalloc(&out, 8 * 256 * 512);
  run<0>("f64 fma", out); run<1>("f32 fma", out); run<2>("int mad", out); run<3>("lds read", out);
  return 0;
}
