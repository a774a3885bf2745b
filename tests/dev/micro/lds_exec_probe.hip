// Microbenchmark: does the LDS pipe of gfx950 spend fewer cycles on a ds_read_b128 / ds_write_b64 when part of the wave is masked off?
// Every wave of a 512-thread workgroup (8 waves, as the cluster assembly kernel runs) issues N reads with `active` lanes enabled.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_exec_probe tests/dev/micro/lds_exec_probe.hip && /tmp/lds_exec_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>   // 0: ds_read_b128, 1: ds_read_b64, 2: ds_write_b64, 3: ds_write_b128, 4: two ds_write_b64, 5: ds_write_b64 at the staging stride
__global__ __launch_bounds__(512) void k_probe(double* out, int n, int active, unsigned long long* cycles) {
  __shared__ double sm[8 * 1024 + 64 * 29];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 8 * 1024; i += 512) sm[i] = i * 0.5;
  __syncthreads();
  double acc0 = 0.0, acc1 = 0.0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  if (lane < active) {
    const double* base = sm + wave * 1024 + lane * ((MODE == 0 || MODE == 3) ? 2 : (MODE == 5 ? 29 : 1));     // natural stride of the access width; 5: the staging's row stride
    for (int it = 0; it < n; it++) {
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int off = ((it + u) & 3) * 128;
        if (MODE == 0) {
          const double2 v = *reinterpret_cast<const double2*>(base + off);
          acc0 += v.x; acc1 += v.y;
        } else if (MODE == 1) {
          acc0 += base[off];
        } else if (MODE == 2 || MODE == 5) {
          const_cast<double*>(base)[MODE == 5 ? (off & 127) % 3 : off] = acc0 + it;
        } else if (MODE == 3) {
          *reinterpret_cast<double2*>(const_cast<double*>(base) + off) = make_double2(acc0 + it, acc1);
        } else {          // 4: two 8-byte stores to separate addresses (ds_write2_b64 if the compiler pairs them)
          const_cast<double*>(base)[off] = acc0 + it;
          const_cast<double*>(base)[off + 64] = acc1 + it;
        }
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) cycles[blockIdx.x * 8 + wave] = t1 - t0;
  out[(size_t)blockIdx.x * 512 + threadIdx.x] = acc0 + acc1 + (MODE == 2 ? sm[threadIdx.x] : 0.0);
}

template <int MODE>
static void run(const char* what) {
  const int nblk = 256 * 4, n = 2000;
  double* out; unsigned long long* cyc;
  hipMalloc(&out, (size_t)nblk * 512 * 8); hipMalloc(&cyc, nblk * 8 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int active : {64, 48, 32}) {
    hipLaunchKernelGGL(k_probe<MODE>, dim3(nblk), dim3(512), 0, 0, out, n, active, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_probe<MODE>, dim3(nblk), dim3(512), 0, 0, out, n, active, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per CU: 4 workgroups x 8 waves x n x 8 instructions
    const double instr_per_cu = 4.0 * 8 * n * 8;
    printf("%s active %2d lanes: %.3f ms  = %.2f ns per wave instruction per CU (%.1f cycles at 2.4 GHz)\n", what, active, ms, ms * 1e6 / instr_per_cu, ms * 1e6 / instr_per_cu * 2.4);
  }
  hipFree(out); hipFree(cyc);
}
int main() {
  run<0>("ds_read_b128");
  run<1>("ds_read_b64 ");
  run<2>("ds_write_b64");
  run<3>("ds_write_b128");
  run<4>("2 x ds_write_b64 (write2?)");
  run<5>("ds_write_b64 stride 29");
  return 0;
}
