// Microbenchmark: what does moving a double between lanes cost on gfx950 WITHOUT going through LDS memory -- the price of the verdict's direction (a) for the
// element phase (stages 2-3 of the sum factorisation as MFMA products with the fragments permuted in registers instead of 64 ds_read_b128 per element)?
// Eight waves per workgroup, as the cluster assembly kernel runs; every wave issues N exchanges of a 64-bit value (two 32-bit halves), either as a dependent
// chain (latency) or as eight independent chains (throughput).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lane_exchange_probe tests/dev/micro/lane_exchange_probe.hip && /tmp/lane_exchange_probe
#include <hip/hip_runtime.h>
#include <cstdio>

// MODE 0: v_permlane32_swap (lanes 0-31 of one register <-> lanes 32-63 of another), 1: v_permlane16_swap, 2: DPP row_shr:1 (v_mov_b32_dpp),
//      3: ds_bpermute_b32 (the LDS crossbar, no memory), 4: ds_swizzle (butterfly), 5: ds_read_b64 from LDS memory (what the kernel does now, for reference)
template <int MODE>
__device__ __forceinline__ double exchange(double v, double w, const double* sm, int lane) {
  unsigned lo = __double2loint(v), hi = __double2hiint(v), lo2 = __double2loint(w), hi2 = __double2hiint(w);
  if (MODE == 0) {
    auto a = __builtin_amdgcn_permlane32_swap(lo, lo2, false, false);
    auto b = __builtin_amdgcn_permlane32_swap(hi, hi2, false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
  } else if (MODE == 1) {
    auto a = __builtin_amdgcn_permlane16_swap(lo, lo2, false, false);
    auto b = __builtin_amdgcn_permlane16_swap(hi, hi2, false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
  } else if (MODE == 2) {
    const int a = __builtin_amdgcn_update_dpp(0, (int)lo, 0x111, 0xf, 0xf, true);
    const int b = __builtin_amdgcn_update_dpp(0, (int)hi, 0x111, 0xf, 0xf, true);
    return __hiloint2double(b, a) + w;
  } else if (MODE == 3) {
    const int idx = ((lane + 17) & 63) * 4;
    const int a = __builtin_amdgcn_ds_bpermute(idx, (int)lo);
    const int b = __builtin_amdgcn_ds_bpermute(idx, (int)hi);
    return __hiloint2double(b, a) + w;
  } else if (MODE == 4) {
    const int a = __builtin_amdgcn_ds_swizzle((int)lo, 0x041f);      // swap with the neighbour lane
    const int b = __builtin_amdgcn_ds_swizzle((int)hi, 0x041f);
    return __hiloint2double(b, a) + w;
  } else {
    return sm[(lane * 2 + (int)(v == 12345.0)) & 1023] + w;
  }
}

template <int MODE, bool CHAIN>
__global__ __launch_bounds__(512) void k_probe(double* out, int n) {
  __shared__ double sm[8 * 1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 8 * 1024; i += 512) sm[i] = i * 1e-6;
  __syncthreads();
  const double* base = sm + wave * 1024;
  double a[8];
#pragma unroll
  for (int u = 0; u < 8; u++) a[u] = lane * 0.25 + u;
  for (int it = 0; it < n; it++) {
    if (CHAIN) {
#pragma unroll
      for (int u = 0; u < 8; u++) a[0] = exchange<MODE>(a[0], a[1], base, lane);          // eight dependent exchanges
    } else {
#pragma unroll
      for (int u = 0; u < 8; u++) a[u] = exchange<MODE>(a[u], a[(u + 1) & 7], base, lane);   // eight exchanges whose inputs are a step old
    }
  }
  double s = 0.0;
#pragma unroll
  for (int u = 0; u < 8; u++) s += a[u];
  out[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE>
static void run(const char* what) {
  const int nblk = 256 * 4, n = 2000;
  double* out;
  hipMalloc(&out, (size_t)nblk * 512 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float ms[2];
  for (int chain = 0; chain < 2; chain++) {
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      if (chain) hipLaunchKernelGGL((k_probe<MODE, true>), dim3(nblk), dim3(512), 0, 0, out, n);
      else hipLaunchKernelGGL((k_probe<MODE, false>), dim3(nblk), dim3(512), 0, 0, out, n);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms[chain], e0, e1);
    }
  }
  // per CU: 4 workgroups x 8 waves x n x 8 exchanges of one double (each: the exchange instructions of both halves + one v_add_f64)
  const double per_cu = 4.0 * 8 * n * 8;
  printf("%-34s independent: %6.2f cycles per exchanged double and CU;  dependent chain: %6.2f (at 2.4 GHz; one v_add_f64 included)\n", what,
         ms[0] * 1e6 / per_cu * 2.4, ms[1] * 1e6 / per_cu * 2.4);
  hipFree(out);
}
int main() {
  run<0>("v_permlane32_swap x 2");
  run<1>("v_permlane16_swap x 2");
  run<2>("v_mov_b32_dpp row_shr:1 x 2");
  run<3>("ds_bpermute_b32 x 2");
  run<4>("ds_swizzle_b32 x 2");
  run<5>("ds_read_b64 (LDS memory)");
  return 0;
}
