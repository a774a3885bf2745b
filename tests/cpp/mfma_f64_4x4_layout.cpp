#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(int* rec) {   // rec[l*8 + t] = (pa<<8|pb) pairs feeding output lane l
  const int l = threadIdx.x;
  int cnt = 0;
  for (int pa = 0; pa < 64; pa++)
    for (int pb = 0; pb < 64; pb++) {
      double d = __builtin_amdgcn_mfma_f64_4x4x4f64(l == pa ? 1.0 : 0.0, l == pb ? 1.0 : 0.0, 0.0, 0, 0, 0);
      if (d != 0.0 && cnt < 8) rec[l * 8 + cnt++] = (pa << 8) | pb;
    }
  for (; cnt < 8; cnt++) rec[l * 8 + cnt] = -1;
}
int main() {
  int* d; hipMalloc(&d, 64 * 8 * 4);
  k<<<1, 64>>>(d);
  std::vector<int> h(512);
  hipMemcpy(h.data(), d, 2048, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; l++) {
    printf("D lane %2d <-", l);
    for (int t = 0; t < 8; t++) if (h[l * 8 + t] >= 0) printf(" (A%d,B%d)", h[l * 8 + t] >> 8, h[l * 8 + t] & 255);
    printf("\n");
  }
  return 0;
}
