// Calibration of the FETCH_SIZE / WRITE_SIZE counters on gfx950 for the access mix of the SpMV kernel (k_spmv_lx): every kernel reads
// (or writes) a KNOWN number of distinct bytes exactly once, so counter x 1024 / bytes is the factor to apply to that access width.
//   k_read16    16-byte loads per lane (the double2 value stream)           k_read4     4-byte loads per lane (the ushort2 column stream)
//   k_read8     8-byte loads per lane, coalesced                            k_gather8   8-byte gathers through a SORTED index list with gaps
//   k_write8    8-byte stores per lane                                      (the tile's distinct x entries: 1 of every 2..5 entries)
// build: hipcc --offload-arch=gfx950 -O3 tests/cpp/fetch_calibration_probe.cpp -o /tmp/fetch_cal ; run under
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace ...   and   rocprofv3 --pmc WRITE_SIZE --kernel-trace ...   (tests/fetch_calibration.sh)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void k_read16(const double2* __restrict__ a, size_t n, double* out) {
  double s = 0.0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const double2 v = a[i]; s += v.x + v.y; }
  if (s == 1.2345e300) *out = s;
}
__global__ void k_read4(const ushort2* __restrict__ a, size_t n, double* out) {
  unsigned s = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const ushort2 v = a[i]; s += v.x + v.y; }
  if (s == 0xdeadbeefu) *out = s;
}
__global__ void k_read8(const double* __restrict__ a, size_t n, double* out) {
  double s = 0.0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += a[i];
  if (s == 1.2345e300) *out = s;
}
__global__ void k_gather8(const double* __restrict__ a, const int* __restrict__ idx, size_t n, double* out) {
  double s = 0.0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += a[idx[i]];
  if (s == 1.2345e300) *out = s;
}
__global__ void k_write8(double* __restrict__ a, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = (double)i;
}

int main() {
  const size_t bytes = (size_t)1 << 30;                       // 1 GiB per stream: far beyond L2 + MALL
  void *buf, *buf2;
  double* out;
  if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&buf2, bytes) != hipSuccess || hipMalloc(&out, 8) != hipSuccess) return 1;
  hipMemset(buf, 0, bytes);
  hipMemset(buf2, 0, bytes);
  // sorted gather list: from a 1 GiB array of doubles take entry k, then skip 1..4 entries (distinct, ascending): the line-level footprint is
  // what the hardware fetches, the USEFUL bytes are 8 per index
  const size_t ndbl = bytes / 8;
  std::vector<int> idx;
  idx.reserve(ndbl / 3);
  unsigned r = 12345u;
  for (size_t k = 0; k < ndbl;) {
    idx.push_back((int)k);
    r = r * 1664525u + 1013904223u;
    k += 2 + (r >> 30);                                        // step 2..5
  }
  int* didx;
  hipMalloc(&didx, idx.size() * sizeof(int));
  hipMemcpy(didx, idx.data(), idx.size() * sizeof(int), hipMemcpyHostToDevice);
  // distinct 64-byte and 128-byte lines the gather touches
  size_t l64 = 0, l128 = 0, last64 = (size_t)-1, last128 = (size_t)-1;
  for (int k : idx) {
    const size_t a = (size_t)k * 8;
    if (a / 64 != last64) { l64++; last64 = a / 64; }
    if (a / 128 != last128) { l128++; last128 = a / 128; }
  }
  const dim3 grid(256 * 8), block(256);
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(k_read16, grid, block, 0, 0, (const double2*)buf, bytes / 16, out);
    hipLaunchKernelGGL(k_read4, grid, block, 0, 0, (const ushort2*)buf2, bytes / 4, out);
    hipLaunchKernelGGL(k_read8, grid, block, 0, 0, (const double*)buf, bytes / 8, out);
    hipLaunchKernelGGL(k_gather8, grid, block, 0, 0, (const double*)buf2, didx, idx.size(), out);
    hipLaunchKernelGGL(k_write8, grid, block, 0, 0, (double*)buf, bytes / 8);
  }
  hipDeviceSynchronize();
  printf("CAL bytes_stream %zu gather_indices %zu gather_useful_bytes %zu gather_index_bytes %zu gather_lines64_bytes %zu gather_lines128_bytes %zu\n", bytes,
         idx.size(), idx.size() * 8, idx.size() * 4, l64 * 64, l128 * 128);
  return 0;
}
