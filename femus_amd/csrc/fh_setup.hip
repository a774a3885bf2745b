// Setup work that runs on the device but belongs to host-side modules (fh_mesh.cpp stays plain host C++ so that the sanitizer pass of
// tests/asan_host.sh can rebuild it without the device compiler): the prolongator builder of fh_build_prolongator.
#include "fh_internal.h"
#include <algorithm>
#include <vector>

// ---------------------------------------------------------------------------------------------------------------------
// The same prolongator built ON THE DEVICE (round 4).  The host loop above gives a fine row to the first (coarse element, child, local
// node) that visits it; in loop order that is the MINIMUM of the linear index (iel * nch + j) * nc + i over all visits, so the owner of a
// row is an atomicMin.  Row lengths come from a (child, node) table of non-zero counts, the host scans them, and one thread per row then
// writes its columns at their rank among the coarse element's dofs (sorted CSR order without a sort) with the boundary rule applied.
// Nothing but the 4-byte row lengths visits the host; the column copy there is fetched only if host code asks (fh_hcol).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int PL_NONE = 0x7f7f7f7f;   // a row nobody visits (byte pattern of the memset)
__global__ __launch_bounds__(256) void k_pl_owner(size_t n, int nc, int nl, const int* __restrict__ child, const int* __restrict__ f_ed, int* __restrict__ owner) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  const int slot = (int)(t / nc), i = (int)(t % nc);
  const int jel = child[slot];
  if (jel < 0) return;
  atomicMin(&owner[f_ed[(size_t)jel * nl + i]], (int)t);
}
__global__ __launch_bounds__(256) void k_pl_len(int nf, int nc, int nch, const int* __restrict__ owner, const char* __restrict__ refined, const int* __restrict__ cnt,
                                                int* __restrict__ len) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= nf) return;
  const int o = owner[r];
  if (o == PL_NONE) {
    len[r] = 0;
    return;
  }
  const int slot = o / nc, i = o % nc, iel = slot / nch, j = slot % nch;
  len[r] = refined[iel] ? cnt[j * nc + i] : 1;
}
__global__ __launch_bounds__(256) void k_pl_fill(int nf, int nc, int nch, int nl, const int* __restrict__ owner, const char* __restrict__ refined,
                                                 const int* __restrict__ c_ed, const int* __restrict__ cnt, const int* __restrict__ nzk,
                                                 const double* __restrict__ EP, const char* __restrict__ bf, const char* __restrict__ bc,
                                                 const int* __restrict__ rowptr, int* __restrict__ col, double* __restrict__ val) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= nf) return;
  const int o = owner[r];
  if (o == PL_NONE) return;
  const int slot = o / nc, i = o % nc, iel = slot / nch, j = slot % nch;
  const int* cd = c_ed + (size_t)iel * nl;
  const int p = rowptr[r];
  const bool rowb = bf && bf[r];
  if (!refined[iel]) {
    const int c = cd[i];
    col[p] = c;
    val[p] = (rowb || (bc && bc[c])) ? 0.0 : 1.0;
    return;
  }
  const int n = cnt[j * nc + i];
  const int* nz = nzk + (size_t)(j * nc + i) * nc;
  const double* pr = EP + (size_t)(j * nc + i) * nc;
  for (int a = 0; a < n; a++) {
    const int k = nz[a], c = cd[k];
    int rank = 0;
    for (int b = 0; b < n; b++) rank += cd[nz[b]] < c ? 1 : 0;
    col[p + rank] = c;
    val[p + rank] = (rowb || (bc && bc[c])) ? 0.0 : pr[k];     // pattern kept, value zeroed
  }
}

namespace {
struct DevBuf {   // scratch device arrays of one setup routine, freed on every exit path
  std::vector<void*> p;
  ~DevBuf() {
    for (void* q : p)
      if (q) hipFree(q);
  }
  template <class T>
  int get(T** out, size_t n) {
    void* q = nullptr;
    if (hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) return 1;
    p.push_back(q);
    *out = (T*)q;
    return 0;
  }
};
}   // namespace

int fh_prolongator_device(fh_ctx_t ctx, int nl, int nc, int nch, int nel_c, const int* child, const char* refined, const int* c_ed, size_t n_fed, const int* f_ed,
                          int nf, int ncc, const std::vector<double>& EP, const char* bf, const char* bc, fh_mat_t* out, const fh_mesh_dev* cdev,
                          const fh_mesh_dev* fdev) {
  std::vector<int> cnt((size_t)nch * nc, 0), nzk((size_t)nch * nc * nc, 0);
  for (int ji = 0; ji < nch * nc; ji++)
    for (int k = 0; k < nc; k++)
      if (EP[(size_t)ji * nc + k] != 0.0) nzk[(size_t)ji * nc + cnt[ji]++] = k;
  hipStream_t st = ctx->stream;
  DevBuf B;
  int *d_child, *d_fed, *d_ced, *d_owner, *d_len, *d_cnt, *d_nzk;
  char *d_ref, *d_bf = nullptr, *d_bc = nullptr;
  double* d_EP;
  const size_t nslot = (size_t)nel_c * nch;
  const bool resident = cdev && fdev;      // the element tables, child lists and split marks are in device memory already
  if ((!resident && (B.get(&d_child, nslot) || B.get(&d_fed, n_fed) || B.get(&d_ced, (size_t)nel_c * nl) || B.get(&d_ref, (size_t)nel_c))) ||
      B.get(&d_owner, (size_t)nf) || B.get(&d_len, (size_t)nf) || B.get(&d_cnt, cnt.size()) || B.get(&d_nzk, nzk.size()) || B.get(&d_EP, EP.size())) {
    fh_set_error("fh_build_prolongator: out of device memory");
    return 2;
  }
  if (resident) {
    d_child = cdev->d_child;
    d_ced = cdev->d_elem_dof;
    d_ref = cdev->d_refined;
    d_fed = fdev->d_elem_dof;
  } else {
    FH_CHECK_HIP(hipMemcpyAsync(d_child, child, nslot * sizeof(int), hipMemcpyHostToDevice, st));
    FH_CHECK_HIP(hipMemcpyAsync(d_fed, f_ed, n_fed * sizeof(int), hipMemcpyHostToDevice, st));
    FH_CHECK_HIP(hipMemcpyAsync(d_ced, c_ed, (size_t)nel_c * nl * sizeof(int), hipMemcpyHostToDevice, st));
    FH_CHECK_HIP(hipMemcpyAsync(d_ref, refined, (size_t)nel_c, hipMemcpyHostToDevice, st));
  }
  FH_CHECK_HIP(hipMemcpyAsync(d_cnt, cnt.data(), cnt.size() * sizeof(int), hipMemcpyHostToDevice, st));
  FH_CHECK_HIP(hipMemcpyAsync(d_nzk, nzk.data(), nzk.size() * sizeof(int), hipMemcpyHostToDevice, st));
  FH_CHECK_HIP(hipMemcpyAsync(d_EP, EP.data(), EP.size() * sizeof(double), hipMemcpyHostToDevice, st));
  FH_CHECK_HIP(hipMemsetAsync(d_owner, 0x7f, (size_t)nf * sizeof(int), st));          // PL_NONE
  if (bf) {
    if (B.get(&d_bf, (size_t)nf) || B.get(&d_bc, (size_t)ncc)) {
      fh_set_error("fh_build_prolongator: out of device memory");
      return 2;
    }
    FH_CHECK_HIP(hipMemcpyAsync(d_bf, bf, (size_t)nf, hipMemcpyHostToDevice, st));
    FH_CHECK_HIP(hipMemcpyAsync(d_bc, bc, (size_t)ncc, hipMemcpyHostToDevice, st));
  }
  const size_t nvis = nslot * nc;
  FH_REQUIRE(nvis < (size_t)PL_NONE, "fh_build_prolongator: %zu visits do not fit the owner index", nvis);
  if (nvis) hipLaunchKernelGGL(k_pl_owner, dim3((unsigned)((nvis + 255) / 256)), dim3(256), 0, st, nvis, nc, nl, d_child, d_fed, d_owner);
  if (nf) hipLaunchKernelGGL(k_pl_len, dim3(fh_div_up(nf, 256)), dim3(256), 0, st, nf, nc, nch, d_owner, d_ref, d_cnt, d_len);
  std::vector<int> rp((size_t)nf + 1, 0);
  if (nf) FH_CHECK_HIP(hipMemcpyAsync(rp.data() + 1, d_len, (size_t)nf * sizeof(int), hipMemcpyDeviceToHost, st));
  FH_CHECK_HIP(hipStreamSynchronize(st));
  int64_t tot = 0;
  for (int r = 0; r < nf; r++) {
    tot += rp[r + 1];
    rp[r + 1] = (int)tot;
  }
  FH_REQUIRE(tot < 2147483647ll, "fh_build_prolongator: nnz overflows int32");
  fh_mat_t P = nullptr;
  if (fh_mat_alloc_device_pattern(ctx, nf, ncc, std::move(rp), &P)) {
    fh_mat_destroy(P);
    return 2;
  }
  if (nf) hipLaunchKernelGGL(k_pl_fill, dim3(fh_div_up(nf, 256)), dim3(256), 0, st, nf, nc, nch, nl, d_owner, d_ref, d_ced, d_cnt, d_nzk, d_EP, d_bf, d_bc,
                             P->d_rowptr, P->d_col, P->d_val);
  FH_CHECK_HIP(hipGetLastError());
  FH_CHECK_HIP(hipStreamSynchronize(st));
  FH_TRY(fh_mat_build_rowblocks(P, ctx->spmv_tile));
  *out = P;
  return 0;
}
