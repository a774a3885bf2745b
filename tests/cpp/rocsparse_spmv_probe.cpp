// comparison probe (not a test, not product): the vendor library's CSR SpMV (rocSPARSE csrmv, with and without its adaptive
// analysis) and this library's fh_spmv on the SAME fine-level matrix of config C2 (64^3 HEX27/Q2, 135 M non-zeros), same
// device, HIP-event timing, algorithmic bytes of SURVEY 8(d).    usage: rocsparse_spmv_probe [coarse n = 8] [levels = 4]
#include <hip/hip_runtime.h>
#include <rocsparse/rocsparse.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../include/femus_hip.h"

#define CK(x) do { if ((x) != 0) { fprintf(stderr, "failed: %s (%s)\n", #x, fh_last_error()); return 1; } } while (0)
#define RS(x) do { rocsparse_status s_ = (x); if (s_ != rocsparse_status_success) { fprintf(stderr, "rocsparse failed: %s = %d\n", #x, (int)s_); return 1; } } while (0)

int main(int argc, char** argv) {
  const int n0 = argc > 1 ? atoi(argv[1]) : 8, nlev = argc > 2 ? atoi(argv[2]) : 4, reps = 50;
  fh_ctx_t ctx;
  CK(fh_init(0, &ctx));
  const double lo[3] = {0, 0, 0}, hi[3] = {1, 1, 1};
  fh_mesh_t m;
  CK(fh_mesh_box(n0, n0, n0, lo, hi, &m));
  for (int l = 1; l < nlev; l++) {
    fh_mesh_t f;
    CK(fh_mesh_refine(m, &f));
    fh_mesh_destroy(m);
    m = f;
  }
  int dim, nel, nnode, nloc, own[3], lev;
  fh_mesh_info(m, &dim, &nel, &nnode, &nloc, own, &lev);
  std::vector<int> ed((size_t)nel * nloc), rp(nnode + 1), col;
  std::vector<double> xy((size_t)nnode * dim);
  fh_mesh_get(m, ed.data(), xy.data(), nullptr);
  CK(fh_pattern_from_elements(nel, nloc, ed.data(), nnode, rp.data(), nullptr));
  col.resize(rp[nnode]);
  CK(fh_pattern_from_elements(nel, nloc, ed.data(), nnode, rp.data(), col.data()));
  fh_mat_t A;
  CK(fh_mat_create_csr(ctx, nnode, nnode, rp.data(), col.data(), nullptr, &A));
  fh_assembler_t as;
  CK(fh_assembler_create(ctx, 0, 2, 3, nel, nloc, ed.data(), nnode, xy.data(), A, &as));
  fh_vec_t x, y, res;
  CK(fh_vec_create(ctx, nnode, nnode, 0, nullptr, 0, &x));
  CK(fh_vec_create(ctx, nnode, nnode, 0, nullptr, 0, &y));
  CK(fh_vec_create(ctx, nnode, nnode, 0, nullptr, 0, &res));
  const double f[2] = {1.0, 0.0};
  CK(fh_assemble_poisson(as, nullptr, 0, f, A, res));
  std::vector<double> hx(nnode);
  unsigned long long st = 12345;
  for (int i = 0; i < nnode; i++) {
    st = st * 6364136223846793005ULL + 1442695040888963407ULL;
    hx[i] = (double)(st >> 11) / 9007199254740992.0 * 2.0 - 1.0;
  }
  CK(fh_vec_upload(x, hx.data()));
  const double bytes = (double)fh_spmv_algorithmic_bytes(A);
  hipStream_t stream = (hipStream_t)fh_stream(ctx);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms;
  // this library
  for (int i = 0; i < 5; i++) CK(fh_spmv(A, x, y, 0, nullptr, nullptr, 0.0));
  hipEventRecord(e0, stream);
  for (int i = 0; i < reps; i++) CK(fh_spmv(A, x, y, 0, nullptr, nullptr, 0.0));
  hipEventRecord(e1, stream);
  hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<double> y_fh(nnode);
  CK(fh_vec_download(y, y_fh.data()));
  printf("{\"rows\": %d, \"nnz\": %d, \"algorithmic_bytes\": %.0f,\n \"femus_hip_spmv\": {\"ms\": %.4f, \"GBps\": %.1f},\n", nnode, rp[nnode], bytes,
         ms / reps, bytes / (ms / reps) / 1e6);
  // rocSPARSE
  const int *d_rp, *d_col;
  const double* d_val;
  CK(fh_mat_dev_ptrs(A, &d_rp, &d_col, &d_val));
  rocsparse_handle h;
  RS(rocsparse_create_handle(&h));
  RS(rocsparse_set_stream(h, stream));
  rocsparse_mat_descr descr;
  RS(rocsparse_create_mat_descr(&descr));
  const double alpha = 1.0, beta = 0.0;
  double maxdiff[2] = {0, 0};
  for (int variant = 0; variant < 2; variant++) {
    rocsparse_mat_info info = nullptr;
    if (variant == 1) {
      RS(rocsparse_create_mat_info(&info));
      RS(rocsparse_dcsrmv_analysis(h, rocsparse_operation_none, nnode, nnode, rp[nnode], descr, d_val, d_rp, d_col, info));
    }
    for (int i = 0; i < 5; i++)
      RS(rocsparse_dcsrmv(h, rocsparse_operation_none, nnode, nnode, rp[nnode], &alpha, descr, d_val, d_rp, d_col, info, fh_vec_dev_ptr(x), &beta,
                          fh_vec_dev_ptr(y)));
    hipEventRecord(e0, stream);
    for (int i = 0; i < reps; i++)
      RS(rocsparse_dcsrmv(h, rocsparse_operation_none, nnode, nnode, rp[nnode], &alpha, descr, d_val, d_rp, d_col, info, fh_vec_dev_ptr(x), &beta,
                          fh_vec_dev_ptr(y)));
    hipEventRecord(e1, stream);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<double> y_rs(nnode);
    CK(fh_vec_download(y, y_rs.data()));
    double ymax = 0;
    for (int i = 0; i < nnode; i++) {
      maxdiff[variant] = std::max(maxdiff[variant], std::abs(y_rs[i] - y_fh[i]));
      ymax = std::max(ymax, std::abs(y_fh[i]));
    }
    printf(" \"rocsparse_dcsrmv_%s\": {\"ms\": %.4f, \"GBps\": %.1f, \"max_rel_diff_vs_femus_hip\": %.2e}%s\n", variant ? "adaptive" : "default",
           ms / reps, bytes / (ms / reps) / 1e6, maxdiff[variant] / ymax, variant ? "}" : ",");
    if (info) rocsparse_destroy_mat_info(info);
  }
  return 0;
}
