// The per-element crossing of the reference interface at size: an UNCHANGED FEMuS assembly callback calls
// KK->add_matrix_blocked / RES->add_vector_blocked once per element (applications/001_Poisson/main.cpp:283-609).  This program
// drives exactly that loop through the abstract classes -- first on a matrix that only knows its size (init with nnz bounds, the
// pattern grows as with MatSetValues), then again after zero() on the frozen device pattern (the pinned staging ring) -- and
// compares both with adding the same elements one after the other through the immediate C-ABI call.  The results must have the
// same bits.  Wall-clock times of the two loops (including close()) are printed.
//   usage: element_loop_adapters nx ny nz
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <vector>
#include "HipBackend.hpp"

using namespace femus;
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  const int nx = atoi(argv[1]), ny = atoi(argv[2]), nz = atoi(argv[3]);
  const int fe = 2, geom = nz ? 0 : 1;
  const double lo[3] = {0, 0, 0}, hi[3] = {1, 1, 1};
  fh_mesh_t msh;
  hip_check(fh_mesh_box(nx, ny, nz, lo, hi, &msh), "mesh");
  int dim, nel, nnode, nloc, own[3], lev;
  fh_mesh_info(msh, &dim, &nel, &nnode, &nloc, own, &lev);
  std::vector<int> elem_dof((size_t)nel * nloc), ff((size_t)nel * 2 * dim);
  std::vector<double> coords((size_t)nnode * dim);
  fh_mesh_get(msh, elem_dof.data(), coords.data(), ff.data());
  const int nc = nloc;
  std::vector<double> Kall((size_t)nel * nc * nc), Fall((size_t)nel * nc);
  std::vector<int> rp(nnode + 1), col;
  hip_check(fh_pattern_from_elements(nel, nloc, elem_dof.data(), nnode, rp.data(), nullptr), "pattern");
  col.resize(rp[nnode]);
  hip_check(fh_pattern_from_elements(nel, nloc, elem_dof.data(), nnode, rp.data(), col.data()), "pattern");
  {
    // element integrals (the reference computes them on the host inside the callback; here the device's "element matrices" mode)
    fh_mat_t tmp;
    hip_check(fh_mat_create_csr(hip_context(), nnode, nnode, rp.data(), col.data(), nullptr, &tmp), "tmp");
    fh_assembler_t as;
    hip_check(fh_assembler_create(hip_context(), geom, fe, 3, nel, nloc, elem_dof.data(), nnode, coords.data(), tmp, &as), "assembler");
    const double params[4] = {1.0, 0, 0, 0};
    hip_check(fh_element_matrices_poisson(as, nullptr, 0, params, Kall.data(), Fall.data()), "element matrices");
    fh_assembler_destroy(as);
    fh_mat_destroy(tmp);
  }
  SparseMatrix* KK = SparseMatrix::build().release();
  NumericVector* RES = NumericVector::build().release();
  std::vector<int> d_nnz(nnode, dim == 3 ? 125 : 25), o_nnz(nnode, 0);     // GetSparsityPatternSize upper bounds
  KK->init(nnode, nnode, nnode, nnode, d_nnz, o_nnz);
  RES->init(nnode, nnode, false, SERIAL);

  std::vector<double> Jac(nc * nc), Res(nc);
  std::vector<int> l2GMap(nc);
  auto element_loop = [&]() {       // the callback's tail, as in 001_Poisson
    KK->zero();
    RES->zero();
    for (int iel = 0; iel < nel; iel++) {
      for (int i = 0; i < nc; i++) l2GMap[i] = elem_dof[(size_t)iel * nloc + i];
      Jac.assign(Kall.begin() + (size_t)iel * nc * nc, Kall.begin() + (size_t)(iel + 1) * nc * nc);
      Res.assign(Fall.begin() + (size_t)iel * nc, Fall.begin() + (size_t)(iel + 1) * nc);
      RES->add_vector_blocked(Res, l2GMap);
      KK->add_matrix_blocked(Jac, l2GMap, l2GMap);
    }
    RES->close();
    KK->close();
  };
  auto values = [&](std::vector<double>& a, std::vector<double>& r) {
    HipMatrix* h = static_cast<HipMatrix*>(KK);
    int m, n, nnz;
    fh_mat_size(h->handle(), &m, &n, &nnz);
    a.resize(nnz);
    hip_check(fh_mat_get_values_csr(h->handle(), a.data()), "values");
    RES->localize(r);
    if (nnz != rp[nnode]) { std::cout << "pattern of the grown matrix differs: " << nnz << " vs " << rp[nnode] << std::endl; exit(3); }
    std::vector<int> rp2(m + 1), col2(nnz);
    fh_mat_get_pattern(h->handle(), rp2.data(), col2.data());
    if (rp2 != rp || col2 != col) { std::cout << "pattern of the grown matrix differs" << std::endl; exit(3); }
  };
  double t0 = now();
  element_loop();
  const double t_first = now() - t0;
  std::vector<double> a1, r1, a2, r2;
  values(a1, r1);
  t0 = now();
  element_loop();
  const double t_second = now() - t0;
  values(a2, r2);
  int64_t blocks = 0, rings = 0;
  fh_mat_stage_stats(static_cast<HipMatrix*>(KK)->handle(), &blocks, &rings);

  // the elements one after the other through the immediate call
  fh_mat_t B;
  fh_vec_t rb;
  hip_check(fh_mat_create_csr(hip_context(), nnode, nnode, rp.data(), col.data(), nullptr, &B), "B");
  hip_check(fh_vec_create(hip_context(), nnode, nnode, 0, nullptr, 0, &rb), "rb");
  t0 = now();
  for (int iel = 0; iel < nel; iel++) {
    const int* d = elem_dof.data() + (size_t)iel * nloc;
    hip_check(fh_vec_add_values(rb, nc, d, Fall.data() + (size_t)iel * nc), "fh_vec_add_values");
    hip_check(fh_mat_add_block(B, nc, d, nc, d, Kall.data() + (size_t)iel * nc * nc), "fh_mat_add_block");
  }
  const double t_immediate = now() - t0;
  std::vector<double> a3(rp[nnode]), r3(nnode);
  hip_check(fh_mat_get_values_csr(B, a3.data()), "values");
  hip_check(fh_vec_download(rb, r3.data()), "values");
  const bool same = !memcmp(a1.data(), a3.data(), a3.size() * 8) && !memcmp(a2.data(), a3.data(), a3.size() * 8) &&
                    !memcmp(r1.data(), r3.data(), r3.size() * 8) && !memcmp(r2.data(), r3.data(), r3.size() * 8);
  // and the batched device assembly (other summation inside the element kernel: close, not identical)
  double asum = 0, amax = 0;
  for (double v : a3) { asum += v; amax = std::max(amax, std::fabs(v)); }
  printf("elements %d  dofs %d  nnz %d\n", nel, nnode, rp[nnode]);
  printf("first_loop_s %.4f  (pattern grown on the host, per element %.2f us)\n", t_first, 1e6 * t_first / nel);
  printf("second_loop_s %.4f  (staging ring, per element %.2f us; %lld blocks in %lld rings)\n", t_second, 1e6 * t_second / nel,
         (long long)blocks, (long long)rings);
  printf("immediate_loop_s %.4f  (per element %.2f us)\n", t_immediate, 1e6 * t_immediate / nel);
  printf("value_sum %.17g  max %.17g\n", asum, amax);
  printf("bit_identical %d\n", same ? 1 : 0);
  // an entry outside the pattern is reported by close(): checked by the unit test through the C ABI (the adapter aborts)
  fh_mat_destroy(B);
  fh_vec_destroy(rb);
  delete KK;
  delete RES;
  fh_mesh_destroy(msh);
  return same ? 0 : 1;
}
