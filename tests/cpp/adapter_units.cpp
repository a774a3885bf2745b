// Member-by-member checks of the adapter classes against hand-computed values (the reference has no unit tests of
// SparseMatrix / NumericVector; these read like the calls FEMuS makes: LinearEquation.cpp, Solution.cpp, Mesh.cpp).
#include <cmath>
#include <cstdio>
#include <iostream>
#include <vector>
#include "../../femus_amd/csrc/adapters/HipBackend.hpp"

using namespace femus;
static int fails = 0;
#define CHECK(cond)                                                          \
  do {                                                                       \
    if (!(cond)) {                                                           \
      std::cout << "FAIL line " << __LINE__ << ": " #cond << std::endl;      \
      fails++;                                                               \
    }                                                                        \
  } while (0)

int main() {
  // ---- NumericVector ----------------------------------------------------------------------------------------------
  NumericVector* v = NumericVector::build().release();
  v->init(6, 6, false, SERIAL);
  CHECK(v->size() == 6 && v->local_size() == 6 && v->first_local_index() == 0 && v->last_local_index() == 6);
  *v = std::vector<double>{1, -2, 3, -4, 5, -6};
  CHECK(v->l1_norm() == 21. && v->linfty_norm() == 6. && std::fabs(v->l2_norm() - std::sqrt(91.)) < 1e-14);
  CHECK(v->min() == -6. && v->max() == 5. && v->sum() == -3.);
  CHECK((*v)(2) == 3.);
  v->set(0, 10.);
  v->add(0, 0.5);
  CHECK((*v)(0) == 10.5);
  std::vector<double> vals;
  v->get({1, 3}, vals);
  CHECK(vals[0] == -2. && vals[1] == -4.);
  NumericVector* w = NumericVector::build().release();
  w->init(*v);
  *w = 2.0;
  w->add(0.5, *v);   // w = 2 + 0.5 v
  CHECK((*w)(1) == 1. && (*w)(4) == 4.5);
  *w += *v;
  *w -= *v;
  CHECK((*w)(4) == 4.5);
  w->scale(2.);
  w->add(1.);
  CHECK((*w)(4) == 10.);
  w->abs();
  CHECK(w->min() >= 0.);
  std::unique_ptr<NumericVector> c = v->clone();
  CHECK((*c)(5) == -6. && c->dot(*v) == v->dot(*v));
  w->pointwise_mult(*v, *v);
  CHECK((*w)(3) == 16.);
  w->zero();
  w->add_vector_blocked(std::vector<double>{1., 2., 3.}, std::vector<int>{1, 1, 4});
  CHECK((*w)(1) == 3. && (*w)(4) == 3.);
  w->insert_vector_blocked(std::vector<double>{7.}, std::vector<int>{4});
  w->close();
  CHECK((*w)(4) == 7. && w->closed());
  std::vector<double> loc;
  w->localize(loc);
  CHECK(loc.size() == 6 && loc[1] == 3.);

  // ---- SparseMatrix: reference-style init + per-entry insertion, pattern frozen at close() ---------------------
  SparseMatrix* A = SparseMatrix::build().release();
  std::vector<int> nnz(6, 3), onz(6, 0);
  A->init(6, 6, 6, 6, nnz, onz);
  for (int i = 0; i < 6; i++) {   // 1-D Laplacian stencil by 2x2 element blocks
    if (i + 1 < 6) A->add_matrix_blocked(std::vector<double>{1., -1., -1., 1.}, std::vector<int>{i, i + 1}, std::vector<int>{i, i + 1});
  }
  A->close();
  CHECK(A->closed() && A->m() == 6 && A->n() == 6 && A->row_start() == 0 && A->row_stop() == 6);
  CHECK((*A)(0, 0) == 1. && (*A)(2, 2) == 2. && (*A)(2, 3) == -1. && (*A)(0, 5) == 0.);
  int cols[8];
  double rv[8];
  CHECK(A->MatGetRowM(2, cols, rv) == 3 && cols[0] == 1 && rv[1] == 2.);
  CHECK(A->linfty_norm() == 4. && A->l1_norm() == 4.);
  NumericVector* y = NumericVector::build().release();
  y->init(*v);
  y->matrix_mult(*v, *A);   // v = (10.5,-2,3,-4,5,-6)
  CHECK((*y)(0) == 12.5 && (*y)(5) == -11.);
  A->vector_mult_add(*y, *v);
  CHECK((*y)(0) == 25.);
  y->resid(*v, *v, *A);     // v - A v
  CHECK((*y)(0) == 10.5 - 12.5);
  y->matrix_mult_transpose(*v, *A);
  CHECK((*y)(0) == 12.5);   // symmetric
  A->add(1, 2, 0.25);
  A->set(5, 5, 9.);
  CHECK((*A)(1, 2) == -0.75 && (*A)(5, 5) == 9.);
  A->insert_row(3, 2, std::vector<int>{2, 4}, std::vector<double>{5., 6.}.data());
  CHECK((*A)(3, 2) == 5. && (*A)(3, 4) == 6. && (*A)(3, 3) == 2.);
  A->mat_zero_rows(std::vector<int>{0, 5}, 1.);
  CHECK((*A)(0, 0) == 1. && (*A)(0, 1) == 0. && (*A)(5, 4) == 0. && (*A)(5, 5) == 1.);
  std::vector<double> dv;
  A->matrix_get_diagonal_values(std::vector<int>{0, 3}, dv);
  CHECK(dv[0] == 1. && dv[1] == 2.);
  A->get_diagonal(*y);
  CHECK((*y)(3) == 2.);
  SparseMatrix* At = SparseMatrix::build().release();
  A->get_transpose(*At);
  CHECK((*At)(2, 3) == 5. && (*At)(3, 2) == -1.);
  A->get_transpose(*A);   // in place, as LinearImplicitSystem.cpp:1024 does
  CHECK((*A)(2, 3) == 5.);
  A->zero();
  CHECK(A->linfty_norm() == 0. && A->MatGetRowM(2) == 3);   // pattern kept

  // ---- products: matrix_PtAP, matrix_ABC, Left/Right -------------------------------------------------------------
  HipMatrix *P = new HipMatrix(), *B = new HipMatrix();
  P->init_pattern(4, 2, std::vector<int>{0, 1, 2, 3, 4}, std::vector<int>{0, 0, 1, 1});
  for (int i = 0; i < 4; i++) P->set(i, i / 2, 1.0);   // aggregation prolongator
  B->init_pattern(4, 4, std::vector<int>{0, 2, 5, 8, 10}, std::vector<int>{0, 1, 0, 1, 2, 1, 2, 3, 2, 3});
  const double bvals[10] = {2, -1, -1, 2, -1, -1, 2, -1, -1, 2};
  {
    int k = 0;
    const int bc[10] = {0, 1, 0, 1, 2, 1, 2, 3, 2, 3};
    const int br[10] = {0, 0, 1, 1, 1, 2, 2, 2, 3, 3};
    for (; k < 10; k++) B->set(br[k], bc[k], bvals[k]);
  }
  SparseMatrix* C = SparseMatrix::build().release();
  C->matrix_PtAP(*P, *B, false);
  CHECK(C->m() == 2 && (*C)(0, 0) == 2. && (*C)(0, 1) == -1. && (*C)(1, 1) == 2.);
  C->matrix_PtAP(*P, *B, true);
  CHECK((*C)(1, 0) == -1.);
  SparseMatrix *Pt = SparseMatrix::build().release(), *D = SparseMatrix::build().release();
  P->get_transpose(*Pt);
  D->matrix_ABC(*Pt, *B, *P, false);
  CHECK((*D)(0, 0) == 2. && (*D)(0, 1) == -1.);
  B->matrix_RightMatMult(*P);   // B <- B P  (4x2)
  CHECK(B->n() == 2 && (*B)(1, 0) == 1. && (*B)(1, 1) == -1.);
  B->matrix_LeftMatMult(*Pt);   // B <- P^T B (2x2)
  CHECK(B->m() == 2 && (*B)(0, 0) == 2.);

  delete v; delete w; delete y; delete A; delete At; delete P; delete B; delete C; delete Pt; delete D;
  std::cout << (fails ? "ADAPTER UNITS FAILED" : "ADAPTER UNITS OK") << std::endl;
  return fails ? 1 : 0;
}
