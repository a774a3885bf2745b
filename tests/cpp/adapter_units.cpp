// Member-by-member checks of the adapter classes against hand-computed values (the reference has no unit tests of
// SparseMatrix / NumericVector; these read like the calls FEMuS makes: LinearEquation.cpp, Solution.cpp, Mesh.cpp).
#include <cmath>
#include <cstdio>
#include <iostream>
#include <vector>
#include "HipBackend.hpp"

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

  // ---- the rest of the reference's pure virtuals (NumericVector.hpp:160-169, :275-279, :301-323; SparseMatrix.hpp:81, :113, :174-207)
  {
    NumericVector *a = NumericVector::build().release(), *b = NumericVector::build().release();
    a->init(5, 5, false, SERIAL);
    b->init(5, 5, false, SERIAL);
    *a = std::vector<double>{1, 2, 3, 4, 5};
    b->zero();
    b->insert(std::vector<double>{7., 8.}, std::vector<int>{4, 0});
    CHECK((*b)(4) == 7. && (*b)(0) == 8. && (*b)(2) == 0.);
    b->add_vector(std::vector<double>{1., 1.}, std::vector<int>{4, 1});
    CHECK((*b)(4) == 8. && (*b)(1) == 1.);
    b->add_vector(*a, std::vector<int>{0, 1, 2, 3, 4});
    CHECK((*b)(0) == 9. && (*b)(3) == 4.);
    b->insert(*a, std::vector<int>{4, 3, 2, 1, 0});
    CHECK((*b)(4) == 1. && (*b)(0) == 5.);
    a->swap(*b);
    CHECK((*a)(0) == 5. && (*b)(0) == 1.);
    b->localize(*a);
    CHECK((*a)(4) == 5.);
    std::vector<double> all;
    a->localize_to_all(all);
    CHECK(all.size() == 5 && all[2] == 3.);
    a->localize_to_one(all, 0);
    CHECK(all[1] == 2.);
    *b = 0.;
    a->localize(*b, std::vector<int>{1, 3});
    CHECK((*b)(1) == 2. && (*b)(3) == 4. && (*b)(0) == 0.);
    *a *= 2.;
    *a /= 4.;
    CHECK((*a)(3) == 2.);
    a->close();
    CHECK(a->closed() && a->type() == SERIAL);
    a->BinaryPrint("/tmp/femus_hip_adapter_units_vec.bin");          // SaveSolution / LoadSolution file of one variable
    *b = 0.;
    b->BinaryLoad("/tmp/femus_hip_adapter_units_vec.bin");
    CHECK((*b)(3) == (*a)(3) && (*b)(0) == (*a)(0));
    delete a;
    delete b;
  }
  {
    HipMatrix X, Y, Z;
    X.init_pattern(2, 2, std::vector<int>{0, 1, 3}, std::vector<int>{0, 0, 1});
    X.set(0, 0, 1.); X.set(1, 0, 2.); X.set(1, 1, 3.);
    Y.init_pattern(2, 2, std::vector<int>{0, 2, 3}, std::vector<int>{0, 1, 1});
    Y.set(0, 0, 10.); Y.set(0, 1, 20.); Y.set(1, 1, 30.);
    X.matrix_add(0.5, Y, "different_nonzero_pattern");          // union pattern
    CHECK(X(0, 0) == 6. && X(0, 1) == 10. && X(1, 0) == 2. && X(1, 1) == 18. && X.MatGetRowM(0) == 2);
    X.add(1.0, Y);
    CHECK(X(0, 1) == 30.);
    X.matrix_set_diagonal_values(std::vector<int>{0, 1}, 0.);
    CHECK(X(0, 0) == 0. && X(1, 1) == 0. && X(1, 0) == 2.);
    double tol = 1e-15;
    X.RemoveZeroEntries(tol);
    CHECK(X.MatGetRowM(0) == 1 && X.MatGetRowM(1) == 1);
    X.matrix_set_off_diagonal_values_blocked(std::vector<int>{0}, std::vector<int>{1}, 4.5);
    CHECK(X(0, 1) == 4.5);
    std::vector<double> dv;
    Y.matrix_get_diagonal_values(std::vector<int>{0, 1}, dv);
    CHECK(dv[0] == 10. && dv[1] == 30.);
    // 2 x 2 block matrix [[Y, 0], [0, Y]] (SparseMatrix::init(nr, nc, blocks))
    std::vector<SparseMatrix*> blocks = {&Y, nullptr, nullptr, &Y};
    Z.init(2, 2, blocks);
    CHECK(Z.m() == 4 && Z.n() == 4 && Z(2, 3) == 20. && Z(0, 1) == 20. && Z(1, 2) == 0. && Z.row_start() == 0 && Z.row_stop() == 4);
    // matrix_ABC with reuse: numeric only on the kept plan
    HipMatrix I2, R;
    I2.init_pattern(2, 2, std::vector<int>{0, 1, 2}, std::vector<int>{0, 1});
    I2.set(0, 0, 1.); I2.set(1, 1, 2.);
    R.matrix_ABC(I2, Y, I2, false);
    CHECK(R(0, 1) == 40. && R(1, 1) == 120.);
    Y.set(0, 1, 1.);
    R.matrix_ABC(I2, Y, I2, true);
    CHECK(R(0, 1) == 2.);
  }
  // ---- several ranks, seen from one: planner + a GHOSTED vector whose close() refreshes the ghosts through its exchange plan
  // (PetscVector.hpp:595-612) and whose dot / norms go through the plan's all-reduce.  The "other rank" is this one: a host-staged
  // transport that hands the sent entries back.
  {
    const int64_t gid[4] = {10, 11, 12, 13};
    const int owner[4] = {0, 0, 0, 0};
    const unsigned char need[4] = {1, 1, 1, 1};
    fh_dd_plan_t plan = nullptr;
    hip_check(fh_dd_plan_create(0, 1, 4, gid, owner, need, nullptr, nullptr, &plan), "fh_dd_plan_create");
    int no = 0, ng = 0, ns = 0;
    fh_dd_plan_sizes(plan, &no, &ng, &ns);
    CHECK(no == 4 && ng == 0 && ns == 0);
    fh_dd_plan_destroy(plan);
    struct Self {
      static int exchange(void*, const double* send, const int* sc, double* recv, const int* rc) {
        for (int k = 0; k < sc[0] && k < rc[0]; k++) recv[k] = send[k];
        return 0;
      }
      static int allreduce(void*, double*, int) { return 0; }
    };
    const int send_counts[1] = {2}, recv_counts[1] = {2}, send_idx[2] = {0, 2};
    fh_halo_t halo = nullptr;
    hip_check(fh_halo_create_host(hip_context(), 0, 1, Self::exchange, Self::allreduce, nullptr, send_counts, send_idx, recv_counts, &halo), "halo");
    HipVector g;
    g.init(8, 4, std::vector<int>{6, 7}, false, GHOSTED);     // owned global 0..3, ghosts = global 6 and 7 ("owned elsewhere")
    g.attach_halo(halo);
    g.set(0, 1.5); g.set(1, 2.5); g.set(2, 3.5); g.set(3, 4.5);
    CHECK(!g.closed());
    g.close();                                                   // ghost refresh
    CHECK(g.closed() && g(6) == 1.5 && g(7) == 3.5 && g.type() == GHOSTED);
    HipVector g2;
    g2.init(g);                                                  // layout incl. ghosts and plan
    g2 = static_cast<const NumericVector&>(g);
    CHECK(g2.halo() == halo && std::fabs(g2.dot(g) - (1.5 * 1.5 + 2.5 * 2.5 + 3.5 * 3.5 + 4.5 * 4.5)) < 1e-14);
    // a product with a ghosted operand refreshes the ghosts itself: y = A x, A = [0 .. | picks ghost 1]
    HipMatrix Ag;
    Ag.init_pattern(4, 6, std::vector<int>{0, 1, 2, 3, 4}, std::vector<int>{5, 0, 1, 4});
    Ag.set(0, 5, 1.); Ag.set(1, 0, 1.); Ag.set(2, 1, 1.); Ag.set(3, 4, 2.);
    g.set(2, 9.);                                                // owner value changes; the ghost copy (global 7) is stale until the product
    HipVector yg;
    yg.init(4, 4, false, PARALLEL);
    yg.matrix_mult(g, Ag);
    CHECK(yg(0) == 9. && yg(1) == 1.5 && yg(2) == 2.5 && yg(3) == 3.0);
    g.clear(); g2.clear();
    fh_halo_destroy(halo);
  }
  // ---- LinearEquationSolver::Solve: one-level solve of the level's system with the Dirichlet rows derived from _Bdc -----------
  {
    Mesh msh1;
    for (int t = 0; t < 5; t++) msh1._dofOffset[t] = {0u, 4u};
    Solution sol1(&msh1);
    NumericVector* flag = NumericVector::build().release();
    flag->init(4, 4, false, SERIAL);
    *flag = std::vector<double>{0., 2., 2., 0.};          // ends are Dirichlet
    sol1._Bdc.push_back(flag);
    LinearEquationSolver* ls = LinearEquationSolver::build(0, &sol1, FEMuS_DEFAULT).release();
    std::vector<unsigned> idx(1, 0u), typ(1, 2u), vars(1, 0u);
    char nm[] = "u";
    std::vector<char*> names(1, nm);
    std::vector<bool> sp;
    ls->InitPde(idx, typ, names, &sol1._Bdc, 1, sp);
    CHECK(ls->KKoffset[1][0] == 4u && ls->KKIndex[1] == 4);
    static_cast<HipMatrix*>(ls->_KK)->init_pattern(4, 4, std::vector<int>{0, 2, 5, 8, 10}, std::vector<int>{0, 1, 0, 1, 2, 1, 2, 3, 2, 3});
    const int br[10] = {0, 0, 1, 1, 1, 2, 2, 2, 3, 3}, bc[10] = {0, 1, 0, 1, 2, 1, 2, 3, 2, 3};
    const double bv[10] = {2, -1, -1, 2, -1, -1, 2, -1, -1, 2};
    for (int k = 0; k < 10; k++) ls->_KK->set(br[k], bc[k], bv[k]);
    *ls->_RES = std::vector<double>{5., 1., 1., 5.};
    ls->SetEpsZero();
    ls->SetTolerances(1e-12, 1e-50, 1e50, 10, 10);
    ls->Solve(vars, true);
    // rows 0 and 3 become identity rows with zero residual: 2 x1 - x2 = 1, -x1 + 2 x2 = 1 -> x1 = x2 = 1
    CHECK(std::fabs((*ls->_EPS)(1) - 1.) < 1e-13 && std::fabs((*ls->_EPS)(2) - 1.) < 1e-13 && (*ls->_EPS)(0) == 0. && (*ls->_EPS)(3) == 0.);
    CHECK(ls->_RES->linfty_norm() < 1e-13);
    CHECK(static_cast<LinearEquationSolverHip*>(ls)->bdc_index() == (std::vector<int>{0, 3}));
    ls->DeletePde();
    delete ls;
    delete flag;
  }

  delete v; delete w; delete y; delete A; delete At; delete P; delete B; delete C; delete Pt; delete D;
  std::cout << (fails ? "ADAPTER UNITS FAILED" : "ADAPTER UNITS OK") << std::endl;
  return fails ? 1 : 0;
}
