// Adapter test written the way a FEMuS application drives the algebra layer (applications/001_Poisson/main.cpp:283-609 for
// the assembly callback, LinearImplicitSystem::MGsolve / Vcycle, LinearImplicitSystem.cpp:288-411, 468-497 for the solver):
// everything goes through the abstract SparseMatrix / NumericVector / LinearEquationSolver interface and the factories.
// Mesh, DOF maps and Dirichlet flags (FEMuS-owned in a real build) come from the C-ABI mesh helpers.
//   usage: poisson_adapters nx ny nz nlevels out.bin
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>
#include "HipBackend.hpp"

using namespace femus;

int main(int argc, char** argv) {
  if (argc < 6) return 2;
  const int nx = atoi(argv[1]), ny = atoi(argv[2]), nz = atoi(argv[3]), nlev = atoi(argv[4]);
  const int fe = 2, geom = nz ? 0 : 1;
  const double lo[3] = {0, 0, 0}, hi[3] = {1, 1, 1};
  std::vector<fh_mesh_t> msh(nlev);
  hip_check(fh_mesh_box(nx, ny, nz, lo, hi, &msh[0]), "mesh");
  for (int l = 1; l < nlev; l++) hip_check(fh_mesh_refine(msh[l - 1], &msh[l]), "refine");

  // ---- what FEMuS owns in a real build, per level: the mesh's dof offsets, the Solution with its boundary flag vector _Bdc
  // (2 = free, 0 = Dirichlet: MultiLevelSolution::GenerateBdc, MultiLevelSolution.cpp:725-840) --------------------------------
  std::vector<Mesh*> fmesh(nlev);
  std::vector<Solution*> fsol(nlev);
  std::vector<int> ndof(nlev);
  for (int l = 0; l < nlev; l++) {
    int dim, nel, nnode, nloc, own[3], lev;
    fh_mesh_info(msh[l], &dim, &nel, &nnode, &nloc, own, &lev);
    ndof[l] = nnode;
    fmesh[l] = new Mesh();
    for (int t = 0; t < 5; t++) fmesh[l]->_dofOffset[t] = {0u, (unsigned)(t == 0 ? own[0] : nnode)};     // one rank
    fsol[l] = new Solution(fmesh[l]);
    NumericVector* flag = NumericVector::build().release();
    flag->init(nnode, nnode, false, SERIAL);
    *flag = 2.;
    int nb = nnode;
    std::vector<int> bdc(nnode);
    hip_check(fh_mesh_dirichlet_dofs(msh[l], fe, &nb, bdc.data()), "bdc");
    bdc.resize(nb);
    flag->insert(std::vector<double>(nb, 0.), bdc);
    flag->close();
    fsol[l]->_Bdc.push_back(flag);
  }

  // ---- LinearImplicitSystem::init: one LinearEquationSolver per level through the factory, InitPde, prolongators --------------
  std::vector<LinearEquationSolver*> LinSolver(nlev);
  std::vector<SparseMatrix*> PP(nlev, nullptr);
  std::vector<unsigned> SolPdeIndex(1, 0u), SolType(1, (unsigned)fe);
  char name_u[] = "u";
  std::vector<char*> SolName(1, name_u);
  std::vector<bool> sparsity;
  for (int l = 0; l < nlev; l++) {
    const int nnode = ndof[l];
    LinSolver[l] = LinearEquationSolver::build(l, fsol[l], FEMuS_DEFAULT).release();
    LinearEquationSolver* ls = LinSolver[l];
    ls->InitPde(SolPdeIndex, SolType, SolName, &fsol[l]->_Bdc, nlev, sparsity);     // creates _KK, _RES, _RESC, _EPS, _EPSC
    if (l == nlev - 1) {
      std::vector<int> d_nnz(nnode, 125), o_nnz(nnode, 0);     // GetSparsityPatternSize upper bounds
      ls->_KK->init(nnode, nnode, nnode, nnode, d_nnz, o_nnz);
    }
    ls->set_solver_type(RICHARDSON);
    // argv[6]: "sor" = SOR_PRECOND (the choice of 001_Poisson/main.cpp:242), "ilu" = ILU_PRECOND, "mlu" = MLU_PRECOND (MUMPS through PCLU, the
    // SetPreconditionerFineGrids choice of 18 applications), default JACOBI_PRECOND
    const std::string pc = argc > 6 ? argv[6] : "jacobi";
    ls->set_preconditioner_type(pc == "sor" ? SOR_PRECOND : pc == "ilu" ? ILU_PRECOND : pc == "mlu" ? MLU_PRECOND : JACOBI_PRECOND);
    ls->SetRichardsonScaleFactor(pc == "jacobi" ? 2. / 3. : pc == "mlu" ? 1.0 : 0.8);
    if (l > 0) {
      fh_mat_t P;
      hip_check(fh_build_prolongator(hip_context(), msh[l - 1], msh[l], fe, 1, &P), "BuildProlongatorMatrix");
      HipMatrix* hp = new HipMatrix();
      hp->adopt(P);
      PP[l] = hp;
    }
  }
  // ---- the assembly callback on the finest level: per-element add_*_blocked through the virtual interface ------
  const int top = nlev - 1;
  int dim, nel, nnode, nloc, own[3], lev;
  fh_mesh_info(msh[top], &dim, &nel, &nnode, &nloc, own, &lev);
  std::vector<int> elem_dof((size_t)nel * nloc), ff((size_t)nel * 2 * dim);
  std::vector<double> coords((size_t)nnode * dim);
  fh_mesh_get(msh[top], elem_dof.data(), coords.data(), ff.data());
  const int nc = nloc;
  std::vector<double> Kall((size_t)nel * nc * nc), Fall((size_t)nel * nc);
  {
    // element integrals: the batched device kernel in "element matrices" mode (the reference computes them on the host)
    std::vector<int> rp(nnode + 1), col;
    hip_check(fh_pattern_from_elements(nel, nloc, elem_dof.data(), nnode, rp.data(), nullptr), "pattern");
    col.resize(rp[nnode]);
    hip_check(fh_pattern_from_elements(nel, nloc, elem_dof.data(), nnode, rp.data(), col.data()), "pattern");
    fh_mat_t tmp;
    hip_check(fh_mat_create_csr(hip_context(), nnode, nnode, rp.data(), col.data(), nullptr, &tmp), "tmp");
    fh_assembler_t as;
    hip_check(fh_assembler_create(hip_context(), geom, fe, 3, nel, nloc, elem_dof.data(), nnode, coords.data(), tmp, &as), "assembler");
    const double params[4] = {1.0, 0, 0, 0};
    hip_check(fh_element_matrices_poisson(as, nullptr, 0, params, Kall.data(), Fall.data()), "element matrices");
    fh_assembler_destroy(as);
    fh_mat_destroy(tmp);
  }
  SparseMatrix* KK = LinSolver[top]->_KK;
  NumericVector* RES = LinSolver[top]->_RES;
  KK->zero();
  RES->zero();
  std::vector<double> Jac(nc * nc), Res(nc);
  std::vector<int> l2GMap(nc);
  for (int iel = 0; iel < nel; iel++) {
    for (int i = 0; i < nc; i++) l2GMap[i] = elem_dof[(size_t)iel * nloc + i];
    Jac.assign(Kall.begin() + (size_t)iel * nc * nc, Kall.begin() + (size_t)(iel + 1) * nc * nc);
    Res.assign(Fall.begin() + (size_t)iel * nc, Fall.begin() + (size_t)(iel + 1) * nc);
    RES->add_vector_blocked(Res, l2GMap);
    KK->add_matrix_blocked(Jac, l2GMap, l2GMap);
  }
  RES->close();
  KK->close();

  // ---- MGsolve: Galerkin chain, MGInit, MGSetLevel, Vcycle --------------------------------------------------------
  for (int i = top; i > 0; i--) LinSolver[i - 1]->_KK->matrix_PtAP(*PP[i], *LinSolver[i]->_KK, false);
  LinSolver[top]->MGInit(MULTIPLICATIVE, nlev, GMRES);
  LinSolver[top]->SetTolerances(1e-12, 1e-50, 1e50, 40, 30);
  std::vector<unsigned> vars(1, 0);
  {
    // where the unknowns of the coarsest level lie (inside FEMuS: Mesh::GetTopology()->_Sol[0..dim-1]): lets the exact coarse solve dissect its
    // dense problem when that is large enough (optional, backend-specific)
    int d0, e0, n0, l0, o0[3], v0;
    fh_mesh_info(msh[0], &d0, &e0, &n0, &l0, o0, &v0);
    std::vector<int> ed0((size_t)e0 * l0), ff0((size_t)e0 * 2 * d0);
    std::vector<double> xy0((size_t)n0 * d0);
    fh_mesh_get(msh[0], ed0.data(), xy0.data(), ff0.data());
    static_cast<LinearEquationSolverHip*>(LinSolver[0])->SetLevelCoordinates(d0, xy0);
  }
  for (int i = 0; i < nlev; i++) LinSolver[i]->MGSetLevel(LinSolver[top], top, vars, PP[i], PP[i], i ? 2 : 1, i ? 2 : 0);
  LinSolver[top]->SetEpsZero();
  double res = 0;
  int it = 0;
  for (; it < 6; it++) {
    LinSolver[top]->MGSolve(it == 0);
    res = LinSolver[top]->_RES->l2_norm();
    std::cout << "       *************** Linear iteration " << it + 1 << "  Linear Res  L2norm u = " << res << std::endl;
    if (res < 1e-11) break;
  }
  std::vector<double> sol;
  LinSolver[top]->_EPS->localize(sol);
  std::cout << "||u||_2 = " << LinSolver[top]->_EPS->l2_norm() << "  max = " << LinSolver[top]->_EPS->max() << "  KK(0,0) = " << (*KK)(0, 0)
            << std::endl;
  FILE* f = fopen(argv[5], "wb");
  fwrite(sol.data(), sizeof(double), sol.size(), f);
  fclose(f);
  // BuildBdcIndex gave the rows of the mesh helper's Dirichlet list
  {
    int nb = ndof[top];
    std::vector<int> bdc(nb);
    hip_check(fh_mesh_dirichlet_dofs(msh[top], fe, &nb, bdc.data()), "bdc");
    bdc.resize(nb);
    if (static_cast<LinearEquationSolverHip*>(LinSolver[top])->bdc_index() != bdc) {
      std::cout << "BuildBdcIndex: wrong Dirichlet rows" << std::endl;
      return 3;
    }
  }
  LinSolver[top]->MGClear();
  for (int l = 0; l < nlev; l++) {
    LinSolver[l]->DeletePde();
    delete LinSolver[l];
    delete PP[l];
    delete fsol[l]->_Bdc[0];
    delete fsol[l];
    delete fmesh[l];
    fh_mesh_destroy(msh[l]);
  }
  return 0;
}
