// Adapter test written the way applications/003_NavierStokes drives the library: NonLinearImplicitSystem::MGsolve with F_CYCLE
// (NonLinearImplicitSystem.cpp:157-361) over the abstract SparseMatrix / NumericVector / LinearEquationSolver interface, the
// FEMuS_ASM solver type for the level smoothers (SteadyNavierStokesParallel/main.cpp:166-167, 187-188), the cavity boundary
// conditions of main.cpp:365-390, and the batched Taylor-Hood residual/Jacobian call in place of the adept callback.
// Mesh, dof maps and element blocks (FEMuS-owned in a real build) come from the C-ABI mesh helpers.
// With the last argument `stab` = 1 the run is the application's own discretisation (main.cpp:96-108, :390-925): equal-order LAGRANGE FIRST velocity and
// pressure, the Franca-Frey stabilised callback (fh_assemble_navier_stokes_stab) and the Reynolds continuation of the callback's call counter (:485-489;
// `nu` is ignored).  The application erases its coarse levels (:92: ONE level, every linear solve the exact one): nlevels = 1 is its real configuration.
// With an eleventh argument -- a Gambit file -- the run is the reference's KNOWN-ANSWER TEST, unittests/testNSSteadyDD/main.cpp: that mesh (input/nsbenc.neu), Q2
// velocity with AddSolution("P", DISCONTINUOUS_POLYNOMIAL, FIRST) (:97), its boundary conditions (:290-392) and initial velocity (:281-287), solver type
// FEMuS_DEFAULT with SetSolverFineGrids(GMRES) + SetPreconditionerFineGrids(ILU_PRECOND) (:145-152), nonlinear F-cycle; `n` is ignored, `nlevels` = 4 gives the
// level whose norms the test stores (:202-244).
//   usage: navier_stokes_adapters n nlevels nu out.bin [nschur nblock lsolver outer_pre coloured stab [mesh.neu]]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <map>
#include <vector>
#include "HipBackend.hpp"

using namespace femus;

static bool SetBoundaryConditionCavityFlow(const double* x, const char name, double& value, const int FaceName) {
  bool test = 1;
  value = 0.;
  if (name == 'V') {
    if (4 == FaceName) {          // the x = -0.5 wall of the generated box (group 1 of box10x10.neu)
      if (x[1] < 0.5 && x[1] > -0.5) value = 1.;
    }
  }
  if (name == 'P') {
    test = 0;
    if (x[0] < -.5 + 1.e-08 && x[1] < -.5 + 1.e-08) test = 1;
  }
  return test;
}

// unittests/testNSSteadyDD/main.cpp:290-392 (faces: 1 inflow, 2 outflow, 3 walls, 4 cylinder) and :281-287
static double InflowProfile(const double y) { return 1.5 * 0.2 * (4.0 / 0.1681) * y * (0.41 - y); }
static bool SetBoundaryConditionCylinder(const double* x, const char name, double& value, const int FaceName) {
  value = 0.;
  if (name == 'P') return false;
  if (FaceName == 2) return false;
  if (name == 'U' && FaceName == 1) value = InflowProfile(x[1]);
  return true;
}

int main(int argc, char** argv) {
  if (argc < 5) return 2;
  const int n = atoi(argv[1]), nlev = atoi(argv[2]);
  const double nu = atof(argv[3]);
  const int nschur = argc > 5 ? atoi(argv[5]) : 0, nblock = argc > 6 ? atoi(argv[6]) : 4, lsolver = argc > 7 ? atoi(argv[7]) : 0, outer_pre = argc > 8 ? atoi(argv[8]) : 0,
            coloured = argc > 9 ? atoi(argv[9]) : 0;     // 1: the library's own block smoother (exact inverses in colour order) instead of PCASM as the reference sets it
  const int stab = argc > 10 ? atoi(argv[10]) : 0;       // 1: the application's equal-order stabilised callback with its Reynolds continuation
  const char* neu = argc > 11 ? argv[11] : nullptr;      // the known-answer test: Gambit mesh, discontinuous piecewise-linear pressure
  const bool pw = neu != nullptr;
  const int geom = 1, nvars = 3;
  const int fe[3] = {stab ? 0 : 2, stab ? 0 : 2, pw ? 4 : 0};
  const char names[3] = {'U', 'V', 'P'};
  const double lo[3] = {-0.5, -0.5, 0}, hi[3] = {0.5, 0.5, 1};
  std::vector<fh_mesh_t> msh(nlev);
  if (pw) hip_check(fh_mesh_read_gambit(neu, 1.0, &msh[0]), "ReadCoarseMesh");
  else hip_check(fh_mesh_box(n, n, 0, lo, hi, &msh[0]), "mesh");
  for (int l = 1; l < nlev; l++) hip_check(fh_mesh_refine(msh[l - 1], &msh[l]), "refine");

  std::vector<LinearEquationSolver*> LinSolver(nlev);
  std::vector<Mesh*> fmesh(nlev);           // FEMuS-owned in a real build: dof offsets of the families, Solution with the _Bdc flag vectors
  std::vector<Solution*> fsol(nlev);
  std::vector<unsigned> SolPdeIndex = {0u, 1u, 2u}, SolType = {stab ? 0u : 2u, stab ? 0u : 2u, pw ? 4u : 0u};
  char nU[] = "U", nV[] = "V", nP[] = "P";
  std::vector<char*> SolName = {nU, nV, nP};
  std::vector<bool> sparsity;
  std::vector<SparseMatrix*> PP(nlev, nullptr), PPsol(nlev, nullptr);
  std::vector<NumericVector*> Sol(nlev);
  std::vector<fh_ns_assembler_t> as(nlev);
  std::vector<std::vector<int>> offs(nlev, std::vector<int>(nvars + 1)), bdc(nlev);
  for (int l = 0; l < nlev; l++) {
    int dim, nel, nnode, nloc, own[3], lev, nd;
    fh_mesh_info(msh[l], &dim, &nel, &nnode, &nloc, own, &lev);
    hip_check(fh_system_elem_dofs(msh[l], nvars, fe, &nd, offs[l].data(), nullptr), "GetSystemDof");
    std::vector<int> es((size_t)nel * nd), ed((size_t)nel * nloc), ff((size_t)nel * 4);
    std::vector<double> xy((size_t)nnode * dim);
    hip_check(fh_system_elem_dofs(msh[l], nvars, fe, &nd, offs[l].data(), es.data()), "GetSystemDof");
    fh_mesh_get(msh[l], ed.data(), xy.data(), ff.data());
    const int ndof = offs[l][nvars];
    // FEMuS_ASM solver: element blocks around every pressure dof
    fmesh[l] = new Mesh();
    for (int t = 0; t < 5; t++) fmesh[l]->_dofOffset[t] = {0u, (unsigned)(t == 0 ? own[0] : t < 3 ? nnode : t == 3 ? nel : 3 * nel)};     // one rank
    fmesh[l]->_elementDofNumber[4] = 3;
    // the element tables BuildASMIndex reads through the Mesh interface (FEMuS's own Mesh has them)
    fmesh[l]->_elementOffset = {0u, (unsigned)nel};
    fmesh[l]->_elementMaterial.assign(nel, 2);                                            // fluid
    fmesh[l]->_nloc = nloc;
    fmesh[l]->_elementDof.assign(ed.begin(), ed.end());
    fmesh[l]->_elementDofNumber[0] = 4; fmesh[l]->_elementDofNumber[1] = 8; fmesh[l]->_elementDofNumber[2] = 9;
    {
      std::vector<std::vector<unsigned> > near_vertex(own[0]);                            // Elem.cpp:494-528: elements around the vertices of an element
      for (int iel = 0; iel < nel; iel++)
        for (int i = 0; i < 4; i++) near_vertex[ed[(size_t)iel * nloc + i]].push_back(iel);
      fmesh[l]->_el._elementNearElement.resize(nel);
      for (int iel = 0; iel < nel; iel++) {
        std::map<unsigned, bool> els;
        for (int i = 0; i < 4; i++)
          for (unsigned jel : near_vertex[ed[(size_t)iel * nloc + i]])
            if ((int)jel != iel) els[jel] = true;
        fmesh[l]->_el._elementNearElement[iel].push_back(iel);
        for (auto& kv : els) fmesh[l]->_el._elementNearElement[iel].push_back(kv.first);
      }
    }
    fsol[l] = new Solution(fmesh[l]);
    for (int k = 0; k < nvars; k++) {
      NumericVector* flag = NumericVector::build().release();
      const int nk = offs[l][k + 1] - offs[l][k];
      flag->init(nk, nk, false, SERIAL);
      *flag = 2.;                                            // free; Dirichlet nodes get 0 below (MultiLevelSolution::GenerateBdc)
      fsol[l]->_Bdc.push_back(flag);
    }
    LinearEquationSolverHip* ls = static_cast<LinearEquationSolverHip*>(LinearEquationSolver::build(l, fsol[l], pw ? FEMuS_DEFAULT : FEMuS_ASM).release());
    LinearEquationSolverHipAsm* lsa = pw ? nullptr : static_cast<LinearEquationSolverHipAsm*>(ls);
    LinSolver[l] = ls;
    ls->InitPde(SolPdeIndex, SolType, SolName, &fsol[l]->_Bdc, nlev, sparsity);     // _KK, _RES, _RESC, _EPS, _EPSC; KKoffset = offs[l]
    for (int k = 0; k <= nvars; k++)
      if ((int)ls->KKoffset[k][0] != offs[l][k]) {
        std::cout << "KKoffset differs from fh_system_elem_dofs" << std::endl;
        return 3;
      }
    // the smoother exactly as SteadyNavierStokesParallel/main.cpp:155-179 sets it up: blocks of `nblock` elements, `nschur` Schur variables
    // (the application: 0 and 4), GMRES around the block-Schwarz preconditioner on every level above the coarsest
    if (lsa) {
      lsa->SetNumberOfSchurVariables((unsigned short)nschur);
      lsa->SetElementBlockNumber((unsigned)nblock);
    }
    Sol[l] = NumericVector::build().release();
    Sol[l]->init(ndof, ndof, false, SERIAL);
    // sparsity from the element couplings of the stacked variables
    std::vector<int> rp(ndof + 1), col;
    hip_check(fh_pattern_from_elements(nel, nd, es.data(), ndof, rp.data(), nullptr), "pattern");
    col.resize(rp[ndof]);
    hip_check(fh_pattern_from_elements(nel, nd, es.data(), ndof, rp.data(), col.data()), "pattern");
    fh_mat_t K;
    hip_check(fh_mat_create_csr(hip_context(), ndof, ndof, rp.data(), col.data(), nullptr, &K), "KK");
    static_cast<HipMatrix*>(ls->_KK)->adopt(K);
    if (pw) hip_check(fh_ns_pw_assembler_create(hip_context(), geom, 3, nel, nloc, ed.data(), nnode, xy.data(), K, &as[l]), "assembler");
    else if (stab) hip_check(fh_ns_stab_assembler_create(hip_context(), geom, 3, nel, nloc, ed.data(), nnode, own[0], xy.data(), K, &as[l]), "assembler");
    else hip_check(fh_ns_assembler_create(hip_context(), geom, 3, nel, nloc, ed.data(), nnode, own[0], xy.data(), K, &as[l]), "assembler");
    // GenerateBdc: boundary faces in element order, nodes of the face, boundary function at the node
    std::map<int, double> val;
    for (int k = 0; k < nvars; k++) {
      if (fe[k] == 4) continue;                             // the element-owned pressure carries no boundary condition
      const int nck = fe[k] == 2 ? 9 : 4;
      for (int iel = 0; iel < nel; iel++)
        for (int f = 0; f < 4; f++) {
          const int flag = ff[(size_t)iel * 4 + f];
          if (flag >= -1) continue;
          int nfn = 0, loc[9];
          hip_check(fh_fe_face_nodes(geom, 2, f, &nfn, loc), "face nodes");
          for (int q = 0; q < nfn; q++) {
            if (loc[q] >= nck) continue;
            const int node = ed[(size_t)iel * nloc + loc[q]];
            double v;
            const bool dirichlet = pw ? SetBoundaryConditionCylinder(&xy[(size_t)node * dim], names[k], v, -(flag + 1))
                                      : SetBoundaryConditionCavityFlow(&xy[(size_t)node * dim], names[k], v, -(flag + 1));
            if (dirichlet) val[offs[l][k] + node] = v;
          }
        }
    }
    std::vector<double> vals;
    for (auto& kv : val) {
      bdc[l].push_back(kv.first);
      vals.push_back(kv.second);
    }
    for (int k = 0; k < nvars; k++) {                       // the flag vectors: 0 at the Dirichlet nodes of each variable
      std::vector<int> nodes;
      for (int row : bdc[l])
        if (row >= offs[l][k] && row < offs[l][k + 1]) nodes.push_back(row - offs[l][k]);
      fsol[l]->_Bdc[k]->insert(std::vector<double>(nodes.size(), 0.), nodes);
      fsol[l]->_Bdc[k]->close();
    }
    Sol[l]->zero();
    if (pw && l == 0) {                                     // Initialize("U", InitVariableU): the inflow parabola everywhere (the F-cycle prolongs upwards from here)
      std::vector<double> u0(nnode);
      std::vector<int> rows(nnode);
      for (int i = 0; i < nnode; i++) {
        rows[i] = offs[l][0] + i;
        u0[i] = InflowProfile(xy[(size_t)i * dim + 1]);
      }
      Sol[l]->insert_vector_blocked(u0, rows);
    }
    Sol[l]->insert_vector_blocked(vals, bdc[l]);
    ls->set_solver_type(lsolver == 0 ? GMRES : RICHARDSON);                       // SetSolverFineGrids(GMRES)
    if (lsolver) ls->SetRichardsonScaleFactor(0.6);
    ls->set_preconditioner_type(ILU_PRECOND);         // SetPreconditionerFineGrids(ILU_PRECOND): one ILU(0) application per block
    if (lsa) lsa->SetAsmExactInColourOrder(coloured != 0);
    if (l > 0) {
      for (int copy = 0; copy < 2; copy++) {
        fh_mat_t P;
        hip_check(fh_build_system_prolongator(hip_context(), msh[l - 1], msh[l], nvars, fe, &P), "BuildProlongatorMatrix");
        HipMatrix* hp = new HipMatrix();
        hp->adopt(P);
        (copy ? PPsol : PP)[l] = hp;
      }
      PP[l]->mat_zero_rows(bdc[l], 0.);                                                       // ZeroInterpolatorDirichletNodes
      hip_check(fh_mat_zero_cols(static_cast<HipMatrix*>(PP[l])->handle(), (int)bdc[l - 1].size(), bdc[l - 1].data()), "zero cols");
    }
  }

  // ---- NonLinearImplicitSystem::MGsolve, F_CYCLE --------------------------------------------------------------------------
  std::vector<unsigned> vars = {0, 1, 2};
  int total_newton = 0, max_linear = 0;
  unsigned counter = 0;                  // the callback's own call counter (main.cpp:387, :485-489)
  for (int ig = 0; ig < nlev; ig++) {
    for (int it = 0; it < (outer_pre == 1 ? 90 : 30); it++) {          // SetMaxNumberOfNonLinearIterations(90) in the application
      LinearEquationSolver* top = LinSolver[ig];
      top->SetResZero();
      double IRe = nu;
      if (stab) {
        const double DRe = 1 + (counter * counter) * 5;
        IRe = (DRe * (counter + 1) < 10000) ? 1. / (DRe * (counter + 1)) : 1. / 10000.;
        std::cout << "iteration=" << counter << " Reynolds Number = " << 1. / IRe << std::endl;
        counter++;
        hip_check(fh_assemble_navier_stokes_stab(as[ig], static_cast<HipVector*>(Sol[ig])->handle(), IRe, static_cast<HipMatrix*>(top->_KK)->handle(),
                                                 static_cast<HipVector*>(top->_RES)->handle()),
                  "assemble");
      } else
        hip_check(fh_assemble_navier_stokes(as[ig], static_cast<HipVector*>(Sol[ig])->handle(), nu, static_cast<HipMatrix*>(top->_KK)->handle(),
                                            static_cast<HipVector*>(top->_RES)->handle()),
                  "assemble");
      for (int i = ig; i > 0; i--) LinSolver[i - 1]->_KK->matrix_PtAP(*PP[i], *LinSolver[i]->_KK, it > 0);
      if (outer_pre == 1) {
        // SteadyNavierStokesParallel/main.cpp:148-185: SetOuterSolver(PREONLY), one pre- and one post-smoothing step, at most two
        // linear iterations (V-cycles) per nonlinear one (LinearImplicitSystem.cpp:385-411)
        top->MGInit(MULTIPLICATIVE, ig + 1, PREONLY);
        top->SetTolerances(1e-12, 1e-20, 1e50, 4, 30);
        for (int i = 0; i <= ig; i++) LinSolver[i]->MGSetLevel(top, ig, vars, PP[i], PP[i], 1, i ? 1 : 0);
        top->SetEpsZero();
        for (int lin = 0; lin < 2; lin++) top->MGSolve(lin == 0);
      } else {
        top->MGInit(MULTIPLICATIVE, ig + 1, outer_pre == 2 ? FGMRES : GMRES);     // 2: the flexible form (cycles with GMRES level solvers)
        top->SetTolerances(1e-11, 1e-50, 1e50, 60, 30);
        // (the known-answer test: four GMRES iterations per smoothing step, SetTolerances(1.e-12, 1.e-20, 1.e+50, 4), main.cpp:153)
        for (int i = 0; i <= ig; i++) LinSolver[i]->MGSetLevel(top, ig, vars, PP[i], PP[i], i ? (pw ? 4 : 2) : 1, i ? (pw ? 4 : 2) : 0);
        top->SetEpsZero();
        top->MGSolve(true);
      }
      *Sol[ig] += *top->_EPS;                                                                  // Solution::UpdateSol
      max_linear = std::max(max_linear, static_cast<LinearEquationSolverHip*>(top)->last_iterations());
      top->MGClear();
      total_newton++;
      // HasNonLinearConverged: max over variables of ||Eps_k|| / ||Sol_k||
      std::vector<double> eps, sol;
      top->_EPS->localize(eps);
      Sol[ig]->localize(sol);
      double worst = 0.;
      for (int k = 0; k < nvars; k++) {
        double ne = 0., ns = 0.;
        for (int i = offs[ig][k]; i < offs[ig][k + 1]; i++) {
          ne += eps[i] * eps[i];
          ns += sol[i] * sol[i];
        }
        worst = std::max(worst, std::sqrt(ne) / (std::sqrt(ns) + 1.e-50));
      }
      std::cout << "     ********* Level Max " << ig + 1 << " Nonlinear iteration " << it + 1 << " Eps_l2norm/Sol_l2norm = " << worst << std::endl;
      if (worst < 1.e-10 && (!stab || IRe == 1. / 10000.)) break;          // (the continuation has to have reached its final Reynolds number)
    }
    if (ig + 1 < nlev) Sol[ig + 1]->matrix_mult(*Sol[ig], *PPsol[ig + 1]);                    // ProlongatorSol
  }
  std::vector<double> sol;
  Sol[nlev - 1]->localize(sol);
  std::cout << "newton steps = " << total_newton << "  max linear iterations = " << max_linear << std::endl;
  FILE* f = fopen(argv[4], "wb");
  fwrite(sol.data(), sizeof(double), sol.size(), f);
  fclose(f);
  for (int l = 0; l < nlev; l++) {
    fh_ns_assembler_destroy(as[l]);
    LinSolver[l]->MGClear();
    LinSolver[l]->DeletePde();
    delete LinSolver[l];
    for (NumericVector* f2 : fsol[l]->_Bdc) delete f2;
    delete fsol[l];
    delete fmesh[l];
    delete PP[l];
    delete PPsol[l];
    delete Sol[l];
    fh_mesh_destroy(msh[l]);
  }
  return 0;
}
