// What a FEMuS build gets from its own library and the stand-alone (mirror) build has to provide itself: the three factories
// (NumericVector.cpp:35-56, SparseMatrix.cpp:42-63, LinearEquationSolver.cpp:40-74 -- INTEGRATION.md shows the case a maintainer
// adds to each) and the members of LinearEquation the hot path uses (LinearEquation.cpp:40-56, :107-342, :378-405).
#include "HipBackend.hpp"

namespace femus {

std::unique_ptr<NumericVector> NumericVector::build(const SolverPackage solver_package) {
  if (solver_package != HIP_SOLVERS) {
    std::cout << "SolverPackage solver_package:  Something is wrong here" << std::endl;
    abort();
  }
  return std::unique_ptr<NumericVector>(new HipVector());
}
std::unique_ptr<SparseMatrix> SparseMatrix::build(const SolverPackage solver_package) {
  if (solver_package != HIP_SOLVERS) {
    std::cout << "SolverPackage solver_package:  Something is wrong here" << std::endl;
    abort();
  }
  return std::unique_ptr<SparseMatrix>(new HipMatrix());
}
std::unique_ptr<LinearEquationSolver> LinearEquationSolver::build(const unsigned& igrid, Solution* other_solution,
                                                                  const LinearEquationSolverType& smoother_type, const SolverPackage solver_package) {
  if (solver_package != HIP_SOLVERS) {
    std::cout << "SolverPackage solver_package:  Something is wrong here" << std::endl;
    abort();
  }
  switch (smoother_type) {
    case FEMuS_DEFAULT: return std::unique_ptr<LinearEquationSolver>(new LinearEquationSolverHip(igrid, other_solution));
    case FEMuS_ASM: return std::unique_ptr<LinearEquationSolver>(new LinearEquationSolverHipAsm(igrid, other_solution));
    default: break;
  }
  std::cout << "LinearEquationSolver::build: this smoother type is not served by the HIP backend" << std::endl;
  abort();
}

LinearEquation::LinearEquation(Solution* other_solution)
    : _KK(NULL), _KKamr(NULL), _RES(NULL), _RESC(NULL), _EPS(NULL), _EPSC(NULL), _gridn(0), _solution(other_solution), _Bdc(NULL),
      _msh(other_solution ? other_solution->GetMesh() : NULL) {}

LinearEquation::~LinearEquation() {}

void LinearEquation::InitPde(const std::vector<unsigned>& SolPdeIndex_other, const std::vector<unsigned>& SolType_other,
                             const std::vector<char*>& SolName_other, std::vector<NumericVector*>* Bdc_other, const unsigned& other_gridn,
                             std::vector<bool>& /*SparsityPattern_other*/) {
  _SolPdeIndex = SolPdeIndex_other;
  _gridn = other_gridn;
  _SolType = SolType_other;
  _SolName = SolName_other;
  _Bdc = Bdc_other;
  const int np = n_processors(), ip = processor_id();
  const unsigned nvar = (unsigned)_SolPdeIndex.size();
  // KKIndex: variable sizes summed over the ranks, in offset mode; KKoffset[k][p]: first row of variable k on rank p -- the rows of
  // a rank hold its variables one after another (LinearEquation.cpp:212-237)
  KKIndex.assign(nvar + 1u, 0);
  for (unsigned i = 1; i <= nvar; i++) KKIndex[i] = KKIndex[i - 1] + (int)_msh->_dofOffset[_SolType[_SolPdeIndex[i - 1]]][np];
  KKoffset.assign(nvar + 1u, std::vector<unsigned>(np, 0u));
  for (int p = 0; p < np; p++) {
    if (p > 0) KKoffset[0][p] = KKoffset[nvar][p - 1];
    for (unsigned j = 1; j <= nvar; j++) {
      const unsigned t = _SolType[_SolPdeIndex[j - 1]];
      KKoffset[j][p] = KKoffset[j - 1][p] + (_msh->_dofOffset[t][p + 1] - _msh->_dofOffset[t][p]);
    }
  }
  const int N = KKIndex[nvar], n_local = (int)(KKoffset[nvar][ip] - KKoffset[0][ip]);
  // the level's vectors and matrix through the factories (:273-339); one rank: SERIAL vectors, no ghost list
  for (NumericVector** v : {&_EPS, &_EPSC, &_RES, &_RESC}) {
    *v = NumericVector::build().release();
    (*v)->init(N, n_local, false, np == 1 ? SERIAL : PARALLEL);
  }
  _KK = SparseMatrix::build().release();
}

void LinearEquation::DeletePde() {
  delete _KK;
  delete _KKamr;
  delete _EPS;
  delete _EPSC;
  delete _RES;
  delete _RESC;
  _KK = _KKamr = NULL;
  _EPS = _EPSC = _RES = _RESC = NULL;
}

void LinearEquation::SetResZero() { _RES->zero(); }
void LinearEquation::SetEpsZero() {
  _EPS->zero();
  _EPSC->zero();
}
void LinearEquation::SumEpsCToEps() { *_EPS += *_EPSC; }

}  // namespace femus
