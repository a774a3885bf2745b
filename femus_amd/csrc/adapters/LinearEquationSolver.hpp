// Mirror of the multigrid part of FEMuS's LinearEquationSolver (src/08_algebra.../03_solvers_with_preconditioner/
// LinearEquationSolver.hpp:54-261) as the hot path uses it: per-level object owning _KK, _RES, _RESC, _EPS, _EPSC,
// with MGInit / MGSetLevel / MGSolve / MGClear.  Mesh, Solution and the _Bdc flag vectors stay with FEMuS; here the
// Dirichlet list (the result of BuildBdcIndex, LinearEquationSolverPetsc.cpp:53-90) is handed over directly.
#pragma once
#include <memory>
#include <vector>
#include "NumericVector.hpp"
#include "SparseMatrix.hpp"

namespace femus {

enum MgSmootherType { FULL = 0, MULTIPLICATIVE, ADDITIVE, KASKADE };                    // MgSmootherEnum.hpp
enum SolverType { CG = 0, GMRES = 7, RICHARDSON = 12, PREONLY = 14 };                    // SolverTypeEnum (subset, own ids)
enum PreconditionerType { JACOBI_PRECOND = 2, SOR_PRECOND = 4, MLU_PRECOND = 14 };       // PrecondtypeEnum (subset)
enum LinearEquationSolverType { FEMuS_DEFAULT = 0, FEMuS_ASM = 1 };                      // LinearEquationSolverEnum.hpp (subset)

class LinearEquationSolver {
 public:
  virtual ~LinearEquationSolver() {}
  // LinearEquationSolver.cpp:40-74: FEMuS_ASM returns the block Schwarz variant (LinearEquationSolverPetscAsm in the reference)
  static std::unique_ptr<LinearEquationSolver> build(const unsigned& igrid, const SolverPackage solver_package = HIP_SOLVERS,
                                                     const LinearEquationSolverType smoother_type = FEMuS_DEFAULT);
  // ASM / Vanka options (LinearEquationSolver.hpp:176-204): accepted by every solver, used by the FEMuS_ASM one
  virtual void SetElementBlockNumber(const unsigned& block_elemet_number) {}
  virtual void SetNumberOfSchurVariables(const unsigned short& NSchurVar) {}
  // the per-level algebra objects of LinearEquation (LinearEquation.hpp): raw pointers, owned by this object
  SparseMatrix* _KK = nullptr;
  NumericVector *_RES = nullptr, *_RESC = nullptr, *_EPS = nullptr, *_EPSC = nullptr;

  virtual void SetTolerances(const double& rtol, const double& atol, const double& divtol, const unsigned& maxits,
                             const unsigned& restart) = 0;                                                        // :118
  virtual void SetRichardsonScaleFactor(const double& richardsonScaleFactor) = 0;                                 // :162
  void SetSolverType(const SolverType st) { _solver_type = st; }
  void SetPreconditionerType(const PreconditionerType pct) { _preconditioner_type = pct; }
  // result of BuildBdcIndex for this level (sorted system rows with _Bdc < 1.5)
  virtual void SetBdcIndex(const std::vector<int>& bdc) = 0;
  virtual void MGInit(const MgSmootherType& mg_smoother_type, const unsigned& levelMax, const SolverType& mgSolverType) = 0;   // :101
  virtual void MGSetLevel(LinearEquationSolver* LinSolver, const unsigned& levelMax, const std::vector<unsigned>& variable_to_be_solved,
                          SparseMatrix* PP, SparseMatrix* RR, const unsigned& npre, const unsigned& npost) = 0;                 // :106
  virtual void MGSolve(const bool ksp_clean) = 0;                                                                               // :112
  virtual void MGClear() = 0;                                                                                                   // :104
  void SetEpsZero() { _EPS->zero(); _EPSC->zero(); }                                                                           // LinearEquation.cpp
  void SetResZero() { _RES->zero(); }

 protected:
  SolverType _solver_type = RICHARDSON;
  PreconditionerType _preconditioner_type = JACOBI_PRECOND;
};

}  // namespace femus
