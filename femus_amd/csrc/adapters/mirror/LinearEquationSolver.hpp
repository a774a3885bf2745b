// Mirror of FEMuS's per-level solver object as the hot path uses it: LinearEquation (03_solvers/LinearEquation.hpp:41-300: the
// level's operators _KK/_KKamr, _RES/_RESC/_EPS/_EPSC, the variable offsets KKoffset/KKIndex and the Dirichlet flag vectors _Bdc)
// and LinearEquationSolver on top of it (03_solvers/LinearEquationSolver.hpp:54-261: factory, Solve, SetTolerances, the MG calls),
// with the members' names, signatures and access the real classes have.  Mesh and Solution are reduced to what these two classes
// read from them (Mesh::_dofOffset, Mesh.hpp; the Solution's mesh pointer and its _Bdc vectors, Solution.hpp).
#pragma once
#include <cstdio>
#include <memory>
#include <string>
#include <vector>
#include "FemusEnums.hpp"
#include "NumericVector.hpp"
#include "SparseMatrix.hpp"

namespace femus {

// ParallelObject.hpp:33-67
class ParallelObject {
 public:
  ParallelObject() : _nprocs(1), _iproc(0) {}
  int n_processors() const { return _nprocs; }
  int processor_id() const { return _iproc; }

 protected:
  int _nprocs, _iproc;
};

// Mesh.hpp: _dofOffset[soltype][proc] .. [proc + 1] is the range of mesh dofs of a family a rank owns (soltype 0 linear, 1 serendipity,
// 2 biquadratic, 3 / 4 discontinuous)
// Elem.hpp:471-477: the elements sharing a vertex with an element, itself first (`BuildElementNearElement`, Elem.cpp:494-528)
class elem {
 public:
  unsigned GetElementNearElementSize(const unsigned& iel, const unsigned& layers) const { return (layers == 0) ? 1 : (unsigned)_elementNearElement[iel].size(); }
  unsigned GetElementNearElement(const unsigned& iel, const unsigned& j) const { return _elementNearElement[iel][j]; }
  std::vector<std::vector<unsigned> > _elementNearElement;     // (MyMatrix <unsigned> in the FEMuS tree)
};

class Mesh {
 public:
  std::vector<unsigned> _dofOffset[5];
  unsigned GetLevel() const { return _level; }
  unsigned _level = 0;
  // what BuildASMIndex reads (Mesh.hpp:167, 183, 211, 234, 424, 496); the element tables are filled by whoever owns the mesh
  elem* GetMeshElements() const { return const_cast<elem*>(&_el); }
  unsigned GetNumberOfElements() const { return (unsigned)_elementMaterial.size(); }
  unsigned GetElementOffset(const unsigned iproc_in) const { return _elementOffset[iproc_in]; }
  short unsigned GetElementMaterial(const unsigned& iel) const { return _elementMaterial[iel]; }
  unsigned GetElementDofNumber(const unsigned& iel, const unsigned& type) const { return _elementDofNumber[type]; }
  unsigned GetSolutionDof(const unsigned& i, const unsigned& iel, const short unsigned& solType) const {
    // Lagrange families share the biquadratic node ids; the discontinuous ones belong to the element: constant = element id, linear =
    // i * (number of elements) + iel on one rank (Mesh.cpp:1057-1070)
    return solType < 3 ? _elementDof[(size_t)iel * _nloc + i] : (solType == 4 ? i * GetNumberOfElements() + iel : iel);
  }
  unsigned BisectionSearch_find_processor_of_dof(const unsigned& dof, const short unsigned& solType) const {
    unsigned p = 0;
    while (p + 2 < _dofOffset[solType].size() && dof >= _dofOffset[solType][p + 1]) p++;
    return p;
  }
  elem _el;
  std::vector<unsigned> _elementOffset;        // [nprocs + 1]
  std::vector<short unsigned> _elementMaterial;
  unsigned _elementDofNumber[5] = {0, 0, 0, 1, 1};
  std::vector<unsigned> _elementDof;           // [nel * _nloc]
  unsigned _nloc = 0;
};

// Solution.hpp: the mesh it lives on and, per solution, the boundary flag vector (2 free, 1 AMR-constrained, 0 Dirichlet;
// MultiLevelSolution.cpp:725-840)
class Solution {
 public:
  explicit Solution(Mesh* msh) : _msh(msh) {}
  Mesh* GetMesh() { return _msh; }
  std::vector<NumericVector*> _Bdc;

 private:
  Mesh* _msh;
};

class LinearEquation : public ParallelObject {
 public:
  LinearEquation(Solution* other_solution);                                    // LinearEquation.cpp:40-56
  ~LinearEquation();
  // LinearEquation.cpp:107-342: stores the variable lists, builds KKIndex / KKoffset (:212-237) and creates _EPS, _EPSC, _RES, _RESC
  // and _KK through the factories (:273-339).  The sparsity pre-count of the real class (GetSparsityPatternSize, :407-548) stays
  // with FEMuS: here the caller initialises _KK (SparseMatrix::init) itself.
  void InitPde(const std::vector<unsigned>& _SolPdeIndex, const std::vector<unsigned>& SolType, const std::vector<char*>& SolName,
               std::vector<NumericVector*>* Bdc_other, const unsigned& other_gridn, std::vector<bool>& SparsityPattern_other);
  void DeletePde();
  inline const Mesh* GetMeshFromLinEq() const { return _msh; }
  unsigned GetSystemDof(const unsigned& index_sol, const unsigned& kkindex_sol, const unsigned& i, const unsigned& iel) const {   // LinearEquation.cpp:76-85
    const unsigned soltype = _SolType[index_sol];
    const unsigned idof = _msh->GetSolutionDof(i, iel, soltype);
    const unsigned isubdom = _msh->BisectionSearch_find_processor_of_dof(idof, soltype);
    return KKoffset[kkindex_sol][isubdom] + idof - _msh->_dofOffset[soltype][isubdom];
  }
  std::vector<std::vector<unsigned> > KKoffset;   // [nvars + 1][nprocs]
  std::vector<int> KKIndex;                       // [nvars + 1]
  void SwapMatrices() { SparseMatrix* t = _KK; _KK = _KKamr; _KKamr = t; }
  SparseMatrix* _KK;
  SparseMatrix* _KKamr;
  void SetResZero();
  NumericVector *_RES, *_RESC;
  void SetEpsZero();
  void SumEpsCToEps();
  NumericVector *_EPS, *_EPSC;

 protected:
  unsigned _gridn;
  std::vector<unsigned> _SolPdeIndex;
  Solution* _solution;
  std::vector<unsigned> _SolType;
  std::vector<char*> _SolName;
  const std::vector<NumericVector*>* _Bdc;

 private:
  const Mesh* _msh;
};

class Preconditioner;
class FieldSplitTree;

class LinearEquationSolver : public LinearEquation {
 public:
  LinearEquationSolver(const unsigned& igrid, Solution* other_solution);        // LinearEquationSolver.hpp:264-277
  virtual ~LinearEquationSolver();
  virtual void Clear() {}
  // LinearEquationSolver.cpp:40-74: FEMuS_ASM returns the block Schwarz variant (LinearEquationSolverPetscAsm in the reference)
  static std::unique_ptr<LinearEquationSolver> build(const unsigned& igrid, Solution* other_solution, const LinearEquationSolverType& smoother_type,
                                                     const SolverPackage solver_package = LSOLVER);
  bool initialized() const { return _is_initialized; }
  void SetPrintSolverInfo(const bool& printInfo) { _printSolverInfo = printInfo; }
  virtual void Solve(const std::vector<unsigned>& VariableTobeSolved, const bool& ksp_clean) = 0;                 // :113
  virtual void SetTolerances(const double& rtol, const double& atol, const double& divtol, const unsigned& maxits,
                             const unsigned& restart) = 0;                                                        // :128
  // :132 of the real header is `virtual KSP* GetKSP()` -- a PETSc type in the abstract interface; a non-PETSc backend keeps the
  // base-class default there (warn and abort)
  void set_solver_type(const SolverType st) { _levelSolverType = st; }
  SolverType solver_type() const { return _levelSolverType; }
  virtual void MGInit(const MgSmootherType& mg_smoother_type, const unsigned& levelMax, const SolverType& mgSolverType) {
    std::cout << "Warning InitMG(...) is not available for this smoother\n";
    abort();
  }
  virtual void MGClear() {
    std::cout << "Warning ClearMG() is not available for this smoother\n";
    abort();
  }
  virtual void MGSetLevel(LinearEquationSolver* LinSolver, const unsigned& levelMax, const std::vector<unsigned>& variable_to_be_solved,
                          SparseMatrix* PP, SparseMatrix* RR, const unsigned& npre, const unsigned& npost) = 0;                 // :166
  virtual void MGSolve(const bool ksp_clean) = 0;                                                                               // :172
  void set_preconditioner_type(const PreconditionerType pct) { _preconditioner_type = pct; }                                   // LinearEquationSolver.cpp
  PreconditionerType preconditioner_type() const { return _preconditioner_type; }
  virtual void SetRichardsonScaleFactor(const double& richardsonScaleFactor) = 0;                                               // :210
  // ASM / Vanka options (:218-247): accepted by every solver, used by the FEMuS_ASM one
  virtual void SetElementBlockNumber(const unsigned& block_elemet_number) {
    std::cout << "Warning SetElementBlockNumber(const unsigned &) is not available for this smoother\n";
  }
  virtual void SetElementBlockNumber(const char all[], const unsigned& overlap = 1) {
    std::cout << "Warning SetElementBlockNumber(const char [], const unsigned & ) is not available for this smoother\n";
  }
  virtual void SetNumberOfSchurVariables(const unsigned short& NSchurVar) {
    std::cout << "Warning SetNumberOfSchurVariables(const unsigned short &) is not available for this smoother\n";
  }
  virtual void SetFieldSplitTree(FieldSplitTree* fieldSplitTree) {
    std::cout << "SetFieldSplitTree(const FieldSpliTreeStructure & fieldSplitTree) is not available for this smoother\n";
  }

 protected:
  bool _is_initialized;
  bool _printSolverInfo;
  SolverType _levelSolverType;
  SolverType _mgSolverType;
  PreconditionerType _preconditioner_type;
  Preconditioner* _preconditioner;
  bool same_preconditioner;
};

inline LinearEquationSolver::LinearEquationSolver(const unsigned& igrid, Solution* other_solution)
    : LinearEquation(other_solution), _is_initialized(false), _printSolverInfo(false), _levelSolverType(GMRES), _mgSolverType(GMRES),
      _preconditioner_type(ILU_PRECOND), _preconditioner(NULL), same_preconditioner(false) {}
inline LinearEquationSolver::~LinearEquationSolver() { this->Clear(); }

}  // namespace femus
