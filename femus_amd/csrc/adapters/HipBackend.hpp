// Concrete MI355X backend classes behind FEMuS's algebra interface: thin shells over the C-ABI (include/femus_hip.h).
//   HipVector : NumericVector      (replaces PetscVector,  src/03_algebra/00_vectors/PetscVector.{hpp,cpp})
//   HipMatrix : SparseMatrix       (replaces PetscMatrix,  src/03_algebra/01_matrices/PetscMatrix.{hpp,cpp})
//   LinearEquationSolverHip        (replaces LinearEquationSolverPetsc, 03_solvers/LinearEquationSolverPetsc.{hpp,cpp})
// Error convention of the reference: print and abort.
#pragma once
#include <map>
#include "../../../include/femus_hip.h"
#include "LinearEquationSolver.hpp"

namespace femus {

fh_ctx_t hip_context();          // process-wide context (FemusInit equivalent); aborts when no device is usable
void hip_check(int rc, const char* what);

class HipVector : public NumericVector {
 public:
  HipVector() {}
  HipVector(const HipVector&) = delete;
  HipVector& operator=(const HipVector& o) { this->operator=(static_cast<const NumericVector&>(o)); return *this; }
  ~HipVector() override { clear(); }
  std::unique_ptr<NumericVector> clone() const override;
  void clear() override;
  void close() override { _is_closed = true; }
  void init(const int N, const int n_local, const bool fast = false, const ParallelType type = AUTOMATIC) override;
  void init(const int N, const bool fast = false, const ParallelType type = AUTOMATIC) override { init(N, N, fast, type); }
  void init(const int N, const int n_local, const std::vector<int>& ghost, const bool fast = false,
            const ParallelType type = AUTOMATIC) override;
  void init(const NumericVector& other, const bool fast = false) override;
  void set(const int i, const double value) override;
  void add(const int i, const double value) override;
  void zero() override;
  NumericVector& operator=(const double s) override;
  NumericVector& operator=(const NumericVector& V) override;
  NumericVector& operator=(const std::vector<double>& v) override;
  double min() const override;
  double max() const override;
  double sum() const override;
  double l1_norm() const override;
  double l2_norm() const override;
  double linfty_norm() const override;
  int size() const override { return _n_global; }
  int local_size() const override { return _n_local; }
  int first_local_index() const override { return _first; }
  int last_local_index() const override { return _first + _n_local; }
  double operator()(const int i) const override;
  void get(const std::vector<int>& index, std::vector<double>& values) const override;
  NumericVector& operator+=(const NumericVector& V) override { add(1., V); return *this; }
  NumericVector& operator-=(const NumericVector& V) override { add(-1., V); return *this; }
  void add(const double s) override;
  void add(const NumericVector& V) override { add(1., V); }
  void add(const double a, const NumericVector& v) override;
  void add_vector_blocked(const std::vector<double>& v, const std::vector<int>& dof_indices) override;
  void add_vector_blocked(const std::vector<double>& v, const std::vector<unsigned>& dof_indices) override;
  void insert_vector_blocked(const std::vector<double>& v, const std::vector<int>& dof_indices) override;
  void add_vector(const NumericVector& v, const SparseMatrix& A) override;
  void resid(const NumericVector& rhs, const NumericVector& v, const SparseMatrix& A) override;
  void matrix_mult(const NumericVector& v, const SparseMatrix& A) override;
  void matrix_mult_transpose(const NumericVector& v, const SparseMatrix& A) override;
  void scale(const double factor) override;
  void abs() override;
  double dot(const NumericVector&) const override;
  void localize(std::vector<double>& v_local) const override;
  void localize_to_all(std::vector<double>& v_local) const override { localize(v_local); }   // nprocs = 1
  void pointwise_mult(const NumericVector& vec1, const NumericVector& vec2) override;
  fh_vec_t handle() const { return _v; }

 private:
  fh_vec_t _v = nullptr;
  int _n_global = 0, _n_local = 0, _first = 0;
};

class HipMatrix : public SparseMatrix {
 public:
  HipMatrix() {}
  ~HipMatrix() override { clear(); }
  void clear() override;
  void init(const int m, const int n, const int m_l, const int n_l, const std::vector<int>& n_nz, const std::vector<int>& n_oz) override;
  // fast path: the CSR pattern is known up front (fh_pattern_from_elements) -- no host staging at all
  void init_pattern(const int m, const int n, const std::vector<int>& rowptr, const std::vector<int>& col);
  void adopt(fh_mat_t handle);                  // take ownership of a C-ABI matrix (prolongators, PtAP results)
  void set(const int i, const int j, const double value) override;
  void add(const int i, const int j, const double value) override;
  void zero() override;
  void close() const override;
  double operator()(const int i, const int j) const override;
  int MatGetRowM(const int i_val, int* cols = NULL, double* vals = NULL) override;
  bool closed() const override { return _closed; }
  int m() const override { return _m; }
  int n() const override { return _n; }
  int row_start() const override { return 0; }
  int row_stop() const override { return _m; }
  void insert_row(const int row, const int ncols, const std::vector<int>& cols, double* values) override;
  void add_matrix_blocked(const std::vector<double>& mat_value, const std::vector<int>& rows, const std::vector<int>& cols) override;
  void add_matrix_blocked(const std::vector<double>& mat_value, const std::vector<unsigned>& rows, const std::vector<unsigned>& cols) override;
  void matrix_PtAP(const SparseMatrix& mat_P, const SparseMatrix& mat_A, const bool& reuse) override;
  void matrix_ABC(const SparseMatrix& mat_A, const SparseMatrix& mat_B, const SparseMatrix& mat_C, const bool& reuse) override;
  void matrix_RightMatMult(const SparseMatrix& mat_A) override;
  void matrix_LeftMatMult(const SparseMatrix& mat_A) override;
  void matrix_get_diagonal_values(const std::vector<int>& index, std::vector<double>& value) const override;
  double l1_norm() const override;
  double linfty_norm() const override;
  void get_diagonal(NumericVector& dest) const override;
  void get_transpose(SparseMatrix& dest) const override;
  void mat_zero_rows(const std::vector<int>& index, const double& diagonal_value) const override;
  fh_mat_t handle() const { close(); return _A; }

 private:
  // before the first close() entries are staged on the host (the reference relies on MatSetValues growing the
  // pattern); close() freezes the pattern into a device CSR, after which add/insert go straight to the device
  mutable fh_mat_t _A = nullptr;
  mutable std::vector<std::map<int, double>> _stage;
  mutable bool _closed = false;
  int _m = 0, _n = 0;
};

class LinearEquationSolverHip : public LinearEquationSolver {
 public:
  explicit LinearEquationSolverHip(const unsigned& igrid) : _level(igrid) {}
  ~LinearEquationSolverHip() override;
  void SetTolerances(const double& rtol, const double& atol, const double& divtol, const unsigned& maxits, const unsigned& restart) override;
  void SetRichardsonScaleFactor(const double& s) override { _richardsonScaleFactor = s; }
  void SetBdcIndex(const std::vector<int>& bdc) override { _bdcIndex = bdc; }
  void MGInit(const MgSmootherType& mg_smoother_type, const unsigned& levelMax, const SolverType& mgSolverType) override;
  void MGSetLevel(LinearEquationSolver* LinSolver, const unsigned& levelMax, const std::vector<unsigned>& variable_to_be_solved,
                  SparseMatrix* PP, SparseMatrix* RR, const unsigned& npre, const unsigned& npost) override;
  void MGSolve(const bool ksp_clean) override;
  void MGClear() override;
  int last_iterations() const { return _its; }
  double last_residual() const { return _rnorm; }

 protected:
  // smoother of this level as handed to fh_mg_set_level; the ASM variant overrides it
  virtual int smoother_id() const;
  virtual void attach_smoother_data(fh_mg_t, int) {}

 private:
  void SetPenalty();                 // LinearEquationSolverPetsc.cpp:428-436
  void ZerosBoundaryResiduals();     // :417-424
  unsigned _level;
  std::vector<int> _bdcIndex;
  double _rtol = 1e-5, _abstol = 1e-50, _dtol = 1e5, _richardsonScaleFactor = 0.5;   // LinearEquationSolverPetsc.hpp:139-146
  int _maxits = 1000, _restart = 30;
  // top-level (the object MGInit was called on) owns the cycle
  fh_mg_t _mg = nullptr;
  unsigned _levelMax = 0;
  SolverType _mgSolverType = GMRES;
  bool _needs_setup = true;
  int _its = 0;
  double _rnorm = 0.;
};

// Block Schwarz smoother (LinearEquationSolverPetscAsm, petsc_asm/LinearEquationSolverPetscAsm.cpp): the blocks
// BuildASMIndex derives from the mesh (:91-276) are handed over as dof lists, e.g. from fh_mesh_vertex_patches
class LinearEquationSolverHipAsm : public LinearEquationSolverHip {
 public:
  explicit LinearEquationSolverHipAsm(const unsigned& igrid) : LinearEquationSolverHip(igrid) {}
  void SetElementBlockNumber(const unsigned& n) override { _elementBlockNumber = n; }
  void SetNumberOfSchurVariables(const unsigned short& n) override { _NSchurVar = n; }
  void SetAsmBlocks(const std::vector<int>& ptr, const std::vector<int>& dofs) { _blockPtr = ptr; _blockDofs = dofs; }

 protected:
  int smoother_id() const override { return FH_SMOOTH_VANKA; }
  void attach_smoother_data(fh_mg_t mg, int level) override;

 private:
  unsigned _elementBlockNumber = 1;
  unsigned short _NSchurVar = 1;
  std::vector<int> _blockPtr, _blockDofs;
};

}  // namespace femus
