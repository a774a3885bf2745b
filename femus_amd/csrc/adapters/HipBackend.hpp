// Concrete MI355X backend classes behind FEMuS's algebra interface: thin shells over the C-ABI (include/femus_hip.h).
//   HipVector : NumericVector      (replaces PetscVector,  src/03_algebra/00_vectors/PetscVector.{hpp,cpp})
//   HipMatrix : SparseMatrix       (replaces PetscMatrix,  src/03_algebra/01_matrices/PetscMatrix.{hpp,cpp})
//   LinearEquationSolverHip        (replaces LinearEquationSolverPetsc, 03_solvers/LinearEquationSolverPetsc.{hpp,cpp})
// Error convention of the reference: print and abort.
//
// The FEMuS headers are included by their plain names, exactly as a source file inside the FEMuS tree includes them: with
// -I adapters/mirror the mirrored interface is used (stand-alone build, tests/cpp), with the -I list of the FEMuS tree the SAME
// sources derive from the reference's own classes (tests/test_adapters_vs_reference_headers.py compiles them that way).
#pragma once
#include <map>
#include "femus_hip.h"
#include "NumericVector.hpp"
#include "SparseMatrix.hpp"
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
  // close(): the synchronisation point callers rely on (PetscVector.hpp:595-612): pending adds are on the device already; a
  // GHOSTED vector with an exchange plan attached (attach_halo) refreshes its ghost entries from their owners here
  void close() override;
  void closeWithMinValues() override { not_served("closeWithMinValues"); }
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
  void insert(const std::vector<double>& v, const std::vector<int>& dof_indices) override { insert_vector_blocked(v, dof_indices); }
  void insert(const NumericVector& V, const std::vector<int>& dof_indices) override;
  void insert(const DenseVector&, const std::vector<int>&) override { not_served("insert(DenseVector)"); }
  void insert(const DenseSubVector&, const std::vector<int>&) override { not_served("insert(DenseSubVector)"); }
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
  void add_vector(const std::vector<double>& v, const std::vector<int>& dof_indices) override { add_vector_blocked(v, dof_indices); }
  void add_vector(const NumericVector& V, const std::vector<int>& dof_indices) override;
  void add_vector(const DenseVector&, const std::vector<unsigned int>&) override { not_served("add_vector(DenseVector)"); }
  void add_vector(const NumericVector& v, const SparseMatrix& A) override;
  void resid(const NumericVector& rhs, const NumericVector& v, const SparseMatrix& A) override;
  void matrix_mult(const NumericVector& v, const SparseMatrix& A) override;
  void matrix_mult_transpose(const NumericVector& v, const SparseMatrix& A) override;
  void scale(const double factor) override;
  void abs() override;
  double dot(const NumericVector&) const override;
  void swap(NumericVector& v) override;
  void localize(std::vector<double>& v_local) const override;
  void localize(NumericVector& v_local) const override;
  void localize(NumericVector& v_local, const std::vector<int>& send_list) const override;
  void localize(const int first_local_idx, const int last_local_idx, const std::vector<int>& send_list) override;
  void localize_to_one(std::vector<double>& v_local, const int proc_id = 0) const override;
  void localize_to_all(std::vector<double>& v_local) const override;
  void pointwise_mult(const NumericVector& vec1, const NumericVector& vec2) override;
  // MultiLevelSolution::SaveSolution / LoadSolution (MultiLevelSolution.cpp:1070-1126): PETSc's binary Vec layout
  void BinaryPrint(const char* fileName) override;
  void BinaryLoad(const char* fileName) override;
  // the device vector with every staged add applied (add_vector_blocked only stages, as VecSetValues does until VecAssemblyEnd)
  fh_vec_t handle() const { return rv(); }
  // several ranks (one per GPU): the exchange plan that refreshes this vector's ghosts (built from the ghost list it was
  // initialised with, fh_halo_create*) and reduces dot products / norms / min / max over the ranks.  Not owned.  Collective: the
  // ranks exchange their owned sizes here, which gives this rank's first global index (PetscVector's ownership range)
  void attach_halo(fh_halo_t halo);
  fh_halo_t halo() const { return _halo; }

 private:
  static void not_served(const char* what);
  double all_sum(double local) const;
  double all_extreme(double local, bool want_max) const;
  fh_vec_t rv() const;               // flushes the staged adds
  fh_vec_t _v = nullptr;
  fh_halo_t _halo = nullptr;
  mutable bool _pending = false;     // add_vector_blocked calls staged since the last flush
  int _n_global = 0, _n_local = 0, _first = 0;
};

class HipMatrix : public SparseMatrix {
 public:
  HipMatrix() {}
  ~HipMatrix() override { clear(); }
  void clear() override;
  using SparseMatrix::init;          // the non-pure overloads of the base class stay visible
  void init(const int m, const int n, const int m_l, const int n_l, const std::vector<int>& n_nz, const std::vector<int>& n_oz) override;
  void init(const int nr, const int nc, const std::vector<SparseMatrix*>& P) override;
  // fast path: the CSR pattern is known up front (fh_pattern_from_elements) -- no host staging at all
  void init_pattern(const int m, const int n, const std::vector<int>& rowptr, const std::vector<int>& col);
  void adopt(fh_mat_t handle);                  // take ownership of a C-ABI matrix (prolongators, PtAP results)
  void set(const int i, const int j, const double value) override;
  void add(const int i, const int j, const double value) override;
  void zero() override;
  void close() const override;
  double operator()(const int i, const int j) const override;
  int MatGetRowM(const int i_val, int* cols = NULL, double* vals = NULL) override;
  void RemoveZeroEntries(double& tolerance) override;
  bool closed() const override { return _closed; }
  void update_sparsity_pattern_old(const Graph&) override { not_served("update_sparsity_pattern_old(Graph)"); }
  void update_sparsity_pattern(const Graph&) override { not_served("update_sparsity_pattern(Graph)"); }
  void update_sparsity_pattern(int m, int n, int m_l, int n_l, const std::vector<int> n_oz, const std::vector<int> n_nz) override {
    init(m, n, m_l, n_l, n_nz, n_oz);
  }
  int m() const override { return _m; }
  int n() const override { return _n; }
  // rows this rank owns, in the global numbering: [row_start, row_stop).  A matrix over [owned | ghost] columns of several ranks
  // carries its offset (set_row_range); one rank owns everything
  int row_start() const override { return _row_start; }
  int row_stop() const override { return _row_start + _m; }
  void set_row_range(int first_row) { _row_start = first_row; }
  void add_matrix(const DenseMatrix&, const std::vector<unsigned int>&, const std::vector<unsigned int>&) override { not_served("add_matrix(DenseMatrix)"); }
  void add_matrix(const DenseMatrix&, const std::vector<unsigned int>&) override { not_served("add_matrix(DenseMatrix)"); }
  void insert_row(const int row, const int ncols, const std::vector<int>& cols, double* values) override;
  void add_matrix_blocked(const std::vector<double>& mat_value, const std::vector<int>& rows, const std::vector<int>& cols) override;
  void add_matrix_blocked(const std::vector<double>& mat_value, const std::vector<unsigned>& rows, const std::vector<unsigned>& cols) override;
  void matrix_set_off_diagonal_values_blocked(const std::vector<int>& index_rows, const std::vector<int>& index_cols, const double& value) override;
  void matrix_set_off_diagonal_values_blocked(const std::vector<int>& index_rows, const std::vector<int>& index_cols,
                                              const std::vector<double>& value) override;
  void matrix_add(const double a_in, SparseMatrix& X_in, const char pattern[]) override;
  void matrix_PtAP(const SparseMatrix& mat_P, const SparseMatrix& mat_A, const bool& reuse) override;
  void matrix_ABC(const SparseMatrix& mat_A, const SparseMatrix& mat_B, const SparseMatrix& mat_C, const bool& reuse) override;
  void matrix_RightMatMult(const SparseMatrix& mat_A) override;
  void matrix_LeftMatMult(const SparseMatrix& mat_A) override;
  void matrix_get_diagonal_values(const std::vector<int>& index, std::vector<double>& value) const override;
  void matrix_set_diagonal_values(NumericVector& D) override;
  void matrix_set_diagonal_values(const std::vector<int>& index, const double& value) override;
  void matrix_set_diagonal_values(const std::vector<int>& index, const std::vector<double>& value) override;
  void add(const double c, SparseMatrix& B) override { matrix_add(c, B, "different_nonzero_pattern"); }
  double l1_norm() const override;
  double linfty_norm() const override;
  void get_diagonal(NumericVector& dest) const override;
  void get_transpose(SparseMatrix& dest) const override;
  void mat_zero_rows(const std::vector<int>& index, const double& diagonal_value) const override;
  void print_personal(std::ostream& os = std::cout) const override;
  void print_hdf5(const std::string name = "NULL") const override { not_served("print_hdf5"); }
  fh_mat_t handle() const { close(); return _A; }

 private:
  // before the first close() the pattern is not known (the reference relies on MatSetValues growing it): inserted entries are
  // kept per row, added element blocks are logged as they come; close() builds the union pattern on all host cores, freezes it
  // into a device CSR and replays the log through the staged add.  After that add_matrix_blocked stages into the pinned ring of
  // fh_mat_stage_block and close() flushes -- no device call per element in either phase
  mutable fh_mat_t _A = nullptr;
  mutable std::vector<std::map<int, double>> _stage;
  mutable std::vector<int> _logHdr;        // per block: nrow, ncol (rows then cols in _logIdx, values row-major in _logVal)
  mutable std::vector<int> _logIdx;
  mutable std::vector<double> _logVal;
  mutable bool _pending = false;           // blocks staged on the device side since the last flush
  mutable bool _closed = false;
  void first_close() const;
  int _row_start = 0;
  static void not_served(const char* what);
  void to_host(std::vector<int>& rp, std::vector<int>& col, std::vector<double>& val) const;
};

class LinearEquationSolverHip : public LinearEquationSolver {
 public:
  LinearEquationSolverHip(const unsigned& igrid, Solution* other_solution) : LinearEquationSolver(igrid, other_solution), _level(igrid) {}
  ~LinearEquationSolverHip() override;
  // one-level solve of this level's system, as the reference's smoother-solver (LinearEquationSolverPetsc.cpp:94-160): SetPenalty,
  // ZerosBoundaryResiduals, Krylov solve with the level's preconditioner, EPS += EPSC, RES -= KK EPSC
  void Solve(const std::vector<unsigned>& VariableTobeSolved, const bool& ksp_clean) override;
  void SetTolerances(const double& rtol, const double& atol, const double& divtol, const unsigned& maxits, const unsigned& restart) override;
  void SetRichardsonScaleFactor(const double& s) override { _richardsonScaleFactor = s; }
  void MGInit(const MgSmootherType& mg_smoother_type, const unsigned& levelMax, const SolverType& mgSolverType) override;
  void MGSetLevel(LinearEquationSolver* LinSolver, const unsigned& levelMax, const std::vector<unsigned>& variable_to_be_solved,
                  SparseMatrix* PP, SparseMatrix* RR, const unsigned& npre, const unsigned& npost) override;
  void MGSolve(const bool ksp_clean) override;
  void MGClear() override;
  void SetMulticolourSor(const bool on) { _multicolourSor = on; }     // SOR_PRECOND in colour order (faster on the device, other history)
  // Backend-specific, optional: where the unknowns of THIS level's system lie (row-major [rows][dim], the rows of _KK in their order).  Given for
  // the coarsest level, the exact coarse solve of the cycle (PCLU in the reference) dissects its dense problem with them (fh_mg_set_coarse_coords);
  // inside FEMuS: the entries of Mesh::GetTopology()->_Sol[0..dim-1] at GetSolutionDof of every system row
  void SetLevelCoordinates(const int dim, const std::vector<double>& xyz) { _coordDim = dim; _coords = xyz; }
  int last_iterations() const { return _its; }
  double last_residual() const { return _rnorm; }
  const std::vector<int>& bdc_index() const { return _bdcIndex; }

 protected:
  // smoother of this level as handed to fh_mg_set_level; the ASM variant overrides it
  virtual int smoother_id() const;
  virtual void attach_smoother_data(fh_mg_t, int, const std::vector<unsigned>&) {}

 private:
  // LinearEquationSolverPetsc.hpp:81 / .cpp:53-90: the sorted system rows that are Dirichlet (flag < 1.5) or belong to variables
  // not solved for, derived from _Bdc, KKoffset and the mesh's dof offsets; built once per level (_bdcIndexIsInitialized)
  void BuildBdcIndex(const std::vector<unsigned>& variable_to_be_solved);     // HipBackendBdc.cpp
  void SetPenalty();                 // LinearEquationSolverPetsc.cpp:428-436
  void ZerosBoundaryResiduals();     // :417-424
  unsigned _level;
  std::vector<int> _bdcIndex;
  bool _bdcIndexIsInitialized = false;
  bool _multicolourSor = false;
  double _rtol = 1e-5, _abstol = 1e-50, _dtol = 1e5, _richardsonScaleFactor = 0.5;   // LinearEquationSolverPetsc.hpp:139-146
  int _maxits = 1000, _restart = 30;
  // top-level (the object MGInit was called on) owns the cycle
  fh_mg_t _mg = nullptr;
  int _coordDim = 0;
  std::vector<double> _coords;
  fh_mg_t _one = nullptr;            // one-level solver object of Solve()
  unsigned _levelMax = 0;
  bool _needs_setup = true;
  int _its = 0;
  double _rnorm = 0.;
};

// Block Schwarz smoother (LinearEquationSolverPetscAsm, petsc_asm/LinearEquationSolverPetscAsm.cpp): the blocks come from the
// reference's own setters -- SetElementBlockNumber / SetNumberOfSchurVariables -- through BuildASMIndex (HipBackendBdc.cpp, :91-276 of
// the reference); SetAsmBlocks (not a FEMuS member) hands over other dof lists instead, e.g. fh_mesh_vertex_patches
class LinearEquationSolverHipAsm : public LinearEquationSolverHip {
 public:
  LinearEquationSolverHipAsm(const unsigned& igrid, Solution* other_solution) : LinearEquationSolverHip(igrid, other_solution) {}
  using LinearEquationSolver::SetElementBlockNumber;
  void SetElementBlockNumber(const unsigned& n) override { _elementBlockNumber = n; }
  void SetNumberOfSchurVariables(const unsigned short& n) override { _NSchurVar = n; }
  void SetAsmBlocks(const std::vector<int>& ptr, const std::vector<int>& dofs, int exact_first = 0) {      // not a FEMuS member: overrides BuildASMIndex
    _blockPtr = ptr; _blockDofs = dofs; _blockExactCount = exact_first; _blocksGiven = true;
  }
  void SetAsmExactInColourOrder(bool on) { _exactColoured = on; }                // not a FEMuS member: see smoother_id
  void BuildASMIndex(const std::vector<unsigned>& variable_to_be_solved);       // petsc_asm/LinearEquationSolverPetscAsm.cpp:91-276

 protected:
  // SetPreconditionerFineGrids(ILU_PRECOND) -- what the Navier-Stokes applications set -- gives PCASM as the reference configures it: basic /
  // multiplicative over the blocks in index order, one ILU(0) application per block -- an EXACT sub-solve on the blocks of solid / porous elements,
  // which come first (`_blockTypeRange[1]`, LinearEquationSolverPetscAsm.cpp:298-322) -- (FH_SMOOTH_ASM).  Any other preconditioner type, or
  // SetAsmExactInColourOrder(true) (not a FEMuS member), gives this library's variant: exact block inverses, damped, colour order (FH_SMOOTH_VANKA)
  int smoother_id() const override { return (_preconditioner_type == ILU_PRECOND && !_exactColoured) ? FH_SMOOTH_ASM : FH_SMOOTH_VANKA; }
  void attach_smoother_data(fh_mg_t mg, int level, const std::vector<unsigned>& variable_to_be_solved) override;

 private:
  bool _blocksGiven = false, _exactColoured = false;
  unsigned _elementBlockNumber = 1;
  unsigned short _NSchurVar = 1;
  std::vector<int> _blockPtr, _blockDofs;
  int _blockExactCount = 0;      // leading blocks with the exact sub-solve (solid + porous element blocks)
};

}  // namespace femus
