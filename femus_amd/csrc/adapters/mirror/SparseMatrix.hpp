// Mirror of FEMuS's abstract sparse-matrix interface (src/03_algebra/01_matrices/SparseMatrix.hpp:48-282): every pure virtual of the
// real class with its exact signature; see NumericVector.hpp in this directory for the ground rules.
#pragma once
#include <iostream>
#include <memory>
#include <vector>
#include "FemusEnums.hpp"
#include "NumericVector.hpp"

namespace femus {

class DenseMatrix;       // named by the interface only (DenseMatrix.hpp, Graph.hpp of the reference)
class Graph;

class SparseMatrix {
 public:
  SparseMatrix() : _is_initialized(false) {}
  virtual ~SparseMatrix() {}
  virtual void clear() = 0;                                                                        // :59
  static std::unique_ptr<SparseMatrix> build(const SolverPackage solver_package = LSOLVER);        // SparseMatrix.cpp:42-63
  virtual void init(const int m, const int n, const int m_l, const int n_l, const int /*nnz*/ = 30, const int /*noz*/ = 10) {   // :65
    _m = m; _n = n; _m_l = m_l; _n_l = n_l;
  }
  virtual void init(const int m, const int n) { _m = m; _n = n; }              // :76
  virtual void init() {}                                                       // :84
  // :73-74  m x n global, m_l x n_l local, per-row diagonal / off-diagonal counts (upper bounds, as for MatCreateAIJ)
  virtual void init(const int m, const int n, const int m_l, const int n_l, const std::vector<int>& n_nz, const std::vector<int>& n_oz) = 0;
  virtual void init(const int nr, const int nc, const std::vector<SparseMatrix*>& P) = 0;          // :81 block matrix of nr x nc matrices
  virtual void set(const int i, const int j, const double value) = 0;          // :90
  virtual void add(const int i, const int j, const double value) = 0;          // :93
  virtual void zero() = 0;                                                     // :96 (keeps the pattern)
  virtual void close() const = 0;                                              // :102
  virtual double operator()(const int i, const int j) const = 0;               // :108
  virtual int MatGetRowM(const int i_val, int* cols = NULL, double* vals = NULL) = 0;   // :111
  virtual void RemoveZeroEntries(double& tolerance) = 0;                       // :113
  virtual bool initialized() const { return _is_initialized; }
  virtual bool closed() const = 0;
  virtual void update_sparsity_pattern_old(const Graph&) = 0;                  // :130
  virtual void update_sparsity_pattern(const Graph&) = 0;                      // :133
  virtual void update_sparsity_pattern(int m, int n, int m_l, int n_l, const std::vector<int> n_oz, const std::vector<int> n_nz) = 0;   // :136
  virtual int m() const = 0;
  virtual int n() const = 0;
  virtual int row_start() const = 0;
  virtual int row_stop() const = 0;
  virtual void add_matrix(const DenseMatrix& dm, const std::vector<unsigned int>& rows, const std::vector<unsigned int>& cols) = 0;   // :154
  virtual void add_matrix(const DenseMatrix& dm, const std::vector<unsigned int>& dof_indices) = 0;                                    // :159
  virtual void insert_row(const int row, const int ncols, const std::vector<int>& cols, double* values) = 0;   // :162
  virtual void add_matrix_blocked(const std::vector<double>& mat_value, const std::vector<int>& rows, const std::vector<int>& cols) = 0;            // :165
  virtual void add_matrix_blocked(const std::vector<double>& mat_value, const std::vector<unsigned>& rows, const std::vector<unsigned>& cols) = 0;  // :169
  virtual void matrix_set_off_diagonal_values_blocked(const std::vector<int>& index_rows, const std::vector<int>& index_cols, const double& value) = 0;               // :174
  virtual void matrix_set_off_diagonal_values_blocked(const std::vector<int>& index_rows, const std::vector<int>& index_cols, const std::vector<double>& value) = 0;  // :177
  virtual void matrix_add(const double a_in, SparseMatrix& X_in, const char pattern[]) = 0;                     // :180
  virtual void matrix_PtAP(const SparseMatrix& mat_P, const SparseMatrix& mat_A, const bool& reuse) = 0;        // :183
  virtual void matrix_ABC(const SparseMatrix& mat_A, const SparseMatrix& mat_B, const SparseMatrix& mat_C, const bool& reuse) = 0;    // :186
  virtual void matrix_RightMatMult(const SparseMatrix& mat_A) = 0;                                              // :189  this = this * A
  virtual void matrix_LeftMatMult(const SparseMatrix& mat_A) = 0;                                               // :191  this = A * this
  virtual void matrix_get_diagonal_values(const std::vector<int>& index, std::vector<double>& value) const = 0; // :195
  virtual void matrix_set_diagonal_values(NumericVector& D) = 0;                                                // :198
  virtual void matrix_set_diagonal_values(const std::vector<int>& index, const double& value) = 0;              // :201
  virtual void matrix_set_diagonal_values(const std::vector<int>& index, const std::vector<double>& value) = 0; // :204
  virtual void add(const double /*c*/, SparseMatrix& /*B*/) = 0;               // :207  A += c B
  virtual double l1_norm() const = 0;
  virtual double linfty_norm() const = 0;
  // non-virtual helpers of the reference (SparseMatrix.cpp:70-81): forward to NumericVector
  void vector_mult(NumericVector& dest, const NumericVector& arg) const {
    dest.zero();
    dest.add_vector(arg, *this);
  }
  void vector_mult_add(NumericVector& dest, const NumericVector& arg) const { dest.add_vector(arg, *this); }
  virtual void get_diagonal(NumericVector& dest) const = 0;                    // :224
  virtual void get_transpose(SparseMatrix& dest) const = 0;                    // :227 (dest may be *this)
  virtual void mat_zero_rows(const std::vector<int>& index, const double& diagonal_value) const = 0;   // :229
  virtual void print(std::ostream& os = std::cout) const { print_personal(os); }
  virtual void print_personal(std::ostream& os = std::cout) const = 0;         // :245
  virtual void print_hdf5(const std::string name = "NULL") const = 0;          // :248

 protected:
  int _m = 0, _n = 0, _m_l = 0, _n_l = 0;
  bool _is_initialized;
};

}  // namespace femus
