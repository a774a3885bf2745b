// Mirror of FEMuS's abstract sparse-matrix interface for the hot path
// (src/03_algebra/01_matrices/SparseMatrix.hpp:48-282); see NumericVector.hpp for the ground rules.
#pragma once
#include <memory>
#include <vector>
#include "NumericVector.hpp"

namespace femus {

class SparseMatrix {
 public:
  virtual ~SparseMatrix() {}
  static std::unique_ptr<SparseMatrix> build(const SolverPackage solver_package = HIP_SOLVERS);   // SparseMatrix.cpp:42-63
  virtual void clear() = 0;                                                                        // :59
  // :65-74  m x n global, m_l x n_l local, per-row diagonal / off-diagonal counts (upper bounds, as for MatCreateAIJ)
  virtual void init(const int m, const int n, const int m_l, const int n_l, const std::vector<int>& n_nz,
                    const std::vector<int>& n_oz) = 0;
  virtual void set(const int i, const int j, const double value) = 0;          // :90
  virtual void add(const int i, const int j, const double value) = 0;          // :93
  virtual void zero() = 0;                                                     // :96 (keeps the pattern)
  virtual void close() const = 0;                                              // :102
  virtual double operator()(const int i, const int j) const = 0;               // :108
  virtual int MatGetRowM(const int i_val, int* cols = NULL, double* vals = NULL) = 0;   // :111
  virtual bool closed() const = 0;
  virtual int m() const = 0;
  virtual int n() const = 0;
  virtual int row_start() const = 0;
  virtual int row_stop() const = 0;
  virtual void insert_row(const int row, const int ncols, const std::vector<int>& cols, double* values) = 0;   // :162
  virtual void add_matrix_blocked(const std::vector<double>& mat_value, const std::vector<int>& rows,
                                  const std::vector<int>& cols) = 0;                                            // :165
  virtual void add_matrix_blocked(const std::vector<double>& mat_value, const std::vector<unsigned>& rows,
                                  const std::vector<unsigned>& cols) = 0;                                       // :169
  virtual void matrix_PtAP(const SparseMatrix& mat_P, const SparseMatrix& mat_A, const bool& reuse) = 0;        // :183
  virtual void matrix_ABC(const SparseMatrix& mat_A, const SparseMatrix& mat_B, const SparseMatrix& mat_C, const bool& reuse) = 0;    // :186
  virtual void matrix_RightMatMult(const SparseMatrix& mat_A) = 0;                                              // :189  this = this * A
  virtual void matrix_LeftMatMult(const SparseMatrix& mat_A) = 0;                                               // :191  this = A * this
  virtual void matrix_get_diagonal_values(const std::vector<int>& index, std::vector<double>& value) const = 0; // :195
  virtual double l1_norm() const = 0;
  virtual double linfty_norm() const = 0;
  virtual void get_diagonal(NumericVector& dest) const = 0;                    // :224
  virtual void get_transpose(SparseMatrix& dest) const = 0;                    // :227 (dest may be *this)
  virtual void mat_zero_rows(const std::vector<int>& index, const double& diagonal_value) const = 0;   // :229
  // non-virtual helpers of the reference (SparseMatrix.cpp:70-81): forward to NumericVector
  void vector_mult(NumericVector& dest, const NumericVector& arg) const {
    dest.zero();
    dest.add_vector(arg, *this);
  }
  void vector_mult_add(NumericVector& dest, const NumericVector& arg) const { dest.add_vector(arg, *this); }
};

}  // namespace femus
