// Mirror of FEMuS's abstract vector interface for the hot path (names, argument meaning and error behaviour of
// src/03_algebra/00_vectors/NumericVector.hpp:51-353).  Written from the interface description, not copied: only the
// members the assembly / multigrid path calls are declared; everything aborts on error like the reference
// (CHKERRABORT / abort()).  A FEMuS maintainer keeps the original header and only adds the HipVector backend
// (see INTEGRATION.md); this mirror lets the adapters and their tests build without the FEMuS tree.
#pragma once
#include <memory>
#include <vector>

namespace femus {

enum ParallelType { AUTOMATIC = 0, SERIAL, PARALLEL, GHOSTED, INVALID_PARALLELIZATION };   // ParallelTypeEnum.hpp
enum SolverPackage { PETSC_SOLVERS = 0, HIP_SOLVERS = 7 };                                  // SolverPackageEnum.hpp (+ new id)

class SparseMatrix;

class NumericVector {
 public:
  virtual ~NumericVector() {}
  static std::unique_ptr<NumericVector> build(const SolverPackage solver_package = HIP_SOLVERS);   // NumericVector.cpp:35-56
  virtual std::unique_ptr<NumericVector> clone() const = 0;
  virtual void clear() = 0;
  virtual void close() = 0;                                                    // :96 (ghost refresh point)
  virtual void init(const int N, const int n_local, const bool fast = false, const ParallelType type = AUTOMATIC) = 0;      // :107
  virtual void init(const int N, const bool fast = false, const ParallelType type = AUTOMATIC) = 0;                         // :113
  virtual void init(const int N, const int n_local, const std::vector<int>& ghost, const bool fast = false,
                    const ParallelType type = AUTOMATIC) = 0;                                                                // :120
  virtual void init(const NumericVector& other, const bool fast = false) = 0;                                                // :129
  virtual void set(const int i, const double value) = 0;                       // :146
  virtual void add(const int i, const double value) = 0;                       // :148
  virtual void zero() = 0;                                                     // :151
  virtual NumericVector& operator=(const double s) = 0;                        // :153
  virtual NumericVector& operator=(const NumericVector& V) = 0;                // :155
  virtual NumericVector& operator=(const std::vector<double>& v) = 0;          // :157
  virtual double min() const = 0;
  virtual double max() const = 0;
  virtual double sum() const = 0;
  virtual double l1_norm() const = 0;
  virtual double l2_norm() const = 0;
  virtual double linfty_norm() const = 0;
  virtual int size() const = 0;
  virtual int local_size() const = 0;
  virtual int first_local_index() const = 0;
  virtual int last_local_index() const = 0;
  virtual double operator()(const int i) const = 0;                            // :224 (owned or ghost, host-synchronous)
  virtual void get(const std::vector<int>& index, std::vector<double>& values) const = 0;   // :236
  virtual NumericVector& operator+=(const NumericVector& V) = 0;
  virtual NumericVector& operator-=(const NumericVector& V) = 0;
  virtual void add(const double s) = 0;
  virtual void add(const NumericVector& V) = 0;
  virtual void add(const double a, const NumericVector& v) = 0;                // :262
  virtual void add_vector_blocked(const std::vector<double>& v, const std::vector<int>& dof_indices) = 0;        // :265
  virtual void add_vector_blocked(const std::vector<double>& v, const std::vector<unsigned>& dof_indices) = 0;   // :268
  virtual void insert_vector_blocked(const std::vector<double>& v, const std::vector<int>& dof_indices) = 0;     // :271
  virtual void add_vector(const NumericVector& v, const SparseMatrix& A) = 0;  // :281  y += A v
  virtual void resid(const NumericVector& rhs, const NumericVector& v, const SparseMatrix& A) = 0;   // :282  r = rhs - A v
  virtual void matrix_mult(const NumericVector& v, const SparseMatrix& A) = 0; // :283  y = A v
  virtual void matrix_mult_transpose(const NumericVector& v, const SparseMatrix& A) = 0;   // :284  y = A^T v
  virtual void scale(const double factor) = 0;
  virtual void abs() = 0;
  virtual double dot(const NumericVector&) const = 0;
  virtual void localize(std::vector<double>& v_local) const = 0;               // :308
  virtual void localize_to_all(std::vector<double>& v_local) const = 0;        // :323
  virtual void pointwise_mult(const NumericVector& vec1, const NumericVector& vec2) = 0;
  bool closed() const { return _is_closed; }
  bool initialized() const { return _is_initialized; }

 protected:
  bool _is_closed = false, _is_initialized = false;
};

}  // namespace femus
