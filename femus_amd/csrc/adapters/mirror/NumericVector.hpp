// Mirror of FEMuS's abstract vector interface (src/03_algebra/00_vectors/NumericVector.hpp:51-353): every pure virtual of the real
// class with its exact signature, plus the few non-pure members the hot path relies on.  Written from the interface, not copied;
// it exists so that the adapters and their test applications build and run without the FEMuS tree.  The SAME adapter sources are
// compiled against the real headers by tests/test_adapters_vs_reference_headers.py (HipVector : femus::NumericVector of the
// reference); a FEMuS maintainer keeps the original header.
#pragma once
#include <cstdlib>
#include <iostream>
#include <memory>
#include <set>
#include <vector>
#include "FemusEnums.hpp"

namespace femus {

class SparseMatrix;
class DenseVector;       // dense element vectors of the reference (DenseVector.hpp); only named by the interface here
class DenseSubVector;

class NumericVector {
 public:
  NumericVector(const ParallelType type = AUTOMATIC) : _is_closed(false), _is_initialized(false), _type(type) {}
  virtual ~NumericVector() { clear(); }
  static std::unique_ptr<NumericVector> build(const SolverPackage solver_package = LSOLVER);   // NumericVector.cpp:35-56
  virtual std::unique_ptr<NumericVector> clone() const = 0;                                      // :82
  virtual void clear() { _is_closed = false; _is_initialized = false; }                         // :90
  virtual void close() = 0;                                                                      // :96 (ghost refresh point)
  virtual void closeWithMinValues() = 0;                                                         // :97
  virtual void init(const int, const int, const bool = false, const ParallelType = AUTOMATIC) = 0;                           // :107
  virtual void init(const int, const bool = false, const ParallelType = AUTOMATIC) = 0;                                      // :113
  virtual void init(const int /*N*/, const int /*n_local*/, const std::vector<int>& /*ghost*/, const bool /*fast*/ = false,
                    const ParallelType = AUTOMATIC) = 0;                                                                      // :120
  virtual void init(const NumericVector& other, const bool fast = false) = 0;                                                // :129
  virtual void set(const int i, const double value) = 0;                       // :146
  virtual void add(const int i, const double value) = 0;                       // :148
  virtual void zero() = 0;                                                     // :151
  virtual NumericVector& operator=(const double s) = 0;                        // :153
  virtual NumericVector& operator=(const NumericVector& V) = 0;                // :155
  virtual NumericVector& operator=(const std::vector<double>& v) = 0;          // :157
  virtual void insert(const std::vector<double>& v, const std::vector<int>& dof_indices) = 0;   // :160
  virtual void insert(const NumericVector& V, const std::vector<int>& dof_indices) = 0;         // :163
  virtual void insert(const DenseVector& V, const std::vector<int>& dof_indices) = 0;           // :166
  virtual void insert(const DenseSubVector& V, const std::vector<int>& dof_indices) = 0;        // :169
  virtual bool initialized() const { return _is_initialized; }
  virtual bool closed() const { return _is_closed; }
  ParallelType type() const { return _type; }
  ParallelType& type() { return _type; }
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
  virtual double el(const int i) const { return (*this)(i); }
  virtual void get(const std::vector<int>& index, std::vector<double>& values) const {   // :236
    values.resize(index.size());
    for (size_t k = 0; k < index.size(); k++) values[k] = (*this)(index[k]);
  }
  virtual NumericVector& operator+=(const NumericVector& V) = 0;
  virtual NumericVector& operator-=(const NumericVector& V) = 0;
  NumericVector& operator*=(const double a) { this->scale(a); return *this; }
  NumericVector& operator/=(const double a) { this->scale(1. / a); return *this; }
  virtual void add(const double s) = 0;
  virtual void add(const NumericVector& V) = 0;
  virtual void add(const double a, const NumericVector& v) = 0;                // :262
  virtual void add_vector_blocked(const std::vector<double>& v, const std::vector<int>& dof_indices) = 0;        // :265
  virtual void add_vector_blocked(const std::vector<double>& v, const std::vector<unsigned>& dof_indices) = 0;   // :268
  virtual void insert_vector_blocked(const std::vector<double>& v, const std::vector<int>& dof_indices) = 0;     // :271
  virtual void add_vector(const std::vector<double>& v, const std::vector<int>& dof_indices) = 0;                // :275
  virtual void add_vector(const NumericVector& V, const std::vector<int>& dof_indices) = 0;                      // :278
  virtual void add_vector(const NumericVector& v, const SparseMatrix& A) = 0;  // :281  y += A v
  virtual void resid(const NumericVector& rhs, const NumericVector& v, const SparseMatrix& A) = 0;   // :282  r = rhs - A v
  virtual void matrix_mult(const NumericVector& v, const SparseMatrix& A) = 0; // :283  y = A v
  virtual void matrix_mult_transpose(const NumericVector& v, const SparseMatrix& A) = 0;   // :284  y = A^T v
  virtual void add_vector(const DenseVector& V, const std::vector<unsigned int>& dof_indices) = 0;               // :290
  virtual void scale(const double factor) = 0;
  virtual void abs() = 0;
  virtual double dot(const NumericVector&) const = 0;
  virtual void swap(NumericVector& v) {                                        // :301 (NumericVector.cpp)
    std::swap(_is_closed, v._is_closed);
    std::swap(_is_initialized, v._is_initialized);
    std::swap(_type, v._type);
  }
  virtual void localize(std::vector<double>& v_local) const = 0;               // :308
  virtual void localize(NumericVector& v_local) const = 0;                     // :310
  virtual void localize(NumericVector& v_local, const std::vector<int>& send_list) const = 0;                    // :313
  virtual void localize(const int first_local_idx, const int last_local_idx, const std::vector<int>& send_list) = 0;   // :316
  virtual void localize_to_one(std::vector<double>& v_local, const int proc_id = 0) const = 0;                   // :320
  virtual void localize_to_all(std::vector<double>& v_local) const = 0;        // :323
  virtual void pointwise_mult(const NumericVector& vec1, const NumericVector& vec2) = 0;                         // :328
  virtual void BinaryPrint(const char* fileName) {                              // :345 (SaveSolution writes one such file per variable)
    std::cout << "BinaryPrint is not available for this vector type\n";
    abort();
  }
  virtual void BinaryLoad(const char* fileName) {                               // :350
    std::cout << "BinaryLoad is not available for this vector type\n";
    abort();
  }

 protected:
  bool _is_closed, _is_initialized;
  ParallelType _type;
};

}  // namespace femus
