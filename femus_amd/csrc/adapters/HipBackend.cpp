#include "HipBackend.hpp"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <thread>

namespace femus {

void hip_check(int rc, const char* what) {
  if (rc != 0) {   // the reference's convention: CHKERRABORT / std::cout << ...; abort();
    std::cout << "femus_hip error in " << what << ": " << fh_last_error() << std::endl;
    abort();
  }
}

fh_ctx_t hip_context() {
  static fh_ctx_t ctx = nullptr;
  if (!ctx) {
    const char* dev = getenv("FEMUS_HIP_DEVICE");
    hip_check(fh_init(dev ? atoi(dev) : 0, &ctx), "fh_init");
  }
  return ctx;
}

static const HipVector& hv(const NumericVector& v) { return static_cast<const HipVector&>(v); }   // one backend per process,
static const HipMatrix& hm(const SparseMatrix& A) { return static_cast<const HipMatrix&>(A); }     // as PetscMatrix.cpp:735-739

// =============================== HipVector ===============================
void HipVector::clear() {
  if (_v) fh_vec_destroy(_v);
  _v = nullptr;
  _halo = nullptr;
  _pending = false;
  _is_initialized = _is_closed = false;
}
std::unique_ptr<NumericVector> HipVector::clone() const {
  HipVector* c = new HipVector();
  c->init(*this, true);
  c->operator=(static_cast<const NumericVector&>(*this));
  return std::unique_ptr<NumericVector>(c);
}
void HipVector::init(const int N, const int n_local, const bool, const ParallelType ptype) {
  clear();
  _type = ptype;
  hip_check(fh_vec_create(hip_context(), N, n_local, 0, nullptr, 0, &_v), "HipVector::init");
  _n_global = N;
  _n_local = n_local;
  _first = 0;
  _is_initialized = true;
}
void HipVector::init(const int N, const int n_local, const std::vector<int>& ghost, const bool, const ParallelType ptype) {
  clear();
  _type = ptype;
  hip_check(fh_vec_create(hip_context(), N, n_local, 0, ghost.data(), (int)ghost.size(), &_v), "HipVector::init(ghosted)");
  _n_global = N;
  _n_local = n_local;
  _first = 0;
  _is_initialized = true;
}
void HipVector::init(const NumericVector& other, const bool) {
  clear();
  hip_check(fh_vec_duplicate(hv(other).handle(), &_v), "HipVector::init(other)");
  fh_vec_size(_v, &_n_global, &_n_local, &_first, nullptr);
  _type = hv(other).type();
  _halo = hv(other).halo();          // same layout, same ghosts: same plan (init(other) clones the layout incl. ghosts, PetscVector.hpp:572-592)
  _is_initialized = true;
}
fh_vec_t HipVector::rv() const {
  if (_pending) {
    _pending = false;
    hip_check(fh_vec_flush(_v), "HipVector: flush of the staged adds");
  }
  return _v;
}
// ownership range of this rank: the owned sizes of the ranks before it (VecGetOwnershipRange; PetscVector.hpp:683-707)
void HipVector::attach_halo(fh_halo_t halo) {
  _halo = halo;
  if (!halo) return;
  int rank = 0, nranks = 1;
  hip_check(fh_halo_rank(halo, &rank, &nranks), "attach_halo");
  std::vector<double> sizes(nranks, 0.);
  sizes[rank] = _n_local;
  hip_check(fh_halo_allreduce_sum(halo, sizes.data(), nranks), "attach_halo: owned sizes");
  double first = 0., total = 0.;
  for (int r = 0; r < nranks; r++) {
    if (r < rank) first += sizes[r];
    total += sizes[r];
  }
  if (nranks > 1 && (int)total != _n_global) {      // (a one-rank plan may be a self exchange standing in for absent neighbours)
    std::cout << "HipVector::attach_halo: the owned sizes of the ranks add up to " << (long)total << ", the vector has " << _n_global << std::endl;
    abort();
  }
  _first = (int)first;
  hip_check(fh_vec_set_first(rv(), _first), "attach_halo: ownership offset");
}
void HipVector::not_served(const char* what) {
  std::cout << "HipVector::" << what << " is not served by the HIP backend" << std::endl;
  abort();
}
void HipVector::close() {
  if (_v) rv();          // VecAssemblyBegin/End: the staged adds of this rank
  if (_halo && _v) {
    // ... and the adds other ranks staged for entries this rank owns (ADD_VALUES to an off-process index is shipped to the owner and summed there,
    // PetscVector.cpp:131-153): one flag over the ranks decides whether the reverse exchange runs at all
    int mine = 0;
    hip_check(fh_vec_ghost_adds_pending(rv(), &mine), "HipVector::close (ghost adds)");
    if (all_sum((double)mine) > 0.) hip_check(fh_halo_reverse_add(_halo, rv()), "HipVector::close (ghost adds to their owners)");
    hip_check(fh_halo_update(_halo, rv()), "HipVector::close (ghost refresh)");   // VecGhostUpdateBegin/End, PetscVector.hpp:604-610
  }
  _is_closed = true;
}
double HipVector::all_sum(double local) const {      // VecDot / VecNorm over the ranks (Parallel.hpp:351-377)
  if (_halo) hip_check(fh_halo_allreduce_sum(_halo, &local, 1), "HipVector: all-reduce");
  return local;
}
double HipVector::all_extreme(double local, bool want_max) const {   // VecMin / VecMax / NORM_INFINITY are global (PetscVector.cpp)
  if (!_halo) return local;
  int rank = 0, nranks = 1;
  hip_check(fh_halo_rank(_halo, &rank, &nranks), "HipVector: reduce");
  std::vector<double> all(nranks, 0.);
  all[rank] = local;
  hip_check(fh_halo_allreduce_sum(_halo, all.data(), nranks), "HipVector: reduce");
  return want_max ? *std::max_element(all.begin(), all.end()) : *std::min_element(all.begin(), all.end());
}
void HipVector::insert(const NumericVector& V, const std::vector<int>& dof) {
  std::vector<double> v;
  V.localize(v);
  if (v.size() != dof.size()) { std::cout << "HipVector::insert: size mismatch" << std::endl; abort(); }
  insert_vector_blocked(v, dof);
}
void HipVector::add_vector(const NumericVector& V, const std::vector<int>& dof) {
  std::vector<double> v;
  V.localize(v);
  if (v.size() != dof.size()) { std::cout << "HipVector::add_vector: size mismatch" << std::endl; abort(); }
  add_vector_blocked(v, dof);
}
void HipVector::swap(NumericVector& other) {
  HipVector& o = static_cast<HipVector&>(other);
  NumericVector::swap(other);
  std::swap(_v, o._v);
  std::swap(_halo, o._halo);
  std::swap(_pending, o._pending);
  std::swap(_n_global, o._n_global);
  std::swap(_n_local, o._n_local);
  std::swap(_first, o._first);
}
void HipVector::localize(NumericVector& v_local) const {     // PetscVector.cpp: copy of the whole vector into v_local (one rank: the owned part)
  static_cast<HipVector&>(v_local) = static_cast<const NumericVector&>(*this);
}
void HipVector::localize(NumericVector& v_local, const std::vector<int>& send_list) const {
  std::vector<double> vals;
  get(send_list, vals);
  v_local.insert(vals, send_list);
  v_local.close();
}
void HipVector::localize(const int first_local_idx, const int last_local_idx, const std::vector<int>&) {
  if (first_local_idx != _first || last_local_idx + 1 != _first + _n_local) not_served("localize(first, last, send_list) with a new layout");
  close();
}
void HipVector::localize_to_one(std::vector<double>& v_local, const int) const { localize_to_all(v_local); }
void HipVector::localize_to_all(std::vector<double>& v_local) const {
  // every rank gets the whole vector (PetscVector.cpp: VecScatterCreateToAll): owned part here, summed into place over the ranks
  std::vector<double> own;
  localize(own);
  v_local.assign(_n_global, 0.);
  std::copy(own.begin(), own.end(), v_local.begin() + _first);
  if (_halo && _n_global != _n_local) hip_check(fh_halo_allreduce_sum(_halo, v_local.data(), _n_global), "localize_to_all");
}
void HipVector::set(const int i, const double value) { hip_check(fh_vec_set_values(rv(), 1, &i, &value), "HipVector::set"); _is_closed = false; }
void HipVector::add(const int i, const double value) {
  hip_check(fh_vec_stage_values(_v, 1, &i, &value), "HipVector::add");
  _pending = true;
  _is_closed = false;
}
void HipVector::zero() { hip_check(fh_vec_zero(rv()), "HipVector::zero"); }
NumericVector& HipVector::operator=(const double s) { hip_check(fh_vec_fill(rv(), s), "HipVector::operator=(double)"); return *this; }
NumericVector& HipVector::operator=(const NumericVector& V) { hip_check(fh_vec_copy(rv(), hv(V).handle()), "HipVector::operator=(vector)"); return *this; }
NumericVector& HipVector::operator=(const std::vector<double>& v) {
  if ((int)v.size() != _n_local) { std::cout << "HipVector::operator=(std::vector): size mismatch" << std::endl; abort(); }
  hip_check(fh_vec_upload(rv(), v.data()), "HipVector::operator=(std::vector)");
  return *this;
}
double HipVector::min() const { double r; hip_check(fh_vec_reduce(rv(), 1, &r), "min"); return all_extreme(r, false); }
double HipVector::max() const { double r; hip_check(fh_vec_reduce(rv(), 2, &r), "max"); return all_extreme(r, true); }
double HipVector::sum() const { double r; hip_check(fh_vec_reduce(rv(), 0, &r), "sum"); return all_sum(r); }
double HipVector::l1_norm() const { double r; hip_check(fh_vec_norm(rv(), 1, &r), "l1_norm"); return all_sum(r); }
double HipVector::l2_norm() const { double r; hip_check(fh_vec_norm(rv(), 2, &r), "l2_norm"); return _halo ? sqrt(all_sum(r * r)) : r; }
double HipVector::linfty_norm() const { double r; hip_check(fh_vec_norm(rv(), 0, &r), "linfty_norm"); return all_extreme(r, true); }
double HipVector::operator()(const int i) const { double r; hip_check(fh_vec_get_values(rv(), 1, &i, &r), "operator()"); return r; }
void HipVector::get(const std::vector<int>& index, std::vector<double>& values) const {
  values.resize(index.size());
  hip_check(fh_vec_get_values(rv(), (int)index.size(), index.data(), values.data()), "HipVector::get");
}
void HipVector::add(const double s) { hip_check(fh_vec_shift(rv(), s), "HipVector::add(s)"); }
void HipVector::add(const double a, const NumericVector& v) { hip_check(fh_vec_axpy(rv(), a, hv(v).handle()), "HipVector::add(a,v)"); }
void HipVector::add_vector_blocked(const std::vector<double>& v, const std::vector<int>& dof) {
  // staged in the pinned ring (VecSetValues into the stash); applied in call order by the next member that needs the values
  hip_check(fh_vec_stage_values(_v, (int)dof.size(), dof.data(), v.data()), "add_vector_blocked");
  _pending = true;
  _is_closed = false;
}
void HipVector::add_vector_blocked(const std::vector<double>& v, const std::vector<unsigned>& dof) {
  std::vector<int> d(dof.begin(), dof.end());
  add_vector_blocked(v, d);
}
void HipVector::insert_vector_blocked(const std::vector<double>& v, const std::vector<int>& dof) {
  hip_check(fh_vec_set_values(rv(), (int)dof.size(), dof.data(), v.data()), "insert_vector_blocked");
  _is_closed = false;
}
// the operand's exchange plan (if any) refreshes its ghosts inside the product, overlapped with the rows that need none: MatMult on
// an MPIAIJ matrix (PetscVector.cpp:182-247)
void HipVector::add_vector(const NumericVector& v, const SparseMatrix& A) {
  hip_check(fh_spmv_ghosted(hm(A).handle(), hv(v).halo(), hv(v).handle(), rv(), 1, nullptr, nullptr, 0.), "add_vector(v,A)");
}
void HipVector::resid(const NumericVector& rhs, const NumericVector& v, const SparseMatrix& A) {
  hip_check(fh_spmv_ghosted(hm(A).handle(), hv(v).halo(), hv(v).handle(), rv(), 2, hv(rhs).handle(), nullptr, 0.), "resid");
}
void HipVector::matrix_mult(const NumericVector& v, const SparseMatrix& A) {
  hip_check(fh_spmv_ghosted(hm(A).handle(), hv(v).halo(), hv(v).handle(), rv(), 0, nullptr, nullptr, 0.), "matrix_mult");
}
void HipVector::matrix_mult_transpose(const NumericVector& v, const SparseMatrix& A) {
  hip_check(fh_spmv_transpose(hm(A).handle(), hv(v).handle(), rv()), "matrix_mult_transpose");
}
void HipVector::scale(const double f) { hip_check(fh_vec_scale(rv(), f), "scale"); }
void HipVector::abs() { hip_check(fh_vec_abs(rv()), "abs"); }
double HipVector::dot(const NumericVector& o) const { double r; hip_check(fh_vec_dot(rv(), hv(o).handle(), &r), "dot"); return all_sum(r); }
void HipVector::localize(std::vector<double>& out) const {
  out.resize(_n_local);
  hip_check(fh_vec_download(rv(), out.data()), "localize");
}
void HipVector::BinaryPrint(const char* fileName) { hip_check(fh_vec_binary_print(rv(), fileName), "BinaryPrint"); }
void HipVector::BinaryLoad(const char* fileName) { hip_check(fh_vec_binary_load(rv(), fileName), "BinaryLoad"); }
void HipVector::pointwise_mult(const NumericVector& a, const NumericVector& b) {
  hip_check(fh_vec_pointwise_mult(rv(), hv(a).handle(), hv(b).handle()), "pointwise_mult");
}

// =============================== HipMatrix ===============================
void HipMatrix::clear() {
  if (_A) fh_mat_destroy(_A);
  _A = nullptr;
  _stage.clear();
  _logHdr.clear();
  _logIdx.clear();
  _logVal.clear();
  _pending = false;
  _closed = false;
}
void HipMatrix::init(const int m, const int n, const int, const int, const std::vector<int>&, const std::vector<int>&) {
  clear();
  _m = m;
  _n = n;
  _stage.assign(m, std::map<int, double>());   // the nnz counts are upper bounds only; the pattern grows until close()
}
void HipMatrix::not_served(const char* what) {
  std::cout << "HipMatrix::" << what << " is not served by the HIP backend" << std::endl;
  abort();
}
void HipMatrix::to_host(std::vector<int>& rp, std::vector<int>& col, std::vector<double>& val) const {
  close();
  int m = 0, n = 0, nnz = 0;
  fh_mat_size(_A, &m, &n, &nnz);
  rp.resize(m + 1);
  col.resize(nnz);
  val.resize(nnz);
  hip_check(fh_mat_get_pattern(_A, rp.data(), col.data()), "HipMatrix: pattern");
  hip_check(fh_mat_get_values_csr(_A, val.data()), "HipMatrix: values");
}
// block matrix of nr x nc matrices (PetscMatrix.cpp: MatCreateNest + conversion): blocks may be NULL; merged into one CSR
void HipMatrix::init(const int nr, const int nc, const std::vector<SparseMatrix*>& P) {
  if ((int)P.size() != nr * nc) { std::cout << "HipMatrix::init(nr, nc, blocks): wrong number of blocks" << std::endl; abort(); }
  std::vector<int> roff(nr + 1, 0), coff(nc + 1, 0);
  for (int i = 0; i < nr; i++)
    for (int j = 0; j < nc; j++)
      if (P[i * nc + j]) {
        roff[i + 1] = P[i * nc + j]->m();
        coff[j + 1] = P[i * nc + j]->n();
      }
  for (int i = 0; i < nr; i++) roff[i + 1] += roff[i];
  for (int j = 0; j < nc; j++) coff[j + 1] += coff[j];
  std::vector<int> rp(roff[nr] + 1, 0), col;
  std::vector<double> val;
  std::vector<std::vector<int>> brp(nr * nc), bcol(nr * nc);
  std::vector<std::vector<double>> bval(nr * nc);
  for (int b = 0; b < nr * nc; b++)
    if (P[b]) static_cast<const HipMatrix*>(P[b])->to_host(brp[b], bcol[b], bval[b]);
  for (int i = 0; i < nr; i++)
    for (int r = 0; r < roff[i + 1] - roff[i]; r++) {
      for (int j = 0; j < nc; j++) {
        const int b = i * nc + j;
        if (!P[b]) continue;
        for (int k = brp[b][r]; k < brp[b][r + 1]; k++) {
          col.push_back(coff[j] + bcol[b][k]);
          val.push_back(bval[b][k]);
        }
      }
      rp[roff[i] + r + 1] = (int)col.size();
    }
  clear();
  _m = roff[nr];
  _n = coff[nc];
  hip_check(fh_mat_create_csr(hip_context(), _m, _n, rp.data(), col.data(), val.data(), &_A), "HipMatrix::init(blocks)");
  _closed = true;
}
void HipMatrix::RemoveZeroEntries(double& tolerance) {       // PetscMatrix.cpp: entries with |value| <= tolerance leave the pattern
  std::vector<int> rp, col, nrp, ncol;
  std::vector<double> val, nval;
  to_host(rp, col, val);
  nrp.assign(_m + 1, 0);
  for (int i = 0; i < _m; i++) {
    for (int k = rp[i]; k < rp[i + 1]; k++)
      if (fabs(val[k]) > tolerance) {
        ncol.push_back(col[k]);
        nval.push_back(val[k]);
      }
    nrp[i + 1] = (int)ncol.size();
  }
  const int m = _m, n = _n;
  clear();
  _m = m;
  _n = n;
  hip_check(fh_mat_create_csr(hip_context(), m, n, nrp.data(), ncol.data(), nval.data(), &_A), "RemoveZeroEntries");
  _closed = true;
}
// this += a X (MatAXPY, PetscMatrix.cpp): the union pattern on the host, values summed
void HipMatrix::matrix_add(const double a, SparseMatrix& X, const char[]) {
  std::vector<int> rp, col, xrp, xcol, nrp, ncol;
  std::vector<double> val, xval, nval;
  to_host(rp, col, val);
  static_cast<HipMatrix&>(X).to_host(xrp, xcol, xval);
  if (X.m() != _m || X.n() != _n) { std::cout << "HipMatrix::matrix_add: shapes differ" << std::endl; abort(); }
  nrp.assign(_m + 1, 0);
  for (int i = 0; i < _m; i++) {
    int p = rp[i], q = xrp[i];
    while (p < rp[i + 1] || q < xrp[i + 1]) {
      const int cp = p < rp[i + 1] ? col[p] : _n, cq = q < xrp[i + 1] ? xcol[q] : _n;
      const int c = std::min(cp, cq);
      double v = 0.;
      if (cp == c) v += val[p++];
      if (cq == c) v += a * xval[q++];
      ncol.push_back(c);
      nval.push_back(v);
    }
    nrp[i + 1] = (int)ncol.size();
  }
  const int m = _m, n = _n;
  clear();
  _m = m;
  _n = n;
  hip_check(fh_mat_create_csr(hip_context(), m, n, nrp.data(), ncol.data(), nval.data(), &_A), "matrix_add");
  _closed = true;
}
void HipMatrix::matrix_set_off_diagonal_values_blocked(const std::vector<int>& rows, const std::vector<int>& cols, const double& value) {
  std::vector<double> v(rows.size() * cols.size(), value);
  matrix_set_off_diagonal_values_blocked(rows, cols, v);
}
void HipMatrix::matrix_set_off_diagonal_values_blocked(const std::vector<int>& rows, const std::vector<int>& cols, const std::vector<double>& value) {
  // PetscMatrix.cpp: MatSetValuesBlocked(..., INSERT_VALUES) on the block rows x cols
  close();
  for (size_t i = 0; i < rows.size(); i++) {
    std::vector<double> v(value.begin() + i * cols.size(), value.begin() + (i + 1) * cols.size());
    hip_check(fh_mat_insert_row(_A, rows[i], (int)cols.size(), cols.data(), v.data()), "matrix_set_off_diagonal_values_blocked");
  }
}
void HipMatrix::matrix_set_diagonal_values(NumericVector& D) {
  std::vector<double> d;
  D.localize(d);
  std::vector<int> idx(d.size());
  for (size_t i = 0; i < d.size(); i++) idx[i] = (int)i;
  matrix_set_diagonal_values(idx, d);
}
void HipMatrix::matrix_set_diagonal_values(const std::vector<int>& index, const double& value) {
  std::vector<double> v(index.size(), value);
  matrix_set_diagonal_values(index, v);
}
void HipMatrix::matrix_set_diagonal_values(const std::vector<int>& index, const std::vector<double>& value) {
  close();
  for (size_t k = 0; k < index.size(); k++) hip_check(fh_mat_insert_row(_A, index[k], 1, &index[k], &value[k]), "matrix_set_diagonal_values");
}
void HipMatrix::print_personal(std::ostream& os) const {
  std::vector<int> rp, col;
  std::vector<double> val;
  to_host(rp, col, val);
  for (int i = 0; i < _m; i++)
    for (int k = rp[i]; k < rp[i + 1]; k++) os << i << " " << col[k] << " " << val[k] << "\n";
}
void HipMatrix::init_pattern(const int m, const int n, const std::vector<int>& rowptr, const std::vector<int>& col) {
  clear();
  _m = m;
  _n = n;
  hip_check(fh_mat_create_csr(hip_context(), m, n, rowptr.data(), col.data(), nullptr, &_A), "HipMatrix::init_pattern");
  _closed = true;
}
void HipMatrix::adopt(fh_mat_t h) {
  clear();
  _A = h;
  fh_mat_size(_A, &_m, &_n, nullptr);
  _closed = true;
}
void HipMatrix::close() const {
  if (!_A && !(_stage.empty() && _logHdr.empty())) first_close();
  if (_A && _pending) {       // MatAssemblyBegin/End: the blocks staged since the last close
    _pending = false;
    hip_check(fh_mat_flush(_A), "HipMatrix::close (staged add_matrix_blocked calls)");
  }
  _closed = true;
}
// first close(): the union pattern of everything inserted and added so far (PETSc grows it inside MatSetValues), on all host cores
void HipMatrix::first_close() const {
  const size_t nblk = _logHdr.size() / 2;
  std::vector<int64_t> ptr(_m + 1, 0);
  {
    size_t io = 0;
    for (size_t b = 0; b < nblk; b++) {
      const int nr = _logHdr[2 * b], nc = _logHdr[2 * b + 1];
      for (int i = 0; i < nr; i++) {
        const int r = _logIdx[io + i];
        if (r < 0 || r >= _m) { std::cout << "HipMatrix::add_matrix_blocked: row " << r << " out of range" << std::endl; abort(); }
        ptr[r + 1] += nc;
      }
      io += (size_t)nr + nc;
    }
  }
  for (int i = 0; i < (int)_stage.size(); i++) ptr[i + 1] += (int64_t)_stage[i].size();
  for (int i = 0; i < _m; i++) ptr[i + 1] += ptr[i];
  std::vector<int> cand(ptr[_m]);
  {
    std::vector<int64_t> cur(ptr.begin(), ptr.end() - 1);
    size_t io = 0;
    for (size_t b = 0; b < nblk; b++) {
      const int nr = _logHdr[2 * b], nc = _logHdr[2 * b + 1];
      const int* cols = _logIdx.data() + io + nr;
      for (int i = 0; i < nr; i++) {
        const int r = _logIdx[io + i];
        std::copy(cols, cols + nc, cand.begin() + cur[r]);
        cur[r] += nc;
      }
      io += (size_t)nr + nc;
    }
    for (int i = 0; i < (int)_stage.size(); i++)
      for (auto& kv : _stage[i]) cand[cur[i]++] = kv.first;
  }
  // sort + unique per row, rows split over the host cores
  std::vector<int> len(_m, 0);
  const int nth = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  auto work = [&](int t) {
    const int r0 = (int)((int64_t)_m * t / nth), r1 = (int)((int64_t)_m * (t + 1) / nth);
    for (int r = r0; r < r1; r++) {
      int* b = cand.data() + ptr[r];
      int* e = cand.data() + ptr[r + 1];
      std::sort(b, e);
      len[r] = (int)(std::unique(b, e) - b);
    }
  };
  {
    std::vector<std::thread> th;
    for (int t = 1; t < nth; t++) th.emplace_back(work, t);
    work(0);
    for (auto& t : th) t.join();
  }
  std::vector<int> rp(_m + 1, 0);
  for (int r = 0; r < _m; r++) {
    if ((int64_t)rp[r] + len[r] > 2147483647LL) { std::cout << "HipMatrix::close: more than 2^31 non-zeros" << std::endl; abort(); }
    rp[r + 1] = rp[r] + len[r];
  }
  std::vector<int> col(rp[_m]);
  for (int r = 0; r < _m; r++) {
    if (len[r] && (cand[ptr[r]] < 0 || cand[ptr[r] + len[r] - 1] >= _n)) { std::cout << "HipMatrix: column out of range in row " << r << std::endl; abort(); }
    std::copy(cand.begin() + ptr[r], cand.begin() + ptr[r] + len[r], col.begin() + rp[r]);
  }
  std::vector<int>().swap(cand);
  std::vector<double> val(rp[_m], 0.);
  for (int i = 0; i < (int)_stage.size(); i++)       // inserted entries (set / insert_row before the first close)
    for (auto& kv : _stage[i]) val[std::lower_bound(col.begin() + rp[i], col.begin() + rp[i + 1], kv.first) - col.begin()] = kv.second;
  hip_check(fh_mat_create_csr(hip_context(), _m, _n, rp.data(), col.data(), val.data(), &_A), "HipMatrix::close");
  _stage.clear();
  _stage.shrink_to_fit();
  // the logged blocks, in the order they were added
  size_t io = 0, vo = 0;
  for (size_t b = 0; b < nblk; b++) {
    const int nr = _logHdr[2 * b], nc = _logHdr[2 * b + 1];
    hip_check(fh_mat_stage_block(_A, nr, _logIdx.data() + io, nc, _logIdx.data() + io + nr, _logVal.data() + vo), "HipMatrix::close (replay)");
    io += (size_t)nr + nc;
    vo += (size_t)nr * nc;
  }
  if (nblk) _pending = true;
  std::vector<int>().swap(_logHdr);
  std::vector<int>().swap(_logIdx);
  std::vector<double>().swap(_logVal);
}
void HipMatrix::set(const int i, const int j, const double v) {
  if (_A) hip_check(fh_mat_insert_row(handle(), i, 1, &j, &v), "HipMatrix::set");
  else if (!_logHdr.empty()) { close(); set(i, j, v); }       // an insert after adds: the adds come first
  else _stage[i][j] = v;
}
void HipMatrix::add(const int i, const int j, const double v) {
  const std::vector<int> r(1, i), c(1, j);
  add_matrix_blocked(std::vector<double>(1, v), r, c);
}
void HipMatrix::zero() {
  if (_A) {
    close();
    hip_check(fh_mat_zero(_A), "HipMatrix::zero");
  } else {
    for (auto& r : _stage) for (auto& kv : r) kv.second = 0.;
    std::fill(_logVal.begin(), _logVal.end(), 0.);       // the pattern of what was added stays (MatZeroEntries keeps it)
  }
}
double HipMatrix::operator()(const int i, const int j) const {
  close();
  int nc = 0;
  hip_check(fh_mat_get_row(_A, i, &nc, nullptr, nullptr), "operator()");
  std::vector<int> c(nc);
  std::vector<double> v(nc);
  hip_check(fh_mat_get_row(_A, i, &nc, c.data(), v.data()), "operator()");
  auto it = std::lower_bound(c.begin(), c.end(), j);
  return (it != c.end() && *it == j) ? v[it - c.begin()] : 0.;
}
int HipMatrix::MatGetRowM(const int i, int* cols, double* vals) {
  close();
  int nc = 0;
  hip_check(fh_mat_get_row(_A, i, &nc, cols, vals), "MatGetRowM");
  return nc;
}
void HipMatrix::insert_row(const int row, const int ncols, const std::vector<int>& cols, double* values) {
  if (!_A && !_logHdr.empty()) close();
  if (_A) hip_check(fh_mat_insert_row(handle(), row, ncols, cols.data(), values), "insert_row");
  else for (int k = 0; k < ncols; k++) _stage[row][cols[k]] = values[k];
}
// the per-element crossing (PetscMatrix.cpp:699-729): nothing touches the device here.  With a device pattern the block goes into the
// pinned ring of fh_mat_stage_block (a full ring leaves asynchronously); before the first close() it is logged on the host
void HipMatrix::add_matrix_blocked(const std::vector<double>& mat, const std::vector<int>& rows, const std::vector<int>& cols) {
  if (mat.size() != rows.size() * cols.size()) { std::cout << "HipMatrix::add_matrix_blocked: size mismatch" << std::endl; abort(); }
  if (_A) {
    hip_check(fh_mat_stage_block(_A, (int)rows.size(), rows.data(), (int)cols.size(), cols.data(), mat.data()), "add_matrix_blocked");
    _pending = true;
  } else {
    _logHdr.push_back((int)rows.size());
    _logHdr.push_back((int)cols.size());
    _logIdx.insert(_logIdx.end(), rows.begin(), rows.end());
    _logIdx.insert(_logIdx.end(), cols.begin(), cols.end());
    _logVal.insert(_logVal.end(), mat.begin(), mat.end());
  }
  _closed = false;
}
void HipMatrix::add_matrix_blocked(const std::vector<double>& mat, const std::vector<unsigned>& rows, const std::vector<unsigned>& cols) {
  std::vector<int> r(rows.begin(), rows.end()), c(cols.begin(), cols.end());
  add_matrix_blocked(mat, r, c);
}
void HipMatrix::matrix_PtAP(const SparseMatrix& P, const SparseMatrix& A, const bool& reuse) {
  fh_mat_t out = (reuse && _A) ? _A : nullptr;
  if (!reuse) clear();
  hip_check(fh_mat_ptap(hm(P).handle(), hm(A).handle(), &out), "matrix_PtAP");
  _A = out;
  fh_mat_size(_A, &_m, &_n, nullptr);
  _closed = true;
}
void HipMatrix::matrix_ABC(const SparseMatrix& A, const SparseMatrix& B, const SparseMatrix& C, const bool& reuse) {
  fh_mat_t out = (reuse && _A) ? _A : nullptr;       // MAT_REUSE_MATRIX: numeric only (PetscMatrix.cpp:833-856)
  if (!out) clear();
  hip_check(fh_mat_abc(hm(A).handle(), hm(B).handle(), hm(C).handle(), &out), "matrix_ABC");
  _A = out;
  fh_mat_size(_A, &_m, &_n, nullptr);
  _closed = true;
}
void HipMatrix::matrix_RightMatMult(const SparseMatrix& A) {
  fh_mat_t out = nullptr;
  hip_check(fh_mat_matmul(handle(), hm(A).handle(), &out), "matrix_RightMatMult");
  adopt(out);
}
void HipMatrix::matrix_LeftMatMult(const SparseMatrix& A) {
  fh_mat_t out = nullptr;
  hip_check(fh_mat_matmul(hm(A).handle(), handle(), &out), "matrix_LeftMatMult");
  adopt(out);
}
void HipMatrix::matrix_get_diagonal_values(const std::vector<int>& index, std::vector<double>& value) const {
  value.resize(index.size());
  for (size_t k = 0; k < index.size(); k++) value[k] = (*this)(index[k], index[k]);
}
double HipMatrix::l1_norm() const { close(); double r; hip_check(fh_mat_norm(_A, 1, &r), "l1_norm"); return r; }
double HipMatrix::linfty_norm() const { close(); double r; hip_check(fh_mat_norm(_A, 0, &r), "linfty_norm"); return r; }
void HipMatrix::get_diagonal(NumericVector& dest) const { close(); hip_check(fh_mat_get_diagonal(_A, hv(dest).handle()), "get_diagonal"); }
void HipMatrix::get_transpose(SparseMatrix& dest) const {
  close();
  fh_mat_t t = nullptr;
  hip_check(fh_mat_transpose(_A, &t), "get_transpose");
  static_cast<HipMatrix&>(dest).adopt(t);    // also valid for dest == *this (PetscMatrix.cpp:1051-1053)
}
void HipMatrix::mat_zero_rows(const std::vector<int>& index, const double& diag) const {
  close();
  hip_check(fh_mat_zero_rows(_A, (int)index.size(), index.data(), diag), "mat_zero_rows");
}

// =============================== LinearEquationSolverHip ===============================
LinearEquationSolverHip::~LinearEquationSolverHip() {
  if (_mg) fh_mg_destroy(_mg);
  if (_one) fh_mg_destroy(_one);
  // _KK, _RES, ... belong to LinearEquation (DeletePde, LinearEquation.cpp:378-405)
}
void LinearEquationSolverHip::SetTolerances(const double& rtol, const double& atol, const double& divtol, const unsigned& maxits,
                                            const unsigned& restart) {
  _rtol = rtol;
  _abstol = atol;
  _dtol = divtol;
  _maxits = (int)maxits;
  _restart = (int)restart;
}
void LinearEquationSolverHip::MGInit(const MgSmootherType& mg_smoother_type, const unsigned& levelMax, const SolverType& mgSolverType) {
  int cycle = FH_CYCLE_MULTIPLICATIVE;                 // PCMGSetType, LinearEquationSolverPetsc.cpp:199-214
  if (mg_smoother_type == FULL) cycle = FH_CYCLE_FULL;
  else if (mg_smoother_type == ADDITIVE) cycle = FH_CYCLE_ADDITIVE;
  else if (mg_smoother_type == KASKADE) cycle = FH_CYCLE_KASKADE;
  else if (mg_smoother_type != MULTIPLICATIVE) {
    std::cout << "Wrong mg_type for the HIP backend" << std::endl;
    abort();
  }
  if (_mg) fh_mg_destroy(_mg);
  _mg = nullptr;
  _levelMax = levelMax;
  _mgSolverType = mgSolverType;
  hip_check(fh_mg_create(hip_context(), (int)levelMax, &_mg), "MGInit");
  hip_check(fh_mg_set_cycle_type(_mg, cycle), "MGInit (cycle type)");
  _needs_setup = true;
}
void LinearEquationSolverHip::SetPenalty() {
  static_cast<HipMatrix*>(_KK)->mat_zero_rows(_bdcIndex, 1.);
}
void LinearEquationSolverHip::ZerosBoundaryResiduals() {
  if (_bdcIndex.empty()) return;
  std::vector<double> zeros(_bdcIndex.size(), 0.);
  _RES->insert_vector_blocked(zeros, _bdcIndex);
}
void LinearEquationSolverHip::MGSetLevel(LinearEquationSolver* LinSolver, const unsigned&, const std::vector<unsigned>& variable_to_be_solved,
                                         SparseMatrix* PP, SparseMatrix* RR, const unsigned& npre, const unsigned& npost) {
  LinearEquationSolverHip* top = static_cast<LinearEquationSolverHip*>(LinSolver);
  if (!_bdcIndexIsInitialized) BuildBdcIndex(variable_to_be_solved);      // LinearEquationSolverPetsc.cpp:223
  SetPenalty();
  if (_level != 0 && _levelSolverType == PREONLY) {   // LinearEquationSolverPetsc.cpp:245-248
    _levelSolverType = RICHARDSON;
    _richardsonScaleFactor = 1.;
  }
  const int smoother = smoother_id();
  if (_level != 0 && smoother != FH_SMOOTH_VANKA && smoother != FH_SMOOTH_ASM && _preconditioner_type != JACOBI_PRECOND && _preconditioner_type != SOR_PRECOND &&
      _preconditioner_type != ILU_PRECOND && _preconditioner_type != IDENTITY_PRECOND && _preconditioner_type != LU_PRECOND &&
      _preconditioner_type != MLU_PRECOND) {
    std::cout << "HIP backend: level preconditioner must be JACOBI_PRECOND, SOR_PRECOND, ILU_PRECOND, LU_PRECOND / MLU_PRECOND or IDENTITY_PRECOND (or the FEMuS_ASM solver)" << std::endl;
    abort();
  }
  if (_level != 0 && _levelSolverType != RICHARDSON && _levelSolverType != GMRES) {     // GMRES is the reference's default level solver
    std::cout << "HIP backend: the level solver must be RICHARDSON or GMRES (SetSolverFineGrids)" << std::endl;
    abort();
  }
  if (_level != 0) attach_smoother_data(top->_mg, (int)_level, variable_to_be_solved);
  fh_mat_t P = PP ? static_cast<HipMatrix*>(PP)->handle() : nullptr;
  fh_mat_t R = (RR && RR != PP) ? static_cast<HipMatrix*>(RR)->handle() : nullptr;   // RR == PP means "use PP^T"
  hip_check(fh_mg_set_level(top->_mg, (int)_level, static_cast<HipMatrix*>(_KK)->handle(), _level ? P : nullptr, _level ? R : nullptr,
                            smoother, _richardsonScaleFactor, (int)npre, (int)npost),
            "MGSetLevel");
  if (_coordDim > 0 && !_coords.empty() && (_level == 0 || smoother == FH_SMOOTH_LU))     // the exact solves cut their dissection at coordinate layers
    hip_check(fh_mg_set_level_coords(top->_mg, (int)_level, _coordDim, (int)(_coords.size() / (size_t)_coordDim), _coords.data()), "MGSetLevel: coordinates of the level");
  if (_level != 0)      // KSPGMRES with KSPGMRESSetRestart(_restart) and npre / npost iterations, or KSPRICHARDSON (LinearEquationSolverPetsc.cpp:238-250, 501-519)
    hip_check(fh_mg_set_level_solver(top->_mg, (int)_level, _levelSolverType == GMRES ? FH_LEVEL_GMRES : FH_LEVEL_RICHARDSON, _restart > 0 ? _restart : 30),
              "MGSetLevel: level solver");
  top->_needs_setup = true;
}
// PCSOR runs in the natural row order as PETSc does (level-scheduled); SetMulticolourSor(true) selects the colour order instead
int LinearEquationSolverHip::smoother_id() const {
  if (_preconditioner_type == SOR_PRECOND) return _multicolourSor ? FH_SMOOTH_GS_COLOR : FH_SMOOTH_SOR;
  if (_preconditioner_type == ILU_PRECOND) return FH_SMOOTH_ILU0;
  if (_preconditioner_type == IDENTITY_PRECOND) return FH_SMOOTH_IDENTITY;
  if (_preconditioner_type == LU_PRECOND || _preconditioner_type == MLU_PRECOND) return FH_SMOOTH_LU;      // PCLU (MUMPS), PetscPreconditioner.cpp:147-160
  return FH_SMOOTH_JACOBI;
}
void LinearEquationSolverHipAsm::attach_smoother_data(fh_mg_t mg, int level, const std::vector<unsigned>& variable_to_be_solved) {
  if (!_blocksGiven) BuildASMIndex(variable_to_be_solved);      // LinearEquationSolverPetscAsm::MGSetLevel builds them with the Dirichlet index
  if (_blockPtr.size() < 2) {
    std::cout << "HIP backend: FEMuS_ASM level " << level << " has no blocks" << std::endl;
    abort();
  }
  hip_check(fh_mg_set_level_patches(mg, level, (int)_blockPtr.size() - 1, _blockPtr.data(), _blockDofs.data()), "MGSetLevel: ASM blocks");
  if (smoother_id() == FH_SMOOTH_ASM) hip_check(fh_mg_set_level_patches_exact(mg, level, _blockExactCount), "MGSetLevel: exact ASM blocks");
}
void LinearEquationSolverHip::MGSolve(const bool) {
  if (_needs_setup) {
    hip_check(fh_mg_setup(_mg), "MGSolve: setup");
    _needs_setup = false;
  }
  ZerosBoundaryResiduals();
  const int outer = (_mgSolverType == PREONLY) ? FH_OUTER_PREONLY : (_mgSolverType == RICHARDSON) ? FH_OUTER_RICHARDSON
                    : (_mgSolverType == CG) ? FH_OUTER_CG : (_mgSolverType == FGMRES) ? FH_OUTER_FGMRES : FH_OUTER_GMRES;
  hip_check(fh_mg_solve(_mg, static_cast<HipVector*>(_RES)->handle(), static_cast<HipVector*>(_EPSC)->handle(), outer, _rtol, _abstol, _dtol,
                        _maxits, _restart, &_its, &_rnorm),
            "MGSolve");
  _RESC->matrix_mult(*_EPSC, *_KK);   // LinearEquationSolverPetsc.cpp:333-335
  *_RES -= *_RESC;
  *_EPS += *_EPSC;
}
void LinearEquationSolverHip::MGClear() {
  if (_mg) fh_mg_destroy(_mg);
  _mg = nullptr;
}
void LinearEquationSolverHip::Solve(const std::vector<unsigned>& variable_to_be_solved, const bool& ksp_clean) {
  if (!_bdcIndexIsInitialized) BuildBdcIndex(variable_to_be_solved);
  fh_mat_t KK = static_cast<HipMatrix*>(_KK)->handle();
  if (ksp_clean || !_one) {       // this->Clear(); SetPenalty(); this->Init(KK, KK)  (LinearEquationSolverPetsc.cpp:101-107)
    if (_one) fh_mg_destroy(_one);
    _one = nullptr;
    SetPenalty();
    KK = static_cast<HipMatrix*>(_KK)->handle();
    // a one-level hierarchy: the "cycle" is the exact solve of this level (what the reference reaches with its default
    // GMRES + ILU/MLU level solver at convergence): the sparse exact solve for symmetric operators of any size, the dense inverse otherwise
    hip_check(fh_mg_create(hip_context(), 1, &_one), "Solve");
    hip_check(fh_mg_set_level(_one, 0, KK, nullptr, nullptr, FH_SMOOTH_JACOBI, 1.0, 1, 0), "Solve");
    if (_coordDim > 0 && !_coords.empty())
      hip_check(fh_mg_set_coarse_coords(_one, _coordDim, (int)(_coords.size() / (size_t)_coordDim), _coords.data()), "Solve: coordinates of the level");
    hip_check(fh_mg_setup(_one), "Solve: factorisation of the level operator");
  }
  ZerosBoundaryResiduals();
  hip_check(fh_mg_solve(_one, static_cast<HipVector*>(_RES)->handle(), static_cast<HipVector*>(_EPSC)->handle(), FH_OUTER_PREONLY, _rtol, _abstol,
                        _dtol, _maxits, _restart, &_its, &_rnorm),
            "Solve");
  *_EPS += *_EPSC;                     // :123-126
  _RESC->matrix_mult(*_EPSC, *_KK);
  *_RES -= *_RESC;
}

}  // namespace femus
