// Batched form of the per-element crossings of the reference interface:
//   SparseMatrix::add_matrix_blocked  (SparseMatrix.hpp:165-171, PetscMatrix.cpp:699-729: MatSetValues ADD_VALUES per element)
//   NumericVector::add_vector_blocked (NumericVector.hpp:265-269, PetscVector.cpp:132-153: VecSetValues ADD_VALUES per element)
// An application such as applications/001_Poisson/main.cpp:283-609 calls them once per element.  PETSc keeps such adds in a stash
// until MatAssemblyEnd / VecAssemblyEnd (close()); this file does the same on the device's side of the bus:
//   stage:  the call appends (rows, cols, values) to a pinned host ring -- a few memcpys, no device call at all;
//   flush:  when a ring is full or the object is closed the ring goes to the device in two asynchronous copies (integers, values)
//           followed by ONE kernel; the host goes on filling the second ring meanwhile.
// The kernel gives every touched matrix row to one wave, which keeps the row in LDS and adds the staged contributions to it in
// the order of the calls (positions by bisection in the row's sorted columns): no atomics between rows, a fixed summation order,
// and therefore the same bits as adding the elements one after the other.  Entries outside the fixed pattern raise an error at
// the flush unless the value is zero (PETSc would allocate; the device pattern is fixed).
#include "fh_internal.h"
#include <unordered_map>

namespace {

constexpr size_t MAT_CAP_D = 4u << 20;    // doubles per ring (32 MB)
constexpr size_t MAT_CAP_I = 3u << 20;    // ints per ring (12 MB)
constexpr size_t VEC_CAP_D = 1u << 20;
constexpr size_t VEC_CAP_I = 4u << 20;
constexpr int ROW_CAP = 2048;             // longest row the LDS kernel keeps (24 KB of LDS per wave)

extern "C" int fh_vec_flush(fh_vec_t v);

struct ring_s {
  int* h_i = nullptr;        // pinned: [block chunks ... | touch lists]
  double* h_d = nullptr;     // pinned
  int* d_i = nullptr;
  double* d_d = nullptr;
  size_t ni = 0, nd = 0;     // used
  size_t touch = 0;          // ints the touch lists of the staged blocks will need at the flush
  std::vector<int> chunk;    // start of every block's chunk in h_i
  hipEvent_t done = nullptr;
  bool in_flight = false;
};

}  // namespace

struct fh_stage_s {
  fh_ctx_t ctx = nullptr;
  size_t cap_d = 0, cap_i = 0;
  ring_s ring[2];
  int cur = 0;
  int* d_err = nullptr;
  int* h_err = nullptr;
  bool pending = false;          // staged or in flight since the last completed flush
  std::vector<int> cnt, slot;    // per target row / entry: contributions in the ring being flushed, index into the touched list
  std::vector<int> touched;
  std::unordered_map<int, int> ghost_slot;   // vectors: global index of a ghost -> local slot
  bool ghost_built = false;
  int64_t n_flush = 0, n_blocks = 0;
};

static void stage_free_impl(fh_stage_s* s) {
  if (!s) return;
  for (auto& r : s->ring) {
    if (r.in_flight) hipEventSynchronize(r.done);
    if (r.done) hipEventDestroy(r.done);
    if (r.h_i) hipHostFree(r.h_i);
    if (r.h_d) hipHostFree(r.h_d);
    if (r.d_i) hipFree(r.d_i);
    if (r.d_d) hipFree(r.d_d);
  }
  if (s->d_err) hipFree(s->d_err);
  if (s->h_err) hipHostFree(s->h_err);
  delete s;
}
void fh_stage_free(fh_stage_s* s) { stage_free_impl(s); }

static int stage_create(fh_ctx_t c, size_t cap_d, size_t cap_i, int ntarget, fh_stage_s** out) {
  fh_stage_s* s = new fh_stage_s();
  s->ctx = c;
  s->cap_d = cap_d;
  s->cap_i = cap_i;
  for (auto& r : s->ring) {
    if (hipHostMalloc(&r.h_i, cap_i * sizeof(int)) != hipSuccess || hipHostMalloc(&r.h_d, cap_d * sizeof(double)) != hipSuccess ||
        hipMalloc(&r.d_i, cap_i * sizeof(int)) != hipSuccess || hipMalloc(&r.d_d, cap_d * sizeof(double)) != hipSuccess ||
        hipEventCreateWithFlags(&r.done, hipEventDisableTiming) != hipSuccess) {
      stage_free_impl(s);
      fh_set_error("staging ring: out of pinned host or device memory");
      return 1;
    }
  }
  if (hipMalloc(&s->d_err, 4 * sizeof(int)) != hipSuccess || hipHostMalloc(&s->h_err, 4 * sizeof(int)) != hipSuccess) {
    stage_free_impl(s);
    fh_set_error("staging ring: out of memory");
    return 1;
  }
  hipMemsetAsync(s->d_err, 0, 4 * sizeof(int), c->stream);
  s->cnt.assign(ntarget, 0);
  s->slot.assign(ntarget, 0);
  *out = s;
  return 0;
}

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
// chunk of a block in the integer ring: {nrow, ncol, first value (doubles), 0, rows[nrow], cols[ncol]}
// touch lists: trow[nt], tptr[nt+1], tlist[2*ncontrib] = (chunk start, local row) in the order of the calls
template <bool LDS_ROW>
__global__ __launch_bounds__(64) void k_stage_flush_rows(const int* __restrict__ rowptr, const int* __restrict__ col, double* __restrict__ val,
                                                          const int* __restrict__ ints, const double* __restrict__ vals, int trow_off,
                                                          int tptr_off, int tlist_off, int* __restrict__ err) {
  __shared__ double acc[LDS_ROW ? ROW_CAP : 1];
  __shared__ int lc[LDS_ROW ? ROW_CAP : 1];
  const int t = blockIdx.x, lane = threadIdx.x;
  const int r = ints[trow_off + t];
  const int s = rowptr[r], len = rowptr[r + 1] - s;
  if (LDS_ROW) {
    for (int k = lane; k < len; k += 64) {
      acc[k] = val[s + k];
      lc[k] = col[s + k];
    }
    __syncthreads();
  }
  const int c0 = ints[tptr_off + t], c1 = ints[tptr_off + t + 1];
  for (int c = c0; c < c1; c++) {
    const int chunk = ints[tlist_off + 2 * c], i = ints[tlist_off + 2 * c + 1];
    const int nrow = ints[chunk], ncol = ints[chunk + 1], voff = ints[chunk + 2];
    const int* cols = ints + chunk + 4 + nrow;
    const double* v = vals + voff + (size_t)i * ncol;
    for (int j = lane; j < ncol; j += 64) {
      const int cj = cols[j];
      const double vj = v[j];
      int lo = 0, hi = len;
      while (lo < hi) {
        const int mid = lo + ((hi - lo) >> 1);
        const int cm = LDS_ROW ? lc[mid] : col[s + mid];
        if (cm < cj) lo = mid + 1;
        else hi = mid;
      }
      const bool found = lo < len && (LDS_ROW ? lc[lo] : col[s + lo]) == cj;
      if (found) {
        // columns of one block are distinct in every caller of the reference, so lanes hit distinct entries; the atomic form keeps
        // the sum right (if not its rounding order) should a caller repeat a column
        if (LDS_ROW) atomicAdd(&acc[lo], vj);
        else atomicAdd(&val[s + lo], vj);
      } else if (vj != 0.0) {
        if (atomicExch(&err[0], 1) == 0) {
          err[1] = r;
          err[2] = cj;
        }
      }
    }
    if (!LDS_ROW) __threadfence();     // the next contribution may add to the same entries from other lanes
    __syncthreads();
  }
  if (LDS_ROW)
    for (int k = lane; k < len; k += 64) val[s + k] = acc[k];
}

// vectors: one thread per distinct target entry, its staged values added in the order of the calls
__global__ __launch_bounds__(256) void k_stage_flush_vec(double* __restrict__ y, double* __restrict__ gacc, int n_local, const int* __restrict__ ints,
                                                          const double* __restrict__ vals, int tidx_off, int tptr_off, int tlist_off, int nt) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= nt) return;
  const int idx = ints[tidx_off + t];
  double* tgt = idx < n_local ? y + idx : gacc + (idx - n_local);      // adds to a ghost entry go to the accumulator the owner receives
  double a = *tgt;
  for (int c = ints[tptr_off + t]; c < ints[tptr_off + t + 1]; c++) a += vals[ints[tlist_off + c]];
  *tgt = a;
}

// ------------------------------------------------------------------------------------------------
// rings
// ------------------------------------------------------------------------------------------------
static int ring_wait(ring_s& r) {
  if (r.in_flight) {
    FH_CHECK_HIP(hipEventSynchronize(r.done));
    r.in_flight = false;
  }
  return 0;
}

static int ring_issued(fh_stage_s* s, ring_s& r) {
  FH_CHECK_HIP(hipGetLastError());
  FH_CHECK_HIP(hipEventRecord(r.done, s->ctx->stream));
  r.in_flight = true;
  r.ni = r.nd = r.touch = 0;
  r.chunk.clear();
  s->cur ^= 1;
  s->n_flush++;
  return ring_wait(s->ring[s->cur]);     // the ring staged into next must have left the host
}

// sends the current ring of a matrix and starts the kernel; asynchronous
static int mat_issue(fh_mat_t A) {
  fh_stage_s* s = A->stage;
  ring_s& r = s->ring[s->cur];
  if (r.chunk.empty()) return 0;
  // touch lists: rows in first-touch order, their contributions in call order (a counting sort over the staged rows)
  s->touched.clear();
  for (int ch : r.chunk) {
    const int nrow = r.h_i[ch];
    const int* rows = r.h_i + ch + 4;
    for (int i = 0; i < nrow; i++)
      if (s->cnt[rows[i]]++ == 0) {
        s->slot[rows[i]] = (int)s->touched.size();
        s->touched.push_back(rows[i]);
      }
  }
  const int nt = (int)s->touched.size();
  const size_t trow_off = r.ni, tptr_off = trow_off + nt, tlist_off = tptr_off + nt + 1;
  int* trow = r.h_i + trow_off;
  int* tptr = r.h_i + tptr_off;
  int* tlist = r.h_i + tlist_off;
  int acc = 0;
  for (int t = 0; t < nt; t++) {
    trow[t] = s->touched[t];
    tptr[t] = acc;
    acc += s->cnt[s->touched[t]];
    s->cnt[s->touched[t]] = 0;      // reused as the fill cursor below, then left at zero for the next flush
  }
  tptr[nt] = acc;
  FH_REQUIRE(tlist_off + 2 * (size_t)acc <= s->cap_i, "staging ring: touch lists overflow the ring (internal)");
  for (int ch : r.chunk) {
    const int nrow = r.h_i[ch];
    const int* rows = r.h_i + ch + 4;
    for (int i = 0; i < nrow; i++) {
      const int t = s->slot[rows[i]];
      const int p = tptr[t] + s->cnt[rows[i]]++;
      tlist[2 * p] = ch;
      tlist[2 * p + 1] = i;
    }
  }
  for (int t = 0; t < nt; t++) s->cnt[s->touched[t]] = 0;
  const size_t ni_all = tlist_off + 2 * (size_t)acc;
  hipStream_t st = s->ctx->stream;
  FH_CHECK_HIP(hipMemcpyAsync(r.d_i, r.h_i, ni_all * sizeof(int), hipMemcpyHostToDevice, st));
  FH_CHECK_HIP(hipMemcpyAsync(r.d_d, r.h_d, r.nd * sizeof(double), hipMemcpyHostToDevice, st));
  if (A->max_row <= ROW_CAP)
    hipLaunchKernelGGL(k_stage_flush_rows<true>, dim3(nt), dim3(64), 0, st, A->d_rowptr, A->d_col, A->d_val, r.d_i, r.d_d, (int)trow_off,
                       (int)tptr_off, (int)tlist_off, s->d_err);
  else
    hipLaunchKernelGGL(k_stage_flush_rows<false>, dim3(nt), dim3(64), 0, st, A->d_rowptr, A->d_col, A->d_val, r.d_i, r.d_d, (int)trow_off,
                       (int)tptr_off, (int)tlist_off, s->d_err);
  A->at_valid = false;
  return ring_issued(s, r);
}

static int stage_rows(fh_mat_t A, int nrow, const int* rows, int ncol, const int* cols, const double* vals) {
  fh_stage_s* s = A->stage;
  const size_t need_i = 4 + (size_t)nrow + ncol, need_t = 4 * (size_t)nrow + 1, need_d = (size_t)nrow * ncol;
  {
    ring_s& r = s->ring[s->cur];
    if (r.ni + need_i + r.touch + need_t > s->cap_i || r.nd + need_d > s->cap_d) FH_TRY(mat_issue(A));
  }
  ring_s& r = s->ring[s->cur];
  int* h = r.h_i + r.ni;
  h[0] = nrow;
  h[1] = ncol;
  h[2] = (int)r.nd;
  h[3] = 0;
  memcpy(h + 4, rows, (size_t)nrow * sizeof(int));
  memcpy(h + 4 + nrow, cols, (size_t)ncol * sizeof(int));
  memcpy(r.h_d + r.nd, vals, need_d * sizeof(double));
  r.chunk.push_back((int)r.ni);
  r.ni += need_i;
  r.touch += need_t;
  r.nd += need_d;
  s->pending = true;
  s->n_blocks++;
  return 0;
}

extern "C" int fh_mat_stage_block(fh_mat_t A, int nrow, const int* rows, int ncol, const int* cols, const double* vals) {
  FH_REQUIRE(A && nrow >= 0 && ncol >= 0, "fh_mat_stage_block: bad arguments");
  if (nrow == 0 || ncol == 0) return 0;
  FH_REQUIRE(rows && cols && vals, "fh_mat_stage_block: null array");
  for (int i = 0; i < nrow; i++) FH_REQUIRE(rows[i] >= 0 && rows[i] < A->m, "fh_mat_stage_block: row %d out of range", rows[i]);
  for (int j = 0; j < ncol; j++) FH_REQUIRE(cols[j] >= 0 && cols[j] < A->n, "fh_mat_stage_block: column %d out of range", cols[j]);
  if (!A->stage) FH_TRY(stage_create(A->ctx, MAT_CAP_D, MAT_CAP_I, A->m, &A->stage));
  // a block that would not fit half a ring goes in slices of rows (same order of additions: row slices touch disjoint entries
  // unless the caller repeats a row, and then the slices still arrive in order)
  const size_t max_d = A->stage->cap_d / 2, max_i = A->stage->cap_i / 4;
  FH_REQUIRE((size_t)ncol <= max_d && 5 + (size_t)ncol <= max_i, "fh_mat_stage_block: %d columns in one block", ncol);
  int per = nrow;
  while ((size_t)per * ncol > max_d || 5 * (size_t)per + ncol + 5 > max_i) per = (per + 1) / 2;
  for (int i0 = 0; i0 < nrow; i0 += per) {
    const int n = std::min(per, nrow - i0);
    FH_TRY(stage_rows(A, n, rows + i0, ncol, cols, vals + (size_t)i0 * ncol));
  }
  return 0;
}

static int stage_finish(fh_stage_s* s, const char* who) {
  hipStream_t st = s->ctx->stream;
  FH_CHECK_HIP(hipMemcpyAsync(s->h_err, s->d_err, 4 * sizeof(int), hipMemcpyDeviceToHost, st));
  FH_CHECK_HIP(hipStreamSynchronize(st));
  for (auto& r : s->ring) r.in_flight = false;
  s->pending = false;
  if (s->h_err[0]) {
    const int row = s->h_err[1], colm = s->h_err[2];
    hipMemsetAsync(s->d_err, 0, 4 * sizeof(int), st);
    fh_set_error("%s: entry (%d,%d) is outside the pattern", who, row, colm);
    return 2;
  }
  return 0;
}

extern "C" int fh_mat_flush(fh_mat_t A) {
  if (A) A->val_gen++;
  FH_REQUIRE(A, "fh_mat_flush: null matrix");
  if (!A->stage || !A->stage->pending) return 0;
  FH_TRY(mat_issue(A));
  return stage_finish(A->stage, "fh_mat_add_block");
}

extern "C" int fh_mat_stage_stats(fh_mat_t A, int64_t* blocks, int64_t* flushes) {
  FH_REQUIRE(A, "fh_mat_stage_stats: null matrix");
  if (blocks) *blocks = A->stage ? A->stage->n_blocks : 0;
  if (flushes) *flushes = A->stage ? A->stage->n_flush : 0;
  return 0;
}

// ------------------------------------------------------------------------------------------------
// vectors
// ------------------------------------------------------------------------------------------------
static int vec_issue(fh_vec_t v) {
  fh_stage_s* s = v->stage;
  ring_s& r = s->ring[s->cur];
  if (r.nd == 0) return 0;
  const int n = (int)r.nd;          // ints [0, n): local slot of every staged value
  s->touched.clear();
  for (int k = 0; k < n; k++)
    if (s->cnt[r.h_i[k]]++ == 0) {
      s->slot[r.h_i[k]] = (int)s->touched.size();
      s->touched.push_back(r.h_i[k]);
    }
  const int nt = (int)s->touched.size();
  const size_t tidx_off = n, tptr_off = tidx_off + nt, tlist_off = tptr_off + nt + 1;
  int* tidx = r.h_i + tidx_off;
  int* tptr = r.h_i + tptr_off;
  int* tlist = r.h_i + tlist_off;
  int acc = 0;
  for (int t = 0; t < nt; t++) {
    tidx[t] = s->touched[t];
    tptr[t] = acc;
    acc += s->cnt[s->touched[t]];
    s->cnt[s->touched[t]] = 0;
  }
  tptr[nt] = acc;
  for (int k = 0; k < n; k++) tlist[tptr[s->slot[r.h_i[k]]] + s->cnt[r.h_i[k]]++] = k;
  for (int t = 0; t < nt; t++) s->cnt[s->touched[t]] = 0;
  hipStream_t st = s->ctx->stream;
  bool to_ghost = false;
  for (int t = 0; t < nt && !to_ghost; t++) to_ghost = tidx[t] >= v->n_local;
  if (to_ghost) {
    if (!v->d_gacc) {
      FH_CHECK_HIP(hipMalloc(&v->d_gacc, (size_t)v->nghost * sizeof(double)));
      FH_CHECK_HIP(hipMemsetAsync(v->d_gacc, 0, (size_t)v->nghost * sizeof(double), st));
    }
    v->gacc_dirty = true;
  }
  FH_CHECK_HIP(hipMemcpyAsync(r.d_i, r.h_i, (tlist_off + n) * sizeof(int), hipMemcpyHostToDevice, st));
  FH_CHECK_HIP(hipMemcpyAsync(r.d_d, r.h_d, (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_stage_flush_vec, dim3(fh_div_up(nt, 256)), dim3(256), 0, st, v->d, v->d_gacc, v->n_local, r.d_i, r.d_d, (int)tidx_off, (int)tptr_off,
                     (int)tlist_off, nt);
  return ring_issued(s, r);
}

extern "C" int fh_vec_stage_values(fh_vec_t v, int n, const int* idx, const double* vals) {
  FH_REQUIRE(v && n >= 0, "fh_vec_stage_values: bad arguments");
  if (n == 0) return 0;
  FH_REQUIRE(idx && vals, "fh_vec_stage_values: null array");
  if (!v->stage) FH_TRY(stage_create(v->ctx, VEC_CAP_D, VEC_CAP_I, v->n_local + v->nghost, &v->stage));
  fh_stage_s* s = v->stage;
  if (v->nghost && !s->ghost_built) {
    for (int k = 0; k < v->nghost; k++) s->ghost_slot.emplace(v->ghost_idx[k], v->n_local + k);
    s->ghost_built = true;
  }
  // all indices are resolved before anything is staged: a failing call leaves nothing behind
  static thread_local std::vector<int> loc;
  loc.resize(n);
  for (int k = 0; k < n; k++) {
    const int g = idx[k];
    int l = -1;
    if (g >= v->first_local && g < v->first_local + v->n_local) l = g - v->first_local;
    else if (v->nghost) {
      auto it = s->ghost_slot.find(g);
      if (it != s->ghost_slot.end()) l = it->second;
    }
    FH_REQUIRE(l >= 0, "vector index %d is neither owned nor a ghost on this rank", g);
    loc[k] = l;
  }
  const size_t ring_vals = std::min(s->cap_d, (s->cap_i - 2) / 4);     // 4 ints per staged value: slot + its share of the touch lists
  for (int k0 = 0; k0 < n;) {
    ring_s* r = &s->ring[s->cur];
    if (r->nd >= ring_vals) {
      FH_TRY(vec_issue(v));
      continue;
    }
    const int m = (int)std::min<size_t>(ring_vals - r->nd, (size_t)(n - k0));
    memcpy(r->h_i + r->nd, loc.data() + k0, (size_t)m * sizeof(int));
    memcpy(r->h_d + r->nd, vals + k0, (size_t)m * sizeof(double));
    r->nd += m;
    s->pending = true;
    k0 += m;
  }
  s->n_blocks++;
  return 0;
}

extern "C" int fh_vec_ghost_adds(fh_vec_t v, double* out) {
  FH_REQUIRE(v && (out || v->nghost == 0), "fh_vec_ghost_adds: null argument");
  if (v->stage && v->stage->pending) FH_TRY(fh_vec_flush(v));
  if (!v->nghost) return 0;
  if (!v->d_gacc) {
    memset(out, 0, (size_t)v->nghost * sizeof(double));
    return 0;
  }
  FH_CHECK_HIP(hipMemcpyAsync(out, v->d_gacc, (size_t)v->nghost * sizeof(double), hipMemcpyDeviceToHost, v->ctx->stream));
  FH_CHECK_HIP(hipStreamSynchronize(v->ctx->stream));
  return 0;
}

extern "C" int fh_vec_ghost_adds_pending(fh_vec_t v, int* pending) {
  FH_REQUIRE(v && pending, "fh_vec_ghost_adds_pending: null argument");
  if (v->stage && v->stage->pending) FH_TRY(fh_vec_flush(v));
  *pending = v->gacc_dirty ? 1 : 0;
  return 0;
}

extern "C" int fh_vec_flush(fh_vec_t v) {
  FH_REQUIRE(v, "fh_vec_flush: null vector");
  if (!v->stage || !v->stage->pending) return 0;
  FH_TRY(vec_issue(v));
  return stage_finish(v->stage, "fh_vec_add_values");
}
