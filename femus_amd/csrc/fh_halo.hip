// Multi-GPU halo exchange and scalar reductions (C1, C3, C4 of SURVEY 2.1) over RCCL / xGMI.
// Replaces what PETSc does inside VecGhostUpdateBegin/End (src/03_algebra/00_vectors/PetscVector.hpp:595-612),
// the MPIAIJ MatMult scatter, and VecDot/VecNorm all-reduces (Parallel.hpp:351-377).
//
// One rank per GPU.  The exchange is a neighbour all-to-all: a pack kernel gathers the owned interface entries
// into one send buffer (grouped by destination), a single RCCL group of ncclSend/ncclRecv pairs moves each
// neighbour's slice over its direct xGMI link, and the receives land straight in the ghost tail of the vector
// ([owned | ghosts grouped by source rank]).  Communication runs on the context's second stream so interior work
// queued on the compute stream overlaps with it; events join the two streams.
#include "fh_internal.h"
#include <rccl/rccl.h>

struct fh_halo_s {
  fh_ctx_t ctx = nullptr;
  int rank = 0, nranks = 1;
  ncclComm_t comm = nullptr;
  bool owns_comm = false;
  std::vector<int> send_counts, recv_counts, send_off, recv_off;
  int nsend = 0, nrecv = 0;
  int* d_send_idx = nullptr;
  double* d_sendbuf = nullptr;
  double* d_scalars = nullptr;
  hipEvent_t ev_packed = nullptr, ev_done = nullptr, ev_d2h = nullptr;
  // begin/end pair in flight (the exchange runs on the communication stream while the caller queues interior work)
  double* pend_vd = nullptr;
  int pend_owned = 0;
  bool pending = false;
  // statistics (fh_halo_stats): exchanges started, payload sent; with the context option "halo_profile" also the duration of the
  // exchanges (pack done -> ghosts landed) and the part of it the compute stream really waited for (exposed)
  int64_t n_updates = 0, bytes_sent = 0, n_allreduce = 0;
  double exchange_ms = 0.0, exposed_ms = 0.0, allreduce_ms = 0.0;
  hipEvent_t evt_begin = nullptr, evt_ready = nullptr;
  // host-staged transport (fh_halo_create_host): the exchange itself is the caller's function (MPI_Neighbor_alltoallv, sockets ...)
  fh_exchange_fn exchange = nullptr;
  fh_allreduce_fn allreduce = nullptr;
  void* user = nullptr;
  double *h_send = nullptr, *h_recv = nullptr;   // pinned
};

// a one-rank plan without communicator or transport has nothing to exchange
static inline bool halo_inert(fh_halo_t h) { return h->nranks == 1 && !h->comm && !h->exchange; }

#define FH_CHECK_NCCL(expr)                                                                     \
  do {                                                                                          \
    ncclResult_t _r = (expr);                                                                   \
    if (_r != ncclSuccess) {                                                                    \
      fh_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, ncclGetErrorString(_r));  \
      return 1;                                                                                 \
    }                                                                                           \
  } while (0)

__global__ __launch_bounds__(256) void k_pack(const double* __restrict__ v, const int* __restrict__ idx, double* __restrict__ buf, int n) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) buf[i] = v[idx[i]];
}

extern "C" int fh_halo_unique_id(char id128[128]) {
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
  ncclUniqueId id;
  FH_CHECK_NCCL(ncclGetUniqueId(&id));
  memcpy(id128, &id, 128);
  return 0;
}

static int halo_create(fh_ctx_t ctx, int rank, int nranks, const char* id128, ncclComm_t shared, const int* send_counts, const int* send_idx,
                       const int* recv_counts, fh_halo_t* out, int plan_ranks = 0);

extern "C" int fh_halo_create(fh_ctx_t ctx, int rank, int nranks, const char id128[128], const int* send_counts, const int* send_idx,
                              const int* recv_counts, fh_halo_t* out) {
  return halo_create(ctx, rank, nranks, id128, nullptr, send_counts, send_idx, recv_counts, out);
}

static int host_buffers(fh_halo_t h) {
  FH_CHECK_HIP(hipHostMalloc((void**)&h->h_send, std::max<size_t>(std::max(h->nsend, 256), 1) * sizeof(double), hipHostMallocDefault));
  FH_CHECK_HIP(hipHostMalloc((void**)&h->h_recv, std::max<size_t>(h->nrecv, 1) * sizeof(double), hipHostMallocDefault));
  return 0;
}

extern "C" int fh_halo_create_shared(fh_halo_t parent, const int* send_counts, const int* send_idx, const int* recv_counts, fh_halo_t* out) {
  FH_REQUIRE(parent, "fh_halo_create_shared: null parent");
  if (parent->exchange) {
    FH_TRY(halo_create(parent->ctx, parent->rank, 1, nullptr, nullptr, send_counts, send_idx, recv_counts, out, parent->nranks));
    (*out)->exchange = parent->exchange;
    (*out)->allreduce = parent->allreduce;
    (*out)->user = parent->user;
    return host_buffers(*out);
  }
  return halo_create(parent->ctx, parent->rank, parent->nranks, nullptr, parent->comm, send_counts, send_idx, recv_counts, out);
}

// version of the RCCL this process really calls (ncclGetVersion of whichever librccl the dynamic loader bound: the launcher reports it with the path)
extern "C" int fh_rccl_version(int* version) {
  FH_REQUIRE(version, "fh_rccl_version: null argument");
  int v = 0;
  const ncclResult_t r = ncclGetVersion(&v);
  FH_REQUIRE(r == ncclSuccess, "fh_rccl_version: ncclGetVersion failed: %s", ncclGetErrorString(r));
  *version = v;
  return 0;
}

extern "C" int fh_halo_create_host(fh_ctx_t ctx, int rank, int nranks, fh_exchange_fn exchange, fh_allreduce_fn allreduce, void* user,
                                   const int* send_counts, const int* send_idx, const int* recv_counts, fh_halo_t* out) {
  FH_REQUIRE(exchange && allreduce, "fh_halo_create_host: null transport function");
  FH_TRY(halo_create(ctx, rank, 1, nullptr, nullptr, send_counts, send_idx, recv_counts, out, nranks));   // no RCCL communicator
  (*out)->exchange = exchange;
  (*out)->allreduce = allreduce;
  (*out)->user = user;
  return host_buffers(*out);
}

static int halo_create(fh_ctx_t ctx, int rank, int comm_ranks, const char* id128, ncclComm_t shared, const int* send_counts, const int* send_idx,
                       const int* recv_counts, fh_halo_t* out, int plan_ranks) {
  FH_REQUIRE(ctx && out && send_counts && recv_counts, "fh_halo_create: null argument");
  const int nranks = plan_ranks > 0 ? plan_ranks : comm_ranks;     // host transport: the plan spans plan_ranks, no RCCL communicator
  FH_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "fh_halo_create: bad rank %d of %d", rank, nranks);
  FH_CHECK_HIP(hipSetDevice(ctx->device));      // the communicator binds to the CURRENT device: make it this context's (one rank = one device)
  fh_halo_t h = new fh_halo_s();
  h->ctx = ctx;
  h->rank = rank;
  h->nranks = nranks;
  h->send_counts.assign(send_counts, send_counts + nranks);
  h->recv_counts.assign(recv_counts, recv_counts + nranks);
  h->send_off.assign(nranks + 1, 0);
  h->recv_off.assign(nranks + 1, 0);
  for (int r = 0; r < nranks; r++) {
    FH_REQUIRE(send_counts[r] >= 0 && recv_counts[r] >= 0, "fh_halo_create: negative count");
    h->send_off[r + 1] = h->send_off[r] + send_counts[r];
    h->recv_off[r + 1] = h->recv_off[r] + recv_counts[r];
  }
  h->nsend = h->send_off[nranks];
  h->nrecv = h->recv_off[nranks];
  FH_REQUIRE(h->nsend == 0 || send_idx, "fh_halo_create: null send index list");
  FH_CHECK_HIP(hipMalloc(&h->d_send_idx, std::max(h->nsend, 1) * sizeof(int)));
  FH_CHECK_HIP(hipMalloc(&h->d_sendbuf, std::max(h->nsend, 1) * sizeof(double)));
  FH_CHECK_HIP(hipMalloc(&h->d_scalars, 256 * sizeof(double)));
  if (h->nsend) FH_CHECK_HIP(hipMemcpy(h->d_send_idx, send_idx, h->nsend * sizeof(int), hipMemcpyHostToDevice));
  FH_CHECK_HIP(hipEventCreateWithFlags(&h->ev_packed, hipEventDisableTiming));
  FH_CHECK_HIP(hipEventCreateWithFlags(&h->ev_d2h, hipEventDisableTiming));
  FH_CHECK_HIP(hipEventCreate(&h->ev_done));       // timing-enabled: halo_profile reads it
  FH_CHECK_HIP(hipEventCreate(&h->evt_begin));
  FH_CHECK_HIP(hipEventCreate(&h->evt_ready));
  // "halo_self_rccl": a ONE-rank plan still gets a communicator and sends its interface entries to itself through
  // ncclSend/ncclRecv -- the only way to execute the RCCL calls of this file on a box with a single GPU (hardware preflight)
  if (comm_ranks > 1 || (plan_ranks == 0 && ctx->halo_self_rccl)) {
    if (shared) {
      h->comm = shared;
    } else {
      FH_REQUIRE(id128 != nullptr, "fh_halo_create: null ncclUniqueId");
      ncclUniqueId id;
      memcpy(&id, id128, 128);
      FH_CHECK_NCCL(ncclCommInitRank(&h->comm, nranks, id, rank));
      h->owns_comm = true;
    }
  }
  *out = h;
  return 0;
}

extern "C" int fh_halo_sizes(fh_halo_t h, int* nsend, int* nrecv) {
  FH_REQUIRE(h, "fh_halo_sizes: null plan");
  if (nsend) *nsend = h->nsend;
  if (nrecv) *nrecv = h->nrecv;
  return 0;
}

int fh_halo_update_ptr(fh_halo_t h, double* vd, int n_owned);
int fh_halo_begin_ptr(fh_halo_t h, double* vd, int n_owned, bool prepacked = false);
int fh_halo_end_ptr(fh_halo_t h);

extern "C" int fh_halo_update(fh_halo_t h, fh_vec_t v) {
  FH_REQUIRE(h && v, "fh_halo_update: null argument");
  FH_REQUIRE(v->nghost == h->nrecv, "fh_halo_update: vector has %d ghosts, plan receives %d", v->nghost, h->nrecv);
  return fh_halo_update_ptr(h, v->d, v->n_local);
}

// halo_profile: wall time of a collective as the compute stream sees it (synchronises, measurement only)
struct AllreduceTimer {
  fh_halo_t h;
  bool on;
  explicit AllreduceTimer(fh_halo_t hh) : h(hh), on(hh->ctx->halo_profile != 0) {
    if (on) hipEventRecord(h->evt_begin, h->ctx->stream);
  }
  ~AllreduceTimer() {
    if (!on) return;
    float t = 0.f;
    if (hipEventRecord(h->evt_ready, h->ctx->stream) == hipSuccess && hipEventSynchronize(h->evt_ready) == hipSuccess &&
        hipEventElapsedTime(&t, h->evt_begin, h->evt_ready) == hipSuccess)
      h->allreduce_ms += t;
  }
};

static int host_allreduce(fh_halo_t h, double* d, int n) {
  std::vector<double> buf(n);
  FH_CHECK_HIP(hipMemcpyAsync(buf.data(), d, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->ctx->stream));
  FH_CHECK_HIP(hipStreamSynchronize(h->ctx->stream));
  FH_REQUIRE(h->allreduce(h->user, buf.data(), n) == 0, "host transport: the all-reduce function failed");
  FH_CHECK_HIP(hipMemcpyAsync(d, buf.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, h->ctx->stream));
  FH_CHECK_HIP(hipStreamSynchronize(h->ctx->stream));
  return 0;
}

extern "C" int fh_halo_allreduce_vec(fh_halo_t h, fh_vec_t v) {
  FH_REQUIRE(h && v, "fh_halo_allreduce_vec: null argument");
  if (halo_inert(h) || v->n_local == 0) return 0;
  h->n_allreduce++;
  AllreduceTimer timer(h);
  if (h->allreduce) return host_allreduce(h, v->d, v->n_local);
  FH_CHECK_NCCL(ncclAllReduce(v->d, v->d, v->n_local, ncclDouble, ncclSum, h->comm, h->ctx->stream));
  return 0;
}

// sum over the ranks of the values of a matrix with the SAME pattern on every rank (the replicated level's operator: every rank adds
// its share P^T A_0 P; the reference gets the same sum from MatPtAP over the distributed rows, PetscMatrix.cpp:733-751)
extern "C" int fh_halo_allreduce_mat(fh_halo_t h, fh_mat_t A) {
  if (A) A->val_gen++;
  FH_REQUIRE(h && A, "fh_halo_allreduce_mat: null argument");
  if (halo_inert(h) || A->nnz == 0) return 0;
  h->n_allreduce++;
  AllreduceTimer timer(h);
  A->at_valid = false;
  if (h->allreduce) return host_allreduce(h, A->d_val, A->nnz);
  FH_CHECK_NCCL(ncclAllReduce(A->d_val, A->d_val, A->nnz, ncclDouble, ncclSum, h->comm, h->ctx->stream));
  return 0;
}

int fh_halo_allreduce_ptr(fh_halo_t h, double* d, int n) {
  if (halo_inert(h) || n == 0) return 0;
  h->n_allreduce++;
  AllreduceTimer timer(h);
  if (h->allreduce) return host_allreduce(h, d, n);
  FH_CHECK_NCCL(ncclAllReduce(d, d, n, ncclDouble, ncclSum, h->comm, h->ctx->stream));
  return 0;
}

// start the exchange of the ghosts of vd ([owned | ghost]): pack on the compute stream, transfer on the communication stream.
// Work queued on the compute stream after this call and before fh_halo_end_ptr overlaps with the exchange; it must not read
// the ghost tail.
// the send plan of an exchange, for producers that fill the send buffer themselves (the first smoother sweep writes its interface
// entries straight into it: one launch less per level and cycle); follow with fh_halo_begin_ptr(..., prepacked = true)
void fh_halo_send_plan(fh_halo_t h, const int** send_idx, double** sendbuf, int* nsend) {
  const bool live = h && !halo_inert(h);
  *send_idx = live ? h->d_send_idx : nullptr;
  *sendbuf = live ? h->d_sendbuf : nullptr;
  *nsend = live ? h->nsend : 0;
}

int fh_halo_begin_ptr(fh_halo_t h, double* vd, int n_owned, bool prepacked) {
  if (halo_inert(h)) return 0;
  FH_REQUIRE(!h->pending, "fh_halo_begin: the previous exchange of this plan has not been ended");
  fh_ctx_t c = h->ctx;
  const bool prof = c->halo_profile != 0;
  if (h->nsend && !prepacked) {
    int nb = std::max(1, std::min(fh_div_up(h->nsend, 256), c->num_cu * 4));
    hipLaunchKernelGGL(k_pack, dim3(nb), dim3(256), 0, c->stream, vd, h->d_send_idx, h->d_sendbuf, h->nsend);
    FH_CHECK_HIP(hipGetLastError());
  }
  FH_CHECK_HIP(hipEventRecord(h->ev_packed, c->stream));
  FH_CHECK_HIP(hipStreamWaitEvent(c->comm_stream, h->ev_packed, 0));
  if (prof) FH_CHECK_HIP(hipEventRecord(h->evt_begin, c->comm_stream));
  if (h->exchange) {   // host-staged: pack -> pinned host (communication stream) ; the caller's exchange runs in fh_halo_end_ptr
    if (h->nsend) FH_CHECK_HIP(hipMemcpyAsync(h->h_send, h->d_sendbuf, (size_t)h->nsend * sizeof(double), hipMemcpyDeviceToHost, c->comm_stream));
    FH_CHECK_HIP(hipEventRecord(h->ev_d2h, c->comm_stream));
  } else {
    FH_CHECK_NCCL(ncclGroupStart());
    ncclResult_t bad = ncclSuccess;           // a group that was opened is always closed, also when a call inside it fails
    for (int r = 0; r < h->nranks && bad == ncclSuccess; r++) {     // r == rank: only a self plan ("halo_self_rccl") has entries there
      if (h->send_counts[r]) bad = ncclSend(h->d_sendbuf + h->send_off[r], h->send_counts[r], ncclDouble, r, h->comm, c->comm_stream);
      if (h->recv_counts[r] && bad == ncclSuccess)
        bad = ncclRecv(vd + n_owned + h->recv_off[r], h->recv_counts[r], ncclDouble, r, h->comm, c->comm_stream);
    }
    const ncclResult_t closed = ncclGroupEnd();
    FH_CHECK_NCCL(bad);
    FH_CHECK_NCCL(closed);
    FH_CHECK_HIP(hipEventRecord(h->ev_done, c->comm_stream));
  }
  h->pend_vd = vd;
  h->pend_owned = n_owned;
  h->pending = true;
  h->n_updates++;
  h->bytes_sent += (int64_t)h->nsend * (int64_t)sizeof(double);
  return 0;
}

// consumers of the ghosts queued on the compute stream after this call see the received values
int fh_halo_end_ptr(fh_halo_t h) {
  if (halo_inert(h)) return 0;
  FH_REQUIRE(h->pending, "fh_halo_end: no exchange in flight");
  fh_ctx_t c = h->ctx;
  const bool prof = c->halo_profile != 0;
  h->pending = false;
  if (prof) FH_CHECK_HIP(hipEventRecord(h->evt_ready, c->stream));     // the compute stream has run out of overlapped work here
  if (h->exchange) {
    FH_CHECK_HIP(hipEventSynchronize(h->ev_d2h));                      // only the send buffer: the overlapped kernels keep running
    FH_REQUIRE(h->exchange(h->user, h->h_send, h->send_counts.data(), h->h_recv, h->recv_counts.data()) == 0,
               "host transport: the exchange function failed");
    if (h->nrecv)
      FH_CHECK_HIP(hipMemcpyAsync(h->pend_vd + h->pend_owned, h->h_recv, (size_t)h->nrecv * sizeof(double), hipMemcpyHostToDevice, c->comm_stream));
    FH_CHECK_HIP(hipEventRecord(h->ev_done, c->comm_stream));
  }
  FH_CHECK_HIP(hipStreamWaitEvent(c->stream, h->ev_done, 0));
  if (prof) {
    float t_x = 0.f, t_e = 0.f;
    FH_CHECK_HIP(hipEventSynchronize(h->ev_done));
    FH_CHECK_HIP(hipEventSynchronize(h->evt_ready));
    FH_CHECK_HIP(hipEventElapsedTime(&t_x, h->evt_begin, h->ev_done));
    FH_CHECK_HIP(hipEventElapsedTime(&t_e, h->evt_ready, h->ev_done));   // negative: the ghosts were there before they were needed
    h->exchange_ms += t_x;
    h->exposed_ms += std::max(0.f, std::min(t_e, t_x));
  }
  return 0;
}

int fh_halo_update_ptr(fh_halo_t h, double* vd, int n_owned) {
  FH_TRY(fh_halo_begin_ptr(h, vd, n_owned));
  return fh_halo_end_ptr(h);
}

// operator application on a distributed level: y = op(A, x) with x = [owned | ghost].  What MatMult does for an MPIAIJ matrix
// behind NumericVector::matrix_mult (PetscVector.cpp:203-214): start the ghost scatter, multiply the rows that need no ghost
// while it is in flight (compute stream), wait, multiply the rest.  n_own = owned entries of x (the operator's columns below
// n_own are local).  h == NULL: plain product.
int fh_dev_halo_spmv(fh_halo_t h, fh_mat_t A, double* x, int n_own, double* y, int mode, const double* b, const double* dinv, double omega,
                     bool prepacked) {
  if (!h) return fh_dev_spmv(A, x, y, mode, b, dinv, omega);
  if (!A->ctx->halo_overlap) {
    FH_TRY(fh_halo_begin_ptr(h, x, n_own, prepacked));
    FH_TRY(fh_halo_end_ptr(h));
    return fh_dev_spmv(A, x, y, mode, b, dinv, omega);
  }
  FH_TRY(fh_halo_begin_ptr(h, x, n_own, prepacked));
  const int rc = fh_dev_spmv_part(A, n_own, 0, x, y, mode, b, dinv, omega);
  if (rc) {                       // the exchange in flight is completed all the same: a plan left pending would refuse every later exchange
    std::string msg = fh_last_error();
    fh_halo_end_ptr(h);
    fh_set_error("%s", msg.c_str());
    return rc;
  }
  FH_TRY(fh_halo_end_ptr(h));
  return fh_dev_spmv_part(A, n_own, 1, x, y, mode, b, dinv, omega);
}

extern "C" int fh_spmv_ghosted(fh_mat_t A, fh_halo_t halo, fh_vec_t x, fh_vec_t y, int mode, fh_vec_t b, fh_vec_t dinv, double omega) {
  FH_REQUIRE(A && x && y, "fh_spmv_ghosted: null argument");
  FH_REQUIRE(x->n_local + x->nghost >= A->n, "fh_spmv_ghosted: x has %d entries, matrix has %d columns", x->n_local + x->nghost, A->n);
  FH_REQUIRE(y->n_local >= A->m, "fh_spmv_ghosted: y has %d entries, matrix has %d rows", y->n_local, A->m);
  FH_REQUIRE(!halo || x->nghost == halo->nrecv, "fh_spmv_ghosted: vector has %d ghosts, plan receives %d", x->nghost, halo ? halo->nrecv : 0);
  FH_REQUIRE(mode >= 0 && mode <= 3, "fh_spmv_ghosted: unknown mode %d", mode);
  FH_REQUIRE(mode < 2 || (b && b->n_local >= A->m), "fh_spmv_ghosted: mode %d needs b", mode);
  FH_REQUIRE(mode < 3 || (dinv && dinv->n_local >= A->m && A->m <= A->n), "fh_spmv_ghosted: mode 3 needs dinv and owned rows over [owned | ghost] columns");
  return fh_dev_halo_spmv(halo, A, x->d, x->n_local, y->d, mode, b ? b->d : nullptr, dinv ? dinv->d : nullptr, omega, false);
}

extern "C" int fh_halo_begin(fh_halo_t h, fh_vec_t v) {
  FH_REQUIRE(h && v, "fh_halo_begin: null argument");
  FH_REQUIRE(v->nghost == h->nrecv, "fh_halo_begin: vector has %d ghosts, plan receives %d", v->nghost, h->nrecv);
  return fh_halo_begin_ptr(h, v->d, v->n_local);
}

extern "C" int fh_halo_end(fh_halo_t h) {
  FH_REQUIRE(h, "fh_halo_end: null argument");
  return fh_halo_end_ptr(h);
}

// ghost -> owner, ADD: the accumulated adds to ghost entries (fh_vec_s::d_gacc, filled by the staged add path) travel against the direction of
// the ghost update and are added to the owners' entries, source ranks in ascending order (an owned entry may be a ghost of several ranks:
// one unpack launch per source rank keeps the sum deterministic).  Collective over the plan's ranks.  VecAssemblyBegin/End for the stash of
// off-process ADD_VALUES (PetscVector.cpp:131-153, PetscVector.hpp:595-612).
__global__ __launch_bounds__(256) void k_unpack_add(double* __restrict__ v, const int* __restrict__ idx, const double* __restrict__ buf, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) v[idx[i]] += buf[i];
}

extern "C" int fh_halo_reverse_add(fh_halo_t h, fh_vec_t v) {
  FH_REQUIRE(h && v, "fh_halo_reverse_add: null argument");
  if (halo_inert(h)) return 0;
  FH_REQUIRE(!h->pending, "fh_halo_reverse_add: an exchange of this plan is in flight");
  FH_REQUIRE(v->nghost == h->nrecv, "fh_halo_reverse_add: the vector has %d ghosts, the plan receives %d", v->nghost, h->nrecv);
  fh_ctx_t c = h->ctx;
  if (!v->d_gacc && v->nghost) {          // this rank staged nothing for a ghost: it sends zeros (the exchange is collective)
    FH_CHECK_HIP(hipMalloc(&v->d_gacc, (size_t)v->nghost * sizeof(double)));
    FH_CHECK_HIP(hipMemsetAsync(v->d_gacc, 0, (size_t)v->nghost * sizeof(double), c->stream));
  }
  FH_CHECK_HIP(hipEventRecord(h->ev_packed, c->stream));
  FH_CHECK_HIP(hipStreamWaitEvent(c->comm_stream, h->ev_packed, 0));
  if (h->exchange) {
    if (h->nrecv) FH_CHECK_HIP(hipMemcpyAsync(h->h_recv, v->d_gacc, (size_t)h->nrecv * sizeof(double), hipMemcpyDeviceToHost, c->comm_stream));
    FH_CHECK_HIP(hipStreamSynchronize(c->comm_stream));
    FH_REQUIRE(h->exchange(h->user, h->h_recv, h->recv_counts.data(), h->h_send, h->send_counts.data()) == 0, "host transport: the exchange function failed");
    if (h->nsend) FH_CHECK_HIP(hipMemcpyAsync(h->d_sendbuf, h->h_send, (size_t)h->nsend * sizeof(double), hipMemcpyHostToDevice, c->comm_stream));
  } else {
    FH_CHECK_NCCL(ncclGroupStart());
    ncclResult_t bad = ncclSuccess;
    for (int r = 0; r < h->nranks && bad == ncclSuccess; r++) {
      if (h->recv_counts[r]) bad = ncclSend(v->d_gacc + h->recv_off[r], h->recv_counts[r], ncclDouble, r, h->comm, c->comm_stream);
      if (h->send_counts[r] && bad == ncclSuccess) bad = ncclRecv(h->d_sendbuf + h->send_off[r], h->send_counts[r], ncclDouble, r, h->comm, c->comm_stream);
    }
    const ncclResult_t closed = ncclGroupEnd();
    FH_CHECK_NCCL(bad);
    FH_CHECK_NCCL(closed);
  }
  FH_CHECK_HIP(hipEventRecord(h->ev_done, c->comm_stream));
  FH_CHECK_HIP(hipStreamWaitEvent(c->stream, h->ev_done, 0));
  for (int r = 0; r < h->nranks; r++)
    if (h->send_counts[r]) {
      hipLaunchKernelGGL(k_unpack_add, dim3(fh_div_up(h->send_counts[r], 256)), dim3(256), 0, c->stream, v->d, h->d_send_idx + h->send_off[r],
                         h->d_sendbuf + h->send_off[r], h->send_counts[r]);
      FH_CHECK_HIP(hipGetLastError());
    }
  if (v->nghost) FH_CHECK_HIP(hipMemsetAsync(v->d_gacc, 0, (size_t)v->nghost * sizeof(double), c->stream));
  v->gacc_dirty = false;
  return 0;
}

extern "C" int fh_halo_allreduce_ms(fh_halo_t h, int reset, double* ms) {
  FH_REQUIRE(h && ms, "fh_halo_allreduce_ms: null argument");
  *ms = h->allreduce_ms;
  if (reset) h->allreduce_ms = 0.0;
  return 0;
}

extern "C" int fh_halo_allreduce_count(fh_halo_t h, int reset, int64_t* n) {
  FH_REQUIRE(h, "fh_halo_allreduce_count: null argument");
  if (n) *n = h->n_allreduce;
  if (reset) h->n_allreduce = 0;
  return 0;
}

extern "C" int fh_halo_stats(fh_halo_t h, int reset, int64_t* n_updates, int64_t* bytes_sent, double* exchange_ms, double* exposed_ms) {
  FH_REQUIRE(h, "fh_halo_stats: null argument");
  if (n_updates) *n_updates = h->n_updates;
  if (bytes_sent) *bytes_sent = h->bytes_sent;
  if (exchange_ms) *exchange_ms = h->exchange_ms;
  if (exposed_ms) *exposed_ms = h->exposed_ms;
  if (reset) {
    h->n_updates = h->bytes_sent = 0;
    h->exchange_ms = h->exposed_ms = 0.0;
  }
  return 0;
}

extern "C" int fh_halo_allreduce_sum(fh_halo_t h, double* vals, int n) {
  FH_REQUIRE(h && vals && n >= 0, "fh_halo_allreduce_sum: bad arguments");
  if (halo_inert(h) || n == 0) return 0;
  h->n_allreduce++;
  if (h->allreduce) {
    FH_REQUIRE(h->allreduce(h->user, vals, n) == 0, "host transport: the all-reduce function failed");
    return 0;
  }
  fh_ctx_t c = h->ctx;
  double* d = h->d_scalars;        // 256 doubles live with the plan (dot products, norms); longer arrays (localize_to_all) get their own
  if (n > 256) FH_CHECK_HIP(hipMalloc(&d, (size_t)n * sizeof(double)));
  hipError_t e = hipMemcpyAsync(d, vals, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream);
  ncclResult_t nr = ncclSuccess;
  if (e == hipSuccess) nr = ncclAllReduce(d, d, n, ncclDouble, ncclSum, h->comm, c->stream);
  if (e == hipSuccess && nr == ncclSuccess) e = hipMemcpyAsync(vals, d, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess && nr == ncclSuccess) e = hipStreamSynchronize(c->stream);
  if (n > 256) hipFree(d);
  FH_REQUIRE(nr == ncclSuccess, "fh_halo_allreduce_sum: ncclAllReduce failed: %s", ncclGetErrorString(nr));
  FH_CHECK_HIP(e);
  return 0;
}

extern "C" int fh_halo_rank(fh_halo_t h, int* rank, int* nranks) {
  FH_REQUIRE(h, "fh_halo_rank: null plan");
  if (rank) *rank = h->rank;
  if (nranks) *nranks = h->nranks;
  return 0;
}

extern "C" int fh_halo_destroy(fh_halo_t h) {
  if (!h) return 0;
  hipStreamSynchronize(h->ctx->stream);
  hipStreamSynchronize(h->ctx->comm_stream);
  if (h->comm && h->owns_comm) ncclCommDestroy(h->comm);
  hipFree(h->d_send_idx);
  hipFree(h->d_sendbuf);
  hipFree(h->d_scalars);
  if (h->h_send) hipHostFree(h->h_send);
  if (h->h_recv) hipHostFree(h->h_recv);
  for (hipEvent_t e : {h->ev_packed, h->ev_done, h->ev_d2h, h->evt_begin, h->evt_ready})
    if (e) hipEventDestroy(e);
  delete h;
  return 0;
}
