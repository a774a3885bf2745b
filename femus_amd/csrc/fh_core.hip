// Context, device vectors and BLAS-1 kernels (K13 of SURVEY 2.1) for gfx950.
// Replaces PetscVector (src/03_algebra/00_vectors/PetscVector.cpp) behind the C-ABI of include/femus_hip.h.
// All kernels are HBM-bound streaming kernels: 16-byte accesses per lane, grid-stride, wave64 shuffles
// for reductions, one partial per workgroup, second pass in one workgroup (deterministic order).
#include <cstdlib>
#include "fh_internal.h"
#include <chrono>
#include <cstdlib>
#include <cstdarg>
#include <cmath>

static thread_local char g_err[1024] = "";

void fh_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* fh_last_error(void) { return g_err; }

bool fh_trace_on() {
  static const bool on = [] {
    const char* e = getenv("FEMUS_HIP_TRACE");
    return e && *e && *e != '0';
  }();
  return on;
}

void fh_trace_print(const char* fmt, ...) {
  static const auto t0 = std::chrono::steady_clock::now();
  static double last = 0.0;
  const double now = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  char msg[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(msg, sizeof(msg), fmt, ap);
  va_end(ap);
  fprintf(stderr, "[femus_hip %9.3f s, +%8.3f] %s\n", now, now - last, msg);
  fflush(stderr);
  last = now;
}
extern "C" const char* fh_version(void) { return "femus_hip 0.1 (gfx950)"; }

extern "C" int fh_device_count(int* n) {
  FH_REQUIRE(n, "fh_device_count: null output");
  int ndev = 0;
  const hipError_t e = hipGetDeviceCount(&ndev);
  *n = (e == hipSuccess) ? ndev : 0;
  return 0;
}

extern "C" int fh_init(int device, fh_ctx_t* out) {
  FH_REQUIRE(out != nullptr, "fh_init: null output");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    fh_set_error("fh_init: no HIP device available (%s); libfemus_hip has no CPU fallback",
                 e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    return 1;
  }
  FH_REQUIRE(device >= 0 && device < ndev, "fh_init: device %d out of range (count %d)", device, ndev);
  FH_CHECK_HIP(hipSetDevice(device));
  fh_ctx_t c = new fh_ctx_s();
  c->device = device;
  hipDeviceProp_t prop;
  FH_CHECK_HIP(hipGetDeviceProperties(&prop, device));
  c->num_cu = prop.multiProcessorCount;
  FH_CHECK_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  FH_CHECK_HIP(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
  FH_CHECK_HIP(hipEventCreate(&c->ev0));
  FH_CHECK_HIP(hipEventCreate(&c->ev1));
  FH_CHECK_HIP(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
  if (const char* e = getenv("FEMUS_HIP_COARSE_ND")) c->coarse_nd = atoi(e);   // A/B measurements of the dissected coarse solve (blocks; 0: one dense inverse)
  if (const char* e = getenv("FEMUS_HIP_CARRY")) c->assemble_carry = atoi(e);   // A/B measurements of the carried rows of the fused assembly (0: none)
  if (const char* e = getenv("FEMUS_HIP_POISON")) c->debug_poison = atoi(e);   // whole test suites in poison mode (see debug_poison)
  FH_TRY(fh_reserve_reduction(c, 4096));
  *out = c;
  return 0;
}

extern "C" int fh_finalize(fh_ctx_t c) {
  if (!c) return 0;
  hipSetDevice(c->device);
  hipStreamSynchronize(c->stream);
  hipStreamSynchronize(c->comm_stream);
  if (c->d_red) hipFree(c->d_red);
  if (c->h_red) hipHostFree(c->h_red);
  hipEventDestroy(c->ev0);
  hipEventDestroy(c->ev1);
  hipEventDestroy(c->ev_join);
  hipStreamDestroy(c->stream);
  hipStreamDestroy(c->comm_stream);
  delete c;
  return 0;
}

extern "C" int fh_device_name(fh_ctx_t c, char* buf, int buflen) {
  hipDeviceProp_t prop;
  FH_CHECK_HIP(hipGetDeviceProperties(&prop, c->device));
  snprintf(buf, buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
  return 0;
}

extern "C" int fh_sync(fh_ctx_t c) {
  FH_CHECK_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

extern "C" void* fh_stream(fh_ctx_t c) { return (void*)c->stream; }

// phase markers for kernel traces: a one-thread kernel whose NAME carries the id (k_phase_marker<3>), so that a rocprofv3 --kernel-trace of a
// run can be cut into phases (profiles/summarize.py): which launches of a kernel ran inside the cycle, which were issued one by one ...
template <int ID>
__global__ void k_phase_marker(int* sink) {
  if (sink) *sink = ID;
}
extern "C" int fh_profile_marker(fh_ctx_t c, int id) {
  FH_REQUIRE(c && id >= 0 && id < 16, "fh_profile_marker: id must be 0 .. 15");
#define FH_MARK(I) case I: hipLaunchKernelGGL(k_phase_marker<I>, dim3(1), dim3(1), 0, c->stream, (int*)nullptr); break;
  switch (id) {
    FH_MARK(0) FH_MARK(1) FH_MARK(2) FH_MARK(3) FH_MARK(4) FH_MARK(5) FH_MARK(6) FH_MARK(7)
    FH_MARK(8) FH_MARK(9) FH_MARK(10) FH_MARK(11) FH_MARK(12) FH_MARK(13) FH_MARK(14) FH_MARK(15)
  }
#undef FH_MARK
  FH_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int fh_timer_start(fh_ctx_t c) {
  FH_CHECK_HIP(hipEventRecord(c->ev0, c->stream));
  return 0;
}

extern "C" int fh_timer_stop(fh_ctx_t c, double* ms) {
  FH_CHECK_HIP(hipEventRecord(c->ev1, c->stream));
  FH_CHECK_HIP(hipEventSynchronize(c->ev1));
  float f = 0.f;
  FH_CHECK_HIP(hipEventElapsedTime(&f, c->ev0, c->ev1));
  *ms = (double)f;
  return 0;
}

// ------------------------------------------------------------------------------------------------
// recorded launch sequences: the device-only calls made between fh_graph_begin and fh_graph_end (SpMV family, vector algebra, fh_assemble_*
// after their first call -- nothing that synchronises, allocates or copies to the host) become one hipGraph that fh_graph_launch replays
// ------------------------------------------------------------------------------------------------
struct fh_graph_s {
  fh_ctx_t ctx = nullptr;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
};

extern "C" int fh_graph_begin(fh_ctx_t c) {
  FH_REQUIRE(c, "fh_graph_begin: null context");
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  FH_CHECK_HIP(hipStreamIsCapturing(c->stream, &st));
  FH_REQUIRE(st == hipStreamCaptureStatusNone, "fh_graph_begin: a recording is already open on this context");
  FH_CHECK_HIP(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
  return 0;
}

extern "C" int fh_graph_end(fh_ctx_t c, fh_graph_t* out) {
  FH_REQUIRE(c && out, "fh_graph_end: null argument");
  *out = nullptr;
  hipGraph_t g = nullptr;
  const hipError_t e = hipStreamEndCapture(c->stream, &g);
  if (e != hipSuccess || !g) {
    // an invalidated recording: clear the sticky error; if the runtime leaves the stream in its capture state all the same, the context
    // gets a fresh compute stream (every call reads ctx->stream when it runs; a handle taken with fh_stream() before is stale then)
    (void)hipGetLastError();
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    const hipError_t q = hipStreamIsCapturing(c->stream, &st);
    (void)hipGetLastError();
    if (q != hipSuccess || st != hipStreamCaptureStatusNone) {
      hipStream_t fresh = nullptr;
      if (hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking) == hipSuccess) {
        (void)hipStreamDestroy(c->stream);
        (void)hipGetLastError();
        c->stream = fresh;
      }
    }
  }
  FH_REQUIRE(e == hipSuccess && g, "fh_graph_end: the recording is invalid (%s): a call inside it synchronised, allocated or copied to the host",
             hipGetErrorString(e));
  hipGraphExec_t ex = nullptr;
  if (hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) != hipSuccess) {
    hipGraphDestroy(g);
    FH_REQUIRE(false, "fh_graph_end: the graph could not be instantiated");
  }
  fh_graph_s* G = new fh_graph_s;
  G->ctx = c;
  G->graph = g;
  G->exec = ex;
  *out = G;
  return 0;
}

extern "C" int fh_graph_launch(fh_graph_t g) {
  FH_REQUIRE(g && g->exec, "fh_graph_launch: null graph");
  FH_CHECK_HIP(hipGraphLaunch(g->exec, g->ctx->stream));
  return 0;
}

extern "C" int fh_graph_destroy(fh_graph_t g) {
  if (!g) return 0;
  hipStreamSynchronize(g->ctx->stream);
  if (g->exec) hipGraphExecDestroy(g->exec);
  if (g->graph) hipGraphDestroy(g->graph);
  delete g;
  return 0;
}

extern "C" int fh_set_option(fh_ctx_t c, const char* name, double value) {
  c->opt_gen++;
  if (!strcmp(name, "spmv_tile")) c->spmv_tile = (int)value;
  else if (!strcmp(name, "spmv_xcd_remap")) c->spmv_xcd_remap = (int)value;
  else if (!strcmp(name, "spmv_kernel")) c->spmv_kernel = (int)value;
  else if (!strcmp(name, "spmv_nt")) c->spmv_nt = (int)value;
  else if (!strcmp(name, "spmv_share")) c->spmv_share = (int)value;
  else if (!strcmp(name, "spmv_threads")) c->spmv_threads = (int)value;
  else if (!strcmp(name, "assemble_emap")) c->assemble_emap = (int)value;
  else if (!strcmp(name, "asm_debug")) c->asm_debug = (int)value;
  else if (!strcmp(name, "debug_poison")) c->debug_poison = (int)value;
  else if (!strcmp(name, "assemble_two_pass")) c->assemble_two_pass = (int)value;
  else if (!strcmp(name, "assemble_sym")) c->assemble_sym = (int)value;
  else if (!strcmp(name, "assemble_mfma")) c->assemble_mfma = (int)value;
  else if (!strcmp(name, "assemble_kpad")) c->assemble_kpad = (int)value;
  else if (!strcmp(name, "assemble_rows2")) c->assemble_rows2 = (int)value;
  else if (!strcmp(name, "assemble_sf")) c->assemble_sf = (int)value;
  else if (!strcmp(name, "assemble_rows_nt")) c->assemble_rows_nt = (int)value;
  else if (!strcmp(name, "assemble_fused")) c->assemble_fused = (int)value;
  else if (!strcmp(name, "assemble_sf_grid")) c->assemble_sf_grid = std::max(1, (int)value);
  else if (!strcmp(name, "assemble_sumfac")) c->assemble_sumfac = (int)value;
  else if (!strcmp(name, "galerkin_mfma")) c->galerkin_mfma = (int)value;
  else if (!strcmp(name, "gj_block")) c->gj_block = (int)value;
  else if (!strcmp(name, "vanka_persistent")) c->vanka_persistent = (int)value;
  else if (!strcmp(name, "coarse_reduce")) c->coarse_reduce = (int)value;
  else if (!strcmp(name, "coarse_nd")) c->coarse_nd = (int)value;
  else if (!strcmp(name, "coarse_nd_min")) c->coarse_nd_min = (int)value;
  else if (!strcmp(name, "coarse_direct")) c->coarse_direct = (int)value;
  else if (!strcmp(name, "gmres_device")) c->gmres_device = (int)value;
  else if (!strcmp(name, "galerkin_macro")) c->galerkin_macro = (int)value;
  else if (!strcmp(name, "vanka_fused")) c->vanka_fused = (int)value;
  else if (!strcmp(name, "coarse_direct_min")) c->coarse_direct_min = (int)value;
  else if (!strcmp(name, "coarse_nd_streams")) c->coarse_nd_streams = (int)value;
  else if (!strcmp(name, "patch_invert_lds")) c->patch_invert_lds = (int)value;
  else if (!strcmp(name, "gj_mfma")) c->gj_mfma = (int)value;
  else if (!strcmp(name, "gj_symmetric")) c->gj_symmetric = (int)value;
  else if (!strcmp(name, "assemble_affine")) c->assemble_affine = (int)value;
  else if (!strcmp(name, "assemble_carry")) c->assemble_carry = (int)value;
  else if (!strcmp(name, "tri_runs")) c->tri_runs = (int)value;
  else if (!strcmp(name, "ilu_ahead")) c->ilu_ahead = (int)value;
  else if (!strcmp(name, "use_graph")) c->use_graph = (int)value;
  else if (!strcmp(name, "mg_reuse_graph")) c->mg_reuse_graph = (int)value;
  else if (!strcmp(name, "spgemm_slot_map")) c->spgemm_slot_map = (int)value;
  else if (!strcmp(name, "device_setup")) c->device_setup = (int)value;
  else if (!strcmp(name, "spgemm_device_symbolic")) c->spgemm_device_symbolic = (int)value;
  else if (!strcmp(name, "halo_overlap")) c->halo_overlap = (int)value;
  else if (!strcmp(name, "halo_profile")) c->halo_profile = (int)value;
  else if (!strcmp(name, "halo_self_rccl")) c->halo_self_rccl = (int)value;
  else {
    fh_set_error("fh_set_option: unknown option '%s'", name);
    return 2;
  }
  return 0;
}

int fh_reserve_reduction(fh_ctx_t c, size_t n) {
  if (n <= c->red_cap) return 0;
  if (c->d_red) FH_CHECK_HIP(hipFree(c->d_red));
  if (c->h_red) FH_CHECK_HIP(hipHostFree(c->h_red));
  FH_CHECK_HIP(hipMalloc(&c->d_red, n * sizeof(double)));
  FH_CHECK_HIP(hipHostMalloc(&c->h_red, n * sizeof(double)));
  c->red_cap = n;
  return 0;
}

// ------------------------------------------------------------------------------------------------
// vectors
// ------------------------------------------------------------------------------------------------
extern "C" int fh_vec_create(fh_ctx_t c, int n_global, int n_local, int first_local, const int* ghost_idx, int nghost,
                             fh_vec_t* out) {
  FH_REQUIRE(c && out, "fh_vec_create: null argument");
  FH_REQUIRE(n_local >= 0 && nghost >= 0 && n_global >= n_local, "fh_vec_create: bad sizes");
  fh_vec_t v = new fh_vec_s();
  v->ctx = c;
  v->n_global = n_global;
  v->n_local = n_local;
  v->first_local = first_local;
  v->nghost = nghost;
  size_t tot = (size_t)n_local + nghost;
  FH_CHECK_HIP(hipMalloc(&v->d, (tot + 2) * sizeof(double)));
  FH_CHECK_HIP(hipMemsetAsync(v->d, 0, (tot + 2) * sizeof(double), c->stream));
  if (nghost > 0) {
    v->ghost_idx.assign(ghost_idx, ghost_idx + nghost);
    FH_CHECK_HIP(hipMalloc(&v->d_ghost_idx, nghost * sizeof(int)));
    FH_CHECK_HIP(hipMemcpy(v->d_ghost_idx, ghost_idx, nghost * sizeof(int), hipMemcpyHostToDevice));
  }
  *out = v;
  return 0;
}

extern "C" int fh_vec_duplicate(fh_vec_t s, fh_vec_t* out) {
  return fh_vec_create(s->ctx, s->n_global, s->n_local, s->first_local, s->ghost_idx.data(), s->nghost, out);
}

extern "C" int fh_vec_destroy(fh_vec_t v) {
  if (!v) return 0;
  hipStreamSynchronize(v->ctx->stream);
  fh_stage_free(v->stage);
  if (v->d) hipFree(v->d);
  if (v->d_ghost_idx) hipFree(v->d_ghost_idx);
  if (v->d_gacc) hipFree(v->d_gacc);
  delete v;
  return 0;
}

extern "C" int fh_vec_size(fh_vec_t v, int* n_global, int* n_local, int* first_local, int* nghost) {
  if (n_global) *n_global = v->n_global;
  if (n_local) *n_local = v->n_local;
  if (first_local) *first_local = v->first_local;
  if (nghost) *nghost = v->nghost;
  return 0;
}

extern "C" int fh_vec_set_first(fh_vec_t v, int first_local) {
  FH_REQUIRE(v && first_local >= 0 && (int64_t)first_local + v->n_local <= v->n_global, "fh_vec_set_first: offset %d does not fit", first_local);
  v->first_local = first_local;
  return 0;
}

extern "C" double* fh_vec_dev_ptr(fh_vec_t v) { return v->d; }

static inline int stream_grid(fh_ctx_t c, int64_t n, int per_thread) {
  int64_t blocks = (n + 256 * (int64_t)per_thread - 1) / (256 * (int64_t)per_thread);
  int64_t cap = (int64_t)c->num_cu * 8;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

__global__ __launch_bounds__(256) void k_fill(double* __restrict__ y, double s, int n) {
  int i = (blockIdx.x * 256 + threadIdx.x) * 2;
  const int stride = gridDim.x * 512;
  for (; i + 1 < n; i += stride) *reinterpret_cast<double2*>(y + i) = make_double2(s, s);
  if (i < n) y[i] = s;
}

// y = a*x + b*y  (covers axpy, aypx, copy, scale)
__global__ __launch_bounds__(256) void k_axpby(double* __restrict__ y, const double* __restrict__ x, double a, double b, int n) {
  int i = (blockIdx.x * 256 + threadIdx.x) * 2;
  const int stride = gridDim.x * 512;
  for (; i + 1 < n; i += stride) {
    double2 xv = *reinterpret_cast<const double2*>(x + i);
    double2 yv = *reinterpret_cast<const double2*>(y + i);
    // BLAS semantics: with b == 0 the old y is not referenced (0 * NaN of an uninitialised buffer would be NaN)
    yv.x = (b == 0.0) ? a * xv.x : a * xv.x + b * yv.x;
    yv.y = (b == 0.0) ? a * xv.y : a * xv.y + b * yv.y;
    *reinterpret_cast<double2*>(y + i) = yv;
  }
  if (i < n) y[i] = (b == 0.0) ? a * x[i] : a * x[i] + b * y[i];
}

template <int OP>   // 0: scale, 1: shift, 2: abs
__global__ __launch_bounds__(256) void k_unary(double* __restrict__ y, double s, int n) {
  int i = blockIdx.x * 256 + threadIdx.x;
  const int stride = gridDim.x * 256;
  for (; i < n; i += stride) {
    const double v = y[i];
    double r;
    if (OP == 0) r = v * s;
    else if (OP == 1) r = v + s;
    else r = __builtin_fabs(v);
    y[i] = r;
  }
}

__global__ __launch_bounds__(256) void k_pmult(double* __restrict__ w, const double* __restrict__ a, const double* __restrict__ b, int n) {
  int i = blockIdx.x * 256 + threadIdx.x;
  const int stride = gridDim.x * 256;
  for (; i < n; i += stride) w[i] = a[i] * b[i];
}

__global__ void k_scatter_set(double* __restrict__ y, const int* __restrict__ idx, const double* __restrict__ v, int n, int add) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    if (add) atomicAdd(&y[idx[i]], v[i]);   // duplicates allowed (add_vector_blocked)
    else y[idx[i]] = v[i];
  }
}

__global__ void k_gather(const double* __restrict__ y, const int* __restrict__ idx, double* __restrict__ v, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = y[idx[i]];
}

extern "C" int fh_vec_zero(fh_vec_t v) {
  FH_CHECK_HIP(hipMemsetAsync(v->d, 0, ((size_t)v->n_local + v->nghost) * sizeof(double), v->ctx->stream));
  return 0;
}

extern "C" int fh_vec_fill(fh_vec_t v, double s) {
  int n = v->n_local + v->nghost;
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_fill, dim3(stream_grid(v->ctx, n, 2)), dim3(256), 0, v->ctx->stream, v->d, s, n);
  FH_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int fh_vec_copy(fh_vec_t dst, fh_vec_t src) {
  FH_REQUIRE(dst->n_local == src->n_local, "fh_vec_copy: size mismatch %d vs %d", dst->n_local, src->n_local);
  size_t n = (size_t)src->n_local + (dst->nghost == src->nghost ? src->nghost : 0);
  FH_CHECK_HIP(hipMemcpyAsync(dst->d, src->d, n * sizeof(double), hipMemcpyDeviceToDevice, dst->ctx->stream));
  return 0;
}

extern "C" int fh_vec_upload(fh_vec_t v, const double* h) {
  FH_CHECK_HIP(hipMemcpyAsync(v->d, h, (size_t)v->n_local * sizeof(double), hipMemcpyHostToDevice, v->ctx->stream));
  FH_CHECK_HIP(hipStreamSynchronize(v->ctx->stream));
  return 0;
}

extern "C" int fh_vec_download(fh_vec_t v, double* h) {
  FH_CHECK_HIP(hipMemcpyAsync(h, v->d, (size_t)v->n_local * sizeof(double), hipMemcpyDeviceToHost, v->ctx->stream));
  FH_CHECK_HIP(hipStreamSynchronize(v->ctx->stream));
  return 0;
}

// global index -> local slot (owned or ghost); -1 when not present on this rank
static int vec_local_slot(fh_vec_t v, int g) {
  if (g >= v->first_local && g < v->first_local + v->n_local) return g - v->first_local;
  for (int k = 0; k < v->nghost; k++)
    if (v->ghost_idx[k] == g) return v->n_local + k;
  return -1;
}

static int vec_indexed(fh_vec_t v, int n, const int* idx, const double* vals_in, double* vals_out, int mode) {
  if (n <= 0) return 0;
  fh_ctx_t c = v->ctx;
  std::vector<int> loc(n);
  for (int i = 0; i < n; i++) {
    loc[i] = vec_local_slot(v, idx[i]);
    FH_REQUIRE(loc[i] >= 0, "vector index %d is neither owned nor a ghost on this rank", idx[i]);
  }
  int* d_idx = nullptr;
  double* d_val = nullptr;
  FH_CHECK_HIP(hipMalloc(&d_idx, n * sizeof(int)));
  FH_CHECK_HIP(hipMalloc(&d_val, n * sizeof(double)));
  FH_CHECK_HIP(hipMemcpyAsync(d_idx, loc.data(), n * sizeof(int), hipMemcpyHostToDevice, c->stream));
  if (mode <= 1) {
    FH_CHECK_HIP(hipMemcpyAsync(d_val, vals_in, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_scatter_set, dim3(fh_div_up(n, 256)), dim3(256), 0, c->stream, v->d, d_idx, d_val, n, mode);
  } else {
    hipLaunchKernelGGL(k_gather, dim3(fh_div_up(n, 256)), dim3(256), 0, c->stream, v->d, d_idx, d_val, n);
    FH_CHECK_HIP(hipMemcpyAsync(vals_out, d_val, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  }
  FH_CHECK_HIP(hipGetLastError());
  FH_CHECK_HIP(hipStreamSynchronize(c->stream));
  hipFree(d_idx);
  hipFree(d_val);
  return 0;
}

extern "C" int fh_vec_set_values(fh_vec_t v, int n, const int* idx, const double* vals) { return vec_indexed(v, n, idx, vals, nullptr, 0); }
// immediate form of the staged add (fh_stage.hip): values added in the order given, duplicates included
extern "C" int fh_vec_add_values(fh_vec_t v, int n, const int* idx, const double* vals) {
  FH_TRY(fh_vec_stage_values(v, n, idx, vals));
  return fh_vec_flush(v);
}
extern "C" int fh_vec_get_values(fh_vec_t v, int n, const int* idx, double* vals) { return vec_indexed(v, n, idx, nullptr, vals, 2); }

extern "C" int fh_vec_axpy(fh_vec_t y, double a, fh_vec_t x) {
  FH_REQUIRE(y->n_local == x->n_local, "fh_vec_axpy: size mismatch");
  int n = y->n_local;
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_axpby, dim3(stream_grid(y->ctx, n, 2)), dim3(256), 0, y->ctx->stream, y->d, x->d, a, 1.0, n);
  FH_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int fh_vec_aypx(fh_vec_t y, double a, fh_vec_t x) {
  FH_REQUIRE(y->n_local == x->n_local, "fh_vec_aypx: size mismatch");
  int n = y->n_local;
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_axpby, dim3(stream_grid(y->ctx, n, 2)), dim3(256), 0, y->ctx->stream, y->d, x->d, 1.0, a, n);
  FH_CHECK_HIP(hipGetLastError());
  return 0;
}

static int vec_unary(fh_vec_t v, int op, double s) {
  int n = v->n_local;
  if (n == 0) return 0;
  dim3 grid(stream_grid(v->ctx, n, 1));
  if (op == 0) hipLaunchKernelGGL(k_unary<0>, grid, dim3(256), 0, v->ctx->stream, v->d, s, n);
  else if (op == 1) hipLaunchKernelGGL(k_unary<1>, grid, dim3(256), 0, v->ctx->stream, v->d, s, n);
  else hipLaunchKernelGGL(k_unary<2>, grid, dim3(256), 0, v->ctx->stream, v->d, s, n);
  FH_CHECK_HIP(hipGetLastError());
  return 0;
}
extern "C" int fh_vec_scale(fh_vec_t v, double s) { return vec_unary(v, 0, s); }
extern "C" int fh_vec_shift(fh_vec_t v, double s) { return vec_unary(v, 1, s); }
extern "C" int fh_vec_abs(fh_vec_t v) { return vec_unary(v, 2, 0.0); }

extern "C" int fh_vec_pointwise_mult(fh_vec_t w, fh_vec_t a, fh_vec_t b) {
  FH_REQUIRE(w->n_local == a->n_local && w->n_local == b->n_local, "fh_vec_pointwise_mult: size mismatch");
  int n = w->n_local;
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_pmult, dim3(stream_grid(w->ctx, n, 1)), dim3(256), 0, w->ctx->stream, w->d, a->d, b->d, n);
  FH_CHECK_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------
// reductions: op 0 sum(x*y), 1 sum|x|, 2 max|x|, 3 sum x, 4 min x, 5 max x
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double red_combine(int op, double a, double b) {
  if (op == 2 || op == 5) return fmax(a, b);
  if (op == 4) return fmin(a, b);
  return a + b;
}
__device__ __forceinline__ double red_identity(int op) {
  if (op == 4) return INFINITY;
  if (op == 5) return -INFINITY;
  return 0.0;
}

__device__ __forceinline__ double block_reduce(double v, int op) {
  __shared__ double sm[4];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = red_combine(op, v, __shfl_down(v, off, 64));
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) sm[w] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    v = sm[0];
    for (int k = 1; k < (int)(blockDim.x >> 6); k++) v = red_combine(op, v, sm[k]);
  }
  return v;
}

__global__ __launch_bounds__(256) void k_reduce1(const double* __restrict__ x, const double* __restrict__ y, int n, int op,
                                                 double* __restrict__ part) {
  double acc = red_identity(op);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    double a = x[i];
    double t = (op == 0) ? a * y[i] : (op == 1 || op == 2) ? __builtin_fabs(a) : a;
    acc = red_combine(op, acc, t);
  }
  acc = block_reduce(acc, op);
  if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_reduce2(double* __restrict__ part, int nb, int op) {
  double acc = red_identity(op);
  for (int i = threadIdx.x; i < nb; i += 256) acc = red_combine(op, acc, part[i]);
  acc = block_reduce(acc, op);
  if (threadIdx.x == 0) part[0] = acc;
}

static int vec_reduce(fh_vec_t x, fh_vec_t y, int op, double* out) {
  fh_ctx_t c = x->ctx;
  int n = x->n_local;
  if (n == 0) {
    *out = (op == 4) ? INFINITY : (op == 5) ? -INFINITY : 0.0;
    return 0;
  }
  int nb = stream_grid(c, n, 4);
  FH_TRY(fh_reserve_reduction(c, nb));
  hipLaunchKernelGGL(k_reduce1, dim3(nb), dim3(256), 0, c->stream, x->d, y ? y->d : x->d, n, op, c->d_red);
  hipLaunchKernelGGL(k_reduce2, dim3(1), dim3(256), 0, c->stream, c->d_red, nb, op);
  FH_CHECK_HIP(hipGetLastError());
  FH_CHECK_HIP(hipMemcpyAsync(c->h_red, c->d_red, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  FH_CHECK_HIP(hipStreamSynchronize(c->stream));
  *out = c->h_red[0];
  return 0;
}

extern "C" int fh_vec_dot(fh_vec_t x, fh_vec_t y, double* out) {
  FH_REQUIRE(x->n_local == y->n_local, "fh_vec_dot: size mismatch");
  return vec_reduce(x, y, 0, out);
}

extern "C" int fh_vec_norm(fh_vec_t x, int kind, double* out) {
  if (kind == 2) {
    double s;
    FH_TRY(vec_reduce(x, x, 0, &s));
    *out = sqrt(s);
    return 0;
  }
  if (kind == 1) return vec_reduce(x, nullptr, 1, out);
  if (kind == 0) return vec_reduce(x, nullptr, 2, out);
  fh_set_error("fh_vec_norm: unknown kind %d", kind);
  return 2;
}

extern "C" int fh_vec_reduce(fh_vec_t x, int kind, double* out) {
  FH_REQUIRE(kind >= 0 && kind <= 2, "fh_vec_reduce: unknown kind %d", kind);
  return vec_reduce(x, nullptr, 3 + kind, out);
}
