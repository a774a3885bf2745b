// Device CSR matrices and the SpMV family (K7, K8, K9, K10 of SURVEY 2.1) for gfx950.
// Replaces PetscMatrix / MatMult / MatMultAdd / MatMultTranspose / MatZeroRows
// (src/03_algebra/01_matrices/PetscMatrix.cpp, src/03_algebra/00_vectors/PetscVector.cpp:182-247).
//
// SpMV design ("CSR-stream", HBM-bound): rows are grouped at setup into row blocks of <= TILE non-zeros.
// One 256-thread workgroup per block streams its contiguous slice of (val, col) with fully coalesced
// 16 B + 8 B loads per lane, multiplies by the gathered x entries (L1/L2/Infinity-Cache hits: x is
// 17 MB on the 64^3 level) and parks the products in LDS; a wave-level segmented reduction (G lanes per
// row, shuffles) then produces the rows.  The epilogue fuses residual / Jacobi-sweep forms, so a smoother
// sweep moves the matrix exactly once.  Blocks are renumbered so that consecutive row blocks (which share
// x lines) run on the same XCD and hit the same 4 MiB L2.
#include "fh_internal.h"
#include <atomic>
#include <algorithm>
#include <cmath>
#include <thread>

// ------------------------------------------------------------------------------------------------
// creation / destruction
// ------------------------------------------------------------------------------------------------
extern "C" int fh_mat_create_csr(fh_ctx_t c, int m, int n, const int* rowptr, const int* col, const double* val, fh_mat_t* out) {
  FH_GUARD_BEGIN
  FH_REQUIRE(c && out && rowptr, "fh_mat_create_csr: null argument");
  FH_REQUIRE(m >= 0 && n >= 0 && rowptr[0] == 0, "fh_mat_create_csr: bad sizes");
  const int nnz = rowptr[m];
  FH_REQUIRE(nnz == 0 || col != nullptr, "fh_mat_create_csr: null column array");
  fh_mat_t A = new fh_mat_s();
  static std::atomic<uint64_t> next_uid{1};
  A->uid = next_uid++;
  A->ctx = c;
  A->m = m;
  A->n = n;
  A->nnz = nnz;
  A->h_rowptr.assign(rowptr, rowptr + m + 1);
  A->h_col.assign(col, col + nnz);
  int maxrow = 0;
  for (int i = 0; i < m; i++) {
    FH_REQUIRE(rowptr[i + 1] >= rowptr[i], "fh_mat_create_csr: rowptr not monotone at row %d", i);
    maxrow = std::max(maxrow, rowptr[i + 1] - rowptr[i]);
    for (int k = rowptr[i]; k < rowptr[i + 1]; k++) {
      FH_REQUIRE(col[k] >= 0 && col[k] < n, "fh_mat_create_csr: column %d out of range in row %d", col[k], i);
      FH_REQUIRE(k == rowptr[i] || col[k] > col[k - 1], "fh_mat_create_csr: columns of row %d not strictly sorted", i);
    }
  }
  A->max_row = maxrow;
  // +2 padding: the streaming kernel reads (val,col) in aligned pairs and may touch one element past the end
  FH_CHECK_HIP(hipMalloc(&A->d_rowptr, ((size_t)m + 1) * sizeof(int)));
  FH_CHECK_HIP(hipMalloc(&A->d_col, ((size_t)nnz + 2) * sizeof(int)));
  FH_CHECK_HIP(hipMalloc(&A->d_val, ((size_t)nnz + 2) * sizeof(double)));
  FH_CHECK_HIP(hipMemsetAsync(A->d_col, 0, ((size_t)nnz + 2) * sizeof(int), c->stream));
  FH_CHECK_HIP(hipMemsetAsync(A->d_val, 0, ((size_t)nnz + 2) * sizeof(double), c->stream));
  FH_CHECK_HIP(hipMemcpyAsync(A->d_rowptr, rowptr, ((size_t)m + 1) * sizeof(int), hipMemcpyHostToDevice, c->stream));
  if (nnz) FH_CHECK_HIP(hipMemcpyAsync(A->d_col, col, (size_t)nnz * sizeof(int), hipMemcpyHostToDevice, c->stream));
  if (val && nnz) FH_CHECK_HIP(hipMemcpyAsync(A->d_val, val, (size_t)nnz * sizeof(double), hipMemcpyHostToDevice, c->stream));
  FH_CHECK_HIP(hipStreamSynchronize(c->stream));
  FH_TRY(fh_mat_build_rowblocks(A, c->spmv_tile));
  *out = A;
  return 0;
  FH_GUARD_END("fh_mat_create_csr")
}

int fh_mat_fetch_host_cols(fh_mat_t A) {
  A->h_col.resize((size_t)A->nnz);
  if (A->nnz) {
    FH_CHECK_HIP(hipMemcpyAsync(A->h_col.data(), A->d_col, (size_t)A->nnz * sizeof(int), hipMemcpyDeviceToHost, A->ctx->stream));
    FH_CHECK_HIP(hipStreamSynchronize(A->ctx->stream));
  }
  return 0;
}

// A matrix whose pattern is produced ON THE DEVICE: row pointers (scanned on the host) are given, the column array is allocated and left to
// the caller's kernels; values start at zero.  The host column copy is fetched only if host code asks for it (fh_hcol).  The caller
// finishes with fh_mat_build_rowblocks once the columns are written.
int fh_mat_alloc_device_pattern(fh_ctx_t c, int m, int n, std::vector<int>&& rp, fh_mat_t* out) {
  fh_mat_t A = new fh_mat_s();
  static std::atomic<uint64_t> next_uid{(uint64_t)1 << 40};        // (apart from the counter of fh_mat_create_csr)
  A->uid = next_uid++;
  A->ctx = c;
  A->m = m;
  A->n = n;
  A->h_rowptr = std::move(rp);
  A->nnz = A->h_rowptr[m];
  int maxrow = 0;
  for (int r = 0; r < m; r++) maxrow = std::max(maxrow, A->h_rowptr[r + 1] - A->h_rowptr[r]);
  A->max_row = maxrow;
  *out = A;
  FH_CHECK_HIP(hipMalloc(&A->d_rowptr, ((size_t)m + 1) * sizeof(int)));
  FH_CHECK_HIP(hipMalloc(&A->d_col, ((size_t)A->nnz + 2) * sizeof(int)));
  FH_CHECK_HIP(hipMalloc(&A->d_val, ((size_t)A->nnz + 2) * sizeof(double)));
  FH_CHECK_HIP(hipMemsetAsync(A->d_col + A->nnz, 0, 2 * sizeof(int), c->stream));
  FH_CHECK_HIP(hipMemsetAsync(A->d_val, 0, ((size_t)A->nnz + 2) * sizeof(double), c->stream));
  FH_CHECK_HIP(hipMemcpyAsync(A->d_rowptr, A->h_rowptr.data(), ((size_t)m + 1) * sizeof(int), hipMemcpyHostToDevice, c->stream));
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// Finite-element pattern on the DEVICE (round 4; LinearEquation::GetSparsityPatternSize + SparseMatrix::init, LinearEquation.cpp:407-548): row r
// holds the dofs of all elements around dof r.  node -> element lists by a counting pass, then one wave per row: the <= 1024 candidate
// columns into LDS, bitonic sort, distinct keys -> length (first launch) / columns (second launch).  The host only scans the row lengths;
// the column array never visits it unless host code asks (fh_hcol).  Same pattern as fh_pattern_from_elements (sorted, unique).
// ------------------------------------------------------------------------------------------------------------------
constexpr int PE_CAP = 1024;
__global__ __launch_bounds__(256) void k_pe_count(size_t n, const int* __restrict__ elem_dof, int m, int ncols, int* __restrict__ cnt, int* __restrict__ err) {
  const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  const int d = elem_dof[k];
  if (d < 0 || d >= ncols) { atomicExch(err, 1); return; }
  if (d < m) atomicAdd(&cnt[d], 1);
}
__global__ __launch_bounds__(256) void k_pe_fill(size_t n, int nloc, const int* __restrict__ elem_dof, int m, int* __restrict__ cur, int* __restrict__ adj) {
  const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  const int d = elem_dof[k];
  if (d >= 0 && d < m) adj[atomicAdd(&cur[d], 1)] = (int)(k / nloc);
}
template <bool FILL>
__global__ __launch_bounds__(256) void k_pe_rows(int m, const int* __restrict__ aptr, const int* __restrict__ adj, const int* __restrict__ elem_dof, int nloc,
                                                 const int* __restrict__ rowptr, int* __restrict__ rowlen, int* __restrict__ col, int* __restrict__ err) {
  __shared__ int keys[4][PE_CAP];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + wave;
  if (r >= m) return;
  int* key = keys[wave];
  const int a0 = aptr[r], na = aptr[r + 1] - a0, n = na * nloc;
  if (n > PE_CAP) {
    if (lane == 0) atomicExch(err, 2);
    return;
  }
  int np = 64;
  while (np < n) np <<= 1;
  for (int k = lane; k < np; k += 64) key[k] = k < n ? elem_dof[(size_t)adj[a0 + k / nloc] * nloc + k % nloc] : 0x7fffffff;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  for (int size = 2; size <= np; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = lane; t < (np >> 1); t += 64) {
        const int lo = ((t / stride) * (stride << 1)) + (t % stride), hi = lo + stride;
        const bool up = ((lo & size) == 0);
        const int a = key[lo], c = key[hi];
        if ((a > c) == up) {
          key[lo] = c;
          key[hi] = a;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  const int per = np >> 6, k0 = lane * per;
  int mine = 0;
  for (int k = k0; k < k0 + per && k < n; k++) mine += (k == 0 || key[k] != key[k - 1]) ? 1 : 0;
  int incl = mine;
  for (int d = 1; d < 64; d <<= 1) {
    const int v = __shfl_up(incl, d, 64);
    if (lane >= d) incl += v;
  }
  const int total = __shfl(incl, 63, 64);
  if (!FILL) {
    if (lane == 0) rowlen[r] = total;
    return;
  }
  int o = rowptr[r] + incl - mine;
  for (int k = k0; k < k0 + per && k < n; k++)
    if (k == 0 || key[k] != key[k - 1]) col[o++] = key[k];
}

// elem_dof: host array [nel * nloc], or null when dev_elem_dof -- the same array already in device memory (not owned) -- is given
static int mat_create_from_elements_impl(fh_ctx_t c, int nel, int nloc, const int* elem_dof, const int* dev_elem_dof, int m, int n, fh_mat_t* out) {
  FH_GUARD_BEGIN
  FH_REQUIRE(c && out && nel >= 0 && nloc > 0 && m >= 0 && n >= m && (elem_dof || dev_elem_dof || nel == 0), "fh_mat_create_from_elements: bad arguments");
  const size_t ne = (size_t)nel * nloc;
  int *d_ed = nullptr, *d_ed_own = nullptr, *d_cnt = nullptr, *d_adj = nullptr, *d_err = nullptr, *d_len = nullptr;
  auto cleanup = [&]() {
    for (int* p : {d_ed_own, d_cnt, d_adj, d_err, d_len})
      if (p) hipFree(p);
  };
  std::vector<int> fetched;        // host copy of a device-only table, made only if the host builder has to serve a row
  if (dev_elem_dof) {
    d_ed = const_cast<int*>(dev_elem_dof);
  } else {
    FH_CHECK_HIP(hipMalloc(&d_ed_own, std::max<size_t>(ne, 1) * sizeof(int)));
    d_ed = d_ed_own;
  }
  FH_CHECK_HIP(hipMalloc(&d_cnt, ((size_t)m + 2) * sizeof(int)));
  FH_CHECK_HIP(hipMalloc(&d_err, sizeof(int)));
  FH_CHECK_HIP(hipMalloc(&d_len, ((size_t)m + 1) * sizeof(int)));
  if (ne && !dev_elem_dof) FH_CHECK_HIP(hipMemcpyAsync(d_ed, elem_dof, ne * sizeof(int), hipMemcpyHostToDevice, c->stream));
  FH_CHECK_HIP(hipMemsetAsync(d_cnt, 0, ((size_t)m + 2) * sizeof(int), c->stream));
  FH_CHECK_HIP(hipMemsetAsync(d_err, 0, sizeof(int), c->stream));
  const unsigned gb = (unsigned)((ne + 255) / 256);
  if (ne) hipLaunchKernelGGL(k_pe_count, dim3(gb), dim3(256), 0, c->stream, ne, d_ed, m, n, d_cnt, d_err);
  std::vector<int> aptr((size_t)m + 1, 0);
  if (m) FH_CHECK_HIP(hipMemcpyAsync(aptr.data() + 1, d_cnt, (size_t)m * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  int err = 0;
  FH_CHECK_HIP(hipMemcpyAsync(&err, d_err, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  FH_CHECK_HIP(hipStreamSynchronize(c->stream));
  if (err) {
    cleanup();
    fh_set_error("fh_mat_create_from_elements: a dof of an element is out of range");
    return 2;
  }
  int64_t tot = 0;
  for (int r = 0; r < m; r++) {
    tot += aptr[r + 1];
    aptr[r + 1] = (int)tot;
  }
  FH_REQUIRE(tot < 2147483647ll, "fh_mat_create_from_elements: adjacency overflows int32");
  int* d_aptr = nullptr;
  FH_CHECK_HIP(hipMalloc(&d_aptr, ((size_t)m + 1) * sizeof(int)));
  FH_CHECK_HIP(hipMalloc(&d_adj, std::max<size_t>((size_t)tot, 1) * sizeof(int)));
  FH_CHECK_HIP(hipMemcpyAsync(d_aptr, aptr.data(), ((size_t)m + 1) * sizeof(int), hipMemcpyHostToDevice, c->stream));
  FH_CHECK_HIP(hipMemcpyAsync(d_cnt, aptr.data(), (size_t)m * sizeof(int), hipMemcpyHostToDevice, c->stream));        // cursors
  if (ne) hipLaunchKernelGGL(k_pe_fill, dim3(gb), dim3(256), 0, c->stream, ne, nloc, d_ed, m, d_cnt, d_adj);
  if (m) hipLaunchKernelGGL(k_pe_rows<false>, dim3(fh_div_up(m, 4)), dim3(256), 0, c->stream, m, d_aptr, d_adj, d_ed, nloc, (const int*)nullptr, d_len, (int*)nullptr, d_err);
  std::vector<int> rp((size_t)m + 1, 0);
  if (m) FH_CHECK_HIP(hipMemcpyAsync(rp.data() + 1, d_len, (size_t)m * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  FH_CHECK_HIP(hipMemcpyAsync(&err, d_err, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  FH_CHECK_HIP(hipStreamSynchronize(c->stream));
  if (err) {               // a node with more than PE_CAP candidate columns: the host builder serves it
    if (!elem_dof) {
      fetched.resize(ne);
      FH_CHECK_HIP(hipMemcpy(fetched.data(), d_ed, ne * sizeof(int), hipMemcpyDeviceToHost));
      elem_dof = fetched.data();
    }
    hipFree(d_aptr);
    cleanup();
    std::vector<int> hrp((size_t)n + 1), hcol;
    FH_TRY(fh_pattern_from_elements(nel, nloc, elem_dof, n, hrp.data(), nullptr));
    hcol.resize((size_t)hrp[n]);
    FH_TRY(fh_pattern_from_elements(nel, nloc, elem_dof, n, hrp.data(), hcol.data()));
    return fh_mat_create_csr(c, m, n, hrp.data(), hcol.data(), nullptr, out);
  }
  tot = 0;
  for (int r = 0; r < m; r++) {
    tot += rp[r + 1];
    rp[r + 1] = (int)tot;
  }
  if (tot >= 2147483647ll) {
    hipFree(d_aptr);
    cleanup();
    fh_set_error("fh_mat_create_from_elements: nnz overflows int32");
    return 2;
  }
  fh_mat_t A = nullptr;
  if (fh_mat_alloc_device_pattern(c, m, n, std::move(rp), &A)) {
    hipFree(d_aptr);
    cleanup();
    fh_mat_destroy(A);
    return 2;
  }
  if (m) hipLaunchKernelGGL(k_pe_rows<true>, dim3(fh_div_up(m, 4)), dim3(256), 0, c->stream, m, d_aptr, d_adj, d_ed, nloc, A->d_rowptr, (int*)nullptr, A->d_col, d_err);
  FH_CHECK_HIP(hipGetLastError());
  FH_CHECK_HIP(hipStreamSynchronize(c->stream));
  hipFree(d_aptr);
  cleanup();
  FH_TRY(fh_mat_build_rowblocks(A, c->spmv_tile));
  *out = A;
  return 0;
  FH_GUARD_END("fh_mat_create_from_elements")
}

extern "C" int fh_mat_create_from_elements(fh_ctx_t c, int nel, int nloc, const int* elem_dof, int m, int n, fh_mat_t* out) {
  FH_REQUIRE(elem_dof || nel == 0, "fh_mat_create_from_elements: bad arguments");
  return mat_create_from_elements_impl(c, nel, nloc, elem_dof, nullptr, m, n, out);
}

int fh_mesh_host_arrays(fh_mesh_t m, int* dim, int* geom, int* nel, int* nnode, int* nloc, int* n_linear, const int** elem_dof, const double** coords);

extern "C" int fh_mat_create_from_mesh(fh_ctx_t c, fh_mesh_t mesh, int fe, fh_mat_t* out) {
  FH_REQUIRE(c && mesh && out && fe >= 0 && fe <= 3, "fh_mat_create_from_mesh: bad arguments (fe: 0 linear, 1 serendipity, 2 biquadratic, 3 piecewise constant)");
  int dim, geom, nel, nnode, nloc, nlin;
  const int* ed;
  const double* xy;
  FH_TRY(fh_mesh_host_arrays(mesh, &dim, &geom, &nel, &nnode, &nloc, &nlin, &ed, &xy));
  if (fe == 3) {       // element-owned dofs: every element couples with itself only
    std::vector<int> iota(nel);
    for (int e = 0; e < nel; e++) iota[e] = e;
    return mat_create_from_elements_impl(c, nel, 1, iota.data(), nullptr, nel, nel, out);
  }
  fh_mesh_dev* dev = nullptr;
  FH_TRY(fh_mesh_device(c, mesh, &dev));
  if (fe == 2) return mat_create_from_elements_impl(c, nel, nloc, nullptr, dev->d_elem_dof, nnode, nnode, out);
  // linear / serendipity: the family's nodes are the first 2^dim (4 / 8) resp. vertex + edge (8 / 20) local nodes of every element -- a strided copy of the
  // table -- and own the leading node ids (Mesh::GetSolutionDof, Mesh.cpp:1026-1050)
  int own[3] = {0, 0, 0};
  FH_TRY(fh_mesh_info(mesh, nullptr, nullptr, nullptr, nullptr, own, nullptr));
  const int nv = fe == 0 ? 1 << dim : (dim == 3 ? 20 : 8);
  if (fe == 1) nlin = own[1];
  int* d_lin = nullptr;
  FH_CHECK_HIP(hipMalloc(&d_lin, std::max<size_t>((size_t)nel * nv, 1) * sizeof(int)));
  hipError_t e = nel ? hipMemcpy2DAsync(d_lin, nv * sizeof(int), dev->d_elem_dof, nloc * sizeof(int), nv * sizeof(int), nel, hipMemcpyDeviceToDevice, c->stream) : hipSuccess;
  int rc = e == hipSuccess ? mat_create_from_elements_impl(c, nel, nv, nullptr, d_lin, nlin, nlin, out) : 1;
  if (e != hipSuccess) fh_set_error("fh_mat_create_from_mesh: %s", hipGetErrorString(e));
  hipStreamSynchronize(c->stream);
  hipFree(d_lin);
  return rc;
}

extern "C" int fh_mat_destroy(fh_mat_t A) {
  if (!A) return 0;
  hipStreamSynchronize(A->ctx->stream);
  fh_stage_free(A->stage);
  if (A->plan && A->plan_destroy) A->plan_destroy(A->plan);
  if (A->At) fh_mat_destroy(A->At);
  if (A->d_tperm) hipFree(A->d_tperm);
  if (A->d_rowptr) hipFree(A->d_rowptr);
  if (A->d_col) hipFree(A->d_col);
  if (A->d_val) hipFree(A->d_val);
  if (A->d_rowblk) hipFree(A->d_rowblk);
  if (A->d_uptr) hipFree(A->d_uptr);
  if (A->d_ucols) hipFree(A->d_ucols);
  if (A->d_lcol) hipFree(A->d_lcol);
  if (A->d_tile_s) hipFree(A->d_tile_s);
  if (A->d_blkinfo) hipFree(A->d_blkinfo);
  if (A->d_blkinfo_split) hipFree(A->d_blkinfo_split);
  if (A->d_diagpos) hipFree(A->d_diagpos);
  delete A;
  return 0;
}

extern "C" int fh_mat_size(fh_mat_t A, int* m, int* n, int* nnz) {
  if (m) *m = A->m;
  if (n) *n = A->n;
  if (nnz) *nnz = A->nnz;
  return 0;
}

extern "C" int fh_mat_dev_ptrs(fh_mat_t A, const int** rowptr, const int** col, const double** val) {
  FH_REQUIRE(A, "fh_mat_dev_ptrs: null matrix");
  if (rowptr) *rowptr = A->d_rowptr;
  if (col) *col = A->d_col;
  if (val) *val = A->d_val;
  return 0;
}

extern "C" int64_t fh_spmv_algorithmic_bytes(fh_mat_t A) {
  return 12ll * A->nnz + 4ll * (A->m + 1) + 8ll * A->n + 8ll * A->m;
}

// Bytes the LDS-staged kernel (spmv_kernel 3) really touches per product, from the sizes of its arrays: the value stream (8 nnz), the 16-bit
// local columns (2 nnz), the distinct-column lists (4 per entry of ucols), the 32-byte block descriptors, the row pointers of the block's
// rows, y (and b, D^-1, x_row for the fused forms) -- and x: `lo` counts every entry of x once (all tiles that share a column find it in the
// L2), `hi` counts every tile's gather as a miss.  The algorithmic figure (12 B per non-zero) sits between the two for FE matrices.
extern "C" int fh_spmv_expected_bytes(fh_mat_t A, int mode, int64_t* lo, int64_t* hi) {
  FH_REQUIRE(A && lo && hi && mode >= 0 && mode <= 3, "fh_spmv_expected_bytes: bad arguments");
  fh_ctx_t c = A->ctx;
  if (A->tile != c->spmv_tile || (A->tile_kernel != 3 && A->tile_kernel != 4)) FH_TRY(fh_mat_build_rowblocks(A, c->spmv_tile));
  if (A->lx_tile != A->tile) FH_TRY(fh_mat_build_localcols(A));
  const int64_t fixed = 10ll * A->nnz + 4ll * A->nu_total + 32ll * A->nblk + 4ll * ((int64_t)A->m + A->nblk) + 8ll * A->m +
                        (mode == 1 || mode == 2 ? 8ll * A->m : mode == 3 ? 24ll * A->m : 0);
  *lo = fixed + 8ll * A->n;
  *hi = fixed + 8ll * A->nu_total;
  return 0;
}

// row blocks: greedy, <= tile non-zeros and <= 512 rows per block; a row longer than the tile is alone
int fh_mat_build_rowblocks(fh_mat_t A, int tile) {
  FH_REQUIRE(tile == 256 || tile == 512 || tile == 1024 || tile == 2048 || tile == 4096, "spmv_tile must be 256..4096, power of two (got %d)", tile);
  const int maxrows = (A->ctx->spmv_kernel == 2) ? 128 : (A->ctx->spmv_kernel == 4) ? 255 : 512;
  std::vector<int> blk;
  blk.push_back(0);
  int acc = 0, rows = 0;
  for (int i = 0; i < A->m; i++) {
    int len = A->h_rowptr[i + 1] - A->h_rowptr[i];
    if (rows > 0 && (acc + len > tile || rows >= maxrows)) {
      blk.push_back(i);
      acc = 0;
      rows = 0;
    }
    acc += len;
    rows++;
  }
  blk.push_back(A->m);
  if (A->m == 0) blk.assign(1, 0);
  A->nblk = (int)blk.size() - 1;
  A->tile = tile;
  A->tile_kernel = A->ctx->spmv_kernel;
  A->h_rowblk = blk;
  A->lx_tile = 0;   // local-column data (if any) no longer matches the row blocks
  A->split_nown = -1;
  if (A->d_rowblk) FH_CHECK_HIP(hipFree(A->d_rowblk));
  FH_CHECK_HIP(hipMalloc(&A->d_rowblk, blk.size() * sizeof(int)));
  FH_CHECK_HIP(hipMemcpy(A->d_rowblk, blk.data(), blk.size() * sizeof(int), hipMemcpyHostToDevice));
  return 0;
}

extern "C" int fh_mat_zero(fh_mat_t A) {
  if (A) A->val_gen++;
  FH_CHECK_HIP(hipMemsetAsync(A->d_val, 0, (size_t)A->nnz * sizeof(double), A->ctx->stream));
  A->at_valid = false;
  return 0;
}

extern "C" int fh_mat_set_values_csr(fh_mat_t A, const double* val) {
  if (A) A->val_gen++;
  FH_CHECK_HIP(hipMemcpyAsync(A->d_val, val, (size_t)A->nnz * sizeof(double), hipMemcpyHostToDevice, A->ctx->stream));
  FH_CHECK_HIP(hipStreamSynchronize(A->ctx->stream));
  A->at_valid = false;
  return 0;
}

extern "C" int fh_mat_get_values_csr(fh_mat_t A, double* val) {
  FH_CHECK_HIP(hipMemcpyAsync(val, A->d_val, (size_t)A->nnz * sizeof(double), hipMemcpyDeviceToHost, A->ctx->stream));
  FH_CHECK_HIP(hipStreamSynchronize(A->ctx->stream));
  return 0;
}

extern "C" int fh_mat_get_pattern(fh_mat_t A, int* rowptr, int* col) {
  if (rowptr) memcpy(rowptr, A->h_rowptr.data(), ((size_t)A->m + 1) * sizeof(int));
  if (col) memcpy(col, fh_hcol(A).data(), (size_t)A->nnz * sizeof(int));
  return 0;
}

// ------------------------------------------------------------------------------------------------
// per-element crossings of the reference interface (slow path; the batched assembler is the fast one)
// ------------------------------------------------------------------------------------------------
static int host_find(const fh_mat_t A, int row, int c) {
  const int* b = fh_hcol(A).data() + A->h_rowptr[row];
  const int* e = fh_hcol(A).data() + A->h_rowptr[row + 1];
  const int* p = std::lower_bound(b, e, c);
  if (p == e || *p != c) return -1;
  return (int)(p - fh_hcol(A).data());
}

__global__ void k_apply_entries(double* __restrict__ val, const int* __restrict__ pos, const double* __restrict__ v, int n, int add) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && pos[i] >= 0) {
    if (add) atomicAdd(&val[pos[i]], v[i]);
    else val[pos[i]] = v[i];
  }
}

static int apply_entries(fh_mat_t A, const std::vector<int>& pos, const double* vals, int add) {
  fh_ctx_t c = A->ctx;
  int n = (int)pos.size();
  if (!n) return 0;
  int* d_pos = nullptr;
  double* d_v = nullptr;
  FH_CHECK_HIP(hipMalloc(&d_pos, n * sizeof(int)));
  FH_CHECK_HIP(hipMalloc(&d_v, n * sizeof(double)));
  FH_CHECK_HIP(hipMemcpyAsync(d_pos, pos.data(), n * sizeof(int), hipMemcpyHostToDevice, c->stream));
  FH_CHECK_HIP(hipMemcpyAsync(d_v, vals, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_apply_entries, dim3(fh_div_up(n, 256)), dim3(256), 0, c->stream, A->d_val, d_pos, d_v, n, add);
  FH_CHECK_HIP(hipGetLastError());
  FH_CHECK_HIP(hipStreamSynchronize(c->stream));
  hipFree(d_pos);
  hipFree(d_v);
  A->at_valid = false;
  return 0;
}

// immediate form of the staged add (fh_stage.hip): the block is on the device when the call returns
extern "C" int fh_mat_add_block(fh_mat_t A, int nrow, const int* rows, int ncol, const int* cols, const double* vals) {
  if (A) A->val_gen++;
  FH_TRY(fh_mat_stage_block(A, nrow, rows, ncol, cols, vals));
  return fh_mat_flush(A);
}

extern "C" int fh_mat_insert_row(fh_mat_t A, int row, int ncols, const int* cols, const double* vals) {
  if (A) A->val_gen++;
  FH_REQUIRE(row >= 0 && row < A->m, "fh_mat_insert_row: row %d out of range", row);
  std::vector<int> pos(ncols);
  for (int j = 0; j < ncols; j++) {
    pos[j] = host_find(A, row, cols[j]);
    FH_REQUIRE(pos[j] >= 0, "fh_mat_insert_row: entry (%d,%d) is outside the pattern", row, cols[j]);
  }
  return apply_entries(A, pos, vals, 0);
}

extern "C" int fh_mat_get_row(fh_mat_t A, int row, int* ncols, int* cols, double* vals) {
  FH_REQUIRE(row >= 0 && row < A->m, "fh_mat_get_row: row %d out of range", row);
  int s = A->h_rowptr[row], e = A->h_rowptr[row + 1];
  if (ncols) *ncols = e - s;
  if (cols) memcpy(cols, fh_hcol(A).data() + s, (size_t)(e - s) * sizeof(int));
  if (vals && e > s) {
    FH_CHECK_HIP(hipMemcpyAsync(vals, A->d_val + s, (size_t)(e - s) * sizeof(double), hipMemcpyDeviceToHost, A->ctx->stream));
    FH_CHECK_HIP(hipStreamSynchronize(A->ctx->stream));
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// K5: Dirichlet rows / columns
// ------------------------------------------------------------------------------------------------
// two rows per wave (32 lanes each: a Q2 row has 27 ... 125 entries), four waves per workgroup
__global__ __launch_bounds__(256) void k_zero_rows(const int* __restrict__ rowptr, const int* __restrict__ col, double* __restrict__ val,
                                                   const int* __restrict__ rows, int nrows, double diag) {
  const int r = blockIdx.x * 8 + (threadIdx.x >> 5), lane = threadIdx.x & 31;
  if (r >= nrows) return;
  const int row = rows[r];
  const int e = rowptr[row + 1];
  for (int k = rowptr[row] + lane; k < e; k += 32) val[k] = (col[k] == row) ? diag : 0.0;
}

__global__ __launch_bounds__(256) void k_mask_set(unsigned char* __restrict__ mask, const int* __restrict__ idx, int n) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) mask[idx[i]] = 1;
}

__global__ __launch_bounds__(256) void k_zero_cols(const int* __restrict__ col, double* __restrict__ val, const unsigned char* __restrict__ mask, int nnz) {
  for (int k = blockIdx.x * 256 + threadIdx.x; k < nnz; k += gridDim.x * 256)
    if (mask[col[k]]) val[k] = 0.0;
}

extern "C" int fh_mat_zero_rows(fh_mat_t A, int n, const int* rows, double diag) {
  if (n <= 0) return 0;
  fh_ctx_t c = A->ctx;
  for (int i = 0; i < n; i++) FH_REQUIRE(rows[i] >= 0 && rows[i] < A->m, "fh_mat_zero_rows: row %d out of range", rows[i]);
  int* d_rows = nullptr;
  FH_CHECK_HIP(hipMalloc(&d_rows, n * sizeof(int)));
  FH_CHECK_HIP(hipMemcpyAsync(d_rows, rows, n * sizeof(int), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_zero_rows, dim3(fh_div_up(n, 8)), dim3(256), 0, c->stream, A->d_rowptr, A->d_col, A->d_val, d_rows, n, diag);
  FH_CHECK_HIP(hipGetLastError());
  FH_CHECK_HIP(hipStreamSynchronize(c->stream));
  hipFree(d_rows);
  A->at_valid = false;
  return 0;
}

// ---- device-resident index list: the result of BuildBdcIndex (LinearEquationSolverPetsc.cpp:53-90) is built once per level
// (_bdcIndexIsInitialized) and reused by SetPenalty / ZerosBoundaryResiduals at every assembly; kept on the device, these two
// calls need no host traffic and no synchronisation
struct fh_index_s {
  fh_ctx_t ctx = nullptr;
  int n = 0, max_index = -1;
  bool has_negative = false;     // -1 entries are allowed for gather maps ("no source": the target gets 0)
  int* d = nullptr;
};

extern "C" int fh_index_create(fh_ctx_t ctx, int n, const int* idx, fh_index_t* out) {
  FH_REQUIRE(ctx && out && n >= 0 && (n == 0 || idx), "fh_index_create: bad arguments");
  fh_index_s* x = new fh_index_s();
  x->ctx = ctx;
  x->n = n;
  for (int i = 0; i < n; i++) {
    FH_REQUIRE(idx[i] >= -1, "fh_index_create: index %d (only -1 is allowed as 'none')", idx[i]);
    if (idx[i] < 0) x->has_negative = true;
    x->max_index = std::max(x->max_index, idx[i]);
  }
  FH_CHECK_HIP(hipMalloc(&x->d, std::max(n, 1) * sizeof(int)));
  if (n) FH_CHECK_HIP(hipMemcpy(x->d, idx, (size_t)n * sizeof(int), hipMemcpyHostToDevice));
  *out = x;
  return 0;
}

extern "C" int fh_index_destroy(fh_index_t x) {
  if (!x) return 0;
  hipStreamSynchronize(x->ctx->stream);
  hipFree(x->d);
  delete x;
  return 0;
}

extern "C" int fh_mat_zero_rows_index(fh_mat_t A, fh_index_t rows, double diag) {
  FH_REQUIRE(A && rows, "fh_mat_zero_rows_index: null argument");
  FH_REQUIRE(!rows->has_negative, "fh_mat_zero_rows_index: the list holds 'none' entries");
  FH_REQUIRE(rows->max_index < A->m, "fh_mat_zero_rows_index: row %d out of range", rows->max_index);
  if (rows->n == 0) return 0;
  hipLaunchKernelGGL(k_zero_rows, dim3(fh_div_up(rows->n, 8)), dim3(256), 0, A->ctx->stream, A->d_rowptr, A->d_col, A->d_val, rows->d, rows->n, diag);
  FH_CHECK_HIP(hipGetLastError());
  A->at_valid = false;
  return 0;
}

__global__ __launch_bounds__(256) void k_set_index(double* __restrict__ v, const int* __restrict__ idx, int n, double value) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) v[idx[i]] = value;
}

extern "C" int fh_vec_set_index(fh_vec_t v, fh_index_t idx, double value) {
  FH_REQUIRE(v && idx, "fh_vec_set_index: null argument");
  FH_REQUIRE(!idx->has_negative, "fh_vec_set_index: the list holds 'none' entries");
  FH_REQUIRE(idx->max_index < v->n_local, "fh_vec_set_index: index %d is not an owned entry", idx->max_index);
  if (idx->n == 0) return 0;
  hipLaunchKernelGGL(k_set_index, dim3(fh_div_up(idx->n, 256)), dim3(256), 0, v->ctx->stream, v->d, idx->d, idx->n, value);
  FH_CHECK_HIP(hipGetLastError());
  return 0;
}

// gathers along a device-resident map: dst[k] = (map[k] >= 0) ? src[map[k]] : 0.  Used to take the owned rows of an operator that
// was assembled / projected on a rank's extended box (adaptive levels on several ranks) without host traffic.
__global__ __launch_bounds__(256) void k_gather_map(double* __restrict__ dst, const double* __restrict__ src, const int* __restrict__ map, int n) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int j = map[i];
    dst[i] = (j >= 0) ? src[j] : 0.0;
  }
}

extern "C" int fh_mat_gather_values(fh_mat_t dst, fh_mat_t src, fh_index_t map) {
  if (dst) dst->val_gen++;
  FH_REQUIRE(dst && src && map, "fh_mat_gather_values: null argument");
  FH_REQUIRE(map->n == dst->nnz && map->max_index < src->nnz, "fh_mat_gather_values: map has %d entries (target nnz %d), largest source %d (source nnz %d)",
             map->n, dst->nnz, map->max_index, src->nnz);
  if (dst->nnz == 0) return 0;
  const int nb = std::min(fh_div_up(dst->nnz, 256), dst->ctx->num_cu * 16);
  hipLaunchKernelGGL(k_gather_map, dim3(nb), dim3(256), 0, dst->ctx->stream, dst->d_val, src->d_val, map->d, dst->nnz);
  FH_CHECK_HIP(hipGetLastError());
  dst->at_valid = false;
  return 0;
}

// ---- owned-row operators of a domain-decomposed level, built on the device from the operator of the rank's extended box ----------
// The reference's multi-rank matrices hold the rows a rank owns over [owned | ghost] columns (MPIAIJ behind PetscMatrix::init,
// PetscMatrix.cpp:162-203; ghost lists LinearEquation.cpp:239-280).  Here the rank's complete local operator already lives on the
// device; these entry points cut the owned rows out of it: integer pattern work on the host copies of the pattern, values by a
// device gather along a map that is kept, so that every re-preparation (new values, same pattern) is one kernel per operator.

// mask[c] = 1 for every column that one of the listed rows touches (the halo a set of owned rows reads)
extern "C" int fh_mat_col_mask(fh_mat_t A, int nrows, const int* rows, unsigned char* mask /* [A->n], or-ed into */) {
  FH_REQUIRE(A && mask && nrows >= 0 && (nrows == 0 || rows), "fh_mat_col_mask: bad arguments");
  for (int i = 0; i < nrows; i++) {
    const int r = rows[i];
    FH_REQUIRE(r >= 0 && r < A->m, "fh_mat_col_mask: row %d out of range", r);
    for (int k = A->h_rowptr[r]; k < A->h_rowptr[r + 1]; k++) mask[fh_hcol(A)[k]] = 1;
  }
  return 0;
}

// rowmask[r] = 1 for every row with an entry in a masked column (the rows a transposed operator restricted to those columns reads)
extern "C" int fh_mat_row_mask(fh_mat_t A, const unsigned char* colmask /* [A->n] */, unsigned char* rowmask /* [A->m], or-ed into */) {
  FH_REQUIRE(A && colmask && rowmask, "fh_mat_row_mask: null argument");
  for (int r = 0; r < A->m; r++) {
    if (rowmask[r]) continue;
    for (int k = A->h_rowptr[r]; k < A->h_rowptr[r + 1]; k++)
      if (colmask[fh_hcol(A)[k]]) {
        rowmask[r] = 1;
        break;
      }
  }
  return 0;
}

// map[k] for every non-zero k = (r, c) of dst: position of (src_row[r], src_col[c]) in src, or -1 when src has no such entry
// (NULL row / column lists = identity).  Feeds fh_mat_gather_values.
extern "C" int fh_mat_value_map(fh_mat_t dst, fh_mat_t src, const int* src_row, const int* src_col, fh_index_t* out) {
  FH_GUARD_BEGIN
  FH_REQUIRE(dst && src && out, "fh_mat_value_map: null argument");
  std::vector<int> map((size_t)dst->nnz, -1);
  for (int r = 0; r < dst->m; r++) {
    const int sr = src_row ? src_row[r] : r;
    if (sr < 0) continue;
    FH_REQUIRE(sr < src->m, "fh_mat_value_map: source row %d out of range", sr);
    const int* b = fh_hcol(src).data() + src->h_rowptr[sr];
    const int* e = fh_hcol(src).data() + src->h_rowptr[sr + 1];
    for (int k = dst->h_rowptr[r]; k < dst->h_rowptr[r + 1]; k++) {
      const int sc = src_col ? src_col[fh_hcol(dst)[k]] : fh_hcol(dst)[k];
      if (sc < 0) continue;
      const int* q = std::lower_bound(b, e, sc);
      if (q != e && *q == sc) map[k] = (int)(q - fh_hcol(src).data());
    }
  }
  return fh_index_create(dst->ctx, dst->nnz, map.data(), out);
  FH_GUARD_END("fh_mat_value_map")
}

// dst = the listed rows of src (in list order) with columns renumbered by newcol[src->n] (entries whose new column is < 0 are
// dropped -- they must hold zeros, which fh_mat_restrict_check verifies on the device); *map as in fh_mat_value_map, values gathered
extern "C" int fh_mat_restrict(fh_mat_t src, int nrows, const int* rows, const int* newcol, int ncols_new, fh_mat_t* out, fh_index_t* map_out) {
  FH_GUARD_BEGIN
  FH_REQUIRE(src && out && map_out && newcol && nrows >= 0 && (nrows == 0 || rows) && ncols_new >= 0, "fh_mat_restrict: bad arguments");
  std::vector<int> rp(nrows + 1, 0);
  for (int i = 0; i < nrows; i++) {
    const int r = rows[i];
    FH_REQUIRE(r >= 0 && r < src->m, "fh_mat_restrict: row %d out of range", r);
    int cnt = 0;
    for (int k = src->h_rowptr[r]; k < src->h_rowptr[r + 1]; k++) {
      const int c = newcol[fh_hcol(src)[k]];
      FH_REQUIRE(c < ncols_new, "fh_mat_restrict: new column %d out of range (%d columns)", c, ncols_new);
      cnt += c >= 0;
    }
    rp[i + 1] = rp[i] + cnt;
  }
  std::vector<int> col((size_t)rp[nrows]), map((size_t)rp[nrows]);
  std::vector<std::pair<int, int>> buf;
  for (int i = 0; i < nrows; i++) {
    const int r = rows[i];
    buf.clear();
    for (int k = src->h_rowptr[r]; k < src->h_rowptr[r + 1]; k++) {
      const int c = newcol[fh_hcol(src)[k]];
      if (c >= 0) buf.emplace_back(c, k);
    }
    std::sort(buf.begin(), buf.end());
    for (size_t t = 0; t < buf.size(); t++) {
      FH_REQUIRE(t == 0 || buf[t].first != buf[t - 1].first, "fh_mat_restrict: two columns of row %d map to the same new column", r);
      col[rp[i] + t] = buf[t].first;
      map[rp[i] + t] = buf[t].second;
    }
  }
  fh_mat_t D = nullptr;
  FH_TRY(fh_mat_create_csr(src->ctx, nrows, ncols_new, rp.data(), col.data(), nullptr, &D));
  fh_index_t M = nullptr;
  int rc = fh_index_create(src->ctx, (int)map.size(), map.data(), &M);
  if (rc) {
    fh_mat_destroy(D);
    return rc;
  }
  rc = fh_mat_gather_values(D, src, M);
  if (rc) {
    fh_mat_destroy(D);
    fh_index_destroy(M);
    return rc;
  }
  *out = D;
  *map_out = M;
  return 0;
  FH_GUARD_END("fh_mat_restrict")
}

// largest |value| among the entries of the listed rows of src that fh_mat_restrict drops (new column < 0): must be 0 for the owned
// rows of a level operator -- an owned row reads nothing outside its halo
__global__ __launch_bounds__(256) void k_dropped_max(const int* __restrict__ rowptr, const int* __restrict__ col, const double* __restrict__ val,
                                                     const int* __restrict__ rows, int nrows, const int* __restrict__ newcol, double* __restrict__ out) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  double mx = 0.0;
  if (i < nrows) {
    const int r = rows[i];
    for (int k = rowptr[r] + lane; k < rowptr[r + 1]; k += 64)
      if (newcol[col[k]] < 0) mx = fmax(mx, fabs(val[k]));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, 64));
  if (lane == 0 && mx > 0.0) atomicMax(reinterpret_cast<unsigned long long*>(out), (unsigned long long)__double_as_longlong(mx));   // positive doubles order like integers
}

extern "C" int fh_mat_restrict_check(fh_mat_t src, int nrows, const int* rows, const int* newcol, double* max_dropped) {
  FH_REQUIRE(src && newcol && max_dropped && nrows >= 0 && (nrows == 0 || rows), "fh_mat_restrict_check: bad arguments");
  *max_dropped = 0.0;
  if (nrows == 0) return 0;
  fh_ctx_t c = src->ctx;
  int *d_rows = nullptr, *d_new = nullptr;
  double* d_out = nullptr;
  FH_CHECK_HIP(hipMalloc(&d_rows, (size_t)nrows * sizeof(int)));
  FH_CHECK_HIP(hipMalloc(&d_new, (size_t)std::max(src->n, 1) * sizeof(int)));
  FH_CHECK_HIP(hipMalloc(&d_out, sizeof(double)));
  FH_CHECK_HIP(hipMemcpyAsync(d_rows, rows, (size_t)nrows * sizeof(int), hipMemcpyHostToDevice, c->stream));
  FH_CHECK_HIP(hipMemcpyAsync(d_new, newcol, (size_t)src->n * sizeof(int), hipMemcpyHostToDevice, c->stream));
  FH_CHECK_HIP(hipMemsetAsync(d_out, 0, sizeof(double), c->stream));
  hipLaunchKernelGGL(k_dropped_max, dim3(fh_div_up(nrows, 4)), dim3(256), 0, c->stream, src->d_rowptr, src->d_col, src->d_val, d_rows, nrows, d_new, d_out);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(max_dropped, d_out, sizeof(double), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  hipFree(d_rows);
  hipFree(d_new);
  hipFree(d_out);
  FH_CHECK_HIP(e);
  return 0;
}

extern "C" int fh_vec_gather(fh_vec_t dst, fh_vec_t src, fh_index_t map) {
  FH_REQUIRE(dst && src && map, "fh_vec_gather: null argument");
  FH_REQUIRE(map->n <= dst->n_local + dst->nghost && map->max_index < src->n_local + src->nghost, "fh_vec_gather: map does not fit the vectors");
  if (map->n == 0) return 0;
  const int nb = std::min(fh_div_up(map->n, 256), dst->ctx->num_cu * 16);
  hipLaunchKernelGGL(k_gather_map, dim3(nb), dim3(256), 0, dst->ctx->stream, dst->d, src->d, map->d, map->n);
  FH_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int fh_mat_zero_cols(fh_mat_t A, int n, const int* cols) {
  if (A) A->val_gen++;
  if (n <= 0 || A->nnz == 0) return 0;
  fh_ctx_t c = A->ctx;
  for (int i = 0; i < n; i++) FH_REQUIRE(cols[i] >= 0 && cols[i] < A->n, "fh_mat_zero_cols: column %d out of range", cols[i]);
  int* d_idx = nullptr;
  unsigned char* d_mask = nullptr;
  FH_CHECK_HIP(hipMalloc(&d_idx, n * sizeof(int)));
  FH_CHECK_HIP(hipMalloc(&d_mask, (size_t)A->n));
  FH_CHECK_HIP(hipMemsetAsync(d_mask, 0, (size_t)A->n, c->stream));
  FH_CHECK_HIP(hipMemcpyAsync(d_idx, cols, n * sizeof(int), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_mask_set, dim3(fh_div_up(n, 256)), dim3(256), 0, c->stream, d_mask, d_idx, n);
  int nb = std::min(fh_div_up(A->nnz, 256), c->num_cu * 8);
  hipLaunchKernelGGL(k_zero_cols, dim3(nb), dim3(256), 0, c->stream, A->d_col, A->d_val, d_mask, A->nnz);
  FH_CHECK_HIP(hipGetLastError());
  FH_CHECK_HIP(hipStreamSynchronize(c->stream));
  hipFree(d_idx);
  hipFree(d_mask);
  A->at_valid = false;
  return 0;
}

// position of the diagonal entry of every row (-1: none), once per matrix: the pattern of a matrix never changes after its creation
__global__ __launch_bounds__(256) void k_diag_pos(const int* __restrict__ rowptr, const int* __restrict__ col, int* __restrict__ pos, int m) {
  int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= m) return;
  int lo = rowptr[r], hi = rowptr[r + 1] - 1, p = -1;
  while (lo <= hi) {  // sorted columns
    int mid = lo + ((hi - lo) >> 1);   // (lo + hi) overflows beyond 2^30 non-zeros
    int cc = col[mid];
    if (cc == r) { p = mid; break; }
    if (cc < r) lo = mid + 1; else hi = mid - 1;
  }
  pos[r] = p;
}

__global__ __launch_bounds__(256) void k_get_diag(const int* __restrict__ pos, const double* __restrict__ val, double* __restrict__ d, int m, int invert) {
  int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= m) return;
  const int p = pos[r];
  double v = p >= 0 ? val[p] : 0.0;
  if (invert) v = (v == 0.0) ? 1.0 : 1.0 / v;   // PCJACOBI: zero diagonal -> 1
  d[r] = v;
}

int fh_dev_get_diag(fh_mat_t A, double* d, int invert) {
  if (A->m == 0) return 0;
  if (!A->d_diagpos) {
    FH_CHECK_HIP(hipMalloc(&A->d_diagpos, (size_t)A->m * sizeof(int)));
    hipLaunchKernelGGL(k_diag_pos, dim3(fh_div_up(A->m, 256)), dim3(256), 0, A->ctx->stream, A->d_rowptr, A->d_col, A->d_diagpos, A->m);
  }
  hipLaunchKernelGGL(k_get_diag, dim3(fh_div_up(A->m, 256)), dim3(256), 0, A->ctx->stream, A->d_diagpos, A->d_val, d, A->m, invert);
  FH_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int fh_mat_get_diagonal(fh_mat_t A, fh_vec_t d) {
  FH_REQUIRE(d->n_local >= std::min(A->m, A->n), "fh_mat_get_diagonal: vector too short");
  return fh_dev_get_diag(A, d->d, 0);
}

// ------------------------------------------------------------------------------------------------
// transpose: symbolic part on the host (integer setup work), values gathered on the device
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gather_perm(double* __restrict__ dst, const double* __restrict__ src, const int* __restrict__ perm, int n) {
  for (int k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256) dst[k] = src[perm[k]];
}

static int build_transpose_host(fh_mat_t A, fh_mat_t* out, int** d_perm_out) {
  const int m = A->m, n = A->n, nnz = A->nnz;
  std::vector<int> trp(n + 1, 0), tcol(nnz), perm(nnz);
  for (int k = 0; k < nnz; k++) trp[fh_hcol(A)[k] + 1]++;
  for (int j = 0; j < n; j++) trp[j + 1] += trp[j];
  std::vector<int> cur(trp.begin(), trp.end() - 1);
  for (int i = 0; i < m; i++)
    for (int k = A->h_rowptr[i]; k < A->h_rowptr[i + 1]; k++) {
      int p = cur[fh_hcol(A)[k]]++;
      tcol[p] = i;   // rows visited in increasing order => sorted columns in the transpose
      perm[p] = k;
    }
  fh_mat_t At = nullptr;
  FH_TRY(fh_mat_create_csr(A->ctx, n, m, trp.data(), tcol.data(), nullptr, &At));
  int* d_perm = nullptr;
  FH_CHECK_HIP(hipMalloc(&d_perm, std::max(nnz, 1) * sizeof(int)));
  if (nnz) FH_CHECK_HIP(hipMemcpy(d_perm, perm.data(), nnz * sizeof(int), hipMemcpyHostToDevice));
  *out = At;
  *d_perm_out = d_perm;
  FH_TRACE("build_transpose: %d x %d, %d non-zeros", m, n, nnz);
  return 0;
}

// the same on the DEVICE (round 4): column counts by a counting pass, the host scans them, entries dropped into their transposed row through an
// atomic cursor and every transposed row (<= TR_CAP entries) sorted by its column (= the source row) in LDS, the position in the source riding
// along -- identical pattern and permutation as the host loop (which visits the rows in ascending order); longer rows: the host loop.
constexpr int TR_CAP = 1024;
__global__ __launch_bounds__(256) void k_tr_count(int nnz, const int* __restrict__ col, int* __restrict__ cnt) {
  for (int k = blockIdx.x * 256 + threadIdx.x; k < nnz; k += gridDim.x * 256) atomicAdd(&cnt[col[k]], 1);
}
__global__ __launch_bounds__(256) void k_tr_fill(int m, const int* __restrict__ rowptr, const int* __restrict__ col, int* __restrict__ cur, int* __restrict__ tcol,
                                                 int* __restrict__ perm) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= m) return;
  for (int k = rowptr[i] + lane; k < rowptr[i + 1]; k += 64) {
    const int p = atomicAdd(&cur[col[k]], 1);
    tcol[p] = i;
    perm[p] = k;
  }
}
__global__ __launch_bounds__(256) void k_tr_sort(int n, const int* __restrict__ trp, int* __restrict__ tcol, int* __restrict__ perm) {
  __shared__ int keys[4][TR_CAP], pay[4][TR_CAP];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + wave;
  if (j >= n) return;
  const int s = trp[j], len = trp[j + 1] - s;
  if (len <= 1) return;
  int *key = keys[wave], *val = pay[wave];
  int np = 64;
  while (np < len) np <<= 1;
  for (int k = lane; k < np; k += 64) {
    key[k] = k < len ? tcol[s + k] : 0x7fffffff;
    val[k] = k < len ? perm[s + k] : 0;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  for (int size = 2; size <= np; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = lane; t < (np >> 1); t += 64) {
        const int lo = ((t / stride) * (stride << 1)) + (t % stride), hi = lo + stride;
        const bool up = ((lo & size) == 0);
        const int a = key[lo], c = key[hi];
        if ((a > c) == up) {
          key[lo] = c;
          key[hi] = a;
          const int v = val[lo];
          val[lo] = val[hi];
          val[hi] = v;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  for (int k = lane; k < len; k += 64) {
    tcol[s + k] = key[k];
    perm[s + k] = val[k];
  }
}

static int build_transpose(fh_mat_t A, fh_mat_t* out, int** d_perm_out) {
  const int m = A->m, n = A->n, nnz = A->nnz;
  fh_ctx_t c = A->ctx;
  if (nnz == 0) return build_transpose_host(A, out, d_perm_out);
  int* d_cnt = nullptr;
  FH_CHECK_HIP(hipMalloc(&d_cnt, ((size_t)n + 1) * sizeof(int)));
  FH_CHECK_HIP(hipMemsetAsync(d_cnt, 0, ((size_t)n + 1) * sizeof(int), c->stream));
  hipLaunchKernelGGL(k_tr_count, dim3(std::min(fh_div_up(nnz, 256), c->num_cu * 16)), dim3(256), 0, c->stream, nnz, A->d_col, d_cnt);
  std::vector<int> trp((size_t)n + 1, 0);
  FH_CHECK_HIP(hipMemcpyAsync(trp.data() + 1, d_cnt, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  FH_CHECK_HIP(hipStreamSynchronize(c->stream));
  int maxrow = 0;
  for (int j = 0; j < n; j++) {
    maxrow = std::max(maxrow, trp[j + 1]);
    trp[j + 1] += trp[j];
  }
  if (maxrow > TR_CAP) {
    hipFree(d_cnt);
    return build_transpose_host(A, out, d_perm_out);
  }
  fh_mat_t At = new fh_mat_s();
  static std::atomic<uint64_t> next_uid{(uint64_t)1 << 41};
  At->uid = next_uid++;
  At->ctx = c;
  At->m = n;
  At->n = m;
  At->nnz = nnz;
  At->max_row = maxrow;
  At->h_rowptr = trp;
  int* d_perm = nullptr;
  FH_CHECK_HIP(hipMalloc(&At->d_rowptr, ((size_t)n + 1) * sizeof(int)));
  FH_CHECK_HIP(hipMalloc(&At->d_col, ((size_t)nnz + 2) * sizeof(int)));
  FH_CHECK_HIP(hipMalloc(&At->d_val, ((size_t)nnz + 2) * sizeof(double)));
  FH_CHECK_HIP(hipMalloc(&d_perm, (size_t)nnz * sizeof(int)));
  FH_CHECK_HIP(hipMemsetAsync(At->d_col + nnz, 0, 2 * sizeof(int), c->stream));
  FH_CHECK_HIP(hipMemsetAsync(At->d_val, 0, ((size_t)nnz + 2) * sizeof(double), c->stream));
  FH_CHECK_HIP(hipMemcpyAsync(At->d_rowptr, trp.data(), ((size_t)n + 1) * sizeof(int), hipMemcpyHostToDevice, c->stream));
  FH_CHECK_HIP(hipMemcpyAsync(d_cnt, trp.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, c->stream));          // cursors
  hipLaunchKernelGGL(k_tr_fill, dim3(fh_div_up(m, 4)), dim3(256), 0, c->stream, m, A->d_rowptr, A->d_col, d_cnt, At->d_col, d_perm);
  hipLaunchKernelGGL(k_tr_sort, dim3(fh_div_up(n, 4)), dim3(256), 0, c->stream, n, At->d_rowptr, At->d_col, d_perm);
  FH_CHECK_HIP(hipGetLastError());
  FH_CHECK_HIP(hipStreamSynchronize(c->stream));
  hipFree(d_cnt);
  FH_TRY(fh_mat_build_rowblocks(At, c->spmv_tile));
  *out = At;
  *d_perm_out = d_perm;
  FH_TRACE("build_transpose (device): %d x %d, %d non-zeros", m, n, nnz);
  return 0;
}

static int gather_transpose_values(fh_mat_t A, fh_mat_t At, const int* d_perm) {
  if (A->nnz == 0) return 0;
  int nb = std::min(fh_div_up(A->nnz, 256), A->ctx->num_cu * 8);
  hipLaunchKernelGGL(k_gather_perm, dim3(nb), dim3(256), 0, A->ctx->stream, At->d_val, A->d_val, d_perm, A->nnz);
  FH_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int fh_mat_transpose(fh_mat_t A, fh_mat_t* out) {
  int* d_perm = nullptr;
  FH_TRY(build_transpose(A, out, &d_perm));
  FH_TRY(gather_transpose_values(A, *out, d_perm));
  FH_CHECK_HIP(hipStreamSynchronize(A->ctx->stream));
  hipFree(d_perm);
  return 0;
}

int fh_mat_refresh_transpose(fh_mat_t A) {
  if (!A->At) FH_TRY(build_transpose(A, &A->At, &A->d_tperm));
  if (!A->at_valid) {
    FH_TRY(gather_transpose_values(A, A->At, A->d_tperm));
    A->at_valid = true;
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// norms (l1 = max column sum, linfty = max row sum) -- setup/diagnostic only: via the host
// ------------------------------------------------------------------------------------------------
extern "C" int fh_mat_norm(fh_mat_t A, int kind, double* out) {
  std::vector<double> v(A->nnz);
  if (A->nnz) FH_TRY(fh_mat_get_values_csr(A, v.data()));
  double best = 0.0;
  if (kind == 0) {
    for (int i = 0; i < A->m; i++) {
      double s = 0.0;
      for (int k = A->h_rowptr[i]; k < A->h_rowptr[i + 1]; k++) s += fabs(v[k]);
      best = std::max(best, s);
    }
  } else if (kind == 1) {
    std::vector<double> cs(A->n, 0.0);
    for (int k = 0; k < A->nnz; k++) cs[fh_hcol(A)[k]] += fabs(v[k]);
    for (double s : cs) best = std::max(best, s);
  } else {
    fh_set_error("fh_mat_norm: unknown kind %d", kind);
    return 2;
  }
  *out = best;
  return 0;
}

// ------------------------------------------------------------------------------------------------
// SpMV kernels
// ------------------------------------------------------------------------------------------------
// epilogue: MODE 0: y = s ; 1: y += s ; 2: y = b - s ; 3: y = x + omega*dinv*(b - s)
template <int MODE>
__device__ __forceinline__ void spmv_store(double s, int r, const double* __restrict__ x, double* __restrict__ y,
                                           const double* __restrict__ b, const double* __restrict__ dinv, double omega) {
  if (MODE == 0) y[r] = s;
  else if (MODE == 1) y[r] += s;
  else if (MODE == 2) y[r] = b[r] - s;
  else y[r] = x[r] + omega * dinv[r] * (b[r] - s);
}

template <int TILE, int MODE>
__global__ __launch_bounds__(256) void k_spmv_stream(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                     const double* __restrict__ val, const int* __restrict__ rowblk, int nblk, int q,
                                                     const double* __restrict__ x, double* __restrict__ y,
                                                     const double* __restrict__ b, const double* __restrict__ dinv, double omega) {
  __shared__ double prod[TILE + 2];
  // XCD-aware logical block id: physical block p runs on XCD p%8; give each XCD a contiguous range of row blocks
  int blk = (q > 0) ? (int)(blockIdx.x & 7) * q + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  if (blk >= nblk) return;
  const int tid = threadIdx.x;
  const int r0 = rowblk[blk], r1 = rowblk[blk + 1];
  const int s = rowptr[r0], e = rowptr[r1];
  if (r1 - r0 == 1 && e - s > TILE) {
    // one long row: CSR-vector over the whole workgroup
    double acc = 0.0;
    for (int k = s + tid; k < e; k += 256) acc += val[k] * x[col[k]];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((tid & 63) == 0) prod[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) spmv_store<MODE>(prod[0] + prod[1] + prod[2] + prod[3], r0, x, y, b, dinv, omega);
    return;
  }
  // ---- stream phase: aligned (16 B val, 8 B col) pairs --------------------------------------------
  const int s2 = s & ~1;
  constexpr int ITER = TILE / 512 + 1;
  double2 v[ITER];
  int2 c[ITER];
#pragma unroll
  for (int k = 0; k < ITER; k++) {
    int i = s2 + 2 * tid + k * 512;
    if (i < e) {
      v[k] = *reinterpret_cast<const double2*>(val + i);
      c[k] = *reinterpret_cast<const int2*>(col + i);
    }
  }
#pragma unroll
  for (int k = 0; k < ITER; k++) {
    int i = s2 + 2 * tid + k * 512;
    if (i < e) {
      double p0 = v[k].x * x[c[k].x];
      if (i >= s) prod[i - s] = p0;
      if (i + 1 < e) prod[i + 1 - s] = v[k].y * x[c[k].y];
    }
  }
  __syncthreads();
  // ---- segmented reduction: G lanes per row ---------------------------------------------------------
  const int nrows = r1 - r0;
  int G = (nrows <= 16) ? 16 : (nrows <= 32) ? 8 : (nrows <= 64) ? 4 : (nrows <= 128) ? 2 : 1;
  const int gl = tid & (G - 1);
  const int rows_per_pass = 256 / G;
  const int npass = (nrows + rows_per_pass - 1) / rows_per_pass;   // uniform trip count: every lane joins the shuffles
  for (int p = 0; p < npass; p++) {
    const int rr = p * rows_per_pass + tid / G;
    const bool live = rr < nrows;
    const int r = r0 + (live ? rr : 0);
    double acc = 0.0;
    if (live) {
      const int a = rowptr[r] - s, z = rowptr[r + 1] - s;
      for (int k = a + gl; k < z; k += G) acc += prod[k];
    }
    for (int off = G >> 1; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (live && gl == 0) spmv_store<MODE>(acc, r, x, y, b, dinv, omega);
  }
}

// ------------------------------------------------------------------------------------------------
// tile-local column compaction: for every row block the sorted list of distinct columns (ucols) and, per non-zero,
// the 16-bit index into that list.  The SpMV then gathers each needed x entry ONCE per tile into LDS (sorted, hence
// mostly coalesced) and the 135M per-non-zero gathers become LDS reads; the column stream shrinks from 4 to 2 bytes.
// Integer setup work on host threads, done lazily at the first product with the matrix.
// ------------------------------------------------------------------------------------------------
// On the DEVICE since round 4 (the host version, 16 threads, took 0.39 s for the 135 M non-zeros of the bench's fine level -- half of the
// first preparation): one workgroup per row block sorts the block's columns in LDS (bitonic, <= 4096 keys), marks the first of every run,
// scans the marks and writes the distinct columns to a scratch slab and the 16-bit local index of every non-zero; the host only scans the
// 67 k per-block counts.  Same lists, same indices as the host loop gave.
template <int TILE>
__global__ __launch_bounds__(256) void k_localcols(const int* __restrict__ blk, const int* __restrict__ rowptr, const int* __restrict__ col, int tile,
                                                   int* __restrict__ uscratch, int* __restrict__ ucount, unsigned short* __restrict__ lcol) {
  __shared__ int key[TILE];
  __shared__ int uniq[TILE];
  __shared__ int wsum[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int s = rowptr[blk[b]], e = rowptr[blk[b + 1]], n = e - s;
  if (n > tile) {                       // single long row: handled by the global-column path
    if (tid == 0) ucount[b] = 0;
    return;
  }
  int np = 64;
  while (np < n) np <<= 1;
  for (int k = tid; k < np; k += 256) key[k] = k < n ? col[s + k] : 0x7fffffff;
  __syncthreads();
  for (int size = 2; size <= np; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < (np >> 1); t += 256) {
        const int lo = ((t / stride) * (stride << 1)) + (t % stride), hi = lo + stride;
        const bool up = ((lo & size) == 0);
        const int a = key[lo], c = key[hi];
        if ((a > c) == up) {
          key[lo] = c;
          key[hi] = a;
        }
      }
      __syncthreads();
    }
  // distinct keys: exclusive scan of the "first of its run" marks, 256 threads x (np / 256) consecutive entries
  const int per = np >> 8 ? np >> 8 : 1, k0 = tid * per;
  int mine = 0;
  for (int k = k0; k < k0 + per && k < n; k++) mine += (k == 0 || key[k] != key[k - 1]) ? 1 : 0;
  int incl = mine;
  const int lane = tid & 63, wave = tid >> 6;
  for (int d = 1; d < 64; d <<= 1) {
    const int v = __shfl_up(incl, d, 64);
    if (lane >= d) incl += v;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int base = incl - mine;
  for (int w = 0; w < wave; w++) base += wsum[w];
  const int total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  for (int k = k0; k < k0 + per && k < n; k++)
    if (k == 0 || key[k] != key[k - 1]) uniq[base++] = key[k];
  __syncthreads();
  for (int k = tid; k < total; k += 256) uscratch[(size_t)b * tile + k] = uniq[k];
  if (tid == 0) ucount[b] = total;
  for (int k = tid; k < n; k += 256) {
    const int c = col[s + k];
    int lo = 0, hi = total - 1;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (uniq[mid] < c) lo = mid + 1; else hi = mid;
    }
    lcol[s + k] = (unsigned short)lo;
  }
}
__global__ __launch_bounds__(256) void k_localcols_compact(const int* __restrict__ uscratch, const int* __restrict__ uptr, int tile, int* __restrict__ ucols) {
  const int b = blockIdx.x;
  const int o = uptr[b], n = uptr[b + 1] - o;
  for (int k = threadIdx.x; k < n; k += 256) ucols[o + k] = uscratch[(size_t)b * tile + k];
}

int fh_mat_build_localcols(fh_mat_t A) {
  const int nblk = A->nblk;
  const std::vector<int>& blk = A->h_rowblk;
  fh_ctx_t c = A->ctx;
  std::vector<int> uptr(nblk + 1, 0);
  if (A->d_uptr) FH_CHECK_HIP(hipFree(A->d_uptr));
  if (A->d_ucols) FH_CHECK_HIP(hipFree(A->d_ucols));
  if (A->d_lcol) FH_CHECK_HIP(hipFree(A->d_lcol));
  A->d_uptr = nullptr;
  A->d_ucols = nullptr;
  A->d_lcol = nullptr;
  FH_CHECK_HIP(hipMalloc(&A->d_lcol, ((size_t)A->nnz + 2) * sizeof(unsigned short)));
  FH_CHECK_HIP(hipMemsetAsync(A->d_lcol, 0, ((size_t)A->nnz + 2) * sizeof(unsigned short), c->stream));
  {
    int *d_scratch = nullptr, *d_count = nullptr;
    FH_CHECK_HIP(hipMalloc(&d_scratch, std::max<size_t>((size_t)nblk * A->tile, 1) * sizeof(int)));
    FH_CHECK_HIP(hipMalloc(&d_count, (size_t)(nblk + 1) * sizeof(int)));
    if (nblk) {
      hipLaunchKernelGGL(k_localcols<4096>, dim3(nblk), dim3(256), 0, c->stream, A->d_rowblk, A->d_rowptr, A->d_col, A->tile, d_scratch, d_count, A->d_lcol);
      FH_CHECK_HIP(hipGetLastError());
      FH_CHECK_HIP(hipMemcpyAsync(uptr.data() + 1, d_count, (size_t)nblk * sizeof(int), hipMemcpyDeviceToHost, c->stream));
      FH_CHECK_HIP(hipStreamSynchronize(c->stream));
    }
    for (int b = 0; b < nblk; b++) uptr[b + 1] += uptr[b];
    A->nu_total = uptr[nblk];
    FH_CHECK_HIP(hipMalloc(&A->d_uptr, uptr.size() * sizeof(int)));
    FH_CHECK_HIP(hipMalloc(&A->d_ucols, ((size_t)uptr[nblk] + 1) * sizeof(int)));
    FH_CHECK_HIP(hipMemcpyAsync(A->d_uptr, uptr.data(), uptr.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    if (nblk) {
      hipLaunchKernelGGL(k_localcols_compact, dim3(nblk), dim3(256), 0, c->stream, d_scratch, A->d_uptr, A->tile, A->d_ucols);
      FH_CHECK_HIP(hipGetLastError());
    }
    FH_CHECK_HIP(hipStreamSynchronize(c->stream));
    hipFree(d_scratch);
    hipFree(d_count);
  }
  {
    std::vector<int> ts(nblk + 1);
    for (int b = 0; b <= nblk; b++) ts[b] = A->h_rowptr[blk[b]];
    if (A->d_tile_s) FH_CHECK_HIP(hipFree(A->d_tile_s));
    FH_CHECK_HIP(hipMalloc(&A->d_tile_s, ts.size() * sizeof(int)));
    FH_CHECK_HIP(hipMemcpy(A->d_tile_s, ts.data(), ts.size() * sizeof(int), hipMemcpyHostToDevice));
  }
  {
    // one 32-byte descriptor per row block {first row, end row, first nnz, end nnz, first unique column, #unique columns}: the
    // kernel starts from ONE scalar load instead of the chain rowblk -> rowptr (and uptr), i.e. one memory round trip less
    // before the matrix stream can be issued
    std::vector<int> info((size_t)nblk * 8 + 8, 0);
    for (int b = 0; b < nblk; b++) {
      int* d = &info[(size_t)b * 8];
      d[0] = blk[b];
      d[1] = blk[b + 1];
      d[2] = A->h_rowptr[blk[b]];
      d[3] = A->h_rowptr[blk[b + 1]];
      d[4] = uptr[b];
      d[5] = uptr[b + 1] - uptr[b];
    }
    if (A->d_blkinfo) FH_CHECK_HIP(hipFree(A->d_blkinfo));
    FH_CHECK_HIP(hipMalloc(&A->d_blkinfo, info.size() * sizeof(int)));
    FH_CHECK_HIP(hipMemcpy(A->d_blkinfo, info.data(), info.size() * sizeof(int), hipMemcpyHostToDevice));
  }
  A->lx_tile = A->tile;
  A->split_nown = -1;
  FH_TRACE("fh_mat_build_localcols: %d x %d, %d non-zeros, %d row blocks", A->m, A->n, A->nnz, nblk);
  return 0;
}

// SHARE: the x tile and the products use the SAME LDS buffer (one more barrier) -> half the LDS per tile, so twice the
// matrix bytes are in flight per CU at the same residency
template <int TILE, int MODE, bool SHARE, int NT>
__global__ __launch_bounds__(NT) void k_spmv_lx(const int* __restrict__ rowptr, const int* __restrict__ col, const unsigned short* __restrict__ lcol,
                                                 const double* __restrict__ val, const int* __restrict__ blkinfo,
                                                 const int* __restrict__ ucols, int nblk, int q, int chunk, const double* __restrict__ x,
                                                 double* __restrict__ y, const double* __restrict__ b, const double* __restrict__ dinv, double omega) {
  __shared__ double prod[TILE + 2];
  __shared__ double xs_own[SHARE ? 1 : TILE];
  double* xs = SHARE ? prod : xs_own;
  __shared__ int rps[512 + 4];   // a row block holds at most 512 rows (fh_mat_build_rowblocks)
  // XCD-aware order (workgroup p runs on XCD p % 8).  q > 0: XCD k walks the row blocks in chunks of `chunk` consecutive blocks
  // (neighbouring blocks share x lines -> L2 reuse), the chunks of the 8 XCDs interleaved so that the device as a whole still
  // advances through the matrix front to back; chunk == q is the fully contiguous split
  int blk = (int)blockIdx.x;
  if (q > 0) {
    const int xcd = (int)(blockIdx.x & 7), pos = (int)(blockIdx.x >> 3);
    blk = ((pos / chunk) * 8 + xcd) * chunk + pos % chunk;
  }
  if (blk >= nblk) return;
  const int tid = threadIdx.x;
  const int4 d0 = *reinterpret_cast<const int4*>(blkinfo + (size_t)blk * 8);
  const int2 d1 = *reinterpret_cast<const int2*>(blkinfo + (size_t)blk * 8 + 4);
  const int r0 = d0.x, r1 = d0.y, s = d0.z, e = d0.w;
  if (r1 - r0 == 1 && e - s > TILE) {
    double acc = 0.0;
    for (int k = s + tid; k < e; k += NT) acc += val[k] * x[col[k]];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((tid & 63) == 0) prod[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
      double tot = 0.0;
      for (int w = 0; w < NT / 64; w++) tot += prod[w];
      spmv_store<MODE>(tot, r0, x, y, b, dinv, omega);
    }
    return;
  }
  // row pointers of the tile: issued now, parked in LDS before the first barrier (no dependent global load in the reduction)
  const int nrows = r1 - r0;
  // epilogue operands of the row this lane will finish (first reduction pass): fetched now, with the stream, instead of as a
  // last dependent round trip after the reduction
  const int G = (nrows <= 16) ? 16 : (nrows <= 32) ? 8 : (nrows <= 64) ? 4 : (nrows <= 128) ? 2 : 1;
  double eb = 0.0, ed = 0.0, ex = 0.0;
  if (MODE >= 1) {
    const int rr0 = tid / G;
    if (rr0 < nrows) {
      if (MODE == 1) eb = y[r0 + rr0];
      if (MODE >= 2) eb = b[r0 + rr0];
      if (MODE == 3) {
        ed = dinv[r0 + rr0];
        ex = x[r0 + rr0];
      }
    }
  }
  int rp0 = 0, rp1 = 0, rp2 = 0;
  if (tid <= nrows) rp0 = rowptr[r0 + tid];
  int rp3 = 0, rp4 = 0;
  if (tid + NT <= nrows) rp1 = rowptr[r0 + tid + NT];
  if (tid + 2 * NT <= nrows) rp2 = rowptr[r0 + tid + 2 * NT];
  if (tid + 3 * NT <= nrows) rp3 = rowptr[r0 + tid + 3 * NT];
  if (tid + 4 * NT <= nrows) rp4 = rowptr[r0 + tid + 4 * NT];
  // ---- matrix stream first (longest latency): 16 B of values + 4 B of local columns per lane and step ----
  const int s2 = s & ~1;
  constexpr int ITER = TILE / (2 * NT) + 1;
  double2 v[ITER];
  ushort2 lc[ITER];
#pragma unroll
  for (int k = 0; k < ITER; k++) {
    const int i = s2 + 2 * tid + k * (2 * NT);
    if (i < e) {
      v[k] = *reinterpret_cast<const double2*>(val + i);
      lc[k] = *reinterpret_cast<const ushort2*>(lcol + i);
    }
  }
  // ---- x entries of this tile -> LDS (sorted distinct columns: neighbouring lanes share cache lines) ----
  const int u0 = d1.x, nu = d1.y;
  for (int j = tid; j < nu; j += NT) xs[j] = x[ucols[u0 + j]];
  if (tid <= nrows) rps[tid] = rp0;
  if (tid + NT <= nrows) rps[tid + NT] = rp1;
  if (tid + 2 * NT <= nrows) rps[tid + 2 * NT] = rp2;
  if (tid + 3 * NT <= nrows) rps[tid + 3 * NT] = rp3;
  if (tid + 4 * NT <= nrows) rps[tid + 4 * NT] = rp4;
  __syncthreads();
  if (SHARE) {
    double p0[ITER], p1[ITER];
#pragma unroll
    for (int k = 0; k < ITER; k++) {
      const int i = s2 + 2 * tid + k * (2 * NT);
      p0[k] = p1[k] = 0.0;
      if (i < e) {
        p0[k] = v[k].x * xs[lc[k].x];
        p1[k] = v[k].y * xs[lc[k].y];
      }
    }
    __syncthreads();                      // every lane has read its x values: the buffer may now hold the products
#pragma unroll
    for (int k = 0; k < ITER; k++) {
      const int i = s2 + 2 * tid + k * (2 * NT);
      if (i < e) {
        if (i >= s) prod[i - s] = p0[k];
        if (i + 1 < e) prod[i + 1 - s] = p1[k];
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < ITER; k++) {
      const int i = s2 + 2 * tid + k * (2 * NT);
      if (i < e) {
        if (i >= s) prod[i - s] = v[k].x * xs[lc[k].x];
        if (i + 1 < e) prod[i + 1 - s] = v[k].y * xs[lc[k].y];
      }
    }
  }
  __syncthreads();
  const int gl = tid & (G - 1);
  const int rows_per_pass = NT / G;
  const int npass = (nrows + rows_per_pass - 1) / rows_per_pass;
  for (int p = 0; p < npass; p++) {
    const int rr = p * rows_per_pass + tid / G;
    const bool live = rr < nrows;
    const int r = r0 + (live ? rr : 0);
    double acc = 0.0;
    if (live) {
      const int a = rps[rr] - s, z = rps[rr + 1] - s;
      for (int k = a + gl; k < z; k += G) acc += prod[k];
    }
    for (int off = G >> 1; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (live && gl == 0) {
      if (p == 0 && MODE >= 1) {
        if (MODE == 1) y[r] = eb + acc;
        else if (MODE == 2) y[r] = eb - acc;
        else y[r] = ex + omega * ed * (eb - acc);
      } else {
        spmv_store<MODE>(acc, r, x, y, b, dinv, omega);
      }
    }
  }
}

template <int TILE, int NT>
static void launch_lx(fh_mat_t A, int mode, const double* x, double* y, const double* b, const double* dinv, double omega,
                      const int* blkinfo = nullptr, int nblk = -1) {
  fh_ctx_t c = A->ctx;
  if (!blkinfo) {          // all row blocks in matrix order; otherwise a sub-list of descriptors (interior / interface split)
    blkinfo = A->d_blkinfo;
    nblk = A->nblk;
  }
  if (nblk <= 0) return;
  int q = 0, grid = nblk, chunk = 1;
  if (c->spmv_xcd_remap && nblk >= 64) {
    // spmv_xcd_remap: 1 = contiguous eighth per XCD, n > 1 = chunks of n consecutive row blocks per XCD, interleaved
    chunk = (c->spmv_xcd_remap == 1) ? (nblk + 7) / 8 : c->spmv_xcd_remap;
    const int per_round = 8 * chunk;
    q = ((nblk + per_round - 1) / per_round) * chunk;     // positions per XCD
    grid = 8 * q;
  }
#define FH_LAUNCH(MODE, SH) \
  hipLaunchKernelGGL((k_spmv_lx<TILE, MODE, SH, NT>), dim3(grid), dim3(NT), 0, c->stream, A->d_rowptr, A->d_col, A->d_lcol, A->d_val, \
                     blkinfo, A->d_ucols, nblk, q, chunk, x, y, b, dinv, omega)
  if (c->spmv_share) {
    switch (mode) {
      case 0: FH_LAUNCH(0, true); break;
      case 1: FH_LAUNCH(1, true); break;
      case 2: FH_LAUNCH(2, true); break;
      default: FH_LAUNCH(3, true); break;
    }
  } else {
    switch (mode) {
      case 0: FH_LAUNCH(0, false); break;
      case 1: FH_LAUNCH(1, false); break;
      case 2: FH_LAUNCH(2, false); break;
      default: FH_LAUNCH(3, false); break;
    }
  }
#undef FH_LAUNCH
}

// interior / interface split for an operator over [owned | ghost] columns (PETSc keeps the two column ranges of an MPIAIJ matrix
// as two matrices so that MatMult can multiply the local part while the scatter is in flight, PetscVector.cpp:203-214 ->
// MatMult; here the row BLOCKS are classified): the 32-byte block descriptors are copied in a permuted order, blocks whose
// rows read no column >= n_own first.  The kernel is unchanged -- every descriptor is self-contained.
static int build_split(fh_mat_t A, int n_own) {
  const int nblk = A->nblk;
  const std::vector<int>& blk = A->h_rowblk;
  std::vector<int> info((size_t)nblk * 8 + 8, 0);
  FH_CHECK_HIP(hipMemcpy(info.data(), A->d_blkinfo, (size_t)nblk * 8 * sizeof(int), hipMemcpyDeviceToHost));
  std::vector<int> order;
  order.reserve(nblk);
  std::vector<char> ghost(nblk, 0);
  for (int b = 0; b < nblk; b++)
    for (int r = blk[b]; r < blk[b + 1] && !ghost[b]; r++)     // columns are sorted inside a row: its last one is its largest
      if (A->h_rowptr[r + 1] > A->h_rowptr[r] && fh_hcol(A)[A->h_rowptr[r + 1] - 1] >= n_own) ghost[b] = 1;
  for (int b = 0; b < nblk; b++)
    if (!ghost[b]) order.push_back(b);
  const int n_int = (int)order.size();
  for (int b = 0; b < nblk; b++)
    if (ghost[b]) order.push_back(b);
  std::vector<int> perm((size_t)nblk * 8 + 8, 0);
  for (int k = 0; k < nblk; k++) memcpy(&perm[(size_t)k * 8], &info[(size_t)order[k] * 8], 8 * sizeof(int));
  if (A->d_blkinfo_split) FH_CHECK_HIP(hipFree(A->d_blkinfo_split));
  A->d_blkinfo_split = nullptr;
  FH_CHECK_HIP(hipMalloc(&A->d_blkinfo_split, perm.size() * sizeof(int)));
  FH_CHECK_HIP(hipMemcpy(A->d_blkinfo_split, perm.data(), perm.size() * sizeof(int), hipMemcpyHostToDevice));
  A->nblk_int = n_int;
  A->split_nown = n_own;
  A->split_tile = A->tile;
  return 0;
}

// ------------------------------------------------------------------------------------------------
// persistent, software-pipelined form of k_spmv_lx: every workgroup walks a contiguous range of row blocks and keeps
// TWO tiles of matrix data in flight (the loads of tile t+1 and the column list of tile t+2 are issued before tile t is
// reduced), so the HBM stream does not drain while a tile waits for its x gather, its barriers and its row reduction.
// LDS is double-buffered by tile parity: two barriers per tile.
// ------------------------------------------------------------------------------------------------
template <int TILE, int MODE>
__global__ __launch_bounds__(256) void k_spmv_lxp(const int* __restrict__ rowptr, const unsigned short* __restrict__ lcol,
                                                  const double* __restrict__ val, const int* __restrict__ rowblk, const int* __restrict__ tile_s,
                                                  const int* __restrict__ uptr, const int* __restrict__ ucols, int nblk, int tpb, int q,
                                                  const double* __restrict__ x, double* __restrict__ y, const double* __restrict__ b,
                                                  const double* __restrict__ dinv, double omega) {
  constexpr int ITER = TILE / 512 + 1;
  constexpr int NX = TILE / 256;
  __shared__ double prod[2][TILE + 2];
  __shared__ double xs[2][TILE];
  __shared__ int rps[2][260];
  const int tid = threadIdx.x;
  const int chunk = (q > 0) ? (int)(blockIdx.x & 7) * q + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int first = chunk * tpb;
  const int last = min(first + tpb, nblk);
  if (first >= last) return;

  // registers of the tile being loaded ("N") and of the tile after it (column list only, "NN")
  double2 vN[ITER];
  ushort2 lN[ITER];
  double xN[NX];
  int rpN = 0, ucNN[NX];
  int sN, eN, r0N, r1N, nuN;

  auto load_ucols = [&](int t, int (&uc)[NX]) {
    const int u0 = uptr[t], nu = uptr[t + 1] - u0;
#pragma unroll
    for (int k = 0; k < NX; k++) {
      const int j = tid + k * 256;
      uc[k] = (j < nu) ? ucols[u0 + j] : -1;
    }
  };
  auto issue_tile = [&](int t, const int (&uc)[NX]) {
    sN = tile_s[t];
    eN = tile_s[t + 1];
    r0N = rowblk[t];
    r1N = rowblk[t + 1];
    nuN = uptr[t + 1] - uptr[t];
    const int s2 = sN & ~1;
#pragma unroll
    for (int k = 0; k < ITER; k++) {
      const int i = s2 + 2 * tid + k * 512;
      if (i < eN) {
        vN[k] = *reinterpret_cast<const double2*>(val + i);
        lN[k] = *reinterpret_cast<const ushort2*>(lcol + i);
      }
    }
#pragma unroll
    for (int k = 0; k < NX; k++) xN[k] = (uc[k] >= 0) ? x[uc[k]] : 0.0;
    rpN = (tid <= r1N - r0N) ? rowptr[r0N + tid] : 0;
  };

  {
    int uc0[NX];
    load_ucols(first, uc0);
    issue_tile(first, uc0);
    if (first + 1 < last) load_ucols(first + 1, ucNN);
  }
  for (int t = first; t < last; t++) {
    // ---- current tile := the one in flight; start the next one ----
    double2 vC[ITER];
    ushort2 lC[ITER];
    double xC[NX];
#pragma unroll
    for (int k = 0; k < ITER; k++) { vC[k] = vN[k]; lC[k] = lN[k]; }
#pragma unroll
    for (int k = 0; k < NX; k++) xC[k] = xN[k];
    const int rpC = rpN, s = sN, e = eN, r0 = r0N, r1 = r1N, nu = nuN;
    if (t + 1 < last) {
      int ucT[NX];
#pragma unroll
      for (int k = 0; k < NX; k++) ucT[k] = ucNN[k];
      if (t + 2 < last) load_ucols(t + 2, ucNN);
      issue_tile(t + 1, ucT);
    }
    // ---- process tile t ----
    const int par = t & 1;
    const int nrows = r1 - r0;
#pragma unroll
    for (int k = 0; k < NX; k++) {
      const int j = tid + k * 256;
      if (j < nu) xs[par][j] = xC[k];
    }
    if (tid <= nrows) rps[par][tid] = rpC;
    __syncthreads();
    const int s2 = s & ~1;
#pragma unroll
    for (int k = 0; k < ITER; k++) {
      const int i = s2 + 2 * tid + k * 512;
      if (i < e) {
        if (i >= s) prod[par][i - s] = vC[k].x * xs[par][lC[k].x];
        if (i + 1 < e) prod[par][i + 1 - s] = vC[k].y * xs[par][lC[k].y];
      }
    }
    __syncthreads();
    const int G = (nrows <= 16) ? 16 : (nrows <= 32) ? 8 : (nrows <= 64) ? 4 : (nrows <= 128) ? 2 : 1;
    const int gl = tid & (G - 1);
    const int rr = tid / G;                       // nrows <= 255 and 256/G >= nrows for every G above: one pass
    const bool live = rr < nrows;
    double acc = 0.0;
    if (live) {
      const int a = rps[par][rr] - s, z = rps[par][rr + 1] - s;
      for (int k = a + gl; k < z; k += G) acc += prod[par][k];
    }
    for (int off = G >> 1; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (live && gl == 0) spmv_store<MODE>(acc, r0 + rr, x, y, b, dinv, omega);
  }
}

template <int TILE>
static void launch_lxp(fh_mat_t A, int mode, const double* x, double* y, const double* b, const double* dinv, double omega) {
  fh_ctx_t c = A->ctx;
  const int target_blocks = c->num_cu * 5;
  int tpb = std::max(1, (A->nblk + target_blocks - 1) / target_blocks);
  int nchunks = (A->nblk + tpb - 1) / tpb;
  int q = 0, grid = nchunks;
  if (c->spmv_xcd_remap && nchunks >= 64) {
    q = (nchunks + 7) / 8;
    grid = 8 * q;
  }
#define FH_LAUNCH(MODE) \
  hipLaunchKernelGGL((k_spmv_lxp<TILE, MODE>), dim3(grid), dim3(256), 0, c->stream, A->d_rowptr, A->d_lcol, A->d_val, A->d_rowblk, A->d_tile_s, \
                     A->d_uptr, A->d_ucols, A->nblk, tpb, q, x, y, b, dinv, omega)
  switch (mode) {
    case 0: FH_LAUNCH(0); break;
    case 1: FH_LAUNCH(1); break;
    case 2: FH_LAUNCH(2); break;
    default: FH_LAUNCH(3); break;
  }
#undef FH_LAUNCH
}

// wave-granular CSR-stream: every wave owns one row block of <= WT non-zeros and runs on its own (no workgroup
// barrier), so slow gathers of one tile do not hold back its neighbours; NT=1 streams (val, col) with non-temporal
// loads so the once-read matrix does not evict x from L2.
template <int WT, int MODE, int NT>
__global__ __launch_bounds__(256) void k_spmv_wave(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                   const double* __restrict__ val, const int* __restrict__ rowblk, int nblk, int q,
                                                   const double* __restrict__ x, double* __restrict__ y,
                                                   const double* __restrict__ b, const double* __restrict__ dinv, double omega) {
  __shared__ double prod_all[4][WT + 2];
  __shared__ int rp_all[4][132];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int pb = blockIdx.x * 4 + wave;     // physical wave-tile id
  // XCD-aware: workgroup p runs on XCD p%8; keep consecutive row blocks on one XCD
  int blk = pb;
  if (q > 0) {
    const int wg = blockIdx.x;
    blk = ((wg & 7) * q + (wg >> 3)) * 4 + wave;
  }
  if (blk >= nblk) return;
  double* prod = prod_all[wave];
  int* rp = rp_all[wave];
  const int r0 = rowblk[blk], r1 = rowblk[blk + 1];
  const int nrows = r1 - r0;
  // row pointers of the tile -> LDS (also gives s, e)
  for (int k = lane; k <= nrows; k += 64) rp[k] = rowptr[r0 + k];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int s = rp[0], e = rp[nrows];
  if (nrows == 1 && e - s > WT) {
    double acc = 0.0;
    for (int k = s + lane; k < e; k += 64) acc += val[k] * x[col[k]];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (lane == 0) spmv_store<MODE>(acc, r0, x, y, b, dinv, omega);
    return;
  }
  const int s2 = s & ~1;
  constexpr int ITER = WT / 128 + 1;
  double2 v[ITER];
  int2 c[ITER];
#pragma unroll
  for (int k = 0; k < ITER; k++) {
    const int i = s2 + 2 * lane + k * 128;
    if (i < e) {
      if (NT) {
        typedef double d2v __attribute__((ext_vector_type(2)));
        typedef int i2v __attribute__((ext_vector_type(2)));
        const d2v vv = __builtin_nontemporal_load(reinterpret_cast<const d2v*>(val + i));
        const i2v cc = __builtin_nontemporal_load(reinterpret_cast<const i2v*>(col + i));
        v[k] = make_double2(vv.x, vv.y);
        c[k] = make_int2(cc.x, cc.y);
      } else {
        v[k] = *reinterpret_cast<const double2*>(val + i);
        c[k] = *reinterpret_cast<const int2*>(col + i);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < ITER; k++) {
    const int i = s2 + 2 * lane + k * 128;
    if (i < e) {
      const double p0 = v[k].x * x[c[k].x];
      if (i >= s) prod[i - s] = p0;
      if (i + 1 < e) prod[i + 1 - s] = v[k].y * x[c[k].y];
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int G = (nrows <= 4) ? 16 : (nrows <= 8) ? 8 : (nrows <= 16) ? 4 : (nrows <= 32) ? 2 : 1;
  const int gl = lane & (G - 1);
  const int rows_per_pass = 64 / G;
  const int npass = (nrows + rows_per_pass - 1) / rows_per_pass;
  for (int p = 0; p < npass; p++) {
    const int rr = p * rows_per_pass + lane / G;
    const bool live = rr < nrows;
    double acc = 0.0;
    if (live) {
      const int a = rp[rr] - s, z = rp[rr + 1] - s;
      for (int k = a + gl; k < z; k += G) acc += prod[k];
    }
    for (int off = G >> 1; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (live && gl == 0) spmv_store<MODE>(acc, r0 + rr, x, y, b, dinv, omega);
  }
}

// classic CSR-vector: LANES lanes per row (kept for A/B measurements and very small matrices)
template <int LANES, int MODE>
__global__ __launch_bounds__(256) void k_spmv_vector(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                     const double* __restrict__ val, int m, const double* __restrict__ x,
                                                     double* __restrict__ y, const double* __restrict__ b,
                                                     const double* __restrict__ dinv, double omega) {
  const int r = (blockIdx.x * 256 + threadIdx.x) / LANES;
  const int gl = threadIdx.x & (LANES - 1);
  const bool live = r < m;
  double acc = 0.0;
  if (live) {
    const int s = rowptr[r], e = rowptr[r + 1];
    for (int k = s + gl; k < e; k += LANES) acc += val[k] * x[col[k]];
  }
#pragma unroll
  for (int off = LANES >> 1; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (live && gl == 0) spmv_store<MODE>(acc, r, x, y, b, dinv, omega);
}

template <int TILE>
static void launch_stream(fh_mat_t A, int mode, const double* x, double* y, const double* b, const double* dinv, double omega) {
  fh_ctx_t c = A->ctx;
  int q = 0, grid = A->nblk;
  if (c->spmv_xcd_remap && A->nblk >= 64) {
    q = (A->nblk + 7) / 8;
    grid = 8 * q;
  }
#define FH_LAUNCH(MODE) \
  hipLaunchKernelGGL((k_spmv_stream<TILE, MODE>), dim3(grid), dim3(256), 0, c->stream, A->d_rowptr, A->d_col, A->d_val, \
                     A->d_rowblk, A->nblk, q, x, y, b, dinv, omega)
  switch (mode) {
    case 0: FH_LAUNCH(0); break;
    case 1: FH_LAUNCH(1); break;
    case 2: FH_LAUNCH(2); break;
    default: FH_LAUNCH(3); break;
  }
#undef FH_LAUNCH
}

int fh_dev_spmv(fh_mat_t A, const double* x, double* y, int mode, const double* b, const double* dinv, double omega) {
  fh_ctx_t c = A->ctx;
  if (A->m == 0) return 0;
  FH_REQUIRE(mode >= 0 && mode <= 3, "fh_spmv: unknown mode %d", mode);
  FH_REQUIRE(x != y, "fh_spmv: x and y must not alias");
  if (c->spmv_kernel == 1) {
    const int grid = fh_div_up((int64_t)A->m * 16, 256);
#define FH_LAUNCHV(MODE) \
  hipLaunchKernelGGL((k_spmv_vector<16, MODE>), dim3(grid), dim3(256), 0, c->stream, A->d_rowptr, A->d_col, A->d_val, A->m, x, y, b, dinv, omega)
    switch (mode) {
      case 0: FH_LAUNCHV(0); break;
      case 1: FH_LAUNCHV(1); break;
      case 2: FH_LAUNCHV(2); break;
      default: FH_LAUNCHV(3); break;
    }
#undef FH_LAUNCHV
  } else if (c->spmv_kernel == 2) {
    if (A->tile != c->spmv_tile || A->tile_kernel != 2) FH_TRY(fh_mat_build_rowblocks(A, c->spmv_tile));
    const int nwg = fh_div_up(A->nblk, 4);
    int q = 0, grid = nwg;
    if (c->spmv_xcd_remap && nwg >= 64) {
      q = (nwg + 7) / 8;
      grid = 8 * q;
    }
    const int nt = c->spmv_nt;
#define FH_LW(WT, MODE, NT) \
  hipLaunchKernelGGL((k_spmv_wave<WT, MODE, NT>), dim3(grid), dim3(256), 0, c->stream, A->d_rowptr, A->d_col, A->d_val, A->d_rowblk, \
                     A->nblk, q, x, y, b, dinv, omega)
#define FH_LW_MODE(WT, NT)                 \
  switch (mode) {                          \
    case 0: FH_LW(WT, 0, NT); break;       \
    case 1: FH_LW(WT, 1, NT); break;       \
    case 2: FH_LW(WT, 2, NT); break;       \
    default: FH_LW(WT, 3, NT); break;      \
  }
    FH_REQUIRE(A->tile <= 1024, "spmv_kernel 2 needs spmv_tile <= 1024");
    if (A->tile == 256) { if (nt) { FH_LW_MODE(256, 1) } else { FH_LW_MODE(256, 0) } }
    else if (A->tile == 512) { if (nt) { FH_LW_MODE(512, 1) } else { FH_LW_MODE(512, 0) } }
    else { if (nt) { FH_LW_MODE(1024, 1) } else { FH_LW_MODE(1024, 0) } }
#undef FH_LW_MODE
#undef FH_LW
  } else if (c->spmv_kernel == 4 && A->max_row <= c->spmv_tile) {
    FH_REQUIRE(c->spmv_tile == 1024 || c->spmv_tile == 2048, "spmv_kernel 4 needs spmv_tile 1024 or 2048");
    if (A->tile != c->spmv_tile || A->tile_kernel != 4) FH_TRY(fh_mat_build_rowblocks(A, c->spmv_tile));
    if (A->lx_tile != A->tile) FH_TRY(fh_mat_build_localcols(A));
    if (A->tile == 1024) launch_lxp<1024>(A, mode, x, y, b, dinv, omega);
    else launch_lxp<2048>(A, mode, x, y, b, dinv, omega);
  } else if (c->spmv_kernel == 3 || c->spmv_kernel == 4) {
    FH_REQUIRE(c->spmv_tile == 1024 || c->spmv_tile == 2048 || c->spmv_tile == 4096, "spmv_kernel 3 needs spmv_tile 1024, 2048 or 4096");
    if (A->tile != c->spmv_tile || (A->tile_kernel != 3 && A->tile_kernel != 4)) FH_TRY(fh_mat_build_rowblocks(A, c->spmv_tile));
    if (A->lx_tile != A->tile) FH_TRY(fh_mat_build_localcols(A));
    if (c->spmv_threads == 128) {
      if (A->tile == 1024) launch_lx<1024, 128>(A, mode, x, y, b, dinv, omega);
      else if (A->tile == 2048) launch_lx<2048, 128>(A, mode, x, y, b, dinv, omega);
      else launch_lx<4096, 128>(A, mode, x, y, b, dinv, omega);
    } else {
      if (A->tile == 1024) launch_lx<1024, 256>(A, mode, x, y, b, dinv, omega);
      else if (A->tile == 2048) launch_lx<2048, 256>(A, mode, x, y, b, dinv, omega);
      else launch_lx<4096, 256>(A, mode, x, y, b, dinv, omega);
    }
  } else {
    if (A->tile != c->spmv_tile || A->tile_kernel != 0) FH_TRY(fh_mat_build_rowblocks(A, c->spmv_tile));
    FH_REQUIRE(A->tile >= 1024, "spmv_kernel 0 needs spmv_tile >= 1024");
    if (A->tile == 1024) launch_stream<1024>(A, mode, x, y, b, dinv, omega);
    else if (A->tile == 2048) launch_stream<2048>(A, mode, x, y, b, dinv, omega);
    else launch_stream<4096>(A, mode, x, y, b, dinv, omega);
  }
  FH_CHECK_HIP(hipGetLastError());
  return 0;
}

int fh_dev_spmv_part(fh_mat_t A, int n_own, int part, const double* x, double* y, int mode, const double* b, const double* dinv, double omega) {
  fh_ctx_t c = A->ctx;
  if (A->m == 0) return 0;
  FH_REQUIRE(mode >= 0 && mode <= 3, "fh_spmv: unknown mode %d", mode);
  FH_REQUIRE(x != y, "fh_spmv: x and y must not alias");
  FH_REQUIRE(part == 0 || part == 1, "fh_dev_spmv_part: part %d", part);
  const bool lx = (c->spmv_kernel == 3 || (c->spmv_kernel == 4 && A->max_row > c->spmv_tile)) && c->spmv_threads == 256 &&
                  (c->spmv_tile == 1024 || c->spmv_tile == 2048 || c->spmv_tile == 4096);
  if (!lx) return part == 0 ? 0 : fh_dev_spmv(A, x, y, mode, b, dinv, omega);     // other kernels: no split, everything after the exchange
  if (A->tile != c->spmv_tile || (A->tile_kernel != 3 && A->tile_kernel != 4)) FH_TRY(fh_mat_build_rowblocks(A, c->spmv_tile));
  if (A->lx_tile != A->tile) FH_TRY(fh_mat_build_localcols(A));
  if (A->split_nown != n_own || A->split_tile != A->tile || !A->d_blkinfo_split) FH_TRY(build_split(A, n_own));
  const int* info = A->d_blkinfo_split + (part == 0 ? 0 : (size_t)A->nblk_int * 8);
  const int nb = part == 0 ? A->nblk_int : A->nblk - A->nblk_int;
  if (A->tile == 1024) launch_lx<1024, 256>(A, mode, x, y, b, dinv, omega, info, nb);
  else if (A->tile == 2048) launch_lx<2048, 256>(A, mode, x, y, b, dinv, omega, info, nb);
  else launch_lx<4096, 256>(A, mode, x, y, b, dinv, omega, info, nb);
  FH_CHECK_HIP(hipGetLastError());
  return 0;
}

/* interior / interface block counts of the split used on distributed levels (diagnostics, tests) */
extern "C" int fh_mat_split_info(fh_mat_t A, int n_own_cols, int* nblk_interior, int* nblk_interface) {
  FH_REQUIRE(A, "fh_mat_split_info: null argument");
  fh_ctx_t c = A->ctx;
  if (A->tile != c->spmv_tile || (A->tile_kernel != 3 && A->tile_kernel != 4)) FH_TRY(fh_mat_build_rowblocks(A, c->spmv_tile));
  if (A->lx_tile != A->tile) FH_TRY(fh_mat_build_localcols(A));
  if (A->split_nown != n_own_cols || A->split_tile != A->tile || !A->d_blkinfo_split) FH_TRY(build_split(A, n_own_cols));
  if (nblk_interior) *nblk_interior = A->nblk_int;
  if (nblk_interface) *nblk_interface = A->nblk - A->nblk_int;
  return 0;
}

extern "C" int fh_spmv(fh_mat_t A, fh_vec_t x, fh_vec_t y, int mode, fh_vec_t b, fh_vec_t dinv, double omega) {
  FH_REQUIRE(A && x && y, "fh_spmv: null argument");
  FH_REQUIRE(x->n_local + x->nghost >= A->n, "fh_spmv: x has %d entries, matrix has %d columns", x->n_local + x->nghost, A->n);
  FH_REQUIRE(y->n_local >= A->m, "fh_spmv: y has %d entries, matrix has %d rows", y->n_local, A->m);
  FH_REQUIRE(mode < 2 || (b && b->n_local >= A->m), "fh_spmv: mode %d needs b", mode);
  // Jacobi sweep: x_new[i] = x[i] + omega dinv[i] (b[i] - (A x)[i]) for the rows of A; on a distributed level A holds the owned
  // rows over [owned | ghost] columns (m <= n) and row i sits at column i
  FH_REQUIRE(mode < 3 || (dinv && dinv->n_local >= A->m && A->m <= A->n),
             "fh_spmv: mode 3 needs dinv and a square matrix (or owned rows over [owned | ghost] columns)");
  return fh_dev_spmv(A, x->d, y->d, mode, b ? b->d : nullptr, dinv ? dinv->d : nullptr, omega);
}

extern "C" int fh_spmv_transpose(fh_mat_t A, fh_vec_t x, fh_vec_t y) {
  FH_REQUIRE(x->n_local >= A->m && y->n_local >= A->n, "fh_spmv_transpose: size mismatch");
  FH_TRY(fh_mat_refresh_transpose(A));
  return fh_dev_spmv(A->At, x->d, y->d, 0, nullptr, nullptr, 0.0);
}
