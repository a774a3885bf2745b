// Geometric multigrid on gfx950 (K8-K13 of SURVEY 2.1; a15-a18 of SURVEY 8).
// Replaces the PETSc PCMG wiring of
//   src/08_algebra.../03_solvers_with_preconditioner/LinearEquationSolverPetsc.cpp:185-353 (MGInit/MGSetLevel/MGSolve)
// with the cycle the reference states itself in LinearImplicitSystem::MGStep (LinearImplicitSystem.cpp:1397-1562):
//   pre-smooth -> residual -> restrict (R = PP^T, LinearImplicitSystem.cpp:379-382) -> recurse -> prolong+add -> post-smooth.
// PETSc behaviours restated: multiplicative V-cycle, smoother = KSPRICHARDSON(scale omega) + PCJACOBI
// with a FIXED iteration count and zero initial guess on the way down (first sweep needs no SpMV), level 0 = exact solve
// (PREONLY + LU, LinearEquationSolverPetsc.hpp:131-134), outer KSP = PREONLY / RICHARDSON(0.99999) / left-preconditioned
// GMRES(restart) with classical Gram-Schmidt and Knoll initial guess (:294-335) / PCG.
//
// MI355X design: every smoother sweep is ONE fused CSR-stream SpMV (matrix read once, x_new = x + omega*dinv*(b - A x));
// restriction uses the explicit transpose (built once) so it is a plain coalesced SpMV; the coarse solve is a dense
// GEMV with the inverse factored once per assembly; the whole cycle (~25 short launches, launch-bound on the coarse
// levels) is captured in a hipGraph and replayed.
#include "fh_internal.h"
#include "fh_trisolve.h"
#include <algorithm>
#include <memory>
#include <cmath>

int fh_dev_get_diag(fh_mat_t A, double* d, int invert);
int fh_halo_update_ptr(fh_halo_t h, double* vd, int n_owned);
int fh_halo_end_ptr(fh_halo_t h);
int fh_halo_allreduce_ptr(fh_halo_t h, double* d, int n);
int fh_direct_solve_ptr(fh_direct_t d, const double* b, double* x);
uint64_t fh_direct_generation(fh_direct_t d);      // changes whenever the object re-analysed its operator (new device buffers, new launch shapes)

struct MgLevel {
  fh_mat_t A = nullptr, P = nullptr, R = nullptr;
  bool R_given = false;      // restriction handed in by the caller; otherwise R = the cached explicit transpose of P, refreshed at every setup
  uint64_t A_uid = 0;        // the matrix the colourings below were built for (pattern caches: dropped when another matrix is installed)
  int n = 0, smoother = 0, npre = 2, npost = 2;
  double omega = 2.0 / 3.0;
  double *dinv = nullptr, *x = nullptr, *x2 = nullptr, *b = nullptr, *r = nullptr;
  // multicolour Gauss-Seidel (SOR) smoother: rows grouped by colour of the matrix graph
  int ncolors = 0;
  std::vector<int> color_ptr;
  int* d_color_rows = nullptr;
  // natural-order sweeps (FH_SMOOTH_SOR, FH_SMOOTH_ILU0): level schedules of the matrix graph, ILU(0) factors
  fh_tri_t tri = nullptr;
  // block Schwarz (Vanka) smoother: dof patches, their colours and dense inverses
  int npatch = 0, vanka_ncolors = 0, max_patch = 0;
  int npatch_exact = 0;       // FH_SMOOTH_ASM: the first npatch_exact blocks get the EXACT sub-solve (MLU_PRECOND on the solid / porous blocks, LinearEquationSolverPetscAsm.cpp:298-307)
  std::vector<int> h_pptr, h_pdofs, vcolor_ptr;
  std::vector<int64_t> h_poff;
  int *d_pptr = nullptr, *d_pdofs = nullptr, *d_porder = nullptr, *d_pflag = nullptr, *d_pcptr = nullptr;
  int4* d_pdesc = nullptr;    // per patch dof: {row, first entry, end of the row, 0} -- one load instead of the chain dof -> row pointer (k_vanka_color_fused)
  unsigned* d_pbar = nullptr;      // arrival / exit counters of the persistent sweep
  int vanka_maxcolor = 0;          // patches of the largest colour
  int pbar_len = 0;
  int64_t* d_poff = nullptr;
  double* d_pinv = nullptr;
  // FH_SMOOTH_ASM (PCASM as the reference configures it): scratch of the block ILU(0) products and the pattern masks of the blocks; order_kind
  // says how d_porder / d_pcptr were made (0: greedy colours, 1: dependency levels of the blocks in their index order)
  double* d_pscr = nullptr;
  unsigned char* d_pmask = nullptr;
  int order_kind = -1;
  // distributed level: operator = owned rows over [owned | ghost] columns; halo refreshes the ghosts
  fh_halo_t halo = nullptr;
  bool replicated_below = false;
  int ncols = 0;
  int buf_n = -1;            // size the work vectors were allocated for (kept across preparations)
  double* buf_base = nullptr;   // dinv, x, x2, b, r: one allocation
  // FH_SMOOTH_LU: B = A^-1 by the sparse exact solve (fh_direct.hip); coordinates of the level's unknowns let it cut at coordinate layers
  fh_direct_t direct = nullptr;
  uint64_t direct_uid = 0;
  std::vector<double> xyz;
  int xyz_dim = 0;
  // level solver: 0 = Richardson(omega) around the sweep preconditioner, 1 = GMRES (fixed iteration count, left-preconditioned)
  int solver = 0, gm_restart = 30, gm_m = 0;
  double* gm_buf = nullptr;  // (gm_m + 1) basis vectors of ncols + 2 entries
  double** gm_dV = nullptr;  // their device pointer table
  double* gm_small = nullptr;   // partial sums, Hessenberg matrix, reduced right-hand side, solution of the least-squares problem
  int gm_nb = 0;             // workgroups of the dot-product launches
};

struct fh_mg_s {
  fh_ctx_t ctx = nullptr;
  int nlevels = 0;
  std::vector<MgLevel> lv;
  double* d_ainv = nullptr;   // dense inverse of the coarsest operator, row-major n0 x n0
  double* d_gjwork = nullptr; // panels of the blocked inversion, kept with d_ainv across preparations
  double* d_gjwork2 = nullptr;   // second pivot-inverse buffer (inside d_gjwork)
  int ainv_n = -1;
  // unknowns of the coarsest level that are coupled to nothing (Dirichlet rows after SetPenalty, whose columns the Galerkin product has
  // emptied too) are solved by their diagonal; the dense inverse holds the na remaining ones.  d_act: their indices, then the others
  int na = -1;
  int* d_act = nullptr;
  int* d_hit = nullptr;       // row / column coupling marks of the last test
  int hit_n = 0;
  std::vector<int> h_act;
  // nested dissection of the coupled unknowns (coarse_nd): [interior block 0 | ... | interior block k-1 | separator], see nd_factor
  std::vector<double> coarse_xyz;     // coordinates of the unknowns of level 0 (fh_mg_set_coarse_coords), [n0 * coarse_dim]
  int coarse_dim = 0;
  std::vector<int> h_act_raw;         // the coupled / uncoupled lists before the dissection reordered the coupled part
  int nd_key = -1, coords_version = 0;
  int cycle_type = 0;                 // FH_CYCLE_*: PCMGSetType
  bool capturable = true;             // no distributed level: the cycle is replayed from a captured graph
  fh_direct_t direct0 = nullptr;      // sparse exact solve of level 0 (more coupled unknowns than the dense inverse holds, or option coarse_direct)
  uint64_t direct0_uid = 0;
  bool direct0_active = false;
  uint64_t nd_A_uid = 0;              // the level-0 matrix the dissection was computed on (another pattern may not be separated by the cached separator)
  bool nd_tables_valid = false;
  std::vector<int> nd_off;            // offsets of the blocks inside the coupled unknowns, nd_off[k] = first separator unknown, nd_off[k + 1] = na
  bool nd_active = false;             // the last factorisation produced the block form (the cycle solves with it)
  double* d_nd = nullptr;             // block inverses, separator inverse, W, W^T, work space
  size_t nd_cap = 0;
  double *d_nd_sinv = nullptr, *d_nd_w = nullptr, *d_nd_wt = nullptr, *d_nd_t = nullptr, *d_nd_xs = nullptr;
  std::vector<double*> nd_dinv;       // per block
  int64_t* d_nd_rowoff = nullptr;     // per interior unknown: where its row of the block inverse starts (doubles from d_nd)
  int* d_nd_rowinfo = nullptr;        // per interior unknown: block offset, block size
  void* d_nd_desc = nullptr;          // per interior block: matrix, size, work space of its inversion (InvDesc), for the batched launches
  int nd_rows_cap = 0;
  std::vector<hipStream_t> nd_streams;
  std::vector<hipEvent_t> nd_events;
  bool setup_done = false;
  hipGraph_t graph = nullptr;
  hipGraphExec_t gexec = nullptr;
  uint64_t graph_sig = 0;     // what the captured cycle was recorded for (every pointer, size and option a launch of the cycle carries)
  // Krylov workspace
  std::vector<double*> kv;
  int kv_n = 0;
  double** d_V = nullptr;     // device copy of the basis pointers (GMRES), kept with kv
  int d_V_n = 0;
  double* d_gm = nullptr;     // state block of the device-resident GMRES (gm_state_doubles(restart)); h_gm = pinned mirror of its header
  double* h_gm = nullptr;
  size_t gm_cap = 0;
  int64_t cycle_bytes = 0;
};

// ------------------------------------------------------------------------------------------------
// small kernels
// ------------------------------------------------------------------------------------------------
// sweep 1 from a zero guess; on a distributed level the interface entries go straight into the send buffer of the exchange that
// follows (the pack kernel fused into the sweep: nsend > 0)
__global__ __launch_bounds__(256) void k_first_sweep(double* __restrict__ x, const double* __restrict__ b, const double* __restrict__ dinv,
                                                     double omega, int n, const int* __restrict__ send_idx, double* __restrict__ sendbuf, int nsend) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) x[i] = omega * dinv[i] * b[i];
  for (int k = blockIdx.x * 256 + threadIdx.x; k < nsend; k += gridDim.x * 256) {
    const int i = send_idx[k];
    sendbuf[k] = omega * dinv[i] * b[i];
  }
}

// one colour of a Gauss-Seidel sweep on A z = r: z_i = dinv_i (r_i - sum_{j != i} a_ij z_j) for the rows of the colour
// (rows of one colour are mutually uncoupled, so the in-place update is race-free); 16 lanes per row
__global__ __launch_bounds__(256) void k_gs_color(const int* __restrict__ rows, int nrows, const int* __restrict__ rowptr, const int* __restrict__ col,
                                                  const double* __restrict__ val, const double* __restrict__ dinv, const double* __restrict__ r,
                                                  double* z) {
  const int rr = (blockIdx.x * 256 + threadIdx.x) >> 4;
  const int gl = threadIdx.x & 15;
  const bool live = rr < nrows;
  const int i = live ? rows[rr] : 0;
  double acc = 0.0;
  if (live)
    for (int k = rowptr[i] + gl; k < rowptr[i + 1]; k += 16) {
      const int j = col[k];
      if (j != i) acc += val[k] * z[j];
    }
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (live && gl == 0) z[i] = dinv[i] * (r[i] - acc);
}

// ------------------------------------------------------------------------------------------------
// block Schwarz (Vanka) smoother for saddle-point systems: x_p += omega A_pp^-1 (b - A x)_p for every patch p, patches of one
// colour concurrently (patches of a colour neither share a dof nor read one another's dofs, so the order inside a colour is
// irrelevant), colours in sequence = multiplicative Schwarz.  GPU form of the element-block ASM smoother the reference
// selects with FEMuS_ASM (petsc_asm/LinearEquationSolverPetscAsm.cpp:91-345).
// ------------------------------------------------------------------------------------------------
// dense copy of A restricted to the patch, stored TRANSPOSED (column-major) so that the apply kernel reads it coalesced
__global__ __launch_bounds__(256) void k_patch_extract(const int* __restrict__ pptr, const int* __restrict__ pdofs, const int64_t* __restrict__ poff,
                                                       const int* __restrict__ rowptr, const int* __restrict__ col, const double* __restrict__ val,
                                                       double* __restrict__ M) {
  const int p = blockIdx.x;
  const int* d = pdofs + pptr[p];
  const int np = pptr[p + 1] - pptr[p];
  double* Mp = M + poff[p];
  for (int t = threadIdx.x; t < np * np; t += 256) {
    const int a = t / np, b = t % np;    // entry (row a, col b) of the patch matrix
    const int r = d[a], c = d[b];
    int lo = rowptr[r], hi = rowptr[r + 1] - 1;
    double v = 0.0;
    while (lo <= hi) {
      const int mid = lo + ((hi - lo) >> 1);   // (lo + hi) overflows beyond 2^30 non-zeros
      const int cc = col[mid];
      if (cc == c) {
        v = val[mid];
        break;
      }
      if (cc < c) lo = mid + 1; else hi = mid - 1;
    }
    Mp[(size_t)a * np + b] = v;
  }
}

// in-place inverse by Gauss-Jordan with partial (row) pivoting, one workgroup per patch, row-major in global memory (the
// patch matrix is L2-resident); on exit the matrix is transposed in place for the apply kernel.  flag[p] = 1: singular.
__global__ __launch_bounds__(256) void k_patch_invert(const int* __restrict__ pptr, const int64_t* __restrict__ poff, double* __restrict__ M,
                                                      int* __restrict__ flag, int skip_upto) {
  __shared__ int piv[512];
  __shared__ double red_v[256];
  __shared__ int red_i[256];
  __shared__ double colk[512];
  const int p = blockIdx.x, tid = threadIdx.x;
  const int n = pptr[p + 1] - pptr[p];
  if (n <= skip_upto) return;                   // inverted by k_patch_invert_lds
  double* A = M + poff[p];
  bool singular = false;
  for (int k = 0; k < n; k++) {
    // pivot search: largest |A[i][k]|, i >= k, smallest index on ties
    double best = -1.0;
    int bi = k;
    for (int i = k + tid; i < n; i += 256) {
      const double v = fabs(A[(size_t)i * n + k]);
      if (v > best) {
        best = v;
        bi = i;
      }
    }
    red_v[tid] = best;
    red_i[tid] = bi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (tid < s) {
        const double v2 = red_v[tid + s];
        const int i2 = red_i[tid + s];
        if (v2 > red_v[tid] || (v2 == red_v[tid] && i2 < red_i[tid])) {
          red_v[tid] = v2;
          red_i[tid] = i2;
        }
      }
      __syncthreads();
    }
    const int pr = red_i[0];
    const double pmax = red_v[0];
    __syncthreads();
    if (tid == 0) piv[k] = pr;
    if (!(pmax > 0.0)) {
      singular = true;
      break;
    }
    if (pr != k)
      for (int j = tid; j < n; j += 256) {
        const double t = A[(size_t)k * n + j];
        A[(size_t)k * n + j] = A[(size_t)pr * n + j];
        A[(size_t)pr * n + j] = t;
      }
    __syncthreads();
    const double pv = 1.0 / A[(size_t)k * n + k];
    for (int i = tid; i < n; i += 256) colk[i] = A[(size_t)i * n + k];
    __syncthreads();
    for (int j = tid; j < n; j += 256) A[(size_t)k * n + j] = (j == k) ? pv : A[(size_t)k * n + j] * pv;
    __syncthreads();
    for (int t = tid; t < n * n; t += 256) {
      const int i = t / n, j = t % n;
      if (i == k) continue;
      const double f = colk[i];
      const double akj = A[(size_t)k * n + j];
      A[t] = (j == k) ? -f * akj : A[t] - f * akj;
    }
    __syncthreads();
  }
  if (singular) {
    if (tid == 0) flag[p] = 1;
    return;
  }
  // undo the row interchanges as column interchanges, last first
  for (int k = n - 1; k >= 0; k--) {
    const int pr = piv[k];
    if (pr != k)
      for (int i = tid; i < n; i += 256) {
        const double t = A[(size_t)i * n + k];
        A[(size_t)i * n + k] = A[(size_t)i * n + pr];
        A[(size_t)i * n + pr] = t;
      }
    __syncthreads();
  }
  // transpose in place
  for (int t = tid; t < n * n; t += 256) {
    const int i = t / n, j = t % n;
    if (i < j) {
      const double a = A[(size_t)i * n + j];
      A[(size_t)i * n + j] = A[(size_t)j * n + i];
      A[(size_t)j * n + i] = a;
    }
  }
}

// the same inversion for patches of at most PLDS_MAX dofs, whole matrix in LDS, ONE wave per patch (no workgroup barriers; the same
// pivot rule -- largest |a_ik|, smallest row on ties -- and the same arithmetic per entry as k_patch_invert, so both give the same bits)
constexpr int PLDS_MAX = 96;
__device__ __forceinline__ void wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__global__ __launch_bounds__(64) void k_patch_invert_lds(const int* __restrict__ pptr, const int64_t* __restrict__ poff, double* __restrict__ M,
                                                         int* __restrict__ flag, int nmax) {
  extern __shared__ double pl_smem[];
  const int p = blockIdx.x, lane = threadIdx.x;
  const int n = pptr[p + 1] - pptr[p];
  if (n > PLDS_MAX) return;                      // left to k_patch_invert
  const int ld = n | 1;
  double* As = pl_smem;                          // [n][ld]
  double* colk = pl_smem + nmax * (nmax | 1);    // nmax: the largest patch this launch inverts (sizes the LDS of every workgroup)
  int* piv = reinterpret_cast<int*>(colk + nmax);
  double* A = M + poff[p];
  for (int t = lane; t < n * n; t += 64) As[(t / n) * ld + t % n] = A[t];
  wave_sync_lds();
  for (int k = 0; k < n; k++) {
    double best = -1.0, bval = 0.0;
    int bi = k;
    for (int i = k + lane; i < n; i += 64) {
      const double v = As[i * ld + k];
      if (fabs(v) > best) {
        best = fabs(v);
        bval = v;
        bi = i;
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const double v2 = __shfl_xor(best, off, 64), w2 = __shfl_xor(bval, off, 64);
      const int i2 = __shfl_xor(bi, off, 64);
      if (v2 > best || (v2 == best && i2 < bi)) {
        best = v2;
        bval = w2;
        bi = i2;
      }
    }
    const int pr = bi;
    if (!(best > 0.0)) {
      if (lane == 0) flag[p] = 1;
      return;
    }
    if (lane == 0) piv[k] = pr;
    const double pv = 1.0 / bval;
    // column k of the other rows (after the interchange), then the interchange and the scaled pivot row, every lane its own columns
    for (int i = lane; i < n; i += 64) colk[i] = (i == pr) ? As[k * ld + k] : As[i * ld + k];
    wave_sync_lds();
    for (int j = lane; j < n; j += 64) {
      const double akj = As[pr * ld + j];
      if (pr != k) As[pr * ld + j] = As[k * ld + j];
      As[k * ld + j] = (j == k) ? pv : akj * pv;
    }
    wave_sync_lds();
    for (int j = lane; j < n; j += 64) {
      const double akj = As[k * ld + j];
      for (int i0 = 0; i0 < n; i0 += 8) {          // eight rows at a time: all loads issued before the first store
        double a[8], f[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int i = min(i0 + u, n - 1);
          a[u] = As[i * ld + j];
          f[u] = colk[i];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int i = i0 + u;
          if (i < n && i != k) As[i * ld + j] = (j == k) ? -f[u] * akj : a[u] - f[u] * akj;
        }
      }
    }
    wave_sync_lds();
  }
  // undo the row interchanges as column interchanges, last first
  for (int k = n - 1; k >= 0; k--) {
    const int pr = piv[k];
    if (pr != k)
      for (int i = lane; i < n; i += 64) {
        const double t = As[i * ld + k];
        As[i * ld + k] = As[i * ld + pr];
        As[i * ld + pr] = t;
      }
    wave_sync_lds();
  }
  // transposed back to global memory
  for (int t = lane; t < n * n; t += 64) A[t] = As[(t % n) * ld + t / n];
}

// one colour of the sweep; one workgroup of 64 per patch.  r = b - A x of the whole level is formed once per colour by the
// fused SpMV (patches of a colour do not read each other's dofs, so it is the exact residual for every patch of the colour);
// the patch gathers its rows of r, applies the dense inverse and updates x
__global__ __launch_bounds__(64) void k_vanka_color(const int* __restrict__ order, int npat, const int* __restrict__ pptr, const int* __restrict__ pdofs,
                                                    const int64_t* __restrict__ poff, const double* __restrict__ Minv, const double* __restrict__ r,
                                                    double* x, double omega) {
  extern __shared__ double rp[];
  if ((int)blockIdx.x >= npat) return;
  const int p = order[blockIdx.x], lane = threadIdx.x;
  const int* d = pdofs + pptr[p];
  const int np = pptr[p + 1] - pptr[p];
  for (int a = lane; a < np; a += 64) rp[a] = r[d[a]];
  __syncthreads();
  const double* Mi = Minv + poff[p];     // transposed inverse: Mi[b * np + a] = inv[a][b]
  for (int a = lane; a < np; a += 64) {
    double s = 0.0;
    for (int c = 0; c < np; c++) s += Mi[(size_t)c * np + a] * rp[c];
    x[d[a]] += omega * s;
  }
}

// The same colour in ONE launch (round 5, default: option vanka_fused 1): every patch forms the residual of ITS OWN rows (4 lanes per row, 16 rows at a
// time) instead of reading a residual of the whole level that a separate SpMV launch has just made -- exact for the colour, because its patches do not
// read each other's dofs.  Half the launches of a sweep (the cycles of config 4 are launch-latency bound: ~380 launches of 4-9 us) and none of the
// residual rows nobody reads.  One workgroup of four waves per patch: 8 lanes per row, the dense inverse applied a quarter of the columns per wave.
__global__ __launch_bounds__(256) void k_patch_desc(int n, const int* __restrict__ pdofs, const int* __restrict__ rowptr, int4* __restrict__ desc) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  const int r = pdofs[k];
  desc[k] = make_int4(r, rowptr[r], rowptr[r + 1], 0);
}
__global__ __launch_bounds__(256) void k_vanka_color_fused(const int* __restrict__ order, int npat, const int* __restrict__ pptr, const int4* __restrict__ pdesc,
                                                           const int64_t* __restrict__ poff, const double* __restrict__ Minv, const int* __restrict__ col,
                                                           const double* __restrict__ val, const double* __restrict__ b, double* x, double omega, int max_patch) {
  extern __shared__ double vf_smem[];
  double* rp = vf_smem;                          // [max_patch] residual of the patch rows
  double* part = vf_smem + max_patch;            // [4][max_patch] partial products of the four waves
  if ((int)blockIdx.x >= npat) return;
  const int p = order[blockIdx.x], tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, sub = tid & 7, rl = tid >> 3;
  const int p0 = pptr[p], np = pptr[p + 1] - p0;
  const int4* d = pdesc + p0;
  const double* Mi = Minv + poff[p];             // transposed inverse: Mi[c * np + a] = inv[a][c]
  const int c0 = (np * wave) >> 2, c1 = (np * (wave + 1)) >> 2;          // this wave's quarter of the columns
  // the step is a chain of dependent memory round trips (cycles of config 4: 192 colour steps of 15-25 us): what does not depend on x goes out first --
  // this thread's entries of the inverse (patches of <= 64 dofs: at most sixteen), the row descriptors ({row, first, end} in one load)
  constexpr int MR = 16;
  double mreg[MR];
  const bool small = np <= 64;
  if (small) {
#pragma unroll
    for (int q = 0; q < MR; q++) mreg[q] = (lane < np && c0 + q < c1) ? Mi[(size_t)(c0 + q) * np + lane] : 0.0;
  }
  for (int a0 = 0; a0 < np; a0 += 32) {          // 32 rows at a time, 8 lanes per row (one wave per patch and 4 lanes per row was latency bound: 184 instead
    const int a = a0 + rl;                       //  of 120 ms per linear solve of config 4)
    double acc = 0.0, bb = 0.0;
    if (a < np) {
      const int4 q = d[a];
      bb = b[q.x];
      for (int kk = q.y + sub; kk < q.z; kk += 8) acc += val[kk] * x[col[kk]];
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    if (a < np && sub == 0) rp[a] = bb - acc;
  }
  __syncthreads();
  if (small) {
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < MR; q++) s += mreg[q] * rp[min(c0 + q, np - 1)];          // (entries beyond c1 are zero)
    if (lane < np) part[wave * max_patch + lane] = s;
  } else {
    for (int a = lane; a < np; a += 64) {
      double s = 0.0;
      for (int c = c0; c < c1; c++) s += Mi[(size_t)c * np + a] * rp[c];
      part[wave * max_patch + a] = s;
    }
  }
  __syncthreads();
  for (int a = tid; a < np; a += 256) x[d[a].x] += omega * (((part[a] + part[max_patch + a]) + part[2 * max_patch + a]) + part[3 * max_patch + a]);
}

// ALL colours of ALL sweeps in one launch (fh_set_option(vanka_persistent, 1 | 2); off by default): a grid of resident one-wave workgroups walks
// the colours together, a device-wide barrier between two colours instead of a launch boundary.  Every patch forms the residual
// of its own rows (4 lanes per row, shuffle reduction), exact for the colour because its patches do not read each other's dofs;
// pass A of a step (patch dofs, row extents: independent of x) is issued BEFORE the barrier wait, so that after the barrier only
// the x-dependent part is on the critical path.  The grid is sized by the host to fit the device many times over (one wave and
// <= 4 KB of LDS per workgroup), which is what makes the spin barrier safe.  bar[0]: arrivals (monotone), bar[1]: exits; the
// last workgroup to leave zeroes both, so the next launch finds them clean.
// Barrier of the resident grid.  MODE 1: one arrival counter (bar[0], monotone over the launch; bar[1] counts exits and the last
// workgroup out zeroes both).  MODE 2: one flag per workgroup (plain stores, no read-modify-write on a shared address), workgroup 0
// polls them 64 at a time and publishes the step in bar[0]; at the end every workgroup clears its flag, workgroup 0 then bar[0].
__device__ __forceinline__ void wave_lds_sync() {      // LDS written by one lane, read by another lane of the same wave
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int MODE>
__device__ __forceinline__ void vanka_grid_barrier(unsigned* bar, unsigned step) {
  __threadfence();                                   // release: this workgroup's x updates reach the other XCDs' view
  __syncthreads();
  if (MODE == 1) {
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = step * gridDim.x;
      while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
  } else {
    unsigned* flags = bar + 2;
    if (threadIdx.x == 0) __hip_atomic_store(flags + blockIdx.x, step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (blockIdx.x == 0) {
      if (threadIdx.x < 64) {
        for (unsigned w = threadIdx.x; w < gridDim.x; w += 64)
          while (__hip_atomic_load(flags + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != step) __builtin_amdgcn_s_sleep(1);
      }
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_store(bar, step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (threadIdx.x == 0) {
      while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != step) __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
  __threadfence();                                   // acquire: drop stale lines of x before the next colour reads it
}

template <int MODE>
__global__ __launch_bounds__(256) void k_vanka_persistent(const int* __restrict__ order, const int* __restrict__ cptr, int ncolors, int nsweeps,
                                                          const int* __restrict__ pptr, const int* __restrict__ pdofs,
                                                          const int64_t* __restrict__ poff, const double* __restrict__ Minv,
                                                          const int* __restrict__ rowptr, const int* __restrict__ col, const double* __restrict__ val,
                                                          const double* __restrict__ b, double* x, double omega, unsigned* bar, int max_patch) {
  extern __shared__ double rp_all[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, sub = lane & 3, rl = lane >> 2;
  double* rp = rp_all + (size_t)wave * max_patch;     // one patch per wave
  const int nsteps = nsweeps * ncolors;
  for (int st = 0; st < nsteps; st++) {
    const int k = st % ncolors;
    const int c0 = cptr[k], npat = cptr[k + 1] - c0;
    if (st > 0) vanka_grid_barrier<MODE>(bar, (unsigned)st);
    for (int q = blockIdx.x * 4 + wave; q < npat; q += gridDim.x * 4) {
      const int p = order[c0 + q];
      const int* d = pdofs + pptr[p];
      const int np = pptr[p + 1] - pptr[p];
      for (int a0 = 0; a0 < np; a0 += 16) {            // 16 rows at a time, 4 lanes per row
        const int a = a0 + rl;
        double acc = 0.0;
        int row = 0;
        if (a < np) {
          row = d[a];
          const int ke = rowptr[row + 1];
          for (int kk = rowptr[row] + sub; kk < ke; kk += 4) acc += val[kk] * x[col[kk]];
        }
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        if (a < np && sub == 0) rp[a] = b[row] - acc;
      }
      wave_lds_sync();
      const double* Mi = Minv + poff[p];               // transposed inverse: Mi[c * np + a] = inv[a][c]
      for (int a = lane; a < np; a += 64) {
        double s = 0.0;
        for (int c = 0; c < np; c++) s += Mi[(size_t)c * np + a] * rp[c];
        x[d[a]] += omega * s;
      }
      wave_lds_sync();
    }
  }
  // leave with clean counters for the next launch
  if (MODE == 1) {
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned left = __hip_atomic_fetch_add(bar + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (left == gridDim.x - 1) {
        __hip_atomic_store(bar, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(bar + 1, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  } else if (nsteps > 1) {
    vanka_grid_barrier<MODE>(bar, (unsigned)nsteps);  // everybody has read the last published step ...
    vanka_grid_barrier<MODE>(bar, 0u);                // ... and the flags and the step go back to zero
  }
}

// y = Ainv b, one wave per row, 16-byte loads
__global__ __launch_bounds__(256) void k_dense_gemv(const double* __restrict__ M, const double* __restrict__ b, double* __restrict__ y, int n) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n) return;
  const double* m = M + (size_t)row * n;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;          // four loads of the row in flight per lane
  int k = lane;
  for (; k + 192 < n; k += 256) {
    a0 += m[k] * b[k];
    a1 += m[k + 64] * b[k + 64];
    a2 += m[k + 128] * b[k + 128];
    a3 += m[k + 192] * b[k + 192];
  }
  for (; k < n; k += 64) a0 += m[k] * b[k];
  double acc = (a0 + a1) + (a2 + a3);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (lane == 0) y[row] = acc;
}

__global__ __launch_bounds__(256) void k_csr_to_dense(const int* __restrict__ rowptr, const int* __restrict__ col, const double* __restrict__ val,
                                                      double* __restrict__ D, int n) {
  const int row = blockIdx.x;
  for (int k = rowptr[row] + threadIdx.x; k < rowptr[row + 1]; k += 256) D[(size_t)row * n + col[k]] = val[k];
}

// ------------------------------------------------------------------------------------------------
// dense inverse of the coarsest operator: BLOCKED in-place Gauss-Jordan without pivoting (the operator is SPD on the free
// dofs and the identity on Dirichlet rows).  Per pivot block of NB columns: save the column panel, invert the NB x NB pivot
// in LDS, form the new row panel D^-1 A[k,:], rank-NB update of all other rows as a tiled FP64 GEMM (64x64 tiles, 4x4
// register blocks, operands staged in LDS), and the pivot-column panel -C D^-1.  2 n^3 flops in n/NB steps of 5 launches
// instead of 3 n launches of rank-1 updates.
// ------------------------------------------------------------------------------------------------
constexpr int GJ_NB = 32;   // pivot block (64 measured slower twice, also with the MFMA update: pivot-block inversion 34 -> 192 us, row panel 38 -> 144 us per step)
constexpr int GJ_KS = 32;   // K slice of the update staged in LDS at a time

// in-place inverse of the NB x NB block M (LDS, row stride NB + 1; rows / columns >= nb are identity padding) by Gauss-Jordan with
// PARTIAL PIVOTING, all 256 threads of the workgroup.  The inverse of a block does not depend on how it is computed, so the
// callers (general and symmetric sweeps) are unchanged; what pivoting buys is a stable inverse of blocks that are not positive
// definite -- saddle-point operators carry zero diagonal entries (the reference factors level 0 with a pivoted LU,
// LinearEquationSolverPetsc.hpp:131-134).  A pivot column without any entry above 1e-300 raises *flag (singular block).
__device__ __forceinline__ void gj_invert_block(double (*M)[GJ_NB + 1], double* colk, int* piv, int nb, int tid, int* flag) {
  for (int k = 0; k < nb; k++) {
    if (tid < 64) {            // wave 0: largest |M[i][k]|, i in [k, nb), smallest index on ties
      double v = (tid >= k && tid < nb) ? fabs(M[tid][k]) : -1.0;
      int idx = tid;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const double v2 = __shfl_xor(v, off, 64);
        const int i2 = __shfl_xor(idx, off, 64);
        if (v2 > v || (v2 == v && i2 < idx)) {
          v = v2;
          idx = i2;
        }
      }
      if (tid == 0) {
        piv[k] = idx;
        if (!(v > 1e-300)) atomicOr(flag, 1);
      }
    }
    __syncthreads();
    const int pr = piv[k];
    if (pr != k && tid < GJ_NB) {
      const double t = M[k][tid];
      M[k][tid] = M[pr][tid];
      M[pr][tid] = t;
    }
    __syncthreads();
    if (tid < GJ_NB) colk[tid] = M[tid][k];
    __syncthreads();
    const double p = 1.0 / colk[k];
#pragma unroll
    for (int idx = tid; idx < GJ_NB * GJ_NB; idx += 256) {
      const int i = idx / GJ_NB, j = idx % GJ_NB;
      if (i != k) {
        const double f = colk[i] * p;
        M[i][j] = (j == k) ? -f : M[i][j] - f * M[k][j];
      }
    }
    __syncthreads();
    if (tid < GJ_NB) M[k][tid] = (tid == k) ? p : M[k][tid] * p;
    __syncthreads();
  }
  for (int k = nb - 1; k >= 0; k--) {      // the row interchanges come back as column interchanges, last first
    const int pr = piv[k];
    if (pr != k && tid < GJ_NB) {
      const double t = M[tid][k];
      M[tid][k] = M[tid][pr];
      M[tid][pr] = t;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_check_finite(const double* __restrict__ D, size_t n, int* __restrict__ flag) {
  bool bad = false;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) bad |= !isfinite(D[i]);
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 2);
}

__global__ __launch_bounds__(256) void k_gjb_save_panel(const double* __restrict__ D, double* __restrict__ Cp, double* __restrict__ CpT, int n, int kb, int nb) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= n * nb) return;
  const int i = idx / nb, t = idx % nb;
  const double v = D[(size_t)i * n + kb + t];
  Cp[(size_t)i * GJ_NB + t] = v;
  CpT[(size_t)t * n + i] = v;
}

// one workgroup; a single-wave version (no workgroup barriers) was measured 3x slower: 16 LDS read-modify-writes per lane and step
// instead of 4.  Index arithmetic on the compile-time block size (the run-time nb only guards).
__global__ __launch_bounds__(256) void k_gjb_pivot(const double* __restrict__ D, double* __restrict__ Dinv, int n, int kb, int nb, int* __restrict__ flag) {
  __shared__ double M[GJ_NB][GJ_NB + 1];
  __shared__ double colk[GJ_NB];
  __shared__ int piv[GJ_NB];
  const int tid = threadIdx.x;
  for (int idx = tid; idx < GJ_NB * GJ_NB; idx += 256) {
    const int i = idx / GJ_NB, j = idx % GJ_NB;
    M[i][j] = (i < nb && j < nb) ? D[(size_t)(kb + i) * n + kb + j] : (i == j ? 1.0 : 0.0);   // identity padding: inert
  }
  __syncthreads();
  gj_invert_block(M, colk, piv, nb, tid, flag);
  for (int idx = tid; idx < nb * nb; idx += 256) Dinv[(idx / nb) * GJ_NB + idx % nb] = M[idx / nb][idx % nb];
}

// rows of the pivot block: A[kb+s, j] <- sum_t Dinv[s,t] * A_old[kb+t, j] (j outside the pivot columns), Dinv inside
__global__ __launch_bounds__(64) void k_gjb_row_panel(double* __restrict__ D, const double* __restrict__ Dinv, const double* __restrict__ Cp,
                                                      int n, int kb, int nb) {
  __shared__ double Ds[GJ_NB][GJ_NB + 1];
  const int tid = threadIdx.x;
  for (int idx = tid; idx < GJ_NB * GJ_NB; idx += 64) {
    const int a = idx / GJ_NB, b = idx % GJ_NB;
    Ds[a][b] = (a < nb && b < nb) ? Dinv[a * GJ_NB + b] : 0.0;
  }
  __syncthreads();
  const int j = blockIdx.x * 64 + tid;
  if (j >= n) return;
  if (j >= kb && j < kb + nb) {
    for (int s2 = 0; s2 < nb; s2++) D[(size_t)(kb + s2) * n + j] = Ds[s2][j - kb];
    return;
  }
  double old[GJ_NB];        // compile-time trip counts: with the run-time bound nb the array lived in scratch memory
#pragma unroll
  for (int t = 0; t < GJ_NB; t++) old[t] = (t < nb) ? D[(size_t)(kb + t) * n + j] : 0.0;
  for (int s2 = 0; s2 < nb; s2++) {
    double acc = 0.0;
#pragma unroll
    for (int t = 0; t < GJ_NB; t++) acc += Ds[s2][t] * old[t];
    D[(size_t)(kb + s2) * n + j] = acc;
  }
}

// all other rows, columns outside the pivot block: A[i,j] -= sum_t Cp[i,t] * R[t,j]   (R = the new row panel)
__global__ __launch_bounds__(256) void k_gjb_update(double* __restrict__ D, const double* __restrict__ Cp, int n, int kb, int nb) {
  __shared__ double Cs[64][GJ_KS + 1];
  __shared__ double Rs[GJ_KS][64 + 2];
  const int tid = threadIdx.x;
  const int ti = blockIdx.y * 64, tj = blockIdx.x * 64;
  const int ty = tid >> 4, tx = tid & 15;
  double acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b2 = 0; b2 < 4; b2++) acc[a][b2] = 0.0;
  for (int t0 = 0; t0 < nb; t0 += GJ_KS) {
    for (int idx = tid; idx < 64 * GJ_KS; idx += 256) {
      const int r = idx / GJ_KS, t = t0 + idx % GJ_KS;
      const int i = ti + r;
      Cs[r][idx % GJ_KS] = (i < n && t < nb) ? Cp[(size_t)i * GJ_NB + t] : 0.0;
    }
    for (int idx = tid; idx < GJ_KS * 64; idx += 256) {
      const int t = t0 + idx / 64, c = idx % 64;
      const int j = tj + c;
      Rs[idx / 64][c] = (j < n && t < nb) ? D[(size_t)(kb + t) * n + j] : 0.0;
    }
    __syncthreads();
#pragma unroll 8
    for (int t = 0; t < GJ_KS; t++) {
      double cv[4], rv[4];
#pragma unroll
      for (int a = 0; a < 4; a++) cv[a] = Cs[ty * 4 + a][t];
#pragma unroll
      for (int b2 = 0; b2 < 4; b2++) rv[b2] = Rs[t][tx * 4 + b2];
#pragma unroll
      for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b2 = 0; b2 < 4; b2++) acc[a][b2] += cv[a] * rv[b2];
    }
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < 4; a++) {
    const int i = ti + ty * 4 + a;
    if (i >= n || (i >= kb && i < kb + nb)) continue;
#pragma unroll
    for (int b2 = 0; b2 < 4; b2++) {
      const int j = tj + tx * 4 + b2;
      if (j >= n || (j >= kb && j < kb + nb)) continue;
      D[(size_t)i * n + j] -= acc[a][b2];
    }
  }
}

// the same update on the FP64 matrix cores (v_mfma_f64_16x16x4): 64 x 64 output tile per workgroup, 32 x 32 per wave as 2 x 2
// MFMA tiles, K slices of 32 staged in LDS k-major (row stride 80 doubles = 16 mod 32: conflict-free fragment reads; A fragment:
// lane = 16 k + i, B fragment: lane = 16 k + j, C: col = lane & 15, row = (lane >> 4) + 4 reg).  CpT = the saved column panel
// transposed (t-major), so that both operands load coalesced.
typedef double gj_d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_gjb_update_mfma(double* __restrict__ D, const double* __restrict__ CpT, int n, int kb, int nb) {
  constexpr int LD = 80;
  __shared__ double Cs[GJ_KS][LD], Rs[GJ_KS][LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ti = blockIdx.y * 64, tj = blockIdx.x * 64;
  const int wi = (wave >> 1) * 32, wj = (wave & 1) * 32;
  const int kk = lane >> 4, li = lane & 15;
  // the tile of D this workgroup updates is read FIRST (its HBM latency then overlaps the operand staging and the MFMAs) and
  // serves as the accumulator: D - Cp R = D + (-Cp) R, the sign goes onto the A operand
  gj_d4 acc[2][2];
  bool live[2][4][2];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int i = ti + wi + a * 16 + kk + 4 * r;
#pragma unroll
      for (int b = 0; b < 2; b++) {
        const int j = tj + wj + b * 16 + li;
        live[a][r][b] = i < n && j < n && !(i >= kb && i < kb + nb) && !(j >= kb && j < kb + nb);
        acc[a][b][r] = live[a][r][b] ? D[(size_t)i * n + j] : 0.0;
      }
    }
  for (int t0 = 0; t0 < nb; t0 += GJ_KS) {
    for (int idx = tid; idx < GJ_KS * 64; idx += 256) {
      const int k = idx >> 6, c = idx & 63, t = t0 + k;
      Cs[k][c] = (ti + c < n && t < nb) ? -CpT[(size_t)t * n + ti + c] : 0.0;
      Rs[k][c] = (tj + c < n && t < nb) ? D[(size_t)(kb + t) * n + tj + c] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int k0 = 0; k0 < GJ_KS; k0 += 4) {
      const double a0 = Cs[k0 + kk][wi + li], a1 = Cs[k0 + kk][wi + 16 + li];
      const double b0 = Rs[k0 + kk][wj + li], b1 = Rs[k0 + kk][wj + 16 + li];
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int i = ti + wi + a * 16 + kk + 4 * r;
#pragma unroll
      for (int b = 0; b < 2; b++) {
        const int j = tj + wj + b * 16 + li;
        if (live[a][r][b]) D[(size_t)i * n + j] = acc[a][b][r];
      }
    }
}

// ------------------------------------------------------------------------------------------------
// SYMMETRIC coarse operators (Poisson, AMR: checked entry by entry before use): the sweep operator on pivot blocks,
//   A_kk <- -A_kk^-1,   A_ko <- A_kk^-1 A_ko (and its transpose),   A_oo <- A_oo - A_ok A_kk^-1 A_ko,
// keeps the working matrix symmetric through all steps and ends in -A^-1, so only the UPPER block triangle is updated: half the
// flops and half the HBM traffic of the general Gauss-Jordan above.  PT = the old pivot rows for ALL columns (taken from the
// rows right of the pivot block and from the columns above it), RT = the new row panel, both k-major for the MFMA fragments.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_csr_symmetry(const int* __restrict__ rowptr, const int* __restrict__ col, const double* __restrict__ val, int n,
                                                      double tol, int* __restrict__ flag) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;     // one wave per row
  if (i >= n) return;
  double dmax = 0.0;
  for (int k = rowptr[i] + lane; k < rowptr[i + 1]; k += 64) dmax = fmax(dmax, fabs(val[k]));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) dmax = fmax(dmax, __shfl_xor(dmax, off, 64));
  for (int k = rowptr[i] + lane; k < rowptr[i + 1]; k += 64) {
    const int j = col[k];
    if (j == i) continue;
    int lo = rowptr[j], hi = rowptr[j + 1] - 1;
    double vt = 0.0;
    while (lo <= hi) {
      const int mid = lo + ((hi - lo) >> 1);   // (lo + hi) overflows beyond 2^30 non-zeros
      if (col[mid] == i) { vt = val[mid]; break; }
      if (col[mid] < i) lo = mid + 1; else hi = mid - 1;
    }
    if (fabs(val[k] - vt) > tol * dmax) atomicOr(flag, 1);
  }
}

__global__ __launch_bounds__(256) void k_gjs_gather_panel(const double* __restrict__ D, double* __restrict__ PT, int n, int kb, int nb) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= n * GJ_NB) return;
  const int j = idx / GJ_NB, t = idx % GJ_NB;            // t fastest: for j < kb the 32 entries D[j][kb..] are contiguous
  if (t >= nb) return;
  double v;
  if (j < kb) v = D[(size_t)j * n + kb + t];             // column above the pivot block (upper triangle)
  else if (j < kb + nb) v = D[(size_t)(kb + min(t, j - kb)) * n + kb + max(t, j - kb)];
  else v = D[(size_t)(kb + t) * n + j];                  // row right of the pivot block
  PT[(size_t)t * n + j] = v;
}

__global__ __launch_bounds__(256) void k_gjs_pivot(const double* __restrict__ PT, double* __restrict__ Dinv, int n, int kb, int nb, int* __restrict__ flag) {
  __shared__ double M[GJ_NB][GJ_NB + 1];
  __shared__ double colk[GJ_NB];
  __shared__ int piv[GJ_NB];
  const int tid = threadIdx.x;
  for (int idx = tid; idx < GJ_NB * GJ_NB; idx += 256) {
    const int i = idx / GJ_NB, j = idx % GJ_NB;
    M[i][j] = (i < nb && j < nb) ? PT[(size_t)i * n + kb + j] : (i == j ? 1.0 : 0.0);
  }
  __syncthreads();
  gj_invert_block(M, colk, piv, nb, tid, flag);
  for (int idx = tid; idx < nb * nb; idx += 256) Dinv[(idx / nb) * GJ_NB + idx % nb] = M[idx / nb][idx % nb];
}

// new row panel R = Dinv * PT for the columns outside the pivot block -> RT, the matrix row (j right of the block), the matrix
// column (j above it: the transpose); -Dinv into the pivot block
__global__ __launch_bounds__(256) void k_gjs_row_panel(double* __restrict__ D, const double* __restrict__ Dinv, const double* __restrict__ PT,
                                                       double* __restrict__ RT, int n, int kb, int nb) {
  // 64 columns x 4 groups of 8 output rows per workgroup (one wave per group): 4x the waves of a column-per-thread layout
  __shared__ double Ds[GJ_NB][GJ_NB + 1];
  const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
  for (int idx = tid; idx < GJ_NB * GJ_NB; idx += 256) {
    const int a = idx / GJ_NB, b = idx % GJ_NB;
    Ds[a][b] = (a < nb && b < nb) ? Dinv[a * GJ_NB + b] : 0.0;
  }
  __syncthreads();
  const int j = blockIdx.x * 64 + tx;
  if (j >= n) return;
  const int s_lo = ty * (GJ_NB / 4), s_hi = min(nb, s_lo + GJ_NB / 4);
  if (j >= kb && j < kb + nb) {
    for (int s2 = s_lo; s2 < s_hi; s2++) {
      D[(size_t)(kb + s2) * n + j] = -Ds[s2][j - kb];
      RT[(size_t)s2 * n + j] = 0.0;
    }
    return;
  }
  double old[GJ_NB];
#pragma unroll
  for (int t = 0; t < GJ_NB; t++) old[t] = (t < nb) ? PT[(size_t)t * n + j] : 0.0;
  for (int s2 = s_lo; s2 < s_hi; s2++) {
    double acc = 0.0;
#pragma unroll
    for (int t = 0; t < GJ_NB; t++) acc += Ds[s2][t] * old[t];
    RT[(size_t)s2 * n + j] = acc;
    if (j > kb) D[(size_t)(kb + s2) * n + j] = acc;
    else D[(size_t)j * n + kb + s2] = acc;
  }
}

// upper block triangle: A[i][j] -= sum_t PT[t][i] * RT[t][j]   (i, j outside the pivot block)
// Look-ahead: the workgroup that owns the diagonal tile with the NEXT pivot block inverts that block right after its update
// (the tile order is rotated so that it is scheduled first), which takes the sequential 32-step inversion (23 us) off the
// critical path of every step but the first.
__global__ __launch_bounds__(256) void k_gjs_update_mfma(double* __restrict__ D, const double* __restrict__ PT, const double* __restrict__ RT, int n,
                                                         int kb, int nb, double* __restrict__ Dinv_next, int kb_next, int nb_next, int* __restrict__ flag) {
  const int nt = gridDim.x, t_next = (kb_next < n) ? kb_next / 64 : 0;
  const int by = (blockIdx.y + t_next) % nt, bx = (blockIdx.x + t_next) % nt;
  if (by > bx) return;                                    // lower block triangle: not maintained
  constexpr int LD = 80;
  __shared__ double Cs[GJ_KS][LD], Rs[GJ_KS][LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ti = by * 64, tj = bx * 64;
  const int wi = (wave >> 1) * 32, wj = (wave & 1) * 32;
  const int kk = lane >> 4, li = lane & 15;
  gj_d4 acc[2][2];
  bool live[2][4][2];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int i = ti + wi + a * 16 + kk + 4 * r;
#pragma unroll
      for (int b = 0; b < 2; b++) {
        const int j = tj + wj + b * 16 + li;
        live[a][r][b] = i < n && j < n && !(i >= kb && i < kb + nb) && !(j >= kb && j < kb + nb);
        acc[a][b][r] = live[a][r][b] ? D[(size_t)i * n + j] : 0.0;
      }
    }
  for (int t0 = 0; t0 < nb; t0 += GJ_KS) {
    for (int idx = tid; idx < GJ_KS * 64; idx += 256) {
      const int k = idx >> 6, c = idx & 63, t = t0 + k;
      Cs[k][c] = (ti + c < n && t < nb) ? -PT[(size_t)t * n + ti + c] : 0.0;
      Rs[k][c] = (tj + c < n && t < nb) ? RT[(size_t)t * n + tj + c] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int k0 = 0; k0 < GJ_KS; k0 += 4) {
      const double a0 = Cs[k0 + kk][wi + li], a1 = Cs[k0 + kk][wi + 16 + li];
      const double b0 = Rs[k0 + kk][wj + li], b1 = Rs[k0 + kk][wj + 16 + li];
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int i = ti + wi + a * 16 + kk + 4 * r;
#pragma unroll
      for (int b = 0; b < 2; b++) {
        const int j = tj + wj + b * 16 + li;
        if (live[a][r][b]) D[(size_t)i * n + j] = acc[a][b][r];
      }
    }
  if (!(kb_next < n && by == bx && by == t_next)) return;
  // ---- this workgroup holds the updated next pivot block in its accumulators: invert it (same elimination as k_gjs_pivot) ----
  double (*M)[GJ_NB + 1] = reinterpret_cast<double (*)[GJ_NB + 1]>(&Cs[0][0]);     // 32 x 33 doubles inside Cs (32 x 80)
  double* colk = &Rs[0][0];
  __syncthreads();
  for (int idx = tid; idx < GJ_NB * GJ_NB; idx += 256) M[idx / GJ_NB][idx % GJ_NB] = (idx / GJ_NB == idx % GJ_NB) ? 1.0 : 0.0;   // identity padding
  __syncthreads();
  const int o = kb_next - ti;                              // offset of the block inside the tile (0 or 32)
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int il = wi + a * 16 + kk + 4 * r - o;
#pragma unroll
      for (int b = 0; b < 2; b++) {
        const int jl = wj + b * 16 + li - o;
        if (il >= 0 && il < nb_next && jl >= 0 && jl < nb_next) M[il][jl] = acc[a][b][r];
      }
    }
  __syncthreads();
  gj_invert_block(M, colk, reinterpret_cast<int*>(&Rs[1][0]), nb_next, tid, flag);
  for (int idx = tid; idx < nb_next * nb_next; idx += 256) Dinv_next[(idx / nb_next) * GJ_NB + idx % nb_next] = M[idx / nb_next][idx % nb_next];
}

// the upper triangle holds -A^-1: negate and mirror (64 x 64 tiles through LDS, both directions coalesced)
__global__ __launch_bounds__(256) void k_gjs_finish(double* __restrict__ D, int n) {
  if (blockIdx.y > blockIdx.x) return;
  __shared__ double Ts[64][65];
  const int ti = blockIdx.y * 64, tj = blockIdx.x * 64;
  for (int idx = threadIdx.x; idx < 64 * 64; idx += 256) {
    const int r = idx >> 6, c = idx & 63, i = ti + r, j = tj + c;
    double v = 0.0;
    if (i < n && j < n) {
      v = (i <= j) ? -D[(size_t)i * n + j] : 0.0;
      if (i <= j) D[(size_t)i * n + j] = v;
    }
    Ts[r][c] = v;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 64 * 64; idx += 256) {
    const int r = idx >> 6, c = idx & 63;              // writes D[tj + r][ti + c] = Ts[c][r]
    const int i = tj + r, j = ti + c;
    if (i < n && j < n && j < i) D[(size_t)i * n + j] = Ts[c][r];
  }
}

// ------------------------------------------------------------------------------------------------
// Symmetric sweep with pivot blocks of 128 (default for symmetric operators; option gj_block): the same three updates as above,
//   A_kk <- -A_kk^-1,  A_ko <- A_kk^-1 A_ko,  A_oo <- A_oo - A_ok A_kk^-1 A_ko     (upper block triangle, ends in -A^-1)
// in n / 128 steps of TWO launches.  With rank-32 updates every step streamed the whole upper triangle (97 MB at n = 4913) for
// 0.8 GFLOP -- 154 steps of ~69 us; a rank-128 update does 3.1 GFLOP per pass over the same bytes, i.e. it is bound by the FP64
// matrix cores and not by HBM, and there are 39 of them.
//   k_inv_panel   one workgroup per 32 columns: gathers the pivot rows PT (from the upper triangle: a row right of the block, a
//                 column above it), R = A_kk^-1 PT on the matrix cores (A_kk^-1 comes from the look-ahead below), writes PT, RT, the
//                 row panel of D and -A_kk^-1 into the pivot block
//   k_inv_update  128 x 128 tiles of the upper block triangle, K = 128 staged through LDS in double-buffered chunks of 16 (one
//                 barrier per chunk), 4 waves x (4 x 4) v_mfma_f64_16x16x4 tiles; tiles of the pivot block column copy the column
//                 panel out of RT (transposed through LDS); the workgroup that owns the NEXT pivot block inverts it right after its
//                 update (tile order rotated so that it is scheduled first): the sequential inversion stays off the critical path
//   inversion of a 128 x 128 block: every thread keeps an 8 x 8 sub-block in registers, pivot row and column go through a
//                 double-buffered LDS line (one barrier per pivot), WITHOUT pivoting; a pivot below 1e-10 of the block's largest
//                 diagonal entry raises flag bit 2 and the host repeats the whole factorisation with the pivoted 32-wide sweep.
// ------------------------------------------------------------------------------------------------
constexpr int IB = 128;          // pivot block and tile
constexpr int IKC = 16;          // k rows staged per chunk
constexpr int ILD = 144;         // LDS row stride in doubles: 2 * ILD mod 64 = 32, the two k rows a half-wave reads hit disjoint banks
constexpr int IPN = 32;          // columns per workgroup of the panel kernel
constexpr int IPLD = 48;         // its B stride: 2 * 48 mod 64 = 32

// In-register inverse of the symmetric block src (nb x nb, upper entries valid, row stride ld, read past the L1: the caller may just
// have written it) -> dst (row-major 128 x 128, rows / columns >= nb identity) and dstT (its transpose).  256 threads, thread
// (ty, tx) owns rows ty*8.., columns tx*8...  lines: 2 x 2 x 128 doubles of LDS.
__device__ __forceinline__ void inv128_block(const double* src, size_t ld, int nb, double* __restrict__ dst, double* __restrict__ dstT, double* lines,
                                            unsigned long long* dmax_bits, int* flag) {
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  double M[8][8];
  double dloc = 0.0;
#pragma unroll
  for (int r = 0; r < 8; r++)
#pragma unroll
    for (int c = 0; c < 8; c++) {
      const int i = ty * 8 + r, j = tx * 8 + c;
      double v = (i == j) ? 1.0 : 0.0;
      if (i < nb && j < nb) v = __hip_atomic_load(src + (size_t)min(i, j) * ld + max(i, j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      M[r][c] = v;
      if (i == j) dloc = fmax(dloc, fabs(v));
    }
  if (tid == 0) *dmax_bits = 0ull;
  __syncthreads();
  if (ty == tx) atomicMax(dmax_bits, (unsigned long long)__double_as_longlong(dloc));      // non-negative doubles order like their bits
  __syncthreads();
  const double tiny = 1e-10 * __longlong_as_double((long long)*dmax_bits);
  bool bad = false;
#pragma unroll 1
  for (int k8 = 0; k8 < 16; k8++) {
    if (k8 * 8 >= nb) break;
#pragma unroll
    for (int kk = 0; kk < 8; kk++) {
      const int k = k8 * 8 + kk;
      double* rowb = lines + (kk & 1) * 256;       // k and kk have the same parity
      double* colb = rowb + 128;
      if (ty == k8) {
#pragma unroll
        for (int c = 0; c < 8; c++) rowb[tx * 8 + c] = M[kk][c];
      }
      if (tx == k8) {
#pragma unroll
        for (int r = 0; r < 8; r++) colb[ty * 8 + r] = M[r][kk];
      }
      __syncthreads();
      const double piv = rowb[k];
      bad |= !(fabs(piv) > tiny);
      const double p = 1.0 / piv;
      double rk[8], f[8];
#pragma unroll
      for (int c = 0; c < 8; c++) rk[c] = rowb[tx * 8 + c];
#pragma unroll
      for (int r = 0; r < 8; r++) f[r] = colb[ty * 8 + r] * p;
#pragma unroll
      for (int r = 0; r < 8; r++)
#pragma unroll
        for (int c = 0; c < 8; c++) M[r][c] -= f[r] * rk[c];
      if (tx == k8) {                               // column k of the other rows
#pragma unroll
        for (int r = 0; r < 8; r++) M[r][kk] = -f[r];
      }
      if (ty == k8) {                               // row k
#pragma unroll
        for (int c = 0; c < 8; c++) M[kk][c] = rk[c] * p;
        if (tx == k8) M[kk][kk] = p;
      }
    }
  }
  if (bad) atomicOr(flag, 4);
#pragma unroll
  for (int r = 0; r < 8; r++)
#pragma unroll
    for (int c = 0; c < 8; c++) {
      dst[(ty * 8 + r) * IB + tx * 8 + c] = M[r][c];
      dstT[(tx * 8 + c) * IB + ty * 8 + r] = M[r][c];
    }
}

__global__ __launch_bounds__(256) void k_inv_first(const double* __restrict__ D, int n, int nb, double* __restrict__ Dinv, int* __restrict__ flag) {
  __shared__ double lines[512];
  __shared__ unsigned long long dmax_bits;
  inv128_block(D, (size_t)n, nb, Dinv, Dinv + IB * IB, lines, &dmax_bits, flag);
}

// Dinv: [0, IB*IB) the inverse of the pivot block (row-major), [IB*IB, 2 IB*IB) its transpose
__device__ __forceinline__ void inv_panel_body(double* __restrict__ D, const double* __restrict__ Dinv, double* __restrict__ PT, double* __restrict__ RT,
                                               int n, int kb, int nb, int bxi) {
  __shared__ double As[IKC][ILD];
  __shared__ double Bs[IB][IPLD];              // the whole gathered panel of this workgroup: 128 x 32 (+ padding)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tj = bxi * IPN;
  const bool inside = tj >= kb && tj < kb + IB, left = tj < kb;
  // ---- gather PT[t][tj + jj], t < nb: a column above the block (contiguous in t), the symmetric pivot block, or a row right of it ----
  if (left) {
    for (int idx = tid; idx < IB * IPN; idx += 256) {
      const int jj = idx >> 7, t = idx & 127, j = tj + jj;
      Bs[t][jj] = (t < nb && j < n) ? D[(size_t)j * n + kb + t] : 0.0;
    }
  } else {
    for (int idx = tid; idx < IB * IPN; idx += 256) {
      const int t = idx >> 5, jj = idx & 31, j = tj + jj;
      double v = 0.0;
      if (t < nb && j < n) v = inside ? D[(size_t)(kb + min(t, j - kb)) * n + kb + max(t, j - kb)] : D[(size_t)(kb + t) * n + j];
      Bs[t][jj] = v;
    }
  }
  __syncthreads();
  for (int idx = tid; idx < IB * IPN; idx += 256) {
    const int t = idx >> 5, jj = idx & 31;
    if (tj + jj < n) PT[(size_t)t * n + tj + jj] = Bs[t][jj];
  }
  if (inside) {               // the pivot block takes -A_kk^-1 (upper part); its columns of RT are never read
    for (int idx = tid; idx < IB * IPN; idx += 256) {
      const int s2 = idx >> 5, jj = idx & 31, j = tj + jj;
      if (s2 < nb && j < kb + nb && kb + s2 <= j) D[(size_t)(kb + s2) * n + j] = -Dinv[s2 * IB + (j - kb)];
    }
    return;
  }
  // ---- R = Dinv * PT: wave w the rows 32 w .. 32 w + 31, all 32 columns; A[k][i] = Dinv^T[k][i] streamed through LDS ----
  const int kk = lane >> 4, li = lane & 15, wi = wave * 32;
  gj_d4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++) acc[a][b] = gj_d4{0.0, 0.0, 0.0, 0.0};
  const double* DinvT = Dinv + IB * IB;
  const int nchunk = (nb + IKC - 1) / IKC;
  for (int ch = 0; ch < nchunk; ch++) {
    __syncthreads();
    {
      const int kr = tid >> 4, c8 = (tid & 15) * 8;
      const double* src = DinvT + (size_t)(ch * IKC + kr) * IB + c8;
#pragma unroll
      for (int q = 0; q < 8; q += 2) *reinterpret_cast<double2*>(&As[kr][c8 + q]) = *reinterpret_cast<const double2*>(src + q);
    }
    __syncthreads();
#pragma unroll
    for (int k0 = 0; k0 < IKC; k0 += 4) {
      const double a0 = As[k0 + kk][wi + li], a1 = As[k0 + kk][wi + 16 + li];
      const double b0 = Bs[ch * IKC + k0 + kk][li], b1 = Bs[ch * IKC + k0 + kk][16 + li];
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
  }
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int s2 = wi + a * 16 + kk + 4 * r;
#pragma unroll
      for (int b = 0; b < 2; b++) {
        const int j = tj + b * 16 + li;
        if (j < n) {
          const double v = (s2 < nb) ? acc[a][b][r] : 0.0;
          RT[(size_t)s2 * n + j] = v;
          if (!left && s2 < nb) D[(size_t)(kb + s2) * n + j] = v;
        }
      }
    }
}

// one dense matrix per launch (k_inv_panel / k_inv_update) or several beside each other (k_inv_panel_b / k_inv_update_b: blockIdx.z names the
// matrix, the grid is sized for the largest; the dissected coarse solve inverts its interior blocks this way)
// (InvDesc: fh_internal.h)

__global__ __launch_bounds__(256) void k_inv_panel(double* __restrict__ D, const double* __restrict__ Dinv, double* __restrict__ PT, double* __restrict__ RT,
                                                   int n, int kb, int nb) {
  inv_panel_body(D, Dinv, PT, RT, n, kb, nb, blockIdx.x);
}

__global__ __launch_bounds__(256) void k_inv_panel_b(const InvDesc* __restrict__ desc, int kb, int odd) {
  const InvDesc q = desc[blockIdx.z];
  if (kb >= q.n || (int)blockIdx.x * IPN >= q.n) return;
  inv_panel_body(q.D, odd ? q.Dv1 : q.Dv0, q.PT, q.RT, q.n, kb, min(IB, q.n - kb), blockIdx.x);
}

__device__ __forceinline__ void inv_update_body(double* __restrict__ D, const double* __restrict__ PT, const double* __restrict__ RT, int n, int kb,
                                                int nb, double* __restrict__ Dinv_next, int* __restrict__ flag, int nt, int bxi, int byi) {
  extern __shared__ __attribute__((aligned(16))) double iu_smem[];
  const int kblk = kb / IB, kb_next = kb + IB;
  const int t_next = (kb_next < n) ? kblk + 1 : 0;
  const int by = (byi + t_next) % nt, bx = (bxi + t_next) % nt;
  if (by > bx) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ti = by * IB, tj = bx * IB;
  if (by == kblk) return;                                  // pivot block and row panel: written by k_inv_panel
  if (bx == kblk) {                                         // column panel above the pivot block: D[ti + i][kb + s] = RT[s][ti + i]
    double (*Ts)[65] = reinterpret_cast<double (*)[65]>(iu_smem);
    for (int h = 0; h < 4; h++) {                           // four 64 x 64 quarters through LDS, both directions coalesced
      const int s0 = (h >> 1) * 64, i0 = (h & 1) * 64;
      __syncthreads();
      for (int idx = tid; idx < 64 * 64; idx += 256) {
        const int s2 = s0 + (idx >> 6), i = ti + i0 + (idx & 63);
        Ts[idx >> 6][idx & 63] = (s2 < nb && i < n) ? RT[(size_t)s2 * n + i] : 0.0;
      }
      __syncthreads();
      for (int idx = tid; idx < 64 * 64; idx += 256) {
        const int i = ti + i0 + (idx >> 6), s2 = s0 + (idx & 63);
        if (s2 < nb && i < n) D[(size_t)i * n + kb + s2] = Ts[idx & 63][idx >> 6];
      }
    }
    return;
  }
  double* As = iu_smem;                       // [2][IKC][ILD]
  double* Bs = iu_smem + 2 * IKC * ILD;       // [2][IKC][ILD]
  const int kk = lane >> 4, li = lane & 15;
  const int wi = (wave >> 1) * 64, wj = (wave & 1) * 64;
  gj_d4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int i = ti + wi + a * 16 + kk + 4 * r;
#pragma unroll
      for (int b = 0; b < 4; b++) {
        const int j = tj + wj + b * 16 + li;
        acc[a][b][r] = (i < n && j < n) ? D[(size_t)i * n + j] : 0.0;
      }
    }
  // staging: thread -> k row tid >> 4, eight columns (tid & 15) * 8 of A (= -PT) and of B (= RT)
  const int skr = tid >> 4, sc8 = (tid & 15) * 8;
  double2 ra[4], rb[4];
  auto gload = [&](int ch) {
    const int t = ch * IKC + skr;
    const double* pa = PT + (size_t)t * n + ti + sc8;
    const double* pb = RT + (size_t)t * n + tj + sc8;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int ca = ti + sc8 + 2 * q, cb = tj + sc8 + 2 * q;
      double2 va = make_double2(0.0, 0.0), vb = make_double2(0.0, 0.0);
      if (t < nb) {
        if (ca + 1 < n) { va.x = pa[2 * q]; va.y = pa[2 * q + 1]; } else if (ca < n) va.x = pa[2 * q];
        if (cb + 1 < n) { vb.x = pb[2 * q]; vb.y = pb[2 * q + 1]; } else if (cb < n) vb.x = pb[2 * q];
      }
      ra[q] = make_double2(-va.x, -va.y);
      rb[q] = vb;
    }
  };
  auto lstore = [&](int buf) {
    double* da = As + (buf * IKC + skr) * ILD + sc8;
    double* db = Bs + (buf * IKC + skr) * ILD + sc8;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      *reinterpret_cast<double2*>(da + 2 * q) = ra[q];
      *reinterpret_cast<double2*>(db + 2 * q) = rb[q];
    }
  };
  const int nchunk = (nb + IKC - 1) / IKC;
  gload(0);
  lstore(0);
  __syncthreads();
  for (int ch = 0; ch < nchunk; ch++) {
    const int buf = ch & 1;
    if (ch + 1 < nchunk) gload(ch + 1);
    const double* A0 = As + buf * IKC * ILD + wi + li;
    const double* B0 = Bs + buf * IKC * ILD + wj + li;
#pragma unroll
    for (int k0 = 0; k0 < IKC; k0 += 4) {
      double av[4], bv[4];
#pragma unroll
      for (int x = 0; x < 4; x++) {
        av[x] = A0[(k0 + kk) * ILD + x * 16];
        bv[x] = B0[(k0 + kk) * ILD + x * 16];
      }
#pragma unroll
      for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
    }
    if (ch + 1 < nchunk) lstore(buf ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int i = ti + wi + a * 16 + kk + 4 * r;
#pragma unroll
      for (int b = 0; b < 4; b++) {
        const int j = tj + wj + b * 16 + li;
        if (i < n && j < n) D[(size_t)i * n + j] = acc[a][b][r];
      }
    }
  if (!(kb_next < n && by == bx && by == t_next)) return;
  // ---- this workgroup has just written the next pivot block: invert it (off the critical path of the sweep) ----
  __threadfence();
  __syncthreads();
  unsigned long long* dmax_bits = reinterpret_cast<unsigned long long*>(iu_smem + 512);
  inv128_block(D + (size_t)kb_next * n + kb_next, (size_t)n, min(IB, n - kb_next), Dinv_next, Dinv_next + IB * IB, iu_smem, dmax_bits, flag);
}

__global__ __launch_bounds__(256) void k_inv_update(double* __restrict__ D, const double* __restrict__ PT, const double* __restrict__ RT, int n, int kb,
                                                    int nb, double* __restrict__ Dinv_next, int* __restrict__ flag) {
  inv_update_body(D, PT, RT, n, kb, nb, Dinv_next, flag, gridDim.x, blockIdx.x, blockIdx.y);
}

__global__ __launch_bounds__(256) void k_inv_update_b(const InvDesc* __restrict__ desc, int kb, int odd) {
  const InvDesc q = desc[blockIdx.z];
  const int nt = (q.n + IB - 1) / IB;
  if (kb >= q.n || (int)blockIdx.x >= nt || (int)blockIdx.y >= nt) return;
  inv_update_body(q.D, q.PT, q.RT, q.n, kb, min(IB, q.n - kb), odd ? q.Dv0 : q.Dv1, q.flg + 1, nt, blockIdx.x, blockIdx.y);
}

__global__ __launch_bounds__(256) void k_inv_first_b(const InvDesc* __restrict__ desc) {
  __shared__ double lines[512];
  __shared__ unsigned long long dmax_bits;
  const InvDesc q = desc[blockIdx.x];
  inv128_block(q.D, (size_t)q.n, min(IB, q.n), q.Dv0, q.Dv0 + IB * IB, lines, &dmax_bits, q.flg + 1);
}

// the upper triangle holds -A^-1: negate and mirror -- k_gjs_finish for several matrices (blockIdx.z)
__global__ __launch_bounds__(256) void k_gjs_finish_b(const InvDesc* __restrict__ desc) {
  const InvDesc q = desc[blockIdx.z];
  const int n = q.n, nt = (n + 63) / 64;
  if (blockIdx.y > blockIdx.x || (int)blockIdx.x >= nt) return;
  double* D = q.D;
  __shared__ double Ts[64][65];
  const int ti = blockIdx.y * 64, tj = blockIdx.x * 64;
  for (int idx = threadIdx.x; idx < 64 * 64; idx += 256) {
    const int r = idx >> 6, c = idx & 63, i = ti + r, j = tj + c;
    double v = 0.0;
    if (i < n && j < n) {
      v = (i <= j) ? -D[(size_t)i * n + j] : 0.0;
      if (i <= j) D[(size_t)i * n + j] = v;
    }
    Ts[r][c] = v;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 64 * 64; idx += 256) {
    const int r = idx >> 6, c = idx & 63;
    const int i = tj + r, j = ti + c;
    if (i < n && j < n && j < i) D[(size_t)i * n + j] = Ts[c][r];
  }
}

// pivot columns of all other rows: A[i, kb+t] <- - sum_s Cp[i,s] * Dinv[s,t]
__global__ __launch_bounds__(256) void k_gjb_col_panel(double* __restrict__ D, const double* __restrict__ Dinv, const double* __restrict__ Cp,
                                                       int n, int kb, int nb) {
  __shared__ double Ds[GJ_NB][GJ_NB + 1];
  const int tid = threadIdx.x;
  for (int idx = tid; idx < nb * nb; idx += 256) Ds[idx / nb][idx % nb] = Dinv[(idx / nb) * GJ_NB + idx % nb];
  __syncthreads();
  const int idx = blockIdx.x * 256 + tid;
  if (idx >= n * nb) return;
  const int i = idx / nb, t = idx % nb;
  if (i >= kb && i < kb + nb) return;
  double acc = 0.0;
  for (int s2 = 0; s2 < nb; s2++) acc += Cp[(size_t)i * GJ_NB + s2] * Ds[s2][t];
  D[(size_t)i * n + kb + t] = -acc;
}

// V^T w for nvec basis vectors (GMRES classical Gram-Schmidt): partials[j*nb + block]
__global__ __launch_bounds__(256) void k_multidot(const double* const* __restrict__ V, const double* __restrict__ w, int nvec, int n,
                                                  double* __restrict__ part) {
  __shared__ double sm[4];
  for (int j = 0; j < nvec; j++) {
    const double* v = V[j];
    double acc = 0.0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) acc += v[i] * w[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[(size_t)j * gridDim.x + blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_multidot_final(double* __restrict__ part, int nvec, int nb) {
  __shared__ double sm[4];
  const int j = blockIdx.x;
  double acc = 0.0;
  for (int i = threadIdx.x; i < nb; i += 256) acc += part[(size_t)j * nb + i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[(size_t)nvec * nb + j] = sm[0] + sm[1] + sm[2] + sm[3];
}

// w -= sum_j h[j] V_j   (h on the device, right behind the partials)
__global__ __launch_bounds__(256) void k_multiaxpy(double* __restrict__ w, const double* const* __restrict__ V, const double* __restrict__ h,
                                                   double sign, int nvec, int n) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    double acc = w[i];
    for (int j = 0; j < nvec; j++) acc += sign * h[j] * V[j][i];
    w[i] = acc;
  }
}

__global__ __launch_bounds__(256) void k_axpby2(double* y, const double* x, double a, double b, int n) {   // x may alias y
  // BLAS semantics: with b == 0 the old y is NOT referenced (it may be uninitialised memory: 0 * NaN = NaN)
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) y[i] = (b == 0.0) ? a * x[i] : a * x[i] + b * y[i];
}

// ---- device-resident GMRES (the default outer solver): the Hessenberg column, the Givens rotations, the residual estimate and the convergence test live in
// a small state block on the device; the host reads {done, rn} back ONCE per iteration (one synchronisation instead of two, no arithmetic on the host) ----
// state layout (doubles): [0] reference norm beta0  [1] rtol  [2] atol  [3] dtol  [4] rn  [5] scale of the next basis vector (1 / h_{k+1,k}, or 1 / beta)
//                         [6] iterations done  [7] done flag  [8] maxit  [9] kused  [10] last norm  [11] restart   [16 ..] g, cs, sn, y, H (row-major, restart columns)
constexpr int GM_HDR = 16;
__device__ __forceinline__ double* gm_g(double* S) { return S + GM_HDR; }
__device__ __forceinline__ double* gm_cs(double* S, int m) { return S + GM_HDR + (m + 1); }
__device__ __forceinline__ double* gm_sn(double* S, int m) { return S + GM_HDR + (m + 1) + m; }
__device__ __forceinline__ double* gm_y(double* S, int m) { return S + GM_HDR + (m + 1) + 2 * m; }
__device__ __forceinline__ double* gm_H(double* S, int m) { return S + GM_HDR + (m + 1) + 3 * m; }
static size_t gm_state_doubles(int m) { return (size_t)GM_HDR + (m + 1) + 3 * (size_t)m + (size_t)(m + 1) * m; }

// squared norm, partial sums per block
__global__ __launch_bounds__(256) void k_sqnorm_part(const double* __restrict__ w, int n, double* __restrict__ part) {
  __shared__ double sm[4];
  double acc = 0.0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) acc += w[i] * w[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}
__global__ __launch_bounds__(256) void k_sum_part(const double* __restrict__ part, int nb, double* __restrict__ out) {
  __shared__ double sm[4];
  double acc = 0.0;
  for (int i = threadIdx.x; i < nb; i += 256) acc += part[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = sm[0] + sm[1] + sm[2] + sm[3];
}
// y = s[0] * x (s on the device)
__global__ __launch_bounds__(256) void k_scale_dev(double* __restrict__ y, const double* __restrict__ x, const double* __restrict__ s, int n) {
  const double a = s[0];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) y[i] = a * x[i];
}
// Knoll guess done: beta0 = ||M^-1 b|| (sq = its square, summed over the ranks)
__global__ void k_gm_begin(double* __restrict__ S, const double* __restrict__ sq, double rtol, double atol, double dtol, int maxit, int restart) {
  S[0] = sqrt(sq[0]);
  S[1] = rtol; S[2] = atol; S[3] = dtol;
  S[4] = 0.0; S[5] = 0.0; S[6] = 0.0; S[7] = 0.0;
  S[8] = (double)maxit; S[9] = 0.0; S[10] = 0.0; S[11] = (double)restart;
}
// start of a restart cycle: beta = ||v0|| (sq = its square); converged / diverged / out of iterations -> done, otherwise g = beta e_0 and the scale 1 / beta
__global__ void k_gm_restart(double* __restrict__ S, const double* __restrict__ sq) {
  const int m = (int)S[11];
  const double beta = sqrt(sq[0]);
  S[4] = beta;
  S[10] = beta;
  S[9] = 0.0;
  double* g = gm_g(S);
  for (int i = 0; i <= m; i++) g[i] = 0.0;
  g[0] = beta;
  const bool stop = beta <= fmax(S[1] * S[0], S[2]) || S[6] >= S[8] || beta > S[3] * S[0];
  S[7] = stop ? 1.0 : 0.0;
  S[5] = (stop || beta == 0.0) ? 0.0 : 1.0 / beta;
}
// iteration k: h[0..k] = V^T w (before the orthogonalisation), wsq = ||w||^2 after it -> column k of the Hessenberg matrix, rotations, residual estimate,
// convergence test; at the end of a restart cycle (converged or k == restart - 1) the back substitution y = H^-1 g as well.  The statements follow the
// host loop of the flexible variant below one for one (same operations in the same order).
__global__ void k_gm_step(double* __restrict__ S, const double* __restrict__ h, const double* __restrict__ wsq, int k) {
  const int m = (int)S[11];
  double* g = gm_g(S);
  double* cs = gm_cs(S, m);
  double* sn = gm_sn(S, m);
  double* y = gm_y(S, m);
  double* H = gm_H(S, m);
  const double wn = sqrt(wsq[0]);
  for (int j = 0; j <= k; j++) H[(size_t)j * m + k] = h[j];
  H[(size_t)(k + 1) * m + k] = wn;
  S[10] = wn;
  S[5] = wn != 0.0 ? 1.0 / wn : 0.0;
  for (int j = 0; j < k; j++) {
    const double a = H[(size_t)j * m + k], bb = H[(size_t)(j + 1) * m + k];
    H[(size_t)j * m + k] = cs[j] * a + sn[j] * bb;
    H[(size_t)(j + 1) * m + k] = -sn[j] * a + cs[j] * bb;
  }
  const double a = H[(size_t)k * m + k], bb = H[(size_t)(k + 1) * m + k];
  const double d = hypot(a, bb);
  bool done = false;
  if (d == 0.0) {          // column k of the Hessenberg matrix vanished entirely: nothing to rotate, nothing more to gain
    cs[k] = 1.0;
    sn[k] = 0.0;
    H[(size_t)k * m + k] = 1.0;
    g[k + 1] = 0.0;
    S[6] += 1.0;
    S[4] = 0.0;
    done = true;
  } else {
    cs[k] = a / d;
    sn[k] = bb / d;
    H[(size_t)k * m + k] = d;
    H[(size_t)(k + 1) * m + k] = 0.0;
    g[k + 1] = -sn[k] * g[k];
    g[k] = cs[k] * g[k];
    S[6] += 1.0;
    const double rn = fabs(g[k + 1]);
    S[4] = rn;
    done = rn <= fmax(S[1] * S[0], S[2]) || S[6] >= S[8] || wn == 0.0 || rn > S[3] * S[0];
  }
  const int kused = k + 1;
  S[9] = (double)kused;
  S[7] = done ? 1.0 : 0.0;
  if (done || k == m - 1) {
    for (int i = kused - 1; i >= 0; i--) {
      double s2 = g[i];
      for (int j = i + 1; j < kused; j++) s2 -= H[(size_t)i * m + j] * y[j];
      y[i] = s2 / H[(size_t)i * m + i];
    }
  }
}

static inline int sgrid(fh_ctx_t c, int n) { return std::max(1, std::min(fh_div_up(n, 256), c->num_cu * 8)); }

static inline int halo_spmv(fh_halo_t h, fh_mat_t A, double* x, int n_own, double* y, int mode, const double* b, const double* dinv, double omega,
                            bool prepacked = false) {
  return fh_dev_halo_spmv(h, A, x, n_own, y, mode, b, dinv, omega, prepacked);
}

// ------------------------------------------------------------------------------------------------
// API
// ------------------------------------------------------------------------------------------------
static void free_level_colors(MgLevel& L);
static void free_level_patch_setup(MgLevel& L);
extern "C" int fh_mg_create(fh_ctx_t ctx, int nlevels, fh_mg_t* out) {
  FH_REQUIRE(ctx && out && nlevels >= 1, "fh_mg_create: bad arguments");
  fh_mg_t mg = new fh_mg_s();
  mg->ctx = ctx;
  mg->nlevels = nlevels;
  mg->lv.resize(nlevels);
  *out = mg;
  return 0;
}

extern "C" int fh_mg_set_level(fh_mg_t mg, int level, fh_mat_t A, fh_mat_t P, fh_mat_t R, int smoother, double omega, int npre, int npost) {
  FH_REQUIRE(mg && A, "fh_mg_set_level: null argument");
  FH_REQUIRE(level >= 0 && level < mg->nlevels, "fh_mg_set_level: level %d out of range", level);
  FH_REQUIRE(A->m <= A->n, "fh_mg_set_level: operator must be square (or owned rows x local columns on a distributed level)");
  FH_REQUIRE(level == 0 || P != nullptr, "fh_mg_set_level: level %d needs an interpolation matrix", level);
  FH_REQUIRE(!P || P->m == A->m, "fh_mg_set_level: interpolation has %d rows, operator has %d", P ? P->m : 0, A->m);
  FH_REQUIRE(smoother >= FH_SMOOTH_JACOBI && smoother <= FH_SMOOTH_ASM,
             "fh_mg_set_level: unknown smoother %d (0 = Richardson+Jacobi, 1 = Richardson+multicolour SOR, 2 = block Schwarz / Vanka, "
             "3 = Richardson+SOR in natural order, 4 = Richardson+ILU(0), 5 = no preconditioner, 6 = exact solve, 7 = PCASM basic / multiplicative with ILU(0) blocks)", smoother);
  FH_REQUIRE(npre >= 0 && npost >= 0, "fh_mg_set_level: negative sweep count");
  MgLevel& L = mg->lv[level];
  if (L.A_uid != A->uid) {   // another matrix (also one that landed on the address of a destroyed one): its graph may differ
    free_level_colors(L);
    free_level_patch_setup(L);
    fh_tri_destroy(L.tri);
    L.tri = nullptr;
    L.A_uid = A->uid;
  }
  L.A = A;
  L.P = P;
  L.R = R;
  L.R_given = R != nullptr;
  L.n = A->m;
  L.ncols = A->n;
  L.smoother = smoother;
  L.omega = omega;
  L.npre = npre;
  L.npost = npost;
  mg->setup_done = false;
  return 0;
}

static int capture_cycle(fh_mg_t mg);
extern "C" int fh_mg_set_cycle_type(fh_mg_t mg, int type) {
  FH_REQUIRE(mg, "fh_mg_set_cycle_type: null argument");
  FH_REQUIRE(type >= FH_CYCLE_MULTIPLICATIVE && type <= FH_CYCLE_KASKADE, "fh_mg_set_cycle_type: unknown type %d (0 multiplicative, 1 full, 2 additive, 3 kaskade)", type);
  const bool changed = mg->cycle_type != type;
  mg->cycle_type = type;
  if (changed && mg->setup_done) FH_TRY(capture_cycle(mg));      // the captured launch sequence is the old type's
  return 0;
}

extern "C" int fh_mg_set_level_solver(fh_mg_t mg, int level, int solver, int restart) {
  FH_REQUIRE(mg && level >= 0 && level < mg->nlevels, "fh_mg_set_level_solver: bad level %d", level);
  FH_REQUIRE(solver == FH_LEVEL_RICHARDSON || solver == FH_LEVEL_GMRES, "fh_mg_set_level_solver: unknown level solver %d", solver);
  FH_REQUIRE(restart >= 1, "fh_mg_set_level_solver: restart must be positive");
  mg->lv[level].solver = solver;
  mg->lv[level].gm_restart = restart;
  mg->setup_done = false;
  return 0;
}

extern "C" int fh_mg_set_level_distributed(fh_mg_t mg, int level, fh_halo_t halo, int replicated_below) {
  FH_REQUIRE(mg && level >= 0 && level < mg->nlevels, "fh_mg_set_level_distributed: bad level");
  mg->lv[level].halo = halo;
  mg->lv[level].replicated_below = replicated_below != 0;
  mg->setup_done = false;
  return 0;
}

static void free_level_gmres(MgLevel& L) {
  if (L.gm_buf) hipFree(L.gm_buf);
  if (L.gm_dV) hipFree(L.gm_dV);
  if (L.gm_small) hipFree(L.gm_small);
  L.gm_buf = nullptr;
  L.gm_dV = nullptr;
  L.gm_small = nullptr;
  L.gm_m = 0;
}

static void free_level_buffers(MgLevel& L) {
  if (L.buf_base) hipFree(L.buf_base);         // the five work vectors of the level are one allocation (one fill per preparation)
  L.buf_base = nullptr;
  for (double** p : {&L.dinv, &L.x, &L.x2, &L.b, &L.r}) *p = nullptr;
  free_level_gmres(L);
  L.buf_n = -1;
}

// everything a launch of the captured cycle carries: a repeated preparation of the same hierarchy (MGsolve prepares before every
// solve, LinearImplicitSystem.cpp:347-383) finds the same pointers, sizes and options and replays the graph it already has
static uint64_t cycle_signature(fh_mg_t mg) {
  uint64_t h = 1469598103934665603ull;
  auto mix = [&](uint64_t v) {
    for (int k = 0; k < 8; k++) {
      h ^= (v >> (8 * k)) & 0xff;
      h *= 1099511628211ull;
    }
  };
  auto mixp = [&](const void* p) { mix((uint64_t)(uintptr_t)p); };
  auto mixm = [&](fh_mat_t M) {
    mixp(M);
    if (M) {
      mix(M->uid);
      mixp(M->d_val);
      mixp(M->d_blkinfo);
      mixp(M->d_rowblk);
      mix((uint64_t)M->nblk);
      mix((uint64_t)M->tile);
      mix((uint64_t)M->lx_tile);
    }
  };
  mix((uint64_t)mg->nlevels);
  mix((uint64_t)mg->cycle_type);
  mix((uint64_t)mg->ctx->opt_gen);
  mixp(mg->d_ainv);
  mixp(mg->d_nd);
  mix((uint64_t)(mg->nd_active ? mg->nd_off.size() : 0));
  if (mg->nd_active)
    for (int o : mg->nd_off) mix((uint64_t)o);          // another dissection of the same size keeps no captured pointer
  mix((uint64_t)mg->na);
  mix((uint64_t)mg->direct0_active);
  mixp(mg->direct0);
  mix(fh_direct_generation(mg->direct0));     // a re-analysed sparse solve frees and reallocates every buffer the captured launches refer to
  mixp(mg->d_act);
  for (int l = 0; l < mg->nlevels; l++) {
    MgLevel& L = mg->lv[l];
    mixm(L.A);
    mixm(L.P);
    mixm(L.R);
    mix((uint64_t)L.smoother);
    mix((uint64_t)L.npatch_exact);
    mix((uint64_t)L.npre);
    mix((uint64_t)L.npost);
    uint64_t ob;
    memcpy(&ob, &L.omega, 8);
    mix(ob);
    mix((uint64_t)L.n);
    mix((uint64_t)L.ncols);
    for (double* p : {L.dinv, L.x, L.x2, L.b, L.r}) mixp(p);
    mixp(L.direct);
    mix(fh_direct_generation(L.direct));
    mixp(L.tri);
    mixp(L.d_color_rows);
    mix((uint64_t)L.ncolors);
    mixp(L.d_pinv);
    mixp(L.d_porder);
    mix((uint64_t)L.vanka_ncolors);
    mixp(L.halo);
    mix((uint64_t)L.solver);
    mix((uint64_t)L.gm_restart);
    mixp(L.gm_buf);
  }
  return h ? h : 1;
}

static void free_level_colors(MgLevel& L) {
  if (L.d_color_rows) hipFree(L.d_color_rows);
  L.d_color_rows = nullptr;
  L.ncolors = 0;
}

// device side of the patch smoother (colouring by the matrix graph, inverses): rebuilt by the next setup; the patch lists stay
static void free_level_patch_setup(MgLevel& L) {
  for (void** q : {(void**)&L.d_pptr, (void**)&L.d_pdofs, (void**)&L.d_porder, (void**)&L.d_pflag, (void**)&L.d_poff, (void**)&L.d_pinv, (void**)&L.d_pcptr,
                   (void**)&L.d_pbar, (void**)&L.d_pscr, (void**)&L.d_pmask, (void**)&L.d_pdesc})
    if (*q) {
      hipFree(*q);
      *q = nullptr;
    }
  L.vanka_ncolors = 0;
}

static void free_level_patches(MgLevel& L) {
  free_level_patch_setup(L);
  L.npatch = 0;
}

extern "C" int fh_mg_set_level_patches(fh_mg_t mg, int level, int npatch, const int* ptr, const int* dofs) {
  FH_REQUIRE(mg && level >= 0 && level < mg->nlevels && npatch > 0 && ptr && dofs, "fh_mg_set_level_patches: bad arguments");
  MgLevel& L = mg->lv[level];
  free_level_patches(L);
  L.npatch = npatch;
  L.h_pptr.assign(ptr, ptr + npatch + 1);
  L.h_pdofs.assign(dofs, dofs + ptr[npatch]);
  L.h_poff.assign(npatch + 1, 0);
  L.max_patch = 0;
  for (int p = 0; p < npatch; p++) {
    const int np = ptr[p + 1] - ptr[p];
    FH_REQUIRE(np > 0 && np <= 512, "fh_mg_set_level_patches: patch %d has %d dofs (1..512 supported)", p, np);
    L.max_patch = std::max(L.max_patch, np);
    L.h_poff[p + 1] = L.h_poff[p] + (int64_t)np * np;
  }
  L.npatch_exact = 0;
  mg->setup_done = false;
  return 0;
}

extern "C" int fh_mg_set_level_patches_exact(fh_mg_t mg, int level, int nfirst) {
  FH_REQUIRE(mg && level >= 0 && level < mg->nlevels, "fh_mg_set_level_patches_exact: bad arguments");
  MgLevel& L = mg->lv[level];
  FH_REQUIRE(nfirst >= 0 && nfirst <= L.npatch, "fh_mg_set_level_patches_exact: %d of %d blocks", nfirst, L.npatch);
  L.npatch_exact = nfirst;
  mg->setup_done = false;
  return 0;
}

// greedy colouring in patch order; two patches conflict when a dof of one appears in the matrix rows of the other
// sequential = false: greedy colours (patches of a colour neither share nor read each other's dofs; colours in any order -- the Vanka smoother).
// sequential = true (FH_SMOOTH_ASM): the blocks keep their INDEX ORDER, as PCASM's multiplicative composition visits them: block p gets the
// dependency level 1 + max level of the earlier blocks it conflicts with, so a level holds blocks whose order among themselves does not matter and
// the levels, run one after the other, give exactly the sequential sweep (level scheduling, as for the natural-order SOR / ILU sweeps)
static int color_patches(MgLevel& L, bool sequential) {
  const int n = L.A->m, np = L.npatch;
  const std::vector<int>&rp = L.A->h_rowptr, &cl = fh_hcol(L.A);
  for (int d : L.h_pdofs) FH_REQUIRE(d >= 0 && d < n, "Vanka smoother: patch dof %d out of range", d);
  std::vector<int> optr(n + 1, 0), rptr(n + 1, 0);
  std::vector<std::vector<int>> reads(np);
  std::vector<int> mark(n, -1);
  for (int p = 0; p < np; p++) {
    for (int q = L.h_pptr[p]; q < L.h_pptr[p + 1]; q++) {
      const int r = L.h_pdofs[q];
      optr[r + 1]++;
      for (int k = rp[r]; k < rp[r + 1]; k++)
        if (mark[cl[k]] != p) {
          mark[cl[k]] = p;
          reads[p].push_back(cl[k]);
        }
    }
    for (int c : reads[p]) rptr[c + 1]++;
  }
  for (int i = 0; i < n; i++) {
    optr[i + 1] += optr[i];
    rptr[i + 1] += rptr[i];
  }
  std::vector<int> owner(optr[n]), reader(rptr[n]), oc(optr.begin(), optr.end() - 1), rc(rptr.begin(), rptr.end() - 1);
  for (int p = 0; p < np; p++) {
    for (int q = L.h_pptr[p]; q < L.h_pptr[p + 1]; q++) owner[oc[L.h_pdofs[q]]++] = p;
    for (int c : reads[p]) reader[rc[c]++] = p;
  }
  std::vector<int> color(np, -1), used;
  int ncol = 0;
  for (int p = 0; p < np; p++) {
    used.assign(ncol + 1, 0);
    for (int c : reads[p])
      for (int k = optr[c]; k < optr[c + 1]; k++)
        if (color[owner[k]] >= 0) used[color[owner[k]]] = 1;
    for (int q = L.h_pptr[p]; q < L.h_pptr[p + 1]; q++) {
      const int d = L.h_pdofs[q];
      for (int k = rptr[d]; k < rptr[d + 1]; k++)
        if (color[reader[k]] >= 0) used[color[reader[k]]] = 1;
    }
    int c = 0;
    if (sequential) {
      for (int k = 0; k < ncol; k++)
        if (used[k]) c = k + 1;
    } else
      while (used[c]) c++;
    color[p] = c;
    ncol = std::max(ncol, c + 1);
  }
  L.vanka_ncolors = ncol;
  L.vcolor_ptr.assign(ncol + 1, 0);
  for (int p = 0; p < np; p++) L.vcolor_ptr[color[p] + 1]++;
  for (int c = 0; c < ncol; c++) L.vcolor_ptr[c + 1] += L.vcolor_ptr[c];
  std::vector<int> order(np), pos(L.vcolor_ptr.begin(), L.vcolor_ptr.end() - 1);
  for (int p = 0; p < np; p++) order[pos[color[p]]++] = p;
  auto up = [&](void** d, const void* h, size_t bytes) -> int {
    FH_CHECK_HIP(hipMalloc(d, bytes ? bytes : 8));
    if (bytes) FH_CHECK_HIP(hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice));
    return 0;
  };
  FH_TRY(up((void**)&L.d_pptr, L.h_pptr.data(), L.h_pptr.size() * sizeof(int)));
  FH_TRY(up((void**)&L.d_pdofs, L.h_pdofs.data(), L.h_pdofs.size() * sizeof(int)));
  FH_CHECK_HIP(hipMalloc(&L.d_pdesc, std::max<size_t>(L.h_pdofs.size(), 1) * sizeof(int4)));
  hipLaunchKernelGGL(k_patch_desc, dim3(fh_div_up((int64_t)L.h_pdofs.size(), 256)), dim3(256), 0, L.A->ctx->stream, (int)L.h_pdofs.size(), L.d_pdofs, L.A->d_rowptr, L.d_pdesc);
  FH_CHECK_HIP(hipGetLastError());
  FH_TRY(up((void**)&L.d_poff, L.h_poff.data(), L.h_poff.size() * sizeof(int64_t)));
  FH_TRY(up((void**)&L.d_porder, order.data(), order.size() * sizeof(int)));
  FH_TRY(up((void**)&L.d_pcptr, L.vcolor_ptr.data(), L.vcolor_ptr.size() * sizeof(int)));
  L.pbar_len = 2 + 1024;
  FH_CHECK_HIP(hipMalloc(&L.d_pbar, L.pbar_len * sizeof(unsigned)));
  FH_CHECK_HIP(hipMemset(L.d_pbar, 0, L.pbar_len * sizeof(unsigned)));
  L.vanka_maxcolor = 0;
  for (int c = 0; c < ncol; c++) L.vanka_maxcolor = std::max(L.vanka_maxcolor, L.vcolor_ptr[c + 1] - L.vcolor_ptr[c]);
  FH_CHECK_HIP(hipMalloc(&L.d_pflag, (size_t)np * sizeof(int)));
  FH_CHECK_HIP(hipMalloc(&L.d_pinv, (size_t)L.h_poff[np] * sizeof(double)));
  if (sequential) {
    FH_CHECK_HIP(hipMalloc(&L.d_pscr, (size_t)L.h_poff[np] * sizeof(double)));
    FH_CHECK_HIP(hipMalloc(&L.d_pmask, (size_t)L.h_poff[np]));
  }
  L.order_kind = sequential ? 1 : 0;
  return 0;
}

// FH_SMOOTH_ASM: the sub-solve of a block is ONE application of ILU(0) of the block matrix in its natural (ascending dof) order with
// PCFactorSetZeroPivot(1e-16) and MAT_SHIFT_NONZERO (LinearEquationSolverPetscAsm.cpp:278-335): the block matrix A_pp in M is replaced by the
// product L~ U~ of its incomplete factors (restarted on A_pp + shift I, shift = 100 eps then doubled, when a pivot fails
// |u_kk| > 1e-16 sum_{j>k} |u_kj|), which the patch inversion then turns into the dense operator (L~ U~)^-1 the sweep applies.  One workgroup
// per block on global memory (setup); O: scratch of the same layout, mask: which entries of the block lie in the pattern of A.
__global__ __launch_bounds__(256) void k_patch_ilu0(const int* __restrict__ pptr, const int* __restrict__ pdofs, const int64_t* __restrict__ poff,
                                                    const int* __restrict__ rowptr, const int* __restrict__ col, double* __restrict__ M, double* __restrict__ O,
                                                    unsigned char* __restrict__ mask, int* __restrict__ flag, int first) {
  __shared__ double s_shift, s_piv;
  __shared__ int s_fail;
  const int p = blockIdx.x + first, tid = threadIdx.x;
  const int* d = pdofs + pptr[p];
  const int np = pptr[p + 1] - pptr[p];
  double* Mp = M + poff[p];
  double* Op = O + poff[p];
  unsigned char* mk = mask + poff[p];
  for (int t = tid; t < np * np; t += 256) {
    const int r = d[t / np], c = d[t % np];
    int lo = rowptr[r], hi = rowptr[r + 1] - 1, found = 0;
    while (lo <= hi) {
      const int mid = lo + ((hi - lo) >> 1);
      const int cc = col[mid];
      if (cc == c) { found = 1; break; }
      if (cc < c) lo = mid + 1; else hi = mid - 1;
    }
    mk[t] = (unsigned char)found;
    Op[t] = Mp[t];
  }
  if (tid == 0) s_shift = 0.0;
  __syncthreads();
  for (int attempt = 0; attempt < 64; attempt++) {
    const double shift = s_shift;
    for (int t = tid; t < np * np; t += 256) Mp[t] = Op[t] + ((t / np == t % np) ? shift : 0.0);
    if (tid == 0) s_fail = 0;
    __syncthreads();
    for (int k = 0; k < np; k++) {
      if (tid == 0) {
        const double piv = Mp[(size_t)k * np + k];
        double rs = 0.0;
        for (int j = k + 1; j < np; j++)
          if (mk[(size_t)k * np + j]) rs += fabs(Mp[(size_t)k * np + j]);
        if (!(fabs(piv) > 1e-16 * rs)) s_fail = 1;
        s_piv = piv;
      }
      __syncthreads();
      if (s_fail) break;
      const double piv = s_piv;
      for (int i = k + 1 + tid; i < np; i += 256)
        if (mk[(size_t)i * np + k]) Mp[(size_t)i * np + k] /= piv;
      __syncthreads();
      const int w = np - k - 1;
      for (int t = tid; t < w * w; t += 256) {
        const int i = k + 1 + t / w, j = k + 1 + t % w;
        if (mk[(size_t)i * np + k] && mk[(size_t)k * np + j] && mk[(size_t)i * np + j]) Mp[(size_t)i * np + j] -= Mp[(size_t)i * np + k] * Mp[(size_t)k * np + j];
      }
      __syncthreads();
    }
    if (!s_fail) break;
    __syncthreads();
    if (tid == 0) s_shift = shift == 0.0 ? 100.0 * 2.220446049250313e-16 : 2.0 * shift;
    __syncthreads();
  }
  if (s_fail) {
    if (tid == 0) flag[p] = 2;
    return;
  }
  // M <- L~ U~ (entries outside the pattern of the factors are zero, so the dense product needs no masks)
  for (int t = tid; t < np * np; t += 256) {
    const int i = t / np, j = t % np, kmax = i < j ? i : j;
    double a = 0.0;
    for (int k = 0; k <= kmax; k++) a += (k == i ? 1.0 : Mp[(size_t)i * np + k]) * Mp[(size_t)k * np + j];
    Op[t] = a;
  }
  __syncthreads();
  for (int t = tid; t < np * np; t += 256) Mp[t] = Op[t];
}

// numeric part of the smoother setup (every fh_mg_setup, i.e. every Newton iteration): extract and invert the patch matrices
static int factor_patches(fh_mg_t mg, MgLevel& L) {
  fh_ctx_t c = mg->ctx;
  FH_CHECK_HIP(hipMemsetAsync(L.d_pflag, 0, (size_t)L.npatch * sizeof(int), c->stream));
  hipLaunchKernelGGL(k_patch_extract, dim3(L.npatch), dim3(256), 0, c->stream, L.d_pptr, L.d_pdofs, L.d_poff, L.A->d_rowptr, L.A->d_col, L.A->d_val,
                     L.d_pinv);
  // (blocks below npatch_exact keep A_pp: their sub-solve is the exact one, the reference's MLU_PRECOND on the solid / porous blocks)
  if (L.smoother == FH_SMOOTH_ASM && L.npatch > L.npatch_exact)
    hipLaunchKernelGGL(k_patch_ilu0, dim3(L.npatch - L.npatch_exact), dim3(256), 0, c->stream, L.d_pptr, L.d_pdofs, L.d_poff, L.A->d_rowptr, L.A->d_col, L.d_pinv, L.d_pscr,
                       L.d_pmask, L.d_pflag, L.npatch_exact);
  // patches of at most PLDS_MAX dofs: one wave each with the matrix in LDS; the others (and everything with patch_invert_lds = 0): the
  // workgroup kernel on the matrix in global memory
  int nsmall = 0;
  for (int p = 0; p < L.npatch; p++) nsmall += (L.h_pptr[p + 1] - L.h_pptr[p] <= PLDS_MAX) ? 1 : 0;
  if (c->patch_invert_lds && nsmall > 0) {
    constexpr size_t lds_max = ((size_t)PLDS_MAX * (PLDS_MAX | 1) + PLDS_MAX) * sizeof(double) + PLDS_MAX * sizeof(int);
    static bool attr_set[64] = {};
    if (!attr_set[c->device & 63]) {
      FH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_patch_invert_lds), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
      attr_set[c->device & 63] = true;
    }
    int nmax = 1;                                   // the LDS a launch asks for follows the largest small patch of this level
    for (int p = 0; p < L.npatch; p++) {
      const int np = L.h_pptr[p + 1] - L.h_pptr[p];
      if (np <= PLDS_MAX) nmax = std::max(nmax, np);
    }
    const size_t lds = ((size_t)nmax * (nmax | 1) + nmax) * sizeof(double) + nmax * sizeof(int);
    hipLaunchKernelGGL(k_patch_invert_lds, dim3(L.npatch), dim3(64), lds, c->stream, L.d_pptr, L.d_poff, L.d_pinv, L.d_pflag, nmax);
  }
  if (!c->patch_invert_lds || nsmall < L.npatch)
    hipLaunchKernelGGL(k_patch_invert, dim3(L.npatch), dim3(256), 0, c->stream, L.d_pptr, L.d_poff, L.d_pinv, L.d_pflag,
                       c->patch_invert_lds ? PLDS_MAX : 0);
  FH_CHECK_HIP(hipGetLastError());
  std::vector<int> flag(L.npatch);
  FH_CHECK_HIP(hipMemcpyAsync(flag.data(), L.d_pflag, flag.size() * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  FH_CHECK_HIP(hipStreamSynchronize(c->stream));
  for (int p = 0; p < L.npatch; p++)
    FH_REQUIRE(flag[p] == 0, flag[p] == 2 ? "ASM smoother: ILU(0) of block %d found no usable pivots within 64 shifts" : "Vanka smoother: the matrix of patch %d is singular", p);
  return 0;
}

// multiplicative Schwarz sweeps over the colours of the patches on A x = b (x updated in place; rwork: a residual vector)
static int vanka_apply(fh_mg_t mg, MgLevel& L, double* x, const double* b, double* rwork, double omega, int nsweeps) {
  fh_ctx_t c = mg->ctx;
  if (c->vanka_persistent && nsweeps > 0 && L.vanka_ncolors > 0) {
    // four waves and 4 * max_patch * 8 <= 16 KB of LDS per workgroup, at most one workgroup per CU: the whole grid is resident
    const int grid = std::max(1, std::min(fh_div_up(L.vanka_maxcolor, 4), c->num_cu));
    FH_REQUIRE(grid <= L.pbar_len - 2, "Vanka smoother: barrier buffer too small");
    const size_t lds = (size_t)4 * L.max_patch * sizeof(double);
    if (c->vanka_persistent == 1)
      hipLaunchKernelGGL(k_vanka_persistent<1>, dim3(grid), dim3(256), lds, c->stream, L.d_porder, L.d_pcptr, L.vanka_ncolors, nsweeps, L.d_pptr, L.d_pdofs,
                         L.d_poff, L.d_pinv, L.A->d_rowptr, L.A->d_col, L.A->d_val, b, x, omega, L.d_pbar, L.max_patch);
    else
      hipLaunchKernelGGL(k_vanka_persistent<2>, dim3(grid), dim3(256), lds, c->stream, L.d_porder, L.d_pcptr, L.vanka_ncolors, nsweeps, L.d_pptr, L.d_pdofs,
                         L.d_poff, L.d_pinv, L.A->d_rowptr, L.A->d_col, L.A->d_val, b, x, omega, L.d_pbar, L.max_patch);
    FH_CHECK_HIP(hipGetLastError());
    return 0;
  }
  for (int s = 0; s < nsweeps; s++)
    for (int k = 0; k < L.vanka_ncolors; k++) {
      const int np = L.vcolor_ptr[k + 1] - L.vcolor_ptr[k];
      if (np == 0) continue;
      if (c->vanka_fused) {
        hipLaunchKernelGGL(k_vanka_color_fused, dim3(np), dim3(256), (size_t)5 * L.max_patch * sizeof(double), c->stream, L.d_porder + L.vcolor_ptr[k], np, L.d_pptr,
                           L.d_pdesc, L.d_poff, L.d_pinv, L.A->d_col, L.A->d_val, b, x, omega, L.max_patch);
        continue;
      }
      FH_TRY(fh_dev_spmv(L.A, x, rwork, 2, b, nullptr, 0.0));                       // r = b - A x
      hipLaunchKernelGGL(k_vanka_color, dim3(np), dim3(64), (size_t)L.max_patch * sizeof(double), c->stream, L.d_porder + L.vcolor_ptr[k], np,
                         L.d_pptr, L.d_pdofs, L.d_poff, L.d_pinv, rwork, x, omega);
    }
  FH_CHECK_HIP(hipGetLastError());
  return 0;
}
static int vanka_sweeps(fh_mg_t mg, MgLevel& L, int nsweeps) { return vanka_apply(mg, L, L.x, L.b, L.r, L.omega, nsweeps); }

// row i is decoupled when it has a non-zero diagonal and no other non-zero entry, and no other row has a non-zero in column i; one wave per row
__global__ __launch_bounds__(256) void k_coarse_coupling(const int* __restrict__ rowptr, const int* __restrict__ col, const double* __restrict__ val, int n,
                                                         int* __restrict__ rowhit, int* __restrict__ colhit) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= n) return;
  int hit = 0, diag = 0;
  for (int k = rowptr[i] + lane; k < rowptr[i + 1]; k += 64) {
    const int j = col[k];
    const bool nz = val[k] != 0.0;
    if (j == i) diag |= nz ? 1 : 0;
    else if (nz && j < n) {
      hit = 1;
      colhit[j] = 1;          // benign race: every writer stores 1
    }
  }
  hit = __any(hit);
  diag = __any(diag);
  if (lane == 0) rowhit[i] = hit | (diag ? 0 : 2);
}

__global__ __launch_bounds__(256) void k_csr_to_dense_sub(const int* __restrict__ rowptr, const int* __restrict__ col, const double* __restrict__ val,
                                                          double* __restrict__ D, int na, const int* __restrict__ act, const int* __restrict__ pos) {
  const int i = blockIdx.x, row = act[i];
  for (int k = rowptr[row] + threadIdx.x; k < rowptr[row + 1]; k += 256) {
    const int j = pos[col[k]];
    if (j >= 0) D[(size_t)i * na + j] = val[k];
  }
}

// coarse solve with decoupled unknowns, two launches: bc = b[act[0 .. na)] (k_gather_act), then blocks [0, ceil(na / 4)): y[act[i]] = sum_k M[i][k] bc[k],
// one wave per row; the blocks behind them: y[j] = dinv[j] b[j] for the n - na others (act[na ...])
__global__ __launch_bounds__(256) void k_gather_act(const double* __restrict__ b, const int* __restrict__ act, int na, double* __restrict__ bc) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k < na) bc[k] = b[act[k]];
}

__global__ __launch_bounds__(256) void k_dense_gemv_sub(const double* __restrict__ M, const double* __restrict__ bc, const double* __restrict__ b,
                                                        double* __restrict__ y, int na, int n, const int* __restrict__ act, const double* __restrict__ dinv) {
  const int nbr = (na + 3) >> 2;
  if ((int)blockIdx.x >= nbr) {
    const int t = ((int)blockIdx.x - nbr) * 256 + threadIdx.x + na;
    if (t < n) {
      const int j = act[t];
      y[j] = dinv[j] * b[j];
    }
    return;
  }
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= na) return;
  const double* m = M + (size_t)row * na;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;          // four loads of the row in flight per lane
  int k = lane;
  for (; k + 192 < na; k += 256) {
    a0 += m[k] * bc[k];
    a1 += m[k + 64] * bc[k + 64];
    a2 += m[k + 128] * bc[k + 128];
    a3 += m[k + 192] * bc[k + 192];
  }
  for (; k < na; k += 64) a0 += m[k] * bc[k];
  double acc = (a0 + a1) + (a2 + a3);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (lane == 0) y[act[row]] = acc;
}

__global__ __launch_bounds__(256) void k_fill_value(double* __restrict__ v, double a, int n) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) v[i] = a;
}

// ------------------------------------------------------------------------------------------------
// Nested dissection of the dense coarse problem (option coarse_nd = k interior blocks, default 4; needs fh_mg_set_coarse_coords).
// What bounds the dense inverse is its SERIAL pivot chain: n pivots of ~0.7 us whatever the matrix size.  With the coupled unknowns
// ordered [I_0 | ... | I_{k-1} | S] -- S a vertex separator, no entry between two interior blocks --
//     A = [A_II A_IS; A_SI A_SS],  A_II block diagonal,   Sc = A_SS - A_SI A_II^-1 A_IS,   W = A_II^-1 A_IS
// the k block inverses run BESIDE each other (one stream each, chains of n / k pivots), then Sc^-1 (|S| pivots), and the cycle solves
//     t = b_S - W^T b_I,   x_S = Sc^-1 t,   x_I = A_II^-1 b_I - W x_S                    (three launches, exact like the full inverse)
// over 39 instead of 91 MB (bench hierarchy: 4 blocks of 735, separator 435).  Symmetric operators only (W^T = A_SI A_II^-1); an
// unusable pivot in any block falls back to the full inverse with its own fall-backs.
// The separator comes from the coordinates (host, once per pattern): the set is halved across the principal axis of its coordinates at
// a layer boundary next to the median, and the side with fewer unknowns coupled to the other side gives them up as separator.
// ------------------------------------------------------------------------------------------------
namespace {
struct NdGraph {
  std::vector<int> ptr, adj;        // coupling graph over the coupled unknowns (positions 0 .. n), both directions
};

static void nd_split(const NdGraph& G, const double* xyz, int dim, const std::vector<int>& set, int depth, std::vector<std::vector<int> >& blocks,
                     std::vector<int>& sep, std::vector<int>& side /* scratch, size n, zero */) {
  if (depth == 0 || set.size() < 64) {
    blocks.push_back(set);
    return;
  }
  // principal axis of the set
  double mean[3] = {0, 0, 0};
  for (int u : set)
    for (int d = 0; d < dim; d++) mean[d] += xyz[(size_t)u * dim + d];
  for (int d = 0; d < dim; d++) mean[d] /= (double)set.size();
  double C[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  for (int u : set) {
    double x[3] = {0, 0, 0};
    for (int d = 0; d < dim; d++) x[d] = xyz[(size_t)u * dim + d] - mean[d];
    for (int i = 0; i < dim; i++)
      for (int j = 0; j < dim; j++) C[i][j] += x[i] * x[j];
  }
  double v[3] = {0, 0, 0};
  int dmax = 0;
  for (int d = 1; d < dim; d++)
    if (C[d][d] > C[dmax][dmax] * (1.0 + 1e-9)) dmax = d;
  v[dmax] = 1.0;
  for (int it = 0; it < 60; it++) {
    double u[3] = {0, 0, 0}, nrm = 0.0;
    for (int i = 0; i < dim; i++)
      for (int j = 0; j < dim; j++) u[i] += C[i][j] * v[j];
    for (int i = 0; i < dim; i++) nrm += u[i] * u[i];
    nrm = sqrt(nrm);
    if (!(nrm > 0.0)) break;
    for (int i = 0; i < dim; i++) v[i] = u[i] / nrm;
  }
  std::vector<std::pair<double, int> > key(set.size());
  double span = 0.0;
  for (size_t k = 0; k < set.size(); k++) {
    double t = 0.0;
    for (int d = 0; d < dim; d++) t += v[d] * (xyz[(size_t)set[k] * dim + d] - mean[d]);
    key[k] = std::make_pair(t, set[k]);
    span = std::max(span, fabs(t));
  }
  const double q = span > 0.0 ? span * 1e-9 : 1.0;
  for (auto& kv : key) kv.first = std::floor(kv.first / q + 0.5);        // layers across the axis: equal keys
  std::sort(key.begin(), key.end());
  // candidate cuts: the layer boundaries next to the median on both sides
  const size_t half = set.size() / 2;
  size_t c_lo = half, c_hi = half;
  while (c_lo > 0 && key[c_lo - 1].first == key[c_lo].first) c_lo--;
  while (c_hi < set.size() && c_hi > 0 && key[c_hi - 1].first == key[c_hi].first) c_hi++;
  size_t best_cut = 0, best_cnt = (size_t)-1;
  int best_side = 0;
  for (size_t cut : {c_lo, c_hi}) {
    if (cut == 0 || cut >= set.size()) continue;
    for (size_t k = 0; k < set.size(); k++) side[key[k].second] = k < cut ? 1 : 2;
    size_t cntA = 0, cntB = 0;
    for (size_t k = 0; k < set.size(); k++) {
      const int u = key[k].second, mine = side[u];
      bool touches = false;
      for (int e = G.ptr[u]; e < G.ptr[u + 1] && !touches; e++) touches = side[G.adj[e]] == 3 - mine;
      if (touches) (mine == 1 ? cntA : cntB)++;
    }
    for (int which = 1; which <= 2; which++) {
      const size_t cnt = which == 1 ? cntA : cntB;
      const size_t rest = (which == 1 ? cut : set.size() - cut) - cnt;         // a side must keep unknowns
      if (rest == 0) continue;
      if (cnt < best_cnt) {
        best_cnt = cnt;
        best_cut = cut;
        best_side = which;
      }
    }
    for (size_t k = 0; k < set.size(); k++) side[key[k].second] = 0;
  }
  if (best_side == 0) {            // no usable cut (one layer): the set stays one block
    blocks.push_back(set);
    return;
  }
  for (size_t k = 0; k < set.size(); k++) side[key[k].second] = k < best_cut ? 1 : 2;
  std::vector<int> A, B;
  for (size_t k = 0; k < set.size(); k++) {
    const int u = key[k].second, mine = side[u];
    bool touches = false;
    if (mine == best_side)
      for (int e = G.ptr[u]; e < G.ptr[u + 1] && !touches; e++) touches = side[G.adj[e]] == 3 - mine;
    if (touches) sep.push_back(u);
    else (mine == 1 ? A : B).push_back(u);
  }
  for (size_t k = 0; k < set.size(); k++) side[key[k].second] = 0;
  std::sort(A.begin(), A.end());
  std::sort(B.begin(), B.end());
  nd_split(G, xyz, dim, A, depth - 1, blocks, sep, side);
  nd_split(G, xyz, dim, B, depth - 1, blocks, sep, side);
}
}  // namespace

constexpr int ND_ROW = 512;      // entries of a separator row staged in LDS

__global__ __launch_bounds__(256) void k_csr_to_dense_blk(const int* __restrict__ rowptr, const int* __restrict__ col, const double* __restrict__ val,
                                                          double* __restrict__ D, int off, int nb, const int* __restrict__ act, const int* __restrict__ pos) {
  const int i = blockIdx.x, row = act[off + i];
  for (int k = rowptr[row] + threadIdx.x; k < rowptr[row + 1]; k += 256) {
    const int j = pos[col[k]] - off;
    if (j >= 0 && j < nb) D[(size_t)i * nb + j] = val[k];
  }
}

// W[p][c] = sum over the entries (j, v) of separator row c inside interior block i of v * Binv_i[pos(j)][p]   (A_IS = A_SI^T, Binv symmetric);
// grid (separator unknowns, blocks).  Written as W (interior x separator) and as its transpose.
__device__ __forceinline__ void nd_w_body(const int* __restrict__ rowptr, const int* __restrict__ col, const double* __restrict__ val,
                                          const int* __restrict__ act, const int* __restrict__ pos, const double* __restrict__ Binv, int off, int nb,
                                          int nI, int ns, double* __restrict__ W, double* __restrict__ WT, int* __restrict__ flag) {
  __shared__ int ej[ND_ROW];
  __shared__ double ev[ND_ROW];
  __shared__ int ne;
  const int c = blockIdx.x, row = act[nI + c];
  if (threadIdx.x == 0) {        // the entries of the row inside the block, in the order of the row (the sums below do not depend on lane timing)
    int m = 0;
    for (int k = rowptr[row]; k < rowptr[row + 1]; k++) {
      const int j = pos[col[k]] - off;
      if (j >= 0 && j < nb && val[k] != 0.0) {
        if (m < ND_ROW) {
          ej[m] = j;
          ev[m] = val[k];
        }
        m++;
      }
    }
    if (m > ND_ROW) atomicOr(flag, 8);          // a row with more entries than the staging holds: the caller falls back to the full inverse
    ne = min(m, ND_ROW);
  }
  __syncthreads();
  const int m = ne;
  for (int p = threadIdx.x; p < nb; p += 256) {
    double acc = 0.0;
    for (int e = 0; e < m; e++) acc += ev[e] * Binv[(size_t)ej[e] * nb + p];
    W[(size_t)(off + p) * ns + c] = acc;
    WT[(size_t)c * nI + off + p] = acc;
  }
}

__global__ __launch_bounds__(256) void k_nd_w(const int* __restrict__ rowptr, const int* __restrict__ col, const double* __restrict__ val,
                                              const int* __restrict__ act, const int* __restrict__ pos, const double* __restrict__ Binv, int off, int nb,
                                              int nI, int ns, double* __restrict__ W, double* __restrict__ WT, int* __restrict__ flag) {
  nd_w_body(rowptr, col, val, act, pos, Binv, off, nb, nI, ns, W, WT, flag);
}

// all interior blocks in one launch: grid (separator unknowns, blocks)
__global__ __launch_bounds__(256) void k_nd_w_b(const int* __restrict__ rowptr, const int* __restrict__ col, const double* __restrict__ val,
                                                const int* __restrict__ act, const int* __restrict__ pos, const InvDesc* __restrict__ desc, int nI, int ns,
                                                double* __restrict__ W, double* __restrict__ WT, int* __restrict__ flag) {
  const InvDesc q = desc[blockIdx.y];
  nd_w_body(rowptr, col, val, act, pos, q.D, q.off, q.n, nI, ns, W, WT, flag);
}

// Sc[c1][c2] = A_SS[c1][c2] - sum over the interior entries (j, v) of separator row c1 of v * W[pos(j)][c2]
__global__ __launch_bounds__(256) void k_nd_schur(const int* __restrict__ rowptr, const int* __restrict__ col, const double* __restrict__ val,
                                                  const int* __restrict__ act, const int* __restrict__ pos, const double* __restrict__ W, int nI, int ns,
                                                  double* __restrict__ S, int* __restrict__ flag) {
  __shared__ int ej[ND_ROW];
  __shared__ double ev[ND_ROW];
  __shared__ int ne;
  const int c1 = blockIdx.x, row = act[nI + c1];
  if (threadIdx.x == 0) {
    int m = 0;
    for (int k = rowptr[row]; k < rowptr[row + 1]; k++) {
      const int j = pos[col[k]];
      if (j >= 0 && j < nI && val[k] != 0.0) {
        if (m < ND_ROW) {
          ej[m] = j;
          ev[m] = val[k];
        }
        m++;
      }
    }
    if (m > ND_ROW) atomicOr(flag, 8);
    ne = min(m, ND_ROW);
  }
  __syncthreads();
  const int m = ne;
  for (int c2 = threadIdx.x; c2 < ns; c2 += 256) {
    double acc = 0.0;
    for (int e = 0; e < m; e++) acc += ev[e] * W[(size_t)ej[e] * ns + c2];
    S[(size_t)c1 * ns + c2] -= acc;
  }
}

// the three launches of the block solve (bc = b gathered at the coupled unknowns): one wave per row
__device__ __forceinline__ double nd_wave_dot(const double* __restrict__ m, const double* __restrict__ v, int n, int lane) {
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  int k = lane;
  for (; k + 192 < n; k += 256) {
    a0 += m[k] * v[k];
    a1 += m[k + 64] * v[k + 64];
    a2 += m[k + 128] * v[k + 128];
    a3 += m[k + 192] * v[k + 192];
  }
  for (; k < n; k += 64) a0 += m[k] * v[k];
  double acc = (a0 + a1) + (a2 + a3);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  return acc;
}

__global__ __launch_bounds__(256) void k_nd_t(const double* __restrict__ WT, const double* __restrict__ bc, int nI, int ns, double* __restrict__ t) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= ns) return;
  const double acc = nd_wave_dot(WT + (size_t)c * nI, bc, nI, lane);
  if (lane == 0) t[c] = bc[nI + c] - acc;
}

__global__ __launch_bounds__(256) void k_nd_xs(const double* __restrict__ Sinv, const double* __restrict__ t, int ns, int nI, const int* __restrict__ act,
                                               double* __restrict__ xs, double* __restrict__ y) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= ns) return;
  const double acc = nd_wave_dot(Sinv + (size_t)c * ns, t, ns, lane);
  if (lane == 0) {
    xs[c] = acc;
    y[act[nI + c]] = acc;
  }
}

__global__ __launch_bounds__(256) void k_nd_xi(const double* __restrict__ base, const int64_t* __restrict__ rowoff, const int* __restrict__ rowinfo,
                                               const double* __restrict__ W, const double* __restrict__ bc, const double* __restrict__ xs,
                                               const double* __restrict__ b, double* __restrict__ y, int nI, int ns, int na, int n,
                                               const int* __restrict__ act, const double* __restrict__ dinv) {
  const int nbr = (nI + 3) >> 2;
  if ((int)blockIdx.x >= nbr) {              // the unknowns coupled to nothing: their diagonal
    const int t = ((int)blockIdx.x - nbr) * 256 + threadIdx.x + na;
    if (t < n) {
      const int j = act[t];
      y[j] = dinv[j] * b[j];
    }
    return;
  }
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (p >= nI) return;
  const int off = rowinfo[2 * p], nb = rowinfo[2 * p + 1];
  const double a = nd_wave_dot(base + rowoff[p], bc + off, nb, lane);
  const double w = nd_wave_dot(W + (size_t)p * ns, xs, ns, lane);
  if (lane == 0) y[act[p]] = a - w;
}

// the unpivoted symmetric sweep with pivot blocks of 128 on ONE dense matrix (n x n, leading dimension n) on a given stream;
// work: 2 n IB + 4 IB IB doubles; flag[1] collects bit 2 when a pivot block has no usable diagonal pivot
static size_t inv128_work_doubles(int n) { return (size_t)2 * n * IB + (size_t)4 * IB * IB; }

static int invert_sym128(fh_ctx_t c, hipStream_t st, double* D, int n, double* work, int* flg) {
  double* PT = work;
  double* RT = PT + (size_t)n * IB;
  double* Dv[2] = {RT + (size_t)n * IB, RT + (size_t)n * IB + 2 * IB * IB};
  const int ntb = fh_div_up(n, IB), nt = fh_div_up(n, 64);
  constexpr size_t upd_lds = (size_t)4 * IKC * ILD * sizeof(double);
  static bool attr_set[64] = {};
  if (!attr_set[c->device & 63]) {
    FH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_inv_update), hipFuncAttributeMaxDynamicSharedMemorySize, (int)upd_lds));
    attr_set[c->device & 63] = true;
  }
  hipLaunchKernelGGL(k_inv_first, dim3(1), dim3(256), 0, st, D, n, std::min(IB, n), Dv[0], flg + 1);
  for (int kb = 0, step = 0; kb < n; kb += IB, step++) {
    const int nb = std::min(IB, n - kb);
    hipLaunchKernelGGL(k_inv_panel, dim3(fh_div_up(n, IPN)), dim3(256), 0, st, D, Dv[step & 1], PT, RT, n, kb, nb);
    hipLaunchKernelGGL(k_inv_update, dim3(ntb, ntb), dim3(256), upd_lds, st, D, PT, RT, n, kb, nb, Dv[(step + 1) & 1], flg + 1);
  }
  hipLaunchKernelGGL(k_gjs_finish, dim3(nt, nt), dim3(256), 0, st, D, n);
  FH_CHECK_HIP(hipGetLastError());
  return 0;
}

// the batched form for other translation units (fh_direct.hip): k dense symmetric matrices (descriptors on the device) inverted beside each other
// on the compute stream, nmax = the largest order; work per matrix: fh_inv_work_doubles(n), two flag ints per matrix (flag[1] != 0: no usable pivot)
size_t fh_inv_work_doubles(int n) { return inv128_work_doubles(n); }
int fh_inv_sym_batched(fh_ctx_t c, const InvDesc* desc, int k, int nmax) {
  if (k <= 0 || nmax <= 0) return 0;
  constexpr size_t upd_lds = (size_t)4 * IKC * ILD * sizeof(double);
  static bool attr_set[64] = {};
  if (!attr_set[c->device & 63]) {
    FH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_inv_update_b), hipFuncAttributeMaxDynamicSharedMemorySize, (int)upd_lds));
    attr_set[c->device & 63] = true;
  }
  const int ntb = fh_div_up(nmax, IB), nt64 = fh_div_up(nmax, 64);
  for (int z0 = 0; z0 < k; z0 += 32768) {          // gridDim.z <= 65535
    const int kz = std::min(k - z0, 32768);
    hipLaunchKernelGGL(k_inv_first_b, dim3(kz), dim3(256), 0, c->stream, desc + z0);
    for (int kb = 0, step = 0; kb < nmax; kb += IB, step++) {
      hipLaunchKernelGGL(k_inv_panel_b, dim3(fh_div_up(nmax, IPN), 1, kz), dim3(256), 0, c->stream, desc + z0, kb, step & 1);
      hipLaunchKernelGGL(k_inv_update_b, dim3(ntb, ntb, kz), dim3(256), upd_lds, c->stream, desc + z0, kb, step & 1);
    }
    hipLaunchKernelGGL(k_gjs_finish_b, dim3(nt64, nt64, kz), dim3(256), 0, c->stream, desc + z0);
  }
  FH_CHECK_HIP(hipGetLastError());
  return 0;
}

// block form of the coarse solve (see the note above nd_split).  Returns 0 with mg->nd_active set, or 0 with it cleared when a block
// could not be inverted without pivoting (the caller goes on with the full inverse); non-zero: an error of the runtime
static int nd_factor(fh_mg_t mg, int n, int nfull) {
  fh_ctx_t c = mg->ctx;
  MgLevel& L0 = mg->lv[0];
  mg->nd_active = false;
  const int k = (int)mg->nd_off.size() - 2;
  if (k < 2) return 0;
  const int nI = mg->nd_off[k], ns = n - nI;
  // layout of the buffer: block inverses | separator inverse | W | W^T | t | xs | work of the blocks | work of the separator | flags
  std::vector<size_t> boff(k + 1, 0);
  size_t tot = 0;
  for (int i = 0; i < k; i++) {
    const size_t nb = (size_t)(mg->nd_off[i + 1] - mg->nd_off[i]);
    boff[i] = tot;
    tot += nb * nb;
  }
  boff[k] = tot;
  const size_t o_sinv = tot;
  tot += (size_t)ns * ns;
  const size_t n_mat = tot;                 // everything that is zeroed before the operator is copied in
  if (n_mat >= (size_t)2147483647) return 0;      // beyond the fill kernel's 32-bit length: the caller goes on with the other paths (sparse exact solve)
  const size_t o_w = tot;
  tot += (size_t)nI * ns;
  const size_t o_wt = tot;
  tot += (size_t)nI * ns;
  const size_t n_result = tot;              // ... checked for Inf / NaN at the end
  const size_t o_t = tot;
  tot += (size_t)ns + 8;
  const size_t o_xs = tot;
  tot += (size_t)ns + 8;
  std::vector<size_t> woff(k + 1, 0);
  for (int i = 0; i < k; i++) {
    woff[i] = tot;
    tot += inv128_work_doubles(mg->nd_off[i + 1] - mg->nd_off[i]);
  }
  woff[k] = tot;
  tot += inv128_work_doubles(std::max(ns, 1));
  const size_t o_flags = tot;
  tot += (size_t)(k + 2) + 8;               // two ints per matrix
  if (mg->nd_cap < tot) {
    if (mg->d_nd) FH_CHECK_HIP(hipFree(mg->d_nd));
    mg->d_nd = nullptr;
    mg->nd_cap = 0;
    FH_CHECK_HIP(hipMalloc(&mg->d_nd, tot * sizeof(double)));
    mg->nd_cap = tot;
    mg->nd_tables_valid = false;
  }
  double* base = mg->d_nd;
  mg->nd_dinv.assign(k, nullptr);
  for (int i = 0; i < k; i++) mg->nd_dinv[i] = base + boff[i];
  mg->d_nd_sinv = base + o_sinv;
  mg->d_nd_w = base + o_w;
  mg->d_nd_wt = base + o_wt;
  mg->d_nd_t = base + o_t;
  mg->d_nd_xs = base + o_xs;
  int* flags = reinterpret_cast<int*>(base + o_flags);        // [2 i], [2 i + 1] per matrix; the last pair: W / Schur staging overflow
  if (!mg->nd_tables_valid) {
    if (mg->nd_rows_cap < nI) {
      if (mg->d_nd_rowoff) FH_CHECK_HIP(hipFree(mg->d_nd_rowoff));
      if (mg->d_nd_rowinfo) FH_CHECK_HIP(hipFree(mg->d_nd_rowinfo));
      mg->d_nd_rowoff = nullptr;
      mg->d_nd_rowinfo = nullptr;
      mg->nd_rows_cap = 0;
      FH_CHECK_HIP(hipMalloc(&mg->d_nd_rowoff, (size_t)std::max(nI, 1) * sizeof(int64_t)));
      FH_CHECK_HIP(hipMalloc(&mg->d_nd_rowinfo, (size_t)2 * std::max(nI, 1) * sizeof(int)));
      mg->nd_rows_cap = nI;
    }
    std::vector<int64_t> ro(nI);
    std::vector<int> ri((size_t)2 * nI);
    for (int i = 0; i < k; i++) {
      const int off = mg->nd_off[i], nb = mg->nd_off[i + 1] - off;
      for (int p = 0; p < nb; p++) {
        ro[off + p] = (int64_t)(boff[i] + (size_t)p * nb);
        ri[2 * (off + p)] = off;
        ri[2 * (off + p) + 1] = nb;
      }
    }
    FH_CHECK_HIP(hipMemcpy(mg->d_nd_rowoff, ro.data(), ro.size() * sizeof(int64_t), hipMemcpyHostToDevice));
    FH_CHECK_HIP(hipMemcpy(mg->d_nd_rowinfo, ri.data(), ri.size() * sizeof(int), hipMemcpyHostToDevice));
    std::vector<InvDesc> hd(k);
    for (int i = 0; i < k; i++) {
      const int nb = mg->nd_off[i + 1] - mg->nd_off[i];
      double* w = base + woff[i];
      hd[i] = InvDesc{mg->nd_dinv[i], nb, w, w + (size_t)nb * IB, w + (size_t)2 * nb * IB, w + (size_t)2 * nb * IB + 2 * IB * IB, flags + 2 * i, mg->nd_off[i]};
    }
    if (mg->d_nd_desc) FH_CHECK_HIP(hipFree(mg->d_nd_desc));
    mg->d_nd_desc = nullptr;
    FH_CHECK_HIP(hipMalloc(&mg->d_nd_desc, hd.size() * sizeof(InvDesc)));
    FH_CHECK_HIP(hipMemcpy(mg->d_nd_desc, hd.data(), hd.size() * sizeof(InvDesc), hipMemcpyHostToDevice));
    mg->nd_tables_valid = true;
  }
  while ((int)mg->nd_streams.size() < k) {
    hipStream_t st;
    FH_CHECK_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    mg->nd_streams.push_back(st);
  }
  while ((int)mg->nd_events.size() < k + 1) {
    hipEvent_t ev;
    FH_CHECK_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    mg->nd_events.push_back(ev);
  }
  const int* act = mg->d_act;
  const int* pos = mg->d_act + nfull;
  hipLaunchKernelGGL(k_fill_value, dim3(c->num_cu * 4), dim3(256), 0, c->stream, base, 0.0, (int)n_mat);
  FH_CHECK_HIP(hipMemsetAsync(flags, 0, (size_t)(2 * (k + 2)) * sizeof(int), c->stream));
  for (int i = 0; i < k; i++) {
    const int off = mg->nd_off[i], nb = mg->nd_off[i + 1] - off;
    hipLaunchKernelGGL(k_csr_to_dense_blk, dim3(nb), dim3(256), 0, c->stream, L0.A->d_rowptr, L0.A->d_col, L0.A->d_val, mg->nd_dinv[i], off, nb, act, pos);
  }
  if (ns > 0)
    hipLaunchKernelGGL(k_csr_to_dense_blk, dim3(ns), dim3(256), 0, c->stream, L0.A->d_rowptr, L0.A->d_col, L0.A->d_val, mg->d_nd_sinv, nI, ns, act, pos);
  FH_CHECK_HIP(hipGetLastError());
  // the block inverses beside each other: ONE launch per step for all of them (blockIdx.z = block; default), or one stream per block
  // (coarse_nd_streams = 1; beside each other only where the runtime gives the streams distinct hardware queues)
  if (c->coarse_nd_streams == 0) {
    int nmax = 0;
    for (int i = 0; i < k; i++) nmax = std::max(nmax, mg->nd_off[i + 1] - mg->nd_off[i]);
    const InvDesc* desc = static_cast<const InvDesc*>(mg->d_nd_desc);
    constexpr size_t upd_lds = (size_t)4 * IKC * ILD * sizeof(double);
    static bool attr_set[64] = {};
    if (!attr_set[c->device & 63]) {
      FH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_inv_update_b), hipFuncAttributeMaxDynamicSharedMemorySize, (int)upd_lds));
      attr_set[c->device & 63] = true;
    }
    const int ntb = fh_div_up(nmax, IB), nt64 = fh_div_up(nmax, 64);
    hipLaunchKernelGGL(k_inv_first_b, dim3(k), dim3(256), 0, c->stream, desc);
    for (int kb = 0, step = 0; kb < nmax; kb += IB, step++) {
      hipLaunchKernelGGL(k_inv_panel_b, dim3(fh_div_up(nmax, IPN), 1, k), dim3(256), 0, c->stream, desc, kb, step & 1);
      hipLaunchKernelGGL(k_inv_update_b, dim3(ntb, ntb, k), dim3(256), upd_lds, c->stream, desc, kb, step & 1);
    }
    hipLaunchKernelGGL(k_gjs_finish_b, dim3(nt64, nt64, k), dim3(256), 0, c->stream, desc);
    if (ns > 0)
      hipLaunchKernelGGL(k_nd_w_b, dim3(ns, k), dim3(256), 0, c->stream, L0.A->d_rowptr, L0.A->d_col, L0.A->d_val, act, pos, desc, nI, ns, mg->d_nd_w, mg->d_nd_wt,
                         flags + 2 * (k + 1));
    FH_CHECK_HIP(hipGetLastError());
  } else {
  FH_CHECK_HIP(hipEventRecord(mg->nd_events[k], c->stream));
  for (int i = 0; i < k; i++) {
    const int nb = mg->nd_off[i + 1] - mg->nd_off[i];
    hipStream_t sti = c->coarse_nd_streams == 2 ? c->stream : mg->nd_streams[i];      // 2: one block after the other on the compute stream (measurements)
    FH_CHECK_HIP(hipStreamWaitEvent(sti, mg->nd_events[k], 0));
    FH_TRY(invert_sym128(c, sti, mg->nd_dinv[i], nb, base + woff[i], flags + 2 * i));
    if (ns > 0)          // W of this block on its own stream as well: it needs nothing but the block inverse
      hipLaunchKernelGGL(k_nd_w, dim3(ns), dim3(256), 0, sti, L0.A->d_rowptr, L0.A->d_col, L0.A->d_val, act, pos, mg->nd_dinv[i], mg->nd_off[i], nb,
                         nI, ns, mg->d_nd_w, mg->d_nd_wt, flags + 2 * (k + 1));
    FH_CHECK_HIP(hipEventRecord(mg->nd_events[i], sti));
    FH_CHECK_HIP(hipStreamWaitEvent(c->stream, mg->nd_events[i], 0));
  }
  }
  if (ns > 0) {
    hipLaunchKernelGGL(k_nd_schur, dim3(ns), dim3(256), 0, c->stream, L0.A->d_rowptr, L0.A->d_col, L0.A->d_val, act, pos, mg->d_nd_w, nI, ns, mg->d_nd_sinv,
                       flags + 2 * (k + 1));
    FH_CHECK_HIP(hipGetLastError());
    FH_TRY(invert_sym128(c, c->stream, mg->d_nd_sinv, ns, base + woff[k], flags + 2 * k));
  }
  hipLaunchKernelGGL(k_check_finite, dim3(std::min(fh_div_up((int64_t)n_result, 256), c->num_cu * 8)), dim3(256), 0, c->stream, base, n_result,
                     flags + 2 * (k + 1) + 1);
  FH_CHECK_HIP(hipGetLastError());
  std::vector<int> hf((size_t)2 * (k + 2), 0);
  FH_CHECK_HIP(hipMemcpyAsync(hf.data(), flags, hf.size() * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  FH_CHECK_HIP(hipStreamSynchronize(c->stream));
  bool ok = true;
  for (int v : hf) ok = ok && v == 0;
  mg->nd_active = ok;          // anything else: the full inverse with its own fall-backs and error messages decides
  return 0;
}

static int coarse_factor(fh_mg_t mg) {
  fh_ctx_t c = mg->ctx;
  MgLevel& L0 = mg->lv[0];
  const int nfull = L0.n;
  // ---- unknowns coupled to nothing leave the dense problem (exact: the operator is block diagonal with respect to them) ----
  int n = nfull;
  int sym_known = -1;                  // 1 / 0: the operator passed / failed the symmetry test of this preparation
  mg->nd_active = false;
  if (!c->coarse_reduce) mg->nd_off.clear();
  if (c->coarse_reduce && nfull > 0) {
    if (mg->hit_n < nfull) {           // kept across preparations (an allocation and its release cost more than the test itself)
      if (mg->d_hit) FH_CHECK_HIP(hipFree(mg->d_hit));
      mg->d_hit = nullptr;
      mg->hit_n = 0;
      FH_CHECK_HIP(hipMalloc(&mg->d_hit, ((size_t)2 * nfull + 2) * sizeof(int)));
      mg->hit_n = nfull;
    }
    int* d_hit = mg->d_hit;
    FH_CHECK_HIP(hipMemsetAsync(d_hit, 0, ((size_t)2 * nfull + 2) * sizeof(int), c->stream));
    hipLaunchKernelGGL(k_coarse_coupling, dim3(fh_div_up(nfull, 4)), dim3(256), 0, c->stream, L0.A->d_rowptr, L0.A->d_col, L0.A->d_val, nfull, d_hit,
                       d_hit + nfull);
    // the symmetry test of the dissected solve in the same host round trip (flag behind the marks)
    hipLaunchKernelGGL(k_csr_symmetry, dim3(fh_div_up(nfull, 4)), dim3(256), 0, c->stream, L0.A->d_rowptr, L0.A->d_col, L0.A->d_val, nfull, 1e-12,
                       d_hit + 2 * nfull);
    FH_CHECK_HIP(hipGetLastError());
    std::vector<int> hit((size_t)2 * nfull + 2);
    FH_CHECK_HIP(hipMemcpyAsync(hit.data(), d_hit, hit.size() * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    FH_CHECK_HIP(hipStreamSynchronize(c->stream));
    sym_known = hit[(size_t)2 * nfull] == 0 ? 1 : 0;
    std::vector<int> act, rest;
    for (int i = 0; i < nfull; i++) (hit[i] == 0 && hit[nfull + i] == 0 ? rest : act).push_back(i);
    n = (int)act.size();
    act.insert(act.end(), rest.begin(), rest.end());
    const int nd_key = c->coarse_nd * 1024 + (mg->coords_version & 1023);
    if (act != mg->h_act_raw || !mg->d_act || nd_key != mg->nd_key || mg->nd_A_uid != L0.A->uid) {
      mg->nd_A_uid = L0.A->uid;
      if (mg->d_act) FH_CHECK_HIP(hipFree(mg->d_act));
      mg->d_act = nullptr;
      mg->h_act_raw = act;
      mg->nd_key = nd_key;
      mg->nd_off.clear();
      mg->nd_tables_valid = false;
      if (c->coarse_nd >= 2 && n >= c->coarse_nd_min && mg->coarse_dim >= 1 && (int)mg->coarse_xyz.size() == nfull * mg->coarse_dim) {
        // nested dissection of the coupled unknowns (host, once per pattern): coupling graph from the pattern of the operator
        std::vector<int> rp(nfull + 1), posn(nfull, -1);
        FH_CHECK_HIP(hipMemcpy(rp.data(), L0.A->d_rowptr, rp.size() * sizeof(int), hipMemcpyDeviceToHost));
        std::vector<int> cl(rp[nfull]);
        FH_CHECK_HIP(hipMemcpy(cl.data(), L0.A->d_col, cl.size() * sizeof(int), hipMemcpyDeviceToHost));
        for (int i = 0; i < n; i++) posn[act[i]] = i;
        std::vector<std::pair<int, int> > ed;
        for (int i = 0; i < n; i++)
          for (int k = rp[act[i]]; k < rp[act[i] + 1]; k++) {
            const int j = cl[k] < nfull ? posn[cl[k]] : -1;
            if (j >= 0 && j != i) {
              ed.emplace_back(i, j);
              ed.emplace_back(j, i);
            }
          }
        std::sort(ed.begin(), ed.end());
        ed.erase(std::unique(ed.begin(), ed.end()), ed.end());
        NdGraph G;
        G.ptr.assign(n + 1, 0);
        for (auto& e : ed) G.ptr[e.first + 1]++;
        for (int i = 0; i < n; i++) G.ptr[i + 1] += G.ptr[i];
        G.adj.resize(ed.size());
        for (size_t k = 0; k < ed.size(); k++) G.adj[k] = ed[k].second;
        std::vector<double> xyz((size_t)n * mg->coarse_dim);
        for (int i = 0; i < n; i++)
          for (int d = 0; d < mg->coarse_dim; d++) xyz[(size_t)i * mg->coarse_dim + d] = mg->coarse_xyz[(size_t)act[i] * mg->coarse_dim + d];
        int depth = 0;
        while ((1 << (depth + 1)) <= c->coarse_nd) depth++;
        std::vector<int> all(n), side(n, 0), sep;
        for (int i = 0; i < n; i++) all[i] = i;
        std::vector<std::vector<int> > blocks;
        nd_split(G, xyz.data(), mg->coarse_dim, all, depth, blocks, sep, side);
        if (blocks.size() >= 2) {
          std::sort(sep.begin(), sep.end());
          std::vector<int> order;
          for (auto& b : blocks) {
            mg->nd_off.push_back((int)order.size());
            order.insert(order.end(), b.begin(), b.end());
          }
          mg->nd_off.push_back((int)order.size());          // first separator unknown
          order.insert(order.end(), sep.begin(), sep.end());
          mg->nd_off.push_back((int)order.size());          // = n
          FH_REQUIRE((int)order.size() == n, "coarse_factor: the dissection lost unknowns (%d of %d)", (int)order.size(), n);
          std::vector<int> act2(act);
          for (int i = 0; i < n; i++) act2[i] = act[order[i]];
          act.swap(act2);
        }
      }
      std::vector<int> both(act);
      both.resize((size_t)2 * nfull, -1);                  // [nfull, 2 nfull): position of an unknown in the dense problem, -1 = not in it
      for (int i = 0; i < n; i++) both[nfull + act[i]] = i;
      FH_CHECK_HIP(hipMalloc(&mg->d_act, both.size() * sizeof(int)));
      FH_CHECK_HIP(hipMemcpy(mg->d_act, both.data(), both.size() * sizeof(int), hipMemcpyHostToDevice));
      mg->h_act = act;
    }
  }
  mg->na = n;
  if (n == 0) return 0;
  // more coupled unknowns than the dense inverse is meant for (or asked for): the sparse exact solve -- symmetric operators on its unpivoted fronts,
  // unsymmetric / indefinite ones on pivoted fronts (round 5); only a singular operator is refused and goes on to the dense path and its own limits
  mg->direct0_active = false;
  if (c->coarse_direct == 2 || (c->coarse_direct == 1 && n > c->coarse_direct_min)) {
    if (!mg->direct0 || mg->direct0_uid != L0.A->uid) {
      if (mg->direct0) fh_direct_destroy(mg->direct0);
      mg->direct0 = nullptr;
      const bool have_xyz = mg->coarse_dim >= 1 && (int)mg->coarse_xyz.size() == nfull * mg->coarse_dim;
      FH_TRY(fh_direct_create(c, L0.A, have_xyz ? mg->coarse_dim : 0, have_xyz ? mg->coarse_xyz.data() : nullptr, 0, &mg->direct0));
      mg->direct0_uid = L0.A->uid;
    }
    if (fh_direct_factor(mg->direct0) == 0) {
      mg->direct0_active = true;
      return 0;
    }
    FH_TRACE("coarse_factor: the sparse exact solve refused the operator (%s); dense path", fh_last_error());
  }
  if (!mg->nd_off.empty() && c->gj_symmetric && c->gj_block >= IB) {
    // block form first: needs a symmetric operator (entry-by-entry check on the sparse form, taken with the coupling marks above)
    if (sym_known == 1) {
      FH_TRY(nd_factor(mg, n, nfull));
      if (mg->nd_active) return 0;
    }
  }
  FH_REQUIRE(n <= 16384, "coarse level has %d coupled unknowns: the dense direct solve supports at most 16384 (the sparse exact solve, option coarse_direct, serves operators of any size: %s)", n,
             c->coarse_direct ? "it refused this operator as singular" : "it is switched off");
  if (mg->ainv_n != n) {      // a repeated preparation of the same hierarchy keeps its buffers (the 193 MB allocation cost 5-10 ms)
    if (mg->d_ainv) FH_CHECK_HIP(hipFree(mg->d_ainv));
    if (mg->d_gjwork) FH_CHECK_HIP(hipFree(mg->d_gjwork));
    mg->d_ainv = nullptr;
    mg->d_gjwork = nullptr;
    FH_CHECK_HIP(hipMalloc(&mg->d_ainv, (size_t)n * n * sizeof(double)));
    // panels PT, RT of the widest sweep (2 x n x 128), then the pivot inverses: 2 x (block + transpose) of 128 x 128, flags
    FH_CHECK_HIP(hipMalloc(&mg->d_gjwork, ((size_t)2 * n * IB + 4 * IB * IB + 8) * sizeof(double)));
    mg->d_gjwork2 = mg->d_gjwork + (size_t)2 * n * GJ_NB + GJ_NB * GJ_NB + 8;
    mg->ainv_n = n;
  }
  hipLaunchKernelGGL(k_fill_value, dim3(c->num_cu * 8), dim3(256), 0, c->stream, mg->d_ainv, 0.0, n * n);      // (the runtime's memset runs at 0.6 TB/s)
  auto to_dense = [&]() {
    if (n == nfull) hipLaunchKernelGGL(k_csr_to_dense, dim3(n), dim3(256), 0, c->stream, L0.A->d_rowptr, L0.A->d_col, L0.A->d_val, mg->d_ainv, n);
    else hipLaunchKernelGGL(k_csr_to_dense_sub, dim3(n), dim3(256), 0, c->stream, L0.A->d_rowptr, L0.A->d_col, L0.A->d_val, mg->d_ainv, n, mg->d_act,
                            mg->d_act + nfull);
  };
  to_dense();
  double* colk = mg->d_gjwork;   // column panel (n x NB), its transpose / the row panel, pivot inverse (NB x NB), flag
  double* Cp = colk;
  double* CpT = colk + (size_t)n * GJ_NB;
  double* Dinv = colk + (size_t)2 * n * GJ_NB;
  const int nt = fh_div_up(n, 64);
  // flags behind everything else in the work buffer: [0] unsymmetric, [1] bit 0: singular pivot block, bit 1: non-finite inverse, bit 2: the
  // unpivoted 128-wide sweep met a pivot it cannot use
  int* d_flag = reinterpret_cast<int*>(mg->d_gjwork + (size_t)2 * n * IB + 4 * IB * IB);
  FH_CHECK_HIP(hipMemsetAsync(d_flag, 0, 2 * sizeof(int), c->stream));
  // the factorisation must end in a usable inverse: a pivot block without a usable pivot, or Inf / NaN anywhere in the result, is an
  // error of fh_mg_setup, not a silent part of every later cycle
  auto finish = [&]() -> int {
    int h[2] = {0, 0};
    hipLaunchKernelGGL(k_check_finite, dim3(std::min(fh_div_up((int64_t)n * n, 256), c->num_cu * 8)), dim3(256), 0, c->stream, mg->d_ainv, (size_t)n * n,
                       d_flag + 1);
    FH_CHECK_HIP(hipGetLastError());
    FH_CHECK_HIP(hipMemcpyAsync(h, d_flag, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    FH_CHECK_HIP(hipStreamSynchronize(c->stream));
    FH_REQUIRE(!(h[1] & 1), "fh_mg_setup: the coarsest operator (%d unknowns) is singular to working precision (no pivot in a %d x %d block)", n, GJ_NB, GJ_NB);
    FH_REQUIRE(!(h[1] & 2), "fh_mg_setup: the inverse of the coarsest operator (%d unknowns) contains Inf / NaN", n);
    return 0;
  };
  if (c->gj_symmetric) {
    // symmetric operator? (entry-by-entry check on the sparse form, 1e-12 of the row's largest entry)
    int h_flag = 0;
    hipLaunchKernelGGL(k_csr_symmetry, dim3(fh_div_up(nfull, 4)), dim3(256), 0, c->stream, L0.A->d_rowptr, L0.A->d_col, L0.A->d_val, nfull, 1e-12, d_flag);
    FH_CHECK_HIP(hipMemcpyAsync(&h_flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    FH_CHECK_HIP(hipStreamSynchronize(c->stream));
    if (h_flag == 0 && c->gj_block >= IB) {
      // pivot blocks of 128, two launches per step (see k_inv_update)
      double* PT = mg->d_gjwork;
      double* RT = PT + (size_t)n * IB;
      double* Dv[2] = {RT + (size_t)n * IB, RT + (size_t)n * IB + 2 * IB * IB};
      int* flg = d_flag;
      const int ntb = fh_div_up(n, IB);
      constexpr size_t upd_lds = (size_t)4 * IKC * ILD * sizeof(double);
      static bool attr_set[64] = {};
      if (!attr_set[c->device & 63]) {
        FH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_inv_update), hipFuncAttributeMaxDynamicSharedMemorySize, (int)upd_lds));
        attr_set[c->device & 63] = true;
      }
      hipLaunchKernelGGL(k_inv_first, dim3(1), dim3(256), 0, c->stream, mg->d_ainv, n, std::min(IB, n), Dv[0], flg + 1);
      for (int kb = 0, step = 0; kb < n; kb += IB, step++) {
        const int nb = std::min(IB, n - kb);
        hipLaunchKernelGGL(k_inv_panel, dim3(fh_div_up(n, IPN)), dim3(256), 0, c->stream, mg->d_ainv, Dv[step & 1], PT, RT, n, kb, nb);
        hipLaunchKernelGGL(k_inv_update, dim3(ntb, ntb), dim3(256), upd_lds, c->stream, mg->d_ainv, PT, RT, n, kb, nb, Dv[(step + 1) & 1], flg + 1);
      }
      hipLaunchKernelGGL(k_gjs_finish, dim3(nt, nt), dim3(256), 0, c->stream, mg->d_ainv, n);
      FH_CHECK_HIP(hipGetLastError());
      int hf[2] = {0, 0};
      FH_CHECK_HIP(hipMemcpyAsync(hf, flg, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
      FH_CHECK_HIP(hipStreamSynchronize(c->stream));
      if (!(hf[1] & 4)) return finish();
      // a pivot block without a usable diagonal pivot (the operator is symmetric but not definite): start again with the pivoted sweep
      FH_CHECK_HIP(hipMemsetAsync(d_flag, 0, 2 * sizeof(int), c->stream));
      hipLaunchKernelGGL(k_fill_value, dim3(c->num_cu * 8), dim3(256), 0, c->stream, mg->d_ainv, 0.0, n * n);      // (the runtime's memset runs at 0.6 TB/s)
      to_dense();
    }
    if (h_flag == 0) {
      double *PT = Cp, *RT = CpT;
      double* Dinv2[2] = {Dinv, mg->d_gjwork2};          // pivot inverse of this step / of the next one (look-ahead)
      for (int kb = 0, step = 0; kb < n; kb += GJ_NB, step++) {
        const int nb = std::min(GJ_NB, n - kb);
        const int kb_next = kb + GJ_NB, nb_next = std::max(0, std::min(GJ_NB, n - kb_next));
        hipLaunchKernelGGL(k_gjs_gather_panel, dim3(fh_div_up((int64_t)n * GJ_NB, 256)), dim3(256), 0, c->stream, mg->d_ainv, PT, n, kb, nb);
        if (step == 0) hipLaunchKernelGGL(k_gjs_pivot, dim3(1), dim3(256), 0, c->stream, PT, Dinv2[0], n, kb, nb, d_flag + 1);
        hipLaunchKernelGGL(k_gjs_row_panel, dim3(fh_div_up(n, 64)), dim3(256), 0, c->stream, mg->d_ainv, Dinv2[step & 1], PT, RT, n, kb, nb);
        hipLaunchKernelGGL(k_gjs_update_mfma, dim3(nt, nt), dim3(256), 0, c->stream, mg->d_ainv, PT, RT, n, kb, nb, Dinv2[(step + 1) & 1], kb_next,
                           nb_next, d_flag + 1);
      }
      hipLaunchKernelGGL(k_gjs_finish, dim3(nt, nt), dim3(256), 0, c->stream, mg->d_ainv, n);
      FH_CHECK_HIP(hipGetLastError());
      return finish();
    }
  }
  for (int kb = 0; kb < n; kb += GJ_NB) {
    const int nb = std::min(GJ_NB, n - kb);
    hipLaunchKernelGGL(k_gjb_save_panel, dim3(fh_div_up((int64_t)n * nb, 256)), dim3(256), 0, c->stream, mg->d_ainv, Cp, CpT, n, kb, nb);
    hipLaunchKernelGGL(k_gjb_pivot, dim3(1), dim3(256), 0, c->stream, mg->d_ainv, Dinv, n, kb, nb, d_flag + 1);
    hipLaunchKernelGGL(k_gjb_row_panel, dim3(fh_div_up(n, 64)), dim3(64), 0, c->stream, mg->d_ainv, Dinv, Cp, n, kb, nb);
    if (c->gj_mfma) hipLaunchKernelGGL(k_gjb_update_mfma, dim3(nt, nt), dim3(256), 0, c->stream, mg->d_ainv, CpT, n, kb, nb);
    else hipLaunchKernelGGL(k_gjb_update, dim3(nt, nt), dim3(256), 0, c->stream, mg->d_ainv, Cp, n, kb, nb);
    hipLaunchKernelGGL(k_gjb_col_panel, dim3(fh_div_up((int64_t)n * nb, 256)), dim3(256), 0, c->stream, mg->d_ainv, Dinv, Cp, n, kb, nb);
  }
  FH_CHECK_HIP(hipGetLastError());
  return finish();
}

static int run_cycle(fh_mg_t mg);

// greedy colouring of the matrix graph (host, integer setup work): coupled rows get different colours.  "Coupled" is symmetric: row i
// reading x_j keeps j out of i's colour whether or not row j reads x_i (an unsymmetric pattern -- a convection term, a one-sided
// constraint -- would otherwise let i and j share a colour, and row i would race with row j's update)
static int color_rows(MgLevel& L) {
  fh_mat_t A = L.A;
  const int m = A->m;
  std::vector<int> tptr(m + 1, 0);
  for (int i = 0; i < m; i++)
    for (int k = A->h_rowptr[i]; k < A->h_rowptr[i + 1]; k++) {
      const int j = fh_hcol(A)[k];
      if (j < m && j != i) tptr[j + 1]++;
    }
  for (int i = 0; i < m; i++) tptr[i + 1] += tptr[i];
  std::vector<int> trow(tptr[m]), tpos(tptr.begin(), tptr.end() - 1);
  for (int i = 0; i < m; i++)
    for (int k = A->h_rowptr[i]; k < A->h_rowptr[i + 1]; k++) {
      const int j = fh_hcol(A)[k];
      if (j < m && j != i) trow[tpos[j]++] = i;      // row i reads column j
    }
  std::vector<int> color(m, -1), mark;
  int nc = 0;
  for (int i = 0; i < m; i++) {
    mark.assign(nc + 1, 0);
    for (int k = A->h_rowptr[i]; k < A->h_rowptr[i + 1]; k++) {
      const int j = fh_hcol(A)[k];
      if (j < m && j != i && color[j] >= 0) mark[color[j]] = 1;
    }
    for (int k = tptr[i]; k < tptr[i + 1]; k++)
      if (color[trow[k]] >= 0) mark[color[trow[k]]] = 1;
    int c = 0;
    while (c < nc && mark[c]) c++;
    color[i] = c;
    nc = std::max(nc, c + 1);
  }
  L.color_ptr.assign(nc + 1, 0);
  for (int i = 0; i < m; i++) L.color_ptr[color[i] + 1]++;
  for (int c = 0; c < nc; c++) L.color_ptr[c + 1] += L.color_ptr[c];
  std::vector<int> rows(m), pos(L.color_ptr.begin(), L.color_ptr.end() - 1);
  for (int i = 0; i < m; i++) rows[pos[color[i]]++] = i;
  FH_CHECK_HIP(hipMalloc(&L.d_color_rows, std::max(m, 1) * sizeof(int)));
  FH_CHECK_HIP(hipMemcpy(L.d_color_rows, rows.data(), (size_t)m * sizeof(int), hipMemcpyHostToDevice));
  L.ncolors = nc;
  return 0;
}

// (re)capture of the cycle: un-captured warm-up run, then one cycle recorded on the internal buffers and kept for replay
static int capture_cycle(fh_mg_t mg) {
  fh_ctx_t c = mg->ctx;
  if (mg->gexec) {
    hipGraphExecDestroy(mg->gexec);
    mg->gexec = nullptr;
  }
  if (mg->graph) {
    hipGraphDestroy(mg->graph);
    mg->graph = nullptr;
  }
  mg->graph_sig = 0;
  FH_TRY(run_cycle(mg));   // un-captured warm-up: builds lazily created row blocks, validates the launches
  FH_CHECK_HIP(hipStreamSynchronize(c->stream));
  FH_TRACE("capture_cycle: warm-up cycle done");
  // distributed cycles are NOT captured: stream capture of the grouped ncclSend/ncclRecv (forked communication stream) was tried on
  // this stack (RCCL 2.26.6 of the PyTorch wheel, one-rank self exchange) and segfaults inside the library at capture time; the
  // launches of a distributed cycle are issued one by one
  if (c->use_graph && mg->capturable) {
    FH_CHECK_HIP(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    int rc = run_cycle(mg);
    hipError_t e = hipStreamEndCapture(c->stream, &mg->graph);
    if (rc) return rc;
    FH_CHECK_HIP(e);
    FH_CHECK_HIP(hipGraphInstantiate(&mg->gexec, mg->graph, nullptr, nullptr, 0));
    mg->graph_sig = cycle_signature(mg);      // after the warm-up: lazily built row blocks exist now
  }
  return 0;
}

extern "C" int fh_mg_setup(fh_mg_t mg) {
  FH_GUARD_BEGIN
  fh_ctx_t c = mg->ctx;
  for (int l = 0; l < mg->nlevels; l++) FH_REQUIRE(mg->lv[l].A, "fh_mg_setup: level %d has not been set", l);
  bool distributed = false;
  for (int l = 0; l < mg->nlevels; l++) {
    MgLevel& L = mg->lv[l];
    distributed |= (L.halo != nullptr);
    FH_REQUIRE(L.halo || L.A->m == L.A->n, "fh_mg_setup: level %d is not square and has no halo", l);
    FH_REQUIRE(!L.halo || L.R || l == 0, "fh_mg_setup: distributed level %d needs an explicit restriction matrix", l);
  }
  for (int l = 1; l < mg->nlevels; l++)
    FH_REQUIRE(mg->lv[l].P->n == mg->lv[l - 1].ncols, "fh_mg_setup: interpolation of level %d has %d columns, level %d has %d local entries", l,
               mg->lv[l].P->n, l - 1, mg->lv[l - 1].ncols);
  mg->cycle_bytes = 0;
  for (int l = 0; l < mg->nlevels; l++) {
    MgLevel& L = mg->lv[l];
    const size_t nb = ((size_t)L.ncols + 2) * sizeof(double);
    const size_t nbd = ((size_t)L.ncols + 2 + 1) & ~(size_t)1;      // doubles per vector, even: every vector stays 16-byte aligned
    if (L.buf_n != L.ncols) {      // a repeated preparation keeps its work vectors (and with them the captured cycle)
      free_level_buffers(L);
      FH_CHECK_HIP(hipMalloc(&L.buf_base, 5 * nbd * sizeof(double)));
      int slot = 0;
      for (double** p : {&L.dinv, &L.x, &L.x2, &L.b, &L.r}) *p = L.buf_base + (size_t)(slot++) * nbd;
      L.buf_n = L.ncols;
    }
    if (c->debug_poison) {
      for (double** p : {&L.dinv, &L.x, &L.x2, &L.b, &L.r}) {
        if (p != &L.dinv) FH_CHECK_HIP(hipMemsetAsync(*p, 0xFF, nb, c->stream));
        else hipLaunchKernelGGL(k_fill_value, dim3(sgrid(c, L.ncols + 2)), dim3(256), 0, c->stream, *p, 0.0, L.ncols + 2);
      }
    } else {      // zeroed by ONE fill kernel: the runtime's memset reaches 0.6 TB/s (five vectors of the finest level: 0.14 ms), five launches cost 20 us per level
      FH_REQUIRE(5 * nbd < ((size_t)1 << 31), "fh_mg_setup: level %d is too large for the 32-bit fill", l);
      hipLaunchKernelGGL(k_fill_value, dim3(sgrid(c, (int)std::min<size_t>(5 * nbd, (size_t)1 << 30))), dim3(256), 0, c->stream, L.buf_base, 0.0, (int)(5 * nbd));
    }
    FH_TRY(fh_dev_get_diag(L.A, L.dinv, 1));
    if (l > 0 && L.smoother == FH_SMOOTH_IDENTITY)      // PCNONE: B = I, the Jacobi kernels with a unit "inverse diagonal"
      hipLaunchKernelGGL(k_fill_value, dim3(sgrid(c, L.n)), dim3(256), 0, c->stream, L.dinv, 1.0, L.n);
    if (l > 0 && L.solver == FH_LEVEL_GMRES) {
      const int m = std::max(1, std::min(std::max(L.npre, L.npost), L.gm_restart));
      if (L.gm_m != m || !L.gm_buf) {
        free_level_gmres(L);
        const size_t vs = (size_t)L.ncols + 2;
        L.gm_nb = sgrid(c, L.n);
        FH_CHECK_HIP(hipMalloc(&L.gm_buf, (size_t)(m + 1) * vs * sizeof(double)));
        FH_CHECK_HIP(hipMalloc(&L.gm_dV, (size_t)(m + 1) * sizeof(double*)));
        FH_CHECK_HIP(hipMalloc(&L.gm_small, ((size_t)(m + 2) * L.gm_nb + m + 2 + (size_t)m * (m + 1) + (m + 1) + m + 2) * sizeof(double)));
        std::vector<double*> tab(m + 1);
        for (int j = 0; j <= m; j++) tab[j] = L.gm_buf + (size_t)j * vs;
        FH_CHECK_HIP(hipMemcpy(L.gm_dV, tab.data(), tab.size() * sizeof(double*), hipMemcpyHostToDevice));
        L.gm_m = m;
      }
      FH_CHECK_HIP(hipMemsetAsync(L.gm_buf, 0, (size_t)(L.gm_m + 1) * ((size_t)L.ncols + 2) * sizeof(double), c->stream));
    } else if (L.gm_buf) {
      free_level_gmres(L);
    }
    if (L.smoother == FH_SMOOTH_GS_COLOR && l > 0 && L.ncolors == 0) FH_TRY(color_rows(L));
    if ((L.smoother == FH_SMOOTH_SOR || L.smoother == FH_SMOOTH_ILU0) && l > 0) {
      if (!L.tri) FH_TRY(fh_tri_create(L.A, &L.tri));                       // level schedules: once per pattern
      if (L.smoother == FH_SMOOTH_ILU0) FH_TRY(fh_tri_ilu_factor(L.tri, L.A));   // numeric factorisation: every setup
    }
    if (L.smoother == FH_SMOOTH_LU && l > 0) {
      FH_REQUIRE(!L.halo, "fh_mg_setup: level %d: the exact solve as level preconditioner serves undistributed levels", l);
      if (!L.direct || L.direct_uid != L.A->uid) {
        if (L.direct) fh_direct_destroy(L.direct);
        L.direct = nullptr;
        const bool have_xyz = L.xyz_dim >= 1 && (int)L.xyz.size() == L.n * L.xyz_dim;
        FH_TRY(fh_direct_create(c, L.A, have_xyz ? L.xyz_dim : 0, have_xyz ? L.xyz.data() : nullptr, 0, &L.direct));
        L.direct_uid = L.A->uid;
      }
      FH_TRY(fh_direct_factor(L.direct));
    }
    if ((L.smoother == FH_SMOOTH_VANKA || L.smoother == FH_SMOOTH_ASM) && l > 0) {
      FH_REQUIRE(L.npatch > 0 && !L.halo, "fh_mg_setup: level %d uses the block smoother but has no patches (fh_mg_set_level_patches)", l);
      const int kind = L.smoother == FH_SMOOTH_ASM ? 1 : 0;
      if (L.d_pinv && L.order_kind != kind) free_level_patch_setup(L);
      if (!L.d_pinv) FH_TRY(color_patches(L, kind == 1));
      FH_TRY(factor_patches(mg, L));
    }
    if (l > 0) {
      if (!L.R_given) {
        // restriction = transpose of the interpolation (LinearImplicitSystem.cpp:379-382): P's cached explicit transpose, whose
        // values are re-gathered here whenever P was edited since (zero_rows / zero_cols / new values clear its validity flag)
        FH_TRY(fh_mat_refresh_transpose(L.P));
        L.R = L.P->At;
      }
      FH_REQUIRE(L.R->m == mg->lv[l - 1].n && (L.R->n == L.n || L.R->n == L.ncols), "fh_mg_setup: restriction of level %d has the wrong shape", l);
      const int64_t bA = fh_spmv_algorithmic_bytes(L.A), n8 = 8ll * L.n;
      // algorithmic bytes of the cycle on this level (SURVEY 8d model, zero-guess first sweep needs no SpMV):
      if (L.npre > 0) mg->cycle_bytes += 3 * n8 + (int64_t)(L.npre - 1) * (bA + 2 * n8);
      mg->cycle_bytes += bA + n8;                                            // residual
      mg->cycle_bytes += fh_spmv_algorithmic_bytes(L.R) + fh_spmv_algorithmic_bytes(L.P) + n8;   // restrict, prolong+add
      mg->cycle_bytes += (int64_t)L.npost * (bA + 2 * n8);
    }
  }
  FH_TRACE("fh_mg_setup: levels set up");
  FH_TRY(coarse_factor(mg));
  FH_TRACE("fh_mg_setup: coarse level factored");
  mg->cycle_bytes += 8ll * mg->lv[0].n * mg->lv[0].n + 16ll * mg->lv[0].n;
  mg->setup_done = true;
  mg->capturable = !distributed;
  const uint64_t sig = cycle_signature(mg);
  if (mg->gexec && mg->graph_sig == sig && c->use_graph && mg->capturable && c->mg_reuse_graph) return 0;   // same launches: the graph stays
  FH_TRY(capture_cycle(mg));
  return 0;
  FH_GUARD_END("fh_mg_setup")
}

// z = B r: forward then backward Gauss-Seidel from a zero guess over the colours of the matrix graph
static int gs_color_apply(fh_mg_t mg, MgLevel& L, const double* r, double* z) {
  fh_ctx_t c = mg->ctx;
  FH_CHECK_HIP(hipMemsetAsync(z, 0, (size_t)L.ncols * sizeof(double), c->stream));
  for (int pass = 0; pass < 2; pass++)
    for (int k = 0; k < L.ncolors; k++) {
      const int col = pass == 0 ? k : L.ncolors - 1 - k;
      const int nr = L.color_ptr[col + 1] - L.color_ptr[col];
      if (nr == 0) continue;
      hipLaunchKernelGGL(k_gs_color, dim3(fh_div_up((int64_t)nr * 16, 256)), dim3(256), 0, c->stream, L.d_color_rows + L.color_ptr[col], nr,
                         L.A->d_rowptr, L.A->d_col, L.A->d_val, L.dinv, r, z);
    }
  FH_CHECK_HIP(hipGetLastError());
  return 0;
}

// Richardson(scale omega) + a sweep preconditioner: x <- x + omega * B (b - A x).  B = forward then backward Gauss-Seidel from a
// zero guess (PCSOR's local symmetric sweep, PetscPreconditioner.cpp:219-222) over the colours (FH_SMOOTH_GS_COLOR) or in the natural
// row order as PETSc runs it (FH_SMOOTH_SOR), or the ILU(0) solve (FH_SMOOTH_ILU0, PetscPreconditioner.cpp:91-115)
static int gs_sweeps(fh_mg_t mg, MgLevel& L, int nsweeps, bool zero_guess) {
  fh_ctx_t c = mg->ctx;
  for (int s = 0; s < nsweeps; s++) {
    const bool first = zero_guess && s == 0;
    if (first) {
      FH_CHECK_HIP(hipMemcpyAsync(L.r, L.b, (size_t)L.n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    } else {
      FH_TRY(halo_spmv(L.halo, L.A, L.x, L.n, L.r, 2, L.b, nullptr, 0.0));
    }
    double* z = L.x2;
    if (L.smoother == FH_SMOOTH_ASM) {
      // Richardson around PCASM: z = B r from zero (x2), the residual of the block sweep goes through dinv (unused by this smoother)
      FH_CHECK_HIP(hipMemsetAsync(z, 0, (size_t)L.ncols * sizeof(double), c->stream));
      FH_TRY(vanka_apply(mg, L, z, L.r, L.dinv, 1.0, 1));
      hipLaunchKernelGGL(k_axpby2, dim3(sgrid(c, L.n)), dim3(256), 0, c->stream, L.x, z, L.omega, first ? 0.0 : 1.0, L.n);
      continue;
    }
    if (L.smoother == FH_SMOOTH_SOR || L.smoother == FH_SMOOTH_ILU0 || L.smoother == FH_SMOOTH_LU) {
      // z = B r in the natural row order, as the reference's PCSOR / PCILU apply it (level-scheduled, fh_trisolve.hip); PCLU: z = A^-1 r
      if (L.smoother == FH_SMOOTH_SOR) FH_TRY(fh_tri_ssor_apply(L.tri, L.A, L.dinv, L.r, z));
      else if (L.smoother == FH_SMOOTH_LU) FH_TRY(fh_direct_solve_ptr(L.direct, L.r, z));
      else FH_TRY(fh_tri_ilu_apply(L.tri, L.A, L.r, z));
      hipLaunchKernelGGL(k_axpby2, dim3(sgrid(c, L.n)), dim3(256), 0, c->stream, L.x, z, L.omega, first ? 0.0 : 1.0, L.n);
      continue;
    }
    FH_TRY(gs_color_apply(mg, L, L.r, z));
    hipLaunchKernelGGL(k_axpby2, dim3(sgrid(c, L.n)), dim3(256), 0, c->stream, L.x, z, L.omega, first ? 0.0 : 1.0, L.n);
  }
  FH_CHECK_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------
// GMRES as the level solver (FH_LEVEL_GMRES): what `SetSolverFineGrids(GMRES)` -- the reference's default `_levelSolverType`, and
// what 003_NavierStokes sets -- makes of a level (LinearEquationSolverPetsc.cpp:238-250, 501-502): exactly npre / npost iterations
// (PCMG skips the convergence test of its smoothers), left-preconditioned by the level's sweep preconditioner B (Jacobi, SOR,
// ILU(0), colour sweep, one multiplicative pass over the patches), classical Gram-Schmidt, restart _restart.  Minimises
// ||B (b - A x)||_2 over x0 + K_m(BA, B r0).  Everything stays on the device and on the stream -- dot products into device
// scalars, the (m + 1) x m least-squares problem in one single-thread kernel -- so the cycle remains one captured graph.
// ------------------------------------------------------------------------------------------------
// v <- v / sqrt(s2[0]), the norm goes to *hout (a zero norm -- lucky breakdown -- gives the zero vector and a zero entry)
__global__ __launch_bounds__(256) void k_gm_normalize(double* __restrict__ v, const double* __restrict__ s2, double* __restrict__ hout, int n) {
  const double nrm = sqrt(fmax(s2[0], 0.0));
  const double inv = nrm > 0.0 ? 1.0 / nrm : 0.0;
  if (blockIdx.x == 0 && threadIdx.x == 0) *hout = nrm;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) v[i] *= inv;
}
__global__ void k_gm_copy(double* __restrict__ dst, const double* __restrict__ src, int k) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < k) dst[i] = src[i];
}
// least-squares solution of min || beta e1 - H y ||, H (m + 1) x m stored by columns of length ld (Givens rotations, one thread)
__global__ void k_gm_solve(double* __restrict__ H, int ld, int m, const double* __restrict__ beta, double* __restrict__ g, double* __restrict__ y) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  for (int i = 0; i <= m; i++) g[i] = 0.0;
  g[0] = *beta;
  for (int k = 0; k < m; k++) {
    double* hk = H + (size_t)k * ld;
    // (rotations 0 .. k-1 have been applied to this column as they were formed: see below)
    const double a = hk[k], b2 = hk[k + 1];
    const double d = hypot(a, b2);
    const double cs = d > 0.0 ? a / d : 1.0, sn = d > 0.0 ? b2 / d : 0.0;
    hk[k] = d;
    hk[k + 1] = 0.0;
    const double t = cs * g[k] + sn * g[k + 1];
    g[k + 1] = -sn * g[k] + cs * g[k + 1];
    g[k] = t;
    for (int j = k + 1; j < m; j++) {          // the same rotation on the later columns
      double* hj = H + (size_t)j * ld;
      const double u = cs * hj[k] + sn * hj[k + 1];
      hj[k + 1] = -sn * hj[k] + cs * hj[k + 1];
      hj[k] = u;
    }
  }
  for (int k = m - 1; k >= 0; k--) {
    double acc = g[k];
    for (int j = k + 1; j < m; j++) acc -= H[(size_t)j * ld + k] * y[j];
    const double d = H[(size_t)k * ld + k];
    y[k] = d != 0.0 ? acc / d : 0.0;
  }
}
__global__ __launch_bounds__(256) void k_scale_by(double* __restrict__ z, const double* __restrict__ r, const double* __restrict__ dinv, int n) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) z[i] = dinv[i] * r[i];
}

static int gs_color_apply(fh_mg_t mg, MgLevel& L, const double* r, double* z);

// z = B r with the level's sweep preconditioner
static int level_precond(fh_mg_t mg, MgLevel& L, const double* r, double* z) {
  fh_ctx_t c = mg->ctx;
  switch (L.smoother) {
    case FH_SMOOTH_SOR: return fh_tri_ssor_apply(L.tri, L.A, L.dinv, r, z);
    case FH_SMOOTH_ILU0: return fh_tri_ilu_apply(L.tri, L.A, r, z);
    case FH_SMOOTH_LU: return fh_direct_solve_ptr(L.direct, r, z);
    case FH_SMOOTH_GS_COLOR: return gs_color_apply(mg, L, r, z);
    case FH_SMOOTH_VANKA:
    case FH_SMOOTH_ASM:           // PCApply_ASM: the blocks in order on the residual of r with the corrections so far, from zero
      FH_CHECK_HIP(hipMemsetAsync(z, 0, (size_t)L.ncols * sizeof(double), c->stream));
      return vanka_apply(mg, L, z, r, L.x2, 1.0, 1);
    default:
      hipLaunchKernelGGL(k_scale_by, dim3(sgrid(c, L.n)), dim3(256), 0, c->stream, z, r, L.dinv, L.n);
      return 0;
  }
}

static int gmres_smooth(fh_mg_t mg, MgLevel& L, int nits, bool zero_guess) {
  fh_ctx_t c = mg->ctx;
  const int n = L.n, nb = L.gm_nb, ld = L.gm_m + 1;
  const size_t vs = (size_t)L.ncols + 2;
  double* part = L.gm_small;                               // [(gm_m + 2) * nb + gm_m + 2]
  double* Hm = part + (size_t)(L.gm_m + 2) * nb + L.gm_m + 2;   // gm_m columns of length ld
  double* g = Hm + (size_t)L.gm_m * ld;
  double* y = g + ld;
  double* beta = y + L.gm_m;
  auto V = [&](int j) { return L.gm_buf + (size_t)j * vs; };
  auto dots = [&](int nvec, const double* w) -> int {      // V[0..nvec)^T w -> part[nvec * nb ...]; summed over the ranks on a distributed level
    hipLaunchKernelGGL(k_multidot, dim3(nb), dim3(256), 0, c->stream, (const double* const*)L.gm_dV, w, nvec, n, part);
    hipLaunchKernelGGL(k_multidot_final, dim3(nvec), dim3(256), 0, c->stream, part, nvec, nb);
    if (L.halo) FH_TRY(fh_halo_allreduce_ptr(L.halo, part + (size_t)nvec * nb, nvec));
    return 0;
  };
  int done = 0;
  while (done < nits) {
    const int m = std::min(L.gm_m, nits - done);
    const bool zg = zero_guess && done == 0;
    if (zg) FH_TRY(level_precond(mg, L, L.b, V(0)));
    else {
      FH_TRY(halo_spmv(L.halo, L.A, L.x, n, L.r, 2, L.b, nullptr, 0.0));
      FH_TRY(level_precond(mg, L, L.r, V(0)));
    }
    // beta = ||V0||, V0 <- V0 / beta : the dot kernel takes its vectors from the pointer table, so V0 . V0 = table entry 0 against V0
    FH_TRY(dots(1, V(0)));
    hipLaunchKernelGGL(k_gm_normalize, dim3(sgrid(c, n)), dim3(256), 0, c->stream, V(0), part + (size_t)nb, beta, n);
    FH_CHECK_HIP(hipMemsetAsync(Hm, 0, (size_t)L.gm_m * ld * sizeof(double), c->stream));
    for (int j = 0; j < m; j++) {
      FH_TRY(halo_spmv(L.halo, L.A, V(j), n, L.r, 0, nullptr, nullptr, 0.0));
      FH_TRY(level_precond(mg, L, L.r, V(j + 1)));
      FH_TRY(dots(j + 1, V(j + 1)));                        // h = V^T w  (classical Gram-Schmidt, no refinement: PETSc's default)
      hipLaunchKernelGGL(k_gm_copy, dim3(fh_div_up(j + 1, 64)), dim3(64), 0, c->stream, Hm + (size_t)j * ld, part + (size_t)(j + 1) * nb, j + 1);
      hipLaunchKernelGGL(k_multiaxpy, dim3(nb), dim3(256), 0, c->stream, V(j + 1), (const double* const*)L.gm_dV, Hm + (size_t)j * ld, -1.0, j + 1, n);
      // ||w||: table entry j + 1 is w itself
      hipLaunchKernelGGL(k_multidot, dim3(nb), dim3(256), 0, c->stream, (const double* const*)(L.gm_dV + j + 1), V(j + 1), 1, n, part);
      hipLaunchKernelGGL(k_multidot_final, dim3(1), dim3(256), 0, c->stream, part, 1, nb);
      if (L.halo) FH_TRY(fh_halo_allreduce_ptr(L.halo, part + (size_t)nb, 1));
      hipLaunchKernelGGL(k_gm_normalize, dim3(sgrid(c, n)), dim3(256), 0, c->stream, V(j + 1), part + (size_t)nb, Hm + (size_t)j * ld + j + 1, n);
    }
    hipLaunchKernelGGL(k_gm_solve, dim3(1), dim3(1), 0, c->stream, Hm, ld, m, beta, g, y);
    if (zg) FH_CHECK_HIP(hipMemsetAsync(L.x, 0, (size_t)L.ncols * sizeof(double), c->stream));
    hipLaunchKernelGGL(k_multiaxpy, dim3(nb), dim3(256), 0, c->stream, L.x, (const double* const*)L.gm_dV, y, 1.0, m, n);
    done += m;
  }
  FH_CHECK_HIP(hipGetLastError());
  return 0;
}

// one multiplicative V-cycle on the internal buffers: input lv[top].b, output lv[top].x
// distributed levels: ghosts of the operand are refreshed before every operator application (MPIAIJ MatMult semantics)
// npre / npost iterations of a level's smoother on L.x for the right-hand side L.b; zero_guess: L.x is taken as zero (sweep 1 of the
// Richardson/Jacobi smoother is then the diagonal scaling PETSc's Richardson does from a zero guess).  *packed: the send buffer of the level's
// exchange already holds the interface entries of L.x (the first sweep writes them).
static int smooth_level(fh_mg_t mg, MgLevel& L, int nits, bool zero_guess, bool* packed) {
  fh_ctx_t c = mg->ctx;
  *packed = false;
  if (nits == 0) {
    if (zero_guess) FH_CHECK_HIP(hipMemsetAsync(L.x, 0, (size_t)L.ncols * sizeof(double), c->stream));
    return 0;
  }
  if (L.solver == FH_LEVEL_GMRES) return gmres_smooth(mg, L, nits, zero_guess);
  if (L.smoother == FH_SMOOTH_VANKA) {
    if (zero_guess) FH_CHECK_HIP(hipMemsetAsync(L.x, 0, (size_t)L.ncols * sizeof(double), c->stream));
    return vanka_sweeps(mg, L, nits);
  }
  if (L.smoother == FH_SMOOTH_GS_COLOR || L.smoother == FH_SMOOTH_SOR || L.smoother == FH_SMOOTH_ILU0 || L.smoother == FH_SMOOTH_LU ||
      L.smoother == FH_SMOOTH_ASM)
    return gs_sweeps(mg, L, nits, zero_guess);
  int s = 0;
  if (zero_guess) {
    // sweep 1 from a zero guess: x = omega D^-1 b ; the others: fused Jacobi SpMV, ping-pong x <-> x2
    const int* sidx = nullptr;
    double* sbuf = nullptr;
    int nsend = 0;
    if (L.halo) fh_halo_send_plan(L.halo, &sidx, &sbuf, &nsend);
    hipLaunchKernelGGL(k_first_sweep, dim3(sgrid(c, L.n)), dim3(256), 0, c->stream, L.x, L.b, L.dinv, L.omega, L.n, sidx, sbuf, nsend);
    *packed = L.halo != nullptr;             // the exchange of this x needs no pack launch
    s = 1;
  }
  for (; s < nits; s++) {
    FH_TRY(halo_spmv(L.halo, L.A, L.x, L.n, L.x2, 3, L.b, L.dinv, L.omega, *packed));
    *packed = false;
    std::swap(L.x, L.x2);
  }
  return 0;
}

// the exact solve of level 0: x = A_0^-1 b
static int coarse_solve(fh_mg_t mg) {
  fh_ctx_t c = mg->ctx;
  MgLevel& L0 = mg->lv[0];
  if (mg->direct0_active) return fh_direct_solve_ptr(mg->direct0, L0.b, L0.x);
  if (mg->nd_active) {
    const int k = (int)mg->nd_off.size() - 2, nI = mg->nd_off[k], ns = mg->na - nI;
    hipLaunchKernelGGL(k_gather_act, dim3(fh_div_up(std::max(mg->na, 1), 256)), dim3(256), 0, c->stream, L0.b, mg->d_act, mg->na, L0.r);
    if (ns > 0) {
      hipLaunchKernelGGL(k_nd_t, dim3(fh_div_up(ns, 4)), dim3(256), 0, c->stream, mg->d_nd_wt, L0.r, nI, ns, mg->d_nd_t);
      hipLaunchKernelGGL(k_nd_xs, dim3(fh_div_up(ns, 4)), dim3(256), 0, c->stream, mg->d_nd_sinv, mg->d_nd_t, ns, nI, mg->d_act, mg->d_nd_xs, L0.x);
    }
    hipLaunchKernelGGL(k_nd_xi, dim3(fh_div_up(nI, 4) + fh_div_up(L0.n - mg->na, 256)), dim3(256), 0, c->stream, mg->d_nd, mg->d_nd_rowoff, mg->d_nd_rowinfo,
                       mg->d_nd_w, L0.r, mg->d_nd_xs, L0.b, L0.x, nI, ns, mg->na, L0.n, mg->d_act, L0.dinv);
  } else if (mg->na == L0.n)
    hipLaunchKernelGGL(k_dense_gemv, dim3(fh_div_up(L0.n, 4)), dim3(256), 0, c->stream, mg->d_ainv, L0.b, L0.x, L0.n);
  else {
    hipLaunchKernelGGL(k_gather_act, dim3(fh_div_up(std::max(mg->na, 1), 256)), dim3(256), 0, c->stream, L0.b, mg->d_act, mg->na, L0.r);
    hipLaunchKernelGGL(k_dense_gemv_sub, dim3(fh_div_up(mg->na, 4) + fh_div_up(L0.n - mg->na, 256)), dim3(256), 0, c->stream, mg->d_ainv, L0.r, L0.b, L0.x,
                       mg->na, L0.n, mg->d_act, L0.dinv);
  }
  FH_CHECK_HIP(hipGetLastError());
  return 0;
}

// b_{l-1} = R v on level l (v = the level's residual, or its right-hand side): the restriction reads ghost entries, except into a replicated
// level (owned part, then all-reduce)
static int restrict_into(fh_mg_t mg, int l, double* v) {
  MgLevel& L = mg->lv[l];
  FH_TRY(halo_spmv(L.replicated_below ? nullptr : L.halo, L.R, v, L.n, mg->lv[l - 1].b, 0, nullptr, nullptr, 0.0));
  if (L.halo && L.replicated_below) FH_TRY(fh_halo_allreduce_ptr(L.halo, mg->lv[l - 1].b, mg->lv[l - 1].n));
  return 0;
}

// PCMGMCycle_Private from level `from` down (PC_MG_MULTIPLICATIVE with one cycle per level): x_from starts at zero (zero_guess) or holds a guess
static int mcycle(fh_mg_t mg, int from, bool zero_guess) {
  if (from == 0) return coarse_solve(mg);
  for (int l = from; l >= 1; l--) {
    MgLevel& L = mg->lv[l];
    bool packed = false;
    FH_TRY(smooth_level(mg, L, L.npre, zero_guess || l < from, &packed));
    FH_TRY(halo_spmv(L.halo, L.A, L.x, L.n, L.r, 2, L.b, nullptr, 0.0, packed));     // r = b - A x (ghosts of x refreshed)
    FH_TRY(restrict_into(mg, l, L.r));
  }
  FH_TRY(coarse_solve(mg));
  for (int l = 1; l <= from; l++) {
    MgLevel& L = mg->lv[l];
    MgLevel& Lc = mg->lv[l - 1];
    FH_TRY(halo_spmv(Lc.halo, L.P, Lc.x, Lc.n, L.x, 1, nullptr, nullptr, 0.0));      // x += P x_{l-1} (reads ghost coarse values)
    bool packed = false;
    FH_TRY(smooth_level(mg, L, L.npost, false, &packed));
  }
  return 0;
}

// one application of the multigrid preconditioner to lv[top].b -> lv[top].x, PCMG's four forms (PCMGSetType, LinearEquationSolverPetsc.cpp:199-214;
// PETSc mg.c / fmg.c: PCMGMCycle_Private, PCMGACycle_Private, PCMGFCycle_Private, PCMGKCycle_Private)
static int run_cycle(fh_mg_t mg) {
  const int top = mg->nlevels - 1;
  if (mg->cycle_type == FH_CYCLE_MULTIPLICATIVE) {
    FH_TRY(mcycle(mg, top, true));
    FH_CHECK_HIP(hipGetLastError());
    return 0;
  }
  // the other three restrict the RIGHT-HAND SIDE through all levels first
  for (int l = top; l >= 1; l--) FH_TRY(restrict_into(mg, l, mg->lv[l].b));
  if (mg->cycle_type == FH_CYCLE_ADDITIVE) {
    // every level solves for itself from zero with its down smoother, the corrections are interpolated upwards and added
    for (int l = top; l >= 1; l--) {
      bool packed = false;
      FH_TRY(smooth_level(mg, mg->lv[l], mg->lv[l].npre, true, &packed));
    }
    FH_TRY(coarse_solve(mg));
    for (int l = 1; l <= top; l++) FH_TRY(halo_spmv(mg->lv[l - 1].halo, mg->lv[l].P, mg->lv[l - 1].x, mg->lv[l - 1].n, mg->lv[l].x, 1, nullptr, nullptr, 0.0));
  } else if (mg->cycle_type == FH_CYCLE_FULL) {
    // coarsest solve, then per level: interpolate the solution as the guess and run one multiplicative cycle from there
    FH_TRY(coarse_solve(mg));
    for (int l = 1; l <= top; l++) {
      FH_TRY(halo_spmv(mg->lv[l - 1].halo, mg->lv[l].P, mg->lv[l - 1].x, mg->lv[l - 1].n, mg->lv[l].x, 0, nullptr, nullptr, 0.0));   // x_l = P x_{l-1}
      FH_TRY(mcycle(mg, l, false));
    }
  } else {   // FH_CYCLE_KASKADE: coarsest solve, then interpolate and smooth (down smoother) on the way up, no coarse-grid correction
    FH_TRY(coarse_solve(mg));
    for (int l = 1; l <= top; l++) {
      FH_TRY(halo_spmv(mg->lv[l - 1].halo, mg->lv[l].P, mg->lv[l - 1].x, mg->lv[l - 1].n, mg->lv[l].x, 0, nullptr, nullptr, 0.0));
      bool packed = false;
      FH_TRY(smooth_level(mg, mg->lv[l], mg->lv[l].npre, false, &packed));
    }
  }
  FH_CHECK_HIP(hipGetLastError());
  return 0;
}

// x_out = M^-1 b_in with raw device pointers
// b_in may BE the cycle's own right-hand-side buffer (lv[top].b: a caller inside this file wrote it there) and x_out may be null (the result stays
// in lv[top].x, read there by the caller before the next cycle): the Krylov loops save the two vector copies per application that way
static int apply_cycle(fh_mg_t mg, const double* b_in, double* x_out) {
  fh_ctx_t c = mg->ctx;
  const int top = mg->nlevels - 1;
  MgLevel& L = mg->lv[top];
  if (b_in != L.b) FH_CHECK_HIP(hipMemcpyAsync(L.b, b_in, (size_t)L.n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  if (mg->gexec) {
    // the graph bakes in the x / x2 roles of the capture run, and that run left L.x (host side) pointing at the buffer it ended in;
    // run_cycle is never called un-captured while the graph exists, so the copy below reads the buffer the replay writes
    FH_CHECK_HIP(hipGraphLaunch(mg->gexec, c->stream));
  } else {
    FH_TRY(run_cycle(mg));
  }
  if (x_out && x_out != L.x) FH_CHECK_HIP(hipMemcpyAsync(x_out, L.x, (size_t)L.n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  return 0;
}

static int not_recording(fh_ctx_t c, const char* who) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  FH_CHECK_HIP(hipStreamIsCapturing(c->stream, &st));
  FH_REQUIRE(st == hipStreamCaptureStatusNone, "%s: not inside fh_graph_begin / fh_graph_end (the cycle replays its own graph)", who);
  return 0;
}

extern "C" int fh_mg_vcycle(fh_mg_t mg, fh_vec_t b, fh_vec_t x) {
  FH_REQUIRE(mg && mg->setup_done, "fh_mg_vcycle: fh_mg_setup has not been called");
  FH_TRY(not_recording(mg->ctx, "fh_mg_vcycle"));
  const int n = mg->lv[mg->nlevels - 1].n;
  FH_REQUIRE(b->n_local >= n && x->n_local >= n, "fh_mg_vcycle: vectors too short");
  return apply_cycle(mg, b->d, x->d);
}

extern "C" int64_t fh_mg_cycle_algorithmic_bytes(fh_mg_t mg) { return mg->cycle_bytes; }

// coordinates of the unknowns of level 0 (any dimension 1..3): lets the exact coarse solve dissect its dense problem (option coarse_nd);
// without them it inverts one dense matrix
extern "C" int fh_mg_set_coarse_coords(fh_mg_t mg, int dim, int n, const double* coords) {
  FH_REQUIRE(mg && dim >= 1 && dim <= 3 && n >= 0 && (coords || n == 0), "fh_mg_set_coarse_coords: bad arguments");
  mg->coarse_xyz.assign(coords, coords + (size_t)n * dim);
  mg->coarse_dim = dim;
  mg->coords_version++;
  return 0;
}

// what the last fh_mg_setup made of the coarsest level: unknowns in the dense problem, interior blocks of the dissection (0: one dense
// inverse), separator size, largest block
// coordinates of the unknowns of a level >= 1 whose preconditioner is the exact solve (FH_SMOOTH_LU): optional, as fh_mg_set_coarse_coords for level 0
extern "C" int fh_mg_set_level_coords(fh_mg_t mg, int level, int dim, int n, const double* coords) {
  FH_REQUIRE(mg && level >= 0 && level < mg->nlevels && dim >= 1 && dim <= 3 && n >= 0 && (coords || n == 0), "fh_mg_set_level_coords: bad arguments");
  if (level == 0) return fh_mg_set_coarse_coords(mg, dim, n, coords);
  MgLevel& L = mg->lv[level];
  L.xyz.assign(coords, coords + (size_t)n * dim);
  L.xyz_dim = dim;
  if (L.direct) {            // the tree was cut with other (or no) coordinates
    fh_direct_destroy(L.direct);
    L.direct = nullptr;
  }
  return 0;
}

extern "C" int fh_mg_coarse_info(fh_mg_t mg, int* n_dense, int* nd_blocks, int* nd_separator, int* nd_largest_block) {
  FH_REQUIRE(mg && mg->setup_done, "fh_mg_coarse_info: fh_mg_setup has not been called");
  const int k = mg->nd_active ? (int)mg->nd_off.size() - 2 : 0;
  if (n_dense) *n_dense = mg->na;
  if (nd_blocks) *nd_blocks = k;
  if (nd_separator) *nd_separator = k ? mg->na - mg->nd_off[k] : 0;
  int big = 0;
  for (int i = 0; i < k; i++) big = std::max(big, mg->nd_off[i + 1] - mg->nd_off[i]);
  if (nd_largest_block) *nd_largest_block = big;
  return 0;
}

extern "C" int fh_mg_destroy(fh_mg_t mg) {
  if (!mg) return 0;
  hipStreamSynchronize(mg->ctx->stream);
  if (mg->gexec) hipGraphExecDestroy(mg->gexec);
  if (mg->graph) hipGraphDestroy(mg->graph);
  for (auto& L : mg->lv) {
    free_level_buffers(L);
    free_level_colors(L);
    free_level_patches(L);
    fh_tri_destroy(L.tri);
    L.tri = nullptr;
    if (L.direct) fh_direct_destroy(L.direct);
    L.direct = nullptr;
  }
  if (mg->direct0) fh_direct_destroy(mg->direct0);
  if (mg->d_ainv) hipFree(mg->d_ainv);
  if (mg->d_act) hipFree(mg->d_act);
  if (mg->d_hit) hipFree(mg->d_hit);
  if (mg->d_gjwork) hipFree(mg->d_gjwork);
  if (mg->d_nd) hipFree(mg->d_nd);
  if (mg->d_nd_rowoff) hipFree(mg->d_nd_rowoff);
  if (mg->d_nd_rowinfo) hipFree(mg->d_nd_rowinfo);
  if (mg->d_nd_desc) hipFree(mg->d_nd_desc);
  for (hipStream_t st : mg->nd_streams) hipStreamDestroy(st);
  for (hipEvent_t ev : mg->nd_events) hipEventDestroy(ev);
  for (double* p : mg->kv) hipFree(p);
  if (mg->d_V) hipFree(mg->d_V);
  if (mg->d_gm) hipFree(mg->d_gm);
  if (mg->h_gm) hipHostFree(mg->h_gm);
  delete mg;
  return 0;
}

// ------------------------------------------------------------------------------------------------
// outer solvers
// ------------------------------------------------------------------------------------------------
static int krylov_reserve(fh_mg_t mg, int nvec, int n) {
  if ((int)mg->kv.size() >= nvec && mg->kv_n == n) return 0;
  for (double* p : mg->kv) hipFree(p);
  mg->kv.assign(nvec, nullptr);
  for (int i = 0; i < nvec; i++) {
    FH_CHECK_HIP(hipMalloc(&mg->kv[i], ((size_t)n + 2) * sizeof(double)));
    // zero: ghost tails are read by the SpMV before any write.  debug_poison fills with NaN bit patterns instead, so that a test
    // can show that nothing ELSE of a work vector is read before it is written (tests/test_gpu_multigrid.py)
    FH_CHECK_HIP(hipMemsetAsync(mg->kv[i], mg->ctx->debug_poison ? 0xFF : 0, ((size_t)n + 2) * sizeof(double), mg->ctx->stream));
  }
  mg->kv_n = n;
  return 0;
}

static int dev_dot(fh_ctx_t c, const double* x, const double* y, int n, double* out) {
  fh_vec_s vx, vy;
  vx.ctx = vy.ctx = c;
  vx.n_local = vy.n_local = n;
  vx.d = const_cast<double*>(x);
  vy.d = const_cast<double*>(y);
  return fh_vec_dot(&vx, &vy, out);
}

static int dev_axpby(fh_ctx_t c, double* y, const double* x, double a, double b, int n) {
  hipLaunchKernelGGL(k_axpby2, dim3(sgrid(c, n)), dim3(256), 0, c->stream, y, x, a, b, n);
  FH_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int fh_mg_solve(fh_mg_t mg, fh_vec_t bv, fh_vec_t xv, int outer, double rtol, double atol, double dtol, int maxit, int restart,
                           int* iterations, double* final_residual) {
  FH_REQUIRE(mg && mg->setup_done, "fh_mg_solve: fh_mg_setup has not been called");
  fh_ctx_t c = mg->ctx;
  const int top = mg->nlevels - 1;
  fh_mat_t A = mg->lv[top].A;
  const int n = A->m;                       // owned rows
  const int ncols = mg->lv[top].ncols;      // owned + ghosts on a distributed level
  fh_halo_t HL = mg->lv[top].halo;
  FH_REQUIRE(bv->n_local >= n && xv->n_local + xv->nghost >= ncols, "fh_mg_solve: vectors too short");
  // distributed forms of the two global operations (MatMult with ghost refresh, VecDot with all-reduce)
  auto spmv = [&](double* xin, double* yout, int mode, const double* bb) -> int {
    return halo_spmv(HL, A, xin, n, yout, mode, bb, nullptr, 0.0);
  };
  auto dot = [&](const double* u, const double* w2, double* out) -> int {
    FH_TRY(dev_dot(c, u, w2, n, out));
    if (HL) FH_TRY(fh_halo_allreduce_sum(HL, out, 1));
    return 0;
  };
  FH_REQUIRE(outer >= 0 && outer <= 4, "fh_mg_solve: unknown outer solver %d", outer);
  double* b = bv->d;
  double* x = xv->d;
  int its = 0;
  double rn = 0.0;

  if (outer == FH_OUTER_PREONLY) {
    // exactly one cycle per MGSolve (LinearEquationSolverPetsc.cpp:310-313)
    FH_TRY(apply_cycle(mg, b, x));
    its = 1;
  } else if (outer == FH_OUTER_RICHARDSON) {
    // x <- x + 0.99999 M^-1 (b - A x), x0 = 0.  The scale of the OUTER Richardson is fixed by the reference itself: MGInit sets
    // _richardsonScaleFactor = .99999 around SetSolver(_ksp) and restores the user's value afterwards (:190-193), so
    // SetRichardsonScaleFactor only ever reaches the level smoothers (omega of fh_mg_set_level)
    FH_TRY(krylov_reserve(mg, 2, ncols));
    double *r = mg->kv[0], *z = mg->kv[1];
    FH_CHECK_HIP(hipMemsetAsync(x, 0, (size_t)n * sizeof(double), c->stream));
    double bn;
    FH_TRY(dot(b, b, &bn));
    bn = sqrt(bn);
    for (;;) {
      FH_TRY(spmv(x, r, 2, b));
      FH_TRY(dot(r, r, &rn));
      rn = sqrt(rn);
      if (rn <= std::max(rtol * bn, atol) || its >= maxit || rn > dtol * bn) break;
      FH_TRY(apply_cycle(mg, r, z));
      FH_TRY(dev_axpby(c, x, z, 0.99999, 1.0, n));
      its++;
    }
  } else if (outer == FH_OUTER_CG) {
    FH_TRY(krylov_reserve(mg, 4, ncols));
    double *r = mg->kv[0], *z = mg->kv[1], *p = mg->kv[2], *Ap = mg->kv[3];
    FH_CHECK_HIP(hipMemsetAsync(x, 0, (size_t)n * sizeof(double), c->stream));
    FH_CHECK_HIP(hipMemcpyAsync(r, b, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    double bn, rz, rz_new, pAp;
    FH_TRY(dot(b, b, &bn));
    bn = sqrt(bn);
    rn = bn;
    FH_TRY(apply_cycle(mg, r, z));
    FH_CHECK_HIP(hipMemcpyAsync(p, z, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    FH_TRY(dot(r, z, &rz));
    while (rn > std::max(rtol * bn, atol) && its < maxit && rn <= dtol * bn) {
      FH_TRY(spmv(p, Ap, 0, nullptr));
      FH_TRY(dot(p, Ap, &pAp));
      const double alpha = rz / pAp;
      FH_TRY(dev_axpby(c, x, p, alpha, 1.0, n));
      FH_TRY(dev_axpby(c, r, Ap, -alpha, 1.0, n));
      FH_TRY(dot(r, r, &rn));
      rn = sqrt(rn);
      its++;
      FH_TRY(apply_cycle(mg, r, z));
      FH_TRY(dot(r, z, &rz_new));
      FH_TRY(dev_axpby(c, p, z, 1.0, rz_new / rz, n));
      rz = rz_new;
    }
  } else if (outer == FH_OUTER_FGMRES) {
    // flexible GMRES(restart) (KSPFGMRES, LinearEquationSolverPetsc.cpp:506-507): RIGHT preconditioning with the vectors z_k = M^-1 v_k
    // kept, so the cycle may be a different operator at every application (GMRES level solvers); classical Gram-Schmidt, Knoll
    // guess x0 = M^-1 b, convergence on the TRUE residual norm against ||b|| (KSPConvergedDefault with a non-zero guess)
    FH_REQUIRE(restart >= 1 && restart <= 200, "fh_mg_solve: restart %d out of range", restart);
    FH_TRY(krylov_reserve(mg, 2 * restart + 2, ncols));
    double** Vv = mg->kv.data();                       // v_0 .. v_restart
    double** Zv = mg->kv.data() + restart + 1;         // z_0 .. z_{restart-1}
    double* w = mg->kv[2 * restart + 1];
    const int nb = sgrid(c, n);
    FH_TRY(fh_reserve_reduction(c, (size_t)(restart + 2) * (nb + 1) + 64));
    if (mg->d_V_n < 2 * restart + 1) {
      if (mg->d_V) FH_CHECK_HIP(hipFree(mg->d_V));
      mg->d_V = nullptr;
      mg->d_V_n = 0;
      FH_CHECK_HIP(hipMalloc(&mg->d_V, (2 * restart + 1) * sizeof(double*)));
      mg->d_V_n = 2 * restart + 1;
    }
    double** d_V = mg->d_V;
    double** d_Z = mg->d_V + restart + 1;
    FH_CHECK_HIP(hipMemcpy(d_V, mg->kv.data(), (2 * restart + 1) * sizeof(double*), hipMemcpyHostToDevice));
    std::vector<double> H((size_t)(restart + 1) * restart, 0.0), g(restart + 1), cs(restart), sn(restart), y(restart);
    FH_TRY(apply_cycle(mg, b, x));
    double bnorm;
    FH_TRY(dot(b, b, &bnorm));
    bnorm = sqrt(bnorm);
    bool done = false;
    while (!done) {
      FH_TRY(spmv(x, Vv[0], 2, b));                            // v0 = b - A x
      double beta;
      FH_TRY(dot(Vv[0], Vv[0], &beta));
      beta = sqrt(beta);
      rn = beta;
      if (beta <= std::max(rtol * bnorm, atol) || its >= maxit || beta > dtol * bnorm) break;
      FH_TRY(dev_axpby(c, Vv[0], Vv[0], 0.0, 1.0 / beta, n));
      std::fill(g.begin(), g.end(), 0.0);
      g[0] = beta;
      int kused = 0;
      for (int k = 0; k < restart; k++) {
        FH_TRY(apply_cycle(mg, Vv[k], Zv[k]));                 // z_k = M^-1 v_k
        FH_TRY(spmv(Zv[k], w, 0, nullptr));                    // w = A z_k
        hipLaunchKernelGGL(k_multidot, dim3(nb), dim3(256), 0, c->stream, (const double* const*)d_V, w, k + 1, n, c->d_red);
        hipLaunchKernelGGL(k_multidot_final, dim3(k + 1), dim3(256), 0, c->stream, c->d_red, k + 1, nb);
        if (HL) FH_TRY(fh_halo_allreduce_ptr(HL, c->d_red + (size_t)(k + 1) * nb, k + 1));
        hipLaunchKernelGGL(k_multiaxpy, dim3(nb), dim3(256), 0, c->stream, w, (const double* const*)d_V, c->d_red + (size_t)(k + 1) * nb, -1.0,
                           k + 1, n);
        FH_CHECK_HIP(hipMemcpyAsync(c->h_red, c->d_red + (size_t)(k + 1) * nb, (k + 1) * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        FH_CHECK_HIP(hipStreamSynchronize(c->stream));
        for (int j = 0; j <= k; j++) H[(size_t)j * restart + k] = c->h_red[j];
        double wn;
        FH_TRY(dot(w, w, &wn));
        wn = sqrt(wn);
        H[(size_t)(k + 1) * restart + k] = wn;
        if (wn != 0.0) FH_TRY(dev_axpby(c, Vv[k + 1], w, 1.0 / wn, 0.0, n));
        else FH_CHECK_HIP(hipMemsetAsync(Vv[k + 1], 0, (size_t)n * sizeof(double), c->stream));
        for (int j = 0; j < k; j++) {
          const double a = H[(size_t)j * restart + k], bb = H[(size_t)(j + 1) * restart + k];
          H[(size_t)j * restart + k] = cs[j] * a + sn[j] * bb;
          H[(size_t)(j + 1) * restart + k] = -sn[j] * a + cs[j] * bb;
        }
        const double a = H[(size_t)k * restart + k], bb = H[(size_t)(k + 1) * restart + k];
        const double d = hypot(a, bb);
        if (d == 0.0) {
          cs[k] = 1.0;
          sn[k] = 0.0;
          H[(size_t)k * restart + k] = 1.0;
          g[k + 1] = 0.0;
          its++;
          kused = k + 1;
          rn = 0.0;
          done = true;
          break;
        }
        cs[k] = a / d;
        sn[k] = bb / d;
        H[(size_t)k * restart + k] = d;
        H[(size_t)(k + 1) * restart + k] = 0.0;
        g[k + 1] = -sn[k] * g[k];
        g[k] = cs[k] * g[k];
        its++;
        kused = k + 1;
        rn = fabs(g[k + 1]);
        if (rn <= std::max(rtol * bnorm, atol) || its >= maxit || wn == 0.0 || rn > dtol * bnorm) {
          done = true;
          break;
        }
      }
      for (int i = kused - 1; i >= 0; i--) {
        double s2 = g[i];
        for (int j = i + 1; j < kused; j++) s2 -= H[(size_t)i * restart + j] * y[j];
        y[i] = s2 / H[(size_t)i * restart + i];
      }
      // x += Z y
      FH_CHECK_HIP(hipMemcpyAsync(c->d_red, y.data(), kused * sizeof(double), hipMemcpyHostToDevice, c->stream));
      hipLaunchKernelGGL(k_multiaxpy, dim3(nb), dim3(256), 0, c->stream, x, (const double* const*)d_Z, c->d_red, 1.0, kused, n);
      FH_CHECK_HIP(hipStreamSynchronize(c->stream));
    }
  } else if (c->gmres_device) {
    // left-preconditioned GMRES(restart), classical Gram-Schmidt, Knoll guess x0 = M^-1 b (LinearEquationSolverPetsc.cpp:294-335), device-resident: the
    // Hessenberg matrix, the rotations, the residual estimate and the convergence test stay on the device (k_gm_*); per iteration the host enqueues
    // [A v, cycle, V^T w, w -= V h, ||w||^2, k_gm_step, v_{k+1} = w / h_{k+1,k}] and reads {done, rn, iterations} back once.  Same arithmetic in the same
    // order as the host-driven form below (option gmres_device 0), which it replaces as the default.
    FH_REQUIRE(restart >= 1 && restart <= 200, "fh_mg_solve: restart %d out of range", restart);
    FH_TRY(krylov_reserve(mg, restart + 3, ncols));
    double* t = mg->lv[mg->nlevels - 1].b;
    const int nb = sgrid(c, n);
    FH_TRY(fh_reserve_reduction(c, (size_t)(restart + 2) * (nb + 1) + nb + 64));
    if (mg->d_V_n < restart + 1) {
      if (mg->d_V) FH_CHECK_HIP(hipFree(mg->d_V));
      mg->d_V = nullptr;
      mg->d_V_n = 0;
      FH_CHECK_HIP(hipMalloc(&mg->d_V, (restart + 1) * sizeof(double*)));
      mg->d_V_n = restart + 1;
    }
    double** d_V = mg->d_V;
    FH_CHECK_HIP(hipMemcpyAsync(d_V, mg->kv.data(), (restart + 1) * sizeof(double*), hipMemcpyHostToDevice, c->stream));
    if (mg->gm_cap < gm_state_doubles(restart)) {
      if (mg->d_gm) FH_CHECK_HIP(hipFree(mg->d_gm));
      mg->d_gm = nullptr;
      mg->gm_cap = 0;
      FH_CHECK_HIP(hipMalloc(&mg->d_gm, gm_state_doubles(restart) * sizeof(double)));
      mg->gm_cap = gm_state_doubles(restart);
    }
    if (!mg->h_gm) FH_CHECK_HIP(hipHostMalloc(&mg->h_gm, GM_HDR * sizeof(double)));
    double* S = mg->d_gm;
    // scratch inside the reduction buffer: partial sums [0, (restart + 1) * nb), the projections h behind them, then the partials of ||w||^2 and its sum
    double* hcol_of_k = nullptr;
    double* sqp = c->d_red + (size_t)(restart + 2) * (nb + 1);
    double* sq1 = sqp + nb;
    auto sqnorm = [&](const double* v) -> int {          // sq1[0] = ||v||^2 over all ranks
      hipLaunchKernelGGL(k_sqnorm_part, dim3(nb), dim3(256), 0, c->stream, v, n, sqp);
      hipLaunchKernelGGL(k_sum_part, dim3(1), dim3(256), 0, c->stream, sqp, nb, sq1);
      if (HL) FH_TRY(fh_halo_allreduce_ptr(HL, sq1, 1));
      return 0;
    };
    auto readback = [&]() -> int {
      FH_CHECK_HIP(hipMemcpyAsync(mg->h_gm, S, GM_HDR * sizeof(double), hipMemcpyDeviceToHost, c->stream));
      FH_CHECK_HIP(hipStreamSynchronize(c->stream));
      return 0;
    };
    // Knoll: x0 = M^-1 b ; reference norm = ||M^-1 b||
    FH_TRY(apply_cycle(mg, b, x));
    FH_TRY(sqnorm(x));
    hipLaunchKernelGGL(k_gm_begin, dim3(1), dim3(1), 0, c->stream, S, sq1, rtol, atol, dtol, maxit, restart);
    bool done = false;
    while (!done) {
      FH_TRY(spmv(x, t, 2, b));                                // t = b - A x
      FH_TRY(apply_cycle(mg, t, mg->kv[0]));                  // v0 = M^-1 t
      FH_TRY(sqnorm(mg->kv[0]));
      hipLaunchKernelGGL(k_gm_restart, dim3(1), dim3(1), 0, c->stream, S, sq1);
      hipLaunchKernelGGL(k_scale_dev, dim3(nb), dim3(256), 0, c->stream, mg->kv[0], mg->kv[0], S + 5, n);
      FH_CHECK_HIP(hipGetLastError());
      FH_TRY(readback());
      rn = mg->h_gm[4];
      if (mg->h_gm[7] != 0.0) break;
      int kused = 0;
      for (int k = 0; k < restart; k++) {
        FH_TRY(spmv(mg->kv[k], t, 0, nullptr));
        FH_TRY(apply_cycle(mg, t, nullptr));
        double* w = mg->lv[mg->nlevels - 1].x;                // (an un-captured cycle alternates between its two buffers)
        // h = V^T w (one pass), w -= V h, h_{k+1,k} = ||w||
        hipLaunchKernelGGL(k_multidot, dim3(nb), dim3(256), 0, c->stream, (const double* const*)d_V, w, k + 1, n, c->d_red);
        hipLaunchKernelGGL(k_multidot_final, dim3(k + 1), dim3(256), 0, c->stream, c->d_red, k + 1, nb);
        hcol_of_k = c->d_red + (size_t)(k + 1) * nb;
        if (HL) FH_TRY(fh_halo_allreduce_ptr(HL, hcol_of_k, k + 1));
        hipLaunchKernelGGL(k_multiaxpy, dim3(nb), dim3(256), 0, c->stream, w, (const double* const*)d_V, hcol_of_k, -1.0, k + 1, n);
        FH_TRY(sqnorm(w));
        hipLaunchKernelGGL(k_gm_step, dim3(1), dim3(1), 0, c->stream, S, hcol_of_k, sq1, k);
        // v_{k+1} = w / h_{k+1,k} (zero on a happy breakdown: the scale is 0 then); never used when the test above said stop
        hipLaunchKernelGGL(k_scale_dev, dim3(nb), dim3(256), 0, c->stream, mg->kv[k + 1], w, S + 5, n);
        FH_CHECK_HIP(hipGetLastError());
        FH_TRY(readback());
        its = (int)mg->h_gm[6];
        rn = mg->h_gm[4];
        kused = k + 1;
        if (mg->h_gm[7] != 0.0) {
          done = true;
          break;
        }
      }
      // x += V y (y from the back substitution inside the last k_gm_step)
      hipLaunchKernelGGL(k_multiaxpy, dim3(nb), dim3(256), 0, c->stream, x, (const double* const*)d_V, S + GM_HDR + (restart + 1) + 2 * restart, 1.0, kused, n);
      FH_CHECK_HIP(hipGetLastError());
    }
  } else {
    // the same solver driven from the host (option gmres_device 0): two synchronisations per iteration, the small dense algebra on the host
    FH_REQUIRE(restart >= 1 && restart <= 200, "fh_mg_solve: restart %d out of range", restart);
    FH_TRY(krylov_reserve(mg, restart + 3, ncols));
    // t = the cycle's own right-hand-side buffer (the products A v land where the cycle reads them), w = wherever the cycle leaves its result
    double* t = mg->lv[mg->nlevels - 1].b;
    double* w = nullptr;
    const int nb = sgrid(c, n);
    FH_TRY(fh_reserve_reduction(c, (size_t)(restart + 2) * (nb + 1) + 64));
    // basis pointers on the device: owned by the solver object (an early error return must not leak them)
    if (mg->d_V_n < restart + 1) {
      if (mg->d_V) FH_CHECK_HIP(hipFree(mg->d_V));
      mg->d_V = nullptr;
      mg->d_V_n = 0;
      FH_CHECK_HIP(hipMalloc(&mg->d_V, (restart + 1) * sizeof(double*)));
      mg->d_V_n = restart + 1;
    }
    double** d_V = mg->d_V;
    FH_CHECK_HIP(hipMemcpy(d_V, mg->kv.data(), (restart + 1) * sizeof(double*), hipMemcpyHostToDevice));
    std::vector<double> H((size_t)(restart + 1) * restart, 0.0), g(restart + 1), cs(restart), sn(restart), y(restart);
    // Knoll: x0 = M^-1 b ; reference norm = ||M^-1 b||
    FH_TRY(apply_cycle(mg, b, x));
    double beta0;
    FH_TRY(dot(x, x, &beta0));
    beta0 = sqrt(beta0);
    bool done = false;
    while (!done) {
      FH_TRY(spmv(x, t, 2, b));                                // t = b - A x
      FH_TRY(apply_cycle(mg, t, mg->kv[0]));                  // v0 = M^-1 t
      double beta;
      FH_TRY(dot(mg->kv[0], mg->kv[0], &beta));
      beta = sqrt(beta);
      rn = beta;
      if (beta <= std::max(rtol * beta0, atol) || its >= maxit || beta > dtol * beta0) break;
      FH_TRY(dev_axpby(c, mg->kv[0], mg->kv[0], 0.0, 1.0 / beta, n));
      std::fill(g.begin(), g.end(), 0.0);
      g[0] = beta;
      int kused = 0;
      for (int k = 0; k < restart; k++) {
        FH_TRY(spmv(mg->kv[k], t, 0, nullptr));
        FH_TRY(apply_cycle(mg, t, nullptr));
        w = mg->lv[mg->nlevels - 1].x;                  // (an un-captured cycle alternates between its two buffers)
        // h = V^T w (one pass), w -= V h, h_{k+1,k} = ||w||
        hipLaunchKernelGGL(k_multidot, dim3(nb), dim3(256), 0, c->stream, (const double* const*)d_V, w, k + 1, n, c->d_red);
        hipLaunchKernelGGL(k_multidot_final, dim3(k + 1), dim3(256), 0, c->stream, c->d_red, k + 1, nb);
        if (HL) FH_TRY(fh_halo_allreduce_ptr(HL, c->d_red + (size_t)(k + 1) * nb, k + 1));
        hipLaunchKernelGGL(k_multiaxpy, dim3(nb), dim3(256), 0, c->stream, w, (const double* const*)d_V, c->d_red + (size_t)(k + 1) * nb, -1.0,
                           k + 1, n);
        FH_CHECK_HIP(hipMemcpyAsync(c->h_red, c->d_red + (size_t)(k + 1) * nb, (k + 1) * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        FH_CHECK_HIP(hipStreamSynchronize(c->stream));
        for (int j = 0; j <= k; j++) H[(size_t)j * restart + k] = c->h_red[j];
        double wn;
        FH_TRY(dot(w, w, &wn));
        wn = sqrt(wn);
        H[(size_t)(k + 1) * restart + k] = wn;
        // happy breakdown (w = 0: the Krylov space is invariant): the next basis vector is never used, but it must not stay
        // uninitialised / stale either
        if (wn != 0.0) FH_TRY(dev_axpby(c, mg->kv[k + 1], w, 1.0 / wn, 0.0, n));
        else FH_CHECK_HIP(hipMemsetAsync(mg->kv[k + 1], 0, (size_t)n * sizeof(double), c->stream));
        for (int j = 0; j < k; j++) {
          const double a = H[(size_t)j * restart + k], bb = H[(size_t)(j + 1) * restart + k];
          H[(size_t)j * restart + k] = cs[j] * a + sn[j] * bb;
          H[(size_t)(j + 1) * restart + k] = -sn[j] * a + cs[j] * bb;
        }
        const double a = H[(size_t)k * restart + k], bb = H[(size_t)(k + 1) * restart + k];
        const double d = hypot(a, bb);
        if (d == 0.0) {          // column k of the Hessenberg matrix vanished entirely: nothing to rotate, nothing more to gain
          cs[k] = 1.0;
          sn[k] = 0.0;
          H[(size_t)k * restart + k] = 1.0;      // keeps the back substitution finite; g[k] stays, y[k] = g[k]
          g[k + 1] = 0.0;
          its++;
          kused = k + 1;
          rn = 0.0;
          done = true;
          break;
        }
        cs[k] = a / d;
        sn[k] = bb / d;
        H[(size_t)k * restart + k] = d;
        H[(size_t)(k + 1) * restart + k] = 0.0;
        g[k + 1] = -sn[k] * g[k];
        g[k] = cs[k] * g[k];
        its++;
        kused = k + 1;
        rn = fabs(g[k + 1]);
        if (rn <= std::max(rtol * beta0, atol) || its >= maxit || wn == 0.0 || rn > dtol * beta0) {   // dtol: KSP_DIVERGED_DTOL at every iteration
          done = true;
          break;
        }
      }
      for (int i = kused - 1; i >= 0; i--) {
        double s = g[i];
        for (int j = i + 1; j < kused; j++) s -= H[(size_t)i * restart + j] * y[j];
        y[i] = s / H[(size_t)i * restart + i];
      }
      // x += V y
      FH_CHECK_HIP(hipMemcpyAsync(c->d_red, y.data(), kused * sizeof(double), hipMemcpyHostToDevice, c->stream));
      hipLaunchKernelGGL(k_multiaxpy, dim3(nb), dim3(256), 0, c->stream, x, (const double* const*)d_V, c->d_red, 1.0, kused, n);
      FH_CHECK_HIP(hipStreamSynchronize(c->stream));
    }
  }
  FH_CHECK_HIP(hipStreamSynchronize(c->stream));
  if (iterations) *iterations = its;
  if (final_residual) *final_residual = rn;
  return 0;
}
