// Sparse exact solve on gfx950: multifrontal factorisation over a nested-dissection tree (round 4).
// What it serves: the reference hands its exact solves to MUMPS through PETSc -- the coarsest multigrid level (KSPPREONLY + PCLU,
// 03_solvers/LinearEquationSolverPetsc.hpp:131-138, LinearEquationSolverPetsc.cpp:237-287), `Solve(vars, ksp_clean)` on one level, and
// MLU_PRECOND / LU_PRECOND as the preconditioner of a level solver (PetscPreconditioner.cpp:147-160).  Up to round 3 this library inverted
// ONE dense matrix (<= 16 384 coupled unknowns, one dissection level at most); this file removes the limit for symmetric operators.
//
//   symbolic (host, once per pattern and coupled set)
//     coupled unknowns -> coupling graph -> recursive bisection into a binary tree: leaves of <= `leaf` unknowns, inner nodes = vertex
//     separators.  With coordinates the cut is a layer boundary across the principal axis next to the median (for Q2 unknowns: one plane
//     of nodes at an element boundary); without, the layers are the levels of a breadth-first search from a pseudo-peripheral unknown.
//     Unknowns are renumbered in post-order (a node's own unknowns are contiguous, ancestors come later).  Front of node t: own unknowns
//     S_t (s) and the boundary B_t (b) = ancestors' unknowns reached by fill, from the usual recurrence
//     struct(t) = (adj(S_t) u struct(children)) \ subtree(t).
//   numeric (device, every factorisation; nodes of equal height in the tree batched into the same launches)
//     D_t = A[S,S] + children's updates, E_t = A[S,B] + ..., U_t = children's updates on [B,B]      (assembly by index maps, extend-add)
//     D_t <- D_t^-1 (the batched symmetric 128-block inverse of fh_mg.hip), W_t = D_t^-1 E_t, U_t <- U_t - E_t^T W_t   (FP64 matrix cores)
//   solve (device, graph-capturable: fixed launch sequence and buffers), front vectors y_t = [r_t ; u_t] flow up the tree as the matrices did
//     up:    y_t = [b[S_t] ; 0] + children's u (gather, fixed order: deterministic) ; z_t = D_t^-1 r_t ; u_t -= W_t^T r_t
//     down:  x[S_t] = z_t - W_t x[B_t]
// Unknowns whose ROW holds nothing but its diagonal entry (the Dirichlet rows SetPenalty leaves, LinearEquationSolverPetsc.cpp:428-436) are solved
// first, x_d = b_d / a_dd, and their columns move to the right-hand side of the others (b_c - A_cd x_d): the operator [A_cc A_cd; 0 D] needs
// A_cc symmetric only, whether or not the Dirichlet columns were zeroed as well.
// Symmetric A_cc: the fast path above (symmetric block inverse without pivoting, half of the update tiles).  UNSYMMETRIC or INDEFINITE A_cc (round 5; every
// Navier-Stokes Jacobian): the same tree on the symmetrised pattern, fronts [D E; F U] with D^-1 by blocked Gauss-Jordan with partial pivoting inside
// the front and static perturbation of pivots that stay tiny (k_gi_*), W = D^-1 E, V = (F D^-1)^T, U -= F W; up sweep z = D^-1 r, u -= V^T r; steps of
// iterative refinement behind a perturbed factorisation.  A symmetric operator whose unpivoted fronts break down is factored again on this path.
#include "fh_internal.h"
#include <algorithm>
#include <cmath>
#include <numeric>

namespace {

struct DNode {
  int parent = -1, child[2] = {-1, -1};
  int height = 0;                // 0 = leaf
  int s = 0, b = 0;              // own unknowns, boundary unknowns
  int own_off = 0;               // first own unknown in the permuted numbering
  size_t D_off = 0, W_off = 0;   // into the factor buffer: D (s x s), W (s x b)
  size_t V_off = 0;              //   general fronts: V = (F D^-1)^T (s x b)
  size_t E_off = 0, U_off = 0;   // into the transient buffer: E (s x b), U (b x b)
  size_t F_off = 0, X_off = 0;   //   general fronts: F^T (s x b) = A[B, S]^T, working copy of D (s x s)
  size_t y_off = 0;              // front vector [s + b]
  size_t bidx_off = 0;           // boundary unknowns (permuted numbering, ascending)
  std::vector<int> own, bnd;     // host: permuted indices
};

struct Graph {
  std::vector<int> ptr, adj;
};

// ---- bisection of a set of graph vertices into (A, B, separator) --------------------------------------------------------------------
// keys = layer index of every vertex of the set (coordinates: quantised projection on the principal axis; none: BFS levels)
static void layer_keys(const Graph& G, const double* xyz, int dim, const std::vector<int>& set, std::vector<int>& mark /* size n, all -1 */,
                       std::vector<std::pair<double, int> >& key) {
  key.resize(set.size());
  if (xyz) {
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
      nrm = std::sqrt(nrm);
      if (!(nrm > 0.0)) break;
      for (int i = 0; i < dim; i++) v[i] = u[i] / nrm;
    }
    double span = 0.0;
    for (size_t k = 0; k < set.size(); k++) {
      double t = 0.0;
      for (int d = 0; d < dim; d++) t += v[d] * (xyz[(size_t)set[k] * dim + d] - mean[d]);
      key[k] = std::make_pair(t, set[k]);
      span = std::max(span, std::fabs(t));
    }
    const double q = span > 0.0 ? span * 1e-9 : 1.0;
    for (auto& kv : key) kv.first = std::floor(kv.first / q + 0.5);
    return;
  }
  // graph only: breadth-first levels from a pseudo-peripheral vertex of the set (two sweeps); other components follow behind
  for (int u : set) mark[u] = -2;                       // in the set, not reached
  auto bfs = [&](int start, int base, std::vector<int>& order) {
    size_t head = order.size();
    mark[start] = base;
    order.push_back(start);
    while (head < order.size()) {
      const int u = order[head++];
      for (int e = G.ptr[u]; e < G.ptr[u + 1]; e++) {
        const int w = G.adj[e];
        if (mark[w] == -2) {
          mark[w] = mark[u] + 1;
          order.push_back(w);
        }
      }
    }
  };
  std::vector<int> order;
  order.reserve(set.size());
  bfs(set[0], 0, order);
  const int far = order.back();
  for (int u : order) mark[u] = -2;
  order.clear();
  int base = 0;
  bfs(far, 0, order);
  for (int u : set)
    if (mark[u] == -2) {                               // another component
      base = mark[order.back()] + 1;
      bfs(u, base, order);
    }
  for (size_t k = 0; k < set.size(); k++) key[k] = std::make_pair((double)mark[set[k]], set[k]);
  for (int u : set) mark[u] = -1;
}

// returns false when the set cannot be cut (one layer)
static bool bisect(const Graph& G, const double* xyz, int dim, const std::vector<int>& set, std::vector<int>& mark, std::vector<int>& side /* size n, zero */,
                   std::vector<int>& A, std::vector<int>& B, std::vector<int>& S) {
  std::vector<std::pair<double, int> > key;
  layer_keys(G, xyz, dim, set, mark, key);
  std::sort(key.begin(), key.end());
  const size_t half = set.size() / 2;
  size_t c_lo = half, c_hi = half;
  while (c_lo > 0 && key[c_lo - 1].first == key[c_lo].first) c_lo--;
  while (c_hi < set.size() && c_hi > 0 && key[c_hi - 1].first == key[c_hi].first) c_hi++;
  size_t best_cut = 0, best_cnt = (size_t)-1;
  int best_side = 0;
  for (size_t cut : {c_lo, c_hi}) {
    if (cut == 0 || cut >= set.size()) continue;
    for (size_t k = 0; k < set.size(); k++) side[key[k].second] = k < cut ? 1 : 2;
    size_t cnt[3] = {0, 0, 0};
    for (size_t k = 0; k < set.size(); k++) {
      const int u = key[k].second, mine = side[u];
      bool touches = false;
      for (int e = G.ptr[u]; e < G.ptr[u + 1] && !touches; e++) touches = side[G.adj[e]] == 3 - mine;
      if (touches) cnt[mine]++;
    }
    for (int which = 1; which <= 2; which++) {
      const size_t rest = (which == 1 ? cut : set.size() - cut) - cnt[which];
      if (rest == 0) continue;
      if (cnt[which] < best_cnt) {
        best_cnt = cnt[which];
        best_cut = cut;
        best_side = which;
      }
    }
    for (size_t k = 0; k < set.size(); k++) side[key[k].second] = 0;
  }
  if (best_side == 0) return false;
  for (size_t k = 0; k < set.size(); k++) side[key[k].second] = k < best_cut ? 1 : 2;
  A.clear(), B.clear(), S.clear();
  for (size_t k = 0; k < set.size(); k++) {
    const int u = key[k].second, mine = side[u];
    bool touches = false;
    if (mine == best_side)
      for (int e = G.ptr[u]; e < G.ptr[u + 1] && !touches; e++) touches = side[G.adj[e]] == 3 - mine;
    if (touches) S.push_back(u);
    else (mine == 1 ? A : B).push_back(u);
  }
  for (size_t k = 0; k < set.size(); k++) side[key[k].second] = 0;
  std::sort(A.begin(), A.end());
  std::sort(B.begin(), B.end());
  std::sort(S.begin(), S.end());
  return !A.empty() && !B.empty();
}

static int build_tree(const Graph& G, const double* xyz, int dim, const std::vector<int>& set, int leaf, std::vector<int>& mark, std::vector<int>& side,
                      std::vector<DNode>& nodes) {
  std::vector<int> A, B, S;
  if ((int)set.size() <= leaf || !bisect(G, xyz, dim, set, mark, side, A, B, S)) {
    DNode t;
    t.own = set;
    nodes.push_back(std::move(t));
    return (int)nodes.size() - 1;
  }
  const int ca = build_tree(G, xyz, dim, A, leaf, mark, side, nodes);
  const int cb = build_tree(G, xyz, dim, B, leaf, mark, side, nodes);
  DNode t;
  t.own = S;                      // (may be empty: disconnected halves -- a node without unknowns that only joins two subtrees)
  t.child[0] = ca;
  t.child[1] = cb;
  t.height = 1 + std::max(nodes[ca].height, nodes[cb].height);
  nodes.push_back(std::move(t));
  const int me = (int)nodes.size() - 1;
  nodes[ca].parent = me;
  nodes[cb].parent = me;
  return me;
}

// ---- device kernels -------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_dd_coupling(const int* __restrict__ rowptr, const int* __restrict__ col, const double* __restrict__ val, int n,
                                                     int* __restrict__ rowhit) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= n) return;
  int hit = 0;
  for (int k = rowptr[i] + lane; k < rowptr[i + 1]; k += 64) hit |= (col[k] != i && col[k] < n && val[k] != 0.0) ? 1 : 0;
  hit = __any(hit);
  if (lane == 0) rowhit[i] = hit;
}

// symmetry of the coupled block: entries (i, j) with both rows coupled
__global__ __launch_bounds__(256) void k_dd_symmetry(const int* __restrict__ rowptr, const int* __restrict__ col, const double* __restrict__ val, int n, double tol,
                                                     const int* __restrict__ rowhit, int* __restrict__ flag) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= n || !rowhit[i]) return;
  double dmax = 0.0;
  for (int k = rowptr[i] + lane; k < rowptr[i + 1]; k += 64) dmax = fmax(dmax, fabs(val[k]));
  for (int d = 32; d >= 1; d >>= 1) dmax = fmax(dmax, __shfl_xor(dmax, d, 64));
  int bad = 0;
  for (int k = rowptr[i] + lane; k < rowptr[i + 1]; k += 64) {
    const int j = col[k];
    if (j >= n || j == i || !rowhit[j]) continue;
    int lo = rowptr[j], hi = rowptr[j + 1] - 1;
    double t = 0.0;
    while (lo <= hi) {
      const int mid = lo + ((hi - lo) >> 1);
      if (col[mid] == i) { t = val[mid]; break; }
      if (col[mid] < i) lo = mid + 1; else hi = mid - 1;
    }
    if (fabs(t - val[k]) > tol * dmax) bad = 1;
  }
  if (__any(bad) && lane == 0) atomicExch(flag, 1);
}

// assembly of the fronts from the operator: dst[k] (offset into fac or tmp, bit 62 selects tmp) <- val[src[k]]
__global__ __launch_bounds__(256) void k_dd_assemble(size_t n, const int* __restrict__ src, const long long* __restrict__ dst, const double* __restrict__ val,
                                                     double* __restrict__ fac, double* __restrict__ tmp) {
  const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  const long long d = dst[k];
  const double v = val[src[k]];
  if (d & (1ll << 62)) tmp[d & ~(1ll << 62)] = v;
  else fac[d] = v;
}

struct EaDesc {          // extend-add of one child's update matrix into its parent's front
  const double* U;       // child: b_c x b_c
  int bc;
  const int* cmap;       // [b_c] position of the child's boundary unknown in the parent's front [own | boundary]
  double *D, *E, *Up;    // parent
  int sp, bp;
  double* Ft;            // parent's F^T (s x b), general fronts only (null: symmetric, the entries below the own block mirror E)
};
__global__ __launch_bounds__(256) void k_dd_extend_add(const EaDesc* __restrict__ desc) {
  const EaDesc q = desc[blockIdx.z];
  const int i = blockIdx.y * 16 + (threadIdx.x >> 4), j = blockIdx.x * 16 + (threadIdx.x & 15);
  if (i >= q.bc || j >= q.bc) return;
  const int pi = q.cmap[i], pj = q.cmap[j];
  const double v = q.U[(size_t)i * q.bc + j];
  if (pi < q.sp) {
    if (pj < q.sp) q.D[(size_t)pi * q.sp + pj] += v;
    else q.E[(size_t)pi * q.bp + (pj - q.sp)] += v;
  } else if (pj >= q.sp)
    q.Up[(size_t)(pi - q.sp) * q.bp + (pj - q.sp)] += v;
  else if (q.Ft)
    q.Ft[(size_t)pj * q.bp + (pi - q.sp)] += v;
}

// C (M x N, ldc) += alpha P^T Q with P (K x M, ldp), Q (K x N, ldq), all row-major: 64 x 64 tile per workgroup on v_mfma_f64_16x16x4 (operand
// staging k-major with row stride 80 doubles: conflict-free fragment reads; fragments as in k_gjb_update_mfma of fh_mg.hip)
struct GemmDesc {
  const double *P, *Q;
  double* C;
  int M, N, K, ldp, ldq, ldc;
  double alpha;
  int upper_only;        // 1: C is symmetric and only tiles with tile column >= tile row are computed (mirrored by k_dd_mirror)
  int transP;            // 1: P is given as M x K (row-major, ldp): C += alpha P Q
};
typedef double dd_d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_dd_gemm_tn(const GemmDesc* __restrict__ desc) {
  constexpr int LD = 80, KS = 16;
  __shared__ double Ps[KS][LD], Qs[KS][LD];
  const GemmDesc q = desc[blockIdx.z];
  const int ti = blockIdx.y * 64, tj = blockIdx.x * 64;
  if (ti >= q.M || tj >= q.N || (q.upper_only && tj < ti)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = (wave >> 1) * 32, wj = (wave & 1) * 32;
  const int kk = lane >> 4, li = lane & 15;
  dd_d4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
      for (int r = 0; r < 4; r++) acc[a][b][r] = 0.0;
  for (int k0 = 0; k0 < q.K; k0 += KS) {
    for (int idx = tid; idx < KS * 64; idx += 256) {
      const int k = idx >> 6, c = idx & 63, kr = k0 + k;
      if (!q.transP) Ps[k][c] = (kr < q.K && ti + c < q.M) ? q.P[(size_t)kr * q.ldp + ti + c] : 0.0;
      Qs[k][c] = (kr < q.K && tj + c < q.N) ? q.Q[(size_t)kr * q.ldq + tj + c] : 0.0;
    }
    if (q.transP)                    // consecutive threads along K: sixteen-element runs of a row of P
      for (int idx = tid; idx < KS * 64; idx += 256) {
        const int c = idx >> 4, k = idx & 15, kr = k0 + k;
        Ps[k][c] = (kr < q.K && ti + c < q.M) ? q.P[(size_t)(ti + c) * q.ldp + kr] : 0.0;
      }
    __syncthreads();
#pragma unroll
    for (int k4 = 0; k4 < KS; k4 += 4) {
      const double a0 = Ps[k4 + kk][wi + li], a1 = Ps[k4 + kk][wi + 16 + li];
      const double b0 = Qs[k4 + kk][wj + li], b1 = Qs[k4 + kk][wj + 16 + li];
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
        if (i < q.M && j < q.N) q.C[(size_t)i * q.ldc + j] += q.alpha * acc[a][b][r];
      }
    }
}
// lower tiles of a symmetric C from the upper ones (after k_dd_gemm_tn with upper_only)
__global__ __launch_bounds__(256) void k_dd_mirror(const GemmDesc* __restrict__ desc) {
  const GemmDesc q = desc[blockIdx.z];
  const int ti = blockIdx.y * 64, tj = blockIdx.x * 64;
  if (!q.upper_only || ti >= q.M || tj >= q.N || tj <= ti) return;          // strictly upper tiles are copied below the diagonal
  for (int idx = threadIdx.x; idx < 64 * 64; idx += 256) {
    const int i = ti + (idx >> 6), j = tj + (idx & 63);
    if (i < q.M && j < q.N) q.C[(size_t)j * q.ldc + i] = q.C[(size_t)i * q.ldc + j];
  }
}
// inside the diagonal tiles the MFMA computed both halves from the same products in a different order: make them bit-symmetric
__global__ __launch_bounds__(256) void k_dd_mirror_diag(const GemmDesc* __restrict__ desc) {
  const GemmDesc q = desc[blockIdx.z];
  const int t0 = blockIdx.x * 64;
  if (!q.upper_only || t0 >= q.M) return;
  for (int idx = threadIdx.x; idx < 64 * 64; idx += 256) {
    const int i = t0 + (idx >> 6), j = t0 + (idx & 63);
    if (i < j && j < q.M) q.C[(size_t)j * q.ldc + i] = q.C[(size_t)i * q.ldc + j];
  }
}

// ---- general fronts (round 5): the own block D of a front inverted by BLOCKED Gauss-Jordan WITH PARTIAL PIVOTING over the whole block -----------------
// What MUMPS does for the reference's unsymmetric / indefinite level operators (every Navier-Stokes Jacobian is a saddle point with an empty pressure
// block): threshold pivoting inside the front; a pivot that stays below `tiny` = sqrt(eps) of the front's largest entry is replaced by +-tiny (static
// perturbation, counted in flag[2]; fh_direct_solve then adds steps of iterative refinement).  Per block step of 32 columns, batched over the fronts of one
// tree height (blockIdx.z / blockIdx.y = front): (1) LU with partial pivoting of the column panel below the diagonal -> the 32 pivot rows, (2) whole-row
// interchanges, (3) the block Gauss-Jordan step of fh_mg.hip's general sweep (pivot block inverted in LDS, row panel, rank-32 update on the matrix cores,
// column panel).  At the end X = (P D)^-1 and D^-1 = X P: the columns of X scattered by the row permutation.
constexpr int GI_NB = 32;
constexpr double GI_TINY = 1.5e-8;       // sqrt(machine epsilon): a pivot below GI_TINY x (largest entry of the front) is replaced (MUMPS / SuperLU_DIST static pivoting)
struct GInvDesc {
  double* M;          // n x n working matrix, row-major (in: D, out of the sweep: (P D)^-1)
  double* out;        // n x n: D^-1
  int n;
  double *Cp, *CpT;   // saved column panel, n x 32 and 32 x n
  double* Dinv;       // 32 x 32 inverse of the pivot block
  double* panel;      // n x 32 copy of the column panel for the pivot search
  int *piv, *rowid;   // [n] pivot row of every step ; [n] original row that sits in working row m at the end
  int* flag;          // [0]: a pivot block without usable pivot, [2]: number of perturbed pivots
  double* scale;      // [1] largest |entry| of D
};
static size_t ginv_work_doubles(int n) { return (size_t)n * GI_NB * 3 + GI_NB * GI_NB + 8; }
static size_t ginv_work_ints(int n) { return (size_t)2 * n + 8; }

__global__ __launch_bounds__(256) void k_gi_scale(const GInvDesc* __restrict__ desc) {
  __shared__ double sm[4];
  const GInvDesc q = desc[blockIdx.x];
  double a = 0.0;
  for (size_t k = threadIdx.x; k < (size_t)q.n * q.n; k += 256) a = fmax(a, fabs(q.M[k]));
  for (int off = 32; off > 0; off >>= 1) a = fmax(a, __shfl_xor(a, off, 64));
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) q.scale[0] = fmax(fmax(sm[0], sm[1]), fmax(sm[2], sm[3]));
}

// (1) pivot rows of block step kb: LU with partial pivoting on a copy of the column panel (rows kb .. n-1), one workgroup per front
__global__ __launch_bounds__(256) void k_gi_panel(const GInvDesc* __restrict__ desc, int kb) {
  __shared__ double smv[4];
  __shared__ int smi[4];
  __shared__ double prow[GI_NB];
  __shared__ int s_pr;
  const GInvDesc q = desc[blockIdx.x];
  if (kb >= q.n) return;
  const int n = q.n, nb = min(GI_NB, n - kb), nr = n - kb, tid = threadIdx.x;
  double* P = q.panel;
  for (int idx = tid; idx < nr * GI_NB; idx += 256) {
    const int r = idx / GI_NB, c = idx % GI_NB;
    P[idx] = c < nb ? q.M[(size_t)(kb + r) * n + kb + c] : 0.0;
  }
  __syncthreads();
  const double tiny = GI_TINY * q.scale[0];
  for (int j = 0; j < nb; j++) {
    double v = -1.0;
    int idx = j;
    for (int r = j + tid; r < nr; r += 256) {
      const double a = fabs(P[(size_t)r * GI_NB + j]);
      if (a > v) { v = a; idx = r; }
    }
    for (int off = 32; off > 0; off >>= 1) {
      const double v2 = __shfl_xor(v, off, 64);
      const int i2 = __shfl_xor(idx, off, 64);
      if (v2 > v || (v2 == v && i2 < idx)) { v = v2; idx = i2; }
    }
    if ((tid & 63) == 0) { smv[tid >> 6] = v; smi[tid >> 6] = idx; }
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < 4; w++)
        if (smv[w] > v || (smv[w] == v && smi[w] < idx)) { v = smv[w]; idx = smi[w]; }
      if (!(v > 0.0)) idx = j;
      s_pr = idx;
      q.piv[kb + j] = kb + idx;
    }
    __syncthreads();
    const int pr = s_pr;
    if (tid < GI_NB) {               // interchange, then keep the pivot row in LDS
      const double a = P[(size_t)j * GI_NB + tid], b = P[(size_t)pr * GI_NB + tid];
      P[(size_t)j * GI_NB + tid] = b;
      P[(size_t)pr * GI_NB + tid] = a;
      double pv = b;
      if (tid == j && fabs(pv) < tiny) {           // static perturbation (the pivot block's own inversion applies the same rule)
        pv = pv < 0.0 ? -tiny : tiny;
        P[(size_t)j * GI_NB + tid] = pv;
      }
      prow[tid] = pv;
    }
    __syncthreads();
    const double pinv = 1.0 / prow[j];
    for (int r = j + 1 + tid; r < nr; r += 256) {
      double* row = P + (size_t)r * GI_NB;
      const double f = row[j] * pinv;
      if (f != 0.0)
        for (int c = j + 1; c < nb; c++) row[c] -= f * prow[c];
    }
    __syncthreads();
  }
}
// (2) whole-row interchanges of the step (every thread its own column: no synchronisation between the 32 swaps)
__global__ __launch_bounds__(256) void k_gi_swap(const GInvDesc* __restrict__ desc, int kb) {
  const GInvDesc q = desc[blockIdx.y];
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (kb >= q.n || c >= q.n) return;
  const int nb = min(GI_NB, q.n - kb);
  for (int j = 0; j < nb; j++) {
    const int pr = q.piv[kb + j];
    if (pr != kb + j) {
      const double a = q.M[(size_t)(kb + j) * q.n + c], b = q.M[(size_t)pr * q.n + c];
      q.M[(size_t)(kb + j) * q.n + c] = b;
      q.M[(size_t)pr * q.n + c] = a;
    }
  }
}
// (3a) the column panel saved (both orientations), (3b) the pivot block inverted in LDS -- Gauss-Jordan with partial pivoting inside the block (after the
// interchanges the largest entry already sits on the diagonal) and the static perturbation
__global__ __launch_bounds__(256) void k_gi_save_panel(const GInvDesc* __restrict__ desc, int kb) {
  const GInvDesc q = desc[blockIdx.y];
  if (kb >= q.n) return;
  const int nb = min(GI_NB, q.n - kb);
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= q.n * GI_NB) return;
  const int i = idx / GI_NB, t = idx % GI_NB;
  const double v = t < nb ? q.M[(size_t)i * q.n + kb + t] : 0.0;
  q.Cp[(size_t)i * GI_NB + t] = v;
  q.CpT[(size_t)t * q.n + i] = v;
}
__global__ __launch_bounds__(256) void k_gi_pivot(const GInvDesc* __restrict__ desc, int kb) {
  __shared__ double M[GI_NB][GI_NB + 1];
  __shared__ double colk[GI_NB];
  __shared__ int piv[GI_NB];
  const GInvDesc q = desc[blockIdx.x];
  if (kb >= q.n) return;
  const int n = q.n, nb = min(GI_NB, n - kb), tid = threadIdx.x;
  for (int idx = tid; idx < GI_NB * GI_NB; idx += 256) {
    const int i = idx / GI_NB, j = idx % GI_NB;
    M[i][j] = (i < nb && j < nb) ? q.M[(size_t)(kb + i) * n + kb + j] : (i == j ? 1.0 : 0.0);
  }
  __syncthreads();
  const double tiny = GI_TINY * q.scale[0];
  for (int k = 0; k < nb; k++) {
    if (tid < 64) {
      double v = (tid >= k && tid < nb) ? fabs(M[tid][k]) : -1.0;
      int idx = tid;
      for (int off = 32; off > 0; off >>= 1) {
        const double v2 = __shfl_xor(v, off, 64);
        const int i2 = __shfl_xor(idx, off, 64);
        if (v2 > v || (v2 == v && i2 < idx)) { v = v2; idx = i2; }
      }
      if (tid == 0) piv[k] = (v > 0.0) ? idx : k;
    }
    __syncthreads();
    const int pr = piv[k];
    if (pr != k && tid < GI_NB) {
      const double t = M[k][tid];
      M[k][tid] = M[pr][tid];
      M[pr][tid] = t;
    }
    __syncthreads();
    if (tid == 0 && fabs(M[k][k]) < tiny) {
      M[k][k] = M[k][k] < 0.0 ? -tiny : tiny;
      atomicAdd(q.flag + 2, 1);
    }
    __syncthreads();
    if (tid < GI_NB) colk[tid] = M[tid][k];
    __syncthreads();
    const double p = 1.0 / colk[k];
    for (int idx = tid; idx < GI_NB * GI_NB; idx += 256) {
      const int i = idx / GI_NB, j = idx % GI_NB;
      if (i != k) {
        const double f = colk[i] * p;
        M[i][j] = (j == k) ? -f : M[i][j] - f * M[k][j];
      }
    }
    __syncthreads();
    if (tid < GI_NB) M[k][tid] = (tid == k) ? p : M[k][tid] * p;
    __syncthreads();
  }
  for (int k = nb - 1; k >= 0; k--) {      // the row interchanges come back as column interchanges, last first
    const int pr = piv[k];
    if (pr != k && tid < GI_NB) {
      const double t = M[tid][k];
      M[tid][k] = M[tid][pr];
      M[tid][pr] = t;
    }
    __syncthreads();
  }
  for (int idx = tid; idx < GI_NB * GI_NB; idx += 256) q.Dinv[idx] = M[idx / GI_NB][idx % GI_NB];
  if (tid == 0) {
    bool bad = false;
    for (int k = 0; k < nb; k++) bad |= !isfinite(M[k][k]);
    if (bad) atomicOr(q.flag, 1);
  }
}
// (3c) rows of the pivot block: A[kb+s, j] <- sum_t Dinv[s,t] A_old[kb+t, j] (j outside the pivot columns), Dinv inside
__global__ __launch_bounds__(64) void k_gi_row_panel(const GInvDesc* __restrict__ desc, int kb) {
  __shared__ double Ds[GI_NB][GI_NB + 1];
  const GInvDesc q = desc[blockIdx.y];
  if (kb >= q.n) return;
  const int n = q.n, nb = min(GI_NB, n - kb), tid = threadIdx.x;
  for (int idx = tid; idx < GI_NB * GI_NB; idx += 64) Ds[idx / GI_NB][idx % GI_NB] = q.Dinv[idx];
  __syncthreads();
  const int j = blockIdx.x * 64 + tid;
  if (j >= n) return;
  if (j >= kb && j < kb + nb) {
    for (int s2 = 0; s2 < nb; s2++) q.M[(size_t)(kb + s2) * n + j] = Ds[s2][j - kb];
    return;
  }
  double old[GI_NB];
#pragma unroll
  for (int t = 0; t < GI_NB; t++) old[t] = (t < nb) ? q.M[(size_t)(kb + t) * n + j] : 0.0;
  for (int s2 = 0; s2 < nb; s2++) {
    double acc = 0.0;
#pragma unroll
    for (int t = 0; t < GI_NB; t++) acc += Ds[s2][t] * old[t];
    q.M[(size_t)(kb + s2) * n + j] = acc;
  }
}
// (3d) all other rows, columns outside the pivot block: A[i,j] -= sum_t Cp[i,t] R[t,j] (R = the new row panel), 64 x 64 tiles on v_mfma_f64_16x16x4
__global__ __launch_bounds__(256) void k_gi_update(const GInvDesc* __restrict__ desc, int kb) {
  constexpr int LD = 80;
  __shared__ double Cs[GI_NB][LD], Rs[GI_NB][LD];
  const GInvDesc q = desc[blockIdx.z];
  const int n = q.n;
  const int ti = blockIdx.y * 64, tj = blockIdx.x * 64;
  if (kb >= n || ti >= n || tj >= n) return;
  const int nb = min(GI_NB, n - kb);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = (wave >> 1) * 32, wj = (wave & 1) * 32;
  const int kk = lane >> 4, li = lane & 15;
  dd_d4 acc[2][2];
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
        acc[a][b][r] = live[a][r][b] ? q.M[(size_t)i * n + j] : 0.0;
      }
    }
  for (int idx = tid; idx < GI_NB * 64; idx += 256) {
    const int k = idx >> 6, c = idx & 63;
    Cs[k][c] = (ti + c < n && k < nb) ? -q.CpT[(size_t)k * n + ti + c] : 0.0;
    Rs[k][c] = (tj + c < n && k < nb) ? q.M[(size_t)(kb + k) * n + tj + c] : 0.0;
  }
  __syncthreads();
#pragma unroll
  for (int k0 = 0; k0 < GI_NB; k0 += 4) {
    const double a0 = Cs[k0 + kk][wi + li], a1 = Cs[k0 + kk][wi + 16 + li];
    const double b0 = Rs[k0 + kk][wj + li], b1 = Rs[k0 + kk][wj + 16 + li];
    acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
  }
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int i = ti + wi + a * 16 + kk + 4 * r;
#pragma unroll
      for (int b = 0; b < 2; b++) {
        const int j = tj + wj + b * 16 + li;
        if (live[a][r][b]) q.M[(size_t)i * n + j] = acc[a][b][r];
      }
    }
}
// (3e) the pivot columns of all other rows: A[i, kb+t] <- -sum_s Cp[i,s] Dinv[s,t]
__global__ __launch_bounds__(256) void k_gi_col_panel(const GInvDesc* __restrict__ desc, int kb) {
  __shared__ double Ds[GI_NB][GI_NB + 1];
  const GInvDesc q = desc[blockIdx.y];
  if (kb >= q.n) return;
  const int n = q.n, nb = min(GI_NB, n - kb);
  for (int idx = threadIdx.x; idx < GI_NB * GI_NB; idx += 256) Ds[idx / GI_NB][idx % GI_NB] = q.Dinv[idx];
  __syncthreads();
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= n * GI_NB) return;
  const int i = idx / GI_NB, t = idx % GI_NB;
  if (t >= nb || (i >= kb && i < kb + nb)) return;
  double acc = 0.0;
  for (int s2 = 0; s2 < nb; s2++) acc += q.Cp[(size_t)i * GI_NB + s2] * Ds[s2][t];
  q.M[(size_t)i * n + kb + t] = -acc;
}
// the end: working row m holds original row rowid[m]; D^-1 = X P, i.e. column m of X is column rowid[m] of D^-1
__global__ void k_gi_rowid(const GInvDesc* __restrict__ desc) {
  const GInvDesc q = desc[blockIdx.x];
  if (threadIdx.x != 0) return;
  for (int m = 0; m < q.n; m++) q.rowid[m] = m;
  for (int k = 0; k < q.n; k++) {
    const int pr = q.piv[k];
    if (pr != k) {
      const int t = q.rowid[k];
      q.rowid[k] = q.rowid[pr];
      q.rowid[pr] = t;
    }
  }
}
__global__ __launch_bounds__(256) void k_gi_scatter(const GInvDesc* __restrict__ desc) {
  const GInvDesc q = desc[blockIdx.z];
  const int i = blockIdx.y * 16 + (threadIdx.x >> 4), m = blockIdx.x * 16 + (threadIdx.x & 15);
  if (i >= q.n || m >= q.n) return;
  q.out[(size_t)i * q.n + q.rowid[m]] = q.M[(size_t)i * q.n + m];
}
__global__ __launch_bounds__(256) void k_gi_copyin(const GInvDesc* __restrict__ desc) {
  const GInvDesc q = desc[blockIdx.y];
  for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < (size_t)q.n * q.n; k += (size_t)gridDim.x * 256) q.M[k] = q.out[k];
}
// all fronts of one height: desc[0 .. cnt) on the device, nmax = the largest order
static int ginv_batched(fh_ctx_t c, const GInvDesc* desc, int cnt, int nmax) {
  if (cnt <= 0 || nmax <= 0) return 0;
  for (int z0 = 0; z0 < cnt; z0 += 16384) {
    const int kz = std::min(16384, cnt - z0);
    const GInvDesc* d = desc + z0;
    hipLaunchKernelGGL(k_gi_copyin, dim3(std::min(1024, fh_div_up(nmax * nmax, 256)), kz), dim3(256), 0, c->stream, d);
    hipLaunchKernelGGL(k_gi_scale, dim3(kz), dim3(256), 0, c->stream, d);
    for (int kb = 0; kb < nmax; kb += GI_NB) {
      hipLaunchKernelGGL(k_gi_panel, dim3(kz), dim3(256), 0, c->stream, d, kb);
      hipLaunchKernelGGL(k_gi_swap, dim3(fh_div_up(nmax, 256), kz), dim3(256), 0, c->stream, d, kb);
      hipLaunchKernelGGL(k_gi_save_panel, dim3(fh_div_up(nmax * GI_NB, 256), kz), dim3(256), 0, c->stream, d, kb);
      hipLaunchKernelGGL(k_gi_pivot, dim3(kz), dim3(256), 0, c->stream, d, kb);
      hipLaunchKernelGGL(k_gi_row_panel, dim3(fh_div_up(nmax, 64), kz), dim3(64), 0, c->stream, d, kb);
      const int nt = fh_div_up(nmax, 64);
      hipLaunchKernelGGL(k_gi_update, dim3(nt, nt, kz), dim3(256), 0, c->stream, d, kb);
      hipLaunchKernelGGL(k_gi_col_panel, dim3(fh_div_up(nmax * GI_NB, 256), kz), dim3(256), 0, c->stream, d, kb);
    }
    hipLaunchKernelGGL(k_gi_rowid, dim3(kz), dim3(64), 0, c->stream, d);
    const int g = fh_div_up(nmax, 16);
    hipLaunchKernelGGL(k_gi_scatter, dim3(g, g, kz), dim3(256), 0, c->stream, d);
    FH_CHECK_HIP(hipGetLastError());
  }
  return 0;
}

struct SolveNode {
  const double *D, *W, *V;   // V = W on symmetric fronts
  int s, b, own_off;
  long long y_off;
  const int* bidx;           // [b] boundary unknowns (permuted numbering)
  const long long* gat[2];   // [s + b] per child: offset into y of the child's entry that lands here, -1 = none
};
// right-hand side into the permuted numbering: b_c - A_cd (b_d / a_dd) (dinv = 1 / a_dd on the decoupled unknowns, 0 elsewhere) / solution back
__global__ __launch_bounds__(256) void k_dd_gather(int na, const int* __restrict__ p2o, const double* __restrict__ b, const int* __restrict__ rowptr,
                                                   const int* __restrict__ col, const double* __restrict__ val, const double* __restrict__ dinv, int n,
                                                   double* __restrict__ bp) {
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (k >= na) return;
  const int o = p2o[k];
  double a = 0.0;
  for (int e = rowptr[o] + lane; e < rowptr[o + 1]; e += 64) {
    const int j = col[e];
    const double dj = j < n ? dinv[j] : 0.0;
    if (dj != 0.0) a += val[e] * (b[j] * dj);
  }
  for (int d = 32; d >= 1; d >>= 1) a += __shfl_xor(a, d, 64);
  if (lane == 0) bp[k] = b[o] - a;
}
__global__ __launch_bounds__(256) void k_dd_scatter(int n, int na, const int* __restrict__ p2o, const double* __restrict__ xp, const double* __restrict__ b,
                                                    const double* __restrict__ dinv, double* __restrict__ x) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  const int o = p2o[k];
  x[o] = k < na ? xp[k] : b[o] * dinv[o];
}
// up, step 1: y_t = [b[S_t] ; 0] + the children's u (child 0 first)
__global__ __launch_bounds__(256) void k_dd_up_assemble(const SolveNode* __restrict__ nodes, const int* __restrict__ list, const double* __restrict__ bp,
                                                        double* __restrict__ y) {
  const SolveNode q = nodes[list[blockIdx.y]];
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= q.s + q.b) return;
  double v = k < q.s ? bp[q.own_off + k] : 0.0;
  if (q.gat[0]) {
    const long long g = q.gat[0][k];
    if (g >= 0) v += y[g];
  }
  if (q.gat[1]) {
    const long long g = q.gat[1][k];
    if (g >= 0) v += y[g];
  }
  y[q.y_off + k] = v;
}
// up, step 2: blocks [0, ceil(s / 4)): z_t = D^-1 r_t (one wave per row) ; blocks behind: u_t[j] -= sum_k V[k][j] r_t[k] (64 columns per block; V = W when symmetric)
__global__ __launch_bounds__(256) void k_dd_up_apply(const SolveNode* __restrict__ nodes, const int* __restrict__ list, double* __restrict__ y,
                                                     double* __restrict__ zp) {
  __shared__ double part[4][64];
  const SolveNode q = nodes[list[blockIdx.y]];
  const int nrow = (q.s + 3) / 4, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const double* r = y + q.y_off;
  if ((int)blockIdx.x < nrow) {
    const int i = blockIdx.x * 4 + wave;
    if (i >= q.s) return;
    const double* row = q.D + (size_t)i * q.s;
    double a = 0.0;
    for (int k = lane; k < q.s; k += 64) a += row[k] * r[k];
    for (int d = 32; d >= 1; d >>= 1) a += __shfl_xor(a, d, 64);
    if (lane == 0) zp[q.own_off + i] = a;
    return;
  }
  const int j0 = ((int)blockIdx.x - nrow) * 64;
  if (j0 >= q.b) return;
  const int j = j0 + lane;
  double a = 0.0;
  if (j < q.b)
    for (int k = wave; k < q.s; k += 4) a += q.V[(size_t)k * q.b + j] * r[k];
  part[wave][lane] = a;
  __syncthreads();
  if (wave == 0 && j < q.b) y[q.y_off + q.s + j] -= ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
}
// down: x[S_t] = z_t - W_t x[B_t], one wave per row
__global__ __launch_bounds__(256) void k_dd_down(const SolveNode* __restrict__ nodes, const int* __restrict__ list, const double* __restrict__ zp,
                                                 double* __restrict__ xp) {
  const SolveNode q = nodes[list[blockIdx.y]];
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= q.s) return;
  const double* row = q.W + (size_t)i * q.b;
  double a = 0.0;
  for (int k = lane; k < q.b; k += 64) a += row[k] * xp[q.bidx[k]];
  for (int d = 32; d >= 1; d >>= 1) a += __shfl_xor(a, d, 64);
  if (lane == 0) xp[q.own_off + i] = zp[q.own_off + i] - a;
}
__global__ __launch_bounds__(256) void k_dd_fill(double* __restrict__ p, size_t n, double v) {
  for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t)gridDim.x * 256) p[k] = v;
}
__global__ __launch_bounds__(256) void k_dd_diag_rest(int nrest, const int* __restrict__ rest, const int* __restrict__ rowptr, const int* __restrict__ col,
                                                      const double* __restrict__ val, double* __restrict__ dinv /* [n], zeroed */, int* __restrict__ flag) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= nrest) return;
  const int i = rest[k];
  double d = 0.0;
  for (int e = rowptr[i]; e < rowptr[i + 1]; e++)
    if (col[e] == i) d = val[e];
  if (d == 0.0) atomicExch(flag, 1);
  dinv[i] = d != 0.0 ? 1.0 / d : 0.0;
}
__global__ __launch_bounds__(256) void k_dd_diag_all(int n, const int* __restrict__ rowptr, const int* __restrict__ col, const double* __restrict__ val, double* __restrict__ diag) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double d = 0.0;
  for (int e = rowptr[i]; e < rowptr[i + 1]; e++)
    if (col[e] == i) d = val[e];
  diag[i] = d;
}
__global__ __launch_bounds__(256) void k_dd_check(const double* __restrict__ p, size_t n, int* __restrict__ flag) {
  int bad = 0;
  for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t)gridDim.x * 256) bad |= !isfinite(p[k]);
  if (bad) atomicExch(flag, 1);
}

}  // namespace

struct fh_direct_s {
  fh_ctx_t ctx = nullptr;
  fh_mat_t A = nullptr;
  uint64_t A_uid = 0;
  int n = 0, na = 0;               // unknowns, coupled unknowns
  int dim = 0, leaf = 256;
  std::vector<double> xyz;         // host copy of the coordinates ([n * dim]) or empty
  // symbolic
  std::vector<int> act;            // the coupled list the tree was made for (original indices), then the decoupled ones
  std::vector<DNode> nodes;
  std::vector<std::vector<int> > by_height;     // node ids
  int root = -1, max_height = 0;
  size_t fac_doubles = 0, tmp_doubles = 0, y_doubles = 0, nmap = 0;
  // device
  int *d_hit = nullptr, *d_p2o = nullptr, *d_flags = nullptr;
  int* d_asm_src = nullptr;
  long long* d_asm_dst = nullptr;
  double *d_fac = nullptr, *d_tmp = nullptr, *d_work = nullptr, *d_y = nullptr, *d_bp = nullptr, *d_zp = nullptr, *d_xp = nullptr, *d_dinv_rest = nullptr;
  size_t work_doubles = 0;
  int* d_int = nullptr;            // bidx lists, cmaps, per-height node lists
  long long* d_gat = nullptr;
  void *d_inv = nullptr, *d_gemm = nullptr, *d_ea = nullptr, *d_snodes = nullptr;
  std::vector<int> h_list_off;     // per height: offset of its node list in d_int
  size_t list_base = 0;
  std::vector<size_t> inv_off, gemm1_off, gemm2_off, gemm3_off, ea_off[2];    // per height: first descriptor
  std::vector<int> inv_cnt, inv_nmax, gemm_cnt, gemm_maxM1, gemm_maxN, gemm_maxM2, ea_cnt[2], ea_maxb[2], max_sb, max_s, max_b;
  bool factored = false;
  int n_fronts = 0, largest_front = 0;
  // general (unsymmetric / indefinite) operators
  bool general = false;            // the layout below was made for the pivoted path
  int force_general = 0;           // fh_direct_set_general: 1 = always the pivoted path
  int zero_rule = 0;           // placement of zero-diagonal unknowns in the tree of the general fronts: 0 lowest ancestor if lonely, 1 highest neighbour front (direct_symbolic)
  int perturbed = 0;               // pivots replaced by the static perturbation in the last factorisation
  int refine = 0;                  // steps of iterative refinement in every solve (2 behind a perturbed factorisation)
  void* d_ginv = nullptr;
  double* d_gwork = nullptr;
  int* d_giwork = nullptr;
  double *d_rr = nullptr, *d_dx = nullptr;      // refinement: residual, correction
  std::vector<size_t> ginv_off;
  std::vector<int> ginv_cnt;
  uint64_t generation = 0;         // bumped by every symbolic analysis: whoever captured launches on these buffers (fh_mg's cycle graph) compares it
};
static uint64_t g_direct_generation = 0;

static void direct_free_device(fh_direct_t d) {
  for (void* p : {(void*)d->d_p2o, (void*)d->d_asm_src, (void*)d->d_asm_dst, (void*)d->d_fac, (void*)d->d_tmp, (void*)d->d_work, (void*)d->d_y, (void*)d->d_bp,
                  (void*)d->d_zp, (void*)d->d_xp, (void*)d->d_dinv_rest, (void*)d->d_int, (void*)d->d_gat, d->d_inv, d->d_gemm, d->d_ea, d->d_snodes, d->d_ginv,
                  (void*)d->d_gwork, (void*)d->d_giwork, (void*)d->d_rr, (void*)d->d_dx})
    if (p) hipFree(p);
  d->d_ginv = nullptr; d->d_gwork = nullptr; d->d_giwork = nullptr; d->d_rr = d->d_dx = nullptr;
  d->d_p2o = nullptr; d->d_asm_src = nullptr; d->d_asm_dst = nullptr; d->d_fac = d->d_tmp = d->d_work = d->d_y = d->d_bp = d->d_zp = d->d_xp = d->d_dinv_rest = nullptr;
  d->d_int = nullptr; d->d_gat = nullptr; d->d_inv = d->d_gemm = d->d_ea = d->d_snodes = nullptr;
}

extern "C" int fh_direct_create(fh_ctx_t ctx, fh_mat_t A, int dim, const double* coords, int leaf, fh_direct_t* out) {
  FH_GUARD_BEGIN
  FH_REQUIRE(ctx && A && out && A->m == A->n, "fh_direct_create: a square matrix is needed");
  FH_REQUIRE(dim == 0 || (dim >= 1 && dim <= 3 && coords), "fh_direct_create: coordinates need dim 1..3");
  fh_direct_t d = new fh_direct_s();
  d->ctx = ctx;
  d->A = A;
  d->n = A->m;
  d->dim = coords ? dim : 0;
  d->leaf = leaf > 0 ? leaf : 256;
  if (coords) d->xyz.assign(coords, coords + (size_t)A->m * dim);
  FH_CHECK_HIP(hipMalloc(&d->d_hit, ((size_t)2 * std::max(d->n, 1) + 2) * sizeof(int)));
  FH_CHECK_HIP(hipMalloc(&d->d_flags, 8 * sizeof(int)));
  *out = d;
  return 0;
  FH_GUARD_END("fh_direct_create")
}

extern "C" int fh_direct_destroy(fh_direct_t d) {
  if (!d) return 0;
  hipStreamSynchronize(d->ctx->stream);
  direct_free_device(d);
  if (d->d_hit) hipFree(d->d_hit);
  if (d->d_flags) hipFree(d->d_flags);
  delete d;
  return 0;
}

// tree, fronts, maps for the coupled set `act` (first na entries of d->act)
static int direct_symbolic(fh_direct_t d) {
  fh_mat_t A = d->A;
  const int n = d->n, na = d->na;
  const bool gen = d->general;
  direct_free_device(d);
  d->generation = ++g_direct_generation;
  d->nodes.clear();
  d->by_height.clear();
  d->factored = false;
  std::vector<int> rp(n + 1);
  FH_CHECK_HIP(hipMemcpy(rp.data(), A->d_rowptr, rp.size() * sizeof(int), hipMemcpyDeviceToHost));
  std::vector<int> cl(std::max(rp[n], 1));
  if (rp[n]) FH_CHECK_HIP(hipMemcpy(cl.data(), A->d_col, (size_t)rp[n] * sizeof(int), hipMemcpyDeviceToHost));
  // coupling graph over the coupled unknowns (positions in act), both directions
  std::vector<int> posn(n, -1);
  for (int i = 0; i < na; i++) posn[d->act[i]] = i;
  Graph G;
  {
    std::vector<std::pair<int, int> > ed;
    for (int i = 0; i < na; i++)
      for (int k = rp[d->act[i]]; k < rp[d->act[i] + 1]; k++) {
        const int j = cl[k] < n ? posn[cl[k]] : -1;
        if (j >= 0 && j != i) {
          ed.emplace_back(i, j);
          ed.emplace_back(j, i);
        }
      }
    std::sort(ed.begin(), ed.end());
    ed.erase(std::unique(ed.begin(), ed.end()), ed.end());
    G.ptr.assign(na + 1, 0);
    for (auto& e : ed) G.ptr[e.first + 1]++;
    for (int i = 0; i < na; i++) G.ptr[i + 1] += G.ptr[i];
    G.adj.resize(ed.size());
    for (size_t k = 0; k < ed.size(); k++) G.adj[k] = ed[k].second;
  }
  std::vector<double> xyz;
  if (d->dim) {
    xyz.resize((size_t)na * d->dim);
    for (int i = 0; i < na; i++)
      for (int k = 0; k < d->dim; k++) xyz[(size_t)i * d->dim + k] = d->xyz[(size_t)d->act[i] * d->dim + k];
  }
  std::vector<int> all(na), mark(na, -1), side(na, 0);
  std::iota(all.begin(), all.end(), 0);
  if (na > 0) d->root = build_tree(G, d->dim ? xyz.data() : nullptr, d->dim, all, d->leaf, mark, side, d->nodes);
  std::vector<DNode>& N = d->nodes;
  const int nn = (int)N.size();
  if (gen && na > 0) {
    // pivoting stays inside a front: an unknown with a ZERO diagonal entry (a pressure, a multiplier) none of whose neighbours lies in its own front would
    // meet an empty row there.  It moves up to the lowest ancestor that owns one of its neighbours (it couples to that subtree and to ancestors only, so
    // the separator property of the tree is kept) -- the structural part of what MUMPS does by delaying pivots.
    std::vector<double> diag(std::max(n, 1));
    double* d_diag = nullptr;
    FH_CHECK_HIP(hipMalloc(&d_diag, std::max<size_t>(n, 1) * sizeof(double)));
    hipLaunchKernelGGL(k_dd_diag_all, dim3(fh_div_up(n, 256)), dim3(256), 0, d->ctx->stream, n, A->d_rowptr, A->d_col, A->d_val, d_diag);
    const hipError_t e1 = hipMemcpyAsync(diag.data(), d_diag, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, d->ctx->stream);
    const hipError_t e2 = hipStreamSynchronize(d->ctx->stream);
    hipFree(d_diag);
    FH_CHECK_HIP(e1);
    FH_CHECK_HIP(e2);
    std::vector<int> owner(na, -1);
    for (int t = 0; t < nn; t++)
      for (int u : N[t].own) owner[u] = t;
    // Two placements.  zero_rule 0 (tried first; continuous pressures are served by it with the smaller fronts): an unknown none of whose neighbours lies
    // in its own front moves to the lowest ancestor that owns one.  zero_rule 1 (taken when the first factorisation had to perturb a pivot):
    // Round 5 (the Q2 / discontinuous-pressure Jacobian of unittests/testNSSteadyDD): "no neighbour in its front" is not enough -- a front that owns the
    // three pressure functions of an element but only some of its velocity nodes holds a singular pivot block (the element's divergence constraint reaches
    // velocities of an ancestor).  A zero-diagonal unknown therefore goes to the HIGHEST front that owns one of its neighbours: every unknown it couples to
    // is then eliminated in or below its own front.  An unknown and its neighbours in one row of the operator lie on one root path of the tree (the
    // separators separate), so the destination is an ancestor and the tree stays an elimination tree.
    std::vector<int> depth(nn, 0);
    for (int t = nn - 1; t >= 0; t--) depth[t] = N[t].parent >= 0 ? depth[N[t].parent] + 1 : 0;      // parents are numbered after their children
    int moved = 0;
    for (int t = 0; t < nn; t++) {
      if (N[t].parent < 0) continue;
      std::vector<int> keep;
      for (int u : N[t].own) {
        int dest = -1;
        if (diag[d->act[u]] == 0.0 && d->zero_rule == 1) {
          int best = t;
          for (int e = G.ptr[u]; e < G.ptr[u + 1]; e++)
            if (depth[owner[G.adj[e]]] < depth[best]) best = owner[G.adj[e]];
          if (best != t) {
            bool ancestor = false;
            for (int a = N[t].parent; a >= 0 && !ancestor; a = N[a].parent) ancestor = a == best;
            if (ancestor) dest = best;
          }
        } else if (diag[d->act[u]] == 0.0) {      // first attempt (smaller fronts): only an unknown with no neighbour in its own front moves, to the lowest ancestor that owns one
          bool lonely = true;
          for (int e = G.ptr[u]; e < G.ptr[u + 1] && lonely; e++) lonely = owner[G.adj[e]] != t;
          if (lonely)
            for (int a = N[t].parent; a >= 0 && dest < 0; a = N[a].parent)
              for (int e = G.ptr[u]; e < G.ptr[u + 1] && dest < 0; e++)
                if (owner[G.adj[e]] == a) dest = a;
        }
        if (dest >= 0) {
          N[dest].own.push_back(u);
          owner[u] = dest;
          moved++;
        } else
          keep.push_back(u);
      }
      N[t].own.swap(keep);
    }
    for (int t = 0; t < nn; t++) std::sort(N[t].own.begin(), N[t].own.end());
    if (moved) FH_TRACE("fh_direct: %d unknowns with a zero diagonal entry moved to the highest front among their neighbours", moved);
  }
  // post-order numbering: build_tree pushes children before their parent, so node order IS a post-order
  std::vector<int> perm(na), node_of(na);          // perm[new] = position in act ; inverse below
  {
    int next = 0;
    for (int t = 0; t < nn; t++) {
      N[t].own_off = next;
      N[t].s = (int)N[t].own.size();
      for (int u : N[t].own) perm[next++] = u;
    }
    FH_REQUIRE(next == na, "fh_direct: the dissection lost unknowns (%d of %d)", next, na);
  }
  std::vector<int> inv(na);
  for (int k = 0; k < na; k++) inv[perm[k]] = k;
  for (int t = 0; t < nn; t++)
    for (int k = 0; k < N[t].s; k++) {
      N[t].own[k] = N[t].own_off + k;
      node_of[N[t].own_off + k] = t;
    }
  // graph in the permuted numbering, boundary sets bottom-up
  std::vector<int> pptr(na + 1, 0), padj(G.adj.size());
  for (int k = 0; k < na; k++) pptr[k + 1] = pptr[k] + (G.ptr[perm[k] + 1] - G.ptr[perm[k]]);
  for (int k = 0; k < na; k++) {
    int o = pptr[k];
    for (int e = G.ptr[perm[k]]; e < G.ptr[perm[k] + 1]; e++) padj[o++] = inv[G.adj[e]];
    std::sort(padj.begin() + pptr[k], padj.begin() + pptr[k + 1]);
  }
  for (int t = 0; t < nn; t++) {
    const int hi = N[t].own_off + N[t].s;           // everything >= hi that is reached lies in an ancestor
    std::vector<int> st;
    for (int k = N[t].own_off; k < hi; k++)
      for (int e = pptr[k]; e < pptr[k + 1]; e++)
        if (padj[e] >= hi) st.push_back(padj[e]);
    for (int ch : N[t].child)
      if (ch >= 0)
        for (int v : N[ch].bnd)
          if (v >= hi) st.push_back(v);
    std::sort(st.begin(), st.end());
    st.erase(std::unique(st.begin(), st.end()), st.end());
    N[t].bnd.swap(st);
    N[t].b = (int)N[t].bnd.size();
  }
  // layout
  d->by_height.assign((size_t)(nn ? N[d->root].height : 0) + 1, std::vector<int>());
  d->max_height = nn ? N[d->root].height : 0;
  size_t fac = 0, tmp = 0, yo = 0, bo = 0;
  d->largest_front = 0;
  for (int t = 0; t < nn; t++) {
    DNode& q = N[t];
    d->by_height[q.height].push_back(t);
    q.D_off = fac; fac += (size_t)q.s * q.s;
    q.W_off = fac; fac += (size_t)q.s * q.b;
    q.V_off = q.W_off;
    if (gen) { q.V_off = fac; fac += (size_t)q.s * q.b; }
    q.E_off = tmp; tmp += (size_t)q.s * q.b;
    q.U_off = tmp; tmp += (size_t)q.b * q.b;
    if (gen) {
      q.F_off = tmp; tmp += (size_t)q.s * q.b;
      q.X_off = tmp; tmp += (size_t)q.s * q.s;
    }
    q.y_off = yo; yo += (size_t)q.s + q.b;
    q.bidx_off = bo; bo += (size_t)q.b;
    d->largest_front = std::max(d->largest_front, q.s + q.b);
  }
  d->n_fronts = nn;
  d->fac_doubles = fac; d->tmp_doubles = tmp; d->y_doubles = yo;
  // assembly map: operator entry -> front entry
  std::vector<int> src;
  std::vector<long long> dst;
  for (int k = 0; k < na; k++) {
    const int orow = d->act[perm[k]];
    const DNode& q = N[node_of[k]];
    const int li = k - q.own_off;
    for (int e = rp[orow]; e < rp[orow + 1]; e++) {
      const int pj0 = cl[e] < n ? posn[cl[e]] : -1;
      if (pj0 < 0) continue;
      const int pj = inv[pj0];
      if (node_of[pj] == node_of[k]) {
        src.push_back(e);
        dst.push_back((long long)(q.D_off + (size_t)li * q.s + (pj - q.own_off)));
      } else if (pj > k) {
        const auto it = std::lower_bound(q.bnd.begin(), q.bnd.end(), pj);
        FH_REQUIRE(it != q.bnd.end() && *it == pj, "fh_direct: an entry of the operator falls outside its front");
        src.push_back(e);
        dst.push_back((long long)(q.E_off + (size_t)li * q.b + (it - q.bnd.begin())) | (1ll << 62));
      } else if (gen) {          // row k lies in the boundary of the column's front: entry (k, pj) of F, stored transposed
        const DNode& qc = N[node_of[pj]];
        const auto it = std::lower_bound(qc.bnd.begin(), qc.bnd.end(), k);
        FH_REQUIRE(it != qc.bnd.end() && *it == k, "fh_direct: an entry of the operator falls outside its front");
        src.push_back(e);
        dst.push_back((long long)(qc.F_off + (size_t)(pj - qc.own_off) * qc.b + (it - qc.bnd.begin())) | (1ll << 62));
      }
    }
  }
  d->nmap = src.size();
  // integer tables: boundary lists, child maps, node lists per height ; gather tables of the up sweep
  std::vector<int> ints(bo, 0);
  for (int t = 0; t < nn; t++) std::copy(N[t].bnd.begin(), N[t].bnd.end(), ints.begin() + N[t].bidx_off);
  std::vector<size_t> cmap_off(nn, 0);
  for (int t = 0; t < nn; t++) {
    if (N[t].parent < 0) continue;
    const DNode& p = N[N[t].parent];
    cmap_off[t] = ints.size();
    for (int v : N[t].bnd) {
      int pos;
      if (v >= p.own_off && v < p.own_off + p.s) pos = v - p.own_off;
      else {
        const auto it = std::lower_bound(p.bnd.begin(), p.bnd.end(), v);
        FH_REQUIRE(it != p.bnd.end() && *it == v, "fh_direct: a child's boundary unknown is missing in its parent's front");
        pos = p.s + (int)(it - p.bnd.begin());
      }
      ints.push_back(pos);
    }
  }
  d->h_list_off.assign(d->max_height + 2, 0);
  d->list_base = ints.size();
  for (int h = 0; h <= d->max_height; h++) {
    d->h_list_off[h] = (int)(ints.size() - d->list_base);
    for (int t : d->by_height[h]) ints.push_back(t);
  }
  d->h_list_off[d->max_height + 1] = (int)(ints.size() - d->list_base);
  std::vector<long long> gat;
  std::vector<size_t> gat_off[2] = {std::vector<size_t>(nn, (size_t)-1), std::vector<size_t>(nn, (size_t)-1)};
  for (int t = 0; t < nn; t++)
    for (int w = 0; w < 2; w++) {
      const int ch = N[t].child[w];
      if (ch < 0 || N[ch].b == 0) continue;
      gat_off[w][t] = gat.size();
      gat.resize(gat.size() + (size_t)N[t].s + N[t].b, -1);
      for (int k = 0; k < N[ch].b; k++) gat[gat_off[w][t] + ints[cmap_off[ch] + k]] = (long long)(N[ch].y_off + N[ch].s + k);
    }
  // ---- device ----
  auto up = [&](void** dp, const void* h, size_t bytes) -> int {
    FH_CHECK_HIP(hipMalloc(dp, bytes ? bytes : 8));
    if (bytes) FH_CHECK_HIP(hipMemcpy(*dp, h, bytes, hipMemcpyHostToDevice));
    return 0;
  };
  std::vector<int> p2o(n);
  for (int k = 0; k < na; k++) p2o[k] = d->act[perm[k]];
  for (int k = na; k < n; k++) p2o[k] = d->act[k];
  FH_TRY(up((void**)&d->d_p2o, p2o.data(), p2o.size() * sizeof(int)));
  FH_TRY(up((void**)&d->d_asm_src, src.data(), src.size() * sizeof(int)));
  FH_TRY(up((void**)&d->d_asm_dst, dst.data(), dst.size() * sizeof(long long)));
  FH_TRY(up((void**)&d->d_int, ints.data(), ints.size() * sizeof(int)));
  FH_TRY(up((void**)&d->d_gat, gat.data(), gat.size() * sizeof(long long)));
  FH_CHECK_HIP(hipMalloc(&d->d_fac, std::max<size_t>(fac, 1) * sizeof(double)));
  FH_CHECK_HIP(hipMalloc(&d->d_tmp, std::max<size_t>(tmp, 1) * sizeof(double)));
  FH_CHECK_HIP(hipMalloc(&d->d_y, std::max<size_t>(yo, 1) * sizeof(double)));
  for (double** p : {&d->d_bp, &d->d_zp, &d->d_xp}) FH_CHECK_HIP(hipMalloc(p, std::max<size_t>(na, 1) * sizeof(double)));
  FH_CHECK_HIP(hipMalloc(&d->d_dinv_rest, std::max<size_t>(n, 1) * sizeof(double)));        // [n]: 1 / a_dd on the decoupled unknowns, 0 elsewhere
  // descriptors per height
  const int H = d->max_height + 1;
  std::vector<InvDesc> hinv;
  std::vector<GemmDesc> hg;
  std::vector<EaDesc> hea;
  std::vector<SolveNode> hs(nn);
  d->inv_off.assign(H, 0); d->inv_cnt.assign(H, 0); d->inv_nmax.assign(H, 0);
  d->gemm1_off.assign(H, 0); d->gemm2_off.assign(H, 0); d->gemm_cnt.assign(H, 0); d->gemm_maxM1.assign(H, 0); d->gemm_maxN.assign(H, 0); d->gemm_maxM2.assign(H, 0);
  d->max_sb.assign(H, 0); d->max_s.assign(H, 0); d->max_b.assign(H, 0);
  for (int w = 0; w < 2; w++) { d->ea_off[w].assign(H, 0); d->ea_cnt[w].assign(H, 0); d->ea_maxb[w].assign(H, 0); }
  size_t work = 0, gwork = 0, giwork = 0;
  for (int h = 0; h < H; h++) {
    size_t wh = 0, gh = 0, gi = 0;
    for (int t : d->by_height[h])
      if (N[t].s > 0) {
        wh += fh_inv_work_doubles(N[t].s);
        gh += ginv_work_doubles(N[t].s);
        gi += ginv_work_ints(N[t].s);
      }
    work = std::max(work, wh);
    gwork = std::max(gwork, gh);
    giwork = std::max(giwork, gi);
  }
  std::vector<GInvDesc> hginv;
  d->ginv_off.assign(H, 0);
  d->ginv_cnt.assign(H, 0);
  if (gen) {
    FH_CHECK_HIP(hipMalloc(&d->d_gwork, std::max<size_t>(gwork, 1) * sizeof(double)));
    FH_CHECK_HIP(hipMalloc(&d->d_giwork, std::max<size_t>(giwork, 1) * sizeof(int)));
    FH_CHECK_HIP(hipMalloc(&d->d_rr, std::max<size_t>(n, 1) * sizeof(double)));
    FH_CHECK_HIP(hipMalloc(&d->d_dx, std::max<size_t>(n, 1) * sizeof(double)));
    for (int h = 0; h < H; h++) {
      d->ginv_off[h] = hginv.size();
      size_t wo = 0, io = 0;
      for (int t : d->by_height[h]) {
        const DNode& q = N[t];
        if (q.s == 0) continue;
        double* w = d->d_gwork + wo;
        int* iw = d->d_giwork + io;
        wo += ginv_work_doubles(q.s);
        io += ginv_work_ints(q.s);
        GInvDesc g;
        g.M = d->d_tmp + q.X_off; g.out = d->d_fac + q.D_off; g.n = q.s;
        g.Cp = w; g.CpT = w + (size_t)q.s * GI_NB; g.panel = w + (size_t)2 * q.s * GI_NB; g.Dinv = w + (size_t)3 * q.s * GI_NB; g.scale = g.Dinv + GI_NB * GI_NB;
        g.piv = iw; g.rowid = iw + q.s; g.flag = d->d_flags + 4;
        hginv.push_back(g);
      }
      d->ginv_cnt[h] = (int)(hginv.size() - d->ginv_off[h]);
    }
  }
  d->work_doubles = work;
  FH_CHECK_HIP(hipMalloc(&d->d_work, std::max<size_t>(work, 1) * sizeof(double)));
  int* flags = d->d_flags;
  for (int h = 0; h < H; h++) {
    d->inv_off[h] = hinv.size();
    size_t wo = 0;
    for (int t : d->by_height[h]) {
      const DNode& q = N[t];
      d->max_sb[h] = std::max(d->max_sb[h], q.s + q.b);
      d->max_s[h] = std::max(d->max_s[h], q.s);
      d->max_b[h] = std::max(d->max_b[h], q.b);
      if (q.s == 0) continue;
      double* w = d->d_work + wo;
      wo += fh_inv_work_doubles(q.s);
      hinv.push_back(InvDesc{d->d_fac + q.D_off, q.s, w, w + (size_t)q.s * 128, w + (size_t)2 * q.s * 128, w + (size_t)2 * q.s * 128 + 2 * 128 * 128, flags + 2, 0});
      d->inv_nmax[h] = std::max(d->inv_nmax[h], q.s);
    }
    d->inv_cnt[h] = (int)(hinv.size() - d->inv_off[h]);
  }
  for (int h = 0; h < H; h++) {            // W = D^-1 E  (D symmetric: P = D)
    d->gemm1_off[h] = hg.size();
    for (int t : d->by_height[h]) {
      const DNode& q = N[t];
      if (q.s == 0 || q.b == 0) continue;
      // symmetric: W = D^-1 E with P = D^-1 = its transpose; general: P = D^-1 read transposed
      hg.push_back(GemmDesc{d->d_fac + q.D_off, d->d_tmp + q.E_off, d->d_fac + q.W_off, q.s, q.b, q.s, q.s, q.b, q.b, 1.0, 0, gen ? 1 : 0});
      d->gemm_maxM1[h] = std::max(d->gemm_maxM1[h], q.s);
      d->gemm_maxN[h] = std::max(d->gemm_maxN[h], q.b);
    }
    d->gemm_cnt[h] = (int)(hg.size() - d->gemm1_off[h]);
  }
  for (int h = 0; h < H; h++) {            // U -= E^T W
    d->gemm2_off[h] = hg.size();
    for (int t : d->by_height[h]) {
      const DNode& q = N[t];
      if (q.s == 0 || q.b == 0) continue;
      // symmetric: U -= E^T W (upper tiles, mirrored); general: U -= F W with F^T stored, every tile
      hg.push_back(GemmDesc{d->d_tmp + (gen ? q.F_off : q.E_off), d->d_fac + q.W_off, d->d_tmp + q.U_off, q.b, q.b, q.s, q.b, q.b, q.b, -1.0, gen ? 0 : 1, 0});
      d->gemm_maxM2[h] = std::max(d->gemm_maxM2[h], q.b);
    }
  }
  d->gemm3_off.assign(H, 0);
  if (gen)
    for (int h = 0; h < H; h++) {          // V = (F D^-1)^T = (D^-1)^T F^T
      d->gemm3_off[h] = hg.size();
      for (int t : d->by_height[h]) {
        const DNode& q = N[t];
        if (q.s == 0 || q.b == 0) continue;
        hg.push_back(GemmDesc{d->d_fac + q.D_off, d->d_tmp + q.F_off, d->d_fac + q.V_off, q.s, q.b, q.s, q.s, q.b, q.b, 1.0, 0, 0});
      }
    }
  for (int w = 0; w < 2; w++)
    for (int h = 0; h < H; h++) {
      d->ea_off[w][h] = hea.size();
      for (int t : d->by_height[h]) {
        const int ch = N[t].child[w];
        if (ch < 0 || N[ch].b == 0) continue;
        const DNode& q = N[t];
        hea.push_back(EaDesc{d->d_tmp + N[ch].U_off, N[ch].b, d->d_int + cmap_off[ch], d->d_fac + q.D_off, d->d_tmp + q.E_off, d->d_tmp + q.U_off, q.s, q.b,
                             gen ? d->d_tmp + q.F_off : nullptr});
        d->ea_maxb[w][h] = std::max(d->ea_maxb[w][h], N[ch].b);
      }
      d->ea_cnt[w][h] = (int)(hea.size() - d->ea_off[w][h]);
    }
  for (int t = 0; t < nn; t++) {
    const DNode& q = N[t];
    hs[t] = SolveNode{d->d_fac + q.D_off, d->d_fac + q.W_off, d->d_fac + q.V_off, q.s, q.b, q.own_off, (long long)q.y_off, d->d_int + q.bidx_off,
                      {gat_off[0][t] == (size_t)-1 ? nullptr : d->d_gat + gat_off[0][t], gat_off[1][t] == (size_t)-1 ? nullptr : d->d_gat + gat_off[1][t]}};
  }
  FH_TRY(up(&d->d_inv, hinv.data(), hinv.size() * sizeof(InvDesc)));
  FH_TRY(up(&d->d_ginv, hginv.data(), hginv.size() * sizeof(GInvDesc)));
  FH_TRY(up(&d->d_gemm, hg.data(), hg.size() * sizeof(GemmDesc)));
  FH_TRY(up(&d->d_ea, hea.data(), hea.size() * sizeof(EaDesc)));
  FH_TRY(up(&d->d_snodes, hs.data(), hs.size() * sizeof(SolveNode)));
  FH_TRACE("fh_direct: %d coupled unknowns (%d decoupled), %d fronts, height %d, largest front %d, factor %.1f MB + %.1f MB transient%s", na, n - na, nn,
           d->max_height, d->largest_front, fac * 8e-6, tmp * 8e-6, gen ? " (general fronts: pivoted)" : "");
  return 0;
}

// numeric factorisation on the current layout (symmetric or general fronts); *broke = a front had no usable pivot on the symmetric path
static int direct_numeric(fh_direct_t d, bool* broke) {
  fh_ctx_t c = d->ctx;
  fh_mat_t A = d->A;
  const int n = d->n, na = d->na;
  const bool gen = d->general;
  *broke = false;
  FH_CHECK_HIP(hipMemsetAsync(d->d_flags, 0, 8 * sizeof(int), c->stream));
  FH_CHECK_HIP(hipMemsetAsync(d->d_dinv_rest, 0, (size_t)n * sizeof(double), c->stream));
  if (n > na) {
    hipLaunchKernelGGL(k_dd_diag_rest, dim3(fh_div_up(n - na, 256)), dim3(256), 0, c->stream, n - na, d->d_p2o + na, A->d_rowptr, A->d_col, A->d_val, d->d_dinv_rest,
                       d->d_flags);
  }
  const int fgrid = c->num_cu * 8;
  hipLaunchKernelGGL(k_dd_fill, dim3(fgrid), dim3(256), 0, c->stream, d->d_fac, d->fac_doubles, 0.0);
  hipLaunchKernelGGL(k_dd_fill, dim3(fgrid), dim3(256), 0, c->stream, d->d_tmp, d->tmp_doubles, 0.0);
  if (d->nmap)
    hipLaunchKernelGGL(k_dd_assemble, dim3((unsigned)((d->nmap + 255) / 256)), dim3(256), 0, c->stream, d->nmap, d->d_asm_src, d->d_asm_dst, A->d_val, d->d_fac, d->d_tmp);
  FH_CHECK_HIP(hipGetLastError());
  const InvDesc* inv = static_cast<const InvDesc*>(d->d_inv);
  const GInvDesc* ginv = static_cast<const GInvDesc*>(d->d_ginv);
  const GemmDesc* gm = static_cast<const GemmDesc*>(d->d_gemm);
  const EaDesc* ea = static_cast<const EaDesc*>(d->d_ea);
  for (int h = 0; h <= d->max_height; h++) {
    for (int w = 0; w < 2; w++)          // children's updates, child 0 first (fixed order of the sums)
      if (d->ea_cnt[w][h]) {
        const int g = fh_div_up(d->ea_maxb[w][h], 16);
        for (int z0 = 0; z0 < d->ea_cnt[w][h]; z0 += 32768)
          hipLaunchKernelGGL(k_dd_extend_add, dim3(g, g, std::min(32768, d->ea_cnt[w][h] - z0)), dim3(256), 0, c->stream, ea + d->ea_off[w][h] + z0);
      }
    if (!gen) {
      if (d->inv_cnt[h]) FH_TRY(fh_inv_sym_batched(c, inv + d->inv_off[h], d->inv_cnt[h], d->inv_nmax[h]));
    } else if (d->ginv_cnt[h])
      FH_TRY(ginv_batched(c, ginv + d->ginv_off[h], d->ginv_cnt[h], d->inv_nmax[h]));
    if (d->gemm_cnt[h]) {
      for (int z0 = 0; z0 < d->gemm_cnt[h]; z0 += 32768) {
        const int kz = std::min(32768, d->gemm_cnt[h] - z0);
        const dim3 g1(fh_div_up(d->gemm_maxN[h], 64), fh_div_up(d->gemm_maxM1[h], 64), kz);
        hipLaunchKernelGGL(k_dd_gemm_tn, g1, dim3(256), 0, c->stream, gm + d->gemm1_off[h] + z0);                 // W = D^-1 E
        if (gen) hipLaunchKernelGGL(k_dd_gemm_tn, g1, dim3(256), 0, c->stream, gm + d->gemm3_off[h] + z0);        // V = (F D^-1)^T
        const int g2 = fh_div_up(d->gemm_maxM2[h], 64);
        hipLaunchKernelGGL(k_dd_gemm_tn, dim3(g2, g2, kz), dim3(256), 0, c->stream, gm + d->gemm2_off[h] + z0);    // U -= E^T W  /  U -= F W
        if (!gen) {
          hipLaunchKernelGGL(k_dd_mirror_diag, dim3(g2, 1, kz), dim3(256), 0, c->stream, gm + d->gemm2_off[h] + z0);
          hipLaunchKernelGGL(k_dd_mirror, dim3(g2, g2, kz), dim3(256), 0, c->stream, gm + d->gemm2_off[h] + z0);
        }
      }
    }
    FH_CHECK_HIP(hipGetLastError());
  }
  hipLaunchKernelGGL(k_dd_check, dim3(fgrid), dim3(256), 0, c->stream, d->d_fac, d->fac_doubles, d->d_flags + 1);
  FH_CHECK_HIP(hipGetLastError());
  int hf[8];
  FH_CHECK_HIP(hipMemcpyAsync(hf, d->d_flags, sizeof(hf), hipMemcpyDeviceToHost, c->stream));
  FH_CHECK_HIP(hipStreamSynchronize(c->stream));
  FH_REQUIRE(hf[0] == 0, "fh_direct_factor: a decoupled unknown has a zero diagonal entry (singular operator)");
  if (!gen) {
    *broke = hf[3] != 0 || hf[1] != 0;
    return 0;
  }
  FH_REQUIRE(hf[4] == 0 && hf[1] == 0, "fh_direct_factor: a front stayed singular under pivoting and perturbation (singular operator)");
  d->perturbed = hf[6];
  d->refine = hf[6] ? 3 : 0;
  if (hf[6]) FH_TRACE("fh_direct: %d pivots perturbed (static pivoting); every solve adds %d steps of iterative refinement", hf[6], d->refine);
  return 0;
}

extern "C" int fh_direct_factor(fh_direct_t d) {
  FH_GUARD_BEGIN
  FH_REQUIRE(d, "fh_direct_factor: null argument");
  fh_ctx_t c = d->ctx;
  fh_mat_t A = d->A;
  const int n = d->n;
  d->factored = false;
  d->perturbed = 0;
  d->refine = 0;
  if (n == 0) { d->factored = true; return 0; }
  // coupled unknowns and the symmetry test, one host round trip
  FH_CHECK_HIP(hipMemsetAsync(d->d_hit, 0, ((size_t)2 * n + 2) * sizeof(int), c->stream));
  hipLaunchKernelGGL(k_dd_coupling, dim3(fh_div_up(n, 4)), dim3(256), 0, c->stream, A->d_rowptr, A->d_col, A->d_val, n, d->d_hit);
  hipLaunchKernelGGL(k_dd_symmetry, dim3(fh_div_up(n, 4)), dim3(256), 0, c->stream, A->d_rowptr, A->d_col, A->d_val, n, 1e-12, d->d_hit, d->d_hit + 2 * n);
  FH_CHECK_HIP(hipGetLastError());
  std::vector<int> hit((size_t)2 * n + 2);
  FH_CHECK_HIP(hipMemcpyAsync(hit.data(), d->d_hit, hit.size() * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  FH_CHECK_HIP(hipStreamSynchronize(c->stream));
  const bool want_general = d->force_general || hit[(size_t)2 * n] != 0;      // unsymmetric coupled block: fronts [D E; F U] with pivoting
  std::vector<int> act, rest;
  for (int i = 0; i < n; i++) (hit[i] == 0 ? rest : act).push_back(i);
  const int na = (int)act.size();
  act.insert(act.end(), rest.begin(), rest.end());
  // (a layout made for general fronts also serves a symmetric operator; it is kept until the pattern changes: no flip-flop between the two)
  if (act != d->act || d->A_uid != A->uid || !d->d_p2o || (want_general && !d->general)) {
    d->act = act;
    d->na = na;
    d->A_uid = A->uid;
    d->general = want_general;
    FH_TRY(direct_symbolic(d));
  }
  bool broke = false;
  FH_TRY(direct_numeric(d, &broke));
  if (broke) {          // symmetric but not definite (a saddle point with symmetric blocks, a shifted operator): the pivoted path
    FH_TRACE("fh_direct: a symmetric front has no usable pivot without pivoting -- general fronts");
    d->general = true;
    FH_TRY(direct_symbolic(d));
    FH_TRY(direct_numeric(d, &broke));
  }
  if (d->general && d->perturbed > 0 && d->zero_rule == 0) {
    // a pivot block was singular to working precision although the operator need not be: unknowns with a zero diagonal entry (element-owned
    // pressures, multipliers) sat below velocities they constrain.  Place them in the highest front among their neighbours and factor again; the
    // placement is kept for this object.
    FH_TRACE("fh_direct: %d pivots perturbed -- zero-diagonal unknowns move to the highest front among their neighbours", d->perturbed);
    d->zero_rule = 1;
    d->perturbed = 0;
    d->refine = 0;
    FH_TRY(direct_symbolic(d));
    FH_TRY(direct_numeric(d, &broke));
  }
  d->factored = true;
  return 0;
  FH_GUARD_END("fh_direct_factor")
}

// internal: identifies the device buffers and launch shapes a captured solve refers to (fh_mg.hip's cycle signature)
// The number of refinement steps is part of the launch sequence of a solve and is set by every NUMERIC factorisation (a later Jacobian of the same
// pattern may perturb a pivot where the captured one did not, or the other way round): it is part of the identity a captured cycle compares.
uint64_t fh_direct_generation(fh_direct_t d) { return d ? (d->generation << 3) | (uint64_t)(d->refine & 7) : 0; }
extern "C" int fh_direct_stats(fh_direct_t d, int* general_fronts, int* perturbed_pivots, int* refinement_steps) {
  FH_REQUIRE(d, "fh_direct_stats: null argument");
  if (general_fronts) *general_fronts = d->general ? 1 : 0;
  if (perturbed_pivots) *perturbed_pivots = d->perturbed;
  if (refinement_steps) *refinement_steps = d->refine;
  return 0;
}
extern "C" int fh_direct_set_general(fh_direct_t d, int on) {
  FH_REQUIRE(d, "fh_direct_set_general: null argument");
  d->force_general = on ? 1 : 0;
  return 0;
}

// one sweep up and down the tree: x = (factors)^-1 b on raw device pointers
static int direct_sweeps(fh_direct_t d, const double* b, double* x) {
  fh_ctx_t c = d->ctx;
  const int n = d->n, na = d->na;
  const SolveNode* sn = static_cast<const SolveNode*>(d->d_snodes);
  const int* lists = d->d_int + d->list_base;
  if (na)
    hipLaunchKernelGGL(k_dd_gather, dim3(fh_div_up(na, 4)), dim3(256), 0, c->stream, na, d->d_p2o, b, d->A->d_rowptr, d->A->d_col, d->A->d_val, d->d_dinv_rest, n,
                       d->d_bp);
  for (int h = 0; h <= d->max_height && na; h++) {
    const int cnt = d->h_list_off[h + 1] - d->h_list_off[h];
    if (!cnt || !d->max_sb[h]) continue;
    const int* list = lists + d->h_list_off[h];
    hipLaunchKernelGGL(k_dd_up_assemble, dim3(fh_div_up(d->max_sb[h], 256), cnt), dim3(256), 0, c->stream, sn, list, d->d_bp, d->d_y);
    hipLaunchKernelGGL(k_dd_up_apply, dim3(fh_div_up(d->max_s[h], 4) + fh_div_up(d->max_b[h], 64), cnt), dim3(256), 0, c->stream, sn, list, d->d_y, d->d_zp);
  }
  for (int h = d->max_height; h >= 0 && na; h--) {
    const int cnt = d->h_list_off[h + 1] - d->h_list_off[h];
    if (!cnt || !d->max_s[h]) continue;
    hipLaunchKernelGGL(k_dd_down, dim3(fh_div_up(d->max_s[h], 4), cnt), dim3(256), 0, c->stream, sn, lists + d->h_list_off[h], d->d_zp, d->d_xp);
  }
  hipLaunchKernelGGL(k_dd_scatter, dim3(fh_div_up(n, 256)), dim3(256), 0, c->stream, n, na, d->d_p2o, d->d_xp, b, d->d_dinv_rest, x);
  FH_CHECK_HIP(hipGetLastError());
  return 0;
}
// r = b - A x (one wave per row), x += dx
__global__ __launch_bounds__(256) void k_dd_residual(const int* __restrict__ rowptr, const int* __restrict__ col, const double* __restrict__ val, int n,
                                                     const double* __restrict__ b, const double* __restrict__ x, double* __restrict__ r) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= n) return;
  double a = 0.0;
  for (int k = rowptr[i] + lane; k < rowptr[i + 1]; k += 64)
    if (col[k] < n) a += val[k] * x[col[k]];
  for (int off = 32; off >= 1; off >>= 1) a += __shfl_xor(a, off, 64);
  if (lane == 0) r[i] = b[i] - a;
}
__global__ __launch_bounds__(256) void k_dd_add(double* __restrict__ x, const double* __restrict__ dx, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) x[i] += dx[i];
}
// x = A^-1 b on raw device pointers (fixed launch sequence: capturable); behind a perturbed factorisation `refine` steps of iterative refinement
int fh_direct_solve_ptr(fh_direct_t d, const double* b, double* x) {
  FH_REQUIRE(d && d->factored, "fh_direct_solve: fh_direct_factor has not succeeded");
  fh_ctx_t c = d->ctx;
  const int n = d->n;
  if (n == 0) return 0;
  FH_TRY(direct_sweeps(d, b, x));
  for (int it = 0; it < d->refine; it++) {
    hipLaunchKernelGGL(k_dd_residual, dim3(fh_div_up(n, 4)), dim3(256), 0, c->stream, d->A->d_rowptr, d->A->d_col, d->A->d_val, n, b, x, d->d_rr);
    FH_TRY(direct_sweeps(d, d->d_rr, d->d_dx));
    hipLaunchKernelGGL(k_dd_add, dim3(fh_div_up(n, 256)), dim3(256), 0, c->stream, x, d->d_dx, n);
    FH_CHECK_HIP(hipGetLastError());
  }
  return 0;
}

extern "C" int fh_direct_solve(fh_direct_t d, fh_vec_t b, fh_vec_t x) {
  FH_REQUIRE(d && b && x && b->n_local >= d->n && x->n_local >= d->n, "fh_direct_solve: vectors too short");
  FH_REQUIRE(b->d != x->d, "fh_direct_solve: b and x must be different vectors");
  return fh_direct_solve_ptr(d, b->d, x->d);
}

extern "C" int fh_direct_info(fh_direct_t d, int* coupled, int* fronts, int* height, int* largest_front, int64_t* factor_doubles) {
  FH_REQUIRE(d, "fh_direct_info: null argument");
  if (coupled) *coupled = d->na;
  if (fronts) *fronts = d->n_fronts;
  if (height) *height = d->max_height;
  if (largest_front) *largest_front = d->largest_front;
  if (factor_doubles) *factor_doubles = (int64_t)d->fac_doubles;
  return 0;
}
