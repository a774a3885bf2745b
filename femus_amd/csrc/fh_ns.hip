// Steady Navier-Stokes residual and Newton Jacobian for Taylor-Hood (Q2 velocity / Q1 pressure) elements on gfx950
// (a21 of SURVEY 8; BASELINE config "003_NavierStokes lid-driven cavity").
// Replaces, as ONE batched call, the per-element loop of
//   src/08_equations/assemble/03_navier_stokes.hpp:187-413   (Gauss loop :330-395)
// whose Jacobian the reference gets by replaying an adept tape per element
//   src/08_equations/assemble/Assemble_jacobian.cpp:39-72 (compute_jacobian_outside_integration_loop)
// The Jacobian here is the hand-derived d aRes / d sol of the same residual:
//   aResV[k][i] = sum_g ( nu grad phi_i . grad u_k + phi_i (u . grad) u_k - p d_k phi_i ) w
//   aResP[i]    = sum_g -(div u) psi_i w                         Res = -aRes (:401-409)
//   d aResV[k][i] / d u_m[j] = [ delta_km ( nu grad phi_i . grad phi_j + phi_i (u . grad phi_j) ) + phi_i phi_j d_m u_k ] w
//   d aResV[k][i] / d p[j]   = -psi_j d_k phi_i w ;   d aResP[i] / d u_m[j] = -psi_i d_m phi_j w ;   d aResP / d p = 0
// System rows: variables stacked [U | V | (W) | P] (LinearEquation::GetSystemDof, KKoffset, LinearEquation.cpp:76-85).
//
// Two deterministic passes, as for the Poisson path: (1) one workgroup per element integrates the dense nd x nd element
// Jacobian (nd = 22 in 2-D, 89 in 3-D) with per-Gauss-point quantities staged in LDS; (2) one wave per CSR row adds the
// element rows that touch it in ascending element order (add_matrix_blocked order), accumulating in LDS.
// The boundary integral of :201-326 (pressure traction on faces whose normal velocity is not Dirichlet) vanishes for the
// enclosed-flow configurations served here and is not built.
#include "fh_internal.h"
#include "fh_fe.h"
#include <algorithm>

struct fh_ns_assembler_s {
  fh_ctx_t ctx = nullptr;
  int geom = 0, dim = 2, nv = 9, np = 4, nd = 22, nloc = 9, nel = 0, nnode = 0, nq1 = 0, ng = 0, ndof = 0;
  int *d_elem_dof = nullptr, *d_elem_sys = nullptr;
  double *d_coords = nullptr, *d_w = nullptr, *d_phi = nullptr, *d_dphi = nullptr, *d_psi = nullptr;
  double *d_K = nullptr, *d_F = nullptr;
  int *d_adj_ptr = nullptr, *d_adj_ei = nullptr;   // row -> (element * nd + local row), ascending
  int max_row = 0;
};

struct NsParams {
  const int* elem_dof;
  const double* coords;
  const double *w, *phi, *dphi, *psi;
  const double* sol;     // system vector [U|V|(W)|P] or null
  double* K;             // [nel][nd*nd]
  double* F;             // [nel][nd]
  int nel, nloc, ng, nnode;
  double nu;
};

template <int DIM>
struct NsCfg {
  static constexpr int NV = (DIM == 2) ? 9 : 27;
  static constexpr int NP = (DIM == 2) ? 4 : 8;
  static constexpr int ND = DIM * NV + NP;
  static constexpr int NT = (DIM == 2) ? 64 : 256;
  static constexpr int EPT = (ND * ND + NT - 1) / NT;
};

template <int DIM>
__global__ __launch_bounds__(NsCfg<DIM>::NT) void k_ns_elem(NsParams P) {
  using C = NsCfg<DIM>;
  constexpr int NV = C::NV, NP = C::NP, ND = C::ND, NT = C::NT, EPT = C::EPT;
  __shared__ double xv[NV * DIM], uv[DIM * NV], pr[NP];
  __shared__ double G[NV * DIM], Jm[DIM * DIM], sc[DIM + DIM * DIM + 2];   // u[DIM], gu[DIM*DIM], p, w
  const int e = blockIdx.x, tid = threadIdx.x;
  const int* ed = P.elem_dof + (size_t)e * P.nloc;
  for (int t = tid; t < NV * DIM; t += NT) {
    const int n = t / DIM, d = t % DIM;
    xv[t] = P.coords[(size_t)ed[n] * DIM + d];
  }
  for (int t = tid; t < DIM * NV; t += NT) uv[t] = P.sol ? P.sol[(size_t)(t / NV) * P.nnode + ed[t % NV]] : 0.0;
  for (int t = tid; t < NP; t += NT) pr[t] = P.sol ? P.sol[(size_t)DIM * P.nnode + ed[t]] : 0.0;
  // this thread's entries of the element Jacobian: (row, col) -> (variable, node)
  int er[EPT], ec[EPT];
  double acc[EPT];
#pragma unroll
  for (int k = 0; k < EPT; k++) {
    const int idx = tid + k * NT;
    er[k] = (idx < ND * ND) ? idx / ND : -1;
    ec[k] = (idx < ND * ND) ? idx % ND : 0;
    acc[k] = 0.0;
  }
  double racc = 0.0;
  __syncthreads();
  for (int g = 0; g < P.ng; g++) {
    const double* dph = P.dphi + (size_t)g * NV * DIM;   // [node][dim]
    const double* ph = P.phi + (size_t)g * NV;
    const double* ps = P.psi + (size_t)g * NP;
    // Jacobian of the map: Jm[a][b] = sum_n d phi_n / d xi_a * x_n[b]   (ElemType.hpp:1462-1472 / :1206-1213)
    if (tid < DIM * DIM) {
      const int a = tid / DIM, b = tid % DIM;
      double s = 0.0;
      for (int n = 0; n < NV; n++) s += dph[n * DIM + a] * xv[n * DIM + b];
      Jm[tid] = s;
    }
    __syncthreads();
    double JI[DIM][DIM], det;   // JI[b][a] = d xi_a / d x_b
    if (DIM == 2) {
      det = Jm[0] * Jm[3] - Jm[1] * Jm[2];
      const double id = 1.0 / det;
      JI[0][0] = Jm[3] * id;
      JI[0][1] = -Jm[1] * id;
      JI[1][0] = -Jm[2] * id;
      JI[1][1] = Jm[0] * id;
    } else {
      const double a00 = Jm[0], a01 = Jm[1], a02 = Jm[2], a10 = Jm[3], a11 = Jm[4], a12 = Jm[5], a20 = Jm[6], a21 = Jm[7], a22 = Jm[8];
      det = a00 * (a11 * a22 - a12 * a21) + a01 * (a12 * a20 - a10 * a22) + a02 * (a10 * a21 - a11 * a20);
      const double id = 1.0 / det;
      // inverse of Jm (rows a, cols b): inv[b][a] ; JI[b][a] = inv(Jm)[b][a]
      JI[0][0] = (a11 * a22 - a12 * a21) * id;
      JI[0][1] = (a02 * a21 - a01 * a22) * id;
      JI[0][2] = (a01 * a12 - a02 * a11) * id;
      JI[1][0] = (a12 * a20 - a10 * a22) * id;
      JI[1][1] = (a00 * a22 - a02 * a20) * id;
      JI[1][2] = (a02 * a10 - a00 * a12) * id;
      JI[2][0] = (a10 * a21 - a11 * a20) * id;
      JI[2][1] = (a01 * a20 - a00 * a21) * id;
      JI[2][2] = (a00 * a11 - a01 * a10) * id;
    }
    // physical gradients: G[n][b] = sum_a d phi_n / d xi_a * JI[b][a]
    for (int t = tid; t < NV * DIM; t += NT) {
      const int n = t / DIM, b = t % DIM;
      double s = 0.0;
#pragma unroll
      for (int a = 0; a < DIM; a++) s += dph[n * DIM + a] * JI[b][a];
      G[t] = s;
    }
    __syncthreads();
    // solution at the Gauss point: u_k, d_j u_k, p
    if (tid < DIM) {
      double s = 0.0;
      for (int n = 0; n < NV; n++) s += uv[tid * NV + n] * ph[n];
      sc[tid] = s;
    } else if (tid < DIM + DIM * DIM) {
      const int k = (tid - DIM) / DIM, j = (tid - DIM) % DIM;
      double s = 0.0;
      for (int n = 0; n < NV; n++) s += uv[k * NV + n] * G[n * DIM + j];
      sc[tid] = s;
    } else if (tid == DIM + DIM * DIM) {
      double s = 0.0;
      for (int n = 0; n < NP; n++) s += pr[n] * ps[n];
      sc[tid] = s;
    }
    __syncthreads();
    const double wq = det * P.w[g];
    const double* ug = sc;
    const double* gu = sc + DIM;
    const double pg = sc[DIM + DIM * DIM];
#pragma unroll
    for (int k = 0; k < EPT; k++) {
      if (er[k] < 0) continue;
      const int r = er[k], c = ec[k];
      const int kr = (r < DIM * NV) ? r / NV : DIM, i = r - kr * NV;
      const int kc = (c < DIM * NV) ? c / NV : DIM, j = c - kc * NV;
      double v = 0.0;
      if (kr < DIM && kc < DIM) {
        v = ph[i] * ph[j] * gu[kr * DIM + kc];
        if (kr == kc) {
          double lap = 0.0, adv = 0.0;
#pragma unroll
          for (int d = 0; d < DIM; d++) {
            lap += G[i * DIM + d] * G[j * DIM + d];
            adv += ug[d] * G[j * DIM + d];
          }
          v += P.nu * lap + ph[i] * adv;
        }
      } else if (kr < DIM && kc == DIM) {
        v = -ps[j] * G[i * DIM + kr];
      } else if (kr == DIM && kc < DIM) {
        v = -ps[i] * G[j * DIM + kc];
      }
      acc[k] += v * wq;
    }
    if (tid < ND) {
      const int r = tid;
      const int kr = (r < DIM * NV) ? r / NV : DIM, i = r - kr * NV;
      double v;
      if (kr < DIM) {
        double lap = 0.0, adv = 0.0;
#pragma unroll
        for (int d = 0; d < DIM; d++) {
          lap += G[i * DIM + d] * gu[kr * DIM + d];
          adv += ug[d] * gu[kr * DIM + d];
        }
        v = -(P.nu * lap + ph[i] * adv - pg * G[i * DIM + kr]);
      } else {
        double div = 0.0;
#pragma unroll
        for (int d = 0; d < DIM; d++) div += gu[d * DIM + d];
        v = div * ps[i];
      }
      racc += v * wq;
    }
    __syncthreads();
  }
  double* Ke = P.K + (size_t)e * ND * ND;
#pragma unroll
  for (int k = 0; k < EPT; k++)
    if (er[k] >= 0) Ke[tid + k * NT] = acc[k];
  if (tid < ND) P.F[(size_t)e * ND + tid] = racc;
}

// pass 2: CSR row r <- sum over (element, local row) in ascending element order; one wave per row, LDS accumulator
__global__ __launch_bounds__(64) void k_sys_row_gather(const int* __restrict__ adj_ptr, const int* __restrict__ adj_ei, const int* __restrict__ elem_sys,
                                                       const double* __restrict__ K, const double* __restrict__ F, int nd,
                                                       const int* __restrict__ rowptr, const int* __restrict__ col, double* __restrict__ val,
                                                       double* __restrict__ res, int nrows) {
  extern __shared__ double rowacc[];
  const int r = blockIdx.x, lane = threadIdx.x;
  if (r >= nrows) return;
  const int rs = rowptr[r], len = rowptr[r + 1] - rs;
  for (int t = lane; t < len; t += 64) rowacc[t] = 0.0;
  __syncthreads();
  double f = 0.0;
  for (int q = adj_ptr[r]; q < adj_ptr[r + 1]; q++) {
    const int ei = adj_ei[q];
    const int e = ei / nd, i = ei - e * nd;
    const int* es = elem_sys + (size_t)e * nd;
    const double* Kr = K + ((size_t)e * nd + i) * nd;
    for (int j = lane; j < nd; j += 64) {
      const int c = es[j];
      int lo = 0, hi = len - 1;
      while (lo < hi) {
        const int mid = lo + ((hi - lo) >> 1);   // (lo + hi) overflows beyond 2^30 non-zeros
        if (col[rs + mid] < c) lo = mid + 1; else hi = mid;
      }
      rowacc[lo] += Kr[j];
    }
    if (lane == 0) f += F[(size_t)e * nd + i];
    __syncthreads();
  }
  for (int t = lane; t < len; t += 64) val[rs + t] = rowacc[t];
  if (lane == 0 && res) res[r] = f;
}

extern "C" int fh_ns_assembler_create(fh_ctx_t ctx, int geom, int gauss_order, int nel, int nloc, const int* elem_dof, int nnode, int n_vertex_nodes,
                                      const double* coords, fh_mat_t A, fh_ns_assembler_t* out) {
  FH_GUARD_BEGIN
  FH_REQUIRE(ctx && elem_dof && coords && A && out, "fh_ns_assembler_create: null argument");
  FH_REQUIRE(geom == 0 || geom == 1, "fh_ns_assembler_create: geom must be 0 (hex) or 1 (quad)");
  FH_REQUIRE(nloc == fhfe::nloc_of(geom), "fh_ns_assembler_create: nloc %d does not match the geometry", nloc);
  fh_ns_assembler_t as = new fh_ns_assembler_s();
  as->ctx = ctx;
  as->geom = geom;
  as->dim = fhfe::dim_of(geom);
  as->nv = fhfe::ndofs_of(geom, fhfe::FE_BIQUADRATIC);
  as->np = fhfe::ndofs_of(geom, fhfe::FE_LINEAR);
  as->nd = as->dim * as->nv + as->np;
  as->nloc = nloc;
  as->nel = nel;
  as->nnode = nnode;
  as->nq1 = n_vertex_nodes;
  as->ndof = as->dim * nnode + n_vertex_nodes;
  FH_REQUIRE(A->m == as->ndof && A->n == as->ndof, "fh_ns_assembler_create: matrix is %d x %d, the system has %d rows", A->m, A->n, as->ndof);
  std::vector<double> w, phi, dphi, w1, psi, dpsi;
  FH_REQUIRE(fhfe::shape_tables(geom, fhfe::FE_BIQUADRATIC, gauss_order, w, phi, dphi) == 0 &&
                 fhfe::shape_tables(geom, fhfe::FE_LINEAR, gauss_order, w1, psi, dpsi) == 0,
             "fh_ns_assembler_create: unsupported Gauss rule %d", gauss_order);
  as->ng = (int)w.size();
  const int nd = as->nd;
  std::vector<int> es((size_t)nel * nd);
  for (int e = 0; e < nel; e++) {
    const int* ed = elem_dof + (size_t)e * nloc;
    for (int i = 0; i < nloc; i++) FH_REQUIRE(ed[i] >= 0 && ed[i] < nnode, "fh_ns_assembler_create: node id %d out of range", ed[i]);
    for (int i = 0; i < as->np; i++) FH_REQUIRE(ed[i] < n_vertex_nodes, "fh_ns_assembler_create: vertex node %d is not a linear dof", ed[i]);
    int p = 0;
    for (int k = 0; k < as->dim; k++)
      for (int i = 0; i < as->nv; i++) es[(size_t)e * nd + p++] = k * nnode + ed[i];
    for (int i = 0; i < as->np; i++) es[(size_t)e * nd + p++] = as->dim * nnode + ed[i];
  }
  // row -> (element, local row) adjacency in ascending element order
  std::vector<int> aptr(as->ndof + 1, 0);
  for (size_t k = 0; k < es.size(); k++) aptr[es[k] + 1]++;
  for (int r = 0; r < as->ndof; r++) aptr[r + 1] += aptr[r];
  FH_REQUIRE((int64_t)nel * nd < 2147483647ll, "fh_ns_assembler_create: too many element rows");
  std::vector<int> aei(aptr[as->ndof]), cur(aptr.begin(), aptr.end() - 1);
  for (int e = 0; e < nel; e++)
    for (int i = 0; i < nd; i++) aei[cur[es[(size_t)e * nd + i]]++] = e * nd + i;
  // every element coupling must be in the pattern of A
  for (int e = 0; e < nel; e++)
    for (int i = 0; i < nd; i++) {
      const int r = es[(size_t)e * nd + i];
      for (int j = 0; j < nd; j++) {
        const int c = es[(size_t)e * nd + j];
        FH_REQUIRE(std::binary_search(fh_hcol(A).begin() + A->h_rowptr[r], fh_hcol(A).begin() + A->h_rowptr[r + 1], c),
                   "fh_ns_assembler_create: entry (%d, %d) is not in the matrix pattern", r, c);
      }
    }
  as->max_row = A->max_row;
  if (as->max_row == 0)
    for (int r = 0; r < A->m; r++) as->max_row = std::max(as->max_row, A->h_rowptr[r + 1] - A->h_rowptr[r]);
  FH_REQUIRE((size_t)as->max_row * sizeof(double) <= 64 * 1024, "fh_ns_assembler_create: rows of %d entries exceed the LDS accumulator", as->max_row);
  auto up = [&](void** d, const void* h, size_t bytes) -> int {
    FH_CHECK_HIP(hipMalloc(d, bytes ? bytes : 8));
    if (bytes) FH_CHECK_HIP(hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice));
    return 0;
  };
  FH_TRY(up((void**)&as->d_elem_dof, elem_dof, (size_t)nel * nloc * sizeof(int)));
  FH_TRY(up((void**)&as->d_elem_sys, es.data(), es.size() * sizeof(int)));
  FH_TRY(up((void**)&as->d_coords, coords, (size_t)nnode * as->dim * sizeof(double)));
  FH_TRY(up((void**)&as->d_w, w.data(), w.size() * sizeof(double)));
  FH_TRY(up((void**)&as->d_phi, phi.data(), phi.size() * sizeof(double)));
  FH_TRY(up((void**)&as->d_dphi, dphi.data(), dphi.size() * sizeof(double)));
  FH_TRY(up((void**)&as->d_psi, psi.data(), psi.size() * sizeof(double)));
  FH_TRY(up((void**)&as->d_adj_ptr, aptr.data(), aptr.size() * sizeof(int)));
  FH_TRY(up((void**)&as->d_adj_ei, aei.data(), aei.size() * sizeof(int)));
  FH_CHECK_HIP(hipMalloc(&as->d_K, std::max<size_t>((size_t)nel * nd * nd, 1) * sizeof(double)));
  FH_CHECK_HIP(hipMalloc(&as->d_F, std::max<size_t>((size_t)nel * nd, 1) * sizeof(double)));
  if (ctx->debug_poison) {   // tests: the row pass must read nothing the element kernel has not written
    FH_CHECK_HIP(hipMemset(as->d_K, 0xFF, std::max<size_t>((size_t)nel * nd * nd, 1) * sizeof(double)));
    FH_CHECK_HIP(hipMemset(as->d_F, 0xFF, std::max<size_t>((size_t)nel * nd, 1) * sizeof(double)));
  }
  *out = as;
  return 0;
  FH_GUARD_END("fh_ns_assembler_create")
}

extern "C" int fh_ns_assembler_destroy(fh_ns_assembler_t as) {
  if (!as) return 0;
  hipStreamSynchronize(as->ctx->stream);
  for (void* q : {(void*)as->d_elem_dof, (void*)as->d_elem_sys, (void*)as->d_coords, (void*)as->d_w, (void*)as->d_phi, (void*)as->d_dphi,
                  (void*)as->d_psi, (void*)as->d_K, (void*)as->d_F, (void*)as->d_adj_ptr, (void*)as->d_adj_ei})
    if (q) hipFree(q);
  delete as;
  return 0;
}

static int ns_element_pass(fh_ns_assembler_t as, fh_vec_t sol, double nu) {
  FH_REQUIRE(!sol || sol->n_local >= as->ndof, "navier-stokes assembly: solution vector has %d entries, the system has %d", sol ? sol->n_local : 0, as->ndof);
  NsParams P;
  P.elem_dof = as->d_elem_dof;
  P.coords = as->d_coords;
  P.w = as->d_w;
  P.phi = as->d_phi;
  P.dphi = as->d_dphi;
  P.psi = as->d_psi;
  P.sol = sol ? sol->d : nullptr;
  P.K = as->d_K;
  P.F = as->d_F;
  P.nel = as->nel;
  P.nloc = as->nloc;
  P.ng = as->ng;
  P.nnode = as->nnode;
  P.nu = nu;
  if (as->nel == 0) return 0;
  if (as->dim == 2) hipLaunchKernelGGL(k_ns_elem<2>, dim3(as->nel), dim3(NsCfg<2>::NT), 0, as->ctx->stream, P);
  else hipLaunchKernelGGL(k_ns_elem<3>, dim3(as->nel), dim3(NsCfg<3>::NT), 0, as->ctx->stream, P);
  FH_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int fh_assemble_navier_stokes(fh_ns_assembler_t as, fh_vec_t sol, double nu, fh_mat_t A, fh_vec_t res) {
  FH_REQUIRE(as && A && res, "fh_assemble_navier_stokes: null argument");
  FH_REQUIRE(A->m == as->ndof && res->n_local >= as->ndof, "fh_assemble_navier_stokes: size mismatch");
  FH_TRY(ns_element_pass(as, sol, nu));
  if (as->ndof > 0)
    hipLaunchKernelGGL(k_sys_row_gather, dim3(as->ndof), dim3(64), (size_t)as->max_row * sizeof(double), as->ctx->stream, as->d_adj_ptr, as->d_adj_ei,
                       as->d_elem_sys, as->d_K, as->d_F, as->nd, A->d_rowptr, A->d_col, A->d_val, res->d, as->ndof);
  FH_CHECK_HIP(hipGetLastError());
  A->at_valid = false;
  return 0;
}

extern "C" int fh_ns_element_matrices(fh_ns_assembler_t as, fh_vec_t sol, double nu, double* K, double* F) {
  FH_REQUIRE(as && K && F, "fh_ns_element_matrices: null argument");
  FH_TRY(ns_element_pass(as, sol, nu));
  FH_CHECK_HIP(hipMemcpyAsync(K, as->d_K, (size_t)as->nel * as->nd * as->nd * sizeof(double), hipMemcpyDeviceToHost, as->ctx->stream));
  FH_CHECK_HIP(hipMemcpyAsync(F, as->d_F, (size_t)as->nel * as->nd * sizeof(double), hipMemcpyDeviceToHost, as->ctx->stream));
  FH_CHECK_HIP(hipStreamSynchronize(as->ctx->stream));
  return 0;
}
