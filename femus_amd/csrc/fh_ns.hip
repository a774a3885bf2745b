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
extern "C" int fh_fe_tables_d2(int geom, int fe, int order, double* d2phi);

struct fh_ns_assembler_s {
  fh_ctx_t ctx = nullptr;
  int geom = 0, dim = 2, nv = 9, np = 4, nd = 22, nloc = 9, nel = 0, nnode = 0, nq1 = 0, ng = 0, ndof = 0;
  int *d_elem_dof = nullptr, *d_elem_sys = nullptr;
  double *d_coords = nullptr, *d_w = nullptr, *d_phi = nullptr, *d_dphi = nullptr, *d_psi = nullptr;
  double *d_K = nullptr, *d_F = nullptr;
  int *d_adj_ptr = nullptr, *d_adj_ei = nullptr;   // row -> (element * nd + local row), ascending
  int max_row = 0;
  int kind = 0;                 // 0: Taylor-Hood (03_navier_stokes.hpp), 1: equal-order linear with the Franca-Frey stabilisation (the application's callback),
                                // 2: Q2 velocity with the discontinuous piecewise-linear pressure (unittests/testNSSteadyDD), 3: scalar advection-diffusion (its temperature system)
  double* d_d2phi = nullptr;    // kind 1: second reference derivatives [ng][nv][nh]
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

// PW = false: continuous linear pressure on the vertex nodes (Taylor-Hood, 03_navier_stokes.hpp); PW = true: DISCONTINUOUS_POLYNOMIAL FIRST, the pressure
// space of the reference's known-answer test (unittests/testNSSteadyDD/main.cpp:97; callback :396-726, the same weak form): psi = 1, xi, eta (, zeta) in
// REFERENCE coordinates (quadpwLinear / hexpwLinear::eval_phi, Quadrilateral.cpp:188-200), dofs owned by the element: i * nel + iel behind the velocities
template <int DIM, bool PW = false>
struct NsCfg {
  static constexpr int NV = (DIM == 2) ? 9 : 27;
  static constexpr int NP = PW ? DIM + 1 : ((DIM == 2) ? 4 : 8);
  static constexpr int ND = DIM * NV + NP;
  static constexpr int NT = (DIM == 2) ? 64 : 256;
  static constexpr int EPT = (ND * ND + NT - 1) / NT;
};

template <int DIM, bool PW = false>
__global__ __launch_bounds__((DIM == 2) ? 64 : 256) void k_ns_elem(NsParams P) {
  using C = NsCfg<DIM, PW>;
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
  for (int t = tid; t < NP; t += NT) pr[t] = P.sol ? P.sol[(size_t)DIM * P.nnode + (PW ? (size_t)t * P.nel + e : (size_t)ed[t])] : 0.0;
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

// ------------------------------------------------------------------------------------------------------------------
// Scalar advection-diffusion in a given velocity field (kind 3): the temperature callback of the reference's known-answer test,
// unittests/testNSSteadyDD/main.cpp AssembleMatrixResT (:730-880): T and the velocities are LAGRANGE SECOND,
//   F[i]   += (-IPe grad phi_i . grad T - (u . grad T) phi_i) w          (:851)
//   B[i,j] += (IPe grad phi_i . grad phi_j + (u . grad phi_j) phi_i) w  (:854-864)
// One workgroup per element as k_ns_elem; the velocity is read from a stacked vector [U | V | (W) | ...] of stride nnode.
// ------------------------------------------------------------------------------------------------------------------
struct AdvDiffParams {
  const int* elem_dof;
  const double* coords;
  const double *w, *phi, *dphi;
  const double* sol;     // T [nnode] or null
  const double* vel;     // [dim * nnode (+ ...)] or null (pure diffusion)
  double* K;             // [nel][nv*nv]
  double* F;             // [nel][nv]
  int nel, nloc, ng, nnode;
  double ipe;
};
template <int DIM>
__global__ __launch_bounds__((DIM == 2) ? 64 : 256) void k_advdiff_elem(AdvDiffParams P) {
  constexpr int NV = (DIM == 2) ? 9 : 27, NT = (DIM == 2) ? 64 : 256, EPT = (NV * NV + NT - 1) / NT;
  __shared__ double xv[NV * DIM], uv[DIM * NV], tv[NV];
  __shared__ double G[NV * DIM], Jm[DIM * DIM], sc[2 * DIM];   // u[DIM], grad T[DIM]
  const int e = blockIdx.x, tid = threadIdx.x;
  const int* ed = P.elem_dof + (size_t)e * P.nloc;
  for (int t = tid; t < NV * DIM; t += NT) xv[t] = P.coords[(size_t)ed[t / DIM] * DIM + t % DIM];
  for (int t = tid; t < DIM * NV; t += NT) uv[t] = P.vel ? P.vel[(size_t)(t / NV) * P.nnode + ed[t % NV]] : 0.0;
  for (int t = tid; t < NV; t += NT) tv[t] = P.sol ? P.sol[ed[t]] : 0.0;
  double acc[EPT];
#pragma unroll
  for (int k = 0; k < EPT; k++) acc[k] = 0.0;
  double racc = 0.0;
  __syncthreads();
  for (int g = 0; g < P.ng; g++) {
    const double* dph = P.dphi + (size_t)g * NV * DIM;
    const double* ph = P.phi + (size_t)g * NV;
    if (tid < DIM * DIM) {
      const int a = tid / DIM, b = tid % DIM;
      double s = 0.0;
      for (int n = 0; n < NV; n++) s += dph[n * DIM + a] * xv[n * DIM + b];
      Jm[tid] = s;
    }
    __syncthreads();
    double JI[DIM][DIM], det;
    if (DIM == 2) {
      det = Jm[0] * Jm[3] - Jm[1] * Jm[2];
      const double id = 1.0 / det;
      JI[0][0] = Jm[3] * id; JI[0][1] = -Jm[1] * id; JI[1][0] = -Jm[2] * id; JI[1][1] = Jm[0] * id;
    } else {
      const double a00 = Jm[0], a01 = Jm[1], a02 = Jm[2], a10 = Jm[3], a11 = Jm[4], a12 = Jm[5], a20 = Jm[6], a21 = Jm[7], a22 = Jm[8];
      det = a00 * (a11 * a22 - a12 * a21) + a01 * (a12 * a20 - a10 * a22) + a02 * (a10 * a21 - a11 * a20);
      const double id = 1.0 / det;
      JI[0][0] = (a11 * a22 - a12 * a21) * id; JI[0][1] = (a02 * a21 - a01 * a22) * id; JI[0][2] = (a01 * a12 - a02 * a11) * id;
      JI[1][0] = (a12 * a20 - a10 * a22) * id; JI[1][1] = (a00 * a22 - a02 * a20) * id; JI[1][2] = (a02 * a10 - a00 * a12) * id;
      JI[2][0] = (a10 * a21 - a11 * a20) * id; JI[2][1] = (a01 * a20 - a00 * a21) * id; JI[2][2] = (a00 * a11 - a01 * a10) * id;
    }
    for (int t = tid; t < NV * DIM; t += NT) {
      const int n = t / DIM, b = t % DIM;
      double s = 0.0;
#pragma unroll
      for (int a = 0; a < DIM; a++) s += dph[n * DIM + a] * JI[b][a];
      G[t] = s;
    }
    __syncthreads();
    if (tid < DIM) {
      double s = 0.0;
      for (int n = 0; n < NV; n++) s += uv[tid * NV + n] * ph[n];
      sc[tid] = s;
    } else if (tid < 2 * DIM) {
      const int d = tid - DIM;
      double s = 0.0;
      for (int n = 0; n < NV; n++) s += tv[n] * G[n * DIM + d];
      sc[tid] = s;
    }
    __syncthreads();
    const double wq = det * P.w[g];
    const double* ug = sc;
    const double* gt = sc + DIM;
#pragma unroll
    for (int k = 0; k < EPT; k++) {
      const int idx = tid + k * NT;
      if (idx >= NV * NV) continue;
      const int i = idx / NV, j = idx % NV;
      double lap = 0.0, adv = 0.0;
#pragma unroll
      for (int d = 0; d < DIM; d++) {
        lap += G[i * DIM + d] * G[j * DIM + d];
        adv += ug[d] * G[j * DIM + d];
      }
      acc[k] += (P.ipe * lap + adv * ph[i]) * wq;
    }
    if (tid < NV) {
      double lap = 0.0, adv = 0.0;
#pragma unroll
      for (int d = 0; d < DIM; d++) {
        lap += G[tid * DIM + d] * gt[d];
        adv += ug[d] * gt[d];
      }
      racc += (-P.ipe * lap - adv * ph[tid]) * wq;
    }
    __syncthreads();
  }
  double* Ke = P.K + (size_t)e * NV * NV;
#pragma unroll
  for (int k = 0; k < EPT; k++)
    if (tid + k * NT < NV * NV) Ke[tid + k * NT] = acc[k];
  if (tid < NV) P.F[(size_t)e * NV + tid] = racc;
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

// ------------------------------------------------------------------------------------------------------------------
// The callback the application ships (round 5): applications/003_NavierStokes/SteadyNavierStokesParallel/main.cpp:390-925 -- equal-order LAGRANGE FIRST
// velocity and pressure (:98-108) with the Franca-Frey stabilisation (:677-868).  Per Gauss point, with u, g_kj = d_j u_k, H_k = Hessian of u_k, p, q = grad p:
//   Res_k = -q_k - sum_j u_j g_kj + IRe sum_j (H_k[jj] + H_j[kj])                                    strong residual (:742-755)
//   Rek = |u| / (4 sl IRe), sl = sqrt(lambda_k) = sqrt(6) / h_k (SetLambda for LAGRANGE FIRST, :1070-1082, :1262)
//   Rek <= 1e-15: tau = 1 / (4 sl^2 IRe), delta = 0 ; Rek < 1: tau = 1 / (4 sl^2 IRe), delta = |u|^2 / (4 sl^2 IRe) ; else tau = 1 / (|u| sl), delta = |u| / sl
//   aRhs[k][i] = { -phi_i sum_j u_j g_kj - IRe sum_j d_j phi_i (g_kj + g_jk) + (p - delta div u) d_k phi_i + Res_k tau (u . grad phi_i)
//                  - IRe tau [ Res_k lap phi_i + sum_n Res_n d_n d_k phi_i ] } W                     (:768-792, the two least-squares lines folded)
//   aRhs[p][i] = { div u phi_i - tau grad phi_i . Res } W                                            (:796-803)
// Rhs = aRhs goes to the residual vector, KKloc = -d aRhs / d Soli to the matrix (:884-910).  The reference differentiates with an adept tape; here the
// derivative is written out (tau and delta depend on u through |u| and through the branch taken):
//   dRes_k / d u_c[j] = -phi_j g_kc - [k = c] (u . grad phi_j) + IRe ([k = c] lap phi_j + d_k d_c phi_j) ;  dRes_k / d p[j] = -d_k phi_j
// One workgroup of 64 threads per element; nd = 12 (QUAD4) / 32 (HEX8).
// ------------------------------------------------------------------------------------------------------------------
struct NsStabParams {
  const int* elem_dof;
  const double* coords;
  const double *w, *phi, *dphi, *d2phi;
  const double* sol;     // system vector [U|V|(W)|P] on the vertex nodes, or null
  double* K;             // [nel][nd*nd]
  double* F;             // [nel][nd]
  int nel, nloc, ng, nq1;
  double ire;
};
template <int DIM>
__global__ __launch_bounds__(64) void k_ns_stab_elem(NsStabParams P) {
  constexpr int NV = (DIM == 2) ? 4 : 8, NH = (DIM == 2) ? 3 : 6, ND = (DIM + 1) * NV, NT = 64, EPT = (ND * ND + NT - 1) / NT;
  __shared__ double xv[NV * DIM], uv[(DIM + 1) * NV];
  __shared__ double G[NV * DIM], N[NV * NH], B[NV], L[NV], Jm[DIM * DIM];
  __shared__ double sc[64];
  // layout of sc: u[DIM] | gu[DIM*DIM] | res[DIM] | dtau[DIM] | ddel[DIM] | p, div, tau, delta, W
  constexpr int S_U = 0, S_GU = DIM, S_RES = DIM + DIM * DIM, S_DT = S_RES + DIM, S_DD = S_DT + DIM, S_P = S_DD + DIM, S_DIV = S_P + 1, S_TAU = S_P + 2,
                S_DEL = S_P + 3, S_W = S_P + 4, S_HU = S_P + 5, S_GP = S_HU + DIM * NH;
  static_assert(S_GP + DIM <= 64, "k_ns_stab_elem: scalar block");
  const int e = blockIdx.x, tid = threadIdx.x;
  const int* ed = P.elem_dof + (size_t)e * P.nloc;
  for (int t = tid; t < NV * DIM; t += NT) xv[t] = P.coords[(size_t)ed[t / DIM] * DIM + t % DIM];
  for (int t = tid; t < (DIM + 1) * NV; t += NT) uv[t] = P.sol ? P.sol[(size_t)(t / NV) * P.nq1 + ed[t % NV]] : 0.0;
  int er[EPT], ec[EPT];
  double acc[EPT];
#pragma unroll
  for (int k = 0; k < EPT; k++) {
    const int idx = tid + k * NT;
    er[k] = (idx < ND * ND) ? idx / ND : -1;
    ec[k] = (idx < ND * ND) ? idx % ND : 0;
    acc[k] = 0.0;
  }
  double racc = 0.0, sl = 0.0;
  // second-derivative index of (a, b): ElemType.hpp:1232-1244 (xx, yy, xy) / :1509-1534 (xx, yy, zz, xy, yz, zx)
  auto hidx = [](int a, int b) -> int {
    if (a == b) return a;
    if (DIM == 2) return 2;
    const int s2 = a + b;
    return s2 == 1 ? 3 : s2 == 3 ? 4 : 5;
  };
  __syncthreads();
  for (int g = 0; g < P.ng; g++) {
    const double* dph = P.dphi + (size_t)g * NV * DIM;
    const double* ph = P.phi + (size_t)g * NV;
    const double* d2 = P.d2phi + (size_t)g * NV * NH;
    if (tid < DIM * DIM) {
      const int a = tid / DIM, b = tid % DIM;
      double s = 0.0;
      for (int n = 0; n < NV; n++) s += dph[n * DIM + a] * xv[n * DIM + b];
      Jm[tid] = s;
    }
    __syncthreads();
    double JI[DIM][DIM], det;   // JI[b][a] = d xi_a / d x_b (the reference's JacI)
    if (DIM == 2) {
      det = Jm[0] * Jm[3] - Jm[1] * Jm[2];
      const double id = 1.0 / det;
      JI[0][0] = Jm[3] * id; JI[0][1] = -Jm[1] * id; JI[1][0] = -Jm[2] * id; JI[1][1] = Jm[0] * id;
    } else {
      const double a00 = Jm[0], a01 = Jm[1], a02 = Jm[2], a10 = Jm[3], a11 = Jm[4], a12 = Jm[5], a20 = Jm[6], a21 = Jm[7], a22 = Jm[8];
      det = a00 * (a11 * a22 - a12 * a21) + a01 * (a12 * a20 - a10 * a22) + a02 * (a10 * a21 - a11 * a20);
      const double id = 1.0 / det;
      JI[0][0] = (a11 * a22 - a12 * a21) * id; JI[0][1] = (a02 * a21 - a01 * a22) * id; JI[0][2] = (a01 * a12 - a02 * a11) * id;
      JI[1][0] = (a12 * a20 - a10 * a22) * id; JI[1][1] = (a00 * a22 - a02 * a20) * id; JI[1][2] = (a02 * a10 - a00 * a12) * id;
      JI[2][0] = (a10 * a21 - a11 * a20) * id; JI[2][1] = (a01 * a20 - a00 * a21) * id; JI[2][2] = (a00 * a11 - a01 * a10) * id;
    }
    if (g == 0) {                  // sqrt(lambda_k) of SetLambda for LAGRANGE FIRST: hk = (scale * Weight(0) / GaussWeight(0))^(1/dim) = (scale * det)^(1/dim)
      const double area = (DIM == 2 ? 4.0 : 8.0) * det;
      const double hk = (DIM == 2) ? sqrt(area) : cbrt(area);
      sl = sqrt(6.0 / (hk * hk));
    }
    for (int t = tid; t < NV * DIM; t += NT) {      // physical gradients
      const int n = t / DIM, b = t % DIM;
      double s = 0.0;
#pragma unroll
      for (int a = 0; a < DIM; a++) s += dph[n * DIM + a] * JI[b][a];
      G[t] = s;
    }
    for (int t = tid; t < NV * NH; t += NT) {       // physical Hessians: sum_r sum_c Href[r][c] JacI[a][c] JacI[b][r] (the map's own second derivatives left out, as there)
      const int n = t / NH, m = t % NH;
      int a, b;
      if (m < DIM) { a = m; b = m; }
      else if (DIM == 2) { a = 0; b = 1; }
      else { a = m == 3 ? 0 : m == 4 ? 1 : 2; b = m == 3 ? 1 : m == 4 ? 2 : 0; }
      double out = 0.0;
#pragma unroll
      for (int r = 0; r < DIM; r++) {
        double row = 0.0;
#pragma unroll
        for (int c = 0; c < DIM; c++) row += d2[n * NH + hidx(r, c)] * JI[a][c];
        out += row * JI[b][r];
      }
      N[t] = out;
    }
    __syncthreads();
    // solution at the Gauss point
    if (tid < DIM) {
      double s = 0.0;
      for (int n = 0; n < NV; n++) s += uv[tid * NV + n] * ph[n];
      sc[S_U + tid] = s;
    } else if (tid < DIM + DIM * DIM) {
      const int k = (tid - DIM) / DIM, j = (tid - DIM) % DIM;
      double s = 0.0;
      for (int n = 0; n < NV; n++) s += uv[k * NV + n] * G[n * DIM + j];
      sc[S_GU + k * DIM + j] = s;
    } else if (tid < DIM + DIM * DIM + DIM * NH) {
      const int q = tid - DIM - DIM * DIM, k = q / NH, m = q % NH;
      double s = 0.0;
      for (int n = 0; n < NV; n++) s += uv[k * NV + n] * N[n * NH + m];
      sc[S_HU + q] = s;
    } else if (tid < DIM + DIM * DIM + DIM * NH + DIM) {
      const int j = tid - (DIM + DIM * DIM + DIM * NH);
      double s = 0.0;
      for (int n = 0; n < NV; n++) s += uv[DIM * NV + n] * G[n * DIM + j];
      sc[S_GP + j] = s;
    } else if (tid == DIM + DIM * DIM + DIM * NH + DIM) {
      double s = 0.0;
      for (int n = 0; n < NV; n++) s += uv[DIM * NV + n] * ph[n];
      sc[S_P] = s;
      sc[S_W] = det * P.w[g];
    }
    __syncthreads();
    if (tid == 0) {               // strong residual, stabilisation parameters and their derivatives with respect to u
      const double ire = P.ire;
      double a2 = 0.0, div = 0.0;
      for (int k = 0; k < DIM; k++) { a2 += sc[S_U + k] * sc[S_U + k]; div += sc[S_GU + k * DIM + k]; }
      const double an = sqrt(a2);
      for (int k = 0; k < DIM; k++) {
        double r = -sc[S_GP + k];
        for (int j = 0; j < DIM; j++) r += -sc[S_U + j] * sc[S_GU + k * DIM + j] + ire * (sc[S_HU + k * NH + j] + sc[S_HU + j * NH + hidx(k, j)]);
        sc[S_RES + k] = r;
      }
      const double rek = an / (4.0 * sl * ire);
      double tau = 1.0 / (sl * sl * 4.0 * ire), del = 0.0;
      for (int k = 0; k < DIM; k++) { sc[S_DT + k] = 0.0; sc[S_DD + k] = 0.0; }
      if (rek > 1.0e-15) {
        if (rek >= 1.0) {
          tau = 1.0 / (an * sl);
          del = an / sl;
          for (int k = 0; k < DIM; k++) { sc[S_DT + k] = -sc[S_U + k] / (an * an * an * sl); sc[S_DD + k] = sc[S_U + k] / (an * sl); }
        } else {                 // xi = Rek: tau = Rek / (|u| sl) = 1 / (4 sl^2 IRe), delta = Rek |u| / sl = |u|^2 / (4 sl^2 IRe)
          tau = rek / (an * sl);
          del = (rek * an) / sl;
          for (int k = 0; k < DIM; k++) sc[S_DD + k] = 2.0 * sc[S_U + k] / (4.0 * sl * sl * ire);
        }
      }
      sc[S_DIV] = div; sc[S_TAU] = tau; sc[S_DEL] = del;
    }
    if (tid >= 32 && tid < 32 + NV) {      // per node: u . grad phi and the Laplacian of phi
      const int n = tid - 32;
      double b = 0.0, l = 0.0;
      for (int j = 0; j < DIM; j++) { b += sc[S_U + j] * G[n * DIM + j]; l += N[n * NH + j]; }
      B[n] = b;
      L[n] = l;
    }
    __syncthreads();
    const double ire = P.ire, W = sc[S_W], tau = sc[S_TAU], del = sc[S_DEL], div = sc[S_DIV];
    const double* u = sc + S_U;
    const double* gu = sc + S_GU;
    const double* rs = sc + S_RES;
    const double* dt = sc + S_DT;
    const double* dd = sc + S_DD;
    // least-squares test factor of (component k, node i): Res_k lap phi_i + sum_n Res_n d_n d_k phi_i
    auto lsq = [&](int k, int i, const double* r) -> double {
      double s = r[k] * L[i];
      for (int n = 0; n < DIM; n++) s += r[n] * N[i * NH + hidx(n, k)];
      return s;
    };
#pragma unroll
    for (int q = 0; q < EPT; q++) {
      if (er[q] < 0) continue;
      const int kr = er[q] / NV, i = er[q] % NV, kc = ec[q] / NV, j = ec[q] % NV;
      double J;
      double dres[DIM];            // d Res_n / d (this column's dof)
      if (kc < DIM) {
        for (int n = 0; n < DIM; n++) dres[n] = -ph[j] * gu[n * DIM + kc] + (n == kc ? -B[j] + ire * L[j] : 0.0) + ire * N[j * NH + hidx(n, kc)];
      } else {
        for (int n = 0; n < DIM; n++) dres[n] = -G[j * DIM + n];
      }
      if (kr < DIM) {
        const int k = kr;
        if (kc < DIM) {
          const int c = kc;
          double lap = 0.0;
          for (int n = 0; n < DIM; n++) lap += G[i * DIM + n] * G[j * DIM + n];
          J = -ph[i] * (ph[j] * gu[k * DIM + c] + (k == c ? B[j] : 0.0)) - ire * ((k == c ? lap : 0.0) + G[i * DIM + c] * G[j * DIM + k]) -
              (dd[c] * ph[j] * div + del * G[j * DIM + c]) * G[i * DIM + k] + dres[k] * tau * B[i] +
              rs[k] * (dt[c] * ph[j] * B[i] + tau * ph[j] * G[i * DIM + c]) - ire * dt[c] * ph[j] * lsq(k, i, rs) - ire * tau * lsq(k, i, dres);
        } else {
          J = ph[j] * G[i * DIM + k] + dres[k] * tau * B[i] - ire * tau * lsq(k, i, dres);
        }
      } else {
        double gr = 0.0, gd = 0.0;
        for (int n = 0; n < DIM; n++) { gr += G[i * DIM + n] * rs[n]; gd += G[i * DIM + n] * dres[n]; }
        if (kc < DIM) J = G[j * DIM + kc] * ph[i] - dt[kc] * ph[j] * gr - tau * gd;
        else J = -tau * gd;
      }
      acc[q] += -J * W;
    }
    if (tid < ND) {
      const int kr = tid / NV, i = tid % NV;
      double r;
      if (kr < DIM) {
        const int k = kr;
        double adv = 0.0, lp = 0.0;
        for (int n = 0; n < DIM; n++) { adv += u[n] * gu[k * DIM + n]; lp += G[i * DIM + n] * (gu[k * DIM + n] + gu[n * DIM + k]); }
        r = -ph[i] * adv - ire * lp + (sc[S_P] - del * div) * G[i * DIM + k] + rs[k] * tau * B[i] - ire * tau * lsq(k, i, rs);
      } else {
        double gr = 0.0;
        for (int n = 0; n < DIM; n++) gr += G[i * DIM + n] * rs[n];
        r = div * ph[i] - tau * gr;
      }
      racc += r * W;
    }
    __syncthreads();
  }
  double* Ke = P.K + (size_t)e * ND * ND;
#pragma unroll
  for (int q = 0; q < EPT; q++)
    if (er[q] >= 0) Ke[er[q] * ND + ec[q]] = acc[q];
  if (tid < ND) P.F[(size_t)e * ND + tid] = racc;
}

// pw: the pressure is the discontinuous piecewise-linear space (kind 2), n_vertex_nodes is not used then
static int ns_assembler_create_impl(fh_ctx_t ctx, int geom, int gauss_order, int nel, int nloc, const int* elem_dof, int nnode, int n_vertex_nodes,
                                    const double* coords, fh_mat_t A, bool pw, fh_ns_assembler_t* out, bool scalar = false) {
  FH_GUARD_BEGIN
  FH_REQUIRE(ctx && elem_dof && coords && A && out, "fh_ns_assembler_create: null argument");
  FH_REQUIRE(geom == 0 || geom == 1, "fh_ns_assembler_create: geom must be 0 (hex) or 1 (quad)");
  FH_REQUIRE(nloc == fhfe::nloc_of(geom), "fh_ns_assembler_create: nloc %d does not match the geometry", nloc);
  fh_ns_assembler_t as = new fh_ns_assembler_s();
  as->ctx = ctx;
  as->kind = scalar ? 3 : pw ? 2 : 0;
  as->geom = geom;
  as->dim = fhfe::dim_of(geom);
  as->nv = fhfe::ndofs_of(geom, fhfe::FE_BIQUADRATIC);
  as->np = scalar ? 0 : pw ? as->dim + 1 : fhfe::ndofs_of(geom, fhfe::FE_LINEAR);
  as->nd = scalar ? as->nv : as->dim * as->nv + as->np;
  as->nloc = nloc;
  as->nel = nel;
  as->nnode = nnode;
  as->nq1 = pw ? 0 : n_vertex_nodes;
  FH_REQUIRE((int64_t)as->dim * nnode + (pw ? (int64_t)as->np * nel : (int64_t)n_vertex_nodes) < 2147483647ll, "fh_ns_assembler_create: the system does not fit 32-bit ids");
  as->ndof = scalar ? nnode : as->dim * nnode + (pw ? as->np * nel : n_vertex_nodes);
  FH_REQUIRE(A->m == as->ndof && A->n == as->ndof, "fh_ns_assembler_create: matrix is %d x %d, the system has %d rows", A->m, A->n, as->ndof);
  std::vector<double> w, phi, dphi, w1, psi, dpsi;
  FH_REQUIRE(fhfe::shape_tables(geom, fhfe::FE_BIQUADRATIC, gauss_order, w, phi, dphi) == 0 &&
                 fhfe::shape_tables(geom, fhfe::FE_LINEAR, gauss_order, w1, psi, dpsi) == 0,
             "fh_ns_assembler_create: unsupported Gauss rule %d", gauss_order);
  as->ng = (int)w.size();
  if (pw) {      // GetPhi(ig) of the discontinuous family: 1 and the reference coordinates of the Gauss point
    std::vector<double> gw(as->ng), gx((size_t)as->ng * as->dim);
    FH_REQUIRE(fhfe::gauss_table(geom, gauss_order, gw.data(), gx.data()) == 0, "fh_ns_assembler_create: unsupported Gauss rule %d", gauss_order);
    psi.assign((size_t)as->ng * as->np, 1.0);
    for (int g = 0; g < as->ng; g++)
      for (int d = 0; d < as->dim; d++) psi[(size_t)g * as->np + 1 + d] = gx[(size_t)d * as->ng + g];      // gauss_table: one array per coordinate
  }
  const int nd = as->nd;
  std::vector<int> es((size_t)nel * nd);
  for (int e = 0; e < nel; e++) {
    const int* ed = elem_dof + (size_t)e * nloc;
    for (int i = 0; i < nloc; i++) FH_REQUIRE(ed[i] >= 0 && ed[i] < nnode, "fh_ns_assembler_create: node id %d out of range", ed[i]);
    for (int i = 0; i < as->np && !pw; i++) FH_REQUIRE(ed[i] < n_vertex_nodes, "fh_ns_assembler_create: vertex node %d is not a linear dof", ed[i]);
    int p = 0;
    for (int k = 0; k < (scalar ? 1 : as->dim); k++)
      for (int i = 0; i < as->nv; i++) es[(size_t)e * nd + p++] = k * nnode + ed[i];
    for (int i = 0; i < as->np; i++) es[(size_t)e * nd + p++] = as->dim * nnode + (pw ? i * nel + e : ed[i]);
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

extern "C" int fh_ns_assembler_create(fh_ctx_t ctx, int geom, int gauss_order, int nel, int nloc, const int* elem_dof, int nnode, int n_vertex_nodes,
                                      const double* coords, fh_mat_t A, fh_ns_assembler_t* out) {
  return ns_assembler_create_impl(ctx, geom, gauss_order, nel, nloc, elem_dof, nnode, n_vertex_nodes, coords, A, false, out);
}

extern "C" int fh_advdiff_assembler_create(fh_ctx_t ctx, int geom, int gauss_order, int nel, int nloc, const int* elem_dof, int nnode, const double* coords,
                                          fh_mat_t A, fh_ns_assembler_t* out) {
  return ns_assembler_create_impl(ctx, geom, gauss_order, nel, nloc, elem_dof, nnode, 0, coords, A, false, out, true);
}

extern "C" int fh_assemble_advection_diffusion(fh_ns_assembler_t as, fh_vec_t sol, fh_vec_t velocity, double inverse_peclet, fh_mat_t A, fh_vec_t res) {
  FH_REQUIRE(as && A && res, "fh_assemble_advection_diffusion: null argument");
  FH_REQUIRE(as->kind == 3, "fh_assemble_advection_diffusion: this assembler was not created by fh_advdiff_assembler_create");
  FH_REQUIRE(A->m == as->ndof && res->n_local >= as->ndof, "fh_assemble_advection_diffusion: size mismatch");
  FH_REQUIRE(!sol || sol->n_local >= as->nnode, "fh_assemble_advection_diffusion: the state has %d entries, the mesh %d nodes", sol ? sol->n_local : 0, as->nnode);
  FH_REQUIRE(!velocity || velocity->n_local >= as->dim * as->nnode, "fh_assemble_advection_diffusion: the velocity vector has %d entries, %d x %d are needed",
             velocity ? velocity->n_local : 0, as->dim, as->nnode);
  AdvDiffParams P;
  P.elem_dof = as->d_elem_dof; P.coords = as->d_coords; P.w = as->d_w; P.phi = as->d_phi; P.dphi = as->d_dphi;
  P.sol = sol ? sol->d : nullptr;
  P.vel = velocity ? velocity->d : nullptr;
  P.K = as->d_K; P.F = as->d_F;
  P.nel = as->nel; P.nloc = as->nloc; P.ng = as->ng; P.nnode = as->nnode;
  P.ipe = inverse_peclet;
  if (as->nel) {
    if (as->dim == 2) hipLaunchKernelGGL(k_advdiff_elem<2>, dim3(as->nel), dim3(64), 0, as->ctx->stream, P);
    else hipLaunchKernelGGL(k_advdiff_elem<3>, dim3(as->nel), dim3(256), 0, as->ctx->stream, P);
  }
  if (as->ndof > 0)
    hipLaunchKernelGGL(k_sys_row_gather, dim3(as->ndof), dim3(64), (size_t)as->max_row * sizeof(double), as->ctx->stream, as->d_adj_ptr, as->d_adj_ei,
                       as->d_elem_sys, as->d_K, as->d_F, as->nd, A->d_rowptr, A->d_col, A->d_val, res->d, as->ndof);
  FH_CHECK_HIP(hipGetLastError());
  A->at_valid = false;
  return 0;
}

extern "C" int fh_ns_pw_assembler_create(fh_ctx_t ctx, int geom, int gauss_order, int nel, int nloc, const int* elem_dof, int nnode, const double* coords,
                                         fh_mat_t A, fh_ns_assembler_t* out) {
  return ns_assembler_create_impl(ctx, geom, gauss_order, nel, nloc, elem_dof, nnode, 0, coords, A, true, out);
}

extern "C" int fh_ns_assembler_destroy(fh_ns_assembler_t as) {
  if (!as) return 0;
  hipStreamSynchronize(as->ctx->stream);
  for (void* q : {(void*)as->d_elem_dof, (void*)as->d_elem_sys, (void*)as->d_coords, (void*)as->d_w, (void*)as->d_phi, (void*)as->d_dphi,
                  (void*)as->d_psi, (void*)as->d_K, (void*)as->d_F, (void*)as->d_adj_ptr, (void*)as->d_adj_ei, (void*)as->d_d2phi})
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
  if (as->kind == 2) {
    if (as->dim == 2) hipLaunchKernelGGL((k_ns_elem<2, true>), dim3(as->nel), dim3(NsCfg<2, true>::NT), 0, as->ctx->stream, P);
    else hipLaunchKernelGGL((k_ns_elem<3, true>), dim3(as->nel), dim3(NsCfg<3, true>::NT), 0, as->ctx->stream, P);
  } else if (as->dim == 2) hipLaunchKernelGGL(k_ns_elem<2>, dim3(as->nel), dim3(NsCfg<2>::NT), 0, as->ctx->stream, P);
  else hipLaunchKernelGGL(k_ns_elem<3>, dim3(as->nel), dim3(NsCfg<3>::NT), 0, as->ctx->stream, P);
  FH_CHECK_HIP(hipGetLastError());
  return 0;
}

static int ns_stab_element_pass(fh_ns_assembler_t as, fh_vec_t sol, double ire) {
  FH_REQUIRE(!sol || sol->n_local >= as->ndof, "navier-stokes assembly: solution vector has %d entries, the system has %d", sol ? sol->n_local : 0, as->ndof);
  FH_REQUIRE(ire > 0.0, "navier-stokes assembly: the inverse Reynolds number must be positive");
  NsStabParams P;
  P.elem_dof = as->d_elem_dof;
  P.coords = as->d_coords;
  P.w = as->d_w;
  P.phi = as->d_phi;
  P.dphi = as->d_dphi;
  P.d2phi = as->d_d2phi;
  P.sol = sol ? sol->d : nullptr;
  P.K = as->d_K;
  P.F = as->d_F;
  P.nel = as->nel;
  P.nloc = as->nloc;
  P.ng = as->ng;
  P.nq1 = as->nq1;
  P.ire = ire;
  if (as->nel == 0) return 0;
  if (as->dim == 2) hipLaunchKernelGGL(k_ns_stab_elem<2>, dim3(as->nel), dim3(64), 0, as->ctx->stream, P);
  else hipLaunchKernelGGL(k_ns_stab_elem<3>, dim3(as->nel), dim3(64), 0, as->ctx->stream, P);
  FH_CHECK_HIP(hipGetLastError());
  return 0;
}

// equal-order linear velocity and pressure on the vertex nodes, variables stacked [U | V | (W) | P] (each n_vertex_nodes long)
extern "C" int fh_ns_stab_assembler_create(fh_ctx_t ctx, int geom, int gauss_order, int nel, int nloc, const int* elem_dof, int nnode, int n_vertex_nodes,
                                           const double* coords, fh_mat_t A, fh_ns_assembler_t* out) {
  FH_GUARD_BEGIN
  FH_REQUIRE(ctx && elem_dof && coords && A && out, "fh_ns_stab_assembler_create: null argument");
  FH_REQUIRE(geom == 0 || geom == 1, "fh_ns_stab_assembler_create: geom must be 0 (hex) or 1 (quad)");
  FH_REQUIRE(nloc >= fhfe::ndofs_of(geom, fhfe::FE_LINEAR), "fh_ns_stab_assembler_create: nloc %d is less than the vertices of the element", nloc);
  fh_ns_assembler_t as = new fh_ns_assembler_s();
  struct Guard {
    fh_ns_assembler_t p;
    ~Guard() { if (p) fh_ns_assembler_destroy(p); }
  } guard{as};
  as->ctx = ctx;
  as->kind = 1;
  as->geom = geom;
  as->dim = fhfe::dim_of(geom);
  as->nv = as->np = fhfe::ndofs_of(geom, fhfe::FE_LINEAR);
  as->nd = (as->dim + 1) * as->nv;
  as->nloc = nloc;
  as->nel = nel;
  as->nnode = nnode;
  as->nq1 = n_vertex_nodes;
  as->ndof = (as->dim + 1) * n_vertex_nodes;
  FH_REQUIRE(A->m == as->ndof && A->n == as->ndof, "fh_ns_stab_assembler_create: matrix is %d x %d, the system has %d rows", A->m, A->n, as->ndof);
  std::vector<double> w, phi, dphi;
  FH_REQUIRE(fhfe::shape_tables(geom, fhfe::FE_LINEAR, gauss_order, w, phi, dphi) == 0, "fh_ns_stab_assembler_create: unsupported Gauss rule %d", gauss_order);
  as->ng = (int)w.size();
  const int nd = as->nd, nv = as->nv, nh = as->dim == 2 ? 3 : 6;
  std::vector<double> d2t((size_t)nh * as->ng * nv), d2((size_t)as->ng * nv * nh);
  FH_TRY(fh_fe_tables_d2(geom, fhfe::FE_LINEAR, gauss_order, d2t.data()));       // one [ng][nv] table per second derivative -> [ng][nv][nh]
  for (int k = 0; k < nh; k++)
    for (int g = 0; g < as->ng; g++)
      for (int j = 0; j < nv; j++) d2[((size_t)g * nv + j) * nh + k] = d2t[((size_t)k * as->ng + g) * nv + j];
  std::vector<int> es((size_t)nel * nd);
  for (int e = 0; e < nel; e++) {
    const int* ed = elem_dof + (size_t)e * nloc;
    for (int i = 0; i < nv; i++) FH_REQUIRE(ed[i] >= 0 && ed[i] < n_vertex_nodes, "fh_ns_stab_assembler_create: vertex node %d is not a linear dof", ed[i]);
    int p = 0;
    for (int k = 0; k <= as->dim; k++)
      for (int i = 0; i < nv; i++) es[(size_t)e * nd + p++] = k * n_vertex_nodes + ed[i];
  }
  std::vector<int> aptr(as->ndof + 1, 0);
  for (size_t k = 0; k < es.size(); k++) aptr[es[k] + 1]++;
  for (int r = 0; r < as->ndof; r++) aptr[r + 1] += aptr[r];
  FH_REQUIRE((int64_t)nel * nd < 2147483647ll, "fh_ns_stab_assembler_create: too many element rows");
  std::vector<int> aei(aptr[as->ndof]), cur(aptr.begin(), aptr.end() - 1);
  for (int e = 0; e < nel; e++)
    for (int i = 0; i < nd; i++) aei[cur[es[(size_t)e * nd + i]]++] = e * nd + i;
  for (int e = 0; e < nel; e++)
    for (int i = 0; i < nd; i++) {
      const int r = es[(size_t)e * nd + i];
      for (int j = 0; j < nd; j++) {
        const int c = es[(size_t)e * nd + j];
        FH_REQUIRE(std::binary_search(fh_hcol(A).begin() + A->h_rowptr[r], fh_hcol(A).begin() + A->h_rowptr[r + 1], c),
                   "fh_ns_stab_assembler_create: entry (%d, %d) is not in the matrix pattern", r, c);
      }
    }
  as->max_row = A->max_row;
  if (as->max_row == 0)
    for (int r = 0; r < A->m; r++) as->max_row = std::max(as->max_row, A->h_rowptr[r + 1] - A->h_rowptr[r]);
  FH_REQUIRE((size_t)as->max_row * sizeof(double) <= 64 * 1024, "fh_ns_stab_assembler_create: rows of %d entries exceed the LDS accumulator", as->max_row);
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
  FH_TRY(up((void**)&as->d_d2phi, d2.data(), d2.size() * sizeof(double)));
  FH_TRY(up((void**)&as->d_adj_ptr, aptr.data(), aptr.size() * sizeof(int)));
  FH_TRY(up((void**)&as->d_adj_ei, aei.data(), aei.size() * sizeof(int)));
  FH_CHECK_HIP(hipMalloc(&as->d_K, std::max<size_t>((size_t)nel * nd * nd, 1) * sizeof(double)));
  FH_CHECK_HIP(hipMalloc(&as->d_F, std::max<size_t>((size_t)nel * nd, 1) * sizeof(double)));
  guard.p = nullptr;
  *out = as;
  return 0;
  FH_GUARD_END("fh_ns_stab_assembler_create")
}

extern "C" int fh_assemble_navier_stokes_stab(fh_ns_assembler_t as, fh_vec_t sol, double inverse_reynolds, fh_mat_t A, fh_vec_t res) {
  FH_REQUIRE(as && A && res, "fh_assemble_navier_stokes_stab: null argument");
  FH_REQUIRE(as->kind == 1, "fh_assemble_navier_stokes_stab: this assembler was created for the Taylor-Hood form (fh_assemble_navier_stokes)");
  FH_REQUIRE(A->m == as->ndof && res->n_local >= as->ndof, "fh_assemble_navier_stokes_stab: size mismatch");
  FH_TRY(ns_stab_element_pass(as, sol, inverse_reynolds));
  if (as->ndof > 0)
    hipLaunchKernelGGL(k_sys_row_gather, dim3(as->ndof), dim3(64), (size_t)as->max_row * sizeof(double), as->ctx->stream, as->d_adj_ptr, as->d_adj_ei,
                       as->d_elem_sys, as->d_K, as->d_F, as->nd, A->d_rowptr, A->d_col, A->d_val, res->d, as->ndof);
  FH_CHECK_HIP(hipGetLastError());
  A->at_valid = false;
  return 0;
}

extern "C" int fh_assemble_navier_stokes(fh_ns_assembler_t as, fh_vec_t sol, double nu, fh_mat_t A, fh_vec_t res) {
  FH_REQUIRE(as && A && res, "fh_assemble_navier_stokes: null argument");
  FH_REQUIRE(as->kind == 0 || as->kind == 2, "fh_assemble_navier_stokes: this assembler was created for the stabilised equal-order form (fh_assemble_navier_stokes_stab)");
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
  if (as->kind == 1) FH_TRY(ns_stab_element_pass(as, sol, nu));
  else FH_TRY(ns_element_pass(as, sol, nu));
  FH_CHECK_HIP(hipMemcpyAsync(K, as->d_K, (size_t)as->nel * as->nd * as->nd * sizeof(double), hipMemcpyDeviceToHost, as->ctx->stream));
  FH_CHECK_HIP(hipMemcpyAsync(F, as->d_F, (size_t)as->nel * as->nd * sizeof(double), hipMemcpyDeviceToHost, as->ctx->stream));
  FH_CHECK_HIP(hipStreamSynchronize(as->ctx->stream));
  return 0;
}
