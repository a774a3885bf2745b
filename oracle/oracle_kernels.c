/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Plain-C restatement of the three timed loops of the FEMuS hot path,
 * used (a) as a second checker beside oracle/femus_oracle.py and (b) as bench.py's `cpu_baseline` ("port").
 * Never linked into or called by the product (femus_amd/).
 *
 *   oc_assemble_poisson : element loop of src/08_equations/assemble/
 *                         00_poisson_eqn_with_all_dirichlet_bc_AD_or_nonAD_separate.hpp:111-215 with
 *                         elem_type::Jacobian (src/02_reference_geom_elements/03_fe_evaluations_at_quadrature/
 *                         ElemType.hpp:1183-1248, 1438-1537) and MatSetValues-style scatter (PetscMatrix.cpp:699-729:
 *                         per-row binary search, ADD_VALUES) in element order.
 *   oc_spmv             : MatMult / MatMultAdd / residual / one Richardson+Jacobi sweep (PetscVector.cpp:182-247,
 *                         03_solvers/LinearEquationSolverPetsc.cpp:516-519) on CSR, rows split over OpenMP threads
 *                         (the reference splits rows over MPI ranks).
 * Tables (phi, dphi, w) are passed in by the caller (oracle/femus_oracle.py builds them).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

void oc_set_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n);
#endif
}

int oc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

static void jacobian(int dim, int nc, const double* dphi_g /* [nc][dim] */, const double* x /* [nc][dim] */, double wg,
                     double* weight, double* grad /* [nc][dim] */) {
  double J[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, I[3][3], det;
  for (int n = 0; n < nc; n++)
    for (int a = 0; a < dim; a++)
      for (int b = 0; b < dim; b++) J[a][b] += dphi_g[n * dim + a] * x[n * dim + b];
  if (dim == 2) {
    det = J[0][0] * J[1][1] - J[0][1] * J[1][0];
    I[0][0] = J[1][1] / det;
    I[0][1] = -J[0][1] / det;
    I[1][0] = -J[1][0] / det;
    I[1][1] = J[0][0] / det;
  } else {
    det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) + J[0][1] * (J[1][2] * J[2][0] - J[1][0] * J[2][2]) +
          J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
    I[0][0] = (-J[1][2] * J[2][1] + J[1][1] * J[2][2]) / det;
    I[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) / det;
    I[0][2] = (-J[0][2] * J[1][1] + J[0][1] * J[1][2]) / det;
    I[1][0] = (J[1][2] * J[2][0] - J[1][0] * J[2][2]) / det;
    I[1][1] = (-J[0][2] * J[2][0] + J[0][0] * J[2][2]) / det;
    I[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) / det;
    I[2][0] = (-J[1][1] * J[2][0] + J[1][0] * J[2][1]) / det;
    I[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) / det;
    I[2][2] = (-J[0][1] * J[1][0] + J[0][0] * J[1][1]) / det;
  }
  *weight = det * wg;
  for (int n = 0; n < nc; n++)
    for (int a = 0; a < dim; a++) {
      double s = dphi_g[n * dim + 0] * I[a][0];
      for (int b = 1; b < dim; b++) s += dphi_g[n * dim + b] * I[a][b];
      grad[n * dim + a] = s;
    }
}

static double source(int kind, double p0, double p1, const double* xg, int dim) {
  if (kind == 0) return p0;
  if (kind == 3) { /* p0 * sum_d prod_{e != d} x_e (p1 - x_e) */
    double sum = 0.0;
    for (int d = 0; d < dim; d++) {
      double pr = 1.0;
      for (int e = 0; e < dim; e++)
        if (e != d) pr *= xg[e] * (p1 - xg[e]);
      sum += pr;
    }
    return p0 * sum;
  }
  double r = p0;
  for (int d = 0; d < dim; d++) r *= (kind == 1) ? sin(p1 * xg[d]) : cos(p1 * xg[d]);
  return r;
}

/* elements [e0, e1): element matrices/vectors; if rowptr != NULL they are added into the CSR (val) and res in
 * element order, otherwise written to Kout[e-e0][nc][nc], Fout[e-e0][nc].  Sequential (one rank of the reference). */
int oc_assemble_poisson(int dim, int nc, int ng, const double* w, const double* phi /* [ng][nc] */, const double* dphi /* [ng][nc][dim] */,
                        int e0, int e1, int nloc, const int* elem_dof, const double* coords /* [nnode][dim] */, const double* sol,
                        int source_kind, double p0, double p1, const int* rowptr, const int* col, double* val, double* res,
                        double* Kout, double* Fout) {
  double x[27 * 3], u[27], grad[27 * 3], K[27 * 27], F[27];
  int dofs[27];
  for (int e = e0; e < e1; e++) {
    for (int n = 0; n < nc; n++) {
      dofs[n] = elem_dof[(size_t)e * nloc + n];
      for (int d = 0; d < dim; d++) x[n * dim + d] = coords[(size_t)dofs[n] * dim + d];
      u[n] = sol ? sol[dofs[n]] : 0.0;
    }
    memset(K, 0, sizeof(double) * nc * nc);
    memset(F, 0, sizeof(double) * nc);
    for (int g = 0; g < ng; g++) {
      double weight;
      jacobian(dim, nc, dphi + (size_t)g * nc * dim, x, w[g], &weight, grad);
      double gu[3] = {0, 0, 0}, xg[3] = {0, 0, 0};
      for (int i = 0; i < nc; i++)
        for (int d = 0; d < dim; d++) {
          gu[d] += grad[i * dim + d] * u[i];
          xg[d] += x[i * dim + d] * phi[g * nc + i];
        }
      const double f = source(source_kind, p0, p1, xg, dim);
      for (int i = 0; i < nc; i++) {
        double wl = 0.0;
        for (int d = 0; d < dim; d++) wl += grad[i * dim + d] * gu[d];
        F[i] += (-f * phi[g * nc + i] - wl) * weight;
        for (int j = 0; j < nc; j++) {
          double wlj = 0.0;
          for (int d = 0; d < dim; d++) wlj += grad[i * dim + d] * grad[j * dim + d];
          K[i * nc + j] += wlj * weight;
        }
      }
    }
    if (rowptr) {
      for (int i = 0; i < nc; i++) {
        const int row = dofs[i];
        res[row] += F[i];
        for (int j = 0; j < nc; j++) {
          int lo = rowptr[row], hi = rowptr[row + 1] - 1, target = dofs[j], pos = -1;
          while (lo <= hi) {
            int mid = (lo + hi) >> 1;
            if (col[mid] == target) { pos = mid; break; }
            if (col[mid] < target) lo = mid + 1; else hi = mid - 1;
          }
          if (pos < 0) return 1;
          val[pos] += K[i * nc + j];
        }
      }
    } else {
      memcpy(Kout + (size_t)(e - e0) * nc * nc, K, sizeof(double) * nc * nc);
      memcpy(Fout + (size_t)(e - e0) * nc, F, sizeof(double) * nc);
    }
  }
  return 0;
}

/* The same loop over ALL elements [0, nel) on every core of the host (timed CPU baseline only): each thread takes a contiguous
 * range of elements -- the owner-computes split of the reference's MPI ranks (Mesh.cpp:589-616) -- and the adds into rows shared
 * with a neighbouring range are atomic, standing in for the stash exchange of MatAssemblyBegin/End.  The summation order is not
 * the sequential one, so parity tests keep using oc_assemble_poisson. */
int oc_assemble_poisson_omp(int dim, int nc, int ng, const double* w, const double* phi, const double* dphi, int nel, int nloc,
                            const int* elem_dof, const double* coords, const double* sol, int source_kind, double p0, double p1,
                            const int* rowptr, const int* col, double* val, double* res) {
  int bad = 0;
#pragma omp parallel
  {
    double x[27 * 3], u[27], grad[27 * 3], K[27 * 27], F[27];
    int dofs[27];
#pragma omp for schedule(static)
    for (int e = 0; e < nel; e++) {
      for (int n = 0; n < nc; n++) {
        dofs[n] = elem_dof[(size_t)e * nloc + n];
        for (int d = 0; d < dim; d++) x[n * dim + d] = coords[(size_t)dofs[n] * dim + d];
        u[n] = sol ? sol[dofs[n]] : 0.0;
      }
      memset(K, 0, sizeof(double) * nc * nc);
      memset(F, 0, sizeof(double) * nc);
      for (int g = 0; g < ng; g++) {
        double weight;
        jacobian(dim, nc, dphi + (size_t)g * nc * dim, x, w[g], &weight, grad);
        double gu[3] = {0, 0, 0}, xg[3] = {0, 0, 0};
        for (int i = 0; i < nc; i++)
          for (int d = 0; d < dim; d++) {
            gu[d] += grad[i * dim + d] * u[i];
            xg[d] += x[i * dim + d] * phi[g * nc + i];
          }
        const double f = source(source_kind, p0, p1, xg, dim);
        for (int i = 0; i < nc; i++) {
          double wl = 0.0;
          for (int d = 0; d < dim; d++) wl += grad[i * dim + d] * gu[d];
          F[i] += (-f * phi[g * nc + i] - wl) * weight;
          for (int j = 0; j < nc; j++) {
            double wlj = 0.0;
            for (int d = 0; d < dim; d++) wlj += grad[i * dim + d] * grad[j * dim + d];
            K[i * nc + j] += wlj * weight;
          }
        }
      }
      for (int i = 0; i < nc; i++) {
        const int row = dofs[i];
#pragma omp atomic
        res[row] += F[i];
        for (int j = 0; j < nc; j++) {
          int lo = rowptr[row], hi = rowptr[row + 1] - 1, target = dofs[j], pos = -1;
          while (lo <= hi) {
            int mid = (lo + hi) >> 1;
            if (col[mid] == target) { pos = mid; break; }
            if (col[mid] < target) lo = mid + 1; else hi = mid - 1;
          }
          if (pos < 0) {
#pragma omp atomic write
            bad = 1;
            continue;
          }
#pragma omp atomic
          val[pos] += K[i * nc + j];
        }
      }
    }
  }
  return bad;
}

/* mode 0: y = A x ; 1: y += A x ; 2: y = b - A x ; 3: y = x + omega*dinv*(b - A x) */
void oc_spmv(int m, const int* rowptr, const int* col, const double* val, const double* x, double* y, int mode, const double* b,
             const double* dinv, double omega) {
#pragma omp parallel for schedule(static)
  for (int r = 0; r < m; r++) {
    double s = 0.0;
    for (int k = rowptr[r]; k < rowptr[r + 1]; k++) s += val[k] * x[col[k]];
    if (mode == 0) y[r] = s;
    else if (mode == 1) y[r] += s;
    else if (mode == 2) y[r] = b[r] - s;
    else y[r] = x[r] + omega * dinv[r] * (b[r] - s);
  }
}

void oc_first_sweep(int n, double* x, const double* b, const double* dinv, double omega) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; i++) x[i] = omega * dinv[i] * b[i];
}
