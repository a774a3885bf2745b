// Galerkin triple product C = P^T A P (K6 of SURVEY 2.1, a15 of SURVEY 8) for gfx950.
// Replaces SparseMatrix::matrix_PtAP (src/03_algebra/01_matrices/PetscMatrix.cpp:733-751 -> MatPtAP) as used by the
// Galerkin chain of LinearImplicitSystem::MGsolve (LinearImplicitSystem.cpp:347-370).
//
// Split as PETSc does into symbolic and numeric:
//   symbolic (integer, once per pattern, host threads): patterns of AP = A*P and C = R*(AP) with R = P^T;
//   numeric  (FP, every assembly, device): two launches of one "owner-computes" SpGEMM kernel -- one wave per
//            output row, each lane owns output slots of the known sorted pattern and accumulates
//            sum_k A[i,k] * B[k,c] in increasing k with a binary search in the short sorted row B[k,:].
//            No atomics, no hash tables: deterministic, bit-reproducible operators.
#include "fh_internal.h"
#include <algorithm>
#include <thread>

struct PtapPlan {
  fh_mat_t AP = nullptr;   // m x nc work matrix (pattern + values)
  int m = 0, n = 0, nc = 0;
  int a_nnz = 0, p_nnz = 0;
};

static void destroy_plan(void* p) {
  PtapPlan* plan = (PtapPlan*)p;
  if (!plan) return;
  if (plan->AP) fh_mat_destroy(plan->AP);
  delete plan;
}

// C = A*B numeric on a given pattern of C
__global__ __launch_bounds__(256) void k_spgemm_numeric(const int* __restrict__ a_rp, const int* __restrict__ a_col, const double* __restrict__ a_val,
                                                        const int* __restrict__ b_rp, const int* __restrict__ b_col, const double* __restrict__ b_val,
                                                        const int* __restrict__ c_rp, const int* __restrict__ c_col, double* __restrict__ c_val,
                                                        int m) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= m) return;
  const int as = a_rp[row], ae = a_rp[row + 1];
  const int cs = c_rp[row], ce = c_rp[row + 1];
  for (int t = cs + lane; t < ce; t += 64) {
    const int c = c_col[t];
    double acc = 0.0;
    for (int ka = as; ka < ae; ka++) {
      const int k = a_col[ka];
      int lo = b_rp[k], hi = b_rp[k + 1] - 1;
      while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        const int cc = b_col[mid];
        if (cc == c) {
          acc += a_val[ka] * b_val[mid];
          break;
        }
        if (cc < c) lo = mid + 1; else hi = mid - 1;
      }
    }
    c_val[t] = acc;
  }
}

static int spgemm_numeric(fh_mat_t A, fh_mat_t B, fh_mat_t C) {
  if (C->m == 0) return 0;
  hipLaunchKernelGGL(k_spgemm_numeric, dim3(fh_div_up(C->m, 4)), dim3(256), 0, C->ctx->stream, A->d_rowptr, A->d_col, A->d_val, B->d_rowptr,
                     B->d_col, B->d_val, C->d_rowptr, C->d_col, C->d_val, C->m);
  FH_CHECK_HIP(hipGetLastError());
  C->at_valid = false;
  return 0;
}

// pattern of A*B on the host, rows split over threads (marker arrays, then sort per row)
static void spgemm_symbolic(int m, int ncols, const std::vector<int>& a_rp, const std::vector<int>& a_col, const std::vector<int>& b_rp,
                            const std::vector<int>& b_col, std::vector<int>& c_rp, std::vector<int>& c_col) {
  const int nthreads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  c_rp.assign(m + 1, 0);
  std::vector<std::vector<int>> chunks(nthreads);
  std::vector<int> bounds(nthreads + 1);
  for (int t = 0; t <= nthreads; t++) bounds[t] = (int)((int64_t)m * t / nthreads);
  auto work = [&](int t) {
    std::vector<int> marker(ncols, -1), buf;
    std::vector<int>& out = chunks[t];
    for (int i = bounds[t]; i < bounds[t + 1]; i++) {
      buf.clear();
      for (int ka = a_rp[i]; ka < a_rp[i + 1]; ka++) {
        const int k = a_col[ka];
        for (int kb = b_rp[k]; kb < b_rp[k + 1]; kb++) {
          const int j = b_col[kb];
          if (marker[j] != i) {
            marker[j] = i;
            buf.push_back(j);
          }
        }
      }
      std::sort(buf.begin(), buf.end());
      c_rp[i + 1] = (int)buf.size();
      out.insert(out.end(), buf.begin(), buf.end());
    }
  };
  std::vector<std::thread> th;
  for (int t = 0; t < nthreads; t++) th.emplace_back(work, t);
  for (auto& x : th) x.join();
  for (int i = 0; i < m; i++) c_rp[i + 1] += c_rp[i];
  c_col.resize(c_rp[m]);
  for (int t = 0; t < nthreads; t++)
    if (!chunks[t].empty()) std::copy(chunks[t].begin(), chunks[t].end(), c_col.begin() + c_rp[bounds[t]]);
}

int fh_mat_refresh_transpose(fh_mat_t A);

extern "C" int fh_mat_ptap(fh_mat_t P, fh_mat_t A, fh_mat_t* Cio) {
  FH_REQUIRE(P && A && Cio, "fh_mat_ptap: null argument");
  FH_REQUIRE(A->m == A->n && P->m == A->m, "fh_mat_ptap: shapes do not conform (A %dx%d, P %dx%d)", A->m, A->n, P->m, P->n);
  FH_TRY(fh_mat_refresh_transpose(P));     // R = P^T with current values
  fh_mat_t R = P->At;
  fh_mat_t C = *Cio;
  PtapPlan* plan = nullptr;
  if (!C) {
    plan = new PtapPlan();
    plan->m = A->m;
    plan->n = A->n;
    plan->nc = P->n;
    plan->a_nnz = A->nnz;
    plan->p_nnz = P->nnz;
    std::vector<int> ap_rp, ap_col, c_rp, c_col;
    spgemm_symbolic(A->m, P->n, A->h_rowptr, A->h_col, P->h_rowptr, P->h_col, ap_rp, ap_col);
    FH_TRY(fh_mat_create_csr(A->ctx, A->m, P->n, ap_rp.data(), ap_col.data(), nullptr, &plan->AP));
    spgemm_symbolic(R->m, P->n, R->h_rowptr, R->h_col, ap_rp, ap_col, c_rp, c_col);
    FH_TRY(fh_mat_create_csr(A->ctx, P->n, P->n, c_rp.data(), c_col.data(), nullptr, &C));
    C->plan = plan;
    C->plan_destroy = destroy_plan;
    *Cio = C;
  } else {
    plan = (PtapPlan*)C->plan;
    FH_REQUIRE(plan != nullptr, "fh_mat_ptap: the output matrix was not created by fh_mat_ptap (no reusable plan)");
    FH_REQUIRE(plan->m == A->m && plan->nc == P->n && plan->a_nnz == A->nnz && plan->p_nnz == P->nnz,
               "fh_mat_ptap: reuse with operands of a different pattern");
  }
  FH_TRY(spgemm_numeric(A, P, plan->AP));
  FH_TRY(spgemm_numeric(R, plan->AP, C));
  return 0;
}

extern "C" int fh_mat_matmul(fh_mat_t A, fh_mat_t B, fh_mat_t* Cout) {
  FH_REQUIRE(A && B && Cout, "fh_mat_matmul: null argument");
  FH_REQUIRE(A->n == B->m, "fh_mat_matmul: shapes do not conform (A %dx%d, B %dx%d)", A->m, A->n, B->m, B->n);
  std::vector<int> rp, col;
  spgemm_symbolic(A->m, B->n, A->h_rowptr, A->h_col, B->h_rowptr, B->h_col, rp, col);
  fh_mat_t C = nullptr;
  FH_TRY(fh_mat_create_csr(A->ctx, A->m, B->n, rp.data(), col.data(), nullptr, &C));
  FH_TRY(spgemm_numeric(A, B, C));
  *Cout = C;
  return 0;
}
