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

// product lists of C = A*B on fixed patterns: for every entry of C the elementary products that make it up, as pairs
// (offset of the A entry inside its row, index of the B entry), stored contiguously per C entry in increasing-k order.  Built
// once on the device (the only place a search is needed); the numeric product is then a segmented sum with no search, no
// atomics and a fixed summation order: 6 bytes per elementary product (6.5 GB for A*P plus 10 GB for R*(AP) at 64^3 Q2 -- HBM
// is 288 GB) streamed per re-assembly instead of 12x as many binary searches.
struct SlotMap {
  unsigned short* pa = nullptr;     // [nprod] offset of the A entry in its row
  int* pb = nullptr;                // [nprod] index of the B entry
  long long* rowbase = nullptr;     // [m+1] first elementary product of every row
  int* segptr = nullptr;            // [nnz(C)+m] per row: clen+1 offsets relative to rowbase[row]
  unsigned short* slot = nullptr;   // alternative for long B rows: per elementary product, in (row, ka, kb) order, its position in the C row
  long long nprod = 0;
  int max_crow = 0, max_arow = 0;
  void release() {
    for (void** q : {(void**)&pa, (void**)&pb, (void**)&rowbase, (void**)&segptr, (void**)&slot})
      if (*q) {
        hipFree(*q);
        *q = nullptr;
      }
  }
};

struct PtapPlan {
  fh_mat_t AP = nullptr;   // m x nc work matrix (pattern + values)
  int m = 0, n = 0, nc = 0;
  int a_nnz = 0, p_nnz = 0;
  SlotMap map_ap, map_c;
};

static void destroy_plan(void* p) {
  PtapPlan* plan = (PtapPlan*)p;
  if (!plan) return;
  if (plan->AP) fh_mat_destroy(plan->AP);
  plan->map_ap.release();
  plan->map_c.release();
  delete plan;
}

__device__ __forceinline__ void wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int row_slot(const int* __restrict__ c_col, int cs, int clen, int c) {
  int lo = 0, hi = clen - 1;
  while (lo < hi) {
    const int mid = lo + ((hi - lo) >> 1);   // (lo + hi) overflows beyond 2^30 non-zeros
    if (c_col[cs + mid] < c) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// pass 1: number of elementary products per row and per C entry (segptr holds counts, shifted by one, turned into offsets here).
// Lanes: groups of gl = 2^k lanes, one group per A entry, the lanes of a group over that entry's B row -- B rows of a prolongator hold 1 to
// 27 entries (8 on average), a whole wave per A entry left 7 lanes of 8 idle in the column searches, which are the cost of this pass
// (round 5: 6.5 -> see profiles/r05_amr_probe.json for the adaptive hierarchy's first preparation).
__global__ __launch_bounds__(256) void k_spgemm_segcount(const int* __restrict__ a_rp, const int* __restrict__ a_col, const int* __restrict__ b_rp,
                                                         const int* __restrict__ b_col, const int* __restrict__ c_rp, const int* __restrict__ c_col,
                                                         long long* __restrict__ rowcount, int* __restrict__ segptr, int m, int max_crow, int gl_log2) {
  extern __shared__ int cnts[];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + w;
  if (row >= m) return;
  int* cnt = cnts + (size_t)w * (max_crow + 1);
  const int cs = c_rp[row], clen = c_rp[row + 1] - cs;
  for (int t = lane; t <= clen; t += 64) cnt[t] = 0;
  wave_sync_lds();
  const int gl = 1 << gl_log2, ng = 64 >> gl_log2, g = lane >> gl_log2, sub = lane & (gl - 1);
  const int ae = a_rp[row + 1];
  for (int ka = a_rp[row] + g; ka < ae; ka += ng) {
    const int k = a_col[ka];
    const int bs = b_rp[k], blen = b_rp[k + 1] - bs;
    for (int t = sub; t < blen; t += gl) atomicAdd(&cnt[row_slot(c_col, cs, clen, b_col[bs + t]) + 1], 1);
  }
  wave_sync_lds();
  if (lane == 0) {
    int run = 0;
    for (int t = 1; t <= clen; t++) {
      run += cnt[t];
      cnt[t] = run;
    }
    rowcount[row + 1] = run;
  }
  wave_sync_lds();
  int* sp = segptr + cs + row;
  for (int t = lane; t <= clen; t += 64) sp[t] = cnt[t];
}

// pass 2: fill the lists.  The searches of ng A entries run side by side (the same lane groups as pass 1); the list positions are then handed
// out group by group, so the order inside every list is fixed by the patterns alone (A entries in chunks of ng; inside a chunk by round of
// gl B entries, then by A entry): the numeric product sums in a fixed order -- deterministic, bit-reproducible operators.
__global__ __launch_bounds__(256) void k_spgemm_segfill(const int* __restrict__ a_rp, const int* __restrict__ a_col, const int* __restrict__ b_rp,
                                                        const int* __restrict__ b_col, const int* __restrict__ c_rp, const int* __restrict__ c_col,
                                                        const long long* __restrict__ rowbase, const int* __restrict__ segptr,
                                                        unsigned short* __restrict__ pa, int* __restrict__ pb, int m, int max_crow, int gl_log2) {
  extern __shared__ int cnts[];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + w;
  if (row >= m) return;
  int* cur = cnts + (size_t)w * (max_crow + 1);
  const int cs = c_rp[row], clen = c_rp[row + 1] - cs;
  const int* sp = segptr + cs + row;
  for (int t = lane; t < clen; t += 64) cur[t] = sp[t];
  wave_sync_lds();
  const long long base = rowbase[row];
  const int as = a_rp[row], ae = a_rp[row + 1];
  const int gl = 1 << gl_log2, ng = 64 >> gl_log2, g = lane >> gl_log2, sub = lane & (gl - 1);
  for (int ka0 = as; ka0 < ae; ka0 += ng) {
    const int ka = ka0 + g;
    int bs = 0, blen = 0;
    if (ka < ae) {
      const int k = a_col[ka];
      bs = b_rp[k];
      blen = b_rp[k + 1] - bs;
    }
    for (int t0 = 0; __any(t0 < blen); t0 += gl) {
      const int t = t0 + sub;
      const bool has = t < blen;
      const int s = has ? row_slot(c_col, cs, clen, b_col[bs + t]) : 0;
      for (int gg = 0; gg < ng; gg++) {          // distinct slots inside one B row: the lanes of a group never meet
        if (has && g == gg) {
          const int q = cur[s];
          cur[s] = q + 1;
          pa[base + q] = (unsigned short)(ka - as);
          pb[base + q] = bs + t;
        }
        wave_sync_lds();
      }
    }
  }
}

#ifndef SPGEMM_LANES_PER_ENTRY
#define SPGEMM_LANES_PER_ENTRY 4   // measured on the fine-level A*P: 1 lane 5.27 ms, 2: 4.64, 4: 4.29, 8: 4.98
#endif
// numeric product: one wave per output row; groups of lanes own output entries and sum their lists in a fixed order
__global__ __launch_bounds__(256) void k_spgemm_numeric_map(const int* __restrict__ a_rp, const double* __restrict__ a_val, const double* __restrict__ b_val,
                                                            const int* __restrict__ c_rp, double* __restrict__ c_val,
                                                            const long long* __restrict__ rowbase, const int* __restrict__ segptr,
                                                            const unsigned short* __restrict__ pa, const int* __restrict__ pb, int m, int max_arow) {
  extern __shared__ double arow[];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + w;
  if (row >= m) return;
  double* av = arow + (size_t)w * max_arow;
  const int as = a_rp[row], alen = a_rp[row + 1] - as;
  for (int t = lane; t < alen; t += 64) av[t] = a_val[as + t];
  wave_sync_lds();
  const int cs = c_rp[row], clen = c_rp[row + 1] - cs;
  const int* sp = segptr + cs + row;
  const long long base = rowbase[row];
  // G lanes per output entry: they read the entry's product list interleaved (consecutive positions, G x fewer cache lines per
  // load instruction than one lane per entry) and combine with shuffles -- a fixed order, so the result is deterministic
  constexpr int G = SPGEMM_LANES_PER_ENTRY;
  const int sub = lane & (G - 1), ent = lane / G;
  for (int t0 = 0; t0 < clen; t0 += 64 / G) {
    const int t = t0 + ent;
    double acc = 0.0;
    if (t < clen) {
      const long long q1 = base + sp[t + 1];
      for (long long q = base + sp[t] + sub; q < q1; q += G) acc += av[pa[q]] * b_val[pb[q]];
    }
#pragma unroll
    for (int off = 1; off < G; off <<= 1) acc += __shfl_xor(acc, off, 64);
    if (t < clen && sub == 0) c_val[cs + t] = acc;
  }
}

// ---- variant for products whose B rows are long (R * (AP): ~100 entries per row): the lanes spread over the B row, the A
// entries are taken in sequence, the output row is accumulated in LDS; the map gives every elementary product its position in
// the output row (2 bytes each)
__global__ __launch_bounds__(256) void k_spgemm_count(const int* __restrict__ a_rp, const int* __restrict__ a_col, const int* __restrict__ b_rp,
                                                      long long* __restrict__ rowcount, int m) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= m) return;
  long long c = 0;
  for (int ka = a_rp[row] + lane; ka < a_rp[row + 1]; ka += 64) c += b_rp[a_col[ka] + 1] - b_rp[a_col[ka]];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
  if (lane == 0) rowcount[row + 1] = c;
}

__global__ __launch_bounds__(256) void k_spgemm_fill(const int* __restrict__ a_rp, const int* __restrict__ a_col, const int* __restrict__ b_rp,
                                                     const int* __restrict__ b_col, const int* __restrict__ c_rp, const int* __restrict__ c_col,
                                                     const long long* __restrict__ rowbase, unsigned short* __restrict__ slot, int m) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= m) return;
  const int cs = c_rp[row], clen = c_rp[row + 1] - cs;
  long long off = rowbase[row];
  for (int ka = a_rp[row]; ka < a_rp[row + 1]; ka++) {
    const int k = a_col[ka];
    const int bs = b_rp[k], blen = b_rp[k + 1] - bs;
    for (int t = lane; t < blen; t += 64) slot[off + t] = (unsigned short)row_slot(c_col, cs, clen, b_col[bs + t]);
    off += blen;
  }
}

__global__ __launch_bounds__(256) void k_spgemm_numeric_slots(const int* __restrict__ a_rp, const int* __restrict__ a_col, const double* __restrict__ a_val,
                                                              const int* __restrict__ b_rp, const double* __restrict__ b_val,
                                                              const int* __restrict__ c_rp, double* __restrict__ c_val,
                                                              const long long* __restrict__ rowbase, const unsigned short* __restrict__ slot, int m,
                                                              int max_crow) {
  extern __shared__ double arow[];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + w;
  if (row >= m) return;
  double* acc = arow + (size_t)w * max_crow;
  const int cs = c_rp[row], clen = c_rp[row + 1] - cs;
  for (int t = lane; t < clen; t += 64) acc[t] = 0.0;
  wave_sync_lds();
  long long off = rowbase[row];
  for (int ka = a_rp[row]; ka < a_rp[row + 1]; ka++) {
    const int k = a_col[ka];
    const double a = a_val[ka];
    const int bs = b_rp[k], blen = b_rp[k + 1] - bs;
    for (int t = lane; t < blen; t += 64) acc[slot[off + t]] += a * b_val[bs + t];
    off += blen;
    wave_sync_lds();
  }
  for (int t = lane; t < clen; t += 64) c_val[cs + t] = acc[t];
}

static int build_slot_map_long(fh_mat_t A, fh_mat_t B, fh_mat_t C, SlotMap& M) {
  fh_ctx_t c = A->ctx;
  const int m = C->m;
  int max_crow = 0;
  for (int r = 0; r < m; r++) max_crow = std::max(max_crow, C->h_rowptr[r + 1] - C->h_rowptr[r]);
  if (m == 0 || max_crow == 0 || max_crow > 2000) return 0;
  M.max_crow = max_crow;
  FH_CHECK_HIP(hipMalloc(&M.rowbase, ((size_t)m + 1) * sizeof(long long)));
  FH_CHECK_HIP(hipMemsetAsync(M.rowbase, 0, sizeof(long long), c->stream));
  hipLaunchKernelGGL(k_spgemm_count, dim3(fh_div_up(m, 4)), dim3(256), 0, c->stream, A->d_rowptr, A->d_col, B->d_rowptr, M.rowbase, m);
  std::vector<long long> rb((size_t)m + 1);
  FH_CHECK_HIP(hipMemcpyAsync(rb.data(), M.rowbase, rb.size() * sizeof(long long), hipMemcpyDeviceToHost, c->stream));
  FH_CHECK_HIP(hipStreamSynchronize(c->stream));
  FH_TRACE("slot map (long rows): products counted");
  for (int r = 0; r < m; r++) rb[r + 1] += rb[r];
  M.nprod = rb[m];
  size_t free_b = 0, total_b = 0;
  hipMemGetInfo(&free_b, &total_b);
  if (M.nprod == 0 || (size_t)M.nprod * 2 > free_b / 2) {
    M.release();
    return 0;
  }
  FH_CHECK_HIP(hipMemcpyAsync(M.rowbase, rb.data(), rb.size() * sizeof(long long), hipMemcpyHostToDevice, c->stream));
  FH_CHECK_HIP(hipMalloc(&M.slot, (size_t)M.nprod * sizeof(unsigned short)));
  hipLaunchKernelGGL(k_spgemm_fill, dim3(fh_div_up(m, 4)), dim3(256), 0, c->stream, A->d_rowptr, A->d_col, B->d_rowptr, B->d_col, C->d_rowptr,
                     C->d_col, M.rowbase, M.slot, m);
  FH_CHECK_HIP(hipGetLastError());
  FH_CHECK_HIP(hipStreamSynchronize(c->stream));
  FH_TRACE("slot map (long rows): slots filled (%lld products)", (long long)M.nprod);
  return 0;
}

static int build_slot_map(fh_mat_t A, fh_mat_t B, fh_mat_t C, SlotMap& M) {
  fh_ctx_t c = A->ctx;
  M.release();
  if (B->m > 0 && (double)B->nnz / B->m >= 32.0) return build_slot_map_long(A, B, C, M);   // long B rows: lanes over the B row
  const int m = C->m;
  int max_crow = 0, max_arow = 0;
  for (int r = 0; r < m; r++) {
    max_crow = std::max(max_crow, C->h_rowptr[r + 1] - C->h_rowptr[r]);
    max_arow = std::max(max_arow, A->h_rowptr[r + 1] - A->h_rowptr[r]);
  }
  // 4 waves x (max_crow + 1) ints / 4 waves x max_arow doubles must fit in 64 KB of LDS; A-row offsets are 16-bit
  if (m == 0 || max_crow == 0 || max_crow > 4000 || max_arow > 2000) return 0;
  M.max_crow = max_crow;
  M.max_arow = std::max(max_arow, 1);
  FH_CHECK_HIP(hipMalloc(&M.rowbase, ((size_t)m + 1) * sizeof(long long)));
  FH_CHECK_HIP(hipMalloc(&M.segptr, ((size_t)C->nnz + m + 2) * sizeof(int)));
  FH_CHECK_HIP(hipMemsetAsync(M.rowbase, 0, sizeof(long long), c->stream));
  const size_t lds = (size_t)4 * (max_crow + 1) * sizeof(int);
  // lanes per A entry: the power of two next to the mean length of a B row, 8 to 64
  int gl_log2 = 3;
  while (gl_log2 < 6 && (double)(1 << gl_log2) < (double)B->nnz / std::max(B->m, 1)) gl_log2++;
  hipLaunchKernelGGL(k_spgemm_segcount, dim3(fh_div_up(m, 4)), dim3(256), lds, c->stream, A->d_rowptr, A->d_col, B->d_rowptr, B->d_col, C->d_rowptr,
                     C->d_col, M.rowbase, M.segptr, m, max_crow, gl_log2);
  std::vector<long long> rb((size_t)m + 1);
  FH_CHECK_HIP(hipMemcpyAsync(rb.data(), M.rowbase, rb.size() * sizeof(long long), hipMemcpyDeviceToHost, c->stream));
  FH_CHECK_HIP(hipStreamSynchronize(c->stream));
  FH_TRACE("slot map: products counted");
  for (int r = 0; r < m; r++) {
    if (rb[r + 1] > 2000000000ll) {   // segment offsets inside a row are 32-bit
      M.release();
      return 0;
    }
    rb[r + 1] += rb[r];
  }
  M.nprod = rb[m];
  size_t free_b = 0, total_b = 0;
  hipMemGetInfo(&free_b, &total_b);
  if (M.nprod == 0 || (size_t)M.nprod * 6 > free_b / 2) {   // keep the searching kernel when the lists would not fit comfortably
    M.release();
    return 0;
  }
  FH_CHECK_HIP(hipMemcpyAsync(M.rowbase, rb.data(), rb.size() * sizeof(long long), hipMemcpyHostToDevice, c->stream));
  FH_TRACE("slot map: scanned");
  FH_CHECK_HIP(hipMalloc(&M.pa, (size_t)M.nprod * sizeof(unsigned short)));
  FH_CHECK_HIP(hipMalloc(&M.pb, (size_t)M.nprod * sizeof(int)));
  hipLaunchKernelGGL(k_spgemm_segfill, dim3(fh_div_up(m, 4)), dim3(256), lds, c->stream, A->d_rowptr, A->d_col, B->d_rowptr, B->d_col, C->d_rowptr,
                     C->d_col, M.rowbase, M.segptr, M.pa, M.pb, m, max_crow, gl_log2);
  FH_CHECK_HIP(hipGetLastError());
  FH_CHECK_HIP(hipStreamSynchronize(c->stream));
  FH_TRACE("slot map: lists filled (%lld products)", (long long)M.nprod);
  return 0;
}

static int spgemm_numeric_map(fh_mat_t A, fh_mat_t B, fh_mat_t C, const SlotMap& M) {
  if (M.slot) {
    hipLaunchKernelGGL(k_spgemm_numeric_slots, dim3(fh_div_up(C->m, 4)), dim3(256), (size_t)4 * M.max_crow * sizeof(double), C->ctx->stream,
                       A->d_rowptr, A->d_col, A->d_val, B->d_rowptr, B->d_val, C->d_rowptr, C->d_val, M.rowbase, M.slot, C->m, M.max_crow);
    FH_CHECK_HIP(hipGetLastError());
    C->at_valid = false;
    return 0;
  }
  hipLaunchKernelGGL(k_spgemm_numeric_map, dim3(fh_div_up(C->m, 4)), dim3(256), (size_t)4 * M.max_arow * sizeof(double), C->ctx->stream, A->d_rowptr,
                     A->d_val, B->d_val, C->d_rowptr, C->d_val, M.rowbase, M.segptr, M.pa, M.pb, C->m, M.max_arow);
  FH_CHECK_HIP(hipGetLastError());
  C->at_valid = false;
  return 0;
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
        const int mid = lo + ((hi - lo) >> 1);   // (lo + hi) overflows beyond 2^30 non-zeros
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

// ------------------------------------------------------------------------------------------------------------------
// Pattern of A*B on the DEVICE (round 4): one wave per row.  The candidate columns (the B rows named by the A row) go through a hash set
// in LDS (SY_TAB slots, linear probing, ds atomic compare-and-swap), which leaves the distinct columns; they are compacted, sorted (bitonic,
// the next power of two above their count) and written.  First launch: row lengths; the host scans them; second launch: columns.  A row with
// more than SY_CAP distinct columns raises a flag and the host builder (below) serves the product.  Short B rows (prolongators) are walked
// one per lane, long ones (operators) with the lanes across the row.
// ------------------------------------------------------------------------------------------------------------------
constexpr int SY_TAB = 2048, SY_CAP = 1024, SY_EMPTY = 0x7fffffff;
template <bool FILL>
__global__ __launch_bounds__(256) void k_spgemm_symbolic(int m, const int* __restrict__ a_rp, const int* __restrict__ a_col, const int* __restrict__ b_rp,
                                                         const int* __restrict__ b_col, int lanes_over_b, const int* __restrict__ c_rp,
                                                         int* __restrict__ rowlen, int* __restrict__ c_col, int* __restrict__ err) {
  __shared__ int tab[4][SY_TAB];
  __shared__ int cmp[FILL ? 4 : 1][FILL ? SY_CAP : 1];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + wave;
  if (row >= m) return;
  int* T = tab[wave];
  for (int k = lane; k < SY_TAB; k += 64) T[k] = SY_EMPTY;
  wave_sync_lds();
  int mine = 0;
  bool fail = false;
  auto insert = [&](int j) {
    unsigned h = ((unsigned)j * 2654435761u) >> 21;       // 11 bits
    for (int p = 0; p < SY_TAB; p++) {
      const int old = atomicCAS(&T[h], SY_EMPTY, j);
      if (old == SY_EMPTY) {
        mine++;
        return;
      }
      if (old == j) return;
      h = (h + 1) & (SY_TAB - 1);
    }
    fail = true;
  };
  const int as = a_rp[row], ae = a_rp[row + 1];
  if (lanes_over_b) {
    for (int ka = as; ka < ae; ka++) {
      const int k = a_col[ka];
      const int be = b_rp[k + 1];
      for (int kb = b_rp[k] + lane; kb < be && !fail; kb += 64) insert(b_col[kb]);
    }
  } else {
    for (int ka = as + lane; ka < ae; ka += 64) {
      const int k = a_col[ka];
      const int be = b_rp[k + 1];
      for (int kb = b_rp[k]; kb < be && !fail; kb++) insert(b_col[kb]);
    }
  }
  wave_sync_lds();
  int incl = mine;
  for (int d = 1; d < 64; d <<= 1) {
    const int v = __shfl_up(incl, d, 64);
    if (lane >= d) incl += v;
  }
  const int total = __shfl(incl, 63, 64);
  if (__ballot(fail) != 0ull || total > SY_CAP) {
    if (lane == 0) atomicExch(err, 2);
    return;
  }
  if (!FILL) {
    if (lane == 0) rowlen[row] = total;
    return;
  }
  int* K = cmp[FILL ? wave : 0];
  // compaction: each lane owns SY_TAB/64 consecutive slots
  constexpr int PER = SY_TAB / 64;
  int cnt = 0;
  for (int k = 0; k < PER; k++) cnt += T[lane * PER + k] != SY_EMPTY ? 1 : 0;
  int ci = cnt;
  for (int d = 1; d < 64; d <<= 1) {
    const int v = __shfl_up(ci, d, 64);
    if (lane >= d) ci += v;
  }
  int o = ci - cnt;
  for (int k = 0; k < PER; k++) {
    const int v = T[lane * PER + k];
    if (v != SY_EMPTY) K[o++] = v;
  }
  int np = 64;
  while (np < total) np <<= 1;
  for (int k = total + lane; k < np; k += 64) K[k] = SY_EMPTY;
  wave_sync_lds();
  for (int size = 2; size <= np; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = lane; t < (np >> 1); t += 64) {
        const int lo = ((t / stride) * (stride << 1)) + (t % stride), hi = lo + stride;
        const bool up = ((lo & size) == 0);
        const int a = K[lo], c = K[hi];
        if ((a > c) == up) {
          K[lo] = c;
          K[hi] = a;
        }
      }
      wave_sync_lds();
    }
  const int cs = c_rp[row];
  for (int k = lane; k < total; k += 64) c_col[cs + k] = K[k];
}

// C = pattern(A*B) as a matrix with zero values; returns 1 (and *Cout == nullptr) when a row exceeds the device limits
static int spgemm_symbolic_device(fh_mat_t A, fh_mat_t B, fh_mat_t* Cout) {
  fh_ctx_t c = A->ctx;
  const int m = A->m;
  *Cout = nullptr;
  if (m == 0 || A->nnz == 0 || B->nnz == 0) return 1;
  const int lanes_over_b = (double)B->nnz / std::max(B->m, 1) >= 24.0 ? 1 : 0;
  int *d_len = nullptr, *d_err = nullptr;
  FH_CHECK_HIP(hipMalloc(&d_len, ((size_t)m + 1) * sizeof(int)));
  FH_CHECK_HIP(hipMalloc(&d_err, sizeof(int)));
  FH_CHECK_HIP(hipMemsetAsync(d_err, 0, sizeof(int), c->stream));
  hipLaunchKernelGGL(k_spgemm_symbolic<false>, dim3(fh_div_up(m, 4)), dim3(256), 0, c->stream, m, A->d_rowptr, A->d_col, B->d_rowptr, B->d_col, lanes_over_b,
                     (const int*)nullptr, d_len, (int*)nullptr, d_err);
  std::vector<int> rp((size_t)m + 1, 0);
  int err = 0;
  FH_CHECK_HIP(hipMemcpyAsync(rp.data() + 1, d_len, (size_t)m * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  FH_CHECK_HIP(hipMemcpyAsync(&err, d_err, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  FH_CHECK_HIP(hipStreamSynchronize(c->stream));
  FH_TRACE("spgemm symbolic: row lengths counted (%d rows)", m);
  int64_t tot = 0;
  for (int r = 0; r < m && !err; r++) {
    tot += rp[r + 1];
    if (tot >= 2147483647ll) err = 3;
    rp[r + 1] = (int)tot;
  }
  if (err) {
    hipFree(d_len);
    hipFree(d_err);
    return 1;
  }
  fh_mat_t C = nullptr;
  if (fh_mat_alloc_device_pattern(c, m, B->n, std::move(rp), &C)) {
    hipFree(d_len);
    hipFree(d_err);
    fh_mat_destroy(C);
    return 2;
  }
  FH_TRACE("spgemm symbolic: scanned, pattern allocated");
  hipLaunchKernelGGL(k_spgemm_symbolic<true>, dim3(fh_div_up(m, 4)), dim3(256), 0, c->stream, m, A->d_rowptr, A->d_col, B->d_rowptr, B->d_col, lanes_over_b,
                     C->d_rowptr, (int*)nullptr, C->d_col, d_err);
  FH_CHECK_HIP(hipGetLastError());
  FH_CHECK_HIP(hipStreamSynchronize(c->stream));
  FH_TRACE("spgemm symbolic: columns filled");
  hipFree(d_len);
  hipFree(d_err);
  FH_TRY(fh_mat_build_rowblocks(C, c->spmv_tile));
  FH_TRACE("spgemm symbolic: row blocks");
  *Cout = C;
  return 0;
}

// pattern of A*B (zero values): on the device, or on the host when a row is beyond the device kernel's limits
static int spgemm_pattern(fh_mat_t A, fh_mat_t B, fh_mat_t* Cout) {
  if (A->ctx->spgemm_device_symbolic) {
    const int rc = spgemm_symbolic_device(A, B, Cout);
    if (rc == 0) return 0;
    if (rc != 1) return rc;
  }
  std::vector<int> rp, col;
  spgemm_symbolic(A->m, B->n, A->h_rowptr, fh_hcol(A), B->h_rowptr, fh_hcol(B), rp, col);
  return fh_mat_create_csr(A->ctx, A->m, B->n, rp.data(), col.data(), nullptr, Cout);
}

int fh_mat_refresh_transpose(fh_mat_t A);

// D = R * A * P with a reusable plan attached to D: the product A*P and the two slot maps are kept, so that a repeated call with
// operands of the same pattern is numeric only
static int triple_product(fh_mat_t R, fh_mat_t A, fh_mat_t P, fh_mat_t* Cio, const char* who) {
  fh_mat_t C = *Cio;
  PtapPlan* plan = nullptr;
  if (!C) {
    plan = new PtapPlan();
    plan->m = A->m;
    plan->n = A->n;
    plan->nc = P->n;
    plan->a_nnz = A->nnz;
    plan->p_nnz = P->nnz;
    FH_TRACE("%s: first product of %d x %d (%d non-zeros) with %d x %d", who, A->m, A->n, A->nnz, P->m, P->n);
    if (int rc = spgemm_pattern(A, P, &plan->AP)) {
      delete plan;
      return rc;
    }
    FH_TRACE("%s: pattern of A P (%d non-zeros)", who, plan->AP->nnz);
    if (int rc = spgemm_pattern(R, plan->AP, &C)) {
      destroy_plan(plan);
      return rc;
    }
    FH_TRACE("%s: pattern of R A P (%d non-zeros)", who, C->nnz);
    C->plan = plan;
    C->plan_destroy = destroy_plan;
    *Cio = C;
    if (A->ctx->spgemm_slot_map) {
      FH_TRY(build_slot_map(A, P, plan->AP, plan->map_ap));
      FH_TRY(build_slot_map(R, plan->AP, C, plan->map_c));
      FH_TRACE("%s: slot maps (%lld + %lld products)", who, (long long)plan->map_ap.nprod, (long long)plan->map_c.nprod);
    }
  } else {
    plan = (PtapPlan*)C->plan;
    FH_REQUIRE(plan != nullptr, "%s: the output matrix was not created by this product (no reusable plan)", who);
    FH_REQUIRE(plan->m == A->m && plan->n == A->n && plan->nc == P->n && plan->a_nnz == A->nnz && plan->p_nnz == P->nnz && C->m == R->m,
               "%s: reuse with operands of a different pattern", who);
  }
  if (plan->map_ap.pa || plan->map_ap.slot) FH_TRY(spgemm_numeric_map(A, P, plan->AP, plan->map_ap)); else FH_TRY(spgemm_numeric(A, P, plan->AP));
  if (plan->map_c.pa || plan->map_c.slot) FH_TRY(spgemm_numeric_map(R, plan->AP, C, plan->map_c)); else FH_TRY(spgemm_numeric(R, plan->AP, C));
  C->val_gen++;
  return 0;
}

extern "C" int fh_mat_ptap(fh_mat_t P, fh_mat_t A, fh_mat_t* Cio) {
  FH_GUARD_BEGIN
  FH_REQUIRE(P && A && Cio, "fh_mat_ptap: null argument");
  FH_REQUIRE(A->m == A->n && P->m == A->m, "fh_mat_ptap: shapes do not conform (A %dx%d, P %dx%d)", A->m, A->n, P->m, P->n);
  FH_TRY(fh_mat_refresh_transpose(P));     // R = P^T with current values
  return triple_product(P->At, A, P, Cio, "fh_mat_ptap");
  FH_GUARD_END("fh_mat_ptap")
}

extern "C" int fh_mat_abc(fh_mat_t A, fh_mat_t B, fh_mat_t C, fh_mat_t* D) {
  FH_GUARD_BEGIN
  FH_REQUIRE(A && B && C && D, "fh_mat_abc: null argument");
  FH_REQUIRE(A->n == B->m && B->n == C->m, "fh_mat_abc: shapes do not conform (A %dx%d, B %dx%d, C %dx%d)", A->m, A->n, B->m, B->n, C->m, C->n);
  return triple_product(A, B, C, D, "fh_mat_abc");
  FH_GUARD_END("fh_mat_abc")
}

extern "C" int fh_mat_matmul(fh_mat_t A, fh_mat_t B, fh_mat_t* Cout) {
  FH_GUARD_BEGIN
  FH_REQUIRE(A && B && Cout, "fh_mat_matmul: null argument");
  FH_REQUIRE(A->n == B->m, "fh_mat_matmul: shapes do not conform (A %dx%d, B %dx%d)", A->m, A->n, B->m, B->n);
  fh_mat_t C = nullptr;
  FH_TRY(spgemm_pattern(A, B, &C));
  FH_TRY(spgemm_numeric(A, B, C));
  *Cout = C;
  return 0;
  FH_GUARD_END("fh_mat_matmul")
}
