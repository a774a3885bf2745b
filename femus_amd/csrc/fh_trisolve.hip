// Natural-order sweeps on gfx950: the reference's PCSOR and PCILU level smoothers (a18 of SURVEY 8).
//   PCSOR  (PetscPreconditioner.cpp:219-222; 001_Poisson/main.cpp:240-242): PETSc's default omega_sor = 1, ONE local symmetric
//          sweep from a zero guess -- forward Gauss-Seidel over the rows in their natural order, then backward (MatSOR with
//          SOR_LOCAL_SYMMETRIC_SWEEP | SOR_ZERO_INITIAL_GUESS); on several ranks only the rank-local diagonal block takes part.
//   PCILU  (PetscPreconditioner.cpp:91-115): ILU(0) of the rank-local block in natural order, zero pivot 1e-16 and
//          MAT_SHIFT_NONZERO (LinearEquationSolverPetsc.cpp:444-446): a pivot with |p| <= zeropivot * (row sum of |a_ij|) restarts the
//          factorisation of A + shift I, shift = 100 eps first, doubled at every further restart (PETSc's MatPivotCheck_nz; PETSc
//          3.20.2 is not under /root/reference -- restated from its documented behaviour).
// A sequential sweep is a sparse triangular solve.  MI355X form: LEVEL SCHEDULING -- row i goes to level 1 + max(level of the rows
// j < i it reads), all rows of a level are independent, one launch per level (captured into the cycle's hipGraph like every other
// launch), 16 lanes per row.  The arithmetic per row is the sequential one (same operands, sum taken by 16 lanes instead of one),
// so a sweep agrees with the sequential sweep to rounding -- unlike the multicolour ordering (FH_SMOOTH_GS_COLOR), whose iteration
// history differs from the reference's.  Integer setup (levels, diagonal positions) on the host, once per pattern.
#include "fh_internal.h"
#include "fh_trisolve.h"
#ifndef TRI_STAMP
#define TRI_STAMP 0      // dev builds: shader-clock stamps of two waves over 32 steps of a long run (FEMUS_TRI_STAMP=1), printed when a plan is destroyed
#endif
// dev builds (TRI_STAMP 1): FEMUS_TRI_DBG switches stages of the run kernel off -- 1 no operand gathers, 4 no arithmetic, 8 no row-table loads; wrong results, timing only
#define TRI_ON(bit) (!TRI_STAMP || !(P.dbg & (bit)))
#if TRI_STAMP
static long long* g_tri_stamp = nullptr;
static void tri_stamp_report();
#endif
#include <algorithm>
#include <cmath>

// A level of at most TRI_SMALL rows is "small": a launch per level then costs more than the level's work (a two-dimensional stacked system of 70 000 unknowns
// has 1 500 levels of 46 rows on average: 3 000 launches of 5 us per triangular solve).  Consecutive small levels are swept by ONE workgroup of 1 024 threads
// (64 rows at a time, 16 lanes each) with a workgroup barrier between the levels: all its waves sit on one CU and share its L1, so the barrier's
// workgroup-scope release / acquire is all the ordering the rows of the next level need (round 5).
constexpr int TRI_SMALL = 256;
constexpr int TRI_RING = 2;                     // levels whose values the run kernel keeps in LDS (a ring indexed by the level number): this level and the one before.  (Eight levels, so that
                                                // most operands come from LDS instead of the gather, measured the same: the level is bound by the L1's requests for the rows' own lines)
constexpr int64_t TRI_SMALL_WORK = 32768;      // entries of a small level (all of its rows, both triangles)

// rows of the local block only: columns >= m are ghosts (block Jacobi across ranks, as PCSOR / PCILU are local)
static void schedule(const std::vector<int>& rp, const std::vector<int>& col, int m, bool forward, std::vector<int>& ptr, std::vector<int>& rows, std::vector<int>& lev) {
  lev.assign(m, 0);
  int nlev = 0;
  if (forward) {
    for (int i = 0; i < m; i++) {
      int l = 0;
      for (int k = rp[i]; k < rp[i + 1] && col[k] < i; k++) l = std::max(l, lev[col[k]] + 1);
      lev[i] = l;
      nlev = std::max(nlev, l + 1);
    }
  } else {
    for (int i = m - 1; i >= 0; i--) {
      int l = 0;
      for (int k = rp[i + 1] - 1; k >= rp[i] && col[k] > i; k--)
        if (col[k] < m) l = std::max(l, lev[col[k]] + 1);
      lev[i] = l;
      nlev = std::max(nlev, l + 1);
    }
  }
  ptr.assign(nlev + 1, 0);
  for (int i = 0; i < m; i++) ptr[lev[i] + 1]++;
  for (int l = 0; l < nlev; l++) ptr[l + 1] += ptr[l];
  rows.resize(m);
  std::vector<int> pos(ptr.begin(), ptr.end() - 1);
  for (int i = 0; i < m; i++) rows[pos[lev[i]]++] = i;      // ascending row index inside a level
}

static int tri_fill(fh_mat_t A, fh_tri_t T) {
  T->m = A->m;
  T->A_uid = A->uid;
  std::vector<int> rows, brows, flev, blev;
  schedule(A->h_rowptr, fh_hcol(A), A->m, true, T->fptr, rows, flev);
  FH_CHECK_HIP(hipMalloc(&T->d_frows, std::max(A->m, 1) * sizeof(int)));
  FH_CHECK_HIP(hipMemcpy(T->d_frows, rows.data(), (size_t)A->m * sizeof(int), hipMemcpyHostToDevice));
  schedule(A->h_rowptr, fh_hcol(A), A->m, false, T->bptr, brows, blev);
  FH_CHECK_HIP(hipMalloc(&T->d_brows, std::max(A->m, 1) * sizeof(int)));
  FH_CHECK_HIP(hipMemcpy(T->d_brows, brows.data(), (size_t)A->m * sizeof(int), hipMemcpyHostToDevice));
  // small = few rows AND little work: a level of 256 rows with 1 200 entries each (stacked three-dimensional systems) would keep ONE compute unit busy for
  // four passes of 75 steps where a launch spreads it over sixteen workgroups
  auto segments = [&](const std::vector<int>& ptr, const std::vector<int>& lrows, std::vector<int>& seg) {
    seg.clear();
    const int nl = (int)ptr.size() - 1;
    auto small = [&](int l) {
      if (!A->ctx->tri_runs || A->nnz >= (1 << 28) || ptr[l + 1] - ptr[l] > TRI_SMALL) return false;
      int64_t work = 0;
      for (int q = ptr[l]; q < ptr[l + 1]; q++) work += A->h_rowptr[lrows[q] + 1] - A->h_rowptr[lrows[q]];
      return work <= TRI_SMALL_WORK;
    };
    for (int l = 0; l < nl;) {
      if (!small(l)) {
        seg.insert(seg.end(), {l, 1, 0});
        l++;
        continue;
      }
      int e = l;
      while (e < nl && small(e)) e++;
      seg.insert(seg.end(), {l, e - l, 1});
      l = e;
    }
  };
  segments(T->fptr, rows, T->fseg);
  segments(T->bptr, brows, T->bseg);
  // operand sources of the run kernel: an entry whose column was computed in the level JUST BEFORE, inside the same run, reads the workgroup's LDS copy of that
  // level (rank of the column among the level's rows); every other entry reads z in global memory -- written at least two barriers earlier, or by another launch
  auto sources = [&](const std::vector<int>& ptr, const std::vector<int>& seg, const std::vector<int>& lvrows, const std::vector<int>& lev, bool forward, int** d_src) -> int {
    const int m = A->m, nl = (int)ptr.size() - 1;
    std::vector<int> run_first(std::max(nl, 1), -1);       // first level of the run a level belongs to (-1: a level with a launch of its own)
    for (size_t q = 0; q < seg.size(); q += 3)
      if (seg[q + 2])
        for (int l = seg[q]; l < seg[q] + seg[q + 1]; l++) run_first[l] = seg[q];
    std::vector<int> rank(m, 0);
    for (int l = 0; l < nl; l++)
      for (int k = ptr[l]; k < ptr[l + 1]; k++) rank[lvrows[k]] = k - ptr[l];
    const std::vector<int>& col = fh_hcol(A);
    std::vector<int> src(std::max<size_t>(col.size(), 1));
    for (int i = 0; i < m; i++)
      for (int k = A->h_rowptr[i]; k < A->h_rowptr[i + 1]; k++) {
        const int j = col[k];
        const bool takes = forward ? j < i : (j > i && j < m);
        // from LDS: the column was computed at most TRI_RING - 1 levels before, inside the same run (the ring slot of a level is its number modulo TRI_RING)
        const bool lds = takes && run_first[lev[i]] >= 0 && lev[j] < lev[i] && lev[i] - lev[j] < TRI_RING && lev[j] >= run_first[lev[i]];
        src[k] = lds ? -((lev[j] % TRI_RING) * TRI_SMALL + rank[j] + 1) : j;
      }
    FH_CHECK_HIP(hipMalloc(d_src, src.size() * sizeof(int)));
    FH_CHECK_HIP(hipMemcpy(*d_src, src.data(), src.size() * sizeof(int), hipMemcpyHostToDevice));
    return 0;
  };
  // per row in level order: {row, first entry of the sweep's triangle, its end, w}; forward: the entries left of the diagonal, w = position of the diagonal;
  // backward: the entries right of the diagonal without the ghost columns, w = start of the whole row (the lanes keep the entries they have in the
  // one-launch-per-level kernels, which walk the whole row: the same partial sums, the same bits -- the run kernel just does not load the other triangle)
  int64_t tri_len_f = 0, tri_len_b = 0;
  std::vector<int> hist_f(1024, 0), hist_b(1024, 0);
  auto level_rows = [&](const std::vector<int>& lvrows, const std::vector<int>& dp, bool forward, int** d_lv) -> int {
    std::vector<int> lv((size_t)std::max(A->m, 1) * 4, 0);
    const std::vector<int>& col = fh_hcol(A);
    for (int k = 0; k < A->m; k++) {
      const int i = lvrows[k];
      const int* b = col.data() + A->h_rowptr[i];
      const int* e = col.data() + A->h_rowptr[i + 1];
      lv[(size_t)k * 4 + 0] = i;
      const int lower = (int)(std::lower_bound(b, e, i) - b), upper = (int)(std::lower_bound(b, e, A->m) - std::upper_bound(b, e, i));
      (forward ? tri_len_f : tri_len_b) += forward ? lower : upper;
      (forward ? hist_f : hist_b)[std::min(forward ? lower : upper, 1023)]++;
      if (forward) {
        lv[(size_t)k * 4 + 1] = A->h_rowptr[i];
        lv[(size_t)k * 4 + 2] = (int)(std::lower_bound(b, e, i) - col.data());
        lv[(size_t)k * 4 + 3] = std::max(dp[i], 0);
      } else {
        lv[(size_t)k * 4 + 1] = (int)(std::upper_bound(b, e, i) - col.data());
        lv[(size_t)k * 4 + 2] = (int)(std::lower_bound(b, e, A->m) - col.data());
        lv[(size_t)k * 4 + 3] = A->h_rowptr[i];
      }
    }
    FH_CHECK_HIP(hipMalloc(d_lv, lv.size() * sizeof(int)));
    FH_CHECK_HIP(hipMemcpy(*d_lv, lv.data(), lv.size() * sizeof(int), hipMemcpyHostToDevice));
    return 0;
  };
  FH_TRY(sources(T->fptr, T->fseg, rows, flev, true, &T->d_fsrc));
  FH_TRY(sources(T->bptr, T->bseg, brows, blev, false, &T->d_bsrc));
  FH_CHECK_HIP(hipMalloc(&T->d_fptr, T->fptr.size() * sizeof(int)));
  FH_CHECK_HIP(hipMemcpy(T->d_fptr, T->fptr.data(), T->fptr.size() * sizeof(int), hipMemcpyHostToDevice));
  FH_CHECK_HIP(hipMalloc(&T->d_bptr, T->bptr.size() * sizeof(int)));
  FH_CHECK_HIP(hipMemcpy(T->d_bptr, T->bptr.data(), T->bptr.size() * sizeof(int), hipMemcpyHostToDevice));
  std::vector<int> dpos(A->m, -1);
  for (int i = 0; i < A->m; i++) {
    const int* b = fh_hcol(A).data() + A->h_rowptr[i];
    const int* e = fh_hcol(A).data() + A->h_rowptr[i + 1];
    const int* q = std::lower_bound(b, e, i);
    if (q != e && *q == i) dpos[i] = (int)(q - fh_hcol(A).data());
  }
  T->h_diagpos = dpos;
  FH_CHECK_HIP(hipMalloc(&T->d_prog, 2 * sizeof(unsigned long long)));
  FH_CHECK_HIP(hipMemset(T->d_prog, 0, 2 * sizeof(unsigned long long)));
  FH_TRY(level_rows(rows, dpos, true, &T->d_flv));
  FH_TRY(level_rows(brows, dpos, false, &T->d_blv));
  // register slots per lane and direction by the 90th percentile of the lengths of the direction's triangles (measured on the stacked two-dimensional system of the
  // known-answer test -- lower triangles: median 18, 90 % within 26, longest 33; upper: 12, 34, 76 --: lower sweep 0.62 / 0.65 / 0.69 ms with 2 / 3 / 4 slots, upper
  // sweep 0.73 / 0.67 / 0.64: a slot is instructions in every step whether a row fills it or not, a tail in the loop is a round trip to the L2 that the whole level
  // waits for)
  const double mean_f = (double)tri_len_f / std::max(A->m, 1), mean_b = (double)tri_len_b / std::max(A->m, 1);
  auto pct = [&](const std::vector<int>& h, double f) {
    int64_t acc = 0;
    for (int v = 0; v < 1024; v++) {
      acc += h[v];
      if ((double)acc >= f * A->m) return v;
    }
    return 1023;
  };
  FH_TRACE("fh_tri_create: lower p50 / p75 / p90 / max %d / %d / %d / %d, upper %d / %d / %d / %d", pct(hist_f, 0.5), pct(hist_f, 0.75), pct(hist_f, 0.9), pct(hist_f, 1.0), pct(hist_b, 0.5), pct(hist_b, 0.75), pct(hist_b, 0.9), pct(hist_b, 1.0));
  T->run_pf = pct(hist_f, 0.9) <= 32 ? 2 : 4;
  T->run_pb = pct(hist_b, 0.9) <= 32 ? 2 : 4;
  FH_TRACE("fh_tri_create: %d rows, mean lower / upper triangle %.1f / %.1f entries, %d / %d register slots per lane", A->m, mean_f, mean_b, T->run_pf, T->run_pb);
  if (const char* e = getenv("FEMUS_TRI_PF")) T->run_pf = T->run_pb = atoi(e) == 2 ? 2 : atoi(e) == 3 ? 3 : 4;      // measurements
  FH_CHECK_HIP(hipMalloc(&T->d_diagpos, std::max(A->m, 1) * sizeof(int)));
  FH_CHECK_HIP(hipMemcpy(T->d_diagpos, dpos.data(), (size_t)A->m * sizeof(int), hipMemcpyHostToDevice));
  FH_CHECK_HIP(hipMalloc(&T->d_t, std::max(A->m, 1) * sizeof(double)));
  return 0;
}

int fh_tri_create(fh_mat_t A, fh_tri_t* out) {
#if TRI_STAMP
  if (getenv("FEMUS_TRI_STAMP") && !g_tri_stamp) {
    hipMalloc(&g_tri_stamp, 512 * sizeof(long long));
    hipMemset(g_tri_stamp, 0, 512 * sizeof(long long));
  }
#endif
  fh_tri_t T = new fh_tri_s();
  const int rc = tri_fill(A, T);
  if (rc) {
    fh_tri_destroy(T);
    return rc;
  }
  *out = T;
  return 0;
}

void fh_tri_destroy(fh_tri_t T) {
  if (!T) return;
#if TRI_STAMP
  tri_stamp_report();
#endif
  for (void* p : {(void*)T->d_frows, (void*)T->d_brows, (void*)T->d_diagpos, (void*)T->d_lu, (void*)T->d_flag, (void*)T->d_t, (void*)T->d_fptr, (void*)T->d_bptr, (void*)T->d_fsrc, (void*)T->d_bsrc, (void*)T->d_flv, (void*)T->d_blv, (void*)T->d_prog, (void*)T->d_ppofs, (void*)T->d_ppos})
    if (p) hipFree(p);
  delete T;
}

// ---- symmetric Gauss-Seidel ------------------------------------------------------------------------------------------------
// forward, zero guess: t_i = r_i - sum_{j < i} a_ij z_j, z_i = dinv_i t_i   (entries right of the diagonal multiply zeros)
__global__ __launch_bounds__(256) void k_gs_fwd(const int* __restrict__ rows, int nrows, const int* __restrict__ rowptr, const int* __restrict__ col,
                                                const double* __restrict__ val, const double* __restrict__ dinv, const double* __restrict__ r,
                                                double* z, double* __restrict__ t) {
  const int rr = (blockIdx.x * 256 + threadIdx.x) >> 4, gl = threadIdx.x & 15;
  const bool live = rr < nrows;
  const int i = live ? rows[rr] : 0;
  double acc = 0.0;
  if (live)
    for (int k = rowptr[i] + gl; k < rowptr[i + 1]; k += 16) {
      const int j = col[k];
      if (j < i) acc += val[k] * z[j];
    }
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (live && gl == 0) {
    const double ti = r[i] - acc;
    t[i] = ti;
    z[i] = dinv[i] * ti;
  }
}

// backward: z_i = dinv_i (t_i - sum_{j > i, j local} a_ij z_j): the part left of the diagonal is the forward sweep's (kept in t, as PETSc's
// MatSOR keeps it), so a row reads nothing that a LATER row of the sweep overwrites -- with an unsymmetric pattern (a_ij != 0, a_ji == 0) row j < i
// is not ordered after row i by the schedule, and reading z_j in place would pick up its new value
__global__ __launch_bounds__(256) void k_gs_bwd(const int* __restrict__ rows, int nrows, const int* __restrict__ rowptr, const int* __restrict__ col,
                                                const double* __restrict__ val, const double* __restrict__ dinv, const double* __restrict__ t,
                                                double* z, int m) {
  const int rr = (blockIdx.x * 256 + threadIdx.x) >> 4, gl = threadIdx.x & 15;
  const bool live = rr < nrows;
  const int i = live ? rows[rr] : 0;
  double acc = 0.0;
  if (live)
    for (int k = rowptr[i] + gl; k < rowptr[i + 1]; k += 16) {
      const int j = col[k];
      if (j > i && j < m) acc += val[k] * z[j];
    }
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (live && gl == 0) z[i] = dinv[i] * (t[i] - acc);
}

// ---- runs of small levels in one workgroup: the row bodies of the four kernels above / below, levels separated by a workgroup barrier ----
constexpr int TRI_AHEAD = 1024, TRI_POLL_CAP = 1 << 15;
static unsigned tri_launch_id = 0;
// the prefetching workgroup pays for runs of many levels (env FEMUS_TRI_PREFETCH: 0 never, else the block count to launch)
static int tri_blocks(int nl) {
  static int v = -1;
  if (v < 0) v = getenv("FEMUS_TRI_PREFETCH") ? atoi(getenv("FEMUS_TRI_PREFETCH")) : 9;
  return (v > 1 && nl >= 64) ? v : 1;
}
static int tri_ahead() {
  static int v = -1;
  if (v < 0) v = getenv("FEMUS_TRI_AHEAD") ? atoi(getenv("FEMUS_TRI_AHEAD")) : TRI_AHEAD;
  return v;
}
static int tri_dbg() {
  static int v = -1;
  if (v < 0) v = getenv("FEMUS_TRI_DBG") ? atoi(getenv("FEMUS_TRI_DBG")) : 0;
  return v;
}
struct TriRun {
  const int *rows, *lptr, *rowptr, *src, *diagpos;
  const int4* lv;
  const double *val, *dinv, *r, *t_in;
  double *z, *t_out;
  int l0, nl, m;
  long long* stamp;        // dev builds (TRI_STAMP): shader-clock stamps of two waves over 32 steps
  unsigned long long* prog;      // progress of the main workgroup: (launch id << 32) | first table row of the level in work, ~0 in the low half when done
  unsigned id;
  int ahead;               // table rows the prefetching workgroup keeps in front of the level in work
  int nblk;                // workgroups of the launch: 1, or TRI_PF_BLOCK + 1 with the prefetching one
  int dbg;                 // timing aids (FEMUS_TRI_DBG, wrong results): 1 no operand gathers, 2 no entry loads, 4 no arithmetic, 8 no row-table loads
};

#if TRI_STAMP
static void tri_stamp_report() {
  if (!g_tri_stamp) return;
  long long h[512];
  hipDeviceSynchronize();
  hipMemcpy(h, g_tri_stamp, sizeof(h), hipMemcpyDeviceToHost);
  for (int w = 0; w < 2; w++) {
    double d[6] = {0, 0, 0, 0, 0, 0};
    int cnt = 0;
    for (int st = 0; st < 31; st++) {
      const long long* a = h + w * 256 + st * 8;
      if (!a[0] || !a[8]) continue;
      for (int k = 0; k < 5; k++) d[k] += (double)(a[k + 1] - a[k]);
      d[5] += (double)(a[8] - a[0]);
      cnt++;
    }
    if (cnt) fprintf(stderr, "[tri stamp] wave %d, %d steps: wait + C issue %.0f, A + Z issue %.0f, level %.0f, - %.0f, barrier %.0f; step %.0f (s_memtime ticks; the stamps themselves slow the stamped steps)\n", w * 8, cnt, d[0] / cnt, d[1] / cnt, d[2] / cnt,
                     d[3] / cnt, d[4] / cnt, d[5] / cnt);
  }
  hipMemset(g_tri_stamp, 0, 512 * sizeof(long long));
}
#endif
// The run kernel (round 6).  A two-dimensional stacked system of 70 000 unknowns has 1 500 levels of 46 rows; a level that loads its rows when it gets to them
// costs the chain level pointer -> row -> entries -> operands, four trips to the L2 (0.7 us, round 5).  Here:
//  * the row, the bounds of the sweep's triangle and the lane phase come in ONE 16-byte load from a table in level order (fh_tri_s::d_flv / d_blv), and only the
//    sweep's triangle is loaded at all;
//  * a software pipeline: in the step that computes level l the workgroup asks for the table rows of level l + 3 (stage A, into a register set of their own), the
//    first 16 PF entries per row / right-hand side / diagonal of level l + 2 (stage C) and the global operands of level l + 1 (stage Z: written two barriers ago
//    or earlier) -- all at the START of the step, behind one wait for what the step before asked for -- into three slot sets used round robin (six steps per
//    trip, straight-line code: no register moves);
//  * operands computed by the level JUST BEFORE cannot be loaded ahead: they are read from the workgroup's LDS copy of that level (P.src < 0);
//  * after the barrier a level is: LDS reads, the sums (sixteen lanes per row, the tree by DPP row shifts), a store;
//  * a second workgroup of the launch on the same XCD touches the rows' lines TRI_AHEAD table rows ahead (tri_prefetch).
// The lane's entries are added in the order of the one-launch-per-level kernels: the same bits (option tri_runs = 0, tests/test_gpu_multigrid.py).
#if TRI_STAMP
#define TRI_T(k) if (stamping) { P.stamp[sbase + (size_t)(l + PHV - P.l0 - 200) * 8 + (k)] = (long long)__builtin_amdgcn_s_memtime(); }
#else
#define TRI_T(k)
#endif
// PF: entries per lane held in registers (16 PF entries of the triangle per row, the rest in a loop): 2 or 4 per direction, chosen in tri_fill
#ifndef TRI_SKIP
#define TRI_SKIP 1
#endif
// A slot = what the pipeline holds of one level for this lane.  Nothing in it is TESTED in the step that loads it (a select on a value a step too early is a wait for
// the load just issued, with every other load of the step queued behind it -- the ISA of the first version of this pipeline showed two such full round trips per
// level: the `k < re ? c : none` select of stage C, and register copies of stage A's row behind the barrier, put there by the guards around the unrolled steps):
// c[] holds the sources as loaded, nv says how many of the slots belong to the row (the others are not loaded).
// byte offsets as unsigned 32-bit values on a uniform base: one shift per address instead of a sign extension and a 64-bit multiply-add (runs are only planned
// for matrices of less than 2^28 entries)
template <class T>
__device__ __forceinline__ T tri_ld(const T* base, int idx) {
  return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + (size_t)((unsigned)idx * (unsigned)sizeof(T)));
}
template <int PF>
struct TriSlot {
  int i, lo, hi, w, active, nv, first, b, n, c[PF];      // lo, hi: the triangle's entries; w: see level_rows; first: this lane's first entry; b, n: first row / rows of the level (wave-uniform)
  double v[PF], zq[PF], e0, e1;
};
template <int KIND>
__device__ __forceinline__ bool tri_takes(int j, int i, int m) {
  return (KIND == 0 || KIND == 2) ? (j < i) : (j > i && j < m);
}
// stage A: this lane group's row of a level whose pointer pair (b, b + n) is at hand (loaded a step ahead); a group beyond the level repeats its last row and stores
// nothing.  The row goes into a register set of its own (two of them, alternating): it is asked for at the START of a step and used at the start of the next one,
// a whole step later -- loaded straight into the slot that the step's level still computes from, it could only be asked for at the step's end and was waited for,
// a round trip to the L2, every level
struct TriRow {
  int i, lo, hi, w, active, b, n;
};
__device__ __forceinline__ void tri_stage_a(const TriRun& P, bool in_run, int b, int n, int grp, TriRow& R) {
  R.active = (in_run && grp < n) ? 1 : 0;
  R.b = b;
  R.n = in_run ? n : 0;
  const int4 q = P.lv[b + min(grp, n - 1)];
  R.i = q.x; R.lo = q.y; R.hi = q.z; R.w = q.w;
}
// stage C: the lane's first entries (only those the row has: a load instruction costs the memory pipe of the ONE compute unit its cycles whether its lanes
// carry an entry or repeat the last one, and that pipe is what bounds a level), the right-hand side and the diagonal
template <int KIND, int PF>
__device__ __forceinline__ void tri_stage_c(const TriRun& P, int gl, const TriRow& R, TriSlot<PF>& S) {
  S.i = R.i; S.lo = R.lo; S.hi = R.hi; S.w = R.w; S.active = R.active; S.b = R.b; S.n = R.n;
#if TRI_SKIP
  if (__builtin_amdgcn_ballot_w64(S.active != 0) == 0ull) return;        // a wave without a row in that level (levels hold 46 rows on average, the workgroup 64 groups)
#endif
  const int i = S.i;
  // the lane's entries are those at positions = gl (mod 16) counted from the start of the whole row, as in the kernels that walk the whole row
  S.first = (KIND == 0 || KIND == 2) ? S.lo + gl : S.lo + ((S.w + gl - S.lo) & 15);
  const int len = S.active ? S.hi - S.first : 0;
  S.nv = len <= 0 ? 0 : min((len + 15) >> 4, PF);
#pragma unroll
  for (int q = 0; q < PF; q++)
    if (q < S.nv) {
      S.c[q] = tri_ld(P.src, S.first + 16 * q);
      S.v[q] = tri_ld(P.val, S.first + 16 * q);
    }
  if (S.active) {
    S.e0 = KIND == 1 ? tri_ld(P.t_in, i) : KIND == 3 ? tri_ld((const double*)P.z, i) : tri_ld(P.r, i);
    S.e1 = (KIND == 0 || KIND == 1) ? tri_ld(P.dinv, i) : KIND == 3 ? tri_ld(P.val, S.lo - 1) : 1.0;      // the diagonal of U sits just before the row's upper entries (inverted in stage Z, a step later: off the level's critical path)
  }
}
// stage Z: operands from global memory (entries of levels at least two back, or of another launch)
template <int KIND, int PF>
__device__ __forceinline__ void tri_stage_z(const TriRun& P, TriSlot<PF>& S) {
#if TRI_SKIP
  if (__builtin_amdgcn_ballot_w64(S.active != 0) == 0ull) return;
#endif
#pragma unroll
  for (int q = 0; q < PF; q++)
    if (q < S.nv && S.c[q] >= 0) S.zq[q] = tri_ld((const double*)P.z, S.c[q]);
  if (KIND == 3 && S.active) S.e1 = 1.0 / S.e1;
}
template <int KIND>
__device__ __forceinline__ double tri_store(const TriRun& P, int i, double acc, double e0, double e1) {
  double zi;
  if (KIND == 0) {
    const double ti = e0 - acc;
    P.t_out[i] = ti;
    zi = e1 * ti;
  } else if (KIND == 1) {
    zi = e1 * (e0 - acc);
  } else if (KIND == 2) {
    zi = e0 - acc;
  } else {
    zi = (e0 - acc) * e1;        // e1 = 1 / u_ii, formed a step ahead (PETSc's factored matrices store the inverted diagonal and multiply, MatSolve_SeqAIJ)
  }
  P.z[i] = zi;
  return zi;
}
// sum over the 16 lanes of a row group into its lane 0: the tree of `for (off = 8; off; off >>= 1) acc += __shfl_down(acc, off)` (same operands, same order, same bits)
// with the lane exchange as DPP row shifts -- a row of the DPP unit IS 16 lanes -- instead of four dependent ds_bpermute round trips through the LDS pipe on
// the critical path of every level
template <int N>
__device__ __forceinline__ double tri_row_from(double v) {      // lane i <- lane i + N of the same 16 lanes (lanes past the row end read zero: their sums are not used)
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x100 + N, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x100 + N, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double tri_reduce16(double acc) {
  acc += tri_row_from<8>(acc);
  acc += tri_row_from<4>(acc);
  acc += tri_row_from<2>(acc);
  acc += tri_row_from<1>(acc);
  return acc;
}
// the level held by slot S; zp / zc: LDS copies of the previous / of this level
template <int KIND, int PF>
__device__ __forceinline__ void tri_level(const TriRun& P, int gl, int grp, const TriSlot<PF>& S, const double* zp, double* zc) {
  if (S.active) {                                                       // first 64 rows: from the registers the pipeline filled
    const int i = S.i;
    double zz[PF];
#pragma unroll
    for (int q = 0; q < PF; q++)
      if (q < S.nv && S.c[q] < 0) zz[q] = zp[-S.c[q] - 1];      // all LDS reads at once, one wait (the operands of the level just before)
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < PF; q++) {
      const int c = S.c[q];
      const double op = c < 0 ? zz[q] : S.zq[q];
      if (q < S.nv) acc += S.v[q] * op;
    }
    for (int k = S.first + 16 * PF; k < S.hi; k += 16) {
      const int j = P.src[k];
      acc += P.val[k] * (j < 0 ? zp[-j - 1] : P.z[j]);
    }
    acc = tri_reduce16(acc);
    if (gl == 0) zc[grp] = tri_store<KIND>(P, i, acc, S.e0, S.e1);
  }
  for (int r0 = 64; r0 < S.n; r0 += 64) {                               // the rest of a level of more than 64 rows
    const int rr = r0 + grp;
    const bool live = rr < S.n;
    const int i = live ? P.rows[S.b + rr] : 0;
    double acc = 0.0;
    if (live)
      for (int k = P.rowptr[i] + gl; k < P.rowptr[i + 1]; k += 16) {
        const int j = P.src[k];
        if (j < 0) acc += P.val[k] * zp[-j - 1];
        else if (tri_takes<KIND>(j, i, P.m)) acc += P.val[k] * P.z[j];
      }
    acc = tri_reduce16(acc);
    if (live && gl == 0) {
      const double e0 = KIND == 1 ? P.t_in[i] : KIND == 3 ? P.z[i] : P.r[i];
      const double e1 = (KIND == 0 || KIND == 1) ? P.dinv[i] : KIND == 3 ? 1.0 / P.val[P.diagpos[i]] : 1.0;
      zc[rr] = tri_store<KIND>(P, i, acc, e0, e1);
    }
  }
}

// The run's rows stream 25 MB per sweep through ONE compute unit with a look-ahead of a level or two: what the pipeline cannot hide is the distance to HBM.  A second
// workgroup of the same launch -- workgroups go round the eight XCDs in turn, so number 8 shares the L2 of number 0 -- walks the level-ordered row table ahead of
// the main one and touches the lines of the rows' entries (values and sources) TRI_AHEAD table rows in front of the level in work, which it reads from a progress
// word the main workgroup stores once per level.  It never holds the main workgroup up (nothing waits for it), leaves when the main one is done or when it sees no
// progress for a few milliseconds, and nothing it loads is used.

__device__ void tri_prefetch(const TriRun& P) {
  const int first = P.lptr[P.l0], end = P.lptr[P.l0 + P.nl];
  int pos = first, polls = 0;
  double acc = 0.0;
  int iacc = 0;
  while (pos < end) {
    const unsigned long long w = __hip_atomic_load(P.prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int cur = first;
    if ((unsigned)(w >> 32) == P.id) {
      if ((unsigned)w == 0xffffffffu) break;
      cur = (int)(unsigned)w;
    }
    const int limit = min(end, cur + P.ahead);
    if (pos >= limit) {
      if (++polls > TRI_POLL_CAP) break;
      __builtin_amdgcn_s_sleep(8);
      continue;
    }
    const int e = pos + (int)threadIdx.x;
    if (e < limit) {
      const int4 q = P.lv[e];
      for (int k0 = q.y & ~7; k0 < q.z; k0 += 64) {          // one load per 64-byte line, eight lines in flight
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = (k0 + 8 * u < q.z) ? P.val[k0 + 8 * u] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; u++) acc += v[u];
      }
      for (int k0 = q.y & ~15; k0 < q.z; k0 += 64) {
        int c[4];
#pragma unroll
        for (int u = 0; u < 4; u++) c[u] = (k0 + 16 * u < q.z) ? P.src[k0 + 16 * u] : 0;
#pragma unroll
        for (int u = 0; u < 4; u++) iacc ^= c[u];
      }
    }
    pos = min(pos + (int)blockDim.x, limit);
  }
  if (acc == 1.2345678e-301 && iacc == 0x7fffffff) P.prog[1] = 0;      // never: keeps the loads alive
}

template <int KIND, int PF>      // 0: Gauss-Seidel forward, 1: backward, 2: ILU lower, 3: ILU upper
__global__ __launch_bounds__(1024) void k_tri_run(TriRun P) {
  __shared__ double zl[TRI_RING * TRI_SMALL];     // z of the rows of the last TRI_RING levels, by level number modulo TRI_RING and rank inside the level
  const int gl = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int l0 = P.l0, lend = P.l0 + P.nl;
  if (blockIdx.x != 0) {
    if ((int)blockIdx.x == P.nblk - 1) tri_prefetch(P);
    return;
  }
  TriSlot<PF> S0, S1, S2;
  TriRow R0, R1;
  auto lp = [&](int L) { return P.lptr[min(L, lend)]; };        // level pointers, clamped to the run (lend itself is the end of the last level)
  const int fb = lp(l0), fn = lp(l0 + 1) - fb;      // what stage A repeats beyond the end of the run (an existing level, nothing stored)
  auto rows_of = [&](int L, int b, int e, TriRow& R) {         // stage A of level L with its pointer pair (b, e)
    const bool in = L < lend;
    tri_stage_a(P, in, in ? b : fb, in ? e - b : fn, grp, R);
  };
  // prologue: level l0 complete, l0 + 1 up to its entries, l0 + 2 its rows (in R0: l0 + 2 is even from l0), the pointer pair of l0 + 3
  int nb, ne;
  {
    const int p0 = lp(l0), p1 = lp(l0 + 1), p2 = lp(l0 + 2), p3 = lp(l0 + 3);
    rows_of(l0, p0, p1, R0);
    rows_of(l0 + 1, p1, p2, R1);
    tri_stage_c<KIND, PF>(P, gl, R0, S0);
    tri_stage_c<KIND, PF>(P, gl, R1, S1);
    rows_of(l0 + 2, p2, p3, R0);
    nb = p3;
    ne = lp(l0 + 4);
  }
  tri_stage_z<KIND, PF>(P, S0);
  // step for level l in slot CUR: entries of l + 2 (C, from the rows asked for a step ago), rows of l + 3 (A), operands of l + 1 (Z), the level itself, pointer pair
  // of l + 5.  EVERYTHING a step asks for is asked for at its start, behind one wait for what the step before asked for: every load has a whole step to come back.
  // The LDS buffers alternate with the level, the row sets too, the slots with a period of three: six steps per trip, NOT guarded one by one -- the steps past the
  // end of the run find inactive slots and only meet at the barriers -- so that the trip is straight-line code.
#define TRI_STEP(PH, CUR, NZ, NC, RC, RA)                                              \
  {                                                                                    \
    [[maybe_unused]] constexpr int PHV = PH;                                           \
    [[maybe_unused]] const bool stamping = TRI_STAMP && P.stamp && (threadIdx.x == 0 || threadIdx.x == 512) && l + PH - P.l0 >= 200 && l + PH - P.l0 < 232; \
    [[maybe_unused]] const size_t sbase = threadIdx.x == 0 ? 0 : 256;                  \
    TRI_T(0)                                                                           \
    if (threadIdx.x == 0 && P.nblk > 1) __hip_atomic_store(P.prog, ((unsigned long long)P.id << 32) | (unsigned)CUR.b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
    if (TRI_ON(2)) tri_stage_c<KIND, PF>(P, gl, RC, NC);                               \
    TRI_T(1)                                                                           \
    if (TRI_ON(8)) rows_of(l + PH + 3, nb, ne, RA);                                    \
    nb = ne;                                                                           \
    ne = lp(l + PH + 5);                                                               \
    if (TRI_ON(1)) tri_stage_z<KIND, PF>(P, NZ);                                       \
    TRI_T(2)                                                                           \
    if (TRI_ON(4)) tri_level<KIND, PF>(P, gl, grp, CUR, zl, zl + ((l + PH) % TRI_RING) * TRI_SMALL); \
    TRI_T(3)                                                                           \
    TRI_T(4)                                                                           \
    __syncthreads();                                                                   \
    TRI_T(5)                                                                           \
  }
  for (int l = l0; l < lend; l += 6) {
    TRI_STEP(0, S0, S1, S2, R0, R1)
    TRI_STEP(1, S1, S2, S0, R1, R0)
    TRI_STEP(2, S2, S0, S1, R0, R1)
    TRI_STEP(3, S0, S1, S2, R1, R0)
    TRI_STEP(4, S1, S2, S0, R0, R1)
    TRI_STEP(5, S2, S0, S1, R1, R0)
  }
#undef TRI_STEP
  if (threadIdx.x == 0 && P.nblk > 1) __hip_atomic_store(P.prog, ((unsigned long long)P.id << 32) | 0xffffffffull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// z = B r, B = one symmetric Gauss-Seidel sweep of A's local block from z = 0
int fh_tri_ssor_apply(fh_tri_t T, fh_mat_t A, const double* dinv, const double* r, double* z) {
  hipStream_t s = A->ctx->stream;
  const int nf = (int)T->fptr.size() - 1, nb = (int)T->bptr.size() - 1;
  (void)nf; (void)nb;
  TriRun P = {nullptr, nullptr, A->d_rowptr, nullptr, T->d_diagpos, nullptr, A->d_val, dinv, r, T->d_t, z, T->d_t, 0, 0, A->m, nullptr, T->d_prog, 0u, tri_ahead(), 1, tri_dbg()};
  for (size_t q = 0; q < T->fseg.size(); q += 3) {
    const int l = T->fseg[q];
    if (T->fseg[q + 2]) {
      P.rows = T->d_frows; P.lptr = T->d_fptr; P.src = T->d_fsrc; P.lv = reinterpret_cast<const int4*>(T->d_flv); P.l0 = l; P.nl = T->fseg[q + 1];
      P.id = ++tri_launch_id; P.nblk = tri_blocks(P.nl);
      if (T->run_pf == 2) hipLaunchKernelGGL((k_tri_run<0, 2>), dim3(P.nblk), dim3(1024), 0, s, P);
      else if (T->run_pf == 3) hipLaunchKernelGGL((k_tri_run<0, 3>), dim3(P.nblk), dim3(1024), 0, s, P);
      else hipLaunchKernelGGL((k_tri_run<0, 4>), dim3(P.nblk), dim3(1024), 0, s, P);
    } else {
      const int n = T->fptr[l + 1] - T->fptr[l];
      hipLaunchKernelGGL(k_gs_fwd, dim3(fh_div_up((int64_t)n * 16, 256)), dim3(256), 0, s, T->d_frows + T->fptr[l], n, A->d_rowptr, A->d_col, A->d_val, dinv,
                         r, z, T->d_t);
    }
  }
  for (size_t q = 0; q < T->bseg.size(); q += 3) {
    const int l = T->bseg[q];
    if (T->bseg[q + 2]) {
      P.rows = T->d_brows; P.lptr = T->d_bptr; P.src = T->d_bsrc; P.lv = reinterpret_cast<const int4*>(T->d_blv); P.l0 = l; P.nl = T->bseg[q + 1];
      P.id = ++tri_launch_id; P.nblk = tri_blocks(P.nl);
      if (T->run_pb == 2) hipLaunchKernelGGL((k_tri_run<1, 2>), dim3(P.nblk), dim3(1024), 0, s, P);
      else if (T->run_pb == 3) hipLaunchKernelGGL((k_tri_run<1, 3>), dim3(P.nblk), dim3(1024), 0, s, P);
      else hipLaunchKernelGGL((k_tri_run<1, 4>), dim3(P.nblk), dim3(1024), 0, s, P);
    } else {
      const int n = T->bptr[l + 1] - T->bptr[l];
      hipLaunchKernelGGL(k_gs_bwd, dim3(fh_div_up((int64_t)n * 16, 256)), dim3(256), 0, s, T->d_brows + T->bptr[l], n, A->d_rowptr, A->d_col, A->d_val, dinv,
                         T->d_t, z, A->m);
    }
  }
  FH_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---- ILU(0) ----------------------------------------------------------------------------------------------------------------
// one row per 16-lane group, rows of one level together.  Row i: for k < i in the row, ascending: l_ik = a_ik / u_kk, then
// a_ij -= l_ik u_kj for the j > k that row k AND row i hold (IKJ form of the reference's MatLUFactorNumeric on the fixed pattern).
__device__ __forceinline__ void tri_group_sync() {      // LDS operations of one wave execute in order; keep the compiler from reordering them
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// LCOL: the columns of the row live in LDS beside its values (rows of up to 800 entries): the position search of every update -- seven dependent loads per
// entry of row k -- then runs at LDS latency instead of cache latency (round 5: the factorisation is a chain of such searches, 120 us per level of 46 rows before)
template <bool LCOL>
__global__ __launch_bounds__(256) void k_ilu_factor(const int* __restrict__ rows, int nrows, const int* __restrict__ rowptr, const int* __restrict__ col,
                                                    const int* __restrict__ diagpos, double* lu, int m, int maxrow, double zeropivot,
                                                    int* __restrict__ flag) {
  extern __shared__ double tri_rows[];             // [16 groups][maxrow]: the row being eliminated lives in LDS
  const int rr = (blockIdx.x * 256 + threadIdx.x) >> 4, gl = threadIdx.x & 15;
  if (rr >= nrows) return;
  double* w = tri_rows + (size_t)(threadIdx.x >> 4) * maxrow;
  int* wc = reinterpret_cast<int*>(tri_rows + (size_t)16 * maxrow) + (size_t)(threadIdx.x >> 4) * maxrow;
  const int i = rows[rr];
  const int rs = rowptr[i], re = rowptr[i + 1];
  for (int p = rs + gl; p < re; p += 16) {
    w[p - rs] = lu[p];
    if (LCOL) wc[p - rs] = col[p];
  }
  tri_group_sync();
  // what step p + 1 needs of its pivot row k' (final since an earlier launch: its diagonal position, u_k'k', its end, the lane's first entry right of the
  // diagonal) is loaded while step p updates the row -- the chain diagpos[k] -> lu[dk] -> col / lu[q] is otherwise paid once per lower entry
  int k = rs < re ? (LCOL ? wc[0] : col[rs]) : i;
  int dk = k < i ? diagpos[k] : 0, ke = k < i ? rowptr[k + 1] : 0;
  double ukk = k < i ? 1.0 / lu[dk] : 1.0;          // the INVERTED pivot, formed ahead of the chain: the multiplier is a product, as in MatLUFactorNumeric_SeqAIJ
  int jq = (k < i && dk + 1 + gl < ke) ? col[dk + 1 + gl] : -1;
  double uq = (k < i && dk + 1 + gl < ke) ? lu[dk + 1 + gl] : 0.0;
  for (int p = rs; p < re; p++) {
    if (k >= i) break;
    const int k2 = p + 1 < re ? (LCOL ? wc[p + 1 - rs] : col[p + 1]) : i;
    const int dk2 = k2 < i ? diagpos[k2] : 0, ke2 = k2 < i ? rowptr[k2 + 1] : 0;
    const double ukk2 = k2 < i ? 1.0 / lu[dk2] : 1.0;
    const int jq2 = (k2 < i && dk2 + 1 + gl < ke2) ? col[dk2 + 1 + gl] : -1;
    const double uq2 = (k2 < i && dk2 + 1 + gl < ke2) ? lu[dk2 + 1 + gl] : 0.0;
    const double lik = w[p - rs] * ukk;             // row k is final: it was factored by an earlier launch (lower level)
    tri_group_sync();
    if (gl == 0) w[p - rs] = lik;
    for (int q = dk + 1 + gl; q < ke; q += 16) {
      const bool first = q == dk + 1 + gl;
      const int j = first ? jq : col[q];
      const double u = first ? uq : lu[q];
      if (j >= m) continue;                         // ghost column: not part of the local block
      int lo = p + 1, hi = re - 1;
      while (lo <= hi) {
        const int mid = lo + ((hi - lo) >> 1);   // (lo + hi) overflows beyond 2^30 non-zeros
        const int cc = LCOL ? wc[mid - rs] : col[mid];
        if (cc == j) {
          w[mid - rs] -= lik * u;                    // distinct j per lane: distinct slots
          break;
        }
        if (cc < j) lo = mid + 1; else hi = mid - 1;
      }
    }
    tri_group_sync();
    k = k2; dk = dk2; ke = ke2; ukk = ukk2; jq = jq2; uq = uq2;
  }
  for (int p = rs + gl; p < re; p += 16) lu[p] = w[p - rs];
  if (gl == 0) {
    const int di = diagpos[i];
    double rsum = 0.0;
    for (int p = (di < 0 ? rs : di + 1); p < re; p++) rsum += fabs(w[p - rs]);   // the U part of the row without its diagonal (PETSc: sctx.rs)
    if (di < 0 || !(fabs(w[di - rs]) > zeropivot * rsum)) atomicOr(flag, 1);
  }
}

// The same elimination with the pivot rows asked for AHEAD of the chain (round 6).  What a pivot costs above is two dependent trips to memory -- diagpos[k] /
// rowptr[k + 1], then u_kk and the row's entries -- behind a look-ahead of one pivot: 2.8 us per lower entry, 89 us for a level of 46 rows of 31 lower entries.
// The pivot rows are final (earlier launches), so nothing but the arithmetic depends on the order: first the descriptors of ALL pivots of the row (diagonal
// position, end, u_kk) go into LDS, sixteen pivots per trip; then the elimination runs over them with the lane's first two entries of pivots p + 1 .. p + 3 in three
// register sets (loop unrolled by three), asked for three steps before they are used.  Same operations on every entry in the same order: the same bits.
struct IluPivot {
  int j0, j1;
  double u0, u1;
};
__global__ __launch_bounds__(256) void k_ilu_factor_ahead(const int* __restrict__ rows, int nrows, const int* __restrict__ rowptr, const int* __restrict__ col,
                                                          const int* __restrict__ diagpos, double* lu, int m, int maxrow, double zeropivot,
                                                          int* __restrict__ flag) {
  extern __shared__ double tri_rows[];             // per group: the row's values, the pivots' u_kk; then its columns, the pivots' diagonal positions and ends
  const int rr = (blockIdx.x * 256 + threadIdx.x) >> 4, gl = threadIdx.x & 15, g = threadIdx.x >> 4;
  if (rr >= nrows) return;
  double* w = tri_rows + (size_t)g * maxrow;
  double* pu = tri_rows + (size_t)(16 + g) * maxrow;
  int* ibase = reinterpret_cast<int*>(tri_rows + (size_t)32 * maxrow);
  int* wc = ibase + (size_t)g * maxrow;
  int* pd = ibase + (size_t)(16 + g) * maxrow;
  int* pe = ibase + (size_t)(32 + g) * maxrow;
  const int i = rows[rr];
  const int rs = rowptr[i], re = rowptr[i + 1], di = diagpos[i];
  const int nlow = max(di - rs, 0);                 // entries left of the diagonal = pivots (the columns are sorted)
  for (int p = rs + gl; p < re; p += 16) {
    w[p - rs] = lu[p];
    wc[p - rs] = col[p];
  }
  tri_group_sync();
  for (int t = gl; t < nlow; t += 16) {
    const int k = wc[t], dk = diagpos[k];
    pd[t] = dk;
    pe[t] = rowptr[k + 1];
    pu[t] = 1.0 / lu[dk];      // the inverted pivot (PETSc keeps the diagonal of the factor inverted and multiplies)
  }
  tri_group_sync();
  auto fetch = [&](int t, IluPivot& v) {            // the lane's first two entries right of pivot t's diagonal
    v.j0 = v.j1 = -1;
    v.u0 = v.u1 = 0.0;
    if (t < nlow) {
      const int q0 = pd[t] + 1 + gl, ke = pe[t];
      if (q0 < ke) {
        v.j0 = col[q0];
        v.u0 = lu[q0];
      }
      if (q0 + 16 < ke) {
        v.j1 = col[q0 + 16];
        v.u1 = lu[q0 + 16];
      }
    }
  };
  const int nrow = re - rs;
  auto update = [&](int t, int j, double u, double lik) {      // a_ij -= l_ik u_kj where row i holds column j (right of the pivot's position)
    if (j < 0 || j >= m) return;                               // no entry / ghost column: not part of the local block
    int lo = t + 1, hi = nrow - 1;
    while (lo <= hi) {
      const int mid = (lo + hi) >> 1;
      const int cc = wc[mid];
      if (cc == j) {
        w[mid] -= lik * u;                                     // distinct j per lane: distinct slots
        break;
      }
      if (cc < j) lo = mid + 1; else hi = mid - 1;
    }
  };
  IluPivot v0, v1, v2;
  fetch(0, v0);
  fetch(1, v1);
  fetch(2, v2);
#define ILU_STEP(T, V)                                                                     \
  if ((T) < nlow) {                                                                        \
    const double lik = w[T] * pu[T];                                                       \
    tri_group_sync();                                                                      \
    if (gl == 0) w[T] = lik;                                                               \
    update(T, V.j0, V.u0, lik);                                                            \
    update(T, V.j1, V.u1, lik);                                                            \
    for (int q = pd[T] + 33 + gl; q < pe[T]; q += 16) update(T, col[q], lu[q], lik);       \
    tri_group_sync();                                                                      \
    fetch((T) + 3, V);                                                                     \
  }
  for (int t = 0; t < nlow; t += 3) {
    ILU_STEP(t, v0)
    ILU_STEP(t + 1, v1)
    ILU_STEP(t + 2, v2)
  }
#undef ILU_STEP
  for (int p = rs + gl; p < re; p += 16) lu[p] = w[p - rs];
  if (gl == 0) {
    double rsum = 0.0;
    for (int p = (di < 0 ? rs : di + 1); p < re; p++) rsum += fabs(w[p - rs]);   // the U part of the row without its diagonal (PETSc: sctx.rs)
    if (di < 0 || !(fabs(w[di - rs]) > zeropivot * rsum)) atomicOr(flag, 1);
  }
}

// The elimination with a PLAN (round 6): what is left per pivot above is the position search -- six dependent LDS reads per entry of the pivot row, two or three
// entries per lane, every one of the row's ~ 30 pivots: 2.5 us per pivot.  Where an entry of pivot row k lands in row i depends on the pattern alone, so the
// searches are done once (k_ilu_plan, all rows at once, when the first factorisation of a pattern is asked for) and kept as one byte per (pivot, entry): the
// factorisation then reads u_kj and a byte and updates w[byte].  Same operations on every entry in the same order as the searching kernels: the same bits.
__global__ __launch_bounds__(256) void k_ilu_plan(int m, const int* __restrict__ rowptr, const int* __restrict__ col, const int* __restrict__ diagpos,
                                                  const int* __restrict__ ppofs, unsigned char* __restrict__ ppos) {
  const int i = (blockIdx.x * 256 + threadIdx.x) >> 4, gl = threadIdx.x & 15;
  if (i >= m) return;
  const int rs = rowptr[i], re = rowptr[i + 1], nlow = max(diagpos[i] - rs, 0);
  for (int t = 0; t < nlow; t++) {
    const int k = col[rs + t], dk = diagpos[k], ke = rowptr[k + 1], o = ppofs[rs + t];
    for (int q = dk + 1 + gl; q < ke; q += 16) {
      const int j = col[q];
      int pos = 255;
      if (j < m) {
        int lo = rs + t + 1, hi = re - 1;
        while (lo <= hi) {
          const int mid = lo + ((hi - lo) >> 1);
          const int cc = col[mid];
          if (cc == j) {
            pos = mid - rs;
            break;
          }
          if (cc < j) lo = mid + 1; else hi = mid - 1;
        }
      }
      ppos[o + (q - dk - 1)] = (unsigned char)pos;
    }
  }
}
struct IluPlanned {
  int b0, b1;
  double u0, u1;
};
__global__ __launch_bounds__(256) void k_ilu_factor_plan(const int* __restrict__ rows, int nrows, const int* __restrict__ rowptr, const int* __restrict__ col,
                                                         const int* __restrict__ diagpos, const int* __restrict__ ppofs,
                                                         const unsigned char* __restrict__ ppos, double* lu, int maxrow, double zeropivot,
                                                         int* __restrict__ flag) {
  extern __shared__ double tri_rows[];             // per group: the row's values, the pivots' u_kk; then the pivots' diagonal positions, ends and plan offsets
  const int rr = (blockIdx.x * 256 + threadIdx.x) >> 4, gl = threadIdx.x & 15, g = threadIdx.x >> 4;
  if (rr >= nrows) return;
  double* w = tri_rows + (size_t)g * maxrow;
  double* pu = tri_rows + (size_t)(16 + g) * maxrow;
  int* ibase = reinterpret_cast<int*>(tri_rows + (size_t)32 * maxrow);
  int* pd = ibase + (size_t)g * maxrow;
  int* pe = ibase + (size_t)(16 + g) * maxrow;
  int* po = ibase + (size_t)(32 + g) * maxrow;
  const int i = rows[rr];
  const int rs = rowptr[i], re = rowptr[i + 1], di = diagpos[i];
  const int nlow = max(di - rs, 0);                 // entries left of the diagonal = pivots (the columns are sorted)
  for (int p = rs + gl; p < re; p += 16) w[p - rs] = lu[p];
  for (int t = gl; t < nlow; t += 16) {
    const int k = col[rs + t], dk = diagpos[k];
    pd[t] = dk;
    pe[t] = rowptr[k + 1];
    po[t] = ppofs[rs + t];
    pu[t] = 1.0 / lu[dk];      // the inverted pivot (PETSc keeps the diagonal of the factor inverted and multiplies)
  }
  tri_group_sync();
  auto fetch = [&](int t, IluPlanned& v) {          // the lane's first two entries right of pivot t's diagonal: value and place in this row
    v.b0 = v.b1 = 255;
    v.u0 = v.u1 = 0.0;
    if (t < nlow) {
      const int q0 = pd[t] + 1 + gl, ke = pe[t], o = po[t] + gl;
      if (q0 < ke) {
        v.b0 = ppos[o];
        v.u0 = lu[q0];
      }
      if (q0 + 16 < ke) {
        v.b1 = ppos[o + 16];
        v.u1 = lu[q0 + 16];
      }
    }
  };
  IluPlanned v0, v1, v2;
  fetch(0, v0);
  fetch(1, v1);
  fetch(2, v2);
#define ILU_STEP(T, V)                                                                     \
  if ((T) < nlow) {                                                                        \
    const double lik = w[T] * pu[T];                                                       \
    tri_group_sync();                                                                      \
    if (gl == 0) w[T] = lik;                                                               \
    if (V.b0 != 255) w[V.b0] -= lik * V.u0;                                                \
    if (V.b1 != 255) w[V.b1] -= lik * V.u1;                                                \
    for (int q = pd[T] + 33 + gl; q < pe[T]; q += 16) {                                    \
      const int b = ppos[po[T] + (q - pd[T] - 1)];                                         \
      if (b != 255) w[b] -= lik * lu[q];                                                   \
    }                                                                                      \
    tri_group_sync();                                                                      \
    fetch((T) + 3, V);                                                                     \
  }
  for (int t = 0; t < nlow; t += 3) {
    ILU_STEP(t, v0)
    ILU_STEP(t + 1, v1)
    ILU_STEP(t + 2, v2)
  }
#undef ILU_STEP
  for (int p = rs + gl; p < re; p += 16) lu[p] = w[p - rs];
  if (gl == 0) {
    double rsum = 0.0;
    for (int p = (di < 0 ? rs : di + 1); p < re; p++) rsum += fabs(w[p - rs]);   // the U part of the row without its diagonal (PETSc: sctx.rs)
    if (di < 0 || !(fabs(w[di - rs]) > zeropivot * rsum)) atomicOr(flag, 1);
  }
}

__global__ __launch_bounds__(256) void k_ilu_shift(double* __restrict__ lu, const int* __restrict__ diagpos, int m, double shift) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < m && diagpos[i] >= 0) lu[diagpos[i]] += shift;
}

int fh_tri_ilu_factor(fh_tri_t T, fh_mat_t A) {
  fh_ctx_t c = A->ctx;
  for (int i = 0; i < A->m; i++) FH_REQUIRE(T->h_diagpos[i] >= 0, "ILU(0): row %d has no diagonal entry in the pattern", i);
  if (!T->d_lu) FH_CHECK_HIP(hipMalloc(&T->d_lu, ((size_t)A->nnz + 2) * sizeof(double)));
  if (!T->d_flag) FH_CHECK_HIP(hipMalloc(&T->d_flag, sizeof(int)));
  const int nf = (int)T->fptr.size() - 1;
  const int maxrow = std::max(A->max_row, 1);
  // the row being eliminated lives in LDS, sixteen rows per workgroup: 128 bytes per entry of the longest row.  512 entries fit the 64 kB a kernel gets
  // without asking; up to 1 200 (stacked three-dimensional systems after a Galerkin product) the kernel asks for more of the CU's 160 kB
  FH_REQUIRE(maxrow <= 1200, "ILU(0): a row with %d entries (at most 1200 are served)", maxrow);
  // what the device grants a workgroup on request (160 kB on MI355X): the columns join the values in LDS only where both fit, and a row too long even for the
  // values alone is refused with the limit in the message
  int lds_max = 0, lds_optin = 0;
  if (hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, c->device) != hipSuccess) lds_max = 0;
  if (hipDeviceGetAttribute(&lds_optin, hipDeviceAttributeSharedMemPerBlockOptin, c->device) != hipSuccess) lds_optin = 0;
  (void)hipGetLastError();
  lds_max = std::max(std::max(lds_max, lds_optin), 64 * 1024);
  const bool lcol = maxrow <= 800 && (size_t)16 * maxrow * (sizeof(double) + sizeof(int)) <= (size_t)lds_max;
  // pivot descriptors in LDS too (k_ilu_factor_ahead): 28 bytes per entry of the longest row and group
  const bool ahead = c->ilu_ahead && (size_t)16 * maxrow * 28 <= (size_t)lds_max;
  if (ahead && c->ilu_ahead >= 2 && T->plan_state == 0) {       // the elimination plan of this pattern, once
    T->plan_state = -1;
    if (maxrow <= 254) {
      const std::vector<int>& hc = fh_hcol(A);
      std::vector<int> pofs((size_t)std::max(A->nnz, 1), 0);
      int64_t total = 0;
      for (int i = 0; i < A->m; i++)
        for (int p = A->h_rowptr[i]; p < T->h_diagpos[i]; p++) {
          const int k = hc[p];
          pofs[p] = (int)std::min<int64_t>(total, 2147483647);
          total += A->h_rowptr[k + 1] - T->h_diagpos[k] - 1;
        }
      if (total < 2147483647ll) {
        FH_CHECK_HIP(hipMalloc(&T->d_ppofs, pofs.size() * sizeof(int)));
        FH_CHECK_HIP(hipMemcpyAsync(T->d_ppofs, pofs.data(), pofs.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
        FH_CHECK_HIP(hipMalloc(&T->d_ppos, (size_t)std::max<int64_t>(total, 1) + 64));
        hipLaunchKernelGGL(k_ilu_plan, dim3(fh_div_up((int64_t)A->m * 16, 256)), dim3(256), 0, c->stream, A->m, A->d_rowptr, A->d_col, T->d_diagpos, T->d_ppofs,
                           T->d_ppos);
        FH_CHECK_HIP(hipGetLastError());
        FH_CHECK_HIP(hipStreamSynchronize(c->stream));
        T->plan_state = 1;
        FH_TRACE("fh_tri_ilu_factor: elimination plan of %d rows, %lld bytes", A->m, (long long)total);
      }
    }
  }
  const bool planned = ahead && c->ilu_ahead >= 2 && T->plan_state == 1;
  const size_t lds = ahead ? (size_t)16 * maxrow * 28 : (size_t)16 * maxrow * (sizeof(double) + (lcol ? sizeof(int) : 0));
  FH_REQUIRE(lds <= (size_t)lds_max, "ILU(0): a row with %d entries needs %zu bytes of LDS per workgroup, the device grants %d", maxrow, lds, lds_max);
  if (lds > 64 * 1024)
    FH_CHECK_HIP(hipFuncSetAttribute(planned ? reinterpret_cast<const void*>(&k_ilu_factor_plan) : ahead ? reinterpret_cast<const void*>(&k_ilu_factor_ahead) : lcol ? reinterpret_cast<const void*>(&k_ilu_factor<true>) : reinterpret_cast<const void*>(&k_ilu_factor<false>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  double shift = 0.0;
  for (int attempt = 0; attempt < 40; attempt++) {
    FH_CHECK_HIP(hipMemcpyAsync(T->d_lu, A->d_val, (size_t)A->nnz * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    if (shift != 0.0) hipLaunchKernelGGL(k_ilu_shift, dim3(fh_div_up(A->m, 256)), dim3(256), 0, c->stream, T->d_lu, T->d_diagpos, A->m, shift);
    FH_CHECK_HIP(hipMemsetAsync(T->d_flag, 0, sizeof(int), c->stream));
    for (int l = 0; l < nf; l++) {
      const int n = T->fptr[l + 1] - T->fptr[l];
      if (planned)
        hipLaunchKernelGGL(k_ilu_factor_plan, dim3(fh_div_up((int64_t)n * 16, 256)), dim3(256), lds, c->stream, T->d_frows + T->fptr[l], n, A->d_rowptr, A->d_col,
                           T->d_diagpos, T->d_ppofs, T->d_ppos, T->d_lu, maxrow, 1e-16, T->d_flag);
      else if (ahead)
        hipLaunchKernelGGL(k_ilu_factor_ahead, dim3(fh_div_up((int64_t)n * 16, 256)), dim3(256), lds, c->stream, T->d_frows + T->fptr[l], n, A->d_rowptr,
                           A->d_col, T->d_diagpos, T->d_lu, A->m, maxrow, 1e-16, T->d_flag);
      else if (lcol)
        hipLaunchKernelGGL(k_ilu_factor<true>, dim3(fh_div_up((int64_t)n * 16, 256)), dim3(256), lds, c->stream, T->d_frows + T->fptr[l], n, A->d_rowptr,
                           A->d_col, T->d_diagpos, T->d_lu, A->m, maxrow, 1e-16, T->d_flag);
      else
        hipLaunchKernelGGL(k_ilu_factor<false>, dim3(fh_div_up((int64_t)n * 16, 256)), dim3(256), lds, c->stream, T->d_frows + T->fptr[l], n, A->d_rowptr,
                           A->d_col, T->d_diagpos, T->d_lu, A->m, maxrow, 1e-16, T->d_flag);
    }
    FH_CHECK_HIP(hipGetLastError());
    int h = 0;
    FH_CHECK_HIP(hipMemcpyAsync(&h, T->d_flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    FH_CHECK_HIP(hipStreamSynchronize(c->stream));
    if (!h) {
      T->shift = shift;
      return 0;
    }
    shift = (shift == 0.0) ? 100.0 * 2.220446049250313e-16 : 2.0 * shift;      // MAT_SHIFT_NONZERO: restart with A + shift I
  }
  fh_set_error("ILU(0): zero pivot even after 40 diagonal shifts");
  return 2;
}

// y_i = r_i - sum_{j < i} l_ij y_j
__global__ __launch_bounds__(256) void k_ilu_lsolve(const int* __restrict__ rows, int nrows, const int* __restrict__ rowptr, const int* __restrict__ col,
                                                    const double* __restrict__ lu, const double* __restrict__ r, double* z) {
  const int rr = (blockIdx.x * 256 + threadIdx.x) >> 4, gl = threadIdx.x & 15;
  const bool live = rr < nrows;
  const int i = live ? rows[rr] : 0;
  double acc = 0.0;
  if (live)
    for (int k = rowptr[i] + gl; k < rowptr[i + 1]; k += 16) {
      const int j = col[k];
      if (j < i) acc += lu[k] * z[j];
    }
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (live && gl == 0) z[i] = r[i] - acc;
}

// z_i = (y_i - sum_{j > i, local} u_ij z_j) / u_ii
__global__ __launch_bounds__(256) void k_ilu_usolve(const int* __restrict__ rows, int nrows, const int* __restrict__ rowptr, const int* __restrict__ col,
                                                    const int* __restrict__ diagpos, const double* __restrict__ lu, double* z, int m) {
  const int rr = (blockIdx.x * 256 + threadIdx.x) >> 4, gl = threadIdx.x & 15;
  const bool live = rr < nrows;
  const int i = live ? rows[rr] : 0;
  double acc = 0.0;
  if (live)
    for (int k = rowptr[i] + gl; k < rowptr[i + 1]; k += 16) {
      const int j = col[k];
      if (j > i && j < m) acc += lu[k] * z[j];
    }
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (live && gl == 0) z[i] = (z[i] - acc) * (1.0 / lu[diagpos[i]]);      // as the run kernel: times the inverted diagonal
}

int fh_tri_ilu_apply(fh_tri_t T, fh_mat_t A, const double* r, double* z) {
  hipStream_t s = A->ctx->stream;
  const int nf = (int)T->fptr.size() - 1, nb = (int)T->bptr.size() - 1;
  (void)nf; (void)nb;
  TriRun P = {nullptr, nullptr, A->d_rowptr, nullptr, T->d_diagpos, nullptr, T->d_lu, nullptr, r, nullptr, z, nullptr, 0, 0, A->m, nullptr, T->d_prog, 0u, tri_ahead(), 1, tri_dbg()};
  for (size_t q = 0; q < T->fseg.size(); q += 3) {
    const int l = T->fseg[q];
    if (T->fseg[q + 2]) {
      P.rows = T->d_frows; P.lptr = T->d_fptr; P.src = T->d_fsrc; P.lv = reinterpret_cast<const int4*>(T->d_flv); P.l0 = l; P.nl = T->fseg[q + 1];
#if TRI_STAMP
      P.stamp = (g_tri_stamp && P.nl > 300) ? g_tri_stamp : nullptr;
#endif
      P.id = ++tri_launch_id; P.nblk = tri_blocks(P.nl);
      if (T->run_pf == 2) hipLaunchKernelGGL((k_tri_run<2, 2>), dim3(P.nblk), dim3(1024), 0, s, P);
      else if (T->run_pf == 3) hipLaunchKernelGGL((k_tri_run<2, 3>), dim3(P.nblk), dim3(1024), 0, s, P);
      else hipLaunchKernelGGL((k_tri_run<2, 4>), dim3(P.nblk), dim3(1024), 0, s, P);
    } else {
      const int n = T->fptr[l + 1] - T->fptr[l];
      hipLaunchKernelGGL(k_ilu_lsolve, dim3(fh_div_up((int64_t)n * 16, 256)), dim3(256), 0, s, T->d_frows + T->fptr[l], n, A->d_rowptr, A->d_col, T->d_lu, r, z);
    }
  }
  for (size_t q = 0; q < T->bseg.size(); q += 3) {
    const int l = T->bseg[q];
    if (T->bseg[q + 2]) {
      P.rows = T->d_brows; P.lptr = T->d_bptr; P.src = T->d_bsrc; P.lv = reinterpret_cast<const int4*>(T->d_blv); P.l0 = l; P.nl = T->bseg[q + 1];
      P.id = ++tri_launch_id; P.nblk = tri_blocks(P.nl);
      if (T->run_pb == 2) hipLaunchKernelGGL((k_tri_run<3, 2>), dim3(P.nblk), dim3(1024), 0, s, P);
      else if (T->run_pb == 3) hipLaunchKernelGGL((k_tri_run<3, 3>), dim3(P.nblk), dim3(1024), 0, s, P);
      else hipLaunchKernelGGL((k_tri_run<3, 4>), dim3(P.nblk), dim3(1024), 0, s, P);
    } else {
      const int n = T->bptr[l + 1] - T->bptr[l];
      hipLaunchKernelGGL(k_ilu_usolve, dim3(fh_div_up((int64_t)n * 16, 256)), dim3(256), 0, s, T->d_brows + T->bptr[l], n, A->d_rowptr, A->d_col,
                         T->d_diagpos, T->d_lu, z, A->m);
    }
  }
  FH_CHECK_HIP(hipGetLastError());
  return 0;
}
