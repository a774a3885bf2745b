// level-scheduled natural-order sweeps (fh_trisolve.hip): symmetric Gauss-Seidel (PCSOR) and ILU(0) (PCILU)
#pragma once
#include "fh_internal.h"

struct fh_tri_s {
  int m = 0;
  uint64_t A_uid = 0;
  std::vector<int> fptr, bptr;          // level pointers of the forward (rows j < i first) / backward schedule
  std::vector<int> h_diagpos;
  int *d_frows = nullptr, *d_brows = nullptr, *d_diagpos = nullptr;
  // runs of consecutive SMALL levels go to one workgroup each (fh_trisolve.hip: k_tri_run): level pointers on the device, and per sweep direction the
  // segments {first level, number of levels, 1 = run of small levels / 0 = one large level}
  int *d_fptr = nullptr, *d_bptr = nullptr;
  std::vector<int> fseg, bseg;
  // where an entry's operand z_j comes from inside a run (round 6): >= 0 the column j itself (global memory), < 0: -(rank + 1) of row j in the level just before --
  // the run kernel keeps the values of the previous level in LDS, so the one link of a level's dependency chain that cannot be loaded ahead is an LDS read
  int *d_fsrc = nullptr, *d_bsrc = nullptr;
  int *d_flv = nullptr, *d_blv = nullptr;      // per row in level order: {row, first entry, end, diagonal position} (one 16-byte load instead of a chain of three)
  // elimination plan of the ILU(0) factorisation (built at the first factorisation): for every entry left of a diagonal (a pivot of its row) where the pivot row's
  // entries right of ITS diagonal land in the row -- one byte each (255: nowhere), d_ppofs[p] = first byte of the pivot at entry p
  int* d_ppofs = nullptr;
  unsigned char* d_ppos = nullptr;
  int plan_state = 0;                    // 0 not tried, 1 built, -1 not served (rows of more than 254 entries, or more than 2^31 bytes)
  unsigned long long* d_prog = nullptr;  // progress word of the run kernel's main workgroup, read by its prefetching workgroup (fh_trisolve.hip)
  int run_pf = 2, run_pb = 2;           // register slots per lane of the run kernel, forward / backward sweep: 2 while 90 % of the lower / upper triangles have at most 32 entries, else 4
  double* d_lu = nullptr;               // ILU(0) factors on A's pattern: strict lower part = L (unit diagonal), rest = U
  int* d_flag = nullptr;
  double* d_t = nullptr;                // symmetric sweep: t = r - L z of the forward half, read by the backward half
  double shift = 0.0;                   // diagonal shift the last factorisation needed (MAT_SHIFT_NONZERO)
};
typedef fh_tri_s* fh_tri_t;

int fh_tri_create(fh_mat_t A, fh_tri_t* out);
void fh_tri_destroy(fh_tri_t T);
int fh_tri_ssor_apply(fh_tri_t T, fh_mat_t A, const double* dinv, const double* r, double* z);
int fh_tri_ilu_factor(fh_tri_t T, fh_mat_t A);
int fh_tri_ilu_apply(fh_tri_t T, fh_mat_t A, const double* r, double* z);
