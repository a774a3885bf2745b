// Internal declarations shared by the translation units of libfemus_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>
#include <algorithm>
#include <vector>
#include "../../include/femus_hip.h"

void fh_set_error(const char* fmt, ...);

#define FH_CHECK_HIP(expr)                                                                        \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess) {                                                                       \
      fh_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e));     \
      return 1;                                                                                   \
    }                                                                                             \
  } while (0)

#define FH_REQUIRE(cond, ...)                                                                     \
  do {                                                                                            \
    if (!(cond)) {                                                                                \
      fh_set_error(__VA_ARGS__);                                                                  \
      return 2;                                                                                   \
    }                                                                                             \
  } while (0)

#define FH_TRY(expr)                                                                              \
  do {                                                                                            \
    int _r = (expr);                                                                              \
    if (_r) return _r;                                                                            \
  } while (0)

// no C++ exception leaves an entry point whose host side allocates in proportion to the problem (meshes, patterns, plans, symbolic
// products): std::bad_alloc and friends become an error code + fh_last_error, as every other failure of the C ABI
#define FH_GUARD_BEGIN try {
#define FH_GUARD_END(who)                                                        \
  }                                                                              \
  catch (const std::bad_alloc&) {                                                \
    fh_set_error("%s: out of host memory", who);                                 \
    return 3;                                                                    \
  }                                                                              \
  catch (const std::exception& e) {                                              \
    fh_set_error("%s: %s", who, e.what());                                       \
    return 3;                                                                    \
  }

// setup diagnostics: FEMUS_HIP_TRACE=1 prints the stages of the (host-side) setup calls with wall-clock times to stderr
#define FH_TRACE(...)                                                                             \
  do {                                                                                            \
    if (fh_trace_on()) fh_trace_print(__VA_ARGS__);                                               \
  } while (0)
bool fh_trace_on();
void fh_trace_print(const char* fmt, ...);

// copy of a whole host vector into a caller's array (an empty vector may have a null data(): memcpy with a null source is undefined)
template <class T>
static inline void fh_copy_out(T* dst, const std::vector<T>& v) {
  if (!v.empty()) std::copy(v.begin(), v.end(), dst);
}

struct fh_ctx_s {
  int device = 0;
  hipStream_t stream = nullptr;       // compute stream
  hipStream_t comm_stream = nullptr;  // halo / collective stream
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_join = nullptr;
  int num_cu = 256;
  // reduction scratch
  double* d_red = nullptr;            // device partials
  double* h_red = nullptr;            // pinned host
  size_t red_cap = 0;
  // options
  int spmv_tile = 2048;               // nnz per row block (LDS tile)
  int spmv_xcd_remap = 32;            // XCD-aware row-block order: 0 none, 1 one contiguous eighth per XCD, n > 1 interleaved chunks of n blocks
  int spmv_kernel = 3;                // 0: csr-stream (workgroup tiles), 1: csr-vector, 2: csr-stream (wave tiles), 3: csr-stream with LDS-staged x
  int spmv_threads = 256;             // kernel 3 workgroup size (128 or 256)
  int spmv_share = 1;                 // kernel 3: x tile and products share one LDS buffer
  int spmv_nt = 0;                    // non-temporal matrix stream
  int assemble_emap = 1;
  int asm_debug = 0;
  int debug_poison = 0;              // work buffers of the multigrid / Krylov solvers start as NaN instead of zero (tests)
  int assemble_mfma = 12;            // HEX27/Q2, 64 Gauss points: element matrices on the FP64 matrix cores, value = waves per workgroup (0 = off)
  int assemble_rows_nt = 1;          // row pass: bit 0 = non-temporal loads of the element rows (read once: 1.372 -> 1.361 ms per assembly), bit 1 = non-temporal stores of the matrix values (no gain)
  int assemble_sf_grid = 1;          // workgroups of the sum-factorised element kernel per resident slot (1 = persistent grid; > 1: shorter workgroups, the dispatcher balances them)
  int assemble_sf = 8;               // HEX27/Q2, 64 Gauss points, tensor-product tables: element matrices by sum factorisation on the vector ALU, value = waves per workgroup (0 = off: matrix-core kernel)
  int assemble_sumfac = 1;           // matrix-core element kernel: map Jacobian by sum factorisation (tensor-product tables)
  int assemble_rows2 = 1;            // row pass: two rows per 32-lane group when no row has more than 128 entries
  int assemble_kpad = 1;             // HEX27/Q2 two-pass assembly: element rows padded to 32 doubles (whole 64-byte lines per row)
  int assemble_sym = 1;              // symmetric-tile HEX27/Q2 element kernel (2 elements per wave)
  int assemble_two_pass = 1;         // 1: element matrices + row gather (default), 0: coloured scatter
  int assemble_fused = 1;            // HEX27/Q2 meshes whose elements come in sibling groups of eight: fused cluster assembly (rows complete inside a group go straight to the CSR arrays)
  int ilu_ahead = 2;                 // ILU(0) factorisation with the pivot rows asked for ahead of the elimination chain: 2 with the positions from a plan built once per pattern (k_ilu_factor_plan), 1 searched (k_ilu_factor_ahead); 0: the round-5 kernel (A/B of the bitwise test)
  int tri_runs = 1;                  // natural-order sweeps: runs of small levels in one workgroup (fh_trisolve.hip k_tri_run); 0: one launch per level (the A/B of the bitwise test)
  int assemble_carry = -1;           // fused cluster assembly: rows whose elements all lie in one SUPER-cluster of 8^k consecutive clusters (k = value / 3) are accumulated in the CSR array
                                     // itself by the one workgroup that walks the super-cluster (store / load-add-store, ascending cluster order), not through the partial-row buffer;
                                     // -1 = 64 or 8 clusters where every workgroup gets at least two super-clusters, 0 = off, 3 / 6 = forced (read when an assembler is created)
  int assemble_affine = 0;           // opt-in: affine HEX27/Q2 elements through precomputed reference matrices instead of quadrature
  int gj_symmetric = 1;              // coarse dense inverse: symmetric sweep on the upper block triangle when the operator is symmetric
  int galerkin_mfma = 1;             // element-wise Galerkin product on the FP64 matrix cores (0: sparse child tables on the vector ALU)
  int patch_invert_lds = 1;          // block smoother setup: patches of <= 96 dofs are inverted by one wave each in LDS (0: workgroup kernel on global memory)
  int coarse_nd = 8;                 // coarsest level: interior blocks of the nested dissection of the coupled unknowns (block inverses beside each other + separator Schur complement); needs coordinates (fh_mg_set_coarse_coords) and a symmetric operator; 0 / 1: one dense inverse
  int coarse_nd_streams = 0;         // ... the block inverses: 0 = one launch per step for all blocks (block index as a grid dimension), 1 = one stream per block, 2 = one after the other (measurements)
  int coarse_nd_min = 1024;          // ... only from this many coupled unknowns on
  int vanka_fused = 1;               // block smoothers (Vanka / PCASM): 1 = a colour in one launch, every patch forms the residual of its own rows; 0 = level residual (SpMV) + patch launch
  int galerkin_macro = 1;            // element-wise Galerkin product after a FUSED assembly: 1 = from the macro rows it left behind (k_galerkin_macro), 0 = re-create the element rows
  int gmres_device = 1;              // outer GMRES of fh_mg_solve: 1 = Hessenberg / rotations / convergence test on the device, one read-back per iteration; 0 = driven from the host
  int coarse_direct = 1;             // coarsest level: 1 = the sparse exact solve (fh_direct.hip) when more than coarse_direct_min unknowns are coupled, 2 = always, 0 = never (dense inverse, <= 16384)
  int coarse_direct_min = 8192;
  int coarse_reduce = 1;             // coarsest level: unknowns coupled to nothing (Dirichlet rows) are solved by their diagonal, the dense inverse holds the rest
  int vanka_persistent = 0;          // block smoother: all colours of a sweep in one launch with device-wide barriers (1: arrival counter, 2: flag per
                                     // workgroup); measured 2.7x SLOWER than residual SpMV + patch kernel per colour, kept as an option
  int gj_block = 128;                // symmetric coarse inverse: pivot blocks of 128 with rank-128 matrix-core updates (0: the 32-wide pivoted sweep)
  int gj_mfma = 1;                   // coarse dense inverse: rank-NB updates on the FP64 matrix cores
  int use_graph = 1;
  int mg_reuse_graph = 1;            // a repeated fh_mg_setup of an unchanged hierarchy keeps the captured cycle
  int opt_gen = 0;                   // bumped by every fh_set_option (captured launches depend on the options)
  int device_setup = 1;              // prolongators are built on the device (0: host loops; identical matrices)
  int spgemm_device_symbolic = 1;    // patterns of sparse products are built on the device (0: host builder)
  int spgemm_slot_map = 1;           // Galerkin products stream a precomputed slot map instead of searching
  int halo_overlap = 1;              // distributed operators: rows without ghost columns run while the ghost exchange is in flight
  int halo_profile = 0;              // time every exchange and the part of it the compute stream waited for (fh_halo_stats; synchronises)
  int halo_self_rccl = 0;            // one-rank plans exchange with themselves through RCCL (hardware preflight on a single GPU)
};

// pinned staging rings of the per-element add path (fh_stage.hip); owned by the matrix / vector that staged
struct fh_stage_s;
void fh_stage_free(fh_stage_s* s);

struct fh_vec_s {
  fh_ctx_t ctx = nullptr;
  fh_stage_s* stage = nullptr;
  int n_global = 0, n_local = 0, first_local = 0, nghost = 0;
  double* d = nullptr;                // [n_local + nghost]
  std::vector<int> ghost_idx;         // global indices of ghosts (host copy)
  int* d_ghost_idx = nullptr;
  // staged ADDS to ghost entries (VecSetValues(ADD_VALUES) on an off-process index, PetscVector.cpp:131-153) collect here, not in the ghost
  // tail: fh_halo_reverse_add ships them to the owners, which add them to their owned entries (VecAssemblyBegin/End)
  double* d_gacc = nullptr;           // [nghost], allocated by the first such add
  bool gacc_dirty = false;
};

struct fh_mat_s {
  fh_ctx_t ctx = nullptr;
  fh_stage_s* stage = nullptr;
  uint64_t uid = 0;                   // unique per created matrix (never reused, unlike the address): keys caches built from the pattern
  uint64_t val_gen = 0;               // bumped by every writer of the values except the Dirichlet-row replacement (fh_mat_zero_rows*): whoever keeps something derived from
                                      // the values of one moment (the macro rows of a fused assembly, read back by fh_assembler_galerkin) compares it
  int m = 0, n = 0, nnz = 0;
  int* d_rowptr = nullptr;
  int* d_col = nullptr;
  double* d_val = nullptr;
  std::vector<int> h_rowptr, h_col;   // host copy of the pattern (setup-time integer work)
  // CSR-stream row blocks
  int tile = 0;
  int tile_kernel = 0;
  int nblk = 0;
  int* d_rowblk = nullptr;
  std::vector<int> h_rowblk;
  // tile-local column compaction (spmv_kernel 3): unique columns per row block + 16-bit local indices
  int* d_uptr = nullptr;
  int* d_ucols = nullptr;
  unsigned short* d_lcol = nullptr;
  int* d_tile_s = nullptr;             // first non-zero of every row block (persistent pipelined kernel)
  int* d_blkinfo = nullptr;            // 8 ints per row block: r0, r1, s, e, u0, nu (one descriptor load instead of a pointer chain)
  int lx_tile = 0;
  int64_t nu_total = 0;                // sum over the row blocks of their distinct columns (entries of d_ucols)
  int max_row = 0;
  // interior / interface split of the row blocks for operators over [owned | ghost] columns (fh_dev_spmv_part):
  // descriptors permuted so that blocks without a ghost column come first
  int split_nown = -1;                 // number of owned columns the split was built for (-1: none)
  int split_tile = 0;
  int nblk_int = 0;
  int* d_blkinfo_split = nullptr;
  int* d_diagpos = nullptr;            // position of every row's diagonal entry (-1: none), built by the first fh_dev_get_diag
  // cached explicit transpose for matrix_mult_transpose
  fh_mat_t At = nullptr;
  int* d_tperm = nullptr;             // At.val[k] = val[tperm[k]]
  bool at_valid = false;
  // attached reusable product plan (fh_mat_ptap)
  void* plan = nullptr;
  void (*plan_destroy)(void*) = nullptr;
};

// descriptor of one dense symmetric matrix of the batched 128-block inverse (fh_mg.hip: k_inv_*_b; fh_inv_sym_batched)
struct InvDesc {
  double* D;             // n x n, leading dimension n; replaced by its inverse
  int n;
  double *PT, *RT, *Dv0, *Dv1;   // work: panels 2 x (n x 128), pivot-block inverses 2 x 2 x 128 x 128 (fh_inv_work_doubles(n) doubles from PT)
  int* flg;              // two ints; flg[1] != 0: a pivot block had no usable diagonal pivot
  int off;               // first unknown of the block in the dissected ordering (k_nd_w)
};
size_t fh_inv_work_doubles(int n);
int fh_inv_sym_batched(fh_ctx_t c, const InvDesc* d_desc, int k, int nmax);

// host copy of the column indices: matrices whose pattern was built on the device (fh_mat_create_from_elements) fetch it at the first host use
int fh_mat_fetch_host_cols(fh_mat_t A);
int fh_mat_alloc_device_pattern(fh_ctx_t c, int m, int n, std::vector<int>&& rp, fh_mat_t* out);   // columns left to the caller's kernels
static inline const std::vector<int>& fh_hcol(fh_mat_t A) {
  if (A->h_col.size() != (size_t)A->nnz) fh_mat_fetch_host_cols(A);
  return A->h_col;
}

// device-resident copy of a mesh (fh_meshdev.hip): made by fh_mesh_refine_device (or uploaded at the first use), read by the set-up calls that
// take a mesh -- the element table of a 64^3 level is 28 MB, its coordinates 51 MB, and every one of those calls used to upload them again
struct fh_mesh_dev {
  fh_ctx_t ctx = nullptr;
  int nel = 0, nnode = 0, nloc = 0, dim = 0, nf = 0;
  int* d_elem_dof = nullptr;      // [nel * nloc]
  double* d_coords = nullptr;     // [nnode * dim]
  int* d_face_flag = nullptr;     // [nel * nf]
  int* d_elem_level = nullptr;    // [nel]
  int* d_child = nullptr;         // [nel * nch], set when this mesh was refined on the device
  char* d_refined = nullptr;      // [nel]
};
struct fh_refine_tables {         // reference-element tables of one geometry (fh_mesh.cpp: refine_tables)
  int nv, ne, nc, nch, nf, dim;
  int f2c[8][8], edge_v[12][2], face_v[6][4], face_diag[6][4];
  unsigned char cof[6][8];        // child j touches face f
  std::vector<double> EP;         // element prolongator [nch * nc][nc]
  std::vector<int> cnt, nzk;      // non-zero weights of every row: how many, which columns
};
struct fh_refine_result {
  int nel = 0, nnode = 0, own[3] = {0, 0, 0};
  std::vector<int> elem_dof, face_flag, elem_level, child;
  std::vector<char> refined;
  std::vector<double> coords;
  fh_mesh_dev* dev = nullptr;
};
void fh_meshdev_free(fh_mesh_dev* d);
int fh_meshdev_upload(fh_ctx_t ctx, int nel, int nnode, int nloc, int dim, int nf, const int* elem_dof, const double* coords, const int* face_flag,
                      const int* elem_level, fh_mesh_dev** out);
int fh_meshdev_refine(fh_ctx_t ctx, const fh_refine_tables& tables, fh_mesh_dev* coarse, int level_c, const unsigned char* flags, fh_refine_result* out);
// the device copy of a mesh on this context (uploaded now if the mesh has none), its host arrays beside it
int fh_mesh_device(fh_ctx_t ctx, fh_mesh_t m, fh_mesh_dev** dev);

// kernels / helpers implemented across TUs
int fh_reserve_reduction(fh_ctx_t ctx, size_t ndoubles);
int fh_mat_build_rowblocks(fh_mat_t A, int tile);
int fh_mat_build_localcols(fh_mat_t A);
int fh_mat_refresh_transpose(fh_mat_t A);   // re-gather values into the cached transpose
int fh_dev_spmv(fh_mat_t A, const double* x, double* y, int mode, const double* b, const double* dinv, double omega);
// part 0: the row blocks that read no column >= n_own_cols (no ghost), part 1: the others; part 0 + part 1 = fh_dev_spmv
int fh_dev_spmv_part(fh_mat_t A, int n_own_cols, int part, const double* x, double* y, int mode, const double* b, const double* dinv, double omega);

// y = op(A, x) for an operator over [owned | ghost] columns: ghost exchange of x overlapped with the rows that need no ghost
int fh_dev_halo_spmv(fh_halo_t h, fh_mat_t A, double* x, int n_own, double* y, int mode, const double* b, const double* dinv, double omega,
                     bool prepacked = false);
void fh_halo_send_plan(fh_halo_t h, const int** send_idx, double** sendbuf, int* nsend);

static inline int fh_div_up(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
