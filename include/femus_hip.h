/*
 * femus_hip.h -- C-ABI of libfemus_hip.so: the MI355X (gfx950) backend for the FEMuS
 * assembly + geometric-multigrid hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.  The three adapter
 * classes in femus_amd/csrc/adapters/ (HipMatrix : SparseMatrix, HipVector : NumericVector,
 * LinearEquationSolverHip : LinearEquationSolver) are thin shells over these entry points; a FEMuS
 * maintainer binds them exactly the same way.
 *
 * Each entry point cites the reference interface it replaces (paths relative to the FEMuS tree;
 * "03_solvers/" abbreviates
 * src/08_algebra_dependent_on_Mesh_and_Solution_but_independent_of_Systems/03_solvers_with_preconditioner/).
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; fh_last_error() gives the message.
 *     (The reference aborts on error -- CHKERRABORT / abort(); the C++ adapters do the same on !=0.)
 *   - indices are 32-bit (PetscVector.hpp:536 asserts sizeof(PetscInt)==sizeof(int)); values are IEEE double.
 *   - host pointers unless the name says "_dev".  All device work is queued on the context's stream;
 *     functions that return numbers to the host synchronise that stream.
 *   - there is NO CPU fallback: without a usable HIP device fh_init fails and nothing else works.
 */
#ifndef FEMUS_HIP_H
#define FEMUS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fh_ctx_s* fh_ctx_t;
typedef struct fh_vec_s* fh_vec_t;
typedef struct fh_mat_s* fh_mat_t;
typedef struct fh_mg_s* fh_mg_t;
typedef struct fh_mesh_s* fh_mesh_t;
typedef struct fh_halo_s* fh_halo_t;
typedef struct fh_graph_s* fh_graph_t;

/* ---- context --------------------------------------------------------------------------------
 * replaces FemusInit (src/00_utils/00_application_initialization/FemusInit.cpp:46-74: PetscInitialize) */
int fh_device_count(int* n);                         /* visible HIP devices (0 when there is none or no driver) */
int fh_init(int device, fh_ctx_t* ctx);
int fh_finalize(fh_ctx_t ctx);
const char* fh_last_error(void);
const char* fh_version(void);
int fh_device_name(fh_ctx_t ctx, char* buf, int buflen);
int fh_sync(fh_ctx_t ctx);
void* fh_stream(fh_ctx_t ctx);                       /* hipStream_t of the compute stream */
/* HIP-event timing on the compute stream (bench.py: roofline.achieved is measured with these) */
/* measurement aid: launches the one-thread kernel k_phase_marker<id> (id 0 .. 15) on the compute stream; a kernel trace of the run can be cut
 * into phases at these names (profiles/summarize.py, bench.py's in-cycle sweep time) */
int fh_profile_marker(fh_ctx_t ctx, int id);
int fh_timer_start(fh_ctx_t ctx);
int fh_timer_stop(fh_ctx_t ctx, double* milliseconds);
/* Recorded launch sequences (hipGraph): the device-only calls between fh_graph_begin and fh_graph_end -- SpMV family, vector algebra,
 * fh_assemble_* after their first call; nothing that synchronises, allocates or copies to the host, and no fh_mg_* call (the cycle keeps
 * its own graph) -- are recorded instead of executed; fh_graph_launch replays them on the compute stream.  No counterpart in the
 * reference (PETSc issues every operation eagerly); bench.py times the fused sweep inside such a recording, as the cycle runs it.
 * A recording that contains a forbidden call ends in an error of fh_graph_end; the context stays usable (it may get a new compute stream:
 * a handle taken with fh_stream() earlier is stale then). */
int fh_graph_begin(fh_ctx_t ctx);
int fh_graph_end(fh_ctx_t ctx, fh_graph_t* graph);
int fh_graph_launch(fh_graph_t graph);
int fh_graph_destroy(fh_graph_t graph);
/* runtime tuning knobs (the reference honours the PETSc options DB, 03_solvers/LinearEquationSolverPetsc.cpp:251-254);
 * names (default): "spmv_tile" (2048), "spmv_xcd_remap" (32), "spmv_kernel" (3), "assemble_two_pass" (1), "assemble_emap" (1),
 * "assemble_mfma" (12: HEX27/Q2 element matrices on the FP64 matrix cores, value = waves per workgroup, 0 = vector kernel),
 * "assemble_kpad" (1: element rows of the two-pass buffer padded to 256 bytes; read when an assembler is created),
 * "assemble_sumfac" (1: map Jacobian by sum factorisation in that kernel), "assemble_sym" (1), "assemble_affine" (0, see fh_assembler_affine_count), "assemble_fused" (1, see fh_assembler_fused_info), "assemble_carry" (-1, see fh_assembler_carry_info), "ilu_ahead" (2: ILU(0) factorisation with the pivot rows loaded ahead of the elimination chain and the update positions from a plan built once per pattern; 1: positions searched; 0: the one-pivot look-ahead), "tri_runs" (1: natural-order sweeps take runs of small levels in one workgroup; 0: one launch per level -- read when a level solver is set up), "gj_mfma" (1: coarse dense inverse updates on the
 * matrix cores), "gj_symmetric" (1: symmetric sweep on the upper block triangle when the coarse operator is symmetric), "spgemm_slot_map" (1), "spgemm_device_symbolic" (1: patterns of sparse products on the device), "device_setup" (1: prolongators built on the device; 0: host loops, identical matrices), "use_graph" (1), "asm_debug" (0),
 * "debug_poison" (0; tests: work buffers of the solvers and the element-row buffers start as NaN bit patterns instead of zero),
 * "galerkin_macro" (1: fh_assembler_galerkin after a fused assembly reads the macro rows that assembly left behind), "vanka_fused" (1: block smoothers
 * run a colour / dependency level in one launch, every patch forming the residual of its own rows),
 * "gmres_device" (1: the outer GMRES of fh_mg_solve keeps its Hessenberg matrix, rotations and convergence test on the device and reads one status word back
 * per iteration; 0: the same solver driven from the host, two synchronisations per iteration),
 * "halo_overlap" (1: on distributed levels the rows without ghost columns are multiplied while the ghost exchange is in flight),
 * "halo_profile" (0; see fh_halo_stats), "halo_self_rccl" (0; 1: a one-rank exchange plan created afterwards gets an RCCL
 * communicator and sends to itself -- executes ncclCommInitRank / ncclSend / ncclRecv / ncclAllReduce on a single GPU).
 * Returns non-zero for unknown names */
int fh_set_option(fh_ctx_t ctx, const char* name, double value);

/* ---- vectors: NumericVector (src/03_algebra/00_vectors/NumericVector.hpp:51-353, PetscVector.cpp) ----
 * layout: [n_local owned entries | nghost ghost entries] ; ghost_idx are GLOBAL indices (PetscVector.hpp:515-569).
 * first_local is this rank's offset into the global numbering. */
int fh_vec_create(fh_ctx_t ctx, int n_global, int n_local, int first_local, const int* ghost_idx, int nghost, fh_vec_t* v);
int fh_vec_duplicate(fh_vec_t src, fh_vec_t* v);                 /* NumericVector::init(other) :129 */
int fh_vec_destroy(fh_vec_t v);
int fh_vec_size(fh_vec_t v, int* n_global, int* n_local, int* first_local, int* nghost);
int fh_vec_set_first(fh_vec_t v, int first_local);               /* ownership offset learnt after creation (HipVector::attach_halo) */
int fh_vec_zero(fh_vec_t v);                                     /* zero() :151 */
int fh_vec_fill(fh_vec_t v, double s);                           /* operator=(double) :153 */
int fh_vec_copy(fh_vec_t dst, fh_vec_t src);                     /* operator=(NumericVector) :155 */
int fh_vec_upload(fh_vec_t v, const double* host_owned);         /* operator=(std::vector) :157 (owned part) */
int fh_vec_download(fh_vec_t v, double* host_owned);             /* localize(std::vector) :308 (owned part) */
int fh_vec_set_values(fh_vec_t v, int n, const int* idx, const double* vals);   /* set(i,v) :146 / insert :160 */
int fh_vec_add_values(fh_vec_t v, int n, const int* idx, const double* vals);   /* add(i,v) :148, add_vector_blocked :265 */
int fh_vec_get_values(fh_vec_t v, int n, const int* idx, double* vals);         /* operator()(i) :224, get() :236 (owned+ghost) */
/* add_vector_blocked once per element (PetscVector.cpp:132-153: VecSetValues into the stash, applied by VecAssemblyBegin/End =
 * close(), PetscVector.hpp:595-612): fh_vec_stage_values appends to a pinned host ring, fh_vec_flush adds everything staged in
 * the order of the calls (see fh_mat_stage_block).  fh_vec_add_values = stage + flush. */
int fh_vec_stage_values(fh_vec_t v, int n, const int* idx, const double* vals);
int fh_vec_flush(fh_vec_t v);
int fh_vec_axpy(fh_vec_t y, double a, fh_vec_t x);               /* add(a,v) :262, +=, -= :243-245 */
int fh_vec_aypx(fh_vec_t y, double a, fh_vec_t x);               /* y = a*y + x (used by resid) */
int fh_vec_shift(fh_vec_t v, double s);                          /* add(s) :258 */
int fh_vec_scale(fh_vec_t v, double s);                          /* scale :294 */
int fh_vec_abs(fh_vec_t v);                                      /* abs :296 */
int fh_vec_pointwise_mult(fh_vec_t w, fh_vec_t a, fh_vec_t b);   /* pointwise_mult :328 */
int fh_vec_dot(fh_vec_t x, fh_vec_t y, double* out);             /* dot :298 (local part; see fh_halo_allreduce) */
int fh_vec_norm(fh_vec_t x, int kind, double* out);              /* kind 1: l1 :200, 2: l2 :202, 0: linfty :204 */
int fh_vec_reduce(fh_vec_t x, int kind, double* out);            /* kind 0: sum :197, 1: min :193, 2: max :195 */
double* fh_vec_dev_ptr(fh_vec_t v);                              /* device pointer (owned then ghosts) */
/* device CSR arrays of a matrix (interoperation with other device libraries; read-only use) */
int fh_mat_dev_ptrs(fh_mat_t A, const int** rowptr, const int** col, const double** val);

/* ---- matrices: SparseMatrix (src/03_algebra/01_matrices/SparseMatrix.hpp:48-282, PetscMatrix.cpp) ----
 * device CSR, sorted columns, fixed pattern once created.  rowptr[m+1], col[nnz], val[nnz] (val may be NULL = zeros).
 * Replaces init(m,n,m_l,n_l,d_nnz,o_nnz) :65-74 + the implicit pattern growth of MatSetValues: callers give
 * the pattern (fh_pattern_from_elements) instead of a per-row count. */
int fh_mat_create_csr(fh_ctx_t ctx, int m, int n, const int* rowptr, const int* col, const double* val, fh_mat_t* A);
int fh_mat_destroy(fh_mat_t A);                                  /* clear() :59 */
int fh_mat_size(fh_mat_t A, int* m, int* n, int* nnz);           /* m() :140, n() :143 */
int fh_mat_zero(fh_mat_t A);                                     /* zero() :96 -- keeps the pattern */
int fh_mat_set_values_csr(fh_mat_t A, const double* val);        /* bulk upload of all values (pattern order) */
int fh_mat_get_values_csr(fh_mat_t A, double* val);              /* bulk download */
int fh_mat_get_pattern(fh_mat_t A, int* rowptr, int* col);
/* add_matrix_blocked(vals, rows, cols) :165-171 (PetscMatrix.cpp:699-729): A[rows[i],cols[j]] += vals[i*ncol+j] */
int fh_mat_add_block(fh_mat_t A, int nrow, const int* rows, int ncol, const int* cols, const double* vals);
/* the same add as PETSc performs it for an application that calls add_matrix_blocked once per element
 * (applications/001_Poisson/main.cpp:283-609): MatSetValues keeps the block in a stash and MatAssemblyBegin/End -- close(),
 * PetscMatrix.hpp:237-244 -- applies it.  fh_mat_stage_block appends the block to a pinned host ring (no device call, no
 * synchronisation; a full ring leaves for the device asynchronously while the second ring fills), fh_mat_flush applies everything
 * staged and returns once it is on the device: one kernel per ring adds the blocks row by row IN THE ORDER OF THE CALLS, so the
 * result has the bits of adding the elements one after the other.  An entry outside the pattern with a non-zero value is reported
 * by the flush.  Between a stage and the flush the staged blocks are not part of the matrix: every other entry point sees the
 * values of the last flush.  fh_mat_add_block = stage + flush. */
int fh_mat_stage_block(fh_mat_t A, int nrow, const int* rows, int ncol, const int* cols, const double* vals);
int fh_mat_flush(fh_mat_t A);
int fh_mat_stage_stats(fh_mat_t A, int64_t* blocks_staged, int64_t* rings_sent);
/* insert_row(row, ncols, cols, vals) :162 -- INSERT semantics */
int fh_mat_insert_row(fh_mat_t A, int row, int ncols, const int* cols, const double* vals);
int fh_mat_get_row(fh_mat_t A, int row, int* ncols, int* cols, double* vals);    /* MatGetRowM :111 */
/* mat_zero_rows(index, diag) :229 (PetscMatrix.cpp:1073-1077) == MatZeroRows with MAT_KEEP_NONZERO_PATTERN;
 * also SetPenalty (03_solvers/LinearEquationSolverPetsc.cpp:428-436).  diag==0: plain zero rows. */
int fh_mat_zero_rows(fh_mat_t A, int n, const int* rows, double diag);
int fh_mat_zero_cols(fh_mat_t A, int n, const int* cols);        /* get_transpose+mat_zero_rows+get_transpose, LinearImplicitSystem.cpp:1101-1106 */
/* the Dirichlet list of a level kept on the device: BuildBdcIndex builds it once (_bdcIndexIsInitialized,
 * LinearEquationSolverPetsc.cpp:53-90) and SetPenalty (:428-436) / ZerosBoundaryResiduals (:417-424) reuse it at every assembly;
 * these two calls are asynchronous (no host traffic, no synchronisation) */
typedef struct fh_index_s* fh_index_t;
int fh_index_create(fh_ctx_t ctx, int n, const int* idx, fh_index_t* index);
int fh_index_destroy(fh_index_t index);
int fh_mat_zero_rows_index(fh_mat_t A, fh_index_t rows, double diag);
int fh_vec_set_index(fh_vec_t v, fh_index_t idx, double value);   /* v[idx] = value (owned entries) */
/* device-side gathers along such a list used as a map (-1 = no source, the target gets 0): dst.val[k] = src.val[map[k]] over the
 * non-zeros of dst, dst[i] = src[map[i]] for vectors.  They take a rank's owned rows out of an operator / residual that was
 * assembled and projected on its extended box (adaptive levels on several ranks) without host traffic. */
int fh_mat_gather_values(fh_mat_t dst, fh_mat_t src, fh_index_t map);
int fh_vec_gather(fh_vec_t dst, fh_vec_t src, fh_index_t map);
/* owned-row operators of a domain-decomposed level cut out of the operator of the rank's extended box on the device (the MPIAIJ
 * row ownership of PetscMatrix::init, PetscMatrix.cpp:162-203, with the ghost lists of LinearEquation.cpp:239-280):
 *   fh_mat_col_mask   mask[c] |= 1 for every column the listed rows touch (their halo)
 *   fh_mat_row_mask   rowmask[r] |= 1 for every row with an entry in a masked column
 *   fh_mat_restrict   dst = listed rows of src, columns renumbered by newcol[] (< 0: dropped, must hold zeros), values gathered;
 *                     *map feeds fh_mat_gather_values(dst, src, map) at every re-preparation
 *   fh_mat_restrict_check   largest |value| among the dropped entries of those rows (device reduction)
 *   fh_mat_value_map  map of an EXISTING dst pattern into src: entry (r, c) <- (src_row[r], src_col[c]) or -1 */
int fh_mat_col_mask(fh_mat_t A, int nrows, const int* rows, unsigned char* mask /* [n] */);
int fh_mat_row_mask(fh_mat_t A, const unsigned char* colmask /* [n] */, unsigned char* rowmask /* [m] */);
int fh_mat_restrict(fh_mat_t src, int nrows, const int* rows, const int* newcol /* [src n] */, int ncols_new, fh_mat_t* dst, fh_index_t* map);
int fh_mat_restrict_check(fh_mat_t src, int nrows, const int* rows, const int* newcol, double* max_dropped);
int fh_mat_value_map(fh_mat_t dst, fh_mat_t src, const int* src_row /* [dst m] or NULL */, const int* src_col /* [dst n] or NULL */, fh_index_t* map);
int fh_mat_get_diagonal(fh_mat_t A, fh_vec_t d);                 /* get_diagonal :224 */
int fh_mat_transpose(fh_mat_t A, fh_mat_t* At);                  /* get_transpose :227 (PetscMatrix.cpp:1031-1070) */
/* matrix_PtAP(P, A, reuse) :183 (PetscMatrix.cpp:733-751): C = P^T A P.  *C==NULL: symbolic+numeric; else numeric reuse */
int fh_mat_ptap(fh_mat_t P, fh_mat_t A, fh_mat_t* C);
/* general sparse product C = A*B (symbolic + numeric); matrix_ABC :186 (PetscMatrix.cpp:833-856), matrix_RightMatMult :189,
 * matrix_LeftMatMult :191 are two / one of these */
int fh_mat_matmul(fh_mat_t A, fh_mat_t B, fh_mat_t* C);
/* matrix_ABC(A, B, C, reuse) :186 (PetscMatrix.cpp:833-856): D = A B C.  *D == NULL: symbolic + numeric, the plan stays attached to D;
 * else numeric only (operands of the same patterns) */
int fh_mat_abc(fh_mat_t A, fh_mat_t B, fh_mat_t C, fh_mat_t* D);
int fh_mat_norm(fh_mat_t A, int kind, double* out);              /* kind 1: l1_norm :211, 0: linfty_norm :214 */

/* SpMV family (NumericVector::matrix_mult :283, add_vector(v,A) :281, resid :282, matrix_mult_transpose :284;
 * PetscVector.cpp:182-247).  mode 0: y = A x ; 1: y += A x ; 2: y = b - A x (b given) ;
 * 3: y = x + omega*dinv.*(b - A x) (one Richardson/Jacobi sweep, 03_solvers/LinearEquationSolverPetsc.cpp:516-519 + PCJACOBI) */
int fh_spmv(fh_mat_t A, fh_vec_t x, fh_vec_t y, int mode, fh_vec_t b, fh_vec_t dinv, double omega);
int fh_spmv_transpose(fh_mat_t A, fh_vec_t x, fh_vec_t y);       /* y = A^T x via the cached explicit transpose */
/* operators over [owned | ghost] columns (distributed levels): number of row blocks that read no column >= n_own_cols
 * (multiplied while the ghost exchange is in flight) and of those that do (multiplied after it) -- the MPIAIJ diagonal /
 * off-diagonal split of MatMult at row-block granularity */
int fh_mat_split_info(fh_mat_t A, int n_own_cols, int* nblk_interior, int* nblk_interface);
/* algorithmic bytes of one y=Ax with this matrix: 12 nnz + 4 (m+1) + 8 n + 8 m */
int64_t fh_spmv_algorithmic_bytes(fh_mat_t A);
/* bytes of the arrays the LDS-staged product kernel reads and writes (mode: 0 y = Ax, 1 y += Ax, 2 r = b - Ax, 3 Jacobi sweep); lo: x counted once,
 * hi: every row block's gather of x counted (no reuse between blocks) */
int fh_spmv_expected_bytes(fh_mat_t A, int mode, int64_t* lo, int64_t* hi);

/* ---- FE tables (a1-a3, a6) -------------------------------------------------------------------
 * geom: 0 = hex (HEX27 geometry), 1 = quad (QUAD9), 2 = line (EDGE3), 3 = triangle (TRI7), 4 = tetrahedron (TET15: 3d/Tetrahedron.cpp; TetBiquadratic is summed
 * from barycentric products and agrees with the reference's expanded polynomials to rounding, every other table bit for bit; no second derivatives for it),
 * 5 = prism (WEDGE21; no serendipity second derivatives).  fe = FEMuS SolType ids (FEFamily order): 0 = linear (Q1), 1 = serendipity (QuadQuadratic / HexQuadratic,
 * 8 / 20 nodes: Quadrilateral.cpp:113-161, Hexahedron.cpp:167-256), 2 = biquadratic (Q2), 3 = piecewise constant (quad0 / hex0, one dof per element).
 * Dofs of one variable on one process (Mesh::GetSolutionDof, Mesh.cpp:1021-1074): local node i of the element for 0 / 1 / 2 -- nodes are numbered vertices, edge
 * mid-points, the rest, so families 0 and 1 own the leading own[0] / own[1] node ids of fh_mesh_info --, the element itself for 3 (and i * nel + iel for 4,
 * discontinuous linear: fh_system_elem_dofs / fh_build_system_prolongator only).  Tables, element prolongators, patterns, prolongators and system dof maps serve
 * 0 .. 3; the Poisson assembler, the Neumann faces and fh_fe_jacobian serve the Lagrange families 0 .. 2; the adaptive-refinement constraints 0 and 2.
 * gauss_order: index 0..4 of quadrature_interface.cpp:36-57 ("seventh" -> 3).
 * Gauss: src/02_reference_geom_elements/02_quadrature/ ; basis: 01_fe/ ; tables: 03_fe_evaluations_at_quadrature/ElemType.cpp:576-741 */
int fh_fe_gauss(int geom, int gauss_order, int* ng, double* w, double* x /* [dim*ng], x[d*ng+ig] */);
int fh_fe_tables(int geom, int fe, int gauss_order, int* ng, int* nc, double* phi /* [ng*nc] */, double* dphi /* [dim][ng*nc] */);
/* second derivatives at the Gauss points, one [ng*nc] table per derivative in the order of elem_type's members (ElemType.cpp:637-741):
 * 3-D _d2phidxi2, _d2phideta2, _d2phidzeta2, _d2phidxideta, _d2phidetadzeta, _d2phidzetadxi; 2-D _d2phidxi2, _d2phideta2, _d2phidxideta */
int fh_fe_tables_d2(int geom, int fe, int gauss_order, double* d2phi /* [3 or 6][ng*nc] */);
/* elem_type::Jacobian(vt, ig, Weight, phi, gradphi, nablaphi) (ElemType.hpp:1183-1248 2-D, :1438-1537 3-D; public wrappers :675-679, :846-850) for
 * EVERY element and Gauss point of a mesh in one launch: vt = the coordinates of the element's first nc nodes.  Host arrays in and out:
 * weight[nel*ng]; gradphi[(e*ng + ig)*nc*dim + dim*j + d]; nablaphi[(e*ng + ig)*nc*nh + nh*j + k], nh = 3 (xx, yy, xy) or 6 (xx, yy, zz, xy,
 * yz, zx) -- the reference's optional Hessians (:1509-1534, :1232-1244); an output passed as NULL is not computed.  phi is fh_fe_tables' */
int fh_fe_jacobian(fh_ctx_t ctx, int geom, int fe, int gauss_order, int nel, int nloc, const int* elem_dof, int nnode, const double* coords,
                   double* weight, double* gradphi, double* nablaphi);
int fh_fe_elem_prolongator(int geom, int fe, int* nchild, int* nc, double* P /* [nchild*nc*nc], |.|<1e-14 -> 0 (ElemType.cpp:439-532) */);

/* ---- mesh + DOF maps (a8-a10): box generator, uniform refinement, first-touch numbering, nprocs=1 ----
 * MeshGeneration.cpp:790-849,979-1075 ; MeshRefinement.cpp:240-294,356-417,513-620 ; Mesh.cpp:517-559 */
int fh_mesh_box(int nx, int ny, int nz, const double lo[3], const double hi[3], fh_mesh_t* mesh);
/* Gambit neutral file: GambitIO::read (src/06_mesh/00_single_level/01_input/01_from_external_file/GambitIO.cpp:93-352)
 * for HEX27 / QUAD9 meshes: nodes, elements (Gambit -> FEMuS local order, :55-80), groups/materials, boundary sets (set n ->
 * face flag -n-1, :337), then the element and node numbering of Mesh::ReadCoarseMesh (Mesh.cpp:239-250).  Lref scales the
 * coordinates.  Errors carry the reference's messages in fh_last_error. */
int fh_mesh_read_gambit(const char* path, double Lref, fh_mesh_t* mesh);
int fh_mesh_refine(fh_mesh_t coarse, fh_mesh_t* fine);
/* selective (adaptive) refinement -- MeshRefinement::RefineMesh with an AMR flag per element (MeshRefinement.cpp:197-493):
 * elements of the current level with flags[iel] != 0 are split, every other element is carried over unchanged, which
 * makes the new level non-homogeneous.  flags == NULL refines every element of the current level. */
int fh_mesh_refine_flagged(fh_mesh_t coarse, const unsigned char* flags /* [nel] or NULL */, fh_mesh_t* fine);
/* The same refinement ON THE DEVICE (round 5; MeshRefinement.cpp:240-294, 356-417, 513-620 + Mesh.cpp:517-559): the final node numbering is
 * the rank of every node's first touch in (class, element, local node) order, so an atomicMin per touched node (edges and faces found
 * through a device hash table), one scan and one pass that writes ids and coordinates replace the host's ordered loops.  Same arrays as
 * fh_mesh_refine_flagged, bit for bit (numbering, boundary flags, coordinates); the fine mesh keeps a device copy that
 * fh_mat_create_from_mesh / fh_assembler_create_mesh / fh_build_prolongator read instead of uploading the element table again. */
int fh_mesh_refine_device(fh_ctx_t ctx, fh_mesh_t coarse, const unsigned char* flags /* [nel] or NULL */, fh_mesh_t* fine);
/* the point FlagElementsToRefine evaluates the user's flag function at: mean of the element vertices (:88-101) */
int fh_mesh_elem_centroids(fh_mesh_t mesh, double* xc /* [nel*3] */);
/* Elem::GetElementLevel per element, Mesh::GetIfHomogeneous */
int fh_mesh_elem_levels(fh_mesh_t mesh, int* levels /* [nel] or NULL */, int* homogeneous /* or NULL */);
/* Elem::GetElementGroup / GetElementMaterial per element (Gambit group and material numbers, inherited by the children of a refinement: what a
 * SetRefinementFlag(x, ElemGroupNumber, level) callback and MeshASMPartitioning read); a generated box: group 1, material 2 */
int fh_mesh_elem_groups(fh_mesh_t mesh, int* group /* [nel] or NULL */, int* material /* [nel] or NULL */);
int fh_mesh_destroy(fh_mesh_t mesh);
/* domain decomposition: faces of a sub-box that are artificial cuts, not physical boundary (bit f = local face f); call on the
 * coarse mesh before refining (flags are inherited, MeshRefinement.cpp:271-278) */
int fh_mesh_clear_boundary_faces(fh_mesh_t mesh, unsigned face_mask);
int fh_mesh_set_coords(fh_mesh_t mesh, const double* coords /* [nnode*dim], this mesh's numbering */);
int fh_mesh_info(fh_mesh_t mesh, int* dim, int* nel, int* nnode, int* nloc, int own_size[3], int* level);
int fh_mesh_get(fh_mesh_t mesh, int* elem_dof /* [nel*nloc] */, double* coords /* [nnode*dim] */, int* face_flag /* [nel*nfaces] */);
int fh_mesh_child_elems(fh_mesh_t coarse, int* child /* [nel*nchild] */);
/* boundary flags: Dirichlet dofs = nodes on faces with flag < -1 (MultiLevelSolution.cpp:725-840, value < 1.5 in
 * 03_solvers/LinearEquationSolverPetsc.cpp:53-90).  Returns the sorted list; *n in: capacity, out: count */
int fh_mesh_dirichlet_dofs(fh_mesh_t mesh, int fe, int* n, int* dofs);

/* ---- sparsity (a11): LinearEquation::GetSparsityPatternSize (03_solvers/LinearEquation.cpp:407-548) ----
 * CSR pattern of the element-connectivity graph: two-call protocol (rowptr first, then col). */
int fh_pattern_from_elements(int nel, int nloc, const int* elem_dof, int ndof, int* rowptr /* [ndof+1] */, int* col /* NULL on first call */);
/* the same pattern built ON THE DEVICE and made a matrix at once (m owned rows over n columns, values zero): node -> element lists by a counting
 * pass, one wave per row sorts the candidate columns in LDS.  The column array stays on the device (host code that asks for it fetches it). */
int fh_mat_create_from_elements(fh_ctx_t ctx, int nel, int nloc, const int* elem_dof, int m, int n, fh_mat_t* out);
/* the same for a square operator of one variable on a mesh (fe 0 .. 3), from the mesh's device copy (made now if it has none) */
int fh_mat_create_from_mesh(fh_ctx_t ctx, fh_mesh_t mesh, int fe, fh_mat_t* out);

/* ---- prolongator (a14): LinearImplicitSystem::BuildProlongatorMatrix (LinearImplicitSystem.cpp:761-909) ----
 * builds P (fine x coarse) from the element prolongator; zero_bdc != 0 also applies
 * ZeroInterpolatorDirichletNodes (LinearImplicitSystem.cpp:1032-1120). */
int fh_build_prolongator(fh_ctx_t ctx, fh_mesh_t coarse, fh_mesh_t fine, int fe, int zero_bdc, fh_mat_t* P);

/* ---- adaptive refinement projection (a22) -----------------------------------------------------
 * Hanging-node constraints of a non-homogeneous level: Mesh::GetAMRRestrictionAndAMRSolidMark (Mesh.cpp:1352-1830),
 * source of Mesh::GetAmrRestrictionMap (Mesh.hpp:677-691).  Two-call protocol: with hanging == NULL only the counts are
 * returned; then hanging[n] (sorted), ptr[n+1], master[nnz], weight[nnz] with chains through intermediate levels
 * resolved.  Hanging dofs are the "AMR artificial Dirichlet" rows (flag 1 of MultiLevelSolution.cpp:725-760): the caller
 * merges them into the Dirichlet list used by mat_zero_rows / ZeroInterpolatorDirichletNodes. */
/* Which map the two functions below build.  mode 0 (default): exactly the reference's -- every pair of levels (ilevel < jlevel) writes
 * restriction[master][hanging] (Mesh.cpp:1489-1590), then the genealogy walk (:1711-1801), whose "alreadyFound" rule drops the
 * path through an intermediate hanging node when the node is also a direct son of the master: at a node on the interfaces with two
 * coarser levels at once (3-D edges with a level jump of two) the row does NOT sum to one.  mode 1: such a node keeps only its
 * description to the coarsest level and chains are expanded completely, so every row sums to one and Q2 polynomials are reproduced
 * (the consistent variant; not what the reference computes).  Both agree wherever level jumps are single.  Inherited by refined
 * meshes; call on the coarse mesh. */
int fh_mesh_set_amr_mode(fh_mesh_t mesh, int mode);
int fh_mesh_amr_constraints(fh_mesh_t mesh, int fe, int* n_hanging, int* nnz, int* hanging, int* ptr, int* master, double* weight);
/* LinearImplicitSystem::BuildAmrProlongatorMatrix (LinearImplicitSystem.cpp:912-1028): P_amr (n x n), identity rows for
 * regular dofs, master weights (+ an explicit zero diagonal) for hanging dofs.  Use sites: P[l] <- P[l] * P_amr[l-1]
 * (:247-262, fh_mat_matmul), RES <- P_amr^T RES and KK <- P_amr^T KK P_amr (:329-342, fh_spmv_transpose / fh_mat_ptap),
 * EPS <- P_amr EPS (:487-491, fh_spmv). */
int fh_build_amr_prolongator(fh_ctx_t ctx, fh_mesh_t mesh, int fe, fh_mat_t* P_amr);

/* ---- batched assembly (a4, a7, a12): the per-element callback of
 * src/08_equations/assemble/00_poisson_eqn_with_all_dirichlet_bc_AD_or_nonAD_separate.hpp:106-228 as ONE call.
 * KK->zero(); RES->zero(); element loop {Jacobian (ElemType.hpp:1183-1248,1438-1537); Res,Jac; add_*_blocked}; close.
 * source_kind: 0 constant f=p[0]; 1: f = p[0]*prod_d sin(p[1]*x_d) ; 2: f = p[0]*prod_d cos(p[1]*x_d);
 * 3: f = p[0]*sum_d prod_{e!=d} x_e (p[1]-x_e)  (Laplacian of the Q2 polynomial -(p[0]/2) prod_d x_d (p[1]-x_d)).
 * sol may be NULL (= 0).  A must carry the pattern from fh_pattern_from_elements. */
typedef struct fh_assembler_s* fh_assembler_t;
int fh_assembler_create(fh_ctx_t ctx, int geom, int fe, int gauss_order, int nel, int nloc, const int* elem_dof,
                        int nnode, const double* coords /* [nnode*dim] */, fh_mat_t A, fh_assembler_t* as);
/* the same from a mesh: element table and coordinates come from the mesh's device copy (made now if it has none), no upload */
int fh_assembler_create_mesh(fh_ctx_t ctx, fh_mesh_t mesh, int fe, int gauss_order, fh_mat_t A, fh_assembler_t* as);
int fh_assembler_destroy(fh_assembler_t as);
/* Galerkin coarse operator PP^T KK PP of a UNIFORMLY refined level, element by element, from the element matrices the fine assembler
 * holds since its last assembly (LinearImplicitSystem.cpp:347-370 calls the sparse product SparseMatrix::matrix_PtAP): child[nel_coarse * 2^dim]
 * = the fine elements of every coarse element (fh_mesh_child_elems), fbdc / cbdc = the Dirichlet nodes whose interpolation rows / columns are
 * zero (ZeroInterpolatorDirichletNodes).  Ac must carry the finite-element pattern of the coarse mesh (the pattern `coarse` was created on). */
int fh_assembler_galerkin(fh_assembler_t fine, fh_assembler_t coarse, const int* child, int nfb, const int* fbdc, int ncb, const int* cbdc, fh_mat_t Ac);
int fh_assemble_poisson(fh_assembler_t as, fh_vec_t sol, int source_kind, const double* params, fh_mat_t A, fh_vec_t res);
int fh_assembler_info(fh_assembler_t as, int* ncolors, int64_t* algorithmic_bytes, double* flops);
/* element-level entry (tests): K[nel*nc*nc], F[nel*nc] for the given elements, no scatter */
int fh_element_matrices_poisson(fh_assembler_t as, fh_vec_t sol, int source_kind, const double* params, double* K, double* F);
/* Optional fast path (fh_set_option(ctx, "assemble_affine", 1), default 0 = every element by quadrature as the reference does):
 * HEX27 / Q2 elements whose geometry is affine (parallelepipeds; classified once from the coordinates, tolerance 1e-12 of the
 * element size) are assembled from nine reference matrices M_ab = sum_g w_g d_a phi_i d_b phi_j as K = sum_ab det (J^-1 J^-T)_ab M_ab,
 * which equals the quadrature loop up to summation order; curved elements keep the quadrature kernel.  Two-pass mode only. */
int fh_assembler_affine_count(fh_assembler_t as, int* n_affine, int* n_general);
/* Fused cluster assembly (fh_set_option(ctx, "assemble_fused", 1), the default; read when an assembler is created and at every assembly):
 * HEX27 / Q2 meshes whose elements come in groups of eight consecutive elements with one common local topology -- the children of one
 * coarse element as MeshRefinement.cpp:240-294 numbers them -- are assembled group by group: the CSR rows whose elements all lie in the
 * group are written once, straight from the eight element matrices; the others are summed by a second pass over a partial-row buffer.
 * Same element matrices as the two-pass path, sums grouped per cluster (deterministic).  active = 0: the mesh does not offer the structure
 * (or the option is off) and the two-pass path runs.  partial_entries = doubles in the partial-row buffer, second_pass_rows = CSR rows summed
 * by the second pass (the scatter of separate.hpp:165-205 through PetscMatrix.cpp:699-729 is what both paths replace). */
int fh_assembler_fused_info(fh_assembler_t as, int* active, int* nclusters, int64_t* partial_entries, int* second_pass_rows);
/* Carried rows of the fused cluster assembly (fh_set_option(ctx, "assemble_carry", v), read when an assembler is created; -1 = automatic, the default):
 * the persistent workgroups of the cluster kernel walk SUPER-clusters of 8^k consecutive clusters (v = 3 k: 3 -> 8 clusters = 64 elements, 6 -> 64 clusters
 * = 512 elements; in a refined mesh the descendants of one ancestor k + 1 levels up) one cluster after the other.  A CSR row all of whose elements lie in one
 * super-cluster is accumulated IN the CSR array by that one workgroup -- the first cluster that holds an entry stores it, the later ones load, add, store,
 * in ascending cluster order, which is the order of the second pass: the values have the bits the partial-row buffer gives -- and only the rows on the
 * surface of a super-cluster go through the partial-row buffer (64 clusters: 33 % of the entries that go there with 0).  Automatic: the largest k <= 2 that
 * leaves every workgroup at least two super-clusters, else 0.  The classes are read from the element lists; nothing about the mesh is assumed beyond the
 * cluster structure itself.  clusters_per_super = 1: nothing is carried; carried_entries = CSR entries of the carried rows. */
int fh_assembler_carry_info(fh_assembler_t as, int* clusters_per_super, int64_t* carried_entries);
/* "assemble_fused" = 1 chooses per assembly: the fused path, unless fh_assembler_galerkin asked for the element rows of the PREVIOUS assembly (a solve that
 * re-prepares its hierarchy after every assembly: the two-pass path leaves the rows in place); 2 = always fused (the rows are re-created when asked for),
 * 0 = never.  path: what the last assembly ran, 1 = fused, 2 = two-pass, 0 = none yet / another path.
 * Both paths are deterministic and use the same element matrices; inside a cluster both add in ascending element order (a row complete inside one
 * cluster has the same bits from both), across clusters the fused path adds cluster sums, the two-pass path single element rows: values agree to
 * rounding (<= 4e-16 relative, tests/test_gpu_fused_assembly.py), so the assembly at which option 1 changes the path differs from its predecessor
 * in the last bits; options 0 and 2 never change path.  (Giving the two-pass row pass the clusters' association was measured: 1.33 -> 1.93 ms.) */
int fh_assembler_last_path(fh_assembler_t as, int* path);

/* Neumann boundary term of the 001_Poisson callback (applications/001_Poisson/main.cpp:560-594): for every listed boundary face
 * res[node_i] += int_face phi_i * tau ds with elem_type::JacobianSur (ElemType.hpp:1089-1138 edges, :1330-1380 quad faces).
 * face_nodes[nfaces*nfn]: node ids of each face in the face element's own local order (fh_fe_face_nodes), nfn = 9/4 (hex faces,
 * Q2/Q1) or 3/2 (quad edges); tau[nfaces]: flux per face.  Contributions to a node are summed in ascending face order. */
int fh_fe_face_nodes(int geom, int fe, int face, int* nfn, int* local_nodes);
/* reference coordinates (-1, 0, 1 per direction) of a local node of the biquadratic element (the X tables of hex_lag / quad_lag) */
int fh_fe_node_ref(int geom, int node, int* xi /* [dim] */);
/* the same as doubles, for every element (geom 3 = the triangle TRI7 of 2d/Triangle.cpp: vertices, edge middles, centre at 0, 1/2, 1, 1/3) */
int fh_fe_node_ref_coords(int geom, int node, double* xi /* [dim]  geom 3 / 4: edges of triangles / faces of tetrahedra (TRI3, TRI6); geom 100 + g names the FACE element itself (101 quadrilateral, 103 triangle, 102 line:
 * prisms have faces of two kinds). */);
int fh_assemble_neumann_faces(fh_ctx_t ctx, int geom, int fe, int gauss_order, int nfaces, const int* face_nodes, const double* tau,
                              int nnode, const double* coords, fh_vec_t res);

/* ---- run-time expressions: femus::ParsedFunction (src/02_calculus/function_parser/ParsedFunction.hpp:25-60,
 * ParsedFunction.cpp:28-80; wraps the third-party "Function Parser for C++", not vendored).  The expression is compiled once
 * into a postfix program; fh_expr_eval is ParsedFunction::operator()(double* x) on the host (boundary values), and
 * fh_assemble_poisson_expr evaluates the same program on the device at every Gauss point (source term "func_source" of
 * applications/001_Poisson/input/input.json, main.cpp:205-211, 472): f = scale * expr(x, y, z, t).  variables: comma separated,
 * e.g. "x,y,z,t".  Constants "pi" and "e" are predefined as in the reference.  A syntax error returns nonzero with the message
 * in fh_last_error (the reference prints it and exits). */
typedef struct fh_expr_s* fh_expr_t;
int fh_expr_compile(const char* expression, const char* variables, fh_expr_t* expr);
int fh_expr_eval(fh_expr_t expr, const double* x, double* value);
int fh_expr_eval_many(fh_expr_t expr, int npts, const double* x /* [npts*nvars] */, double* values);
int fh_expr_program(fh_expr_t expr, int* ncode, int* nconst, int* code /* or NULL */, double* consts /* or NULL */);
int fh_expr_nvars(fh_expr_t expr, int* nvars);      /* number of variables the expression was compiled over */
int fh_expr_destroy(fh_expr_t expr);
int fh_assemble_poisson_expr(fh_assembler_t as, fh_vec_t sol, fh_expr_t source, double scale, fh_mat_t A, fh_vec_t res);
/* Neumann term with the flux as a parsed function of the face Gauss point (x, y, z, t = 0): the parsed-boundary-condition branch of the
 * 001_Poisson callback, `(*bdcfunc)(&xyzt[0])` inside the Gauss loop (applications/001_Poisson/main.cpp:495-553).  face_expr[f] names
 * one of the nexpr expressions (one per boundary face name in the application); everything else as fh_assemble_neumann_faces */
int fh_assemble_neumann_faces_expr(fh_ctx_t ctx, int geom, int fe, int gauss_order, int nfaces, const int* face_nodes, const int* face_expr,
                                   int nexpr, const fh_expr_t* exprs, int nnode, const double* coords, fh_vec_t res);

/* The callback of applications/001_Poisson on a ONE-DIMENSIONAL mesh (main.cpp:355-480 with dim == 1; the shipped input/input1D.json: an EDGE3 box with
 * V = 1, nu = 0.01, main.cpp:392-395): advection-diffusion with the streamline-upwind terms of that loop.  elem_dof[nel*3]: the nodes of every element in
 * EDGE3 order (the two ends, then the middle: MeshGeneration.cpp:231-233), vertices numbered first so that a node id is its dof id in both families;
 * coords[nnode]; fe 0 (two dofs per element) or 1 / 2 (three); sol NULL = 0; source NULL = 0.  KK (pattern holding every (i, j) of every element) and RES are
 * OVERWRITTEN: KK = the Jacobian, RES = the residual, as the callback leaves them before the boundary rows are treated. */
int fh_assemble_advdiff_line(fh_ctx_t ctx, int fe, int gauss_order, int nel, const int* elem_dof, int nnode, const double* coords, fh_vec_t sol, double nu,
                             double velocity, fh_expr_t source, fh_mat_t KK, fh_vec_t RES);

/* The Poisson callback (main.cpp:355-480 with V = 0) through a GENERIC (dim, nc, ng) kernel, for element families whose meshes the caller keeps -- the triangle
 * (geom 3: TRI7 node order of 2d/Triangle.cpp: vertices, edge middles, centre) first: elem_dof[nel*nloc] in the family's local order with the node classes
 * numbered one after the other (a node id is its dof id), coords[nnode*dim].  KK and RES are OVERWRITTEN: KK_ij = sum grad phi_i . grad phi_j w,
 * RES_i = sum (scale f phi_i - grad phi_i . grad Sol) w.  Element matrices first (one wave per element, the entries' places in KK beside them), then one thread per row adding them in
 * ascending element order; geom 0 .. 2 are served too (a cross-check of the
 * tensor-product assemblers on small meshes). */
int fh_assemble_poisson_rows(fh_ctx_t ctx, int geom, int fe, int gauss_order, int nel, int nloc, const int* elem_dof, int nnode, const double* coords, fh_vec_t sol,
                             fh_expr_t source, double scale, fh_mat_t KK, fh_vec_t RES);
/* The same on a mesh of MIXED shapes (applications/001_Poisson/input/cube_all_shapes*.neu: hexahedra, tetrahedra and prisms in one file; quadrilaterals and
 * triangles in two dimensions; the element loop of
 * main.cpp:355-480 asks every element for its own type): elem_geom[nel] = the geom id of every element (at most three different ones, of one dimension), rows of
 * elem_dof padded to nloc.  A row's entries are summed in ascending element order whatever the shapes. */
int fh_assemble_poisson_mixed(fh_ctx_t ctx, int fe, int gauss_order, int nel, int nloc, const int* elem_geom, const int* elem_dof, int nnode, const double* coords,
                              fh_vec_t sol, fh_expr_t source, double scale, fh_mat_t KK, fh_vec_t RES);
/* Open-boundary pressure term of the steady Navier-Stokes residual (src/08_equations/assemble/03_navier_stokes.hpp:185-290): on every listed boundary
 * face  aResV[k][node_i] += phi_i * tau * normal[k] * weight  for the dim velocity components (Q2 face nodes), tau = the prescribed pressure -- one
 * number per face (tau) or expression face_expr[f] of the nexpr expressions evaluated at the face Gauss point, as the bdc callback is (:280) -- and
 * normal / weight from elem_type::JacobianSur at that Gauss point.  res[comp_offset[k] + node] receives scale * the sum (scale = -1: RES = -aRes,
 * :377-381).  Which faces: the application's decision (normal velocity component not Dirichlet, :236-276); fh_fe_face_normals returns the normals it
 * reads for that (host arrays; Gauss point 0 in the reference). */
int fh_assemble_pressure_faces(fh_ctx_t ctx, int geom, int gauss_order, int nfaces, const int* face_nodes, const double* tau, const int* face_expr,
                               int nexpr, const fh_expr_t* exprs, int nnode, const double* coords, const int* comp_offset, double scale, fh_vec_t res);
int fh_fe_face_normals(int geom, int fe, int gauss_order, int gauss_point, int nfaces, const int* face_nodes, int nnode, const double* coords,
                       double* normals /* [nfaces*dim] */);

/* ---- multi-variable systems and the Navier-Stokes Newton path (a9, a21) --------------------------------------------
 * Variables are stacked per rank: system dof = KKoffset[k] + mesh dof (LinearEquation::GetSystemDof, LinearEquation.cpp:76-85,
 * :212-237; nprocs = 1).  fe[k] in {0, 2}, and -- for these two calls -- 4 = DISCONTINUOUS_POLYNOMIAL FIRST (round 5: the pressure of
 * unittests/testNSSteadyDD, 004_Boussinesq, tutorial ex08): dim + 1 functions 1, xi, eta (, zeta) per element, mesh dof of function i of element
 * iel = i * nel + iel (Mesh::GetSolutionDof for solution type 4), interpolated by the element prolongator of that type (ElemType.cpp:446-520:
 * the coarse function at the child's centre for the constant, half the coarse slope for the linear functions; an unrefined element keeps
 * its functions).  elem_sys[nel*nd] lists, per element, the dofs of variable 0, then 1, ... */
int fh_system_elem_dofs(fh_mesh_t mesh, int nvars, const int* fe, int* nd, int* offsets /* [nvars+1] or NULL */, int* elem_sys /* or NULL */);
/* BuildProlongatorMatrix over the system variables (LinearImplicitSystem.cpp:826-909): block-diagonal interpolation; apply
 * ZeroInterpolatorDirichletNodes with fh_mat_zero_rows(fine bdc, 0) / fh_mat_zero_cols(coarse bdc) */
int fh_build_system_prolongator(fh_ctx_t ctx, fh_mesh_t coarse, fh_mesh_t fine, int nvars, const int* fe, fh_mat_t* P);
/* blocks of the Schwarz (Vanka) smoother: one per dof of the LAST variable (the Schur / pressure variable), holding that dof
 * and all dofs of the other variables on the elements that own it (LinearEquationSolverPetscAsm.cpp:134-256).  Two-call
 * protocol (ptr == NULL: counts only) */
int fh_mesh_vertex_patches(fh_mesh_t mesh, int nvars, const int* fe, int* npatch, int* total, int* ptr, int* dofs);
/* steady Navier-Stokes, Taylor-Hood Q2/Q1, variables [U | V | (W) | P]: residual and Newton Jacobian of
 * src/08_equations/assemble/03_navier_stokes.hpp:330-409 as ONE batched call (the reference differentiates the residual with
 * an adept tape per element, Assemble_jacobian.cpp:39-72; here the derivative is written out).  A must carry the pattern of
 * fh_pattern_from_elements over fh_system_elem_dofs.  sol may be NULL (= 0).  nu: kinematic viscosity. */
typedef struct fh_ns_assembler_s* fh_ns_assembler_t;
int fh_ns_assembler_create(fh_ctx_t ctx, int geom, int gauss_order, int nel, int nloc, const int* elem_dof, int nnode, int n_vertex_nodes,
                           const double* coords /* [nnode*dim] */, fh_mat_t A, fh_ns_assembler_t* as);
/* The same weak form with the pressure space of the reference's KNOWN-ANSWER TEST (unittests/testNSSteadyDD/main.cpp:97, callback :396-726; also
 * 004_Boussinesq, 000_tutorial/ex08): DISCONTINUOUS_POLYNOMIAL FIRST -- psi = 1, xi, eta (, zeta) in reference coordinates (quadpwLinear /
 * hexpwLinear, Quadrilateral.cpp:188-200), dim + 1 dofs owned by every element.  Variables stacked [U | V | (W) | P] with the pressure dof of
 * local function i of element e at dim * nnode + i * nel + e (Mesh::GetSolutionDof for solution type 4 on one process); KK must carry the pattern of
 * that element table (fh_mat_create_from_elements).  fh_assemble_navier_stokes assembles it.  tests/test_gpu_ns_known_answer.py reproduces the
 * level-3 norms that test stores (1e-6 there) to 1e-9 with this assembler and the sparse exact solve. */
/* The temperature system of the same test (AssembleMatrixResT, main.cpp:730-880): LAGRANGE SECOND scalar advected by a given velocity field,
 * Res = (-IPe grad phi . grad T - (u . grad T) phi) w, KK = (IPe grad phi_i . grad phi_j + (u . grad phi_j) phi_i) w.  KK: pattern of the mesh
 * (fh_mat_create_from_mesh, fe 2); velocity: a stacked vector [U | V | (W) | ...] of stride nnode (the Navier-Stokes state), or NULL. */
int fh_advdiff_assembler_create(fh_ctx_t ctx, int geom, int gauss_order, int nel, int nloc, const int* elem_dof, int nnode, const double* coords /* [nnode*dim] */,
                                fh_mat_t KK, fh_ns_assembler_t* as);
int fh_assemble_advection_diffusion(fh_ns_assembler_t as, fh_vec_t T /* or NULL */, fh_vec_t velocity /* or NULL */, double inverse_peclet, fh_mat_t KK, fh_vec_t RES);
int fh_ns_pw_assembler_create(fh_ctx_t ctx, int geom, int gauss_order, int nel, int nloc, const int* elem_dof, int nnode, const double* coords /* [nnode*dim] */,
                              fh_mat_t KK, fh_ns_assembler_t* as);
int fh_ns_assembler_destroy(fh_ns_assembler_t as);
int fh_assemble_navier_stokes(fh_ns_assembler_t as, fh_vec_t sol, double nu, fh_mat_t A, fh_vec_t res);
int fh_ns_element_matrices(fh_ns_assembler_t as, fh_vec_t sol, double nu, double* K /* [nel*nd*nd] */, double* F /* [nel*nd] */);
/* The callback applications/003_NavierStokes/SteadyNavierStokesParallel ships (main.cpp:390-925): EQUAL-ORDER LAGRANGE FIRST velocity and pressure on the
 * vertex nodes with the Franca-Frey stabilisation (:677-868; sqrt(lambda_k) = sqrt(6) / h_k as SetLambda stores it for LAGRANGE FIRST, :1070-1082, :1262).
 * Variables [U | V | (W) | P], each n_vertex_nodes long; elem_dof rows need the element's vertices first.  res receives Rhs = aRhs, A the matrix
 * KKloc = -d aRhs / d Soli (:884-910; the reference's adept tape, written out here -- including the derivatives of the stabilisation parameters).
 * inverse_reynolds = the IRe of the call; the application's continuation (:485-489) is the caller's (adapters: HipNavierStokes helpers).
 * fh_ns_element_matrices and fh_ns_assembler_destroy serve both kinds of assembler. */
int fh_ns_stab_assembler_create(fh_ctx_t ctx, int geom, int gauss_order, int nel, int nloc, const int* elem_dof, int nnode, int n_vertex_nodes,
                                const double* coords /* [nnode*dim] */, fh_mat_t A, fh_ns_assembler_t* as);
int fh_assemble_navier_stokes_stab(fh_ns_assembler_t as, fh_vec_t sol, double inverse_reynolds, fh_mat_t A, fh_vec_t res);

/* ---- multigrid: LinearEquationSolver (03_solvers/LinearEquationSolver.hpp:54-261, LinearEquationSolverPetsc.cpp) ----
 * fh_mg_create      <- MGInit   (:185-215)   nlevels, outer solver
 * fh_mg_set_level   <- MGSetLevel (:219-290) operator, interpolation PP (restriction = PP^T when R==NULL,
 *                      LinearImplicitSystem.cpp:379-382), smoother type / omega / npre / npost, Dirichlet list (SetPenalty)
 * fh_mg_setup       <- KSPSetUp: diagonal inverses, explicit transposes, coarse factorisation (PREONLY+LU level 0,
 *                      LinearEquationSolverPetsc.hpp:131-134), work vectors, hipGraph capture of the cycle
 * fh_mg_vcycle      <- one PCMG multiplicative V-cycle application x = M^-1 b
 * fh_mg_solve       <- MGSolve (:294-353): outer solver preconditioned by the cycle
 * fh_mg_destroy     <- MGClear (LinearEquationSolverPetsc.hpp:86-88) */
/* level smoothers, all as Richardson(omega) with a fixed sweep count (LinearEquationSolverPetsc.cpp:516-519):
 *   FH_SMOOTH_JACOBI    PCJACOBI (PetscPreconditioner.cpp:209-212): one fused SpMV per sweep
 *   FH_SMOOTH_GS_COLOR  PCSOR's local symmetric sweep over the colours of the matrix graph (GPU-friendly order; converges to the same
 *                       solution, iteration history differs from the reference's)
 *   FH_SMOOTH_VANKA     block Schwarz on dof patches (FEMuS_ASM)
 *   FH_SMOOTH_SOR       PCSOR exactly as PETSc applies it (:219-222; 001_Poisson/main.cpp:240-242): omega_sor = 1, one local symmetric
 *                       Gauss-Seidel sweep in the NATURAL row order from a zero guess; level-scheduled on the device
 *   FH_SMOOTH_ILU0      PCILU (:91-115): ILU(0) of the local block in natural order, zero pivot 1e-16 with MAT_SHIFT_NONZERO
 *                       (LinearEquationSolverPetsc.cpp:444-446: restart on A + shift I, shift 100 eps then doubled); re-factored by
 *                       every fh_mg_setup; level-scheduled triangular solves
 *   FH_SMOOTH_IDENTITY  PCNONE (IDENTITY_PRECOND, PetscPreconditioner.cpp:75-77): B = I
 *   FH_SMOOTH_LU        PCLU (LU_PRECOND / MLU_PRECOND as the preconditioner of a level solver, PetscPreconditioner.cpp:147-160): B = A^-1 by the sparse
 *                       exact solve (fh_direct_*; symmetric level operators), refactored at every fh_mg_setup; fh_mg_set_level_coords is optional
 *   FH_SMOOTH_ASM       PCASM as FEMuS_ASM configures it (PetscPreconditioner.cpp:179-184: PC_ASM_BASIC + PC_COMPOSITE_MULTIPLICATIVE; sub-solves
 *                       LinearEquationSolverPetscAsm.cpp:278-335): B r = the blocks of fh_mg_set_level_patches visited in their INDEX ORDER from a zero
 *                       correction, block p adding (L~ U~)^-1 (r - A y)_p over its whole (overlapping) dof set, L~ U~ = ILU(0) of the block matrix in
 *                       ascending dof order, zero pivot 1e-16, MAT_SHIFT_NONZERO.  The sequential order is kept by level scheduling (blocks that do
 *                       not touch each other's rows run beside each other).  FH_SMOOTH_VANKA is this library's own variant of the same smoother:
 *                       exact block inverses, damped, in colour order */
enum { FH_SMOOTH_JACOBI = 0, FH_SMOOTH_GS_COLOR = 1, FH_SMOOTH_VANKA = 2, FH_SMOOTH_SOR = 3, FH_SMOOTH_ILU0 = 4, FH_SMOOTH_IDENTITY = 5, FH_SMOOTH_LU = 6,
       FH_SMOOTH_ASM = 7 };
/* outer solver of fh_mg_solve (`SetOuterSolver`, `_mgSolverType`; KSP types of LinearEquationSolverPetsc.cpp:455-529): one cycle,
 * Richardson, left-preconditioned GMRES, CG, and flexible (right-preconditioned) GMRES for cycles that are not a fixed linear
 * operator (GMRES level solvers) */
enum { FH_OUTER_PREONLY = 0, FH_OUTER_RICHARDSON = 1, FH_OUTER_GMRES = 2, FH_OUTER_CG = 3, FH_OUTER_FGMRES = 4 };
int fh_mg_create(fh_ctx_t ctx, int nlevels, fh_mg_t* mg);
int fh_mg_set_level(fh_mg_t mg, int level, fh_mat_t A, fh_mat_t P, fh_mat_t R, int smoother, double omega, int npre, int npost);
/* FH_SMOOTH_VANKA: block Schwarz smoother for saddle-point systems (the FEMuS_ASM choice of the Navier-Stokes applications,
 * petsc_asm/LinearEquationSolverPetscAsm.cpp:91-345): x_p += omega A_pp^-1 (b - A x)_p per patch of dofs, multiplicative over
 * conflict-free colours of patches.  Patches (fh_mesh_vertex_patches) are given per level before fh_mg_setup, which
 * extracts and inverts the patch matrices (dense, partial pivoting) from the level's current operator. */
int fh_mg_set_level_patches(fh_mg_t mg, int level, int npatch, const int* ptr /* [npatch+1] */, const int* dofs);
/* FH_SMOOTH_ASM only: the first `nfirst` blocks of the list get the EXACT sub-solve instead of ILU(0) -- what FEMuS_ASM gives the blocks of the solid and
 * the porous elements, which MeshASMPartitioning::DoPartition puts first (`_blockTypeRange[1]`; MLU_PRECOND on them, LinearEquationSolverPetscAsm.cpp:298-307).
 * After fh_mg_set_level_patches (which resets it to 0), before fh_mg_setup. */
int fh_mg_set_level_patches_exact(fh_mg_t mg, int level, int nfirst);
/* Level solver (`SetSolverFineGrids`, LinearImplicitSystem; `_levelSolverType`, LinearEquationSolverPetsc.cpp:238-250 and 501-519):
 * FH_LEVEL_RICHARDSON  x <- x + omega B (b - A x), npre / npost times (KSPRICHARDSON, the smoother of fh_mg_set_level as B)
 * FH_LEVEL_GMRES       npre / npost iterations of left-preconditioned GMRES with the same B (KSPGMRES, the reference's default
 *                      level solver and the one 003_NavierStokes selects), classical Gram-Schmidt, restart `restart` (FEMuS: 30) */
enum { FH_LEVEL_RICHARDSON = 0, FH_LEVEL_GMRES = 1 };
int fh_mg_set_level_solver(fh_mg_t mg, int level, int solver, int restart);
/* Sparse exact solve (the reference's MUMPS / PCLU through PETSc: coarsest level LinearEquationSolverPetsc.hpp:131-138, `Solve(vars, ksp_clean)`,
 * MLU_PRECOND / LU_PRECOND as level preconditioner PetscPreconditioner.cpp:147-160): multifrontal factorisation over a nested-dissection tree
 * (fh_direct.hip), operators of any size: symmetric ones on unpivoted symmetric fronts, UNSYMMETRIC or INDEFINITE ones (Navier-Stokes Jacobians, saddle
 * points with an empty diagonal block) on general fronts with partial pivoting inside the front and static perturbation of pivots that stay tiny -- what
 * MUMPS does for the reference -- followed by two steps of iterative refinement per solve when a pivot was perturbed.  The path is chosen by an
 * entry-by-entry symmetry test at every factorisation; fh_direct_set_general(d, 1) forces the pivoted path, fh_direct_stats reports which one ran and how
 * many pivots were perturbed.  coords ([n * dim], dim 1..3) let the dissection cut at coordinate layers (for Q2 unknowns:
 * planes of nodes at element boundaries); dim = 0 / coords = NULL: breadth-first level sets of the matrix graph.  leaf <= 0: 256 unknowns per
 * leaf.  fh_direct_factor reads the CURRENT values of A (symbolic work is redone only when the set of coupled unknowns or the matrix changed);
 * unknowns coupled to nothing (penalised Dirichlet rows) are solved by their diagonal.  fh_direct_solve: x = A^-1 b (b != x). */
typedef struct fh_direct_s* fh_direct_t;
int fh_direct_create(fh_ctx_t ctx, fh_mat_t A, int dim, const double* coords, int leaf, fh_direct_t* out);
int fh_direct_factor(fh_direct_t d);
int fh_direct_solve(fh_direct_t d, fh_vec_t b, fh_vec_t x);
int fh_direct_info(fh_direct_t d, int* coupled, int* fronts, int* height, int* largest_front, int64_t* factor_doubles);
int fh_direct_set_general(fh_direct_t d, int on);
int fh_direct_stats(fh_direct_t d, int* general_fronts, int* perturbed_pivots, int* refinement_steps);
int fh_direct_destroy(fh_direct_t d);
/* PCMGSetType (`MgSmootherType` of MGInit, LinearEquationSolverPetsc.cpp:199-214), one application of the preconditioner to b:
 * FH_CYCLE_MULTIPLICATIVE  V-cycle: pre-smooth, restrict the residual, recurse, interpolate-add, post-smooth (default)
 * FH_CYCLE_FULL            b restricted through all levels; coarsest solve; per level: x = P x_coarse as the guess, then one multiplicative cycle from there
 * FH_CYCLE_ADDITIVE        b restricted through all levels; every level solves from zero with its DOWN smoother (npre iterations), level 0 exactly;
 *                          the corrections are interpolated upwards and added
 * FH_CYCLE_KASKADE         b restricted through all levels; coarsest solve; per level: x = P x_coarse, then npre iterations of the down smoother
 * (PETSc's PCMGMCycle_Private / PCMGFCycle_Private / PCMGACycle_Private / PCMGKCycle_Private).  Takes effect at the next cycle. */
enum { FH_CYCLE_MULTIPLICATIVE = 0, FH_CYCLE_FULL = 1, FH_CYCLE_ADDITIVE = 2, FH_CYCLE_KASKADE = 3 };
int fh_mg_set_cycle_type(fh_mg_t mg, int type);
int fh_mg_setup(fh_mg_t mg);
/* coordinates of the unknowns of level 0 ([n * dim], dim 1..3; the dofs of the coarsest mesh, Mesh::GetTopology()->_Sol[0..2]): the exact coarse
 * solve (the reference's PCLU / MUMPS on the coarsest level, LinearEquationSolverPetsc.cpp:237-287) then dissects its dense problem -- block
 * inverses beside each other + a separator Schur complement (option "coarse_nd", symmetric operators) instead of one dense inverse.  Optional. */
int fh_mg_set_coarse_coords(fh_mg_t mg, int dim, int n, const double* coords);
int fh_mg_set_level_coords(fh_mg_t mg, int level, int dim, int n, const double* coords);   /* level >= 1 with FH_SMOOTH_LU; level 0 = fh_mg_set_coarse_coords */
/* what the last fh_mg_setup made of the coarsest level: unknowns in the dense problem (the others are solved by their diagonal), interior
 * blocks of the dissection (0 = one dense inverse), separator size, largest block; any pointer may be NULL */
int fh_mg_coarse_info(fh_mg_t mg, int* n_dense, int* nd_blocks, int* nd_separator, int* nd_largest_block);
int fh_mg_vcycle(fh_mg_t mg, fh_vec_t b, fh_vec_t x);
int fh_mg_solve(fh_mg_t mg, fh_vec_t b, fh_vec_t x, int outer, double rtol, double atol, double dtol, int maxit, int restart,
                int* iterations, double* final_residual);
int fh_mg_destroy(fh_mg_t mg);
int64_t fh_mg_cycle_algorithmic_bytes(fh_mg_t mg);               /* SURVEY 8(d) V-cycle byte model for these levels */
/* distributed levels (one mesh partition per GPU): the level's operator holds the OWNED rows only (m = n_owned) over the
 * local [owned | ghost] column space (n = n_owned + nghost); `halo` refreshes the ghosts before every operator application
 * (what MatMult does for MPIAIJ).  R must be given explicitly for such levels (owned coarse rows x local fine columns).
 * replicated_below != 0: the level below is replicated on every rank (its rhs is summed with an all-reduce after the
 * restriction; the prolongation needs no exchange).  Call before fh_mg_setup. */
int fh_mg_set_level_distributed(fh_mg_t mg, int level, fh_halo_t halo, int replicated_below);

/* ---- multi-GPU halo (C1/C3/C4 of SURVEY 2.1): VecGhostUpdate (PetscVector.hpp:595-612), VecDot/VecNorm allreduce ----
 * one rank per GPU.  The plan is built from each rank's ghost list; the exchange itself is neighbour
 * send/recv over RCCL (xGMI) on the context's communication stream.  comm_id is the 128-byte ncclUniqueId
 * created by rank 0 (fh_halo_unique_id) and distributed by the launcher (torch.distributed / MPI). */
int fh_halo_unique_id(char id128[128]);
int fh_halo_create(fh_ctx_t ctx, int rank, int nranks, const char id128[128],
                   const int* send_counts /* [nranks] */, const int* send_idx /* local owned indices, grouped by dest rank */,
                   const int* recv_counts /* [nranks] */, fh_halo_t* halo);
/* further exchange plans (other multigrid levels) on the communicator of an existing one (a ncclUniqueId makes ONE communicator) */
int fh_halo_create_shared(fh_halo_t parent, const int* send_counts, const int* send_idx, const int* recv_counts, fh_halo_t* halo);
/* ncclGetVersion of the RCCL this process is bound to (a launcher that also loads PyTorch runs on the copy PyTorch bundles: report it) */
int fh_rccl_version(int* version);
/* host-staged transport for the same plans (ranks without an RCCL peer-to-peer path, e.g. several ranks on one GPU, or a
 * launcher that already has MPI): the pack kernel, the ghost layout and every caller stay the same, the bytes travel through
 * pinned host buffers and the two functions given here (e.g. MPI_Neighbor_alltoallv and MPI_Allreduce).  exchange(user, send,
 * send_counts[nranks], recv, recv_counts[nranks]) moves doubles grouped by destination / source rank; allreduce(user, buf, n)
 * sums in place.  Both return 0 on success.  fh_halo_create_shared on such a plan inherits the functions. */
typedef int (*fh_exchange_fn)(void* user, const double* send, const int* send_counts, double* recv, const int* recv_counts);
typedef int (*fh_allreduce_fn)(void* user, double* buf, int n);
int fh_halo_create_host(fh_ctx_t ctx, int rank, int nranks, fh_exchange_fn exchange, fh_allreduce_fn allreduce, void* user,
                        const int* send_counts, const int* send_idx, const int* recv_counts, fh_halo_t* halo);
int fh_halo_update(fh_halo_t halo, fh_vec_t v);                  /* owner -> ghost copies, async on comm stream + join */
/* ghost -> owner, ADD_VALUES: what VecAssemblyBegin/End do with the stash of off-process adds (PetscVector.cpp:131-153, PetscVector.hpp:595-612).
 * Staged adds (fh_vec_stage_values) whose index is a ghost of this rank are collected beside the vector; this call ships them to the owners,
 * which add them to their entries (source ranks in ascending order: deterministic).  Collective over the ranks of the plan; a rank without
 * such adds sends zeros.  fh_vec_ghost_adds_pending flushes the staged adds and says whether this rank has any (the caller reduces the flag
 * over the ranks and skips the exchange when nobody has).  Follow with fh_halo_update to refresh the ghost copies. */
int fh_halo_reverse_add(fh_halo_t halo, fh_vec_t v);
int fh_vec_ghost_adds_pending(fh_vec_t v, int* pending);
int fh_vec_ghost_adds(fh_vec_t v, double* out /* [nghost] the adds collected for the ghost entries since the last fh_halo_reverse_add */);
/* the two halves of fh_halo_update (VecGhostUpdateBegin / VecGhostUpdateEnd, PetscVector.hpp:605-608): begin packs on the compute
 * stream and starts the transfer on the communication stream; device work queued between begin and end overlaps with it and must
 * not read the ghost tail; consumers queued after end see the received values.  One exchange in flight per plan. */
int fh_halo_begin(fh_halo_t halo, fh_vec_t v);
int fh_halo_end(fh_halo_t halo);
/* the SpMV family (modes of fh_spmv) for a GHOSTED operand: A holds this rank's rows over [owned | ghost] columns, the ghosts of x
 * are refreshed first -- what NumericVector::matrix_mult / resid do on an MPIAIJ matrix (PetscVector.cpp:203-247): the rows that
 * read no ghost column are multiplied while the exchange is in flight, the others after it.  halo == NULL: plain fh_spmv. */
int fh_spmv_ghosted(fh_mat_t A, fh_halo_t halo, fh_vec_t x, fh_vec_t y, int mode, fh_vec_t b, fh_vec_t dinv, double omega);
/* counters of a plan since creation / the last reset: exchanges started, payload bytes sent by this rank; with
 * fh_set_option("halo_profile", 1) (synchronises every exchange -- measurement runs only) also the summed duration of the
 * exchanges in ms (pack finished -> ghosts landed) and the part the compute stream actually waited for (exposed = not hidden
 * behind the rows that need no ghost).  Any pointer may be NULL. */
int fh_halo_stats(fh_halo_t halo, int reset, int64_t* n_updates, int64_t* bytes_sent, double* exchange_ms, double* exposed_ms);
int fh_halo_allreduce_count(fh_halo_t halo, int reset, int64_t* n);   /* all-reduces issued through this plan (vectors, matrices, host scalars) */
int fh_halo_allreduce_ms(fh_halo_t halo, int reset, double* ms);   /* with the option "halo_profile": time the compute stream spent in the all-reduces of this plan */
int fh_halo_sizes(fh_halo_t halo, int* nsend, int* nrecv);
int fh_halo_allreduce_vec(fh_halo_t halo, fh_vec_t v);            /* in-place sum over ranks of the owned part (device) */
int fh_halo_allreduce_sum(fh_halo_t halo, double* vals, int n);  /* host values in/out, any length */
int fh_halo_rank(fh_halo_t halo, int* rank, int* nranks);
int fh_halo_allreduce_mat(fh_halo_t halo, fh_mat_t A);            /* in-place sum over ranks of the values of a matrix with one pattern on all ranks */
int fh_halo_destroy(fh_halo_t halo);

/* ---- output and restart files (host-side) --------------------------------------------------------------------
 * fh_write_vtu: VTKWriter::Write(output_path, "biquadratic", vars) (VTKWriter.cpp:36-120, 460-770): one UnstructuredGrid piece of
 * biquadratic cells, base64 "binary" DataArrays as print_data_array emits them.  fe[k] = 2: values[k] has nnode entries; 0: the Q1
 * dofs (vertex count), carried to the biquadratic nodes.
 * fh_vec_binary_print / _load: NumericVector::BinaryPrint / BinaryLoad (NumericVector.hpp:345-353), the per-variable files of
 * MultiLevelSolution::SaveSolution / LoadSolution (MultiLevelSolution.cpp:1070-1126): PETSc's binary Vec layout (big-endian class id
 * 1211214, length, float64 values).  fh_host_binary_*: the same on host arrays (*n: capacity in / length out; values == NULL queries). */
int fh_write_vtu(fh_mesh_t mesh, const char* path, int nfields, const char* const* names, const int* fe, const double* const* values);
/* fh_write_gmv: GMVWriter::Write (GMVWriter.cpp:72-341), the second output of applications/001_Poisson (main.cpp:267-269): binary GMV file
 * ("gmvinput" "ieeei4r8", nodes, cells, the METIS_DD cell variable, node variables).  order 0 = linear (phex8 / quad cells on the vertex
 * nodes), otherwise the reference's quadratic family (phex20 / 8quad; "biquadratic" selects it too, GMVWriter.cpp:102).  fe[k] as above. */
int fh_write_gmv(fh_mesh_t mesh, const char* path, int order, int nfields, const char* const* names, const int* fe, const double* const* values);
/* XDMFWriter::Write(output_path, "biquadratic", vars, time_step) (src/07_mesh_or_solution/01_multiple_levels/01_output/XDMFWriter.cpp:103-445):
 * <prefix>.level<L>.<step>.biquadratic.xmf + .h5 with /NODES_X1..3, /CONNECTIVITY, /DOMAIN_PARTITIONS and one dataset per variable.  HDF5 is
 * opened at run time (dlopen; FEMUS_HIP_HDF5 may name the library): fh_xdmf_available() says whether the call can work. */
int fh_xdmf_available(void);
int fh_write_xdmf(fh_mesh_t mesh, const char* output_path, const char* prefix, int level, int time_step, int nfields, const char* const* names,
                  const int* fe, const double* const* values);
int fh_vec_binary_print(fh_vec_t v, const char* path);
int fh_vec_binary_load(fh_vec_t v, const char* path);
int fh_host_binary_print(const char* path, int n, const double* values);
int fh_host_binary_load(const char* path, int* n, double* values);

/* ---- domain-decomposition planner (host-only, integer work) ---------------------------------------------------------
 * Which nodes of a rank's local mesh it owns, which it needs as ghosts, the [owned | ghost] renumbering and the send lists of the
 * ghost exchange: Mesh::dofmap_* (lowest rank touching a node owns it, Mesh.cpp:517-559; ghost lists :767-795), KKghost_nd
 * (LinearEquation.cpp:239-280), VecCreateGhost (PetscVector.hpp:515-569).  A C++ FEMuS build reaches the multi-GPU path through
 * these calls and HipVector::attach_halo (close() = ghost refresh, dot / norms summed over the ranks).
 *   fh_dd_box_node_keys   box partition (the METIS stand-in): global id and owner rank of every node from its exact coordinates;
 *                         rank (c0,c1,c2) of a p0 x p1 x p2 grid owns the unit cube [c, c+1), nb coarse elements per unit
 *   fh_dd_plan_create     gid[n] / owner[n] / need[n] (1 = a row this rank owns reads the node: fh_mat_col_mask / fh_mat_row_mask) ->
 *                         plan.  One personalised all-to-all of 64-bit ids through the caller's function: called twice, first with
 *                         send == recv == NULL to exchange the counts (send_counts[r] ids go to rank r, recv_counts[r] is filled), then
 *                         with the data (MPI_Alltoall + MPI_Alltoallv in an MPI launcher).  nranks == 1: no communication
 *   fh_dd_plan_get        owned[] (local ids, ascending), ghost[] (by owner rank, then global id), newid[n] (-1: not part of the
 *                         rank's vectors), send_counts / send_idx (positions among the owned entries) / recv_counts for fh_halo_create*
 *   fh_dd_plan_global     the reference's global numbering of the level: rank r owns the contiguous range [offsets[r], offsets[r+1])
 *                         (_dofOffset, Mesh.cpp:735-741; KKoffset, LinearEquation.cpp:212-237) and every ghost's global index (KKghost_nd,
 *                         :239-280) = the list NumericVector::init(N, n_local, ghost, fast, GHOSTED) takes
 *   fh_dd_plan_halo       the device exchange plan of the level (RCCL; parent != NULL: on that plan's communicator / transport)
 *   fh_dd_system_offsets  system rows of several variables on several ranks: rank by rank, variable by variable inside a rank
 *                         (LinearEquation::InitPde, LinearEquation.cpp:212-237): kk_offset[j][p] = first row of variable j on rank p
 *                         (row nvars = end of the rank), kk_index[j] = global size of the variables before j
 *   fh_dd_system_dofs     LinearEquation::GetSystemDof (LinearEquation.cpp:76-85) for a list of mesh dofs of one variable: owner rank by
 *                         bisection of the dof offsets (Mesh.cpp:1004-1018), row = kk_offset[var][p] + idof - dof_offset[var][p] */
/* Arbitrary coarse meshes (MeshMetisPartitioning.cpp:71-113: METIS_PartMeshDual on the coarsest level, children inherit :143-155):
 *   fh_mesh_partition      native k-way partition of the dual graph (recursive bisection by breadth-first growing; METIS is not a dependency)
 *   fh_mesh_rank_elements  a rank's elements: owned first, then the ring sharing a node with them
 *   fh_mesh_submesh        FEMuS-numbered mesh of those elements + for every node the node of the coarse mesh it is
 *   fh_dd_topo_node_keys   global id / owner (lowest rank around, Mesh.cpp:517-559) of every node of a refined level of the sub-mesh from
 *                          the refinement tree alone (entity of the coarse mesh + dyadic offsets in a frame fixed by global ids) */
int fh_mesh_partition(fh_mesh_t coarse, int nparts, int* part /* [nel] */);
/* the same with element weights (adaptive levels: weight = finest-level descendants of a coarse element; replaces the re-partition of a
 * refined level, MeshMetisPartitioning.cpp:41-113 with AMR = true, while children inherit their coarse element's rank, :143-155) */
int fh_mesh_partition_weighted(fh_mesh_t coarse, int nparts, const double* weight /* [nel], > 0 */, int* part /* [nel] */);
int fh_mesh_rank_elements(fh_mesh_t coarse, const int* part, int rank, int* n_owned, int* n_total, int* elems /* may be NULL */);
int fh_mesh_submesh(fh_mesh_t coarse, int nsel, const int* sel, fh_mesh_t* sub, int* node_gid /* [nodes of sub], may be NULL */);
int fh_dd_topo_node_keys(fh_mesh_t coarse, const int* part, int nlevels, const fh_mesh_t* levels, const int* elem_gid0, int level, int64_t* gid, int* owner);
typedef struct fh_dd_plan_s* fh_dd_plan_t;
typedef int (*fh_dd_alltoallv_fn)(void* user, const int64_t* send, const int* send_counts, int64_t* recv, int* recv_counts);
int fh_dd_box_node_keys(int n, const double* coords /* [n*3] */, int level, int nb, const int p[3], int64_t* gid, int* owner);
int fh_dd_plan_create(int rank, int nranks, int n, const int64_t* gid, const int* owner, const unsigned char* need,
                      fh_dd_alltoallv_fn alltoallv, void* user, fh_dd_plan_t* plan);
int fh_dd_plan_sizes(fh_dd_plan_t plan, int* n_owned, int* n_ghost, int* n_send);
int fh_dd_plan_get(fh_dd_plan_t plan, int* owned, int* ghost, int* newid, int* send_counts, int* send_idx, int* recv_counts);
int fh_dd_plan_global(fh_dd_plan_t plan, int64_t* offsets /* [nranks+1] */, int64_t* ghost_global /* [n_ghost] */);
int fh_dd_plan_halo(fh_dd_plan_t plan, fh_ctx_t ctx, const char id128[128], fh_halo_t parent, fh_halo_t* halo);
int fh_dd_plan_destroy(fh_dd_plan_t plan);
int fh_dd_system_offsets(int nvars, int nranks, const int64_t* dof_offset /* [nvars][nranks+1] */, int64_t* kk_offset /* [nvars+1][nranks] */,
                         int64_t* kk_index /* [nvars+1] or NULL */);
int fh_dd_system_dofs(int nvars, int nranks, const int64_t* dof_offset, const int64_t* kk_offset, int var, int n, const int64_t* idof,
                      int64_t* sysdof, int* owner /* or NULL */);

#ifdef __cplusplus
}
#endif
#endif /* FEMUS_HIP_H */
