// Batched Poisson assembly on gfx950 (K1-K4 of SURVEY 2.1; a4, a7, a12 of SURVEY 8).
// Restates, as ONE device pass per element colour, the per-element callback of
//   src/08_equations/assemble/00_poisson_eqn_with_all_dirichlet_bc_AD_or_nonAD_separate.hpp:106-228
// with elem_type::Jacobian from src/02_reference_geom_elements/03_fe_evaluations_at_quadrature/ElemType.hpp:1183-1248
// (2-D) and :1438-1537 (3-D):
//   J = sum_n dphi_hat[g][n] (x) x_n over the nc SOLUTION dofs, det, adjugate/det, Weight = det * w_g,
//   gradphi = dphi_hat . JacI ; Res_i += (-f(x_g) phi_i - grad phi_i . grad u) W ; Jac_ij += (grad phi_i . grad phi_j) W.
//
// Mapping (wave64): LPE lanes own one element (Q2 hex: 64 lanes, Q1 quad: 1 lane -> 64 elements per wave).
//   phase 1  lanes = (Gauss point of the chunk) x (node part): partial J, wave shuffles, J^-1, physical gradients
//            -> LDS gs[g][d][n] (+ weight, grad u, f) ; conflict-free ds_write_b64
//   phase 2  lanes = (TI x TJ) register tiles of the element matrix, LDS-staged basis x basis accumulation with
//            ds_read_b128 broadcasts; 16 FMA per 8 LDS doubles
//   scatter  element colours make A[row,col] += K race-free without atomics (deterministic); the CSR slot of
//            every (i,j) comes from a precomputed element->CSR map (emap, built once per pattern) or, if disabled,
//            from a binary search in the sorted row.
// The small dense K is FP64-VALU work, not an MFMA target (north_star); the scatter side is HBM-bound.
#include "fh_internal.h"
#include "fh_fe.h"
#include "fh_expr_device.h"
#include <algorithm>

struct SfTab { double L[3][4], D[3][4]; };   // l_a(x_k), l'_a(x_k): 1-D quadratic Lagrange basis (nodes -1, 0, +1) at the four abscissae

struct fh_assembler_s {
  fh_ctx_t ctx = nullptr;
  int geom = 0, fe = 0, order = 0, dim = 3, nc = 27, ng = 64, nloc = 27;
  int nel = 0, nnode = 0, ndof = 0;
  int ncolors = 0;
  std::vector<int> color_ptr;    // host
  int* d_color_elems = nullptr;  // elements grouped by colour
  int* d_elem_dof = nullptr;     // [nel*nloc]
  double* d_coords = nullptr;    // [nnode*dim]
  double* d_w = nullptr;         // [ng]
  double* d_phi = nullptr;       // [ng*nc]
  double* d_dphi = nullptr;      // [ng*nc*dim]
  int* d_emap = nullptr;         // [nel*nc*ncp] CSR slot of (i,j), ncp = nc rounded up to 4
  int* d_iota = nullptr;         // identity element list (for the uncoloured element-matrix entry point)
  int ncp = 28;
  // two-pass ("row gather") assembly: element matrices -> Kbuf, then every CSR row sums its <= 8 element rows in element order
  int* d_adj_ptr = nullptr;      // [m+1]
  int* d_adj_ei = nullptr;       // (element << 5) | local row, ascending element order
  unsigned char* d_rowmap = nullptr;   // [nadj*nc] slot of (element row, j) inside the CSR row
  int* d_slot = nullptr;         // [nel*nc] adjacency slot of (element, local row), -1 when the row is not in the matrix
  double* d_Kbuf = nullptr;      // [nadj*kstride] element rows in row-gather order
  size_t kbuf_bytes = 0;
  int nadj = 0;                  // element rows in d_Kbuf; row nadj is the spare one
  // element-wise Galerkin product from the next finer level (fh_assembler_galerkin): children, child interpolation tables, Dirichlet masks
  int* d_gal_child = nullptr;
  uint64_t gal_key = 0;          // hash of (children, fine / coarse Dirichlet nodes) the tables below were made from
  bool gal_children_in_order = false;      // child j of coarse element E is fine element 8 * (its cluster) + j
  unsigned char *d_gal_cnt = nullptr, *d_gal_row = nullptr, *d_gal_fb = nullptr, *d_gal_cb = nullptr;
  unsigned* d_gal_fmask = nullptr;     // [nel * 8] Dirichlet bits of the children's nodes, then [nel] of the coarse element's (k_galerkin_macro)
  double *d_gal_val = nullptr, *d_gal_res = nullptr, *d_gal_dense = nullptr;
  int kstride = 27;              // doubles per element row in d_Kbuf (nc, or 32 for HEX27/Q2: whole 64-byte lines per row)
  double* d_Fbuf = nullptr;      // [nadj]
  bool two_pass = false;
  // fused cluster assembly (k_cluster_q2hex_sf + k_rows_partial): groups of 8 consecutive elements with one common local topology
  // (the children of one coarse element: 125 macro nodes).  Plan = tables of the template + per-cluster destinations and maps.
  bool fused = false;            // plan built and verified
  bool kbuf_valid = false;       // the element-row buffer holds the matrices of the last assembly (the fused path does not write it)
  bool rows_used_since = false;  // the element-wise Galerkin product asked for the element rows since the last assembly: the next assembly keeps them (two-pass)
  int last_path = 0;             // 1: the last assembly ran the fused path, 2: two-pass
  int cl_ncl = 0, cl_nm = 0, cl_ns = 0, cl_nprow = 0;
  int cl_sup_shift = 0;          // log2 of the clusters per super-cluster whose inner rows are carried in the CSR array (0: none)
  int cl_walk_shift = 0;         // log2 of the consecutive clusters a workgroup of the cluster kernel serves one after the other (= cl_sup_shift; measurements: assemble_carry 100 + k walks 2^k clusters with nothing carried)
  size_t cl_ncarried = 0;        // CSR entries of carried rows
  size_t cl_npart = 0;           // entries of the partial-row buffer (without the sink)
  unsigned* d_cl_sinfo = nullptr;
  void *d_cl_dtab = nullptr, *d_cl_fblk = nullptr, *d_cl_oblk = nullptr;      // descriptor tables of the template (see ClParams)
  int *d_cl_vdst = nullptr, *d_cl_fdst = nullptr;      // [ncl][128]: >= 0 CSR offset / row of a complete row, bit 31: offset into the partial-row buffer
  unsigned long long* d_cl_vdst64 = nullptr;           // ... as addresses, for the value array cl_val_base
  double* cl_val_base = nullptr;
  unsigned char *d_cl_map = nullptr, *d_cl_pmap = nullptr;
  unsigned char* d_cl_mapb = nullptr;       // carried plans: destination positions of the slots (k_cluster_maps)
  unsigned short* d_cl_gtab = nullptr;      // [8][27 * 27] template entry that child j OWNS at (local row, local column), 0xffff = another child's (fh_assembler_galerkin from the macro rows)
  bool cl_all_rows = false;                 // every row of every cluster lies in the matrix (no ghost rows, no sink): the macro rows can be read back
  bool macro_valid = false;                 // the matrix cl_val_base and the partial-row buffer hold the macro rows of the last assembly
  fh_mat_t cl_mat_of_macro = nullptr;       // the matrix of that assembly
  uint64_t macro_val_gen = 0;               // ... as long as nobody but SetPenalty has written the matrix since (fh_mat_s::val_gen)
  double* d_Pbuf = nullptr;
  int* d_cl_prow = nullptr;                 // rows of the second pass
  unsigned* d_cl_pstart = nullptr;          // [nprow + 1] their segments of the partial-row buffer
  // arguments of the last assembly (the element-wise Galerkin product re-creates the element rows from them when the fused path ran)
  int last_source_kind = 0;
  double last_params[2] = {1.0, 0.0};
  // optional fast path for AFFINE HEX27/Q2 elements (option assemble_affine): K_e = sum_ab det*B_ab * M_ab with the nine reference
  // matrices M_ab = sum_g w_g d_a phi_i d_b phi_j, B = J^-1 J^-T; curved elements keep the quadrature kernel
  int *d_aff_elems = nullptr, *d_gen_elems = nullptr;
  int n_aff = 0, n_gen = 0;
  double *d_Mab = nullptr, *d_mphi = nullptr;
  // matrix-core element kernel (k_elem_q2hex_mfma): reference gradients T[q][c][n] (q-stride 85, c-stride 28, zero padded) and
  // shape values Phi[q][n] (stride 33)
  double *d_mfT = nullptr, *d_mfPhi = nullptr;
  double* d_mfSFc = nullptr;     // sum-factorised map Jacobian: per-lane 1-D shape values [18][64] (null: tables are not tensor products)
  int* d_mfSFi = nullptr;        // ... and per-lane LDS offsets [7][64]
  // sum-factorised element kernel (k_elem_q2hex_sf): 1-D basis values (uniform operands) and the per-lane tables
  SfTab sf_tab;
  double* d_sfLc = nullptr;      // [SF_NLC][64]
  int* d_sfLi = nullptr;         // [SF_NLI][64]
  // source term given as a compiled expression (fh_expr): device copy of the program of the expression last used
  int* d_prog = nullptr;
  double* d_prog_consts = nullptr;
  int nprog = 0;
  std::vector<int> h_prog;
  std::vector<double> h_prog_consts;
};

struct AsmParams {
  const int* elems;        // element ids of this launch
  int nelems;
  const int* elem_dof;
  int nloc;
  const double* coords;
  const double* w;
  const double* phi;
  const double* dphi;
  int ng;
  const double* sol;       // may be null
  int source_kind;
  double p0, p1;
  const int* prog;         // source_kind 4: postfix program of the source expression f = p0 * expr(x, y, z, t)
  const double* prog_consts;
  int nprog;
  // scatter targets
  const int* rowptr;
  const int* col;
  int nrows;               // rows of the target matrix (owned rows on a distributed level); other rows are skipped
  double* val;
  double* res;
  const int* emap;         // may be null -> binary search
  int* emap_out;           // non-null: build the map instead of assembling
  int debug;               // profiling aid: bit 0 skips the quadrature loop, bit 1 skips the scatter / stores, bit 3 (host) skips the row pass
  const int* slot;         // non-null with Kout: row i of element e goes to Kout[slot[e*nc+i]*nc + j] (row-gather order), -1 = skip
  double* Kout;            // non-null: write element matrices [e][nc][nc] instead of scattering
  int kstride;             // doubles per element row in Kout (nc; 32 = padded rows of the slot-major buffer)
  int nsink;               // index of the spare row behind the slot-major buffer (rows the matrix does not hold)
  double* Fout;
};

template <int DIM, int NC>
struct AsmCfg {
  static constexpr int TI = (NC == 9) ? 3 : 4;
  static constexpr int TJ = TI;
  static constexpr int NBI = (NC + TI - 1) / TI;
  static constexpr int NBJ = (NC + TJ - 1) / TJ;
  static constexpr int NT = NBI * NBJ;                                        // tiles per element
  static constexpr int LPE = NT <= 1 ? 1 : NT <= 2 ? 2 : NT <= 4 ? 4 : NT <= 8 ? 8 : NT <= 16 ? 16 : NT <= 32 ? 32 : 64;
  static constexpr int EPW = 64 / LPE;                                        // elements per wave
  static constexpr int NSPLIT = (LPE == 64 || NC == 20) ? 4 : 1;              // node parts in phase 1 (HEX20: 8 Gauss points per chunk instead of 32, 47 kB of LDS per workgroup instead of 171)
  static constexpr int GC = LPE / NSPLIT;                                     // Gauss points per chunk
  static constexpr int NPP = (NC + NSPLIT - 1) / NSPLIT;                      // nodes per part
  static constexpr int NCP = NBI * TI;                                        // padded row length in LDS
  static constexpr int WAVES = 4;
  static constexpr int EPB = EPW * WAVES;                                     // elements per block
  // per-element LDS (doubles): gradients of the chunk, weights, grad u, f, coordinates, solution
  static constexpr int GS = GC * DIM * NCP;
  static constexpr int LDS_RAW = GS + GC + GC * NCP + NC * DIM + NC;
  static constexpr int LDS_PER_ELEM = (LDS_RAW + 1) / 2 * 2;                  // keep every element slab 16-byte aligned
};

__device__ __forceinline__ double source_eval(int kind, double p0, double p1, const double* xg, int dim) {
  if (kind == 0) return p0;
  if (kind == 3) {   // p0 * sum_d prod_{e != d} x_e (p1 - x_e): Laplacian of -(p0/2) prod_d x_d (p1 - x_d), a Q2 polynomial
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

// SRC: 0 constant source, 1 closed-form source (kinds 1-3), 2 compiled expression (kind 4) ; OUT: 0 scatter through emap, 1 scatter with binary search,
// 2 build emap (symbolic pass), 3 write element matrices
template <int DIM, int NC, int SRC, int OUT>
__global__ __launch_bounds__(256) void k_assemble_poisson(AsmParams P) {
  using C = AsmCfg<DIM, NC>;
  constexpr int TI = C::TI, TJ = C::TJ, NBJ = C::NBJ, LPE = C::LPE, EPW = C::EPW, NSPLIT = C::NSPLIT, GC = C::GC, NPP = C::NPP,
                NCP = C::NCP;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  __shared__ int s_dof[C::EPB][NC];

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int eiw = lane / LPE;              // element within the wave
  const int l = lane % LPE;                // lane within the element group
  const int slot = wave * EPW + eiw;       // element slot within the block
  const int eidx = blockIdx.x * C::EPB + slot;
  const bool elive = eidx < P.nelems;
  const int e = P.elems[elive ? eidx : (P.nelems - 1)];

  double* base = smem + (size_t)slot * C::LDS_PER_ELEM;
  double* gs = base;                       // [GC][DIM][NCP]
  double* ws = gs + C::GS;                 // [GC]
  double* rs = ws + GC;                    // [GC][NCP] residual integrand per (Gauss point, node)
  double* xe = rs + GC * NCP;              // [NC][DIM]
  double* ue = xe + NC * DIM;              // [NC]

  // ---- gather: dof ids, coordinates, solution (a8, a9: GetSolutionDof / GetSystemDof, nprocs = 1) ----------
  for (int n = l; n < NC; n += LPE) {
    const int dof = P.elem_dof[(size_t)e * P.nloc + n];
    s_dof[slot][n] = dof;
    for (int d = 0; d < DIM; d++) xe[n * DIM + d] = P.coords[(size_t)dof * DIM + d];
    ue[n] = P.sol ? P.sol[dof] : 0.0;
  }
  // zero the padding columns of gs / rs once
  if (NCP > NC) {
    for (int k = l; k < GC * DIM * (NCP - NC); k += LPE) {
      const int row = k / (NCP - NC), c = NC + k % (NCP - NC);
      gs[row * NCP + c] = 0.0;
    }
    for (int k = l; k < GC * (NCP - NC); k += LPE) rs[(k / (NCP - NC)) * NCP + NC + k % (NCP - NC)] = 0.0;
  }
  __syncthreads();

  // tile owned by this lane in phase 2
  const bool tlive = l < C::NT;
  const int ib = tlive ? l / NBJ : 0, jb = tlive ? l % NBJ : 0;
  const int i0 = ib * TI, j0 = jb * TJ;
  double K[TI][TJ];
  double F[TI];
#pragma unroll
  for (int a = 0; a < TI; a++) {
    F[a] = 0.0;
#pragma unroll
    for (int b = 0; b < TJ; b++) K[a][b] = 0.0;
  }

  const int q = l / NSPLIT, part = l % NSPLIT;
  const int n0 = part * NPP;
  const int nchunk = (P.debug & 1) ? 0 : (P.ng + GC - 1) / GC;
#pragma unroll 1
  for (int ch = 0; ch < nchunk; ch++) {
    // ---------------- phase 1: geometry at Gauss point g = ch*GC + q -------------------------------
    {
      const int g = ch * GC + q;
      const bool glive = g < P.ng;
      const int gg = glive ? g : 0;
      double dh[NPP][DIM];                 // reference gradients of this lane's nodes
      double J[DIM][DIM];
#pragma unroll
      for (int a = 0; a < DIM; a++)
#pragma unroll
        for (int b = 0; b < DIM; b++) J[a][b] = 0.0;
#pragma unroll
      for (int k = 0; k < NPP; k++) {
        const int n = n0 + k;
        if (n < NC) {
#pragma unroll
          for (int a = 0; a < DIM; a++) {
            dh[k][a] = P.dphi[((size_t)gg * NC + n) * DIM + a];
#pragma unroll
            for (int b = 0; b < DIM; b++) J[a][b] += dh[k][a] * xe[n * DIM + b];   // Jac[a][b] += dphi_a * vt[b][n]
          }
        } else {
#pragma unroll
          for (int a = 0; a < DIM; a++) dh[k][a] = 0.0;
        }
      }
      if (NSPLIT > 1) {
#pragma unroll
        for (int off = 1; off < NSPLIT; off <<= 1)
#pragma unroll
          for (int a = 0; a < DIM; a++)
#pragma unroll
            for (int b = 0; b < DIM; b++) J[a][b] += __shfl_xor(J[a][b], off, 64);
      }
      // adjugate / det as in the reference, with ONE reciprocal (9 fp64 divisions are ~350 instructions per Gauss point)
      double det, JI[DIM][DIM];
      if constexpr (DIM == 2) {
        det = J[0][0] * J[1][1] - J[0][1] * J[1][0];
        const double rd = 1.0 / det;
        JI[0][0] = J[1][1] * rd;
        JI[0][1] = -J[0][1] * rd;
        JI[1][0] = -J[1][0] * rd;
        JI[1][1] = J[0][0] * rd;
      } else {
        det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) + J[0][1] * (J[1][2] * J[2][0] - J[1][0] * J[2][2]) +
              J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
        const double rd = 1.0 / det;
        JI[0][0] = (-J[1][2] * J[2][1] + J[1][1] * J[2][2]) * rd;
        JI[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * rd;
        JI[0][2] = (-J[0][2] * J[1][1] + J[0][1] * J[1][2]) * rd;
        JI[1][0] = (J[1][2] * J[2][0] - J[1][0] * J[2][2]) * rd;
        JI[1][1] = (-J[0][2] * J[2][0] + J[0][0] * J[2][2]) * rd;
        JI[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * rd;
        JI[2][0] = (-J[1][1] * J[2][0] + J[1][0] * J[2][1]) * rd;
        JI[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * rd;
        JI[2][2] = (-J[0][1] * J[1][0] + J[0][0] * J[1][1]) * rd;
      }
      const double weight = glive ? det * P.w[gg] : 0.0;
      double gu[DIM], xg[DIM], ph[NPP];
#pragma unroll
      for (int a = 0; a < DIM; a++) gu[a] = xg[a] = 0.0;
#pragma unroll
      for (int k = 0; k < NPP; k++) {
        const int n = n0 + k;
        ph[k] = 0.0;
        if (n < NC) {
          const double un = ue[n];
          ph[k] = P.phi[(size_t)gg * NC + n];
#pragma unroll
          for (int a = 0; a < DIM; a++) {
            double s = dh[k][0] * JI[a][0];
#pragma unroll
            for (int b = 1; b < DIM; b++) s += dh[k][b] * JI[a][b];   // gradphi[a] = sum_b dphi_b * JacI[a][b]
            gs[(q * DIM + a) * NCP + n] = glive ? s : 0.0;
            gu[a] += s * un;
            if (SRC != 0) xg[a] += xe[n * DIM + a] * ph[k];
          }
        }
      }
      if (NSPLIT > 1) {
#pragma unroll
        for (int off = 1; off < NSPLIT; off <<= 1)
#pragma unroll
          for (int a = 0; a < DIM; a++) {
            gu[a] += __shfl_xor(gu[a], off, 64);
            if (SRC != 0) xg[a] += __shfl_xor(xg[a], off, 64);
          }
      }
      // residual integrand of this lane's nodes: (-f phi_n - grad phi_n . grad u) W  (summed over g in phase 2)
      double fq;
      if (SRC == 0) fq = P.p0;
      else if (SRC == 1) fq = source_eval(P.source_kind, P.p0, P.p1, xg, DIM);
      else {
        double x4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int a = 0; a < DIM; a++) x4[a] = xg[a];
        fq = P.p0 * fh_expr_device_eval(P.prog, P.nprog, P.prog_consts, x4);
      }
#pragma unroll
      for (int k = 0; k < NPP; k++) {
        const int n = n0 + k;
        if (n < NC) {
          double wl = 0.0;
#pragma unroll
          for (int a = 0; a < DIM; a++) {
            double s = dh[k][0] * JI[a][0];
#pragma unroll
            for (int b = 1; b < DIM; b++) s += dh[k][b] * JI[a][b];
            wl += s * gu[a];
          }
          rs[q * NCP + n] = (-fq * ph[k] - wl) * weight;
        }
      }
      if (part == 0) ws[q] = weight;
    }
    __syncthreads();
    // ---------------- phase 2: K += (grad phi_i . grad phi_j) W ; F += (-f phi_i - grad phi_i . grad u) W ---------
    if (tlive) {
#pragma unroll 2
      for (int gq = 0; gq < GC; gq++) {
        const double wq = ws[gq];
#pragma unroll
        for (int d = 0; d < DIM; d++) {
          const double* row = gs + (gq * DIM + d) * NCP;
          double av[TI], bv[TJ];
          if constexpr (TI == 4) {   // rows are 32-byte aligned (NCP % 4 == 0, slabs 16-byte aligned): two ds_read_b128 per side
            const double2* ra = reinterpret_cast<const double2*>(__builtin_assume_aligned(row + i0, 16));
            const double2* rb = reinterpret_cast<const double2*>(__builtin_assume_aligned(row + j0, 16));
            const double2 a0 = ra[0], a1 = ra[1], b0 = rb[0], b1 = rb[1];
            av[0] = a0.x; av[1] = a0.y; av[2] = a1.x; av[3] = a1.y;
            bv[0] = b0.x * wq; bv[1] = b0.y * wq; bv[2] = b1.x * wq; bv[3] = b1.y * wq;
          } else {
#pragma unroll
            for (int a = 0; a < TI; a++) av[a] = row[i0 + a];
#pragma unroll
            for (int b = 0; b < TJ; b++) bv[b] = row[j0 + b] * wq;
          }
#pragma unroll
          for (int a = 0; a < TI; a++)
#pragma unroll
            for (int b = 0; b < TJ; b++) K[a][b] += av[a] * bv[b];
        }
        if (jb == 0) {
#pragma unroll
          for (int a = 0; a < TI; a++) F[a] += rs[gq * NCP + i0 + a];
        }
      }
    }
    __syncthreads();
  }

  if (!elive || !tlive || (P.debug & 2)) return;
  // ---------------- output ---------------------------------------------------------------------------------------
  if constexpr (OUT == 3) {
#pragma unroll
    for (int a = 0; a < TI; a++) {
      const int i = i0 + a;
      if (i >= NC) continue;
#pragma unroll
      for (int b = 0; b < TJ; b++)
        if (j0 + b < NC) {
          if (P.slot) {
            const int sl = P.slot[(size_t)e * NC + i];
            if (sl >= 0) P.Kout[(size_t)sl * P.kstride + j0 + b] = K[a][b];
          } else {
            P.Kout[((size_t)eidx * NC + i) * NC + j0 + b] = K[a][b];
          }
        }
      if (jb == 0) {
        if (P.slot) {
          const int sl = P.slot[(size_t)e * NC + i];
          if (sl >= 0) P.Fout[sl] = F[a];
        } else {
          P.Fout[(size_t)eidx * NC + i] = F[a];
        }
      }
    }
    return;
  }
#pragma unroll
  for (int a = 0; a < TI; a++) {
    const int i = i0 + a;
    if (i >= NC) continue;
    const int row = s_dof[slot][i];
    if (row >= P.nrows) continue;     // row owned by another rank
    const int rs = P.rowptr[row], re = P.rowptr[row + 1];
#pragma unroll
    for (int b = 0; b < TJ; b++) {
      const int j = j0 + b;
      if (j >= NC) continue;
      int pos;
      if constexpr (OUT == 0) {
        pos = P.emap[((size_t)e * NC + i) * NCP + j];
      } else {
        const int target = s_dof[slot][j];
        int lo = rs, hi = re - 1;
        pos = -1;
        while (lo <= hi) {
          const int mid = lo + ((hi - lo) >> 1);   // (lo + hi) overflows beyond 2^30 non-zeros
          const int cc = P.col[mid];
          if (cc == target) { pos = mid; break; }
          if (cc < target) lo = mid + 1; else hi = mid - 1;
        }
      }
      if constexpr (OUT == 2) P.emap_out[((size_t)e * NC + i) * NCP + j] = pos;
      else if (pos >= 0) P.val[pos] += K[a][b];      // add_matrix_blocked: race-free inside one colour
    }
    if (OUT != 2 && jb == 0) P.res[row] += F[a];     // add_vector_blocked
  }
}

// ------------------------------------------------------------------------------------------------------------------
// pass 2 of the two-pass assembly: one 32-lane group per CSR row.  The row's accumulators live in LDS; for every adjacent
// element (ascending element order = the order of the reference's sequential element loop) lane j adds K_e[i][j] into its
// slot, then the finished row is written once, contiguously.  No colours, no atomics, no read-modify-write of HBM:
// the coloured scatter touches 27 of the 125 entries of a row per visit and pays for whole 128-byte lines (measured 3.0 ms
// on the 64^3 level); this pass moves each matrix value exactly once.
// BUILD=true fills rowmap (symbolic pass, once per pattern).
// ------------------------------------------------------------------------------------------------------------------
template <int NC, bool BUILD>
__global__ __launch_bounds__(256) void k_row_assemble(const int* __restrict__ rowptr, const int* __restrict__ col, int m,
                                                      const int* __restrict__ adj_ptr, const int* __restrict__ adj_ei,
                                                      unsigned char* __restrict__ rowmap, const int* __restrict__ elem_dof, int nloc,
                                                      const double* __restrict__ Kbuf, int kstride, const double* __restrict__ Fbuf,
                                                      double* __restrict__ val, double* __restrict__ res) {
  __shared__ double acc[8][256];
  const int sub = threadIdx.x >> 5, lane = threadIdx.x & 31;
  const int r = blockIdx.x * 8 + sub;
  if (r >= m) return;
  const int rs = rowptr[r], len = rowptr[r + 1] - rs;
  if (!BUILD)
    for (int p = lane; p < len; p += 32) acc[sub][p] = 0.0;
  double facc = 0.0;
  const int a0 = adj_ptr[r], a1 = adj_ptr[r + 1];
  if (BUILD) {
    for (int a = a0; a < a1; a++) {
      const int ei = adj_ei[a];
      const int e = ei >> 5;
      for (int j = lane; j < NC; j += 32) {
        const int target = elem_dof[(size_t)e * nloc + j];
        int lo = rs, hi = rs + len - 1, pos = 0;
        while (lo <= hi) {
          const int mid = lo + ((hi - lo) >> 1);   // (lo + hi) overflows beyond 2^30 non-zeros
          const int cc = col[mid];
          if (cc == target) { pos = mid - rs; break; }
          if (cc < target) lo = mid + 1; else hi = mid - 1;
        }
        rowmap[(size_t)a * NC + j] = (unsigned char)pos;
      }
    }
  } else {
    // batches of RB adjacent elements: all loads of a batch are issued before the (ordered) LDS accumulation
    constexpr int RB = 4;
    for (int ab = a0; ab < a1; ab += RB) {
      double k[RB];
      int pp[RB];
      double f[RB];
#pragma unroll
      for (int t = 0; t < RB; t++) {
        const int a = ab + t;
        k[t] = 0.0;
        pp[t] = 0;
        f[t] = 0.0;
        if (a < a1) {
          if (lane < NC) {
            k[t] = Kbuf[(size_t)a * kstride + lane];     // slot-major: the rows of one CSR row are contiguous
            pp[t] = rowmap[(size_t)a * NC + lane];
          }
          if (lane == 0) f[t] = Fbuf[a];
        }
      }
#pragma unroll
      for (int t = 0; t < RB; t++) {
        if (ab + t < a1 && lane < NC) acc[sub][pp[t]] += k[t];   // distinct slots within one element row; in-order LDS per wave
        facc += f[t];
      }
    }
  }
  if (!BUILD) {
    for (int p = lane; p < len; p += 32) val[rs + p] = acc[sub][p];
    if (lane == 0) res[r] = facc;
  }
}

// The same pass for matrices whose rows have at most 128 entries (Q2 on hexes: 125): every 32-lane group works on TWO rows at
// once -- the pass is bound by the dependent round trips adjacency -> element rows -> store of each row, so the loads of both
// rows are issued together (twice the bytes in flight at the same LDS footprint).
template <int NC, int NT>
__global__ __launch_bounds__(256) void k_row_assemble2_t(const int* __restrict__ rowptr, int m, const int* __restrict__ adj_ptr,
                                                       const unsigned char* __restrict__ rowmap, const double* __restrict__ Kbuf, int kstride,
                                                       const double* __restrict__ Fbuf, double* __restrict__ val, double* __restrict__ res) {
  __shared__ double acc[8][2][128];
  const int sub = threadIdx.x >> 5, lane = threadIdx.x & 31;
  const int r0 = (blockIdx.x * 8 + sub) * 2;
  if (r0 >= m) return;
  const bool two = r0 + 1 < m;
  const int rsA = rowptr[r0], rsB = rowptr[r0 + 1], reB = two ? rowptr[r0 + 2] : rsB;
  const int aA0 = adj_ptr[r0], aB0 = adj_ptr[r0 + 1], aB1 = two ? adj_ptr[r0 + 2] : aB0;
  const int lenA = rsB - rsA, lenB = reB - rsB;
  for (int p = lane; p < 128; p += 32) {
    acc[sub][0][p] = 0.0;
    acc[sub][1][p] = 0.0;
  }
  double fA = 0.0, fB = 0.0;
  constexpr int RB = 4;
  const int nA = aB0 - aA0, nB = aB1 - aB0;
  for (int ab = 0; ab < max(nA, nB); ab += RB) {
    double kA[RB], kB[RB], gA[RB], gB[RB];
    int pA[RB], pB[RB];
#pragma unroll
    for (int t = 0; t < RB; t++) {
      const int iA = ab + t, iB = ab + t;
      kA[t] = kB[t] = gA[t] = gB[t] = 0.0;
      pA[t] = pB[t] = 0;
      if (iA < nA) {
        const int a = aA0 + iA;
        if (lane < NC) {
          kA[t] = NT ? __builtin_nontemporal_load(&Kbuf[(size_t)a * kstride + lane]) : Kbuf[(size_t)a * kstride + lane];
          pA[t] = rowmap[(size_t)a * NC + lane];
        }
        if (lane == 0) gA[t] = Fbuf[a];
      }
      if (iB < nB) {
        const int a = aB0 + iB;
        if (lane < NC) {
          kB[t] = NT ? __builtin_nontemporal_load(&Kbuf[(size_t)a * kstride + lane]) : Kbuf[(size_t)a * kstride + lane];
          pB[t] = rowmap[(size_t)a * NC + lane];
        }
        if (lane == 0) gB[t] = Fbuf[a];
      }
    }
#pragma unroll
    for (int t = 0; t < RB; t++) {          // ascending element order per row, as in the one-row kernel
      if (ab + t < nA && lane < NC) acc[sub][0][pA[t]] += kA[t];
      if (ab + t < nB && lane < NC) acc[sub][1][pB[t]] += kB[t];
      fA += gA[t];
      fB += gB[t];
    }
  }
  if (NT & 2) {
    for (int p = lane; p < lenA; p += 32) __builtin_nontemporal_store(acc[sub][0][p], &val[rsA + p]);
    for (int p = lane; p < lenB; p += 32) __builtin_nontemporal_store(acc[sub][1][p], &val[rsB + p]);
  } else {
    for (int p = lane; p < lenA; p += 32) val[rsA + p] = acc[sub][0][p];
    for (int p = lane; p < lenB; p += 32) val[rsB + p] = acc[sub][1][p];
  }
  if (lane == 0) {
    res[r0] = fA;
    if (two) res[r0 + 1] = fB;
  }
}

template <int NC>
static int launch_rows(fh_assembler_t as, fh_mat_t A, double* res, bool build) {
  if (A->m == 0) return 0;
  const dim3 grid(fh_div_up(A->m, 8)), block(256);
  if (build)
    hipLaunchKernelGGL((k_row_assemble<NC, true>), grid, block, 0, as->ctx->stream, A->d_rowptr, A->d_col, A->m, as->d_adj_ptr, as->d_adj_ei,
                       as->d_rowmap, as->d_elem_dof, as->nloc, nullptr, as->kstride, nullptr, nullptr, nullptr);
  else if (A->max_row <= 128 && as->ctx->assemble_rows2)
  {
    const int nt = as->ctx->assemble_rows_nt;
    const dim3 g2(fh_div_up(A->m, 16));
#define FH_ROWS2(NTV) hipLaunchKernelGGL((k_row_assemble2_t<NC, NTV>), g2, block, 0, as->ctx->stream, A->d_rowptr, A->m, as->d_adj_ptr, as->d_rowmap, \
                                          as->d_Kbuf, as->kstride, as->d_Fbuf, A->d_val, res)
    if (nt == 1) FH_ROWS2(1);
    else if (nt == 2) FH_ROWS2(2);
    else if (nt == 3) FH_ROWS2(3);
    else FH_ROWS2(0);
#undef FH_ROWS2
  }
  else
    hipLaunchKernelGGL((k_row_assemble<NC, false>), grid, block, 0, as->ctx->stream, A->d_rowptr, A->d_col, A->m, as->d_adj_ptr, as->d_adj_ei,
                       as->d_rowmap, as->d_elem_dof, as->nloc, as->d_Kbuf, as->kstride, as->d_Fbuf, A->d_val, res);
  FH_CHECK_HIP(hipGetLastError());
  return 0;
}

static int dispatch_rows(fh_assembler_t as, fh_mat_t A, double* res, bool build) {
  switch (as->nc) {
    case 27: return launch_rows<27>(as, A, res, build);
    case 20: return launch_rows<20>(as, A, res, build);
    case 9: return launch_rows<9>(as, A, res, build);
    case 8: return launch_rows<8>(as, A, res, build);
    case 4: return launch_rows<4>(as, A, res, build);
  }
  fh_set_error("assembler: unsupported nc %d", as->nc);
  return 2;
}

// ------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------------------------
// HEX27 / Q2 element-matrix kernel for the two-pass assembly: TWO elements per wave, symmetric tiles only.
// K_e is symmetric (the reference's Jac[i][j] = Jac[j][i] bit for bit), so only the 28 upper 4x4 tiles are accumulated and
// mirrored on output: half the FP64 work of the full 49-tile version.  Lanes 0-31 own element A, 32-63 element B:
//   phase 1  (8 Gauss points of the chunk) x (4 node parts) per element: partial J, shuffles, J^-1, grad phi -> LDS,
//            residual integrand per (g, node) -> LDS
//   phase 2  lane < 28 of each half: one upper tile, 16 FMA per 8 LDS doubles (2+2 ds_read_b128)
// Waves are independent (own LDS slab, wave-level barriers only); one wave per workgroup.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int SRC>
__global__ __launch_bounds__(64) void k_elem_q2hex_sym(AsmParams P) {
  constexpr int NC = 27, NCP = 28, DIM = 3, GC = 8, NPP = 7;
  constexpr int GS = GC * DIM * NCP;                 // 672
  constexpr int SLAB = GS + GC * NCP + GC + NC * DIM + NC + 4;   // gs, rs, ws, xe, ue (+pad) = 1016 -> even
  __shared__ __attribute__((aligned(16))) double smem[2 * SLAB];
  const int lane = threadIdx.x;
  const int half = lane >> 5, l = lane & 31;
  const int eidx = blockIdx.x * 2 + half;
  const bool elive = eidx < P.nelems;
  const int e = P.elems[elive ? eidx : (P.nelems - 1)];
  double* gs = smem + half * SLAB;
  double* rs = gs + GS;
  double* ws = rs + GC * NCP;
  double* xe = ws + GC;
  double* ue = xe + NC * DIM;

  if (l < NC) {
    const int dof = P.elem_dof[(size_t)e * P.nloc + l];
#pragma unroll
    for (int d = 0; d < DIM; d++) xe[l * DIM + d] = P.coords[(size_t)dof * DIM + d];
    ue[l] = P.sol ? P.sol[dof] : 0.0;
  }
  if (l < GC * DIM) gs[l * NCP + NC] = 0.0;           // padding column of the gradient rows
  if (l < GC) rs[l * NCP + NC] = 0.0;
  wave_lds_sync();

  // upper tile (ib <= jb) owned by this lane: l = 0..27
  int ib = 0, jb = 0;
  {
    int t = (l < 28) ? l : 0, row = 0;
    while (t >= 7 - row) { t -= 7 - row; row++; }
    ib = row;
    jb = row + t;
  }
  const bool tlive = l < 28;
  const int i0 = ib * 4, j0 = jb * 4;
  double K[4][4], F[4];
#pragma unroll
  for (int a = 0; a < 4; a++) {
    F[a] = 0.0;
#pragma unroll
    for (int b = 0; b < 4; b++) K[a][b] = 0.0;
  }
  const int q = l >> 2, part = l & 3, n0 = part * NPP;
  const int nchunk = (P.debug & 1) ? 0 : (P.ng + GC - 1) / GC;
#pragma unroll 1
  for (int ch = 0; ch < nchunk; ch++) {
    {
      const int g = ch * GC + q;
      const bool glive = g < P.ng;
      const int gg = glive ? g : 0;
      double dh[NPP][DIM], J[DIM][DIM];
#pragma unroll
      for (int a = 0; a < DIM; a++)
#pragma unroll
        for (int b = 0; b < DIM; b++) J[a][b] = 0.0;
#pragma unroll
      for (int k = 0; k < NPP; k++) {
        const int n = n0 + k;
        if (n < NC) {
#pragma unroll
          for (int a = 0; a < DIM; a++) {
            dh[k][a] = P.dphi[((size_t)gg * NC + n) * DIM + a];
#pragma unroll
            for (int b = 0; b < DIM; b++) J[a][b] += dh[k][a] * xe[n * DIM + b];
          }
        } else {
#pragma unroll
          for (int a = 0; a < DIM; a++) dh[k][a] = 0.0;
        }
      }
#pragma unroll
      for (int off = 1; off < 4; off <<= 1)
#pragma unroll
        for (int a = 0; a < DIM; a++)
#pragma unroll
          for (int b = 0; b < DIM; b++) J[a][b] += __shfl_xor(J[a][b], off, 64);
      const double det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) + J[0][1] * (J[1][2] * J[2][0] - J[1][0] * J[2][2]) +
                         J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
      const double rd = 1.0 / det;
      double JI[DIM][DIM];
      JI[0][0] = (-J[1][2] * J[2][1] + J[1][1] * J[2][2]) * rd;
      JI[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * rd;
      JI[0][2] = (-J[0][2] * J[1][1] + J[0][1] * J[1][2]) * rd;
      JI[1][0] = (J[1][2] * J[2][0] - J[1][0] * J[2][2]) * rd;
      JI[1][1] = (-J[0][2] * J[2][0] + J[0][0] * J[2][2]) * rd;
      JI[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * rd;
      JI[2][0] = (-J[1][1] * J[2][0] + J[1][0] * J[2][1]) * rd;
      JI[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * rd;
      JI[2][2] = (-J[0][1] * J[1][0] + J[0][0] * J[1][1]) * rd;
      const double weight = glive ? det * P.w[gg] : 0.0;
      double gr[NPP][DIM], gu[DIM] = {0.0, 0.0, 0.0}, xg[DIM] = {0.0, 0.0, 0.0}, ph[NPP];
#pragma unroll
      for (int k = 0; k < NPP; k++) {
        const int n = n0 + k;
        ph[k] = 0.0;
#pragma unroll
        for (int a = 0; a < DIM; a++) gr[k][a] = 0.0;
        if (n < NC) {
          const double un = ue[n];
          ph[k] = P.phi[(size_t)gg * NC + n];
#pragma unroll
          for (int a = 0; a < DIM; a++) {
            const double sgr = glive ? dh[k][0] * JI[a][0] + dh[k][1] * JI[a][1] + dh[k][2] * JI[a][2] : 0.0;
            gr[k][a] = sgr;
            gs[(q * DIM + a) * NCP + n] = sgr;
            gu[a] += sgr * un;
            if (SRC != 0) xg[a] += xe[n * DIM + a] * ph[k];
          }
        }
      }
#pragma unroll
      for (int off = 1; off < 4; off <<= 1)
#pragma unroll
        for (int a = 0; a < DIM; a++) {
          gu[a] += __shfl_xor(gu[a], off, 64);
          if (SRC != 0) xg[a] += __shfl_xor(xg[a], off, 64);
        }
      double fq;
      if (SRC == 0) fq = P.p0;
      else if (SRC == 1) fq = source_eval(P.source_kind, P.p0, P.p1, xg, DIM);
      else {
        double x4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int a = 0; a < DIM; a++) x4[a] = xg[a];
        fq = P.p0 * fh_expr_device_eval(P.prog, P.nprog, P.prog_consts, x4);
      }
#pragma unroll
      for (int k = 0; k < NPP; k++) {
        const int n = n0 + k;
        if (n < NC) rs[q * NCP + n] = (-fq * ph[k] - (gr[k][0] * gu[0] + gr[k][1] * gu[1] + gr[k][2] * gu[2])) * weight;
      }
      if (part == 0) ws[q] = weight;
    }
    wave_lds_sync();
    if (tlive && !(P.debug & 4)) {
#pragma unroll 2
      for (int gq = 0; gq < GC; gq++) {
        const double wq = ws[gq];
#pragma unroll
        for (int d = 0; d < DIM; d++) {
          const double* row = gs + (gq * DIM + d) * NCP;
          const double2* ra = reinterpret_cast<const double2*>(__builtin_assume_aligned(row + i0, 16));
          const double2* rb = reinterpret_cast<const double2*>(__builtin_assume_aligned(row + j0, 16));
          const double2 a0 = ra[0], a1 = ra[1], b0 = rb[0], b1 = rb[1];
          const double av[4] = {a0.x, a0.y, a1.x, a1.y};
          const double bv[4] = {b0.x * wq, b0.y * wq, b1.x * wq, b1.y * wq};
#pragma unroll
          for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 4; b++) K[a][b] += av[a] * bv[b];
        }
        if (ib == jb) {
#pragma unroll
          for (int a = 0; a < 4; a++) F[a] += rs[gq * NCP + i0 + a];
        }
      }
    }
    wave_lds_sync();
  }
  if (!elive || !tlive || (P.debug & 2)) return;
  // output: row i of K_e -> slot of (e, i) in the row-gather buffer (or the plain [e][i][j] layout when no slot map is given)
  int sli[4], slj[4];
#pragma unroll
  for (int a = 0; a < 4; a++) {
    const int i = i0 + a, j = j0 + a;
    sli[a] = (i < NC) ? (P.slot ? P.slot[(size_t)e * NC + i] : eidx * NC + i) : -1;
    slj[a] = (j < NC) ? (P.slot ? P.slot[(size_t)e * NC + j] : eidx * NC + j) : -1;
  }
#pragma unroll
  for (int a = 0; a < 4; a++) {
    const int i = i0 + a;
    if (i >= NC) continue;
#pragma unroll
    for (int b = 0; b < 4; b++) {
      const int j = j0 + b;
      if (j >= NC) continue;
      if (ib == jb) {
        if (b >= a) {               // diagonal tile: upper entries, mirrored
          if (sli[a] >= 0) P.Kout[(size_t)sli[a] * P.kstride + j] = K[a][b];
          if (b > a && slj[b] >= 0) P.Kout[(size_t)slj[b] * P.kstride + i] = K[a][b];
        }
      } else {
        if (sli[a] >= 0) P.Kout[(size_t)sli[a] * P.kstride + j] = K[a][b];
        if (slj[b] >= 0) P.Kout[(size_t)slj[b] * P.kstride + i] = K[a][b];
      }
    }
    if (ib == jb && sli[a] >= 0) P.Fout[sli[a]] = F[a];
  }
}

// ------------------------------------------------------------------------------------------------------------------
// HEX27 / Q2 element matrices on the FP64 matrix cores: ONE element per wave, v_mfma_f64_4x4x4_4b_f64 (four independent
// 4x4x4 products per instruction, 16 cycles, the same 64 MAC/clk/SIMD as the 16x16x4 form but at the granularity of the
// 27 x 27 problem).
//   K_e = sum_q T_q^T D_q T_q,  T_q[c][i] = d phi_i / d xi_c at Gauss point q (table),  D_q = w_q det J (J^-1)^T J^-1  (3 x 3)
//   i.e. K = A^T B with A[(q,c)][i] = T_q[c][i] read straight from the table in LDS and B[(q,c)][j] = sum_c' D_q[c][c'] T_q[c'][j]
//   (3 FMA per value): the 27 x 27 x 192 FMAs of the reference's i/j/gauss loop (`00_poisson_eqn_..._separate.hpp:170-200`).
//   The nodes form 7 groups of 4 (node 27 = zero padding); K_e is symmetric, so only the 28 tiles (ib <= jb) of the 7 x 7 tile grid
//   are needed: 7 instructions of 4 tiles per k-step of 4 (schedule below), 48 k-steps -> 336 MFMAs = 5376 cycles per element.
// Operand layout (measured, tests/cpp/mfma_f64_4x4_layout.cpp): A and B: lane = 16 k + 4 block + r ; D: lane = 16 row + 4 block + col.
//   The lane (k, b, r) of a k-step works on Gauss point q0 + 16 k.  It forms B for its own node li = 4 b + r (column group b,
//   "lo") and for node 16 + li (column group 4 + b, "hi"; block 3 has no such node and takes column group 3 again) and reads A
//   for the row group the schedule gives its block:
//       lo-type (B = lo, columns 0,1,2,3):  t0: rows (0,1,2,3)  t1: (2,0,1,0)  t2: (5,4,4,2)  t3: (6,6,5,4)
//       hi-type (B = hi, columns 4,5,6,3):  t4: rows (4,5,6,5)  t5: (6,4,5,6)  t6: (0,1,2,1)
//   (no instruction has row groups g and g + 4 in different blocks: the two Gauss points of a half-wave sit 16 doubles apart
//   mod 32 in LDS, so such a pair would be a 2-way bank conflict of the A-operand read)
//   28 slots = the 28 unordered tile pairs, each exactly once (an orientation of the pair graph with in-degree 4,4,4,7,3,3,3;
//   a tile with row > column is the transpose of the needed one).  t0 holds the diagonal tiles 0..3: its A operand is the lane's
//   own table value.
// MFMAs do not overlap with vector instructions of other waves on gfx950 (tests/cpp/mfma_f64_overlap_probe.cpp), so the cost is
// MFMA cycles + 4 x (vector instructions): phase B issues 18 FMAs per 21 MFMAs, A operands cost LDS reads only.
// Per element:   phase A  map Jacobian by sum factorisation (three contractions through LDS, lanes = (q1,b,c) / (c,q1,q2) / Gauss
//                         point; the direct 27-node loop when the tables are not tensor products), then per Gauss point the
//                         cofactors, D_q and the source value -> per-wave LDS slab
//                phase B  16 groups of the 4 Gauss points {q0, q0+16, q0+32, q0+48} x 3 directions x 7 MFMAs; operand loads issued
//                         by hand (ds_read_b64 + s_waitcnt, register double buffer)
//                phase C  source integral per node, tiles -> LDS (mirrored, offsets from a table), K_e u for the residual, one
//                         256-byte row per half-wave store off a scalar base address
// Node ids are prefetched two elements ahead, coordinates / solution / slots one element ahead (dependent gathers).
// Tables (shared by the NW waves of the persistent workgroup): T[q][c][n] with q-stride 85 doubles, c-stride 28 (odd q-stride:
// conflict-free for lane = q; 16*85 = 16 mod 32: conflict-free for the k-step pattern), Phi[q][n] with stride 33, the per-lane 1-D
// shape values of the sum factorisation [18][64], the staging offsets [14][64].  Waves never synchronise with each other after
// the table load.  ng == 64 only; other rules keep the vector kernel.
// ------------------------------------------------------------------------------------------------------------------
typedef const int __attribute__((address_space(4)))* fh_ciptr;     // read-only for the launch: uniform accesses become s_load
constexpr int MF_TS = 85, MF_TA = 28, MF_PS = 33, MF_SS = 7, MF_KS = 29;
constexpr int MF_SLAB = 27 * MF_KS;   // 783 doubles per wave: phase-A results (64 x 7), later the 27 x 29 staging of K_e
constexpr int MF_XS = 27 * 4;         // per wave: (x, y, z, u) of the element's nodes
constexpr int MF_WAVE = MF_SLAB + 1 + MF_XS;   // 892 doubles, even: xs stays 16-byte aligned
constexpr int MF_SF = 18 * 64;        // per-lane 1-D shape values of the sum-factorised Jacobian
constexpr size_t mf_lds_bytes(int nw) { return (size_t)(64 * MF_TS + 64 * MF_PS + MF_SF + nw * MF_WAVE) * sizeof(double) + 2 * 7 * 64 * sizeof(int); }
// row group of block b in instruction t: 3 bits each
constexpr unsigned long long mf_rows(int b0, int b1, int b2, int b3) { return (unsigned long long)(b0 | (b1 << 3) | (b2 << 6) | (b3 << 9)); }
constexpr unsigned long long MF_SCHED_LO = mf_rows(0, 1, 2, 3) | (mf_rows(2, 0, 1, 0) << 12) | (mf_rows(5, 4, 4, 2) << 24) | (mf_rows(6, 6, 5, 4) << 36);
constexpr unsigned long long MF_SCHED_HI = mf_rows(4, 5, 6, 5) | (mf_rows(6, 4, 5, 6) << 12) | (mf_rows(0, 1, 2, 1) << 24);
constexpr int MF_NT = 7;              // MFMAs per k-step

__device__ __forceinline__ int mf_rowg(int t, int blk) { return (int)(((t < 4 ? MF_SCHED_LO : MF_SCHED_HI) >> (12 * (t & 3) + 3 * blk)) & 7); }

template <int SRC, int NW>
__global__ __launch_bounds__(NW * 64) void k_elem_q2hex_mfma(AsmParams P, const double* __restrict__ Tg, const double* __restrict__ Phig, const double* __restrict__ SFc,
                                                          const int* __restrict__ SFi) {
  constexpr int NC = 27, DIM = 3;
  extern __shared__ __attribute__((aligned(16))) double mf_smem[];
  double* T = mf_smem;
  double* Phi = T + 64 * MF_TS;
  for (int k = threadIdx.x; k < 64 * MF_TS; k += NW * 64) T[k] = Tg[k];
  for (int k = threadIdx.x; k < 64 * MF_PS; k += NW * 64) Phi[k] = Phig[k];
  double* SFl = Phi + 64 * MF_PS;                         // [18][64] 1-D shape values per lane (sum factorisation)
  if (SFc)
    for (int k = threadIdx.x; k < MF_SF; k += NW * 64) SFl[k] = SFc[k];
  int* KsOff = reinterpret_cast<int*>(SFl + MF_SF + NW * MF_WAVE);   // [2 * MF_NT][64]: where a lane's tile entries go in the staging
  if (threadIdx.x < 64) {
    const int l = threadIdx.x, k4 = l >> 4, b4 = (l >> 2) & 3, r = l & 3;   // D layout: row = lane>>4, block = (lane>>2)&3, col = lane&3
    for (int t = 0; t < MF_NT; t++) {
      const int colg = (t < 4 || b4 == 3) ? b4 : 4 + b4, rowg = mf_rowg(t, b4);
      const int row = 4 * rowg + k4, col = 4 * colg + r;
      // diagonal tiles: the (i, j) and (j, i) sums differ in rounding; keep the upper entries and mirror them, so that K_e is
      // symmetric bit for bit like the reference's Jac (products commute, same summation order)
      const bool live = row < 27 && col < 27 && (rowg != colg || k4 <= r);
      KsOff[(2 * t) * 64 + l] = live ? row * MF_KS + col : -1;
      KsOff[(2 * t + 1) * 64 + l] = (live && row != col) ? col * MF_KS + row : -1;
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double* slab = SFl + MF_SF + wave * MF_WAVE;
  int sfi_x[3] = {0, 0, 0}, sfi_b1 = 0, sfi_b2 = 0, sfi_bv = 0, sfi_b3 = 0;
  if (SFc) {
    sfi_x[0] = SFi[lane]; sfi_x[1] = SFi[64 + lane]; sfi_x[2] = SFi[128 + lane];
    sfi_b1 = SFi[192 + lane]; sfi_b2 = SFi[256 + lane]; sfi_bv = SFi[320 + lane]; sfi_b3 = SFi[384 + lane];
  }
  double* xs = slab + MF_SLAB + 1;
  const int kk = lane >> 4, li = lane & 15, blk = (lane >> 2) & 3, r4 = lane & 3;
  const int ln = lane < NC ? lane : 0;
  // per-lane row groups of the 8 instructions and the table offsets of the A operands
  // LDS byte addresses of this lane's operands (group 0, direction 0); the k-step loop adds immediate offsets
  const unsigned ldsT = (unsigned)(size_t)(__attribute__((address_space(3))) double*)T;
  const unsigned ldsSlab = (unsigned)(size_t)(__attribute__((address_space(3))) double*)slab;
  unsigned aA[MF_NT];                // aA[0] unused: the A operand of t0 is the lane's own table value (rows = columns)
#pragma unroll
  for (int t = 0; t < MF_NT; t++) aA[t] = ldsT + (kk * 16 * MF_TS + 4 * mf_rowg(t, blk) + r4) * 8;
  const unsigned aBlo = ldsT + (kk * 16 * MF_TS + li) * 8;
  const unsigned aBhi = ldsT + (kk * 16 * MF_TS + (li < 12 ? 16 + li : li)) * 8;      // block 3 of the "hi" operand: column group 3 again
  const unsigned aD = ldsSlab + kk * 16 * MF_SS * 8;
  const double wgauss = P.w[lane];
  const fh_ciptr elems = (fh_ciptr)P.elems;
  const int stride = gridDim.x * NW;
  const int idx0 = blockIdx.x * NW + wave;
  if (idx0 >= P.nelems) return;
  // software pipeline over this wave's elements: node ids two elements ahead, coordinates / solution values / output slots
  // one element ahead, so that no phase waits on an HBM round trip (the gathers are dependent loads)
  const int last = P.nelems - 1;
  int e_cur = elems[idx0];
  int e_n = elems[min(idx0 + stride, last)], e_nn = elems[min(idx0 + 2 * stride, last)];
  int sl_cur = (lane < NC) ? (P.slot ? P.slot[(size_t)e_cur * NC + lane] : idx0 * NC + lane) : -1;
  {
    const int dof = P.elem_dof[(size_t)e_cur * P.nloc + ln];
    if (lane < NC) {
      xs[lane * 4 + 0] = P.coords[(size_t)dof * DIM];
      xs[lane * 4 + 1] = P.coords[(size_t)dof * DIM + 1];
      xs[lane * 4 + 2] = P.coords[(size_t)dof * DIM + 2];
      xs[lane * 4 + 3] = P.sol ? P.sol[dof] : 0.0;
    }
  }
  int dof_n = P.elem_dof[(size_t)e_n * P.nloc + ln];
  wave_lds_sync();
#pragma unroll 1
  for (int idx = idx0; idx < P.nelems; idx += stride) {
    // ---- prefetch (consumed at the end of this iteration / in the next one) ----
    const double nx0 = P.coords[(size_t)dof_n * DIM], nx1 = P.coords[(size_t)dof_n * DIM + 1], nx2 = P.coords[(size_t)dof_n * DIM + 2];
    const double nu = P.sol ? P.sol[dof_n] : 0.0;
    const int sl_n = (lane < NC) ? (P.slot ? P.slot[(size_t)e_n * NC + lane] : (idx + stride) * NC + lane) : -1;
    const int dof_nn = P.elem_dof[(size_t)e_nn * P.nloc + ln];
    const int e_nnn = elems[min(idx + 3 * stride, last)];
    // ---- phase A: lane = Gauss point ----
    {
      const int q = lane;
      double J[DIM][DIM] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, xg[DIM] = {0, 0, 0};
      const double* Tq = T + q * MF_TS;
      if (SFc) {
        // sum factorisation: J_q = sum_abc dl_a(q1) l_b(q2) l_c(q3) x_abc ... in three contractions through LDS (U, V alias
        // the slab, which is written only after the last read below):
        //   stage 1, lane = (q1, b, c) < 36:    U [b][c][q1] = sum_a l_a(q1) x_abc,   U'[b][c][q1] = sum_a dl_a(q1) x_abc
        //   stage 2, lane = (c, q1, q2) < 48:   V [c][q12]   = sum_b l_b(q2) U,  Veta = sum_b dl_b(q2) U,  Vxi = sum_b l_b(q2) U'
        //   stage 3, lane = Gauss point:        J[0] = sum_c l_c(q3) Vxi,  J[1] = sum_c l_c(q3) Veta,  J[2] = sum_c dl_c(q3) V
        // 72 FMAs and ~64 LDS operations per lane instead of 243 and 135; every vector is (x, y, z, pad).
        double* U = slab;                  // [2][3][3][4][4] = 288 doubles
        double* V = slab + 288;            // [3][3][16][4]   = 432 doubles
        if (lane < 36) {
          double u[2][3] = {{0, 0, 0}, {0, 0, 0}};
#pragma unroll
          for (int a = 0; a < 3; a++) {
            const double* xn = xs + sfi_x[a];
            const double2 xa = *reinterpret_cast<const double2*>(xn);
            const double x2 = xn[2];
            const double la = SFl[a * 64 + lane], da = SFl[(3 + a) * 64 + lane];
            u[0][0] += la * xa.x; u[0][1] += la * xa.y; u[0][2] += la * x2;
            u[1][0] += da * xa.x; u[1][1] += da * xa.y; u[1][2] += da * x2;
          }
#pragma unroll
          for (int k = 0; k < 2; k++) {
            *reinterpret_cast<double2*>(U + k * 144 + sfi_b1) = make_double2(u[k][0], u[k][1]);
            U[k * 144 + sfi_b1 + 2] = u[k][2];
          }
        }
        wave_lds_sync();
        if (lane < 48) {
          double v[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll
          for (int b = 0; b < 3; b++) {
            const double* u0 = U + b * 48 + sfi_b2;
            const double* u1 = u0 + 144;
            const double2 ua = *reinterpret_cast<const double2*>(u0), va = *reinterpret_cast<const double2*>(u1);
            const double u2 = u0[2], v2 = u1[2];
            const double lb = SFl[(6 + b) * 64 + lane], db = SFl[(9 + b) * 64 + lane];
            v[0][0] += lb * ua.x; v[0][1] += lb * ua.y; v[0][2] += lb * u2;     // V
            v[1][0] += db * ua.x; v[1][1] += db * ua.y; v[1][2] += db * u2;     // Veta
            v[2][0] += lb * va.x; v[2][1] += lb * va.y; v[2][2] += lb * v2;     // Vxi
          }
#pragma unroll
          for (int k = 0; k < 3; k++) {
            *reinterpret_cast<double2*>(V + k * 192 + sfi_bv) = make_double2(v[k][0], v[k][1]);
            V[k * 192 + sfi_bv + 2] = v[k][2];
          }
        }
        wave_lds_sync();
#pragma unroll
        for (int c = 0; c < 3; c++) {
          const double lc = SFl[(12 + c) * 64 + lane], dc = SFl[(15 + c) * 64 + lane];
          const double* v0 = V + c * 64 + sfi_b3;
          const double2 a0 = *reinterpret_cast<const double2*>(v0), a1 = *reinterpret_cast<const double2*>(v0 + 192),
                        a2 = *reinterpret_cast<const double2*>(v0 + 384);
          const double z0 = v0[2], z1 = v0[192 + 2], z2 = v0[384 + 2];
          J[0][0] += lc * a2.x; J[0][1] += lc * a2.y; J[0][2] += lc * z2;
          J[1][0] += lc * a1.x; J[1][1] += lc * a1.y; J[1][2] += lc * z1;
          J[2][0] += dc * a0.x; J[2][1] += dc * a0.y; J[2][2] += dc * z0;
          if (SRC != 0) { xg[0] += lc * a0.x; xg[1] += lc * a0.y; xg[2] += lc * z0; }
        }
      } else {
#pragma unroll 9
        for (int n = 0; n < NC; n++) {
          const double2 xa = *reinterpret_cast<const double2*>(xs + n * 4);   // broadcast reads
          const double x0 = xa.x, x1 = xa.y, x2 = xs[n * 4 + 2];
          const double t0 = Tq[n], t1 = Tq[MF_TA + n], t2 = Tq[2 * MF_TA + n];
          J[0][0] += t0 * x0; J[0][1] += t0 * x1; J[0][2] += t0 * x2;
          J[1][0] += t1 * x0; J[1][1] += t1 * x1; J[1][2] += t1 * x2;
          J[2][0] += t2 * x0; J[2][1] += t2 * x1; J[2][2] += t2 * x2;
          if (SRC != 0) {
            const double ph = Phi[q * MF_PS + n];
            xg[0] += x0 * ph; xg[1] += x1 * ph; xg[2] += x2 * ph;
          }
        }
      }
      // cofactors Cf = det * J^-1 (the reference's Jacobian inverse, `elem_type_template` 3-D branch, without the division)
      double Cf[DIM][DIM];
      Cf[0][0] = -J[1][2] * J[2][1] + J[1][1] * J[2][2];
      Cf[0][1] = J[0][2] * J[2][1] - J[0][1] * J[2][2];
      Cf[0][2] = -J[0][2] * J[1][1] + J[0][1] * J[1][2];
      Cf[1][0] = J[1][2] * J[2][0] - J[1][0] * J[2][2];
      Cf[1][1] = -J[0][2] * J[2][0] + J[0][0] * J[2][2];
      Cf[1][2] = J[0][2] * J[1][0] - J[0][0] * J[1][2];
      Cf[2][0] = -J[1][1] * J[2][0] + J[1][0] * J[2][1];
      Cf[2][1] = J[0][1] * J[2][0] - J[0][0] * J[2][1];
      Cf[2][2] = -J[0][1] * J[1][0] + J[0][0] * J[1][1];
      const double det = J[0][0] * Cf[0][0] + J[0][1] * Cf[1][0] + J[0][2] * Cf[2][0];
      double fq;
      if (SRC == 0) fq = P.p0;
      else if (SRC == 1) fq = source_eval(P.source_kind, P.p0, P.p1, xg, DIM);
      else {
        double x4[4] = {xg[0], xg[1], xg[2], 0.0};
        fq = P.p0 * fh_expr_device_eval(P.prog, P.nprog, P.prog_consts, x4);
      }
      // grad phi_i[a] = sum_c JI[a][c] T[c][i], JI = Cf / det  ->  D[c][c'] = w det sum_a JI[a][c] JI[a][c'] = (w / det) (Cf^T Cf)[c][c']
      const double sc = wgauss / det;
      double* sq = slab + q * MF_SS;
      sq[0] = sc * (Cf[0][0] * Cf[0][0] + Cf[1][0] * Cf[1][0] + Cf[2][0] * Cf[2][0]);
      sq[1] = sc * (Cf[0][0] * Cf[0][1] + Cf[1][0] * Cf[1][1] + Cf[2][0] * Cf[2][1]);
      sq[2] = sc * (Cf[0][0] * Cf[0][2] + Cf[1][0] * Cf[1][2] + Cf[2][0] * Cf[2][2]);
      sq[3] = sc * (Cf[0][1] * Cf[0][1] + Cf[1][1] * Cf[1][1] + Cf[2][1] * Cf[2][1]);
      sq[4] = sc * (Cf[0][1] * Cf[0][2] + Cf[1][1] * Cf[1][2] + Cf[2][1] * Cf[2][2]);
      sq[5] = sc * (Cf[0][2] * Cf[0][2] + Cf[1][2] * Cf[1][2] + Cf[2][2] * Cf[2][2]);
      sq[6] = det * wgauss * fq;
    }
    wave_lds_sync();
    // ---- phase B: the rank-192 update on the matrix cores ----
    // The operand loads are issued by hand (ds_read_b64 with immediate offsets, explicit s_waitcnt): left to the compiler, pairs of
    // them become ds_read2_b64, which costs 8 LDS cycles instead of 2 + 2 (MI355X_MICROARCH.md, LDS table) and made the LDS, not
    // the matrix pipe, the bound of this phase.  Double-buffered in registers: the loads of step s+1 fly during the MFMAs of step s.
    double acc[MF_NT] = {0, 0, 0, 0, 0, 0, 0};
    if (!(P.debug & 1)) {
      double Ab[2][MF_NT], TD[2][12];
#define MF_LD(dst, addr, off) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define MF_LOAD_A(p, g, c)                                                                     \
  _Pragma("unroll") for (int t = 1; t < MF_NT; t++) MF_LD(Ab[p][t], aA[t], ((g) * MF_TS + (c) * MF_TA) * 8)
#define MF_LOAD_TD(p, g)                                                                       \
  _Pragma("unroll") for (int k = 0; k < 3; k++) {                                              \
    MF_LD(TD[p][k], aBlo, ((g) * MF_TS + k * MF_TA) * 8);                                      \
    MF_LD(TD[p][3 + k], aBhi, ((g) * MF_TS + k * MF_TA) * 8);                                  \
  }                                                                                            \
  _Pragma("unroll") for (int k = 0; k < 6; k++) MF_LD(TD[p][6 + k], aD, ((g) * MF_SS + k) * 8)
#define MF_WAIT_A(p)                                                                           \
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(Ab[p][1]), "+v"(Ab[p][2]), "+v"(Ab[p][3]), "+v"(Ab[p][4]), "+v"(Ab[p][5]), "+v"(Ab[p][6]))
#define MF_WAIT_TD(p)                                                                          \
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(TD[p][0]), "+v"(TD[p][1]), "+v"(TD[p][2]), "+v"(TD[p][3]), "+v"(TD[p][4]), "+v"(TD[p][5]), \
               "+v"(TD[p][6]), "+v"(TD[p][7]), "+v"(TD[p][8]), "+v"(TD[p][9]), "+v"(TD[p][10]), "+v"(TD[p][11]))
      MF_LOAD_TD(0, 0);
      MF_LOAD_A(0, 0, 0);
#pragma unroll
      for (int g = 0; g < 16; g++) {
        const int pg = g & 1;
        MF_WAIT_TD(pg);                       // lgkmcnt(0): the A loads of the first step of this group are in as well
        const double d00 = TD[pg][6], d01 = TD[pg][7], d02 = TD[pg][8], d11 = TD[pg][9], d12 = TD[pg][10], d22 = TD[pg][11];
        const double dd[3][3] = {{d00, d01, d02}, {d01, d11, d12}, {d02, d12, d22}};
#pragma unroll
        for (int c = 0; c < DIM; c++) {
          const int s = 3 * g + c, ps = s & 1;
          MF_WAIT_A(ps);
          if (c < 2) MF_LOAD_A(ps ^ 1, g, c + 1);
          else if (g < 15) {
            MF_LOAD_TD(pg ^ 1, g + 1);
            MF_LOAD_A(ps ^ 1, g + 1, 0);
          }
          const double blo = dd[c][0] * TD[pg][0] + dd[c][1] * TD[pg][1] + dd[c][2] * TD[pg][2];
          const double bhi = dd[c][0] * TD[pg][3] + dd[c][1] * TD[pg][4] + dd[c][2] * TD[pg][5];
          acc[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(TD[pg][c], blo, acc[0], 0, 0, 0);      // the four diagonal tiles 0..3
#pragma unroll
          for (int t = 1; t < MF_NT; t++) acc[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(Ab[ps][t], t < 4 ? blo : bhi, acc[t], 0, 0, 0);
        }
      }
#undef MF_LD
#undef MF_LOAD_A
#undef MF_LOAD_TD
#undef MF_WAIT_A
#undef MF_WAIT_TD
    }
    // ---- phase C: source integral per node (lanes i = lane&31, half of the Gauss points each) ----
    double fsrc = 0.0;
    {
      const int i = lane & 31, h = lane >> 5;
#pragma unroll 8
      for (int g = 0; g < 32; g++) {
        const int q = h * 32 + g;
        fsrc += Phi[q * MF_PS + i] * slab[q * MF_SS + 6];
      }
      fsrc += __shfl_xor(fsrc, 32, 64);
    }
    wave_lds_sync();          // every lane is done with the phase-A slab: reuse it as Ks[27][29]
    double* Ks = slab;
    // staging offsets of this lane's tile entries from the table built at kernel start (kept in LDS: as loop invariants in
    // registers they were spilled to scratch and every reload waited on vmcnt(0))
#pragma unroll
    for (int t = 0; t < MF_NT; t++) {
      const int od = KsOff[(2 * t) * 64 + lane], om = KsOff[(2 * t + 1) * 64 + lane];
      if (od >= 0) Ks[od] = acc[t];
      if (om >= 0) Ks[om] = acc[t];
    }
    wave_lds_sync();
    double ku = 0.0;
    if (P.sol) {              // residual: (K_e u)_i, lanes i = lane&31, half of the columns each
      const int i = min(lane & 31, NC - 1), h = lane >> 5;
#pragma unroll
      for (int g = 0; g < 14; g++) {
        const int j = h * 14 + g;
        const double v = (j < NC) ? Ks[i * MF_KS + min(j, NC - 1)] : 0.0;
        ku += v * xs[min(j, NC - 1) * 4 + 3];
      }
      ku += __shfl_xor(ku, 32, 64);
    }
    // the next element's nodes: this consumes the prefetched registers (a vmcnt wait) BEFORE the hand-issued stores below, which
    // the compiler's wait-count bookkeeping does not see; LDS operations of one wave execute in order, so the reads of xs above
    // are done before these writes
    if (lane < NC) {
      xs[lane * 4 + 0] = nx0;
      xs[lane * 4 + 1] = nx1;
      xs[lane * 4 + 2] = nx2;
      xs[lane * 4 + 3] = nu;
    }
    if (!(P.debug & 2)) {
      if (P.kstride >= 28) {   // padded rows (32: one whole 256-byte row per half-wave; 28: 224 bytes = seven whole 32-byte sectors)
        // all LDS reads first; slots through v_readlane into scalar registers, and the stores written by hand so that each
        // half-wave stores its row off a SCALAR base address (address arithmetic on the scalar unit; left to the compiler every
        // store cost 5-9 vector instructions).  The 5 pad entries of a row receive whatever follows in the staging: the row pass
        // never reads them.
        const int j = lane & 31, hrow = lane >> 5;
        const unsigned joff = (unsigned)j * 8u;
        double kv[14];
#pragma unroll
        for (int p = 0; p < 14; p++) kv[p] = Ks[min(2 * p + hrow, NC - 1) * MF_KS + j];
#pragma unroll
        for (int p = 0; p < 14; p++) {
          const int s0 = __builtin_amdgcn_readlane(sl_cur, 2 * p), s1 = __builtin_amdgcn_readlane(sl_cur, min(2 * p + 1, NC - 1));
          const double* b0 = P.Kout + (size_t)s0 * P.kstride;
          const double* b1 = P.Kout + (size_t)s1 * P.kstride;
          if (j >= P.kstride) continue;
          if (hrow == 0) {
            if (s0 >= 0) asm volatile("global_store_dwordx2 %0, %1, %2" ::"v"(joff), "v"(kv[p]), "s"(b0) : "memory");
          } else if (2 * p + 1 < NC) {
            if (s1 >= 0) asm volatile("global_store_dwordx2 %0, %1, %2" ::"v"(joff), "v"(kv[p]), "s"(b1) : "memory");
          }
        }
      } else {
#pragma unroll
        for (int t0 = 0; t0 < NC * NC; t0 += 64) {
          const int t = t0 + lane;
          const int row = (t < NC * NC) ? t / NC : 0;
          const int j = t - row * NC;
          const int s = __shfl(sl_cur, row, 64);
          if (t < NC * NC && s >= 0) P.Kout[(size_t)s * NC + j] = Ks[row * MF_KS + j];
        }
      }
      if (lane < NC && sl_cur >= 0) P.Fout[sl_cur] = -(ku + fsrc);
    }
    wave_lds_sync();          // Ks is the next element's phase-A slab, xs holds the next element's nodes
    sl_cur = sl_n;
    dof_n = dof_nn;
    e_n = e_nn;
    e_nn = e_nnn;
  }
}

template <int SRC, int NW>
static int launch_mfma_one(fh_assembler_t as, const AsmParams& P) {
  constexpr size_t lds = mf_lds_bytes(NW);
  static_assert(lds <= 160 * 1024, "k_elem_q2hex_mfma: LDS budget");
  static bool attr_set[64] = {};      // per device: the attribute belongs to the function on the current device
  const int dev = as->ctx->device & 63;
  if (!attr_set[dev]) {
    FH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_elem_q2hex_mfma<SRC, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set[dev] = true;
  }
  const int per_cu = std::max(1, (int)((size_t)160 * 1024 / lds));
  const int grid = std::max(1, std::min(fh_div_up(P.nelems, NW), as->ctx->num_cu * per_cu));
  hipLaunchKernelGGL((k_elem_q2hex_mfma<SRC, NW>), dim3(grid), dim3(NW * 64), lds, as->ctx->stream, P, as->d_mfT, as->d_mfPhi,
                     as->ctx->assemble_sumfac ? as->d_mfSFc : nullptr, as->d_mfSFi);
  FH_CHECK_HIP(hipGetLastError());
  return 0;
}

// waves per workgroup: the option for the constant source; the closed-form sources need more registers (no spills up to 8 waves)
// and the expression evaluator with its operand stack more still (4 waves: 512 registers per lane)
static int launch_mfma(fh_assembler_t as, const AsmParams& P, int nw) {
  if (P.source_kind == 4) return launch_mfma_one<2, 4>(as, P);
  if (P.source_kind != 0) return nw <= 4 ? launch_mfma_one<1, 4>(as, P) : launch_mfma_one<1, 8>(as, P);
  return nw <= 4 ? launch_mfma_one<0, 4>(as, P) : nw <= 8 ? launch_mfma_one<0, 8>(as, P) : launch_mfma_one<0, 12>(as, P);
}

// ------------------------------------------------------------------------------------------------------------------
// HEX27 / Q2 element matrices by SUM FACTORISATION (default for the 64-point rule when the tables are tensor products, which
// fh_assembler_create verifies against the reference's tables): ONE element per wave, vector FP64 only.
//   phi_i = l_a(xi) l_b(eta) l_c(zeta), Gauss point q = (q1, q2, q3), D_q = w_q det J (J^-1)(J^-1)^T in reference directions, so
//   K[(a,b,c),(a',b',c')] = sum_q1 sum_{X1,X2 in {l, l'}} X1_a(q1) X2_a'(q1) G_{X1X2}[bb'][cc'][q1]
//   G_{l'l'} = sum_q2 (l_b l_b')  e0                      e0 = sum_q3 D00 l_c l_c'      e1 = sum_q3 D01 l_c l_c'
//   G_{l'l}  = sum_q2 (l_b l'_b') e1 + (l_b l_b') e2      e2 = sum_q3 D02 l_c l'_c'     e3 = sum_q3 D11 l_c l_c'
//   G_{ll'}  = sum_q2 (l'_b l_b') e1 + (l_b l_b') e2^T    e4 = sum_q3 D12 l_c l'_c'     e5 = sum_q3 D22 l'_c l'_c'
//   G_{ll}   = sum_q2 (l'_b l'_b') e3 + (l'_b l_b') e4 + (l_b l'_b') e4^T + (l_b l_b') e5          (^T: c and c' exchanged)
// i.e. the i/j/Gauss loop of `00_poisson_eqn_..._separate.hpp:170-200` (27 x 27 x 64 x 3 x 3 products) contracted one direction
// at a time: about 100 + 144 + 120 vector FMAs per lane instead of 336 matrix instructions of 16 cycles + 252 FMAs (same result
// up to the order of the floating-point sums; parity tests at 1e-12).
//   phase A   map Jacobian by sum factorisation (as in k_elem_q2hex_mfma; here the Gauss points sit on the lanes in tensor order
//             q1*16 + q2*4 + q3), cofactors, D_q, source value -> Dq[7][64]
//   stage 1   lane = (q1, q2, c): contracts q3 -> e[6][c c'][q1 q2] (and the source: sE[c][q1 q2])
//   stage 2   lane = unordered pair {(b,c), (b',c')} (45 of the 64 lanes): contracts q2 -> 16 values G in registers
//   stage 3   same lane: contracts q1 -> the 3 x 3 block over (a, a'), mirrored into the 27 x 29 staging (K_e symmetric bit for
//             bit: blocks off the diagonal pair are written twice, the six upper entries of a diagonal pair are mirrored)
//   source    f_i = sum_q phi_i s_q by the same three contractions (lanes (q1,q2,c) -> (q1,b,c) -> node)
//   output    as k_elem_q2hex_mfma: K_e u for the residual, one 256-byte row per half-wave store
// No table of the 3-D basis is read: the 1-D values are uniform operands (kernel argument) or per-lane constants.
// ------------------------------------------------------------------------------------------------------------------
constexpr int SF_ES = 18;                    // doubles per (c,c') block of e (16 + 2: the nine blocks start in distinct banks)
constexpr int SF_NE = 56 * SF_ES;            // e: four symmetric arrays of 8 slots (6 used), two of 12 (9 used) = 1008 doubles
constexpr int SF_R = SF_NE + 64 + 36;        // e, sE[4][16], sF[9][4]; phase A's U / V and the 27 x 29 staging alias e
constexpr int SF_US = 40, SF_VS = 48;        // phase A: U[2][3][SF_US] then V[3][3][SF_VS], one array per coordinate (conflict-free)
constexpr int SF_XT = 4 * 28;                // per wave: x, y, z and u of the element's nodes, 28 doubles each, in TENSOR order a*9 + b*3 + c
constexpr int SF_WAVE = SF_XT + SF_R;   // 1220 doubles per wave (even: 16-byte alignment is kept)
constexpr int SF_NLC = 51, SF_NLI = 18;      // rows of the per-lane tables (layout: fh_assembler_create)
constexpr int SF_TAB = SF_NLC * 64 + SF_NLI * 32;   // doubles: both tables, shared by the waves of the workgroup
constexpr size_t sf_lds_bytes(int nw) { return (size_t)(SF_TAB + nw * SF_WAVE) * sizeof(double); }
static_assert(SF_R >= MF_SLAB && SF_R >= 6 * SF_US + 9 * SF_VS, "k_elem_q2hex_sf: region R holds the staging and phase A's U, V");

__device__ __forceinline__ void sf_ld4(const double* p, double v[4]) {
  const double2 a = *reinterpret_cast<const double2*>(__builtin_assume_aligned(p, 16));
  const double2 b = *reinterpret_cast<const double2*>(__builtin_assume_aligned(p + 2, 16));
  v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
}

// Phases A and stages 1-3 of the sum-factorised element matrix (shared by k_elem_q2hex_sf and the cluster kernel k_cluster_q2hex_sf): from the
// element's nodes in xt (tensor order) to this lane's 3 x 3 block Kb over (a, a') and its source entry fsrc.  R is the wave's scratch region.
// phase stamps of the instrumented build of the cluster kernel (asm_debug bit 7): shader clock at the phase boundaries, summed per wave in scalar registers
struct SfStamps {
  unsigned long long prev, acc[16];
};
template <bool INS>
__device__ __forceinline__ void sf_stamp(SfStamps& st, const int k) {
  if (INS) {
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    st.acc[k] += t - st.prev;
    st.prev = t;
  }
}

// lane l of tabv holds entry l of the 1-D table {L[3][4], D[3][4]} (sf_tab_lanes); a uniform entry is fetched with two v_readlane.  Keeping the 24 doubles
// in scalar registers instead (the kernel argument) left the unrolled stages 2-3 with scalar spills.
__device__ __forceinline__ double sf_tab_lanes(const SfTab& tab, const int lane) {
  double v = 0.0;
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      v = (lane == a * 4 + q) ? tab.L[a][q] : v;
      v = (lane == 12 + a * 4 + q) ? tab.D[a][q] : v;
    }
  return v;
}
__device__ __forceinline__ double sf_tab_get(const double tabv, const int entry) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(tabv), entry), hi = __builtin_amdgcn_readlane(__double2hiint(tabv), entry);
  return __hiloint2double(hi, lo);
}

// Phase A of the sum-factorised element matrix: from the element's nodes in xt (tensor order) to D_q and the source weight at this lane's Gauss point.
// UV = 6 * SF_US + 9 * SF_VS doubles of wave-private scratch (the callers pass region R, or a region of its own when R still holds a staging).
struct SfIdxA {             // per-lane double indices of phase A (rows 0 .. 4 of the integer table): xt gather, U write, U read, V write, V read
  int xn, uo, ui, vo, vi;
};
template <int SRC, bool REGC, bool INS = false>
__device__ __forceinline__ void sf_phase_a(const AsmParams& P, const double* LCl, const SfIdxA ia, const double (&rcA)[19], const double* xt, double* UV, double (&Dr)[7],
                                           SfStamps* stp = nullptr) {
  constexpr int DIM = 3;
  SfStamps dummy_st;
  SfStamps& st = INS ? *stp : dummy_st;
#define SF_C(r) LCl[(r) * 64]
#define SF_CA(r) (REGC ? rcA[r] : SF_C(r))
  // ---- phase A: J_q by three contractions through LDS (U, V: one array per coordinate, every access is conflict-free), then D_q; lane = Gauss point
  //      in tensor order.  Lanes beyond a stage's role count repeat its last role (same values to the same addresses): no divergent branch in the
  //      element loop.  The reads are issued by hand as ds_read_b64: left to the compiler every pair became a ds_read2_b64, which costs 8 LDS cycles
  //      instead of 2 + 2 (MI355X_MICROARCH.md, LDS table) -- and the LDS array is the busiest unit of the element phase (65 % of its cycles) ----
#define SF_LDA(dst, addr, off) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#define SF_WAIT3(n, a, b, c) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(a), "+v"(b), "+v"(c))
  {
    double J[DIM][DIM] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, xg[DIM] = {0, 0, 0};
    double* U = UV;                    // [2][3][SF_US]: (sum_a l_a x, sum_a l'_a x) at [(b*3+c)*4 + q1]
    double* V = UV + 6 * SF_US;         // [3][3][SF_VS]: V, Veta, Vxi at [c*16 + q1 + 4 q2]
    {
      const unsigned axn = (unsigned)(size_t)(__attribute__((address_space(3))) const double*)(xt + ia.xn);             // b*3 + c
      double x[3][3];
#pragma unroll
      for (int a = 0; a < 3; a++)
#pragma unroll
        for (int d = 0; d < 3; d++) SF_LDA(x[a][d], axn, (d * 28 + a * 9) * 8);
      double u[2][3] = {{0, 0, 0}, {0, 0, 0}};
#pragma unroll
      for (int a = 0; a < 3; a++) {
        const double la = SF_CA(a), da = SF_CA(3 + a);
        if (a == 0) SF_WAIT3(6, x[0][0], x[0][1], x[0][2]);
        else if (a == 1) SF_WAIT3(3, x[1][0], x[1][1], x[1][2]);
        else SF_WAIT3(0, x[2][0], x[2][1], x[2][2]);
#pragma unroll
        for (int d = 0; d < 3; d++) {
          u[0][d] += la * x[a][d];
          u[1][d] += da * x[a][d];
        }
      }
      double* uo = U + ia.uo;
#pragma unroll
      for (int k = 0; k < 2; k++)
#pragma unroll
        for (int d = 0; d < 3; d++) uo[(k * 3 + d) * SF_US] = u[k][d];
    }
    wave_lds_sync();
    sf_stamp<INS>(st, 1);
    {
      const unsigned aui = (unsigned)(size_t)(__attribute__((address_space(3))) const double*)(U + ia.ui);              // c*4 + q1
      double u0[3][3], u1[3][3];
#pragma unroll
      for (int b = 0; b < 3; b++)
#pragma unroll
        for (int d = 0; d < 3; d++) {
          SF_LDA(u0[b][d], aui, (d * SF_US + b * 12) * 8);
          SF_LDA(u1[b][d], aui, ((3 + d) * SF_US + b * 12) * 8);
        }
      double v[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll
      for (int b = 0; b < 3; b++) {
        const double lb = SF_CA(6 + b), db = SF_CA(9 + b);
        if (b == 0) { SF_WAIT3(12, u0[0][0], u0[0][1], u0[0][2]); SF_WAIT3(12, u1[0][0], u1[0][1], u1[0][2]); }
        else if (b == 1) { SF_WAIT3(6, u0[1][0], u0[1][1], u0[1][2]); SF_WAIT3(6, u1[1][0], u1[1][1], u1[1][2]); }
        else { SF_WAIT3(0, u0[2][0], u0[2][1], u0[2][2]); SF_WAIT3(0, u1[2][0], u1[2][1], u1[2][2]); }
#pragma unroll
        for (int d = 0; d < 3; d++) {
          v[0][d] += lb * u0[b][d];     // V
          v[1][d] += db * u0[b][d];     // Veta
          v[2][d] += lb * u1[b][d];     // Vxi
        }
      }
      double* vo = V + ia.vo;
#pragma unroll
      for (int k = 0; k < 3; k++)
#pragma unroll
        for (int d = 0; d < 3; d++) vo[(k * 3 + d) * SF_VS] = v[k][d];
    }
    wave_lds_sync();
    sf_stamp<INS>(st, 2);
    {
      const unsigned avi = (unsigned)(size_t)(__attribute__((address_space(3))) const double*)(V + ia.vi);              // q1 + 4 q2
      double v0[3][3], v1[3][3], v2[3][3];
#define SF_LDJ(c)                                                      \
  _Pragma("unroll") for (int d = 0; d < 3; d++) {                      \
    SF_LDA(v0[c][d], avi, (d * SF_VS + (c) * 16) * 8);                 \
    SF_LDA(v1[c][d], avi, ((3 + d) * SF_VS + (c) * 16) * 8);           \
    SF_LDA(v2[c][d], avi, ((6 + d) * SF_VS + (c) * 16) * 8);           \
  }
      SF_LDJ(0);
      SF_LDJ(1);
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const double lc = SF_CA(12 + c), dc = SF_CA(15 + c);
        // nine reads per c; the reads of c = 2 go out once c = 0 has been consumed (eighteen operands live instead of twenty-seven)
        if (c == 0) { SF_WAIT3(9, v0[0][0], v0[0][1], v0[0][2]); SF_WAIT3(9, v1[0][0], v1[0][1], v1[0][2]); SF_WAIT3(9, v2[0][0], v2[0][1], v2[0][2]); }
        else if (c == 1) { SF_WAIT3(9, v0[1][0], v0[1][1], v0[1][2]); SF_WAIT3(9, v1[1][0], v1[1][1], v1[1][2]); SF_WAIT3(9, v2[1][0], v2[1][1], v2[1][2]); }
        else { SF_WAIT3(0, v0[2][0], v0[2][1], v0[2][2]); SF_WAIT3(0, v1[2][0], v1[2][1], v1[2][2]); SF_WAIT3(0, v2[2][0], v2[2][1], v2[2][2]); }
#pragma unroll
        for (int d = 0; d < 3; d++) {
          J[0][d] += lc * v2[c][d];
          J[1][d] += lc * v1[c][d];
          J[2][d] += dc * v0[c][d];
          if (SRC != 0) xg[d] += lc * v0[c][d];
        }
        if (c == 0) { SF_LDJ(2); }
      }
#undef SF_LDJ
    }
    // cofactors Cf = det * J^-1 (the reference's Jacobian inverse, `elem_type_template` 3-D branch, without the division)
    double Cf[DIM][DIM];
    Cf[0][0] = -J[1][2] * J[2][1] + J[1][1] * J[2][2];
    Cf[0][1] = J[0][2] * J[2][1] - J[0][1] * J[2][2];
    Cf[0][2] = -J[0][2] * J[1][1] + J[0][1] * J[1][2];
    Cf[1][0] = J[1][2] * J[2][0] - J[1][0] * J[2][2];
    Cf[1][1] = -J[0][2] * J[2][0] + J[0][0] * J[2][2];
    Cf[1][2] = J[0][2] * J[1][0] - J[0][0] * J[1][2];
    Cf[2][0] = -J[1][1] * J[2][0] + J[1][0] * J[2][1];
    Cf[2][1] = J[0][1] * J[2][0] - J[0][0] * J[2][1];
    Cf[2][2] = -J[0][1] * J[1][0] + J[0][0] * J[1][1];
    const double det = J[0][0] * Cf[0][0] + J[0][1] * Cf[1][0] + J[0][2] * Cf[2][0];
    double fq;
    if (SRC == 0) fq = P.p0;
    else if (SRC == 1) fq = source_eval(P.source_kind, P.p0, P.p1, xg, DIM);
    else {
      double x4[4] = {xg[0], xg[1], xg[2], 0.0};
      fq = P.p0 * fh_expr_device_eval(P.prog, P.nprog, P.prog_consts, x4);
    }
    const double wgauss = SF_CA(18);
    const double sc = wgauss / det;
    Dr[0] = sc * (Cf[0][0] * Cf[0][0] + Cf[1][0] * Cf[1][0] + Cf[2][0] * Cf[2][0]);
    Dr[1] = sc * (Cf[0][0] * Cf[0][1] + Cf[1][0] * Cf[1][1] + Cf[2][0] * Cf[2][1]);
    Dr[2] = sc * (Cf[0][0] * Cf[0][2] + Cf[1][0] * Cf[1][2] + Cf[2][0] * Cf[2][2]);
    Dr[3] = sc * (Cf[0][1] * Cf[0][1] + Cf[1][1] * Cf[1][1] + Cf[2][1] * Cf[2][1]);
    Dr[4] = sc * (Cf[0][1] * Cf[0][2] + Cf[1][1] * Cf[1][2] + Cf[2][1] * Cf[2][2]);
    Dr[5] = sc * (Cf[0][2] * Cf[0][2] + Cf[1][2] * Cf[1][2] + Cf[2][2] * Cf[2][2]);
    Dr[6] = det * wgauss * fq;
  }
  if (INS) {                  // the stamp waits for D_q (the value is consumed by an empty asm statement)
    asm volatile("" ::"v"(Dr[0]), "v"(Dr[5]), "v"(Dr[6]));
    sf_stamp<INS>(st, 3);
  }
#undef SF_LDA
#undef SF_WAIT3
#undef SF_C
#undef SF_CA
}

// Stages 1-3 (and the source integral): from D_q at this lane's Gauss point (Dr, as sf_phase_a leaves it) to this lane's 3 x 3 block Kb over (a, a') and its
// source entry fsrc.  R is the wave's scratch region (e arrays); every lane must be done with whatever aliased it (the callers synchronise the wave first).
template <int SRC, bool REGC, bool INS = false>
__device__ __forceinline__ void sf_stages(const AsmParams& P, const SfTab& tab, const double* LCl, const int* LIl, const double (&rcZ)[8], const double (&rcY)[16], const int esym,
                                          const int ens, const int ensT, const int eout, double* R, const int lane, const double (&Dr)[7], double (&Kb)[3][3], double& fsrc,
                                          SfStamps* stp = nullptr) {
  SfStamps dummy_st;
  SfStamps& st = INS ? *stp : dummy_st;
#define SF_I(r) LIl[(r) * 64]
#define SF_C(r) LCl[(r) * 64]
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int a2 = 0; a2 < 3; a2++) Kb[a][a2] = 0.0;
  fsrc = 0.0;
  if (!(P.debug & 1)) {
    // ---- stage 1: contracts q3 on the FP64 matrix cores.  The Gauss points sit on the lanes as 16 q3 + 4 q1 + q2, which is
    //      the B-operand layout of v_mfma_f64_4x4x4_4b (lane = 16 k + 4 block + column) with k = q3, block = q1, column = q2: D_q
    //      is used where phase A left it.  A (lane = 16 k + 4 block + row) = the products l_c l_c', l_c l'_c', l'_c l'_c' at abscissa
    //      k for four (c, c') slots (per-lane constants, the same for every block); the result lane 16 row + 4 q1 + q2 holds
    //      e[slot][q1][q2].  Symmetric arrays (e0, e1, e3, e5) keep the six slots c <= c' (two instructions of four), e2 and e4 the
    //      nine slots c*3 + c' (three); one more instruction contracts the source.  15 matrix instructions replace 100 vector ones,
    //      the LDS round trip of D_q and a cross-lane sum over q3. ----
    {
      double* eo = R + eout;
      constexpr int EB[6] = {0, 8 * SF_ES, 16 * SF_ES, 28 * SF_ES, 36 * SF_ES, 48 * SF_ES};
#pragma unroll
      for (int kk = 0; kk < 6; kk++) {
        const int ng = (kk == 2 || kk == 4) ? 3 : 2;
        const int a0 = (kk == 2 || kk == 4) ? 2 : kk == 5 ? 5 : 0;      // first constant vector: LL 0,1  LD 2,3,4  DD 5,6
#pragma unroll
        for (int g = 0; g < ng; g++) {
          const double za = REGC ? rcZ[a0 + g] : SF_C(19 + a0 + g);
          eo[EB[kk] + 4 * g * SF_ES] = __builtin_amdgcn_mfma_f64_4x4x4f64(za, Dr[kk], 0.0, 0, 0, 0);
        }
      }
      const double zs = REGC ? rcZ[7] : SF_C(26);
      R[SF_NE + lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(zs, Dr[6], 0.0, 0, 0, 0);      // sE[c][q1*4 + q2], c = lane >> 4
    }
    wave_lds_sync();
    sf_stamp<INS>(st, 4);
    // ---- source, second contraction: lane = (q1, b, c), contracts q2 ----
    {
      double s4[4];
      sf_ld4(R + SF_NE + SF_I(10), s4);
      R[SF_NE + 64 + SF_I(11)] = s4[0] * SF_C(43) + s4[1] * SF_C(44) + s4[2] * SF_C(45) + s4[3] * SF_C(46);
    }
    // ---- stages 2 and 3: lane = pair {(b,c), (b',c')}; per q1: contract q2 into the four G values, then add their part of the
    //      3 x 3 block over (a, a').  The loop is NOT unrolled: one iteration's operands are all that is live. ----
    {
      const double* es = R + esym;      // symmetric arrays: slot of (min(c,c'), max(c,c'))
      const double* en = R + ens;       // e2, e4: slot c*3 + c'
      const double* eT = R + ensT;      //         slot c'*3 + c
      double yLL[4], yLD[4], yDL[4], yDD[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        yLL[q] = REGC ? rcY[q] : SF_C(27 + q); yLD[q] = REGC ? rcY[4 + q] : SF_C(31 + q);
        yDL[q] = REGC ? rcY[8 + q] : SF_C(35 + q); yDD[q] = REGC ? rcY[12 + q] : SF_C(39 + q);
      }
      {
        // Software pipeline by hand, half a q1 at a time (the phase stamps showed a wave paying latency + 16 x 16 cycles of read return + 66 FMAs one
        // after the other per q1): group A = the eight reads behind g0, g1, g2, group B = the eight behind g3; B(q1) is issued before A(q1)'s arithmetic,
        // A(q1 + 1) before B(q1)'s, so eight reads are always in flight under 20 / 46 multiply-adds.  LDS reads of a wave return in order:
        // `s_waitcnt lgkmcnt(8)` = everything but the youngest eight has arrived.  No more registers than the plain loop (sixteen operands live).
        typedef double sf_d2 __attribute__((ext_vector_type(2)));
        const unsigned a_es = (unsigned)(size_t)(__attribute__((address_space(3))) const double*)es;
        const unsigned a_en = (unsigned)(size_t)(__attribute__((address_space(3))) const double*)en;
        const unsigned a_eT = (unsigned)(size_t)(__attribute__((address_space(3))) const double*)eT;
#define SF_LD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#define SF_LOAD_A(q1)                                                                                                                     \
  SF_LD(va[0], a_es, (0 * SF_ES + (q1) * 4) * 8); SF_LD(va[1], a_es, (0 * SF_ES + (q1) * 4) * 8 + 16);                                     \
  SF_LD(va[2], a_es, (8 * SF_ES + (q1) * 4) * 8); SF_LD(va[3], a_es, (8 * SF_ES + (q1) * 4) * 8 + 16);                                     \
  SF_LD(va[4], a_en, (16 * SF_ES + (q1) * 4) * 8); SF_LD(va[5], a_en, (16 * SF_ES + (q1) * 4) * 8 + 16);                                   \
  SF_LD(va[6], a_eT, (16 * SF_ES + (q1) * 4) * 8); SF_LD(va[7], a_eT, (16 * SF_ES + (q1) * 4) * 8 + 16)
#define SF_LOAD_B(q1)                                                                                                                     \
  SF_LD(vb[0], a_es, (28 * SF_ES + (q1) * 4) * 8); SF_LD(vb[1], a_es, (28 * SF_ES + (q1) * 4) * 8 + 16);                                   \
  SF_LD(vb[2], a_en, (36 * SF_ES + (q1) * 4) * 8); SF_LD(vb[3], a_en, (36 * SF_ES + (q1) * 4) * 8 + 16);                                   \
  SF_LD(vb[4], a_eT, (36 * SF_ES + (q1) * 4) * 8); SF_LD(vb[5], a_eT, (36 * SF_ES + (q1) * 4) * 8 + 16);                                   \
  SF_LD(vb[6], a_es, (48 * SF_ES + (q1) * 4) * 8); SF_LD(vb[7], a_es, (48 * SF_ES + (q1) * 4) * 8 + 16)
#define SF_WAIT(v, n)                                                                                                                     \
  asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]))
        sf_d2 va[8], vb[8];
        const double tabv = sf_tab_lanes(tab, lane);
        SF_LOAD_A(0);
#pragma unroll
        for (int q1 = 0; q1 < 4; q1++) {
          if (q1 == 0) { SF_LOAD_B(0); } else if (q1 == 1) { SF_LOAD_B(1); } else if (q1 == 2) { SF_LOAD_B(2); } else { SF_LOAD_B(3); }
          SF_WAIT(va, 8);
          double g0 = 0.0, g1 = 0.0, g2 = 0.0, g3 = 0.0;     // G for (l'l'), (l'l), (ll'), (ll)
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const double v0 = va[q >> 1][q & 1], v1 = va[2 + (q >> 1)][q & 1], v2 = va[4 + (q >> 1)][q & 1], v2T = va[6 + (q >> 1)][q & 1];
            g0 += yLL[q] * v0;
            g1 += yLD[q] * v1; g1 += yLL[q] * v2;
            g2 += yDL[q] * v1; g2 += yLL[q] * v2T;
          }
          if (q1 == 0) { SF_LOAD_A(1); } else if (q1 == 1) { SF_LOAD_A(2); } else if (q1 == 2) { SF_LOAD_A(3); }
          if (q1 < 3) { SF_WAIT(vb, 8); } else { SF_WAIT(vb, 0); }
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const double v3 = vb[q >> 1][q & 1], v4 = vb[2 + (q >> 1)][q & 1], v4T = vb[4 + (q >> 1)][q & 1], v5 = vb[6 + (q >> 1)][q & 1];
            g3 += yDD[q] * v3; g3 += yDL[q] * v4; g3 += yLD[q] * v4T; g3 += yLL[q] * v5;
          }
          const double la[3] = {sf_tab_get(tabv, q1), sf_tab_get(tabv, 4 + q1), sf_tab_get(tabv, 8 + q1)};
          const double da[3] = {sf_tab_get(tabv, 12 + q1), sf_tab_get(tabv, 16 + q1), sf_tab_get(tabv, 20 + q1)};
          double u[3], v[3];
#pragma unroll
          for (int a = 0; a < 3; a++) {
            u[a] = la[a] * g3 + da[a] * g1;
            v[a] = la[a] * g2 + da[a] * g0;
          }
#pragma unroll
          for (int a = 0; a < 3; a++)
#pragma unroll
            for (int a2 = 0; a2 < 3; a2++) { Kb[a][a2] += la[a2] * u[a]; Kb[a][a2] += da[a2] * v[a]; }
        }
#undef SF_LD
#undef SF_LOAD_A
#undef SF_LOAD_B
#undef SF_WAIT
      }
    }
    if (INS) {
      asm volatile("" ::"v"(Kb[0][0]), "v"(Kb[2][2]));
      sf_stamp<INS>(st, 5);
    }
    wave_lds_sync();
    {
      double s4[4];
      sf_ld4(R + SF_NE + 64 + SF_I(16), s4);
      fsrc = s4[0] * SF_C(47) + s4[1] * SF_C(48) + s4[2] * SF_C(49) + s4[3] * SF_C(50);
    }
  }
#undef SF_I
#undef SF_C
}

// Phase A and stages 1-3 in one piece (k_elem_q2hex_sf; the cluster kernel calls the two halves itself): R serves phase A's U / V first, then the e arrays.
template <int SRC, bool REGC, bool INS = false>
__device__ __forceinline__ void sf_element_blocks(const AsmParams& P, const SfTab& tab, const double* LCl, const int* LIl, const double (&rcA)[19], const double (&rcZ)[8],
                                                  const double (&rcY)[16], const int esym, const int ens, const int ensT, const int eout, const double* xt, double* R,
                                                  const int lane, double (&Kb)[3][3], double& fsrc, SfStamps* stp = nullptr) {
  double Dr[7];            // D_q (six entries) and the source weight at this lane's Gauss point
  const SfIdxA ia = {LIl[0], LIl[64], LIl[2 * 64], LIl[3 * 64], LIl[4 * 64]};
  sf_phase_a<SRC, REGC, INS>(P, LCl, ia, rcA, xt, R, Dr, stp);
  wave_lds_sync();
  sf_stages<SRC, REGC, INS>(P, tab, LCl, LIl, rcZ, rcY, esym, ens, ensT, eout, R, lane, Dr, Kb, fsrc, stp);
}

template <int SRC, int NW, bool PAD>
__global__ __launch_bounds__(NW * 64) void k_elem_q2hex_sf(AsmParams P, SfTab tab, const double* __restrict__ lanec, const int* __restrict__ lanei) {
  constexpr int NC = 27, DIM = 3, KS = MF_KS;
  extern __shared__ __attribute__((aligned(16))) double sf_smem[];
  double* SFl = sf_smem;                                  // [SF_NLC][64] doubles, then [SF_NLI][64] ints
  int* SFi = reinterpret_cast<int*>(SFl + SF_NLC * 64);
  for (int k = threadIdx.x; k < SF_NLC * 64; k += NW * 64) SFl[k] = lanec[k];
  for (int k = threadIdx.x; k < SF_NLI * 64; k += NW * 64) SFi[k] = lanei[k];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double* xt = SFl + SF_TAB + wave * SF_WAVE;          // xt[0..27] x, [28..55] y, [56..83] z, [84..111] u   (tensor order)
  double* R = xt + SF_XT;
  // per-lane constants: the hot ones (stages 1-3, staging, output) in registers, the others are read from the workgroup's copy in
  // LDS where they are used (conflict-free, 2 LDS cycles each)
  const double* LCl = SFl + lane;                    // row r of the double table: LCl[r * 64]
  const int* LIl = SFi + lane;                       // row r of the int table: LIl[r * 64]
#define SF_I(r) LIl[(r) * 64]
#define SF_C(r) LCl[(r) * 64]
  constexpr bool REGC = NW <= 8;          // 2 waves per SIMD: 256 registers per lane, room for the per-lane constants of the hot stages
  double rcA[19], rcZ[8], rcY[16];
  if (REGC) {
#pragma unroll
    for (int r = 0; r < 19; r++) rcA[r] = lanec[r * 64 + lane];
#pragma unroll
    for (int r = 0; r < 8; r++) rcZ[r] = lanec[(19 + r) * 64 + lane];
#pragma unroll
    for (int r = 0; r < 16; r++) rcY[r] = lanec[(27 + r) * 64 + lane];
  }
#define SF_CA(r) (REGC ? rcA[r] : SF_C(r))
  const int esym = lanei[5 * 64 + lane], ens = lanei[6 * 64 + lane], ensT = lanei[15 * 64 + lane], based = lanei[7 * 64 + lane], basem = lanei[8 * 64 + lane];
  const int eout = (lane >> 4) * SF_ES + (lane & 15);
  const bool diag = lanei[9 * 64 + lane] != 0;
  // Lanes 0..26 stand for the element's nodes in TENSOR order t = a*9 + b*3 + c (node nodeofl): coordinates, solution values, output
  // slots and the residual entries live on lane t.
  const int tofl = lanei[13 * 64 + lane];              // tensor index of node min(lane & 31, 26)  (PAD = false: rows in node order)
  const int nodeofl = lanei[14 * 64 + lane];           // node of tensor index min(lane, 26)
  const size_t rowbytes = (size_t)P.kstride * sizeof(double);
  const size_t sink = (size_t)P.nsink * rowbytes;      // rows the matrix does not hold (slot < 0) go to a spare row behind the buffer
  const fh_ciptr elems = (fh_ciptr)P.elems;
  const int stride = gridDim.x * NW;
  const int idx0 = blockIdx.x * NW + wave;
  if (idx0 >= P.nelems) return;
  const int last = P.nelems - 1;
  int e_cur = elems[idx0];
  int e_n = elems[min(idx0 + stride, last)], e_nn = elems[min(idx0 + 2 * stride, last)];
  int sl_cur = (lane < NC) ? (P.slot ? P.slot[(size_t)e_cur * NC + nodeofl] : idx0 * NC + nodeofl) : -1;
  size_t ro_cur = sl_cur < 0 ? sink : (size_t)sl_cur * rowbytes;
  {
    const int dof = P.elem_dof[(size_t)e_cur * P.nloc + nodeofl];
    if (lane < NC) {
      xt[lane] = P.coords[(size_t)dof * DIM];
      xt[28 + lane] = P.coords[(size_t)dof * DIM + 1];
      xt[56 + lane] = P.coords[(size_t)dof * DIM + 2];
      xt[84 + lane] = P.sol ? P.sol[dof] : 0.0;
    }
  }
  int dof_n = P.elem_dof[(size_t)e_n * P.nloc + nodeofl];
  asm volatile("" : "+v"(dof_n), "+v"(sl_cur));      // nothing pending at the loop head (see the note before the stores)
  wave_lds_sync();
#pragma unroll 1
  for (int idx = idx0; idx < P.nelems; idx += stride) {
    // ---- prefetch: node ids two elements ahead, coordinates / solution / slots one element ahead (dependent gathers) ----
    const double nx0 = P.coords[(size_t)dof_n * DIM], nx1 = P.coords[(size_t)dof_n * DIM + 1], nx2 = P.coords[(size_t)dof_n * DIM + 2];
    const double nu = P.sol ? P.sol[dof_n] : 0.0;
    const int sl_n = (lane < NC) ? (P.slot ? P.slot[(size_t)e_n * NC + nodeofl] : (idx + stride) * NC + nodeofl) : -1;
    const int dof_nn = P.elem_dof[(size_t)e_nn * P.nloc + nodeofl];
    const int e_nnn = elems[min(idx + 3 * stride, last)];
    double Kb[3][3], fsrc;
    sf_element_blocks<SRC, REGC>(P, tab, LCl, LIl, rcA, rcZ, rcY, esym, ens, ensT, eout, xt, R, lane, Kb, fsrc);
    wave_lds_sync();          // every lane is done with e: reuse it as the staging Ks[27][29], rows AND columns in tensor order
    double* Ks = R;
    {
      // lane (p, p2) holds K[(a,p)][(a2,p2)]: entry (a*9 + p, a2*9 + p2) and its mirror; on a diagonal pair the upper entries go to
      // both places (K_e symmetric bit for bit).  Immediate offsets only; lanes 45..63 repeat the last pair.
      double* kd = Ks + based;       // p * KS + p2
      double* km = Ks + basem;       // p2 * KS + p
#pragma unroll
      for (int a = 0; a < 3; a++)
#pragma unroll
        for (int a2 = 0; a2 < 3; a2++) {
          const double v = (a2 < a) ? (diag ? Kb[a2][a] : Kb[a][a2]) : Kb[a][a2];
          kd[a * 9 * KS + a2 * 9] = v;
          km[a2 * 9 * KS + a * 9] = v;
        }
    }
    wave_lds_sync();
    double ku = 0.0;
    if (P.sol) {              // residual: (K_e u)_t for tensor row t = lane & 31, half of the columns each
      const int h = lane >> 5;
      const double* kr = Ks + min(lane & 31, NC - 1) * KS + h * 14;
      const double* ur = xt + 84 + h * 14;
#pragma unroll
      for (int g = 0; g < 14; g++) {
        const bool live = g < 13 || h == 0;          // tensor column h*14 + g < 27
        const double v = live ? kr[g] : 0.0, uu = live ? ur[g] : 0.0;
        ku += v * uu;
      }
      ku += __shfl_xor(ku, 32, 64);
    }
    if (lane < NC) {          // the next element's nodes
      xt[lane] = nx0;
      xt[28 + lane] = nx1;
      xt[56 + lane] = nx2;
      xt[84 + lane] = nu;
    }
    // Every prefetched value is consumed HERE, before this element's stores are issued: the compiler does not see the hand-written
    // stores in its wait-count bookkeeping, and a vmcnt wait placed after them (for a load issued before them) would also wait for
    // all 27 row stores to reach the L2 -- once per element, at the loop head.  The residual entries go first for the same reason.
    size_t ro_n = sl_n < 0 ? sink : (size_t)sl_n * rowbytes;
    int dof_nn_c = dof_nn;
    asm volatile("" : "+v"(ro_n), "+v"(dof_nn_c));
    if (!(P.debug & 2)) {
      if (lane < NC && sl_cur >= 0) {     // by hand as well: a store the compiler tracks makes it wait for vmcnt(0) at the loop head
        const double fv = -(ku + fsrc);
        double* fp = P.Fout + sl_cur;
        asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(fp), "v"(fv) : "memory");
      }
      if (PAD) {               // P.kstride >= 28
        // Half-wave h stores the tensor rows h*14 + p, its lane j = lane & 31 the column of NODE j (tensor column tofl): one whole
        // 256-byte row per half-wave and store, off a scalar base address.  No branch per row: rows without a slot go to the spare
        // row behind the buffer; each half runs its 14 stores under one exec mask.  The LDS reads are issued by hand: left to the
        // compiler, pairs of them become ds_read2_b64 (8 LDS cycles instead of 2 + 2).
        const int j = lane & 31, hrow = lane >> 5;
        const unsigned joff = (unsigned)j * 8u;
        const unsigned kva = (unsigned)(size_t)(__attribute__((address_space(3))) double*)(Ks + hrow * 14 * KS + tofl);
        const unsigned rlo = (unsigned)ro_cur, rhi = (unsigned)(ro_cur >> 32);
        double kv[14];
#pragma unroll
        for (int p = 0; p < 14; p++) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(kv[p]) : "v"(kva), "n"(p * KS * 8) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kv[0]), "+v"(kv[1]), "+v"(kv[2]), "+v"(kv[3]), "+v"(kv[4]), "+v"(kv[5]), "+v"(kv[6]), "+v"(kv[7]),
                     "+v"(kv[8]), "+v"(kv[9]), "+v"(kv[10]), "+v"(kv[11]), "+v"(kv[12]), "+v"(kv[13]));
        if (j < P.kstride) {
          // non-temporal stores: the element rows are read once, by the row pass, long after they have left every cache (0.693 -> 0.663 ms;
          // asm_debug bit 6 selects the plain stores for comparison)
          if (!(P.debug & 64)) {
            if (hrow == 0) {
#pragma unroll
              for (int p = 0; p < 14; p++) {
                const size_t ro = ((size_t)(unsigned)__builtin_amdgcn_readlane((int)rhi, p) << 32) | (unsigned)__builtin_amdgcn_readlane((int)rlo, p);
                const char* base = reinterpret_cast<const char*>(P.Kout) + ro;
                asm volatile("global_store_dwordx2 %0, %1, %2 nt" ::"v"(joff), "v"(kv[p]), "s"(base) : "memory");
              }
            } else {
#pragma unroll
              for (int p = 0; p < 13; p++) {
                const size_t ro = ((size_t)(unsigned)__builtin_amdgcn_readlane((int)rhi, 14 + p) << 32) | (unsigned)__builtin_amdgcn_readlane((int)rlo, 14 + p);
                const char* base = reinterpret_cast<const char*>(P.Kout) + ro;
                asm volatile("global_store_dwordx2 %0, %1, %2 nt" ::"v"(joff), "v"(kv[p]), "s"(base) : "memory");
              }
            }
          } else
          if (hrow == 0) {
#pragma unroll
            for (int p = 0; p < 14; p++) {
              const size_t ro = ((size_t)(unsigned)__builtin_amdgcn_readlane((int)rhi, p) << 32) | (unsigned)__builtin_amdgcn_readlane((int)rlo, p);
              const char* base = reinterpret_cast<const char*>(P.Kout) + ro;
              asm volatile("global_store_dwordx2 %0, %1, %2" ::"v"(joff), "v"(kv[p]), "s"(base) : "memory");
            }
          } else {
#pragma unroll
            for (int p = 0; p < 13; p++) {
              const size_t ro = ((size_t)(unsigned)__builtin_amdgcn_readlane((int)rhi, 14 + p) << 32) | (unsigned)__builtin_amdgcn_readlane((int)rlo, 14 + p);
              const char* base = reinterpret_cast<const char*>(P.Kout) + ro;
              asm volatile("global_store_dwordx2 %0, %1, %2" ::"v"(joff), "v"(kv[p]), "s"(base) : "memory");
            }
          }
        }
      } else {
#pragma unroll
        for (int t0 = 0; t0 < NC * NC; t0 += 64) {
          const int t = t0 + lane;
          const int row = (t < NC * NC) ? t / NC : 0;
          const int j = t - row * NC;
          const int tr = __shfl(tofl, row, 64), tj = __shfl(tofl, j, 64);
          const int s = __shfl(sl_cur, tr, 64);
          if (t < NC * NC && s >= 0) P.Kout[(size_t)s * NC + j] = Ks[tr * KS + tj];
        }
      }
    }
    wave_lds_sync();          // Ks is the next element's phase-A scratch, xt holds the next element's nodes
    sl_cur = sl_n;
    ro_cur = ro_n;
    dof_n = dof_nn_c;
    e_n = e_nn;
    e_nn = e_nnn;
  }
#undef SF_I
#undef SF_C
#undef SF_CA
}

template <int SRC, int NW, bool PAD>
static int launch_sf_one(fh_assembler_t as, const AsmParams& P) {
  constexpr size_t lds = sf_lds_bytes(NW);
  static_assert(lds <= 160 * 1024, "k_elem_q2hex_sf: LDS budget");
  static bool attr_set[64] = {};
  const int dev = as->ctx->device & 63;
  if (!attr_set[dev]) {
    FH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_elem_q2hex_sf<SRC, NW, PAD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set[dev] = true;
  }
  const int per_cu = std::max(1, (int)((size_t)160 * 1024 / lds));
  const int grid = std::max(1, std::min(fh_div_up(P.nelems, NW), as->ctx->num_cu * per_cu * as->ctx->assemble_sf_grid));
  hipLaunchKernelGGL((k_elem_q2hex_sf<SRC, NW, PAD>), dim3(grid), dim3(NW * 64), lds, as->ctx->stream, P, as->sf_tab, as->d_sfLc, as->d_sfLi);
  FH_CHECK_HIP(hipGetLastError());
  return 0;
}

template <int SRC, bool PAD>
static int launch_sf_src(fh_assembler_t as, const AsmParams& P, int nw) {
  if (nw <= 4) return launch_sf_one<SRC, 4, PAD>(as, P);
  if (nw <= 8) return launch_sf_one<SRC, 8, PAD>(as, P);
  if (nw <= 10) return launch_sf_one<SRC, 10, PAD>(as, P);
  if (nw <= 12) return launch_sf_one<SRC, 12, PAD>(as, P);
  return launch_sf_one<SRC, 13, PAD>(as, P);
}

static int launch_sf(fh_assembler_t as, const AsmParams& P, int nw) {
  if (P.kstride >= 28) {
    if (P.source_kind == 4) return launch_sf_src<2, true>(as, P, nw);
    if (P.source_kind != 0) return launch_sf_src<1, true>(as, P, nw);
    return launch_sf_src<0, true>(as, P, nw);
  }
  if (P.source_kind == 4) return launch_sf_src<2, false>(as, P, nw);
  if (P.source_kind != 0) return launch_sf_src<1, false>(as, P, nw);
  return launch_sf_src<0, false>(as, P, nw);
}

// ------------------------------------------------------------------------------------------------------------------
// FUSED CLUSTER ASSEMBLY (round 4; default where the mesh offers it).  The two-pass design writes every element row to HBM (6.9 kB per
// element) and reads it back 0.7 ms later: 3.6 GB of the 5.7 GB the assembly moves.  Here a workgroup of eight waves owns a CLUSTER of
// eight elements that share one local topology -- the eight children of a coarse element, which the refinement numbers consecutively
// (MeshRefinement.cpp:240-294): 125 "macro" nodes, 4913 distinct (row, column) pairs.  Every wave computes one element matrix exactly as
// k_elem_q2hex_sf does and leaves it in its LDS staging; after a workgroup barrier the 512 threads sum, per macro entry, the 1 / 2 / 4 / 8
// element entries that meet there IN ASCENDING ELEMENT ORDER straight out of the eight stagings and store
//   * rows all of whose elements lie in the cluster (interior macro nodes, and boundary nodes of the domain): once, into the CSR row, in
//     CSR order (a one-byte map per entry, built on the device when the plan is made) -- these rows never visit HBM twice;
//   * the other macro rows: packed (template order, residual entry behind them) into the partial-row buffer, which the second pass
//     k_rows_partial sums per CSR row in ascending cluster order (= ascending element order) and writes once.
// Element entries are the same numbers the two-pass path produces; the sums differ from the reference's strictly sequential element order
// only in their grouping ((e0 + e1) + (e2 + e3) across two clusters instead of ((e0 + e1) + e2) + e3), i.e. by rounding.  Deterministic: no
// atomics, fixed order.  Nothing about the topology is assumed: the template is READ from cluster 0 (first-appearance numbering of its
// 8 x 27 nodes in tensor order) and every other cluster must reproduce it, otherwise the assembler keeps the two-pass path.
// Addresses inside the workgroup's LDS are 14-bit double indices; a descriptor holds the first address, the second (or the index of a
// zero cell) and, for the 49 entries with four or eight contributions, the start of an overflow list.
// ------------------------------------------------------------------------------------------------------------------
constexpr int CL_NE = 8;                         // elements per cluster = waves per workgroup
constexpr int CL_T = CL_NE * 64;                 // threads
constexpr int CL_SPT = 10;                       // slots (macro entries) per thread
constexpr int CL_NS_MAX = CL_T * CL_SPT;         // 5120 >= 4913
constexpr int CL_NM_MAX = 128;                   // macro nodes (125), padded
constexpr int CL_NBLK = 128;                     // address blocks of the entries with more than two contributions (49 used; one per lane 16 .. 31 of the eight waves)
// LDS of the cluster kernel (doubles): the per-lane tables it reads from LDS -- rows CL_LC0 .. SF_NLC - 1 of the constants (the others live in registers) and
// the integer rows --, the eight wave regions (xt + R), phase A's U / V scratch of the waves that run phase A one cluster AHEAD (waves 4 .. 7: their R
// still holds the staging of the current cluster then), a zero cell (operand of absent contributions)
constexpr int CL_LC0 = 43;
constexpr int CL_TAB = (SF_NLC - CL_LC0) * 64 + SF_NLI * 32;
constexpr int CL_UVS = 6 * SF_US + 9 * SF_VS;    // 672 doubles per wave
constexpr int CL_UV = CL_TAB + CL_NE * SF_WAVE;
constexpr int CL_ZC = CL_UV + (CL_NE / 2) * CL_UVS;
// behind them: the sums of the long entries (one cell each, written between the two barriers of the output phase), descriptors (8 B per template entry), row
// destinations of the cluster (8 B), residual destinations (4 B), address blocks (8 x 2 B) of the residual sums and of the long sums
constexpr int CL_LV = CL_ZC + 2;
constexpr size_t CL_OFF_DT = (size_t)(CL_LV + CL_NBLK) * sizeof(double);
constexpr size_t CL_OFF_RB = CL_OFF_DT + (size_t)CL_NS_MAX * 8;
constexpr size_t CL_OFF_FB = CL_OFF_RB + (size_t)CL_NM_MAX * 8;
constexpr size_t CL_OFF_FL = CL_OFF_FB + (size_t)CL_NM_MAX * 4;
constexpr size_t CL_OFF_OV = CL_OFF_FL + (size_t)CL_NM_MAX * 16;
constexpr size_t cl_lds_bytes() { return CL_OFF_OV + (size_t)CL_NBLK * 16; }
static_assert((CL_LV + CL_NBLK) * 8 < (1 << 17), "cluster kernel: LDS byte addresses of the stagings and the long-sum cells fit 17 bits");
static_assert(CL_ZC < (1 << 16), "cluster kernel: 16-bit double indices of the address blocks");
static_assert(cl_lds_bytes() <= 160 * 1024, "cluster kernel: LDS budget");
static_assert((CL_TAB % 2) == 0 && (CL_UVS % 2) == 0, "cluster kernel: 16-byte alignment of the wave regions");

struct ClParams {
  int ncl, ns, nm;
  int sup_shift;               // CARRY kernels: log2 of the clusters per super-cluster (a workgroup walks whole super-clusters, ascending cluster order)
  const uint2* dtab;           // [ns] descriptor of template entry off[r] + j: x = LDS byte address of the first contribution, y = address of the second (or of
                               //      the zero cell); an entry with more than two contributions: x = address of its long-sum cell, y = zero cell
  const uint4* fblk;           // [128] residual entry of macro row r: eight 16-bit double indices (zero cell for absent contributions)
  const uint4* oblk;           // [CL_NBLK] ALL contributions of long entry k (its sum goes to cell CL_LV + k), same format
  const unsigned* sinfo;       // [CL_SPT][CL_T]: r | p << 7 | off[r] << 14 of slot tid + CL_T * i
  const unsigned long long* vdst;   // [ncl][128] address of the row's first entry (CSR array or partial-row buffer)
  const int* fdst;             // [ncl][128] >= 0: row of the residual vector (bit 30: carried row an earlier cluster stored to -- add), bit 31: offset into the partial-row buffer
  const uint4* map;            // [ncl][CL_T]: byte i = template entry of slot tid + CL_T * i inside its row; bits 16 + i of word 2: carried entry to add to (k_cluster_maps)
  const uint4* mapb;           // CARRY kernels: [ncl][CL_T] byte i = position of the slot in its destination row
  double* Pbuf;
  double* res;
  unsigned long long* stamps;  // instrumented build only (asm_debug bit 7): [workgroup][wave][16] summed phase cycles + [15] = clusters done
};

// t + eight contributions, added one after the other in ascending element order (the order of the two-pass row pass inside one cluster)
__device__ __forceinline__ double cl_sum8(const char* S, const uint4 a, double t) {
  const unsigned w[4] = {a.x, a.y, a.z, a.w};
  double v[8];
#pragma unroll
  for (int k = 0; k < 8; k++) v[k] = *reinterpret_cast<const double*>(S + (((w[k >> 1] >> (16 * (k & 1))) & 0xffffu) << 3));
#pragma unroll
  for (int k = 0; k < 8; k++) t += v[k];
  return t;
}

// CARRY: the workgroup walks super-clusters (1 << C.sup_shift consecutive clusters, one after the other); a row whose elements all lie in the super-cluster is
// accumulated in the CSR array itself -- the first cluster that holds an entry stores it, the later ones load, add and store (same workgroup, a barrier in
// between; ascending cluster order = the order of the second pass, so the bits are the ones the partial-row buffer gives) -- and only the rows on the surface
// of the super-cluster go through the partial-row buffer.  (Measured and dropped: the add as a floating-point atomic without return, global_atomic_add_f64 --
// same bits, no round trip for the wave, but 17 M of them per assembly took 1.2 ms.)
template <int SRC, bool INS = false, bool ROT = true, bool CARRY = false>
__global__ __launch_bounds__(CL_T) void k_cluster_q2hex_sf(AsmParams P, SfTab tab, const double* __restrict__ lanec, const int* __restrict__ lanei, ClParams C) {
  constexpr int NC = 27, DIM = 3, KS = MF_KS, NW = CL_NE;
  constexpr bool REGC = true;
  extern __shared__ __attribute__((aligned(16))) double sf_smem[];
  double* SFl = sf_smem;                     // rows CL_LC0 .. of the per-lane constants
  int* SFi = reinterpret_cast<int*>(SFl + (SF_NLC - CL_LC0) * 64);
  char* Sb = reinterpret_cast<char*>(sf_smem);
  uint2* dtab = reinterpret_cast<uint2*>(Sb + CL_OFF_DT);
  unsigned long long* rb = reinterpret_cast<unsigned long long*>(Sb + CL_OFF_RB);
  int* fb = reinterpret_cast<int*>(Sb + CL_OFF_FB);
  uint4* fblk = reinterpret_cast<uint4*>(Sb + CL_OFF_FL);
  uint4* oblk = reinterpret_cast<uint4*>(Sb + CL_OFF_OV);
  const int tid = threadIdx.x;
  for (int k = tid; k < (SF_NLC - CL_LC0) * 64; k += CL_T) SFl[k] = lanec[CL_LC0 * 64 + k];
  for (int k = tid; k < SF_NLI * 64; k += CL_T) SFi[k] = lanei[k];
  for (int k = tid; k < C.ns; k += CL_T) dtab[k] = C.dtab[k];
  if (tid < CL_NM_MAX) fblk[tid] = C.fblk[tid];
  if (tid < CL_NBLK) oblk[tid] = C.oblk[tid];
  if (tid < 2) sf_smem[CL_ZC + tid] = 0.0;
  __syncthreads();
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // waves w and w + 4 share a SIMD (a workgroup's waves go round the four SIMDs).  Left alone, all eight waves run the same phase at the same time and the
  // LDS-bound and the arithmetic-bound phases of an element take turns on the compute unit (measured with the phase stamps: LDS busy 54 %, vector ALU 33 %,
  // together 87 % of the element time).  Waves 4 .. 7 therefore run phase A one cluster AHEAD -- after their staging is written, from a scratch region of
  // their own -- so that on every SIMD one wave is in the short, latency-bound phase A while the other is in the long stages 2-3 (asm_debug bit 8: off)
  const bool ahead = ROT && ((P.debug & 1024) ? wave >= CL_NE / 2 : wave < CL_NE / 2) && !(P.debug & 256);
  double* xt = SFl + CL_TAB + wave * SF_WAVE;
  double* R = xt + SF_XT;
  double* UVa = ahead ? SFl + CL_UV + (wave & (CL_NE / 2 - 1)) * CL_UVS : R;
  const double* LCl = SFl + lane - CL_LC0 * 64;       // row r of the constants at LCl[r * 64], r >= CL_LC0
  const int* LIl = SFi + lane;
  double rcA[19], rcZ[8], rcY[16];
#pragma unroll
  for (int r = 0; r < 19; r++) rcA[r] = lanec[r * 64 + lane];
#pragma unroll
  for (int r = 0; r < 8; r++) rcZ[r] = lanec[(19 + r) * 64 + lane];
#pragma unroll
  for (int r = 0; r < 16; r++) rcY[r] = lanec[(27 + r) * 64 + lane];
  // The per-lane integers of the element stages stay in registers, PACKED (the kernel runs at the 256-register limit of two waves per SIMD, and an LDS
  // look-up each put a dependent round trip in front of every contraction): they are unpacked per cluster from values the compiler must treat as
  // changing (cl_opaque), otherwise it would hoist the unpacked forms out of the loop again.
  //   pk0 = esym | ens << 8 | ensT << 16 | nodeofl << 24 | diag << 29      (slot offsets inside the e arrays < 216; node of tensor index min(lane, 26))
  //   pk1 = based | basem << 16                                           (staging offsets < 783)
  //   pk2 = phase A's five indices, six bits each (xn <= 26, uo / ui < SF_US, vo / vi < SF_VS)
  unsigned pk0 = (unsigned)lanei[5 * 64 + lane] | ((unsigned)lanei[6 * 64 + lane] << 8) | ((unsigned)lanei[15 * 64 + lane] << 16) | ((unsigned)lanei[14 * 64 + lane] << 24) |
                 ((lanei[9 * 64 + lane] != 0 ? 1u : 0u) << 29);
  unsigned pk1 = (unsigned)lanei[7 * 64 + lane] | ((unsigned)lanei[8 * 64 + lane] << 16);
  unsigned pk2 = (unsigned)lanei[lane] | ((unsigned)lanei[64 + lane] << 6) | ((unsigned)lanei[2 * 64 + lane] << 12) | ((unsigned)lanei[3 * 64 + lane] << 18) |
                 ((unsigned)lanei[4 * 64 + lane] << 24);
#define CL_OPAQUE(x) asm volatile("" : "+v"(x))
#define CL_IDXA() SfIdxA{(int)(pk2 & 63u), (int)((pk2 >> 6) & 63u), (int)((pk2 >> 12) & 63u), (int)((pk2 >> 18) & 63u), (int)(pk2 >> 24)}
  const int nodeofl = (int)((pk0 >> 24) & 31u);
  const int cstride = gridDim.x;
  // iteration `it` of this workgroup serves cluster ((blockIdx.x + (it >> sh) * gridDim.x) << sh) + (it & ((1 << sh) - 1)); sh = 0 without CARRY
  const int sh = CARRY ? C.sup_shift : 0;
  const int nsup = C.ncl >> sh;
  if ((int)blockIdx.x >= nsup) return;
  const int nit = ((nsup - 1 - (int)blockIdx.x) / cstride + 1) << sh;
  auto cl_at = [&](int it) {
    it = min(it, nit - 1);
    return ((((int)blockIdx.x + (it >> sh) * cstride)) << sh) + (it & ((1 << sh) - 1));
  };
  int cl = cl_at(0);
  {
    const int dof = P.elem_dof[(size_t)(cl * NW + wave) * P.nloc + nodeofl];
    if (lane < NC) {
      xt[lane] = P.coords[(size_t)dof * DIM];
      xt[28 + lane] = P.coords[(size_t)dof * DIM + 1];
      xt[56 + lane] = P.coords[(size_t)dof * DIM + 2];
      xt[84 + lane] = P.sol ? P.sol[dof] : 0.0;
    }
  }
  int dof_n = P.elem_dof[(size_t)(cl_at(1) * NW + wave) * P.nloc + nodeofl];
  wave_lds_sync();
  SfStamps st;
  if (INS) {
#pragma unroll
    for (int k = 0; k < 16; k++) st.acc[k] = 0;
    st.prev = __builtin_amdgcn_s_memtime();
  }
  int ndone = 0;
  double Dr[7];              // D_q and the source weight at this lane's Gauss point: phase A -> stage 1
  if (ahead) {
    sf_phase_a<SRC, REGC, INS>(P, LCl, CL_IDXA(), rcA, xt, UVa, Dr, &st);
    wave_lds_sync();
  }
#pragma unroll 1
  for (int it = 0; it < nit; it++) {
    cl = cl_at(it);
    sf_stamp<INS>(st, 0);
    CL_OPAQUE(pk0);
    CL_OPAQUE(pk1);
    CL_OPAQUE(pk2);
    const int esym = (int)(pk0 & 255u), ens = (int)((pk0 >> 8) & 255u), ensT = (int)((pk0 >> 16) & 255u);
    const int eout = (lane >> 4) * SF_ES + (lane & 15);
    // ---- prefetch (dependent gathers): coordinates / solution of the wave's next element, node ids of the one after ----
    const int cl_nn = cl_at(it + 2);
    const double nx0 = P.coords[(size_t)dof_n * DIM], nx1 = P.coords[(size_t)dof_n * DIM + 1], nx2 = P.coords[(size_t)dof_n * DIM + 2];
    const double nu = P.sol ? P.sol[dof_n] : 0.0;
    const int dof_nn = P.elem_dof[(size_t)(cl_nn * NW + wave) * P.nloc + nodeofl];
    double Kb[3][3], fsrc;
    if (!ahead) {
      sf_phase_a<SRC, REGC, INS>(P, LCl, CL_IDXA(), rcA, xt, UVa, Dr, &st);
      wave_lds_sync();
    }
    sf_stages<SRC, REGC, INS>(P, tab, LCl, LIl, rcZ, rcY, esym, ens, ensT, eout, R, lane, Dr, Kb, fsrc, &st);
    wave_lds_sync();          // every lane is done with e: reuse it as the staging Ks[27][29], rows AND columns in tensor order
    sf_stamp<INS>(st, 6);
    // this cluster's destinations and maps, and the thread's slot table (the same for every cluster: an L1 / L2 hit): issued only now, behind the stages
    // that need every register, and consumed after the workgroup barrier -- the staging, the residual and the wait at the barrier hide the round trip
    const int tm = tid & (CL_NM_MAX - 1);
    const unsigned long long vd_cur = C.vdst[(size_t)cl * CL_NM_MAX + tm];
    const int fd_cur = C.fdst[(size_t)cl * CL_NM_MAX + tm];
    const uint4 mp_cur = C.map[(size_t)cl * CL_T + tid];
    // CARRY: positions of the slots in their rows; residual destination of the row this lane sums in the output phase (wave * 16 + lane % 16)
    const uint4 mq_cur = CARRY ? C.mapb[(size_t)cl * CL_T + tid] : make_uint4(0u, 0u, 0u, 0u);
    const int fv_own = CARRY ? C.fdst[(size_t)cl * CL_NM_MAX + wave * 16 + (lane & 15)] : 0;
    unsigned sinfo[CL_SPT];
#pragma unroll
    for (int i = 0; i < CL_SPT; i++) sinfo[i] = C.sinfo[i * CL_T + tid];
    double* Ks = R;
    {
      const bool diag = ((pk0 >> 29) & 1u) != 0;
      double* kd = Ks + (pk1 & 0xffffu);       // p * KS + p2
      double* km = Ks + (pk1 >> 16);           // p2 * KS + p
#pragma unroll
      for (int a = 0; a < 3; a++)
#pragma unroll
        for (int a2 = 0; a2 < 3; a2++) {
          const double v = (a2 < a) ? (diag ? Kb[a2][a] : Kb[a][a2]) : Kb[a][a2];
          kd[a * 9 * KS + a2 * 9] = v;
          km[a2 * 9 * KS + a * 9] = v;
        }
    }
    wave_lds_sync();
    sf_stamp<INS>(st, 7);
    double ku = 0.0;
    if (P.sol) {              // residual: (K_e u)_t for tensor row t = lane & 31, half of the columns each
      const int h = lane >> 5;
      const double* kr = Ks + min(lane & 31, NC - 1) * KS + h * 14;
      const double* ur = xt + 84 + h * 14;
#pragma unroll
      for (int g = 0; g < 14; g++) {
        const bool live = g < 13 || h == 0;
        const double v = live ? kr[g] : 0.0, uu = live ? ur[g] : 0.0;
        ku += v * uu;
      }
      ku += __shfl_xor(ku, 32, 64);
    }
    Ks[min(lane & 31, NC - 1) * KS + NC] = -(ku + fsrc);   // residual entry of tensor row t = lane & 31 in the staging's spare column (both halves hold the same value; lanes beyond 26 repeat row 26)
    if (lane < NC) {          // the next element's nodes
      xt[lane] = nx0;
      xt[28 + lane] = nx1;
      xt[56 + lane] = nx2;
      xt[84 + lane] = nu;
    }
    if (tid < CL_NM_MAX) {
      rb[tid] = vd_cur;
      fb[tid] = fd_cur;
    }
    sf_stamp<INS>(st, 8);
    // the address block of this lane's sum of eight in the output phase (a table: read ahead of the barrier, one LDS round trip less behind it)
    const int frow = wave * 16 + (lane & 15);
    uint4 sumblk = (lane & 16) ? oblk[frow] : fblk[frow];
    // CARRY: the residual entry of a carried row that an earlier cluster of the super-cluster stored (same workgroup, a barrier ago): on its way before the barrier
    typedef __attribute__((address_space(1))) double cl_gdouble;      // global address space: a generic (flat) access would also count as an LDS operation
    if (ahead) {              // phase A of the wave's NEXT element (its nodes are in xt now; behind the last cluster: the clamped element again, unused)
      wave_lds_sync();
      sf_phase_a<SRC, REGC, INS>(P, LCl, CL_IDXA(), rcA, xt, UVa, Dr, &st);
    }
    __syncthreads();
    sf_stamp<INS>(st, 9);
    if (!(P.debug & 2)) {
      const unsigned mw[4] = {mp_cur.x, mp_cur.y, mp_cur.z, mp_cur.w};
      const unsigned mq[4] = {mq_cur.x, mq_cur.y, mq_cur.z, mq_cur.w};
      // CARRY: the residual entry of a carried row that an earlier cluster of the super-cluster stored (same workgroup, a barrier ago); consumed behind the stores
      double fold = 0.0;
      if (CARRY && lane < 16 && fv_own >= 0 && (fv_own & 0x40000000))
        fold = __hip_atomic_load((const cl_gdouble*)(C.res + (size_t)(fv_own & 0x3fffffff)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // in stages, every stage's LDS reads independent of each other (two waves per SIMD hide little latency): descriptors and row destinations; the
      // residual entries and the 49 entries with four or eight contributions (one sum of eight per lane, lanes 0 .. 15 / 16 .. 31 of every wave; the long
      // sums go to their cells); a second barrier; both operands of every entry (the zero cell for entries of one element, the cell for a long one); stores.
      uint2 dd[CL_SPT];
      unsigned long long vb[CL_SPT];       // destination of the slot (address of the row, plus the position)
      double accv[CARRY ? CL_SPT : 1];     // CARRY: what the earlier clusters of the super-cluster left at a carried entry
#pragma unroll
      for (int i = 0; i < CL_SPT; i++) {
        const unsigned si = sinfo[i];
        const int j = (mw[i >> 2] >> (8 * (i & 3))) & 255;
        const int p = (si >> 7) & 127;
        const int pos = CARRY ? (int)((mq[i >> 2] >> (8 * (i & 3))) & 255) : p;
        dd[i] = dtab[(si >> 14) + j];
        vb[i] = rb[si & 127] + (unsigned long long)(pos * 8);
      }
      if (CARRY) {      // (issuing these ahead of the descriptor reads was measured: the loads cost 0.05 instead of 0.07 ms, the rest of the phase 0.02 more; skipping the
                        // block by a wave-uniform test of the flags: 0.96 -> 1.45 ms, the ten values then live in scratch)
#pragma unroll
        for (int i = 0; i < CL_SPT; i++) {
          accv[i] = 0.0;
          if (((mw[2] >> (16 + i)) & 1u) && !(P.debug & 512)) accv[i] = __hip_atomic_load((const cl_gdouble*)vb[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (bit 9: timing aid, WRONG values)
        }
      }
      double fsum;
      int fvv;
      {
        const double v = cl_sum8(Sb, sumblk, 0.0);
        const int fv = CARRY ? fv_own : fb[frow];
        double* dst = fv < 0 ? C.Pbuf + (size_t)(fv & 0x7fffffff) : C.res + (size_t)(CARRY ? (fv & 0x3fffffff) : fv);
        if (!CARRY) { if (lane < 16) *dst = v; }       // CARRY: stored behind the second barrier (waiting for `fold` here would wait for the carried entries too)
        if (lane >= 16 && lane < 32) sf_smem[CL_LV + frow] = v;
        fsum = v;
        fvv = fv;
      }
      if (INS) {
        asm volatile("" ::"v"(dd[CL_SPT - 1].x), "v"(vb[CL_SPT - 1]));
        sf_stamp<INS>(st, 12);
      }
      __syncthreads();
      sf_stamp<INS>(st, 14);
      double vv[CL_SPT], ww[CL_SPT];
#pragma unroll
      for (int i = 0; i < CL_SPT; i++) {
        vv[i] = *reinterpret_cast<const double*>(Sb + dd[i].x);
        ww[i] = *reinterpret_cast<const double*>(Sb + dd[i].y);
      }
#pragma unroll
      for (int i = 0; i < CL_SPT; i++) vv[i] += ww[i];
      if (CARRY) {
#pragma unroll
        for (int i = 0; i < CL_SPT; i++) vv[i] = accv[i] + vv[i];      // (earlier clusters, or 0) + this cluster: the second pass's order
      }
      if (INS) {
        asm volatile("" ::"v"(vv[CL_SPT - 1]), "v"(vv[0]));
        sf_stamp<INS>(st, 13);
      }
#pragma unroll
      for (int i = 0; i < CL_SPT; i++) {
        cl_gdouble* dst = (cl_gdouble*)vb[i];
        if (P.debug & 64) *dst = vv[i];                  // bit 6: plain stores (comparison)
        else if (!(P.debug & 4)) __builtin_nontemporal_store(vv[i], dst);
        else if (vv[i] == 1.2345e300) *dst = vv[i];      // timing aid (bit 2): the LDS work without the stores
      }
      if (CARRY && lane < 16) {
        cl_gdouble* dst = (cl_gdouble*)(fvv < 0 ? C.Pbuf + (size_t)(fvv & 0x7fffffff) : C.res + (size_t)(fvv & 0x3fffffff));
        *dst = (fvv >= 0 && (fvv & 0x40000000)) ? fold + fsum : fsum;
      }
      sf_stamp<INS>(st, 15);
    }
    sf_stamp<INS>(st, 10);
    __syncthreads();          // the stagings are the next elements' scratch, rb / fb the next cluster's
    sf_stamp<INS>(st, 11);
    dof_n = dof_nn;
    ndone++;
  }
  if (INS && C.stamps && lane == 0) {
    unsigned long long* o = C.stamps + ((size_t)blockIdx.x * NW + wave) * 20;
#pragma unroll
    for (int k = 0; k < 16; k++) o[k] = st.acc[k];
    o[16] = (unsigned long long)ndone;
  }
}

// Second pass of the fused assembly: every CSR row that is not complete inside one cluster sums its partial macro rows (ascending cluster
// order) in LDS and is written once; rows no element touches are written as zeros.  The partial rows of one CSR row lie BEHIND EACH OTHER in
// the partial-row buffer (the cluster kernel scatters whole packed rows of 28 ... 126 entries there), so a 32-lane group streams one
// contiguous segment per row, two rows at once: [prow, pstart, rowptr] -> [segments + byte maps] -> LDS adds -> store.  Order of the sums:
// two entries of a segment meet in one CSR position only from different partial rows, i.e. >= 28 entries apart; the LDS adds are issued
// 16 entries at a time in ascending segment order and LDS executes a wave's operations in order, so every position receives its
// contributions in ascending cluster order -- deterministic without staging the segment.  Map byte 255 = the residual entry behind a partial row.
constexpr int RP_Q = 7;                      // 32 * 7 = 224 >= the longest segment (8 partial rows of 27 + 1)
template <int NT>
__global__ __launch_bounds__(256) void k_rows_partial(int nprow, const int* __restrict__ prow, const unsigned* __restrict__ pstart, const unsigned char* __restrict__ pmap,
                                                      const double* __restrict__ Pbuf, const int* __restrict__ rowptr, double* __restrict__ val, double* __restrict__ res, int nt) {
  __shared__ double acc[8][2][128];
  const int sub = threadIdx.x >> 5, lane = threadIdx.x & 31;
  const int k0 = (blockIdx.x * 8 + sub) * 2;
  if (k0 >= nprow) return;
  const bool two = k0 + 1 < nprow;
  const size_t sA = pstart[k0], sB = pstart[k0 + 1], sC = two ? pstart[k0 + 2] : sB;
  const int TA = (int)(sB - sA), TB = (int)(sC - sB);
  const int gA = prow[k0], gB = two ? prow[k0 + 1] : gA;
  const int rsA = rowptr[gA], lenA = rowptr[gA + 1] - rsA, rsB = rowptr[gB], lenB = two ? rowptr[gB + 1] - rsB : 0;
  double vA[RP_Q], vB[RP_Q];
  int mA[RP_Q], mB[RP_Q];
#pragma unroll
  for (int q = 0; q < RP_Q; q++) {
    const int e = lane + 32 * q;
    vA[q] = e < TA ? (NT ? __builtin_nontemporal_load(&Pbuf[sA + e]) : Pbuf[sA + e]) : 0.0;
    mA[q] = e < TA ? (int)pmap[sA + e] : 254;
    vB[q] = e < TB ? (NT ? __builtin_nontemporal_load(&Pbuf[sB + e]) : Pbuf[sB + e]) : 0.0;
    mB[q] = e < TB ? (int)pmap[sB + e] : 254;
  }
#pragma unroll
  for (int q = 0; q < 4; q++) {
    acc[sub][0][lane + 32 * q] = 0.0;
    acc[sub][1][lane + 32 * q] = 0.0;
  }
  double fA = 0.0, fB = 0.0;
#pragma unroll
  for (int q = 0; q < RP_Q; q++) {
    if (32 * q < TA || 32 * q < TB) {
#pragma unroll
      for (int h = 0; h < 2; h++) {
        if ((lane >> 4) == h && mA[q] < 128) __hip_atomic_fetch_add(&acc[sub][0][mA[q]], vA[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((lane >> 4) == h && mB[q] < 128) __hip_atomic_fetch_add(&acc[sub][1][mB[q]], vB[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    {   // residual entries (one behind each partial row): added one after the other in ascending segment = cluster order, on every lane alike
      unsigned mk = (unsigned)(__ballot(mA[q] == 255) >> (32 * (sub & 1)));
      while (mk) {
        fA += __shfl(vA[q], __builtin_ctz(mk), 32);
        mk &= mk - 1;
      }
      mk = (unsigned)(__ballot(mB[q] == 255) >> (32 * (sub & 1)));
      while (mk) {
        fB += __shfl(vB[q], __builtin_ctz(mk), 32);
        mk &= mk - 1;
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  for (int p = lane; p < lenA; p += 32) val[rsA + p] = acc[sub][0][p];
  for (int p = lane; p < lenB; p += 32) val[rsB + p] = acc[sub][1][p];
  if (lane == 0) {
    res[gA] = fA;
    if (two) res[gB] = fB;
  }
}

// Plan construction on the device: the map words of one cluster per workgroup; a thread writes the 16-byte words of one thread of the cluster kernel.
// Slot i of a thread (slot tid + CL_T * i = macro row r, position p) computes template entry `e` of the row and stores it at position `q` of the row's destination:
//   complete row: e = the entry whose macro column carries the global column at CSR position p, q = p (the row is written in CSR order)
//   carried row:  e = p, q = the CSR position of that entry's column (stored, or added to what the earlier clusters of the super-cluster left there)
//   partial row:  e = q = p (packed row in the partial-row buffer; pmap gets the CSR position for the second pass)
//   map  (always):        bytes 0 .. 9 = e of slot i; bits 16 + i of word 2: carried entry that an EARLIER cluster has stored already (load, add, store)
//   mapb (carried plans): bytes 0 .. 9 = q of slot i; bits 16 + i of word 2: carried entry to which a LATER cluster still adds (the Galerkin product reads an
//                         entry in the cluster that made the last contribution)
// An entry that cannot be placed raises err (the assembler then keeps the two-pass path).  first_cnt[g] counts the first contributions to carried row g: the
// caller checks that they cover the whole CSR row (a position nobody stores to would keep stale values).
__global__ __launch_bounds__(256) void k_cluster_maps(int ncl, int ns, int nm, const int* __restrict__ cdof, const unsigned* __restrict__ sinfo, const unsigned short* __restrict__ roff,
                                                      const unsigned char* __restrict__ tcol, const int* __restrict__ vdst, int m, const int* __restrict__ rowptr, const int* __restrict__ col,
                                                      const unsigned char* __restrict__ rowcls, const int* __restrict__ aptr, const int* __restrict__ aei,
                                                      const int* __restrict__ elem_dof, int nloc, uint4* __restrict__ map, uint4* __restrict__ mapb, unsigned char* __restrict__ pmap,
                                                      unsigned* __restrict__ first_cnt, int* __restrict__ err) {
  __shared__ int cd[CL_NM_MAX];
  const int c = blockIdx.x;
  if (threadIdx.x < CL_NM_MAX) cd[threadIdx.x] = threadIdx.x < nm ? cdof[(size_t)c * CL_NM_MAX + threadIdx.x] : -1;
  __syncthreads();
  auto find_pos = [&](int g, int target) {       // position of column `target` in CSR row g, -1 if absent
    const int rs = rowptr[g];
    int lo = rs, hi = rowptr[g + 1] - 1;
    while (lo <= hi) {
      const int mid = lo + ((hi - lo) >> 1);
      const int cc = col[mid];
      if (cc == target) return mid - rs;
      if (cc < target) lo = mid + 1; else hi = mid - 1;
    }
    return -1;
  };
  for (int tid = threadIdx.x; tid < CL_T; tid += 256) {
    unsigned w[4] = {0u, 0u, 0u, 0u}, wb[4] = {0u, 0u, 0u, 0u};
    for (int i = 0; i < CL_SPT; i++) {
      const unsigned si = sinfo[i * CL_T + tid];
      const int r = si & 127, p = (si >> 7) & 127, o = si >> 14;
      unsigned byte = (unsigned)p, pos_b = (unsigned)p, cls = 0, acc = 0, last = 1;
      if (r < nm) {
        const int len = roff[r + 1] - o;
        const int vb = vdst[(size_t)c * CL_NM_MAX + r];
        if (vb >= 0) {
          const int g = cd[r];
          cls = rowcls[g];
          if (cls != 2) {                    // complete row: CSR position p holds global column h
            cls = 1;
            const int h = col[vb + p];
            int j = -1;
            for (int jj = 0; jj < len; jj++)
              if (cd[tcol[o + jj]] == h) j = jj;
            if (j < 0) { atomicExch(err, 1); j = 0; }
            byte = (unsigned)j;
          } else {                           // carried row: entry p goes to (is added at) the CSR position of its column
            const int target = cd[tcol[o + p]];
            int pos = find_pos(g, target);
            if (pos < 0 || pos > 127) { atomicExch(err, 1); pos = 0; }
            pos_b = (unsigned)pos;
            bool later = false;
            if (target < m) {        // the elements that hold BOTH nodes: the two element lists are sorted, a merge finds the common ones (<= 8 + 8 steps)
              int a = aptr[g], b = aptr[target];
              const int ae = aptr[g + 1], be = aptr[target + 1];
              while (a < ae && b < be) {
                const int ea = aei[a] >> 5, eb = aei[b] >> 5;
                if (ea == eb) {
                  const int cc = ea / CL_NE;
                  if (cc < c) acc = 1;
                  else if (cc > c) later = true;
                  a++;
                  b++;
                } else if (ea < eb) a++;
                else b++;
              }
            } else {                 // a column outside the matrix (ghost) has no element list: look into the elements around g
              for (int a = aptr[g]; a < aptr[g + 1]; a++) {
                const int e = aei[a] >> 5, cc = e / CL_NE;
                if (cc == c) continue;
                bool has = false;
                for (int n = 0; n < 27; n++) has = has || elem_dof[(size_t)e * nloc + n] == target;
                if (has) { if (cc < c) acc = 1; else later = true; }
              }
            }
            last = later ? 0u : 1u;
            if (!acc) atomicAdd(&first_cnt[g], 1u);
          }
        } else {
          byte = (unsigned)p;
          const int g = cd[r];
          if (g < m) {                       // partial row of the matrix: where does packed entry p go in CSR row g
            int pos = find_pos(g, cd[tcol[o + p]]);
            if (pos < 0 || pos > 127) { atomicExch(err, 1); pos = 0; }
            pmap[(size_t)(vb & 0x7fffffff) + p] = (unsigned char)pos;
          }
        }
      }
      w[i >> 2] |= byte << (8 * (i & 3));
      wb[i >> 2] |= pos_b << (8 * (i & 3));
      w[2] |= acc << (16 + i);
      wb[2] |= (last ? 0u : 1u) << (16 + i);
    }
    map[(size_t)c * CL_T + tid] = make_uint4(w[0], w[1], w[2], w[3]);
    if (mapb) mapb[(size_t)c * CL_T + tid] = make_uint4(wb[0], wb[1], wb[2], wb[3]);
  }
}
// every position of a carried CSR row must receive exactly one first contribution
__global__ __launch_bounds__(256) void k_cl_cover(int m, const int* __restrict__ rowptr, const unsigned char* __restrict__ rowcls, const unsigned* __restrict__ first_cnt,
                                                  int* __restrict__ err, unsigned long long* __restrict__ ncarried) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= m || rowcls[g] != 2) return;
  const unsigned len = (unsigned)(rowptr[g + 1] - rowptr[g]);
  if (first_cnt[g] != len) atomicExch(err, 1);
  atomicAdd(ncarried, (unsigned long long)len);
}


// Plan construction on the device, second half (round 4): what the host loops over all (cluster, macro row) pairs did.  The template of cluster 0
// travels as a kernel argument.
struct ClTmpl {
  unsigned char tm[CL_NE][27];         // macro node of (element of the cluster, tensor index)
  unsigned char first_e[CL_NM_MAX], first_t[CL_NM_MAX], tcnt[CL_NM_MAX];
  unsigned char nodeof[27], tof[27];   // node of a tensor index and back
  unsigned short roff[CL_NM_MAX + 1];
  int nm;
};
// nodes of every cluster in template order + verification: the cluster reproduces the template with nm distinct nodes (err 1 / 2 otherwise)
__global__ __launch_bounds__(128) void k_cl_cdof(int ncl, int nloc, const int* __restrict__ elem_dof, ClTmpl T, int* __restrict__ cdof, int* __restrict__ err) {
  __shared__ int cd[CL_NM_MAX];
  const int c = blockIdx.x, k = threadIdx.x;
  cd[k] = k < T.nm ? elem_dof[(size_t)(c * CL_NE + T.first_e[k]) * nloc + T.nodeof[T.first_t[k]]] : -1;
  __syncthreads();
  for (int idx = k; idx < CL_NE * 27; idx += 128) {
    const int e = idx / 27, t = idx % 27;
    if (elem_dof[(size_t)(c * CL_NE + e) * nloc + T.nodeof[t]] != cd[T.tm[e][t]]) atomicExch(err, 1);
  }
  if (k < T.nm)
    for (int j = 0; j < k; j++)
      if (cd[j] == cd[k]) atomicExch(err, 2);          // two macro nodes, one mesh node
  cdof[(size_t)c * CL_NM_MAX + k] = cd[k];
}
// a macro row is complete (class 1) when all elements around its node lie in this cluster and the CSR row has exactly its columns; carried (class 2) when
// they all lie in this cluster's super-cluster (the clusters c >> sup_shift: one workgroup walks them in ascending order); the others add their
// packed length (+ the residual entry) to the segment of their CSR row
__global__ __launch_bounds__(256) void k_cl_rows(int ncl, int m, const int* __restrict__ cdof, const int* __restrict__ aptr, const int* __restrict__ aei, const int* __restrict__ rowptr,
                                                 ClTmpl T, int sup_shift, unsigned char* __restrict__ complete, unsigned* __restrict__ rowsz) {
  const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= (size_t)ncl * CL_NM_MAX) return;
  const int c = (int)(q / CL_NM_MAX), r = (int)(q & (CL_NM_MAX - 1));
  if (r >= T.nm) return;
  const int g = cdof[q];
  if (g >= m) return;
  const int len = T.roff[r + 1] - T.roff[r];
  if (aptr[g + 1] - aptr[g] == T.tcnt[r] && rowptr[g + 1] - rowptr[g] == len) { complete[g] = 1; return; }
  bool carried = sup_shift > 0;
  for (int a = aptr[g]; a < aptr[g + 1] && carried; a++) carried = (((aei[a] >> 5) / CL_NE) >> sup_shift) == (c >> sup_shift);
  if (carried) complete[g] = 2;          // (every cluster around g arrives at the same answer)
  else atomicAdd(&rowsz[g], (unsigned)len + 1u);
}
// destinations: complete rows point into the CSR arrays; a partial row sits in the segment of its CSR row behind the partial rows of the clusters
// before it -- found by walking the node's (element, local row) list, which is sorted by element: one step per earlier cluster
__global__ __launch_bounds__(256) void k_cl_dst(int ncl, int m, const int* __restrict__ cdof, const int* __restrict__ aptr, const int* __restrict__ aei,
                                                const int* __restrict__ rowptr, ClTmpl T, const unsigned char* __restrict__ complete, const unsigned* __restrict__ rowbase,
                                                unsigned sink, int* __restrict__ vdst, int* __restrict__ fdst) {
  const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= (size_t)ncl * CL_NM_MAX) return;
  const int c = (int)(q / CL_NM_MAX), r = (int)(q & (CL_NM_MAX - 1));
  const int g = r < T.nm ? cdof[q] : -1;
  if (g < 0 || g >= m) {
    vdst[q] = (int)sink;
    fdst[q] = (int)sink;
    return;
  }
  if (complete[g]) {       // complete or carried: the CSR row itself; bit 30 of the residual destination: an earlier cluster has stored there already (add)
    vdst[q] = rowptr[g];
    fdst[q] = g | ((complete[g] == 2 && ((aei[aptr[g]] >> 5) / CL_NE) < c) ? 0x40000000 : 0);
    return;
  }
  unsigned off = rowbase[g];
  int prev = -1;
  for (int a = aptr[g]; a < aptr[g + 1]; a++) {
    const int ei = aei[a], e = ei >> 5, cc = e / CL_NE;
    if (cc >= c) break;
    if (cc != prev) {
      prev = cc;
      const int r2 = T.tm[e % CL_NE][T.tof[ei & 31]];
      off += (unsigned)(T.roff[r2 + 1] - T.roff[r2]) + 1u;
    }
  }
  vdst[q] = (int)(0x80000000u | off);
  fdst[q] = (int)(0x80000000u | (off + (unsigned)(T.roff[r + 1] - T.roff[r])));
}

// Plan of the fused cluster assembly (see k_cluster_q2hex_sf).  Host: template from cluster 0, verification of every cluster, completeness of
// every (cluster, macro row), offsets of the partial rows, the lists of the second pass -- O(nel * 27) integer work; device: the byte maps.
// Returns 0 and leaves as->fused false when the mesh does not offer the structure.
static int cluster_plan_build(fh_assembler_t as, fh_mat_t A, const int* elem_dof, const std::vector<int>& aptr, bool allow_carry = true) {
  fh_ctx_t ctx = as->ctx;
  const int nel = as->nel, nloc = as->nloc, m = A->m, KS = MF_KS;
  int nodeof[27];                        // node of tensor index a*9 + b*3 + c (the staging's row / column order)
  for (int n = 0; n < 27; n++) nodeof[(fhfe::xc(as->geom, n, 0) + 1) * 9 + (fhfe::xc(as->geom, n, 1) + 1) * 3 + fhfe::xc(as->geom, n, 2) + 1] = n;
  if (nel < CL_NE || nel % CL_NE || A->max_row > 128 || A->h_rowptr.empty()) return 0;
  const int ncl = nel / CL_NE;
  // template: first-appearance numbering of the 8 x 27 nodes of cluster 0, tensor order inside an element
  int tm[CL_NE][27], nm = 0;
  {
    std::vector<int> seen;
    for (int e = 0; e < CL_NE; e++)
      for (int t = 0; t < 27; t++) {
        const int g = elem_dof[(size_t)e * nloc + nodeof[t]];
        int k = -1;
        for (int q = 0; q < (int)seen.size(); q++)
          if (seen[q] == g) k = q;
        if (k < 0) { k = (int)seen.size(); seen.push_back(g); }
        tm[e][t] = k;
      }
    nm = (int)seen.size();
  }
  if (nm > CL_NM_MAX - 1) return 0;
  int first_e[CL_NM_MAX], first_t[CL_NM_MAX], tcnt[CL_NM_MAX];
  for (int k = 0; k < nm; k++) { first_e[k] = -1; tcnt[k] = 0; }
  for (int e = 0; e < CL_NE; e++) {
    bool dup[CL_NM_MAX] = {};
    for (int t = 0; t < 27; t++) {
      const int k = tm[e][t];
      if (dup[k]) return 0;                      // a node twice in one element
      dup[k] = true;
      if (first_e[k] < 0) { first_e[k] = e; first_t[k] = t; }
      tcnt[k]++;
    }
  }
  // every cluster reproduces the template with 'nm' distinct nodes: checked on the device, which also lists the nodes of every cluster
  ClTmpl T;
  memset(&T, 0, sizeof(T));
  T.nm = nm;
  for (int e = 0; e < CL_NE; e++)
    for (int t = 0; t < 27; t++) T.tm[e][t] = (unsigned char)tm[e][t];
  for (int k = 0; k < nm; k++) {
    T.first_e[k] = (unsigned char)first_e[k];
    T.first_t[k] = (unsigned char)first_t[k];
    T.tcnt[k] = (unsigned char)tcnt[k];
  }
  for (int t = 0; t < 27; t++) {
    T.nodeof[t] = (unsigned char)nodeof[t];
    T.tof[nodeof[t]] = (unsigned char)t;
  }
  struct Tmp {
    std::vector<void*> p;
    ~Tmp() { for (void* q : p) if (q) hipFree(q); }
  } tmp;
  auto dalloc = [&](void** d, size_t bytes) -> int {
    FH_CHECK_HIP(hipMalloc(d, bytes ? bytes : 8));
    tmp.p.push_back(*d);
    return 0;
  };
  int *d_cdof = nullptr, *d_perr = nullptr;
  FH_TRY(dalloc((void**)&d_cdof, (size_t)ncl * CL_NM_MAX * sizeof(int)));
  FH_TRY(dalloc((void**)&d_perr, sizeof(int)));
  FH_CHECK_HIP(hipMemsetAsync(d_perr, 0, sizeof(int), ctx->stream));
  hipLaunchKernelGGL(k_cl_cdof, dim3(ncl), dim3(128), 0, ctx->stream, ncl, nloc, as->d_elem_dof, T, d_cdof, d_perr);
  {
    int perr = 0;
    FH_CHECK_HIP(hipMemcpyAsync(&perr, d_perr, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    FH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    if (perr) return 0;
  }
  FH_TRACE("cluster plan: clusters verified");
  // template rows: macro columns ascending; contributions in ascending element order
  std::vector<unsigned short> roff(CL_NM_MAX + 1, 0);
  std::vector<unsigned char> tcol;
  int nsingle[CL_NM_MAX] = {};
  struct U2 { unsigned x, y; };
  struct U4 { unsigned short a[8]; };
  std::vector<U2> dtab;
  U4 zblk;
  for (int q = 0; q < 8; q++) zblk.a[q] = (unsigned short)CL_ZC;
  std::vector<U4> fblk(CL_NM_MAX, zblk), oblk;
  auto stag = [&](int e, int tr, int tc) { return (unsigned)(CL_TAB + e * SF_WAVE + SF_XT + tr * KS + tc); };      // double index into the workgroup's LDS
  auto pack = [&](const std::vector<unsigned>& ad, U2* out) -> bool {
    if (ad.empty() || ad.size() > 8) return false;
    out->x = ad[0] * 8;
    out->y = (ad.size() > 1 ? ad[1] : (unsigned)CL_ZC) * 8;
    if (ad.size() > 2) {          // summed ahead of the slots, all contributions in ascending element order, into its own cell
      if ((int)oblk.size() >= CL_NBLK) return false;
      U4 b = zblk;
      for (size_t q = 0; q < ad.size(); q++) b.a[q] = (unsigned short)ad[q];
      out->x = (unsigned)(CL_LV + (int)oblk.size()) * 8;
      out->y = (unsigned)CL_ZC * 8;
      oblk.push_back(b);
    }
    return true;
  };
  int tof_e[CL_NE][CL_NM_MAX];          // tensor index of macro node k in element e, -1 if absent
  for (int e = 0; e < CL_NE; e++) {
    for (int k = 0; k < nm; k++) tof_e[e][k] = -1;
    for (int t = 0; t < 27; t++) tof_e[e][tm[e][t]] = t;
  }
  for (int r = 0; r < nm; r++) {
    bool has[CL_NM_MAX] = {};
    for (int e = 0; e < CL_NE; e++)
      if (tof_e[e][r] >= 0)
        for (int t = 0; t < 27; t++) has[tm[e][t]] = true;
    // entries met by ONE element first (83 %: the kernel skips the second operand for a whole wave of them), then the shared ones
    for (int pass = 0; pass < 2; pass++) {
      for (int k = 0; k < nm; k++)
        if (has[k]) {
          std::vector<unsigned> ad;
          for (int e = 0; e < CL_NE; e++)
            if (tof_e[e][r] >= 0 && tof_e[e][k] >= 0) ad.push_back(stag(e, tof_e[e][r], tof_e[e][k]));
          if ((ad.size() > 1) != (pass == 1)) continue;
          U2 d;
          if (!pack(ad, &d)) return 0;
          dtab.push_back(d);
          tcol.push_back((unsigned char)k);
        }
      if (pass == 0) nsingle[r] = (int)dtab.size() - roff[r];
    }
    roff[r + 1] = (unsigned short)dtab.size();
    if (roff[r + 1] - roff[r] > 127) return 0;
    int nf = 0;
    for (int e = 0; e < CL_NE; e++)
      if (tof_e[e][r] >= 0) fblk[r].a[nf++] = (unsigned short)stag(e, tof_e[e][r], 27);
  }
  for (int r = nm; r < CL_NM_MAX; r++) roff[r + 1] = roff[nm];
  const int ns = (int)dtab.size();
  if (ns > CL_NS_MAX || ns >= (1 << 13)) return 0;
  oblk.resize(CL_NBLK, zblk);
  // slots (what thread tid does in its i-th step: slot tid + CL_T * i): the single-element ranges of all rows, then the shared ranges; a
  // run of consecutive slots is a run of consecutive destinations (coalesced stores)
  std::vector<unsigned> sinfo((size_t)CL_SPT * CL_T, (unsigned)(CL_NM_MAX - 1));       // default = dummy row: destination = sink
  {
    int sidx = 0;
    for (int pass = 0; pass < 2; pass++)
      for (int r = 0; r < nm; r++) {
        const int p0 = pass ? nsingle[r] : 0, p1 = pass ? roff[r + 1] - roff[r] : nsingle[r];
        for (int pp = p0; pp < p1; pp++) sinfo[sidx++] = (unsigned)r | ((unsigned)pp << 7) | ((unsigned)roff[r] << 14);
      }
  }
  FH_TRACE("cluster plan: template");
  // destinations: a macro row is complete when all elements around its node lie in this cluster and the CSR row has exactly its columns.
  // The partial rows of one CSR row follow each other in the partial-row buffer, ascending cluster (= element) order.  Device: completeness and
  // segment sizes per CSR row; host: the scan over the rows (list of the second pass, segment starts); device: the destination of every macro row.
  for (int r = 0; r <= CL_NM_MAX; r++) T.roff[r] = roff[r];
  unsigned char* d_complete = nullptr;
  unsigned *d_rowsz = nullptr, *d_rowbase = nullptr;
  FH_TRY(dalloc((void**)&d_complete, (size_t)m));
  FH_TRY(dalloc((void**)&d_rowsz, (size_t)m * sizeof(unsigned)));
  FH_TRY(dalloc((void**)&d_rowbase, (size_t)m * sizeof(unsigned)));
  FH_CHECK_HIP(hipMemsetAsync(d_complete, 0, (size_t)m, ctx->stream));
  FH_CHECK_HIP(hipMemsetAsync(d_rowsz, 0, (size_t)m * sizeof(unsigned), ctx->stream));
  const unsigned gq = (unsigned)(((size_t)ncl * CL_NM_MAX + 255) / 256);
  // super-clusters (assemble_carry): 8^k consecutive clusters = the descendants of one ancestor k levels up in a refined mesh.  Rows all of whose elements lie in
  // one super-cluster never visit the partial-row buffer.  Automatic choice: the largest k <= 2 that still gives every workgroup of the persistent grid two
  // super-clusters (a workgroup walks a super-cluster alone, in ascending cluster order).  Nothing about the topology is assumed: the classes come from the
  // adjacency lists; a mesh whose consecutive clusters are not neighbours simply has few carried rows.
  int sup_shift = 0;
  {
    const int want = ctx->assemble_carry;
    const int wgs = std::max(1, ctx->num_cu * ctx->assemble_sf_grid);
    if (want < 0) {
      for (int sh : {6, 3})
        if (sup_shift == 0 && ncl % (1 << sh) == 0 && (ncl >> sh) >= 2 * wgs) sup_shift = sh;
    } else if (want > 0) {
      sup_shift = std::min(want % 100, 12);
      while (sup_shift > 0 && ncl % (1 << sup_shift)) sup_shift--;
    }
    as->cl_walk_shift = sup_shift;
    if (want >= 100) sup_shift = 0;               // measurement aid: the walk order of the carried plan, nothing carried
    if (m >= (1 << 30) || !allow_carry) sup_shift = 0;          // bit 30 of a residual destination marks "add"
  }
  hipLaunchKernelGGL(k_cl_rows, dim3(gq), dim3(256), 0, ctx->stream, ncl, m, d_cdof, as->d_adj_ptr, as->d_adj_ei, A->d_rowptr, T, sup_shift, d_complete, d_rowsz);
  std::vector<unsigned char> complete(m, 0);
  std::vector<unsigned> rowsz(m, 0);
  if (m) {
    FH_CHECK_HIP(hipMemcpyAsync(complete.data(), d_complete, (size_t)m, hipMemcpyDeviceToHost, ctx->stream));
    FH_CHECK_HIP(hipMemcpyAsync(rowsz.data(), d_rowsz, (size_t)m * sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
  }
  FH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  std::vector<int> prow;
  std::vector<unsigned> pstart(1, 0), rowbase(m, 0);
  size_t npart = 0;
  for (int g = 0; g < m; g++)
    if (!complete[g]) {
      if (rowsz[g] > 32 * RP_Q) return 0;                // the second pass holds a segment of <= 224 entries in registers
      rowbase[g] = (unsigned)npart;
      prow.push_back(g);
      npart += rowsz[g];
      if (npart + 128 >= ((size_t)1 << 31)) return 0;
      pstart.push_back((unsigned)npart);
    }
  const unsigned sink = 0x80000000u | (unsigned)npart;
  if (m) FH_CHECK_HIP(hipMemcpyAsync(d_rowbase, rowbase.data(), (size_t)m * sizeof(unsigned), hipMemcpyHostToDevice, ctx->stream));
  FH_CHECK_HIP(hipMalloc(&as->d_cl_vdst, (size_t)ncl * CL_NM_MAX * sizeof(int)));
  FH_CHECK_HIP(hipMalloc(&as->d_cl_fdst, (size_t)ncl * CL_NM_MAX * sizeof(int)));
  hipLaunchKernelGGL(k_cl_dst, dim3(gq), dim3(256), 0, ctx->stream, ncl, m, d_cdof, as->d_adj_ptr, as->d_adj_ei, A->d_rowptr, T, d_complete, d_rowbase, sink,
                     as->d_cl_vdst, as->d_cl_fdst);
  FH_CHECK_HIP(hipGetLastError());
  FH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  FH_TRACE("cluster plan: destinations");
  auto up = [&](void** d, const void* h, size_t bytes) -> int {
    FH_CHECK_HIP(hipMalloc(d, bytes ? bytes : 8));
    if (bytes) FH_CHECK_HIP(hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice));
    return 0;
  };
  FH_TRY(up((void**)&as->d_cl_dtab, dtab.data(), dtab.size() * sizeof(U2)));
  FH_TRY(up((void**)&as->d_cl_fblk, fblk.data(), fblk.size() * sizeof(U4)));
  FH_TRY(up((void**)&as->d_cl_oblk, oblk.data(), oblk.size() * sizeof(U4)));
  {   // for the Galerkin product from the macro rows: which template entry child e owns at (local row, local column) -- an entry met by several elements of
      // the cluster belongs to the first of them, so that the eight pseudo child matrices add up to the macro matrix
    std::vector<unsigned short> gtab((size_t)CL_NE * 27 * 27, (unsigned short)0xffff);
    for (int e = 0; e < CL_NE; e++)
      for (int il = 0; il < 27; il++)
        for (int ic = 0; ic < 27; ic++) {
          const int r = tm[e][T.tof[il]], k = tm[e][T.tof[ic]];
          int owner = -1;
          for (int e2 = 0; e2 < CL_NE && owner < 0; e2++)
            if (tof_e[e2][r] >= 0 && tof_e[e2][k] >= 0) owner = e2;
          if (owner != e) continue;
          for (int idx = roff[r]; idx < roff[r + 1]; idx++)
            if (tcol[idx] == k) gtab[((size_t)e * 27 + il) * 27 + ic] = (unsigned short)idx;
        }
    FH_TRY(up((void**)&as->d_cl_gtab, gtab.data(), gtab.size() * sizeof(unsigned short)));
    as->cl_all_rows = (m == as->nnode);
  }
  FH_CHECK_HIP(hipMalloc(&as->d_cl_vdst64, (size_t)ncl * CL_NM_MAX * sizeof(unsigned long long)));
  FH_TRY(up((void**)&as->d_cl_sinfo, sinfo.data(), sinfo.size() * 4));
  FH_TRY(up((void**)&as->d_cl_prow, prow.data(), prow.size() * 4));
  FH_TRY(up((void**)&as->d_cl_pstart, pstart.data(), pstart.size() * 4));
  FH_CHECK_HIP(hipMalloc(&as->d_cl_map, (size_t)ncl * CL_T * 16));
  if (sup_shift > 0 || ctx->assemble_carry >= 200) FH_CHECK_HIP(hipMalloc(&as->d_cl_mapb, (size_t)ncl * CL_T * 16));
  FH_CHECK_HIP(hipMalloc(&as->d_cl_pmap, npart + 128));
  FH_CHECK_HIP(hipMemset(as->d_cl_pmap, 0xFF, npart + 128));      // 255 = residual entry (the map kernel fills in the matrix entries)
  FH_CHECK_HIP(hipMalloc(&as->d_Pbuf, (npart + 128) * sizeof(double)));
  FH_CHECK_HIP(hipMemset(as->d_Pbuf, ctx->debug_poison ? 0xFF : 0, (npart + 128) * sizeof(double)));
  FH_TRACE("cluster plan: uploads and buffers");
  {
    int* d_err = nullptr;
    unsigned char* d_tcol = nullptr;
    unsigned short* d_roff = nullptr;
    unsigned* d_first = nullptr;
    unsigned long long* d_ncar = nullptr;
    FH_TRY(up((void**)&d_tcol, tcol.data(), tcol.size()));
    FH_TRY(up((void**)&d_roff, roff.data(), roff.size() * 2));
    const int zero = 0;
    FH_TRY(up((void**)&d_err, &zero, sizeof(int)));
    FH_TRY(dalloc((void**)&d_first, (size_t)m * sizeof(unsigned)));
    FH_TRY(dalloc((void**)&d_ncar, sizeof(unsigned long long)));
    FH_CHECK_HIP(hipMemsetAsync(d_first, 0, (size_t)m * sizeof(unsigned), ctx->stream));
    FH_CHECK_HIP(hipMemsetAsync(d_ncar, 0, sizeof(unsigned long long), ctx->stream));
    hipLaunchKernelGGL(k_cluster_maps, dim3(ncl), dim3(256), 0, ctx->stream, ncl, ns, nm, d_cdof, as->d_cl_sinfo, d_roff, d_tcol, as->d_cl_vdst, m, A->d_rowptr, A->d_col, d_complete,
                       as->d_adj_ptr, as->d_adj_ei, as->d_elem_dof, nloc, reinterpret_cast<uint4*>(as->d_cl_map), reinterpret_cast<uint4*>(as->d_cl_mapb), as->d_cl_pmap, d_first, d_err);
    if (sup_shift > 0 && m > 0)
      hipLaunchKernelGGL(k_cl_cover, dim3(fh_div_up(m, 256)), dim3(256), 0, ctx->stream, m, A->d_rowptr, d_complete, d_first, d_err, d_ncar);
    FH_CHECK_HIP(hipGetLastError());
    int err = 0;
    unsigned long long ncar = 0;
    FH_CHECK_HIP(hipMemcpyAsync(&err, d_err, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    FH_CHECK_HIP(hipMemcpyAsync(&ncar, d_ncar, sizeof(ncar), hipMemcpyDeviceToHost, ctx->stream));
    FH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    as->cl_ncarried = (size_t)ncar;
    for (void* q : {(void*)d_tcol, (void*)d_roff, (void*)d_err}) hipFree(q);
    if (err && sup_shift > 0) {
      // (a pattern with positions no element contributes to: a carried row would keep stale values there) -- the plan without carried rows
      FH_TRACE("fh_assembler_create: carried rows do not cover their CSR rows -- cluster plan without them");
      for (void** q : {(void**)&as->d_cl_dtab, (void**)&as->d_cl_fblk, (void**)&as->d_cl_sinfo, (void**)&as->d_cl_oblk, (void**)&as->d_cl_gtab, (void**)&as->d_cl_vdst, (void**)&as->d_cl_vdst64,
                       (void**)&as->d_cl_fdst, (void**)&as->d_cl_map, (void**)&as->d_cl_mapb, (void**)&as->d_cl_pmap, (void**)&as->d_Pbuf, (void**)&as->d_cl_prow, (void**)&as->d_cl_pstart}) {
        if (*q) FH_CHECK_HIP(hipFree(*q));
        *q = nullptr;
      }
      return cluster_plan_build(as, A, elem_dof, aptr, false);
    }
    if (err) {
      FH_TRACE("fh_assembler_create: cluster maps could not be placed in the matrix pattern -- two-pass assembly kept");
      return 0;
    }
  }
  as->cl_ncl = ncl;
  as->cl_nm = nm;
  as->cl_ns = ns;
  as->cl_npart = npart;
  as->cl_nprow = (int)prow.size();
  as->cl_sup_shift = sup_shift;
  if (sup_shift > 0 || ctx->assemble_carry < 100) as->cl_walk_shift = sup_shift;
  as->fused = true;
  FH_TRACE("fh_assembler_create: cluster plan (%d clusters, %d macro nodes, %d entries; super-clusters of %d: %zu carried entries; %zu partial entries, %d rows in the second pass)", ncl,
           nm, ns, 1 << sup_shift, as->cl_ncarried, npart, as->cl_nprow);
  return 0;
}

// Instrumented launch (asm_debug bit 7, constant source only): the same kernel with the shader clock read at twelve phase boundaries; prints the
// average cycles per cluster and phase over all waves.  A development aid -- the stamps cost about a tenth of the wave cycles themselves.
template <bool CARRY>
static int launch_cluster_stamped(fh_assembler_t as, const AsmParams& P, const ClParams& C0, int grid) {
  constexpr size_t lds = cl_lds_bytes();
  static bool attr_set[64] = {};
  const int dev = as->ctx->device & 63;
  if (!attr_set[dev]) {
    FH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cluster_q2hex_sf<0, true, true, CARRY>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set[dev] = true;
  }
  ClParams C = C0;
  const size_t nst = (size_t)grid * CL_NE * 20;
  FH_CHECK_HIP(hipMalloc(&C.stamps, nst * sizeof(unsigned long long)));
  FH_CHECK_HIP(hipMemsetAsync(C.stamps, 0, nst * sizeof(unsigned long long), as->ctx->stream));
  hipLaunchKernelGGL((k_cluster_q2hex_sf<0, true, true, CARRY>), dim3(grid), dim3(CL_T), lds, as->ctx->stream, P, as->sf_tab, as->d_sfLc, as->d_sfLi, C);
  FH_CHECK_HIP(hipGetLastError());
  std::vector<unsigned long long> h(nst);
  FH_CHECK_HIP(hipMemcpyAsync(h.data(), C.stamps, nst * sizeof(unsigned long long), hipMemcpyDeviceToHost, as->ctx->stream));
  FH_CHECK_HIP(hipStreamSynchronize(as->ctx->stream));
  FH_CHECK_HIP(hipFree(C.stamps));
  // order of the rows = order of the phases in the loop body of a wave that runs phase A in place (waves 0 .. 3)
  static const int order[16] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 14, 13, 15, 10, 11};
  static const char* name[16] = {"loop top (previous barrier exit -> stamp)", "phase A: U (gather x, first contraction, sync)", "phase A: V (second contraction, sync)",
                                 "phase A: J, cofactors, division, D_q", "stage 1 (15 MFMA, writes, sync)", "source 2nd contraction + stages 2-3 (4 x 16 reads + 66 FMA)",
                                 "source 3rd contraction, sync", "destination loads issued, staging writes (18), sync", "residual K_e u, next nodes, destinations",
                                 "workgroup barrier 1 (waiting for the slowest wave)", "(end of the output phase)", "workgroup barrier 2",
                                 "output: descriptors + destinations read, residual entries and long sums", "output: both operands read, added", "output: barrier behind the long sums", "output: stores issued"};
  for (int half = 0; half < 2; half++) {
    double sum[16] = {0}, ncl = 0, tot = 0;
    for (size_t w = 0; w < (size_t)grid * CL_NE; w++) {
      if ((int)((w % CL_NE) / (CL_NE / 2)) != half) continue;
      for (int k = 0; k < 16; k++) sum[k] += (double)h[w * 20 + k];
      ncl += (double)h[w * 20 + 16];
    }
    for (int k = 0; k < 16; k++) tot += sum[k];
    fprintf(stderr, "k_cluster_q2hex_sf<0> phase stamps, waves %d..%d (%s): %d workgroups, %.0f wave-clusters, %.0f shader-clock ticks per cluster and wave\n", half * 4, half * 4 + 3,
            half && !(P.debug & 256) ? "phase A one cluster ahead, i.e. rows 1-3 run behind row 8" : "phase A in place", grid, ncl, tot / std::max(ncl, 1.0));
    for (int kk = 0; kk < 16; kk++) {
      const int k = order[kk];
      fprintf(stderr, "  %2d %-62s %9.1f  %5.1f %%\n", k, name[k], sum[k] / std::max(ncl, 1.0), 100.0 * sum[k] / std::max(tot, 1.0));
    }
  }
  return 0;
}

template <int SRC, bool CARRY>
static int launch_cluster_one(fh_assembler_t as, const AsmParams& P, const ClParams& C) {
  constexpr size_t lds = cl_lds_bytes();
  static bool attr_set[64] = {};
  const int dev = as->ctx->device & 63;
  if (!attr_set[dev]) {
    FH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cluster_q2hex_sf<SRC, false, true, CARRY>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set[dev] = true;
  }
  const int grid = std::max(1, std::min(C.ncl >> (CARRY ? C.sup_shift : 0), as->ctx->num_cu * as->ctx->assemble_sf_grid));
  if (SRC == 0 && (as->ctx->asm_debug & 128)) return launch_cluster_stamped<CARRY>(as, P, C, grid);     // dev aid: phase cycles of every wave on stderr
  hipLaunchKernelGGL((k_cluster_q2hex_sf<SRC, false, true, CARRY>), dim3(grid), dim3(CL_T), lds, as->ctx->stream, P, as->sf_tab, as->d_sfLc, as->d_sfLi, C);
  FH_CHECK_HIP(hipGetLastError());
  return 0;
}

// row destinations as addresses (the cluster kernel's output phase is bound by vector-instruction issue: no address arithmetic there)
__global__ void k_cluster_vdst(size_t n, const int* __restrict__ vdst, double* Pbuf, double* val, unsigned long long* __restrict__ out) {
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int v = vdst[k];
  out[k] = reinterpret_cast<unsigned long long>(v < 0 ? Pbuf + (size_t)(v & 0x7fffffff) : val + (size_t)v);
}

static int launch_cluster(fh_assembler_t as, const AsmParams& P, fh_mat_t A, double* res) {
  if (as->cl_val_base != A->d_val) {        // first assembly, or another matrix of the same pattern
    const size_t n = (size_t)as->cl_ncl * CL_NM_MAX;
    hipLaunchKernelGGL(k_cluster_vdst, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as->ctx->stream, n, as->d_cl_vdst, as->d_Pbuf, A->d_val, as->d_cl_vdst64);
    FH_CHECK_HIP(hipGetLastError());
    as->cl_val_base = A->d_val;
  }
  ClParams C;
  C.ncl = as->cl_ncl; C.ns = as->cl_ns; C.nm = as->cl_nm;
  C.dtab = reinterpret_cast<const uint2*>(as->d_cl_dtab); C.fblk = reinterpret_cast<const uint4*>(as->d_cl_fblk); C.oblk = reinterpret_cast<const uint4*>(as->d_cl_oblk);
  C.sinfo = as->d_cl_sinfo;
  C.vdst = as->d_cl_vdst64; C.fdst = as->d_cl_fdst; C.map = reinterpret_cast<const uint4*>(as->d_cl_map); C.mapb = reinterpret_cast<const uint4*>(as->d_cl_mapb);
  C.Pbuf = as->d_Pbuf; C.res = res;
  C.stamps = nullptr;
  C.sup_shift = as->cl_walk_shift;
  if (as->cl_sup_shift > 0 || (as->ctx->assemble_carry >= 200 && as->d_cl_mapb)) {      // (>= 200: measurement aid, the CARRY kernel on a plan without carried rows)
    if (P.source_kind == 4) FH_TRY((launch_cluster_one<2, true>(as, P, C)));
    else if (P.source_kind != 0) FH_TRY((launch_cluster_one<1, true>(as, P, C)));
    else FH_TRY((launch_cluster_one<0, true>(as, P, C)));
  } else if (P.source_kind == 4) FH_TRY((launch_cluster_one<2, false>(as, P, C)));
  else if (P.source_kind != 0) FH_TRY((launch_cluster_one<1, false>(as, P, C)));
  else FH_TRY((launch_cluster_one<0, false>(as, P, C)));
  if (as->cl_nprow > 0 && !(as->ctx->asm_debug & (2 | 8))) {
    // plain loads of the partial rows: a segment starts at any multiple of 8 bytes, so neighbouring segments share cache lines -- non-temporal
    // loads fetched those twice (measured 1.36 -> 1.33 ms per assembly; asm_debug bit 5 selects them for comparison)
    if (!(as->ctx->asm_debug & 32))
      hipLaunchKernelGGL(k_rows_partial<0>, dim3(fh_div_up(as->cl_nprow, 16)), dim3(256), 0, as->ctx->stream, as->cl_nprow, as->d_cl_prow, as->d_cl_pstart,
                         as->d_cl_pmap, as->d_Pbuf, A->d_rowptr, A->d_val, res, 0);
    else
      hipLaunchKernelGGL(k_rows_partial<1>, dim3(fh_div_up(as->cl_nprow, 16)), dim3(256), 0, as->ctx->stream, as->cl_nprow, as->d_cl_prow, as->d_cl_pstart,
                         as->d_cl_pmap, as->d_Pbuf, A->d_rowptr, A->d_val, res, 1);
    FH_CHECK_HIP(hipGetLastError());
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// Affine HEX27 / Q2 elements (parallelepipeds: boxes, sheared boxes): the Jacobian of the map is constant, so
//   K_ij = sum_g w_g det grad phi_i . grad phi_j = sum_ab (det B_ab) M_ab(i,j),  B = J^-1 J^-T,  M_ab = sum_g w^_g d_a phi^_i d_b phi^_j
// with the nine reference matrices M_ab built once from the same quadrature tables.  The result equals the quadrature loop of
// the reference up to the order of the floating-point sums (checked against the oracle at 1e-12).  Opt-in (assemble_affine):
// the default assembles every element by quadrature as the reference does.  16 elements per workgroup; every thread keeps the
// M_ab values of its <= 3 matrix entries in registers and applies them to the 16 coefficient sets.
// ------------------------------------------------------------------------------------------------------------------
template <int SRC>
__global__ __launch_bounds__(256) void k_elem_q2hex_affine(AsmParams P, const double* __restrict__ Mab, const double* __restrict__ mphi) {
  constexpr int NC = 27, DIM = 3, EB = 16, NE = NC * NC;
  __shared__ double xs[EB][NC * DIM], us[EB][NC], Jl[EB][9], Cs[EB][9], detw[EB], Ks[NE];
  __shared__ double fgp[(SRC != 0) ? EB : 1][(SRC != 0) ? 64 : 1];
  __shared__ int sl[EB][NC];
  const int tid = threadIdx.x;
  const int e0 = blockIdx.x * EB;
  const int ne = min(EB, P.nelems - e0);
  for (int idx = tid; idx < ne * NC; idx += 256) {
    const int le = idx / NC, n = idx % NC;
    const int e = P.elems[e0 + le];
    const int dof = P.elem_dof[(size_t)e * P.nloc + n];
#pragma unroll
    for (int d = 0; d < DIM; d++) xs[le][n * DIM + d] = P.coords[(size_t)dof * DIM + d];
    us[le][n] = P.sol ? P.sol[dof] : 0.0;
    sl[le][n] = P.slot ? P.slot[(size_t)e * NC + n] : (e0 + le) * NC + n;
  }
  __syncthreads();
  {
    const int le = tid >> 4, ab = tid & 15;
    if (le < ne && ab < 9) {
      const int a = ab / 3, b = ab % 3;
      double s = 0.0;
      for (int n = 0; n < NC; n++) s += P.dphi[(size_t)n * DIM + a] * xs[le][n * DIM + b];   // Gauss point 0
      Jl[le][ab] = s;
    }
  }
  __syncthreads();
  if (tid < ne) {
    const double* J = Jl[tid];
    const double det = J[0] * (J[4] * J[8] - J[5] * J[7]) + J[1] * (J[5] * J[6] - J[3] * J[8]) + J[2] * (J[3] * J[7] - J[4] * J[6]);
    const double rd = 1.0 / det;
    double JI[3][3];
    JI[0][0] = (-J[5] * J[7] + J[4] * J[8]) * rd;
    JI[0][1] = (J[2] * J[7] - J[1] * J[8]) * rd;
    JI[0][2] = (-J[2] * J[4] + J[1] * J[5]) * rd;
    JI[1][0] = (J[5] * J[6] - J[3] * J[8]) * rd;
    JI[1][1] = (-J[2] * J[6] + J[0] * J[8]) * rd;
    JI[1][2] = (J[2] * J[3] - J[0] * J[5]) * rd;
    JI[2][0] = (-J[4] * J[6] + J[3] * J[7]) * rd;
    JI[2][1] = (J[1] * J[6] - J[0] * J[7]) * rd;
    JI[2][2] = (-J[1] * J[3] + J[0] * J[4]) * rd;
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) Cs[tid][a * 3 + b] = det * (JI[0][a] * JI[0][b] + JI[1][a] * JI[1][b] + JI[2][a] * JI[2][b]);
    detw[tid] = det;
  }
  if (SRC != 0) {
    for (int idx = tid; idx < ne * 64; idx += 256) {
      const int le = idx >> 6, g = idx & 63;
      double xg[4] = {0.0, 0.0, 0.0, 0.0};
      for (int n = 0; n < NC; n++) {
        const double ph = P.phi[(size_t)g * NC + n];
#pragma unroll
        for (int d = 0; d < DIM; d++) xg[d] += xs[le][n * DIM + d] * ph;
      }
      const double f = (SRC == 1) ? source_eval(P.source_kind, P.p0, P.p1, xg, DIM) : P.p0 * fh_expr_device_eval(P.prog, P.nprog, P.prog_consts, xg);
      fgp[le][g] = f * P.w[g];
    }
  }
  // M_ab of this thread's entries
  double Mr[3][9];
  int ent[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    ent[k] = tid + k * 256;
#pragma unroll
    for (int ab = 0; ab < 9; ab++) Mr[k][ab] = (ent[k] < NE) ? Mab[(size_t)ab * NE + ent[k]] : 0.0;
  }
  __syncthreads();
  for (int le = 0; le < ne; le++) {
    double c[9];
#pragma unroll
    for (int ab = 0; ab < 9; ab++) c[ab] = Cs[le][ab];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      if (ent[k] >= NE) continue;
      double v = 0.0;
#pragma unroll
      for (int ab = 0; ab < 9; ab++) v += c[ab] * Mr[k][ab];
      Ks[ent[k]] = v;
      const int i = ent[k] / NC, j = ent[k] - i * NC;
      const int s = sl[le][i];
      if (s >= 0) P.Kout[(size_t)s * P.kstride + j] = v;
    }
    __syncthreads();
    if (tid < NC) {
      const int i = tid;
      double ku = 0.0;
      for (int j = 0; j < NC; j++) ku += Ks[i * NC + j] * us[le][j];
      double fs;
      if (SRC == 0) fs = P.p0 * mphi[i];
      else {
        fs = 0.0;
        for (int g = 0; g < 64; g++) fs += fgp[le][g] * P.phi[(size_t)g * NC + i];
      }
      const int s = sl[le][i];
      if (s >= 0) P.Fout[s] = -fs * detw[le] - ku;
    }
    __syncthreads();
  }
}

template <int DIM, int NC>
static int launch_assemble(fh_assembler_t as, const AsmParams& P) {
  using C = AsmCfg<DIM, NC>;
  if (P.nelems <= 0) return 0;
  const dim3 grid(fh_div_up(P.nelems, C::EPB)), block(256);
  const size_t lds = (size_t)C::EPB * C::LDS_PER_ELEM * sizeof(double);
  hipStream_t st = as->ctx->stream;
  const int src = P.source_kind == 0 ? 0 : P.source_kind == 4 ? 2 : 1;
  const int outm = P.Kout ? 3 : P.emap_out ? 2 : P.emap ? 0 : 1;
#define FH_ASM(SRC, OUT) hipLaunchKernelGGL((k_assemble_poisson<DIM, NC, SRC, OUT>), grid, block, lds, st, P)
#define FH_ASM_SRC(OUT) do { if (src == 2) FH_ASM(2, OUT); else if (src == 1) FH_ASM(1, OUT); else FH_ASM(0, OUT); } while (0)
  if (outm == 2) FH_ASM(0, 2);
  else if (outm == 3) FH_ASM_SRC(3);
  else if (outm == 0) FH_ASM_SRC(0);
  else FH_ASM_SRC(1);
#undef FH_ASM_SRC
#undef FH_ASM
  FH_CHECK_HIP(hipGetLastError());
  return 0;
}

static int dispatch_assemble(fh_assembler_t as, const AsmParams& P) {
  if (as->dim == 3 && as->nc == 27 && P.Kout && as->ctx->assemble_sf && as->d_sfLc && P.ng == 64) {
    if (P.nelems <= 0) return 0;
    return launch_sf(as, P, as->ctx->assemble_sf);
  }
  if (as->dim == 3 && as->nc == 27 && P.Kout && as->ctx->assemble_mfma && as->d_mfT && P.ng == 64) {
    if (P.nelems <= 0) return 0;
    return launch_mfma(as, P, as->ctx->assemble_mfma);
  }
  if (as->dim == 3 && as->nc == 27 && P.Kout && as->ctx->assemble_sym) {
    if (P.nelems <= 0) return 0;
    const dim3 grid(fh_div_up(P.nelems, 2)), block(64);
    if (P.source_kind == 4) hipLaunchKernelGGL(k_elem_q2hex_sym<2>, grid, block, 0, as->ctx->stream, P);
    else if (P.source_kind != 0) hipLaunchKernelGGL(k_elem_q2hex_sym<1>, grid, block, 0, as->ctx->stream, P);
    else hipLaunchKernelGGL(k_elem_q2hex_sym<0>, grid, block, 0, as->ctx->stream, P);
    FH_CHECK_HIP(hipGetLastError());
    return 0;
  }
  if (as->dim == 3 && as->nc == 27) return launch_assemble<3, 27>(as, P);
  if (as->dim == 3 && as->nc == 20) return launch_assemble<3, 20>(as, P);      // serendipity (HexQuadratic / QuadQuadratic): the generic tile kernel
  if (as->dim == 2 && as->nc == 8) return launch_assemble<2, 8>(as, P);
  if (as->dim == 3 && as->nc == 8) return launch_assemble<3, 8>(as, P);
  if (as->dim == 2 && as->nc == 9) return launch_assemble<2, 9>(as, P);
  if (as->dim == 2 && as->nc == 4) return launch_assemble<2, 4>(as, P);
  fh_set_error("assembler: unsupported element (dim %d, nc %d)", as->dim, as->nc);
  return 2;
}

// greedy element colouring: elements sharing a node get different colours
static void color_elements(int nel, int nc, int nloc, const int* elem_dof, int ndof, std::vector<int>& color_ptr,
                           std::vector<int>& color_elems) {
  std::vector<int> cnt(ndof + 1, 0);
  for (int e = 0; e < nel; e++)
    for (int l = 0; l < nc; l++) cnt[elem_dof[(size_t)e * nloc + l] + 1]++;
  for (int i = 0; i < ndof; i++) cnt[i + 1] += cnt[i];
  std::vector<int> adj(cnt[ndof]), cur(cnt.begin(), cnt.end() - 1);
  for (int e = 0; e < nel; e++)
    for (int l = 0; l < nc; l++) adj[cur[elem_dof[(size_t)e * nloc + l]]++] = e;
  std::vector<int> color(nel, -1);
  std::vector<unsigned long long> used;
  int ncolors = 0;
  for (int e = 0; e < nel; e++) {
    unsigned long long mask = 0;   // up to 64 colours (hex meshes need 8, unstructured ones ~20-30)
    for (int l = 0; l < nc; l++) {
      const int d = elem_dof[(size_t)e * nloc + l];
      for (int k = cnt[d]; k < cnt[d + 1]; k++) {
        const int c = color[adj[k]];
        if (c >= 0) mask |= 1ull << c;
      }
    }
    int c = 0;
    while (c < 63 && (mask >> c) & 1ull) c++;
    color[e] = c;
    ncolors = std::max(ncolors, c + 1);
  }
  color_ptr.assign(ncolors + 1, 0);
  for (int e = 0; e < nel; e++) color_ptr[color[e] + 1]++;
  for (int c = 0; c < ncolors; c++) color_ptr[c + 1] += color_ptr[c];
  color_elems.resize(nel);
  std::vector<int> pos(color_ptr.begin(), color_ptr.end() - 1);
  for (int e = 0; e < nel; e++) color_elems[pos[color[e]]++] = e;
}

static AsmParams base_params(fh_assembler_t as) {
  AsmParams P;
  memset(&P, 0, sizeof(P));
  P.elem_dof = as->d_elem_dof;
  P.nloc = as->nloc;
  P.coords = as->d_coords;
  P.w = as->d_w;
  P.phi = as->d_phi;
  P.dphi = as->d_dphi;
  P.ng = as->ng;
  P.prog = as->d_prog;
  P.prog_consts = as->d_prog_consts;
  P.nprog = as->nprog;
  P.kstride = as->nc;
  return P;
}

// element colours for the coloured scatter (greedy; host, from the device copy of the connectivity): made at the first use
static int ensure_colors(fh_assembler_t as) {
  if (as->d_color_elems) return 0;
  std::vector<int> ed((size_t)as->nel * as->nloc), celems;
  if (!ed.empty()) FH_CHECK_HIP(hipMemcpy(ed.data(), as->d_elem_dof, ed.size() * sizeof(int), hipMemcpyDeviceToHost));
  color_elements(as->nel, as->nc, as->nloc, ed.data(), as->nnode, as->color_ptr, celems);
  as->ncolors = (int)as->color_ptr.size() - 1;
  FH_CHECK_HIP(hipMalloc(&as->d_color_elems, std::max<size_t>(celems.size(), 1) * sizeof(int)));
  if (!celems.empty()) FH_CHECK_HIP(hipMemcpy(as->d_color_elems, celems.data(), celems.size() * sizeof(int), hipMemcpyHostToDevice));
  FH_TRACE("assembler: element colouring done (%d colours)", as->ncolors);
  return 0;
}

// affine classification + reference matrices of the optional affine path (option assemble_affine, fh_assembler_affine_count): at the first use
static int ensure_affine(fh_assembler_t as) {
  if (as->d_Mab || !(as->two_pass && as->dim == 3 && as->nc == 27 && as->ng == 64)) return 0;
  const int nel = as->nel, nloc = as->nloc, geom = as->geom;
  std::vector<int> edv((size_t)nel * nloc);
  std::vector<double> xv((size_t)as->nnode * 3), w, phi, dphi;
  if (!edv.empty()) FH_CHECK_HIP(hipMemcpy(edv.data(), as->d_elem_dof, edv.size() * sizeof(int), hipMemcpyDeviceToHost));
  if (!xv.empty()) FH_CHECK_HIP(hipMemcpy(xv.data(), as->d_coords, xv.size() * sizeof(double), hipMemcpyDeviceToHost));
  FH_REQUIRE(fhfe::shape_tables(as->geom, as->fe, as->order, w, phi, dphi) == 0, "assembler: unsupported Gauss rule %d", as->order);
  const int* elem_dof = edv.data();
  const double* coords = xv.data();
  auto up = [&](void** d, const void* h, size_t bytes) -> int {
    FH_CHECK_HIP(hipMalloc(d, bytes ? bytes : 8));
    if (bytes) FH_CHECK_HIP(hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice));
    return 0;
  };
  {
    // affine classification (host, geometry is fixed for the life of the assembler): every node at c + sum_a xi_a h_a
    std::vector<int> aff, gen;
    for (int e = 0; e < nel; e++) {
      const int* ed = elem_dof + (size_t)e * nloc;
      const double* x0 = coords + (size_t)ed[0] * 3;
      double h[3][3], hmax = 0.0;
      const int vtx[3] = {1, 3, 4};
      for (int a = 0; a < 3; a++)
        for (int d = 0; d < 3; d++) {
          h[a][d] = 0.5 * (coords[(size_t)ed[vtx[a]] * 3 + d] - x0[d]);
          hmax = std::max(hmax, std::fabs(h[a][d]));
        }
      bool ok = hmax > 0.0;
      for (int n = 0; n < 27 && ok; n++)
        for (int d = 0; d < 3; d++) {
          double ref = x0[d];
          for (int a = 0; a < 3; a++) ref += (fhfe::xc(geom, n, a) + 1) * h[a][d];
          if (std::fabs(coords[(size_t)ed[n] * 3 + d] - ref) > 1e-12 * hmax) ok = false;
        }
      (ok ? aff : gen).push_back(e);
    }
    as->n_aff = (int)aff.size();
    as->n_gen = (int)gen.size();
    FH_TRY(up((void**)&as->d_aff_elems, aff.data(), aff.size() * sizeof(int)));
    FH_TRY(up((void**)&as->d_gen_elems, gen.data(), gen.size() * sizeof(int)));
    std::vector<double> Mab((size_t)9 * 729, 0.0), mphi(27, 0.0);
    for (int g = 0; g < as->ng; g++)
      for (int i = 0; i < 27; i++) {
        mphi[i] += w[g] * phi[(size_t)g * 27 + i];
        for (int j = 0; j < 27; j++)
          for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++)
              Mab[(size_t)(a * 3 + b) * 729 + i * 27 + j] += w[g] * dphi[((size_t)g * 27 + i) * 3 + a] * dphi[((size_t)g * 27 + j) * 3 + b];
      }
    FH_TRY(up((void**)&as->d_Mab, Mab.data(), Mab.size() * sizeof(double)));
    FH_TRY(up((void**)&as->d_mphi, mphi.data(), mphi.size() * sizeof(double)));
    FH_TRACE("assembler: affine classification done (%d affine, %d general)", as->n_aff, as->n_gen);
  }
  return 0;
}

// row -> (element, local row) adjacency on the device (fh_assembler_create)
__global__ __launch_bounds__(256) void k_adj_count(size_t n, int nc, int nloc, const int* __restrict__ elem_dof, int m, int* __restrict__ cnt) {
  const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  const int r = elem_dof[(k / nc) * nloc + k % nc];
  if (r < m) atomicAdd(&cnt[r], 1);
}
__global__ __launch_bounds__(256) void k_adj_fill(size_t n, int nc, int nloc, const int* __restrict__ elem_dof, int m, int* __restrict__ cur, int* __restrict__ aei) {
  const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  const int e = (int)(k / nc), i = (int)(k % nc);
  const int r = elem_dof[(size_t)e * nloc + i];
  if (r < m) aei[atomicAdd(&cur[r], 1)] = (e << 5) | i;
}
__global__ __launch_bounds__(256) void k_adj_sort(int m, int nc, const int* __restrict__ aptr, int* __restrict__ aei, int* __restrict__ slot) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= m) return;
  const int a0 = aptr[r], a1 = aptr[r + 1];
  for (int a = a0 + 1; a < a1; a++) {          // insertion sort: a row has a handful of pairs (8 on a hexahedral grid)
    const int v = aei[a];
    int b = a - 1;
    while (b >= a0 && aei[b] > v) {
      aei[b + 1] = aei[b];
      b--;
    }
    aei[b + 1] = v;
  }
  for (int a = a0; a < a1; a++) slot[(size_t)(aei[a] >> 5) * nc + (aei[a] & 31)] = a;
}

// dev_elem_dof / dev_coords: the same two arrays already in device memory (a mesh's device copy), or null
static int assembler_create_impl(fh_ctx_t ctx, int geom, int fe, int order, int nel, int nloc, const int* elem_dof, int nnode, const double* coords,
                                 const int* dev_elem_dof, const double* dev_coords, fh_mat_t A, fh_assembler_t* out) {
  FH_GUARD_BEGIN
  FH_REQUIRE(ctx && elem_dof && coords && A && out, "fh_assembler_create: null argument");
  FH_REQUIRE(geom == 0 || geom == 1, "fh_assembler_create: geom must be 0 (hex) or 1 (quad)");
  FH_REQUIRE(fe == 0 || fe == 1 || fe == 2, "fh_assembler_create: fe must be 0 (linear), 1 (serendipity) or 2 (biquadratic)");
  FH_REQUIRE(nloc == fhfe::nloc_of(geom), "fh_assembler_create: nloc %d does not match the geometry (%d)", nloc, fhfe::nloc_of(geom));
  fh_assembler_t as = new fh_assembler_s();
  as->ctx = ctx;
  as->geom = geom;
  as->fe = fe;
  as->order = order;
  as->dim = fhfe::dim_of(geom);
  as->nc = fhfe::ndofs_of(geom, fe);
  as->nloc = nloc;
  as->nel = nel;
  as->nnode = nnode;
  as->ndof = A->m;
  FH_REQUIRE(A->n >= 1 && A->m <= A->n, "fh_assembler_create: matrix must be square or owned-rows x local-columns");
  as->ncp = ((as->nc + ((as->nc == 9) ? 2 : 3)) / ((as->nc == 9) ? 3 : 4)) * ((as->nc == 9) ? 3 : 4);
  std::vector<double> w, phi, dphi;
  FH_REQUIRE(fhfe::shape_tables(geom, fe, order, w, phi, dphi) == 0, "fh_assembler_create: unsupported Gauss rule %d", order);
  as->ng = (int)w.size();
  if (!dev_elem_dof)      // (a mesh's own table holds ids of its own numbering)
    for (size_t k = 0; k < (size_t)nel * nloc; k++)
      FH_REQUIRE(elem_dof[k] >= 0 && elem_dof[k] < nnode, "fh_assembler_create: node id %d out of range", elem_dof[k]);
  auto up = [&](void** d, const void* h, size_t bytes) -> int {
    FH_CHECK_HIP(hipMalloc(d, bytes ? bytes : 8));
    if (bytes) FH_CHECK_HIP(hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice));
    return 0;
  };
  auto dup = [&](void** d, const void* src, size_t bytes) -> int {
    FH_CHECK_HIP(hipMalloc(d, bytes ? bytes : 8));
    if (bytes) FH_CHECK_HIP(hipMemcpyAsync(*d, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return 0;
  };
  if (dev_elem_dof && dev_coords) {
    FH_TRY(dup((void**)&as->d_elem_dof, dev_elem_dof, (size_t)nel * nloc * sizeof(int)));
    FH_TRY(dup((void**)&as->d_coords, dev_coords, (size_t)nnode * as->dim * sizeof(double)));
  } else {
    FH_TRY(up((void**)&as->d_elem_dof, elem_dof, (size_t)nel * nloc * sizeof(int)));
    FH_TRY(up((void**)&as->d_coords, coords, (size_t)nnode * as->dim * sizeof(double)));
  }
  FH_TRY(up((void**)&as->d_w, w.data(), w.size() * sizeof(double)));
  FH_TRY(up((void**)&as->d_phi, phi.data(), phi.size() * sizeof(double)));
  FH_TRY(up((void**)&as->d_dphi, dphi.data(), dphi.size() * sizeof(double)));
  if (as->dim == 3 && as->nc == 27 && as->ng == 64) {
    std::vector<double> mfT((size_t)64 * MF_TS, 0.0), mfPhi((size_t)64 * MF_PS, 0.0);
    for (int g = 0; g < 64; g++)
      for (int n = 0; n < 27; n++) {
        mfPhi[(size_t)g * MF_PS + n] = phi[(size_t)g * 27 + n];
        for (int a = 0; a < 3; a++) mfT[(size_t)g * MF_TS + a * MF_TA + n] = dphi[((size_t)g * 27 + n) * 3 + a];
      }
    FH_TRY(up((void**)&as->d_mfT, mfT.data(), mfT.size() * sizeof(double)));
    FH_TRY(up((void**)&as->d_mfPhi, mfPhi.data(), mfPhi.size() * sizeof(double)));
    // Sum factorisation of the map Jacobian (phase A of the kernel): HEX27 shape functions are products of 1-D quadratic
    // Lagrange polynomials and the 64-point rule is a 4 x 4 x 4 tensor grid.  Both facts are CHECKED here against the tables the
    // reference's formulas produced (1e-13); if either fails the kernel keeps the direct 27-node loop.
    {
      double xi[64][3];
      for (int g = 0; g < 64; g++)
        for (int d = 0; d < 3; d++) {
          xi[g][d] = 0.0;
          for (int n = 0; n < 27; n++) xi[g][d] += phi[(size_t)g * 27 + n] * fhfe::xc(geom, n, d);
        }
      std::vector<double> absc;
      for (int g = 0; g < 64; g++) {
        bool seen = false;
        for (double v : absc) seen |= std::fabs(v - xi[g][0]) < 1e-12;
        if (!seen) absc.push_back(xi[g][0]);
      }
      std::sort(absc.begin(), absc.end());
      bool ok = absc.size() == 4;
      int qidx[64][3], seen_q[64] = {0}, nodeof[27];
      for (int k = 0; k < 27; k++) nodeof[k] = -1;
      for (int g = 0; g < 64 && ok; g++) {
        for (int d = 0; d < 3; d++) {
          qidx[g][d] = -1;
          for (int k = 0; k < 4; k++)
            if (std::fabs(absc[k] - xi[g][d]) < 1e-12) qidx[g][d] = k;
          ok &= qidx[g][d] >= 0;
        }
        if (ok) seen_q[qidx[g][0] + 4 * qidx[g][1] + 16 * qidx[g][2]]++;
      }
      for (int k = 0; k < 64 && ok; k++) ok &= seen_q[k] == 1;
      for (int n = 0; n < 27; n++) nodeof[(fhfe::xc(geom, n, 0) + 1) * 9 + (fhfe::xc(geom, n, 1) + 1) * 3 + fhfe::xc(geom, n, 2) + 1] = n;
      for (int k = 0; k < 27; k++) ok &= nodeof[k] >= 0;
      double L1[3][4], D1[3][4];
      for (int k = 0; k < 4 && ok; k++) {
        const double x = absc[k];
        L1[0][k] = 0.5 * x * (x - 1.0); L1[1][k] = 1.0 - x * x; L1[2][k] = 0.5 * x * (x + 1.0);
        D1[0][k] = x - 0.5;             D1[1][k] = -2.0 * x;    D1[2][k] = x + 0.5;
      }
      for (int g = 0; g < 64 && ok; g++)
        for (int n = 0; n < 27; n++) {
          const int a = fhfe::xc(geom, n, 0) + 1, b = fhfe::xc(geom, n, 1) + 1, c = fhfe::xc(geom, n, 2) + 1;
          const int q1 = qidx[g][0], q2 = qidx[g][1], q3 = qidx[g][2];
          ok &= std::fabs(phi[(size_t)g * 27 + n] - L1[a][q1] * L1[b][q2] * L1[c][q3]) <= 1e-13;
          ok &= std::fabs(dphi[((size_t)g * 27 + n) * 3 + 0] - D1[a][q1] * L1[b][q2] * L1[c][q3]) <= 1e-13;
          ok &= std::fabs(dphi[((size_t)g * 27 + n) * 3 + 1] - L1[a][q1] * D1[b][q2] * L1[c][q3]) <= 1e-13;
          ok &= std::fabs(dphi[((size_t)g * 27 + n) * 3 + 2] - L1[a][q1] * L1[b][q2] * D1[c][q3]) <= 1e-13;
        }
      if (ok) {
        // per-lane constants of the three stages (lane roles: see k_elem_q2hex_mfma, phase A)
        std::vector<double> sfc((size_t)18 * 64, 0.0);
        std::vector<int> sfi((size_t)7 * 64, 0);
        for (int l = 0; l < 64; l++) {
          if (l < 36) {                                  // stage 1: lane = (q1, b, c), contracts a
            const int q1 = l & 3, bc = l >> 2, b = bc / 3, c = bc % 3;
            for (int a = 0; a < 3; a++) {
              sfc[(size_t)a * 64 + l] = L1[a][q1];
              sfc[(size_t)(3 + a) * 64 + l] = D1[a][q1];
              sfi[(size_t)a * 64 + l] = nodeof[a * 9 + b * 3 + c] * 4;      // doubles into xs
            }
            sfi[(size_t)3 * 64 + l] = ((b * 3 + c) * 4 + q1) * 4;
          }
          if (l < 48) {                                  // stage 2: lane = (c, q1, q2), contracts b
            const int c2 = l >> 4, q12 = l & 15, q1 = q12 & 3, q2 = q12 >> 2;
            for (int b = 0; b < 3; b++) {
              sfc[(size_t)(6 + b) * 64 + l] = L1[b][q2];
              sfc[(size_t)(9 + b) * 64 + l] = D1[b][q2];
            }
            sfi[(size_t)4 * 64 + l] = (c2 * 4 + q1) * 4;
            sfi[(size_t)5 * 64 + l] = (c2 * 16 + q12) * 4;
          }
          {                                              // stage 3: lane = Gauss point, contracts c
            const int q1 = qidx[l][0], q2 = qidx[l][1], q3 = qidx[l][2];
            for (int c = 0; c < 3; c++) {
              sfc[(size_t)(12 + c) * 64 + l] = L1[c][q3];
              sfc[(size_t)(15 + c) * 64 + l] = D1[c][q3];
            }
            sfi[(size_t)6 * 64 + l] = (q1 + 4 * q2) * 4;
          }
        }
        FH_TRY(up((void**)&as->d_mfSFc, sfc.data(), sfc.size() * sizeof(double)));
        FH_TRY(up((void**)&as->d_mfSFi, sfi.data(), sfi.size() * sizeof(int)));
        // tables of the sum-factorised element kernel (lane roles: see k_elem_q2hex_sf); lanes beyond a stage's role count repeat
        // its last role, so that the kernel needs no divergent branch
        {
          std::vector<double> lc((size_t)SF_NLC * 64, 0.0);
          std::vector<int> li((size_t)SF_NLI * 64, 0);
          int gof[64], tof[27];
          for (int g = 0; g < 64; g++) gof[qidx[g][0] * 16 + qidx[g][1] * 4 + qidx[g][2]] = g;
          for (int t = 0; t < 27; t++) tof[nodeof[t]] = t;        // nodeof[a*9 + b*3 + c] = node; tof = its inverse
          for (int a = 0; a < 3; a++)
            for (int k = 0; k < 4; k++) { as->sf_tab.L[a][k] = L1[a][k]; as->sf_tab.D[a][k] = D1[a][k]; }
          int pr[64][2], np = 0;
          for (int p = 0; p < 9; p++)
            for (int p2 = p; p2 < 9; p2++) { pr[np][0] = p; pr[np][1] = p2; np++; }
          for (int l = 0; l < 64; l++) {
            {                                              // phase A stage 1 and source stage 2: lane = (q1, b, c)
              const int lr = std::min(l, 35), q1 = lr & 3, bc = lr >> 2, b = bc / 3, c = bc % 3;
              for (int a = 0; a < 3; a++) {
                lc[(size_t)a * 64 + l] = L1[a][q1];
                lc[(size_t)(3 + a) * 64 + l] = D1[a][q1];
              }
              li[(size_t)0 * 64 + l] = bc;
              li[(size_t)1 * 64 + l] = lr;
              for (int k = 0; k < 4; k++) lc[(size_t)(43 + k) * 64 + l] = L1[b][k];
              li[(size_t)10 * 64 + l] = c * 16 + q1 * 4;
              li[(size_t)11 * 64 + l] = (b * 3 + c) * 4 + q1;
            }
            {                                              // phase A stage 2: lane = (c, q1, q2)
              const int lr = std::min(l, 47), c2 = lr >> 4, q12 = lr & 15, q1 = q12 & 3, q2 = q12 >> 2;
              for (int b = 0; b < 3; b++) {
                lc[(size_t)(6 + b) * 64 + l] = L1[b][q2];
                lc[(size_t)(9 + b) * 64 + l] = D1[b][q2];
              }
              li[(size_t)2 * 64 + l] = c2 * 4 + q1;
              li[(size_t)3 * 64 + l] = lr;
            }
            {                                              // phase A stage 3: lane = Gauss point, 16 q3 + 4 q1 + q2 (the B-operand layout of stage 1)
              const int q3 = l >> 4, q1 = (l >> 2) & 3, q2 = l & 3;
              for (int c = 0; c < 3; c++) {
                lc[(size_t)(12 + c) * 64 + l] = L1[c][q3];
                lc[(size_t)(15 + c) * 64 + l] = D1[c][q3];
              }
              li[(size_t)4 * 64 + l] = q1 + 4 * q2;
              lc[(size_t)18 * 64 + l] = w[gof[q1 * 16 + q2 * 4 + q3]];
            }
            {                                              // stage 1, A operands: lane = 16 k + 4 block + row, k = q3; eight vectors of four slots
              const int k = l >> 4, r = l & 3;
              static const int symc[6][2] = {{0, 0}, {0, 1}, {0, 2}, {1, 1}, {1, 2}, {2, 2}};
              for (int g = 0; g < 2; g++) {
                const int sl = 4 * g + r;
                lc[(size_t)(19 + g) * 64 + l] = sl < 6 ? L1[symc[sl][0]][k] * L1[symc[sl][1]][k] : 0.0;      // l_c l_c'
                lc[(size_t)(24 + g) * 64 + l] = sl < 6 ? D1[symc[sl][0]][k] * D1[symc[sl][1]][k] : 0.0;      // l'_c l'_c'
              }
              for (int g = 0; g < 3; g++) {
                const int sl = 4 * g + r;
                lc[(size_t)(21 + g) * 64 + l] = sl < 9 ? L1[sl / 3][k] * D1[sl % 3][k] : 0.0;                // l_c l'_c'
              }
              lc[(size_t)26 * 64 + l] = r < 3 ? L1[r][k] : 0.0;                                              // source: l_c
            }
            {                                              // stages 2, 3: lane = unordered pair {(b,c), (b',c')}
              const int lr = std::min(l, np - 1);
              const int p = pr[lr][0], p2 = pr[lr][1];
              const int b = p / 3, c = p % 3, b2 = p2 / 3, c2 = p2 % 3;
              for (int k = 0; k < 4; k++) {
                lc[(size_t)(27 + k) * 64 + l] = L1[b][k] * L1[b2][k];
                lc[(size_t)(31 + k) * 64 + l] = L1[b][k] * D1[b2][k];
                lc[(size_t)(35 + k) * 64 + l] = D1[b][k] * L1[b2][k];
                lc[(size_t)(39 + k) * 64 + l] = D1[b][k] * D1[b2][k];
              }
              {
                const int cl = std::min(c, c2), ch = std::max(c, c2);
                const int symslot = cl == 0 ? ch : cl == 1 ? 2 + ch : 5;          // (0,0) (0,1) (0,2) (1,1) (1,2) (2,2)
                li[(size_t)5 * 64 + l] = symslot * SF_ES;
              }
              li[(size_t)6 * 64 + l] = (c * 3 + c2) * SF_ES;
              li[(size_t)15 * 64 + l] = (c2 * 3 + c) * SF_ES;
              li[(size_t)7 * 64 + l] = p * MF_KS + p2;
              li[(size_t)8 * 64 + l] = p2 * MF_KS + p;
              li[(size_t)9 * 64 + l] = p == p2 ? 1 : 0;
            }
            {                                              // source stage 3: lane = tensor node t = min(l & 31, 26)
              const int t = std::min(l & 31, 26);
              for (int k = 0; k < 4; k++) lc[(size_t)(47 + k) * 64 + l] = L1[t / 9][k];
              li[(size_t)16 * 64 + l] = (t % 9) * 4;
              li[(size_t)12 * 64 + l] = tof[std::min(l >> 1, 26)];           // row stores: column of node l >> 1
              li[(size_t)13 * 64 + l] = tof[std::min(l & 31, 26)];
              li[(size_t)14 * 64 + l] = nodeof[std::min(l, 26)];
            }
          }
          FH_TRY(up((void**)&as->d_sfLc, lc.data(), lc.size() * sizeof(double)));
          FH_TRY(up((void**)&as->d_sfLi, li.data(), li.size() * sizeof(int)));
        }
      }
    }
  }
  FH_TRACE("fh_assembler_create: tables uploaded (nel %d, nnode %d)", nel, nnode);
  // (element colours: made when the coloured scatter or fh_assembler_info asks for them -- ensure_colors; the default two-pass / fused paths need none)
  std::vector<int> iota(nel);
  for (int e = 0; e < nel; e++) iota[e] = e;
  FH_TRY(up((void**)&as->d_iota, iota.data(), iota.size() * sizeof(int)));
  if (ctx->assemble_emap && !(ctx->assemble_two_pass && A->max_row <= 255)) {
    // symbolic phase: CSR slot of every element entry, built on the device with the same kernel
    FH_CHECK_HIP(hipMalloc(&as->d_emap, std::max<size_t>((size_t)nel * as->nc * as->ncp, 1) * sizeof(int)));
    AsmParams P = base_params(as);
    P.elems = as->d_iota;
    P.nelems = nel;
    P.rowptr = A->d_rowptr;
    P.col = A->d_col;
    P.nrows = A->m;
    P.emap_out = as->d_emap;
    P.ng = 0;   // no quadrature needed for the symbolic pass
    FH_TRY(dispatch_assemble(as, P));
    FH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  }
  if (ctx->assemble_two_pass && A->max_row <= 255) {
    // row -> (element, local row) adjacency in ascending element order, on the device (round 4): count per row, host scan of the counts, fill
    // through per-row cursors (any order), then every row sorts its few pairs and tells each (element, local row) its slot -- the arrays the host
    // loop produced; only the row pointers (needed by the cluster plan below) visit the host
    const int m = A->m, nc = as->nc;
    const size_t npair = (size_t)nel * nc;
    std::vector<int> aptr(m + 1, 0);
    {
      int* d_cnt = nullptr;
      FH_CHECK_HIP(hipMalloc(&d_cnt, ((size_t)m + 1) * sizeof(int)));
      FH_CHECK_HIP(hipMemsetAsync(d_cnt, 0, ((size_t)m + 1) * sizeof(int), ctx->stream));
      const unsigned gb = (unsigned)((npair + 255) / 256);
      if (npair) hipLaunchKernelGGL(k_adj_count, dim3(gb), dim3(256), 0, ctx->stream, npair, nc, nloc, as->d_elem_dof, m, d_cnt);
      if (m) FH_CHECK_HIP(hipMemcpyAsync(aptr.data() + 1, d_cnt, (size_t)m * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
      FH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
      int64_t tot = 0;
      for (int r = 0; r < m; r++) {
        tot += aptr[r + 1];
        aptr[r + 1] = (int)tot;
      }
      if (tot >= 2147483647ll) {
        hipFree(d_cnt);
        fh_set_error("fh_assembler_create: the adjacency overflows int32");
        return 2;
      }
      FH_CHECK_HIP(hipMalloc(&as->d_adj_ptr, ((size_t)m + 1) * sizeof(int)));
      FH_CHECK_HIP(hipMalloc(&as->d_adj_ei, std::max<size_t>((size_t)tot, 1) * sizeof(int)));
      FH_CHECK_HIP(hipMalloc(&as->d_slot, std::max<size_t>(npair, 1) * sizeof(int)));
      FH_CHECK_HIP(hipMemcpyAsync(as->d_adj_ptr, aptr.data(), ((size_t)m + 1) * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
      FH_CHECK_HIP(hipMemcpyAsync(d_cnt, aptr.data(), (size_t)m * sizeof(int), hipMemcpyHostToDevice, ctx->stream));      // cursors
      FH_CHECK_HIP(hipMemsetAsync(as->d_slot, 0xFF, std::max<size_t>(npair, 1) * sizeof(int), ctx->stream));                // -1: row not in the matrix
      FH_REQUIRE(nel < (1 << 26), "fh_assembler_create: too many elements for the packed adjacency");
      if (npair) hipLaunchKernelGGL(k_adj_fill, dim3(gb), dim3(256), 0, ctx->stream, npair, nc, nloc, as->d_elem_dof, m, d_cnt, as->d_adj_ei);
      if (m) hipLaunchKernelGGL(k_adj_sort, dim3(fh_div_up(m, 256)), dim3(256), 0, ctx->stream, m, nc, as->d_adj_ptr, as->d_adj_ei, as->d_slot);
      FH_CHECK_HIP(hipGetLastError());
      FH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
      hipFree(d_cnt);
    }
    const size_t nadj = (size_t)aptr[m];
    FH_TRACE("fh_assembler_create: row adjacency built (%zu pairs)", nadj);
    FH_CHECK_HIP(hipMalloc(&as->d_rowmap, std::max<size_t>(nadj * nc, 1)));
    as->kstride = (nc == 27 && ctx->assemble_kpad) ? (ctx->assemble_kpad == 28 ? 28 : 32) : nc;
    as->nadj = (int)nadj;
    as->kbuf_bytes = (nadj + 1) * as->kstride * sizeof(double);      // + one spare row: the sink of rows without a slot
    FH_CHECK_HIP(hipMalloc(&as->d_Kbuf, as->kbuf_bytes));
    FH_CHECK_HIP(hipMalloc(&as->d_Fbuf, std::max<size_t>(nadj, 1) * sizeof(double)));
    if (ctx->debug_poison) {   // tests: the row pass must read nothing the element kernels have not written
      FH_CHECK_HIP(hipMemset(as->d_Kbuf, 0xFF, as->kbuf_bytes));
      FH_CHECK_HIP(hipMemset(as->d_Fbuf, 0xFF, std::max<size_t>(nadj, 1) * sizeof(double)));
    }
    FH_TRACE("fh_assembler_create: element-row buffer allocated (%.2f GB), building the row map", as->kbuf_bytes / 1e9);
    FH_TRY(dispatch_rows(as, A, nullptr, true));
    FH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    FH_TRACE("fh_assembler_create: row map built");
    as->two_pass = true;
    if (ctx->assemble_fused && as->dim == 3 && as->nc == 27 && as->ng == 64 && as->d_sfLc && ctx->assemble_sf) FH_TRY(cluster_plan_build(as, A, elem_dof, aptr));
  }
  *out = as;
  return 0;
  FH_GUARD_END("fh_assembler_create")
}

extern "C" int fh_assembler_create(fh_ctx_t ctx, int geom, int fe, int order, int nel, int nloc, const int* elem_dof, int nnode,
                                   const double* coords, fh_mat_t A, fh_assembler_t* out) {
  return assembler_create_impl(ctx, geom, fe, order, nel, nloc, elem_dof, nnode, coords, nullptr, nullptr, A, out);
}

int fh_mesh_host_arrays(fh_mesh_t m, int* dim, int* geom, int* nel, int* nnode, int* nloc, int* n_linear, const int** elem_dof, const double** coords);

extern "C" int fh_assembler_create_mesh(fh_ctx_t ctx, fh_mesh_t mesh, int fe, int order, fh_mat_t A, fh_assembler_t* out) {
  FH_REQUIRE(ctx && mesh && A && out, "fh_assembler_create_mesh: null argument");
  int dim, geom, nel, nnode, nloc, nlin;
  const int* ed;
  const double* xy;
  FH_TRY(fh_mesh_host_arrays(mesh, &dim, &geom, &nel, &nnode, &nloc, &nlin, &ed, &xy));
  fh_mesh_dev* dev = nullptr;
  FH_TRY(fh_mesh_device(ctx, mesh, &dev));
  return assembler_create_impl(ctx, geom, fe, order, nel, nloc, ed, nnode, xy, dev->d_elem_dof, dev->d_coords, A, out);
}

extern "C" int fh_assembler_destroy(fh_assembler_t as) {
  if (!as) return 0;
  hipStreamSynchronize(as->ctx->stream);
  if (as->d_color_elems) hipFree(as->d_color_elems);
  hipFree(as->d_elem_dof);
  hipFree(as->d_coords);
  hipFree(as->d_w);
  hipFree(as->d_phi);
  hipFree(as->d_dphi);
  if (as->d_emap) hipFree(as->d_emap);
  for (void* q : {(void*)as->d_aff_elems, (void*)as->d_gen_elems, (void*)as->d_Mab, (void*)as->d_mphi, (void*)as->d_mfT, (void*)as->d_mfPhi, (void*)as->d_mfSFc, (void*)as->d_mfSFi, (void*)as->d_sfLc, (void*)as->d_sfLi})
    if (q) hipFree(q);
  if (as->d_prog) hipFree(as->d_prog);
  if (as->d_prog_consts) hipFree(as->d_prog_consts);
  hipFree(as->d_iota);
  for (void* q : {(void*)as->d_cl_dtab, (void*)as->d_cl_fblk, (void*)as->d_cl_sinfo, (void*)as->d_cl_oblk, (void*)as->d_cl_gtab, (void*)as->d_cl_vdst, (void*)as->d_cl_vdst64, (void*)as->d_cl_fdst, (void*)as->d_cl_map, (void*)as->d_cl_mapb, (void*)as->d_cl_pmap,
                  (void*)as->d_Pbuf, (void*)as->d_cl_prow, (void*)as->d_cl_pstart})
    if (q) hipFree(q);
  for (void* q : {(void*)as->d_adj_ptr, (void*)as->d_adj_ei, (void*)as->d_rowmap, (void*)as->d_Kbuf, (void*)as->d_Fbuf, (void*)as->d_slot, (void*)as->d_gal_child,
                  (void*)as->d_gal_cnt, (void*)as->d_gal_row, (void*)as->d_gal_fb, (void*)as->d_gal_cb, (void*)as->d_gal_val, (void*)as->d_gal_res, (void*)as->d_gal_dense,
                  (void*)as->d_gal_fmask})
    if (q) hipFree(q);
  delete as;
  return 0;
}

static int assemble_poisson_core(fh_assembler_t as, fh_vec_t sol, int source_kind, const double* params, fh_mat_t A, fh_vec_t res);

extern "C" int fh_assemble_poisson(fh_assembler_t as, fh_vec_t sol, int source_kind, const double* params, fh_mat_t A, fh_vec_t res) {
  FH_REQUIRE(source_kind >= 0 && source_kind <= 3, "fh_assemble_poisson: unknown source kind %d", source_kind);
  return assemble_poisson_core(as, sol, source_kind, params, A, res);
}

int fh_expr_program(fh_expr_t e, int* ncode, int* nconst, int* code, double* consts);

// source term from a run-time expression: f = scale * expr(x, y, z, t) evaluated at every Gauss point on the device
// (001_Poisson/main.cpp:472 calls the ParsedFunction on the host once per Gauss point and test function)
extern "C" int fh_assemble_poisson_expr(fh_assembler_t as, fh_vec_t sol, fh_expr_t source, double scale, fh_mat_t A, fh_vec_t res) {
  FH_REQUIRE(as && source, "fh_assemble_poisson_expr: null argument");
  {
    // the kernels hand the program a 4-entry point (x, y, z, t): a program compiled over more variables would read past it
    int nv = 0;
    FH_TRY(fh_expr_nvars(source, &nv));
    FH_REQUIRE(nv <= 4, "fh_assemble_poisson_expr: the source expression has %d variables, at most 4 (x, y, z, t) are served", nv);
  }
  int nc = 0, nk = 0;
  FH_TRY(fh_expr_program(source, &nc, &nk, nullptr, nullptr));
  std::vector<int> code(nc);
  std::vector<double> consts(std::max(nk, 1));
  FH_TRY(fh_expr_program(source, &nc, &nk, code.data(), consts.data()));
  consts.resize(nk);
  if (code != as->h_prog || consts != as->h_prog_consts || !as->d_prog) {
    FH_CHECK_HIP(hipStreamSynchronize(as->ctx->stream));
    if (as->d_prog) hipFree(as->d_prog);
    if (as->d_prog_consts) hipFree(as->d_prog_consts);
    FH_CHECK_HIP(hipMalloc(&as->d_prog, std::max<size_t>(code.size(), 1) * sizeof(int)));
    FH_CHECK_HIP(hipMalloc(&as->d_prog_consts, std::max<size_t>(consts.size(), 1) * sizeof(double)));
    FH_CHECK_HIP(hipMemcpy(as->d_prog, code.data(), code.size() * sizeof(int), hipMemcpyHostToDevice));
    if (!consts.empty()) FH_CHECK_HIP(hipMemcpy(as->d_prog_consts, consts.data(), consts.size() * sizeof(double), hipMemcpyHostToDevice));
    as->h_prog = code;
    as->h_prog_consts = consts;
    as->nprog = nc;
  }
  const double params[2] = {scale, 0.0};
  return assemble_poisson_core(as, sol, 4, params, A, res);
}

static int assemble_poisson_core(fh_assembler_t as, fh_vec_t sol, int source_kind, const double* params, fh_mat_t A, fh_vec_t res) {
  FH_REQUIRE(as && A && res, "fh_assemble_poisson: null argument");
  FH_REQUIRE(A->m == as->ndof && res->n_local >= as->ndof, "fh_assemble_poisson: size mismatch");
  FH_REQUIRE(!sol || sol->n_local + sol->nghost >= A->n, "fh_assemble_poisson: solution vector too short (needs owned + ghost entries)");
  if (as->two_pass) {
    // pass 1: all element matrices (one launch, no colours) ; pass 2: rows gather their element rows (zeroing included)
    AsmParams P = base_params(as);
    P.sol = sol ? sol->d : nullptr;
    P.source_kind = source_kind;
    P.p0 = params ? params[0] : 1.0;
    P.p1 = params ? params[1] : 0.0;
    P.elems = as->d_iota;
    P.nelems = as->nel;
    P.Kout = as->d_Kbuf;
    P.kstride = as->kstride;
    P.Fout = as->d_Fbuf;
    P.slot = as->d_slot;
    P.nsink = as->nadj;
    P.debug = as->ctx->asm_debug;
    as->last_source_kind = source_kind;
    as->last_params[0] = P.p0;
    as->last_params[1] = P.p1;
    if (as->ctx->assemble_affine) FH_TRY(ensure_affine(as));
    // assemble_fused 1 (default): the fused path, unless the element rows of the PREVIOUS assembly were asked for by the element-wise Galerkin product
    // (a solve that re-prepares after every assembly: the two-pass path leaves the rows in place, re-creating them would cost 0.66 ms at 64^3);
    // 2: always fused; 0: never
    // (rows_used_since is only raised when the product could NOT be made from the macro rows the fused path leaves behind: fh_assembler_galerkin)
    const bool keep_rows = as->ctx->assemble_fused == 1 && as->rows_used_since;
    as->rows_used_since = false;
    as->macro_valid = false;
    if (as->fused && as->ctx->assemble_fused && !keep_rows && as->ctx->assemble_sf && !(as->ctx->assemble_affine && as->d_Mab && as->n_aff > 0) && !(as->ctx->asm_debug & 16)) {
      as->last_path = 1;
      // fused cluster assembly: complete rows straight into the CSR arrays, the others through the partial-row buffer (the element-row buffer is not written)
      as->kbuf_valid = false;
      FH_TRY(launch_cluster(as, P, A, res->d));
      A->at_valid = false;
      as->macro_valid = !(as->ctx->asm_debug & (2 | 4 | 8));     // (timing aids leave rows unwritten)
      as->macro_val_gen = ++A->val_gen;
      as->cl_mat_of_macro = A;
      return 0;
    }
    as->kbuf_valid = true;
    as->last_path = 2;
    if (as->ctx->assemble_affine && as->d_Mab && as->n_aff > 0) {
      // affine elements through the reference-matrix kernel, the rest (curved ones) through the quadrature kernel
      AsmParams Pa = P;
      Pa.elems = as->d_aff_elems;
      Pa.nelems = as->n_aff;
      const dim3 grid(fh_div_up(as->n_aff, 16)), block(256);
      if (P.source_kind == 4) hipLaunchKernelGGL(k_elem_q2hex_affine<2>, grid, block, 0, as->ctx->stream, Pa, as->d_Mab, as->d_mphi);
      else if (P.source_kind != 0) hipLaunchKernelGGL(k_elem_q2hex_affine<1>, grid, block, 0, as->ctx->stream, Pa, as->d_Mab, as->d_mphi);
      else hipLaunchKernelGGL(k_elem_q2hex_affine<0>, grid, block, 0, as->ctx->stream, Pa, as->d_Mab, as->d_mphi);
      FH_CHECK_HIP(hipGetLastError());
      P.elems = as->d_gen_elems;
      P.nelems = as->n_gen;
    }
    if (as->ctx->asm_debug & 16) {
      // TIMING PROBE ONLY (bit 4): the row pass on a second stream BESIDE the element kernel (it reads the element rows of the previous
      // assembly): do the two kernels share the device?
      static hipStream_t s2 = nullptr;
      static hipEvent_t e0 = nullptr, e1 = nullptr;
      if (!s2) {
        FH_CHECK_HIP(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
        FH_CHECK_HIP(hipEventCreateWithFlags(&e0, hipEventDisableTiming));
        FH_CHECK_HIP(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
      }
      hipStream_t s1 = as->ctx->stream;
      FH_CHECK_HIP(hipEventRecord(e0, s1));
      FH_CHECK_HIP(hipStreamWaitEvent(s2, e0, 0));
      if (as->ctx->asm_debug & 32) {        // bit 5: row pass issued first
        as->ctx->stream = s2;
        const int rc = dispatch_rows(as, A, res->d, false);
        as->ctx->stream = s1;
        FH_TRY(rc);
        FH_TRY(dispatch_assemble(as, P));
      } else {
        FH_TRY(dispatch_assemble(as, P));
        as->ctx->stream = s2;
        const int rc = dispatch_rows(as, A, res->d, false);
        as->ctx->stream = s1;
        FH_TRY(rc);
      }
      FH_CHECK_HIP(hipEventRecord(e1, s2));
      FH_CHECK_HIP(hipStreamWaitEvent(s1, e1, 0));
      A->at_valid = false;
      return 0;
    }
    FH_TRY(dispatch_assemble(as, P));
    if (!(as->ctx->asm_debug & (2 | 8))) FH_TRY(dispatch_rows(as, A, res->d, false));   // bit 3: element matrices only (timing)
    A->at_valid = false;
    return 0;
  }
  // KK->zero(); RES->zero();  (separate.hpp:106-107)
  FH_TRY(fh_mat_zero(A));
  FH_TRY(fh_vec_zero(res));
  AsmParams P = base_params(as);
  P.sol = sol ? sol->d : nullptr;
  P.source_kind = source_kind;
  P.p0 = params ? params[0] : 1.0;
  P.p1 = params ? params[1] : 0.0;
  P.rowptr = A->d_rowptr;
  P.col = A->d_col;
  P.nrows = A->m;
  P.val = A->d_val;
  P.res = res->d;
  P.emap = as->d_emap;
  P.debug = as->ctx->asm_debug;
  FH_TRY(ensure_colors(as));
  for (int c = 0; c < as->ncolors; c++) {
    P.elems = as->d_color_elems + as->color_ptr[c];
    P.nelems = as->color_ptr[c + 1] - as->color_ptr[c];
    FH_TRY(dispatch_assemble(as, P));
  }
  A->at_valid = false;
  return 0;
}

extern "C" int fh_element_matrices_poisson(fh_assembler_t as, fh_vec_t sol, int source_kind, const double* params, double* K, double* F) {
  FH_REQUIRE(as && K && F, "fh_element_matrices_poisson: null argument");
  const size_t nk = (size_t)as->nel * as->nc * as->nc, nf = (size_t)as->nel * as->nc;
  double *dK = nullptr, *dF = nullptr;
  FH_CHECK_HIP(hipMalloc(&dK, std::max<size_t>(nk, 1) * sizeof(double)));
  FH_CHECK_HIP(hipMalloc(&dF, std::max<size_t>(nf, 1) * sizeof(double)));
  AsmParams P = base_params(as);
  P.sol = sol ? sol->d : nullptr;
  P.source_kind = source_kind;
  P.p0 = params ? params[0] : 1.0;
  P.p1 = params ? params[1] : 0.0;
  P.elems = as->d_iota;
  P.nelems = as->nel;
  P.Kout = dK;
  P.Fout = dF;
  FH_TRY(dispatch_assemble(as, P));
  FH_CHECK_HIP(hipMemcpyAsync(K, dK, nk * sizeof(double), hipMemcpyDeviceToHost, as->ctx->stream));
  FH_CHECK_HIP(hipMemcpyAsync(F, dF, nf * sizeof(double), hipMemcpyDeviceToHost, as->ctx->stream));
  FH_CHECK_HIP(hipStreamSynchronize(as->ctx->stream));
  hipFree(dK);
  hipFree(dF);
  return 0;
}

extern "C" int fh_assembler_affine_count(fh_assembler_t as, int* n_affine, int* n_general) {
  FH_REQUIRE(as, "fh_assembler_affine_count: null argument");
  FH_TRY(ensure_affine(as));
  if (n_affine) *n_affine = as->d_Mab ? as->n_aff : 0;
  if (n_general) *n_general = as->d_Mab ? as->n_gen : as->nel;
  return 0;
}

extern "C" int fh_assembler_last_path(fh_assembler_t as, int* path) {
  FH_REQUIRE(as && path, "fh_assembler_last_path: null argument");
  *path = as->last_path;
  return 0;
}

extern "C" int fh_assembler_fused_info(fh_assembler_t as, int* active, int* nclusters, int64_t* partial_entries, int* second_pass_rows) {
  FH_REQUIRE(as, "fh_assembler_fused_info: null argument");
  const bool on = as->fused && as->ctx->assemble_fused && as->ctx->assemble_sf;
  if (active) *active = on ? 1 : 0;
  if (nclusters) *nclusters = as->fused ? as->cl_ncl : 0;
  if (partial_entries) *partial_entries = as->fused ? (int64_t)as->cl_npart : 0;
  if (second_pass_rows) *second_pass_rows = as->fused ? as->cl_nprow : 0;
  return 0;
}

extern "C" int fh_assembler_carry_info(fh_assembler_t as, int* clusters_per_super, int64_t* carried_entries) {
  FH_REQUIRE(as, "fh_assembler_carry_info: null argument");
  const bool on = as->fused && as->ctx->assemble_fused && as->ctx->assemble_sf;
  if (clusters_per_super) *clusters_per_super = on ? 1 << as->cl_sup_shift : 0;
  if (carried_entries) *carried_entries = on ? (int64_t)as->cl_ncarried : 0;
  return 0;
}

extern "C" int fh_assembler_info(fh_assembler_t as, int* ncolors, int64_t* algorithmic_bytes, double* flops) {
  if (ncolors) {
    FH_TRY(ensure_colors(as));
    *ncolors = as->ncolors;
  }
  const int nc = as->nc, dim = as->dim, ng = as->ng;
  // per element reads nc*dim*8 (coords) + nc*4 (dof ids) + nc*8 (u), writes nc*nc*8 (K) + nc*8 (F)
  if (algorithmic_bytes) *algorithmic_bytes = (int64_t)as->nel * (nc * dim * 8 + nc * 4 + nc * 8 + nc * nc * 8 + nc * 8);
  if (flops)
    *flops = (double)as->nel * ng * (nc * dim * dim * 2.0 + 60.0 + nc * dim * dim * 2.0 + (double)nc * nc * (dim * 2.0 + 2.0) + nc * 10.0);
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// Galerkin coarse operator ELEMENT BY ELEMENT (uniform refinement): A = sum_e K_e and every fine element lies in one coarse element, whose
// 27 (9) nodes interpolate its nodes by the fixed child matrix C_j (the rows of the element prolongator, ElemType.cpp:439-532), so
//   PP^T KK PP = sum_E sum_{children j of E} C_j^T K_{child j} C_j          (LinearImplicitSystem.cpp:347-370, PetscMatrix.cpp:733-751)
// -- coarse ELEMENT matrices from the fine ones the assembler still holds, then the same row pass as the assembly.  The general sparse triple
// product streams ~10 elementary products per non-zero (8 ms on the 64^3 level); this reads the element-row buffer once (1.8 GB) and does
// 7 k multiply-adds per child with the 125 non-zeros of C_j.  Rows of the interpolation at fine Dirichlet nodes and its columns at coarse
// Dirichlet nodes are zero (ZeroInterpolatorDirichletNodes, :1032-1120): masks on K_j and on the result.  Same value as the product up to
// the order of the sums.  One wave per coarse element.
// ------------------------------------------------------------------------------------------------------------------
template <int NC, int NCH>
__global__ __launch_bounds__(256) void k_galerkin_elem(int nelc, const int* __restrict__ child, const int* __restrict__ slot_f, const double* __restrict__ Kf, int ks_f,
                                                       const int* __restrict__ edof_f, int nloc_f, const unsigned char* __restrict__ fb,
                                                       const int* __restrict__ slot_c, double* __restrict__ Kc, int ks_c, const int* __restrict__ edof_c, int nloc_c,
                                                       const unsigned char* __restrict__ cb, const unsigned char* __restrict__ tab_cnt,
                                                       const unsigned char* __restrict__ tab_row, const double* __restrict__ tab_val) {
  constexpr int NE = NC * NC, NT = (NE + 63) / 64, LD = NC + 1;
  __shared__ double cval[NCH * NC * 8];
  __shared__ unsigned char crow[NCH * NC * 8], ccnt[NCH * NC];
  __shared__ double slab[4][2][NC * LD];
  for (int k = threadIdx.x; k < NCH * NC * 8; k += 256) { cval[k] = tab_val[k]; crow[k] = tab_row[k]; }
  for (int k = threadIdx.x; k < NCH * NC; k += 256) ccnt[k] = tab_cnt[k];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* Ks = slab[wave][0];
  double* Ts = slab[wave][1];
  for (int E = blockIdx.x * 4 + wave; E < nelc; E += gridDim.x * 4) {
    double acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = 0.0;
    for (int j = 0; j < NCH; j++) {
      const int ej = child[(size_t)E * NCH + j];
      // all gathers of the child first (independent chains in flight), then the LDS writes
      double kin[NT];
#pragma unroll
      for (int t = 0; t < NT; t++) {
        const int idx = min(lane + 64 * t, NE - 1);
        const int i = idx / NC, c = idx - i * NC;
        const int s = slot_f[(size_t)ej * NC + i];
        const bool dead = s < 0 || fb[edof_f[(size_t)ej * nloc_f + i]] || fb[edof_f[(size_t)ej * nloc_f + c]];
        kin[t] = dead ? 0.0 : Kf[(size_t)max(s, 0) * ks_f + c];
      }
#pragma unroll
      for (int t = 0; t < NT; t++) {
        const int idx = lane + 64 * t;
        if (idx < NE) Ks[(idx / NC) * LD + idx % NC] = kin[t];
      }
      wave_lds_sync();
      // T = K_j C_j and K_E += C_j^T T: the (at most 8) non-zeros of a column of C_j, padded with zero weights -- a fixed trip count and
      // NT independent chains per lane instead of one serial chain of dependent LDS reads per entry
      {
        double tv[NT];
        int rbase[NT], cbase[NT];
#pragma unroll
        for (int t = 0; t < NT; t++) {
          const int idx = min(lane + 64 * t, NE - 1);
          rbase[t] = (idx / NC) * LD;
          cbase[t] = (j * NC + idx % NC) * 8;
          tv[t] = 0.0;
        }
#pragma unroll
        for (int q = 0; q < 8; q++)
#pragma unroll
          for (int t = 0; t < NT; t++) tv[t] += Ks[rbase[t] + crow[cbase[t] + q]] * cval[cbase[t] + q];
#pragma unroll
        for (int t = 0; t < NT; t++) {
          const int idx = lane + 64 * t;
          if (idx < NE) Ts[(idx / NC) * LD + idx % NC] = tv[t];
        }
      }
      wave_lds_sync();
      {
        int kcol[NT], abase[NT];
#pragma unroll
        for (int t = 0; t < NT; t++) {
          const int idx = min(lane + 64 * t, NE - 1);
          kcol[t] = idx % NC;
          abase[t] = (j * NC + idx / NC) * 8;
        }
#pragma unroll
        for (int q = 0; q < 8; q++)
#pragma unroll
          for (int t = 0; t < NT; t++) acc[t] += cval[abase[t] + q] * Ts[crow[abase[t] + q] * LD + kcol[t]];
      }
      wave_lds_sync();
    }
#pragma unroll
    for (int t = 0; t < NT; t++) {
      const int idx = lane + 64 * t;
      if (idx < NE) {
        const int a = idx / NC, k = idx - a * NC;
        const int s = slot_c[(size_t)E * NC + a];
        if (s >= 0) {
          const bool dead = cb[edof_c[(size_t)E * nloc_c + a]] || cb[edof_c[(size_t)E * nloc_c + k]];
          Kc[(size_t)s * ks_c + k] = dead ? 0.0 : acc[t];
        }
      }
    }
  }
}

// The same on the FP64 matrix cores (default): per child two small dense products T = K_j C_j and K_E += C_j^T T as v_mfma_f64_16x16x4 tiles
// (27 -> 32 x 32 x 28 padded: 2 x 28 instructions of 64 cycles), operands from LDS (the eight C_j, the gathered K_j, T), the element rows of
// the NEXT child gathered into registers while the current one is multiplied.  One wave per coarse element, one workgroup of four per CU.
// waves per workgroup: they share the child matrices C_j in LDS; the per-wave scratch holds K_j first and T = K_j C_j afterwards (one region), so that
// eight waves -- two per SIMD, one's LDS round trips behind the other's matrix instructions -- fit beside the 57 KB of C
constexpr int GAL_NW = 8;
template <int NC, int NCH>
__global__ __launch_bounds__(GAL_NW * 64) void k_galerkin_mfma(int nelc, const int* __restrict__ child, const int* __restrict__ slot_f, const double* __restrict__ Kf, int ks_f,
                                                       const int* __restrict__ edof_f, int nloc_f, const unsigned char* __restrict__ fb,
                                                       const int* __restrict__ slot_c, double* __restrict__ Kc, int ks_c, const int* __restrict__ edof_c, int nloc_c,
                                                       const unsigned char* __restrict__ cb, const double* __restrict__ Cdense /* [NCH][NC][NC] */) {
  constexpr int NE = NC * NC, NT = (NE + 63) / 64, MT = (NC + 15) / 16, KP = (NC + 3) / 4 * 4, CLD = MT * 16, KLD = KP + 1, TLD = MT * 16;
  extern __shared__ __attribute__((aligned(16))) double gm_smem[];
  double* Cs = gm_smem;                                   // [NCH][KP][CLD], zero padded
  for (int k = threadIdx.x; k < NCH * KP * CLD; k += GAL_NW * 64) {
    const int j = k / (KP * CLD), r = (k / CLD) % KP, c = k % CLD;
    Cs[k] = (r < NC && c < NC) ? Cdense[(j * NC + r) * NC + c] : 0.0;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int WS = (MT * 16 * KLD > KP * TLD) ? MT * 16 * KLD : KP * TLD;    // doubles of scratch per wave
  double* Ks = Cs + NCH * KP * CLD + wave * WS;      // [MT*16][KLD]
  double* Ts = Ks;                                    // [KP][TLD]: the same region, after K_j has been consumed
  __syncthreads();
  const int kk = lane >> 4, li = lane & 15;
  typedef double d4 __attribute__((ext_vector_type(4)));
  for (int E = blockIdx.x * GAL_NW + wave; E < nelc; E += gridDim.x * GAL_NW) {
    d4 KE[MT][MT];
#pragma unroll
    for (int a = 0; a < MT; a++)
#pragma unroll
      for (int b = 0; b < MT; b++) KE[a][b] = d4{0.0, 0.0, 0.0, 0.0};
    double kin[NT];
    auto gather = [&](int j) {      // K_j with the rows / columns of Dirichlet nodes zeroed (kept in registers until the LDS is free)
      const int ej = child[(size_t)E * NCH + j];
      const int ln = min(lane, NC - 1);
      const int sl = slot_f[(size_t)ej * NC + ln];
      const bool dn = sl < 0 || fb[edof_f[(size_t)ej * nloc_f + ln]];
      const unsigned long long dead = __ballot(dn && lane < NC);
#pragma unroll
      for (int t = 0; t < NT; t++) {
        const int idx = min(lane + 64 * t, NE - 1);
        const int i = idx / NC, c = idx - i * NC;
        const int s = __shfl(sl, i, 64);
        const bool d = ((dead >> i) | (dead >> c)) & 1ull;
        kin[t] = d ? 0.0 : Kf[(size_t)max(s, 0) * ks_f + c];
      }
    };
    gather(0);
    for (int j = 0; j < NCH; j++) {
#pragma unroll
      for (int t = 0; t < NT; t++) {
        const int idx = lane + 64 * t;
        if (idx < NE) Ks[(idx / NC) * KLD + idx % NC] = kin[t];
      }
      // the padding of K_j (rows and columns NC .. of the operand tile) is zero: T of the previous child lived here
      for (int q = lane; q < (MT * 16 - NC) * KLD; q += 64) Ks[NC * KLD + q] = 0.0;
      for (int q = lane; q < NC * (KLD - NC); q += 64) Ks[(q / (KLD - NC)) * KLD + NC + q % (KLD - NC)] = 0.0;
      wave_lds_sync();
      if (j + 1 < NCH) gather(j + 1);
      const double* Cj = Cs + j * KP * CLD;
      d4 T[MT][MT];
#pragma unroll
      for (int a = 0; a < MT; a++)
#pragma unroll
        for (int b = 0; b < MT; b++) T[a][b] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int k0 = 0; k0 < KP; k0 += 4) {
        double av[MT], bv[MT];
#pragma unroll
        for (int x = 0; x < MT; x++) {
          av[x] = Ks[(x * 16 + li) * KLD + k0 + kk];
          bv[x] = Cj[(k0 + kk) * CLD + x * 16 + li];
        }
#pragma unroll
        for (int a = 0; a < MT; a++)
#pragma unroll
          for (int b = 0; b < MT; b++) T[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], T[a][b], 0, 0, 0);
      }
      wave_lds_sync();                       // every lane has read its operands of K_j: T may take the region
#pragma unroll
      for (int a = 0; a < MT; a++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int i = a * 16 + kk + 4 * r;
          if (i < KP) {
#pragma unroll
            for (int b = 0; b < MT; b++) Ts[i * TLD + b * 16 + li] = T[a][b][r];
          }
        }
      wave_lds_sync();
#pragma unroll
      for (int k0 = 0; k0 < KP; k0 += 4) {
        double av[MT], bv[MT];
#pragma unroll
        for (int x = 0; x < MT; x++) {
          av[x] = Cj[(k0 + kk) * CLD + x * 16 + li];        // A[row a][k i] = C_j[i][a]
          bv[x] = Ts[(k0 + kk) * TLD + x * 16 + li];
        }
#pragma unroll
        for (int a = 0; a < MT; a++)
#pragma unroll
          for (int b = 0; b < MT; b++) KE[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], KE[a][b], 0, 0, 0);
      }
      wave_lds_sync();
    }
    // rows / columns of coarse Dirichlet nodes are zero; rows without a slot are not stored
    const int ln = min(lane, NC - 1);
    const int slc = slot_c[(size_t)E * NC + ln];
    const unsigned long long cdead = __ballot(lane < NC && cb[edof_c[(size_t)E * nloc_c + ln]]);
#pragma unroll
    for (int a = 0; a < MT; a++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int ra = a * 16 + kk + 4 * r;
        const int s = __shfl(slc, min(ra, NC - 1), 64);
#pragma unroll
        for (int b = 0; b < MT; b++) {
          const int k = b * 16 + li;
          if (ra < NC && k < NC && s >= 0) {
            const bool d = ((cdead >> ra) | (cdead >> k)) & 1ull;
            Kc[(size_t)s * ks_c + k] = d ? 0.0 : KE[a][b][r];
          }
        }
      }
  }
}

template <int NC, int NCH>
static size_t galerkin_mfma_lds() {
  constexpr int MT = (NC + 15) / 16, KP = (NC + 3) / 4 * 4;
  const size_t ws = std::max((size_t)MT * 16 * (KP + 1), (size_t)KP * MT * 16);
  return ((size_t)NCH * KP * MT * 16 + GAL_NW * ws) * sizeof(double);
}

// ------------------------------------------------------------------------------------------------------------------
// Element-wise Galerkin product FROM THE MACRO ROWS of a fused assembly (round 5).  The fused path keeps no element rows; what it leaves behind per cluster
// (= per coarse element) is the 125 x 125 macro matrix K_m = sum_j S_j^T K_j S_j: complete rows in the CSR array, the others as packed rows in the
// partial-row buffer, 4913 entries at the addresses the cluster kernel stored them to.  With C (125 x 27) the interpolation from the coarse element,
// K_E = C^T K_m C = sum_j C_j^T K~_j C_j for ANY split of the macro entries over the children (K~_j = the entries child j owns: an entry met by several
// children belongs to the first, table gtab).  One workgroup of eight waves per coarse element: all threads read the 4913 entries back (the loads of the
// NEXT cluster fly during the products) into a template-ordered LDS array, wave j forms K~_j from it and runs the two 32 x 32 x 28 products of
// k_galerkin_mfma on the matrix cores, the eight results are added in child order and leave as coarse element rows.  39 kB read per coarse element instead
// of 55 kB, and the fused assembly stays the path of every assembly of a solve (LinearImplicitSystem.cpp:288-411: assemble -> Galerkin -> solve, every time).
// Between the assembly and this product only Dirichlet ROWS of the fine matrix may have been replaced (SetPenalty): they are masked here anyway.
// ------------------------------------------------------------------------------------------------------------------
constexpr int GMAC_NW = 8, GMAC_T = GMAC_NW * 64;
constexpr int GMAC_KP = 28;                                             // K of the two products, padded to the matrix instruction's 4
constexpr int GMAC_TN = 4928;                                           // template entries (4913), padded
constexpr int GMAC_RS = 27 * 28;                                        // one wave's result [27][28]
constexpr int GMAC_BUF = GMAC_NW * GMAC_RS > GMAC_TN ? GMAC_NW * GMAC_RS : GMAC_TN;      // the macro matrix, then the eight results (same region)
constexpr size_t gmac_lds_bytes() { return (size_t)GMAC_BUF * sizeof(double) + CL_NM_MAX * sizeof(unsigned long long) + 32 * sizeof(int); }
static_assert(gmac_lds_bytes() <= 160 * 1024, "k_galerkin_macro: LDS budget");

// Operands in registers (second form of the round; the first staged K~_j, C_j and T through LDS like k_galerkin_mfma and took 1.0 ms against that kernel's
// 0.86): wave j always serves child j, so its fragments of C_j -- B operand of T = K~_j C_j AND A operand of R = C_j^T T, the same values -- and the
// template indices of its K~_j fragments are loop invariant (14 doubles + 14 indices per lane); T leaves the first product in exactly the lanes the second
// product wants it as B operand (row 4 m + kk of T = accumulator m / 4, register m % 4 of lane (kk, li)): no LDS round trip between the two products.
// Dirichlet masks of the children's and of the coarse element's 27 nodes (bit = local node), once per hierarchy: the product kernel then needs no chain
// of dependent look-ups (children -> element dofs -> flags) per cluster
__global__ __launch_bounds__(256) void k_gal_masks(int nelc, const int* __restrict__ child, const int* __restrict__ edof_f, int nloc_f, const unsigned char* __restrict__ fb,
                                                    const int* __restrict__ edof_c, int nloc_c, const unsigned char* __restrict__ cb, unsigned* __restrict__ fmask,
                                                    unsigned* __restrict__ cmask) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= nelc * 9) return;
  const int E = k / 9, j = k % 9;
  unsigned m = 0;
  if (j < 8) {
    const int ej = child[(size_t)E * 8 + j];
    for (int n = 0; n < 27; n++) m |= (fb[edof_f[(size_t)ej * nloc_f + n]] ? 1u : 0u) << n;
    fmask[(size_t)E * 8 + j] = m;
  } else {
    for (int n = 0; n < 27; n++) m |= (cb[edof_c[(size_t)E * nloc_c + n]] ? 1u : 0u) << n;
    cmask[E] = m;
  }
}

template <bool INS>
__global__ __launch_bounds__(GMAC_T) void k_galerkin_macro(int nelc, const int* __restrict__ child, int nm, const unsigned* __restrict__ sinfo, const uint4* __restrict__ map,
                                                           const uint4* __restrict__ mapb /* carried plans, else null */, const unsigned long long* __restrict__ vdst, const unsigned short* __restrict__ gtab,
                                                           const unsigned* __restrict__ fmask, const unsigned* __restrict__ cmask,
                                                           const int* __restrict__ slot_c, double* __restrict__ Kc, int ks_c, const double* __restrict__ Cdense /* [8][27][27] */,
                                                           unsigned long long* __restrict__ stamps) {
  constexpr int NC = 27, NCH = 8, NE = NC * NC, MT = 2, NK = GMAC_KP / 4;
  SfStamps st;
  if (INS) {
#pragma unroll
    for (int k = 0; k < 16; k++) st.acc[k] = 0;
  }
  extern __shared__ __attribute__((aligned(16))) double gmac_smem[];
  double* Tm = gmac_smem;                                 // the macro matrix in template order; behind the products: the eight results
  unsigned long long* rbl = reinterpret_cast<unsigned long long*>(gmac_smem + GMAC_BUF);
  int* slds = reinterpret_cast<int*>(rbl + CL_NM_MAX);    // slots of the coarse element's 27 rows
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kk = lane >> 4, li = lane & 15;
  unsigned si[CL_SPT];
#pragma unroll
  for (int i = 0; i < CL_SPT; i++) si[i] = sinfo[i * CL_T + tid];
  // loop-invariant fragments of child `wave`: C_j[4 m + kk][16 x + li] and the template index of K~_j[16 x + li][4 m + kk]
  double creg[NK][MT];
  int gidx[NK][MT];
#pragma unroll
  for (int m = 0; m < NK; m++)
#pragma unroll
    for (int x = 0; x < MT; x++) {
      const int r = 4 * m + kk, c = x * 16 + li;
      creg[m][x] = (r < NC && c < NC) ? Cdense[((size_t)wave * NC + r) * NC + c] : 0.0;
      const unsigned short g = (c < NC && r < NC) ? gtab[(size_t)wave * NE + c * NC + r] : (unsigned short)0xffff;
      gidx[m][x] = g == 0xffff ? -1 : (int)g;
    }
  typedef double d4 __attribute__((ext_vector_type(4)));
  const int stride = gridDim.x;
  int E = blockIdx.x;
  if (E >= nelc) return;
  // pipeline: values of cluster n in registers (tv), map / destinations of cluster n + 1 in registers (mp, vd)
  const int tm = tid & (CL_NM_MAX - 1);
  int cl = child[(size_t)E * NCH] >> 3;
  if (tid < CL_NM_MAX) rbl[tid] = vdst[(size_t)cl * CL_NM_MAX + tm];
  uint4 mp = map[(size_t)cl * CL_T + tid];
  __syncthreads();
  double tv[CL_SPT];
  // where slot i of a cluster's macro matrix is read: position p of the row (complete / partial rows), or -- a row carried in the CSR array across the clusters of a
  // super-cluster -- the CSR position the map byte names, and only in the cluster that made the LAST contribution (the entry holds the sum of all of them there;
  // the pseudo child matrices still add up to the assembled matrix, which is all the coarse operator depends on); -1: nothing to read
  auto slot_pos = [&](const uint4& q4, int i) {
    const unsigned mq[4] = {q4.x, q4.y, q4.z, q4.w};
    const int r = si[i] & 127, p = (si[i] >> 7) & 127;
    if (r >= nm) return -1;
    if (!mapb) return p;
    return ((mq[2] >> (16 + i)) & 1u) ? -1 : (int)((mq[i >> 2] >> (8 * (i & 3))) & 255);
  };
  const uint4 qzero = make_uint4(0u, 0u, 0u, 0u);
  uint4 mq = mapb ? mapb[(size_t)cl * CL_T + tid] : qzero;
#pragma unroll
  for (int i = 0; i < CL_SPT; i++) {
    const int r = si[i] & 127, pos = slot_pos(mq, i);
    tv[i] = pos >= 0 ? reinterpret_cast<const double*>(rbl[r])[pos] : 0.0;
  }
  int En = min(E + stride, nelc - 1);
  int cln = child[(size_t)En * NCH] >> 3;
  unsigned long long vdn = vdst[(size_t)cln * CL_NM_MAX + tm];
  uint4 mpn = map[(size_t)cln * CL_T + tid];
  uint4 mqn = mapb ? mapb[(size_t)cln * CL_T + tid] : qzero;
  const int ln = min(lane, NC - 1);
  unsigned dmask = fmask[(size_t)E * NCH + wave], cdm = cmask[E];      // Dirichlet nodes of this wave's child / of the coarse element (k_gal_masks)
  int slc = slot_c[(size_t)E * NC + ln];
  if (INS) st.prev = __builtin_amdgcn_s_memtime();
  int ndone = 0;
#pragma unroll 1
  for (; E < nelc; E += stride) {
    sf_stamp<INS>(st, 0);
    // ---- the macro matrix of this cluster into LDS (template order); destinations of the next cluster into rbl ----
    {
      const unsigned mw[4] = {mp.x, mp.y, mp.z, mp.w};
#pragma unroll
      for (int i = 0; i < CL_SPT; i++) {
        const int r = si[i] & 127, j = (mw[i >> 2] >> (8 * (i & 3))) & 255;
        if (r < nm) Tm[(si[i] >> 14) + j] = tv[i];
      }
    }
    if (tid < CL_NM_MAX) rbl[tid] = vdn;      // (the loads through the previous content were issued one iteration ago and have landed in tv)
    if (tid < NC) slds[tid] = slc;
    sf_stamp<INS>(st, 1);
    __syncthreads();
    sf_stamp<INS>(st, 2);
    // the NEXT cluster's values: in flight during the products below
    mp = mpn;
    mq = mqn;
#pragma unroll
    for (int i = 0; i < CL_SPT; i++) {
      const int r = si[i] & 127, pos = slot_pos(mq, i);
      tv[i] = pos >= 0 ? reinterpret_cast<const double*>(rbl[r])[pos] : 0.0;
    }
    {
      const int Enn = min(E + 2 * stride, nelc - 1);
      const int clnn = child[(size_t)Enn * NCH] >> 3;
      vdn = vdst[(size_t)clnn * CL_NM_MAX + tm];
      mpn = map[(size_t)clnn * CL_T + tid];
      if (mapb) mqn = mapb[(size_t)clnn * CL_T + tid];
    }
    const unsigned long long cdead = cdm, dead = dmask;
    {   // the next element's masks and slots (consumed one iteration later)
      const int En1 = min(E + stride, nelc - 1);
      dmask = fmask[(size_t)En1 * NCH + wave];
      cdm = cmask[En1];
      slc = slot_c[(size_t)En1 * NC + ln];
    }
    sf_stamp<INS>(st, 3);
    // ---- wave j: T = K~_j C_j (A operand gathered from the macro matrix, rows / columns of Dirichlet nodes zeroed), R_j = C_j^T T ----
    d4 T[MT][MT], KE[MT][MT];
#pragma unroll
    for (int a = 0; a < MT; a++)
#pragma unroll
      for (int b = 0; b < MT; b++) { T[a][b] = d4{0.0, 0.0, 0.0, 0.0}; KE[a][b] = d4{0.0, 0.0, 0.0, 0.0}; }
    {
      double av[2][MT];                      // fragments of K~_j, one k-step ahead of the products
      auto gather = [&](int m, double (&o)[MT]) {
#pragma unroll
        for (int x = 0; x < MT; x++) {
          const int row = x * 16 + li, col = 4 * m + kk;
          const bool d = gidx[m][x] < 0 || (((dead >> min(row, 63)) | (dead >> min(col, 63))) & 1ull);
          const double v = Tm[max(gidx[m][x], 0)];
          o[x] = d ? 0.0 : v;
        }
      };
      gather(0, av[0]);
#pragma unroll
      for (int m = 0; m < NK; m++) {
        if (m + 1 < NK) gather(m + 1, av[(m + 1) & 1]);
#pragma unroll
        for (int a = 0; a < MT; a++)
#pragma unroll
          for (int b = 0; b < MT; b++) T[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[m & 1][a], creg[m][b], T[a][b], 0, 0, 0);
      }
    }
#pragma unroll
    for (int m = 0; m < NK; m++)
#pragma unroll
      for (int a = 0; a < MT; a++)
#pragma unroll
        for (int b = 0; b < MT; b++) KE[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(creg[m][a], T[m / 4][b][m % 4], KE[a][b], 0, 0, 0);
    if (INS) {
      asm volatile("" ::"v"(KE[0][0][0]), "v"(KE[1][1][3]));
      sf_stamp<INS>(st, 4);
    }
    __syncthreads();                       // every wave is done with the macro matrix: the results take the region
    sf_stamp<INS>(st, 5);
    {
      double* Rs = Tm + wave * GMAC_RS;
#pragma unroll
      for (int a = 0; a < MT; a++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int ra = a * 16 + kk + 4 * r;
#pragma unroll
          for (int b = 0; b < MT; b++) {
            const int k = b * 16 + li;
            if (ra < NC && k < NC) Rs[ra * 28 + k] = KE[a][b][r];
          }
        }
    }
    sf_stamp<INS>(st, 6);
    __syncthreads();
    sf_stamp<INS>(st, 7);
    // ---- the eight results added in child order; rows / columns of coarse Dirichlet nodes are zero; rows without a slot are not stored ----
    for (int idx = tid; idx < NE; idx += GMAC_T) {
      const int i = idx / NC, k = idx - i * NC;
      double v = Tm[i * 28 + k];
#pragma unroll
      for (int w = 1; w < GMAC_NW; w++) v += Tm[w * GMAC_RS + i * 28 + k];
      const int s2 = slds[i];
      const bool d = ((cdead >> i) | (cdead >> k)) & 1ull;
      if (s2 >= 0) Kc[(size_t)s2 * ks_c + k] = d ? 0.0 : v;
    }
    sf_stamp<INS>(st, 8);
    __syncthreads();                       // the region is the next cluster's macro matrix
    sf_stamp<INS>(st, 9);
    ndone++;
  }
  if (INS && stamps && lane == 0) {
    unsigned long long* o = stamps + ((size_t)blockIdx.x * GMAC_NW + wave) * 20;
#pragma unroll
    for (int k = 0; k < 16; k++) o[k] = st.acc[k];
    o[16] = (unsigned long long)ndone;
  }
}

// element rows of the last assembly into the element-row buffer (pass 1 of the two-pass path alone)
static int element_rows_again(fh_assembler_t as) {
  FH_REQUIRE(as->two_pass && as->d_Kbuf, "fh_assembler_galerkin: the fine assembler holds no element rows");
  AsmParams P = base_params(as);
  P.sol = nullptr;          // only K_e is needed (the Poisson element matrix does not depend on the solution); the vector of that assembly may be gone by now
  P.source_kind = as->last_source_kind;
  P.p0 = as->last_params[0];
  P.p1 = as->last_params[1];
  P.elems = as->d_iota;
  P.nelems = as->nel;
  P.Kout = as->d_Kbuf;
  P.kstride = as->kstride;
  P.Fout = as->d_Fbuf;
  P.slot = as->d_slot;
  P.nsink = as->nadj;
  FH_TRY(dispatch_assemble(as, P));
  as->kbuf_valid = true;
  return 0;
}

extern "C" int fh_assembler_galerkin(fh_assembler_t fas, fh_assembler_t cas, const int* child, int nfb, const int* fbdc, int ncb, const int* cbdc, fh_mat_t Ac) {
  FH_GUARD_BEGIN
  FH_REQUIRE(fas && cas && child && Ac && (nfb == 0 || fbdc) && (ncb == 0 || cbdc), "fh_assembler_galerkin: null argument");
  FH_REQUIRE(fas->two_pass && cas->two_pass && fas->nc == cas->nc && fas->geom == cas->geom && (fas->nc == 27 || fas->nc == 9),
             "fh_assembler_galerkin: both levels need the two-pass biquadratic assembler");
  const int nch = fhfe::nvert_of(cas->geom), nc = cas->nc;
  FH_REQUIRE(fas->nel == cas->nel * nch, "fh_assembler_galerkin: %d fine elements are not the uniform refinement of %d coarse ones", fas->nel, cas->nel);
  FH_REQUIRE(Ac->m == cas->ndof, "fh_assembler_galerkin: the coarse matrix does not belong to the coarse assembler");
  fh_ctx_t c = cas->ctx;
  auto up = [&](void** d, const void* h, size_t bytes) -> int {
    if (*d) FH_CHECK_HIP(hipFree(*d));
    FH_CHECK_HIP(hipMalloc(d, bytes ? bytes : 8));
    if (bytes) FH_CHECK_HIP(hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice));
    return 0;
  };
  // integer / table set-up, once per hierarchy -- and again when the children or the Dirichlet sets differ from the ones the tables were made from
  uint64_t key = 1469598103934665603ull;
  {
    auto mix = [&](const int* a, size_t n) {
      key = (key ^ (uint64_t)n) * 1099511628211ull;
      for (size_t k = 0; k < n; k++) key = (key ^ (uint64_t)(uint32_t)a[k]) * 1099511628211ull;
    };
    mix(child, (size_t)cas->nel * nch);
    mix(fbdc, (size_t)nfb);
    mix(cbdc, (size_t)ncb);
  }
  if (!cas->d_gal_child || cas->gal_key != key) {
    cas->gal_key = key;
    FH_TRY(up((void**)&cas->d_gal_child, child, (size_t)cas->nel * nch * sizeof(int)));
    std::vector<unsigned char> cnt((size_t)nch * nc, 0), row((size_t)nch * nc * 8, 0);
    std::vector<double> val((size_t)nch * nc * 8, 0.0), phi(nc), dphi((size_t)nc * 3);
    for (int j = 0; j < nch; j++)
      for (int i = 0; i < nc; i++) {                      // fine node i of child j: C_j[i][k] = phi_k at its reference point in the parent
        double pt[3] = {0, 0, 0};
        fhfe::child_node_ref(cas->geom, j, i, pt);
        fhfe::eval_basis(cas->geom, fhfe::FE_BIQUADRATIC, pt, phi.data(), dphi.data());
        for (int k = 0; k < nc; k++)
          if (std::fabs(phi[k]) > 1e-14) {                // the threshold of ElemType.cpp:439-532
            unsigned char& n = cnt[(size_t)j * nc + k];
            FH_REQUIRE(n < 8, "fh_assembler_galerkin: more than 8 fine nodes of a child depend on one coarse node");
            row[((size_t)j * nc + k) * 8 + n] = (unsigned char)i;
            val[((size_t)j * nc + k) * 8 + n] = phi[k];
            n++;
          }
      }
    {
      std::vector<double> dense((size_t)nch * nc * nc, 0.0);
      for (int j = 0; j < nch; j++)
        for (int k = 0; k < nc; k++)
          for (int q = 0; q < cnt[(size_t)j * nc + k]; q++) dense[((size_t)j * nc + row[((size_t)j * nc + k) * 8 + q]) * nc + k] = val[((size_t)j * nc + k) * 8 + q];
      FH_TRY(up((void**)&cas->d_gal_dense, dense.data(), dense.size() * sizeof(double)));
    }
    FH_TRY(up((void**)&cas->d_gal_cnt, cnt.data(), cnt.size()));
    FH_TRY(up((void**)&cas->d_gal_row, row.data(), row.size()));
    FH_TRY(up((void**)&cas->d_gal_val, val.data(), val.size() * sizeof(double)));
    std::vector<unsigned char> fb(fas->nnode, 0), cb(cas->nnode, 0);
    for (int k = 0; k < nfb; k++) {
      FH_REQUIRE(fbdc[k] >= 0 && fbdc[k] < fas->nnode, "fh_assembler_galerkin: fine Dirichlet node %d out of range", fbdc[k]);
      fb[fbdc[k]] = 1;
    }
    for (int k = 0; k < ncb; k++) {
      FH_REQUIRE(cbdc[k] >= 0 && cbdc[k] < cas->nnode, "fh_assembler_galerkin: coarse Dirichlet node %d out of range", cbdc[k]);
      cb[cbdc[k]] = 1;
    }
    FH_TRY(up((void**)&cas->d_gal_fb, fb.data(), fb.size()));
    FH_TRY(up((void**)&cas->d_gal_cb, cb.data(), cb.size()));
    if (!cas->d_gal_res) FH_CHECK_HIP(hipMalloc(&cas->d_gal_res, std::max<size_t>(cas->ndof, 1) * sizeof(double)));
    if (nc == 27) {
      if (cas->d_gal_fmask) FH_CHECK_HIP(hipFree(cas->d_gal_fmask));
      cas->d_gal_fmask = nullptr;
      FH_CHECK_HIP(hipMalloc(&cas->d_gal_fmask, ((size_t)cas->nel * 9 + 1) * sizeof(unsigned)));
      hipLaunchKernelGGL(k_gal_masks, dim3(fh_div_up(cas->nel * 9, 256)), dim3(256), 0, c->stream, cas->nel, cas->d_gal_child, fas->d_elem_dof, fas->nloc, cas->d_gal_fb,
                         cas->d_elem_dof, cas->nloc, cas->d_gal_cb, cas->d_gal_fmask, cas->d_gal_fmask + (size_t)cas->nel * 8);
      FH_CHECK_HIP(hipGetLastError());
    }
    // the macro rows of a fused assembly can stand in for the element rows when child j of coarse element E is element j of ONE cluster (what the
    // refinement's numbering gives: MeshRefinement.cpp:240-294)
    cas->gal_children_in_order = nch == CL_NE;
    for (int E = 0; E < cas->nel && cas->gal_children_in_order; E++)
      for (int j = 0; j < nch; j++)
        if (child[(size_t)E * nch + j] != (child[(size_t)E * nch] / nch) * nch + j || child[(size_t)E * nch] % nch) { cas->gal_children_in_order = false; break; }
  }
  // source of the fine element contributions: the macro rows the fused assembly left in the fine matrix and the partial-row buffer (nothing to re-create,
  // the fused path stays the path of the next assembly), or the element-row buffer of the two-pass path
  // (the macro rows live in the user-visible fine matrix: any writer of its values since the assembly other than the Dirichlet-row replacement -- a scaling, an
  //  in-place product, staged adds -- sends the product back to the element rows, which are re-created from the arguments of the last assembly)
  const bool from_macro = nc == 27 && c->galerkin_mfma && c->galerkin_macro && fas->fused && fas->last_path == 1 && fas->macro_valid && fas->cl_all_rows &&
                          fas->d_cl_gtab && cas->gal_children_in_order && fas->cl_ncl == cas->nel && fas->cl_mat_of_macro != nullptr &&
                          fas->cl_mat_of_macro->d_val == fas->cl_val_base && fas->cl_mat_of_macro->val_gen == fas->macro_val_gen;
  if (!from_macro) {
    fas->rows_used_since = true;                              // (the next assembly of this level keeps its element rows)
    if (!fas->kbuf_valid) FH_TRY(element_rows_again(fas));     // the fused assembly kept no element rows: pass 1 of the two-pass path with the last arguments
  }
  const int grid = std::max(1, std::min(fh_div_up(cas->nel, 4), c->num_cu * 2));
  if (from_macro) {
    static bool attr_set_m[64] = {};
    if (!attr_set_m[c->device & 63]) {
      FH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_galerkin_macro<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)gmac_lds_bytes()));
      FH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_galerkin_macro<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)gmac_lds_bytes()));
      attr_set_m[c->device & 63] = true;
    }
    const int gm_grid = std::max(1, std::min(cas->nel, c->num_cu));          // (two workgroups per compute unit would need 128 registers per lane: 101 spilled)
    if (c->asm_debug & 128) {             // dev aid: phase cycles of every wave on stderr (as for the cluster kernel)
      unsigned long long* d_st = nullptr;
      const size_t nst = (size_t)gm_grid * GMAC_NW * 20;
      FH_CHECK_HIP(hipMalloc(&d_st, nst * sizeof(unsigned long long)));
      FH_CHECK_HIP(hipMemsetAsync(d_st, 0, nst * sizeof(unsigned long long), c->stream));
      hipLaunchKernelGGL(k_galerkin_macro<true>, dim3(gm_grid), dim3(GMAC_T), gmac_lds_bytes(), c->stream, cas->nel, cas->d_gal_child, fas->cl_nm, fas->d_cl_sinfo,
                         reinterpret_cast<const uint4*>(fas->d_cl_map), reinterpret_cast<const uint4*>(fas->d_cl_mapb), fas->d_cl_vdst64, fas->d_cl_gtab, cas->d_gal_fmask, cas->d_gal_fmask + (size_t)cas->nel * 8, cas->d_slot, cas->d_Kbuf,
                         cas->kstride, cas->d_gal_dense, d_st);
      std::vector<unsigned long long> h(nst);
      FH_CHECK_HIP(hipMemcpyAsync(h.data(), d_st, nst * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
      FH_CHECK_HIP(hipStreamSynchronize(c->stream));
      FH_CHECK_HIP(hipFree(d_st));
      static const char* name[10] = {"loop top", "macro matrix into LDS", "barrier", "next cluster's loads issued, look-ups", "K~ fragments gathered, two products (56 MFMA)",
                                     "barrier (all waves done with the macro matrix)", "result into LDS", "barrier", "eight results added, coarse rows stored", "barrier"};
      double sum[10] = {0}, ncl = 0, tot = 0;
      for (size_t w = 0; w < (size_t)gm_grid * GMAC_NW; w++) {
        for (int k = 0; k < 10; k++) sum[k] += (double)h[w * 20 + k];
        ncl += (double)h[w * 20 + 16];
      }
      for (int k = 0; k < 10; k++) tot += sum[k];
      fprintf(stderr, "k_galerkin_macro phase stamps: %.0f shader-clock ticks per cluster and wave\n", tot / std::max(ncl, 1.0));
      for (int k = 0; k < 10; k++) fprintf(stderr, "  %2d %-62s %9.1f  %5.1f %%\n", k, name[k], sum[k] / std::max(ncl, 1.0), 100.0 * sum[k] / std::max(tot, 1.0));
    } else
      hipLaunchKernelGGL(k_galerkin_macro<false>, dim3(gm_grid), dim3(GMAC_T), gmac_lds_bytes(), c->stream, cas->nel, cas->d_gal_child, fas->cl_nm, fas->d_cl_sinfo,
                         reinterpret_cast<const uint4*>(fas->d_cl_map), reinterpret_cast<const uint4*>(fas->d_cl_mapb), fas->d_cl_vdst64, fas->d_cl_gtab, cas->d_gal_fmask, cas->d_gal_fmask + (size_t)cas->nel * 8, cas->d_slot, cas->d_Kbuf,
                         cas->kstride, cas->d_gal_dense, (unsigned long long*)nullptr);
  } else if (c->galerkin_mfma) {
    const size_t lds = nc == 27 ? galerkin_mfma_lds<27, 8>() : galerkin_mfma_lds<9, 4>();
    static bool attr_set[64] = {};
    if (!attr_set[c->device & 63]) {
      FH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_galerkin_mfma<27, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)galerkin_mfma_lds<27, 8>()));
      FH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_galerkin_mfma<9, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)galerkin_mfma_lds<9, 4>()));
      attr_set[c->device & 63] = true;
    }
    const int g1 = std::max(1, std::min(fh_div_up(cas->nel, GAL_NW), c->num_cu));
    if (nc == 27)
      hipLaunchKernelGGL((k_galerkin_mfma<27, 8>), dim3(g1), dim3(GAL_NW * 64), lds, c->stream, cas->nel, cas->d_gal_child, fas->d_slot, fas->d_Kbuf, fas->kstride, fas->d_elem_dof,
                         fas->nloc, cas->d_gal_fb, cas->d_slot, cas->d_Kbuf, cas->kstride, cas->d_elem_dof, cas->nloc, cas->d_gal_cb, cas->d_gal_dense);
    else
      hipLaunchKernelGGL((k_galerkin_mfma<9, 4>), dim3(g1), dim3(GAL_NW * 64), lds, c->stream, cas->nel, cas->d_gal_child, fas->d_slot, fas->d_Kbuf, fas->kstride, fas->d_elem_dof,
                         fas->nloc, cas->d_gal_fb, cas->d_slot, cas->d_Kbuf, cas->kstride, cas->d_elem_dof, cas->nloc, cas->d_gal_cb, cas->d_gal_dense);
  } else if (nc == 27)
    hipLaunchKernelGGL((k_galerkin_elem<27, 8>), dim3(grid), dim3(256), 0, c->stream, cas->nel, cas->d_gal_child, fas->d_slot, fas->d_Kbuf, fas->kstride,
                       fas->d_elem_dof, fas->nloc, cas->d_gal_fb, cas->d_slot, cas->d_Kbuf, cas->kstride, cas->d_elem_dof, cas->nloc, cas->d_gal_cb, cas->d_gal_cnt,
                       cas->d_gal_row, cas->d_gal_val);
  else
    hipLaunchKernelGGL((k_galerkin_elem<9, 4>), dim3(grid), dim3(256), 0, c->stream, cas->nel, cas->d_gal_child, fas->d_slot, fas->d_Kbuf, fas->kstride,
                       fas->d_elem_dof, fas->nloc, cas->d_gal_fb, cas->d_slot, cas->d_Kbuf, cas->kstride, cas->d_elem_dof, cas->nloc, cas->d_gal_cb, cas->d_gal_cnt,
                       cas->d_gal_row, cas->d_gal_val);
  FH_CHECK_HIP(hipGetLastError());
  FH_CHECK_HIP(hipMemsetAsync(cas->d_Fbuf, 0, std::max<size_t>(cas->nadj, 1) * sizeof(double), c->stream));
  FH_TRY(dispatch_rows(cas, Ac, cas->d_gal_res, false));
  cas->kbuf_valid = true;     // the coarse element rows this product wrote (the next coarser product reads them)
  Ac->at_valid = false;       // new values: a cached explicit transpose is stale
  return 0;
  FH_GUARD_END("fh_assembler_galerkin")
}

// ------------------------------------------------------------------------------------------------------------------
// a4 in full: elem_type::Jacobian (ElemType.hpp:1183-1248 2-D, :1438-1537 3-D) for every (element, Gauss point) of a mesh, with the optional
// Hessians `nablaphi` (:1509-1534, :1232-1244): one thread per (element, Gauss point), the reference's accumulation order and bracketing.
// The Hessian formula is the reference's: JacI^T (reference Hessian) JacI, i.e. without the second derivatives of the map (exact on affine elements).
// ------------------------------------------------------------------------------------------------------------------
template <int DIM>
__global__ __launch_bounds__(128) void k_fe_jacobian(int nel, int ng, int nc, int nloc, const int* __restrict__ ed, const double* __restrict__ coords,
                                                     const double* __restrict__ w, const double* __restrict__ dphi, const double* __restrict__ d2phi,
                                                     double* __restrict__ weight, double* __restrict__ gradphi, double* __restrict__ nablaphi) {
  constexpr int NH = DIM == 2 ? 3 : 6;
  const size_t t = (size_t)blockIdx.x * 128 + threadIdx.x;
  if (t >= (size_t)nel * ng) return;
  const int e = (int)(t / ng), g = (int)(t % ng);
  const int* en = ed + (size_t)e * nloc;
  const double* dp = dphi + (size_t)g * nc * DIM;
  double J[DIM][DIM], I[DIM][DIM];
  for (int a = 0; a < DIM; a++)
    for (int b = 0; b < DIM; b++) J[a][b] = 0.0;
  for (int n = 0; n < nc; n++) {
    const double* x = coords + (size_t)en[n] * DIM;
    for (int a = 0; a < DIM; a++)
      for (int b = 0; b < DIM; b++) J[a][b] += dp[n * DIM + a] * x[b];
  }
  double det;
  if (DIM == 2) {
    det = J[0][0] * J[1][1] - J[0][1] * J[1][0];
    I[0][0] = J[1][1] / det;
    I[0][1] = -J[0][1] / det;
    I[1][0] = -J[1][0] / det;
    I[1][1] = J[0][0] / det;
  } else {
    det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) + J[0][1] * (J[1][2] * J[2][0] - J[1][0] * J[2][2]) + J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
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
  if (weight) weight[t] = det * w[g];
  for (int n = 0; n < nc; n++) {
    if (gradphi)
      for (int a = 0; a < DIM; a++) {
        double sum = dp[n * DIM + 0] * I[a][0];
        for (int b = 1; b < DIM; b++) sum += dp[n * DIM + b] * I[a][b];
        gradphi[(t * nc + n) * DIM + a] = sum;
      }
    if (nablaphi) {
      const double* h = d2phi + ((size_t)g * nc + n) * NH;
      double H[DIM][DIM];      // reference Hessian, symmetric
      if (DIM == 2) {
        H[0][0] = h[0]; H[1][1] = h[1]; H[0][1] = H[1][0] = h[2];
      } else {
        H[0][0] = h[0]; H[1][1] = h[1]; H[2][2] = h[2];
        H[0][1] = H[1][0] = h[3]; H[1][2] = H[2][1] = h[4]; H[0][2] = H[2][0] = h[5];
      }
      auto entry = [&](int a, int b) {
        double out = 0.0;
        for (int r = 0; r < DIM; r++) {
          double row = H[r][0] * I[a][0];
          for (int c2 = 1; c2 < DIM; c2++) row += H[r][c2] * I[a][c2];
          out += row * I[b][r];
        }
        return out;
      };
      double* o = nablaphi + (t * nc + n) * NH;
      if (DIM == 2) {
        o[0] = entry(0, 0); o[1] = entry(1, 1); o[2] = entry(0, 1);
      } else {
        o[0] = entry(0, 0); o[1] = entry(1, 1); o[2] = entry(2, 2);
        o[3] = entry(0, 1); o[4] = entry(1, 2); o[5] = entry(2, 0);
      }
    }
  }
}

extern "C" int fh_fe_tables_d2(int geom, int fe, int order, double* d2phi);

extern "C" int fh_fe_jacobian(fh_ctx_t ctx, int geom, int fe, int order, int nel, int nloc, const int* elem_dof, int nnode, const double* coords,
                              double* weight, double* gradphi, double* nablaphi) {
  FH_GUARD_BEGIN
  FH_REQUIRE(ctx && (nel == 0 || (elem_dof && coords)), "fh_fe_jacobian: null argument");
  FH_REQUIRE(geom == 0 || geom == 1, "fh_fe_jacobian: geom must be 0 (hex) or 1 (quad)");
  FH_REQUIRE(fe == 0 || fe == 1 || fe == 2, "fh_fe_jacobian: fe must be 0 (linear), 1 (serendipity) or 2 (biquadratic)");
  FH_REQUIRE(nloc == fhfe::nloc_of(geom), "fh_fe_jacobian: nloc %d does not match the geometry (%d)", nloc, fhfe::nloc_of(geom));
  if (nel == 0) return 0;
  const int dim = fhfe::dim_of(geom), nc = fhfe::ndofs_of(geom, fe), nh = dim == 2 ? 3 : 6;
  std::vector<double> w, phi, dphi;
  FH_REQUIRE(fhfe::shape_tables(geom, fe, order, w, phi, dphi) == 0, "fh_fe_jacobian: unsupported Gauss rule %d", order);
  const int ng = (int)w.size();
  for (size_t k = 0; k < (size_t)nel * nloc; k++) FH_REQUIRE(elem_dof[k] >= 0 && elem_dof[k] < nnode, "fh_fe_jacobian: node id %d out of range", elem_dof[k]);
  std::vector<double> d2((size_t)ng * nc * nh, 0.0);
  if (nablaphi) {
    std::vector<double> tab((size_t)nh * ng * nc);
    FH_TRY(fh_fe_tables_d2(geom, fe, order, tab.data()));
    for (int k = 0; k < nh; k++)
      for (int g = 0; g < ng; g++)
        for (int n = 0; n < nc; n++) d2[((size_t)g * nc + n) * nh + k] = tab[((size_t)k * ng + g) * nc + n];
  }
  struct Bufs {
    std::vector<void*> p;
    ~Bufs() { for (void* q : p) if (q) hipFree(q); }
  } B;
  auto dev = [&](void** d, const void* h, size_t bytes) -> int {
    FH_CHECK_HIP(hipMalloc(d, bytes ? bytes : 8));
    B.p.push_back(*d);
    if (h && bytes) FH_CHECK_HIP(hipMemcpyAsync(*d, h, bytes, hipMemcpyHostToDevice, ctx->stream));
    return 0;
  };
  const size_t npt = (size_t)nel * ng;
  int* d_ed;
  double *d_xy, *d_w, *d_dphi, *d_d2, *d_wt = nullptr, *d_g = nullptr, *d_n = nullptr;
  FH_TRY(dev((void**)&d_ed, elem_dof, (size_t)nel * nloc * sizeof(int)));
  FH_TRY(dev((void**)&d_xy, coords, (size_t)nnode * dim * sizeof(double)));
  FH_TRY(dev((void**)&d_w, w.data(), w.size() * sizeof(double)));
  FH_TRY(dev((void**)&d_dphi, dphi.data(), dphi.size() * sizeof(double)));
  FH_TRY(dev((void**)&d_d2, d2.data(), d2.size() * sizeof(double)));
  if (weight) FH_TRY(dev((void**)&d_wt, nullptr, npt * sizeof(double)));
  if (gradphi) FH_TRY(dev((void**)&d_g, nullptr, npt * nc * dim * sizeof(double)));
  if (nablaphi) FH_TRY(dev((void**)&d_n, nullptr, npt * nc * nh * sizeof(double)));
  const dim3 grid((unsigned)((npt + 127) / 128)), block(128);
  if (dim == 3) hipLaunchKernelGGL(k_fe_jacobian<3>, grid, block, 0, ctx->stream, nel, ng, nc, nloc, d_ed, d_xy, d_w, d_dphi, d_d2, d_wt, d_g, d_n);
  else hipLaunchKernelGGL(k_fe_jacobian<2>, grid, block, 0, ctx->stream, nel, ng, nc, nloc, d_ed, d_xy, d_w, d_dphi, d_d2, d_wt, d_g, d_n);
  FH_CHECK_HIP(hipGetLastError());
  if (weight) FH_CHECK_HIP(hipMemcpyAsync(weight, d_wt, npt * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  if (gradphi) FH_CHECK_HIP(hipMemcpyAsync(gradphi, d_g, npt * nc * dim * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  if (nablaphi) FH_CHECK_HIP(hipMemcpyAsync(nablaphi, d_n, npt * nc * nh * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  FH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
  FH_GUARD_END("fh_fe_jacobian")
}

// ------------------------------------------------------------------------------------------------------------------
// Neumann boundary faces (a5: elem_type::JacobianSur).  One thread per boundary node: it owns the node's (face, local i)
// pairs (ascending face order) and integrates phi_i * tau over each face with the face element's quadrature.
// ------------------------------------------------------------------------------------------------------------------
// NORMAL: the vector form  res[off[k] + node] += scale * int_face phi_i tau n_k ds  for the DIM components (open-boundary pressure term of the
// Navier-Stokes residual, 03_navier_stokes.hpp:185-290, normal = the one JacobianSur returns at each face Gauss point)
template <int DIM, bool NORMAL>
__global__ __launch_bounds__(128) void k_neumann(const int* __restrict__ node_ptr, const int* __restrict__ node_id, const int* __restrict__ pairs,
                                                 int nbn, const int* __restrict__ face_nodes, int nfn, const double* __restrict__ tau,
                                                 const double* __restrict__ coords, const double* __restrict__ w, const double* __restrict__ phi,
                                                 const double* __restrict__ dphi, int ng, double* __restrict__ res,
                                                 const int* __restrict__ face_expr, const int* __restrict__ prog, const int* __restrict__ prog_ptr,
                                                 const double* __restrict__ pconst, const int* __restrict__ const_ptr, int off0, int off1, int off2, double scale) {
  const int t = blockIdx.x * 128 + threadIdx.x;
  if (t >= nbn) return;
  double total = 0.0, totn[3] = {0.0, 0.0, 0.0};
  for (int p = node_ptr[t]; p < node_ptr[t + 1]; p++) {
    const int f = pairs[p] >> 4, i = pairs[p] & 15;
    const int* fn = face_nodes + (size_t)f * nfn;
    double acc = 0.0, accn[3] = {0.0, 0.0, 0.0};
    for (int g = 0; g < ng; g++) {
      double weight, nrm[3] = {0.0, 0.0, 0.0};
      if (DIM == 3) {   // quad face in 3-D: tangents, normal = t1 x t2, det = |normal|  (ElemType.hpp:1330-1380)
        double J[3][2] = {{0, 0}, {0, 0}, {0, 0}};
        for (int n = 0; n < nfn; n++) {
          const double dx = dphi[((size_t)g * nfn + n) * 2 + 0], dy = dphi[((size_t)g * nfn + n) * 2 + 1];
          const double* x = coords + (size_t)fn[n] * 3;
          for (int d = 0; d < 3; d++) {
            J[d][0] += dx * x[d];
            J[d][1] += dy * x[d];
          }
        }
        const double nx = J[1][0] * J[2][1] - J[1][1] * J[2][0];
        const double ny = J[0][1] * J[2][0] - J[2][1] * J[0][0];
        const double nz = J[0][0] * J[1][1] - J[0][1] * J[1][0];
        const double inv = 1.0 / sqrt(nx * nx + ny * ny + nz * nz);
        const double n0 = nx * inv, n1 = ny * inv, n2 = nz * inv;
        const double det = J[0][0] * (J[1][1] * n2 - n1 * J[2][1]) + J[0][1] * (n1 * J[2][0] - J[1][0] * n2) + n0 * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
        weight = det * w[g];
        nrm[0] = n0; nrm[1] = n1; nrm[2] = n2;
      } else {          // edge in 2-D (ElemType.hpp:1089-1138)
        double j0 = 0.0, j1 = 0.0;
        for (int n = 0; n < nfn; n++) {
          const double dx = dphi[(size_t)g * nfn + n];
          const double* x = coords + (size_t)fn[n] * 2;
          j0 += dx * x[0];
          j1 += dx * x[1];
        }
        const double modn = sqrt(j0 * j0 + j1 * j1);
        const double n0 = j1 / modn, n1 = -j0 / modn;
        const double det = j0 * (-n1) - (-n0) * j1;
        weight = det * w[g];
        nrm[0] = n0; nrm[1] = n1;
      }
      double tv;
      if (face_expr) {       // the flux is a parsed function of the Gauss point: (*bdcfunc)(&xyzt[0]), 001_Poisson/main.cpp:524-534
        double xg[4] = {0.0, 0.0, 0.0, 0.0};
        for (int n = 0; n < nfn; n++) {
          const double ph = phi[(size_t)g * nfn + n];
          const double* x = coords + (size_t)fn[n] * DIM;
          for (int d = 0; d < DIM; d++) xg[d] += x[d] * ph;
        }
        const int ex = face_expr[f];
        tv = fh_expr_device_eval(prog + prog_ptr[ex], prog_ptr[ex + 1] - prog_ptr[ex], pconst + const_ptr[ex], xg);
      } else {
        tv = tau[f];
      }
      if (NORMAL) {
#pragma unroll
        for (int k = 0; k < DIM; k++) accn[k] += phi[(size_t)g * nfn + i] * tv * nrm[k] * weight;
      } else {
        acc += phi[(size_t)g * nfn + i] * tv * weight;
      }
    }
    total += acc;
#pragma unroll
    for (int k = 0; k < DIM; k++) totn[k] += accn[k];
  }
  if (NORMAL) {
    const int off[3] = {off0, off1, off2};
#pragma unroll
    for (int k = 0; k < DIM; k++) res[off[k] + node_id[t]] += scale * totn[k];
  } else {
    res[node_id[t]] += total;
  }
}

// tables of the face element of `geom` (quad: the 2-D tables; line: 1-D Lagrange at the 1-D Gauss points): weights, phi[g][n], dphi[g][n][dim-1]
static int face_element_tables(int geom, int fe, int order, int* nfn_out, std::vector<double>& w, std::vector<double>& phi, std::vector<double>& dphi) {
  // geom >= 100: the FACE element itself is named (100 + its geometry: 101 quadrilateral, 103 triangle, 102 line) -- prisms have faces of two kinds
  const int fgeom = geom >= 100 ? geom - 100 : (geom == fhfe::GEOM_HEX) ? fhfe::GEOM_QUAD : (geom == fhfe::GEOM_TET) ? fhfe::GEOM_TRI : fhfe::GEOM_LINE;
  int tmp[9];
  const int nfn = geom >= 100 ? fhfe::ndofs_of(fgeom, fe) : fhfe::face_nodes(geom, fe, 0, tmp);
  *nfn_out = nfn;
  if (fgeom == fhfe::GEOM_TRI) {        // the faces of a tetrahedron: TRI3 / TRI6 with the triangle's rule of the same order
    FH_REQUIRE(fhfe::shape_tables(fhfe::GEOM_TRI, fe, order, w, phi, dphi) == 0, "fh_assemble_neumann_faces: unsupported Gauss rule");
  } else if (fgeom == fhfe::GEOM_QUAD) {
    FH_REQUIRE(fhfe::shape_tables(fhfe::GEOM_QUAD, fe, order, w, phi, dphi) == 0, "fh_assemble_neumann_faces: unsupported Gauss rule");
  } else {
    const int ng1 = order + 1;
    w.resize(ng1);
    std::vector<double> x1(ng1);
    FH_REQUIRE(fhfe::gauss_table(fhfe::GEOM_LINE, order, w.data(), x1.data()) == 0, "fh_assemble_neumann_faces: unsupported Gauss rule");
    phi.resize((size_t)ng1 * nfn);
    dphi.resize((size_t)ng1 * nfn);
    for (int g = 0; g < ng1; g++) {
      const double x = x1[g];
      if (fe == 0) {   // LineLinear: nodes -1, +1 (Edge.hpp:72-78)
        phi[g * 2 + 0] = 0.5 * (1. - x);  phi[g * 2 + 1] = 0.5 * (1. + x);
        dphi[g * 2 + 0] = -0.5;           dphi[g * 2 + 1] = 0.5;
      } else {         // LineBiquadratic: nodes -1, +1, 0 (Edge.hpp:94-100)
        phi[g * 3 + 0] = 0.5 * x * (x - 1.);  phi[g * 3 + 1] = 0.5 * x * (1. + x);  phi[g * 3 + 2] = (1. - x) * (1. + x);
        dphi[g * 3 + 0] = x - 0.5;            dphi[g * 3 + 1] = x + 0.5;            dphi[g * 3 + 2] = -2. * x;
      }
    }
  }
  return 0;
}

// unit normals of boundary faces at one face Gauss point, as elem_type::JacobianSur returns them (host; the applications read them to decide what a
// face contributes, e.g. 03_navier_stokes.hpp:264-275 picks the normal velocity component from the normal at Gauss point 0)
extern "C" int fh_fe_face_normals(int geom, int fe, int order, int gauss_point, int nfaces, const int* face_nodes, int nnode, const double* coords,
                                  double* normals /* [nfaces*dim] */) {
  FH_GUARD_BEGIN
  FH_REQUIRE(geom == 0 || geom == 1, "fh_fe_face_normals: geom must be 0 (hex) or 1 (quad)");
  FH_REQUIRE(fe == 0 || fe == 1 || fe == 2, "fh_fe_face_normals: fe must be 0, 1 or 2");
  FH_REQUIRE(nfaces == 0 || (face_nodes && coords && normals), "fh_fe_face_normals: null argument");
  const int dim = fhfe::dim_of(geom);
  int nfn = 0;
  std::vector<double> w, phi, dphi;
  FH_TRY(face_element_tables(geom, fe, order, &nfn, w, phi, dphi));
  const int g = gauss_point;
  FH_REQUIRE(g >= 0 && g < (int)w.size(), "fh_fe_face_normals: Gauss point %d of %d", g, (int)w.size());
  for (int f = 0; f < nfaces; f++) {
    const int* fn = face_nodes + (size_t)f * nfn;
    for (int n = 0; n < nfn; n++) FH_REQUIRE(fn[n] >= 0 && fn[n] < nnode, "fh_fe_face_normals: node id out of range");
    if (dim == 3) {
      double J[3][2] = {{0, 0}, {0, 0}, {0, 0}};
      for (int n = 0; n < nfn; n++) {
        const double dx = dphi[((size_t)g * nfn + n) * 2 + 0], dy = dphi[((size_t)g * nfn + n) * 2 + 1];
        const double* x = coords + (size_t)fn[n] * 3;
        for (int d = 0; d < 3; d++) {
          J[d][0] += dx * x[d];
          J[d][1] += dy * x[d];
        }
      }
      const double nx = J[1][0] * J[2][1] - J[1][1] * J[2][0];
      const double ny = J[0][1] * J[2][0] - J[2][1] * J[0][0];
      const double nz = J[0][0] * J[1][1] - J[0][1] * J[1][0];
      const double inv = 1.0 / sqrt(nx * nx + ny * ny + nz * nz);
      normals[(size_t)f * 3 + 0] = nx * inv;
      normals[(size_t)f * 3 + 1] = ny * inv;
      normals[(size_t)f * 3 + 2] = nz * inv;
    } else {
      double j0 = 0.0, j1 = 0.0;
      for (int n = 0; n < nfn; n++) {
        const double dx = dphi[(size_t)g * nfn + n];
        const double* x = coords + (size_t)fn[n] * 2;
        j0 += dx * x[0];
        j1 += dx * x[1];
      }
      const double modn = sqrt(j0 * j0 + j1 * j1);
      normals[(size_t)f * 2 + 0] = j1 / modn;
      normals[(size_t)f * 2 + 1] = -j0 / modn;
    }
  }
  return 0;
  FH_GUARD_END("fh_fe_face_normals")
}

static int neumann_faces(fh_ctx_t ctx, int geom, int fe, int order, int nfaces, const int* face_nodes, const double* tau, const int* face_expr, int nexpr,
                         const fh_expr_t* exprs, int nnode, const double* coords, fh_vec_t res, const int* comp_offset = nullptr, double scale = 1.0) {
  FH_REQUIRE(ctx && res && (nfaces == 0 || (face_nodes && (tau || face_expr) && coords)), "fh_assemble_neumann_faces: null argument");
  FH_REQUIRE(geom == 0 || geom == 1 || geom == 3 || geom == 4 || geom == 101 || geom == 102 || geom == 103,
             "fh_assemble_neumann_faces: geom must be 0 (hex), 1 (quad), 3 (triangle), 4 (tetrahedron), or 100 + the face element's own geometry (101 / 102 / 103)");
  FH_REQUIRE(fe == 0 || fe == 1 || fe == 2, "fh_assemble_neumann_faces: fe must be 0, 1 or 2");
  if (nfaces == 0) return 0;
  const int dim = geom >= 100 ? fhfe::dim_of(geom - 100) + 1 : fhfe::dim_of(geom);
  int nfn = 0;
  std::vector<double> w, phi, dphi;
  FH_TRY(face_element_tables(geom, fe, order, &nfn, w, phi, dphi));
  const int ng = (int)w.size();
  // node -> (face, i) pairs, ascending face order
  std::vector<int> cnt(nnode + 1, 0);
  for (size_t k = 0; k < (size_t)nfaces * nfn; k++) {
    FH_REQUIRE(face_nodes[k] >= 0 && face_nodes[k] < nnode, "fh_assemble_neumann_faces: node id out of range");
    cnt[face_nodes[k] + 1]++;
  }
  std::vector<int> node_id, node_ptr(1, 0), pairs;
  for (int n = 0; n < nnode; n++) cnt[n + 1] += cnt[n];
  std::vector<int> cur(cnt.begin(), cnt.end() - 1), flat(cnt[nnode]);
  FH_REQUIRE(nfaces < (1 << 27), "fh_assemble_neumann_faces: too many faces");
  for (int f = 0; f < nfaces; f++)
    for (int i = 0; i < nfn; i++) flat[cur[face_nodes[(size_t)f * nfn + i]]++] = (f << 4) | i;
  for (int n = 0; n < nnode; n++)
    if (cnt[n + 1] > cnt[n]) {
      node_id.push_back(n);
      pairs.insert(pairs.end(), flat.begin() + cnt[n], flat.begin() + cnt[n + 1]);
      node_ptr.push_back((int)pairs.size());
    }
  const int nbn = (int)node_id.size();
  FH_REQUIRE(res->n_local + res->nghost > node_id.back() + (comp_offset ? *std::max_element(comp_offset, comp_offset + dim) : 0),
             "fh_assemble_neumann_faces: vector too short");
  // parsed fluxes: the programs of all expressions back to back
  std::vector<int> h_prog, h_prog_ptr(1, 0), h_const_ptr(1, 0);
  std::vector<double> h_const;
  if (face_expr) {
    FH_REQUIRE(nexpr >= 1 && exprs, "fh_assemble_neumann_faces_expr: no expressions");
    for (int f = 0; f < nfaces; f++) FH_REQUIRE(face_expr[f] >= 0 && face_expr[f] < nexpr, "fh_assemble_neumann_faces_expr: face %d names expression %d of %d", f, face_expr[f], nexpr);
    for (int k = 0; k < nexpr; k++) {
      FH_REQUIRE(exprs[k], "fh_assemble_neumann_faces_expr: null expression");
      int nv = 0, nc = 0, nk = 0;
      FH_TRY(fh_expr_nvars(exprs[k], &nv));
      FH_REQUIRE(nv <= 4, "fh_assemble_neumann_faces_expr: expression %d has %d variables, at most 4 (x, y, z, t) are served", k, nv);
      FH_TRY(fh_expr_program(exprs[k], &nc, &nk, nullptr, nullptr));
      std::vector<int> code(nc);
      std::vector<double> consts(nk);
      FH_TRY(fh_expr_program(exprs[k], &nc, &nk, code.data(), consts.data()));
      h_prog.insert(h_prog.end(), code.begin(), code.end());
      h_const.insert(h_const.end(), consts.begin(), consts.end());
      h_prog_ptr.push_back((int)h_prog.size());
      h_const_ptr.push_back((int)h_const.size());
    }
  }
  void* dv[13] = {nullptr};
  auto up = [&](int slot, const void* h, size_t bytes) -> int {
    FH_CHECK_HIP(hipMalloc(&dv[slot], bytes ? bytes : 8));
    FH_CHECK_HIP(hipMemcpyAsync(dv[slot], h, bytes, hipMemcpyHostToDevice, ctx->stream));
    return 0;
  };
  FH_TRY(up(0, node_ptr.data(), node_ptr.size() * sizeof(int)));
  FH_TRY(up(1, node_id.data(), node_id.size() * sizeof(int)));
  FH_TRY(up(2, pairs.data(), pairs.size() * sizeof(int)));
  FH_TRY(up(3, face_nodes, (size_t)nfaces * nfn * sizeof(int)));
  if (tau) FH_TRY(up(4, tau, (size_t)nfaces * sizeof(double)));
  if (face_expr) {
    FH_TRY(up(8, face_expr, (size_t)nfaces * sizeof(int)));
    FH_TRY(up(9, h_prog.data(), h_prog.size() * sizeof(int)));
    FH_TRY(up(10, h_prog_ptr.data(), h_prog_ptr.size() * sizeof(int)));
    FH_TRY(up(11, h_const.data(), h_const.size() * sizeof(double)));
    FH_TRY(up(12, h_const_ptr.data(), h_const_ptr.size() * sizeof(int)));
  }
  FH_TRY(up(5, coords, (size_t)nnode * dim * sizeof(double)));
  FH_TRY(up(6, w.data(), w.size() * sizeof(double)));
  std::vector<double> tab(phi);
  tab.insert(tab.end(), dphi.begin(), dphi.end());
  FH_TRY(up(7, tab.data(), tab.size() * sizeof(double)));
  const double* d_phi = (const double*)dv[7];
  const double* d_dphi = d_phi + phi.size();
  const dim3 grid(fh_div_up(nbn, 128)), block(128);
  const int o0 = comp_offset ? comp_offset[0] : 0, o1 = comp_offset ? comp_offset[1] : 0, o2 = (comp_offset && dim == 3) ? comp_offset[2] : 0;
#define FH_NEUMANN_LAUNCH(D, N)                                                                                                                   \
  hipLaunchKernelGGL((k_neumann<D, N>), grid, block, 0, ctx->stream, (const int*)dv[0], (const int*)dv[1], (const int*)dv[2], nbn, (const int*)dv[3], \
                     nfn, (const double*)dv[4], (const double*)dv[5], (const double*)dv[6], d_phi, d_dphi, ng, res->d, (const int*)dv[8],          \
                     (const int*)dv[9], (const int*)dv[10], (const double*)dv[11], (const int*)dv[12], o0, o1, o2, scale)
  if (dim == 3 && comp_offset) FH_NEUMANN_LAUNCH(3, true);
  else if (dim == 3) FH_NEUMANN_LAUNCH(3, false);
  else if (comp_offset) FH_NEUMANN_LAUNCH(2, true);
  else FH_NEUMANN_LAUNCH(2, false);
#undef FH_NEUMANN_LAUNCH
  FH_CHECK_HIP(hipGetLastError());
  FH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  for (void* q : dv)
    if (q) hipFree(q);
  return 0;
}

extern "C" int fh_assemble_neumann_faces(fh_ctx_t ctx, int geom, int fe, int order, int nfaces, const int* face_nodes, const double* tau, int nnode,
                                         const double* coords, fh_vec_t res) {
  FH_REQUIRE(nfaces == 0 || tau, "fh_assemble_neumann_faces: null argument");
  return neumann_faces(ctx, geom, fe, order, nfaces, face_nodes, tau, nullptr, 0, nullptr, nnode, coords, res);
}

// Open-boundary pressure term of the steady Navier-Stokes residual (03_navier_stokes.hpp:185-290): on the listed boundary faces (those whose
// normal velocity component is not Dirichlet -- the application's bdc callback decides, :236-262) aResV[k][node_i] += phi_i tau n_k weight with the
// prescribed pressure tau (a number per face, or expression face_expr[f] at the face Gauss point) and the JacobianSur normal; the residual
// vector takes scale * that (scale = -1: RES = -aRes, :425).  comp_offset[k]: where component k of the velocity starts in res.
extern "C" int fh_assemble_pressure_faces(fh_ctx_t ctx, int geom, int order, int nfaces, const int* face_nodes, const double* tau, const int* face_expr,
                                          int nexpr, const fh_expr_t* exprs, int nnode, const double* coords, const int* comp_offset, double scale,
                                          fh_vec_t res) {
  FH_REQUIRE(comp_offset, "fh_assemble_pressure_faces: null component offsets");
  FH_REQUIRE(nfaces == 0 || tau || face_expr, "fh_assemble_pressure_faces: neither a pressure per face nor expressions");
  return neumann_faces(ctx, geom, 2, order, nfaces, face_nodes, face_expr ? nullptr : tau, face_expr, nexpr, exprs, nnode, coords, res, comp_offset, scale);
}

// the flux as a parsed function of the Gauss point (x, y, z, t = 0), as the parsed-boundary-condition branch of the 001_Poisson callback
// evaluates it (`(*bdcfunc)(&xyzt[0])` inside the Gauss loop, applications/001_Poisson/main.cpp:495-553): face_expr[f] names one of `nexpr` expressions
extern "C" int fh_assemble_neumann_faces_expr(fh_ctx_t ctx, int geom, int fe, int order, int nfaces, const int* face_nodes, const int* face_expr, int nexpr,
                                              const fh_expr_t* exprs, int nnode, const double* coords, fh_vec_t res) {
  FH_REQUIRE(nfaces == 0 || face_expr, "fh_assemble_neumann_faces_expr: null argument");
  return neumann_faces(ctx, geom, fe, order, nfaces, face_nodes, nullptr, face_expr, nexpr, exprs, nnode, coords, res);
}

// ------------------------------------------------------------------------------------------------------------------
// The application's callback on a ONE-DIMENSIONAL mesh (applications/001_Poisson/main.cpp:355-480 with dim == 1; its shipped input/input1D.json, an EDGE3
// box): there the callback is not a Poisson problem -- main.cpp:392-395 sets V = 1, nu = 0.01 -- but advection-diffusion with the streamline-upwind terms the
// same loop carries in every dimension (tau = 0 where V = 0, which is why the 2-D / 3-D kernels above never see them):
//   tau   = barNu / V^2,  barNu = (coth(Pe) - 1 / Pe) V h / 2,  Pe = V h / (2 nu),  h = x[1] - x[0]  (directions_of_reference_element, Elem.hpp:149-167)
//   F_i  += (f phi_i - nu phi_i' u' - V u' phi_i + (f - (-nu u'' + V u')) s_i) w,   s_i = (V phi_i' + nu phi_i'') tau
//   B_ij += (nu (phi_i' phi_j' - phi_j'' s_i) + V phi_j' (phi_i + s_i)) w
// with elem_type_1D::Jacobian (ElemType.hpp:994-1035): Jac = sum dphi_n x_n, w = Jac w_g, phi' = dphi / Jac, phi'' = d2phi / Jac^2.
// One thread per ROW: it walks the elements of its node in ascending order, forms its row of each element matrix over the Gauss points and adds it -- the
// grouping of the reference's add_matrix_blocked / add_vector_blocked, no atomics.  The problem sizes of a one-dimensional mesh make everything else moot.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_line_advdiff(int ndof, int nc, int ng, const int* __restrict__ adj_ptr, const int* __restrict__ adj,
                                                     const int* __restrict__ elem_dof, const double* __restrict__ coords, const double* __restrict__ sol,
                                                     const double* __restrict__ w, const double* __restrict__ phi, const double* __restrict__ dphi,
                                                     const double* __restrict__ d2phi, double nu, double V, const int* __restrict__ prog, int nprog,
                                                     const double* __restrict__ pconst, const int* __restrict__ rowptr, const int* __restrict__ col,
                                                     double* __restrict__ val, double* __restrict__ res) {
  const int r = blockIdx.x * 64 + threadIdx.x;
  if (r >= ndof) return;
  const int rs = rowptr[r], re = rowptr[r + 1];
  for (int k = rs; k < re; k++) val[k] = 0.0;
  double racc = 0.0;
  for (int a = adj_ptr[r]; a < adj_ptr[r + 1]; a++) {
    const int e = adj[a] >> 2, i = adj[a] & 3;
    double x[3], u[3];
    int dof[3];
    for (int n = 0; n < nc; n++) {
      dof[n] = elem_dof[e * 3 + n];
      x[n] = coords[dof[n]];
      u[n] = sol ? sol[dof[n]] : 0.0;
    }
    // stabilisation parameter of the element (main.cpp:397-428)
    const double VxiHxi = (x[1] - x[0]) * V;
    const double PeXi = VxiHxi / (2. * nu);
    const double barXi = (fabs(PeXi) < 1.0e-10) ? 0. : 1. / tanh(PeXi) - 1. / PeXi;
    const double barNu = barXi * VxiHxi / 2.;
    const double vL2Norm2 = V * V;
    const double supgTau = (vL2Norm2 > 1.0e-15) ? barNu / vL2Norm2 : 0.;
    double F = 0.0, B[3] = {0.0, 0.0, 0.0};
    for (int g = 0; g < ng; g++) {
      double Jac = 0.0;
      for (int n = 0; n < nc; n++) Jac += dphi[g * nc + n] * x[n];
      const double weight = Jac * w[g], JacI = 1 / Jac;
      double ph[3], gr[3], nb[3], gradSol = 0.0, nablaSol = 0.0, xg[4] = {0.0, 0.0, 0.0, 0.0};
      for (int n = 0; n < nc; n++) {
        ph[n] = phi[g * nc + n];
        gr[n] = dphi[g * nc + n] * JacI;
        nb[n] = d2phi[g * nc + n] * JacI * JacI;
        xg[0] += x[n] * ph[n];
        gradSol += gr[n] * u[n];
        nablaSol += nb[n] * u[n];
      }
      const double lapRhs = nu * gr[i] * gradSol;
      const double advRhs = V * gradSol * ph[i];
      const double resRhs = -nu * nablaSol + V * gradSol;
      const double supgPhi = (V * gr[i] + nu * nb[i]) * supgTau;
      const double src = prog ? fh_expr_device_eval(prog, nprog, pconst, xg) : 0.0;
      F += (src * ph[i] - lapRhs - advRhs + (src - resRhs) * supgPhi) * weight;
      for (int j = 0; j < nc; j++) {
        const double lap = nu * (gr[i] * gr[j] - nb[j] * supgPhi) * weight;
        const double adv = V * gr[j] * (ph[i] + supgPhi) * weight;
        B[j] += lap + adv;
      }
    }
    racc += F;
    for (int j = 0; j < nc; j++)
      for (int k = rs; k < re; k++)
        if (col[k] == dof[j]) {
          val[k] += B[j];
          break;
        }
  }
  res[r] = racc;
}

extern "C" int fh_assemble_advdiff_line(fh_ctx_t ctx, int fe, int order, int nel, const int* elem_dof, int nnode, const double* coords, fh_vec_t sol, double nu,
                                        double velocity, fh_expr_t source, fh_mat_t KK, fh_vec_t RES) {
  FH_GUARD_BEGIN
  FH_REQUIRE(ctx && elem_dof && coords && KK && RES && nel >= 1 && nnode >= 2, "fh_assemble_advdiff_line: null or empty argument");
  FH_REQUIRE(fe == fhfe::FE_LINEAR || fe == fhfe::FE_SERENDIPITY || fe == fhfe::FE_BIQUADRATIC, "fh_assemble_advdiff_line: fe must be 0, 1 or 2");
  FH_REQUIRE(nu > 0.0, "fh_assemble_advdiff_line: the diffusivity must be positive");
  const int nc = fhfe::ndofs_of(fhfe::GEOM_LINE, fe), ndof = KK->m;
  FH_REQUIRE(KK->n == ndof && RES->n_local >= ndof && (!sol || sol->n_local >= ndof), "fh_assemble_advdiff_line: size mismatch");
  std::vector<int> cnt(ndof + 1, 0);
  for (int e = 0; e < nel; e++)
    for (int n = 0; n < nc; n++) {
      const int d = elem_dof[e * 3 + n];
      FH_REQUIRE(d >= 0 && d < ndof && d < nnode, "fh_assemble_advdiff_line: element %d, node %d: dof %d outside the system (the vertices are numbered first)", e, n, d);
      cnt[d + 1]++;
    }
  for (int d = 0; d < ndof; d++) cnt[d + 1] += cnt[d];
  std::vector<int> adj(cnt[ndof]), fill(cnt.begin(), cnt.end() - 1);
  for (int e = 0; e < nel; e++)                         // ascending element order per dof
    for (int n = 0; n < nc; n++) adj[fill[elem_dof[e * 3 + n]]++] = e * 4 + n;
  std::vector<double> w, phi, dphi;
  FH_REQUIRE(fhfe::shape_tables(fhfe::GEOM_LINE, fe, order, w, phi, dphi) == 0, "fh_assemble_advdiff_line: unsupported Gauss rule");
  const int ng = (int)w.size();
  std::vector<double> d2((size_t)ng * nc), x1(ng), t(nc);
  fhfe::gauss_table(fhfe::GEOM_LINE, order, nullptr, x1.data());
  for (int g = 0; g < ng; g++) {
    const double pt[3] = {x1[g], 0.0, 0.0};
    fhfe::eval_basis_d2(fhfe::GEOM_LINE, fe, pt, t.data());
    for (int n = 0; n < nc; n++) d2[(size_t)g * nc + n] = t[n];
  }
  std::vector<int> code;
  std::vector<double> consts;
  if (source) {
    int nv = 0, ncode = 0, nk = 0;
    FH_TRY(fh_expr_nvars(source, &nv));
    FH_REQUIRE(nv <= 4, "fh_assemble_advdiff_line: the source expression has %d variables, at most 4 (x, y, z, t) are served", nv);
    FH_TRY(fh_expr_program(source, &ncode, &nk, nullptr, nullptr));
    code.resize(ncode);
    consts.resize(std::max(nk, 1));
    FH_TRY(fh_expr_program(source, &ncode, &nk, code.data(), consts.data()));
  }
  hipStream_t st = ctx->stream;
  std::vector<void*> dv;
  auto up = [&](const void* h, size_t bytes) -> void* {
    void* d = nullptr;
    if (hipMalloc(&d, std::max<size_t>(bytes, 8)) != hipSuccess) return nullptr;
    dv.push_back(d);
    if (bytes && h) hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st);
    return d;
  };
  int* d_ptr = (int*)up(cnt.data(), cnt.size() * sizeof(int));
  int* d_adj = (int*)up(adj.data(), adj.size() * sizeof(int));
  int* d_ed = (int*)up(elem_dof, (size_t)nel * 3 * sizeof(int));
  double* d_x = (double*)up(coords, (size_t)nnode * sizeof(double));
  double* d_w = (double*)up(w.data(), w.size() * sizeof(double));
  double* d_phi = (double*)up(phi.data(), phi.size() * sizeof(double));
  double* d_dphi = (double*)up(dphi.data(), dphi.size() * sizeof(double));
  double* d_d2 = (double*)up(d2.data(), d2.size() * sizeof(double));
  int* d_code = source ? (int*)up(code.data(), code.size() * sizeof(int)) : nullptr;
  double* d_k = source ? (double*)up(consts.data(), consts.size() * sizeof(double)) : nullptr;
  int rc = 0;
  if (!d_ptr || !d_adj || !d_ed || !d_x || !d_w || !d_phi || !d_dphi || !d_d2 || (source && (!d_code || !d_k))) {
    fh_set_error("fh_assemble_advdiff_line: out of device memory");
    rc = 2;
  } else {
    hipLaunchKernelGGL(k_line_advdiff, dim3(fh_div_up(ndof, 64)), dim3(64), 0, st, ndof, nc, ng, d_ptr, d_adj, d_ed, d_x, sol ? sol->d : nullptr, d_w, d_phi, d_dphi,
                       d_d2, nu, velocity, d_code, (int)code.size(), d_k, KK->d_rowptr, KK->d_col, KK->d_val, RES->d);
    if (hipGetLastError() != hipSuccess) {
      fh_set_error("fh_assemble_advdiff_line: launch failed");
      rc = 2;
    }
    KK->val_gen++;
    KK->at_valid = false;
  }
  hipStreamSynchronize(st);
  for (void* q : dv) hipFree(q);
  return rc;
  FH_GUARD_END("fh_assemble_advdiff_line")
}

// ------------------------------------------------------------------------------------------------------------------
// The Poisson callback through a GENERIC (dim, nc, ng) kernel (round 6): any element family fh_fe has tables for -- the triangle (geom 3) first, whose meshes
// do not go through the tensor-product mesh layer -- with the element table given by the caller (nloc nodes per element in the family's local order; dof id =
// node id, the classes numbered one after the other as every FEMuS mesh is).  Two passes: one wave per element forms the element
// matrix over the Gauss points (elem_type::Jacobian: Jac[a][b] = sum_n dphi_n/dxi_a x_n[b], grad phi_n = Jac^-1 dphi_n, w = det w_g) into a buffer, with the place of every entry in the matrix beside it; one thread per
// ROW then adds the rows of its node's elements in ascending element order (first version: the row thread formed them itself -- 80 ms per call on 54 k TET15 elements, host preparation included):
//   K_ij += grad phi_i . grad phi_j w,   RES_i += (scale f phi_i - grad phi_i . grad u) w        (main.cpp:430-470 with V = 0)
// The grouping of the reference's add_matrix_blocked / add_vector_blocked, no atomics; meant for the sizes such meshes have here, not for the bench (the
// hexahedral paths above are the fast ones).
// ------------------------------------------------------------------------------------------------------------------
constexpr int GEN_NC = 27;
struct GenTab {              // the tables of one element shape: a mesh of mixed shapes (hexahedra, tetrahedra, prisms; quadrilaterals, triangles) names one per element
  int nc, ng;
  const double *w, *phi, *dphi;
};
struct GenTabs {
  GenTab t[3];
};
// First pass: one WAVE per element.  The Gauss points are taken GEN_GC at a time through LDS: (A) lane = Gauss point: Jacobian, its inverse, weight, source value;
// (B) lanes over (Gauss point, node): the node's gradient; (C) lane = Gauss point: grad u; (D) lanes over the pairs i <= j of the element matrix (K_ji = K_ij
// bit for bit: the products commute) and over the residual entries, Gauss points in ascending order.  Every sum is taken in the order of the one-thread-per-row
// kernel this replaces (nodes ascending inside a Gauss point, Gauss points ascending); 53 760 TET15 elements: 26 + 8.5 ms (element rows by one thread each +
// searching row pass) -> 4.7 + 0.8 ms (profiles/r06_shipped_inputs_kernel_summary.md).
// Row i of the element matrix goes to Kb[(e * ncmax + i) * ncmax + j], its residual entry to Fb[e * ncmax + i].
constexpr int GEN_GC = 32;
constexpr int GEN_LDS = GEN_NC * 3 + GEN_NC + GEN_GC * 9 + GEN_GC * 2 + GEN_GC * 3 + GEN_GC * GEN_NC * 3;
__global__ __launch_bounds__(64) void k_poisson_pairs_generic(int nel, int ncmax, int dim, GenTabs tabs, const unsigned char* __restrict__ etab, int nloc,
                                                              const int* __restrict__ elem_dof, const double* __restrict__ coords, const double* __restrict__ sol,
                                                              double scale, const int* __restrict__ prog, int nprog, const double* __restrict__ pconst,
                                                              const int* __restrict__ rowptr, const int* __restrict__ col, double* __restrict__ Kb,
                                                              int* __restrict__ Pos, double* __restrict__ Fb) {
  __shared__ double S[GEN_LDS];
  __shared__ int DOF[GEN_NC];
  double* X = S;                          // [nc][3]
  double* U = X + GEN_NC * 3;             // [nc]
  double* JI = U + GEN_NC;                // [GC][9]
  double* WG = JI + GEN_GC * 9;           // [GC] det w
  double* FS = WG + GEN_GC;               // [GC] scale f(x_g)
  double* GU = FS + GEN_GC;               // [GC][3]
  double* G = GU + GEN_GC * 3;            // [GC][nc][3]
  const int e = blockIdx.x, lane = threadIdx.x;
  const GenTab& T = tabs.t[etab ? etab[e] : 0];
  const int nc = T.nc, ng = T.ng;
  const double *w = T.w, *phi = T.phi, *dphi = T.dphi;
  if (lane < nc) {
    const int dof = elem_dof[(size_t)e * nloc + lane];
    DOF[lane] = dof;
    for (int d = 0; d < 3; d++) X[lane * 3 + d] = d < dim ? coords[(size_t)dof * dim + d] : 0.0;
    U[lane] = sol ? sol[dof] : 0.0;
  }
  // this lane's pairs (i <= j), GEN_NC (GEN_NC + 1) / 2 = 378 at most: six per lane
  constexpr int NPL = (GEN_NC * (GEN_NC + 1) / 2 + 63) / 64;
  const int npair = nc * (nc + 1) / 2;
  int pi[NPL], pj[NPL];
  double acc[NPL];
#pragma unroll
  for (int k = 0; k < NPL; k++) {
    int p = lane + 64 * k, i = 0;
    if (p < npair) {
      while (p >= nc - i) {
        p -= nc - i;
        i++;
      }
      pi[k] = i;
      pj[k] = i + p;
    } else {
      pi[k] = pj[k] = -1;
    }
    acc[k] = 0.0;
  }
  double F = 0.0;
  __syncthreads();
  for (int g0 = 0; g0 < ng; g0 += GEN_GC) {
    const int gc = min(GEN_GC, ng - g0);
    if (lane < gc) {                      // (A)
      const int g = g0 + lane;
      const double* dp = dphi + (size_t)g * nc * dim;
      double J[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, Ji[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, det;
      for (int n = 0; n < nc; n++)
        for (int p = 0; p < dim; p++)
          for (int q = 0; q < dim; q++) J[p][q] += dp[n * dim + p] * X[n * 3 + q];
      if (dim == 1) {
        det = J[0][0];
        Ji[0][0] = 1 / det;
      } else if (dim == 2) {
        det = J[0][0] * J[1][1] - J[0][1] * J[1][0];
        Ji[0][0] = J[1][1] / det;
        Ji[0][1] = -J[0][1] / det;
        Ji[1][0] = -J[1][0] / det;
        Ji[1][1] = J[0][0] / det;
      } else {
        det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) + J[0][1] * (J[1][2] * J[2][0] - J[1][0] * J[2][2]) + J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
        Ji[0][0] = (-J[1][2] * J[2][1] + J[1][1] * J[2][2]) / det;
        Ji[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) / det;
        Ji[0][2] = (-J[0][2] * J[1][1] + J[0][1] * J[1][2]) / det;
        Ji[1][0] = (J[1][2] * J[2][0] - J[1][0] * J[2][2]) / det;
        Ji[1][1] = (-J[0][2] * J[2][0] + J[0][0] * J[2][2]) / det;
        Ji[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) / det;
        Ji[2][0] = (-J[1][1] * J[2][0] + J[1][0] * J[2][1]) / det;
        Ji[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) / det;
        Ji[2][2] = (-J[0][1] * J[1][0] + J[0][0] * J[1][1]) / det;
      }
      for (int q = 0; q < 3; q++)
        for (int p = 0; p < 3; p++) JI[lane * 9 + q * 3 + p] = Ji[q][p];
      WG[lane] = det * w[g];
      double xq[4] = {0, 0, 0, 0};
      for (int n = 0; n < nc; n++) {
        const double ph = phi[(size_t)g * nc + n];
        for (int q = 0; q < dim; q++) xq[q] += X[n * 3 + q] * ph;
      }
      FS[lane] = prog ? scale * fh_expr_device_eval(prog, nprog, pconst, xq) : 0.0;
    }
    __syncthreads();
    for (int t = lane; t < gc * nc; t += 64) {                      // (B)
      const int l = t / nc, n = t - l * nc;
      const double* dp = dphi + ((size_t)(g0 + l) * nc + n) * dim;
      for (int q = 0; q < dim; q++) {
        double sacc = 0.0;
        for (int p = 0; p < dim; p++) sacc += JI[l * 9 + q * 3 + p] * dp[p];
        G[(l * GEN_NC + n) * 3 + q] = sacc;
      }
    }
    __syncthreads();
    if (lane < gc) {                      // (C)
      double gu[3] = {0, 0, 0};
      for (int n = 0; n < nc; n++)
        for (int q = 0; q < dim; q++) gu[q] += G[(lane * GEN_NC + n) * 3 + q] * U[n];
      for (int q = 0; q < 3; q++) GU[lane * 3 + q] = gu[q];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NPL; k++)         // (D)
      if (pi[k] >= 0) {
        double a = acc[k];
        for (int l = 0; l < gc; l++) {
          const double *gi = G + (l * GEN_NC + pi[k]) * 3, *gj = G + (l * GEN_NC + pj[k]) * 3;
          double sacc = 0.0;
          for (int q = 0; q < dim; q++) sacc += gi[q] * gj[q];
          a += sacc * WG[l];
        }
        acc[k] = a;
      }
    if (lane < nc)
      for (int l = 0; l < gc; l++) {
        const double* gi = G + (l * GEN_NC + lane) * 3;
        double lap = 0.0;
        for (int q = 0; q < dim; q++) lap += gi[q] * GU[l * 3 + q];
        F += (FS[l] * phi[(size_t)(g0 + l) * nc + lane] - lap) * WG[l];
      }
    __syncthreads();
  }
  // the entry's place in the matrix beside its value (-1: the pattern does not hold it), so that the row pass adds without searching
  double* out = Kb + (size_t)e * ncmax * ncmax;
  int* pos = Pos + (size_t)e * ncmax * ncmax;
  auto place = [&](int i, int j) {
    const int r = DOF[i], c = DOF[j];
    int at = -1;
    for (int k = rowptr[r], re = rowptr[r + 1]; k < re; k++)
      if (col[k] == c) {
        at = k;
        break;
      }
    return at;
  };
#pragma unroll
  for (int k = 0; k < NPL; k++)
    if (pi[k] >= 0) {
      out[(size_t)pi[k] * ncmax + pj[k]] = acc[k];
      pos[(size_t)pi[k] * ncmax + pj[k]] = place(pi[k], pj[k]);
      if (pi[k] != pj[k]) {
        out[(size_t)pj[k] * ncmax + pi[k]] = acc[k];
        pos[(size_t)pj[k] * ncmax + pi[k]] = place(pj[k], pi[k]);
      }
    }
  if (lane < nc) Fb[(size_t)e * ncmax + lane] = F;
}

// Second pass: one thread per row, its (element, local row) pairs in ascending element order -- the order of the reference's element loop --, every entry added
// at the place the first pass found for it.
__global__ __launch_bounds__(64) void k_poisson_rows_generic(int ndof, int ncmax, GenTabs tabs, const unsigned char* __restrict__ etab,
                                                             const int* __restrict__ adj_ptr, const int* __restrict__ adj, const double* __restrict__ Kb,
                                                             const int* __restrict__ Pos, const double* __restrict__ Fb, const int* __restrict__ rowptr,
                                                             double* __restrict__ val, double* __restrict__ res) {
  const int r = blockIdx.x * 64 + threadIdx.x;
  if (r >= ndof) return;
  const int rs = rowptr[r], re = rowptr[r + 1];
  for (int k = rs; k < re; k++) val[k] = 0.0;
  double racc = 0.0;
  for (int a = adj_ptr[r]; a < adj_ptr[r + 1]; a++) {
    const int e = adj[a] / GEN_NC, i = adj[a] % GEN_NC;
    const int nc = tabs.t[etab ? etab[e] : 0].nc;
    const size_t pr = (size_t)e * ncmax + i;
    racc += Fb[pr];
    const double* B = Kb + pr * ncmax;
    const int* at = Pos + pr * ncmax;
    for (int j = 0; j < nc; j++)
      if (at[j] >= 0) val[at[j]] += B[j];
  }
  res[r] = racc;
}

// shapes[ns] (ns <= 3, one dimension), elem_shape[nel] = index into shapes per element (nullptr: every element is shapes[0])
static int poisson_rows_impl(fh_ctx_t ctx, int ns, const int* shapes, const int* elem_shape, int fe, int order, int nel, int nloc, const int* elem_dof, int nnode,
                             const double* coords, fh_vec_t sol, fh_expr_t source, double scale, fh_mat_t KK, fh_vec_t RES) {
  const int dim = fhfe::dim_of(shapes[0]), ndof = KK->m;
  int ncs[3] = {0, 0, 0};
  for (int k = 0; k < ns; k++) {
    FH_REQUIRE(shapes[k] >= 0 && shapes[k] <= 5, "fh_assemble_poisson_rows: geom must be 0 (hex), 1 (quad), 2 (line), 3 (triangle), 4 (tetrahedron) or 5 (prism)");
    FH_REQUIRE(fhfe::dim_of(shapes[k]) == dim, "fh_assemble_poisson_mixed: the shapes of one mesh have one dimension");
    ncs[k] = fhfe::ndofs_of(shapes[k], fe);
    FH_REQUIRE(nloc >= ncs[k] && ncs[k] <= GEN_NC, "fh_assemble_poisson_rows: %d nodes per element given, the family has %d", nloc, ncs[k]);
  }
  FH_REQUIRE(KK->n == ndof && RES->n_local >= ndof && (!sol || sol->n_local >= ndof), "fh_assemble_poisson_rows: size mismatch");
  std::vector<unsigned char> etab;
  if (elem_shape) {
    etab.resize(nel);
    for (int e = 0; e < nel; e++) {
      FH_REQUIRE(elem_shape[e] >= 0 && elem_shape[e] < ns, "fh_assemble_poisson_mixed: element %d names shape %d of %d", e, elem_shape[e], ns);
      etab[e] = (unsigned char)elem_shape[e];
    }
  }
  auto nc_of = [&](int e) { return ncs[elem_shape ? elem_shape[e] : 0]; };
  std::vector<int> cnt(ndof + 1, 0);
  for (int e = 0; e < nel; e++)
    for (int n = 0; n < nc_of(e); n++) {
      const int d = elem_dof[(size_t)e * nloc + n];
      FH_REQUIRE(d >= 0 && d < ndof && d < nnode, "fh_assemble_poisson_rows: element %d, node %d: dof %d outside the system (the classes are numbered one after the other)", e, n, d);
      cnt[d + 1]++;
    }
  for (int d = 0; d < ndof; d++) cnt[d + 1] += cnt[d];
  std::vector<int> adj(cnt[ndof]), fill(cnt.begin(), cnt.end() - 1);
  FH_REQUIRE((int64_t)nel * GEN_NC < 2147483647ll, "fh_assemble_poisson_rows: too many elements");
  for (int e = 0; e < nel; e++)                         // ascending element order per dof
    for (int n = 0; n < nc_of(e); n++) adj[fill[elem_dof[(size_t)e * nloc + n]]++] = e * GEN_NC + n;
  std::vector<double> w[3], phi[3], dphi[3];
  for (int k = 0; k < ns; k++) FH_REQUIRE(fhfe::shape_tables(shapes[k], fe, order, w[k], phi[k], dphi[k]) == 0, "fh_assemble_poisson_rows: unsupported Gauss rule");
  std::vector<int> code;
  std::vector<double> consts;
  if (source) {
    int nv = 0, ncode = 0, nk = 0;
    FH_TRY(fh_expr_nvars(source, &nv));
    FH_REQUIRE(nv <= 4, "fh_assemble_poisson_rows: the source expression has %d variables, at most 4 (x, y, z, t) are served", nv);
    FH_TRY(fh_expr_program(source, &ncode, &nk, nullptr, nullptr));
    code.resize(ncode);
    consts.resize(std::max(nk, 1));
    FH_TRY(fh_expr_program(source, &ncode, &nk, code.data(), consts.data()));
  }
  hipStream_t st = ctx->stream;
  std::vector<void*> dv;
  bool oom = false;
  auto up = [&](const void* h, size_t bytes) -> void* {
    void* d = nullptr;
    if (hipMalloc(&d, std::max<size_t>(bytes, 8)) != hipSuccess) {
      oom = true;
      return nullptr;
    }
    dv.push_back(d);
    if (bytes && h) hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st);
    return d;
  };
  int* d_ptr = (int*)up(cnt.data(), cnt.size() * sizeof(int));
  int* d_adj = (int*)up(adj.data(), adj.size() * sizeof(int));
  int* d_ed = (int*)up(elem_dof, (size_t)nel * nloc * sizeof(int));
  double* d_x = (double*)up(coords, (size_t)nnode * dim * sizeof(double));
  GenTabs tabs;
  for (int k = 0; k < 3; k++) tabs.t[k] = GenTab{0, 0, nullptr, nullptr, nullptr};
  for (int k = 0; k < ns; k++)
    tabs.t[k] = GenTab{ncs[k], (int)w[k].size(), (const double*)up(w[k].data(), w[k].size() * sizeof(double)), (const double*)up(phi[k].data(), phi[k].size() * sizeof(double)),
                       (const double*)up(dphi[k].data(), dphi[k].size() * sizeof(double))};
  unsigned char* d_etab = elem_shape ? (unsigned char*)up(etab.data(), etab.size()) : nullptr;
  int* d_code = source ? (int*)up(code.data(), code.size() * sizeof(int)) : nullptr;
  double* d_k = source ? (double*)up(consts.data(), consts.size() * sizeof(double)) : nullptr;
  const int ncmax = std::max(ncs[0], std::max(ncs[1], ncs[2]));
  FH_REQUIRE((int64_t)nel * ncmax < 2147483647ll, "fh_assemble_poisson_rows: too many elements");
  double* d_Kb = (double*)up(nullptr, (size_t)nel * ncmax * ncmax * sizeof(double));      // element rows between the two passes
  double* d_Fb = (double*)up(nullptr, (size_t)nel * ncmax * sizeof(double));
  int* d_Pos = (int*)up(nullptr, (size_t)nel * ncmax * ncmax * sizeof(int));
  int rc = 0;
  if (oom) {
    fh_set_error("fh_assemble_poisson_rows: out of device memory");
    rc = 2;
  } else {
    hipLaunchKernelGGL(k_poisson_pairs_generic, dim3(nel), dim3(64), 0, st, nel, ncmax, dim, tabs, d_etab, nloc, d_ed, d_x,
                       sol ? sol->d : nullptr, scale, d_code, (int)code.size(), d_k, KK->d_rowptr, KK->d_col, d_Kb, d_Pos, d_Fb);
    hipLaunchKernelGGL(k_poisson_rows_generic, dim3(fh_div_up(ndof, 64)), dim3(64), 0, st, ndof, ncmax, tabs, d_etab, d_ptr, d_adj, d_Kb, d_Pos, d_Fb, KK->d_rowptr,
                       KK->d_val, RES->d);
    if (hipGetLastError() != hipSuccess) {
      fh_set_error("fh_assemble_poisson_rows: launch failed");
      rc = 2;
    }
    KK->val_gen++;
    KK->at_valid = false;
  }
  hipStreamSynchronize(st);
  for (void* q : dv) hipFree(q);
  return rc;
}

extern "C" int fh_assemble_poisson_rows(fh_ctx_t ctx, int geom, int fe, int order, int nel, int nloc, const int* elem_dof, int nnode, const double* coords,
                                        fh_vec_t sol, fh_expr_t source, double scale, fh_mat_t KK, fh_vec_t RES) {
  FH_GUARD_BEGIN
  FH_REQUIRE(ctx && elem_dof && coords && KK && RES && nel >= 1 && nnode >= 1, "fh_assemble_poisson_rows: null or empty argument");
  FH_REQUIRE(geom >= 0 && geom <= 5, "fh_assemble_poisson_rows: geom must be 0 (hex), 1 (quad), 2 (line), 3 (triangle), 4 (tetrahedron) or 5 (prism)");
  FH_REQUIRE(fe == fhfe::FE_LINEAR || fe == fhfe::FE_SERENDIPITY || fe == fhfe::FE_BIQUADRATIC, "fh_assemble_poisson_rows: fe must be 0, 1 or 2");
  return poisson_rows_impl(ctx, 1, &geom, nullptr, fe, order, nel, nloc, elem_dof, nnode, coords, sol, source, scale, KK, RES);
  FH_GUARD_END("fh_assemble_poisson_rows")
}

// The same on a mesh of MIXED shapes (cube_all_shapes*.neu of applications/001_Poisson: hexahedra, tetrahedra and prisms in one file): elem_geom[nel] names the
// shape of every element (at most three different ones, of one dimension); rows of elem_dof padded to nloc.  The entries of a row are summed in ascending element
// order whatever the shapes, as the reference's element loop does.
extern "C" int fh_assemble_poisson_mixed(fh_ctx_t ctx, int fe, int order, int nel, int nloc, const int* elem_geom, const int* elem_dof, int nnode, const double* coords,
                                         fh_vec_t sol, fh_expr_t source, double scale, fh_mat_t KK, fh_vec_t RES) {
  FH_GUARD_BEGIN
  FH_REQUIRE(ctx && elem_geom && elem_dof && coords && KK && RES && nel >= 1 && nnode >= 1, "fh_assemble_poisson_mixed: null or empty argument");
  FH_REQUIRE(fe == fhfe::FE_LINEAR || fe == fhfe::FE_SERENDIPITY || fe == fhfe::FE_BIQUADRATIC, "fh_assemble_poisson_mixed: fe must be 0, 1 or 2");
  int shapes[3], ns = 0;
  std::vector<int> idx(nel);
  for (int e = 0; e < nel; e++) {
    int k = 0;
    while (k < ns && shapes[k] != elem_geom[e]) k++;
    if (k == ns) {
      FH_REQUIRE(ns < 3, "fh_assemble_poisson_mixed: more than three shapes in one mesh");
      FH_REQUIRE(elem_geom[e] >= 0 && elem_geom[e] <= 5, "fh_assemble_poisson_mixed: element %d: shape %d", e, elem_geom[e]);
      shapes[ns++] = elem_geom[e];
    }
    idx[e] = k;
  }
  return poisson_rows_impl(ctx, ns, shapes, idx.data(), fe, order, nel, nloc, elem_dof, nnode, coords, sol, source, scale, KK, RES);
  FH_GUARD_END("fh_assemble_poisson_mixed")
}
