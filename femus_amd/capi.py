"""Thin object wrappers over the C-ABI (include/femus_hip.h) for tests/ and bench.py.

Method names follow the reference classes they stand for (NumericVector / SparseMatrix /
LinearEquationSolver), so the parity tests read like FEMuS code.  No numerics happen here.
"""
import ctypes
import numpy as np
from ._lib import load_library

GEOM = {"hex": 0, "quad": 1, "line": 2, "tri": 3, "tet": 4, "wedge": 5}
FE = {"linear": 0, "serendipity": 1, "biquadratic": 2, "constant": 3, "pwlinear": 4}        # 4: DISCONTINUOUS_POLYNOMIAL FIRST (system dof maps and prolongators only)
GAUSS_ORDER = {"zero": 0, "first": 0, "second": 1, "third": 1, "fourth": 2, "fifth": 2,
               "sixth": 3, "seventh": 3, "eighth": 4, "ninth": 4}
OUTER = {"preonly": 0, "richardson": 1, "gmres": 2, "cg": 3, "fgmres": 4}
SMOOTH_JACOBI, SMOOTH_GS_COLOR, SMOOTH_VANKA, SMOOTH_SOR, SMOOTH_ILU0, SMOOTH_IDENTITY, SMOOTH_LU, SMOOTH_ASM = 0, 1, 2, 3, 4, 5, 6, 7


class FemusHipError(RuntimeError):
    pass


def _chk(rc):
    if rc != 0:
        raise FemusHipError(load_library().fh_last_error().decode())


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class Context:
    """FemusInit replacement: one HIP device, a compute stream and a communication stream."""

    def __init__(self, device=0):
        self.L = load_library()
        self.h = ctypes.c_void_p()
        _chk(self.L.fh_init(int(device), ctypes.byref(self.h)))

    def close(self):
        if self.h:
            self.L.fh_finalize(self.h)
            self.h = ctypes.c_void_p()

    def device_name(self):
        buf = ctypes.create_string_buffer(256)
        _chk(self.L.fh_device_name(self.h, buf, 256))
        return buf.value.decode()

    def sync(self):
        _chk(self.L.fh_sync(self.h))

    def set_option(self, name, value):
        _chk(self.L.fh_set_option(self.h, name.encode(), float(value)))

    def marker(self, ident):
        """one-thread kernel k_phase_marker<ident> on the compute stream: cuts a kernel trace into phases (profiles/summarize.py)"""
        _chk(self.L.fh_profile_marker(self.h, int(ident)))

    def timer_start(self):
        _chk(self.L.fh_timer_start(self.h))

    def timer_stop(self):
        ms = ctypes.c_double()
        _chk(self.L.fh_timer_stop(self.h, ctypes.byref(ms)))
        return ms.value

    def record(self):
        """context manager: the device-only calls made inside are recorded into a hipGraph instead of executed; `.graph` replays them
        (fh_graph_begin / fh_graph_end)"""
        return _Recording(self)

    # factories
    def vector(self, n_global, n_local=None, first_local=0, ghost=None):
        return Vec(self, n_global, n_global if n_local is None else n_local, first_local, ghost)

    def vector_from(self, array):
        a = _f64(array)
        v = Vec(self, a.size, a.size, 0, None)
        v.upload(a)
        return v

    def matrix_csr(self, m, n, rowptr, col, val=None):
        return Mat.from_csr(self, m, n, rowptr, col, val)

    def matrix_from_elements(self, elem_dof, m, n=None):
        return Mat.from_elements(self, elem_dof, m, n)

    def matrix_from_mesh(self, mesh, fe):
        return Mat.from_mesh(self, mesh, fe)

    def matrix_scipy(self, A):
        A = A.tocsr()
        A.sort_indices()
        return Mat.from_csr(self, A.shape[0], A.shape[1], A.indptr, A.indices, A.data)


class Graph:
    """a recorded launch sequence (fh_graph_t)"""

    def __init__(self, L, handle):
        self.L, self.h = L, handle

    def launch(self):
        _chk(self.L.fh_graph_launch(self.h))

    def destroy(self):
        if self.h:
            _chk(self.L.fh_graph_destroy(self.h))
            self.h = None


class _Recording:
    def __init__(self, ctx):
        self.ctx, self.graph = ctx, None

    def __enter__(self):
        _chk(self.ctx.L.fh_graph_begin(self.ctx.h))
        return self

    def __exit__(self, et, ev, tb):
        h = ctypes.c_void_p()
        rc = self.ctx.L.fh_graph_end(self.ctx.h, ctypes.byref(h))
        if et is None:
            _chk(rc)
            self.graph = Graph(self.ctx.L, h)
        elif rc == 0 and h:
            self.ctx.L.fh_graph_destroy(h)
        return False


class Vec:
    """NumericVector (src/03_algebra/00_vectors/NumericVector.hpp:51)."""

    def __init__(self, ctx, n_global, n_local, first_local=0, ghost=None, handle=None):
        self.ctx, self.L = ctx, ctx.L
        if handle is not None:
            self.h = handle
        else:
            self.h = ctypes.c_void_p()
            g = None if ghost is None else _i32(ghost)
            _chk(self.L.fh_vec_create(ctx.h, int(n_global), int(n_local), int(first_local), _p(g),
                                      0 if g is None else g.size, ctypes.byref(self.h)))
        sz = [ctypes.c_int() for _ in range(4)]
        _chk(self.L.fh_vec_size(self.h, *[ctypes.byref(s) for s in sz]))
        self.n_global, self.n_local, self.first_local, self.nghost = [s.value for s in sz]

    def destroy(self):
        if self.h:
            self.L.fh_vec_destroy(self.h)
            self.h = None

    def clone(self):
        h = ctypes.c_void_p()
        _chk(self.L.fh_vec_duplicate(self.h, ctypes.byref(h)))
        return Vec(self.ctx, 0, 0, handle=h)

    def size(self):
        return self.n_global

    def local_size(self):
        return self.n_local

    def zero(self):
        _chk(self.L.fh_vec_zero(self.h))

    def fill(self, s):
        _chk(self.L.fh_vec_fill(self.h, float(s)))

    def assign(self, other):
        _chk(self.L.fh_vec_copy(self.h, other.h))

    def upload(self, a):
        a = _f64(a)
        assert a.size == self.n_local
        _chk(self.L.fh_vec_upload(self.h, _p(a)))

    def to_numpy(self):
        out = np.empty(self.n_local)
        _chk(self.L.fh_vec_download(self.h, _p(out)))
        return out

    def set(self, idx, vals):
        idx, vals = _i32(np.atleast_1d(idx)), _f64(np.atleast_1d(vals))
        _chk(self.L.fh_vec_set_values(self.h, idx.size, _p(idx), _p(vals)))

    def add_vector_blocked(self, vals, idx):
        idx, vals = _i32(idx), _f64(vals)
        _chk(self.L.fh_vec_add_values(self.h, idx.size, _p(idx), _p(vals)))

    def stage_vector_blocked(self, vals, idx):
        """add_vector_blocked as PETSc performs it: staged until flush() (= close())"""
        idx, vals = _i32(idx), _f64(vals)
        _chk(self.L.fh_vec_stage_values(self.h, idx.size, _p(idx), _p(vals)))

    def flush(self):
        _chk(self.L.fh_vec_flush(self.h))

    def ghost_adds(self, nghost):
        """the staged adds collected for this rank's ghost entries (shipped to their owners by fh_halo_reverse_add)"""
        out = np.zeros(max(int(nghost), 1))
        _chk(self.L.fh_vec_ghost_adds(self.h, _p(out)))
        return out[:nghost]

    def get(self, idx):
        idx = _i32(np.atleast_1d(idx))
        out = np.empty(idx.size)
        _chk(self.L.fh_vec_get_values(self.h, idx.size, _p(idx), _p(out)))
        return out

    def __call__(self, i):
        return float(self.get([i])[0])

    def add(self, a, v=None):
        if v is None:
            _chk(self.L.fh_vec_shift(self.h, float(a)))
        else:
            _chk(self.L.fh_vec_axpy(self.h, float(a), v.h))

    def aypx(self, a, x):
        _chk(self.L.fh_vec_aypx(self.h, float(a), x.h))

    def scale(self, s):
        _chk(self.L.fh_vec_scale(self.h, float(s)))

    def abs(self):
        _chk(self.L.fh_vec_abs(self.h))

    def pointwise_mult(self, a, b):
        _chk(self.L.fh_vec_pointwise_mult(self.h, a.h, b.h))

    def dot(self, other):
        out = ctypes.c_double()
        _chk(self.L.fh_vec_dot(self.h, other.h, ctypes.byref(out)))
        return out.value

    def _norm(self, kind):
        out = ctypes.c_double()
        _chk(self.L.fh_vec_norm(self.h, kind, ctypes.byref(out)))
        return out.value

    def l1_norm(self):
        return self._norm(1)

    def l2_norm(self):
        return self._norm(2)

    def linfty_norm(self):
        return self._norm(0)

    def _reduce(self, kind):
        out = ctypes.c_double()
        _chk(self.L.fh_vec_reduce(self.h, kind, ctypes.byref(out)))
        return out.value

    def sum(self):
        return self._reduce(0)

    def min(self):
        return self._reduce(1)

    def max(self):
        return self._reduce(2)

    # SpMV family (NumericVector.hpp:281-284)
    def matrix_mult(self, vec_in, mat):
        _chk(self.L.fh_spmv(mat.h, vec_in.h, self.h, 0, None, None, 0.0))

    def add_vector(self, vec_in, mat):
        _chk(self.L.fh_spmv(mat.h, vec_in.h, self.h, 1, None, None, 0.0))

    def resid(self, rhs, x, mat):
        _chk(self.L.fh_spmv(mat.h, x.h, self.h, 2, rhs.h, None, 0.0))

    def jacobi_sweep(self, rhs, x, mat, dinv, omega):
        _chk(self.L.fh_spmv(mat.h, x.h, self.h, 3, rhs.h, dinv.h, float(omega)))

    def matrix_mult_transpose(self, vec_in, mat):
        _chk(self.L.fh_spmv_transpose(mat.h, vec_in.h, self.h))


class Mat:
    """SparseMatrix (src/03_algebra/01_matrices/SparseMatrix.hpp:48)."""

    def __init__(self, ctx, handle):
        self.ctx, self.L, self.h = ctx, ctx.L, handle
        m, n, nnz = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _chk(self.L.fh_mat_size(self.h, ctypes.byref(m), ctypes.byref(n), ctypes.byref(nnz)))
        self.m_, self.n_, self.nnz = m.value, n.value, nnz.value

    @classmethod
    def from_csr(cls, ctx, m, n, rowptr, col, val=None):
        rowptr, col = _i32(rowptr), _i32(col)
        v = None if val is None else _f64(val)
        h = ctypes.c_void_p()
        _chk(ctx.L.fh_mat_create_csr(ctx.h, int(m), int(n), _p(rowptr), _p(col), _p(v), ctypes.byref(h)))
        return cls(ctx, h)

    @classmethod
    def from_elements(cls, ctx, elem_dof, m, n=None):
        """finite-element pattern of the element dof table, built on the device (fh_mat_create_from_elements): m owned rows over n columns"""
        ed = _i32(np.ascontiguousarray(elem_dof))
        h = ctypes.c_void_p()
        _chk(ctx.L.fh_mat_create_from_elements(ctx.h, int(ed.shape[0]), int(ed.shape[1]), _p(ed), int(m), int(m if n is None else n), ctypes.byref(h)))
        return cls(ctx, h)

    @classmethod
    def from_mesh(cls, ctx, mesh, fe):
        """the same for the square operator of one variable on a mesh, from the mesh's device copy (fh_mat_create_from_mesh)"""
        h = ctypes.c_void_p()
        _chk(ctx.L.fh_mat_create_from_mesh(ctx.h, mesh.h, FE[fe], ctypes.byref(h)))
        return cls(ctx, h)

    def destroy(self):
        if self.h:
            self.L.fh_mat_destroy(self.h)
            self.h = None

    def m(self):
        return self.m_

    def n(self):
        return self.n_

    def zero(self):
        _chk(self.L.fh_mat_zero(self.h))

    def set_values(self, val):
        val = _f64(val)
        assert val.size == self.nnz
        _chk(self.L.fh_mat_set_values_csr(self.h, _p(val)))

    def values(self):
        out = np.empty(self.nnz)
        _chk(self.L.fh_mat_get_values_csr(self.h, _p(out)))
        return out

    def pattern(self):
        rp, col = np.empty(self.m_ + 1, np.int32), np.empty(self.nnz, np.int32)
        _chk(self.L.fh_mat_get_pattern(self.h, _p(rp), _p(col)))
        return rp, col

    def to_scipy(self):
        import scipy.sparse as sp
        rp, col = self.pattern()
        return sp.csr_matrix((self.values(), col, rp), shape=(self.m_, self.n_))

    def add_matrix_blocked(self, vals, rows, cols):
        rows, cols, vals = _i32(rows), _i32(cols), _f64(vals)
        _chk(self.L.fh_mat_add_block(self.h, rows.size, _p(rows), cols.size, _p(cols), _p(vals)))

    def stage_matrix_blocked(self, vals, rows, cols):
        """add_matrix_blocked as PETSc performs it: staged in the pinned ring until flush() (= close())"""
        rows, cols, vals = _i32(rows), _i32(cols), _f64(vals)
        _chk(self.L.fh_mat_stage_block(self.h, rows.size, _p(rows), cols.size, _p(cols), _p(vals)))

    def flush(self):
        _chk(self.L.fh_mat_flush(self.h))

    def stage_stats(self):
        b, r = ctypes.c_int64(), ctypes.c_int64()
        _chk(self.L.fh_mat_stage_stats(self.h, ctypes.byref(b), ctypes.byref(r)))
        return b.value, r.value

    def insert_row(self, row, cols, vals):
        cols, vals = _i32(cols), _f64(vals)
        _chk(self.L.fh_mat_insert_row(self.h, int(row), cols.size, _p(cols), _p(vals)))

    def get_row(self, row):
        n = ctypes.c_int()
        _chk(self.L.fh_mat_get_row(self.h, int(row), ctypes.byref(n), None, None))
        cols, vals = np.empty(n.value, np.int32), np.empty(n.value)
        _chk(self.L.fh_mat_get_row(self.h, int(row), ctypes.byref(n), _p(cols), _p(vals)))
        return cols, vals

    def mat_zero_rows(self, index, diag):
        index = _i32(index)
        _chk(self.L.fh_mat_zero_rows(self.h, index.size, _p(index), float(diag)))

    def zero_cols(self, index):
        index = _i32(index)
        _chk(self.L.fh_mat_zero_cols(self.h, index.size, _p(index)))

    def get_diagonal(self, dest):
        _chk(self.L.fh_mat_get_diagonal(self.h, dest.h))

    def get_transpose(self):
        h = ctypes.c_void_p()
        _chk(self.L.fh_mat_transpose(self.h, ctypes.byref(h)))
        return Mat(self.ctx, h)

    def matrix_PtAP(self, P, A, reuse=False):
        """self = P^T A P ; call as C = Mat.ptap(P, A) for the first product"""
        h = self.h if reuse else ctypes.c_void_p()
        _chk(self.L.fh_mat_ptap(P.h, A.h, ctypes.byref(h)))
        return self

    @classmethod
    def ptap(cls, P, A):
        h = ctypes.c_void_p()
        _chk(P.L.fh_mat_ptap(P.h, A.h, ctypes.byref(h)))
        return cls(P.ctx, h)

    def matmul(self, B):
        """C = self * B (SpGEMM)"""
        h = ctypes.c_void_p()
        _chk(self.L.fh_mat_matmul(self.h, B.h, ctypes.byref(h)))
        return Mat(self.ctx, h)

    @classmethod
    def matrix_ABC(cls, A, B, C):
        """SparseMatrix::matrix_ABC (SparseMatrix.hpp:186): A*B*C"""
        AB = A.matmul(B)
        out = AB.matmul(C)
        AB.destroy()
        return out

    def ptap_numeric(self, P, A):
        h = ctypes.c_void_p(self.h.value if isinstance(self.h, ctypes.c_void_p) else self.h)
        _chk(self.L.fh_mat_ptap(P.h, A.h, ctypes.byref(h)))

    def l1_norm(self):
        out = ctypes.c_double()
        _chk(self.L.fh_mat_norm(self.h, 1, ctypes.byref(out)))
        return out.value

    def linfty_norm(self):
        out = ctypes.c_double()
        _chk(self.L.fh_mat_norm(self.h, 0, ctypes.byref(out)))
        return out.value

    # ---- owned-row operators of a domain-decomposed level (device-side cut out of the extended-box operator) ----
    def col_mask(self, rows, mask=None):
        """uint8 mask over the columns: 1 where one of `rows` has an entry"""
        rows = _i32(rows)
        if mask is None:
            mask = np.zeros(self.n_, dtype=np.uint8)
        _chk(self.L.fh_mat_col_mask(self.h, rows.size, _p(rows), _p(mask)))
        return mask

    def row_mask(self, colmask, mask=None):
        """uint8 mask over the rows: 1 where the row has an entry in a masked column"""
        cm = np.ascontiguousarray(colmask, dtype=np.uint8)
        assert cm.size == self.n_
        if mask is None:
            mask = np.zeros(self.m_, dtype=np.uint8)
        _chk(self.L.fh_mat_row_mask(self.h, _p(cm), _p(mask)))
        return mask

    def restrict(self, rows, newcol, ncols_new, check=False):
        """(dst, map): dst = rows of self with columns renumbered by newcol (< 0: dropped); map re-gathers the values later"""
        rows, newcol = _i32(rows), _i32(newcol)
        assert newcol.size == self.n_
        if check:
            mx = ctypes.c_double()
            _chk(self.L.fh_mat_restrict_check(self.h, rows.size, _p(rows), _p(newcol), ctypes.byref(mx)))
            assert mx.value == 0.0, "an owned row reads a node outside the halo (|value| %g)" % mx.value
        h, mh = ctypes.c_void_p(), ctypes.c_void_p()
        _chk(self.L.fh_mat_restrict(self.h, rows.size, _p(rows), _p(newcol), int(ncols_new), ctypes.byref(h), ctypes.byref(mh)))
        return Mat(self.ctx, h), Index.from_handle(self.ctx, mh, None)

    def value_map(self, src, src_row=None, src_col=None):
        """Index m with self.val[k] <- src.val[m[k]] for the entry (r, c) of self taken from (src_row[r], src_col[c]) of src"""
        sr = None if src_row is None else _i32(src_row)
        sc = None if src_col is None else _i32(src_col)
        mh = ctypes.c_void_p()
        _chk(self.L.fh_mat_value_map(self.h, src.h, _p(sr), _p(sc), ctypes.byref(mh)))
        return Index.from_handle(self.ctx, mh, self.nnz)

    @classmethod
    def abc(cls, A, B, C):
        """D = A B C with a reusable plan (SparseMatrix::matrix_ABC, SparseMatrix.hpp:186); repeat with D.abc_numeric(A, B, C)"""
        h = ctypes.c_void_p()
        _chk(A.L.fh_mat_abc(A.h, B.h, C.h, ctypes.byref(h)))
        return cls(A.ctx, h)

    def abc_numeric(self, A, B, C):
        h = ctypes.c_void_p(self.h.value if isinstance(self.h, ctypes.c_void_p) else self.h)
        _chk(self.L.fh_mat_abc(A.h, B.h, C.h, ctypes.byref(h)))

    def split_info(self, n_own_cols):
        """(interior, interface) row-block counts of the overlap split for an operator over [owned | ghost] columns"""
        a, b = ctypes.c_int(), ctypes.c_int()
        _chk(self.L.fh_mat_split_info(self.h, int(n_own_cols), ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def spmv_algorithmic_bytes(self):
        return int(self.L.fh_spmv_algorithmic_bytes(self.h))

    def spmv_expected_bytes(self, mode=0):
        """(lo, hi) bytes of the arrays the product kernel touches: x counted once / once per row block (fh_spmv_expected_bytes)"""
        lo, hi = ctypes.c_int64(), ctypes.c_int64()
        _chk(self.L.fh_spmv_expected_bytes(self.h, int(mode), ctypes.byref(lo), ctypes.byref(hi)))
        return lo.value, hi.value


class Mesh:
    """box mesh + uniform refinement in FEMuS numbering (Mesh / MeshRefinement, nprocs = 1)."""

    def __init__(self, L, handle):
        self.L, self.h = L, handle
        dim, nel, nnode, nloc, lev = [ctypes.c_int() for _ in range(5)]
        own = (ctypes.c_int * 3)()
        _chk(L.fh_mesh_info(self.h, ctypes.byref(dim), ctypes.byref(nel), ctypes.byref(nnode), ctypes.byref(nloc), own, ctypes.byref(lev)))
        self.dim, self.nel, self.nnode, self.nloc, self.level = dim.value, nel.value, nnode.value, nloc.value, lev.value
        self.own_size = list(own)
        self.nfaces = 2 * self.dim
        self.geom = "hex" if self.dim == 3 else "quad"

    @classmethod
    def box(cls, nx, ny, nz, lo=(0., 0., 0.), hi=(1., 1., 1.)):
        L = load_library()
        h = ctypes.c_void_p()
        lo_, hi_ = _f64(lo), _f64(hi)
        _chk(L.fh_mesh_box(nx, ny, nz, _p(lo_), _p(hi_), ctypes.byref(h)))
        return cls(L, h)

    @classmethod
    def read_gambit(cls, path, Lref=1.0):
        """MultiLevelMesh::ReadCoarseMesh for a Gambit neutral file (HEX27 / QUAD9)"""
        L = load_library()
        h = ctypes.c_void_p()
        _chk(L.fh_mesh_read_gambit(str(path).encode(), float(Lref), ctypes.byref(h)))
        return cls(L, h)

    def refine(self, ctx=None):
        """uniform refinement; with a context: on the device (fh_mesh_refine_device, same arrays bit for bit, the new mesh stays resident)"""
        if ctx is not None:
            return self.refine_device(ctx)
        h = ctypes.c_void_p()
        _chk(self.L.fh_mesh_refine(self.h, ctypes.byref(h)))
        return Mesh(self.L, h)

    def refine_device(self, ctx, flags=None):
        """MeshRefinement::RefineMesh on the device; flags as refine_flagged (None: every element of the current level)"""
        f = None
        if flags is not None:
            f = np.ascontiguousarray(flags, dtype=np.uint8)
            assert f.shape == (self.nel,)
        h = ctypes.c_void_p()
        _chk(self.L.fh_mesh_refine_device(ctx.h, self.h, None if f is None else _p(f), ctypes.byref(h)))
        return Mesh(self.L, h)

    def partition(self, nparts, weights=None):
        """native k-way partition of the dual graph (the METIS_PartMeshDual of MeshMetisPartitioning.cpp:71-113): part[nel];
        weights[nel] > 0: the parts balance the weight sums (adaptive levels) instead of the element counts"""
        part = np.empty(self.nel, dtype=np.int32)
        if weights is None:
            _chk(self.L.fh_mesh_partition(self.h, int(nparts), _p(part)))
        else:
            w = _f64(weights)
            assert w.shape == (self.nel,)
            _chk(self.L.fh_mesh_partition_weighted(self.h, int(nparts), _p(w), _p(part)))
        return part

    def rank_elements(self, part, rank):
        """(owned elements, ring elements sharing a node with them) of a rank"""
        part = _i32(part)
        no, nt = ctypes.c_int(), ctypes.c_int()
        _chk(self.L.fh_mesh_rank_elements(self.h, _p(part), int(rank), ctypes.byref(no), ctypes.byref(nt), None))
        el = np.empty(nt.value, dtype=np.int32)
        _chk(self.L.fh_mesh_rank_elements(self.h, _p(part), int(rank), ctypes.byref(no), ctypes.byref(nt), _p(el)))
        return el[:no.value], el[no.value:]

    def submesh(self, elems):
        """FEMuS-numbered mesh of a list of elements; returns (mesh, node of this mesh for every node of the sub-mesh)"""
        el = _i32(elems)
        h = ctypes.c_void_p()
        # the node count is not known before the call: at most nloc per element
        gid = np.empty(el.size * self.nloc, dtype=np.int32)
        _chk(self.L.fh_mesh_submesh(self.h, el.size, _p(el), ctypes.byref(h), _p(gid)))
        sub = Mesh(self.L, h)
        return sub, gid[:sub.nnode].copy()

    def topo_node_keys(self, part, levels, elem_gid0, level):
        """global id and owner of every node of levels[level] (a refinement of a sub-mesh of this coarse mesh) -- fh_dd_topo_node_keys"""
        part, eg = _i32(part), _i32(elem_gid0)
        hs = (ctypes.c_void_p * len(levels))(*[m.h for m in levels])
        n = levels[level].nnode
        gid, owner = np.empty(n, dtype=np.int64), np.empty(n, dtype=np.int32)
        _chk(self.L.fh_dd_topo_node_keys(self.h, _p(part), len(levels), hs, _p(eg), int(level), _p(gid), _p(owner)))
        return gid, owner.astype(np.int64)

    def refine_flagged(self, flags):
        """selective refinement (MeshRefinement::RefineMesh with an AMR flag per element)"""
        f = np.ascontiguousarray(flags, dtype=np.uint8)
        assert f.shape == (self.nel,)
        h = ctypes.c_void_p()
        _chk(self.L.fh_mesh_refine_flagged(self.h, _p(f), ctypes.byref(h)))
        return Mesh(self.L, h)

    def elem_centroids(self):
        out = np.empty((self.nel, 3))
        _chk(self.L.fh_mesh_elem_centroids(self.h, _p(out)))
        return out

    def elem_groups(self):
        """(group, material) per element: Gambit numbers, inherited through refinements (a generated box: 1, 2)"""
        g, mt = np.empty(self.nel, np.int32), np.empty(self.nel, np.int32)
        _chk(self.L.fh_mesh_elem_groups(self.h, _p(g), _p(mt)))
        return g, mt

    def elem_levels(self):
        out = np.empty(self.nel, np.int32)
        hom = ctypes.c_int()
        _chk(self.L.fh_mesh_elem_levels(self.h, _p(out), ctypes.byref(hom)))
        return out, bool(hom.value)

    def flag_elements(self, fn):
        """MeshRefinement::FlagElementsToRefine type 1: fn(x[3], level) at the vertex mean of elements of the current level"""
        xc = self.elem_centroids()
        lev, _ = self.elem_levels()
        return np.array([lev[e] == self.level and bool(fn(xc[e], self.level)) for e in range(self.nel)], dtype=np.uint8)

    def set_amr_mode(self, mode):
        """"reference" (default): the restriction map exactly as Mesh::GetAMRRestrictionAndAMRSolidMark builds it; "coarsest": the
        consistent variant (rows sum to one).  Inherited by meshes refined from this one"""
        _chk(self.L.fh_mesh_set_amr_mode(self.h, {"reference": 0, "coarsest": 1}[mode]))
        return self

    def amr_constraints(self, fe):
        """hanging dofs and their master weights: (hanging[n], ptr[n+1], master[nnz], weight[nnz])"""
        n, nnz = ctypes.c_int(0), ctypes.c_int(0)
        _chk(self.L.fh_mesh_amr_constraints(self.h, FE[fe], ctypes.byref(n), ctypes.byref(nnz), None, None, None, None))
        hang, ptr = np.empty(n.value, np.int32), np.empty(n.value + 1, np.int32)
        master, w = np.empty(nnz.value, np.int32), np.empty(nnz.value)
        _chk(self.L.fh_mesh_amr_constraints(self.h, FE[fe], ctypes.byref(n), ctypes.byref(nnz), _p(hang), _p(ptr), _p(master), _p(w)))
        return hang, ptr, master, w

    def clear_boundary_faces(self, mask):
        _chk(self.L.fh_mesh_clear_boundary_faces(self.h, int(mask)))

    def set_coords(self, coords):
        c = _f64(coords)
        assert c.shape == (self.nnode, self.dim)
        _chk(self.L.fh_mesh_set_coords(self.h, _p(c)))

    def destroy(self):
        if self.h:
            self.L.fh_mesh_destroy(self.h)
            self.h = None

    def arrays(self):
        ed = np.empty((self.nel, self.nloc), np.int32)
        xy = np.empty((self.nnode, self.dim))
        ff = np.empty((self.nel, self.nfaces), np.int32)
        _chk(self.L.fh_mesh_get(self.h, _p(ed), _p(xy), _p(ff)))
        return ed, xy, ff

    def child_elems(self):
        out = np.empty((self.nel, 2 ** self.dim), np.int32)
        _chk(self.L.fh_mesh_child_elems(self.h, _p(out)))
        return out

    def n_dofs(self, fe):
        """Mesh::GetSolutionDof on one process: the linear / serendipity families own the leading vertex / vertex + edge node ids, the piecewise constant one the elements"""
        return {"linear": self.own_size[0], "serendipity": self.own_size[1], "biquadratic": self.nnode, "constant": self.nel}[fe]

    def dirichlet_dofs(self, fe):
        n = ctypes.c_int(self.nnode)
        out = np.empty(self.nnode, np.int32)
        _chk(self.L.fh_mesh_dirichlet_dofs(self.h, FE[fe], ctypes.byref(n), _p(out)))
        return out[:n.value].copy()


def pattern_from_elements(elem_dof, ndof):
    L = load_library()
    ed = _i32(elem_dof)
    nel, nloc = ed.shape
    rowptr = np.empty(ndof + 1, np.int32)
    _chk(L.fh_pattern_from_elements(nel, nloc, _p(ed), ndof, _p(rowptr), None))
    col = np.empty(rowptr[-1], np.int32)
    _chk(L.fh_pattern_from_elements(nel, nloc, _p(ed), ndof, _p(rowptr), _p(col)))
    return rowptr, col


_DIM = {"hex": 3, "quad": 2, "line": 1, "tri": 2, "tet": 3, "wedge": 3}


def fe_gauss(geom, order):
    L = load_library()
    dim = _DIM[geom]
    ng = ctypes.c_int()
    _chk(L.fh_fe_gauss(GEOM[geom], GAUSS_ORDER[order], ctypes.byref(ng), None, None))
    w, x = np.empty(ng.value), np.empty((dim, ng.value))
    _chk(L.fh_fe_gauss(GEOM[geom], GAUSS_ORDER[order], ctypes.byref(ng), _p(w), _p(x)))
    return w, x.T.copy()


def fe_tables(geom, fe, order):
    L = load_library()
    dim = _DIM[geom]
    ng, nc = ctypes.c_int(), ctypes.c_int()
    _chk(L.fh_fe_tables(GEOM[geom], FE[fe], GAUSS_ORDER[order], ctypes.byref(ng), ctypes.byref(nc), None, None))
    phi, dphi = np.empty((ng.value, nc.value)), np.empty((dim, ng.value, nc.value))
    _chk(L.fh_fe_tables(GEOM[geom], FE[fe], GAUSS_ORDER[order], ctypes.byref(ng), ctypes.byref(nc), _p(phi), _p(dphi)))
    return phi, np.transpose(dphi, (1, 2, 0)).copy()


def fe_tables_d2(geom, fe, order):
    """second derivatives at the Gauss points, [ng, nc, nh]: (xx) in 1-D, (xx, yy, xy) in 2-D, (xx, yy, zz, xy, yz, zx) in 3-D"""
    L = load_library()
    nh = {"hex": 6, "quad": 3, "line": 1, "tri": 3, "tet": 6, "wedge": 6}[geom]
    phi, _ = fe_tables(geom, fe, order)
    d2 = np.empty((nh,) + phi.shape)
    _chk(L.fh_fe_tables_d2(GEOM[geom], FE[fe], GAUSS_ORDER[order], _p(d2)))
    return np.transpose(d2, (1, 2, 0)).copy()


def fe_jacobian(ctx, mesh, fe, order="seventh", hessians=False, elem_dof=None, coords=None):
    """elem_type::Jacobian for every element and Gauss point (fh_fe_jacobian): weight[nel, ng], gradphi[nel, ng, nc, dim] and, when asked,
    nablaphi[nel, ng, nc, nh]"""
    ed, xy, _ = mesh.arrays()
    ed = _i32(ed if elem_dof is None else elem_dof)
    xy = _f64(xy if coords is None else coords)
    phi, _ = fe_tables(mesh.geom, fe, order)
    ng, nc = phi.shape
    dim, nh = mesh.dim, (3 if mesh.dim == 2 else 6)
    nel = ed.shape[0]
    w, g = np.empty((nel, ng)), np.empty((nel, ng, nc, dim))
    n = np.empty((nel, ng, nc, nh)) if hessians else None
    _chk(ctx.L.fh_fe_jacobian(ctx.h, GEOM[mesh.geom], FE[fe], GAUSS_ORDER[order], nel, ed.shape[1], _p(ed), xy.shape[0], _p(xy), _p(w), _p(g),
                              _p(n) if hessians else None))
    return (w, g, n) if hessians else (w, g)


def fe_elem_prolongator(geom, fe):
    L = load_library()
    nch, nc = ctypes.c_int(), ctypes.c_int()
    _chk(L.fh_fe_elem_prolongator(GEOM[geom], FE[fe], ctypes.byref(nch), ctypes.byref(nc), None))
    P = np.empty((nch.value, nc.value, nc.value))
    _chk(L.fh_fe_elem_prolongator(GEOM[geom], FE[fe], ctypes.byref(nch), ctypes.byref(nc), _p(P)))
    return P


def fe_node_ref_coords(geom, node):
    """reference coordinates of a local node as doubles (any element: the simplices' are 0, 1/2, 1, 1/3)"""
    L = load_library()
    out = np.zeros(3)
    _chk(L.fh_fe_node_ref_coords(GEOM[geom], int(node), _p(out)))
    return out[:_DIM[geom]]


def fe_node_ref(geom, node, d=None):
    """reference coordinates (-1, 0, 1) of a local node of the biquadratic element"""
    dim = 3 if geom == "hex" else 2
    out = np.zeros(3, np.int32)
    _chk(load_library().fh_fe_node_ref(GEOM[geom], int(node), _p(out)))
    return out[:dim].copy() if d is None else int(out[d])


def fe_face_nodes(geom, fe, face):
    L = load_library()
    n = ctypes.c_int()
    out = np.empty(9, np.int32)
    _chk(L.fh_fe_face_nodes(GEOM[geom], FE[fe], int(face), ctypes.byref(n), _p(out)))
    return out[:n.value].copy()


def assemble_poisson_rows(ctx, geom, fe, elem_dof, coords, K, res, sol=None, source=None, scale=1.0, order="seventh"):
    """the Poisson callback through the generic (dim, nc, ng) kernel (fh_assemble_poisson_rows): K_ij = sum grad phi_i . grad phi_j w, res_i = sum (scale f phi_i -
    grad phi_i . grad sol) w; elem_dof[nel, nloc] in the family's local order, node classes numbered one after the other"""
    ed, x = _i32(elem_dof), _f64(coords)
    _chk(ctx.L.fh_assemble_poisson_rows(ctx.h, GEOM[geom], FE[fe], GAUSS_ORDER[order], ed.shape[0], ed.shape[1], _p(ed), x.shape[0], _p(x),
                                        sol.h if sol is not None else None, source.h if source is not None else None, float(scale), K.h, res.h))


def assemble_poisson_mixed(ctx, fe, elem_geom, elem_dof, coords, K, res, sol=None, source=None, scale=1.0, order="seventh"):
    """the same on a mesh of mixed shapes (fh_assemble_poisson_mixed): elem_geom[nel] names of GEOM per element, elem_dof[nel, nloc] padded (the padding is not read)"""
    eg = _i32(np.array([GEOM[g] for g in elem_geom]))
    ed, x = _i32(np.where(np.asarray(elem_dof) < 0, 0, elem_dof)), _f64(coords)
    _chk(ctx.L.fh_assemble_poisson_mixed(ctx.h, FE[fe], GAUSS_ORDER[order], ed.shape[0], ed.shape[1], _p(eg), _p(ed), x.shape[0], _p(x),
                                         sol.h if sol is not None else None, source.h if source is not None else None, float(scale), K.h, res.h))


def assemble_advdiff_line(ctx, fe, elem_dof, coords, K, res, nu, velocity, sol=None, source=None, order="seventh"):
    """the 001_Poisson callback on a one-dimensional EDGE3 mesh (main.cpp:355-480 with dim == 1: advection-diffusion with its streamline-upwind terms):
    K <- Jacobian, res <- residual.  elem_dof[nel, 3] node ids (ends, then middle; vertices numbered first), coords[nnode]"""
    ed, x = _i32(elem_dof), _f64(coords)
    _chk(ctx.L.fh_assemble_advdiff_line(ctx.h, FE[fe], GAUSS_ORDER[order], ed.shape[0], _p(ed), x.size, _p(x), sol.h if sol is not None else None, float(nu),
                                        float(velocity), source.h if source is not None else None, K.h, res.h))


def assemble_neumann(ctx, mesh, fe, res, flux_by_flag, order="seventh"):
    """boundary term of 001_Poisson: faces whose boundary flag is a key of flux_by_flag carry the Neumann flux tau -- a number (the
    mesh-file branch, main.cpp:556-594) or an Expr evaluated at every face Gauss point (the parsed-function branch, main.cpp:495-553)"""
    ed, xy, ff = mesh.arrays()
    xy = _f64(xy)
    for parsed in (False, True):
        faces, taus, exprs = [], [], []
        for f in range(mesh.nfaces):
            loc = fe_face_nodes(mesh.geom, fe, f)
            for flag, tau in flux_by_flag.items():
                if isinstance(tau, Expr) != parsed:
                    continue
                els = np.where(ff[:, f] == flag)[0]
                if els.size:
                    faces.append(ed[els][:, loc])
                    if parsed:
                        if tau not in exprs:
                            exprs.append(tau)
                        taus.append(np.full(els.size, exprs.index(tau)))
                    else:
                        taus.append(np.full(els.size, float(tau)))
        if not faces:
            continue
        fn = _i32(np.concatenate(faces))
        if parsed:
            fx = _i32(np.concatenate(taus))
            hs = (ctypes.c_void_p * len(exprs))(*[e.h for e in exprs])
            _chk(ctx.L.fh_assemble_neumann_faces_expr(ctx.h, GEOM[mesh.geom], FE[fe], GAUSS_ORDER[order], fn.shape[0], _p(fn), _p(fx), len(exprs), hs,
                                                      xy.shape[0], _p(xy), res.h))
        else:
            tv = _f64(np.concatenate(taus))
            _chk(ctx.L.fh_assemble_neumann_faces(ctx.h, GEOM[mesh.geom], FE[fe], GAUSS_ORDER[order], fn.shape[0], _p(fn), _p(tv), xy.shape[0], _p(xy), res.h))


def assemble_neumann_edges(ctx, fe, face_nodes, face_expr, exprs, coords, res, order="seventh"):
    """edge integrals of a parsed flux on explicitly listed EDGES of a two-dimensional mesh of any element shape (triangles: the caller keeps the mesh):
    face_nodes[nfaces, nfn] in the line element's order (ends, then middle), face_expr[nfaces] an index into exprs; fh_assemble_neumann_faces_expr with the
    quadrilateral's face geometry (a line)"""
    fn, fx, xy = _i32(face_nodes), _i32(face_expr), _f64(coords)
    if fn.shape[0] == 0:
        return
    hs = (ctypes.c_void_p * len(exprs))(*[e.h for e in exprs])
    _chk(ctx.L.fh_assemble_neumann_faces_expr(ctx.h, GEOM["quad"], FE[fe], GAUSS_ORDER[order], fn.shape[0], _p(fn), _p(fx), len(exprs), hs, xy.shape[0], _p(xy),
                                              res.h))


def assemble_neumann_faces(ctx, geom, fe, face_nodes, tau, coords, res, order="seventh"):
    """face integrals of a constant flux per face on explicitly listed faces of a mesh the caller keeps (tetrahedra: TRI3 / TRI6 faces in the face element's
    order, vertices then middles): res[node] += int phi tau dS (fh_assemble_neumann_faces)"""
    fn, tv, xy = _i32(face_nodes), _f64(tau), _f64(coords)
    if fn.shape[0] == 0:
        return
    g = {"quadface": 101, "lineface": 102, "triface": 103}.get(geom)           # the face element itself named (prisms have faces of two kinds)
    _chk(ctx.L.fh_assemble_neumann_faces(ctx.h, GEOM[geom] if g is None else g, FE[fe], GAUSS_ORDER[order], fn.shape[0], _p(fn), _p(tv), xy.shape[0], _p(xy), res.h))


def face_normals(mesh, fe, face_nodes, gauss_point=0, order="seventh", coords=None):
    """unit normals of faces at one face Gauss point as elem_type::JacobianSur returns them (fh_fe_face_normals, host)"""
    L = load_library()
    xy = _f64(mesh.arrays()[1] if coords is None else coords)
    fn = _i32(face_nodes)
    out = np.empty((fn.shape[0], mesh.dim))
    _chk(L.fh_fe_face_normals(GEOM[mesh.geom], FE[fe], GAUSS_ORDER[order], int(gauss_point), fn.shape[0], _p(fn), xy.shape[0], _p(xy), _p(out)))
    return out


def assemble_pressure_faces(ctx, mesh, res, face_nodes, tau, comp_offset, scale=-1.0, order="seventh"):
    """open-boundary pressure term of the Navier-Stokes residual (03_navier_stokes.hpp:185-290) on the listed faces (Q2 face nodes, face element
    order): res[comp_offset[k] + node] += scale * int_face phi tau n_k.  tau: one number per face, or a list of (Expr, face mask) pairs"""
    xy = _f64(mesh.arrays()[1])
    fn = _i32(face_nodes)
    if fn.shape[0] == 0:
        return
    off = _i32(comp_offset)
    if isinstance(tau, (list, tuple)) and tau and isinstance(tau[0], tuple):
        exprs = [e for e, _ in tau]
        fx = np.full(fn.shape[0], -1, np.int32)
        for k, (_, mask) in enumerate(tau):
            fx[np.asarray(mask, bool)] = k
        assert (fx >= 0).all(), "every face needs an expression"
        hs = (ctypes.c_void_p * len(exprs))(*[e.h for e in exprs])
        _chk(ctx.L.fh_assemble_pressure_faces(ctx.h, GEOM[mesh.geom], GAUSS_ORDER[order], fn.shape[0], _p(fn), None, _p(fx), len(exprs), hs, xy.shape[0],
                                              _p(xy), _p(off), float(scale), res.h))
    else:
        tv = _f64(np.broadcast_to(np.asarray(tau, float), (fn.shape[0],)).copy())
        _chk(ctx.L.fh_assemble_pressure_faces(ctx.h, GEOM[mesh.geom], GAUSS_ORDER[order], fn.shape[0], _p(fn), _p(tv), None, 0, None, xy.shape[0], _p(xy),
                                              _p(off), float(scale), res.h))


def build_prolongator(ctx, coarse, fine, fe, zero_bdc=True):
    h = ctypes.c_void_p()
    _chk(ctx.L.fh_build_prolongator(ctx.h, coarse.h, fine.h, FE[fe], 1 if zero_bdc else 0, ctypes.byref(h)))
    return Mat(ctx, h)


def build_amr_prolongator(ctx, mesh, fe):
    """LinearImplicitSystem::BuildAmrProlongatorMatrix: P_amr (n x n) of a non-homogeneous level"""
    h = ctypes.c_void_p()
    _chk(ctx.L.fh_build_amr_prolongator(ctx.h, mesh.h, FE[fe], ctypes.byref(h)))
    return Mat(ctx, h)


class Assembler:
    """batched Poisson assembly (the element loop of 00_poisson_eqn_..._separate.hpp:106-228 as one call)."""

    def __init__(self, ctx, mesh, fe, A, order="seventh", elem_dof=None, coords=None):
        self.ctx, self.L = ctx, ctx.L
        if elem_dof is None and mesh is not None:         # element table and coordinates from the mesh's device copy
            self.h = ctypes.c_void_p()
            _chk(self.L.fh_assembler_create_mesh(ctx.h, mesh.h, FE[fe], GAUSS_ORDER[order], A.h, ctypes.byref(self.h)))
            self.nel = mesh.nel
            self.nc = {"linear": 2 ** mesh.dim, "serendipity": 8 if mesh.dim == 2 else 20, "biquadratic": 3 ** mesh.dim}[fe]
            return
        ed, xy = _i32(elem_dof), _f64(coords)
        assert ed.min() >= 0 and ed.max() < xy.shape[0]
        geom = "hex" if xy.shape[1] == 3 else "quad"
        self.h = ctypes.c_void_p()
        _chk(self.L.fh_assembler_create(ctx.h, GEOM[geom], FE[fe], GAUSS_ORDER[order], ed.shape[0], ed.shape[1], _p(ed),
                                        xy.shape[0], _p(xy), A.h, ctypes.byref(self.h)))
        self.nel = ed.shape[0]
        self.nc = {"linear": 2 ** xy.shape[1], "serendipity": 8 if xy.shape[1] == 2 else 20, "biquadratic": 3 ** xy.shape[1]}[fe]

    def destroy(self):
        if self.h:
            self.L.fh_assembler_destroy(self.h)
            self.h = None

    def assemble(self, A, res, sol=None, source_kind=0, params=(1.0,)):
        p = _f64(list(params) + [0.0] * (4 - len(params)))
        _chk(self.L.fh_assemble_poisson(self.h, None if sol is None else sol.h, int(source_kind), _p(p), A.h, res.h))

    def galerkin_from(self, fine, child, fine_bdc, coarse_bdc, Ac):
        """Ac = PP^T KK PP element by element from the element matrices `fine` holds (fh_assembler_galerkin); self = the coarse level"""
        ch, fb, cb = _i32(child), _i32(fine_bdc), _i32(coarse_bdc)
        _chk(self.L.fh_assembler_galerkin(fine.h, self.h, _p(ch), fb.size, _p(fb), cb.size, _p(cb), Ac.h))

    def affine_count(self):
        """(elements the affine fast path would take, elements that need quadrature)"""
        a, g = ctypes.c_int(), ctypes.c_int()
        _chk(self.L.fh_assembler_affine_count(self.h, ctypes.byref(a), ctypes.byref(g)))
        return a.value, g.value

    def last_path(self):
        """what the last assembly ran: "fused", "two-pass" or None"""
        v = ctypes.c_int()
        _chk(self.L.fh_assembler_last_path(self.h, ctypes.byref(v)))
        return {1: "fused", 2: "two-pass"}.get(v.value)

    def fused_info(self):
        """fused cluster assembly: is it the path that runs, clusters, doubles in the partial-row buffer, rows of the second pass"""
        a, n, r = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        pe = ctypes.c_int64()
        _chk(self.L.fh_assembler_fused_info(self.h, ctypes.byref(a), ctypes.byref(n), ctypes.byref(pe), ctypes.byref(r)))
        cs, ce = ctypes.c_int(), ctypes.c_int64()
        _chk(self.L.fh_assembler_carry_info(self.h, ctypes.byref(cs), ctypes.byref(ce)))
        return {"active": bool(a.value), "clusters": n.value, "partial_entries": pe.value, "second_pass_rows": r.value,
                "clusters_per_super": cs.value, "carried_entries": ce.value}

    def assemble_expr(self, A, res, sol, expr, scale=1.0):
        """source term f = scale * expr(x, y, z, t), evaluated on the device at the Gauss points"""
        _chk(self.L.fh_assemble_poisson_expr(self.h, None if sol is None else sol.h, expr.h, float(scale), A.h, res.h))

    def element_matrices(self, sol=None, source_kind=0, params=(1.0,)):
        p = _f64(list(params) + [0.0] * (4 - len(params)))
        K, F = np.empty((self.nel, self.nc, self.nc)), np.empty((self.nel, self.nc))
        _chk(self.L.fh_element_matrices_poisson(self.h, None if sol is None else sol.h, int(source_kind), _p(p), _p(K), _p(F)))
        return K, F

    def info(self, colors=True):
        """colors: also the number of element colours of the coloured scatter (made on demand: the default paths need none)"""
        nco, by, fl = ctypes.c_int(), ctypes.c_int64(), ctypes.c_double()
        _chk(self.L.fh_assembler_info(self.h, ctypes.byref(nco) if colors else None, ctypes.byref(by), ctypes.byref(fl)))
        return {"ncolors": nco.value if colors else None, "algorithmic_bytes": by.value, "flops": fl.value}


class Direct:
    """sparse exact solve (fh_direct_*: multifrontal factorisation over a nested-dissection tree; symmetric operators on unpivoted fronts, unsymmetric /
    indefinite ones on pivoted fronts)"""

    def __init__(self, ctx, A, coords=None, leaf=0):
        self.ctx, self.L, self.A = ctx, ctx.L, A
        self.h = ctypes.c_void_p()
        if coords is not None:
            xy = _f64(np.ascontiguousarray(coords))
            dim = xy.shape[1] if xy.ndim == 2 else 1
            _chk(self.L.fh_direct_create(ctx.h, A.h, int(dim), _p(xy), int(leaf), ctypes.byref(self.h)))
        else:
            _chk(self.L.fh_direct_create(ctx.h, A.h, 0, None, int(leaf), ctypes.byref(self.h)))

    def factor(self):
        _chk(self.L.fh_direct_factor(self.h))
        return self

    def solve(self, b, x):
        _chk(self.L.fh_direct_solve(self.h, b.h, x.h))

    def set_general(self, on=True):
        """force the pivoted (general) fronts also for a symmetric operator"""
        _chk(self.L.fh_direct_set_general(self.h, 1 if on else 0))
        return self

    def stats(self):
        g, p, r = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _chk(self.L.fh_direct_stats(self.h, ctypes.byref(g), ctypes.byref(p), ctypes.byref(r)))
        return {"general": bool(g.value), "perturbed_pivots": p.value, "refinement_steps": r.value}

    def info(self):
        a, f, h, lf = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        fd = ctypes.c_int64()
        _chk(self.L.fh_direct_info(self.h, ctypes.byref(a), ctypes.byref(f), ctypes.byref(h), ctypes.byref(lf), ctypes.byref(fd)))
        return {"coupled": a.value, "fronts": f.value, "height": h.value, "largest_front": lf.value, "factor_doubles": fd.value}

    def destroy(self):
        if self.h:
            self.L.fh_direct_destroy(self.h)
            self.h = None


class Index:
    """device-resident index list (the _bdcIndex of a level): SetPenalty / ZerosBoundaryResiduals without host traffic"""

    def __init__(self, ctx, idx):
        self.ctx, self.L = ctx, ctx.L
        idx = _i32(idx)
        self.n = idx.size
        self.h = ctypes.c_void_p()
        _chk(self.L.fh_index_create(ctx.h, idx.size, _p(idx), ctypes.byref(self.h)))

    @classmethod
    def from_handle(cls, ctx, handle, n):
        self = cls.__new__(cls)
        self.ctx, self.L, self.h, self.n = ctx, ctx.L, handle, n
        return self

    def zero_rows(self, A, diag):
        _chk(self.L.fh_mat_zero_rows_index(A.h, self.h, float(diag)))

    def set(self, v, value):
        _chk(self.L.fh_vec_set_index(v.h, self.h, float(value)))

    def gather_matrix_values(self, dst, src):
        """dst.val[k] = src.val[self[k]] (self[k] = -1: 0)"""
        _chk(self.L.fh_mat_gather_values(dst.h, src.h, self.h))

    def gather_vector(self, dst, src):
        _chk(self.L.fh_vec_gather(dst.h, src.h, self.h))

    def destroy(self):
        if self.h:
            self.L.fh_index_destroy(self.h)
            self.h = None


class Expr:
    """femus::ParsedFunction: a run-time expression compiled to a postfix program (host and device evaluation)"""

    def __init__(self, expression, variables="x,y,z,t"):
        self.L = load_library()
        self.h = ctypes.c_void_p()
        self.nvars = len([v for v in variables.split(",") if v.strip()])
        _chk(self.L.fh_expr_compile(expression.encode(), variables.encode(), ctypes.byref(self.h)))

    def __call__(self, x):
        x = _f64(x)
        if x.ndim == 1:
            out = ctypes.c_double()
            assert x.size >= self.nvars
            _chk(self.L.fh_expr_eval(self.h, _p(x), ctypes.byref(out)))
            return out.value
        assert x.shape[1] == self.nvars
        out = np.empty(x.shape[0])
        _chk(self.L.fh_expr_eval_many(self.h, x.shape[0], _p(x), _p(out)))
        return out

    def program(self):
        """(code, consts): the postfix program the device evaluator runs (fh_expr_program)"""
        nc, nk = ctypes.c_int(), ctypes.c_int()
        _chk(self.L.fh_expr_program(self.h, ctypes.byref(nc), ctypes.byref(nk), None, None))
        code, consts = np.empty(nc.value, np.int32), np.empty(nk.value)
        _chk(self.L.fh_expr_program(self.h, ctypes.byref(nc), ctypes.byref(nk), _p(code), _p(consts)))
        return code, consts

    def n_variables(self):
        n = ctypes.c_int()
        _chk(self.L.fh_expr_nvars(self.h, ctypes.byref(n)))
        return n.value

    def destroy(self):
        if self.h:
            self.L.fh_expr_destroy(self.h)
            self.h = None


def system_elem_dofs(mesh, fes):
    """LinearEquation::GetSystemDof for stacked variables: (nd, offsets[nvars+1], elem_sys[nel, nd])"""
    fe = _i32([FE[f] for f in fes])
    nd = ctypes.c_int()
    off = np.empty(len(fes) + 1, np.int32)
    _chk(mesh.L.fh_system_elem_dofs(mesh.h, len(fes), _p(fe), ctypes.byref(nd), _p(off), None))
    es = np.empty((mesh.nel, nd.value), np.int32)
    _chk(mesh.L.fh_system_elem_dofs(mesh.h, len(fes), _p(fe), ctypes.byref(nd), _p(off), _p(es)))
    return nd.value, off, es


def build_system_prolongator(ctx, coarse, fine, fes):
    fe = _i32([FE[f] for f in fes])
    h = ctypes.c_void_p()
    _chk(ctx.L.fh_build_system_prolongator(ctx.h, coarse.h, fine.h, len(fes), _p(fe), ctypes.byref(h)))
    return Mat(ctx, h)


def vertex_patches(mesh, fes):
    """blocks of the Vanka smoother: (ptr[npatch+1], dofs)"""
    fe = _i32([FE[f] for f in fes])
    n, tot = ctypes.c_int(0), ctypes.c_int(0)
    _chk(mesh.L.fh_mesh_vertex_patches(mesh.h, len(fes), _p(fe), ctypes.byref(n), ctypes.byref(tot), None, None))
    ptr, dofs = np.empty(n.value + 1, np.int32), np.empty(tot.value, np.int32)
    _chk(mesh.L.fh_mesh_vertex_patches(mesh.h, len(fes), _p(fe), ctypes.byref(n), ctypes.byref(tot), _p(ptr), _p(dofs)))
    return ptr, dofs


class NSAssembler:
    """batched steady Navier-Stokes residual + Newton Jacobian, Taylor-Hood (03_navier_stokes.hpp:187-413 as one call)"""

    def __init__(self, ctx, mesh, A, order="seventh"):
        self.ctx, self.L = ctx, ctx.L
        ed, xy, _ = mesh.arrays()
        self.nel = mesh.nel
        self.nd = mesh.dim * mesh.nloc + 2 ** mesh.dim
        self.h = ctypes.c_void_p()
        _chk(self.L.fh_ns_assembler_create(ctx.h, GEOM[mesh.geom], GAUSS_ORDER[order], mesh.nel, mesh.nloc, _p(ed), mesh.nnode,
                                           mesh.own_size[0], _p(xy), A.h, ctypes.byref(self.h)))

    def assemble(self, A, res, sol, nu):
        _chk(self.L.fh_assemble_navier_stokes(self.h, None if sol is None else sol.h, ctypes.c_double(nu), A.h, res.h))

    def element_matrices(self, sol, nu):
        K, F = np.empty((self.nel, self.nd, self.nd)), np.empty((self.nel, self.nd))
        _chk(self.L.fh_ns_element_matrices(self.h, None if sol is None else sol.h, ctypes.c_double(nu), _p(K), _p(F)))
        return K, F

    def destroy(self):
        if self.h:
            self.L.fh_ns_assembler_destroy(self.h)
            self.h = None


class NSPwAssembler(NSAssembler):
    """the same weak form with the DISCONTINUOUS piecewise-linear pressure of unittests/testNSSteadyDD (fh_ns_pw_assembler_create): variables
    [U | V | (W) | P], pressure dof of local function i of element e at dim * nnode + i * nel + e"""

    def __init__(self, ctx, mesh, A, order="seventh"):
        self.ctx, self.L = ctx, ctx.L
        ed, xy, _ = mesh.arrays()
        self.nel = mesh.nel
        self.nd = mesh.dim * mesh.nloc + mesh.dim + 1
        self.h = ctypes.c_void_p()
        _chk(self.L.fh_ns_pw_assembler_create(ctx.h, GEOM[mesh.geom], GAUSS_ORDER[order], mesh.nel, mesh.nloc, _p(ed), mesh.nnode, _p(xy), A.h,
                                              ctypes.byref(self.h)))

    @staticmethod
    def elem_sys(mesh):
        """[nel, nd] system dofs of every element (the table the pattern of KK is made from)"""
        ed = mesh.arrays()[0]
        p = np.arange(mesh.dim + 1)[None, :] * mesh.nel + np.arange(mesh.nel)[:, None] + mesh.dim * mesh.nnode
        return np.concatenate([ed + k * mesh.nnode for k in range(mesh.dim)] + [p], axis=1).astype(np.int32)


class AdvDiffAssembler(NSAssembler):
    """scalar LAGRANGE SECOND advection-diffusion in a given velocity field: the temperature callback of unittests/testNSSteadyDD (AssembleMatrixResT)"""

    def __init__(self, ctx, mesh, A, order="seventh"):
        self.ctx, self.L = ctx, ctx.L
        ed, xy, _ = mesh.arrays()
        self.nel = mesh.nel
        self.nd = mesh.nloc
        self.h = ctypes.c_void_p()
        _chk(self.L.fh_advdiff_assembler_create(ctx.h, GEOM[mesh.geom], GAUSS_ORDER[order], mesh.nel, mesh.nloc, _p(ed), mesh.nnode, _p(xy), A.h,
                                                ctypes.byref(self.h)))

    def assemble(self, A, res, sol, velocity, inverse_peclet):
        _chk(self.L.fh_assemble_advection_diffusion(self.h, None if sol is None else sol.h, None if velocity is None else velocity.h,
                                                    ctypes.c_double(inverse_peclet), A.h, res.h))


class NSStabAssembler(NSAssembler):
    """the callback of applications/003_NavierStokes/SteadyNavierStokesParallel (main.cpp:390-925): equal-order linear velocity / pressure on the vertex
    nodes with the Franca-Frey stabilisation; variables [U | V | (W) | P], each `nq1` long"""

    def __init__(self, ctx, mesh, A, order="seventh"):
        self.ctx, self.L = ctx, ctx.L
        ed, xy, _ = mesh.arrays()
        self.nel = mesh.nel
        self.nq1 = mesh.own_size[0]
        self.nd = (mesh.dim + 1) * 2 ** mesh.dim
        self.h = ctypes.c_void_p()
        _chk(self.L.fh_ns_stab_assembler_create(ctx.h, GEOM[mesh.geom], GAUSS_ORDER[order], mesh.nel, mesh.nloc, _p(ed), mesh.nnode, self.nq1, _p(xy), A.h,
                                                ctypes.byref(self.h)))

    def assemble(self, A, res, sol, inverse_reynolds):
        _chk(self.L.fh_assemble_navier_stokes_stab(self.h, None if sol is None else sol.h, ctypes.c_double(inverse_reynolds), A.h, res.h))


class Multigrid:
    """LinearEquationSolver MG interface: MGInit / MGSetLevel / MGSolve / MGClear."""

    def __init__(self, ctx, nlevels):
        self.ctx, self.L = ctx, ctx.L
        self.h = ctypes.c_void_p()
        _chk(self.L.fh_mg_create(ctx.h, int(nlevels), ctypes.byref(self.h)))

    def set_level(self, level, A, P=None, R=None, smoother=0, omega=2. / 3., npre=2, npost=2):
        _chk(self.L.fh_mg_set_level(self.h, int(level), A.h, None if P is None else P.h, None if R is None else R.h,
                                    int(smoother), float(omega), int(npre), int(npost)))

    def set_level_patches(self, level, ptr, dofs):
        """dof patches of the block Schwarz (Vanka) smoother of a level (smoother=2)"""
        ptr, dofs = _i32(ptr), _i32(dofs)
        _chk(self.L.fh_mg_set_level_patches(self.h, int(level), ptr.size - 1, _p(ptr), _p(dofs)))

    def set_level_patches_exact(self, level, nfirst):
        """FH_SMOOTH_ASM: the first nfirst blocks get the exact sub-solve instead of ILU(0) (FEMuS_ASM's solid / porous blocks)"""
        _chk(self.L.fh_mg_set_level_patches_exact(self.h, int(level), int(nfirst)))

    def set_level_solver(self, level, solver="gmres", restart=30):
        """level solver of the smoother: "richardson" (fixed sweeps, the default) or "gmres" (fixed iterations, left-preconditioned)"""
        _chk(self.L.fh_mg_set_level_solver(self.h, int(level), {"richardson": 0, "gmres": 1}[solver], int(restart)))

    def set_level_distributed(self, level, halo, replicated_below=False):
        _chk(self.L.fh_mg_set_level_distributed(self.h, int(level), None if halo is None else halo.h, 1 if replicated_below else 0))

    def coarse_info(self):
        """(unknowns in the dense coarse problem, interior blocks of its dissection -- 0: one dense inverse --, separator size, largest block)"""
        v = [ctypes.c_int() for _ in range(4)]
        _chk(self.L.fh_mg_coarse_info(self.h, *[ctypes.byref(x) for x in v]))
        return tuple(x.value for x in v)

    def set_coarse_coords(self, coords):
        """coordinates of the unknowns of level 0: the exact coarse solve then dissects its dense problem (fh_mg_set_coarse_coords)"""
        xy = _f64(coords)
        _chk(self.L.fh_mg_set_coarse_coords(self.h, xy.shape[1], xy.shape[0], _p(xy)))


    def set_level_coords(self, level, coords):
        """coordinates of the unknowns of a level whose preconditioner is the exact solve (SMOOTH_LU); level 0 = set_coarse_coords"""
        xy = _f64(np.ascontiguousarray(coords))
        dim = xy.shape[1] if xy.ndim == 2 else 1
        _chk(self.L.fh_mg_set_level_coords(self.h, int(level), int(dim), xy.shape[0], _p(xy)))
    def setup(self):
        _chk(self.L.fh_mg_setup(self.h))

    def set_cycle_type(self, kind):
        """PCMGSetType: "multiplicative" (default), "full", "additive", "kaskade" """
        _chk(self.L.fh_mg_set_cycle_type(self.h, {"multiplicative": 0, "full": 1, "additive": 2, "kaskade": 3}[kind]))

    def vcycle(self, b, x):
        _chk(self.L.fh_mg_vcycle(self.h, b.h, x.h))

    def solve(self, b, x, outer="gmres", rtol=1e-10, atol=1e-50, dtol=1e50, maxit=100, restart=30):
        its, rn = ctypes.c_int(), ctypes.c_double()
        _chk(self.L.fh_mg_solve(self.h, b.h, x.h, OUTER[outer], float(rtol), float(atol), float(dtol), int(maxit), int(restart),
                                ctypes.byref(its), ctypes.byref(rn)))
        return its.value, rn.value

    def cycle_algorithmic_bytes(self):
        return int(self.L.fh_mg_cycle_algorithmic_bytes(self.h))

    def destroy(self):
        if self.h:
            self.L.fh_mg_destroy(self.h)
            self.h = None


def version():
    return load_library().fh_version().decode()


class Halo:
    """neighbour exchange plan over RCCL (VecGhostUpdate / MPIAIJ scatter replacement); one rank per GPU"""

    def sizes(self):
        """(entries this rank sends, ghost entries it receives) per exchange"""
        a, b = ctypes.c_int(), ctypes.c_int()
        _chk(self.L.fh_halo_sizes(self.h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def __init__(self, ctx, rank, nranks, unique_id, send_counts, send_idx, recv_counts, parent=None):
        self.ctx, self.L = ctx, ctx.L
        sc, si, rc = _i32(send_counts), _i32(send_idx), _i32(recv_counts)
        self.h = ctypes.c_void_p()
        if parent is not None:     # same communicator, another exchange plan
            _chk(self.L.fh_halo_create_shared(parent.h, _p(sc), _p(si), _p(rc), ctypes.byref(self.h)))
        else:
            uid = ctypes.create_string_buffer(bytes(unique_id), 128)
            _chk(self.L.fh_halo_create(ctx.h, int(rank), int(nranks), uid, _p(sc), _p(si), _p(rc), ctypes.byref(self.h)))

    @classmethod
    def host(cls, ctx, rank, nranks, comm, send_counts, send_idx, recv_counts, parent=None):
        """same exchange plan over a host-staged transport: `comm` provides alltoallv(list of arrays, dtype) and
        allreduce_sum(array) (SocketComm / TorchComm of femus_amd.dd, or an MPI wrapper)"""
        self = cls.__new__(cls)
        self.ctx, self.L = ctx, ctx.L
        sc, si, rc = _i32(send_counts), _i32(send_idx), _i32(recv_counts)
        self.h = ctypes.c_void_p()
        if parent is not None:
            _chk(self.L.fh_halo_create_shared(parent.h, _p(sc), _p(si), _p(rc), ctypes.byref(self.h)))
            return self
        EX = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int),
                              ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int))
        AR = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.c_int)

        def exchange(user, send, scnt, recv, rcnt):
            try:
                parts, off = [], 0
                for r in range(nranks):
                    parts.append(np.ctypeslib.as_array(send, shape=(off + scnt[r],))[off:off + scnt[r]].copy() if scnt[r] else np.zeros(0))
                    off += scnt[r]
                got = comm.alltoallv(parts, np.float64)
                off = 0
                for r in range(nranks):
                    if rcnt[r]:
                        assert got[r].size == rcnt[r]
                        np.ctypeslib.as_array(recv, shape=(off + rcnt[r],))[off:off + rcnt[r]] = got[r]
                    off += rcnt[r]
                return 0
            except Exception:       # never let an exception cross the C boundary
                import traceback
                traceback.print_exc()
                return 1

        def allreduce(user, buf, n):
            try:
                a = np.ctypeslib.as_array(buf, shape=(n,))
                a[:] = comm.allreduce_sum(a.copy())
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return 1

        self._callbacks = (EX(exchange), AR(allreduce))           # keep the trampolines alive as long as the plan
        _chk(self.L.fh_halo_create_host(ctx.h, int(rank), int(nranks), self._callbacks[0], self._callbacks[1], None, _p(sc), _p(si), _p(rc),
                                        ctypes.byref(self.h)))
        return self

    @staticmethod
    def unique_id():
        buf = ctypes.create_string_buffer(128)
        _chk(load_library().fh_halo_unique_id(buf))
        return buf.raw

    def update(self, v):
        _chk(self.L.fh_halo_update(self.h, v.h))

    def begin(self, v):
        _chk(self.L.fh_halo_begin(self.h, v.h))

    def end(self):
        _chk(self.L.fh_halo_end(self.h))

    def spmv(self, A, x, y, mode=0, b=None, dinv=None, omega=0.0):
        """y = op(A, x) with the ghosts of x refreshed through this plan, exchange overlapped with the interior rows"""
        _chk(self.L.fh_spmv_ghosted(A.h, self.h, x.h, y.h, int(mode), None if b is None else b.h, None if dinv is None else dinv.h, float(omega)))

    def stats(self, reset=False):
        """{'updates', 'bytes_sent', 'exchange_ms', 'exposed_ms'}; the two times need ctx.set_option('halo_profile', 1)"""
        n, b = ctypes.c_int64(), ctypes.c_int64()
        tx, te = ctypes.c_double(), ctypes.c_double()
        _chk(self.L.fh_halo_stats(self.h, int(bool(reset)), ctypes.byref(n), ctypes.byref(b), ctypes.byref(tx), ctypes.byref(te)))
        return {"updates": n.value, "bytes_sent": b.value, "exchange_ms": tx.value, "exposed_ms": te.value}

    def allreduce_count(self, reset=False):
        n = ctypes.c_int64()
        _chk(self.L.fh_halo_allreduce_count(self.h, int(bool(reset)), ctypes.byref(n)))
        return n.value

    def allreduce_ms(self, reset=False):
        ms = ctypes.c_double()
        _chk(self.L.fh_halo_allreduce_ms(self.h, int(bool(reset)), ctypes.byref(ms)))
        return ms.value

    def allreduce_mat(self, A):
        """in-place sum over the ranks of the values of a matrix that has the same pattern on every rank"""
        _chk(self.L.fh_halo_allreduce_mat(self.h, A.h))

    def allreduce_vec(self, v):
        """in-place sum over the ranks of the owned part of a device vector"""
        _chk(self.L.fh_halo_allreduce_vec(self.h, v.h))

    def allreduce_sum(self, vals):
        a = _f64(np.atleast_1d(vals)).copy()
        _chk(self.L.fh_halo_allreduce_sum(self.h, _p(a), a.size))
        return a

    def destroy(self):
        if self.h:
            self.L.fh_halo_destroy(self.h)
            self.h = None
