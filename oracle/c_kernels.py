"""ORACLE -- TEST INFRASTRUCTURE ONLY.  ctypes access to oracle/liboracle_c.so (oracle_kernels.c): the C restatement
of the timed loops, used as a second checker and as bench.py's `cpu_baseline` ("port")."""
import ctypes
import os

import numpy as np

from . import femus_oracle as fo

_HERE = os.path.dirname(os.path.abspath(__file__))
_L = None


def lib():
    global _L
    if _L is None:
        _L = ctypes.CDLL(os.path.join(_HERE, "liboracle_c.so"))
        _L.oc_num_threads.restype = ctypes.c_int
    return _L


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def num_threads():
    return int(lib().oc_num_threads())


def set_threads(n):
    lib().oc_set_threads(int(n))


def assemble_poisson(mesh_elem_dof, coords, fe, geom, e0, e1, sol=None, source_kind=0, p0=1.0, p1=0.0, csr=None, order="seventh"):
    """element range [e0,e1).  csr=(rowptr, col, val, res) -> scatter (sequential, element order); else returns K, F."""
    et = fo.ElemType(geom, fe, order)
    ed = np.ascontiguousarray(mesh_elem_dof, dtype=np.int32)
    xy = np.ascontiguousarray(coords, dtype=np.float64)
    w = np.ascontiguousarray(et.w)
    phi = np.ascontiguousarray(et.phi)
    dphi = np.ascontiguousarray(et.dphi)
    s = None if sol is None else np.ascontiguousarray(sol, dtype=np.float64)
    L = lib()
    c_d = ctypes.c_double
    if csr is None:
        K = np.empty((e1 - e0, et.nc, et.nc))
        F = np.empty((e1 - e0, et.nc))
        rc = L.oc_assemble_poisson(et.dim, et.nc, et.ng, _p(w), _p(phi), _p(dphi), int(e0), int(e1), ed.shape[1], _p(ed), _p(xy), _p(s),
                                   int(source_kind), c_d(p0), c_d(p1), None, None, None, None, _p(K), _p(F))
        assert rc == 0
        return K, F
    rowptr, col, val, res = csr
    rc = L.oc_assemble_poisson(et.dim, et.nc, et.ng, _p(w), _p(phi), _p(dphi), int(e0), int(e1), ed.shape[1], _p(ed), _p(xy), _p(s),
                               int(source_kind), c_d(p0), c_d(p1), _p(rowptr), _p(col), _p(val), _p(res), None, None)
    assert rc == 0


def assemble_poisson_all_cores(mesh_elem_dof, coords, fe, geom, csr, sol=None, source_kind=0, p0=1.0, p1=0.0, order="seventh"):
    """the whole element loop with OpenMP over the usable cores (timed CPU baseline: owner-computes element ranges, atomic adds)"""
    et = fo.ElemType(geom, fe, order)
    ed = np.ascontiguousarray(mesh_elem_dof, dtype=np.int32)
    xy = np.ascontiguousarray(coords, dtype=np.float64)
    w, phi, dphi = np.ascontiguousarray(et.w), np.ascontiguousarray(et.phi), np.ascontiguousarray(et.dphi)
    s = None if sol is None else np.ascontiguousarray(sol, dtype=np.float64)
    rowptr, col, val, res = csr
    c_d = ctypes.c_double
    rc = lib().oc_assemble_poisson_omp(et.dim, et.nc, et.ng, _p(w), _p(phi), _p(dphi), ed.shape[0], ed.shape[1], _p(ed), _p(xy), _p(s),
                                       int(source_kind), c_d(p0), c_d(p1), _p(rowptr), _p(col), _p(val), _p(res))
    assert rc == 0


def spmv(A, x, y, mode=0, b=None, dinv=None, omega=0.0):
    """A: scipy csr with int32 indices"""
    lib().oc_spmv(A.shape[0], _p(A.indptr), _p(A.indices), _p(A.data), _p(x), _p(y), int(mode), _p(b), _p(dinv), ctypes.c_double(omega))


class CVcycle:
    """the multiplicative V(npre,npost) Richardson/Jacobi cycle of femus_oracle.vcycle with the C SpMV (timed CPU baseline)."""

    def __init__(self, A, P, omega=2. / 3., npre=2, npost=2, coarse_solve=None):
        import scipy.sparse.linalg as spla
        self.A = [self._csr(a) for a in A]
        self.P = [None] + [self._csr(p) for p in P[1:]]
        self.R = [None] + [self._csr(p.T) for p in P[1:]]
        self.dinv = [np.ascontiguousarray(fo.jacobi_dinv(a)) for a in self.A]
        self.omega, self.npre, self.npost = omega, npre, npost
        self.lu = spla.splu(self.A[0].tocsc()) if coarse_solve is None else None
        self.coarse_solve = coarse_solve
        n = [a.shape[0] for a in self.A]
        self.x = [np.zeros(k) for k in n]
        self.x2 = [np.zeros(k) for k in n]
        self.b = [np.zeros(k) for k in n]
        self.r = [np.zeros(k) for k in n]

    @staticmethod
    def _csr(a):
        a = a.tocsr()
        a.sort_indices()
        a.indptr = a.indptr.astype(np.int32)
        a.indices = a.indices.astype(np.int32)
        a.data = np.ascontiguousarray(a.data, dtype=np.float64)
        return a

    def apply(self, rhs):
        L = lib()
        top = len(self.A) - 1
        self.b[top][:] = rhs
        for l in range(top, 0, -1):
            if self.npre == 0:
                self.x[l][:] = 0.0
            else:
                L.oc_first_sweep(self.A[l].shape[0], _p(self.x[l]), _p(self.b[l]), _p(self.dinv[l]), ctypes.c_double(self.omega))
                for _ in range(1, self.npre):
                    spmv(self.A[l], self.x[l], self.x2[l], 3, self.b[l], self.dinv[l], self.omega)
                    self.x[l], self.x2[l] = self.x2[l], self.x[l]
            spmv(self.A[l], self.x[l], self.r[l], 2, self.b[l])
            spmv(self.R[l], self.r[l], self.b[l - 1], 0)
        self.x[0][:] = self.lu.solve(self.b[0]) if self.lu is not None else self.coarse_solve(self.b[0])
        for l in range(1, top + 1):
            spmv(self.P[l], self.x[l - 1], self.x[l], 1)
            for _ in range(self.npost):
                spmv(self.A[l], self.x[l], self.x2[l], 3, self.b[l], self.dinv[l], self.omega)
                self.x[l], self.x2[l] = self.x2[l], self.x[l]
        return self.x[top].copy()
