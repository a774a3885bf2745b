"""TEST INFRASTRUCTURE ONLY (oracle).  CPU restatement of the TETRAHEDRON path of applications/001_Poisson (its shipped input3D_Tet_first / _serendipity.json with
input/cube_Tet.neu): Gambit reader for TET10, numbering, refinement, the Poisson callback with P1 / P2, triangle-face integrals, solve -- numpy, loops as the
reference writes them.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.

  read_gambit   GambitIO.cpp:101-330 (TET10: GambitToFemusVertexIndex[1] = {0, 4, 1, 6, 5, 2, 7, 8, 9, 3}, faces as numbered in the file, flag = -(set name) - 1)
  refine        MeshRefinement::RefineMesh with tet_lag::fine2CoarseVertexMapping (Tetrahedron.cpp:81-90); coordinates by the P2 element prolongator
  basis         3d/Tetrahedron.cpp (TetLinear, TetQuadratic), checked against tests/golden/fe_tables.npz
  (the face nodes and the centre FEMuS adds for its TET15 are not restated: the families served are P1 and P2)
"""
import os

import numpy as np

from . import femus_oracle_tri as ot

XC = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [0.5, 0, 0], [0.5, 0.5, 0], [0, 0.5, 0], [0, 0, 0.5], [0.5, 0, 0.5], [0, 0.5, 0.5]])
F2C = np.array([[0, 4, 6, 7], [4, 1, 5, 8], [6, 5, 2, 9], [7, 8, 9, 3], [5, 6, 4, 7], [8, 7, 5, 4], [7, 9, 8, 5], [9, 5, 7, 6]])
FACE = np.array([[0, 2, 1, 6, 5, 4], [0, 1, 3, 4, 8, 7], [1, 2, 3, 5, 9, 8], [2, 0, 3, 6, 7, 9]])
EDGE = ((0, 1), (1, 2), (2, 0), (0, 3), (1, 3), (2, 3))          # local nodes 4 .. 9 sit between these vertices
G2F = (0, 4, 1, 6, 5, 2, 7, 8, 9, 3)
NDOF = {"linear": 4, "serendipity": 10}
_G = None


def gauss(order="seventh"):
    global _G
    if _G is None:
        _G = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "fe_tables.npz"))
    return _G["gauss_w_tet_%s" % order], _G["gauss_x_tet_%s" % order]


def basis(fe, pts):
    pts = np.atleast_2d(pts)
    x, y, z = pts[:, 0], pts[:, 1], pts[:, 2]
    L = [1. - x - y - z, x, y, z]
    one, zero = np.ones_like(x), np.zeros_like(x)
    dL = [np.stack([-one, -one, -one], axis=1), np.stack([one, zero, zero], axis=1), np.stack([zero, one, zero], axis=1), np.stack([zero, zero, one], axis=1)]
    if fe == "linear":
        return np.stack(L, axis=1), np.stack(dL, axis=1)
    phi = [L[a] * (2 * L[a] - 1) for a in range(4)] + [4 * L[a] * L[b] for a, b in EDGE]
    dphi = [(4 * L[a] - 1)[:, None] * dL[a] for a in range(4)] + [4 * (L[a][:, None] * dL[b] + L[b][:, None] * dL[a]) for a, b in EDGE]
    return np.stack(phi, axis=1), np.stack(dphi, axis=1)


def _renumber(raw, nnode):
    new = np.full(nnode, -1, dtype=np.int64)
    k, own = 0, []
    for lo, hi in ((0, 4), (4, 10)):
        for e in range(raw.shape[0]):
            for l in range(lo, hi):
                if new[raw[e, l]] < 0:
                    new[raw[e, l]] = k
                    k += 1
        own.append(k)
    return new, own


def read_gambit(path):
    tok = open(path).read().split()
    p = tok.index("NDFVL") + 1
    nvt, nel, ngroup, nbcd, dim, _ = (int(t) for t in tok[p:p + 6])
    assert dim == 3
    p = tok.index("COORDINATES") + 2
    xyz = np.zeros((nvt, 3))
    for n in range(nvt):
        xyz[n] = [float(t) for t in tok[p + 1:p + 4]]
        p += 4
    p = tok.index("ELEMENTS/CELLS") + 2
    raw = np.zeros((nel, 10), dtype=np.int64)
    for e in range(nel):
        assert int(tok[p + 1]) == 6 and int(tok[p + 2]) == 10, "TET10 elements only"
        for i in range(10):
            raw[e, G2F[i]] = int(tok[p + 3 + i]) - 1
        p += 13
    ff = np.full((nel, 4), -1, dtype=np.int64)
    q = 0
    for _ in range(nbcd):
        q = tok.index("CONDITIONS", q) + 2
        name, nface = int(tok[q]), int(tok[q + 2])
        q += 5
        for _ in range(nface):
            ff[int(tok[q]) - 1, int(tok[q + 2]) - 1] = -name - 1
            q += 3
    new, own = _renumber(raw, nvt)
    xs = np.empty_like(xyz)
    xs[new] = xyz
    return new[raw], xs, ff, own


def elem_prolongator(fe):
    nc = NDOF[fe]
    P = np.zeros((8, nc, nc))
    for j in range(8):
        v = XC[F2C[j]]
        for i in range(nc):
            pt = v[0] + (v[1] - v[0]) * XC[i, 0] + (v[2] - v[0]) * XC[i, 1] + (v[3] - v[0]) * XC[i, 2]
            ph = basis(fe, pt)[0][0]
            P[j, i] = np.where(np.abs(ph) >= 1e-14, ph, 0.0)
    return P


def refine(ed, xs, ff):
    nel = ed.shape[0]
    EP = elem_prolongator("serendipity")
    raw = np.full((8 * nel, 10), -1, dtype=np.int64)
    coords = list(xs)
    fff = np.full((8 * nel, 4), -1, dtype=np.int64)
    edges = {}
    for e in range(nel):
        for j in range(8):
            c = 8 * e + j
            cn = F2C[j]                                        # the child's vertices as local nodes of the father
            raw[c, :4] = ed[e, cn]
            for lf in range(4):                                # a child face all of whose vertices lie on a face of the father carries that face's flag
                for f in range(4):
                    if all(int(cn[v]) in FACE[f] for v in FACE[lf][:3]):
                        fff[c, lf] = ff[e, f]
            for k, (a, b) in enumerate(EDGE):
                key = (min(raw[c, a], raw[c, b]), max(raw[c, a], raw[c, b]))
                if key not in edges:
                    edges[key] = len(coords)
                    coords.append(sum(EP[j, 4 + k, m] * xs[ed[e, m]] for m in range(10)))
                raw[c, 4 + k] = edges[key]
    coords = np.array(coords)
    new, own = _renumber(raw, coords.shape[0])
    used = new >= 0
    xf = np.empty((own[1], 3))
    xf[new[used]] = coords[used]
    return new[raw], xf, fff, own


def n_dofs(own, fe):
    return own[0] if fe == "linear" else own[1]


def assemble(ed, xs, fe, source, sol=None, order="seventh"):
    nc = NDOF[fe]
    ndof = int(ed[:, :nc].max()) + 1
    w, xg = gauss(order)
    PHI, DPHI = basis(fe, xg)
    import scipy.sparse as sp
    rows, cols, vals = [], [], []
    F = np.zeros(ndof)
    u = np.zeros(ndof) if sol is None else sol
    for e in range(ed.shape[0]):
        dof = ed[e, :nc]
        x = xs[dof]
        Ke = np.zeros((nc, nc))
        Fe = np.zeros(nc)
        for g in range(w.size):
            J = DPHI[g].T @ x                                  # J[a][b] = sum_n dphi_n/dxi_a x_n[b]
            det = np.linalg.det(J)
            grad = DPHI[g] @ np.linalg.inv(J).T                # grad phi_n [b] = sum_a Jinv[b][a] dphi_n/dxi_a  (Jinv = inverse of J^T ... see elem_type_3D::Jacobian)
            weight = det * w[g]
            gu = grad.T @ u[dof]
            f = source(PHI[g] @ x)
            Ke += (grad @ grad.T) * weight
            Fe += (f * PHI[g] - grad @ gu) * weight
        rows.append(np.repeat(dof, nc))
        cols.append(np.tile(dof, nc))
        vals.append(Ke.ravel())
        F[dof] += Fe
    K = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(ndof, ndof))
    return K, F


def neumann(ed, xs, ff, fe, flux_by_flag, order="seventh"):
    """triangle-face integrals of the flux (JacobianSur of the TRI3 / TRI6 element on the face's nodes: |t_xi x t_eta| w)"""
    nfn = 3 if fe == "linear" else 6
    w, xg = ot.gauss(order)
    PH, DP = ot.basis(fe, xg)
    nc = NDOF[fe]
    F = np.zeros(int(ed[:, :nc].max()) + 1)
    for e, f in zip(*np.nonzero(ff < -1)):
        if ff[e, f] in flux_by_flag:
            fn = ed[e, FACE[f][:nfn]]
            x = xs[fn]
            for g in range(w.size):
                t = DP[g].T @ x                                # rows: d x / d xi, d x / d eta
                area = np.linalg.norm(np.cross(t[0], t[1]))
                tau = flux_by_flag[ff[e, f]]
                tv = tau(PH[g] @ x) if callable(tau) else tau
                F[fn] += PH[g] * tv * area * w[g]
    return F


def dirichlet(ed, ff, fe, flags):
    nfn = 3 if fe == "linear" else 6
    out = set()
    for e, f in zip(*np.nonzero(ff < -1)):
        if ff[e, f] in flags:
            out.update(int(n) for n in ed[e, FACE[f][:nfn]])
    return np.array(sorted(out), dtype=np.int64)


def solve(mesh0, nlevels, fe, source, dirichlet_flags, flux_by_flag=None):
    """the discrete problem of the finest of nlevels levels, solved directly"""
    import scipy.sparse.linalg as spla
    meshes = [mesh0]
    for _ in range(1, nlevels):
        meshes.append(refine(*meshes[-1][:3]))
    ed, xs, ff, own = meshes[-1]
    ndof = n_dofs(own, fe)
    bdc = dirichlet(ed, ff, fe, set(dirichlet_flags))
    K, F = assemble(ed, xs, fe, source)
    if flux_by_flag:
        F = F + neumann(ed, xs, ff, fe, flux_by_flag)
    K = K.tolil()
    K[bdc, :] = 0.0
    K[bdc, bdc] = 1.0
    F[bdc] = 0.0
    return spla.spsolve(K.tocsc(), F), meshes
