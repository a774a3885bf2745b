"""TEST INFRASTRUCTURE ONLY (oracle).  CPU restatement of the PRISM path of applications/001_Poisson (its shipped input3D_Wedge_first / _second /
_serendipity.json with input/cube_Wedge.neu): Gambit reader for WEDGE18, the triangle-face and centre nodes FEMuS adds (WEDGE21), numbering, refinement, the
Poisson callback with the three Lagrange families, quadrilateral- and triangle-face integrals, solve -- numpy, loops as the reference writes them.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.

  read_gambit   GambitIO.cpp:101-330 (WEDGE18: GambitToFemusVertexIndex[2] :70-74, GambitToFemusFaceIndex[2] = {2, 1, 0, 4, 3} :86, flag = -(set name) - 1),
                Mesh::AddBiquadraticNodesNotInMeshFile (Mesh.cpp:1207-1333; weights Mesh.cpp:115-122)
  refine        MeshRefinement::RefineMesh with wedge_lag::fine2CoarseVertexMapping (Wedge.cpp:118-127); coordinates by the biquadratic element prolongator
  basis         3d/Wedge.cpp (triangle x line for the linear and biquadratic families, the 15-node family in barycentric form), checked against the fixture
"""
import os

import numpy as np

from . import femus_oracle as fo
from . import femus_oracle_tri as ot

XC = np.array([[0, 0, -1], [1, 0, -1], [0, 1, -1], [0, 0, 1], [1, 0, 1], [0, 1, 1], [0.5, 0, -1], [0.5, 0.5, -1], [0, 0.5, -1], [0.5, 0, 1], [0.5, 0.5, 1], [0, 0.5, 1],
               [0, 0, 0], [1, 0, 0], [0, 1, 0], [0.5, 0, 0], [0.5, 0.5, 0], [0, 0.5, 0], [1. / 3., 1. / 3., -1], [1. / 3., 1. / 3., 1], [1. / 3., 1. / 3., 0]])
F2C = np.array([[0, 6, 8, 12, 15, 17], [6, 1, 7, 15, 13, 16], [8, 7, 2, 17, 16, 14], [7, 8, 6, 16, 17, 15], [12, 15, 17, 3, 9, 11], [15, 13, 16, 9, 4, 10],
                [17, 16, 14, 11, 10, 5], [16, 17, 15, 10, 11, 9]])
FACE = [[0, 1, 4, 3, 6, 13, 9, 12, 15], [1, 2, 5, 4, 7, 14, 10, 13, 16], [2, 0, 3, 5, 8, 12, 11, 14, 17], [0, 2, 1, 8, 7, 6, 18], [3, 4, 5, 9, 10, 11, 19]]
EDGE = ((0, 1), (1, 2), (2, 0), (3, 4), (4, 5), (5, 3), (0, 3), (1, 4), (2, 5))      # local nodes 6 .. 14
G2F = (3, 11, 5, 9, 10, 4, 12, 17, 14, 15, 16, 13, 0, 8, 2, 6, 7, 1)
GFACE = (2, 1, 0, 4, 3)
NDOF = {"linear": 6, "serendipity": 15, "biquadratic": 21}
NFN = {"linear": (4, 3), "serendipity": (8, 6), "biquadratic": (9, 7)}
_G = None


def gauss(order="seventh"):
    global _G
    if _G is None:
        _G = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "fe_tables.npz"))
    return _G["gauss_w_wedge_%s" % order], _G["gauss_x_wedge_%s" % order]


def _line(fe, z, k):
    if fe == "linear":
        return fo.lag_linear(z, k), fo.dlag_linear(z, k)
    return fo.lag_biquadratic(z, k), fo.dlag_biquadratic(z, k)


def basis(fe, pts):
    pts = np.atleast_2d(pts)
    npt = pts.shape[0]
    x, y, z = pts[:, 0], pts[:, 1], pts[:, 2]
    if fe != "serendipity":
        tp, tdp = ot.basis(fe, pts[:, :2])                       # triangle part: nodes 0..2 (, 3..5, 6) in the triangle's order
        layers = (0, 2) if fe == "linear" else (0, 2, 1)         # line index of z = -1, +1 (, 0)
        if fe == "linear":
            order = [(t, k) for k in layers for t in range(3)]
        else:                                                    # 0-5 vertices, 6-11 triangle middles at z = -1 / +1, 12-14 vertical middles, 15-17 quadrilateral centres, 18-20 the three centres
            order = [(t, 0) for t in range(3)] + [(t, 2) for t in range(3)] + [(t, 0) for t in range(3, 6)] + [(t, 2) for t in range(3, 6)] + \
                    [(t, 1) for t in range(3)] + [(t, 1) for t in range(3, 6)] + [(6, 0), (6, 2), (6, 1)]
        phi = np.zeros((npt, len(order)))
        dphi = np.zeros((npt, len(order), 3))
        for n, (t, k) in enumerate(order):
            l = np.array([_line(fe, zz, k)[0] for zz in z])
            dl = np.array([_line(fe, zz, k)[1] for zz in z])
            phi[:, n] = tp[:, t] * l
            dphi[:, n, 0] = tdp[:, t, 0] * l
            dphi[:, n, 1] = tdp[:, t, 1] * l
            dphi[:, n, 2] = tp[:, t] * dl
        return phi, dphi
    L = [1. - x - y, x, y]
    phi, dL, dz = [], [], []                                     # per node: value, (d/dL0, d/dL1, d/dL2), d/dz
    zero = np.zeros_like(x)
    for s in (-1.0, 1.0):
        for a in range(3):
            phi.append(L[a] * (2 * L[a] - 2 + s * z) * (1 + s * z) / 2)
            g = [zero, zero, zero]
            g[a] = (4 * L[a] - 2 + s * z) * (1 + s * z) / 2
            dL.append(g)
            dz.append(L[a] * s * (2 * L[a] - 1 + 2 * s * z) / 2)
    for s in (-1.0, 1.0):
        for a, b in ((0, 1), (1, 2), (2, 0)):
            phi.append(2 * L[a] * L[b] * (1 + s * z))
            g = [zero, zero, zero]
            g[a] = 2 * L[b] * (1 + s * z)
            g[b] = 2 * L[a] * (1 + s * z)
            dL.append(g)
            dz.append(2 * s * L[a] * L[b])
    for a in range(3):
        phi.append(L[a] * (1 - z * z))
        g = [zero, zero, zero]
        g[a] = 1 - z * z
        dL.append(g)
        dz.append(-2 * z * L[a])
    P = np.stack(phi, axis=1)
    D = np.stack([np.stack([g[1] - g[0] for g in dL], axis=1), np.stack([g[2] - g[0] for g in dL], axis=1), np.stack(dz, axis=1)], axis=2)
    return P, D


def _renumber(raw, nnode):
    new = np.full(nnode, -1, dtype=np.int64)
    k, own = 0, []
    for lo, hi in ((0, 6), (6, 15), (15, 21)):
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
    p = tok.index("COORDINATES") + 2
    xyz = np.zeros((nvt, 3))
    for n in range(nvt):
        xyz[n] = [float(t) for t in tok[p + 1:p + 4]]
        p += 4
    p = tok.index("ELEMENTS/CELLS") + 2
    raw = np.full((nel, 21), -1, dtype=np.int64)
    for e in range(nel):
        assert int(tok[p + 1]) == 5 and int(tok[p + 2]) == 18, "WEDGE18 elements only"
        for i in range(18):
            raw[e, G2F[i]] = int(tok[p + 3 + i]) - 1
        p += 21
    ff = np.full((nel, 5), -1, dtype=np.int64)
    q = 0
    for _ in range(nbcd):
        q = tok.index("CONDITIONS", q) + 2
        name, nface = int(tok[q]), int(tok[q + 2])
        q += 5
        for _ in range(nface):
            ff[int(tok[q]) - 1, GFACE[int(tok[q + 2]) - 1]] = -name - 1
            q += 3
    # the nodes the file does not hold: one per triangle face (shared by the two prisms it separates), then one centre per element
    coords = list(xyz)
    for e in range(nel):
        for f in (3, 4):
            if raw[e, 15 + f] < 0:
                raw[e, 15 + f] = len(coords)
                mine = set(raw[e, FACE[f][:3]].tolist())
                done = False
                for e2 in range(e + 1, nel):
                    for f2 in (3, 4):
                        if raw[e2, 15 + f2] < 0 and set(raw[e2, FACE[f2][:3]].tolist()) == mine:
                            raw[e2, 15 + f2] = len(coords)
                            done = True
                            break
                    if done:
                        break
                coords.append(None)
    for e in range(nel):
        raw[e, 20] = len(coords)
        coords.append(None)
    coords = np.array([c if c is not None else np.zeros(3) for c in coords])
    W = {18: ([0, 1, 2], [6, 7, 8]), 19: ([3, 4, 5], [9, 10, 11]), 20: ([12, 13, 14], [15, 16, 17])}
    for e in range(nel):
        for j in (18, 19, 20):
            s = np.zeros(3)
            for i in range(18):
                wgt = -1. / 9. if i in W[j][0] else 4. / 9. if i in W[j][1] else 0.0
                s += coords[raw[e, i]] * wgt
            coords[raw[e, j]] = s
    new, own = _renumber(raw, coords.shape[0])
    xs = np.empty_like(coords)
    xs[new] = coords
    return new[raw], xs, ff, own


def elem_prolongator(fe):
    nc = NDOF[fe]
    P = np.zeros((8, nc, nc))
    for j in range(8):
        v = XC[F2C[j]]
        z0, z1 = v[0, 2], v[3, 2]
        for i in range(nc):
            pt = np.array([v[0, 0] + (v[1, 0] - v[0, 0]) * XC[i, 0] + (v[2, 0] - v[0, 0]) * XC[i, 1],
                           v[0, 1] + (v[1, 1] - v[0, 1]) * XC[i, 0] + (v[2, 1] - v[0, 1]) * XC[i, 1], z0 + (z1 - z0) * 0.5 * (XC[i, 2] + 1.0)])
            ph = basis(fe, pt)[0][0]
            P[j, i] = np.where(np.abs(ph) >= 1e-14, ph, 0.0)
    return P


def refine(ed, xs, ff):
    nel = ed.shape[0]
    EP = elem_prolongator("biquadratic")
    raw = np.full((8 * nel, 21), -1, dtype=np.int64)
    coords = list(xs)
    fff = np.full((8 * nel, 5), -1, dtype=np.int64)
    shared = {}

    def node(key, e, j, local):
        if key not in shared:
            shared[key] = len(coords)
            coords.append(sum(EP[j, local, m] * xs[ed[e, m]] for m in range(21)))
        return shared[key]

    for e in range(nel):
        for j in range(8):
            c = 8 * e + j
            cn = F2C[j]
            raw[c, :6] = ed[e, cn]
            for lf in range(5):                                # a child face all of whose vertices lie on a face of the father carries that face's flag
                nv = 4 if lf < 3 else 3
                for f in range(5):
                    if (lf < 3) == (f < 3) and all(int(cn[v]) in FACE[f] for v in FACE[lf][:nv]):
                        fff[c, lf] = ff[e, f]
            for k, (a, b) in enumerate(EDGE):
                raw[c, 6 + k] = node(tuple(sorted((raw[c, a], raw[c, b]))), e, j, 6 + k)
            for f in range(5):
                nv = 4 if f < 3 else 3
                raw[c, 15 + f] = node(tuple(sorted(raw[c, FACE[f][:nv]].tolist())), e, j, 15 + f)
            raw[c, 20] = len(coords)
            coords.append(sum(EP[j, 20, m] * xs[ed[e, m]] for m in range(21)))
    coords = np.array(coords)
    new, own = _renumber(raw, coords.shape[0])
    used = new >= 0
    xf = np.empty((own[2], 3))
    xf[new[used]] = coords[used]
    return new[raw], xf, fff, own


def n_dofs(own, fe):
    return own[{"linear": 0, "serendipity": 1, "biquadratic": 2}[fe]]


def assemble(ed, xs, fe, source, sol=None, order="seventh"):
    import scipy.sparse as sp
    nc = NDOF[fe]
    ndof = int(ed[:, :nc].max()) + 1
    w, xg = gauss(order)
    PHI, DPHI = basis(fe, xg)
    rows, cols, vals = [], [], []
    F = np.zeros(ndof)
    u = np.zeros(ndof) if sol is None else sol
    for e in range(ed.shape[0]):
        dof = ed[e, :nc]
        x = xs[dof]
        Ke = np.zeros((nc, nc))
        Fe = np.zeros(nc)
        for g in range(w.size):
            J = DPHI[g].T @ x
            det = np.linalg.det(J)
            grad = DPHI[g] @ np.linalg.inv(J).T
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
    """face integrals of the flux: quadrilateral faces (QUAD4 / QUAD8 / QUAD9 on the face's nodes) and triangle faces (TRI3 / TRI6 / TRI7): |t_xi x t_eta| w"""
    nq, nt = NFN[fe]
    wq, xq = fo.gauss_table("quad", order)
    xq = np.asarray(xq)
    if xq.shape[0] == 2 and xq.shape[1] != 2:
        xq = xq.T
    out = fo.eval_basis("quad", fe, xq)
    PQ, DQ = out[0], out[1]
    wt, xt = ot.gauss(order)
    PT, DT = ot.basis(fe, xt)
    nc = NDOF[fe]
    F = np.zeros(int(ed[:, :nc].max()) + 1)
    for e, f in zip(*np.nonzero(ff < -1)):
        if ff[e, f] not in flux_by_flag:
            continue
        quad = f < 3
        fn = ed[e, FACE[f][:(nq if quad else nt)]]
        x = xs[fn]
        w, PH, DP = (wq, PQ, DQ) if quad else (wt, PT, DT)
        for g in range(len(w)):
            t = DP[g].T @ x
            area = np.linalg.norm(np.cross(t[0], t[1]))
            tau = flux_by_flag[ff[e, f]]
            tv = tau(PH[g] @ x) if callable(tau) else tau
            F[fn] += PH[g] * tv * area * w[g]
    return F


def dirichlet(ed, ff, fe, flags):
    nq, nt = NFN[fe]
    out = set()
    for e, f in zip(*np.nonzero(ff < -1)):
        if ff[e, f] in flags:
            out.update(int(n) for n in ed[e, FACE[f][:(nq if f < 3 else nt)]])
    return np.array(sorted(out), dtype=np.int64)


def solve(mesh0, nlevels, fe, source, dirichlet_flags, flux_by_flag=None):
    import scipy.sparse.linalg as spla
    meshes = [mesh0]
    for _ in range(1, nlevels):
        meshes.append(refine(*meshes[-1][:3]))
    ed, xs, ff, own = meshes[-1]
    bdc = dirichlet(ed, ff, fe, set(dirichlet_flags))
    K, F = assemble(ed, xs, fe, source)
    if flux_by_flag:
        F = F + neumann(ed, xs, ff, fe, flux_by_flag)
    K = K.tolil()
    K[bdc, :] = 0.0
    K[bdc, bdc] = 1.0
    F[bdc] = 0.0
    return spla.spsolve(K.tocsc(), F), meshes
