"""TEST INFRASTRUCTURE ONLY (oracle).  CPU restatement of the TETRAHEDRON path of applications/001_Poisson (its shipped input3D_Tet_first / _serendipity /
_second.json with input/cube_Tet.neu): Gambit reader for TET10, the face and centre nodes FEMuS adds (TET15), numbering, refinement, the Poisson callback with
P1 / P2 / P2 + bubbles, triangle-face integrals, solve -- numpy, loops as the reference writes them.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.

  read_gambit   GambitIO.cpp:101-330 (TET10: GambitToFemusVertexIndex[1] = {0, 4, 1, 6, 5, 2, 7, 8, 9, 3}, faces as numbered in the file, flag = -(set name) - 1);
                Mesh::AddBiquadraticNodesNotInMeshFile (Mesh.cpp:1207-1333): a node per face (shared by the two tetrahedra it separates: the first that holds it
                creates it), a centre per element, coordinates with the weights of Mesh.cpp:107-113 (faces -1/9, 4/9; centre -1/8, 1/4)
  refine        MeshRefinement::RefineMesh with tet_lag::fine2CoarseVertexMapping (Tetrahedron.cpp:81-90); coordinates by the TET15 element prolongator
  basis         3d/Tetrahedron.cpp (TetLinear, TetQuadratic bit for bit; TetBiquadratic written with barycentric products, 1e-14), checked against
                tests/golden/fe_tables.npz
"""
import os

import numpy as np

from . import femus_oracle_tri as ot

XC = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [0.5, 0, 0], [0.5, 0.5, 0], [0, 0.5, 0], [0, 0, 0.5], [0.5, 0, 0.5], [0, 0.5, 0.5],
               [1. / 3., 1. / 3., 0], [1. / 3., 0, 1. / 3.], [1. / 3., 1. / 3., 1. / 3.], [0, 1. / 3., 1. / 3.], [0.25, 0.25, 0.25]])
F2C = np.array([[0, 4, 6, 7], [4, 1, 5, 8], [6, 5, 2, 9], [7, 8, 9, 3], [5, 6, 4, 7], [8, 7, 5, 4], [7, 9, 8, 5], [9, 5, 7, 6]])
FACE = np.array([[0, 2, 1, 6, 5, 4, 10], [0, 1, 3, 4, 8, 7, 11], [1, 2, 3, 5, 9, 8, 12], [2, 0, 3, 6, 7, 9, 13]])
EDGE = ((0, 1), (1, 2), (2, 0), (0, 3), (1, 3), (2, 3))          # local nodes 4 .. 9 sit between these vertices
G2F = (0, 4, 1, 6, 5, 2, 7, 8, 9, 3)
NDOF = {"linear": 4, "serendipity": 10, "biquadratic": 15}
NFACE = {"linear": 3, "serendipity": 6, "biquadratic": 7}
# Mesh.cpp:107-113: weights of the ten file nodes in a face node (rows 0 .. 3) and in the centre (row 4)
WGT = np.array([[-1, -1, -1, 0, 4, 4, 4, 0, 0, 0], [-1, -1, 0, -1, 4, 0, 0, 4, 4, 0], [0, -1, -1, -1, 0, 4, 0, 0, 4, 4], [-1, 0, -1, -1, 0, 0, 4, 4, 0, 4]]) / 9.0
WGT = np.concatenate([WGT, np.array([[-1. / 8.] * 4 + [1. / 4.] * 6])])
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
    if fe == "biquadratic":
        # TetBiquadratic (Tetrahedron.cpp:325-600) = P2 corrected by the face bubbles m_f = L_a L_b L_c and the volume bubble q = L0 L1 L2 L3:
        # vertices + 3 sum m_f - 4 q, edges - 12 sum m_f + 32 q over the faces that hold them; faces 27 m_f - 108 q; centre 256 q
        def prod(idx):
            v = np.ones_like(x)
            for a in idx:
                v = v * L[a]
            d = np.zeros((x.size, 3))
            for a in idx:
                r = np.ones_like(x)
                for b in idx:
                    if b != a:
                        r = r * L[b]
                d += r[:, None] * dL[a]
            return v, d
        m = [prod(FACE[f][:3]) for f in range(4)]
        q = prod((0, 1, 2, 3))
        for a in range(4):
            fs = [f for f in range(4) if a in FACE[f][:3]]
            phi[a] = phi[a] + 3 * sum(m[f][0] for f in fs) - 4 * q[0]
            dphi[a] = dphi[a] + 3 * sum(m[f][1] for f in fs) - 4 * q[1]
        for k, (a, b) in enumerate(EDGE):
            fs = [f for f in range(4) if a in FACE[f][:3] and b in FACE[f][:3]]
            phi[4 + k] = phi[4 + k] - 12 * sum(m[f][0] for f in fs) + 32 * q[0]
            dphi[4 + k] = dphi[4 + k] - 12 * sum(m[f][1] for f in fs) + 32 * q[1]
        phi += [27 * m[f][0] - 108 * q[0] for f in range(4)] + [256 * q[0]]
        dphi += [27 * m[f][1] - 108 * q[1] for f in range(4)] + [256 * q[1]]
    return np.stack(phi, axis=1), np.stack(dphi, axis=1)


def _renumber(raw, nnode):
    new = np.full(nnode, -1, dtype=np.int64)
    k, own = 0, []
    for lo, hi in ((0, 4), (4, 10), (10, 15)):
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
    raw = np.full((nel, 15), -1, dtype=np.int64)
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
    # the nodes the file does not hold: one per face (shared by the two tetrahedra it separates), then one centre per element
    nn = nvt
    for e in range(nel):
        for f in range(4):
            if raw[e, 10 + f] < 0:
                raw[e, 10 + f] = nn
                mine = set(raw[e, FACE[f][:3]].tolist())
                done = False
                for e2 in range(e + 1, nel):
                    for f2 in range(4):
                        if raw[e2, 10 + f2] < 0 and set(raw[e2, FACE[f2][:3]].tolist()) == mine:
                            raw[e2, 10 + f2] = nn
                            done = True
                            break
                    if done:
                        break
                nn += 1
    for e in range(nel):
        raw[e, 14] = nn
        nn += 1
    coords = np.concatenate([xyz, np.zeros((nn - nvt, 3))])
    for e in range(nel):
        for j in range(10, 15):
            sacc = np.zeros(3)
            for i in range(10):
                sacc += coords[raw[e, i]] * WGT[j - 10][i]
            coords[raw[e, j]] = sacc
    new, own = _renumber(raw, nn)
    xs = np.empty_like(coords)
    xs[new] = coords
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
    EP = elem_prolongator("biquadratic")
    raw = np.full((8 * nel, 15), -1, dtype=np.int64)
    coords = list(xs)
    fff = np.full((8 * nel, 4), -1, dtype=np.int64)
    shared = {}

    def node(key, e, j, local):
        if key not in shared:
            shared[key] = len(coords)
            coords.append(sum(EP[j, local, m] * xs[ed[e, m]] for m in range(15)))
        return shared[key]

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
                raw[c, 4 + k] = node(tuple(sorted((raw[c, a], raw[c, b]))), e, j, 4 + k)
            for f in range(4):
                raw[c, 10 + f] = node(tuple(sorted(raw[c, FACE[f][:3]].tolist())), e, j, 10 + f)
            raw[c, 14] = len(coords)
            coords.append(sum(EP[j, 14, m] * xs[ed[e, m]] for m in range(15)))
    coords = np.array(coords)
    new, own = _renumber(raw, coords.shape[0])
    used = new >= 0
    xf = np.empty((own[2], 3))
    xf[new[used]] = coords[used]
    return new[raw], xf, fff, own


def n_dofs(own, fe):
    return own[{"linear": 0, "serendipity": 1, "biquadratic": 2}[fe]]


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
    nfn = NFACE[fe]
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
    nfn = NFACE[fe]
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
