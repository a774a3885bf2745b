"""TEST INFRASTRUCTURE ONLY (oracle).  CPU restatement of the TRIANGLE path of applications/001_Poisson (a 2-D box with "elem_type": "Tri6"): mesh, numbering,
refinement, the Poisson callback, edge integrals, prolongators, solve -- numpy, loops as the reference writes them.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this package.

  box_mesh      MeshGeneration.cpp:283-650 (case 2, TRI6: lattice i + j (2 nx + 1), two triangles per cell, flags bottom -2 / right -3 / top -4 / left -5),
                Mesh::AddBiquadraticNodesNotInMeshFile (Mesh.cpp:1207-1333: the seventh node, weights -1/9, 4/9 of Mesh.cpp:124), numbering (vertices, middles,
                centres; first touch)
  basis         2d/Triangle.hpp:69-181 (P1, P2, P2 + bubble), checked against tests/golden/fe_tables.npz (the reference's compiled classes)
  refine        MeshRefinement::RefineMesh with tri_lag::fine2CoarseVertexMapping (Triangle.cpp:48-53); coordinates by the biquadratic element prolongator
  assemble      main.cpp:355-480 with dim == 2 (V = 0: the Laplace form), elem_type_2D::Jacobian
"""
import os

import numpy as np

from . import femus_oracle as fo

XC = np.array([[0, 0], [1, 0], [0, 1], [0.5, 0], [0.5, 0.5], [0, 0.5], [1. / 3., 1. / 3.]])
F2C = np.array([[0, 3, 5], [3, 1, 4], [5, 4, 2], [4, 5, 3]])
FACE = np.array([[0, 1, 3], [1, 2, 4], [2, 0, 5]])
NDOF = {"linear": 3, "serendipity": 6, "biquadratic": 7}
_G = None


def gauss(order="seventh"):
    """the reference's triangle rules are data (quadrature_Triangle.cpp): read from the fixture its compiled tables were dumped into"""
    global _G
    if _G is None:
        _G = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "fe_tables.npz"))
    return _G["gauss_w_tri_%s" % order], _G["gauss_x_tri_%s" % order]


def basis(fe, pts):
    """phi[np, nc], dphi[np, nc, 2] in the barycentric form of the families"""
    pts = np.atleast_2d(pts)
    x, y = pts[:, 0], pts[:, 1]
    l0, l1, l2 = 1. - x - y, x, y
    one, zero = np.ones_like(x), np.zeros_like(x)
    if fe == "linear":
        phi = [l0, l1, l2]
        dx = [-one, one, zero]
        dy = [-one, zero, one]
    else:
        phi = [l0 * (2 * l0 - 1), l1 * (2 * l1 - 1), l2 * (2 * l2 - 1), 4 * l0 * l1, 4 * l1 * l2, 4 * l2 * l0]
        dx = [-(4 * l0 - 1), 4 * l1 - 1, zero, 4 * (l0 - l1), 4 * l2, -4 * l2]
        dy = [-(4 * l0 - 1), zero, 4 * l2 - 1, -4 * l1, 4 * l1, 4 * (l0 - l2)]
        if fe == "biquadratic":
            b, bx, by = l0 * l1 * l2, l2 * (l0 - l1), l1 * (l0 - l2)
            cv, ce = 3.0, -12.0                      # vertices + 3 b, middles - 12 b, centre 27 b
            phi = [p + cv * b for p in phi[:3]] + [p + ce * b for p in phi[3:]] + [27 * b]
            dx = [p + cv * bx for p in dx[:3]] + [p + ce * bx for p in dx[3:]] + [27 * bx]
            dy = [p + cv * by for p in dy[:3]] + [p + ce * by for p in dy[3:]] + [27 * by]
    return np.stack(phi, axis=1), np.stack([np.stack(dx, axis=1), np.stack(dy, axis=1)], axis=2)


def _renumber(raw, nnode):
    new = np.full(nnode, -1, dtype=np.int64)
    k = 0
    own = []
    for a, b in ((0, 3), (3, 6), (6, 7)):
        for e in range(raw.shape[0]):
            for l in range(a, b):
                if new[raw[e, l]] < 0:
                    new[raw[e, l]] = k
                    k += 1
        own.append(k)
    return new, own          # nodes no element holds keep -1 (a father's centre is no node of its children)


def box_mesh(nx, ny, lo=(0., 0.), hi=(1., 1.)):
    px = 2 * nx + 1
    idx = lambda i, j: i + j * px
    xy = np.zeros((px * (2 * ny + 1), 2))
    for j in range(2 * ny + 1):
        for i in range(px):
            xy[idx(i, j)] = ((i / (2. * nx)) * (hi[0] - lo[0]) + lo[0], (j / (2. * ny)) * (hi[1] - lo[1]) + lo[1])
    ed, ff = [], []
    for j in range(0, 2 * ny, 2):
        for i in range(0, 2 * nx, 2):
            ed.append([idx(i, j), idx(i + 2, j), idx(i + 2, j + 2), idx(i + 1, j), idx(i + 2, j + 1), idx(i + 1, j + 1)])
            ff.append([-2 if j == 0 else -1, -3 if i == 2 * (nx - 1) else -1, -1])
            ed.append([idx(i, j), idx(i + 2, j + 2), idx(i, j + 2), idx(i + 1, j + 1), idx(i + 1, j + 2), idx(i, j + 1)])
            ff.append([-1, -4 if j == 2 * (ny - 1) else -1, -5 if i == 0 else -1])
    ed = np.array(ed)
    nel, n6 = ed.shape[0], xy.shape[0]
    ed7 = np.concatenate([ed, (n6 + np.arange(nel))[:, None]], axis=1)               # the seventh node, element by element
    wts = np.array([-1. / 9., -1. / 9., -1. / 9., 4. / 9., 4. / 9., 4. / 9.])
    centres = np.zeros((nel, 2))
    for e in range(nel):
        for i in range(6):
            centres[e] += xy[ed[e, i]] * wts[i]
    xy7 = np.concatenate([xy, centres])
    new, own = _renumber(ed7, xy7.shape[0])
    xs = np.empty_like(xy7)
    xs[new] = xy7
    return new[ed7], xs, np.array(ff), own


def elem_prolongator(fe):
    """P[child][local node][coarse function]: the coarse functions at the child's nodes, the child's reference triangle mapped affinely onto its vertices in the father"""
    nc = NDOF[fe]
    P = np.zeros((4, nc, nc))
    for j in range(4):
        v = XC[F2C[j]]
        for i in range(nc):
            pt = v[0] + (v[1] - v[0]) * XC[i, 0] + (v[2] - v[0]) * XC[i, 1]
            ph = basis(fe, pt)[0][0]
            P[j, i] = np.where(np.abs(ph) >= 1e-14, ph, 0.0)
    return P


def refine(ed, xs, ff):
    nel, nn = ed.shape[0], xs.shape[0]
    EP = elem_prolongator("biquadratic")
    raw = np.full((4 * nel, 7), -1, dtype=np.int64)
    coords = list(xs)
    fff = np.full((4 * nel, 3), -1)
    edges = {}
    pairs = ((0, 1), (1, 2), (2, 0))
    for e in range(nel):
        for j in range(4):
            c = 4 * e + j
            raw[c, :3] = ed[e, F2C[j]]
            if j < 3:
                for f in range(3):
                    if j in (pairs[f][0], pairs[f][1]):               # vertex j lies on face f: the child carries the flag on the same local face
                        fff[c, f] = ff[e, f]
            for k, (a, b) in enumerate(pairs):
                key = (min(raw[c, a], raw[c, b]), max(raw[c, a], raw[c, b]))
                if key not in edges:
                    edges[key] = len(coords)
                    coords.append(sum(EP[j, 3 + k, m] * xs[ed[e, m]] for m in range(7)))
                raw[c, 3 + k] = edges[key]
            raw[c, 6] = len(coords)
            coords.append(sum(EP[j, 6, m] * xs[ed[e, m]] for m in range(7)))
    coords = np.array(coords)
    new, own = _renumber(raw, coords.shape[0])
    used = new >= 0
    xf = np.empty((own[2], 2))
    xf[new[used]] = coords[used]
    return new[raw], xf, fff, own


def n_dofs(own, fe):
    return own[{"linear": 0, "serendipity": 1, "biquadratic": 2}[fe]]


def assemble(ed, xs, fe, source, sol=None, order="seventh"):
    """dense K and F = (f phi_i - grad phi_i . grad u) w as the callback leaves them (before the boundary rows)"""
    nc = NDOF[fe]
    ndof = int(ed[:, :nc].max()) + 1
    w, xg = gauss(order)
    PHI, DPHI = basis(fe, xg)
    K = np.zeros((ndof, ndof))
    F = np.zeros(ndof)
    u = np.zeros(ndof) if sol is None else sol
    for e in range(ed.shape[0]):
        dof = ed[e, :nc]
        x = xs[dof]
        Ke = np.zeros((nc, nc))
        Fe = np.zeros(nc)
        for g in range(w.size):
            J = DPHI[g].T @ x                                  # J[a][b] = sum_n dphi_n/dxi_a x_n[b]
            det = J[0, 0] * J[1, 1] - J[0, 1] * J[1, 0]
            Ji = np.array([[J[1, 1], -J[0, 1]], [-J[1, 0], J[0, 0]]]) / det
            grad = DPHI[g] @ Ji.T                              # grad phi_n [b] = sum_a Jinv[b][a] dphi_n/dxi_a
            weight = det * w[g]
            gu = grad.T @ u[dof]
            xq = PHI[g] @ x
            f = source(xq)
            Ke += (grad @ grad.T) * weight
            Fe += (f * PHI[g] - grad @ gu) * weight
        K[np.ix_(dof, dof)] += Ke
        F[dof] += Fe
    return K, F


def neumann(ed, xs, ff, fe, flux_by_flag, order="seventh"):
    """edge integrals of the flux (JacobianSur of the line element on the edge's nodes: ends, then middle)"""
    nfn = 2 if fe == "linear" else 3
    lfe = "linear" if fe == "linear" else "biquadratic"
    w, xg = fo.gauss_table("line", order)
    xg = np.asarray(xg).reshape(-1)
    nodes = (0, 2) if nfn == 2 else (0, 2, 1)
    lag, dlag = (fo.lag_linear, fo.dlag_linear) if nfn == 2 else (fo.lag_biquadratic, fo.dlag_biquadratic)
    nc = NDOF[fe]
    F = np.zeros(int(ed[:, :nc].max()) + 1)
    for e in range(ed.shape[0]):
        for f in range(3):
            if ff[e, f] in flux_by_flag:
                fn = ed[e, FACE[f][:nfn]]
                x = xs[fn]
                for g in range(xg.size):
                    ph = np.array([lag(xg[g], I) for I in nodes])
                    dp = np.array([dlag(xg[g], I) for I in nodes])
                    t = dp @ x
                    tau = flux_by_flag[ff[e, f]]
                    tv = tau(ph @ x) if callable(tau) else tau
                    F[fn] += ph * tv * np.hypot(t[0], t[1]) * w[g]
    return F


def dirichlet(ed, ff, fe, flags):
    nfn = 2 if fe == "linear" else 3
    out = set()
    for e in range(ed.shape[0]):
        for f in range(3):
            if ff[e, f] in flags:
                out.update(int(n) for n in ed[e, FACE[f][:nfn]])
    return np.array(sorted(out), dtype=np.int64)


def solve(nx, ny, nlevels, fe, source, dirichlet_flags=(-2, -3, -4, -5), flux_by_flag=None, values=None):
    """the discrete problem of the finest of nlevels levels, solved directly; values(x): Dirichlet value at a boundary node (default 0)"""
    meshes = [box_mesh(nx, ny)]
    for _ in range(1, nlevels):
        meshes.append(refine(*meshes[-1][:3]))
    ed, xs, ff, own = meshes[-1]
    ndof = n_dofs(own, fe)
    bdc = dirichlet(ed, ff, fe, set(dirichlet_flags))
    sol = np.zeros(ndof)
    if values is not None:
        for n in bdc:
            sol[n] = values(xs[n])
    K, F = assemble(ed, xs, fe, source, sol)
    if flux_by_flag:
        F = F + neumann(ed, xs, ff, fe, flux_by_flag)
    K[bdc, :] = 0.0
    K[bdc, bdc] = 1.0
    F[bdc] = 0.0
    return sol + np.linalg.solve(K, F), meshes
