"""TEST INFRASTRUCTURE ONLY (oracle).  CPU restatement of the one-dimensional case of applications/001_Poisson (its shipped input/input1D.json): the EDGE3
box, its numbering, the callback's advection-diffusion form with the streamline-upwind terms, the boundary rows and the solve -- numpy, loops as the reference
writes them.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.

  box_mesh      MeshGeneration.cpp:78-262 (case 1: nodes i / (2 nx), element i = {2 i, 2 i + 2, 2 i + 1}, face 0 of the first element "left" = -2, face 1 of the
                last "right" = -3) + the renumbering every FEMuS mesh goes through (vertices first, then the middles, each class in order of first appearance)
  tables        1d/Edge.hpp:72-104 (LineLinear, LineBiquadratic) at the line Gauss points: pinned bit for bit by tests/golden/fe_tables.npz (oracle/_ref)
  assemble      main.cpp:355-480 with dim == 1 (V = 1, nu = 0.01 at :392-395); elem_type_1D::Jacobian, ElemType.hpp:994-1035
  solve         LinearImplicitSystem::MGsolve with one level: exact solve of KK EPS = RES, Sol += EPS (boundary rows: MatZeroRows with 1 on the diagonal, RES = 0)
"""
import numpy as np

from . import femus_oracle as fo


def box_mesh(nx, xa=0.0, xb=1.0):
    x = np.array([(i / (2.0 * nx)) * (xb - xa) + xa for i in range(2 * nx + 1)])
    ed = np.array([[2 * i, 2 * i + 2, 2 * i + 1] for i in range(nx)])
    face = np.full((nx, 2), -1)
    face[0, 0] = -2            # "left"
    face[nx - 1, 1] = -3       # "right"
    # renumbering: vertices (local nodes 0, 1) first, then the middles, first appearance walking the elements
    new = np.full(2 * nx + 1, -1)
    k = 0
    for cls in ((0, 1), (2,)):
        for e in range(nx):
            for l in cls:
                if new[ed[e, l]] < 0:
                    new[ed[e, l]] = k
                    k += 1
    xs = np.empty_like(x)
    xs[new] = x
    return new[ed], xs, face, nx + 1          # elem_dof, coords, face flags, number of vertices


def line_tables(fe, order="seventh"):
    w, xg = fo.gauss_table("line", order)
    xg = np.asarray(xg).reshape(-1)
    nodes = (0, 2) if fe == "linear" else (0, 2, 1)          # 1-D index I = xc + 1 of the local nodes (ends -1, +1, middle 0)
    lag, dlag = (fo.lag_linear, fo.dlag_linear) if fe == "linear" else (fo.lag_biquadratic, fo.dlag_biquadratic)
    phi = np.array([[lag(x, I) for I in nodes] for x in xg])
    dphi = np.array([[dlag(x, I) for I in nodes] for x in xg])
    d2phi = np.zeros_like(phi) if fe == "linear" else np.array([[fo.d2lag_biquadratic(x, I) for I in nodes] for x in xg])
    return np.asarray(w), phi, dphi, d2phi


def assemble(elem_dof, coords, fe, sol, source, nu=0.01, V=1.0, order="seventh"):
    """dense KK and RES as the callback leaves them (before the boundary rows)"""
    nc = 2 if fe == "linear" else 3
    ndof = int(elem_dof[:, :nc].max()) + 1
    w, PHI, DPHI, D2PHI = line_tables(fe, order)
    KK = np.zeros((ndof, ndof))
    RES = np.zeros(ndof)
    for e in range(elem_dof.shape[0]):
        dof = elem_dof[e, :nc]
        x = coords[dof]
        u = sol[dof]
        VxiHxi = (x[1] - x[0]) * V
        PeXi = VxiHxi / (2. * nu)
        barXi = 0. if abs(PeXi) < 1.0e-10 else 1. / np.tanh(PeXi) - 1. / PeXi
        barNu = barXi * VxiHxi / 2.
        vL2Norm2 = V * V
        supgTau = barNu / vL2Norm2 if vL2Norm2 > 1.0e-15 else 0.
        F = np.zeros(nc)
        B = np.zeros((nc, nc))
        for g in range(w.size):
            Jac = 0.0
            for n in range(nc):
                Jac += DPHI[g, n] * x[n]
            weight = Jac * w[g]
            JacI = 1 / Jac
            phi = PHI[g]
            gradphi = DPHI[g] * JacI
            nablaphi = D2PHI[g] * JacI * JacI
            xg = gradSol = nablaSol = 0.0
            for n in range(nc):
                xg += x[n] * phi[n]
                gradSol += gradphi[n] * u[n]
                nablaSol += nablaphi[n] * u[n]
            src = source(xg)
            for i in range(nc):
                lapRhs = nu * gradphi[i] * gradSol
                advRhs = V * gradSol * phi[i]
                resRhs = -nu * nablaSol + V * gradSol
                supgPhi = (V * gradphi[i] + nu * nablaphi[i]) * supgTau
                F[i] += (src * phi[i] - lapRhs - advRhs + (src - resRhs) * supgPhi) * weight
                for j in range(nc):
                    lap = nu * (gradphi[i] * gradphi[j] - nablaphi[j] * supgPhi) * weight
                    adv = V * gradphi[j] * (phi[i] + supgPhi) * weight
                    B[i, j] += lap + adv
        RES[dof] += F
        KK[np.ix_(dof, dof)] += B
    return KK, RES


def solve(nx, fe, source, dirichlet_left=0.0, xa=0.0, xb=1.0, nu=0.01, V=1.0, flux_right=None):
    """input1D.json: Dirichlet on "left", Neumann on "right" -- homogeneous there (no boundary term); a parsed flux g adds g(x) to the row of the end
    point, the side "element" being a point (main.cpp:540-549) --, one level"""
    ed, xs, face, nv = box_mesh(nx, xa, xb)
    nc = 2 if fe == "linear" else 3
    ndof = nv if fe == "linear" else xs.size
    sol = np.zeros(ndof)
    left = int(ed[0, 0])                                   # local node 0 of the element whose face 0 carries the flag
    sol[left] = dirichlet_left
    KK, RES = assemble(ed, xs, fe, sol, source, nu, V)
    if flux_right is not None:
        right = int(ed[nx - 1, 1])
        RES[right] += flux_right(xs[right])
    KK[left, :] = 0.0
    KK[left, left] = 1.0
    RES[left] = 0.0
    return sol + np.linalg.solve(KK, RES), xs[:ndof], (KK, RES)


def refine(ed, xs, face):
    """MeshRefinement::RefineMesh on EDGE3 (the rules of femus_oracle.refine on the line): children 2 e + j, vertex v of child j = coarse node
    fine2CoarseVertexMapping[j][v], a new middle per child at the image of xi = -1/2 / +1/2 under the father's quadratic map (the element prolongator's rows
    0.375, -0.125, 0.75), child j inherits the flag of face j; numbering: vertices first, then middles, first touch"""
    nel = ed.shape[0]
    f2c = np.array([[0, 2], [2, 1]])
    rows = {0: (fo.lag_biquadratic(-0.5, 0), fo.lag_biquadratic(-0.5, 2), fo.lag_biquadratic(-0.5, 1)),       # weights of (end 0, end 1, middle)
            1: (fo.lag_biquadratic(0.5, 0), fo.lag_biquadratic(0.5, 2), fo.lag_biquadratic(0.5, 1))}
    raw = np.zeros((2 * nel, 3), dtype=np.int64)
    x = list(xs)
    ff = np.full((2 * nel, 2), -1)
    for e in range(nel):
        for j in range(2):
            c = 2 * e + j
            raw[c, :2] = ed[e, f2c[j]]
            raw[c, 2] = len(x)
            w = rows[j]
            x.append(w[0] * xs[ed[e, 0]] + w[1] * xs[ed[e, 1]] + w[2] * xs[ed[e, 2]])
            ff[c, j] = face[e, j]
    x = np.array(x)
    new = np.full(x.size, -1)
    k = 0
    for a, b in ((0, 2), (2, 3)):
        for c in range(2 * nel):
            for l in range(a, b):
                if new[raw[c, l]] < 0:
                    new[raw[c, l]] = k
                    k += 1
    xf = np.empty_like(x)
    xf[new] = x
    return new[raw], xf, ff, 2 * nel + 1


def solve_levels(nx, nlevels, fe, source, dirichlet_left=0.0, xa=0.0, xb=1.0, nu=0.01, V=1.0):
    """the discrete problem of the FINEST of nlevels levels, solved directly (what the multigrid iteration of the application converges to), and the meshes"""
    meshes = [box_mesh(nx, xa, xb)]
    for _ in range(1, nlevels):
        meshes.append(refine(*meshes[-1][:3]))
    ed, xs, face, nv = meshes[-1]
    ndof = nv if fe == "linear" else xs.size
    sol = np.zeros(ndof)
    left = int(ed[np.where(face[:, 0] == -2)[0][0], 0])
    sol[left] = dirichlet_left
    KK, RES = assemble(ed, xs, fe, sol, source, nu, V)
    KK[left, :] = 0.0
    KK[left, left] = 1.0
    RES[left] = 0.0
    return sol + np.linalg.solve(KK, RES), meshes
