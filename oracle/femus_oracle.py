"""
ORACLE -- TEST INFRASTRUCTURE ONLY.

CPU (numpy/scipy) restatement of the FEMuS hot path named by BASELINE.json:north_star:
FE tables -> element Jacobian / stiffness / residual -> box mesh + uniform refinement + DOF
numbering -> CSR assembly -> Dirichlet rows -> prolongators -> Galerkin coarse operators ->
multiplicative V-cycle (Richardson/Jacobi) -> outer Richardson / GMRES / PCG.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this
package; the product (femus_amd/) never does.

Pinning status (see DESIGN.md "Oracle"):
  * Gauss tables, 1-D/2-D/3-D Lagrange bases, node tables: PINNED against the reference's own
    compiled sources (oracle/_ref/libfemus_ref_fe.so, built by oracle/Makefile from
    /root/reference/src/02_reference_geom_elements/{00_definition,01_fe,02_quadrature}) and against
    tests/golden/fe_tables.npz generated from that library by tests/golden/make_golden.py.
  * Jacobian / element matrix / mesh numbering / prolongator / MG cycle: the reference code for
    these needs boost, a cmake-generated FemusConfig.hpp and PETSc 3.20.2 (not in the image) =>
    "parity unpinned" by a reference run; restated from the cited file:line and checked by
    analytic properties (partition of unity, polynomial exactness, manufactured solution).

All paths below are relative to /root/reference/.
"""
import numpy as np
import scipy.sparse as sp

# ----------------------------------------------------------------------------------------------
# a1. Gauss tables  (src/02_reference_geom_elements/02_quadrature/quadrature_interface.cpp:36-94,
#     1d/quadrature_Line.cpp:10-36, 2d/quadrature_Quadrangle.cpp:11-, 3d/quadrature_Hexahedron.cpp:11-)
# The reference stores tensor-product Gauss-Legendre rules as literals with 14 significant digits
# (e.g. 0.57735026918963).  Rules 0..4 are reproduced as: exact rule -> tensor product -> round to
# 14 significant digits; bit-equality with the reference literals is asserted in tests/test_oracle_ref.py.
# ----------------------------------------------------------------------------------------------
_ORDER_INDEX = {"zero": 0, "first": 0, "second": 1, "third": 1, "fourth": 2, "fifth": 2,
                "sixth": 3, "seventh": 3, "eighth": 4, "ninth": 4}


_WEIGHT_LITERAL_OVERRIDE = {("hex", 4): (0.042091477490532, 0.078911515795071, 0.14794033605678, 0.27735296695391)}


def _round14(v):
    v = np.asarray(v, dtype=np.float64)
    out = np.array([float("%.14g" % x) for x in v.ravel()]).reshape(v.shape)
    return out + 0.0  # turn -0.0 into 0.0


def gauss_order_index(order):
    return _ORDER_INDEX[order]


def gauss_1d_exact(n):
    x, w = np.polynomial.legendre.leggauss(n)
    x = 0.5 * (x - x[::-1])  # symmetrise
    w = 0.5 * (w + w[::-1])
    return x, w


def gauss_table(geom, order):
    """returns (w[ng], x[ng, dim]) in the reference's point order (first coordinate slowest)."""
    n = gauss_order_index(order) + 1
    dim = {"line": 1, "quad": 2, "hex": 3}[geom]
    x1, w1 = gauss_1d_exact(n)
    if dim == 1:
        w = w1.copy()
        x = x1[:, None].copy()
    elif dim == 2:
        w = (w1[:, None] * w1[None, :]).ravel()
        X, Y = np.meshgrid(x1, x1, indexing="ij")
        x = np.stack([X.ravel(), Y.ravel()], axis=1)
    else:
        w = (w1[:, None, None] * w1[None, :, None] * w1[None, None, :]).ravel()
        X, Y, Z = np.meshgrid(x1, x1, x1, indexing="ij")
        x = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
    if n == 1:  # {2},{0} / {4},{0},{0} / {8},{0},{0},{0}
        return w, x * 0.0
    w, x = _round14(w), _round14(x)
    if (geom, n) in _WEIGHT_LITERAL_OVERRIDE:
        # the reference literal differs from round14(exact product) in the last digit for this rule
        # (3d/quadrature_Hexahedron.cpp, Gauss3): weights are classed by how many of the three 1-D
        # indices are inner points of the 4-point rule.
        inner = ((np.arange(n) > 0) & (np.arange(n) < n - 1)).astype(np.int64)
        cls = (inner[:, None, None] + inner[None, :, None] + inner[None, None, :]).ravel()
        w = np.asarray(_WEIGHT_LITERAL_OVERRIDE[(geom, n)])[cls]
    return w, x


# ----------------------------------------------------------------------------------------------
# a2. 1-D Lagrange polynomials (src/02_reference_geom_elements/01_fe/1d/Edge.hpp:72-104) and the
#     tensor-product hex / quad bases (3d/Hexahedron.cpp:95-163, 2d/Quadrilateral.cpp:68-110)
# index i in {0,1,2}: 0 -> node at -1, 1 -> node at 0, 2 -> node at +1
# ----------------------------------------------------------------------------------------------
def lag_linear(x, i):
    return (i == 0) * 0.5 * (1. - x) + (i == 2) * 0.5 * (1. + x)


def dlag_linear(x, i):
    return (i == 0) * (-0.5) + (i == 2) * 0.5 + 0.0 * x


def lag_biquadratic(x, i):
    return (i == 0) * 0.5 * x * (x - 1.) + (i == 1) * (1. - x) * (1. + x) + (i == 2) * 0.5 * x * (1. + x)


def dlag_biquadratic(x, i):
    return (i == 0) * (x - 0.5) + (i == 1) * (-2. * x) + (i == 2) * (x + 0.5)


def d2lag_biquadratic(x, i):
    return (i == 0) * 1.0 + (i == 1) * (-2.) + (i == 2) * 1.0 + 0.0 * x


# "quadratic" 1-D factors of the serendipity families (Edge.hpp:81-91): linear at the end nodes, the bubble at the middle one
def lag_quadratic(x, i):
    return (i == 0) * 0.5 * (1. - x) + (i == 1) * (1. - x) * (1. + x) + (i == 2) * 0.5 * (1. + x)


def dlag_quadratic(x, i):
    return (i == 0) * (-0.5) + (i == 1) * (-2. * x) + (i == 2) * 0.5


def d2lag_quadratic(x, i):
    return (i == 1) * (-2.) + 0.0 * x


# FEMuS local node order (the convention of hex_lag::Xc / quad_lag::Xc): vertices, edge mid-points
# (bottom ring, top ring, vertical), side-face centres (y-,x+,y+,x-), bottom, top, centre.
XC_HEX27 = np.array(
    [[-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, 1], [1, -1, 1], [1, 1, 1], [-1, 1, 1],
     [0, -1, -1], [1, 0, -1], [0, 1, -1], [-1, 0, -1], [0, -1, 1], [1, 0, 1], [0, 1, 1], [-1, 0, 1],
     [-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0],
     [0, -1, 0], [1, 0, 0], [0, 1, 0], [-1, 0, 0], [0, 0, -1], [0, 0, 1], [0, 0, 0]], dtype=np.float64)
XC_QUAD9 = np.array([[-1, -1], [1, -1], [1, 1], [-1, 1], [0, -1], [1, 0], [0, 1], [-1, 0], [0, 0]],
                    dtype=np.float64)


def xc_table(geom):
    return {"hex": XC_HEX27, "quad": XC_QUAD9}[geom]


def ind_table(geom):
    """IND[j][d] in {0,1,2} = 1-D node index of local node j along d (Hexahedron.cpp:40-47)."""
    return (xc_table(geom) + 1).astype(np.int64)


def ndofs(geom, fe):
    """basis::n_dofs of the element families: Lagrange linear / serendipity ("quadratic") / biquadratic, piecewise constant (quad0 / hex0)"""
    dim = xc_table(geom).shape[1]
    return {"linear": 2 ** dim, "serendipity": {2: 8, 3: 20}[dim], "biquadratic": 3 ** dim, "constant": 1}[fe]


def n_vertices(geom):
    return 2 ** xc_table(geom).shape[1]


def class_ranges(geom):
    """local-node ranges of the three C0 Lagrange families: [0,nv) vertices, [nv,ne) edge mids,
    [ne,nc) face/centre (elem::GetElementDofNumber(iel,k), k=0,1,2)."""
    return {"hex": (8, 20, 27), "quad": (4, 8, 9)}[geom]


def eval_basis(geom, fe, pts):
    """phi[np, nc], dphi[np, nc, dim], d2phi[np, nc, ndd] at reference points pts[np, dim].
    d2 order: 3-D (xx, yy, zz, xy, yz, zx); 2-D (xx, yy, xy)."""
    pts = np.atleast_2d(np.asarray(pts, dtype=np.float64))
    dim = pts.shape[1]
    nc = ndofs(geom, fe)
    IND = ind_table(geom)[:nc]
    if fe == "constant":          # quad0 / hex0 (Quadrilateral.hpp:173-, Hexahedron.hpp:196-): the constant one, every derivative zero
        npts = pts.shape[0]
        return np.ones((npts, 1)), np.zeros((npts, 1, dim)), np.zeros((npts, 1, 3 if dim == 2 else 6))
    if fe == "serendipity":
        return _eval_serendipity(geom, pts)
    if fe == "linear":
        L, D = lag_linear, dlag_linear
        D2 = lambda x, i: 0.0 * x
    else:
        L, D, D2 = lag_biquadratic, dlag_biquadratic, d2lag_biquadratic
    npts = pts.shape[0]
    phi = np.empty((npts, nc))
    dphi = np.empty((npts, nc, dim))
    d2 = np.empty((npts, nc, 3 if dim == 2 else 6))
    for j in range(nc):
        l = [L(pts[:, d], IND[j, d]) for d in range(dim)]
        dl = [D(pts[:, d], IND[j, d]) for d in range(dim)]
        d2l = [D2(pts[:, d], IND[j, d]) for d in range(dim)]
        if dim == 2:
            phi[:, j] = l[0] * l[1]
            dphi[:, j, 0] = dl[0] * l[1]
            dphi[:, j, 1] = l[0] * dl[1]
            d2[:, j, 0] = d2l[0] * l[1]
            d2[:, j, 1] = l[0] * d2l[1]
            d2[:, j, 2] = dl[0] * dl[1]
        else:
            phi[:, j] = l[0] * l[1] * l[2]
            dphi[:, j, 0] = dl[0] * l[1] * l[2]
            dphi[:, j, 1] = l[0] * dl[1] * l[2]
            dphi[:, j, 2] = l[0] * l[1] * dl[2]
            d2[:, j, 0] = d2l[0] * l[1] * l[2]
            d2[:, j, 1] = l[0] * d2l[1] * l[2]
            d2[:, j, 2] = l[0] * l[1] * d2l[2]
            d2[:, j, 3] = dl[0] * dl[1] * l[2]
            d2[:, j, 4] = l[0] * dl[1] * dl[2]
            d2[:, j, 5] = dl[0] * l[1] * dl[2]
    return phi, dphi, d2


def _eval_serendipity(geom, pts):
    """QuadQuadratic (Quadrilateral.cpp:113-161) / HexQuadratic (Hexahedron.cpp:167-256), the expressions term by term and in their order: an edge
    function is the plain product of the 1-D "quadratic" factors, a vertex function that product times (-1 + ix x + jx y) resp. (-2 + ix x + jx y + kx z)"""
    dim = pts.shape[1]
    nc = ndofs(geom, "serendipity")
    IND = ind_table(geom)[:nc]
    npts = pts.shape[0]
    phi = np.empty((npts, nc))
    dphi = np.empty((npts, nc, dim))
    d2 = np.empty((npts, nc, 3 if dim == 2 else 6))
    x = [pts[:, d] for d in range(dim)]
    for j in range(nc):
        I = IND[j]
        l = [lag_quadratic(x[d], I[d]) for d in range(dim)]
        dl = [dlag_quadratic(x[d], I[d]) for d in range(dim)]
        sl = [d2lag_quadratic(x[d], I[d]) for d in range(dim)]
        if dim == 2:
            ix, jx = I[0] - 1., I[1] - 1.
            if abs(ix * jx) == 0:
                phi[:, j] = l[0] * l[1]
                dphi[:, j, 0] = dl[0] * l[1]
                dphi[:, j, 1] = l[0] * dl[1]
                d2[:, j, 0] = sl[0] * l[1]
                d2[:, j, 1] = l[0] * sl[1]
                d2[:, j, 2] = dl[0] * dl[1]
            else:
                s = -1. + ix * x[0] + jx * x[1]
                phi[:, j] = s * l[0] * l[1]
                dphi[:, j, 0] = l[1] * (ix * l[0] + s * dl[0])
                dphi[:, j, 1] = l[0] * (jx * l[1] + s * dl[1])
                d2[:, j, 0] = l[1] * (2. * ix * dl[0] + s * sl[0])
                d2[:, j, 1] = l[0] * (2. * jx * dl[1] + s * sl[1])
                d2[:, j, 2] = ix * l[0] * dl[1] + jx * l[1] * dl[0] + s * dl[0] * dl[1]
            continue
        ix, jx, kx = I[0] - 1., I[1] - 1., I[2] - 1.
        if abs(ix * jx * kx) == 0:
            phi[:, j] = l[0] * l[1] * l[2]
            dphi[:, j, 0] = dl[0] * l[1] * l[2]
            dphi[:, j, 1] = l[0] * dl[1] * l[2]
            dphi[:, j, 2] = l[0] * l[1] * dl[2]
            d2[:, j, 0] = sl[0] * l[1] * l[2]
            d2[:, j, 1] = l[0] * sl[1] * l[2]
            d2[:, j, 2] = l[0] * l[1] * sl[2]
            d2[:, j, 3] = dl[0] * dl[1] * l[2]
            d2[:, j, 4] = l[0] * dl[1] * dl[2]
            d2[:, j, 5] = dl[0] * l[1] * dl[2]
        else:
            s = -2. + ix * x[0] + jx * x[1] + kx * x[2]
            phi[:, j] = s * l[0] * l[1] * l[2]
            dphi[:, j, 0] = l[1] * l[2] * (ix * l[0] + s * dl[0])
            dphi[:, j, 1] = l[0] * l[2] * (jx * l[1] + s * dl[1])
            dphi[:, j, 2] = l[0] * l[1] * (kx * l[2] + s * dl[2])
            d2[:, j, 0] = l[1] * l[2] * (2. * ix * dl[0] + s * sl[0])
            d2[:, j, 1] = l[2] * l[0] * (2. * jx * dl[1] + s * sl[1])
            d2[:, j, 2] = l[0] * l[1] * (2. * kx * dl[2] + s * sl[2])
            d2[:, j, 3] = l[2] * (ix * l[0] * dl[1] + jx * l[1] * dl[0] + s * dl[0] * dl[1])
            d2[:, j, 4] = l[0] * (jx * l[1] * dl[2] + kx * l[2] * dl[1] + s * dl[1] * dl[2])
            d2[:, j, 5] = l[1] * (kx * l[2] * dl[0] + ix * l[0] * dl[2] + s * dl[2] * dl[0])
    return phi, dphi, d2


# ----------------------------------------------------------------------------------------------
# a3. FE-at-quadrature tables (03_fe_evaluations_at_quadrature/ElemType.cpp:576-633, 637-741):
#     _phi[ig][j], _dphidxi[ig][j] ... row-major [ng][nc]
# ----------------------------------------------------------------------------------------------
class ElemType:
    def __init__(self, geom, fe, order="seventh"):
        self.geom, self.fe, self.order = geom, fe, order
        self.dim = xc_table(geom).shape[1]
        self.nc = ndofs(geom, fe)
        self.w, self.xg = gauss_table(geom, order)
        self.ng = self.w.size
        self.phi, self.dphi, self.d2phi = eval_basis(geom, fe, self.xg)

    # a4. elem_type_{2,3}D::Jacobian_type<double> (ElemType.hpp:1183-1248, 1438-1537)
    def jacobian(self, vt, ig, nabla=False):
        """vt[dim][>=nc] element node coordinates (SoA).  Returns Weight, phi[nc], gradphi[nc*dim]; with nabla=True also the optional Hessians
        nablaphi[nc*nh] of ElemType.hpp:1232-1244 (2-D: xx, yy, xy) / :1509-1534 (3-D: xx, yy, zz, xy, yz, zx), bracketed as written there."""
        dim, nc = self.dim, self.nc
        Jac = np.zeros((dim, dim))
        for inode in range(nc):  # same accumulation order as the reference loop
            for a in range(dim):
                for b in range(dim):
                    Jac[a, b] += self.dphi[ig, inode, a] * vt[b][inode]
        if dim == 2:
            det = Jac[0, 0] * Jac[1, 1] - Jac[0, 1] * Jac[1, 0]
            JacI = np.array([[Jac[1, 1] / det, -Jac[0, 1] / det], [-Jac[1, 0] / det, Jac[0, 0] / det]])
        else:
            det = (Jac[0, 0] * (Jac[1, 1] * Jac[2, 2] - Jac[1, 2] * Jac[2, 1]) +
                   Jac[0, 1] * (Jac[1, 2] * Jac[2, 0] - Jac[1, 0] * Jac[2, 2]) +
                   Jac[0, 2] * (Jac[1, 0] * Jac[2, 1] - Jac[1, 1] * Jac[2, 0]))
            JacI = np.empty((3, 3))
            JacI[0, 0] = (-Jac[1, 2] * Jac[2, 1] + Jac[1, 1] * Jac[2, 2]) / det
            JacI[0, 1] = (Jac[0, 2] * Jac[2, 1] - Jac[0, 1] * Jac[2, 2]) / det
            JacI[0, 2] = (-Jac[0, 2] * Jac[1, 1] + Jac[0, 1] * Jac[1, 2]) / det
            JacI[1, 0] = (Jac[1, 2] * Jac[2, 0] - Jac[1, 0] * Jac[2, 2]) / det
            JacI[1, 1] = (-Jac[0, 2] * Jac[2, 0] + Jac[0, 0] * Jac[2, 2]) / det
            JacI[1, 2] = (Jac[0, 2] * Jac[1, 0] - Jac[0, 0] * Jac[1, 2]) / det
            JacI[2, 0] = (-Jac[1, 1] * Jac[2, 0] + Jac[1, 0] * Jac[2, 1]) / det
            JacI[2, 1] = (Jac[0, 1] * Jac[2, 0] - Jac[0, 0] * Jac[2, 1]) / det
            JacI[2, 2] = (-Jac[0, 1] * Jac[1, 0] + Jac[0, 0] * Jac[1, 1]) / det
        weight = det * self.w[ig]
        gradphi = np.empty(nc * dim)
        for inode in range(nc):
            for a in range(dim):
                s = self.dphi[ig, inode, 0] * JacI[a, 0]
                for b in range(1, dim):
                    s = s + self.dphi[ig, inode, b] * JacI[a, b]
                gradphi[dim * inode + a] = s
        if not nabla:
            return weight, self.phi[ig].copy(), gradphi
        nh = 3 if dim == 2 else 6
        nablaphi = np.empty(nc * nh)
        pairs = [(0, 0), (1, 1), (0, 1)] if dim == 2 else [(0, 0), (1, 1), (2, 2), (0, 1), (1, 2), (2, 0)]
        for inode in range(nc):
            h = self.d2phi[ig, inode]
            if dim == 2:      # rows of the reference Hessian: (dxi2, dxideta), (dxideta, deta2)
                H = [[h[0], h[2]], [h[2], h[1]]]
            else:             # (dxi2, dxideta, dzetadxi), (dxideta, deta2, detadzeta), (dzetadxi, detadzeta, dzeta2)
                H = [[h[0], h[3], h[5]], [h[3], h[1], h[4]], [h[5], h[4], h[2]]]
            for k, (a, b) in enumerate(pairs):
                out = 0.0
                for r in range(dim):
                    row = H[r][0] * JacI[a, 0]
                    for c in range(1, dim):
                        row = row + H[r][c] * JacI[a, c]
                    out = out + row * JacI[b, r]
                nablaphi[nh * inode + k] = out
        return weight, self.phi[ig].copy(), gradphi, nablaphi


# ----------------------------------------------------------------------------------------------
# a7. Poisson element loop (src/08_equations/assemble/
#     00_poisson_eqn_with_all_dirichlet_bc_AD_or_nonAD_separate.hpp:111-215)
#   Res[i] += (-f(x_g) phi_i - grad phi_i . grad u) w ;  Jac[i,j] += (grad phi_i . grad phi_j) w
# ----------------------------------------------------------------------------------------------
def elem_poisson(et, x, solu, rhs):
    """x[dim][n_geom_nodes], solu[nc]; rhs(x_gss[dim]) -> float.  Pure-python loops, reference order."""
    dim, nc = et.dim, et.nc
    Res = np.zeros(nc)
    Jac = np.zeros(nc * nc)
    for ig in range(et.ng):
        weight, phi, phi_x = et.jacobian(x, ig)
        gradSolu = np.zeros(dim)
        x_gss = np.zeros(dim)
        for i in range(nc):
            for jdim in range(dim):
                gradSolu[jdim] += phi_x[i * dim + jdim] * solu[i]
                x_gss[jdim] += x[jdim][i] * phi[i]
        f = rhs(x_gss)
        for i in range(nc):
            weakLaplace = 0.
            for jdim in range(dim):
                weakLaplace += phi_x[i * dim + jdim] * gradSolu[jdim]
            Res[i] += (-f * phi[i] - weakLaplace) * weight
            for j in range(nc):
                weakLaplacej = 0.
                for kdim in range(dim):
                    weakLaplacej += phi_x[i * dim + kdim] * phi_x[j * dim + kdim]
                Jac[i * nc + j] += weakLaplacej * weight
    return Jac.reshape(nc, nc), Res


def elem_poisson_batch(et, X, U, rhs_vec):
    """Vectorised restatement for many elements: X[nel, dim, nc_geom], U[nel, nc],
    rhs_vec(xg[nel, ng, dim]) -> f[nel, ng].  Gauss-point sum is sequential (reference order)."""
    nel = X.shape[0]
    dim, nc, ng = et.dim, et.nc, et.ng
    Jm = np.einsum("gna,ebn->egab", et.dphi, X[:, :, :nc])
    det = np.linalg.det(Jm)
    JI = np.linalg.inv(Jm)                       # JI[e,g,b,a] : d xi_a / d x_b  laid as inverse of J[a,b]=dx_b/dxi_a
    grad = np.einsum("gna,egba->egnb", et.dphi, JI)   # grad[e,g,n,b] = sum_a dphi[g,n,a] * JI[b,a]
    w = det * et.w[None, :]
    xg = np.einsum("gn,ebn->egb", et.phi, X[:, :, :nc])
    f = rhs_vec(xg)
    gu = np.einsum("egnb,en->egb", grad, U)
    K = np.zeros((nel, nc, nc))
    F = np.zeros((nel, nc))
    for g in range(ng):
        G = grad[:, g]                           # [nel, nc, dim]
        K += np.einsum("eid,ejd->eij", G, G) * w[:, g, None, None]
        F += (-f[:, g, None] * et.phi[g][None, :] - np.einsum("eid,ed->ei", G, gu[:, g])) * w[:, g, None]
    return K, F


# ----------------------------------------------------------------------------------------------
# a6. element prolongator (ElemType.cpp:439-532): for each fine node (child j, local i) of a refined
#     element the coarse basis values |phi| >= 1e-14 at that node.
# ----------------------------------------------------------------------------------------------
def child_node_ref_coords(geom):
    """X[j, i, :] = reference coordinate (coarse element) of local node i of child j.
    child j is the sub-element at coarse vertex j (fine2CoarseVertexMapping, Hexahedron.cpp:75-83)."""
    Xc = xc_table(geom)
    nv = n_vertices(geom)
    return 0.5 * (Xc[:nv, None, :] + Xc[None, :, :])


def fine2coarse_vertex_mapping(geom):
    """f2c[j][v] = coarse local node sitting at vertex v of child j."""
    Xc = xc_table(geom)
    nv = n_vertices(geom)
    X = child_node_ref_coords(geom)[:, :nv, :]
    out = np.empty((nv, nv), dtype=np.int64)
    for j in range(nv):
        for v in range(nv):
            out[j, v] = np.where(np.all(Xc == X[j, v], axis=1))[0][0]
    return out


def elem_prolongator(geom, fe):
    """P[j, i, :] coarse-basis values at (child j, local node i); entries with |.|<1e-14 set to 0
    (they are dropped from the sparse row in the reference)."""
    nc = ndofs(geom, fe)
    X = child_node_ref_coords(geom)[:, :nc, :]
    nv = n_vertices(geom)
    phi, _, _ = eval_basis(geom, fe, X.reshape(-1, X.shape[-1]))
    phi = phi.reshape(nv, nc, nc)
    phi[np.abs(phi) < 1.0e-14] = 0.0
    return phi


# ----------------------------------------------------------------------------------------------
# a10. meshes: box generator, uniform refinement, first-touch node renumbering (nprocs = 1)
#   MeshGeneration.cpp:790-849 (nodes), :979-1075 (elements, boundary flags)
#   MeshRefinement.cpp:240-294 (children, inherited vertices), :356-417 (edge mids),
#   :513-620 (face centres, element centres) ; Mesh.cpp:517-559 (node renumbering)
# ----------------------------------------------------------------------------------------------
class Mesh:
    """single-level mesh in FEMuS numbering (nprocs=1).
    elem_dof[nel, nloc] biquadratic node ids; coords[nnode, dim]; face_flag[nel, nfaces] (<-1 boundary)."""

    def __init__(self, geom, elem_dof, coords, face_flag, level=0):
        self.geom = geom
        self.dim = coords.shape[1]
        self.elem_dof = elem_dof
        self.coords = coords
        self.face_flag = face_flag
        self.level = level
        self.nel = elem_dof.shape[0]
        self.nnode = coords.shape[0]
        self.own_size = None      # [n_vertex_nodes, +edge, +face] cumulative counts (dofOffset of families 0,1,2)
        self.child_elem = None    # set on the coarse mesh by refine(): [nel, nchild]


def face_nodes(geom):
    """local nodes on each face (set semantics of hex_lag::faceDofs / quad_lag::faceDofs); face f is
    identified by its centre node: hex 20+f, quad 4+f."""
    Xc = xc_table(geom)
    if geom == "hex":
        centres = list(range(20, 26))
    else:
        centres = list(range(4, 8))
    out = []
    for c in centres:
        d = np.nonzero(Xc[c])[0][0]
        out.append(np.where(Xc[:, d] == Xc[c, d])[0])
    return out


def _first_touch_renumber(geom, elem_dof, nnode):
    """Mesh.cpp:517-559 for nprocs=1: for class k in (vertices, edges, faces+centre): for iel: for local
    nodes of class k: first touch gets the next id.  Returns mapping old->new and the cumulative counts."""
    nv, ne, nc = class_ranges(geom)
    mapping = np.full(nnode, -1, dtype=np.int64)
    counter = 0
    own = []
    for (a, b) in ((0, nv), (nv, ne), (ne, nc)):
        seq = elem_dof[:, a:b].ravel()            # element-major, local order: the visiting order
        seq = seq[mapping[seq] < 0]               # not yet numbered by an earlier class
        uniq, first = np.unique(seq, return_index=True)
        order = np.argsort(first, kind="stable")
        mapping[uniq[order]] = counter + np.arange(uniq.size)
        counter += uniq.size
        own.append(counter)
    assert counter == nnode
    return mapping, own


def coarse_box_mesh(nx, ny, nz, lo=(0., 0., 0.), hi=(1., 1., 1.)):
    """GenerateCoarseBoxMesh for QUAD9 (nz=0) / HEX27: lexicographic nodes (x fastest), elements i fastest,
    FEMuS local node order, boundary face flags, then the nprocs=1 renumbering."""
    if nz == 0:
        geom, dim = "quad", 2
        n = (nx, ny)
    else:
        geom, dim = "hex", 3
        n = (nx, ny, nz)
    Xc = xc_table(geom)
    npts = [2 * m + 1 for m in n]
    # coordinates: (i / (2 nx)) * (xmax - xmin) + xmin   (MeshGeneration.cpp:844-846)
    axes = [(np.arange(npts[d], dtype=np.float64) / float(2 * n[d])) * (hi[d] - lo[d]) + lo[d] for d in range(dim)]
    if dim == 2:
        J, I = np.meshgrid(np.arange(npts[1]), np.arange(npts[0]), indexing="ij")
        coords = np.stack([axes[0][I.ravel()], axes[1][J.ravel()]], axis=1)
        ej, ei = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
        ei, ej = ei.ravel(), ej.ravel()
        off = (Xc + 1).astype(np.int64)
        elem_dof = (2 * ei[:, None] + off[None, :, 0]) + (2 * ej[:, None] + off[None, :, 1]) * npts[0]
        face_flag = np.full((nx * ny, 4), -1, dtype=np.int64)
        # QUAD9 (MeshGeneration.cpp 2-D branch): j==0 -> face 0 (-2 "bottom"), i==nx-1 -> face 1 (-3 "right"),
        # j==ny-1 -> face 2 (-4 "top"), i==0 -> face 3 (-5 "left")
        face_flag[ej == 0, 0] = -2
        face_flag[ei == nx - 1, 1] = -3
        face_flag[ej == ny - 1, 2] = -4
        face_flag[ei == 0, 3] = -5
    else:
        K, J, I = np.meshgrid(np.arange(npts[2]), np.arange(npts[1]), np.arange(npts[0]), indexing="ij")
        coords = np.stack([axes[0][I.ravel()], axes[1][J.ravel()], axes[2][K.ravel()]], axis=1)
        ek, ej, ei = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
        ei, ej, ek = ei.ravel(), ej.ravel(), ek.ravel()
        off = (Xc + 1).astype(np.int64)
        elem_dof = ((2 * ei[:, None] + off[None, :, 0]) +
                    npts[0] * ((2 * ej[:, None] + off[None, :, 1]) + (2 * ek[:, None] + off[None, :, 2]) * npts[1]))
        face_flag = np.full((nx * ny * nz, 6), -1, dtype=np.int64)
        face_flag[ek == 0, 4] = -2        # "bottom"
        face_flag[ek == nz - 1, 5] = -7   # "top"
        face_flag[ej == 0, 0] = -3        # "front"
        face_flag[ej == ny - 1, 2] = -5   # "behind"
        face_flag[ei == 0, 3] = -6        # "left"
        face_flag[ei == nx - 1, 1] = -4   # "right"
    nnode = coords.shape[0]
    mapping, own = _first_touch_renumber(geom, elem_dof, nnode)
    new_coords = np.empty_like(coords)
    new_coords[mapping] = coords
    m = Mesh(geom, mapping[elem_dof], new_coords, face_flag, level=0)
    m.own_size = own
    return m


def refine(mc):
    """MeshRefinement::RefineMesh for a fully refined level (nprocs=1)."""
    geom = mc.geom
    nv, ne, nc = class_ranges(geom)
    nchild = nv
    f2c = fine2coarse_vertex_mapping(geom)
    Xc = xc_table(geom)
    nel_f = mc.nel * nchild
    ed = np.full((nel_f, nc), -1, dtype=np.int64)
    # children 'nchild*iel + j', vertex v of child j = coarse node f2c[j][v]   (:240-268)
    ed[:, :nv] = mc.elem_dof[:, f2c].reshape(nel_f, nv)
    # boundary flags: child j inherits coarse face f iff vertex j lies on face f; same local face (:271-278)
    nfaces = mc.face_flag.shape[1]
    fn = face_nodes(geom)
    ff = np.full((nel_f, nfaces), -1, dtype=np.int64)
    ffv = ff.reshape(mc.nel, nchild, nfaces)
    for f in range(nfaces):
        for j in range(nchild):
            if j in fn[f]:
                ffv[:, j, f] = mc.face_flag[:, f]
    nnodes = mc.nnode
    # edge mid-points: loop elements, local edges in order; first visit creates the node (:356-417)
    edge_v = np.empty((ne - nv, 2), dtype=np.int64)
    for e in range(nv, ne):
        d = np.where(Xc[e] == 0)[0][0]
        vs = [v for v in range(nv) if all(Xc[v, k] == Xc[e, k] for k in range(Xc.shape[1]) if k != d)]
        edge_v[e - nv] = sorted(vs)
    a = ed[:, edge_v[:, 0]]
    b = ed[:, edge_v[:, 1]]
    key = (np.minimum(a, b) * np.int64(nnodes) + np.maximum(a, b)).ravel()
    uniq, first, inv = np.unique(key, return_index=True, return_inverse=True)
    rank = np.empty(uniq.size, dtype=np.int64)
    rank[np.argsort(first, kind="stable")] = np.arange(uniq.size)
    ed[:, nv:ne] = (nnodes + rank[inv]).reshape(nel_f, ne - nv)
    nnodes += uniq.size
    if geom == "hex":
        # quad-face centres: loop elements, faces 0..5; shared faces matched by their vertices (:526-561)
        fv = np.stack([np.sort(ed[:, [v for v in fn[f] if v < nv]], axis=1) for f in range(6)], axis=1)  # [nel,6,4]
        base = np.int64(nnodes)
        key = ((fv[:, :, 0] * base + fv[:, :, 1]) * base + fv[:, :, 2]).ravel()  # 3 vertices identify a face
        uniq, first, inv = np.unique(key, return_index=True, return_inverse=True)
        rank = np.empty(uniq.size, dtype=np.int64)
        rank[np.argsort(first, kind="stable")] = np.arange(uniq.size)
        ed[:, 20:26] = (nnodes + rank[inv]).reshape(nel_f, 6)
        nnodes += uniq.size
    # element centres, one per element in element order (:598-616)
    ed[:, nc - 1] = nnodes + np.arange(nel_f)
    nnodes += nel_f
    # Mesh.cpp:517-559 renumbering on the fine level
    mapping, own = _first_touch_renumber(geom, ed, nnodes)
    ed = mapping[ed]
    mf = Mesh(geom, ed, np.zeros((nnodes, mc.dim)), ff, level=mc.level + 1)
    mf.own_size = own
    mc.child_elem = np.arange(nel_f).reshape(mc.nel, nchild)
    # fine coordinates = mesh prolongator (biquadratic) x coarse coordinates (MeshRefinement.cpp:468-475)
    P = build_prolongator(mc, mf, "biquadratic")
    mf.coords = np.stack([P @ mc.coords[:, d] for d in range(mc.dim)], axis=1)
    return mf


def build_levels(nx, ny, nz, nlevels, lo=(0., 0., 0.), hi=(1., 1., 1.)):
    ms = [coarse_box_mesh(nx, ny, nz, lo, hi)]
    for _ in range(1, nlevels):
        ms.append(refine(ms[-1]))
    return ms


# a8/a9: Mesh::GetSolutionDof + LinearEquation::GetSystemDof for one variable, nprocs=1:
# biquadratic dof = node id; linear dof = node id (vertex nodes are numbered first); system row = dof.
def n_dofs(mesh, fe):
    """dofs of one variable of the family (nprocs = 1): nodes are numbered vertices, edge mid-points, the rest, so the linear / serendipity families own the
    leading own_size[0] / own_size[1] node ids; the piecewise constant family owns the elements (Mesh::GetSolutionDof, Mesh.cpp:1021-1074)"""
    if fe == "biquadratic":
        return mesh.nnode
    if fe == "constant":
        return mesh.nel
    return mesh.own_size[{"linear": 0, "serendipity": 1}[fe]]


def elem_sys_dof(mesh, fe):
    """Mesh::GetSolutionDof(i, iel, solType) for all elements (Mesh.cpp:1021-1074, nprocs = 1): the element's i-th node for the Lagrange families
    (:1026-1054), the element itself for the piecewise constant one (:1056-1059)"""
    if fe == "constant":
        return np.arange(mesh.nel, dtype=mesh.elem_dof.dtype)[:, None]
    return mesh.elem_dof[:, :ndofs(mesh.geom, fe)]


# a14. global prolongator (LinearImplicitSystem.cpp:761-909 ; fe_prolongation_matrices.cpp:232-287):
# row = fine dof of (child j, local i), cols = coarse element dofs, INSERT semantics (duplicates identical)
def build_prolongator(mc, mf, fe):
    geom = mc.geom
    nc = ndofs(geom, fe)
    EP = elem_prolongator(geom, fe)                         # [nchild, nc, nc]
    nchild = EP.shape[0]
    rows = elem_sys_dof(mf, fe)[mc.child_elem, :]            # [nel_c, nchild, nc]
    cols = elem_sys_dof(mc, fe)                              # [nel_c, nc]
    R = np.broadcast_to(rows[:, :, :, None], (mc.nel, nchild, nc, nc)).ravel()
    C = np.broadcast_to(cols[:, None, None, :], (mc.nel, nchild, nc, nc)).ravel()
    V = np.broadcast_to(EP[None], (mc.nel, nchild, nc, nc)).ravel()
    keep = V != 0.0
    R, C, V = R[keep], C[keep], V[keep]
    nf, ncc = n_dofs(mf, fe), n_dofs(mc, fe)
    key = R * np.int64(ncc) + C
    uniq, first = np.unique(key, return_index=True)          # INSERT: keep one copy
    P = sp.csr_matrix((V[first], (R[first], C[first])), shape=(nf, ncc))
    P.sort_indices()
    return P


# a13. boundary flags (MultiLevelSolution.cpp:725-840): nodes on faces with flag < -1 are Dirichlet
def dirichlet_dofs(mesh, fe):
    nc = ndofs(mesh.geom, fe)
    fn = face_nodes(mesh.geom)
    mark = np.zeros(n_dofs(mesh, fe), dtype=bool)
    for f, nodes in enumerate(fn):
        nodes = nodes[nodes < nc]
        els = np.where(mesh.face_flag[:, f] < -1)[0]
        mark[mesh.elem_dof[els][:, nodes].ravel()] = True
    return np.where(mark)[0]


# a11/a12. sparsity + add_matrix_blocked / add_vector_blocked in element order
def csr_pattern(mesh, fe):
    ed = elem_sys_dof(mesh, fe).astype(np.int64)
    nc = ed.shape[1]
    n = n_dofs(mesh, fe)
    key = np.unique((np.repeat(ed, nc, axis=1) * n + np.tile(ed, (1, nc))).ravel())
    rows = key // n
    indptr = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=n))])
    return indptr.astype(np.int32), (key % n).astype(np.int32)


def assemble_poisson(mesh, fe, rhs_vec, sol=None, order="seventh", chunk=4096):
    """KK->zero; RES->zero; element loop; add_*_blocked (sequential element order for every entry)."""
    et = ElemType(mesh.geom, fe, order)
    ed = elem_sys_dof(mesh, fe)
    nc = et.nc
    n = n_dofs(mesh, fe)
    indptr, indices = csr_pattern(mesh, fe)
    vals = np.zeros(indices.size)
    b = np.zeros(n)
    if sol is None:
        sol = np.zeros(n)
    # position of (row, col) in CSR
    for s in range(0, mesh.nel, chunk):
        e = slice(s, min(s + chunk, mesh.nel))
        X = np.transpose(mesh.coords[mesh.elem_dof[e]], (0, 2, 1))          # [ne, dim, nloc] biquadratic geometry
        K, F = elem_poisson_batch(et, X, sol[ed[e]], rhs_vec)
        rows = np.repeat(ed[e], nc, axis=1).ravel()
        cols = np.tile(ed[e], (1, nc)).ravel()
        pos = _csr_positions(indptr, indices, rows, cols)
        # np.add.at applies in index order = element order, local (i,j) order
        np.add.at(vals, pos, K.ravel())
        np.add.at(b, ed[e].ravel(), F.ravel())
    A = sp.csr_matrix((vals, indices, indptr), shape=(n, n))
    return A, b


def _csr_positions(indptr, indices, rows, cols):
    key = rows.astype(np.int64) * (indices.max() + 1 if indices.size else 1)
    # global sorted key array of the CSR (row-major, sorted columns) allows one searchsorted
    ncol = np.int64(indices.max() + 1)
    rowid = np.repeat(np.arange(indptr.size - 1, dtype=np.int64), np.diff(indptr))
    gkey = rowid * ncol + indices
    q = rows.astype(np.int64) * ncol + cols
    pos = np.searchsorted(gkey, q)
    assert np.all(gkey[pos] == q)
    return pos


# K5. MatZeroRows(KK, bdc, diag=1) keeping the pattern (LinearEquationSolverPetsc.cpp:428-436)
def zero_rows(A, rows, diag):
    A = A.tocsr(copy=True)
    for r in rows:
        A.data[A.indptr[r]:A.indptr[r + 1]] = 0.0
    if diag != 0.0:
        d = sp.csr_matrix((np.full(len(rows), diag), (rows, rows)), shape=A.shape)
        A = (A + d).tocsr()
    A.sort_indices()
    return A


def zero_rows_inplace_pattern(A, rows, diag):
    """same, but strictly keeps A's pattern (needs the diagonal to be in the pattern)."""
    A = A.tocsr(copy=True)
    A.sort_indices()
    for r in rows:
        s, e = A.indptr[r], A.indptr[r + 1]
        A.data[s:e] = 0.0
        k = np.searchsorted(A.indices[s:e], r)
        if diag != 0.0:
            assert A.indices[s + k] == r
            A.data[s + k] = diag
    return A


# LinearImplicitSystem::ZeroInterpolatorDirichletNodes (LinearImplicitSystem.cpp:1032-1120)
def zero_interpolator_dirichlet(P, bdc_f, bdc_c):
    P = P.tocsr(copy=True)
    mf = np.ones(P.shape[0]); mf[bdc_f] = 0.0
    mc = np.ones(P.shape[1]); mc[bdc_c] = 0.0
    P = sp.diags(mf) @ P @ sp.diags(mc)
    P = P.tocsr()
    P.sort_indices()
    return P


# ----------------------------------------------------------------------------------------------
# a15-a18. multigrid hierarchy + cycle.  The arithmetic of the production cycle lives in PETSc 3.20.2
# (contrib/scripts/install_petsc.sh:12; NOT in /root/reference): PCMG multiplicative V-cycle, level
# smoother KSPRICHARDSON(scale omega)+PCJACOBI with a fixed number of iterations, zero initial guess on
# the way down, coarse PREONLY+LU, restriction = P^T (LinearImplicitSystem.cpp:379-382).  The in-repo
# statement of the same cycle is LinearImplicitSystem::MGStep (LinearImplicitSystem.cpp:1397-1562).
# ----------------------------------------------------------------------------------------------
class Hierarchy:
    pass


def build_poisson_hierarchy(nx, ny, nz, nlevels, fe, rhs_vec, order="seventh", lo=(0., 0., 0.), hi=(1., 1., 1.)):
    """MGsolve preparation (LinearImplicitSystem.cpp:318-383): assemble finest KK/RES, Galerkin chain
    KK[l-1] = PP[l]^T KK[l] PP[l] from the un-penalised matrices, then SetPenalty on every level."""
    H = Hierarchy()
    H.meshes = build_levels(nx, ny, nz, nlevels, lo, hi)
    H.fe = fe
    H.bdc = [dirichlet_dofs(m, fe) for m in H.meshes]
    H.P = [None]
    for l in range(1, nlevels):
        P = build_prolongator(H.meshes[l - 1], H.meshes[l], fe)
        H.P.append(zero_interpolator_dirichlet(P, H.bdc[l], H.bdc[l - 1]))
    A, b = assemble_poisson(H.meshes[-1], fe, rhs_vec, order=order)
    H.A_raw = [None] * nlevels
    H.A_raw[-1] = A
    for l in range(nlevels - 1, 0, -1):
        H.A_raw[l - 1] = (H.P[l].T @ H.A_raw[l] @ H.P[l]).tocsr()
    H.A = []
    for l in range(nlevels):
        if l == nlevels - 1:
            H.A.append(zero_rows_inplace_pattern(H.A_raw[l], H.bdc[l], 1.0))
        else:
            H.A.append(zero_rows(H.A_raw[l], H.bdc[l], 1.0))
    H.b = b.copy()
    H.b[H.bdc[-1]] = 0.0          # ZerosBoundaryResiduals (LinearEquationSolverPetsc.cpp:417-424)
    H.b_raw = b
    return H


def jacobi_dinv(A):
    d = A.diagonal().copy()
    d[d == 0.0] = 1.0             # PCJACOBI: zero diagonal entries are replaced by 1
    return 1.0 / d


def smooth(A, dinv, b, x, omega, nsweeps, zero_guess):
    """Richardson(omega) + Jacobi, fixed iteration count: x <- x + omega D^-1 (b - A x)."""
    for it in range(nsweeps):
        if zero_guess and it == 0:
            x = omega * dinv * b
        else:
            x = x + omega * dinv * (b - A @ x)
    return x


def greedy_colors(A):
    """greedy colouring of the matrix graph in row order: coupled rows get different colours (coupling taken from the symmetrised
    pattern: i reading x_j separates them whether or not j reads x_i)"""
    A = A.tocsr()
    n = A.shape[0]
    A = A[:, :n].tocsr()
    S = sp.csr_matrix((np.ones(A.indices.size), A.indices, A.indptr), shape=A.shape)     # the stored pattern, explicit zeros included
    A = (S + S.T).tocsr()
    A.sort_indices()
    color = np.full(n, -1, dtype=np.int64)
    nc = 0
    for i in range(n):
        cols = A.indices[A.indptr[i]:A.indptr[i + 1]]
        used = set(color[cols[(cols != i) & (cols < n)]].tolist())
        c = 0
        while c in used:
            c += 1
        color[i] = c
        nc = max(nc, c + 1)
    return color, nc


def smooth_sor_color(A, dinv, b, x, omega, nsweeps, zero_guess, color, nc):
    """Richardson(omega) + one symmetric Gauss-Seidel sweep (PCSOR, omega_sor = 1) per iteration, colours in place of the
    natural order: x <- x + omega * B (b - A x); B: z = 0, forward colours 0..nc-1, backward nc-1..0 of
    z_i = dinv_i (r_i - sum_{j != i} a_ij z_j)."""
    A = A.tocsr()
    offd = A - sp.diags(A.diagonal())
    rows_of = [np.where(color == c)[0] for c in range(nc)]
    for it in range(nsweeps):
        r = b.copy() if (zero_guess and it == 0) else b - A @ x
        z = np.zeros_like(b)
        for order in (range(nc), range(nc - 1, -1, -1)):
            for c in order:
                rows = rows_of[c]
                z[rows] = dinv[rows] * (r[rows] - offd[rows] @ z)
        x = omega * z if (zero_guess and it == 0) else x + omega * z
    return x


def sor_symmetric_natural(A, dinv, r):
    """z = B r, B = PCSOR as the reference selects it (PetscPreconditioner.cpp:219-222): PETSc defaults omega_sor = 1, one LOCAL
    SYMMETRIC sweep from a zero guess, rows in their natural order (MatSOR, SOR_LOCAL_SYMMETRIC_SWEEP | SOR_ZERO_INITIAL_GUESS):
    forward z_i = (r_i - sum_{j<i} a_ij z_j) / a_ii for i = 0..n-1, then backward z_i = (r_i - sum_{j!=i} a_ij z_j) / a_ii for
    i = n-1..0 with the newest values.  Sequential by definition; written as two triangular solves."""
    import scipy.sparse.linalg as spla
    A = A.tocsr()
    n = A.shape[0]
    A = A[:, :n]                                   # local block (columns beyond n are ghosts)
    D = sp.diags(1.0 / dinv)
    L = sp.tril(A, -1).tocsr()
    U = sp.triu(A, 1).tocsr()
    z = spla.spsolve_triangular((L + D).tocsr(), r, lower=True)
    z = spla.spsolve_triangular((U + D).tocsr(), r - L @ z, lower=False)
    return z


def ilu0_factor(A, zeropivot=1e-16):
    """ILU(0) in natural order on A's pattern -- PCILU as the reference configures it (PetscPreconditioner.cpp:91-115;
    PCFactorSetZeroPivot(1e-16), MAT_SHIFT_NONZERO: LinearEquationSolverPetsc.cpp:444-446).  IKJ elimination restricted to the
    pattern; a pivot with |u_ii| <= zeropivot * sum_{j>i} |u_ij| restarts the factorisation of A + shift I with shift = 100 eps,
    doubled at every further restart (PETSc 3.20.2 MatPivotCheck_nz; not under /root/reference).  Returns (L unit lower, U, shift)."""
    A = A.tocsr().copy()
    n = A.shape[0]
    A = A[:, :n].tocsr()
    A.sort_indices()
    ip, ix = A.indptr, A.indices
    dpos = np.array([ip[i] + np.searchsorted(ix[ip[i]:ip[i + 1]], i) for i in range(n)])
    assert np.all(ix[dpos] == np.arange(n)), "ILU(0): a row without diagonal entry"
    shift = 0.0
    while True:
        v = A.data.copy()
        v[dpos] += shift
        ok = True
        for i in range(n):
            rs, re = ip[i], ip[i + 1]
            cols = ix[rs:re]
            for p in range(rs, re):
                k = ix[p]
                if k >= i:
                    break
                lik = v[p] / v[dpos[k]]
                v[p] = lik
                kc = ix[dpos[k] + 1:ip[k + 1]]
                pos = np.searchsorted(cols, kc)
                hit = (pos < cols.size) & (cols[np.minimum(pos, cols.size - 1)] == kc)
                v[rs + pos[hit]] -= lik * v[dpos[k] + 1:ip[k + 1]][hit]
            if not abs(v[dpos[i]]) > zeropivot * np.abs(v[dpos[i] + 1:re]).sum():    # sctx.rs = the U part without the diagonal
                ok = False
                break
        if ok:
            break
        shift = 100.0 * np.finfo(float).eps if shift == 0.0 else 2.0 * shift
    M = sp.csr_matrix((v, ix.copy(), ip.copy()), shape=(n, n))
    return (sp.tril(M, -1) + sp.identity(n)).tocsr(), sp.triu(M, 0).tocsr(), shift


def ilu0_apply(LU, r):
    import scipy.sparse.linalg as spla
    L, U, _ = LU
    return spla.spsolve_triangular(U, spla.spsolve_triangular(L, r, lower=True, unit_diagonal=True), lower=False)


def smooth_precond(A, b, x, omega, nsweeps, zero_guess, apply_B):
    """Richardson(omega) with a preconditioner application z = B r per iteration (KSPRICHARDSON, fixed iteration count)"""
    for it in range(nsweeps):
        first = zero_guess and it == 0
        z = apply_B(b.copy() if first else b - A @ x)
        x = omega * z if first else x + omega * z
    return x


def smooth_gmres(A, b, x, nits, zero_guess, apply_B, restart=30):
    """KSPGMRES as a level smoother (`SetSolverFineGrids(GMRES)`, LinearEquationSolverPetsc.cpp:501-502; the reference's default
    `_levelSolverType`): exactly `nits` iterations (PCMG skips the convergence test of its smoothers, Appendix A), left preconditioning
    (KSPGMRES default), classical Gram-Schmidt without refinement (PETSc's default orthogonalisation), restart 30.  Minimises
    ||B (b - A x)||_2 over x0 + K_nits(BA, B r0).  A lucky breakdown ends the cycle early.  Parity unpinned (PETSc is not in the image):
    the test anchors are the minimisation property and the agreement with a dense least-squares solve."""
    done = 0
    while done < nits:
        m = min(restart, nits - done)
        r = apply_B(b.copy() if (zero_guess and done == 0) else b - A @ x)
        beta = np.linalg.norm(r)
        if beta == 0.0:
            break
        V = [r / beta]
        Hm = np.zeros((m + 1, m))
        k_used = 0
        for k in range(m):
            w = apply_B(A @ V[k])
            h = np.array([w @ v for v in V])
            for hj, v in zip(h, V):
                w = w - hj * v
            Hm[:k + 1, k] = h
            Hm[k + 1, k] = np.linalg.norm(w)
            k_used = k + 1
            if Hm[k + 1, k] == 0.0:
                break
            V.append(w / Hm[k + 1, k])
        g = np.zeros(k_used + 1)
        g[0] = beta
        y = np.linalg.lstsq(Hm[:k_used + 1, :k_used], g, rcond=None)[0]
        dx = sum(yj * v for yj, v in zip(y, V))
        x = dx if (zero_guess and done == 0) else x + dx
        done += m
    return x


def vcycle(H, level, b, omega=2. / 3., npre=2, npost=2, coarse_solve=None, x=None, smoother="jacobi", level_solver="richardson"):
    """One multiplicative V-cycle applied to rhs b, starting from x (None = zero).  level_solver "gmres": the smoother's sweep
    preconditioner B (jacobi / sor / ilu0) inside `npre` / `npost` GMRES iterations instead of Richardson(omega)."""
    A = H.A[level]
    if level == 0:
        if coarse_solve is None:
            import scipy.sparse.linalg as spla
            if not hasattr(H, "_lu"):
                H._lu = spla.splu(H.A[0].tocsc())
            return H._lu.solve(b)
        return coarse_solve(b)
    if not hasattr(H, "_dinv"):
        H._dinv = [jacobi_dinv(a) for a in H.A]
    dinv = H._dinv[level]
    if smoother == "gs_color":
        if not hasattr(H, "_colors"):
            H._colors = [greedy_colors(a) for a in H.A]
        col, nc = H._colors[level]
        sm = lambda bb, xx, n, zg: smooth_sor_color(A, dinv, bb, xx, omega, n, zg, col, nc)
    elif smoother == "sor":
        sm = lambda bb, xx, n, zg: smooth_precond(A, bb, xx, omega, n, zg, lambda r: sor_symmetric_natural(A, dinv, r))
    elif smoother == "ilu0":
        if not hasattr(H, "_ilu"):
            H._ilu = {}
        if level not in H._ilu:
            H._ilu[level] = ilu0_factor(A)
        sm = lambda bb, xx, n, zg: smooth_precond(A, bb, xx, omega, n, zg, lambda r: ilu0_apply(H._ilu[level], r))
    elif smoother == "identity":          # PCNONE (IDENTITY_PRECOND, PetscPreconditioner.cpp:75-77)
        sm = lambda bb, xx, n, zg: smooth_precond(A, bb, xx, omega, n, zg, lambda r: r)
    else:
        sm = lambda bb, xx, n, zg: smooth(A, dinv, bb, xx, omega, n, zg)
    if level_solver == "gmres":
        if smoother == "sor":
            B = lambda r: sor_symmetric_natural(A, dinv, r)
        elif smoother == "ilu0":
            B = lambda r: ilu0_apply(H._ilu[level], r)
        elif smoother == "jacobi":
            B = lambda r: dinv * r
        elif smoother == "identity":
            B = lambda r: r
        else:
            raise ValueError("gmres level solver: preconditioner %s not restated" % smoother)
        sm = lambda bb, xx, n, zg: smooth_gmres(A, bb, xx, n, zg, B)
    x = sm(b, np.zeros_like(b) if x is None else x, npre, x is None)
    r = b - A @ x
    bc = H.P[level].T @ r
    ec = vcycle(H, level - 1, bc, omega, npre, npost, coarse_solve, smoother=smoother, level_solver=level_solver)
    x = x + H.P[level] @ ec
    x = sm(b, x, npost, False)
    return x


def pcmg_apply(H, b, kind="multiplicative", omega=2. / 3., npre=2, npost=2, coarse_solve=None, smoother="jacobi", level_solver="richardson"):
    """One application of PCMG to b for the four PCMGSetType values MGInit can select (LinearEquationSolverPetsc.cpp:199-214), restated from PETSc
    3.20's mg.c / fmg.c (parity unpinned -- PETSc is not in the image):
      multiplicative  PCMGMCycle_Private = vcycle above
      additive        PCMGACycle_Private: b_{l-1} = R b_l down all levels; every level x_l = smoothd(b_l) from zero (level 0: exact); x_l += P x_{l-1} upwards
      full            PCMGFCycle_Private: b restricted down; x_0 exact; for l = 1..top: x_l = P x_{l-1}, then one multiplicative cycle on level l from that guess
      kaskade         PCMGKCycle_Private: b restricted down; x_0 exact; for l = 1..top: x_l = P x_{l-1}, x_l = smoothd(b_l, x_l)"""
    top = len(H.A) - 1
    kw = dict(omega=omega, npre=npre, npost=npost, coarse_solve=coarse_solve, smoother=smoother, level_solver=level_solver)
    if kind == "multiplicative":
        return vcycle(H, top, b, **kw)
    bs = [None] * (top + 1)
    bs[top] = b
    for l in range(top, 0, -1):
        bs[l - 1] = H.P[l].T @ bs[l]
    exact = lambda rhs: vcycle(H, 0, rhs, **kw)
    # the down smoother alone = the first half of vcycle: npre iterations, no coarse correction, no post-smoothing
    def smoothd(l, rhs, x0):
        return exact(rhs) if l == 0 else _vcycle_presmooth_only(H, l, rhs, x0, kw)
    if kind == "additive":
        xs = [exact(bs[0])] + [smoothd(l, bs[l], None) for l in range(1, top + 1)]
        x = xs[0]
        for l in range(1, top + 1):
            x = xs[l] + H.P[l] @ x
        return x
    x = exact(bs[0])
    for l in range(1, top + 1):
        x = H.P[l] @ x
        if kind == "full":
            x = vcycle(H, l, bs[l], x=x, **kw)
        elif kind == "kaskade":
            x = smoothd(l, bs[l], x)
        else:
            raise ValueError(kind)
    return x


def _vcycle_presmooth_only(H, level, b, x0, kw):
    """npre iterations of the level's smoother from x0 (None = zero): vcycle with the coarse correction forced to zero on a one-level-deep view"""
    kw2 = dict(kw)
    kw2["npost"] = 0
    zero = lambda r: np.zeros_like(r)
    # levels below `level` only enter through the coarse solve: give the recursion a zero correction
    class View:
        pass
    V = View()
    V.A = [H.A[level - 1], H.A[level]]
    V.P = [None, H.P[level]]
    V._dinv = [H._dinv[level - 1], H._dinv[level]] if hasattr(H, "_dinv") else [jacobi_dinv(H.A[level - 1]), jacobi_dinv(H.A[level])]
    kw2["coarse_solve"] = zero
    return vcycle(V, 1, b, x=x0, **kw2)


def solve_richardson_mg(H, rtol=1e-10, maxit=100, **kw):
    """outer Richardson with the V-cycle as the iteration (stationary MG iteration)."""
    A, b = H.A[-1], H.b
    x = np.zeros_like(b)
    bn = np.linalg.norm(b)
    hist = []
    for it in range(maxit):
        r = b - A @ x
        rn = np.linalg.norm(r)
        hist.append(rn)
        if rn <= rtol * bn:
            break
        x = x + vcycle(H, len(H.A) - 1, r, **kw)
    return x, hist


def solve_pcg_mg(H, rtol=1e-10, maxit=100, **kw):
    """PCG preconditioned by one V-cycle on the symmetrised system (Dirichlet rows are identity)."""
    A, b = H.A[-1], H.b
    L = len(H.A) - 1
    x = np.zeros_like(b)
    r = b - A @ x
    z = vcycle(H, L, r, **kw)
    p = z.copy()
    rz = r @ z
    bn = np.linalg.norm(b)
    hist = [np.linalg.norm(r)]
    for it in range(maxit):
        if hist[-1] <= rtol * bn:
            break
        Ap = A @ p
        alpha = rz / (p @ Ap)
        x = x + alpha * p
        r = r - alpha * Ap
        hist.append(np.linalg.norm(r))
        z = vcycle(H, L, r, **kw)
        rz_new = r @ z
        p = z + (rz_new / rz) * p
        rz = rz_new
    return x, hist


def solve_gmres_mg(H, rtol=1e-10, maxit=100, restart=30, **kw):
    """left-preconditioned restarted GMRES, preconditioner = one V-cycle, zero initial guess
    (LinearEquationSolverPetsc.cpp:294-335; KSPGMRES default = classical Gram-Schmidt, left PC,
    convergence on the preconditioned residual norm)."""
    A, b = H.A[-1], H.b
    L = len(H.A) - 1
    M = lambda v: vcycle(H, L, v, **kw)
    x = np.zeros_like(b)
    hist = []
    its = 0
    beta0 = None
    while its < maxit:
        r = M(b - A @ x)
        beta = np.linalg.norm(r)
        if beta0 is None:
            beta0 = np.linalg.norm(M(b))
            hist.append(beta)
        if beta <= rtol * beta0:
            break
        V = [r / beta]
        Hm = np.zeros((restart + 1, restart))
        g = np.zeros(restart + 1)
        g[0] = beta
        cs, sn = np.zeros(restart), np.zeros(restart)
        k_used = 0
        for k in range(restart):
            w = M(A @ V[k])
            h = np.array([w @ v for v in V])       # classical Gram-Schmidt
            for hj, v in zip(h, V):
                w = w - hj * v
            Hm[:k + 1, k] = h
            Hm[k + 1, k] = np.linalg.norm(w)
            V.append(w / Hm[k + 1, k] if Hm[k + 1, k] != 0 else w)
            for j in range(k):
                t = cs[j] * Hm[j, k] + sn[j] * Hm[j + 1, k]
                Hm[j + 1, k] = -sn[j] * Hm[j, k] + cs[j] * Hm[j + 1, k]
                Hm[j, k] = t
            d = np.hypot(Hm[k, k], Hm[k + 1, k])
            cs[k], sn[k] = Hm[k, k] / d, Hm[k + 1, k] / d
            Hm[k, k] = d
            Hm[k + 1, k] = 0.0
            g[k + 1] = -sn[k] * g[k]
            g[k] = cs[k] * g[k]
            its += 1
            k_used = k + 1
            hist.append(abs(g[k + 1]))
            if abs(g[k + 1]) <= rtol * beta0 or its >= maxit:
                break
        y = np.linalg.solve(np.triu(Hm[:k_used, :k_used]), g[:k_used])
        for j in range(k_used):
            x = x + y[j] * V[j]
        if abs(g[k_used]) <= rtol * beta0:
            break
    return x, hist


def solve_fgmres_mg(H, rtol=1e-10, maxit=100, restart=30, knoll=True, **kw):
    """flexible restarted GMRES (KSPFGMRES, a case of LinearEquationSolverPetsc.cpp:506-507): right preconditioning, the preconditioned
    vectors z_k = M v_k are kept so that M (one multigrid cycle, possibly with GMRES level solvers) may differ from application to
    application; classical Gram-Schmidt; convergence on the true residual norm against ||b||; Knoll guess x0 = M b
    (KSPSetInitialGuessKnoll, :308).  PETSc 3.20.2 itself is not under /root/reference: parity unpinned."""
    A, b = H.A[-1], H.b
    L = len(H.A) - 1
    M = lambda v: vcycle(H, L, v, **kw)
    x = M(b) if knoll else np.zeros_like(b)
    bnorm = np.linalg.norm(b)
    hist = []
    its = 0
    while its < maxit:
        r = b - A @ x
        beta = np.linalg.norm(r)
        if not hist:
            hist.append(beta)
        if beta <= rtol * bnorm:
            break
        V, Z = [r / beta], []
        Hm = np.zeros((restart + 1, restart))
        g = np.zeros(restart + 1)
        g[0] = beta
        cs, sn = np.zeros(restart), np.zeros(restart)
        k_used = 0
        for k in range(restart):
            Z.append(M(V[k]))
            w = A @ Z[k]
            h = np.array([w @ v for v in V])       # classical Gram-Schmidt
            for hj, v in zip(h, V):
                w = w - hj * v
            Hm[:k + 1, k] = h
            Hm[k + 1, k] = np.linalg.norm(w)
            V.append(w / Hm[k + 1, k] if Hm[k + 1, k] != 0 else w)
            for j in range(k):
                t = cs[j] * Hm[j, k] + sn[j] * Hm[j + 1, k]
                Hm[j + 1, k] = -sn[j] * Hm[j, k] + cs[j] * Hm[j + 1, k]
                Hm[j, k] = t
            d = np.hypot(Hm[k, k], Hm[k + 1, k])
            cs[k], sn[k] = Hm[k, k] / d, Hm[k + 1, k] / d
            Hm[k, k] = d
            Hm[k + 1, k] = 0.0
            g[k + 1] = -sn[k] * g[k]
            g[k] = cs[k] * g[k]
            its += 1
            k_used = k + 1
            hist.append(abs(g[k + 1]))
            if abs(g[k + 1]) <= rtol * bnorm or its >= maxit:
                break
        y = np.linalg.solve(np.triu(Hm[:k_used, :k_used]), g[:k_used])
        for j in range(k_used):
            x = x + y[j] * Z[j]
        if abs(g[k_used]) <= rtol * bnorm:
            break
    return x, hist


# ----------------------------------------------------------------------------------------------
# deterministic fills used by tests / bench (LCG, seed 12345, uniform [-1,1])
# ----------------------------------------------------------------------------------------------
def system_offsets(dof_offset):
    """KKoffset / KKIndex of LinearEquation::InitPde (LinearEquation.cpp:212-237), loop for loop: dof_offset[k][p] is the reference's
    `_msh->_dofOffset[_SolType[_SolPdeIndex[k]]][p]`"""
    nvars, nprocs = len(dof_offset), len(dof_offset[0]) - 1
    KKIndex = [0] * (nvars + 1)
    for i in range(1, nvars + 1):
        KKIndex[i] = KKIndex[i - 1] + dof_offset[i - 1][nprocs]
    KK = [[0] * nprocs for _ in range(nvars + 1)]
    KK[0][0] = 0
    for j in range(1, nvars + 1):
        KK[j][0] = KK[j - 1][0] + (dof_offset[j - 1][1] - dof_offset[j - 1][0])
    for i in range(1, nprocs):
        KK[0][i] = KK[nvars][i - 1]
        for j in range(1, nvars + 1):
            KK[j][i] = KK[j - 1][i] + (dof_offset[j - 1][i + 1] - dof_offset[j - 1][i])
    return KK, KKIndex


def find_processor_of_dof(offsets, dof, iproc=0):
    """Mesh::BisectionSearch_find_processor_of_dof (Mesh.cpp:1004-1018)"""
    nprocs = len(offsets) - 1
    d0, d1, d = 0, nprocs, iproc
    while dof < offsets[d] or dof >= offsets[d + 1]:
        if dof < offsets[d]:
            d1 = d
        else:
            d0 = d + 1
        d = (d0 + d1) // 2
    return d


def system_dof(dof_offset, KK, var, idof, iproc=0):
    """LinearEquation::GetSystemDof (LinearEquation.cpp:76-85) for the mesh dof `idof` of variable `var`"""
    p = find_processor_of_dof(dof_offset[var], idof, iproc)
    return KK[var][p] + idof - dof_offset[var][p], p


def lcg_fill(n, seed=12345):
    s = np.uint64(seed)
    out = np.empty(n)
    a, c = np.uint64(6364136223846793005), np.uint64(1442695040888963407)
    # vectorised LCG via jump-ahead is overkill; n is small in tests
    state = int(seed)
    mask = (1 << 64) - 1
    for i in range(n):
        state = (state * 6364136223846793005 + 1442695040888963407) & mask
        out[i] = ((state >> 11) / float(1 << 53)) * 2.0 - 1.0
    return out


def algorithmic_spmv_bytes(nnz, nrows, ncols):
    """fp64 values + int32 columns + int32 row pointers, x and y touched once."""
    return 12 * nnz + 4 * (nrows + 1) + 8 * ncols + 8 * nrows


# ----------------------------------------------------------------------------------------------
# a5. elem_type::JacobianSur (ElemType.hpp:1089-1138 edges in 2-D, :1330-1380 quad faces in 3-D) and the Neumann
#     boundary term of applications/001_Poisson/main.cpp:560-594:  F[ilocal] += phi_i * tau * weight
# ----------------------------------------------------------------------------------------------
def face_local_nodes(geom, fe, face, table=None):
    """element-local nodes of `face` in the face element's own node order.  table: optional explicit order
    (e.g. the reference's hex_lag::faceDofs row); default = parametrisation by the free coordinates in cyclic order."""
    if table is not None:
        nfn = {("hex", "linear"): 4, ("hex", "serendipity"): 8, ("hex", "biquadratic"): 9, ("quad", "linear"): 2, ("quad", "serendipity"): 3, ("quad", "biquadratic"): 3}[(geom, fe)]
        return np.asarray(table[:nfn], dtype=np.int64)
    Xc = xc_table(geom)
    d = Xc.shape[1]
    centre = (20 + face) if geom == "hex" else (4 + face)
    d0 = int(np.nonzero(Xc[centre])[0][0])
    sgn = Xc[centre, d0]
    # orientation: the JacobianSur normal of the node order points out of the element, as with the reference's faceDofs tables (the golden
    # fixture facedofs_hex / facedofs_quad holds those; same cyclic order, possibly another starting node)
    if geom == "hex":
        ref = XC_QUAD9[:4] if fe == "linear" else XC_QUAD9[:8] if fe == "serendipity" else XC_QUAD9
        a, b = (d0 + 1) % 3, (d0 + 2) % 3
        if sgn < 0:
            a, b = b, a
        return np.array([np.where((Xc[:, d0] == sgn) & (Xc[:, a] == r[0]) & (Xc[:, b] == r[1]))[0][0] for r in ref])
    ref = np.array([-1.0, 1.0]) if fe == "linear" else np.array([-1.0, 1.0, 0.0])
    a = (d0 + 1) % 2
    if (sgn < 0) if d0 == 0 else (sgn > 0):
        ref = -ref
    return np.array([np.where((Xc[:, d0] == sgn) & (Xc[:, a] == r))[0][0] for r in ref])


def jacobian_sur(geom, fe, order, vt, ig):
    """vt[dim][nfn] face node coordinates; returns Weight, phi[nfn], normal[dim] (reference operation order)."""
    if geom == "hex":
        w, xg = gauss_table("quad", order)
        phi, dphi, _ = eval_basis("quad", fe, xg[ig:ig + 1])
        nfn = phi.shape[1]
        J = np.zeros((3, 3))
        for n in range(nfn):
            for d in range(3):
                J[d, 0] += dphi[0, n, 0] * vt[d][n]
                J[d, 1] += dphi[0, n, 1] * vt[d][n]
        nx = J[1, 0] * J[2, 1] - J[1, 1] * J[2, 0]
        ny = J[0, 1] * J[2, 0] - J[2, 1] * J[0, 0]
        nz = J[0, 0] * J[1, 1] - J[0, 1] * J[1, 0]
        inv = 1.0 / np.sqrt(nx * nx + ny * ny + nz * nz)
        normal = np.array([nx * inv, ny * inv, nz * inv])
        J[:, 2] = normal
        det = (J[0, 0] * (J[1, 1] * J[2, 2] - J[1, 2] * J[2, 1]) + J[0, 1] * (J[1, 2] * J[2, 0] - J[1, 0] * J[2, 2]) +
               J[0, 2] * (J[1, 0] * J[2, 1] - J[1, 1] * J[2, 0]))
        return det * w[ig], phi[0], normal
    w, xg = gauss_table("line", order)
    x = xg[ig, 0]
    if fe == "linear":
        phi = np.array([0.5 * (1. - x), 0.5 * (1. + x)])
        dphi = np.array([-0.5, 0.5])
    else:
        phi = np.array([lag_biquadratic(x, 0), lag_biquadratic(x, 2), lag_biquadratic(x, 1)])
        dphi = np.array([dlag_biquadratic(x, 0), dlag_biquadratic(x, 2), dlag_biquadratic(x, 1)])
    j0 = sum(dphi[n] * vt[0][n] for n in range(phi.size))
    j1 = sum(dphi[n] * vt[1][n] for n in range(phi.size))
    modn = np.sqrt(j0 * j0 + j1 * j1)
    normal = np.array([j1 / modn, -j0 / modn])
    det = j0 * (-normal[1]) - (-normal[0]) * j1
    return det * w[ig], phi, normal


def neumann_rhs(mesh, fe, flux_by_flag, order="seventh", face_tables=None):
    """sum over boundary faces whose flag is in flux_by_flag of  int phi_i tau ds, scattered to the nodes; tau: a number
    (applications/001_Poisson/main.cpp:556-594) or a function of the Gauss point xyzt = (sum_i x_i phi_i, t = 0) evaluated inside the
    Gauss loop as the parsed-function branch does (`(*bdcfunc)(&xyzt[0])`, main.cpp:521-537)"""
    geom = mesh.geom
    out = np.zeros(n_dofs(mesh, fe))
    ng = gauss_table("quad" if geom == "hex" else "line", order)[0].size
    for f in range(mesh.face_flag.shape[1]):
        loc = face_local_nodes(geom, fe, f, None if face_tables is None else face_tables[f])
        for flag, tau in flux_by_flag.items():
            for iel in np.where(mesh.face_flag[:, f] == flag)[0]:
                nodes = mesh.elem_dof[iel, loc]
                vt = [mesh.coords[nodes, d] for d in range(mesh.dim)]
                for ig in range(ng):
                    weight, phi, _ = jacobian_sur(geom, fe, order, vt, ig)
                    tv = tau
                    if callable(tau):
                        xyzt = np.zeros(4)
                        for d in range(mesh.dim):
                            xyzt[d] = float(np.dot(vt[d], phi))
                        tv = tau(xyzt)
                    for i in range(nodes.size):
                        out[nodes[i]] += phi[i] * tv * weight
    return out
