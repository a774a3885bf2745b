"""GPU parity of the callback the application ships -- applications/003_NavierStokes/SteadyNavierStokesParallel/main.cpp:390-925: equal-order LAGRANGE FIRST
velocity / pressure with the Franca-Frey stabilisation, the matrix = minus the derivative of the residual (adept's tape in the reference, written out
in k_ns_stab_elem) -- against the oracle's statement-by-statement restatement (oracle/femus_oracle_ns.py, Jacobian by complex-step differentiation,
checked against central differences and plain loops in tests/test_ns_host.py), and the application's run with its own settings (:96-185)."""
import numpy as np
import pytest
import scipy.sparse.linalg as spla

from femus_amd import capi
from oracle import femus_oracle as fo
from oracle import femus_oracle_ns as ns

pytestmark = pytest.mark.gpu
LO, HI = (-0.5, -0.5, 0.0), (0.5, 0.5, 0.5)


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def _setup(ctx, box, distort, seed=3):
    mo = fo.build_levels(*box, 1, LO, HI)[0]
    mh = capi.Mesh.box(*box, LO, HI)
    rng = np.random.default_rng(seed)
    if distort:
        mo.coords = mo.coords + distort * rng.standard_normal(mo.coords.shape)
        mh.set_coords(mo.coords)
    lay = ns.NSLayoutEqualOrder(mo)
    nd, off, es = capi.system_elem_dofs(mh, ["linear"] * (mo.dim + 1))
    assert nd == lay.nd and np.array_equal(es, lay.elem_sys)
    rp, col = capi.pattern_from_elements(es, lay.n)
    return mo, mh, lay, rp, col, rng


@pytest.mark.parametrize("box", [(4, 3, 0), (2, 2, 2)])
@pytest.mark.parametrize("ire", [1.0, 1e-2, 1e-4])
def test_element_matrices_and_assembly_match_the_oracle(ctx, box, ire):
    """distorted QUAD / HEX elements (the Hessians of the bilinear shape functions do not vanish), a non-trivial state, IRe from Stokes-like to
    Re = 10 000 (both branches of the stabilisation parameter): element matrix and residual to 1e-12 per element, the assembled operator likewise"""
    mo, mh, lay, rp, col, rng = _setup(ctx, box, 0.02)
    A = ctx.matrix_csr(lay.n, lay.n, rp, col)
    asm = capi.NSStabAssembler(ctx, mh, A)
    u = 0.5 * rng.standard_normal(lay.n)
    sol = ctx.vector_from(u)
    K, F = asm.element_matrices(sol, ire)
    et = fo.ElemType(mo.geom, "linear")
    X = np.transpose(mo.coords[mo.elem_dof], (0, 2, 1))
    loc = u[lay.elem_sys]
    Ko, Fo = ns.elem_ns_stab_batch(et, X, loc[:, :lay.dim * lay.nv].reshape(mo.nel, lay.dim, lay.nv), loc[:, lay.dim * lay.nv:], ire)
    kscale = np.abs(Ko).max(axis=(1, 2), keepdims=True)
    assert (np.abs(K - Ko) / kscale).max() <= 1e-12
    assert (np.abs(F - Fo) / np.abs(Fo).max(axis=1, keepdims=True)).max() <= 1e-12
    res = ctx.vector(lay.n)
    asm.assemble(A, res, sol, ire)
    Ao, bo = ns.assemble_ns_stab(mo, lay, u, ire)
    assert abs(A.to_scipy() - Ao).max() <= 1e-12 * abs(Ao).max() and rel(res.to_numpy(), bo) < 1e-12
    # zero state: the branch Rek <= 1e-15 (delta = 0, no derivative of the parameters)
    K0, F0 = asm.element_matrices(None, ire)
    Ko0, Fo0 = ns.elem_ns_stab_batch(et, X, np.zeros((mo.nel, lay.dim, lay.nv)), np.zeros((mo.nel, lay.nv)), ire)
    assert (np.abs(K0 - Ko0) / np.abs(Ko0).max(axis=(1, 2), keepdims=True)).max() <= 1e-12 and abs(F0).max() == 0.0 and abs(Fo0).max() == 0.0
    asm.destroy(), A.destroy(), mh.destroy()


def _cavity_bc_equal_order(mesh, lay):
    """SetBoundaryConditionCavityFlow (main.cpp:365-384) on the vertex nodes: U = 0 and V = 0 on every wall, V = 1 on face 1 for -0.5 < y < 0.5,
    P pinned at the (-0.5, -0.5) corner"""
    fn = fo.face_nodes(mesh.geom)
    nq1 = lay.sizes[0]
    wall = np.zeros(mesh.nnode, dtype=bool)
    face1 = np.zeros(mesh.nnode, dtype=bool)
    for f, nodes in enumerate(fn):
        els = np.where(mesh.face_flag[:, f] < -1)[0]
        wall[mesh.elem_dof[els][:, nodes].ravel()] = True
        els = np.where(mesh.face_flag[:, f] == -5)[0]              # the wall x = -0.5 (as oracle.cavity_bc: the generated box names its faces otherwise than box10x10.neu)
        face1[mesh.elem_dof[els][:, nodes].ravel()] = True
    nodes = np.where(wall[:nq1])[0]
    y = mesh.coords[nodes, 1]
    vval = np.where(face1[nodes] & (y < 0.5) & (y > -0.5), 1.0, 0.0)
    corner = nodes[(mesh.coords[nodes, 0] < -0.5 + 1e-8) & (mesh.coords[nodes, 1] < -0.5 + 1e-8)]
    bdc = np.concatenate([nodes + lay.offset[0], nodes + lay.offset[1], corner + lay.offset[2]])
    val = np.concatenate([np.zeros(nodes.size), vval, np.zeros(corner.size)])
    o = np.argsort(bdc)
    return bdc[o], val[o]


def _application_newton(ctx, nx, with_oracle):
    """main.cpp:63-185: ONE level (the coarse levels are erased: every linear solve is the exact one), the callback's Reynolds continuation, Newton until
    max_k ||Eps_k|| / ||Sol_k|| < 1e-10 at the final Reynolds number (max 90 iterations)"""
    mo = fo.build_levels(nx, nx, 0, 1, LO, HI)[0]
    mh = capi.Mesh.box(nx, nx, 0, LO, HI)
    lay = ns.NSLayoutEqualOrder(mo)
    nd, off, es = capi.system_elem_dofs(mh, ["linear"] * 3)
    rp, col = capi.pattern_from_elements(es, lay.n)
    bdc, val = _cavity_bc_equal_order(mo, lay)
    A = ctx.matrix_csr(lay.n, lay.n, rp, col)
    asm = capi.NSStabAssembler(ctx, mh, A)
    xy = np.concatenate([mo.coords[:sz, :2] for sz in lay.sizes])
    direct = capi.Direct(ctx, A, xy)
    bidx = capi.Index(ctx, bdc)
    sol_h = np.zeros(lay.n)
    sol_h[bdc] = val
    sol, res, eps = ctx.vector_from(sol_h), ctx.vector(lay.n), ctx.vector(lay.n)
    u_o = sol_h.copy()
    hist_d, hist_o = [], []
    converged = False
    norm_k = lambda e, s: max(np.linalg.norm(e[lay.offset[k]:lay.offset[k + 1]]) / np.linalg.norm(s[lay.offset[k]:lay.offset[k + 1]]) for k in range(2))
    for c in range(90):
        ire = ns.reynolds_of_call(c)
        asm.assemble(A, res, sol, ire)
        bidx.zero_rows(A, 1.0)
        bidx.set(res, 0.0)
        direct.factor()
        direct.solve(res, eps)
        sol.add(1.0, eps)
        hist_d.append(norm_k(eps.to_numpy(), sol.to_numpy()))
        if with_oracle:
            Ao, bo = ns.assemble_ns_stab(mo, lay, u_o, ire, pattern=(rp, col))
            Ao = fo.zero_rows(Ao, bdc, 1.0)
            bo[bdc] = 0.0
            d_o = spla.splu(Ao.tocsc()).solve(bo)
            u_o = u_o + d_o
            hist_o.append(norm_k(d_o, u_o))
        if hist_d[-1] < 1e-10 and ire == 1e-4:
            converged = True
            break
    out = (converged, hist_d, hist_o, sol.to_numpy(), u_o, lay, direct.stats())
    direct.destroy(), asm.destroy(), A.destroy(), mh.destroy()
    return out


def test_the_application_newton_history_matches_the_oracle(ctx):
    """the application's loop on a 20 x 20 mesh, device against oracle (scipy's LU) step by step: same Newton history through the whole continuation
    Re = 1, 12, 63, ... 10 000, same solution"""
    converged, hist_d, hist_o, u_d, u_o, lay, st = _application_newton(ctx, 20, True)
    assert converged and st["general"] and len(hist_d) == len(hist_o) <= 25, hist_d
    for a, b in zip(hist_d[:-1], hist_o[:-1]):
        assert abs(a - b) <= 1e-6 * b, (hist_d, hist_o)
    assert hist_o[-1] < 1e-10 and rel(u_d, u_o) < 1e-9


def test_the_application_run_with_its_own_settings(ctx):
    """SteadyNavierStokesParallel at its own size: box10x10 refined to 80 x 80, one level (19 683 unknowns solved exactly per Newton step by the pivoted
    fronts of fh_direct), Q1/Q1 + Franca-Frey, continuation to Re = 10 000: converges well inside the application's 90 iterations, quadratically at the end"""
    converged, hist_d, _, u, _, lay, st = _application_newton(ctx, 80, False)
    assert converged and len(hist_d) <= 30 and st["general"], hist_d
    assert hist_d[-1] < 1e-10 and hist_d[-2] < 1e-4 and hist_d[-3] < 1e-2          # the tail of a Newton iteration with the exact derivative
    v = u[lay.offset[1]:lay.offset[2]]
    assert v.max() == 1.0 and v.min() < -0.1                                       # the moving wall drives a vortex
