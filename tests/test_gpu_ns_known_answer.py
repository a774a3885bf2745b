"""The reference's known-answer test (unittests/testNSSteadyDD/main.cpp, see tests/test_ns_known_answer.py for what it stores and why level 3 is
untouched by the adaptive levels) THROUGH THE DEVICE PATH: Gambit reader -> three refinements ON THE DEVICE -> pattern of the Q2 / discontinuous
piecewise-linear system -> Navier-Stokes assembly kernel (fh_ns_pw_assembler_create) -> Dirichlet rows -> sparse exact solve (general fronts: the
Jacobian has an empty pressure block) -> Newton.  The norms of U, V, P the reference asserts to 1e-6 must come out to 1e-8; and the element
matrices of the kernel against the oracle restatement of the callback at a random state, 1e-12."""
import os

import numpy as np
import pytest

from femus_amd import capi
from oracle import femus_oracle as fo
from oracle import femus_oracle_ns as fns

from test_ns_known_answer import CYLINDER, INFLOW, STORED, WALL, inflow_profile, nodes_on

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def hierarchy(ctx, nref):
    m = capi.Mesh.read_gambit(os.path.join(HERE, "golden", "nsbenc.neu"))
    for _ in range(nref):
        m = m.refine(ctx)
    return m


@pytest.mark.parametrize("nref", [0, 1])
def test_element_matrices_with_the_piecewise_linear_pressure_match_the_oracle(ctx, nref):
    m = hierarchy(ctx, nref)
    ed, xy, ff = m.arrays()
    mo = fo.Mesh("quad", ed, xy, ff, level=nref)
    lay = fns.NSLayoutPwLinear(mo)
    es = capi.NSPwAssembler.elem_sys(m)
    assert np.array_equal(es, lay.elem_sys)
    assert np.array_equal(capi.system_elem_dofs(m, ["biquadratic", "biquadratic", "pwlinear"])[2], lay.elem_sys)      # the library's GetSystemDof for solution type 4
    KK = ctx.matrix_from_elements(es, lay.n)
    asm = capi.NSPwAssembler(ctx, m, KK)
    rng = np.random.default_rng(11)
    x = rng.uniform(-1, 1, lay.n)
    sol, res = ctx.vector_from(x), ctx.vector(lay.n)
    K, F = asm.element_matrices(sol, 0.001)
    loc = x[lay.elem_sys]
    X = np.transpose(mo.coords[mo.elem_dof], (0, 2, 1))
    Jo, Ro = fns.elem_ns_batch(fo.ElemType("quad", "biquadratic", "seventh"), fns.PwLinearPressure("quad", "seventh"), X,
                               loc[:, :18].reshape(mo.nel, 2, 9), loc[:, 18:], 0.001)
    assert abs(K - Jo).max() <= 1e-12 * abs(Jo).max()
    assert abs(F - Ro).max() <= 1e-12 * max(abs(Ro).max(), 1.0)
    # and assembled: the CSR operator and residual against the oracle's
    asm.assemble(KK, res, sol, 0.001)
    Ao, bo = fns.assemble_ns(mo, lay, x, 0.001, etp=fns.PwLinearPressure("quad", "seventh"))
    assert abs(KK.to_scipy() - Ao).max() <= 1e-12 * abs(Ao).max()
    assert np.linalg.norm(res.to_numpy() - bo) <= 1e-12 * np.linalg.norm(bo)
    asm.destroy(); KK.destroy()


def test_level3_norms_of_the_reference_known_answer_test_on_the_device(ctx):
    m = hierarchy(ctx, 3)
    assert m.nel == 98 * 64
    ed, xy, ff = m.arrays()
    mo = fo.Mesh("quad", ed, xy, ff, level=3)                       # node sets of the boundary only (host logic of the test)
    nq2 = m.nnode
    n = 2 * nq2 + 3 * m.nel
    dn = np.unique(np.concatenate([nodes_on(mo, INFLOW), nodes_on(mo, WALL), nodes_on(mo, CYLINDER)]))
    inflow = nodes_on(mo, INFLOW)
    bdc = np.concatenate([dn, dn + nq2]).astype(np.int32)
    x0 = np.zeros(n)
    x0[:nq2] = inflow_profile(xy[:, 1])
    x0[dn] = 0.0
    x0[inflow] = inflow_profile(xy[inflow, 1])
    KK = ctx.matrix_from_elements(capi.NSPwAssembler.elem_sys(m), n)
    asm = capi.NSPwAssembler(ctx, m, KK)
    sol, res, eps = ctx.vector_from(x0), ctx.vector(n), ctx.vector(n)
    bidx = capi.Index(ctx, bdc)
    # where the unknowns lie (nested dissection of the fronts): nodes for the velocities, the element centre for its three pressure functions
    cen = xy[ed[:, 8]]
    d = capi.Direct(ctx, KK, np.concatenate([xy, xy, cen, cen, cen]))
    hist = []
    for it in range(12):
        asm.assemble(KK, res, sol, 0.001)
        bidx.zero_rows(KK, 1.0)                                     # SetPenalty on the Dirichlet rows, their residual entries to zero
        bidx.set(res, 0.0)
        d.factor()
        d.solve(res, eps)
        sol.add(1.0, eps)
        e, s = eps.to_numpy(), sol.to_numpy()
        hist.append(max(np.linalg.norm(e[a:b]) / np.linalg.norm(s[a:b]) for a, b in ((0, nq2), (nq2, 2 * nq2), (2 * nq2, n))))
        if hist[-1] < 1e-12:
            break
    assert hist[-1] < 1e-12, hist
    st = d.stats()
    assert st["general"]                                            # the pivoted fronts served it
    s = sol.to_numpy()
    got = {"U": np.linalg.norm(s[:nq2]), "V": np.linalg.norm(s[nq2:2 * nq2]), "P": np.linalg.norm(s[2 * nq2:])}
    rel = {k: abs(got[k] - STORED[k]) / STORED[k] for k in got}
    print("device path, level-3 norms", got, "relative distance to the stored numbers", rel, "Newton updates", hist, st)
    assert max(rel.values()) < 1e-8, rel
    d.destroy(); asm.destroy(); KK.destroy()


def _bc(x, name, face):          # main.cpp:290-392: faces 1 inflow, 2 outflow (nothing prescribed), 3 walls, 4 cylinder
    if face == 2:
        return False, 0.0
    return True, (inflow_profile(x[1]) if (name == "U" and face == 1) else 0.0)


def test_known_answer_under_the_iteration_limits_of_the_reference_test(ctx):
    """the same run stopped where the reference's test stops it (main.cpp:139-153): at most THREE Newton steps per level, ended by a relative update below
    1e-4; per step TWO linear cycles of at most four outer GMRES iterations (SetMaxNumberOfLinearIterations(2), SetTolerances(..., 4)); ONE GMRES + ILU(0)
    iteration before and after the coarse correction (SetNumberPre/PostSmoothingStep(1)).  The norms it leaves on level 3 pass the reference's own 1e-6 by
    four orders of magnitude -- they are, if anything, closer to the stored numbers than the fully converged ones (the stored numbers came from such a run)."""
    from femus_amd.navier_stokes import NavierStokesPwMG
    ms = [capi.Mesh.read_gambit(os.path.join(HERE, "golden", "nsbenc.neu"))]
    for _ in range(3):
        ms.append(ms[-1].refine(ctx))
    pb = NavierStokesPwMG(ctx, ms, 0.001, _bc, level_gmres_its=1).init()
    x = np.zeros(pb.n[0])
    x[:ms[0].nnode] = inflow_profile(ms[0].arrays()[1][:, 1])
    pb.set_state(0, x)
    pb.mgsolve(tol=1e-4, max_newton=3, lin_rtol=1e-12, lin_maxit=8, restart=4)
    for ig in range(1, 4):
        last = [h for h in pb.history if h[0] == ig][-1]
        assert last[2] < 1e-4 and last[3] <= 8                     # every level above the coarsest ended by the tolerance, inside the limits
    s = pb.SOL[3].to_numpy()
    nq = ms[3].nnode
    got = {"U": np.linalg.norm(s[:nq]), "V": np.linalg.norm(s[nq:2 * nq]), "P": np.linalg.norm(s[2 * nq:])}
    rel = {k: abs(got[k] - STORED[k]) / STORED[k] for k in got}
    print("reference limits, level-3 norms", got, rel, pb.history)
    assert max(rel.values()) < 1e-8, rel
    pb.destroy()


def test_known_answer_through_the_multigrid_path(ctx):
    """the same numbers from the path the reference's test itself takes (main.cpp:139-160): nonlinear F-cycle over levels 0 .. 3, per Newton step the
    Galerkin chain, an exact solve on level 0 and GMRES + ILU(0) level solvers above (SetSolverFineGrids(GMRES), SetPreconditionerFineGrids(ILU_PRECOND),
    four iterations per smoothing step: SetTolerances(..., 4)), one pre- and one post-smoothing step, preconditioning an outer flexible GMRES.  The element-owned
    pressures travel between the levels with the element prolongator of solution type 4 (ElemType.cpp:446-520)."""
    from femus_amd.navier_stokes import NavierStokesPwMG
    ms = [capi.Mesh.read_gambit(os.path.join(HERE, "golden", "nsbenc.neu"))]
    for _ in range(3):
        ms.append(ms[-1].refine(ctx))
    pb = NavierStokesPwMG(ctx, ms, 0.001, _bc, level_gmres_its=4).init()
    x = np.zeros(pb.n[0])
    x[:ms[0].nnode] = inflow_profile(ms[0].arrays()[1][:, 1])          # Initialize("U", InitVariableU) on the coarsest level; the F-cycle prolongs from there
    pb.set_state(0, x)
    assert pb.mgsolve(tol=1e-10, max_newton=20, lin_rtol=1e-10, lin_maxit=100)
    # what the reference's own limits (three Newton steps per level, stop at a relative update of 1e-4) would have kept: every level needed at most that
    for ig in range(1, 4):
        steps = [h for h in pb.history if h[0] == ig]
        assert [h[2] for h in steps][min(2, len(steps) - 1)] < 1e-4
        assert max(h[3] for h in steps) <= 30                                  # outer iterations of a linear solve
    s = pb.SOL[3].to_numpy()
    nq = ms[3].nnode
    got = {"U": np.linalg.norm(s[:nq]), "V": np.linalg.norm(s[nq:2 * nq]), "P": np.linalg.norm(s[2 * nq:])}
    rel = {k: abs(got[k] - STORED[k]) / STORED[k] for k in got}
    print("multigrid path, level-3 norms", got, rel, pb.history)
    assert max(rel.values()) < 1e-8, rel
    pb.destroy()


@pytest.mark.parametrize("box", [(3, 2, 0), (2, 2, 2)])
def test_pressure_prolongator_reproduces_a_linear_pressure(ctx, box):
    """solution type 4 between two levels of an affine box mesh (the block fh_build_system_prolongator makes for fe = 4): p = a + b x + c y (+ d z) on the coarse
    elements arrives as the same function on their children; the system dof map puts function i of element e at i * nel + e"""
    dim = 2 if box[2] == 0 else 3
    mc = capi.Mesh.box(*box, lo=(0., 0., 0.), hi=(3., 1., 2.))
    mf = mc.refine(ctx)
    nd, off, es = capi.system_elem_dofs(mf, ["pwlinear"])
    assert nd == dim + 1 and off[-1] == (dim + 1) * mf.nel
    assert np.array_equal(es, np.arange(dim + 1)[None, :] * mf.nel + np.arange(mf.nel)[:, None])
    Pm = capi.build_system_prolongator(ctx, mc, mf, ["pwlinear"])
    P = Pm.to_scipy()
    Pm.destroy()
    if dim == 3:
        def coefficients3(m, f):
            ed, xy, _ = m.arrays()
            xc = xy[ed[:, 26]]
            h = [0.5 * (xy[ed[:, v], d] - xy[ed[:, 0], d]) for d, v in enumerate((1, 3, 4))]
            return np.concatenate([f[0] + xc @ np.array(f[1:])] + [f[1 + d] * h[d] for d in range(3)])
        f3 = (0.3, -1.7, 2.2, 0.9)
        assert np.allclose(P @ coefficients3(mc, f3), coefficients3(mf, f3), rtol=0, atol=1e-14)
        return

    def coefficients(m, f):
        ed, xy, _ = m.arrays()
        xc = xy[ed[:, 8]]                                          # element centres; half sizes from the vertices (affine elements)
        hx = 0.5 * (xy[ed[:, 1], 0] - xy[ed[:, 0], 0])
        hy = 0.5 * (xy[ed[:, 3], 1] - xy[ed[:, 0], 1])
        a, b, c = f
        return np.concatenate([a + b * xc[:, 0] + c * xc[:, 1], b * hx, c * hy])      # value at the centre, slope per reference unit

    f = (0.3, -1.7, 2.2)
    assert np.allclose(P @ coefficients(mc, f), coefficients(mf, f), rtol=0, atol=1e-14)


def test_piecewise_linear_pressure_in_three_dimensions(ctx):
    """the HEX27 form of the same assembler (hexpwLinear: 1, xi, eta, zeta; nd = 85) on a distorted box against the oracle restatement, and one Newton step's
    exact solve through the pivoted fronts against scipy"""
    import scipy.sparse.linalg as spla
    m = capi.Mesh.box(2, 2, 2)
    _, xy0, _ = m.arrays()
    m.set_coords(xy0 + 0.03 * np.random.default_rng(4).uniform(-1, 1, xy0.shape))
    m = m.refine(ctx)
    ed, xy, ff = m.arrays()
    mo = fo.Mesh("hex", ed, xy, ff, level=1)
    lay = fns.NSLayoutPwLinear(mo)
    assert lay.nd == 85
    es = capi.system_elem_dofs(m, ["biquadratic"] * 3 + ["pwlinear"])[2]
    assert np.array_equal(es, lay.elem_sys)
    KK = ctx.matrix_from_elements(es, lay.n)
    asm = capi.NSPwAssembler(ctx, m, KK)
    x = np.random.default_rng(12).uniform(-1, 1, lay.n)
    sol, res, eps = ctx.vector_from(x), ctx.vector(lay.n), ctx.vector(lay.n)
    etp = fns.PwLinearPressure("hex", "seventh")
    asm.assemble(KK, res, sol, 0.05)
    Ao, bo = fns.assemble_ns(mo, lay, x, 0.05, etp=etp)
    assert abs(KK.to_scipy() - Ao).max() <= 1e-12 * abs(Ao).max()
    assert np.linalg.norm(res.to_numpy() - bo) <= 1e-12 * np.linalg.norm(bo)
    # all velocities fixed on the boundary, the constant pressure mode removed by fixing one pressure function
    nq2 = m.nnode
    wall = fo.dirichlet_dofs(mo, "biquadratic")
    bdc = np.concatenate([wall, wall + nq2, wall + 2 * nq2, [3 * nq2]]).astype(np.int32)
    bidx = capi.Index(ctx, bdc)
    bidx.zero_rows(KK, 1.0)
    bidx.set(res, 0.0)
    d = capi.Direct(ctx, KK, None).factor()
    d.solve(res, eps)
    A, b = KK.to_scipy(), res.to_numpy()
    ref = spla.splu(A.tocsc()).solve(b)
    assert np.linalg.norm(eps.to_numpy() - ref) <= 1e-10 * np.linalg.norm(ref)
    assert d.stats()["perturbed_pivots"] == 0
    d.destroy(); asm.destroy(); KK.destroy()


def test_the_whole_hierarchy_of_the_reference_test_with_its_two_selective_levels(ctx):
    """all six levels of unittests/testNSSteadyDD (main.cpp:55-82: four uniform levels, then two made by SetRefinementFlag :262-280 -- Gambit group 5, the
    elements around the cylinder): the nonlinear F-cycle runs through the non-homogeneous levels too (velocities hanging at the interfaces tied to their masters by
    PPamr, NonLinearImplicitSystem.cpp:213-236; the element-owned pressures need nothing).  The reference stores no number above level 3: checked are the level-3
    norms (untouched by what follows), convergence on every level, the hanging-node relation of the final velocities and the residual of the oracle's
    operator on the finest level."""
    import scipy.sparse as sp
    from femus_amd.navier_stokes import NavierStokesPwMG
    ms = [capi.Mesh.read_gambit(os.path.join(HERE, "golden", "nsbenc.neu"))]
    for l in range(1, 6):
        g, _ = ms[-1].elem_groups()
        lev, _ = ms[-1].elem_levels()
        flags = None if l < 4 else ((g == 5) & (lev == ms[-1].level)).astype(np.uint8)
        ms.append(ms[-1].refine_device(ctx, flags))
    assert [m.nel for m in ms] == [98, 392, 1568, 6272, 10112, 25472]
    pb = NavierStokesPwMG(ctx, ms, 0.001, _bc, level_gmres_its=4).init()
    x = np.zeros(pb.n[0])
    x[:ms[0].nnode] = inflow_profile(ms[0].arrays()[1][:, 1])
    pb.set_state(0, x)
    assert pb.mgsolve(tol=1e-10, max_newton=20, lin_rtol=1e-10, lin_maxit=150)
    s3 = pb.SOL[3].to_numpy()
    nq = ms[3].nnode
    got = {"U": np.linalg.norm(s3[:nq]), "V": np.linalg.norm(s3[nq:2 * nq]), "P": np.linalg.norm(s3[2 * nq:])}
    assert max(abs(got[k] - STORED[k]) / STORED[k] for k in got) < 1e-8
    # finest level: hanging velocities = interpolation of their masters; the oracle's discrete residual vanishes at the free unknowns after the projection
    top = 5
    m = ms[top]
    s = pb.SOL[top].to_numpy()
    hang, ptr, master, w = m.amr_constraints("biquadratic")
    assert hang.size > 0
    for k in range(2):
        u = s[k * m.nnode:(k + 1) * m.nnode]
        interp = np.array([np.dot(w[ptr[i]:ptr[i + 1]], u[master[ptr[i]:ptr[i + 1]]]) for i in range(hang.size)])
        assert np.abs(u[hang] - interp).max() <= 1e-10 * np.abs(u).max()
    ed, xy, ff = m.arrays()
    mo = fo.Mesh("quad", ed, xy, ff, level=top)
    lay = fns.NSLayoutPwLinear(mo)
    _, b = fns.assemble_ns(mo, lay, s, 0.001, etp=fns.PwLinearPressure("quad", "seventh"))
    Pamr = pb.Pamr[top].to_scipy()
    r = Pamr.T @ b
    free = np.setdiff1d(np.arange(lay.n), pb.bdc[top])
    assert np.abs(r[free]).max() <= 1e-9 * np.abs(b).max()
    print("six levels:", [m.nel for m in ms], "unknowns on the finest", lay.n, "newton / linear history", [(h[0], h[1], float("%.1e" % h[2]), h[3]) for h in pb.history])
    pb.destroy()


def _t_bc(mo):
    """main.cpp:375-391: T = 1 on the inflow, 5 on the cylinder, nothing prescribed on the walls and the outflow"""
    inflow, cyl = nodes_on(mo, INFLOW), nodes_on(mo, CYLINDER)
    idx = np.concatenate([inflow, cyl])
    val = np.concatenate([np.ones(inflow.size), 5.0 * np.ones(cyl.size)])
    o = np.argsort(idx)
    return idx[o].astype(np.int32), val[o]


def test_temperature_system_of_the_reference_test(ctx):
    """the second system of unittests/testNSSteadyDD (AssembleMatrixResT, main.cpp:730-880: LAGRANGE SECOND advection-diffusion in the computed velocity field,
    IPe = 1 / 1000 for Fluid(par, 0.001, 1, "Newtonian", 0.001, 1.): Prandtl 1) on level 3: element matrices and the assembled operator against the oracle
    restatement at a random state, then the solve -- exact, and by the cycle the test configures (V-cycle, GMRES + ILU(0) level solvers :186-200) -- against scipy
    on the oracle's operator.  The reference stores only the boundary values of T on this level (its V-cycle runs on the finest level, :186); here the field itself
    is checked."""
    import scipy.sparse.linalg as spla
    from femus_amd import known_answer as ka
    ms = [capi.Mesh.read_gambit(os.path.join(HERE, "golden", "nsbenc.neu"))]
    for _ in range(3):
        ms.append(ms[-1].refine(ctx))
    m = ms[3]
    ed, xy, ff = m.arrays()
    mo = fo.Mesh("quad", ed, xy, ff, level=3)
    nn = m.nnode
    KK = ctx.matrix_from_mesh(m, "biquadratic")
    asm = capi.AdvDiffAssembler(ctx, m, KK)
    rng = np.random.default_rng(21)
    t0, v0 = rng.uniform(-1, 1, nn), rng.uniform(-1, 1, 2 * nn)
    T, V, res = ctx.vector_from(t0), ctx.vector_from(v0), ctx.vector(nn)
    asm.assemble(KK, res, T, V, 1e-3)
    Ao, bo = fns.assemble_advdiff(mo, t0, v0, 1e-3)
    assert abs(KK.to_scipy() - Ao).max() <= 1e-12 * abs(Ao).max()
    assert np.linalg.norm(res.to_numpy() - bo) <= 1e-12 * np.linalg.norm(bo)
    # the velocity field of the flow (level 3, exact Newton as in the known-answer run), then T from its boundary values
    pbn = NavierStokesPwMG_single(ctx, m)
    vel = pbn.SOL[0]
    idx, val = _t_bc(mo)
    t_init = np.zeros(nn)
    t_init[idx] = val
    T.upload(t_init)
    asm.assemble(KK, res, T, vel, 1e-3)
    Ao, bo = fns.assemble_advdiff(mo, t_init, vel.to_numpy(), 1e-3)
    free = np.setdiff1d(np.arange(nn), idx)
    d_ref = np.zeros(nn)
    d_ref[free] = spla.splu(Ao.tocsr()[free][:, free].tocsc()).solve(bo[free])
    t_ref = t_init + d_ref
    bidx = capi.Index(ctx, idx)
    bidx.zero_rows(KK, 1.0)
    bidx.set(res, 0.0)
    eps = ctx.vector(nn)
    d = capi.Direct(ctx, KK, xy).factor()
    d.solve(res, eps)
    assert np.linalg.norm(eps.to_numpy() - d_ref) <= 1e-10 * np.linalg.norm(d_ref)
    d.destroy()
    # the cycle of the test: Galerkin chain over levels 0 .. 3, exact solve below, GMRES + ILU(0) level solvers, one smoothing step before and after
    bdc = []
    for l, ml in enumerate(ms):
        e_l, x_l, f_l = ml.arrays()
        bdc.append(_t_bc(fo.Mesh("quad", e_l, x_l, f_l, level=l))[0])
    A = {3: KK}
    Ps = {}
    for l in range(3, 0, -1):
        P = capi.build_prolongator(ctx, ms[l - 1], ms[l], "biquadratic", zero_bdc=False)
        P.mat_zero_rows(bdc[l], 0.0)
        P.zero_cols(bdc[l - 1])
        Ps[l] = P
        A[l - 1] = capi.Mat.ptap(P, A[l])
        bdc_dev = capi.Index(ctx, bdc[l - 1])
        bdc_dev.zero_rows(A[l - 1], 1.0)
    mg = capi.Multigrid(ctx, 4)
    for l in range(1, 4):
        mg.set_level_solver(l, "gmres", 30)
    for l in range(4):
        mg.set_level(l, A[l], Ps.get(l), None, capi.SMOOTH_ILU0, 1.0, 4 if l else 1, 4 if l else 0)
    mg.setup()
    its, rn = mg.solve(res, eps, outer="fgmres", rtol=1e-12, maxit=60)
    assert its <= 30
    assert np.linalg.norm(eps.to_numpy() - d_ref) <= 1e-9 * np.linalg.norm(d_ref)
    t_dev = t_init + eps.to_numpy()
    assert t_dev.min() > 1.0 - 0.35 and t_dev.max() < 5.0 + 0.35             # no spurious extrema beyond the usual Galerkin wiggles at Pe = 1000
    print("temperature on level 3: outer iterations", its, "range", t_dev.min(), t_dev.max(), "||T||", np.linalg.norm(t_dev), "distance to scipy on the oracle's operator",
          np.linalg.norm(t_dev - t_ref) / np.linalg.norm(t_ref))
    mg.destroy(); pbn.destroy(); asm.destroy()


def NavierStokesPwMG_single(ctx, m):
    """the converged flow on one level (exact Newton), as femus_amd.known_answer.run does it"""
    from femus_amd import known_answer as ka
    from femus_amd.navier_stokes import NavierStokesPwMG
    pb = NavierStokesPwMG(ctx, [m], 0.001, ka.boundary_condition).init()
    x = np.zeros(pb.n[0])
    x[:m.nnode] = ka.inflow_profile(m.arrays()[1][:, 1])
    pb.set_state(0, x)
    assert pb.mgsolve(tol=1e-10, max_newton=20)
    pb.meshes = []            # the mesh belongs to the caller
    return pb


def test_the_reference_unit_test_as_an_application(ctx):
    """femus_amd/app_ns_steady_dd.py = unittests/testNSSteadyDD/main.cpp over the C-ABI, all of it: six levels (two selective), the Navier-Stokes system under
    the test's own iteration limits, the temperature system on the finest level, and the test's four assertions (main.cpp:202-244)"""
    from femus_amd import app_ns_steady_dd as app
    r = app.run(ctx, verbose=False)
    assert r["passed"] and r["elements"] == [98, 392, 1568, 6272, 10112, 25472]
    assert max(r["relative_distance"].values()) < 1e-8, r["relative_distance"]
    lo, hi = r["temperature_range"]
    assert 0.9 < lo <= 1.0 + 1e-9 and 5.0 - 1e-9 <= hi < 5.1


def test_three_dimensional_flow_with_a_selective_level(ctx):
    """HEX27 / hexpwLinear through the same driver: lid-driven box, two uniform levels and one selective level above them (an octant refined), nonlinear F-cycle
    with GMRES + ILU(0) level solvers; the final velocities satisfy the hanging-node relation and the oracle's discrete residual vanishes"""
    from femus_amd.navier_stokes import NavierStokesPwMG
    ms = [capi.Mesh.box(2, 2, 2, lo=(0., 0., 0.), hi=(1., 1., 1.))]
    ms.append(ms[-1].refine(ctx))
    xc = ms[-1].elem_centroids()
    ms.append(ms[-1].refine_device(ctx, ((xc[:, 0] > 0.5) & (xc[:, 1] > 0.5) & (xc[:, 2] > 0.5)).astype(np.uint8)))
    assert not ms[2].elem_levels()[1]

    def bc(x, name, face):                       # every face a wall; the top (z = 1, face name 6 of the box generator) moves in x; the pressure level is fixed below
        return True, (1.0 if (name == "U" and face == 6 and 0.0 < x[0] < 1.0 and 0.0 < x[1] < 1.0) else 0.0)

    pb = NavierStokesPwMG(ctx, ms, 0.05, bc, level_gmres_its=4).init()
    # all-Dirichlet velocities leave the pressure defined up to a constant: fix the constant function of element 0 on every level (FixSolutionAtOnePoint)
    for l in range(3):
        p0 = np.int32(3 * ms[l].nnode)
        pb.bdc[l] = np.sort(np.append(pb.bdc[l], p0)).astype(np.int32)
        pb.bdc_val[l] = np.zeros(pb.bdc[l].size)
    for l in range(1, 3):                        # the cycle's interpolation must not touch the fixed pressure function
        pb.P[l].mat_zero_rows(np.array([3 * ms[l].nnode], np.int32), 0.0)
        pb.P[l].zero_cols(np.array([3 * ms[l - 1].nnode], np.int32))
    pb.set_state(0, np.zeros(pb.n[0]))
    assert pb.mgsolve(tol=1e-10, max_newton=20, lin_rtol=1e-10, lin_maxit=150)
    top = 2
    m = ms[top]
    s = pb.SOL[top].to_numpy()
    hang, ptr, master, w = m.amr_constraints("biquadratic")
    assert hang.size > 0
    for k in range(3):
        u = s[k * m.nnode:(k + 1) * m.nnode]
        interp = np.array([np.dot(w[ptr[i]:ptr[i + 1]], u[master[ptr[i]:ptr[i + 1]]]) for i in range(hang.size)])
        assert np.abs(u[hang] - interp).max() <= 1e-10 * max(np.abs(u).max(), 1e-30)
    ed, xy, ff = m.arrays()
    mo = fo.Mesh("hex", ed, xy, ff, level=top)
    lay = fns.NSLayoutPwLinear(mo)
    _, b = fns.assemble_ns(mo, lay, s, 0.05, etp=fns.PwLinearPressure("hex", "seventh"))
    r = pb.Pamr[top].to_scipy().T @ b
    free = np.setdiff1d(np.arange(lay.n), pb.bdc[top])
    assert np.abs(r[free]).max() <= 1e-9 * np.abs(b).max()
    # the lid drives a recirculation (hanging nodes ON the lid interpolate its discontinuous data: up to 1.125^2 there)
    assert 1.0 <= np.abs(s[:m.nnode]).max() <= 1.125 ** 2 + 1e-12 and s[:m.nnode].min() < -0.01
    pb.destroy()
