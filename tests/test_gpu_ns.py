"""GPU parity for the Navier-Stokes Newton / multigrid path (BASELINE config "003_NavierStokes lid-driven
cavity, Q2/Q1 Taylor-Hood, Newton + GMG-preconditioned GMRES") through the C-ABI against the oracle."""
import os

import numpy as np
import pytest
import scipy.sparse.linalg as spla

from femus_amd import capi
from femus_amd.navier_stokes import NavierStokesMG
from oracle import femus_oracle as fo
from oracle import femus_oracle_ns as ns

pytestmark = pytest.mark.gpu
LO, HI = (-0.5, -0.5, -0.5), (0.5, 0.5, 0.5)


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.mark.parametrize("box", [(3, 2, 0), (2, 1, 2)])
def test_element_jacobian_and_assembly_match_oracle(ctx, box):
    mo = fo.build_levels(*box, 1, LO, HI)[0]
    mh = capi.Mesh.box(*box, LO, HI)
    rng = np.random.default_rng(3)
    mo.coords = mo.coords + 0.02 * rng.standard_normal(mo.coords.shape)     # distorted elements: full Jacobian path
    mh.set_coords(mo.coords)
    lay = ns.NSLayout(mo)
    fes = ["biquadratic"] * mo.dim + ["linear"]
    nd, off, es = capi.system_elem_dofs(mh, fes)
    rp, col = capi.pattern_from_elements(es, lay.n)
    ipo, ico = ns.csr_pattern_sys(lay)
    assert np.array_equal(rp, ipo) and np.array_equal(col, ico)              # sparsity: integer, identical
    A = ctx.matrix_csr(lay.n, lay.n, rp, col)
    asm = capi.NSAssembler(ctx, mh, A)
    u = 0.5 * rng.standard_normal(lay.n)
    sol = ctx.vector_from(u)
    nu = 0.03
    K, F = asm.element_matrices(sol, nu)
    etv, etp = fo.ElemType(mo.geom, "biquadratic"), fo.ElemType(mo.geom, "linear")
    X = np.transpose(mo.coords[mo.elem_dof], (0, 2, 1))
    loc = u[lay.elem_sys]
    Ko, Fo = ns.elem_ns_batch(etv, etp, X, loc[:, :lay.dim * lay.nv].reshape(mo.nel, lay.dim, lay.nv), loc[:, lay.dim * lay.nv:], nu)
    assert abs(K - Ko).max() <= 1e-12 * abs(Ko).max()                        # fp64, summation order / FMA contraction only
    assert abs(F - Fo).max() <= 1e-12 * abs(Fo).max()
    res = ctx.vector(lay.n)
    asm.assemble(A, res, sol, nu)
    Ao, bo = ns.assemble_ns(mo, lay, u, nu)
    assert abs(A.to_scipy() - Ao).max() <= 1e-12 * abs(Ao).max()
    assert rel(res.to_numpy(), bo) < 1e-12
    asm.destroy(), A.destroy(), mh.destroy()


@pytest.mark.parametrize("persistent", [0, 1, 2])
def test_vanka_vcycle_matches_oracle(ctx, persistent):
    """one multiplicative V(2,2) cycle with the block Schwarz smoother on the Jacobian of a non-trivial state; the sweep as one residual + one
    patch launch per colour (default) and as one launch with device-wide barriers between the colours (two barrier forms)"""
    ctx.set_option("vanka_persistent", persistent)
    ctx.set_option("patch_invert_lds", 0 if persistent == 2 else 1)      # patch inverses by the workgroup kernel on global memory as well
    nu, nl = 0.01, 3
    pb = NavierStokesMG(ctx, 4, 4, 0, nl, nu).init()
    ms, lays = ns.build_ns_levels(4, 4, 0, nl, LO, HI)
    bcs = [ns.cavity_bc(m, l) for m, l in zip(ms, lays)]
    rng = np.random.default_rng(5)
    top = nl - 1
    state = 0.3 * rng.standard_normal(lays[top].n)
    state[bcs[top][0]] = bcs[top][1]
    pb.SOL[top].upload(state)
    mg = pb.prepare(top)
    H = ns.newton_step_operators(ms, lays, bcs, top, state, nu, omega=pb.omega, npre=pb.npre, npost=pb.npost)
    for l in range(nl):
        assert np.array_equal(pb.bdc[l], bcs[l][0])
        assert abs(pb.A[(top, l)].to_scipy() - H.A[l]).max() <= 1e-11 * abs(H.A[l]).max()
    assert rel(pb.RES[top].to_numpy(), H.b) < 1e-12
    b = rng.standard_normal(lays[top].n)
    b[bcs[top][0]] = 0.0
    x = ctx.vector(lays[top].n)
    mg.vcycle(ctx.vector_from(b), x)
    ref = ns.vcycle(H, top, b)
    assert rel(x.to_numpy(), ref) < 1e-9
    pb.destroy()
    ctx.set_option("vanka_persistent", 0)
    ctx.set_option("patch_invert_lds", 1)


def test_cavity_newton_fcycle_matches_oracle(ctx):
    """NonLinearImplicitSystem::MGsolve, F-cycle, Re = 100: the same Newton history as the oracle with exact linear solves and
    the same discrete solution to 1e-8 (the FP-solve parity bar of north_star, 1e-10, holds for the linear solves inside)"""
    nu, nl = 0.01, 3
    pb = NavierStokesMG(ctx, 4, 4, 0, nl, nu).init()
    assert pb.mgsolve(tol=1e-10, lin_rtol=1e-11)
    _, lays, sols, hist = ns.solve_cavity(4, 4, nl, nu, LO, HI, linear="direct")
    assert [h[:2] for h in pb.history] == [h[:2] for h in hist]               # same number of Newton steps on every level
    for hg, ho in zip(pb.history, hist):
        assert abs(hg[2] - ho[2]) <= 1e-6 * max(ho[2], 1e-9) + 1e-11
    for l in range(nl):
        assert rel(pb.SOL[l].to_numpy(), sols[l]) < 1e-8
    assert max(h[3] for h in pb.history) <= 30
    # the converged state satisfies the discrete equations: residual of the free rows vanishes
    top = nl - 1
    pb.prepare(top)
    assert pb.RES[top].l2_norm() < 1e-9
    pb.destroy()


def test_cavity_reynolds_1000_by_continuation(ctx):
    """BASELINE config 4 (viscosity 0.001): Newton from rest diverges at Re = 1000 on the 10 x 10 coarse grid, so the coarse
    level is walked down in viscosity and the F-cycle then runs at the target value; checks the primary-vortex strength
    against the published benchmark range (Ghia et al. 1982: minimum of the wall-parallel velocity on the centreline -0.38)"""
    nl = 3
    pb = NavierStokesMG(ctx, 10, 10, 0, nl, 0.01).init()
    for nu in (0.01, 0.004, 0.002, 0.001):
        pb.nu = nu
        assert pb.newton(0, tol=1e-10, max_newton=25)
    for ig in range(1, nl):
        pb.prolongator_sol(ig)
        assert pb.newton(ig, tol=1e-9, max_newton=25, lin_rtol=1e-10, lin_maxit=200)
    _, xy, _ = pb.meshes[-1].arrays()
    sol = pb.SOL[-1].to_numpy()
    off = pb.offsets[-1]
    line = np.where(abs(xy[:, 1]) < 1e-12)[0]                       # horizontal centreline y = 0; moving wall is x = -0.5
    vmin = sol[off[1] + line].min()
    assert -0.41 < vmin < -0.36
    pb.destroy()


def test_config4_at_its_stated_parameters_by_continuation(ctx):
    """BASELINE config 4 exactly as stated: lid-driven cavity, Q2/Q1 Taylor-Hood, 10 x 10 coarse QUAD9 mesh, FOUR levels (80 x 80 elements,
    58 403 unknowns), viscosity 0.001 (Re = 1000), Newton + multigrid-preconditioned GMRES on one GPU.  Newton from rest diverges at this
    viscosity on the coarse grid, so the coarse level is walked down in viscosity first (0.01 -> 0.004 -> 0.002 -> 0.001); the F-cycle
    then runs at the target value on every level.  Checks: every Newton solve converges, the discrete residual of the free rows vanishes
    on the finest level, and the primary vortex has the published strength (Ghia et al. 1982: minimum of the wall-parallel velocity on
    the centreline -0.38 at Re = 1000)."""
    nl = 4
    pb = NavierStokesMG(ctx, 10, 10, 0, nl, 0.01).init()
    for nu in (0.01, 0.004, 0.002, 0.001):
        pb.nu = nu
        assert pb.newton(0, tol=1e-10, max_newton=25)
    for ig in range(1, nl):
        pb.prolongator_sol(ig)
        assert pb.newton(ig, tol=1e-9, max_newton=25, lin_rtol=1e-10, lin_maxit=200)
    assert pb.n[-1] == 58403
    top = nl - 1
    pb.prepare(top)
    assert pb.RES[top].l2_norm() < 1e-8
    _, xy, _ = pb.meshes[-1].arrays()
    sol = pb.SOL[-1].to_numpy()
    off = pb.offsets[-1]
    line = np.where(abs(xy[:, 1]) < 1e-12)[0]
    vmin = sol[off[1] + line].min()
    assert -0.40 < vmin < -0.37
    pb.destroy()


def test_config4_with_the_coarse_levels_erased_from_the_cycles(ctx):
    """the cycles of config 4 stopped at the 40 x 40 level (coarse_level 2: 14 803 unknowns solved exactly on the pivoted fronts, the reference's
    EraseCoarseLevels) instead of running three more levels of colour steps: the same nonlinear F-cycle -- same Newton steps per level, same
    flow -- with fewer GMRES iterations per Newton step (profiles/r06_ns_probe.json: 89 -> 24 ms per linear solve)"""
    nl = 4
    runs = {}
    for c in (0, 2):
        pb = NavierStokesMG(ctx, 10, 10, 0, nl, 0.01)
        pb.coarse_level = c
        pb.init()
        for nu in (0.01, 0.004, 0.002, 0.001):
            pb.nu = nu
            assert pb.newton(0, tol=1e-10, max_newton=25)
        for ig in range(1, nl):
            pb.prolongator_sol(ig)
            assert pb.newton(ig, tol=1e-9, max_newton=25, lin_rtol=1e-10, lin_maxit=200)
        runs[c] = ([sum(1 for h in pb.history if h[0] == l) for l in range(nl)], [h[3] for h in pb.history if h[0] == nl - 1], pb.SOL[-1].to_numpy().copy())
        pb.prepare(nl - 1)
        assert pb.RES[nl - 1].l2_norm() < 1e-8
        pb.destroy()
    assert runs[0][0] == runs[2][0]                                          # Newton steps per level
    assert max(runs[2][1]) < min(runs[0][1])                                 # GMRES iterations per Newton step on the finest level
    assert np.linalg.norm(runs[0][2] - runs[2][2]) <= 1e-8 * np.linalg.norm(runs[0][2])


def test_three_dimensional_cavity_matches_oracle(ctx):
    """HEX27 Taylor-Hood (89 x 89 element Jacobians, Vanka patches of up to 376 dofs): lid on the z = hi face moving in x,
    two levels, Newton + multigrid GMRES against the oracle's Newton with exact linear solves"""
    import scipy.sparse.linalg as spla
    nu, nl = 0.05, 2
    lo, hi = np.array(LO), np.array(HI)

    def bc(x, name, face):
        if name == "P":
            return bool(np.all(x < lo + 1e-8)), 0.0
        return True, (1.0 if (name == "U" and face == 6 and lo[0] < x[0] < hi[0]) else 0.0)

    pb = NavierStokesMG(ctx, 2, 2, 2, nl, nu, boundary_condition=bc).init()
    assert pb.mgsolve(tol=1e-10, lin_rtol=1e-11, lin_maxit=100)
    ms = fo.build_levels(2, 2, 2, nl, LO, HI)
    lays = [ns.NSLayout(m) for m in ms]
    bcs = [ns.cavity_bc(m, l, lid_flag=-7, lid_component=0) for m, l in zip(ms, lays)]
    for l in range(nl):
        assert np.array_equal(pb.bdc[l], bcs[l][0]) and np.array_equal(pb.bdc_val[l], bcs[l][1])
    sols = [np.zeros(l.n) for l in lays]
    for l in range(nl):
        sols[l][bcs[l][0]] = bcs[l][1]
    for ig in range(nl):
        for it in range(20):
            A, b = ns.assemble_ns(ms[ig], lays[ig], sols[ig], nu)
            A = fo.zero_rows(A, bcs[ig][0], 1.0)
            b[bcs[ig][0]] = 0.0
            eps = spla.spsolve(A.tocsc(), b)
            sols[ig] = sols[ig] + eps
            if np.linalg.norm(eps) < 1e-11 * np.linalg.norm(sols[ig]):
                break
        if ig + 1 < nl:
            sols[ig + 1] = ns.block_prolongator(ms[ig], ms[ig + 1], lays[ig], lays[ig + 1]) @ sols[ig]
    assert rel(pb.SOL[-1].to_numpy(), sols[-1]) < 1e-8
    pb.destroy()


@pytest.mark.parametrize("level_solver,n_exact", [("richardson", 0), ("gmres", 0), ("richardson", 5)])
def test_pcasm_as_the_reference_configures_it(ctx, level_solver, n_exact):
    """FH_SMOOTH_ASM: PC_ASM_BASIC + PC_COMPOSITE_MULTIPLICATIVE over the element blocks in their index order with ILU(0) (zero pivot 1e-16,
    MAT_SHIFT_NONZERO) sub-solves (PetscPreconditioner.cpp:179-184, LinearEquationSolverPetscAsm.cpp:278-335): one V(2,2) cycle on the
    Jacobian of a non-trivial state against the oracle's sequential restatement, Richardson and GMRES level solvers; n_exact: the leading blocks
    with the EXACT sub-solve the reference gives the solid / porous blocks (`_blockTypeRange[1]`, :298-307)"""
    nu, nl = 0.01, 3
    pb = NavierStokesMG(ctx, 4, 4, 0, nl, nu).init()
    pb.smoother = capi.SMOOTH_ASM
    ms, lays = ns.build_ns_levels(4, 4, 0, nl, LO, HI)
    bcs = [ns.cavity_bc(m, l) for m, l in zip(ms, lays)]
    rng = np.random.default_rng(5)
    top = nl - 1
    state = 0.3 * rng.standard_normal(lays[top].n)
    state[bcs[top][0]] = bcs[top][1]
    pb.SOL[top].upload(state)
    mg = pb.prepare(top)
    if level_solver == "gmres" or n_exact:
        for l in range(1, nl):
            if level_solver == "gmres":
                mg.set_level_solver(l, "gmres", 30)
            mg.set_level_patches_exact(l, n_exact)
        mg.setup()
    # ILU(0) fills the ALLOCATED pattern (stored zeros included): the pattern is taken from the device operators, the values are the oracle's
    H = ns.newton_step_operators(ms, lays, bcs, top, state, nu, omega=pb.omega, npre=pb.npre, npost=pb.npost, smoother="asm",
                                 patterns=[pb.A[(top, l)].pattern() for l in range(nl)], asm_exact=n_exact)
    b = rng.standard_normal(lays[top].n)
    b[bcs[top][0]] = 0.0
    x = ctx.vector(lays[top].n)
    mg.vcycle(ctx.vector_from(b), x)
    if level_solver == "richardson":
        ref = ns.vcycle(H, top, b)
    else:
        def cyc(level, rhs):
            if level == 0:
                return H.coarse_solve(rhs)
            A, sm = H.A[level], H.smoother[level]
            xx = fo.smooth_gmres(A, rhs, np.zeros_like(rhs), H.npre, True, sm.apply)
            xx = xx + H.P[level] @ cyc(level - 1, H.P[level].T @ (rhs - A @ xx))
            return fo.smooth_gmres(A, rhs, xx, H.npost, False, sm.apply)
        ref = cyc(top, b)
    assert rel(x.to_numpy(), ref) < 1e-9
    pb.destroy()


def _channel_bc(inlet, outlet, p_in, p_out):
    """walls no-slip; U free on the inlet / outlet faces (V, W Dirichlet 0 there); the callback's value for "P" is the pressure of the open face"""
    def bc(x, name, face):
        if name == "P":
            return False, (p_in(x) if face == inlet else p_out(x) if face == outlet else 0.0)
        if name == "U":
            return face not in (inlet, outlet), 0.0
        return True, 0.0
    return bc


@pytest.mark.parametrize("box", [(3, 2, 0), (2, 2, 2)])
def test_open_boundary_pressure_integral_matches_oracle(ctx, box):
    """03_navier_stokes.hpp:185-290 (the boundary integral of the prescribed pressure on faces whose normal velocity is free): face selection by the
    bdc callback at the face centre + the Gauss-point-0 normal, then phi tau n weight at the face Gauss points -- device kernel against the
    oracle's loop-for-loop restatement on curved faces, the pressure a number on one face and a parsed function of the Gauss point on the other"""
    from femus_amd.navier_stokes import open_boundary_faces
    dim = 2 if box[2] == 0 else 3
    mo = fo.build_levels(*box, 2, LO, HI)[-1]
    mh = capi.Mesh.box(*box, LO, HI).refine()
    rng = np.random.default_rng(8)
    x0 = mo.coords.copy()
    mo.coords = mo.coords + 0.01 * rng.standard_normal(mo.coords.shape)
    on_side = np.isclose(abs(x0), 0.5)
    mo.coords[on_side] = x0[on_side]        # boundary nodes move inside their side only: the sides stay planar (the reference's choice of the normal
    mh.set_coords(mo.coords)                # velocity component needs axis-parallel faces, :257-262), the face maps do not stay affine
    lay = ns.NSLayout(mo)
    inlet, outlet = (4, 2) if dim == 2 else (5, 3)          # x = lo / x = hi face names of the box generator
    f_out = (lambda x: 0.3 + x[1] * x[1] - 0.5 * x[0]) if dim == 2 else (lambda x: 0.3 + x[1] * x[2] - 0.5 * x[0])
    bc = _channel_bc(inlet, outlet, lambda x: 1.25, f_out)
    names = ["U", "V", "W"][:dim] + ["P"]
    faces, fnames = open_boundary_faces(mh, names, bc)
    assert faces.shape[0] > 0 and set(fnames.tolist()) == {inlet, outlet}
    e_out = capi.Expr("0.3 + y*y - 0.5*x" if dim == 2 else "0.3 + y*z - 0.5*x", "x,y,z,t")
    fes = ["biquadratic"] * dim + ["linear"]
    _, off, _ = capi.system_elem_dofs(mh, fes)
    res = ctx.vector(lay.n)
    res.fill(2.0)
    sel = fnames == inlet
    capi.assemble_pressure_faces(ctx, mh, res, faces[sel], 1.25, off[:dim])
    capi.assemble_pressure_faces(ctx, mh, res, faces[~sel], [(e_out, np.ones((~sel).sum(), bool))], off[:dim])
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fe_tables.npz"))
    ref = ns.pressure_boundary_residual(mo, lay, bc, face_tables=G["facedofs_quad" if dim == 2 else "facedofs_hex"])     # the reference's own face tables
    assert abs(ns.pressure_boundary_residual(mo, lay, bc) - ref).max() <= 1e-13 * abs(ref).max()
    assert abs(ref).max() > 0
    got = res.to_numpy() - 2.0
    assert abs(got + ref).max() <= 1e-13 * abs(ref).max()          # RES = -aRes
    # a face whose normal velocity IS Dirichlet contributes nothing: all-Dirichlet callback -> no faces, and the oracle agrees
    closed = lambda x, name, face: (name != "P", 0.0)
    assert open_boundary_faces(mh, names, closed)[0].shape[0] == 0
    assert abs(ns.pressure_boundary_residual(mo, lay, closed)).max() == 0.0
    mh.destroy()


def test_pressure_driven_channel_reproduces_poiseuille_flow(ctx):
    """end to end: inlet / outlet open with prescribed pressures 1 and 0, walls no-slip, nu = 0.5 on the unit square -> U = (dp / (2 nu L)) y (1 - y),
    V = 0, P = 1 - x, which Q2/Q1 holds exactly; Newton + multigrid-preconditioned GMRES through the whole driver"""
    nu = 0.5
    bc = _channel_bc(4, 2, lambda x: 1.0, lambda x: 0.0)
    pb = NavierStokesMG(ctx, 2, 2, 0, 3, nu, lo=(0.0, 0.0, 0.0), hi=(1.0, 1.0, 0.0), boundary_condition=bc, open_pressure={4: 1.0, 2: 0.0}).init()
    # (HasNonLinearConverged divides ||Eps_V|| by ||V||, and V = 0 here: the reference's criterion never fires on this flow, so the steps are counted)
    pb.mgsolve(tol=1e-11, max_newton=4, lin_rtol=1e-12)
    top = pb.nlevels - 1
    ed, xy, _ = pb.meshes[top].arrays()
    sol = pb.SOL[top].to_numpy()
    off = pb.offsets[top]
    nq2, nq1 = off[1] - off[0], off[3] - off[2]
    U, V, P = sol[off[0]:off[1]], sol[off[1]:off[2]], sol[off[2]:off[3]]
    y, x = xy[:nq2, 1], xy[:nq2, 0]
    assert abs(U - y * (1 - y) / (2 * nu)).max() <= 1e-9
    assert abs(V).max() <= 1e-9
    assert abs(P - (1.0 - x[:nq1])).max() <= 1e-8
    pb.destroy()
