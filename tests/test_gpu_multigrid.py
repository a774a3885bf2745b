"""GPU parity: multigrid cycle and outer solvers (LinearEquationSolver MG interface) against the oracle.
Tolerance: north_star asks 1e-10 relative on the FP solve; single cycles are compared at 1e-11."""
import numpy as np
import pytest
import scipy.sparse.linalg as spla

import femus_amd
from femus_amd import capi
from femus_amd.poisson import PoissonMG
from oracle import femus_oracle as fo

pytestmark = pytest.mark.gpu
ONE = lambda xg: np.ones(xg.shape[:2])


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def device_hierarchy(ctx, H, omega=2. / 3., npre=2, npost=2, smoother=0):
    nl = len(H.A)
    mg = capi.Multigrid(ctx, nl)
    mats = []
    for l in range(nl):
        A = ctx.matrix_scipy(H.A[l])
        P = ctx.matrix_scipy(H.P[l]) if l > 0 else None
        mats += [A, P]
        mg.set_level(l, A, P, None, smoother, omega, npre, npost)
    mg.setup()
    return mg, mats


@pytest.fixture(scope="module")
def H3():
    return fo.build_poisson_hierarchy(2, 2, 2, 3, "biquadratic", ONE)


@pytest.mark.parametrize("graph", [1, 0])
@pytest.mark.parametrize("npre,npost", [(2, 2), (1, 1), (0, 2), (3, 0)])
def test_vcycle_matches_oracle(ctx, H3, graph, npre, npost):
    ctx.set_option("use_graph", graph)
    try:
        mg, mats = device_hierarchy(ctx, H3, 2. / 3., npre, npost)
        n = H3.A[-1].shape[0]
        rhs = fo.lcg_fill(n, 3)
        b, x = ctx.vector_from(rhs), ctx.vector(n)
        for rep in range(3):          # replays of the captured graph must give the same answer
            mg.vcycle(b, x)
            ref = fo.vcycle(H3, len(H3.A) - 1, rhs, omega=2. / 3., npre=npre, npost=npost)
            assert rel(x.to_numpy(), ref) < 1e-11
        assert mg.cycle_algorithmic_bytes() > 0
        mg.destroy()
    finally:
        ctx.set_option("use_graph", 1)


@pytest.mark.parametrize("sym,mfma,block", [(1, 1, 128), (1, 1, 0), (0, 1, 128), (0, 0, 128)])
@pytest.mark.parametrize("box", [(3, 3, 3), (5, 4, 3)])
def test_coarse_inverse_variants(ctx, box, sym, mfma, block):
    """the dense coarse inverse by the symmetric sweep on the upper block triangle (symmetric operators; pivot blocks of 128 with
    rank-128 updates, or -- gj_block 0 -- the pivoted 32-wide sweep), by the general blocked
    Gauss-Jordan with the rank-32 updates on the matrix cores, and by its vector form: a one-level "hierarchy" makes the cycle the
    coarse solve itself, checked against a direct solve.  Coarse sizes 343 and 693: not multiples of the 32-wide pivot block
    or of the 64-wide update tile."""
    H = fo.build_poisson_hierarchy(*box, 1, "biquadratic", ONE)
    n = H.A[0].shape[0]
    ctx.set_option("gj_symmetric", sym)
    ctx.set_option("gj_mfma", mfma)
    ctx.set_option("gj_block", block)
    try:
        mg, mats = device_hierarchy(ctx, H)
        rhs = fo.lcg_fill(n, 11)
        b, x = ctx.vector_from(rhs), ctx.vector(n)
        mg.vcycle(b, x)
        ref = spla.spsolve(H.A[0].tocsc(), rhs)
        assert rel(x.to_numpy(), ref) < 1e-11
        mg.destroy()
    finally:
        ctx.set_option("gj_symmetric", 1)
        ctx.set_option("gj_mfma", 1)
        ctx.set_option("gj_block", 128)


@pytest.mark.parametrize("reuse", [1, 0])
def test_repeated_setup_with_new_values_keeps_or_rebuilds_the_captured_cycle(ctx, H3, reuse):
    """MGsolve prepares before every solve: a second fh_mg_setup of the same hierarchy replays the cycle it captured the first time
    (same pointers, sizes, options) on the NEW operator values.  All level operators times 4: every sweep, the residual and the exact
    coarse solve scale by a power of two, so the cycle output is exactly a quarter."""
    ctx.set_option("mg_reuse_graph", reuse)
    try:
        mg, mats = device_hierarchy(ctx, H3)
        n = H3.A[-1].shape[0]
        rhs = fo.lcg_fill(n, 9)
        b, x = ctx.vector_from(rhs), ctx.vector(n)
        mg.vcycle(b, x)
        first = x.to_numpy().copy()
        assert rel(first, fo.vcycle(H3, len(H3.A) - 1, rhs)) < 1e-11
        for l in range(len(H3.A)):
            mats[2 * l].set_values(4.0 * H3.A[l].tocsr().data)
        mg.setup()
        mg.vcycle(b, x)
        assert np.array_equal(4.0 * x.to_numpy(), first)
        # a changed smoother count is a different cycle: the graph must not be reused
        mg.set_level(len(H3.A) - 1, mats[-2], mats[-1], None, 0, 2. / 3., 1, 1)
        mg.setup()
        mg.vcycle(b, x)
        top = len(H3.A) - 1
        At, dinv = H3.A[top], fo.jacobi_dinv(H3.A[top])
        xr = fo.smooth(At, dinv, rhs, np.zeros(n), 2. / 3., 1, True)
        xr = xr + H3.P[top] @ fo.vcycle(H3, top - 1, H3.P[top].T @ (rhs - At @ xr))
        xr = fo.smooth(At, dinv, rhs, xr, 2. / 3., 1, False)
        assert rel(4.0 * x.to_numpy(), xr) < 1e-11
        mg.destroy()
    finally:
        ctx.set_option("mg_reuse_graph", 1)


@pytest.mark.parametrize("outer", ["richardson", "gmres", "cg", "fgmres"])
def test_outer_solvers_reach_direct_solution(ctx, H3, outer):
    mg, mats = device_hierarchy(ctx, H3)
    n = H3.A[-1].shape[0]
    xd = spla.spsolve(H3.A[-1].tocsc(), H3.b)
    b, x = ctx.vector_from(H3.b), ctx.vector(n)
    its, rn = mg.solve(b, x, outer=outer, rtol=1e-12, maxit=60)
    assert rel(x.to_numpy(), xd) < 1e-10            # north_star: 1e-10 relative on the FP solve
    assert its <= 30
    xo, hist = {"richardson": fo.solve_richardson_mg, "gmres": fo.solve_gmres_mg, "cg": fo.solve_pcg_mg, "fgmres": fo.solve_fgmres_mg}[outer](H3, rtol=1e-12)
    assert rel(x.to_numpy(), xo) < 1e-10
    assert abs(its - (len(hist) - 1)) <= 2           # same convergence behaviour as the restated algorithm
    if outer == "fgmres":                           # the same algorithm step by step: residual estimate of the last iteration
        assert its == len(hist) - 1 and abs(rn - hist[-1]) <= 1e-6 * hist[0]
    mg.destroy()


@pytest.mark.parametrize("outer", ["gmres", "cg", "richardson", "fgmres"])
def test_solvers_do_not_read_uninitialised_work_memory(ctx, H3, outer):
    """regression: the Krylov work vectors used to be raw allocations and `y = a x + 0 * y` read them -- NaN whenever the
    allocation landed on NaN bit patterns (seen as a rare failure of GMRES solves on freshly booted boxes).  With `debug_poison`
    the work buffers of the hierarchy and of the solvers start as NaN: the results must not change."""
    n = H3.A[-1].shape[0]
    xd = spla.spsolve(H3.A[-1].tocsc(), H3.b)
    ctx.set_option("debug_poison", 1)
    try:
        mg, mats = device_hierarchy(ctx, H3)
        b, x = ctx.vector_from(H3.b), ctx.vector(n)
        mg.vcycle(b, x)
        assert rel(x.to_numpy(), fo.vcycle(H3, len(H3.A) - 1, H3.b)) < 1e-11
        its, rn = mg.solve(b, x, outer=outer, rtol=1e-12, maxit=60)
    finally:
        ctx.set_option("debug_poison", 0)
    assert np.isfinite(x.to_numpy()).all() and np.isfinite(rn)
    assert rel(x.to_numpy(), xd) < 1e-10
    mg.destroy()


def test_preonly_is_one_cycle(ctx, H3):
    mg, mats = device_hierarchy(ctx, H3)
    n = H3.A[-1].shape[0]
    b, x, y = ctx.vector_from(H3.b), ctx.vector(n), ctx.vector(n)
    its, _ = mg.solve(b, x, outer="preonly")
    mg.vcycle(b, y)
    assert its == 1 and np.array_equal(x.to_numpy(), y.to_numpy())


def test_config1_2d_q1_three_levels(ctx):
    """BASELINE configs[0]: 2-D Q1 on 32x32, 3-level V-cycle, npre = npost = 1"""
    H = fo.build_poisson_hierarchy(8, 8, 0, 3, "linear", ONE)
    mg, mats = device_hierarchy(ctx, H, 2. / 3., 1, 1)
    n = H.A[-1].shape[0]
    assert n == 1089
    xd = spla.spsolve(H.A[-1].tocsc(), H.b)
    b, x = ctx.vector_from(H.b), ctx.vector(n)
    its, rn = mg.solve(b, x, outer="gmres", rtol=1e-12, maxit=50, restart=30)
    assert rel(x.to_numpy(), xd) < 1e-10


def test_gmres_restart_path(ctx, H3):
    mg, mats = device_hierarchy(ctx, H3, 0.3, 1, 0)      # deliberately weak smoother -> more iterations than restart
    n = H3.A[-1].shape[0]
    xd = spla.spsolve(H3.A[-1].tocsc(), H3.b)
    b, x = ctx.vector_from(H3.b), ctx.vector(n)
    its, rn = mg.solve(b, x, outer="gmres", rtol=1e-11, maxit=200, restart=5)
    assert its > 5 and rel(x.to_numpy(), xd) < 1e-9


def test_multicolour_sor_smoother(ctx, H3):
    """Richardson + SOR_PRECOND level smoother (applications/001_Poisson/main.cpp:240-242) in its multicolour form:
    cycle parity with the oracle's restatement and convergence of the outer solve"""
    mg, mats = device_hierarchy(ctx, H3, 1.0, 1, 1, smoother=1)
    n = H3.A[-1].shape[0]
    rhs = fo.lcg_fill(n, 3)
    b, x = ctx.vector_from(rhs), ctx.vector(n)
    mg.vcycle(b, x)
    ref = fo.vcycle(H3, len(H3.A) - 1, rhs, omega=1.0, npre=1, npost=1, smoother="gs_color")
    assert rel(x.to_numpy(), ref) < 1e-11
    xd = spla.spsolve(H3.A[-1].tocsc(), H3.b)
    bb, xx = ctx.vector_from(H3.b), ctx.vector(n)
    its, rn = mg.solve(bb, xx, outer="gmres", rtol=1e-12, maxit=40)
    assert rel(xx.to_numpy(), xd) < 1e-10 and its <= 12
    mg.destroy()


def test_config1_richardson_sor_as_in_001_poisson(ctx):
    """BASELINE configs[0] with the smoother choice of the reference application: RICHARDSON + SOR_PRECOND, npre = npost = 1"""
    H = fo.build_poisson_hierarchy(8, 8, 0, 3, "linear", ONE)
    mg, mats = device_hierarchy(ctx, H, 1.0, 1, 1, smoother=1)
    xd = spla.spsolve(H.A[-1].tocsc(), H.b)
    b, x = ctx.vector_from(H.b), ctx.vector(H.A[-1].shape[0])
    its, rn = mg.solve(b, x, outer="gmres", rtol=1e-12, maxit=50)
    assert rel(x.to_numpy(), xd) < 1e-10


# ---- robustness of the setup (advisor findings, round 1) -------------------------------------------------------------------
def _one_level(ctx, M, stored_zeros=False):
    import scipy.sparse as sp
    if stored_zeros:                                           # every entry of the dense matrix is a stored entry, zeros included
        n = M.shape[0]
        S = sp.csr_matrix((M.ravel(), np.tile(np.arange(n), n), np.arange(0, n * n + 1, n)), shape=(n, n))
    else:
        S = sp.csr_matrix(M)
    A = ctx.matrix_scipy(S)
    mg = capi.Multigrid(ctx, 1)
    mg.set_level(0, A, None, None, 0, 1.0, 1, 0)
    return mg, A


@pytest.mark.parametrize("n", [150, 402])
@pytest.mark.parametrize("symmetric", [True, False])
def test_coarse_inverse_of_an_indefinite_operator_with_zero_diagonal(ctx, symmetric, n):
    """saddle-point shape: zero diagonal entries, indefinite -- the pivot blocks are inverted with partial pivoting (the reference
    factors level 0 with a pivoted LU, LinearEquationSolverPetsc.hpp:131-134)"""
    rng = np.random.default_rng(4)
    M = np.zeros((n, n))
    for k in range(0, n, 2):                      # 2 x 2 blocks [[0, 1], [1, 0]]: every diagonal entry is zero
        M[k, k + 1] = M[k + 1, k] = 1.0 + 0.1 * rng.uniform()
    C = 0.05 * rng.uniform(-1, 1, (n, n))
    np.fill_diagonal(C, 0.0)
    M += (C + C.T) if symmetric else C
    assert np.all(np.diag(M) == 0.0)
    mg, A = _one_level(ctx, M)
    mg.setup()
    rhs = rng.uniform(-1, 1, n)
    b, x = ctx.vector_from(rhs), ctx.vector(n)
    mg.vcycle(b, x)
    assert rel(x.to_numpy(), np.linalg.solve(M, rhs)) < 1e-11
    mg.destroy()


def test_singular_coarse_operator_is_an_error_of_setup(ctx):
    n = 96
    M = np.eye(n)
    M[40, 40] = 0.0                                # an empty row and column: singular
    mg, A = _one_level(ctx, M)
    with pytest.raises(capi.FemusHipError, match="singular|Inf"):
        mg.setup()
    mg.destroy()


def test_restriction_follows_in_place_edits_of_the_interpolation(ctx, H3):
    """R = PP^T is derived from PP at every setup: zeroing rows / columns of PP after a first setup (ZeroInterpolatorDirichletNodes,
    LinearImplicitSystem.cpp:1032-1120, works in place) must reach the cycle"""
    import copy
    H = copy.copy(H3)
    mg, mats = device_hierarchy(ctx, H3)
    n = H3.A[-1].shape[0]
    rhs = fo.lcg_fill(n, 5)
    b, x = ctx.vector_from(rhs), ctx.vector(n)
    mg.vcycle(b, x)
    top = len(H3.A) - 1
    rows = np.arange(0, n, 7, dtype=np.int32)
    P_dev = mats[2 * top + 1]
    P_dev.mat_zero_rows(rows, 0.0)
    Pn = H3.P[top].tolil(copy=True)
    Pn[rows, :] = 0.0
    H.P = list(H3.P)
    H.P[top] = Pn.tocsr()
    mg.set_level(top, mats[2 * top], P_dev, None, 0, 2. / 3., 2, 2)
    mg.setup()
    mg.vcycle(b, x)
    ref = fo.vcycle(H, top, rhs)
    assert rel(x.to_numpy(), ref) < 1e-11
    mg.destroy()


def test_colouring_is_rebuilt_when_another_operator_is_installed(ctx):
    """the multicolour SOR ordering belongs to the matrix it was computed for: installing another operator in the same solver
    object must not reuse it"""
    Ha = fo.build_poisson_hierarchy(2, 2, 2, 2, "biquadratic", ONE)
    Hb = fo.build_poisson_hierarchy(2, 2, 2, 2, "biquadratic", ONE)
    # same sizes, another graph: drop the couplings of every third row of the fine operator of Hb to all but its diagonal
    Ab = Hb.A[1].tolil(copy=True)
    for r in range(0, Ab.shape[0], 3):
        d = Ab[r, r]
        Ab[r, :] = 0.0
        Ab[r, r] = d
    Hb.A = [Hb.A[0], Ab.tocsr()]
    n = Ha.A[1].shape[0]
    rhs = fo.lcg_fill(n, 9)
    b, x1, x2 = ctx.vector_from(rhs), ctx.vector(n), ctx.vector(n)
    mg, mats = device_hierarchy(ctx, Ha, 1.0, 1, 1, capi.SMOOTH_GS_COLOR)
    mg.vcycle(b, x1)
    A2, P2, A0 = ctx.matrix_scipy(Hb.A[1]), ctx.matrix_scipy(Hb.P[1]), ctx.matrix_scipy(Hb.A[0])
    mg.set_level(0, A0, None, None, capi.SMOOTH_GS_COLOR, 1.0, 1, 1)
    mg.set_level(1, A2, P2, None, capi.SMOOTH_GS_COLOR, 1.0, 1, 1)
    mg.setup()
    mg.vcycle(b, x1)
    fresh, mats_b = device_hierarchy(ctx, Hb, 1.0, 1, 1, capi.SMOOTH_GS_COLOR)
    fresh.vcycle(b, x2)
    assert rel(x1.to_numpy(), x2.to_numpy()) < 1e-13
    mg.destroy(), fresh.destroy()


# ---- the reference's own level smoothers: PCSOR in natural order, PCILU = ILU(0) (a18) ---------------------------------------------
@pytest.mark.parametrize("smoother,name", [(capi.SMOOTH_SOR, "sor"), (capi.SMOOTH_ILU0, "ilu0")])
@pytest.mark.parametrize("box,fe,nl", [((2, 2, 2), "biquadratic", 3), ((8, 8, 0), "linear", 3)])
def test_natural_order_smoothers_match_the_sequential_oracle(ctx, smoother, name, box, fe, nl):
    """one V(2,1) cycle with Richardson(0.8) + [one symmetric Gauss-Seidel sweep in the natural row order | the ILU(0) solve]: the
    level-scheduled device sweeps against the oracle's SEQUENTIAL sweeps (two triangular solves / IKJ elimination), 1e-11"""
    H = fo.build_poisson_hierarchy(*box, nl, fe, ONE)
    mg, mats = device_hierarchy(ctx, H, 0.8, 2, 1, smoother=smoother)
    n = H.A[-1].shape[0]
    rhs = fo.lcg_fill(n, 21)
    b, x = ctx.vector_from(rhs), ctx.vector(n)
    for rep in range(2):                                   # also as a replayed graph
        mg.vcycle(b, x)
        ref = fo.vcycle(H, nl - 1, rhs, omega=0.8, npre=2, npost=1, smoother=name)
        assert rel(x.to_numpy(), ref) < 1e-11
    mg.destroy()


@pytest.mark.parametrize("box,fe,nl", [((24, 24, 0), "biquadratic", 2), ((3, 3, 3), "biquadratic", 2), ((16, 16, 0), "linear", 3)])
def test_ilu_factorisation_with_the_pivot_rows_asked_for_ahead_gives_the_same_bits(ctx, box, fe, nl):
    """k_ilu_factor_plan (update positions from a plan built once per pattern; option ilu_ahead = 2, the default) and k_ilu_factor_ahead (descriptors of all pivots of a
    row in LDS first, the entries of three pivots ahead in registers, positions searched; 1) do the operations of the kernel with a one-pivot look-ahead (0) in the same order: one V(2,1) cycle with the ILU(0) level solves gives the same bits, and the sequential oracle's
    to 1e-11"""
    H = fo.build_poisson_hierarchy(*box, nl, fe, ONE)
    n = H.A[-1].shape[0]
    rhs = fo.lcg_fill(n, 35)
    out = []
    for ahead in (2, 1, 0):
        ctx.set_option("ilu_ahead", ahead)
        try:
            mg, mats = device_hierarchy(ctx, H, 0.8, 2, 1, smoother=capi.SMOOTH_ILU0)
            b, x = ctx.vector_from(rhs), ctx.vector(n)
            mg.vcycle(b, x)
            out.append(x.to_numpy().copy())
            mg.destroy()
        finally:
            ctx.set_option("ilu_ahead", 2)
    assert np.array_equal(out[0], out[1]) and np.array_equal(out[0], out[2])
    ref = fo.vcycle(H, nl - 1, rhs, omega=0.8, npre=2, npost=1, smoother="ilu0")
    assert rel(out[0], ref) < 1e-11


@pytest.mark.parametrize("smoother", [capi.SMOOTH_SOR, capi.SMOOTH_ILU0])
@pytest.mark.parametrize("box,fe,nl", [((24, 24, 0), "biquadratic", 2), ((3, 3, 3), "biquadratic", 2)])
def test_runs_of_small_levels_give_the_bits_of_a_launch_per_level(ctx, smoother, box, fe, nl):
    """the natural-order sweeps take runs of consecutive small levels in ONE workgroup (software pipeline over the levels, the previous level's values in LDS,
    the 16-lane sums by DPP row shifts); option tri_runs = 0 sweeps every level with a launch of its own (sums by wave shuffles): same operands in the same
    order, so one V(2,1) cycle gives the same bits either way.  2-D: 97 levels of <= 49 rows (all in runs); 3-D: levels beyond the run limits in between"""
    H = fo.build_poisson_hierarchy(*box, nl, fe, ONE)
    n = H.A[-1].shape[0]
    rhs = fo.lcg_fill(n, 33)
    out = []
    for runs in (1, 0):
        ctx.set_option("tri_runs", runs)
        try:
            mg, mats = device_hierarchy(ctx, H, 0.8, 2, 1, smoother=smoother)
            b, x = ctx.vector_from(rhs), ctx.vector(n)
            mg.vcycle(b, x)
            out.append(x.to_numpy().copy())
            mg.destroy()
        finally:
            ctx.set_option("tri_runs", 1)
    assert np.isfinite(out[0]).all() and np.abs(out[0]).max() > 0
    assert np.array_equal(out[0], out[1])


@pytest.mark.parametrize("graph", [1, 0])
@pytest.mark.parametrize("smoother,name", [(capi.SMOOTH_JACOBI, "jacobi"), (capi.SMOOTH_SOR, "sor"), (capi.SMOOTH_ILU0, "ilu0"), (capi.SMOOTH_IDENTITY, "identity")])
@pytest.mark.parametrize("npre,npost", [(2, 1), (1, 1), (3, 0), (4, 4)])
def test_gmres_level_solver_matches_the_oracle(ctx, smoother, name, npre, npost, graph):
    """`SetSolverFineGrids(GMRES)`: the reference's default level solver (and what 003_NavierStokes sets).  One V-cycle whose smoothers
    are npre / npost iterations of left-preconditioned GMRES around Jacobi / the natural-order SOR sweep / the ILU(0) solve, against
    the oracle's restatement (classical Gram-Schmidt, dense least squares), also as a replayed graph; and the smoother does what GMRES
    promises: the preconditioned residual after the cycle's pre-smoothing is the smallest over the Krylov space (checked in the oracle
    against a dense least-squares solve)."""
    H = fo.build_poisson_hierarchy(2, 2, 2, 3, "biquadratic", ONE)
    nl = 3
    ctx.set_option("use_graph", graph)
    try:
        mg = capi.Multigrid(ctx, nl)
        mats = []
        for l in range(nl):
            A = ctx.matrix_scipy(H.A[l])
            P = ctx.matrix_scipy(H.P[l]) if l > 0 else None
            mats += [A, P]
            mg.set_level(l, A, P, None, smoother, 1.0, npre, npost)
            if l > 0:
                mg.set_level_solver(l, "gmres", 3 if (npre, npost) == (4, 4) else 30)      # (4, 4) with restart 3: two GMRES cycles
        mg.setup()
        n = H.A[-1].shape[0]
        rhs = fo.lcg_fill(n, 21)
        b, x = ctx.vector_from(rhs), ctx.vector(n)
        if (npre, npost) == (4, 4):
            import functools
            orig = fo.smooth_gmres
            fo.smooth_gmres = functools.partial(orig, restart=3)
        try:
            ref = fo.vcycle(H, nl - 1, rhs, omega=1.0, npre=npre, npost=npost, smoother=name, level_solver="gmres")
        finally:
            if (npre, npost) == (4, 4):
                fo.smooth_gmres = orig
        for rep in range(2):
            mg.vcycle(b, x)
            assert rel(x.to_numpy(), ref) < 1e-9
        mg.destroy()
    finally:
        ctx.set_option("use_graph", 1)


@pytest.mark.parametrize("smoother,name", [(capi.SMOOTH_JACOBI, "jacobi"), (capi.SMOOTH_ILU0, "ilu0")])
def test_fgmres_around_gmres_smoothed_cycles(ctx, smoother, name):
    """a cycle whose level solvers are GMRES is not a fixed linear operator: the flexible outer solver (KSPFGMRES, right preconditioning,
    z_k kept) is the one that converges to the direct solution at the true-residual tolerance; iteration by iteration the same as the
    oracle's restatement"""
    H = fo.build_poisson_hierarchy(2, 2, 2, 3, "biquadratic", ONE)
    nl = 3
    mg = capi.Multigrid(ctx, nl)
    mats = []
    for l in range(nl):
        A = ctx.matrix_scipy(H.A[l])
        P = ctx.matrix_scipy(H.P[l]) if l > 0 else None
        mats += [A, P]
        mg.set_level(l, A, P, None, smoother, 1.0, 2, 2)
        if l > 0:
            mg.set_level_solver(l, "gmres", 30)
    mg.setup()
    n = H.A[-1].shape[0]
    xd = spla.spsolve(H.A[-1].tocsc(), H.b)
    b, x = ctx.vector_from(H.b), ctx.vector(n)
    its, rn = mg.solve(b, x, outer="fgmres", rtol=1e-12, maxit=40)
    assert rel(x.to_numpy(), xd) < 1e-10
    res = np.linalg.norm(H.b - H.A[-1] @ x.to_numpy())
    assert res <= 2e-12 * np.linalg.norm(H.b) and abs(rn - res) <= 1e-3 * np.linalg.norm(H.b) * 1e-9 + 0.5 * res + 1e-14 * np.linalg.norm(H.b)
    xo, hist = fo.solve_fgmres_mg(H, rtol=1e-12, maxit=40, omega=1.0, npre=2, npost=2, smoother=name, level_solver="gmres")
    assert rel(x.to_numpy(), xo) < 1e-10 and abs(its - (len(hist) - 1)) <= 1
    mg.destroy()


@pytest.mark.parametrize("npre,npost", [(2, 2), (1, 3)])
def test_identity_preconditioner_with_richardson(ctx, H3, npre, npost):
    """IDENTITY_PRECOND (PCNONE): Richardson(omega) without a preconditioner as level smoother -- omega small enough for the operator's
    spectrum -- one V-cycle against the oracle"""
    nl = len(H3.A)
    omega = 1.0 / max(abs(a).sum(axis=1).max() for a in H3.A[1:])
    mg = capi.Multigrid(ctx, nl)
    mats = []
    for l in range(nl):
        A = ctx.matrix_scipy(H3.A[l])
        P = ctx.matrix_scipy(H3.P[l]) if l > 0 else None
        mats += [A, P]
        mg.set_level(l, A, P, None, capi.SMOOTH_IDENTITY, omega, npre, npost)
    mg.setup()
    n = H3.A[-1].shape[0]
    rhs = fo.lcg_fill(n, 33)
    b, x = ctx.vector_from(rhs), ctx.vector(n)
    mg.vcycle(b, x)
    ref = fo.vcycle(H3, nl - 1, rhs, omega=omega, npre=npre, npost=npost, smoother="identity")
    assert rel(x.to_numpy(), ref) < 1e-11
    mg.destroy()


def test_config1_converges_within_the_budget_of_001_poisson(ctx):
    """BASELINE configs[0] exactly as applications/001_Poisson/main.cpp:213-257 + input/input.json set it up: 2-D Q1, 8x8 refined to
    32x32, V_CYCLE with npre = npost = 1, level solver RICHARDSON (scale 0.5, LinearEquationSolverPetsc.hpp:145) + SOR_PRECOND, outer
    GMRES with SetTolerances(1e-12, 1e-20, 1e50, 4): at most 6 linear iterations (max_number_linear_iteration) until the residual's
    l2 norm is below abs_conv_tol = 1e-9 (HasLinearConverged, LinearImplicitSystem.cpp:415-449)"""
    from femus_amd.poisson import PoissonMG
    pb = PoissonMG(ctx, 8, 8, 0, 3, fe="linear", omega=0.5, npre=1, npost=1, smoother=capi.SMOOTH_SOR).init()
    pb.assemble()
    pb.prepare()
    hist = []
    for it in range(6):
        pb.mgsolve(outer="gmres", rtol=1e-12, atol=1e-20, maxit=4)
        hist.append(pb.RES.l2_norm())
        if hist[-1] < 1e-9:
            break
    assert hist[-1] < 1e-9 and len(hist) <= 6, hist
    pb.update_sol()
    H = fo.build_poisson_hierarchy(8, 8, 0, 3, "linear", ONE)
    xd = spla.spsolve(H.A[-1].tocsc(), H.b)
    assert rel(pb.SOL.to_numpy(), xd) < 1e-8
    pb.destroy()


def test_ilu0_with_rows_beyond_the_plan_of_one_byte_positions(ctx):
    """the elimination plan keeps positions inside a row as one byte (rows of at most 254 entries); a matrix with a row of 300 entries is factorised by the
    searching kernel instead -- same factors: one preconditioned vector against the sequential oracle"""
    import scipy.sparse as sp
    n = 320
    rng = np.random.default_rng(5)
    M = sp.lil_matrix((n, n))
    for i in range(n):
        M[i, i] = 8.0 + 0.01 * i
        for j in (i - 2, i - 1, i + 1, i + 2):
            if 0 <= j < n:
                M[i, j] = -1.0 + 0.1 * rng.random()
    wide = 200
    cols = np.sort(rng.choice(np.delete(np.arange(n), wide), 299, replace=False))
    for j in cols:                                             # one row (and its column, for a symmetric pattern) with 300 entries
        M[wide, j] = 0.01 * (1 + rng.random())
        M[j, wide] = 0.01 * (1 + rng.random())
    M = M.tocsr()
    assert np.diff(M.indptr).max() >= 300
    Lo = fo.ilu0_factor(M)
    A0 = ctx.matrix_scipy(sp.identity(n, format="csr"))
    A1 = ctx.matrix_scipy(M)
    P = ctx.matrix_scipy(sp.identity(n, format="csr"))
    mg = capi.Multigrid(ctx, 2)
    mg.set_level(0, A0, None, None, 0, 1.0, 1, 0)
    mg.set_level(1, A1, P, None, capi.SMOOTH_ILU0, 1.0, 1, 0)
    mg.setup()
    rhs = fo.lcg_fill(n, 4)
    b, x = ctx.vector_from(rhs), ctx.vector(n)
    mg.vcycle(b, x)
    z = fo.ilu0_apply(Lo, rhs)
    ref = z + (rhs - M @ z)
    assert rel(x.to_numpy(), ref) < 1e-11
    mg.destroy()


def test_ilu0_shift_on_a_zero_pivot(ctx):
    """MAT_SHIFT_NONZERO with zero pivot 1e-16 (LinearEquationSolverPetsc.cpp:444-446): a pivot that cancels exactly restarts the
    factorisation of A + shift I; device and oracle take the same shift and give the same preconditioned vector"""
    import scipy.sparse as sp
    n = 40
    M = sp.lil_matrix((n, n))
    for i in range(n):
        M[i, i] = 2.0
        if i + 1 < n:
            M[i, i + 1] = M[i + 1, i] = -1.0
    M[0, 0], M[0, 1], M[1, 0], M[1, 1] = 1.0, 1.0, 1.0, 1.0            # u_11 = 1 - 1 * 1 / 1 = 0 exactly
    M = M.tocsr()
    Lo = fo.ilu0_factor(M)
    assert Lo[2] > 0.0
    A0 = ctx.matrix_scipy(sp.identity(n, format="csr"))
    A1 = ctx.matrix_scipy(M)
    P = ctx.matrix_scipy(sp.identity(n, format="csr"))
    mg = capi.Multigrid(ctx, 2)
    mg.set_level(0, A0, None, None, 0, 1.0, 1, 0)
    mg.set_level(1, A1, P, None, capi.SMOOTH_ILU0, 1.0, 1, 0)
    mg.setup()
    rhs = fo.lcg_fill(n, 2)
    b, x = ctx.vector_from(rhs), ctx.vector(n)
    mg.vcycle(b, x)
    # cycle = pre-smooth z = (LU)^-1 b, r = b - A z, coarse "solve" with the identity: x = z + r
    z = fo.ilu0_apply(Lo, rhs)
    ref = z + (rhs - M @ z)
    # the shifted pivot is u_11 = (1 + s) - 1 / (1 + s) ~ 2 s = 4.4e-14, itself only known to eps / (2 s) ~ 0.3 %: the entries that go
    # through it agree to that, the rest to rounding; without the shift the result would be Inf / NaN
    got = x.to_numpy()
    assert np.all(np.isfinite(got)) and rel(got, ref) < 1e-2 and rel(got[5:], ref[5:]) < 1e-9
    mg.destroy()


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("smoother,name", [(capi.SMOOTH_SOR, "sor"), (capi.SMOOTH_ILU0, "ilu0")])
def test_natural_order_sweeps_on_unstructured_patterns(ctx, smoother, name, seed):
    """level scheduling on dependency graphs that are not grids: random diagonally dominant matrices with unsymmetric patterns, rows
    without a lower (or upper) part, long rows, a dense-ish block -- PCSOR's symmetric sweep and the ILU(0) solve against the oracle's
    sequential loops.  Two-level wrapper: pre-smooth z = B b, r = b - A z, identity coarse level: x = z + r."""
    import scipy.sparse as sp
    rng = np.random.default_rng(3000 + seed)
    n = [1, 2, 37, 400, 1500, 5000][seed]
    dens = [1.0, 1.0, 0.3, 0.03, 0.01, 0.002][seed]
    S = sp.random(n, n, density=dens, random_state=np.random.RandomState(seed + 11), format="lil", data_rvs=lambda k: rng.uniform(-1, 1, k))
    if n >= 400:
        S[n // 2, :] = rng.uniform(-1, 1, n) * (rng.uniform(size=n) < min(0.2, 300.0 / n))          # a long row (<= 512 entries per row in the ILU kernel)
        S[:, n // 3] = (rng.uniform(-1, 1, n) * (rng.uniform(size=n) < 0.05)).reshape(-1, 1)
        for i in range(0, n, 7):                                                        # rows with nothing left / right of the diagonal
            S[i, :i] = 0.0
        for i in range(3, n, 11):
            S[i, i + 1:] = 0.0
    S = S.tocsr()
    S.setdiag(0.0)
    S.eliminate_zeros()
    assert np.diff(S.indptr).max() < 500
    M = (S + sp.diags(np.asarray(abs(S).sum(axis=1)).ravel() + 1.0)).tocsr()
    M.sort_indices()
    A0, A1 = ctx.matrix_scipy(sp.identity(n, format="csr")), ctx.matrix_scipy(M)
    P = ctx.matrix_scipy(sp.identity(n, format="csr"))
    mg = capi.Multigrid(ctx, 2)
    mg.set_level(0, A0, None, None, 0, 1.0, 1, 0)
    mg.set_level(1, A1, P, None, smoother, 1.0, 1, 0)
    mg.setup()
    rhs = rng.uniform(-1, 1, n)
    b, x = ctx.vector_from(rhs), ctx.vector(n)
    mg.vcycle(b, x)
    if name == "sor":
        z = fo.sor_symmetric_natural(M, 1.0 / M.diagonal(), rhs)
    else:
        LU = fo.ilu0_factor(M)
        assert LU[2] == 0.0                                   # diagonally dominant: no shift
        z = fo.ilu0_apply(LU, rhs)
    ref = z + (rhs - M @ z)
    assert rel(x.to_numpy(), ref) < 1e-12
    mg.destroy()


def test_multicolour_sweep_on_an_unsymmetric_pattern_is_race_free(ctx):
    """the multicolour Gauss-Seidel smoother on a matrix whose pattern is not symmetric (row i reads x_j, row j does not read x_i): the
    colouring works on the symmetrised graph, so no row shares a colour with a row it reads -- repeated runs are bit-identical and equal
    to the oracle's coloured sweep"""
    import scipy.sparse as sp
    rng = np.random.default_rng(77)
    n = 3000
    S = sp.random(n, n, density=0.004, random_state=np.random.RandomState(5), format="csr", data_rvs=lambda k: rng.uniform(-1, 1, k)).tolil()
    S.setdiag(0.0)
    S = S.tocsr()
    S.eliminate_zeros()
    M = (S + sp.diags(np.asarray(abs(S).sum(axis=1)).ravel() + 1.0)).tocsr()
    M.sort_indices()
    assert (abs(M) - abs(M).T).nnz > 0
    color, nc = fo.greedy_colors(M)
    rows, cols = M.nonzero()
    off = rows != cols
    assert not np.any(color[rows[off]] == color[cols[off]])
    A0, A1 = ctx.matrix_scipy(sp.identity(n, format="csr")), ctx.matrix_scipy(M)
    P = ctx.matrix_scipy(sp.identity(n, format="csr"))
    mg = capi.Multigrid(ctx, 2)
    mg.set_level(0, A0, None, None, 0, 1.0, 1, 0)
    mg.set_level(1, A1, P, None, capi.SMOOTH_GS_COLOR, 0.9, 2, 0)
    mg.setup()
    rhs = rng.uniform(-1, 1, n)
    b, x = ctx.vector_from(rhs), ctx.vector(n)
    mg.vcycle(b, x)
    first = x.to_numpy().copy()
    for _ in range(5):
        mg.vcycle(b, x)
        assert np.array_equal(x.to_numpy(), first)
    z = fo.smooth_sor_color(M, 1.0 / M.diagonal(), rhs, np.zeros(n), 0.9, 2, True, color, nc)
    assert rel(first, z + (rhs - M @ z)) < 1e-12
    mg.destroy()


@pytest.mark.parametrize("n", [1, 2, 3, 31, 32, 33, 63, 64, 65, 97, 256, 257, 384, 511, 769, 1000])
@pytest.mark.parametrize("kind", ["spd", "general", "permuted"])
def test_coarse_inverse_random_sizes(ctx, n, kind):
    """the blocked dense inverse of level 0 for sizes around the 32-wide pivot block: symmetric positive definite (the symmetric
    sweep), a general dense matrix (Gauss-Jordan), and one whose pivot blocks need row exchanges (zero diagonal inside the blocks);
    sparse form = dense pattern"""
    rng = np.random.default_rng(n * 7 + len(kind))
    G = rng.uniform(-1, 1, (n, n))
    if kind == "spd":
        M = G @ G.T + n * np.eye(n)
    elif kind == "general":
        M = G + n * 0.6 * np.eye(n)
    else:
        # inside every 32-block the rows are rotated by one: the diagonal is zero and the block's pivots sit off the diagonal
        M = 0.01 * G
        np.fill_diagonal(M, 0.0)
        for b0 in range(0, n, 32):
            sz = min(32, n - b0)
            for k in range(sz):
                M[b0 + k, b0 + (k + 1) % sz] += 1.0 + 0.1 * k / 32
        if n % 32 == 1:
            M[n - 1, n - 1] = 1.0            # a block of one has nothing to exchange with
    mg, A = _one_level(ctx, M)
    mg.setup()
    for rep in range(2):
        rhs = rng.uniform(-1, 1, n)
        b, x = ctx.vector_from(rhs), ctx.vector(n)
        mg.vcycle(b, x)
        ref = np.linalg.solve(M, rhs)
        assert rel(x.to_numpy(), ref) < 1e-10 * max(1.0, np.linalg.cond(M) / 1e3)
    mg.destroy()


@pytest.mark.parametrize("n", [5, 130, 300, 641])
@pytest.mark.parametrize("kind", ["spd", "general", "all_decoupled", "row_only"])
def test_coarse_solve_leaves_decoupled_unknowns_out_of_the_dense_inverse(ctx, n, kind):
    """unknowns of the coarsest level coupled to nothing (Dirichlet rows after SetPenalty whose columns the Galerkin product emptied: non-zero
    diagonal, zeros stored elsewhere in row and column) are solved by their diagonal and the dense inverse holds the rest (coarse_reduce,
    default) -- same solution as the inverse of the whole operator and as numpy; an unknown whose ROW is empty but whose column is not is
    still coupled and stays in the dense problem"""
    rng = np.random.default_rng(n * 3 + len(kind))
    G = rng.uniform(-1, 1, (n, n))
    M = G @ G.T + n * np.eye(n) if kind != "general" else G + 0.6 * n * np.eye(n)
    dec = rng.choice(n, size=n if kind == "all_decoupled" else max(1, n // 3), replace=False)
    M[dec, :] = 0.0
    if kind != "row_only":
        M[:, dec] = 0.0
    M[dec, dec] = rng.uniform(0.5, 3.0, dec.size)
    sols = []
    for reduce in (1, 0):
        ctx.set_option("coarse_reduce", reduce)
        mg, A = _one_level(ctx, M, stored_zeros=(n % 2 == 0))    # zeros as STORED entries (as after mat_zero_rows) or absent from the pattern
        mg.setup()
        rhs = rng.uniform(-1, 1, n)
        b, x = ctx.vector_from(rhs), ctx.vector(n)
        for rep in range(2):
            mg.vcycle(b, x)
        sols.append((x.to_numpy().copy(), np.linalg.solve(M, rhs)))
        mg.destroy()
    ctx.set_option("coarse_reduce", 1)
    for got, ref in sols:
        assert rel(got, ref) < 1e-11


# ---- nested dissection of the dense coarse problem (fh_mg_set_coarse_coords + option coarse_nd) ------------------------------------------------
def _poisson_level_operator(ctx, n, fe="biquadratic"):
    """the Galerkin operator of the coarse level of a two-level 3-D (or 2-D) box hierarchy -- a real coarsest-level matrix: symmetric, Q2 coupling
    through the elements, Dirichlet unknowns coupled to nothing -- and the coordinates of its unknowns"""
    pb = PoissonMG(ctx, n[0], n[1], n[2], 2, fe=fe).init()
    pb.assemble()
    pb.level_operators()
    A = pb.A[0].to_scipy().tocsr()
    xy = pb.meshes[0].arrays()[1][:pb.ndof[0]].copy()
    pb.destroy()
    return A, xy


@pytest.mark.parametrize("n,fe,nd", [((4, 4, 4), "biquadratic", 2), ((4, 4, 4), "biquadratic", 4), ((6, 4, 3), "biquadratic", 8), ((5, 5, 0), "biquadratic", 4),
                                     ((9, 8, 7), "linear", 4)])
def test_dissected_coarse_solve_is_the_exact_solve(ctx, n, fe, nd):
    """the block form (interior blocks inverted beside each other, separator Schur complement, three launches per solve) gives A^-1 b like the
    one dense inverse and like a direct solve on the host; the Dirichlet unknowns stay out of it; two preparations agree bit for bit"""
    import scipy.sparse.linalg as spla
    A, xy = _poisson_level_operator(ctx, n, fe)
    rng = np.random.default_rng(7)
    rhs = rng.uniform(-1, 1, A.shape[0])
    ref = spla.spsolve(A.tocsc(), rhs)
    got = {}
    for mode in (nd, 0):
        ctx.set_option("coarse_nd", mode)
        ctx.set_option("coarse_nd_min", 16)
        mg = capi.Multigrid(ctx, 1)
        Ad = ctx.matrix_scipy(A)
        mg.set_level(0, Ad, None, None, 0, 1.0, 1, 0)
        mg.set_coarse_coords(xy)
        mg.setup()
        nden, nblk, nsep, big = mg.coarse_info()
        if mode:
            assert nblk >= 2 and 0 < nsep < nden and big < nden, (nden, nblk, nsep, big)
        else:
            assert nblk == 0
        b, x = ctx.vector_from(rhs), ctx.vector(A.shape[0])
        mg.vcycle(b, x)
        first = x.to_numpy().copy()
        mg.setup()                                      # a second preparation of the same operator
        mg.vcycle(b, x)
        assert np.array_equal(first, x.to_numpy())
        got[mode] = first
        mg.destroy()
        Ad.destroy()
    ctx.set_option("coarse_nd", 8)
    ctx.set_option("coarse_nd_min", 1024)
    assert rel(got[nd], ref) < 1e-11 and rel(got[0], ref) < 1e-11


def test_dissection_cuts_a_q2_block_at_element_planes(ctx):
    """the separator of the 8^3-element coarse level of the bench hierarchy (3375 coupled unknowns): four blocks of 15 x 7 x 7 unknowns and
    the three element-boundary planes between them (225 + 2 x 105), or eight blocks of 7 x 7 x 7: layers of Q2 nodes at element boundaries separate, mid-planes do not"""
    A, xy = _poisson_level_operator(ctx, (8, 8, 8))
    Ad = ctx.matrix_scipy(A)
    for blocks, expect in ((4, (3375, 4, 435, 735)), (8, (3375, 8, 631, 343))):       # 8 (the default): 7 x 7 x 7 blocks, 225 + 2 x 105 + 4 x 49
        ctx.set_option("coarse_nd", blocks)
        mg = capi.Multigrid(ctx, 1)
        mg.set_level(0, Ad, None, None, 0, 1.0, 1, 0)
        mg.set_coarse_coords(xy)
        mg.setup()
        assert mg.coarse_info() == expect
        mg.destroy()
    Ad.destroy()


def test_dissected_coarse_solve_falls_back(ctx):
    """symmetric but indefinite (no usable unpivoted pivot in a block) and unsymmetric operators with coordinates set: the one dense inverse
    with its pivoted fall-back serves them, the solution is still exact"""
    rng = np.random.default_rng(3)
    n = 160
    xy = rng.uniform(0, 1, (n, 3))
    M = np.zeros((n, n))
    for k in range(0, n, 2):
        M[k, k + 1] = M[k + 1, k] = 1.0 + 0.1 * rng.uniform()
    C = 0.05 * rng.uniform(-1, 1, (n, n))
    for name, Mk in (("indefinite", M + C + C.T), ("unsymmetric", M + C + 2.0 * np.eye(n))):
        ctx.set_option("coarse_nd_min", 16)
        mg, A = _one_level(ctx, Mk)
        mg.set_coarse_coords(xy)
        mg.setup()
        assert mg.coarse_info()[1] == 0, name
        rhs = rng.uniform(-1, 1, n)
        b, x = ctx.vector_from(rhs), ctx.vector(n)
        mg.vcycle(b, x)
        assert rel(x.to_numpy(), np.linalg.solve(Mk, rhs)) < 1e-10, name
        mg.destroy()
    ctx.set_option("coarse_nd_min", 1024)


def test_hierarchy_solve_with_and_without_the_dissection(ctx):
    """a whole 3-level solve whose coarsest level (6^3 elements, 1331 coupled unknowns) is dissected: same iteration count and the same solution
    (1e-11) as with the one dense inverse; re-preparation keeps the block form"""
    sols = []
    for mode in (4, 0):
        ctx.set_option("coarse_nd", mode)
        pb = PoissonMG(ctx, 6, 6, 6, 3).init()
        pb.assemble()
        pb.prepare()
        assert (pb.mg.coarse_info()[1] >= 2) == (mode > 0)
        its, rn = pb.mgsolve(outer="gmres", rtol=1e-12)
        x = pb.EPSC.to_numpy().copy()
        # (the first assembly ran the fused path; the Galerkin product then asked for element rows, so the later ones run the two-pass path, whose
        # sums associate differently across clusters: same solution to rounding after the first re-assembly, the same bits from then on)
        xs = []
        for _ in range(2):
            pb.assemble()
            pb.prepare()
            assert (pb.mg.coarse_info()[1] >= 2) == (mode > 0)
            its2, _ = pb.mgsolve(outer="gmres", rtol=1e-12)
            assert its2 == its
            xs.append(pb.EPSC.to_numpy().copy())
        assert rel(xs[0], x) < 1e-12 and np.array_equal(xs[0], xs[1])
        sols.append((its, x))
        pb.destroy()
    ctx.set_option("coarse_nd", 8)
    assert sols[0][0] == sols[1][0] and rel(sols[0][1], sols[1][1]) < 1e-11


def test_dissected_coarse_solve_of_a_disconnected_operator(ctx):
    """two bodies that share nothing: the first cut needs no separator at all (two blocks, empty Schur complement); with more blocks the
    separators lie inside the bodies -- exact either way"""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    A1, xy1 = _poisson_level_operator(ctx, (4, 3, 3))
    A = sp.block_diag([A1, A1]).tocsr()
    xy = np.vstack([xy1, xy1 + np.array([10.0, 0.0, 0.0])])
    rng = np.random.default_rng(2)
    rhs = rng.uniform(-1, 1, A.shape[0])
    ref = spla.spsolve(A.tocsc(), rhs)
    Ad = ctx.matrix_scipy(A)
    ctx.set_option("coarse_nd_min", 16)
    for blocks in (2, 4):
        ctx.set_option("coarse_nd", blocks)
        mg = capi.Multigrid(ctx, 1)
        mg.set_level(0, Ad, None, None, 0, 1.0, 1, 0)
        mg.set_coarse_coords(xy)
        mg.setup()
        nden, nblk, nsep, big = mg.coarse_info()
        assert nblk == blocks and (nsep == 0) == (blocks == 2), (nden, nblk, nsep, big)
        b, x = ctx.vector_from(rhs), ctx.vector(A.shape[0])
        mg.vcycle(b, x)
        assert rel(x.to_numpy(), ref) < 1e-11
        mg.destroy()
    Ad.destroy()
    ctx.set_option("coarse_nd", 8)
    ctx.set_option("coarse_nd_min", 1024)


@pytest.mark.parametrize("graph", [1, 0])
@pytest.mark.parametrize("kind", ["full", "additive", "kaskade", "multiplicative"])
@pytest.mark.parametrize("smoother,name,npre,npost", [(capi.SMOOTH_JACOBI, "jacobi", 2, 2), (capi.SMOOTH_JACOBI, "jacobi", 1, 3), (capi.SMOOTH_SOR, "sor", 1, 1)])
def test_pcmg_types_match_the_oracle(ctx, H3, kind, smoother, name, npre, npost, graph):
    """PCMGSetType FULL / ADDITIVE / KASKADE / MULTIPLICATIVE (MGInit's MgSmootherType, LinearEquationSolverPetsc.cpp:199-214): one application of
    the preconditioner against the oracle's restatement of PETSc's four cycle routines, eager and replayed from the captured graph; switching
    the type re-captures the cycle"""
    ctx.set_option("use_graph", graph)
    try:
        mg, mats = device_hierarchy(ctx, H3, 2. / 3., npre, npost, smoother)
        n = H3.A[-1].shape[0]
        rhs = fo.lcg_fill(n, 17)
        b, x = ctx.vector_from(rhs), ctx.vector(n)
        for k in (kind, "multiplicative", kind):
            mg.set_cycle_type(k)
            ref = fo.pcmg_apply(H3, rhs, k, omega=2. / 3., npre=npre, npost=npost, smoother=name)
            for rep in range(2):
                mg.vcycle(b, x)
                assert rel(x.to_numpy(), ref) < 1e-11, k
        mg.destroy()
    finally:
        ctx.set_option("use_graph", 1)


@pytest.mark.parametrize("kind", ["full", "additive", "kaskade"])
def test_pcmg_types_as_preconditioners_of_the_outer_solver(ctx, H3, kind):
    """every PCMG type preconditions GMRES to the direct solution (the additive and cascadic forms are no convergent iterations by themselves)"""
    mg, mats = device_hierarchy(ctx, H3)
    mg.set_cycle_type(kind)
    n = H3.A[-1].shape[0]
    xd = spla.spsolve(H3.A[-1].tocsc(), H3.b)
    b, x = ctx.vector_from(H3.b), ctx.vector(n)
    its, rn = mg.solve(b, x, outer="gmres", rtol=1e-12, maxit=100)
    assert rel(x.to_numpy(), xd) < 1e-10
    assert its <= {"full": 12, "additive": 60, "kaskade": 40}[kind]
    mg.destroy()
