"""GPU parity of the sparse exact solve (fh_direct_*: multifrontal factorisation over a nested-dissection tree) against scipy's sparse LU on
the operators this path meets: penalised Q2 / Q1 Poisson operators of box meshes and of a curved Gambit-like mesh, with coordinates (layer
cuts) and without (breadth-first level sets), from one leaf to the 33^3-node mesh the round-3 verdict names (35 937 unknowns, more than the
16 384 the dense coarse solve holds).  The reference hands these solves to MUMPS through PETSc (LinearEquationSolverPetsc.hpp:131-138)."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from femus_amd import capi
from oracle import femus_oracle as fo

pytestmark = pytest.mark.gpu
ONE = lambda xg: np.ones(xg.shape[:2])


def poisson_operator(box, nl, fe="biquadratic", perturb=0.0, zero_columns=True):
    ms = fo.build_levels(*box, nl)
    m = ms[-1]
    if perturb:
        rng = np.random.default_rng(1)
        m.coords = m.coords + rng.uniform(-perturb, perturb, m.coords.shape)
    A, b = fo.assemble_poisson(m, fe, ONE)
    bdc = fo.dirichlet_dofs(m, fe)
    A = fo.zero_rows_inplace_pattern(A.tocsr(), bdc, 1.0)
    # the Galerkin hierarchy also zeroes the Dirichlet COLUMNS (rows of P): symmetric, the Dirichlet unknowns coupled to nothing
    if zero_columns:
        keep = np.ones(A.shape[0])
        keep[bdc] = 0.0
        D = sp.diags(keep)
        A2 = (D @ A @ D + sp.diags(1.0 - keep)).tocsr()
        A = (A2 + A * 0.0).tocsr()                 # keep the pattern (stored zeros)
    A.sort_indices()
    n = A.shape[0]
    return A, m.coords[:n] if fe == "biquadratic" else m.coords[:n], bdc


@pytest.mark.parametrize("box,nl,fe,leaf,with_coords", [((2, 2, 2), 1, "biquadratic", 0, True), ((4, 4, 4), 1, "biquadratic", 64, True),
                                                         ((4, 4, 4), 1, "biquadratic", 64, False), ((8, 8, 0), 2, "biquadratic", 32, True),
                                                         ((4, 4, 4), 2, "biquadratic", 0, True), ((4, 4, 4), 2, "biquadratic", 100, False),
                                                         ((6, 5, 4), 2, "linear", 50, False)])
def test_sparse_exact_solve_matches_scipy(ctx, box, nl, fe, leaf, with_coords):
    A, xy, bdc = poisson_operator(box, nl, fe, perturb=0.01)
    n = A.shape[0]
    M = ctx.matrix_scipy(A)
    d = capi.Direct(ctx, M, xy if with_coords else None, leaf).factor()
    info = d.info()
    assert info["coupled"] == n - len(bdc) and info["fronts"] >= 1
    lu = spla.splu(A.tocsc())
    rng = np.random.default_rng(5)
    for rep in range(2):
        rhs = rng.uniform(-1, 1, n)
        b, x = ctx.vector_from(rhs), ctx.vector(n)
        d.solve(b, x)
        ref = lu.solve(rhs)
        assert np.linalg.norm(x.to_numpy() - ref) <= 1e-12 * np.linalg.norm(ref)
    # new values on the same pattern: numeric factorisation only
    A2 = A.copy()
    A2.data *= 1.0 + 0.1 * np.sin(np.arange(A2.nnz))          # unsymmetric scaling would be refused: keep it symmetric
    A2 = ((A2 + A2.T) * 0.5).tocsr()
    A2.sort_indices()
    assert np.array_equal(A2.indices, A.indices)
    M.set_values(A2.data)
    d.factor()
    rhs = rng.uniform(-1, 1, n)
    b, x = ctx.vector_from(rhs), ctx.vector(n)
    d.solve(b, x)
    ref = spla.splu(A2.tocsc()).solve(rhs)
    assert np.linalg.norm(x.to_numpy() - ref) <= 1e-11 * np.linalg.norm(ref)
    d.destroy()
    M.destroy()


def test_dirichlet_rows_with_their_columns_left_in_place(ctx):
    """SetPenalty alone (rows zeroed, columns not -- what a directly assembled level holds, LinearEquationSolverPetsc.cpp:428-436): x_d = b_d and the
    Dirichlet columns move to the right-hand side of the free unknowns, whose block is symmetric"""
    A, xy, bdc = poisson_operator((3, 3, 3), 2, perturb=0.01, zero_columns=False)
    n = A.shape[0]
    assert abs(A - A.T).max() > 1e-3
    M = ctx.matrix_scipy(A)
    d = capi.Direct(ctx, M, xy, 80).factor()
    rhs = np.random.default_rng(8).uniform(-1, 1, n)
    b, x = ctx.vector_from(rhs), ctx.vector(n)
    d.solve(b, x)
    ref = spla.splu(A.tocsc()).solve(rhs)
    assert np.linalg.norm(x.to_numpy() - ref) <= 1e-12 * np.linalg.norm(ref)
    assert np.array_equal(x.to_numpy()[bdc], rhs[bdc])
    d.destroy()
    M.destroy()


def test_sparse_exact_solve_beyond_the_dense_limit(ctx):
    """33^3 nodes (Q2 on 16^3 elements): 35 937 unknowns, 29 791 of them coupled -- the size the dense coarse solve (<= 16 384) refuses"""
    A, xy, bdc = poisson_operator((4, 4, 4), 3)
    n = A.shape[0]
    assert n == 33 ** 3
    M = ctx.matrix_scipy(A)
    d = capi.Direct(ctx, M, xy).factor()
    info = d.info()
    assert info["coupled"] == 31 ** 3 and info["height"] >= 5
    rhs = np.random.default_rng(2).uniform(-1, 1, n)
    b, x = ctx.vector_from(rhs), ctx.vector(n)
    d.solve(b, x)
    ref = spla.splu(A.tocsc()).solve(rhs)
    assert np.linalg.norm(x.to_numpy() - ref) <= 1e-12 * np.linalg.norm(ref)
    d.destroy()
    M.destroy()


def test_unsymmetric_and_singular_operators_are_refused(ctx):
    A, xy, bdc = poisson_operator((2, 2, 2), 1)
    U = A.copy().tolil()
    free = np.setdiff1d(np.arange(A.shape[0]), bdc)
    U[free[0], free[1]] = U[free[0], free[1]] + 0.5
    M = ctx.matrix_scipy(U.tocsr())
    d = capi.Direct(ctx, M, xy)
    with pytest.raises(capi.FemusHipError, match="not symmetric"):
        d.factor()
    d.destroy()
    M.destroy()
    Z = A.copy().tocsr()
    Z.data[:] = 0.0
    Z = (Z + sp.diags(np.where(np.isin(np.arange(A.shape[0]), bdc), 1.0, 0.0))).tocsr()
    M = ctx.matrix_scipy(Z)
    d = capi.Direct(ctx, M, xy)
    with pytest.raises(capi.FemusHipError, match="zero diagonal"):
        d.factor()
    d.destroy()
    M.destroy()


def _hierarchy(box, nl):
    return fo.build_poisson_hierarchy(*box, nl, "biquadratic", ONE)


@pytest.mark.parametrize("mode", [2, 0])
def test_coarse_level_through_the_sparse_exact_solve(ctx, mode):
    """the multigrid's exact coarse solve through fh_direct (option coarse_direct 2 = always) gives the cycle of the dense inverse (0 = never) and
    of the oracle; the captured cycle replays it"""
    H = _hierarchy((2, 2, 2), 3)
    ctx.set_option("coarse_direct", mode)
    try:
        nl = len(H.A)
        mg = capi.Multigrid(ctx, nl)
        mats = []
        for l in range(nl):
            A = ctx.matrix_scipy(H.A[l])
            P = ctx.matrix_scipy(H.P[l]) if l > 0 else None
            mats += [A, P]
            mg.set_level(l, A, P, None, capi.SMOOTH_JACOBI, 2. / 3., 2, 2)
        mg.set_coarse_coords(H.meshes[0].coords[:H.A[0].shape[0]])
        mg.setup()
        n = H.A[-1].shape[0]
        rhs = fo.lcg_fill(n, 4)
        b, x = ctx.vector_from(rhs), ctx.vector(n)
        ref = fo.vcycle(H, nl - 1, rhs)
        for rep in range(3):
            mg.vcycle(b, x)
            assert np.linalg.norm(x.to_numpy() - ref) <= 1e-11 * np.linalg.norm(ref)
        mg.destroy()
    finally:
        ctx.set_option("coarse_direct", 1)


def test_a_coarse_level_beyond_the_dense_limit(ctx):
    """two-level cycle whose coarse level has 35 937 unknowns (33^3 nodes): the dense inverse would refuse it (> 16 384); GMRES around the cycle
    reaches the direct solution of the 65^3-node problem"""
    Hs = fo.build_poisson_hierarchy(4, 4, 4, 4, "biquadratic", ONE)
    meshes = Hs.meshes[2:]
    A = [Hs.A[2], Hs.A[3]]
    P = [None, Hs.P[3]]
    mg = capi.Multigrid(ctx, 2)
    A0, A1, P1 = ctx.matrix_scipy(A[0]), ctx.matrix_scipy(A[1]), ctx.matrix_scipy(P[1])
    mg.set_level(0, A0, None, None, capi.SMOOTH_JACOBI, 2. / 3., 2, 2)
    mg.set_level(1, A1, P1, None, capi.SMOOTH_JACOBI, 2. / 3., 2, 2)
    mg.set_coarse_coords(meshes[0].coords[:A[0].shape[0]])
    mg.setup()
    assert A[0].shape[0] == 33 ** 3
    n = A[1].shape[0]
    b, x = ctx.vector_from(Hs.b), ctx.vector(n)
    its, rn = mg.solve(b, x, outer="fgmres", rtol=1e-11, maxit=40)          # (flexible GMRES tests the TRUE residual)
    r = Hs.b - A[1] @ x.to_numpy()
    assert np.linalg.norm(r) <= 1e-10 * np.linalg.norm(Hs.b), (np.linalg.norm(r) / np.linalg.norm(Hs.b), its)
    assert its <= 14
    mg.destroy()


@pytest.mark.parametrize("solver", ["richardson", "gmres"])
def test_exact_solve_as_level_preconditioner(ctx, solver):
    """MLU_PRECOND / LU_PRECOND on a level (PetscPreconditioner.cpp:147-160): Richardson(1) around B = A^-1 makes the level exact after one
    iteration, so a two-level cycle with it equals the direct solution of the fine system (the coarse correction of an exact iterate is zero)"""
    H = _hierarchy((2, 2, 2), 2)
    mg = capi.Multigrid(ctx, 2)
    A0, A1, P1 = ctx.matrix_scipy(H.A[0]), ctx.matrix_scipy(H.A[1]), ctx.matrix_scipy(H.P[1])
    mg.set_level(0, A0, None, None, capi.SMOOTH_JACOBI, 1.0, 1, 1)
    mg.set_level(1, A1, P1, None, capi.SMOOTH_LU, 1.0, 1, 1)
    if solver == "gmres":
        mg.set_level_solver(1, "gmres", 30)
    mg.set_level_coords(1, H.meshes[1].coords[:H.A[1].shape[0]])
    mg.setup()
    n = H.A[1].shape[0]
    rhs = fo.lcg_fill(n, 9)
    b, x = ctx.vector_from(rhs), ctx.vector(n)
    mg.vcycle(b, x)
    ref = spla.splu(H.A[1].tocsc()).solve(rhs)
    assert np.linalg.norm(x.to_numpy() - ref) <= 1e-11 * np.linalg.norm(ref)
    mg.destroy()
