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


def poisson_operator(box, nl, fe="biquadratic", perturb=0.0):
    ms = fo.build_levels(*box, nl)
    m = ms[-1]
    if perturb:
        rng = np.random.default_rng(1)
        m.coords = m.coords + rng.uniform(-perturb, perturb, m.coords.shape)
    A, b = fo.assemble_poisson(m, fe, ONE)
    bdc = fo.dirichlet_dofs(m, fe)
    A = fo.zero_rows_inplace_pattern(A.tocsr(), bdc, 1.0)
    # the Galerkin hierarchy also zeroes the Dirichlet COLUMNS (rows of P): symmetric, the Dirichlet unknowns coupled to nothing
    keep = np.ones(A.shape[0])
    keep[bdc] = 0.0
    D = sp.diags(keep)
    A = (D @ A @ D + sp.diags(1.0 - keep)).tocsr()
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
