"""GPU parity: NumericVector / SparseMatrix entry points of the C-ABI against the oracle (numpy/scipy)."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import femus_oracle as fo

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.fixture(scope="module")
def q2_matrix():
    ms = fo.build_levels(2, 2, 2, 3)
    A, b = fo.assemble_poisson(ms[-1], "biquadratic", lambda xg: np.ones(xg.shape[:2]))
    return ms, A, b


def test_vector_blas1(ctx):
    n = 100003
    a, b = fo.lcg_fill(2000, 1), fo.lcg_fill(2000, 2)
    a, b = np.resize(a, n) * np.linspace(0.5, 1.5, n), np.resize(b, n) + np.linspace(-1, 1, n)
    x, y = ctx.vector_from(a), ctx.vector_from(b)
    assert abs(x.dot(y) - a @ b) <= 1e-13 * np.abs(a * b).sum()
    assert abs(x.l2_norm() - np.linalg.norm(a)) <= 1e-13 * np.linalg.norm(a)
    assert abs(x.l1_norm() - np.abs(a).sum()) <= 1e-13 * np.abs(a).sum()
    assert x.linfty_norm() == np.abs(a).max()
    assert x.max() == a.max() and x.min() == a.min()
    assert abs(x.sum() - a.sum()) <= 1e-12 * np.abs(a).sum()
    y.add(2.5, x)
    assert np.array_equal(y.to_numpy(), 2.5 * a + b) or rel(y.to_numpy(), 2.5 * a + b) < 1e-16
    y.scale(-0.5)
    y.add(1.0)
    ref = (2.5 * a + b) * -0.5 + 1.0
    assert rel(y.to_numpy(), ref) < 1e-15
    w = x.clone()
    w.pointwise_mult(x, y)
    assert rel(w.to_numpy(), a * ref) < 1e-15
    w.abs()
    assert rel(w.to_numpy(), np.abs(a * ref)) < 1e-15
    w.zero()
    assert w.linfty_norm() == 0.0
    w.fill(3.0)
    assert w.sum() == 3.0 * n
    # indexed access: operator()(i), set, add_vector_blocked with repeated indices
    w.set([5, 7], [1.0, 2.0])
    w.add_vector_blocked([1.0, 1.0, 4.0], [5, 5, 9])
    assert w.get([5, 7, 9]).tolist() == [3.0, 2.0, 7.0] and w(0) == 3.0
    # empty and odd-length vectors
    e = ctx.vector(0)
    assert e.l2_norm() == 0.0 and e.sum() == 0.0
    o = ctx.vector_from(np.arange(7.0))
    o.add(1.0, o)
    assert o.to_numpy().tolist() == (2 * np.arange(7.0)).tolist()


@pytest.mark.parametrize("kernel,tile", [(4, 1024), (4, 2048), (3, 2048), (3, 1024), (3, 4096), (0, 1024), (0, 2048), (0, 4096), (1, 2048), (2, 256), (2, 512), (2, 1024)])
def test_spmv_family_q2_matrix(ctx, q2_matrix, tile, kernel, share=None):
    ms, A, b = q2_matrix
    ctx.set_option("spmv_tile", tile)
    ctx.set_option("spmv_kernel", kernel)
    try:
        n = A.shape[0]
        xs = fo.lcg_fill(n, 12345)
        M = ctx.matrix_scipy(A)
        x, y, rhs = ctx.vector_from(xs), ctx.vector(n), ctx.vector_from(b)
        y.matrix_mult(x, M)
        ref = A @ xs
        assert rel(y.to_numpy(), ref) < 1e-14
        y.add_vector(x, M)
        assert rel(y.to_numpy(), 2 * ref) < 1e-14
        y.resid(rhs, x, M)
        assert rel(y.to_numpy(), b - ref) < 1e-14
        dinv = ctx.vector_from(fo.jacobi_dinv(A))
        y.jacobi_sweep(rhs, x, M, dinv, 2. / 3.)
        assert rel(y.to_numpy(), xs + 2. / 3. * fo.jacobi_dinv(A) * (b - ref)) < 1e-14
        M.destroy()
    finally:
        ctx.set_option("spmv_tile", 2048)
        ctx.set_option("spmv_kernel", 3)


@pytest.mark.parametrize("kernel", [4, 3, 0, 2])
def test_spmv_ragged_rows_and_long_row(ctx, kernel):
    """empty rows, 1-entry rows, a row longer than the LDS tile, rectangular shape, odd nnz offsets"""
    ctx.set_option("spmv_kernel", kernel)
    ctx.set_option("spmv_tile", 1024 if kernel == 2 else 2048)
    rng = np.random.default_rng(5)
    m, n = 777, 5000
    rows, cols, vals = [], [], []
    for i in range(m):
        k = [0, 1, 3, 64, 130][i % 5]
        if i == 400:
            k = 4500            # > tile
        c = np.sort(rng.choice(n, size=k, replace=False))
        rows += [i] * k
        cols += c.tolist()
        vals += rng.uniform(-1, 1, k).tolist()
    A = sp.csr_matrix((vals, (rows, cols)), shape=(m, n))
    xs = rng.uniform(-1, 1, n)
    M = ctx.matrix_scipy(A)
    x, y = ctx.vector_from(xs), ctx.vector(m)
    y.fill(9.0)
    y.matrix_mult(x, M)
    assert rel(y.to_numpy(), A @ xs) < 1e-14
    # transpose product through the cached explicit transpose
    z, xt = ctx.vector(n), ctx.vector_from(rng.uniform(-1, 1, m))
    z.matrix_mult_transpose(xt, M)
    assert rel(z.to_numpy(), A.T @ xt.to_numpy()) < 1e-14
    At = M.get_transpose()
    assert (At.to_scipy() != A.T.tocsr()).nnz == 0
    # empty matrix
    E = ctx.matrix_csr(0, 0, [0], [])
    assert E.nnz == 0
    ctx.set_option("spmv_kernel", 3)
    ctx.set_option("spmv_tile", 2048)


def test_matrix_row_ops(ctx, q2_matrix):
    ms, A, b = q2_matrix
    M = ctx.matrix_scipy(A)
    bdc = fo.dirichlet_dofs(ms[-1], "biquadratic")
    M.mat_zero_rows(bdc, 1.0)
    ref = fo.zero_rows_inplace_pattern(A, bdc, 1.0)
    assert np.array_equal(M.values(), ref.data)            # SetPenalty is bit-exact
    d = ctx.vector(A.shape[0])
    M.get_diagonal(d)
    assert np.array_equal(d.to_numpy(), ref.diagonal())
    M.zero_cols(bdc[:50])
    ref2 = ref.tolil()
    ref2[:, bdc[:50]] = 0.0
    assert abs(M.to_scipy() - ref2.tocsr()).max() == 0.0
    cols, vals = M.get_row(100)
    s, e = A.indptr[100], A.indptr[101]
    assert np.array_equal(cols, A.indices[s:e])
    # add_matrix_blocked / insert_row on a zeroed matrix
    M.zero()
    assert M.linfty_norm() == 0.0
    ed = ms[-1].elem_dof[3]
    K = np.arange(27.0 * 27).reshape(27, 27)
    M.add_matrix_blocked(K.ravel(), ed, ed)
    M.add_matrix_blocked(K.ravel(), ed, ed)
    S = M.to_scipy()
    assert np.array_equal(S[ed][:, ed].toarray(), 2 * K)
    M.insert_row(int(ed[0]), ed[:4], [1.0, 2.0, 3.0, 4.0])
    assert np.array_equal(M.to_scipy()[int(ed[0]), ed[:4]].toarray().ravel(), [1.0, 2.0, 3.0, 4.0])
    with pytest.raises(Exception):
        M.insert_row(0, [A.shape[0] - 1], [1.0])          # outside the pattern
    assert abs(ctx.matrix_scipy(A).l1_norm() - abs(A).sum(0).max()) < 1e-12


def test_spmv_lds_buffer_modes(ctx, q2_matrix):
    """kernel 3 with separate and with shared x / product LDS buffers give identical bits (same arithmetic)"""
    ms, A, b = q2_matrix
    n = A.shape[0]
    xs = fo.lcg_fill(n, 77)
    M = ctx.matrix_scipy(A)
    x, y = ctx.vector_from(xs), ctx.vector(n)
    outs = []
    for share in (1, 0):
        ctx.set_option("spmv_share", share)
        y.matrix_mult(x, M)
        outs.append(y.to_numpy().copy())
    ctx.set_option("spmv_share", 1)
    assert np.array_equal(outs[0], outs[1])
    assert rel(outs[0], A @ xs) < 1e-14
