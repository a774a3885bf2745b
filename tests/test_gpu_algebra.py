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


@pytest.mark.parametrize("seed", range(10))
def test_spmv_family_random_shapes(ctx, seed):
    """random CSR matrices of awkward shapes (one row, one column, empty rows in runs, rows longer than several LDS tiles, blocks that
    end exactly on a tile boundary, sizes around the 256-thread and 2048-entry granularities): y = Ax, y += Ax, r = b - Ax, the fused
    Jacobi sweep, y = A^T x and the explicit transpose against scipy"""
    rng = np.random.default_rng(1000 + seed)
    m = int(rng.choice([1, 2, 63, 64, 65, 255, 256, 257, 1000, 2047, 2048, 2049, 20011]))
    n = int(rng.choice([1, 7, 64, 2048, 2049, 30011]))
    style = seed % 5
    lens = np.zeros(m, dtype=np.int64)
    if style == 0:
        lens[:] = rng.integers(0, min(n, 9) + 1, m)
    elif style == 1:                                   # long runs of empty rows, a few heavy ones
        heavy = rng.choice(m, size=max(1, m // 50), replace=False)
        lens[heavy] = rng.integers(1, min(n, 7000) + 1, heavy.size)
    elif style == 2:                                   # every row exactly 2048 / k entries: blocks end on the tile boundary
        lens[:] = min(n, int(rng.choice([1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048])))
    elif style == 3:
        lens[:] = rng.integers(0, min(n, 300) + 1, m)
        lens[rng.integers(m)] = min(n, 9000)
    else:
        lens[:] = min(n, 125)                          # the Q2 row length
    indptr = np.concatenate([[0], np.cumsum(lens)])
    indices = np.concatenate([np.sort(rng.choice(n, size=k, replace=False)) for k in lens] + [np.zeros(0, dtype=np.int64)]).astype(np.int32)
    vals = rng.uniform(-1, 1, indices.size)
    A = sp.csr_matrix((vals, indices, indptr), shape=(m, n))
    M = ctx.matrix_scipy(A)
    xs, bs, y0 = rng.uniform(-1, 1, n), rng.uniform(-1, 1, m), rng.uniform(-1, 1, m)
    scale = np.abs(A) @ np.abs(xs) + np.abs(bs) + np.abs(y0) + 1e-300
    x, b, y = ctx.vector_from(xs), ctx.vector_from(bs), ctx.vector_from(y0)
    y.matrix_mult(x, M)
    assert np.max(np.abs(y.to_numpy() - A @ xs) / scale) < 1e-14
    y.assign(ctx.vector_from(y0))
    y.add_vector(x, M)
    assert np.max(np.abs(y.to_numpy() - (y0 + A @ xs)) / scale) < 1e-14
    y.resid(b, x, M)
    assert np.max(np.abs(y.to_numpy() - (bs - A @ xs)) / scale) < 1e-14
    if m == n:
        dinv = rng.uniform(0.5, 2.0, m)
        y.jacobi_sweep(b, x, M, ctx.vector_from(dinv), 0.7)
        assert np.max(np.abs(y.to_numpy() - (xs + 0.7 * dinv * (bs - A @ xs))) / (scale * 2 + np.abs(xs))) < 1e-14
    xt = rng.uniform(-1, 1, m)
    z = ctx.vector(n)
    z.matrix_mult_transpose(ctx.vector_from(xt), M)
    assert np.max(np.abs(z.to_numpy() - A.T @ xt) / (np.abs(A.T) @ np.abs(xt) + 1e-300)) < 1e-14
    assert (M.get_transpose().to_scipy() != A.T.tocsr()).nnz == 0


@pytest.mark.parametrize("seed", range(8))
def test_spgemm_random_shapes(ctx, seed):
    """C = A B, P^T A P and A B C on random sparse operands (empty rows and columns, dense-ish rows, 1 x 1, tall and wide shapes),
    structurally and numerically against scipy; the numeric re-run on new values keeps the pattern and follows the new values"""
    rng = np.random.default_rng(2000 + seed)

    def rand(m, n, density, empty_rows=0.2):
        A = sp.random(m, n, density=density, random_state=np.random.RandomState(int(rng.integers(1 << 30))), format="csr",
                      data_rvs=lambda k: rng.uniform(0.5, 1.5, k) * rng.choice([-1.0, 1.0], k))
        kill = rng.uniform(size=m) < empty_rows
        A = sp.diags((~kill).astype(float)) @ A
        A.eliminate_zeros()
        A.sort_indices()
        return A.tocsr()

    nf, nc = [(1, 1), (50, 7), (300, 300), (2000, 37), (513, 1200), (4000, 500), (64, 64), (1500, 1500)][seed]
    dens = [1.0, 0.2, 0.02, 0.01, 0.01, 0.003, 0.5, 0.004][seed]
    A = rand(nf, nf, dens)
    P = rand(nf, nc, min(1.0, dens * 2))
    Ad, Pd = ctx.matrix_scipy(A), ctx.matrix_scipy(P)

    def same(Md, S):
        S = S.tocsr()
        S.sort_indices()
        G = Md.to_scipy()
        # the product pattern is structural (PETSc keeps entries that cancel to zero): compare through |A| |B|
        assert G.shape == S.shape
        assert abs(G - S).max() <= 1e-13 * max(abs(S).max(), 1e-300) if S.nnz else G.nnz == 0 or abs(G).max() == 0.0
        return G

    C = Ad.matmul(Pd)
    G = same(C, A @ P)
    struct = (abs(A) @ abs(P)).tocsr()
    assert G.nnz == struct.nnz                                     # structural pattern, no numerical dropping
    Gal = capi_ptap(ctx, Pd, Ad)
    same(Gal, P.T @ A @ P)
    A2 = A.copy()
    A2.data = rng.uniform(-2, 2, A2.nnz)
    Ad.set_values(A2.data)
    Gal.ptap_numeric(Pd, Ad)
    same(Gal, P.T @ A2 @ P)
    Rd = ctx.matrix_scipy(P.T.tocsr())
    D = type(Ad).abc(Rd, Ad, Pd)
    same(D, P.T @ A2 @ P)
    Ad.set_values(A.data)
    D.abc_numeric(Rd, Ad, Pd)
    same(D, P.T @ A @ P)


def capi_ptap(ctx, P, A):
    from femus_amd import capi
    return capi.Mat.ptap(P, A)


def test_recorded_launch_sequences(ctx, q2_matrix):
    """fh_graph_begin / fh_graph_end / fh_graph_launch: a recorded sequence of SpMV-family and vector calls replays to the bits of the eager
    calls, any number of times; a cycle call inside a recording is refused; a recording with a host copy inside is reported as invalid"""
    _, A, _ = q2_matrix
    M = ctx.matrix_scipy(A)
    n = A.shape[0]
    rng = np.random.default_rng(3)
    xs, bs = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
    dinv = 1.0 / A.diagonal()

    def sequence(x, y, b, d):
        y.jacobi_sweep(b, x, M, d, 0.7)
        x.jacobi_sweep(b, y, M, d, 0.7)
        y.resid(b, x, M)

    x, y, b, d = ctx.vector_from(xs), ctx.vector(n), ctx.vector_from(bs), ctx.vector_from(dinv)
    sequence(x, y, b, d)                                   # plans built, then the eager result of TWO applications
    sequence(x, y, b, d)
    eager = (x.to_numpy().copy(), y.to_numpy().copy())
    x.upload(xs)
    with ctx.record() as rec:
        sequence(x, y, b, d)
    assert np.array_equal(x.to_numpy(), xs)               # recording does not execute
    rec.graph.launch()
    rec.graph.launch()
    assert np.array_equal(x.to_numpy(), eager[0]) and np.array_equal(y.to_numpy(), eager[1])
    rec.graph.destroy()
    with pytest.raises(Exception):                         # a host read inside a recording invalidates it
        with ctx.record():
            y.resid(b, x, M)
            y.l2_norm()
    y.resid(b, x, M)                                       # the context works on
    assert np.isfinite(y.l2_norm())
    M.destroy()
