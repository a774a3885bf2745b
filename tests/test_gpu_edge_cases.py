"""GPU: degenerate sizes through the C-ABI -- empty lists, single rows, empty matrix rows, one element, one multigrid level,
zero-length index lists -- the inputs the reference's interface accepts without special casing."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from femus_amd import capi
from femus_amd.poisson import PoissonMG
from oracle import femus_oracle as fo

pytestmark = pytest.mark.gpu


def test_one_by_one_and_empty_rows(ctx):
    A = ctx.matrix_scipy(sp.csr_matrix(np.array([[2.5]])))
    x, y = ctx.vector_from([4.0]), ctx.vector(1)
    y.matrix_mult(x, A)
    assert y.to_numpy()[0] == 10.0
    # a matrix whose first, middle and last rows are empty
    M = sp.csr_matrix(np.array([[0, 0, 0, 0], [1., 0, 2., 0], [0, 0, 0, 0], [0, 0, 0, 0]]))
    B = ctx.matrix_scipy(M)
    x, y = ctx.vector_from([1., 2., 3., 4.]), ctx.vector_from([9., 9., 9., 9.])
    y.matrix_mult(x, B)
    assert np.array_equal(y.to_numpy(), [0., 7., 0., 0.])
    y.matrix_mult_transpose(x, B)
    assert np.array_equal(y.to_numpy(), [2., 0., 4., 0.])
    B.mat_zero_rows(np.zeros(0, np.int32), 1.0)                  # empty Dirichlet list: no-op
    idx = capi.Index(ctx, np.zeros(0, np.int32))
    idx.zero_rows(B, 1.0)
    idx.set(y, 5.0)
    assert np.array_equal(B.to_scipy().toarray(), M.toarray())
    idx.destroy(), A.destroy(), B.destroy()


def test_single_element_meshes_assemble_and_solve(ctx):
    for box in ((1, 1, 0), (1, 1, 1)):
        pb = PoissonMG(ctx, *box, 1).init()                       # one level: the "cycle" is the exact coarse solve
        pb.assemble()
        pb.prepare()
        H = fo.build_poisson_hierarchy(*box, 1, "biquadratic", lambda xg: np.ones(xg.shape[:2]))
        assert abs(pb.A[0].to_scipy() - H.A[0]).max() <= 1e-12 * abs(H.A[0]).max()
        its, rn = pb.mgsolve(outer="gmres", rtol=1e-12, maxit=5)
        pb.update_sol()
        xd = spla.spsolve(H.A[-1].tocsc(), H.b)
        assert np.allclose(pb.SOL.to_numpy(), xd, rtol=1e-10, atol=1e-14)
        assert (np.abs(pb.SOL.to_numpy()) > 1e-14).sum() == 1      # only the element centre is free (pivoted inverse: Dirichlet rows to rounding)
        pb.destroy()


def test_device_index_equals_host_list_path(ctx):
    m = capi.Mesh.box(3, 2, 2)
    n = m.nnode
    ed, _, _ = m.arrays()
    rp, col = capi.pattern_from_elements(ed, n)
    vals = fo.lcg_fill(rp[-1], 5)
    A, B = ctx.matrix_csr(n, n, rp, col, vals), ctx.matrix_csr(n, n, rp, col, vals)
    bdc = m.dirichlet_dofs("biquadratic")
    A.mat_zero_rows(bdc, 1.0)
    idx = capi.Index(ctx, bdc)
    idx.zero_rows(B, 1.0)
    assert abs(A.to_scipy() - B.to_scipy()).max() == 0.0
    v = ctx.vector_from(fo.lcg_fill(n, 6))
    ref = v.to_numpy()
    ref[bdc] = -3.0
    idx.set(v, -3.0)
    assert np.array_equal(v.to_numpy(), ref)
    with pytest.raises(capi.FemusHipError):
        capi.Index(ctx, [n + 5]).zero_rows(A, 1.0)               # out of range: reported, not executed
    idx.destroy(), A.destroy(), B.destroy(), m.destroy()
