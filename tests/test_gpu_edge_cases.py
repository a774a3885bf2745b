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


def test_round4_entry_points_with_degenerate_input(ctx):
    """zero elements / zero faces / empty operands / bad arguments through the entry points added in round 4: no-ops or errors with a message, never a crash"""
    import ctypes
    L = ctx.L
    m = capi.Mesh.box(1, 1, 1)
    ed, xy, ff = m.arrays()
    # batched Jacobian: no elements -> nothing written; a node id out of range -> error with the reference to the argument
    w = np.full(3, 7.0)
    assert L.fh_fe_jacobian(ctx.h, 0, 2, 3, 0, 27, None, 0, None, w.ctypes.data_as(ctypes.c_void_p), None, None) == 0 and np.all(w == 7.0)
    bad = ed.copy()
    bad[0, 5] = m.nnode + 3
    with pytest.raises(RuntimeError, match="out of range"):
        capi.fe_jacobian(ctx, m, "biquadratic", elem_dof=bad)
    # one element, Gauss weights add up to the volume; Hessians of a single trilinear element: pure second derivatives vanish (to the rounding of J^-1)
    w, g, h = capi.fe_jacobian(ctx, m, "linear", hessians=True)
    assert abs(w.sum() - 1.0) <= 1e-14 and abs(h[..., :3]).max() <= 1e-14 and abs(h[..., 3:]).max() > 0.1
    # pressure faces: empty list is a no-op, missing offsets / pressures are refused
    res = ctx.vector(3 * m.nnode + 8)
    capi.assemble_pressure_faces(ctx, m, res, np.zeros((0, 9), np.int32), 1.0, [0, m.nnode, 2 * m.nnode])
    assert res.sum() == 0.0
    fn = np.ascontiguousarray(ed[:1, capi.fe_face_nodes("hex", "biquadratic", 1)], dtype=np.int32)
    assert L.fh_assemble_pressure_faces(ctx.h, 0, 3, 1, fn.ctypes.data_as(ctypes.c_void_p), None, None, 0, None, m.nnode,
                                        xy.ctypes.data_as(ctypes.c_void_p), None, -1.0, res.h) != 0
    capi.assemble_pressure_faces(ctx, m, res, fn, 2.0, [0, m.nnode, 2 * m.nnode], scale=1.0)
    got = res.to_numpy()
    assert abs(got[:m.nnode].sum() - 2.0) <= 1e-13 and abs(got[m.nnode:]).max() <= 1e-15          # face x = 1 of the unit cube: tau * area * (1, 0, 0)
    assert capi.face_normals(m, "biquadratic", np.zeros((0, 9), np.int32)).shape == (0, 3)
    # sparse products with an empty operand (host builder) and a product whose result is empty
    Z = ctx.matrix_scipy(sp.csr_matrix((4, 5)))
    B = ctx.matrix_scipy(sp.random(5, 3, density=0.6, random_state=1, format="csr"))
    assert Z.matmul(B).to_scipy().nnz == 0
    D = ctx.matrix_scipy(sp.csr_matrix(np.array([[0, 1.0], [0, 0]])))
    assert abs(D.matmul(D).to_scipy()).sum() == 0.0
    m.destroy()
