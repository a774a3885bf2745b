"""GPU parity: batched Poisson assembly (element kernel + coloured CSR scatter) against the oracle."""
import numpy as np
import pytest

import femus_amd
from femus_amd import capi
from oracle import femus_oracle as fo

pytestmark = pytest.mark.gpu
ONE = lambda xg: np.ones(xg.shape[:2])


def levels(args, nl):
    ms = [capi.Mesh.box(*args)]
    for _ in range(nl - 1):
        ms.append(ms[-1].refine())
    return ms


def oracle_elem(mo, fe, u, rhs):
    et = fo.ElemType(mo.geom, fe, "seventh")
    ed = fo.elem_sys_dof(mo, fe)
    X = np.transpose(mo.coords[mo.elem_dof], (0, 2, 1))
    return fo.elem_poisson_batch(et, X, u[ed], rhs)


CASES = [((2, 2, 2), 2, "biquadratic"), ((2, 2, 2), 2, "linear"), ((4, 4, 0), 2, "biquadratic"), ((8, 8, 0), 3, "linear"),
         ((3, 2, 1), 2, "biquadratic")]


@pytest.mark.parametrize("args,nl,fe", CASES)
def test_element_matrices_match_oracle(ctx, args, nl, fe):
    m = levels(args, nl)[-1]
    mo = fo.build_levels(*args, nl)[-1]
    ed, xy, _ = m.arrays()
    assert np.array_equal(ed, mo.elem_dof) and np.array_equal(xy, mo.coords)     # integer / coordinate parity first
    n = m.n_dofs(fe)
    rp, col = capi.pattern_from_elements(ed[:, :fo.ndofs(m.geom, fe)], n)
    A = ctx.matrix_csr(n, n, rp, col)
    asm = capi.Assembler(ctx, m, fe, A)
    u = fo.lcg_fill(n, 99)
    sol = ctx.vector_from(u)
    # constant source
    K, F = asm.element_matrices(sol, 0, (1.5,))
    Ko, Fo = oracle_elem(mo, fe, u, lambda xg: 1.5 * np.ones(xg.shape[:2]))
    assert abs(K - Ko).max() <= 1e-12 * abs(Ko).max()      # fp64, FMA contraction / summation order only
    assert abs(F - Fo).max() <= 1e-12 * abs(Fo).max()
    # trigonometric source f = p0 * prod sin(p1 x_d)
    K, F = asm.element_matrices(sol, 1, (-3 * np.pi ** 2, np.pi))
    Ko, Fo = oracle_elem(mo, fe, u, lambda xg: -3 * np.pi ** 2 * np.prod(np.sin(np.pi * xg), axis=-1))
    assert abs(K - Ko).max() <= 1e-12 * abs(Ko).max()
    assert abs(F - Fo).max() <= 1e-12 * abs(Fo).max()
    asm.destroy()
    A.destroy()


def test_distorted_hex27_geometry(ctx):
    """curved / sheared HEX27 elements: perturb the coordinates of a 2x2x2 mesh"""
    m = levels((2, 2, 2), 1)[0]
    ed, xy, _ = m.arrays()
    rng = np.random.default_rng(11)
    xy = xy + rng.uniform(-0.03, 0.03, xy.shape)
    n = m.nnode
    rp, col = capi.pattern_from_elements(ed, n)
    A = ctx.matrix_csr(n, n, rp, col)
    asm = capi.Assembler(ctx, m, "biquadratic", A, elem_dof=ed, coords=xy)
    u = rng.uniform(-1, 1, n)
    K, F = asm.element_matrices(ctx.vector_from(u), 2, (2.0, 1.3))
    et = fo.ElemType("hex", "biquadratic", "seventh")
    X = np.transpose(xy[ed], (0, 2, 1))
    Ko, Fo = fo.elem_poisson_batch(et, X, u[ed], lambda xg: 2.0 * np.prod(np.cos(1.3 * xg), axis=-1))
    assert abs(K - Ko).max() <= 1e-12 * abs(Ko).max()
    assert abs(F - Fo).max() <= 1e-12 * abs(Fo).max()


@pytest.mark.parametrize("sf,mfma", [(0, 0), (0, 4), (0, 8), (0, 12), (4, 12), (5, 12), (8, 12), (10, 12)])
@pytest.mark.parametrize("args,with_sol,kind", [((3, 1, 1), True, 2), ((5, 3, 1), False, 0), ((4, 4, 3), True, 1)])
def test_matrix_core_and_vector_element_kernels_match_oracle(ctx, args, with_sol, kind, sf, mfma):
    """HEX27/Q2, 64-point rule: element matrices from the sum-factorised kernel (assemble_sf = waves per workgroup; the default),
    from the FP64 matrix-core kernel (assemble_sf 0, assemble_mfma = waves per workgroup) and from the vector kernel (both 0) on
    curved elements; element counts that leave waves of the persistent workgroups idle or give them several elements; with and
    without a solution vector (the residual's K_e u term)."""
    m = levels(args, 1)[0]
    ed, xy, _ = m.arrays()
    rng = np.random.default_rng(5)
    xy = xy + rng.uniform(-0.02, 0.02, xy.shape)
    n = m.nnode
    rp, col = capi.pattern_from_elements(ed, n)
    A = ctx.matrix_csr(n, n, rp, col)
    asm = capi.Assembler(ctx, m, "biquadratic", A, elem_dof=ed, coords=xy)
    u = rng.uniform(-1, 1, n) if with_sol else np.zeros(n)
    params = {0: (1.5,), 1: (2.0, 1.3), 2: (2.0, 1.3)}[kind]
    rhs = {0: lambda xg: 1.5 * np.ones(xg.shape[:2]), 1: lambda xg: 2.0 * np.prod(np.sin(1.3 * xg), axis=-1),
           2: lambda xg: 2.0 * np.prod(np.cos(1.3 * xg), axis=-1)}[kind]
    ctx.set_option("assemble_mfma", mfma)
    ctx.set_option("assemble_sf", sf)
    try:
        K, F = asm.element_matrices(ctx.vector_from(u) if with_sol else None, kind, params)
    finally:
        ctx.set_option("assemble_mfma", 12)
        ctx.set_option("assemble_sf", 8)
    et = fo.ElemType("hex", "biquadratic", "seventh")
    Ko, Fo = fo.elem_poisson_batch(et, np.transpose(xy[ed], (0, 2, 1)), u[ed], rhs)
    assert abs(K - Ko).max() <= 1e-12 * abs(Ko).max()
    assert abs(F - Fo).max() <= 1e-12 * abs(Fo).max()
    assert np.array_equal(K, np.transpose(K, (0, 2, 1)))                       # K_e symmetric bit for bit, as the reference's Jac
    asm.destroy()
    A.destroy()


@pytest.mark.parametrize("two_pass,emap", [(1, 1), (0, 1), (0, 0)])
@pytest.mark.parametrize("args,nl,fe", [((2, 2, 2), 3, "biquadratic"), ((8, 8, 0), 3, "linear"), ((2, 2, 2), 2, "linear")])
def test_global_assembly_matches_oracle(ctx, args, nl, fe, two_pass, emap):
    """two_pass=1: element matrices + row gather in element order (default); 0: coloured scatter (emap or binary search)"""
    ctx.set_option("assemble_two_pass", two_pass)
    ctx.set_option("assemble_emap", emap)
    try:
        m = levels(args, nl)[-1]
        mo = fo.build_levels(*args, nl)[-1]
        ed, xy, _ = m.arrays()
        n = m.n_dofs(fe)
        nc = fo.ndofs(m.geom, fe)
        rp, col = capi.pattern_from_elements(ed[:, :nc], n)
        rpo, colo = fo.csr_pattern(mo, fe)
        assert np.array_equal(rp, rpo) and np.array_equal(col, colo)          # CSR pattern: bit-exact
        A = ctx.matrix_csr(n, n, rp, col)
        res = ctx.vector(n)
        asm = capi.Assembler(ctx, m, fe, A)
        assert 1 <= asm.info()["ncolors"] <= 64
        u = fo.lcg_fill(n, 5)
        sol = ctx.vector_from(u)
        Ao, bo = fo.assemble_poisson(mo, fe, ONE, sol=u)
        for rep in range(2):                                                  # second call: zero() + reassembly
            asm.assemble(A, res, sol, 0, (1.0,))
            assert abs(A.values() - Ao.data).max() <= 1e-12 * abs(Ao.data).max()
            assert abs(res.to_numpy() - bo).max() <= 1e-12 * abs(bo).max()
        # deterministic: two assemblies give identical bits (colour order, no atomics)
        v1 = A.values().copy()
        asm.assemble(A, res, sol, 0, (1.0,))
        assert np.array_equal(v1, A.values())
        # Dirichlet rows as in MGSetLevel: SetPenalty + ZerosBoundaryResiduals
        bdc = m.dirichlet_dofs(fe)
        assert np.array_equal(bdc, fo.dirichlet_dofs(mo, fe))
        A.mat_zero_rows(bdc, 1.0)
        ref = fo.zero_rows_inplace_pattern(Ao, bdc, 1.0)
        assert abs(A.values() - ref.data).max() <= 1e-12 * abs(ref.data).max()
        asm.destroy()
    finally:
        ctx.set_option("assemble_emap", 1)
        ctx.set_option("assemble_two_pass", 1)


@pytest.mark.parametrize("sf,sumfac,kpad,mfma", [(8, 1, 1, 12), (8, 1, 0, 12), (8, 1, 28, 12), (10, 1, 1, 12), (4, 1, 1, 12),
                                                  (0, 1, 1, 12), (0, 0, 1, 12), (0, 1, 0, 12), (0, 1, 1, 0), (0, 1, 0, 0)])
def test_global_assembly_options_of_the_hex27_path(ctx, sf, sumfac, kpad, mfma):
    """HEX27/Q2 two-pass assembly with the sum-factorised element kernel (sf = waves per workgroup), or (sf 0) the matrix-core /
    vector element kernels with the Jacobian by sum factorisation or by the direct node loop; element rows padded to 256 bytes,
    224 bytes or not at all: all of them against the oracle on a curved mesh, and bit-identical when repeated (the element-row
    buffers start as NaN, so a row the kernel did not write shows)."""
    m = levels((3, 2, 2), 2)[-1]
    ed, xy, _ = m.arrays()
    rng = np.random.default_rng(17)
    xy = xy + rng.uniform(-0.01, 0.01, xy.shape)
    n = m.nnode
    rp, col = capi.pattern_from_elements(ed, n)
    A = ctx.matrix_csr(n, n, rp, col)
    res = ctx.vector(n)
    u = rng.uniform(-1, 1, n)
    ctx.set_option("assemble_sf", sf)
    ctx.set_option("assemble_sumfac", sumfac)
    ctx.set_option("assemble_kpad", kpad)
    ctx.set_option("assemble_mfma", mfma)
    ctx.set_option("debug_poison", 1)              # element-row buffers start as NaN: the row pass may read only what was written
    try:
        asm = capi.Assembler(ctx, m, "biquadratic", A, elem_dof=ed, coords=xy)      # kpad is read here
        asm.assemble(A, res, ctx.vector_from(u), 1, (2.0, 1.3))
        v1, f1 = A.values().copy(), res.to_numpy().copy()
        asm.assemble(A, res, ctx.vector_from(u), 1, (2.0, 1.3))
        assert np.array_equal(v1, A.values()) and np.array_equal(f1, res.to_numpy())
    finally:
        ctx.set_option("debug_poison", 0)
        ctx.set_option("assemble_sf", 8)
        ctx.set_option("assemble_sumfac", 1)
        ctx.set_option("assemble_kpad", 1)
        ctx.set_option("assemble_mfma", 12)
    et = fo.ElemType("hex", "biquadratic", "seventh")
    Ko, Fo = fo.elem_poisson_batch(et, np.transpose(xy[ed], (0, 2, 1)), u[ed], lambda xg: 2.0 * np.prod(np.sin(1.3 * xg), axis=-1))
    import scipy.sparse as sp
    rows = np.repeat(ed, 27, axis=1).ravel()
    cols = np.tile(ed, (1, 27)).ravel()
    Ao = sp.coo_matrix((Ko.ravel(), (rows, cols)), shape=(n, n)).tocsr()
    Ao.sort_indices()
    bo = np.zeros(n)
    np.add.at(bo, ed.ravel(), Fo.ravel())
    assert np.array_equal(Ao.indptr, rp) and np.array_equal(Ao.indices, col)
    assert abs(v1 - Ao.data).max() <= 1e-12 * abs(Ao.data).max()
    assert abs(f1 - bo).max() <= 1e-12 * abs(bo).max()
    asm.destroy()
    A.destroy()


@pytest.mark.parametrize("args,nl,fe", [((2, 2, 2), 2, "biquadratic"), ((4, 4, 0), 2, "linear")])
def test_expression_source_equals_closed_form_source(ctx, args, nl, fe):
    """the compiled-expression source (fh_assemble_poisson_expr, evaluated on the device at the Gauss points) against the
    built-in closed form of the same function"""
    m = levels(args, nl)[-1]
    ed, xy, _ = m.arrays()
    n = m.n_dofs(fe)
    rp, col = capi.pattern_from_elements(ed[:, :fo.ndofs(m.geom, fe)], n)
    A, B = ctx.matrix_csr(n, n, rp, col), ctx.matrix_csr(n, n, rp, col)
    asm = capi.Assembler(ctx, m, fe, A)
    u = ctx.vector_from(fo.lcg_fill(n, 7))
    r1, r2 = ctx.vector(n), ctx.vector(n)
    asm.assemble(A, r1, u, 1, (-2.5, 1.3))
    text = "sin(1.3*x)*sin(1.3*y)*sin(1.3*z)" if m.dim == 3 else "sin(1.3*x)*sin(1.3*y)"
    e = capi.Expr(text)
    asm.assemble_expr(B, r2, u, e, -2.5)
    assert abs(A.to_scipy() - B.to_scipy()).max() == 0.0
    assert abs(r1.to_numpy() - r2.to_numpy()).max() <= 1e-14 * abs(r1.to_numpy()).max()
    e.destroy(), asm.destroy(), A.destroy(), B.destroy()


@pytest.mark.parametrize("shape", ["box", "sheared", "mixed"])
def test_affine_fast_path_matches_oracle(ctx, shape):
    """opt-in affine-element path (reference matrices instead of quadrature) against the oracle's quadrature loop: axis-aligned
    boxes, a sheared (still affine) mesh, and a mesh where some elements are curved and must fall back to quadrature"""
    args, nl = (3, 2, 2), 2
    m = levels(args, nl)[-1]
    mo = fo.build_levels(*args, nl)[-1]
    ed, xy, _ = m.arrays()
    if shape == "sheared":
        T = np.array([[1.0, 0.3, 0.1], [0.0, 0.8, 0.25], [0.2, 0.0, 1.1]])
        xy = xy @ T.T + np.array([0.5, -1.0, 2.0])
    elif shape == "mixed":
        xy = xy.copy()
        centres = ed[::3, 26]                      # move the centre node of every third element: those become curved
        xy[centres] += 0.01
    m.set_coords(xy)
    mo.coords = xy.copy()
    n = m.nnode
    rp, col = capi.pattern_from_elements(ed, n)
    A = ctx.matrix_csr(n, n, rp, col)
    asm = capi.Assembler(ctx, m, "biquadratic", A)
    na, ng = asm.affine_count()
    assert na + ng == m.nel and (ng == 0) == (shape != "mixed") and na > 0
    u = fo.lcg_fill(n, 11)
    sol, res = ctx.vector_from(u), ctx.vector(n)
    ctx.set_option("assemble_affine", 1)
    try:
        for kind, params, rhs in ((0, (1.5,), lambda xg: 1.5 * np.ones(xg.shape[:2])),
                                  (1, (-2.0, 1.3), lambda xg: -2.0 * np.prod(np.sin(1.3 * xg), axis=-1))):
            asm.assemble(A, res, sol, kind, params)
            Ao, bo = fo.assemble_poisson(mo, "biquadratic", rhs, sol=u)
            assert abs(A.to_scipy() - Ao).max() <= 1e-12 * abs(Ao).max()
            assert abs(res.to_numpy() - bo).max() <= 1e-12 * abs(bo).max()
        e = capi.Expr("exp(x)*y - z")
        asm.assemble_expr(A, res, sol, e, 0.7)
        Ao, bo = fo.assemble_poisson(mo, "biquadratic", lambda xg: 0.7 * (np.exp(xg[..., 0]) * xg[..., 1] - xg[..., 2]), sol=u)
        assert abs(res.to_numpy() - bo).max() <= 1e-12 * abs(bo).max()
        e.destroy()
    finally:
        ctx.set_option("assemble_affine", 0)
    asm.destroy(), A.destroy()


@pytest.mark.parametrize("seed", range(6))
def test_assembly_on_shuffled_meshes(ctx, seed):
    """unstructured numbering: the nodes and the elements of a curved mesh are renumbered at random (nothing of the box generator's
    vertex/edge/face/centre order, element order or locality survives) -- pattern, matrix and residual against the oracle's element
    matrices added in ascending element order, for HEX27/Q2, HEX27/Q1, QUAD9/Q2 and QUAD9/Q1, with a solution and a closed-form source"""
    import scipy.sparse as sp
    rng = np.random.default_rng(500 + seed)
    geom_args, fe = [((3, 2, 2), "biquadratic"), ((5, 4, 3), "biquadratic"), ((4, 3, 2), "linear"), ((7, 5, 0), "biquadratic"),
                     ((9, 6, 0), "linear"), ((1, 1, 1), "biquadratic")][seed]
    m = levels(geom_args, 2)[-1]
    ed, xy, _ = m.arrays()
    geom = "hex" if xy.shape[1] == 3 else "quad"
    nc = {"linear": 2 ** xy.shape[1], "biquadratic": 3 ** xy.shape[1]}[fe]
    h = 1.0 / (2 * max(geom_args))
    xy = xy + rng.uniform(-0.04, 0.04, xy.shape) * h
    nn = xy.shape[0]
    if fe == "biquadratic":
        perm = rng.permutation(nn)                 # new id of node k
    else:
        # Q1 dofs are the vertex nodes, numbered first (FEMuS: dof id = node id for the vertices): shuffle inside the two classes
        nv = int(ed[:, :nc].max()) + 1
        perm = np.concatenate([rng.permutation(nv), nv + rng.permutation(nn - nv)])
    xy2 = np.empty_like(xy)
    xy2[perm] = xy
    ed2 = perm[ed][rng.permutation(ed.shape[0])].astype(np.int32)
    n = nn if fe == "biquadratic" else int(ed2[:, :nc].max()) + 1
    rp, col = capi.pattern_from_elements(ed2[:, :nc], n)
    A = ctx.matrix_csr(n, n, rp, col)
    res = ctx.vector(n)
    u = rng.uniform(-1, 1, n)
    asm = capi.Assembler(ctx, None, fe, A, elem_dof=ed2, coords=xy2)
    asm.assemble(A, res, ctx.vector_from(u), 1, (2.0, 1.3))
    et = fo.ElemType(geom, fe, "seventh")
    X = np.transpose(xy2[ed2], (0, 2, 1))
    Ko, Fo = fo.elem_poisson_batch(et, X, u[ed2[:, :nc]], lambda xg: 2.0 * np.prod(np.sin(1.3 * xg), axis=-1))
    rows = np.repeat(ed2[:, :nc], nc, axis=1).ravel()
    cols = np.tile(ed2[:, :nc], (1, nc)).ravel()
    Ao = sp.coo_matrix((Ko.ravel(), (rows, cols)), shape=(n, n)).tocsr()
    Ao.sort_indices()
    bo = np.zeros(n)
    np.add.at(bo, ed2[:, :nc].ravel(), Fo.ravel())
    assert np.array_equal(Ao.indptr, rp) and np.array_equal(Ao.indices, col)
    assert abs(A.values() - Ao.data).max() <= 1e-12 * abs(Ao.data).max()
    assert abs(res.to_numpy() - bo).max() <= 1e-12 * max(abs(bo).max(), abs(Fo).max())
    asm.destroy()
    A.destroy()


@pytest.mark.parametrize("args", [(2, 2, 2), (3, 2, 0)])
@pytest.mark.parametrize("fe", ["biquadratic", "linear"])
def test_batched_jacobian_with_hessians_matches_oracle(ctx, args, fe):
    """a4 in full: elem_type::Jacobian(vt, ig, Weight, phi, gradphi, nablaphi) for every element and Gauss point (fh_fe_jacobian) on curved elements
    against the oracle's restatement (ElemType.hpp:1183-1248, :1438-1537), the optional Hessians included; on AFFINE elements the Hessians
    reproduce the second derivatives of a quadratic exactly"""
    m = levels(args, 1)[0]
    ed, xy, _ = m.arrays()
    rng = np.random.default_rng(21)
    xyc = xy + rng.uniform(-0.03, 0.03, xy.shape)
    w, g, h = capi.fe_jacobian(ctx, m, fe, hessians=True, coords=xyc)
    w2, g2 = capi.fe_jacobian(ctx, m, fe, coords=xyc)
    assert np.array_equal(w, w2) and np.array_equal(g, g2)
    geom = "hex" if m.dim == 3 else "quad"
    et = fo.ElemType(geom, fe, "seventh")
    scale_g, scale_h = abs(g).max(), abs(h).max()
    for e in range(ed.shape[0]):
        vt = [xyc[ed[e, :et.nc], d] for d in range(m.dim)]
        for ig in range(et.ng):
            wo, _, go, ho = et.jacobian(vt, ig, nabla=True)
            assert abs(w[e, ig] - wo) <= 1e-13 * abs(wo)
            assert abs(g[e, ig].ravel() - go).max() <= 1e-12 * scale_g
            assert abs(h[e, ig].ravel() - ho).max() <= 1e-12 * scale_h
    if fe == "biquadratic":
        # affine elements (a sheared box): sum_j nablaphi_j q(x_j) = Hessian of the quadratic q, sum_j gradphi_j q(x_j) = its gradient
        dim = m.dim
        A = np.eye(dim) + 0.2 * rng.standard_normal((dim, dim))
        xa = xy @ A.T + 0.3
        w, g, h = capi.fe_jacobian(ctx, m, fe, hessians=True, coords=xa)
        Q = rng.standard_normal((dim, dim))
        Q = Q + Q.T
        q = 0.5 * np.einsum("ni,ij,nj->n", xa, Q, xa)
        got = np.einsum("egjk,ej->egk", h, q[ed])
        pairs = [(0, 0), (1, 1), (0, 1)] if dim == 2 else [(0, 0), (1, 1), (2, 2), (0, 1), (1, 2), (2, 0)]
        want = np.array([Q[a, b] for a, b in pairs])
        assert abs(got - want).max() <= 1e-10 * abs(Q).max()
        assert abs(w.sum() - abs(np.linalg.det(A))) <= 1e-12          # the weights add up to the volume of the sheared unit box
