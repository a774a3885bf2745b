"""GPU: nothing in the device path may depend on the box structure or on FEMuS's particular numbering -- randomly
renumbered nodes / reordered elements (what an unstructured Gambit/MED mesh looks like to the backend) and bad inputs."""
import numpy as np
import pytest

import femus_amd
from femus_amd import capi
from oracle import femus_oracle as fo

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fe", ["biquadratic", "linear"])
@pytest.mark.parametrize("two_pass", [1, 0])
def test_randomly_renumbered_mesh(ctx, fe, two_pass):
    ctx.set_option("assemble_two_pass", two_pass)
    try:
        mo = fo.build_levels(3, 2, 2, 2)[-1]
        rng = np.random.default_rng(8)
        nv = mo.own_size[0]
        # keep the FEMuS invariant "vertex nodes first" (needed for the Q1 dof map), shuffle inside the classes, shuffle elements
        perm = np.concatenate([rng.permutation(nv), nv + rng.permutation(mo.nnode - nv)])
        inv = np.empty_like(perm)
        inv[perm] = np.arange(perm.size)
        eorder = rng.permutation(mo.nel)
        ed = inv[mo.elem_dof][eorder]
        xy = np.empty_like(mo.coords)
        xy[inv] = mo.coords
        xy = xy + rng.uniform(-0.01, 0.01, xy.shape)
        m2 = fo.Mesh("hex", ed, xy, mo.face_flag[eorder], level=1)
        m2.own_size = mo.own_size
        nc = fo.ndofs("hex", fe)
        n = fo.n_dofs(m2, fe)
        rp, col = capi.pattern_from_elements(ed[:, :nc].astype(np.int32), n)
        rpo, colo = fo.csr_pattern(m2, fe)
        assert np.array_equal(rp, rpo) and np.array_equal(col, colo)
        A = ctx.matrix_csr(n, n, rp, col)
        res = ctx.vector(n)
        asm = capi.Assembler(ctx, None, fe, A, elem_dof=ed, coords=xy)
        u = fo.lcg_fill(n, 6)
        asm.assemble(A, res, ctx.vector_from(u), 2, (1.5, 0.7))
        Ao, bo = fo.assemble_poisson(m2, fe, lambda xg: 1.5 * np.prod(np.cos(0.7 * xg), axis=-1), sol=u)
        assert abs(A.values() - Ao.data).max() <= 1e-12 * abs(Ao.data).max()
        assert abs(res.to_numpy() - bo).max() <= 1e-12 * abs(bo).max()
        # the operator is usable: SpMV against scipy
        x, y = ctx.vector_from(u), ctx.vector(n)
        y.matrix_mult(x, A)
        assert np.linalg.norm(y.to_numpy() - Ao @ u) <= 1e-12 * np.linalg.norm(Ao @ u)
    finally:
        ctx.set_option("assemble_two_pass", 1)


def test_errors_are_reported_not_crashes(ctx):
    E = femus_amd.FemusHipError
    with pytest.raises(E):
        ctx.matrix_csr(2, 2, [0, 2, 3], [1, 0, 5])                  # unsorted row and column out of range
    with pytest.raises(E):
        ctx.matrix_csr(2, 2, [0, 1, 2], [0, 1]).get_row(7)
    A = ctx.matrix_csr(3, 3, [0, 1, 2, 3], [0, 1, 2], [1.0, 2.0, 3.0])
    x3, y2 = ctx.vector(3), ctx.vector(2)
    with pytest.raises(E):
        y2.matrix_mult(x3, A)                                       # output too short
    with pytest.raises(E):
        x3.matrix_mult(x3, A)                                       # aliasing x and y
    with pytest.raises(E):
        ctx.set_option("no_such_option", 1)
    with pytest.raises(E):
        x3.get([5])                                                 # index neither owned nor ghost
    with pytest.raises(E):
        capi.Multigrid(ctx, 2).setup()                              # levels not set
    with pytest.raises(E):
        A.mat_zero_rows([9], 1.0)
    mg = capi.Multigrid(ctx, 1)
    mg.set_level(0, A)
    mg.setup()
    b, x = ctx.vector_from([1.0, 2.0, 3.0]), ctx.vector(3)
    mg.vcycle(b, x)                                                 # single level = direct solve
    assert np.allclose(x.to_numpy(), [1.0, 1.0, 1.0])


def test_device_pattern_builder_equals_the_host_builder(ctx):
    """fh_mat_create_from_elements (node -> element lists + one wave per row sorting its candidate columns in LDS) gives the pattern of
    fh_pattern_from_elements on boxes, a 2-D mesh, Q1 tables, owned-rows x local-columns shapes and a shuffled element order; the tile-local
    column lists of the SpMV (built on the device since round 4) give the product scipy computes on it"""
    import scipy.sparse as sp
    from femus_amd import capi
    rng = np.random.default_rng(4)
    for box, nl, nc in (((2, 2, 2), 2, 27), ((3, 2, 1), 2, 8), ((5, 4, 0), 2, 9), ((4, 4, 0), 1, 4), ((3, 3, 3), 2, 27)):
        m = capi.Mesh.box(*box)
        for _ in range(nl - 1):
            m = m.refine()
        ed = m.arrays()[0][:, :nc]
        ed = ed[rng.permutation(ed.shape[0])]
        n = int(ed.max()) + 1
        rp, col = capi.pattern_from_elements(ed, n)
        for rows in (n, (2 * n) // 3):
            A = ctx.matrix_from_elements(ed, rows, n)
            rpd, cold = A.pattern()
            assert np.array_equal(rpd, rp[:rows + 1]) and np.array_equal(cold, col[:rp[rows]])
            vals = rng.uniform(-1, 1, cold.size)
            A.set_values(vals)
            x = rng.uniform(-1, 1, n)
            y = ctx.vector(rows)
            y.matrix_mult(ctx.vector_from(x), A)
            ref = sp.csr_matrix((vals, cold, rpd), shape=(rows, n)) @ x
            assert np.abs(y.to_numpy() - ref).max() <= 1e-13 * np.abs(ref).max()
            A.destroy()
