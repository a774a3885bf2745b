"""GPU parity of the fused cluster assembly (k_cluster_q2hex_sf + k_rows_partial; `00_poisson_eqn_..._separate.hpp:165-205` + the scatter of
`PetscMatrix.cpp:699-729`): against the oracle's element loop, against the two-pass path, on meshes that offer the sibling structure and on
meshes that do not, with rows outside the matrix, after a change of the value array, and as the source of the element-wise Galerkin product."""
import numpy as np
import pytest
import scipy.sparse as sp

import femus_amd
from femus_amd import capi
from oracle import femus_oracle as fo

pytestmark = pytest.mark.gpu


def levels(args, nl):
    ms = [capi.Mesh.box(*args)]
    for _ in range(nl - 1):
        ms.append(ms[-1].refine())
    return ms


def oracle_global(ed, xy, u, rhs, n):
    et = fo.ElemType("hex", "biquadratic", "seventh")
    Ko, Fo = fo.elem_poisson_batch(et, np.transpose(xy[ed], (0, 2, 1)), u[ed], rhs)
    rows = np.repeat(ed, 27, axis=1).ravel()
    cols = np.tile(ed, (1, 27)).ravel()
    Ao = sp.coo_matrix((Ko.ravel(), (rows, cols)), shape=(n, n)).tocsr()
    Ao.sort_indices()
    bo = np.zeros(n)
    np.add.at(bo, ed.ravel(), Fo.ravel())
    return Ao, bo


KINDS = {0: ((1.5,), lambda xg: 1.5 * np.ones(xg.shape[:2])), 1: ((2.0, 1.3), lambda xg: 2.0 * np.prod(np.sin(1.3 * xg), axis=-1)),
         2: ((2.0, 1.3), lambda xg: 2.0 * np.prod(np.cos(1.3 * xg), axis=-1))}


@pytest.mark.parametrize("args,nl,kind,with_sol", [((1, 1, 1), 2, 0, True), ((2, 1, 1), 2, 1, True), ((3, 2, 2), 2, 2, False), ((2, 2, 2), 3, 1, True)])
def test_fused_assembly_matches_oracle_and_two_pass(ctx, args, nl, kind, with_sol):
    """curved refined meshes (1, 2, 12 and 64 clusters: one cluster = every row complete; boundary clusters; interior clusters): the fused path
    runs (fused_info), equals the oracle's element loop to 1e-12 and the two-pass path to rounding, is bit-identical when repeated from
    NaN-poisoned buffers, and writes every row (the matrix starts as NaN)."""
    m = levels(args, nl)[-1]
    ed, xy, _ = m.arrays()
    rng = np.random.default_rng(3)
    xy = xy + rng.uniform(-0.01, 0.01, xy.shape) / 2 ** (nl - 1)
    n = m.nnode
    rp, col = capi.pattern_from_elements(ed, n)
    u = rng.uniform(-1, 1, n) if with_sol else np.zeros(n)
    params, rhs = KINDS[kind]
    out = {}
    for fused in (1, 0):
        ctx.set_option("assemble_fused", fused)
        ctx.set_option("debug_poison", 1)
        try:
            A = ctx.matrix_csr(n, n, rp, col, np.full(col.size, np.nan))
            res = ctx.vector_from(np.full(n, np.nan))
            asm = capi.Assembler(ctx, m, "biquadratic", A, elem_dof=ed, coords=xy)
            info = asm.fused_info()
            assert info["active"] == bool(fused) and (not fused or info["clusters"] == m.nel // 8)
            sol = ctx.vector_from(u) if with_sol else None
            asm.assemble(A, res, sol, kind, params)
            v1, f1 = A.values().copy(), res.to_numpy().copy()
            asm.assemble(A, res, sol, kind, params)
            assert np.array_equal(v1, A.values()) and np.array_equal(f1, res.to_numpy())
            out[fused] = (v1, f1)
            asm.destroy(), A.destroy()
        finally:
            ctx.set_option("debug_poison", 0)
            ctx.set_option("assemble_fused", 1)
    Ao, bo = oracle_global(ed, xy, u, rhs, n)
    assert np.array_equal(Ao.indptr, rp) and np.array_equal(Ao.indices, col)
    for fused in (1, 0):
        v, f = out[fused]
        assert np.isfinite(v).all() and np.isfinite(f).all()
        row_scale = np.repeat(np.maximum.reduceat(abs(Ao.data), rp[:-1]), np.diff(rp))     # per row: its largest entry (the diagonal)
        assert (abs(v - Ao.data) / row_scale).max() <= 1e-12
        assert abs(f - bo).max() <= 1e-12 * abs(bo).max()
    assert abs(out[1][0] - out[0][0]).max() <= 4e-16 * abs(Ao.data).max()      # same element matrices, sums grouped per cluster
    assert abs(out[1][1] - out[0][1]).max() <= 1e-15 * abs(bo).max()
    # ... and inside a cluster both paths add in ascending element order: a row all of whose elements lie in ONE cluster has the same bits
    blocks = [set() for _ in range(n)]
    for e in range(ed.shape[0]):
        for nd in ed[e]:
            blocks[nd].add(e // 8)
    one = np.array([len(b) == 1 for b in blocks])
    assert one.any() and np.array_equal(out[1][1][one], out[0][1][one])
    for r in np.where(one)[0]:
        assert np.array_equal(out[1][0][rp[r]:rp[r + 1]], out[0][0][rp[r]:rp[r + 1]])


def test_fused_plan_is_refused_where_the_mesh_has_no_sibling_groups(ctx):
    """an unrefined box (elements are no siblings) and a refined mesh whose elements were shuffled: the template check fails, the two-pass path
    runs and gives the oracle's operator; a refined mesh keeps the fused path when WHOLE sibling groups are permuted"""
    rng = np.random.default_rng(9)
    for case in ("unrefined", "shuffled", "groups"):
        m = levels((4, 2, 2), 1)[0] if case == "unrefined" else levels((2, 2, 1), 2)[-1]
        ed, xy, _ = m.arrays()
        if case == "shuffled":
            ed = ed[rng.permutation(ed.shape[0])]
        elif case == "groups":
            ed = ed.reshape(-1, 8, 27)[rng.permutation(ed.shape[0] // 8)].reshape(-1, 27)
        xy = xy + rng.uniform(-0.01, 0.01, xy.shape)
        n = m.nnode
        rp, col = capi.pattern_from_elements(ed, n)
        A = ctx.matrix_csr(n, n, rp, col)
        res = ctx.vector(n)
        asm = capi.Assembler(ctx, m, "biquadratic", A, elem_dof=ed, coords=xy)
        assert asm.fused_info()["active"] == (case == "groups")
        u = rng.uniform(-1, 1, n)
        asm.assemble(A, res, ctx.vector_from(u), 1, (2.0, 1.3))
        Ao, bo = oracle_global(ed, xy, u, KINDS[1][1], n)
        assert abs(A.values() - Ao.data).max() <= 1e-12 * abs(Ao.data).max()
        assert abs(res.to_numpy() - bo).max() <= 1e-12 * abs(bo).max()
        asm.destroy(), A.destroy()


def test_fused_assembly_with_rows_outside_the_matrix_and_another_value_array(ctx):
    """owned rows x local columns (what a rank of the distributed driver assembles): rows of nodes >= m go nowhere; then the same assembler
    on a second matrix of the same pattern (the row destinations are re-based)"""
    m = levels((2, 2, 1), 2)[-1]
    ed, xy, _ = m.arrays()
    rng = np.random.default_rng(21)
    xy = xy + rng.uniform(-0.01, 0.01, xy.shape)
    n = m.nnode
    mrows = (2 * n) // 3
    rp, col = capi.pattern_from_elements(ed, n)
    rp_o, col_o = rp[:mrows + 1].copy(), col[:rp[mrows]].copy()
    u = rng.uniform(-1, 1, n)
    Ao, bo = oracle_global(ed, xy, u, KINDS[2][1], n)
    A = ctx.matrix_csr(mrows, n, rp_o, col_o)
    B = ctx.matrix_csr(mrows, n, rp_o, col_o)
    res = ctx.vector(n)
    asm = capi.Assembler(ctx, m, "biquadratic", A, elem_dof=ed, coords=xy)
    assert asm.fused_info()["active"]
    for M in (A, B, A):
        asm.assemble(M, res, ctx.vector_from(u), 2, (2.0, 1.3))
        assert abs(M.values() - Ao.data[:rp[mrows]]).max() <= 1e-12 * abs(Ao.data).max()
        assert abs(res.to_numpy()[:mrows] - bo[:mrows]).max() <= 1e-12 * abs(bo).max()
    asm.destroy(), A.destroy(), B.destroy()


def test_elementwise_galerkin_after_a_fused_assembly(ctx):
    """the fused path keeps no element rows: the element-wise Galerkin product is made from the MACRO rows it left behind (complete rows in the matrix,
    the others in the partial-row buffer: k_galerkin_macro) and gives the same coarse operators as after a two-pass assembly -- also after SetPenalty has
    replaced the Dirichlet rows of the fine matrix, and with option galerkin_macro 0 (the element rows re-created by pass 1 of the two-pass path)"""
    from femus_amd.poisson import PoissonMG
    vals = {}
    for fused, macro in ((1, 1), (1, 0), (0, 1)):
        ctx.set_option("assemble_fused", fused)
        ctx.set_option("galerkin_macro", macro)
        try:
            pb = PoissonMG(ctx, 2, 2, 2, 3).init()
            assert pb.asm[-1].fused_info()["active"] == bool(fused)
            pb.assemble()
            pb.prepare()
            vals[(fused, macro)] = [pb.A[l].to_scipy() for l in range(pb.nlevels)]
            pb.prepare()                                  # again, from the fine matrix whose Dirichlet rows SetPenalty has replaced meanwhile
            for a, b in zip([pb.A[l].to_scipy() for l in range(pb.nlevels)], vals[(fused, macro)]):
                assert abs(a - b).max() == 0.0
            pb.destroy()
        finally:
            ctx.set_option("assemble_fused", 1)
            ctx.set_option("galerkin_macro", 1)
    for key in ((1, 1), (1, 0)):
        for a, b in zip(vals[key], vals[(0, 1)]):
            assert abs(a - b).max() <= 1e-13 * abs(b).max()
    # the path of an assembler does not change inside a run: with the macro rows serving the product, assemble_fused 1 (default) and 2 stay fused through
    # assemble -> prepare -> assemble (the flow of every MGsolve, LinearImplicitSystem.cpp:288-411); only with galerkin_macro 0 does mode 1 fall back to
    # the two-pass path after a product has asked for the element rows
    for mode, macro in ((1, 1), (2, 1), (1, 0), (2, 0)):
        ctx.set_option("assemble_fused", mode)
        ctx.set_option("galerkin_macro", macro)
        try:
            pb = PoissonMG(ctx, 2, 2, 2, 3).init()
            pb.assemble()
            assert pb.asm[-1].last_path() == "fused"
            pb.prepare()
            pb.assemble()
            assert pb.asm[-1].last_path() == ("two-pass" if (mode, macro) == (1, 0) else "fused")
            pb.assemble()
            assert pb.asm[-1].last_path() == "fused"            # nobody asked for the rows in between
            pb.prepare()
            for a, b in zip([pb.A[l].to_scipy() for l in range(pb.nlevels)], vals[(0, 1)]):
                assert abs(a - b).max() <= 1e-13 * abs(b).max()
            pb.destroy()
        finally:
            ctx.set_option("assemble_fused", 1)
            ctx.set_option("galerkin_macro", 1)


@pytest.mark.parametrize("args,nl,carry,kind,with_sol", [((2, 2, 2), 3, 3, 1, True), ((2, 2, 2), 3, 6, 0, True), ((3, 2, 2), 3, 6, 2, False), ((1, 1, 1), 3, 3, 1, True),
                                                         ((4, 2, 2), 2, 3, 1, True)])
def test_carried_rows_have_the_bits_of_the_partial_row_buffer(ctx, args, nl, carry, kind, with_sol):
    """assemble_carry: rows all of whose elements lie in one super-cluster (8 / 64 / -- where the cluster count is no multiple -- 32 consecutive clusters) are
    accumulated in the CSR array by the one workgroup that walks the super-cluster: store, then load-add-store in ascending cluster order, which is the order
    of the second pass -- matrix and residual are BIT-IDENTICAL to the plan without carried rows, from NaN-poisoned arrays, twice; fewer entries visit the
    partial-row buffer; and both equal the oracle's element loop to 1e-12.  (4, 2, 2) refined once: consecutive clusters that are no siblings of each other."""
    m = levels(args, nl)[-1]
    ed, xy, _ = m.arrays()
    rng = np.random.default_rng(5)
    xy = xy + rng.uniform(-0.01, 0.01, xy.shape) / 2 ** (nl - 1)
    n = m.nnode
    rp, col = capi.pattern_from_elements(ed, n)
    u = rng.uniform(-1, 1, n) if with_sol else np.zeros(n)
    params, rhs = KINDS[kind]
    out, info = {}, {}
    for c in (0, carry):
        ctx.set_option("assemble_carry", c)
        ctx.set_option("debug_poison", 1)
        try:
            A = ctx.matrix_csr(n, n, rp, col, np.full(col.size, np.nan))
            res = ctx.vector_from(np.full(n, np.nan))
            asm = capi.Assembler(ctx, m, "biquadratic", A, elem_dof=ed, coords=xy)
            info[c] = asm.fused_info()
            assert info[c]["active"]
            sol = ctx.vector_from(u) if with_sol else None
            asm.assemble(A, res, sol, kind, params)
            v1, f1 = A.values().copy(), res.to_numpy().copy()
            A.set_values(np.full(col.size, np.nan))
            asm.assemble(A, res, sol, kind, params)
            assert np.array_equal(v1, A.values()) and np.array_equal(f1, res.to_numpy())
            out[c] = (v1, f1)
            asm.destroy(), A.destroy()
        finally:
            ctx.set_option("debug_poison", 0)
            ctx.set_option("assemble_carry", -1)
    assert info[0]["clusters_per_super"] == 1 and info[0]["carried_entries"] == 0
    assert info[carry]["clusters_per_super"] > 1 and info[carry]["carried_entries"] > 0
    assert info[carry]["partial_entries"] < info[0]["partial_entries"] and info[carry]["second_pass_rows"] < info[0]["second_pass_rows"]
    assert np.isfinite(out[carry][0]).all() and np.isfinite(out[carry][1]).all()
    assert np.array_equal(out[0][0], out[carry][0]) and np.array_equal(out[0][1], out[carry][1])
    Ao, bo = oracle_global(ed, xy, u, rhs, n)
    row_scale = np.repeat(np.maximum.reduceat(abs(Ao.data), rp[:-1]), np.diff(rp))
    assert (abs(out[carry][0] - Ao.data) / row_scale).max() <= 1e-12
    assert abs(out[carry][1] - bo).max() <= 1e-12 * abs(bo).max()


def test_carried_rows_with_a_pattern_that_holds_more_than_the_element_couplings(ctx):
    """a CSR row with a position no element contributes to cannot be carried (nobody would store there): the plan is made again without carried rows (the second
    pass writes whole rows) and gives the oracle's operator with zeros at the extra positions"""
    m = levels((2, 2, 2), 3)[-1]
    ed, xy, _ = m.arrays()
    n = m.nnode
    rp, col = capi.pattern_from_elements(ed, n)
    P = sp.csr_matrix((np.ones(col.size), col, rp), shape=(n, n))
    extra = sp.coo_matrix((np.ones(n - 1), (np.arange(n - 1), (np.arange(n - 1) * 7 + 3) % n)), shape=(n, n)).tocsr()
    P = (P + extra).tocsr()
    P.sort_indices()
    ctx.set_option("assemble_carry", 3)
    try:
        A = ctx.matrix_csr(n, n, P.indptr.astype(np.int32), P.indices.astype(np.int32), np.full(P.nnz, np.nan))
        res = ctx.vector(n)
        asm = capi.Assembler(ctx, m, "biquadratic", A, elem_dof=ed, coords=xy)
        fi = asm.fused_info()
        assert fi["active"] and fi["clusters_per_super"] == 1
        u = np.random.default_rng(2).uniform(-1, 1, n)
        asm.assemble(A, res, ctx.vector_from(u), 1, (2.0, 1.3))
        Ao, bo = oracle_global(ed, xy, u, KINDS[1][1], n)
        got = sp.csr_matrix((A.values(), P.indices, P.indptr), shape=(n, n))
        assert np.isfinite(got.data).all()
        assert abs(got - Ao).max() <= 1e-12 * abs(Ao.data).max()
        assert abs(res.to_numpy() - bo).max() <= 1e-12 * abs(bo).max()
        asm.destroy(), A.destroy()
    finally:
        ctx.set_option("assemble_carry", -1)


@pytest.mark.parametrize("carry", [3, 6])
def test_elementwise_galerkin_from_carried_rows(ctx, carry):
    """k_galerkin_macro reads a carried entry in the cluster that made the last contribution (the sum of all of them): the pseudo child matrices still add
    up to the assembled operator, so the coarse operators equal the ones made without carried rows to rounding -- also after SetPenalty"""
    from femus_amd.poisson import PoissonMG
    vals = {}
    for c in (0, carry):
        ctx.set_option("assemble_carry", c)
        try:
            pb = PoissonMG(ctx, 2, 2, 2, 4).init()
            fi = pb.asm[-1].fused_info()
            assert fi["active"] and (fi["clusters_per_super"] > 1) == (c > 0)
            pb.assemble()
            pb.prepare()
            assert pb.asm[-1].last_path() == "fused"
            vals[c] = [pb.A[l].to_scipy() for l in range(pb.nlevels)]
            pb.prepare()
            for a, b in zip([pb.A[l].to_scipy() for l in range(pb.nlevels)], vals[c]):
                assert abs(a - b).max() == 0.0
            pb.destroy()
        finally:
            ctx.set_option("assemble_carry", -1)
    assert abs(vals[0][-1] - vals[carry][-1]).max() == 0.0
    for a, b in zip(vals[0][:-1], vals[carry][:-1]):
        assert abs(a - b).max() <= 1e-13 * abs(b).max()


def test_galerkin_product_does_not_read_macro_rows_somebody_else_has_written(ctx):
    """the macro rows of a fused assembly live in the user-visible fine matrix: after any writer of its values other than SetPenalty (here: the values
    replaced by twice themselves) the element-wise Galerkin product goes back to the element rows of the assembly -- the coarse operators are the ones of
    the assembled operator, not of what the matrix holds now (fh_mat_s::val_gen)"""
    from femus_amd.poisson import PoissonMG
    pb = PoissonMG(ctx, 2, 2, 2, 3).init()
    pb.assemble()
    pb.prepare()
    ref = [pb.A[l].to_scipy() for l in range(pb.nlevels - 1)]
    pb.assemble()
    assert pb.asm[-1].last_path() == "fused"
    top = pb.A[-1]
    top.set_values(2.0 * top.values())
    pb.level_operators()
    for l in range(pb.nlevels - 1):
        assert abs(pb.A[l].to_scipy() - ref[l]).max() <= 1e-13 * abs(ref[l]).max()
    pb.destroy()
