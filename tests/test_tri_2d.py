"""Triangles (TRI6 box of applications/001_Poisson's generator, TRI7 inside FEMuS).  CPU: the oracle restatement (oracle/femus_oracle_tri.py) -- basis against the
fixture of the reference's compiled classes, mesh and refinement properties, convergence to a manufactured solution.  GPU: the generic (dim, nc, ng) kernel
fh_assemble_poisson_rows against the oracle entry for entry on triangles, and against the tensor-product oracle on quadrilaterals and hexahedra."""
import os

import numpy as np
import pytest


def cross2(a, b):
    return a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]

from oracle import femus_oracle as fo
from oracle import femus_oracle_tri as ot

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fe_tables.npz"))
FES = ["linear", "serendipity", "biquadratic"]


def src(x):
    return 2 * np.pi ** 2 * np.sin(np.pi * x[0]) * np.sin(np.pi * x[1])


@pytest.mark.parametrize("fe", FES)
def test_oracle_triangle_basis_is_the_reference_s(fe):
    for tag, pts in (("sample", G["sample_pts_tri"]), ("gauss7", G["gauss_x_tri_seventh"])):
        ref = G["basis_tri_%s_%s" % (fe, tag)]
        phi, dphi = ot.basis(fe, pts)
        assert np.abs(phi - ref[0]).max() < 4e-15 and np.abs(dphi[:, :, 0] - ref[1]).max() < 4e-15 and np.abs(dphi[:, :, 1] - ref[2]).max() < 4e-15
    assert np.array_equal(ot.XC, G["xc_tri"]) and np.array_equal(ot.F2C, G["f2c_tri"]) and np.array_equal(ot.FACE, G["facedofs_tri"])


def test_oracle_triangle_box_and_its_refinement():
    ed, xs, ff, own = ot.box_mesh(3, 2, (0., 0.), (3., 1.))
    assert ed.shape == (12, 7) and own == [12, 12 + 23, 12 + 23 + 12] and xs.shape[0] == own[2]
    assert np.allclose(xs[ed[:, 3]], 0.5 * (xs[ed[:, 0]] + xs[ed[:, 1]])) and np.allclose(xs[ed[:, 6]], xs[ed[:, :3]].mean(axis=1))
    assert sorted(set(ff.ravel().tolist())) == [-5, -4, -3, -2, -1] and (ff == -2).sum() == 3 and (ff == -3).sum() == 2
    area = 0.5 * cross2(xs[ed[:, 1]] - xs[ed[:, 0]], xs[ed[:, 2]] - xs[ed[:, 0]])
    assert np.isclose(area.sum(), 3.0) and (area > 0).all()       # counter-clockwise
    ef, xf, fff, ownf = ot.refine(ed, xs, ff)
    assert ef.shape == (48, 7) and ownf[0] == own[1] and xf.shape[0] == ownf[2]
    af = 0.5 * cross2(xf[ef[:, 1]] - xf[ef[:, 0]], xf[ef[:, 2]] - xf[ef[:, 0]])
    assert (af > 0).all() and np.isclose(af.sum(), 3.0) and np.allclose(xf[ef[:, 4]], 0.5 * (xf[ef[:, 1]] + xf[ef[:, 2]]))
    assert (fff == -2).sum() == 6 and (fff == -5).sum() == 4
    for f, (a, b) in enumerate(((0, 1), (1, 2), (2, 0))):          # a flagged child edge lies on the boundary it is named after
        for e in np.where(fff[:, f] == -2)[0]:
            assert xf[ef[e, a], 1] == 0.0 and xf[ef[e, b], 1] == 0.0


@pytest.mark.parametrize("fe,rate", [("linear", 2.0), ("serendipity", 3.0), ("biquadratic", 3.0)])
def test_oracle_triangles_converge_to_a_manufactured_solution(fe, rate):
    errs = []
    for nl in (1, 2, 3):
        u, meshes = ot.solve(2, 2, nl, fe, src)
        x = meshes[-1][1][:u.size]
        errs.append(np.abs(u - np.sin(np.pi * x[:, 0]) * np.sin(np.pi * x[:, 1])).max())
    assert np.log2(errs[1] / errs[2]) > rate - 0.3, errs


gpu = pytest.mark.gpu


def _pattern(ctx, ed, nc, ndof):
    from femus_amd import capi
    pairs = sorted({(int(a), int(b)) for e in ed for a in e[:nc] for b in e[:nc]})
    rows = np.array([p[0] for p in pairs])
    indptr = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=ndof))])
    return capi.Mat.from_csr(ctx, ndof, ndof, indptr, np.array([p[1] for p in pairs]))


@gpu
@pytest.mark.parametrize("fe", FES)
def test_generic_kernel_on_triangles_matches_the_oracle(ctx, fe):
    """fh_assemble_poisson_rows on a refined, distorted triangle mesh at a non-trivial state: K and the residual against the oracle's loops (1e-12)"""
    from femus_amd import capi
    ed, xs, ff, own = ot.box_mesh(3, 2, (0., 0.), (1.5, 1.))
    ed, xs, ff, own = ot.refine(ed, xs, ff)
    xs = xs + 0.03 * np.stack([np.sin(3 * xs[:, 1]) * xs[:, 0] * (1.5 - xs[:, 0]), np.sin(2 * xs[:, 0]) * xs[:, 1] * (1 - xs[:, 1])], axis=1)   # curved edges inside
    nc, ndof = ot.NDOF[fe], ot.n_dofs(own, fe)
    u = np.random.default_rng(7).uniform(-1, 1, ndof)
    Ko, Fo = ot.assemble(ed, xs, fe, lambda x: np.exp(x[0]) * (1 + x[1]), u)
    K = _pattern(ctx, ed, nc, ndof)
    RES, SOL = ctx.vector(ndof), ctx.vector_from(u)
    f = capi.Expr("exp(x)*(1+y)", "x,y,z,t")
    for rep in range(2):
        capi.assemble_poisson_rows(ctx, "tri", fe, ed, xs, K, RES, sol=SOL, source=f)
    assert np.abs(K.to_scipy().toarray() - Ko).max() <= 1e-12 * np.abs(Ko).max()
    assert np.abs(RES.to_numpy() - Fo).max() <= 1e-12 * np.abs(Fo).max()
    f.destroy()
    K.destroy()


@gpu
@pytest.mark.parametrize("box,geom,fe", [((3, 2, 0), "quad", "biquadratic"), ((3, 2, 0), "quad", "linear"), ((2, 2, 2), "hex", "serendipity"), ((2, 1, 2), "hex", "biquadratic")])
def test_generic_kernel_on_tensor_product_elements_matches_their_oracle(ctx, box, geom, fe):
    """the same kernel on QUAD9 / HEX27 meshes (curved) against femus_oracle.assemble_poisson: K to 1e-12 -- two implementations of the element loop, one oracle"""
    from femus_amd import capi
    m = fo.coarse_box_mesh(*box)
    m = fo.refine(m)
    rng = np.random.default_rng(2)
    m.coords = m.coords + 0.02 * rng.uniform(-1, 1, m.coords.shape) * (np.abs(m.coords - 0.5) < 0.45).all(axis=1, keepdims=True)
    A, b = fo.assemble_poisson(m, fe, lambda xg: np.zeros(xg.shape[:2]))
    ndof = fo.n_dofs(m, fe)
    nc = fo.ndofs(geom, fe)
    K = _pattern(ctx, m.elem_dof, nc, ndof)
    RES = ctx.vector(ndof)
    capi.assemble_poisson_rows(ctx, geom, fe, m.elem_dof, m.coords, K, RES)
    Ad = A.toarray()
    assert np.abs(K.to_scipy().toarray() - Ad).max() <= 1e-12 * np.abs(Ad).max()
    K.destroy()


TRI_INPUT = """
{
    "multilevel_mesh" : { "first" : { "type" : { "box" : { "nx" : 4, "ny" : 3, "nz" : 0, "xa" : 0., "xb" : 1., "ya" : 0., "yb" : 1., "za" : 0., "zb" : 0.,
                                                            "elem_type" : "Tri6" } } } },
    "multilevel_solution" : { "multilevel_mesh" : { "first" : { "variable" : { "first" : {
              "name" : "T", "fe_order" : "second", "init_func" : "0.", "func_source": "1. + x*y",
              "boundary_conditions" : [ { "facename" : "left", "bdc_type" : "dirichlet", "bdc_func" : "0.5+1./pi*atan(10.*(y-0.8))" },
                                        { "facename" : "top", "bdc_type" : "dirichlet", "bdc_func" : "1." },
                                        { "facename" : "bottom", "bdc_type" : "neumann", "bdc_func" : "0." },
                                        { "facename" : "right", "bdc_type" : "neumann", "bdc_func" : "0.3*y" } ] } } } } },
    "multilevel_problem" : { "multilevel_mesh" : { "first" : { "system" : { "poisson" : { "linear_solver" : {
                "max_number_linear_iteration" : 6, "abs_conv_tol" : 1.e-09,
                "type" : { "multigrid" : { "nlevels" : 3, "npresmoothing" : 1, "npostsmoothing" : 1, "mgtype" : "V_cycle" } } } } } } } }
}
"""


@gpu
@pytest.mark.parametrize("fe_order,fe", [("second", "biquadratic"), ("first", "linear"), ("serendipity", "serendipity")])
def test_the_application_on_a_box_of_triangles(ctx, fe_order, fe):
    """applications/001_Poisson with the boundary conditions of its shipped input.json (a parsed Dirichlet profile on "left", 1 on "top", fluxes on the others;
    the profile softened) on a TRI6 box, three levels: meshes equal to the oracle's (integers; coordinates to rounding), converged under the input's limits,
    and -- iterated on -- the direct solve of the finest level's problem to 1e-10"""
    from femus_amd import app_poisson as app
    cfg = app.load_config(TRI_INPUT)
    cfg["multilevel_solution"]["multilevel_mesh"]["first"]["variable"]["first"]["fe_order"] = fe_order
    p = app.Poisson001(ctx, cfg)
    assert p.tri and p.fe == fe and p.nlevels == 3
    out = p.run()
    ref, meshes = ot.solve(4, 3, 3, fe, lambda x: 1. + x[0] * x[1], dirichlet_flags=(-5, -4), flux_by_flag={-3: lambda x: 0.3 * x[1], -2: lambda x: 0.0},
                           values=lambda x: 0.5 + 1. / np.pi * np.arctan(10. * (x[1] - 0.8)) if x[0] == 0.0 else 1.0)      # the corner: "left" is face 2 of its element, after "top" (face 1)
    for (ed_p, xs_p, ff_p), (ed_o, xs_o, ff_o, _) in zip(out["levels"], meshes):
        assert np.array_equal(ed_p, ed_o) and np.array_equal(ff_p, ff_o) and np.abs(xs_p - xs_o).max() < 1e-14
    assert out["dofs"] == ref.size and out["converged"] and len(out["history"]) <= 7, out["history"]
    assert np.abs(out["solution"] - ref).max() < 1e-8
    p.max_linear, p.abs_tol = 40, 1e-13
    out = p.run()
    assert out["converged"] and np.abs(out["solution"] - ref).max() < 1e-10
    p.destroy()
