"""The one-dimensional input of applications/001_Poisson (input/input1D.json: an EDGE3 box, where the callback is advection-diffusion with its streamline-upwind
terms, main.cpp:392-395).  CPU: the oracle restatement (oracle/femus_oracle_1d.py) against the analytic solution of the boundary-value problem and against
itself under refinement.  GPU: fh_assemble_advdiff_line against the oracle entry for entry, and the shipped input through app_poisson against the oracle's
solve."""
import os

import numpy as np
import pytest

from oracle import femus_oracle_1d as o1

NU, V = 0.01, 1.0


def source(x):
    return 10. * np.exp(-5. * x) - 4. * np.exp(-x)          # "func_source" of input1D.json


def analytic(x):
    """-nu u'' + V u' = 10 exp(-5 x) - 4 exp(-x), u(0) = 0, u'(1) = 0: particular solutions A exp(-k x) with A = -a / (nu k^2 + V k), plus c1 + c2 exp(x / nu)"""
    A1, A2 = -10. / (NU * 25. + 5. * V), 4. / (NU + V)
    c2e = NU * (5. * A1 * np.exp(-5.) + A2 * np.exp(-1.))      # c2 exp(1 / nu), from u'(1) = 0
    c1 = -(A1 + A2) - c2e * np.exp(-1. / NU)
    return A1 * np.exp(-5. * x) + A2 * np.exp(-x) + c1 + c2e * np.exp((x - 1.) / NU)


def test_oracle_mesh_is_the_edge3_box_with_vertices_numbered_first():
    ed, xs, face, nv = o1.box_mesh(4, 0.0, 2.0)
    assert nv == 5 and np.allclose(xs[:nv], [0.0, 0.5, 1.0, 1.5, 2.0]) and np.allclose(xs[nv:], [0.25, 0.75, 1.25, 1.75])
    assert ed.tolist() == [[0, 1, 5], [1, 2, 6], [2, 3, 7], [3, 4, 8]]
    assert face[0].tolist() == [-2, -1] and face[-1].tolist() == [-1, -3]


@pytest.mark.parametrize("fe,rate", [("biquadratic", 3.0), ("linear", 1.5)])
def test_oracle_converges_to_the_analytic_solution(fe, rate):
    """the stabilised form is consistent: the nodal error at the vertices falls with the mesh size (measured rates: > 3 for EDGE3, > 1.5 for the linear family)"""
    errs = []
    for nx in (20, 40, 80, 160):
        u, x, _ = o1.solve(nx, fe, source)
        errs.append(np.abs(u[:nx + 1] - analytic(x[:nx + 1])).max())
    assert errs[-1] < 2e-4 and all(np.log2(errs[k] / errs[k + 1]) > rate - 0.6 for k in range(2, 3)), errs
    u, x, _ = o1.solve(10, "biquadratic", source)                      # the shipped size: ten elements
    assert abs(u[np.argmax(x)] - analytic(1.0)) < 1e-3


def test_oracle_without_velocity_is_the_poisson_form():
    """V = 0: tau = 0 and the loop is the Laplace form of the 2-D / 3-D kernels (stiffness nu phi_i' phi_j', load f phi_i)"""
    ed, xs, face, nv = o1.box_mesh(5)
    K, F = o1.assemble(ed, xs, "biquadratic", np.zeros(xs.size), lambda x: 1.0, nu=1.0, V=0.0)
    assert np.allclose(K, K.T) and np.allclose(K.sum(axis=1), 0.0, atol=1e-12) and np.isclose(F.sum(), 1.0)


def test_oracle_refinement_of_the_edge3_box():
    """children at the father's vertices, new middles at the quarter points, flags inherited by the end children, vertices numbered first"""
    ed, xs, face, nv = o1.box_mesh(3, 0.0, 3.0)
    ef, xf, ff, nvf = o1.refine(ed, xs, face)
    assert nvf == 7 and np.allclose(np.sort(xf[:nvf]), np.arange(7) * 0.5) and np.allclose(np.sort(xf[nvf:]), 0.25 + 0.5 * np.arange(6))
    assert np.allclose(xf[ef[:, 2]], 0.5 * (xf[ef[:, 0]] + xf[ef[:, 1]])) and np.all(xf[ef[:, 0]] < xf[ef[:, 1]])
    assert ff[0].tolist() == [-2, -1] and ff[-1].tolist() == [-1, -3] and (ff[1:-1] == -1).all()
    u2, meshes = o1.solve_levels(10, 2, "biquadratic", source)
    u1, x1, _ = o1.solve(20, "biquadratic", source)            # the refined ten-element box IS the twenty-element box, numbered differently
    xs2 = meshes[-1][1]
    assert np.allclose(u2[np.argsort(xs2)], u1[np.argsort(x1)], atol=1e-12)


gpu = pytest.mark.gpu


@gpu
@pytest.mark.parametrize("fe", ["biquadratic", "linear", "serendipity"])
@pytest.mark.parametrize("nx", [10, 37])
def test_device_assembly_matches_the_oracle(ctx, fe, nx):
    """fh_assemble_advdiff_line: Jacobian and residual at a non-trivial state, every entry against the loops of the oracle (1e-12), on a stretched mesh"""
    from femus_amd import capi
    ed, xs, face, nv = o1.box_mesh(nx, -0.3, 1.7)
    xs = xs + 0.02 * np.sin(3.0 * xs)                                   # elements of different lengths, middles off centre (the quadratic map)
    ofe = "linear" if fe == "linear" else "biquadratic"
    nc = 2 if fe == "linear" else 3
    ndof = nv if fe == "linear" else xs.size
    rng = np.random.default_rng(3)
    u = rng.uniform(-1, 1, ndof)
    Ko, Fo = o1.assemble(ed, xs, ofe, u, source, NU, V)
    pairs = sorted({(int(a), int(b)) for e in ed for a in e[:nc] for b in e[:nc]})
    rows = np.array([p[0] for p in pairs])
    indptr = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=ndof))])
    K = capi.Mat.from_csr(ctx, ndof, ndof, indptr, np.array([p[1] for p in pairs]))
    RES, SOL = ctx.vector(ndof), ctx.vector_from(u)
    src = capi.Expr("10.*exp(-5.*x) - 4.*exp(-x)", "x,y,z,t")
    for rep in range(2):                                                # overwritten, not accumulated
        capi.assemble_advdiff_line(ctx, fe, ed, xs, K, RES, NU, V, sol=SOL, source=src)
    Kd = K.to_scipy().toarray()
    assert np.abs(Kd - Ko).max() <= 1e-12 * np.abs(Ko).max()
    assert np.abs(RES.to_numpy() - Fo).max() <= 1e-12 * max(np.abs(Fo).max(), 1.0)
    src.destroy()
    K.destroy()


SHIPPED_1D_INPUT = """
{
    "multilevel_mesh" : { "first" : { "type" : { "box" : { "nx" : 10, "ny" : 0, "nz" : 0, "xa" : 0., "xb" : 1., "ya" : 0., "yb" : 0., "za" : 0., "zb" : 0.,
                                                            "elem_type" : "Edge3" } } } },
    "multilevel_solution" : { "multilevel_mesh" : { "first" : { "variable" : { "first" : {
              "name" : "T", "fe_order" : "second", "init_func" : "0.", "func_source": "10.*exp(-5.*x) - 4.*exp(-x)",
              "boundary_conditions" : [ { "facename" : "left", "bdc_type" : "dirichlet" }, { "facename" : "right", "bdc_type" : "neumann" } ] } } } } },
    "multilevel_problem" : { "multilevel_mesh" : { "first" : { "system" : { "poisson" : { "linear_solver" : {
                "max_number_linear_iteration" : 6, "abs_conv_tol" : 1.e-09,
                "type" : { "multigrid" : { "nlevels" : 1, "npresmoothing" : 1, "npostsmoothing" : 1, "mgtype" : "V_cycle",
                    "smoother" : { "type" : { "gmres" : { "ksp" : "gmres", "precond" : "ilu", "rtol" : 1.e-12, "atol" : 1.e-20, "divtol" : 1.e+50,
                                                          "max_its" : 4 } } } } } } } } } } }
}
"""


def test_the_configuration_below_is_the_shipped_file():
    from femus_amd import app_poisson as app
    ref_file = "/root/reference/applications/001_Poisson/input/input1D.json"
    if not os.path.exists(ref_file):
        pytest.skip("the reference tree is not here")
    assert app.load_config(ref_file) == app.load_config(SHIPPED_1D_INPUT)


@gpu
def test_the_shipped_one_dimensional_input_of_001_poisson(ctx):
    """applications/001_Poisson/input/input1D.json through app_poisson on the GPU (mesh, numbering, callback form, boundary rows, exact one-level solve)
    against the oracle's solve of the same discrete problem (1e-10) and, loosely, the analytic solution"""
    from femus_amd import app_poisson as app
    p = app.Poisson001(ctx, SHIPPED_1D_INPUT)
    assert p.dim == 1 and p.fe == "biquadratic" and p.nlevels == 1 and p.box[0] == 10
    out = p.run()
    assert out["converged"] and len(out["history"]) <= 6
    ed, xs, face, nv = o1.box_mesh(10)
    assert np.array_equal(out["elem_dof"], ed) and np.array_equal(out["nodes"], xs)
    ref, x, _ = o1.solve(10, "biquadratic", source)
    assert out["dofs"] == 21 and abs(ref).max() > 0.5
    assert np.abs(out["solution"] - ref).max() < 1e-10
    assert np.abs(out["solution"][:nv] - analytic(xs[:nv])).max() < 2e-3
    p.destroy()


@gpu
@pytest.mark.parametrize("fe", ["second", "first"])
def test_one_dimensional_input_with_boundary_values_and_a_flux(ctx, fe):
    """the same application with a Dirichlet VALUE on "left" and a parsed flux on "right" (the point term of main.cpp:540-549), on another interval and mesh,
    both families: against the oracle's solve"""
    from femus_amd import app_poisson as app
    cfg = app.load_config(SHIPPED_1D_INPUT)
    box = cfg["multilevel_mesh"]["first"]["type"]["box"]
    box["nx"], box["xa"], box["xb"] = 23, -0.5, 1.5
    var = cfg["multilevel_solution"]["multilevel_mesh"]["first"]["variable"]["first"]
    var["fe_order"] = fe
    var["boundary_conditions"] = [{"facename": "left", "bdc_type": "dirichlet", "bdc_func": "0.75 + x"}, {"facename": "right", "bdc_type": "neumann", "bdc_func": "0.02 * x"}]
    p = app.Poisson001(ctx, cfg)
    out = p.run()
    assert out["converged"]
    ofe = "linear" if fe == "first" else "biquadratic"
    ref, x, _ = o1.solve(23, ofe, source, dirichlet_left=0.75 - 0.5, xa=-0.5, xb=1.5, flux_right=lambda x: 0.02 * x)
    assert out["dofs"] == ref.size and np.abs(out["solution"] - ref).max() < 1e-10 * max(1.0, np.abs(ref).max())
    assert abs(out["solution"][0] - 0.25) < 1e-14
    p.destroy()


@gpu
@pytest.mark.parametrize("fe,nlevels", [("second", 3), ("first", 3), ("second", 4)])
def test_one_dimensional_input_on_several_levels(ctx, fe, nlevels):
    """input1D.json with "nlevels" raised: EDGE3 refinement (meshes equal to the oracle's, integers and coordinates), Galerkin operators, V-cycles with the
    natural-order sweep under GMRES(4): converges under the input's own limits to the direct solve of the finest level's problem"""
    from femus_amd import app_poisson as app
    cfg = app.load_config(SHIPPED_1D_INPUT)
    cfg["multilevel_solution"]["multilevel_mesh"]["first"]["variable"]["first"]["fe_order"] = fe
    cfg["multilevel_problem"]["multilevel_mesh"]["first"]["system"]["poisson"]["linear_solver"]["type"]["multigrid"]["nlevels"] = nlevels
    p = app.Poisson001(ctx, cfg)
    out = p.run()
    ofe = "linear" if fe == "first" else "biquadratic"
    ref, meshes = o1.solve_levels(10, nlevels, ofe, source)
    for (ed_p, xs_p), (ed_o, xs_o, _, _) in zip(out["levels"], meshes):
        assert np.array_equal(ed_p, ed_o) and np.allclose(xs_p, xs_o, rtol=0, atol=1e-15)
    assert out["converged"] and len(out["history"]) <= 7, out["history"]
    assert np.abs(out["solution"] - ref).max() < 1e-8                       # what ||RES|| < 1e-9 leaves
    p.max_linear, p.abs_tol = 30, 1e-13
    out = p.run()
    assert out["converged"] and np.abs(out["solution"] - ref).max() < 1e-10
    p.destroy()
