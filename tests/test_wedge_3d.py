"""Prisms (the WEDGE18 meshes of applications/001_Poisson: input3D_Wedge_first / _second / _serendipity.json with input/cube_Wedge.neu, a data file of the
application kept in tests/golden).  CPU: the oracle restatement (oracle/femus_oracle_wedge.py) -- basis against the fixture of the reference's compiled classes,
reader / added nodes / refinement properties, the product's host-side mesh code equal to it.  GPU: the generic kernel and the face integrals (quadrilaterals and
triangles) against the oracle, and the shipped inputs through app_poisson against the oracle's direct solve."""
import os

import numpy as np
import pytest

from oracle import femus_oracle_wedge as ow

HERE = os.path.dirname(os.path.abspath(__file__))
MESH = os.path.join(HERE, "golden", "cube_Wedge.neu")
G = np.load(os.path.join(HERE, "golden", "fe_tables.npz"))
gpu = pytest.mark.gpu
FES = ["linear", "serendipity", "biquadratic"]


def volume(ed, xs):
    w, xg = ow.gauss("third")
    P, D = ow.basis("biquadratic", xg)
    return sum(np.linalg.det(D[g].T @ xs[ed[e]]) * w[g] for e in range(ed.shape[0]) for g in range(w.size))


@pytest.mark.parametrize("fe", FES)
def test_oracle_prism_basis_is_the_reference_s(fe):
    for tag, pts in (("sample", G["sample_pts_wedge"]), ("gauss7", G["gauss_x_wedge_seventh"])):
        ref = G["basis_wedge_%s_%s" % (fe, tag)]
        phi, dphi = ow.basis(fe, pts)
        assert np.abs(phi - ref[0]).max() < 4e-15 and max(np.abs(dphi[:, :, d] - ref[1 + d]).max() for d in range(3)) < 4e-15
    assert np.array_equal(ow.XC, G["xc_wedge"]) and np.array_equal(ow.F2C, G["f2c_wedge"])
    assert all(list(G["facedofs_wedge"][f][:len(ow.FACE[f])]) == ow.FACE[f] for f in range(5))


def test_the_mesh_file_is_the_application_s():
    ref_file = "/root/reference/applications/001_Poisson/input/cube_Wedge.neu"
    if not os.path.exists(ref_file):
        pytest.skip("the reference tree is not here")
    assert open(ref_file, "rb").read() == open(MESH, "rb").read()


def test_oracle_reader_and_refinement_and_the_product_s_mesh_code():
    """cube_Wedge.neu: 16 WEDGE18 prisms filling the unit cube, the triangle-face nodes shared by stacked prisms and the centres added (WEDGE21: 165 nodes);
    refined: eight times the elements, the same volume, four times the faces per set; femus_amd/wedge_mesh.py gives the same integers on three levels"""
    from femus_amd import wedge_mesh
    ed, xs, ff, own = ow.read_gambit(MESH)
    assert ed.shape == (16, 21) and own == [27, 93, 165] and np.isclose(volume(ed, xs), 1.0)
    for m, (a, b) in enumerate(ow.EDGE):
        assert np.allclose(xs[ed[:, 6 + m]], 0.5 * (xs[ed[:, a]] + xs[ed[:, b]]))
    assert np.allclose(xs[ed[:, 18]], xs[ed[:, :3]].mean(axis=1)) and np.allclose(xs[ed[:, 20]], xs[ed[:, :6]].mean(axis=1))
    counts = [(ff == f).sum() for f in range(-7, -1)]
    assert sum(counts) == 32 and min(counts) == 4          # four quadrilateral sets of 4, two triangle sets of 8
    a, b = wedge_mesh.read_gambit(MESH), (ed, xs, ff, own)
    for level in range(3):
        assert np.array_equal(a[0], b[0]) and np.abs(a[1] - b[1]).max() < 1e-14 and np.array_equal(a[2], b[2]) and a[3] == b[3]
        if level == 2:
            break
        a, b = wedge_mesh.refine(*a[:3]), ow.refine(*b[:3])
        ef, xf, fff, _ = b
        assert ef.shape[0] == 16 * 8 ** (level + 1) and [(fff == f).sum() for f in range(-7, -1)] == [c * 4 ** (level + 1) for c in counts]
        if level == 0:
            assert np.isclose(volume(ef, xf), 1.0)
        for e, f in zip(*np.nonzero(fff < -1)):
            x = xf[ef[e, ow.FACE[f][:3]]]
            assert any(np.all(np.abs(x[:, d] - v) < 1e-14) for d in range(3) for v in (0.0, 1.0))


@gpu
@pytest.mark.parametrize("fe", FES)
def test_generic_kernel_and_face_integrals_on_prisms_match_the_oracle(ctx, fe):
    """fh_assemble_poisson_rows on the refined cube of prisms, nodes moved (curved geometry), at a non-trivial state; fh_assemble_neumann_faces on its flagged
    quadrilateral and triangle faces: against the oracle's loops, 1e-12"""
    from femus_amd import capi
    from test_tri_2d import _pattern
    ed, xs, ff, own = ow.refine(*ow.read_gambit(MESH)[:3])
    xs = xs + 0.01 * np.sin(5 * xs[:, [1, 2, 0]]) * (xs * (1 - xs)).prod(axis=1, keepdims=True) * 60
    nc, ndof = ow.NDOF[fe], ow.n_dofs(own, fe)
    u = np.random.default_rng(13).uniform(-1, 1, ndof)
    Ko, Fo = ow.assemble(ed, xs, fe, lambda x: np.exp(x[0]) * (1 + x[1]) - x[2], u)
    K = _pattern(ctx, ed, nc, ndof)
    RES, SOL = ctx.vector(ndof), ctx.vector_from(u)
    f = capi.Expr("exp(x)*(1+y)-z", "x,y,z,t")
    capi.assemble_poisson_rows(ctx, "wedge", fe, ed, xs, K, RES, sol=SOL, source=f)
    assert abs(K.to_scipy() - Ko).max() <= 1e-12 * abs(Ko).max()
    assert np.abs(RES.to_numpy() - Fo).max() <= 1e-12 * np.abs(Fo).max()
    flux = {int(fl): 0.1 * k - 0.25 for k, fl in enumerate(sorted(set(ff[ff < -1].tolist())))}            # every boundary set: both kinds of faces
    fno = ow.neumann(ed, xs, ff, fe, flux)
    R2 = ctx.vector(ndof)
    nq, nt = ow.NFN[fe]
    for quad in (True, False):
        faces, taus = [], []
        for e, fl in zip(*np.nonzero(ff < -1)):
            if (fl < 3) == quad:
                faces.append(ed[e, ow.FACE[fl][:(nq if quad else nt)]])
                taus.append(flux[int(ff[e, fl])])
        if faces:
            capi.assemble_neumann_faces(ctx, "quadface" if quad else "triface", fe, np.array(faces), np.array(taus), xs, R2)
    assert np.abs(R2.to_numpy() - fno).max() <= 1e-12 * np.abs(fno).max()
    f.destroy()
    K.destroy()


def _shipped(fe_order, nlevels=4):
    return """
{
    "multilevel_mesh" : { "first" : { "type" : { "filename" : "input/cube_Wedge.neu" } } },
    "multilevel_solution" : { "multilevel_mesh" : { "first" : { "variable" : { "first" : {
              "name" : "T", "fe_order" : "%s", "init_func" : "0.", "func_source": "0.",
              "boundary_conditions" : [ { "facename" : "top", "bdc_type" : "dirichlet" },
                                        { "facename" : "right", "bdc_type" : "neumann", "bdc_func" : "0.2" } ] } } } } },
    "multilevel_problem" : { "multilevel_mesh" : { "first" : { "system" : { "poisson" : { "linear_solver" : {
                "max_number_linear_iteration" : 6, "abs_conv_tol" : 1.e-09,
                "type" : { "multigrid" : { "nlevels" : %d, "npresmoothing" : 1, "npostsmoothing" : 1, "mgtype" : "V_cycle",
                    "smoother" : { "type" : { "gmres" : { "ksp" : "gmres", "precond" : "ilu", "rtol" : 1.e-12, "atol" : 1.e-20, "divtol" : 1.e+50,
                                                          "max_its" : 4 } } } } } } } } } } }
}
""" % (fe_order, nlevels)


@pytest.mark.parametrize("name,fe_order", [("input3D_Wedge_first.json", "first"), ("input3D_Wedge_second.json", "second"), ("input3D_Wedge_serendipity.json", "serendipity")])
def test_the_configurations_below_are_the_shipped_files(name, fe_order):
    from femus_amd import app_poisson as app
    ref_file = "/root/reference/applications/001_Poisson/input/" + name
    if not os.path.exists(ref_file):
        pytest.skip("the reference tree is not here")
    assert app.load_config(ref_file) == app.load_config(_shipped(fe_order))


@gpu
@pytest.mark.parametrize("fe_order,fe,nlevels", [("first", "linear", 4), ("serendipity", "serendipity", 3), ("second", "biquadratic", 3)])
def test_the_shipped_prism_inputs_of_001_poisson(ctx, tmp_path, fe_order, fe, nlevels):
    """applications/001_Poisson/input/input3D_Wedge_first.json (four levels, as shipped), _serendipity.json and _second.json (compared on three of their four
    levels: the oracle's loops over the fourth take minutes) with input/cube_Wedge.neu through app_poisson on the GPU -- SetBoundaryCondition of main.cpp:26-36:
    Dirichlet 0 everywhere but face 3, which carries the flux 0.2 -- against the oracle's direct solve of the finest level's problem"""
    from femus_amd import app_poisson as app
    os.makedirs(tmp_path / "input")
    (tmp_path / "input" / "cube_Wedge.neu").write_bytes(open(MESH, "rb").read())
    p = app.Poisson001(ctx, _shipped(fe_order, nlevels), base_dir=str(tmp_path))
    assert p.wedge and p.fe == fe and p.nlevels == nlevels
    out = p.run()
    assert out["converged"] and len(out["history"]) <= 7, out["history"]
    ref, meshes = ow.solve(ow.read_gambit(MESH), nlevels, fe, lambda x: 0.0, dirichlet_flags=(-2, -3, -5, -6, -7), flux_by_flag={-4: 0.2})
    for (ed_p, xs_p, ff_p), (ed_o, xs_o, ff_o, _) in zip(out["levels"], meshes):
        assert np.array_equal(ed_p, ed_o) and np.array_equal(ff_p, ff_o) and np.abs(xs_p - xs_o).max() < 1e-14
    assert out["dofs"] == ref.size and np.abs(ref).max() > 1e-3
    assert np.abs(out["solution"] - ref).max() < 1e-8
    p.max_linear, p.abs_tol = 40, 1e-13
    out = p.run()
    assert out["converged"] and np.abs(out["solution"] - ref).max() < 1e-10
    p.destroy()


@gpu
@pytest.mark.parametrize("fe_order", ["serendipity", "second"])
def test_the_shipped_prism_inputs_on_all_of_their_four_levels(ctx, tmp_path, fe_order):
    """the two inputs exactly as shipped (four levels: 8 192 prisms): converge under the input's own limits"""
    from femus_amd import app_poisson as app
    os.makedirs(tmp_path / "input")
    (tmp_path / "input" / "cube_Wedge.neu").write_bytes(open(MESH, "rb").read())
    p = app.Poisson001(ctx, _shipped(fe_order, 4), base_dir=str(tmp_path))
    out = p.run()
    assert out["converged"] and len(out["history"]) <= 7 and out["levels"][-1][0].shape[0] == 16 * 8 ** 3, out["history"]
    p.destroy()
