"""Tetrahedra (the TET10 meshes of applications/001_Poisson: input3D_Tet_first / _serendipity / _second.json with input/cube_Tet.neu, a data file of the application kept in
tests/golden).  CPU: the oracle restatement (oracle/femus_oracle_tet.py) -- basis against the fixture of the reference's compiled classes, reader / refinement
properties, the product's host-side mesh code equal to it.  GPU: the generic kernel and the triangle-face integrals against the oracle entry for entry, and
the shipped inputs through app_poisson against the oracle's direct solve."""
import os

import numpy as np
import pytest

from oracle import femus_oracle_tet as oq

HERE = os.path.dirname(os.path.abspath(__file__))
MESH = os.path.join(HERE, "golden", "cube_Tet.neu")
G = np.load(os.path.join(HERE, "golden", "fe_tables.npz"))
gpu = pytest.mark.gpu


def volumes(ed, xs):
    return np.einsum("ij,ij->i", np.cross(xs[ed[:, 1]] - xs[ed[:, 0]], xs[ed[:, 2]] - xs[ed[:, 0]]), xs[ed[:, 3]] - xs[ed[:, 0]]) / 6


@pytest.mark.parametrize("fe", ["linear", "serendipity", "biquadratic"])
def test_oracle_tetrahedron_basis_is_the_reference_s(fe):
    for tag, pts in (("sample", G["sample_pts_tet"]), ("gauss7", G["gauss_x_tet_seventh"])):
        ref = G["basis_tet_%s_%s" % (fe, tag)][:4]
        phi, dphi = oq.basis(fe, pts)
        assert np.abs(phi - ref[0]).max() < 4e-15 and max(np.abs(dphi[:, :, d] - ref[1 + d]).max() for d in range(3)) < 4e-15
    assert np.array_equal(oq.XC, G["xc_tet15"]) and np.array_equal(oq.F2C, G["f2c_tet"]) and np.array_equal(oq.FACE, G["facedofs_tet15"])
    assert np.array_equal(oq.XC[:10], G["xc_tet"]) and np.array_equal(oq.FACE[:, :6], G["facedofs_tet"])
    P = oq.elem_prolongator(fe)
    assert np.allclose(P.sum(axis=2), 1.0, atol=1e-13)


def test_the_mesh_file_is_the_application_s():
    ref_file = "/root/reference/applications/001_Poisson/input/cube_Tet.neu"
    if not os.path.exists(ref_file):
        pytest.skip("the reference tree is not here")
    assert open(ref_file, "rb").read() == open(MESH, "rb").read()


def test_oracle_reader_and_refinement_and_the_product_s_mesh_code():
    """cube_Tet.neu: 105 positively oriented TET10 elements filling the unit cube, middles at the middles, the added face nodes at the faces' centres (234 of
    them: every inner face shared) and the added centres at the centres, six boundary sets of eight faces; refined: eight times
    the elements, the same volume, four times the faces per set, flagged faces on the cube's surface; femus_amd/tet_mesh.py gives the same integers and
    coordinates on three levels"""
    from femus_amd import tet_mesh
    ed, xs, ff, own = oq.read_gambit(MESH)
    assert ed.shape == (105, 15) and own == [39, 206, 545] and np.isclose(volumes(ed, xs).sum(), 1.0) and volumes(ed, xs).min() > 0
    assert np.unique(ed[:, 10:14]).size == (105 * 4 + 48) // 2 and np.unique(ed[:, 14]).size == 105
    for f in range(4):
        assert np.allclose(xs[ed[:, 10 + f]], xs[ed[:, oq.FACE[f][:3]]].mean(axis=1), atol=1e-15)
    assert np.allclose(xs[ed[:, 14]], xs[ed[:, :4]].mean(axis=1), atol=1e-15)
    for m, (a, b) in enumerate(oq.EDGE):
        assert np.allclose(xs[ed[:, 4 + m]], 0.5 * (xs[ed[:, a]] + xs[ed[:, b]]))
    assert [(ff == f).sum() for f in range(-7, -1)] == [8] * 6
    a, b = tet_mesh.read_gambit(MESH), (ed, xs, ff, own)
    for level in range(3):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2]) and a[3] == b[3]
        # file level: the same bits; refined levels: the TET15 element prolongator of the library and the oracle's agree to rounding (other order of the sums)
        assert np.array_equal(a[1], b[1]) if level == 0 else np.abs(a[1] - b[1]).max() < 2e-15
        if level == 2:
            break
        a, b = tet_mesh.refine(*a[:3]), oq.refine(*b[:3])
        ef, xf, fff, _ = b
        assert ef.shape[0] == 105 * 8 ** (level + 1) and np.isclose(volumes(ef, xf).sum(), 1.0) and volumes(ef, xf).min() > 0
        assert [(fff == f).sum() for f in range(-7, -1)] == [8 * 4 ** (level + 1)] * 6
        for e, f in zip(*np.nonzero(fff < -1)):
            x = xf[ef[e, oq.FACE[f][:3]]]
            assert any(np.all(x[:, d] == v) for d in range(3) for v in (0.0, 1.0))


@gpu
@pytest.mark.parametrize("fe", ["linear", "serendipity", "biquadratic"])
def test_generic_kernel_and_face_integrals_on_tetrahedra_match_the_oracle(ctx, fe):
    """fh_assemble_poisson_rows on the refined cube of tetrahedra, nodes moved (curved P2 geometry), at a non-trivial state; fh_assemble_neumann_faces on its
    flagged TRI3 / TRI6 / TRI7 faces: against the oracle's loops, 1e-12"""
    from femus_amd import capi
    from test_tri_2d import _pattern
    ed, xs, ff, own = oq.refine(*oq.read_gambit(MESH)[:3])
    xs = xs + 0.01 * np.sin(5 * xs[:, [1, 2, 0]]) * (xs * (1 - xs)).prod(axis=1, keepdims=True) * 60
    nc, ndof = oq.NDOF[fe], oq.n_dofs(own, fe)
    u = np.random.default_rng(11).uniform(-1, 1, ndof)
    Ko, Fo = oq.assemble(ed, xs, fe, lambda x: np.exp(x[0]) * (1 + x[1]) - x[2], u)
    K = _pattern(ctx, ed, nc, ndof)
    RES, SOL = ctx.vector(ndof), ctx.vector_from(u)
    f = capi.Expr("exp(x)*(1+y)-z", "x,y,z,t")
    capi.assemble_poisson_rows(ctx, "tet", fe, ed, xs, K, RES, sol=SOL, source=f)
    Kd, Ka = K.to_scipy(), Ko
    assert abs(Kd - Ka).max() <= 1e-12 * abs(Ka).max()
    assert np.abs(RES.to_numpy() - Fo).max() <= 1e-12 * np.abs(Fo).max()
    fno = oq.neumann(ed, xs, ff, fe, {-4: 0.2, -6: -1.5})
    nfn = oq.NFACE[fe]
    faces, taus = [], []
    for e, fl in zip(*np.nonzero(ff < -1)):
        if ff[e, fl] in (-4, -6):
            faces.append(ed[e, oq.FACE[fl][:nfn]])
            taus.append(0.2 if ff[e, fl] == -4 else -1.5)
    R2 = ctx.vector(ndof)
    capi.assemble_neumann_faces(ctx, "tet", fe, np.array(faces), np.array(taus), xs, R2)
    assert np.abs(R2.to_numpy() - fno).max() <= 1e-13 * np.abs(fno).max() + 1e-16
    f.destroy()
    K.destroy()


def _shipped(fe_order, nlevels=4):
    return """
{
    "multilevel_mesh" : { "first" : { "type" : { "filename" : "input/cube_Tet.neu" } } },
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


@pytest.mark.parametrize("name,fe_order", [("input3D_Tet_first.json", "first"), ("input3D_Tet_serendipity.json", "serendipity"), ("input3D_Tet_second.json", "second")])
def test_the_configurations_below_are_the_shipped_files(name, fe_order):
    from femus_amd import app_poisson as app
    ref_file = "/root/reference/applications/001_Poisson/input/" + name
    if not os.path.exists(ref_file):
        pytest.skip("the reference tree is not here")
    assert app.load_config(ref_file) == app.load_config(_shipped(fe_order))


@gpu
@pytest.mark.parametrize("fe_order,fe,nlevels", [("first", "linear", 4), ("serendipity", "serendipity", 3), ("second", "biquadratic", 3)])
def test_the_shipped_tetrahedral_inputs_of_001_poisson(ctx, tmp_path, fe_order, fe, nlevels):
    """applications/001_Poisson/input/input3D_Tet_first.json (four levels, as shipped), input3D_Tet_serendipity.json and input3D_Tet_second.json (on three of
    their four levels: the oracle's direct solve of the fourth takes minutes) with input/cube_Tet.neu through app_poisson on the GPU -- SetBoundaryCondition of main.cpp:26-36: Dirichlet 0
    everywhere but face 3, which carries the flux 0.2 -- against the oracle's direct solve of the finest level's problem"""
    from femus_amd import app_poisson as app
    os.makedirs(tmp_path / "input")
    (tmp_path / "input" / "cube_Tet.neu").write_bytes(open(MESH, "rb").read())
    p = app.Poisson001(ctx, _shipped(fe_order, nlevels), base_dir=str(tmp_path))
    assert p.tet and p.fe == fe and p.nlevels == nlevels
    out = p.run()
    assert out["converged"] and len(out["history"]) <= 7, out["history"]
    ref, meshes = oq.solve(oq.read_gambit(MESH), nlevels, fe, lambda x: 0.0, dirichlet_flags=(-2, -3, -5, -6, -7), flux_by_flag={-4: 0.2})
    for (ed_p, xs_p, ff_p), (ed_o, xs_o, ff_o, _) in zip(out["levels"], meshes):
        assert np.array_equal(ed_p, ed_o) and np.array_equal(ff_p, ff_o) and np.abs(xs_p - xs_o).max() < 2e-15
    assert out["dofs"] == ref.size and np.abs(ref).max() > 1e-3
    assert np.abs(out["solution"] - ref).max() < 1e-8
    p.max_linear, p.abs_tol = 40, 1e-13
    out = p.run()
    assert out["converged"] and np.abs(out["solution"] - ref).max() < 1e-10
    p.destroy()


@gpu
@pytest.mark.parametrize("fe_order", ["serendipity", "second"])
def test_the_shipped_quadratic_inputs_on_all_of_their_four_levels(ctx, tmp_path, fe_order):
    """input3D_Tet_serendipity.json / input3D_Tet_second.json exactly as shipped (four levels: 53 760 elements): converges under the input's own limits; its solution at the nodes of
    the three-level problem stays within the discretisation error of that problem's solution (the oracle comparison proper is the three-level test above)"""
    from femus_amd import app_poisson as app
    os.makedirs(tmp_path / "input")
    (tmp_path / "input" / "cube_Tet.neu").write_bytes(open(MESH, "rb").read())
    p4 = app.Poisson001(ctx, _shipped(fe_order, 4), base_dir=str(tmp_path))
    out4 = p4.run()
    assert out4["converged"] and len(out4["history"]) <= 7, out4["history"]
    assert out4["levels"][-1][0].shape[0] == 105 * 8 ** 3
    p3 = app.Poisson001(ctx, _shipped(fe_order, 3), base_dir=str(tmp_path))
    out3 = p3.run()
    n3 = out3["dofs"]                                             # the nodes of level 3 are the first vertices of level 4 (vertices are numbered first, fathers' nodes first)
    x3, x4 = out3["coords"], out4["coords"]
    import scipy.spatial
    d, idx = scipy.spatial.cKDTree(x4).query(x3)
    assert d.max() < 1e-12
    assert np.abs(out4["solution"][idx] - out3["solution"]).max() < 0.1 * np.abs(out3["solution"]).max()         # (5 % where the flux face meets the Dirichlet faces)
    p3.destroy()
    p4.destroy()
