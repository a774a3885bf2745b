"""Meshes of mixed shapes (applications/001_Poisson: input3D.json / input3D_All_first.json with input/cube_all_shapes_Six_boundary_groups.neu -- ten tetrahedra, six
prisms and four hexahedra in one Gambit file, a data file of the application kept in tests/golden).  CPU: the oracle restatement
(oracle/femus_oracle_mixed.py) -- reader / refinement properties, the product's host-side mesh code equal to it.  GPU: the mixed generic kernel and the face integrals
against the oracle entry for entry, and the shipped inputs through app_poisson against the oracle's direct solve.  Below: the two-dimensional Gambit files
(QUAD9 + TRI6, TRI6 alone) through the same reader, refinement and kernel."""
import os

import numpy as np
import pytest

from oracle import femus_oracle_mixed as om

HERE = os.path.dirname(os.path.abspath(__file__))
MESH = os.path.join(HERE, "golden", "cube_all_shapes_Six_boundary_groups.neu")
gpu = pytest.mark.gpu


def volume(kind, ed, xs):
    """the integral of 1 with each element's own tables (the reference's tetrahedron weights sum to 1/6 - 1.07e-9: the cube of tetrahedra alone would miss 6.4e-9)"""
    vol = 0.0
    for s in set(kind.tolist()):
        w, _, DPHI = om.tables(s, "biquadratic")
        for e in np.nonzero(kind == s)[0]:
            x = xs[ed[e, :om.NLOC[s]]]
            dets = np.array([np.linalg.det(DPHI[g].T @ x) for g in range(len(w))])
            assert dets.min() > 0
            vol += float(dets @ w)
    return vol


def test_the_mesh_file_is_the_application_s():
    ref_file = "/root/reference/applications/001_Poisson/input/cube_all_shapes_Six_boundary_groups.neu"
    if not os.path.exists(ref_file):
        pytest.skip("the reference tree is not here")
    assert open(ref_file, "rb").read() == open(MESH, "rb").read()


def test_oracle_reader_and_refinement_and_the_product_s_mesh_code():
    """the file: 10 TET10 + 6 WEDGE18 + 4 HEX27 elements filling the unit cube, positively oriented, 131 nodes + 32 triangle-face nodes (those between a tetrahedron
    and a prism shared) + 16 centres, six boundary sets of five faces; edge nodes at the middles; refined twice: eight times the elements, the same volume, four
    times the faces per set, flagged faces on the cube's surface; femus_amd/mixed_mesh.py gives the same integers (and coordinates to rounding) on three levels"""
    from femus_amd import mixed_mesh
    kind, ed, xs, ff, own = om.read_gambit(MESH)
    assert [(kind == s).sum() for s in ("tet", "wedge", "hex")] == [10, 6, 4] and own == [28, 97, 179]
    assert abs(volume(kind, ed, xs) - 1.0) < 1e-8
    for e in range(20):
        s = kind[e]
        nv = om.CLASSES[s][0]
        assert np.all(ed[e, :om.NLOC[s]] >= 0) and np.all(ed[e, om.NLOC[s]:] == -1)
        for m, (a, b) in enumerate(om.EDGE[s]):
            assert np.allclose(xs[ed[e, nv + m]], 0.5 * (xs[ed[e, a]] + xs[ed[e, b]]), atol=1e-11)
        for f in range(len(om.FACE[s])):
            assert np.allclose(xs[ed[e, om.FACE_LOCAL[s][f]]], xs[ed[e, om.FACE[s][f][:om.NVF[s][f]]]].mean(axis=0), atol=1e-11)
    # a triangle between a tetrahedron and a prism carries ONE node
    tri_nodes = {}
    for e in range(20):
        for f in range(len(om.FACE[kind[e]])):
            if om.NVF[kind[e]][f] == 3:
                tri_nodes.setdefault(int(ed[e, om.FACE_LOCAL[kind[e]][f]]), set()).add(kind[e])
    assert len(tri_nodes) == 32 and any(v == {"tet", "wedge"} for v in tri_nodes.values())
    assert [(ff == f).sum() for f in range(-7, -1)] == [5] * 6
    a, b = mixed_mesh.read_gambit(MESH), (kind, ed, xs, ff, own)
    for level in range(3):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[3], b[3]) and a[4] == b[4]
        assert np.array_equal(a[2], b[2]) if level == 0 else np.abs(a[2] - b[2]).max() < 2e-15
        if level == 2:
            break
        a, b = mixed_mesh.refine(*a[:4]), om.refine(*b[:4])
        kf, ef, xf, fff, _ = b
        assert ef.shape[0] == 20 * 8 ** (level + 1) and abs(volume(kf, ef, xf) - 1.0) < 1e-8
        assert [(fff == f).sum() for f in range(-7, -1)] == [5 * 4 ** (level + 1)] * 6
        for e, f in zip(*np.nonzero(fff < -1)):
            x = xf[ef[e, om.FACE[kf[e]][f][:om.NVF[kf[e]][f]]]]
            assert any(np.all(np.abs(x[:, d] - v) < 1e-14) for d in range(3) for v in (0.0, 1.0))


@gpu
@pytest.mark.parametrize("fe", ["linear", "serendipity", "biquadratic"])
def test_mixed_kernel_and_face_integrals_match_the_oracle(ctx, fe):
    """fh_assemble_poisson_mixed on the refined mesh of three shapes, nodes moved (curved geometry), at a non-trivial state; fh_assemble_neumann_faces on its flagged
    triangles and quadrilaterals: against the oracle's loops, 1e-12"""
    from femus_amd import capi
    kind, ed, xs, ff, own = om.refine(*om.read_gambit(MESH)[:4])
    xs = xs + 0.01 * np.sin(5 * xs[:, [1, 2, 0]]) * (xs * (1 - xs)).prod(axis=1, keepdims=True) * 60
    ndof = om.n_dofs(own, fe)
    u = np.random.default_rng(11).uniform(-1, 1, ndof)
    Ko, Fo = om.assemble(kind, ed, xs, fe, lambda x: np.exp(x[0]) * (1 + x[1]) - x[2], u)
    import scipy.sparse as sp
    pat = sp.csr_matrix(Ko)
    pat.sort_indices()
    K = capi.Mat.from_csr(ctx, ndof, ndof, pat.indptr, pat.indices)
    RES, SOL = ctx.vector(ndof), ctx.vector_from(u)
    f = capi.Expr("exp(x)*(1+y)-z", "x,y,z,t")
    capi.assemble_poisson_mixed(ctx, fe, kind, ed, xs, K, RES, sol=SOL, source=f)
    assert abs(K.to_scipy() - Ko).max() <= 1e-12 * abs(Ko).max()
    assert np.abs(RES.to_numpy() - Fo).max() <= 1e-12 * np.abs(Fo).max()
    fno = om.neumann(kind, ed, xs, ff, fe, {-4: 0.2, -6: -1.5}, ndof)
    R2 = ctx.vector(ndof)
    for nv, name in ((3, "triface"), (4, "quadface")):
        faces, taus = [], []
        for e, fl in zip(*np.nonzero(ff < -1)):
            if ff[e, fl] in (-4, -6) and om.NVF[kind[e]][fl] == nv:
                faces.append(ed[e, om.FACE[kind[e]][fl][:om.NFN[nv][fe]]])
                taus.append(0.2 if ff[e, fl] == -4 else -1.5)
        if faces:
            capi.assemble_neumann_faces(ctx, name, fe, np.array(faces), np.array(taus), xs, R2)
    assert np.abs(R2.to_numpy() - fno).max() <= 1e-13 * np.abs(fno).max() + 1e-16
    f.destroy()
    K.destroy()


def _shipped(fe_order, nlevels=4):
    return """
{
    "multilevel_mesh" : { "first" : { "type" : { "filename" : "input/cube_all_shapes_Six_boundary_groups.neu" } } },
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


@pytest.mark.parametrize("name,fe_order", [("input3D_All_first.json", "first"), ("input3D.json", "second")])
def test_the_configurations_below_are_the_shipped_files(name, fe_order):
    from femus_amd import app_poisson as app
    ref_file = "/root/reference/applications/001_Poisson/input/" + name
    if not os.path.exists(ref_file):
        pytest.skip("the reference tree is not here")
    assert app.load_config(ref_file) == app.load_config(_shipped(fe_order))


@gpu
@pytest.mark.parametrize("fe_order,fe,nlevels", [("first", "linear", 4), ("serendipity", "serendipity", 3), ("second", "biquadratic", 3)])
def test_the_shipped_mixed_shape_inputs_of_001_poisson(ctx, tmp_path, fe_order, fe, nlevels):
    """applications/001_Poisson/input/input3D_All_first.json (four levels, as shipped) and input3D.json (second order; compared on three of its four levels: the
    oracle's direct solve of the fourth takes minutes), and the serendipity family on the same mesh, through app_poisson on the GPU -- SetBoundaryCondition of
    main.cpp:26-36: Dirichlet 0 everywhere but face 3, which carries the flux 0.2 -- against the oracle's direct solve of the finest level's problem"""
    from femus_amd import app_poisson as app
    os.makedirs(tmp_path / "input")
    (tmp_path / "input" / os.path.basename(MESH)).write_bytes(open(MESH, "rb").read())
    p = app.Poisson001(ctx, _shipped(fe_order, nlevels), base_dir=str(tmp_path))
    assert p.mixed and p.fe == fe and p.nlevels == nlevels
    out = p.run()
    assert out["converged"] and len(out["history"]) <= 7, out["history"]
    ref, meshes = om.solve(om.read_gambit(MESH), nlevels, fe, lambda x: 0.0, dirichlet_flags=(-2, -3, -5, -6, -7), flux_by_flag={-4: 0.2})
    for (ed_p, xs_p, ff_p), (_, ed_o, xs_o, ff_o, _) in zip(out["levels"], meshes):
        assert np.array_equal(ed_p, ed_o) and np.array_equal(ff_p, ff_o) and np.abs(xs_p - xs_o).max() < 2e-15
    assert out["dofs"] == ref.size and np.abs(ref).max() > 1e-3
    assert np.abs(out["solution"] - ref).max() < 1e-8
    p.max_linear, p.abs_tol = 40, 1e-13
    out = p.run()
    assert out["converged"] and np.abs(out["solution"] - ref).max() < 1e-10
    p.destroy()


@gpu
def test_input3d_json_on_all_of_its_four_levels(ctx, tmp_path):
    """input3D.json exactly as shipped (second order, four levels: 10 240 elements of three shapes): converges under the input's own limits; its solution at the nodes
    of the three-level problem stays within the discretisation error of that problem's solution"""
    from femus_amd import app_poisson as app
    import scipy.spatial
    os.makedirs(tmp_path / "input")
    (tmp_path / "input" / os.path.basename(MESH)).write_bytes(open(MESH, "rb").read())
    p4 = app.Poisson001(ctx, _shipped("second", 4), base_dir=str(tmp_path))
    out4 = p4.run()
    assert out4["converged"] and len(out4["history"]) <= 7, out4["history"]
    assert out4["levels"][-1][0].shape[0] == 20 * 8 ** 3
    p3 = app.Poisson001(ctx, _shipped("second", 3), base_dir=str(tmp_path))
    out3 = p3.run()
    d, idx = scipy.spatial.cKDTree(out4["coords"]).query(out3["coords"])
    assert d.max() < 1e-12
    assert np.abs(out4["solution"][idx] - out3["solution"]).max() < 0.1 * np.abs(out3["solution"]).max()
    p3.destroy()
    p4.destroy()


@gpu
@pytest.mark.parametrize("fe", ["linear", "serendipity", "biquadratic"])
def test_mixed_kernel_in_two_dimensions_a_quadrilateral_beside_two_triangles(ctx, fe):
    """fh_assemble_poisson_mixed with geom 1 (QUAD9) and 3 (TRI7) in one mesh -- the unit square as one quadrilateral and two triangles sharing curved edges --
    against an element loop over the oracles' tables of each shape (1e-12); then the shapes the call refuses"""
    from femus_amd import capi
    from oracle import femus_oracle as fo, femus_oracle_tri as ot
    xs = np.array([[0, 0], [.5, 0], [.5, 1], [0, 1], [1, 0], [1, 1], [.25, 0], [.53, .5], [.25, 1], [0, .5], [.75, 0], [.77, .52], [1, .5], [.75, 1],
                   [.26, .51], [.68, .17], [.84, .66]], dtype=float)
    kind = np.array(["quad", "tri", "tri"])
    ed = np.full((3, 9), -1, dtype=np.int64)
    ed[0] = [0, 1, 2, 3, 6, 7, 8, 9, 14]
    ed[1, :7] = [1, 4, 2, 10, 11, 7, 15]
    ed[2, :7] = [4, 5, 2, 12, 13, 11, 16]
    fam = {"linear": 0, "serendipity": 1, "biquadratic": 2}[fe]
    ncs = {"quad": (4, 8, 9)[fam], "tri": (3, 6, 7)[fam]}
    ndof = (6, 14, 17)[fam]
    wq, xq = fo.gauss_table("quad", "seventh")
    xq = np.asarray(xq)
    xq = xq.T if (xq.shape[0] == 2 and xq.shape[1] != 2) else xq
    outq = fo.eval_basis("quad", fe, xq)
    wt, xt = ot.gauss("seventh")
    T = {"quad": (np.asarray(wq), outq[0], outq[1]), "tri": (wt,) + tuple(ot.basis(fe, xt))}
    u = np.random.default_rng(5).uniform(-1, 1, ndof)
    src = lambda x: 1.0 + x[0] * x[1]
    Ko, Fo = np.zeros((ndof, ndof)), np.zeros(ndof)
    for e in range(3):
        w, PHI, DPHI = T[kind[e]]
        dof = ed[e, :ncs[kind[e]]]
        x = xs[dof]
        for g in range(len(w)):
            J = DPHI[g].T @ x
            det = np.linalg.det(J)
            assert det > 0
            grad = DPHI[g] @ np.linalg.inv(J).T
            Ko[np.ix_(dof, dof)] += (grad @ grad.T) * det * w[g]
            Fo[dof] += (src(PHI[g] @ x) * PHI[g] - grad @ (grad.T @ u[dof])) * det * w[g]
    import scipy.sparse as sp
    pat = sp.csr_matrix((np.abs(Ko) > 0).astype(float) + np.eye(ndof))
    pat.sort_indices()
    K = capi.Mat.from_csr(ctx, ndof, ndof, pat.indptr, pat.indices)
    RES, SOL = ctx.vector(ndof), ctx.vector_from(u)
    f = capi.Expr("1+x*y", "x,y,z,t")
    capi.assemble_poisson_mixed(ctx, fe, kind, ed, xs, K, RES, sol=SOL, source=f)
    assert abs(K.to_scipy().toarray() - Ko).max() <= 1e-12 * abs(Ko).max()
    assert np.abs(RES.to_numpy() - Fo).max() <= 1e-12 * np.abs(Fo).max()
    with pytest.raises(capi.FemusHipError, match="one dimension"):
        capi.assemble_poisson_mixed(ctx, fe, np.array(["quad", "tet", "tri"]), ed, xs, K, RES)
    with pytest.raises(capi.FemusHipError, match="more than three shapes"):
        capi.assemble_poisson_mixed(ctx, fe, np.array(["hex", "tet", "wedge", "quad"]), np.zeros((4, 27), dtype=np.int64), np.zeros((40, 3)), K, RES)
    f.destroy()
    K.destroy()


def test_the_application_tells_the_mesh_files_apart(tmp_path):
    """Poisson001._gambit_kind on the four Gambit files of the application: the hexahedral file goes to the library's reader (None), the others to the host-side
    mesh modules; a file with an element the readers do not serve is refused by the mixed reader with the element named"""
    from femus_amd import app_poisson as app, mixed_mesh
    g = os.path.join(HERE, "golden")
    kinds = {f: app.Poisson001._gambit_kind(os.path.join(g, f)) for f in ("cube_Hex.neu", "cube_Tet.neu", "cube_Wedge.neu", os.path.basename(MESH))}
    assert kinds == {"cube_Hex.neu": None, "cube_Tet.neu": "tet10", "cube_Wedge.neu": "wedge18", os.path.basename(MESH): "mixed"}
    text = open(MESH).read().replace("       1  6 10 ", "       1  7  5 ", 1)          # a pyramid where the first tetrahedron was
    bad = tmp_path / "bad.neu"
    bad.write_text(text)
    with pytest.raises(ValueError, match="element 1 of Gambit type 7"):
        mixed_mesh.read_gambit(str(bad))
    # several element groups: the single-shape readers refuse (they keep the file's order), the application sends such a file to the mixed reader, which orders
    # the elements by (material, group, index)
    from femus_amd import tet_mesh
    tet = os.path.join(g, "cube_Tet.neu")
    lines = open(tet).read().split("\n")
    k = [i for i, l in enumerate(lines) if "NGRPS" in l][0] + 1
    t = lines[k].split()
    t[2] = "2"
    lines[k] = " ".join(t)
    two = tmp_path / "two_groups.neu"
    two.write_text("\n".join(lines))
    with pytest.raises(ValueError, match="2 element groups"):
        tet_mesh.read_gambit(str(two))
    assert app.Poisson001._gambit_kind(str(two)) == "mixed"


def test_element_groups_order_the_elements():
    """triAMR.neu (applications/MGAMR/ex4/input: eight TRI6 elements in four groups named 5 .. 8, one material): Mesh.cpp:626-690 orders the elements by material,
    group, file index -- file elements 7 8 3 4 5 6 1 2 --, after the triangle centres were added in file order; the oracle's literal bubble sort and the product's
    key sort agree, and the boundary sets follow their elements (set 3: four edges)"""
    from femus_amd import mixed_mesh
    path = os.path.join(HERE, "golden", "triAMR.neu")
    kind, ed, xs, ff, own, group, material = mixed_mesh.read_gambit(path, groups=True)
    assert group.tolist() == [5, 5, 6, 6, 7, 7, 8, 8] and set(material.tolist()) == {2}
    ko, eo, xo, fo_, oo = om.read_gambit(path)
    assert np.array_equal(ed, eo) and np.array_equal(xs, xo) and np.array_equal(ff, fo_) and own == oo
    # the first element of the ordered mesh is element 7 of the file: its vertices are the file's nodes of that line
    tok = open(path).read().split()
    p = tok.index("ELEMENTS/CELLS") + 2 + 6 * 9
    assert tok[p] == "7"
    filenodes = np.array(tok[p + 3:p + 9], dtype=np.int64) - 1
    q = tok.index("COORDINATES") + 2
    xyz = np.array(tok[q:q + 3 * 25], dtype=object).reshape(25, 3)[:, 1:].astype(float)
    assert np.array_equal(xs[ed[0, [0, 3, 1, 4, 2, 5]]], xyz[filenodes])
    assert [(ff == f).sum() for f in (-2, -3, -4)] == [2, 2, 4]


# ---- two dimensions: the Gambit files of QUAD9 and TRI6 elements the reference tree holds (59 files mix the two, 73 hold triangles alone) ------------------------
MESH_2D = {"square_mixed.neu": "applications/MPM_FEM/ex11/input/square_mixed.neu", "tri2.neu": "applications/ISM/ex1/input/tri2.neu",
           "triAMR.neu": "applications/MGAMR/ex4/input/triAMR.neu"}


@pytest.mark.parametrize("name", sorted(MESH_2D))
def test_two_dimensional_gambit_files_reader_refinement_and_the_product_s_mesh_code(name):
    """square_mixed.neu (two QUAD9 + four TRI6 elements on [-1/2, 1/2]^2) and tri2.neu (two TRI6 on the unit square), data files of the reference tree kept in
    tests/golden: the centre FEMuS adds to every triangle at the mean of its vertices, edge nodes at the middles, the area kept by two refinements, boundary edges
    doubled per level and lying on the boundary; femus_amd/mixed_mesh.py gives the oracle's integers (coordinates to rounding) on three levels"""
    from femus_amd import mixed_mesh
    path = os.path.join(HERE, "golden", name)
    ref_file = "/root/reference/" + MESH_2D[name]
    if os.path.exists(ref_file):
        assert open(ref_file, "rb").read() == open(path, "rb").read()
    kind, ed, xs, ff, own = om.read_gambit(path)
    assert xs.shape[1] == 2 and set(kind.tolist()) <= {"quad", "tri"}
    for e in np.nonzero(kind == "tri")[0]:
        assert np.allclose(xs[ed[e, 6]], xs[ed[e, :3]].mean(axis=0), atol=1e-15) and np.all(ed[e, 7:] == -1)
    nb0 = int((ff < -1).sum())

    def area(k, e_, x_):
        tot = 0.0
        for s in set(k.tolist()):
            w, _, DPHI = om.tables(s, "biquadratic")
            for e in np.nonzero(k == s)[0]:
                xe = x_[e_[e, :om.NLOC[s]]]
                dets = np.array([np.linalg.det(DPHI[g].T @ xe) for g in range(len(w))])
                assert dets.min() > 0
                tot += float(dets @ w)
        return tot

    a0 = area(kind, ed, xs)
    assert abs(a0 - 1.0) < 1e-9
    lo, hi = xs.min(axis=0), xs.max(axis=0)
    a, b = mixed_mesh.read_gambit(path), (kind, ed, xs, ff, own)
    for level in range(3):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[3], b[3]) and a[4] == b[4]
        assert np.array_equal(a[2], b[2]) if level == 0 else np.abs(a[2] - b[2]).max() < 2e-15
        if level == 2:
            break
        a, b = mixed_mesh.refine(*a[:4]), om.refine(*b[:4])
        kf, ef, xf, fff, _ = b
        assert ef.shape[0] == ed.shape[0] * 4 ** (level + 1) and abs(area(kf, ef, xf) - a0) < 1e-12 and int((fff < -1).sum()) == nb0 * 2 ** (level + 1)
        for e, f in zip(*np.nonzero(fff < -1)):
            x = xf[ef[e, om.FACE[kf[e]][f][:2]]]
            assert any(np.all(np.abs(x[:, d] - v[d]) < 1e-14) for d in range(2) for v in (lo, hi))


def _config_2d(name, fe_order, nlevels):
    return """
{
    "multilevel_mesh" : { "first" : { "type" : { "filename" : "input/%s" } } },
    "multilevel_solution" : { "multilevel_mesh" : { "first" : { "variable" : { "first" : {
              "name" : "T", "fe_order" : "%s", "init_func" : "0.", "func_source": "1.+x*y",
              "boundary_conditions" : [ { "facename" : "top", "bdc_type" : "dirichlet" } ] } } } } },
    "multilevel_problem" : { "multilevel_mesh" : { "first" : { "system" : { "poisson" : { "linear_solver" : {
                "max_number_linear_iteration" : 8, "abs_conv_tol" : 1.e-10,
                "type" : { "multigrid" : { "nlevels" : %d, "npresmoothing" : 1, "npostsmoothing" : 1, "mgtype" : "V_cycle",
                    "smoother" : { "type" : { "gmres" : { "ksp" : "gmres", "precond" : "ilu", "rtol" : 1.e-12, "atol" : 1.e-20, "divtol" : 1.e+50,
                                                          "max_its" : 4 } } } } } } } } } } }
}
""" % (name, fe_order, nlevels)


@gpu
@pytest.mark.parametrize("name", sorted(MESH_2D))
@pytest.mark.parametrize("fe_order,fe", [("first", "linear"), ("serendipity", "serendipity"), ("second", "biquadratic")])
def test_001_poisson_on_the_two_dimensional_gambit_files(ctx, tmp_path, name, fe_order, fe):
    """applications/001_Poisson with a two-dimensional Gambit file of quadrilaterals and triangles (and of triangles alone) on four levels, source 1 + x y, the
    boundary sets of the file as SetBoundaryCondition treats them (every face name but 3 held at zero, name 3 -- triAMR.neu has it -- with the flux 0.2; that file
    also orders its elements by their four groups): through app_poisson on the GPU against the oracle's direct
    solve of the finest level's problem, 1e-10; meshes equal to the oracle's on every level"""
    from femus_amd import app_poisson as app
    path = os.path.join(HERE, "golden", name)
    os.makedirs(tmp_path / "input")
    (tmp_path / "input" / name).write_bytes(open(path, "rb").read())
    p = app.Poisson001(ctx, _config_2d(name, fe_order, 4), base_dir=str(tmp_path))
    assert p.mixed and p.dim == 2 and p.fe == fe
    p.max_linear, p.abs_tol = 40, 1e-13
    out = p.run()
    assert out["converged"], out["history"]
    flags = set(np.unique(om.read_gambit(path)[3]).tolist()) - {-1}
    ref, meshes = om.solve(om.read_gambit(path), 4, fe, lambda x: 1.0 + x[0] * x[1], dirichlet_flags=tuple(flags - {-4}),
                           flux_by_flag={-4: 0.2} if -4 in flags else None)
    for (ed_p, xs_p, ff_p), (_, ed_o, xs_o, ff_o, _) in zip(out["levels"], meshes):
        assert np.array_equal(ed_p, ed_o) and np.array_equal(ff_p, ff_o) and np.abs(xs_p - xs_o).max() < 2e-15
    assert out["dofs"] == ref.size and np.abs(ref).max() > 1e-3
    assert np.abs(out["solution"] - ref).max() < 1e-10
    p.destroy()


@pytest.mark.skipif(not os.path.isdir("/root/reference/applications"), reason="the reference tree is not here")
def test_the_reader_on_the_small_gambit_files_of_the_reference_tree():
    """every distinct Gambit file of the reference tree up to 40 kB that holds other shapes than HEX27 / QUAD9 alone (the whole tree: tests/dev/sweep_reference_neu.py,
    profiles/r06_gambit_reader_sweep.txt): read and refined once by femus_amd/mixed_mesh.py -- the measure is kept, the boundary faces are multiplied by 4 (edges by
    2); a surface in space is refused with its message"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("sweep", os.path.join(HERE, "dev", "sweep_reference_neu.py"))
    sw = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sw)
    from femus_amd import mixed_mesh
    seen, done, refused = set(), 0, 0
    for root, _, files in os.walk("/root/reference"):
        for f in sorted(files):
            p = os.path.join(root, f)
            if not f.endswith(".neu") or os.path.getsize(p) > 40000:
                continue
            raw = open(p, "rb").read()
            if hash(raw) in seen:
                continue
            seen.add(hash(raw))
            tok = raw.decode(errors="ignore").split()
            q = tok.index("NDFVL") + 1
            k = tok.index("ELEMENTS/CELLS") + 2
            types = set()
            for _ in range(int(tok[q + 1])):
                types.add((int(tok[k + 1]), int(tok[k + 2])))
                k += 3 + int(tok[k + 2])
            if types <= {(4, 27)} or types <= {(2, 9)}:
                continue
            if int(tok[q + 4]) != int(tok[q + 5]):
                with pytest.raises(ValueError, match="a surface in space"):
                    mixed_mesh.read_gambit(p)
                refused += 1
                continue
            m = mixed_mesh.read_gambit(p)
            v0, _ = sw.measure(m[0], m[1], m[2])
            m1 = mixed_mesh.refine(*m[:4])
            v1, _ = sw.measure(m1[0], m1[1], m1[2])
            assert v0 > 0 and abs(v1 - v0) <= 1e-9 * v0, p
            assert int((m1[3] < -1).sum()) == int((m[3] < -1).sum()) * (4 if m[2].shape[1] == 3 else 2), p
            done += 1
    assert done >= 20
