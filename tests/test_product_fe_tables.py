"""Product (libfemus_hip.so, host entry points: no GPU needed) against the fixture dumped from the reference's own compiled
FE / quadrature / GeomElem sources (tests/golden/fe_tables.npz, generator tests/golden/make_golden.py): rows a1-a3, a5 (face
nodes) and a6 of SURVEY 8 -- bit for bit.  The oracle is checked against the same new tables beside it."""
import os

import numpy as np
import pytest

from femus_amd import capi
from oracle import femus_oracle as fo

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fe_tables.npz"))
ORDERS = ["first", "third", "fifth", "seventh", "ninth"]


@pytest.mark.parametrize("geom", ["quad", "hex"])
@pytest.mark.parametrize("order", ORDERS)
def test_product_gauss_tables_bit_exact(geom, order):
    """a1: Gauss::Gauss + tables (quadrature_interface.cpp:36-94, quadrature_Hexahedron.cpp, quadrature_Quadrangle.cpp)"""
    w, x = capi.fe_gauss(geom, order)
    assert np.array_equal(w, G["gauss_w_%s_%s" % (geom, order)])
    assert np.array_equal(x, G["gauss_x_%s_%s" % (geom, order)])


@pytest.mark.parametrize("geom", ["quad", "hex"])
@pytest.mark.parametrize("fe", ["linear", "serendipity", "biquadratic", "constant"])
def test_product_shape_tables_at_quadrature_points_bit_exact(geom, fe):
    """a2/a3: phi and d phi at the 'seventh' Gauss points (ElemType.cpp:576-741 fills its tables with exactly these calls)"""
    phi, dphi = capi.fe_tables(geom, fe, "seventh")
    ref = G["basis_%s_%s_gauss7" % (geom, fe)]
    assert np.array_equal(phi, ref[0])
    for d in range(dphi.shape[2]):
        assert np.array_equal(dphi[:, :, d], ref[1 + d])


@pytest.mark.parametrize("geom", ["quad", "hex"])
@pytest.mark.parametrize("fe", ["linear", "serendipity", "biquadratic", "constant"])
def test_product_second_derivative_tables_bit_exact(geom, fe):
    """a3: the _d2phi* tables elem_type fills for the optional Hessians (ElemType.cpp:637-741).  The linear families do not implement the pure
    second derivatives in the reference (the fixture holds none); they are identically zero"""
    d2 = capi.fe_tables_d2(geom, fe, "seventh")
    ref = G["basis_%s_%s_gauss7" % (geom, fe)]
    idx = [4, 5, 7] if geom == "quad" else [4, 5, 6, 7, 8, 9]
    npure = 2 if geom == "quad" else 3
    for k, which in enumerate(idx):
        if fe == "linear" and k < npure:
            assert np.all(d2[:, :, k] == 0.0)
        else:
            assert np.array_equal(d2[:, :, k], ref[which])
    assert np.array_equal(d2, fo.eval_basis(geom, fe, G["gauss_x_%s_seventh" % geom])[2])


@pytest.mark.parametrize("fe", ["linear", "biquadratic"])
def test_product_line_tables_bit_exact(fe):
    """EDGE3 (round 6, the one-dimensional input of 001_Poisson): LineLinear / LineBiquadratic (1d/Edge.hpp:72-104) at the 'seventh' line Gauss points -- phi, d/dx,
    d2/dx2 -- and the Gauss table itself, against the reference's compiled classes; the serendipity family of a line is the three-node one"""
    w, x = capi.fe_gauss("line", "seventh")
    assert np.array_equal(w, G["gauss_w_line_seventh"]) and np.array_equal(x, G["gauss_x_line_seventh"])
    ref = G["basis_line_%s_gauss7" % fe]
    phi, dphi = capi.fe_tables("line", fe, "seventh")
    d2 = capi.fe_tables_d2("line", fe, "seventh")
    assert np.array_equal(phi, ref[0]) and np.array_equal(dphi[:, :, 0], ref[1]) and np.array_equal(d2[:, :, 0], ref[2])
    if fe == "biquadratic":
        ps, ds = capi.fe_tables("line", "serendipity", "seventh")
        assert np.array_equal(ps, phi) and np.array_equal(ds, dphi)
    assert np.array_equal(G["xc_line"][:, 0], [-1.0, 1.0, 0.0])
    assert [capi.fe_face_nodes("line", fe, f).tolist() for f in range(2)] == [[0], [1]]


@pytest.mark.parametrize("order", ORDERS)
def test_product_triangle_gauss_tables_bit_exact(order):
    """2d/quadrature_Triangle.cpp: the five symmetric rules (1 / 4 / 7 / 13 / 19 points)"""
    w, x = capi.fe_gauss("tri", order)
    assert np.array_equal(w, G["gauss_w_tri_%s" % order]) and np.array_equal(x, G["gauss_x_tri_%s" % order])


@pytest.mark.parametrize("fe", ["linear", "serendipity", "biquadratic"])
def test_product_triangle_tables_bit_exact(fe):
    """TRI7 (round 6): TriLinear / TriQuadratic / TriBiquadratic (2d/Triangle.hpp:69-181) at the 'seventh' points: phi, the two first and the three second
    derivatives against the reference's compiled classes; children, edge nodes and the element prolongator (coarse functions at the children's nodes)"""
    ref = G["basis_tri_%s_gauss7" % fe]
    phi, dphi = capi.fe_tables("tri", fe, "seventh")
    d2 = capi.fe_tables_d2("tri", fe, "seventh")
    assert np.array_equal(phi, ref[0]) and np.array_equal(dphi[:, :, 0], ref[1]) and np.array_equal(dphi[:, :, 1], ref[2])
    for k in range(3):
        assert np.array_equal(d2[:, :, k], ref[3 + k])
    assert np.allclose(phi.sum(axis=1), 1.0, atol=1e-14) and np.allclose(dphi.sum(axis=1), 0.0, atol=1e-13)
    nfn = 2 if fe == "linear" else 3
    assert [capi.fe_face_nodes("tri", fe, f).tolist() for f in range(3)] == [G["facedofs_tri"][f][:nfn].tolist() for f in range(3)]
    P = capi.fe_elem_prolongator("tri", fe)
    nc = phi.shape[1]
    assert P.shape == (4, nc, nc)
    f2c = G["f2c_tri"]
    for j in range(4):
        for i in range(3):                                # a child's vertex is a node of the father: the row is a unit vector (for the families that hold that node)
            if f2c[j][i] < nc:
                e = np.zeros(nc)
                e[f2c[j][i]] = 1.0
                assert np.allclose(P[j, i], e, atol=1e-14)
        assert np.allclose(P[j].sum(axis=1), 1.0, atol=1e-14)     # partition of unity at every child node


@pytest.mark.parametrize("order", ORDERS)
def test_product_tetrahedron_gauss_tables_bit_exact(order):
    """3d/quadrature_Tetrahedron.cpp: the five rules (1 / 5 / 15 / 31 / 45 points)"""
    w, x = capi.fe_gauss("tet", order)
    assert np.array_equal(w, G["gauss_w_tet_%s" % order]) and np.array_equal(x, G["gauss_x_tet_%s" % order])


@pytest.mark.parametrize("fe", ["linear", "serendipity"])
def test_product_tetrahedron_tables_bit_exact(fe):
    """TET10 (round 6): TetLinear / TetQuadratic (3d/Tetrahedron.cpp) at the 'seventh' points: phi, the three first and the six second derivatives against the
    reference's compiled classes; children, face nodes, element prolongator; the P2 + bubble family (TET15) is refused"""
    ref = G["basis_tet_%s_gauss7" % fe]
    phi, dphi = capi.fe_tables("tet", fe, "seventh")
    d2 = capi.fe_tables_d2("tet", fe, "seventh")
    assert np.array_equal(phi, ref[0])
    for d in range(3):
        assert np.array_equal(dphi[:, :, d], ref[1 + d])
    for k in range(6):
        assert np.array_equal(d2[:, :, k], ref[4 + k])
    nfn = 3 if fe == "linear" else 6
    assert [capi.fe_face_nodes("tet", fe, f).tolist() for f in range(4)] == [G["facedofs_tet"][f][:nfn].tolist() for f in range(4)]
    P = capi.fe_elem_prolongator("tet", fe)
    nc = phi.shape[1]
    assert P.shape == (8, nc, nc)
    f2c = G["f2c_tet"]
    for j in range(8):
        for i in range(4):
            if f2c[j][i] < nc:
                e = np.zeros(nc)
                e[f2c[j][i]] = 1.0
                assert np.allclose(P[j, i], e, atol=1e-14)
        assert np.allclose(P[j].sum(axis=1), 1.0, atol=1e-14)


def test_product_tetrahedron_p2_bubble_tables():
    """TET15: TetBiquadratic (3d/Tetrahedron.cpp:325-600) in the product's hierarchical form (vertex / edge / face / centre terms summed in another order than the
    reference's expanded polynomials): values and first derivatives to 1e-14 at the 'seventh' points; the fifteen nodes and the seven nodes of a
    face bit for bit; Kronecker property at the nodes; the element prolongator a partition of unity; the second derivatives are refused"""
    ref = G["basis_tet_biquadratic_gauss7"]
    phi, dphi = capi.fe_tables("tet", "biquadratic", "seventh")
    assert np.allclose(phi, ref[0], rtol=0, atol=1e-14)
    for d in range(3):
        assert np.allclose(dphi[:, :, d], ref[1 + d], rtol=0, atol=2e-14)
    xc = np.array([capi.fe_node_ref_coords("tet", i) for i in range(15)])
    assert np.array_equal(xc, G["xc_tet15"])
    assert [capi.fe_face_nodes("tet", "biquadratic", f).tolist() for f in range(4)] == G["facedofs_tet15"].tolist()
    P = capi.fe_elem_prolongator("tet", "biquadratic")
    assert P.shape == (8, 15, 15)
    for j in range(8):
        assert np.allclose(P[j].sum(axis=1), 1.0, atol=1e-13)
        for i in range(4):
            e = np.zeros(15)
            e[G["f2c_tet"][j][i]] = 1.0
            assert np.allclose(P[j, i], e, atol=1e-14)
    with pytest.raises(capi.FemusHipError):
        capi.fe_tables_d2("tet", "biquadratic", "seventh")


@pytest.mark.parametrize("order", ORDERS)
def test_product_prism_gauss_tables_bit_exact(order):
    """3d/quadrature_Wedge.cpp: the five rules (1 / 8 / 21 / 52 / 95 points)"""
    w, x = capi.fe_gauss("wedge", order)
    assert np.array_equal(w, G["gauss_w_wedge_%s" % order]) and np.array_equal(x, G["gauss_x_wedge_%s" % order])


@pytest.mark.parametrize("fe", ["linear", "serendipity", "biquadratic"])
def test_product_prism_tables_bit_exact(fe):
    """WEDGE21 (round 6): WedgeLinear / WedgeQuadratic / WedgeBiquadratic (3d/Wedge.cpp) at the 'seventh' points: phi and the three first derivatives against the
    reference's compiled classes (biquadratic: the six second derivatives too); children, face nodes, element prolongator"""
    ref = G["basis_wedge_%s_gauss7" % fe]
    phi, dphi = capi.fe_tables("wedge", fe, "seventh")
    assert np.array_equal(phi, ref[0])
    for d in range(3):
        assert np.array_equal(dphi[:, :, d], ref[1 + d])
    if fe == "biquadratic":
        d2 = capi.fe_tables_d2("wedge", fe, "seventh")
        for k in range(6):
            assert np.array_equal(d2[:, :, k], ref[4 + k])
    assert np.allclose(phi.sum(axis=1), 1.0, atol=1e-13)
    nq, nt = {"linear": (4, 3), "serendipity": (8, 6), "biquadratic": (9, 7)}[fe]
    for f in range(5):
        assert capi.fe_face_nodes("wedge", fe, f).tolist() == G["facedofs_wedge"][f][:(nq if f < 3 else nt)].tolist()
    P = capi.fe_elem_prolongator("wedge", fe)
    nc = phi.shape[1]
    assert P.shape == (8, nc, nc)
    f2c = G["f2c_wedge"]
    for j in range(8):
        for i in range(6):
            if f2c[j][i] < nc:
                e = np.zeros(nc)
                e[f2c[j][i]] = 1.0
                assert np.allclose(P[j, i], e, atol=1e-14)
        assert np.allclose(P[j].sum(axis=1), 1.0, atol=1e-13)


def _rows_by_kvert(geom, fe, P):
    """rows of a [child][local node][coarse] element prolongator in the reference's fine-node order KVERT_IND (Hexahedron.cpp:49-71)"""
    kv = G["kvert_ind_%s_%s" % (geom, fe)]
    return np.array([P[j, i] for (j, i) in kv])


@pytest.mark.parametrize("geom", ["quad", "hex"])
def test_product_piecewise_constant_element_prolongator(geom):
    """a6 for quad0 / hex0 (Quadrilateral.hpp:173-, Hexahedron.hpp:196-): the family's own fine-point table hex_const::X / KVERT_IND (Hexahedron.cpp:258-278)
    lists one (child, function 0) pair per child; P[i][0] = 1 at every one of them"""
    ref, kv = G["elem_prol_%s_constant" % geom], G["kvert_ind_%s_constant" % geom]
    nch = 8 if geom == "hex" else 4
    assert kv.tolist() == [[j, 0] for j in range(nch)] and np.array_equal(ref, np.ones((nch, 1)))
    for P in (capi.fe_elem_prolongator(geom, "constant"), fo.elem_prolongator(geom, "constant")):
        assert P.shape == (nch, 1, 1) and np.array_equal(_rows_by_kvert(geom, "constant", P), ref)
    # the fine points are the centres of the children: half the reference coordinates of the coarse vertex the child sits at
    assert np.array_equal(G["xfine_%s_constant" % geom], 0.5 * fo.xc_table(geom)[:nch])


@pytest.mark.parametrize("geom", ["quad", "hex"])
def test_product_serendipity_face_nodes(geom):
    """a5 for the serendipity family: basis::GetFaceDof of QuadQuadratic / HexQuadratic -- EDGE3 / QUAD8 faces, vertices first"""
    fd = G["facedofs_%s_serendipity" % geom]
    nvf = 4 if geom == "hex" else 2
    for f in range(fd.shape[0]):
        got = capi.fe_face_nodes(geom, "serendipity", f)
        assert got.size == fd.shape[1] and set(got.tolist()) == set(fd[f].tolist())
        assert set(got[:nvf].tolist()) == set(fd[f][:nvf].tolist())
        assert set(fo.face_local_nodes(geom, "serendipity", f).tolist()) == set(fd[f].tolist())
        assert capi.fe_face_nodes(geom, "constant", f).size == 0


@pytest.mark.parametrize("geom", ["quad", "hex"])
@pytest.mark.parametrize("fe", ["linear", "serendipity", "biquadratic"])
def test_product_element_prolongator_bit_exact(geom, fe):
    """a6: set_prolongation_OneElement_All_FE (ElemType.cpp:439-532): phi_j(GetX(i)), |.| < 1e-14 dropped"""
    ref = G["elem_prol_%s_%s" % (geom, fe)]
    P = capi.fe_elem_prolongator(geom, fe)
    assert np.array_equal(_rows_by_kvert(geom, fe, P), ref)
    Po = fo.elem_prolongator(geom, fe)
    assert np.array_equal(_rows_by_kvert(geom, fe, Po), ref)
    # every (child, local node) pair denotes one of the reference's fine nodes and carries that node's row
    X = fo.child_node_ref_coords(geom)[:, :P.shape[1], :]
    key = {tuple(x): r for x, r in zip(G["xfine_%s_%s" % (geom, fe)], ref)}
    for j in range(P.shape[0]):
        for i in range(P.shape[1]):
            assert np.array_equal(P[j, i], key[tuple(X[j, i])])
    if geom == "hex" and fe == "biquadratic":
        assert ref.shape == (125, 27) and np.count_nonzero(ref) == 729
    if fe == "serendipity":
        assert ref.shape == ((81, 20) if geom == "hex" else (21, 8))          # HexQuadratic(20, 81), QuadQuadratic(8, 21)


@pytest.mark.parametrize("geom,tag", [("quad", "quad9")])
def test_element_prolongator_rows_are_the_rows_of_the_geomelem_embedding_matrix(geom, tag):
    """the compiled GeomElemQuad9 embedding matrix (float, deprecated refinement path, GeomElemQuad9.cpp:22-) numbers children and
    their nodes differently, but describes the same 25 fine nodes over the same coarse node numbering: the SET of distinct weight
    rows must be identical, bit for bit (the weights are dyadic, exact in float).  GeomElemHex27's matrix follows another coarse
    node numbering (its rows are not rows of the FE prolongator under any child permutation), so it is not part of the fixture."""
    E = G["geomelem_embedding_" + tag]
    nc = E.shape[2]
    rows_ref = {tuple(r) for r in E.reshape(-1, nc)}
    for P in (capi.fe_elem_prolongator(geom, "biquadratic"), fo.elem_prolongator(geom, "biquadratic")):
        rows = {tuple(r) for r in P.reshape(-1, nc)}
        assert rows == rows_ref and len(rows) == (125 if geom == "hex" else 25)


@pytest.mark.parametrize("geom,tag", [("quad", "quad9"), ("hex", "hex27")])
def test_product_face_nodes_against_facedofs_and_geomelem_faces(geom, tag):
    """a5: local nodes of each face -- basis::GetFaceDof (faceDofs, Hexahedron.cpp / Quadrilateral.cpp) and
    GeomElem*::get_nodes_of_face (_faces, GeomElemHex27.cpp:15-22): same face -> same node SET, vertices first in both"""
    faces = G["geomelem_faces_" + tag]
    fd = G["facedofs_" + geom]
    info = G["geomelem_info_" + tag]
    dim = 3 if geom == "hex" else 2
    assert info.tolist() == [dim, 3 ** dim, 2 ** dim, 2 * dim]
    nvf = 4 if geom == "hex" else 2
    for f in range(2 * dim):
        got = capi.fe_face_nodes(geom, "biquadratic", f)
        assert set(got.tolist()) == set(fd[f].tolist()) == set(faces[f].tolist())
        assert set(got[:nvf].tolist()) == set(fd[f][:nvf].tolist())                        # vertices of the face come first
        assert got[-1] == fd[f][-1] == faces[f][-1]                                         # the face's own centre node comes last
        lin = capi.fe_face_nodes(geom, "linear", f)
        assert set(lin.tolist()) == set(fd[f][:nvf].tolist())
        assert set(fo.face_nodes(geom)[f].tolist()) == set(faces[f].tolist())


@pytest.mark.parametrize("order", ORDERS)
def test_tensor_rules_are_products_of_the_line_rule(order):
    """the 1-D rules of the fixture (quadrature_Line.cpp): the product tables are tensor grids of them (what the sum-factorised
    Jacobian of the HEX27 element kernel relies on, checked again at assembler creation)"""
    x1 = np.sort(G["gauss_x_line_" + order][:, 0])
    for geom in ("quad", "hex"):
        _, x = capi.fe_gauss(geom, order)
        for d in range(x.shape[1]):
            assert np.allclose(np.unique(np.round(x[:, d], 13)), np.round(x1, 13), atol=1e-13)
