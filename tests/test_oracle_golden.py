"""Oracle pinning (CPU): the numpy restatement against golden vectors dumped from the reference's own
compiled FE / quadrature sources (tests/golden/fe_tables.npz, made by tests/golden/make_golden.py), and --
when it has been built in this container -- against oracle/_ref/libfemus_ref_fe.so directly."""
import ctypes
import os

import numpy as np
import pytest

from oracle import femus_oracle as fo

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "fe_tables.npz"))
REF_SO = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "libfemus_ref_fe.so")


@pytest.mark.parametrize("geom", ["line", "quad", "hex"])
@pytest.mark.parametrize("order", ["first", "third", "fifth", "seventh", "ninth"])
def test_gauss_tables_bit_exact(geom, order):
    w, x = fo.gauss_table(geom, order)
    assert np.array_equal(w, G["gauss_w_%s_%s" % (geom, order)])
    assert np.array_equal(x, G["gauss_x_%s_%s" % (geom, order)])


@pytest.mark.parametrize("geom", ["quad", "hex"])
def test_node_tables(geom):
    assert np.array_equal(fo.xc_table(geom), G["xc_" + geom])
    assert np.array_equal(fo.ind_table(geom), G["ind_" + geom])
    assert np.array_equal(fo.fine2coarse_vertex_mapping(geom), G["f2c_" + geom])
    fn = fo.face_nodes(geom)
    for f in range(len(fn)):
        assert set(fn[f].tolist()) == set(G["facedofs_" + geom][f].tolist())


@pytest.mark.parametrize("geom", ["quad", "hex"])
@pytest.mark.parametrize("fe", ["linear", "serendipity", "biquadratic", "constant"])
@pytest.mark.parametrize("tag", ["gauss7", "sample"])
def test_basis_bit_exact(geom, fe, tag):
    pts = G["gauss_x_%s_seventh" % geom] if tag == "gauss7" else G["sample_pts_" + geom]
    phi, dphi, d2 = fo.eval_basis(geom, fe, pts)
    ref = G["basis_%s_%s_%s" % (geom, fe, tag)]
    dim = pts.shape[1]
    assert np.array_equal(phi, ref[0])
    for d in range(dim):
        assert np.array_equal(dphi[:, :, d], ref[1 + d])
    if fe != "linear":          # (HexLinear / QuadLinear do not implement the pure second derivatives)
        idx = [4, 5, 7] if dim == 2 else [4, 5, 6, 7, 8, 9]
        for k, which in enumerate(idx):
            assert np.array_equal(d2[:, :, k], ref[which])


@pytest.mark.parametrize("geom", ["quad", "hex"])
def test_elem_prolongator_rows_match_kvert_ind(geom):
    """the reference enumerates fine nodes through KVERT_IND[i] = (child, local node) (Hexahedron.cpp:49-71);
    the restatement indexes the same rows as P[child, local] -- all 125 (25) fine nodes must be covered and
    rows that denote the same fine node must be identical."""
    kv = G["kvert_ind_" + geom]
    P = fo.elem_prolongator(geom, "biquadratic")
    X = fo.child_node_ref_coords(geom)
    seen = {}
    for (j, i) in kv:
        key = tuple(X[j, i])
        assert key not in seen
        seen[key] = P[j, i]
    assert len(seen) == kv.shape[0]
    for j in range(P.shape[0]):
        for i in range(P.shape[1]):
            assert np.array_equal(seen[tuple(X[j, i])], P[j, i])
    nnz = sum(int(np.count_nonzero(r)) for r in seen.values())
    assert nnz == (729 if geom == "hex" else 81)   # 125 rows / 729 nnz for hex Q2


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built (reference absent on this box)")
def test_against_reference_library_directly():
    L = ctypes.CDLL(REF_SO)
    L.ref_eval.restype = ctypes.c_double
    rng = np.random.default_rng(7)
    pts = rng.uniform(-1, 1, (20, 3))
    phi, dphi, _ = fo.eval_basis("hex", "biquadratic", pts)
    for p in range(pts.shape[0]):
        pt = (ctypes.c_double * 3)(*pts[p])
        for j in range(27):
            assert L.ref_eval(b"hex", b"biquadratic", 0, j, pt) == phi[p, j]
            for d in range(3):
                assert L.ref_eval(b"hex", b"biquadratic", 1 + d, j, pt) == dphi[p, j, d]


# ---- frozen regression vectors of the unpinned parts (SURVEY 8c items ii-vi; see tests/golden/make_regression_vectors.py) ----
R = np.load(os.path.join(HERE, "golden", "path_small.npz"))


@pytest.mark.parametrize("tag,box,fe", [("hex_q2", (2, 2, 2), "biquadratic"), ("quad_q1", (4, 4, 0), "linear")])
def test_oracle_reproduces_frozen_path_vectors(tag, box, fe):
    H = fo.build_poisson_hierarchy(*box, 2, fe, lambda xg: np.ones(xg.shape[:2]))
    mf = H.meshes[1]
    assert np.array_equal(mf.elem_dof, R[tag + "_elem_dof_fine"]) and np.array_equal(H.meshes[0].elem_dof, R[tag + "_elem_dof_coarse"])
    assert np.array_equal(mf.coords, R[tag + "_coords_fine"]) and np.array_equal(mf.face_flag, R[tag + "_face_flag_fine"])
    assert np.array_equal(H.bdc[1], R[tag + "_bdc_fine"])
    P = H.P[1].tocoo()
    assert np.array_equal(P.row, R[tag + "_P_row"]) and np.array_equal(P.col, R[tag + "_P_col"]) and np.array_equal(P.data, R[tag + "_P_val"])
    A = H.A[1].tocsr()
    assert np.array_equal(A.indptr, R[tag + "_A_indptr"]) and np.array_equal(A.indices, R[tag + "_A_indices"])
    assert np.allclose(A.data, R[tag + "_A_data"], rtol=1e-14, atol=1e-16)
    assert np.allclose(H.b, R[tag + "_b"], rtol=1e-14, atol=1e-18)
    x = np.linalg.solve(A.toarray(), H.b)
    assert np.allclose(x, R[tag + "_x_dense_lu"], rtol=1e-11, atol=1e-15)


def test_host_mesh_layer_reproduces_frozen_vectors():
    """the product's host mesh code (no GPU needed) against the same frozen integer tables"""
    from femus_amd import capi
    for tag, box in (("hex_q2", (2, 2, 2)), ("quad_q1", (4, 4, 0))):
        m = capi.Mesh.box(*box)
        f = m.refine()
        ed, xy, ff = f.arrays()
        assert np.array_equal(ed, R[tag + "_elem_dof_fine"]) and np.array_equal(xy, R[tag + "_coords_fine"])
        assert np.array_equal(ff, R[tag + "_face_flag_fine"]) and f.own_size == R[tag + "_own_size_fine"].tolist()
        fe = "biquadratic" if tag == "hex_q2" else "linear"
        assert np.array_equal(f.dirichlet_dofs(fe), R[tag + "_bdc_fine"])
        m.destroy(), f.destroy()
