"""Generates tests/golden/fe_tables.npz from the reference's own compiled FE/quadrature sources
(oracle/_ref/libfemus_ref_fe.so, see oracle/Makefile target _ref).  Run in the build container only
(/root/reference must exist):   make -C oracle _ref && python tests/golden/make_golden.py
The fixture is data: Gauss tables, basis values/derivatives at the Gauss points and at fixed sample
points, node tables.  No reference source text is stored."""
import ctypes
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
L = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libfemus_ref_fe.so"))
L.ref_eval.restype = ctypes.c_double
L.ref_geomelem_embedding.restype = ctypes.c_double

out = {}
ORDERS = ["first", "third", "fifth", "seventh", "ninth"]
for geom, dim in (("line", 1), ("quad", 2), ("hex", 3)):
    for order in ORDERS:
        ng = L.ref_gauss(geom.encode(), order.encode(), dim, None, None)
        w = np.zeros(ng)
        x = np.zeros((dim, ng))
        L.ref_gauss(geom.encode(), order.encode(), dim, w.ctypes.data_as(ctypes.c_void_p), x.ctypes.data_as(ctypes.c_void_p))
        out["gauss_w_%s_%s" % (geom, order)] = w
        out["gauss_x_%s_%s" % (geom, order)] = x.T.copy()

REFNAME = {"serendipity": "quadratic"}
rng = np.random.default_rng(20260929)
for geom, dim, nv in (("quad", 2, 4), ("hex", 3, 8)):
    sample = rng.uniform(-1, 1, (7, dim))
    out["sample_pts_%s" % geom] = sample
    for fe in ("linear", "biquadratic", "serendipity", "constant"):
        rfe = REFNAME.get(fe, fe)          # the reference calls the serendipity family "quadratic"
        nc = L.ref_ndofs(geom.encode(), rfe.encode())
        for tag, pts in (("gauss7", out["gauss_x_%s_seventh" % geom]), ("sample", sample)):
            vals = np.zeros((10, pts.shape[0], nc))
            for p in range(pts.shape[0]):
                pt = (ctypes.c_double * 3)(*(list(pts[p]) + [0.0] * (3 - dim)))
                for j in range(nc):
                    for which in range(10):
                        if dim == 2 and which in (3, 6, 8, 9):
                            continue
                        if fe == "linear" and which in (4, 5, 6) and dim == 3:
                            continue  # HexLinear does not implement pure second derivatives
                        if fe == "linear" and which in (4, 5) and dim == 2:
                            continue
                        vals[which, p, j] = L.ref_eval(geom.encode(), rfe.encode(), which, j, pt)
            out["basis_%s_%s_%s" % (geom, fe, tag)] = vals
    nloc = 3 ** dim
    xc = np.zeros((nloc, dim))
    ind = np.zeros((nloc, dim), dtype=np.int64)
    for i in range(nloc):
        b = (ctypes.c_double * 3)()
        L.ref_xcoarse(geom.encode(), b"biquadratic", i, dim, b)
        xc[i] = list(b)[:dim]
        ii = (ctypes.c_int * 3)()
        L.ref_ind(geom.encode(), b"biquadratic", i, dim, ii)
        ind[i] = list(ii)[:dim]
    out["xc_%s" % geom] = xc
    out["ind_%s" % geom] = ind
    out["f2c_%s" % geom] = np.array([[L.ref_fine2coarse_vertex(geom.encode(), b"linear", j, v) for v in range(nv)] for j in range(nv)])
    nfd = 9 if geom == "hex" else 3
    out["facedofs_%s" % geom] = np.array([[L.ref_face_dof(geom.encode(), b"biquadratic", f, k) for k in range(nfd)] for f in range(2 * dim)])
    nfs = 8 if geom == "hex" else 3        # face nodes of the serendipity family (QUAD8 / EDGE3)
    out["facedofs_%s_serendipity" % geom] = np.array([[L.ref_face_dof(geom.encode(), b"quadratic", f, k) for k in range(nfs)] for f in range(2 * dim)])
    nf = L.ref_ndofs_fine(geom.encode(), b"biquadratic")
    kv = np.zeros((nf, 2), dtype=np.int64)
    for i in range(nf):
        b = (ctypes.c_int * 2)()
        L.ref_kvert_ind(geom.encode(), b"biquadratic", i, b)
        kv[i] = list(b)
    out["kvert_ind_%s" % geom] = kv

# EDGE3 (round 6: the one-dimensional input of 001_Poisson): LineLinear / LineBiquadratic at the line Gauss points and at sample points: phi, d/dx, d2/dx2
sample1 = rng.uniform(-1, 1, (7, 1))
out["sample_pts_line"] = sample1
for fe in ("linear", "biquadratic"):
    nc = L.ref_ndofs(b"line", fe.encode())
    for tag, pts in (("gauss7", out["gauss_x_line_seventh"]), ("sample", sample1)):
        vals = np.zeros((3, pts.shape[0], nc))
        for p in range(pts.shape[0]):
            pt = (ctypes.c_double * 3)(float(pts[p, 0]), 0.0, 0.0)
            for j in range(nc):
                for k, which in enumerate((0, 1, 4)):
                    vals[k, p, j] = L.ref_eval(b"line", fe.encode(), which, j, pt)
        out["basis_line_%s_%s" % (fe, tag)] = vals
xc1 = np.zeros((3, 1))
for i in range(3):
    b = (ctypes.c_double * 3)()
    L.ref_xcoarse(b"line", b"biquadratic", i, 1, b)
    xc1[i, 0] = b[0]
out["xc_line"] = xc1

# TRI7 (round 6): the triangle's Gauss rules, TriLinear / TriQuadratic / TriBiquadratic at the 'seventh' points and at sample points (phi, dx, dy, dxx, dyy, dxy),
# node table, selectors, children, edge nodes
for order in ORDERS:
    ng = L.ref_gauss(b"tri", order.encode(), 2, None, None)
    w = np.zeros(ng)
    x = np.zeros((2, ng))
    L.ref_gauss(b"tri", order.encode(), 2, w.ctypes.data_as(ctypes.c_void_p), x.ctypes.data_as(ctypes.c_void_p))
    out["gauss_w_tri_%s" % order] = w
    out["gauss_x_tri_%s" % order] = x.T.copy()
sample_t = rng.uniform(0, 0.5, (7, 2))
out["sample_pts_tri"] = sample_t
for fe in ("linear", "serendipity", "biquadratic"):
    rfe = REFNAME.get(fe, fe)
    nc = L.ref_ndofs(b"tri", rfe.encode())
    for tag, pts in (("gauss7", out["gauss_x_tri_seventh"]), ("sample", sample_t)):
        vals = np.zeros((6, pts.shape[0], nc))
        for p in range(pts.shape[0]):
            pt = (ctypes.c_double * 3)(float(pts[p, 0]), float(pts[p, 1]), 0.0)
            for j in range(nc):
                for k, which in enumerate((0, 1, 2, 4, 5, 7)):
                    if fe == "linear" and which in (4, 5, 7):
                        continue               # TriLinear implements no second derivatives
                    vals[k, p, j] = L.ref_eval(b"tri", rfe.encode(), which, j, pt)
        out["basis_tri_%s_%s" % (fe, tag)] = vals
xct = np.zeros((7, 2))
indt = np.zeros((7, 2), dtype=np.int64)
for i in range(7):
    b = (ctypes.c_double * 3)()
    L.ref_xcoarse(b"tri", b"biquadratic", i, 2, b)
    xct[i] = list(b)[:2]
    ii = (ctypes.c_int * 3)()
    L.ref_ind(b"tri", b"biquadratic", i, 2, ii)
    indt[i] = list(ii)[:2]
out["xc_tri"] = xct
out["ind_tri"] = indt
out["f2c_tri"] = np.array([[L.ref_fine2coarse_vertex(b"tri", b"linear", j, v) for v in range(3)] for j in range(4)])
out["facedofs_tri"] = np.array([[L.ref_face_dof(b"tri", b"biquadratic", f, k) for k in range(3)] for f in range(3)])

# TET10 (round 6): the tetrahedron's Gauss rules, TetLinear / TetQuadratic at the 'seventh' points and at sample points (all ten derivatives slots), node table,
# selectors, children, face nodes (the first six of faceDofs: TRI6 order)
for order in ORDERS:
    ng = L.ref_gauss(b"tet", order.encode(), 3, None, None)
    w = np.zeros(ng)
    x = np.zeros((3, ng))
    L.ref_gauss(b"tet", order.encode(), 3, w.ctypes.data_as(ctypes.c_void_p), x.ctypes.data_as(ctypes.c_void_p))
    out["gauss_w_tet_%s" % order] = w
    out["gauss_x_tet_%s" % order] = x.T.copy()
sample_q = rng.uniform(0, 0.3, (7, 3))
out["sample_pts_tet"] = sample_q
for fe in ("linear", "serendipity"):
    rfe = REFNAME.get(fe, fe)
    nc = L.ref_ndofs(b"tet", rfe.encode())
    for tag, pts in (("gauss7", out["gauss_x_tet_seventh"]), ("sample", sample_q)):
        vals = np.zeros((10, pts.shape[0], nc))
        for p in range(pts.shape[0]):
            pt = (ctypes.c_double * 3)(*[float(v) for v in pts[p]])
            for j in range(nc):
                for which in range(10):
                    if fe == "linear" and which >= 4:
                        continue
                    vals[which, p, j] = L.ref_eval(b"tet", rfe.encode(), which, j, pt)
        out["basis_tet_%s_%s" % (fe, tag)] = vals
xcq = np.zeros((10, 3))
indq = np.zeros((10, 3), dtype=np.int64)
for i in range(10):
    b = (ctypes.c_double * 3)()
    L.ref_xcoarse(b"tet", b"quadratic", i, 3, b)
    xcq[i] = list(b)
    ii = (ctypes.c_int * 3)()
    L.ref_ind(b"tet", b"quadratic", i, 3, ii)
    indq[i] = list(ii)
out["xc_tet"] = xcq
out["ind_tet"] = indq
out["f2c_tet"] = np.array([[L.ref_fine2coarse_vertex(b"tet", b"linear", j, v) for v in range(4)] for j in range(8)])
out["facedofs_tet"] = np.array([[L.ref_face_dof(b"tet", b"quadratic", f, k) for k in range(6)] for f in range(4)])
# TET15 (round 6): TetBiquadratic (values and first derivatives), the fifteen nodes, the seven nodes of a face
for tag, pts in (("gauss7", out["gauss_x_tet_seventh"]), ("sample", sample_q)):
    vals = np.zeros((4, pts.shape[0], 15))
    for p in range(pts.shape[0]):
        pt = (ctypes.c_double * 3)(*[float(v) for v in pts[p]])
        for j in range(15):
            for which in range(4):
                vals[which, p, j] = L.ref_eval(b"tet", b"biquadratic", which, j, pt)
    out["basis_tet_biquadratic_%s" % tag] = vals
xcq15 = np.zeros((15, 3))
for i in range(15):
    b = (ctypes.c_double * 3)()
    L.ref_xcoarse(b"tet", b"biquadratic", i, 3, b)
    xcq15[i] = list(b)
out["xc_tet15"] = xcq15
out["facedofs_tet15"] = np.array([[L.ref_face_dof(b"tet", b"biquadratic", f, k) for k in range(7)] for f in range(4)])

# WEDGE21 (round 6): the prism's Gauss rules, WedgeLinear / WedgeQuadratic / WedgeBiquadratic at the 'seventh' points and at sample points, node table, selectors,
# children, face nodes
for order in ORDERS:
    ng = L.ref_gauss(b"wedge", order.encode(), 3, None, None)
    w = np.zeros(ng)
    x = np.zeros((3, ng))
    L.ref_gauss(b"wedge", order.encode(), 3, w.ctypes.data_as(ctypes.c_void_p), x.ctypes.data_as(ctypes.c_void_p))
    out["gauss_w_wedge_%s" % order] = w
    out["gauss_x_wedge_%s" % order] = x.T.copy()
sample_w = np.concatenate([rng.uniform(0, 0.45, (7, 2)), rng.uniform(-1, 1, (7, 1))], axis=1)
out["sample_pts_wedge"] = sample_w
for fe in ("linear", "serendipity", "biquadratic"):
    rfe = REFNAME.get(fe, fe)
    nc = L.ref_ndofs(b"wedge", rfe.encode())
    for tag, pts in (("gauss7", out["gauss_x_wedge_seventh"]), ("sample", sample_w)):
        vals = np.zeros((10, pts.shape[0], nc))
        for p in range(pts.shape[0]):
            pt = (ctypes.c_double * 3)(*[float(v) for v in pts[p]])
            for j in range(nc):
                for which in range(10):
                    if fe != "biquadratic" and which >= 4:
                        continue
                    vals[which, p, j] = L.ref_eval(b"wedge", rfe.encode(), which, j, pt)
        out["basis_wedge_%s_%s" % (fe, tag)] = vals
xcw = np.zeros((21, 3))
indw = np.zeros((21, 3), dtype=np.int64)
for i in range(21):
    b = (ctypes.c_double * 3)()
    L.ref_xcoarse(b"wedge", b"biquadratic", i, 3, b)
    xcw[i] = list(b)
    ii = (ctypes.c_int * 3)()
    L.ref_ind(b"wedge", b"biquadratic", i, 3, ii)
    indw[i] = list(ii)
out["xc_wedge"] = xcw
out["ind_wedge"] = indw
out["f2c_wedge"] = np.array([[L.ref_fine2coarse_vertex(b"wedge", b"linear", j, v) for v in range(6)] for j in range(8)])
out["facedofs_wedge"] = np.array([[L.ref_face_dof(b"wedge", b"biquadratic", f, k) if (f < 3 or k < 7) else -1 for k in range(9)] for f in range(5)])

# element prolongator as elem_type forms it.  ElemType.cpp itself needs boost and is not compiled, so its two loops are followed
# here on top of the COMPILED basis classes (every number below comes out of a call into the reference's object code):
#   (1) set_fine_coordinates_in_Basis_object (ElemType.cpp:404-432): fine node i = (child, vertex) = KVERT_IND[i] of the linear
#       element; X[i] = sum_k phi^lin_k(Xcoarse[vertex]) * Xcoarse[fine2CoarseVertexMapping[child][k]]
#   (2) set_prolongation_OneElement_All_FE (ElemType.cpp:439-532): P[i][j] = phi_j(X[i]), entries with |.| < 1e-14 dropped
def _xcoarse(geom, fe, i, dim):
    b = (ctypes.c_double * 3)()
    L.ref_xcoarse(geom.encode(), fe.encode(), i, dim, b)
    return np.array(list(b)[:dim])


for geom, dim in (("quad", 2), ("hex", 3)):
    nlin = L.ref_ndofs(geom.encode(), b"linear")
    for fe in ("linear", "biquadratic", "serendipity"):
        rfe = REFNAME.get(fe, fe)
        nc = L.ref_ndofs(geom.encode(), rfe.encode())
        nf = L.ref_ndofs_fine(geom.encode(), rfe.encode())
        X = np.zeros((nf, dim))
        P = np.zeros((nf, nc))
        kv = np.zeros((nf, 2), dtype=np.int64)
        for i in range(nf):
            k2 = (ctypes.c_int * 2)()
            L.ref_kvert_ind(geom.encode(), b"linear", i, k2)
            child, vertex = k2[0], k2[1]
            kv[i] = [child, vertex]
            xvtx = _xcoarse(geom, "linear", vertex, dim)
            pt = (ctypes.c_double * 3)(*(list(xvtx) + [0.0] * (3 - dim)))
            xm = np.zeros(dim)
            for k in range(nlin):
                xv = _xcoarse(geom, "linear", L.ref_fine2coarse_vertex(geom.encode(), b"linear", child, k), dim)
                xm += L.ref_eval(geom.encode(), b"linear", 0, k, pt) * xv
            X[i] = xm
            ptx = (ctypes.c_double * 3)(*(list(xm) + [0.0] * (3 - dim)))
            for j in range(nc):
                v = L.ref_eval(geom.encode(), rfe.encode(), 0, j, ptx)
                P[i, j] = v if abs(v) >= 1.0e-14 else 0.0
        out["xfine_%s_%s" % (geom, fe)] = X
        out["elem_prol_%s_%s" % (geom, fe)] = P
        out["kvert_ind_%s_%s" % (geom, fe)] = kv
    # the discontinuous families carry their own fine points (hex_const::X, quad_const::X) and (child, function) table; the piecewise constant one:
    # P[i][0] = phi_0(X[i]) = 1 for each of the 2^dim children
    nf0 = L.ref_ndofs_fine(geom.encode(), b"constant")
    X0 = np.zeros((nf0, dim))
    kv0 = np.zeros((nf0, 2), dtype=np.int64)
    P0 = np.zeros((nf0, 1))
    for i in range(nf0):
        b = (ctypes.c_double * 3)()
        L.ref_xfine(geom.encode(), b"constant", i, dim, b)
        X0[i] = list(b)[:dim]
        k2 = (ctypes.c_int * 2)()
        L.ref_kvert_ind(geom.encode(), b"constant", i, k2)
        kv0[i] = [k2[0], k2[1]]
        ptx = (ctypes.c_double * 3)(*(list(X0[i]) + [0.0] * (3 - dim)))
        v = L.ref_eval(geom.encode(), b"constant", 0, 0, ptx)
        P0[i, 0] = v if abs(v) >= 1.0e-14 else 0.0
    out["xfine_%s_constant" % geom] = X0
    out["elem_prol_%s_constant" % geom] = P0
    out["kvert_ind_%s_constant" % geom] = kv0

# GeomElem* topology tables of the compiled 00_definition sources: sizes, face -> nodes, float embedding matrices
for geom, fam, tag in (("hex", 2, "hex27"), ("quad", 2, "quad9")):
    dim, nn, nl, nfc = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    L.ref_geomelem_info(geom.encode(), fam, ctypes.byref(dim), ctypes.byref(nn), ctypes.byref(nl), ctypes.byref(nfc))
    out["geomelem_info_%s" % tag] = np.array([dim.value, nn.value, nl.value, nfc.value])
    faces = []
    for f in range(nfc.value):
        buf = (ctypes.c_uint * 16)()
        k = L.ref_geomelem_face_nodes(geom.encode(), fam, f, buf)
        faces.append(list(buf)[:k])
    out["geomelem_faces_%s" % tag] = np.array(faces, dtype=np.int64)
    if tag != "quad9":      # GeomElemHex27's deprecated embedding matrix uses another coarse node numbering: not a table of this path
        continue
    nch = 2 ** dim.value
    E = np.zeros((nch, nn.value, nn.value))
    for c in range(nch):
        for i in range(nn.value):
            for j in range(nn.value):
                E[c, i, j] = L.ref_geomelem_embedding(geom.encode(), fam, c, i, j)
    out["geomelem_embedding_%s" % tag] = E

np.savez_compressed(os.path.join(HERE, "fe_tables.npz"), **out)
print("wrote", os.path.join(HERE, "fe_tables.npz"), len(out), "arrays")
