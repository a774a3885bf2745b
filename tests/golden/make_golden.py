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

rng = np.random.default_rng(20260929)
for geom, dim, nv in (("quad", 2, 4), ("hex", 3, 8)):
    sample = rng.uniform(-1, 1, (7, dim))
    out["sample_pts_%s" % geom] = sample
    for fe in ("linear", "biquadratic"):
        nc = L.ref_ndofs(geom.encode(), fe.encode())
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
                        vals[which, p, j] = L.ref_eval(geom.encode(), fe.encode(), which, j, pt)
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
    nf = L.ref_ndofs_fine(geom.encode(), b"biquadratic")
    kv = np.zeros((nf, 2), dtype=np.int64)
    for i in range(nf):
        b = (ctypes.c_int * 2)()
        L.ref_kvert_ind(geom.encode(), b"biquadratic", i, b)
        kv[i] = list(b)
    out["kvert_ind_%s" % geom] = kv

np.savez_compressed(os.path.join(HERE, "fe_tables.npz"), **out)
print("wrote", os.path.join(HERE, "fe_tables.npz"), len(out), "arrays")
