"""Generates tests/golden/path_small.npz: the items of SURVEY 8(c) that cannot come from a reference run (mesh layer, sparsity,
prolongator, assembled system, dense-LU solution need PETSc / generated headers) written down ONCE from the oracle and frozen, so
that a later change of the oracle or of the product shows up as a diff against a committed artefact.  These vectors are NOT a
pin to the reference (the header of oracle/femus_oracle.py says "parity unpinned" for these parts); the reference-pinned fixture
is fe_tables.npz.   python tests/golden/make_regression_vectors.py"""
import os
import sys

import numpy as np
import scipy.sparse.linalg as spla

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import femus_oracle as fo  # noqa: E402

ONE = lambda xg: np.ones(xg.shape[:2])
out = {}
for tag, box, fe in (("hex_q2", (2, 2, 2), "biquadratic"), ("quad_q1", (4, 4, 0), "linear")):
    H = fo.build_poisson_hierarchy(*box, 2, fe, ONE)
    mc, mf = H.meshes
    out[tag + "_elem_dof_coarse"] = mc.elem_dof.astype(np.int32)
    out[tag + "_elem_dof_fine"] = mf.elem_dof.astype(np.int32)
    out[tag + "_coords_fine"] = mf.coords
    out[tag + "_face_flag_fine"] = mf.face_flag.astype(np.int32)
    out[tag + "_own_size_fine"] = np.array(mf.own_size, np.int32)
    out[tag + "_bdc_fine"] = H.bdc[1].astype(np.int32)
    P = H.P[1].tocoo()
    out[tag + "_P_row"], out[tag + "_P_col"], out[tag + "_P_val"] = P.row.astype(np.int32), P.col.astype(np.int32), P.data
    A = H.A[1].tocsr()
    out[tag + "_A_indptr"], out[tag + "_A_indices"], out[tag + "_A_data"] = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data
    out[tag + "_b"] = H.b
    out[tag + "_b_before_penalty"] = H.b_raw
    out[tag + "_x_dense_lu"] = np.linalg.solve(H.A[1].toarray(), H.b)
    et = fo.ElemType(mf.geom, fe)
    X = np.transpose(mf.coords[mf.elem_dof[:1]], (0, 2, 1))
    K, F = fo.elem_poisson_batch(et, X, np.zeros((1, et.nc)), ONE)
    out[tag + "_K_elem0"], out[tag + "_F_elem0"] = K[0], F[0]
    out[tag + "_elem_prolongator"] = fo.elem_prolongator(mf.geom, fe)
np.savez_compressed(os.path.join(HERE, "path_small.npz"), **out)
print("wrote", len(out), "arrays")
