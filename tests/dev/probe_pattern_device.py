"""Dev: fh_mat_create_from_elements on the element tables of the simplex / mixed meshes against the numpy pattern of app_poisson"""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np
import femus_amd
from femus_amd import capi, tet_mesh, mixed_mesh
ctx = femus_amd.Context(0)
g = os.path.join(os.path.dirname(HERE), "golden")
lv = tet_mesh.read_gambit(os.path.join(g, "cube_Tet.neu"))
for _ in range(3):
    lv = tet_mesh.refine(*lv[:3])
for nc, ndof in ((4, lv[3][0]), (10, lv[3][1]), (15, lv[3][2])):
    ed = np.ascontiguousarray(lv[0][:, :nc])
    t0 = time.perf_counter()
    r = np.repeat(ed, nc, axis=1).ravel().astype(np.int64); c = np.tile(ed, (1, nc)).ravel().astype(np.int64)
    key = np.unique(r * ndof + c); rows, cols = key // ndof, key % ndof
    indptr = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=ndof))])
    t1 = time.perf_counter()
    try:
        K = capi.Mat.from_elements(ctx, ed, ndof)
        ctx.sync(); t2 = time.perf_counter()
        A = K.to_scipy()
        print("tet nc", nc, "ndof", ndof, "numpy %.3f device %.3f" % (t1 - t0, t2 - t1), "same", np.array_equal(A.indptr, indptr) and np.array_equal(A.indices, cols), "max row", np.diff(indptr).max())
        K.destroy()
    except Exception as e:
        print("tet nc", nc, "device failed:", str(e)[:200])
m = mixed_mesh.read_gambit(os.path.join(g, "cube_all_shapes_Six_boundary_groups.neu"))
for _ in range(3):
    m = mixed_mesh.refine(*m[:4])
ed = m[1].copy()
ndof = m[4][2]
pad = ed < 0
ed[pad] = np.broadcast_to(ed[:, :1], ed.shape)[pad]
keys = []
for s in ("hex", "tet", "wedge"):
    e2 = m[1][m[0] == s][:, :mixed_mesh.NLOC[s]]
    nc = e2.shape[1]
    keys.append(np.repeat(e2, nc, axis=1).ravel().astype(np.int64) * ndof + np.tile(e2, (1, nc)).ravel())
key = np.unique(np.concatenate(keys)); rows, cols = key // ndof, key % ndof
indptr = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=ndof))])
try:
    K = capi.Mat.from_elements(ctx, ed, ndof)
    A = K.to_scipy()
    print("mixed ndof", ndof, "same", np.array_equal(A.indptr, indptr) and np.array_equal(A.indices, cols), "max row", np.diff(indptr).max())
except Exception as e:
    print("mixed device failed:", str(e)[:200])
