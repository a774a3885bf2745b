"""dev: first Newton step of the known-answer problem on the device: sparse exact solve against scipy on the same operator"""
import os, sys
import numpy as np
import scipy.sparse.linalg as spla
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import femus_amd
from femus_amd import capi
from oracle import femus_oracle as fo
from test_ns_known_answer import CYLINDER, INFLOW, WALL, inflow_profile, nodes_on
ctx = femus_amd.Context(0)
nref = int(sys.argv[1]) if len(sys.argv) > 1 else 1
m = capi.Mesh.read_gambit("tests/golden/nsbenc.neu")
for _ in range(nref):
    m = m.refine(ctx)
ed, xy, ff = m.arrays()
mo = fo.Mesh("quad", ed, xy, ff, level=nref)
nq2 = m.nnode; n = 2 * nq2 + 3 * m.nel
dn = np.unique(np.concatenate([nodes_on(mo, INFLOW), nodes_on(mo, WALL), nodes_on(mo, CYLINDER)]))
inflow = nodes_on(mo, INFLOW)
bdc = np.concatenate([dn, dn + nq2]).astype(np.int32)
x0 = np.zeros(n); x0[:nq2] = inflow_profile(xy[:, 1]); x0[dn] = 0.0; x0[inflow] = inflow_profile(xy[inflow, 1])
KK = ctx.matrix_from_elements(capi.NSPwAssembler.elem_sys(m), n)
asm = capi.NSPwAssembler(ctx, m, KK)
sol, res, eps = ctx.vector_from(x0), ctx.vector(n), ctx.vector(n)
bidx = capi.Index(ctx, bdc)
cen = xy[ed[:, 8]]
for coords in (np.concatenate([xy, xy, cen, cen, cen]), None):
    d = capi.Direct(ctx, KK, coords)
    asm.assemble(KK, res, sol, 0.001)
    bidx.zero_rows(KK, 1.0); bidx.set(res, 0.0)
    d.factor()
    d.solve(res, eps)
    A = KK.to_scipy(); b = res.to_numpy()
    ref = spla.splu(A.tocsc()).solve(b)
    e = eps.to_numpy()
    print("coords" if coords is not None else "no coords", d.stats(), d.info(), "rel err vs splu", np.linalg.norm(e - ref) / np.linalg.norm(ref),
          "residual", np.linalg.norm(A @ e - b) / np.linalg.norm(b), "by variable", [np.linalg.norm((e - ref)[a:bb]) / max(np.linalg.norm(ref[a:bb]), 1e-300) for a, bb in ((0, nq2), (nq2, 2 * nq2), (2 * nq2, n))])
    d.destroy()
