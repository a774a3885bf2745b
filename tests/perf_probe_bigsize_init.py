"""Probe: the stages of PoissonMG.init at 8^3 -> 128^3 one by one (which one faults / is slow)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import femus_amd as fa
from femus_amd import capi

levels = int(sys.argv[1]) if len(sys.argv) > 1 else 5
ctx = fa.Context(0)
t = [time.time()]


def mark(name):
    ctx.sync()
    print("%-28s %8.2f s" % (name, time.time() - t[0]), flush=True)
    t[0] = time.time()


meshes = [capi.Mesh.box(8, 8, 8)]
for _ in range(1, levels):
    meshes.append(meshes[-1].refine())
mark("meshes")
fe = "biquadratic"
ndof = [m.n_dofs(fe) for m in meshes]
bdc = [m.dirichlet_dofs(fe) for m in meshes]
mark("dirichlet")
P = [None]
for l in range(1, levels):
    P.append(capi.build_prolongator(ctx, meshes[l - 1], meshes[l], fe, zero_bdc=True))
    mark("prolongator %d" % l)
ed, xy, _ = meshes[-1].arrays()
mark("arrays")
rp, col = capi.pattern_from_elements(ed[:, :27], ndof[-1])
mark("pattern nnz=%d" % col.size)
K = ctx.matrix_csr(ndof[-1], ndof[-1], rp, col)
mark("matrix_csr")
asm = capi.Assembler(ctx, meshes[-1], fe, K, "seventh", elem_dof=ed, coords=xy)
mark("assembler")
res, sol = ctx.vector(ndof[-1]), ctx.vector(ndof[-1])
asm.assemble(K, res, sol, 0, (1.0,))
mark("assemble")
one, y = ctx.vector(ndof[-1]), ctx.vector(ndof[-1])
one.upload(np.ones(ndof[-1]))
y.matrix_mult(one, K)
mark("spmv")
print("max |A 1| =", np.abs(y.to_numpy()).max(), " sum(res) =", res.to_numpy().sum(), flush=True)
A4 = capi.Mat.ptap(P[-1], K)
mark("ptap top")
