"""Dev probe: where the FIRST preparation of the bench problem spends its time (host-side setup work).  usage: trace_setup.py [levels]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import femus_amd
from femus_amd import capi
lv = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ctx = femus_amd.Context(0)
T = []
def tick(name, t0):
    ctx.sync()
    T.append((name, time.time() - t0))
    return time.time()
t = time.time()
ms = [capi.Mesh.box(8, 8, 8)]
for _ in range(lv - 1):
    ms.append(ms[-1].refine())
t = tick("meshes (box + refinements)", t)
fe = "biquadratic"
ndof = [m.n_dofs(fe) for m in ms]
bdc = [m.dirichlet_dofs(fe) for m in ms]
t = tick("dirichlet_dofs", t)
P = [None] + [capi.build_prolongator(ctx, ms[l - 1], ms[l], fe, zero_bdc=True) for l in range(1, lv)]
t = tick("build_prolongator (all levels)", t)
A, asm = [], []
for l in range(lv):
    ed, xy, _ = ms[l].arrays()
    t = tick("arrays() level %d" % l, t)
    K = ctx.matrix_from_elements(ed[:, :27], ndof[l])
    t = tick("matrix_from_elements level %d" % l, t)
    asm.append(capi.Assembler(ctx, ms[l], fe, K, "seventh", elem_dof=ed, coords=xy))
    t = tick("Assembler level %d" % l, t)
    A.append(K)
res = ctx.vector(ndof[-1])
asm[-1].assemble(A[-1], res, None, 0, (1.0,))
t = tick("first assembly", t)
bdev = [capi.Index(ctx, b) for b in bdc]
for l in range(lv - 1, 0, -1):
    ch = ms[l - 1].child_elems()
    asm[l - 1].galerkin_from(asm[l], ch, bdc[l], bdc[l - 1], A[l - 1])
    t = tick("galerkin %d -> %d (first)" % (l, l - 1), t)
for l in range(lv):
    bdev[l].zero_rows(A[l], 1.0)
t = tick("SetPenalty", t)
mg = capi.Multigrid(ctx, lv)
mg.set_coarse_coords(ms[0].arrays()[1][:ndof[0]])
for l in range(lv):
    mg.set_level(l, A[l], P[l], None, 0, 2. / 3., 2 if l else 1, 2 if l else 0)
t = tick("mg set_level", t)
mg.setup()
t = tick("mg.setup (first)", t)
mg.setup()
t = tick("mg.setup (second)", t)
x = ctx.vector(ndof[-1])
mg.vcycle(res, x)
t = tick("first cycle", t)
for name, dt in T:
    if dt > 0.002:
        print("%-40s %8.3f s" % (name, dt))
print("total %.3f s" % sum(dt for _, dt in T))
