"""dev: the pieces of LinearImplicitSystem::init of the bench problem, level by level (warm: second repetition)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import femus_amd
from femus_amd import capi
ctx = femus_amd.Context(0)
fe = "biquadratic"
for rep in range(2):
    ms = [capi.Mesh.box(4, 4, 4)]
    for _ in range(4):
        ms.append(ms[-1].refine(ctx))
    ctx.sync()
    def T(label, fn):
        ctx.sync(); t = time.perf_counter(); r = fn(); ctx.sync()
        if rep: print("%-44s %7.2f ms" % (label, (time.perf_counter() - t) * 1e3))
        return r
    for l in range(1, 5):
        T("dirichlet_dofs level %d" % l, lambda: ms[l].dirichlet_dofs(fe))
    Ps = [T("build_prolongator level %d" % l, lambda: capi.build_prolongator(ctx, ms[l - 1], ms[l], fe, zero_bdc=True)) for l in range(1, 5)]
    Ks = [T("matrix_from_mesh level %d" % l, lambda: ctx.matrix_from_mesh(ms[l], fe)) for l in range(5)]
    As = [T("Assembler level %d" % l, lambda: capi.Assembler(ctx, ms[l], fe, Ks[l], "seventh")) for l in range(5)]
    for o in As + Ks + Ps + ms:
        o.destroy()
