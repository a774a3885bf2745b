"""dev: one cycle of config 4 with and without the slot maps of the sparse products: how far apart are the results, level by level?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import femus_amd
from femus_amd import capi
from femus_amd.navier_stokes import NavierStokesMG
res = {}
for opt in (0, 1):
    ctx = femus_amd.Context(0)
    ctx.set_option("spgemm_slot_map", opt)
    nl = 4
    pb = NavierStokesMG(ctx, 10, 10, 0, nl, 0.01).init()
    for v in [0.01, 0.004, 0.002, 0.001]:
        pb.nu = v
        assert pb.newton(0, tol=1e-10, max_newton=25)
    for ig in range(1, nl):
        pb.prolongator_sol(ig)
        assert pb.newton(ig, tol=1e-10, max_newton=25, lin_rtol=1e-10, lin_maxit=200)
    top = nl - 1
    pb.asm[top].assemble(pb.KK[top], pb.RES[top], pb.SOL[top], pb.nu)
    mg = pb.prepare(top)
    x = ctx.vector(pb.n[top])
    mg.vcycle(pb.RES[top], x)
    its, rn = mg.solve(pb.RES[top], pb.EPS[top], outer="gmres", rtol=1e-10, maxit=200)
    res[opt] = dict(sol=pb.SOL[top].to_numpy().copy(), rhs=pb.RES[top].to_numpy().copy(), x=x.to_numpy().copy(), its=its, rn=rn,
                    A=[pb.A[(top, l)].to_scipy().copy() for l in range(nl)])
    if hasattr(mg, "coarse_info"):
        print(opt, mg.coarse_info())
    pb.destroy()
a, b = res[0], res[1]
rel = lambda u, v: np.linalg.norm(u - v) / max(np.linalg.norm(v), 1e-300)
print("iterations", a["its"], b["its"], "final residuals", a["rn"], b["rn"])
print("state SOL", rel(a["sol"], b["sol"]), "rhs", rel(a["rhs"], b["rhs"]), "one cycle", rel(a["x"], b["x"]))
for l in range(4):
    d = abs(a["A"][l] - b["A"][l]).max() / abs(b["A"][l]).max()
    print("level", l, "operators differ by", d)
