"""Dev probe: factorisation / solve times of the sparse exact solve on penalised Q2 Poisson operators.  usage: perf_probe_direct.py [levels ...] [leaf]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import femus_amd
from femus_amd import capi
from femus_amd.poisson import PoissonMG

ctx = femus_amd.Context(0)
leaf = int(os.environ.get("LEAF", "0"))
for lv in [int(v) for v in sys.argv[1:]] or [2, 3]:
    pb = PoissonMG(ctx, 4, 4, 4, lv + 1, coarse="rediscretise").init()          # level lv of a 4^3 coarse box: (8 * 2^(lv-1) + 1)^3 nodes ...
    for l in range(pb.nlevels):
        pb.assemble(l)
    pb.level_operators() if hasattr(pb, "level_operators") else None
    A = pb.A[lv]
    n = A.m()
    xy = pb.meshes[lv].arrays()[1][:n]
    for with_xy in (True, False):
        t0 = time.time()
        d = capi.Direct(ctx, A, xy if with_xy else None, leaf)
        d.factor()
        ctx.sync()
        t_first = time.time() - t0
        ctx.timer_start()
        for _ in range(3): d.factor()
        t_fac = ctx.timer_stop() / 3
        b, x = ctx.vector_from(np.ones(n)), ctx.vector(n)
        d.solve(b, x)
        ctx.timer_start()
        for _ in range(10): d.solve(b, x)
        t_sol = ctx.timer_stop() / 10
        r = ctx.vector(n)
        r.matrix_mult(x, A)
        res = np.linalg.norm(r.to_numpy() - 1.0) / np.sqrt(n)
        print("n %7d  coords %-5s  first %.2f s  refactor %.2f ms  solve %.3f ms  residual %.1e  %s" % (n, with_xy, t_first, t_fac, t_sol, res, d.info()), flush=True)
        d.destroy()
    pb.destroy()
