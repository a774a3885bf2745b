"""timing probe (not a test): the sparse exact solve on GENERAL (pivoted) fronts -- Navier-Stokes Jacobians (Taylor-Hood: empty pressure block) of 40 x 40 and
80 x 80 levels, with and without coordinates; first factorisation (symbolic + numeric), re-factorisation, solve, residual against the operator"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import femus_amd
from femus_amd import capi
from oracle import femus_oracle as fo
from oracle import femus_oracle_ns as ns

ctx = femus_amd.Context(0)
for nx in (40, 80):
    ms, lays = ns.build_ns_levels(nx, nx, 0, 1, (-0.5, -0.5, 0.0), (0.5, 0.5, 0.0))
    m, lay = ms[0], lays[0]
    bc = ns.cavity_bc(m, lay)
    u = 0.3 * np.random.default_rng(1).standard_normal(lay.n)
    A, b = ns.assemble_ns(m, lay, u, 0.01)
    A = fo.zero_rows(A.tocsr(), bc[0], 1.0).tocsr()
    A.sort_indices()
    xy = np.concatenate([m.coords[:sz, :2] for sz in lay.sizes])
    M = ctx.matrix_scipy(A)
    for coords in (True, False):
        d = capi.Direct(ctx, M, xy if coords else None)
        ctx.sync(); t = time.time(); d.factor(); ctx.sync(); t_first = time.time() - t
        ts = []
        for _ in range(3):
            ctx.sync(); t = time.time(); d.factor(); ctx.sync(); ts.append(time.time() - t)
        rhs = np.random.default_rng(2).uniform(-1, 1, lay.n)
        bb, x = ctx.vector_from(rhs), ctx.vector(lay.n)
        d.solve(bb, x)
        ctx.sync(); t = time.time()
        for _ in range(10):
            d.solve(bb, x)
        ctx.sync(); t_solve = (time.time() - t) / 10
        r = np.linalg.norm(A @ x.to_numpy() - rhs) / np.linalg.norm(rhs)
        print("n %6d  coords %-5s  first %.2f s  refactor %.1f ms  solve %.3f ms  residual %.1e  %s %s" % (lay.n, coords, t_first, sorted(ts)[1] * 1e3, t_solve * 1e3, r, d.info(), d.stats()), flush=True)
        d.destroy()
    M.destroy()
