"""Timing probe (not a test): one multigrid cycle of the known-answer problem (four levels, finest 69 792 unknowns in about 1 500 dependency levels; GMRES(4) + ILU(0)
level solvers: the cycle is 90 % natural-order sweeps), on the Jacobian of the initial state.  FEMUS_TRI_DBG switches stages of the run kernel off (wrong results, timing only)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import femus_amd
from femus_amd import app_ns_steady_dd as app
from femus_amd import capi

ctx = femus_amd.Context(0)
ms = app.meshes(ctx, n_uniform=4, n_selective=0)
from femus_amd.navier_stokes import NavierStokesPwMG
ns = NavierStokesPwMG(ctx, ms, app.INVERSE_REYNOLDS, app.boundary_condition, level_gmres_its=4).init()      # the Jacobian of the zero state with the boundary values: same pattern, same levels
for ig in range(4):
    xs = np.zeros(ns.n[ig])
    ns.set_state(ig, xs)
mg = ns.prepare(3)
x = ctx.vector(ns.n[3])
for _ in range(3):
    mg.vcycle(ns.RES[3], x)
ctx.sync()
t = time.time()
for _ in range(10):
    mg.vcycle(ns.RES[3], x)
ctx.sync()
print(json.dumps({"dbg": os.environ.get("FEMUS_TRI_DBG", "0"), "cycle_ms": (time.time() - t) / 10 * 1e3}))
ns.destroy()          # (a TRI_STAMP dev build prints its clock stamps when the plans go)
