"""Timing probe (not a test): the path that carries the parity pin -- unittests/testNSSteadyDD through the device path.
  (a) the whole application as the reference runs it (femus_amd.app_ns_steady_dd: six levels, its iteration limits, the temperature system): wall time per part
  (b) the four uniform levels through the nonlinear F-cycle to convergence (GMRES + ILU(0) level solvers, exact coarse solve): wall time, Newton / outer iterations
  (c) the sparse exact solve of level 3 (femus_amd.known_answer.run, what smoke() runs)
Prints one JSON line.  Under `rocprofv3 --kernel-trace --stats` (tests/profile_known_answer.sh) the kernel table of the same run goes to profiles/."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import femus_amd
from femus_amd import app_ns_steady_dd as app
from femus_amd import known_answer as ka

ctx = femus_amd.Context(0)
part = sys.argv[1] if len(sys.argv) > 1 else "all"
out = {}


def wall(fn):
    ctx.sync()
    t = time.time()
    r = fn()
    ctx.sync()
    return r, time.time() - t


if part in ("all", "warm"):
    ka.run(ctx)                                           # first use of every kernel (code objects, workspaces)
if part in ("all", "app"):
    ms, t_mesh = wall(lambda: app.meshes(ctx))
    ns, t_ns = wall(lambda: app.navier_stokes(ctx, ms, True))
    top = len(ms) - 1
    (tp, hist, _), t_t = wall(lambda: app.temperature(ctx, ms, ns.SOL[top]))
    out["application"] = {"elements": [m.nel for m in ms], "unknowns_finest": int(ns.n[top]), "meshes_s": t_mesh, "navier_stokes_fcycle_s": t_ns, "temperature_s": t_t,
                          "newton_steps": len(ns.history), "outer_iterations": int(sum(h[3] for h in ns.history)),
                          "what": "six levels (four uniform + two selective), the test's own limits: 3 Newton steps per level to 1e-4, 2 cycles of 4 outer iterations, one smoothing step"}
    tp.meshes = []
    tp.destroy()
    ns.destroy()
if part in ("all", "fcycle"):
    ms = app.meshes(ctx, n_uniform=4, n_selective=0)
    ns, t_ns = wall(lambda: app.navier_stokes(ctx, ms, False))
    s = ns.SOL[3].to_numpy()
    nq = ms[3].nnode
    got = {"U": float(np.linalg.norm(s[:nq])), "V": float(np.linalg.norm(s[nq:2 * nq])), "P": float(np.linalg.norm(s[2 * nq:]))}
    # one preconditioner application and one ILU(0) sweep pair of the finest level, timed alone
    mg = ns.prepare(3)
    x = ctx.vector(ns.n[3])
    for _ in range(3):
        mg.vcycle(ns.RES[3], x)
    _, t_cyc = wall(lambda: [mg.vcycle(ns.RES[3], x) for _ in range(10)])
    out["fcycle_four_levels"] = {"unknowns": int(ns.n[3]), "wall_s": t_ns, "newton_steps_per_level": [sum(1 for h in ns.history if h[0] == l) for l in range(4)],
                                 "outer_iterations_finest": [h[3] for h in ns.history if h[0] == 3], "cycle_ms": t_cyc / 10 * 1e3,
                                 "max_relative_distance_to_stored_norms": max(abs(got[k] - ka.STORED[k]) / ka.STORED[k] for k in got)}
    ns.destroy()
if part in ("all", "direct"):
    r, t = wall(lambda: ka.run(ctx))
    out["exact_solve_level_3"] = {"wall_s": t, "newton_steps": r["newton_steps"], "unknowns": r["unknowns"], "max_relative_distance_to_stored_norms": r["max_relative_distance"]}
print(json.dumps(out))
