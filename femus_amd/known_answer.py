"""The reference's own known-answer test for the Navier-Stokes / multigrid path, unittests/testNSSteadyDD/main.cpp, through the device path.

That test (one of the two CTests of the tree that store numbers) solves the flow around a cylinder on input/nsbenc.neu -- Q2 velocity, discontinuous
piecewise-linear pressure, nu = 0.001 -- and asserts the l2 norms of U, V, P on level 3 to 1e-6 (main.cpp:202-244).  run() takes the same mesh file
(tests/golden/nsbenc.neu, a data file of that test) through: Gambit reader -> three refinements on the device -> fh_ns_pw_assembler_create ->
Dirichlet rows -> Newton with the sparse exact solve, and returns the norms beside the stored ones.  No oracle is involved: the numbers to meet are the
reference's.  Used by __graft_entry__.smoke() and reported by bench.py."""
import os

import numpy as np

from . import capi
from .navier_stokes import NavierStokesPwMG

STORED = {"U": 35.68179309424519, "V": 6.86749406268887, "P": 3.10222750612995}
STORED_T = 57.69748694700662          # main.cpp:237: T on level 3 -- the temperature system is solved on the finest level only (V_CYCLE, LinearImplicitSystem.cpp:300-303),
                                      # so level 3 holds Initialize("T") = 0 and the boundary values GenerateBdc wrote: 1 on the inflow, 5 on the cylinder (main.cpp:375-391)
MESH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "nsbenc.neu")


def inflow_profile(y):
    return 1.5 * 0.2 * (4.0 / 0.1681) * y * (0.41 - y)                                   # main.cpp:283-287


def boundary_condition(x, name, face):
    """main.cpp:290-392: faces 1 inflow, 2 outflow (nothing prescribed), 3 walls, 4 cylinder; the pressure carries no condition"""
    if face == 2:
        return False, 0.0
    return True, (inflow_profile(x[1]) if (name == "U" and face == 1) else 0.0)


def run(ctx, mesh_file=MESH):
    m = capi.Mesh.read_gambit(mesh_file)
    for _ in range(3):
        m = m.refine(ctx)
    pb = NavierStokesPwMG(ctx, [m], 0.001, boundary_condition).init()
    x = np.zeros(pb.n[0])
    x[:m.nnode] = inflow_profile(m.arrays()[1][:, 1])
    pb.set_state(0, x)
    ok = pb.mgsolve(tol=1e-10, max_newton=20)
    s = pb.SOL[0].to_numpy()
    nq = m.nnode
    got = {"U": float(np.linalg.norm(s[:nq])), "V": float(np.linalg.norm(s[nq:2 * nq])), "P": float(np.linalg.norm(s[2 * nq:]))}
    from .navier_stokes import generate_bdc
    _, tval = generate_bdc(m, ["T"], ["biquadratic"], np.array([0, nq]), lambda x, name, face: (True, 1.0) if face == 1 else (True, 5.0) if face == 4 else (False, 0.0))
    got["T"] = float(np.linalg.norm(tval))
    stored = dict(STORED, T=STORED_T)
    out = {"test": "unittests/testNSSteadyDD/main.cpp:202-244 (level-3 l2 norms, asserted there to 1e-6)", "converged": bool(ok), "unknowns": int(pb.n[0]),
           "newton_steps": len(pb.history), "norms": got, "stored": stored,
           "max_relative_distance": max(abs(got[k] - stored[k]) / stored[k] for k in got),
           "path": "Gambit reader -> 3 refinements on the device -> Q2 / discontinuous-pressure Navier-Stokes assembly kernel -> sparse exact solve (pivoted fronts) -> Newton"}
    pb.destroy()
    return out
