"""dev: the known-answer problem through the multigrid path (nonlinear F-cycle, GMRES + ILU(0) level solvers or element-block Vanka)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import femus_amd
from femus_amd import capi
from femus_amd.navier_stokes import NavierStokesPwMG
from test_ns_known_answer import STORED, inflow_profile
ctx = femus_amd.Context(0)
ms = [capi.Mesh.read_gambit("tests/golden/nsbenc.neu")]
for _ in range(3):
    ms.append(ms[-1].refine(ctx))

def bc(x, name, face):          # main.cpp:290-392: faces 1 inflow, 2 outflow, 3 walls, 4 cylinder
    if face == 2:
        return False, 0.0
    return True, (inflow_profile(x[1]) if (name == "U" and face == 1) else 0.0)

smoother = sys.argv[1] if len(sys.argv) > 1 else "ilu"
pb = NavierStokesPwMG(ctx, ms, 0.001, bc, level_gmres_its=int(sys.argv[2]) if len(sys.argv) > 2 else 4).init()
if smoother == "vanka":
    pb.smoother_kind = "vanka"
m0 = ms[0]
x = np.zeros(pb.n[0]); x[:m0.nnode] = inflow_profile(m0.arrays()[1][:, 1])
pb.set_state(0, x)
t = time.time()
ok = pb.mgsolve(tol=1e-10, max_newton=20, lin_rtol=1e-10, lin_maxit=100)
print("converged", ok, "%.2f s" % (time.time() - t))
for h in pb.history: print(h)
s = pb.SOL[3].to_numpy(); nq = ms[3].nnode
got = {"U": np.linalg.norm(s[:nq]), "V": np.linalg.norm(s[nq:2 * nq]), "P": np.linalg.norm(s[2 * nq:])}
print(got, {k: abs(got[k] - STORED[k]) / STORED[k] for k in got})
