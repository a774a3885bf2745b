"""Dev stress: assemble + prepare (Galerkin chain, coarse inverse, graph capture) + one V-cycle + a GMRES solve, repeated from
identical inputs, must give bit-identical results; any race in those kernels shows as a run-to-run difference.
usage: stress_prepare_determinism.py [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import femus_amd
from femus_amd import capi
from femus_amd.poisson import PoissonMG

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ctx = femus_amd.Context(0)
bad = 0


def ex4_flag(x, level):
    return bool(x[0] > 0.5 and (level < 2 or x[1] > 0.25))


def amr_meshes(box, nu, ns):
    ms = [capi.Mesh.box(*box)]
    for l in range(1, nu + ns):
        flags = np.ones(ms[-1].nel, np.uint8) if l < nu else ms[-1].flag_elements(ex4_flag)
        ms.append(ms[-1].refine_flagged(flags))
    return ms


cases = [("uniform Q2 8^3 x2 (coarse n=4913: 154 pivot blocks of the dense inverse)", lambda: PoissonMG(ctx, 8, 8, 8, 2).init()),
         ("uniform Q2 2^3 x3", lambda: PoissonMG(ctx, 2, 2, 2, 3).init()),
         ("uniform Q2 4^3 x3 (coarse n=729)", lambda: PoissonMG(ctx, 4, 4, 4, 3).init()),
         ("uniform Q1 3x2x1 x3", lambda: PoissonMG(ctx, 3, 2, 1, 3, fe="linear").init()),
         ("AMR Q1 2^3, 1 uniform + 2 selective", lambda: PoissonMG(ctx, 2, 2, 2, 3, fe="linear", source_kind=3, params=(-2.0, 1.0),
                                                                  meshes=amr_meshes((2, 2, 2), 1, 2)).init()),
         ("AMR Q2 2^3, 1 uniform + 2 selective", lambda: PoissonMG(ctx, 2, 2, 2, 3, source_kind=3, params=(-2.0, 1.0),
                                                                  meshes=amr_meshes((2, 2, 2), 1, 2)).init())]
for name, make in cases:
    pb = make()
    ref = None
    for r in range(reps):
        pb.assemble()
        pb.prepare()
        its, rn = pb.mgsolve(outer="gmres", rtol=1e-13, maxit=60)
        x = pb.EPSC.to_numpy().copy()     # the solve result of this repetition (EPS accumulates)
        a0 = pb.A[0].values().copy()
        cur = (x, a0, its)
        if r == 0:
            # the first assembly runs the fused path, the later ones the two-pass path (the Galerkin product asked for element rows; option
            # assemble_fused = 1): same values to rounding, so repetition 0 is compared loosely and the bitwise reference is repetition 1
            first = cur
            continue
        if ref is None:
            ref = cur
            if not (its == first[2] and abs(x - first[0]).max() <= 1e-11 * max(abs(x).max(), 1e-300) and abs(a0 - first[1]).max() <= 1e-12 * abs(a0).max()):
                bad += 1
                print("MISMATCH (first vs second repetition, beyond rounding)", name, abs(x - first[0]).max(), abs(a0 - first[1]).max(), flush=True)
        elif not (np.array_equal(x, ref[0]) and np.array_equal(a0, ref[1]) and its == ref[2]):
            bad += 1
            print("MISMATCH", name, "rep", r, "its", its, ref[2], "max dx", abs(x - ref[0]).max(), "max dA0", abs(a0 - ref[1]).max(), flush=True)
    print("done", name, "its", ref[2], flush=True)
    pb.destroy()
print("mismatches:", bad)
sys.exit(1 if bad else 0)
