"""Dev probe: fused cluster assembly against the two-pass path -- values on a curved mesh, then timings on the bench problem (64^3).
usage: perf_probe_fused.py [levels]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import femus_amd
from femus_amd.poisson import PoissonMG

levels = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ctx = femus_amd.Context(0)

def run(nx, lv, fused, reps=0):
    ctx.set_option("assemble_fused", fused)
    pb = PoissonMG(ctx, nx, nx, nx, lv).init()
    pb.assemble()
    A = pb.A[-1].to_scipy()
    r = pb.RES.to_numpy() if hasattr(pb, "RES") else None
    t = None
    if reps:
        for _ in range(3): pb.assemble()
        ctx.timer_start()
        for _ in range(reps): pb.assemble()
        t = ctx.timer_stop() / reps
    pb.destroy()
    return A, r, t

for nx, lv in ((2, 2), (3, 3)):
    A0, r0, _ = run(nx, lv, 0)
    A1, r1, _ = run(nx, lv, 1)
    dA = abs(A0 - A1).max() / abs(A0).max()
    dr = abs(r0 - r1).max() / max(abs(r0).max(), 1e-300) if r0 is not None else -1
    print("nx %d levels %d: fused vs two-pass  dA %.2e  dres %.2e  (nnz %d)" % (nx, lv, dA, dr, A0.nnz), flush=True)
    assert dA < 1e-13 and dr < 1e-12

A0, r0, t0 = run(8, levels, 0, 10)
A1, r1, t1 = run(8, levels, 1, 10)
print("8^3 x %d levels: two-pass %.3f ms   fused %.3f ms   dA %.2e dres %.2e" % (levels, t0, t1, abs(A0 - A1).max() / abs(A0).max(), abs(r0 - r1).max() / abs(r0).max()), flush=True)
ctx.set_option("assemble_fused", 1)
pb = PoissonMG(ctx, 8, 8, 8, levels).init()
for dbg, name in ((0, "full"), (2, "cluster kernel without output / second pass"), (8, "cluster kernel alone"), (32, "full, plain loads in the second pass"),
                  (64, "full, plain stores in the cluster kernel"), (96, "full, both plain"), (0, "full")):
    ctx.set_option("asm_debug", dbg)
    for _ in range(3): pb.assemble()
    ctx.timer_start()
    for _ in range(10): pb.assemble()
    print("fused  %-45s %.3f ms" % (name, ctx.timer_stop() / 10), flush=True)
ctx.set_option("asm_debug", 0)
