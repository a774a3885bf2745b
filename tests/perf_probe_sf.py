"""Dev probe: fine-level assembly with the sum-factorised element kernel (assemble_sf = waves per workgroup) against the
matrix-core kernel (assemble_sf = 0), element kernel alone (asm_debug 8: row pass off) and whole assembly; prints the
largest difference of the assembled operators.   usage: perf_probe_sf.py [nw ...]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import femus_amd
from femus_amd.poisson import PoissonMG
ref = None
for nw in [int(v) for v in sys.argv[1:]] or [0, 10, 8]:
    ctx = femus_amd.Context(0)
    ctx.set_option("assemble_sf", nw)
    pb = PoissonMG(ctx, 8, 8, 8, 4).init()
    for _ in range(2): pb.assemble()
    ctx.timer_start()
    for _ in range(10): pb.assemble()
    full = ctx.timer_stop() / 10
    val = pb.A[-1].values()
    res = pb.RES.to_numpy() if hasattr(pb, "RES") else None
    if ref is None: ref = (val.copy(), None if res is None else res.copy())
    dv = abs(val - ref[0]).max() / abs(ref[0]).max()
    dr = 0.0 if res is None else abs(res - ref[1]).max() / max(abs(ref[1]).max(), 1e-300)
    ctx.set_option("asm_debug", 8)
    for _ in range(2): pb.assemble()
    ctx.timer_start()
    for _ in range(10): pb.assemble()
    el = ctx.timer_stop() / 10
    print("assemble_sf %2d: assembly %.3f ms, element kernel alone %.3f ms, max rel diff A %.2e res %.2e" % (nw, full, el, dv, dr), flush=True)
    pb.destroy(); ctx.close()
