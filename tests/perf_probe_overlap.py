"""Dev probe (timing only): does the row pass run BESIDE the element kernel when it is issued on a second stream?
asm_debug bit 4 launches it there against the previous assembly's element rows.  usage: perf_probe_overlap.py [nw ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import femus_amd
from femus_amd.poisson import PoissonMG
for nw in [int(v) for v in sys.argv[1:]] or [8, 6, 4]:
    ctx = femus_amd.Context(0)
    ctx.set_option("assemble_sf", nw)
    pb = PoissonMG(ctx, 8, 8, 8, 4).init()
    out = []
    for dbg in (0, 8, 16, 48):
        ctx.set_option("asm_debug", dbg)
        for _ in range(3): pb.assemble()
        ctx.timer_start()
        for _ in range(20): pb.assemble()
        out.append("%s %.3f" % ({0: "serial", 8: "elements only", 16: "overlapped (elements first)", 48: "overlapped (rows first)"}[dbg], ctx.timer_stop() / 20))
    print("sf waves %d: " % nw + " | ".join(out) + " ms", flush=True)
    ctx.set_option("asm_debug", 0)
