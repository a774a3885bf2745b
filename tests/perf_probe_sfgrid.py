"""Dev probe: element kernel alone (asm_debug 8) and the whole assembly for several grid multipliers of the sum-factorised kernel.
usage: perf_probe_sfgrid.py [mult ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import femus_amd
from femus_amd.poisson import PoissonMG
ctx = femus_amd.Context(0)
pb = PoissonMG(ctx, 8, 8, 8, 4).init()
for rep in range(2):
    for mult in [int(v) for v in sys.argv[1:]] or [1, 2, 4, 8, 16]:
        ctx.set_option("assemble_sf_grid", mult)
        out = []
        for dbg in (8, 0):
            ctx.set_option("asm_debug", dbg)
            for _ in range(3): pb.assemble()
            ctx.timer_start()
            for _ in range(20): pb.assemble()
            out.append("%s %.4f" % ("elements" if dbg else "assembly", ctx.timer_stop() / 20))
        print("grid x%d: " % mult + " | ".join(out) + " ms", flush=True)
