"""Dev probe: assembly with the element rows stored non-temporally (asm_debug bit 6)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import femus_amd
from femus_amd.poisson import PoissonMG
ctx = femus_amd.Context(0)
pb = PoissonMG(ctx, 8, 8, 8, 4).init()
for rep in range(3):
    for dbg in (0, 64, 8, 72):
        ctx.set_option("asm_debug", dbg)
        for _ in range(3): pb.assemble()
        ctx.timer_start()
        for _ in range(20): pb.assemble()
        print("asm_debug %d %.4f ms" % (dbg, ctx.timer_stop() / 20), flush=True)
ctx.set_option("asm_debug", 0)
