"""Dev probe: where does the fine-level assembly time go?  (asm_debug bit 0: no quadrature, bit 1: no scatter)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import femus_amd
from femus_amd.poisson import PoissonMG
ctx = femus_amd.Context(0)
pb = PoissonMG(ctx, 8, 8, 8, 4).init()
for dbg, name in ((0, "full"), (1, "gather+scatter only"), (2, "gather+quadrature only"), (6, "gather+phase1 only"), (3, "gather only")):
    ctx.set_option("asm_debug", dbg)
    for _ in range(2): pb.assemble()
    ctx.timer_start()
    for _ in range(5): pb.assemble()
    print("%-24s %.3f ms" % (name, ctx.timer_stop() / 5), flush=True)
ctx.set_option("asm_debug", 0)
