"""Dev probe: where does the fine-level assembly time go?  (asm_debug bit 0: no quadrature / MFMA phase, bit 1: no output)
usage: perf_probe_assembly.py [assemble_mfma values ...]   (0 = vector kernel)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import femus_amd
from femus_amd.poisson import PoissonMG
ctx = femus_amd.Context(0)
pb = PoissonMG(ctx, 8, 8, 8, 4).init()
variants = [int(v) for v in sys.argv[1:]] or [12, 0]
for nw in variants:
    ctx.set_option("assemble_mfma", abs(nw))
    ctx.set_option("assemble_sumfac", 0 if nw < 0 else 1)       # negative: direct 27-node Jacobian loop
    for dbg, name in ((0, "full"), (1, "no quadrature / MFMA phase"), (2, "no output"), (3, "neither")):
        ctx.set_option("asm_debug", dbg)
        for _ in range(2): pb.assemble()
        ctx.timer_start()
        for _ in range(5): pb.assemble()
        print("assemble_mfma %2d  %-28s %.3f ms" % (nw, name, ctx.timer_stop() / 5), flush=True)
ctx.set_option("asm_debug", 0)
