"""Dev probe: whole assembly with the row pass's non-temporal options (assemble_rows_nt bit 0: loads of the element rows, bit 1: stores of the values)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import femus_amd
from femus_amd.poisson import PoissonMG
ctx = femus_amd.Context(0)
pb = PoissonMG(ctx, 8, 8, 8, 4).init()
for rep in range(3):
    for nt in (0, 1, 2, 3):
        ctx.set_option("assemble_rows_nt", nt)
        for _ in range(3): pb.assemble()
        ctx.timer_start()
        for _ in range(20): pb.assemble()
        print("rows_nt %d assembly %.4f ms" % (nt, ctx.timer_stop() / 20), flush=True)
