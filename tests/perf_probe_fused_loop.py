"""Dev probe for the PMC passes: N fused fine-level assemblies of the bench problem.  usage: perf_probe_fused_loop.py [asm_debug] [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import femus_amd
from femus_amd.poisson import PoissonMG
dbg = int(sys.argv[1]) if len(sys.argv) > 1 else 0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ctx = femus_amd.Context(0)
ctx.set_option("assemble_fused", int(os.environ.get("FEMUS_FUSED", "1")))
pb = PoissonMG(ctx, 8, 8, 8, 4).init()
ctx.set_option("asm_debug", dbg)
for _ in range(reps): pb.assemble()
ctx.timer_start()
for _ in range(reps): pb.assemble()
print("asm_debug %d: %.3f ms per assembly" % (dbg, ctx.timer_stop() / reps))
