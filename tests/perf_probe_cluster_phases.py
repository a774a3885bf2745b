"""Dev probe: where the waves of the fused cluster kernel spend their cycles (asm_debug bit 7 = the instrumented build of k_cluster_q2hex_sf: shader clock at
the phase boundaries, summed per wave in scalar registers; the table goes to stderr).  usage: perf_probe_cluster_phases.py [extra asm_debug bits] [stamps 0|1]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import femus_amd
from femus_amd.poisson import PoissonMG
extra = int(sys.argv[1]) if len(sys.argv) > 1 else 0
stamps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = femus_amd.Context(0)
ctx.set_option("assemble_fused", 2)
if os.environ.get("FEMUS_CARRY"): ctx.set_option("assemble_carry", int(os.environ["FEMUS_CARRY"]))
pb = PoissonMG(ctx, 8, 8, 8, 4).init()
for _ in range(3): pb.assemble()
for dbg, what in ((2, "element phase only"), (8, "cluster kernel alone"), (0, "cluster kernel + second pass")):
    ctx.set_option("asm_debug", dbg | extra)
    for _ in range(2): pb.assemble()
    ts = []
    for rep in range(7):
        ctx.timer_start()
        for _ in range(10): pb.assemble()
        ts.append(ctx.timer_stop() / 10)
    ts.sort()
    print("asm_debug %4d (%s): median %.3f  min %.3f  max %.3f ms per assembly" % (dbg | extra, what, ts[3], ts[0], ts[-1]), flush=True)
if stamps:
    ctx.set_option("asm_debug", 128 | extra)
    pb.assemble()
ctx.set_option("asm_debug", 0)
