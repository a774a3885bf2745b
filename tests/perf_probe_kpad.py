"""Dev probe: fine-level assembly for (assemble_mfma waves per workgroup, padded element rows) pairs; run under
rocprofv3 --kernel-trace for the per-kernel split.   usage: perf_probe_kpad.py "nw,kpad" ..."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import femus_amd
from femus_amd.poisson import PoissonMG
for spec in sys.argv[1:] or ["12,1", "8,1"]:
    nw, kpad = [int(v) for v in spec.split(",")]
    ctx = femus_amd.Context(0)
    ctx.set_option("assemble_kpad", kpad)
    ctx.set_option("assemble_mfma", abs(nw))
    ctx.set_option("assemble_sumfac", 0 if nw < 0 else 1)      # negative: direct 27-node Jacobian loop
    pb = PoissonMG(ctx, 8, 8, 8, 4).init()
    for _ in range(2): pb.assemble()
    ctx.timer_start()
    for _ in range(10): pb.assemble()
    print("nw %d kpad %d full %.3f ms" % (nw, kpad, ctx.timer_stop() / 10), flush=True)
