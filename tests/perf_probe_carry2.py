"""Dev probe: what the carried plan costs the cluster kernel -- walk order alone (assemble_carry 100 + k), carried stores without the loads (asm_debug bit 9)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import femus_amd
from femus_amd.poisson import PoissonMG
ctx = femus_amd.Context(0)
for carry in (0, 200, 203, 206, 6):
    ctx.set_option("assemble_carry", carry)
    pb = PoissonMG(ctx, 8, 8, 8, 4).init()
    row = {}
    for dbg, name in ((0, "assembly"), (8, "cluster_kernel"), (8 | 512, "cluster_kernel_no_carried_loads"), (2, "element_phase")):
        ctx.set_option("asm_debug", dbg)
        for _ in range(5): pb.assemble()
        ts = []
        for _ in range(5):
            ctx.timer_start()
            for _ in range(10): pb.assemble()
            ts.append(ctx.timer_stop() / 10)
        row[name] = round(float(np.median(ts)), 4)
    ctx.set_option("asm_debug", 0)
    print(carry, json.dumps(row), flush=True)
    pb.destroy()
