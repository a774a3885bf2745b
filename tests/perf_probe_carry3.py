"""Dev probe: walk order of the cluster kernel with and without its stores."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import femus_amd
from femus_amd.poisson import PoissonMG
ctx = femus_amd.Context(0)
for carry in (0, 101, 102, 103, 106, 6):
    ctx.set_option("assemble_carry", carry)
    pb = PoissonMG(ctx, 8, 8, 8, 4).init()
    row = {}
    for dbg, name in ((8, "cluster_kernel"), (8 | 4, "cluster_kernel_no_stores"), (8 | 64, "cluster_kernel_plain_stores")):
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
