"""Dev probe: the fused cluster assembly of the bench problem (64^3) with carried rows (assemble_carry 0 / 3 / 6): time per assembly, the cluster kernel alone,
entries that go through the partial-row buffer; bitwise comparison of the three operators.
usage: perf_probe_carry.py [levels] [asm_debug bits to add, e.g. 64 = plain stores]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import femus_amd
from femus_amd.poisson import PoissonMG

levels = int(sys.argv[1]) if len(sys.argv) > 1 else 4
extra = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ctx = femus_amd.Context(0)
out = {}
ref = None
for carry in ([int(c) for c in os.environ["CARRY_LIST"].split(",")] if "CARRY_LIST" in os.environ else (0, 3, 6, -1)):
    ctx.set_option("assemble_carry", carry)
    pb = PoissonMG(ctx, 8, 8, 8, levels).init()
    fi = pb.asm[-1].fused_info()
    pb.assemble()
    v = pb.A[-1].values().copy()
    r = pb.RES.to_numpy().copy()
    if ref is None:
        ref = (v, r)
    same = bool(np.array_equal(v, ref[0]) and np.array_equal(r, ref[1]))
    row = {"info": fi, "bitwise_equal_to_carry_0": same}
    for dbg, name in ((0, "assembly_ms"), (8, "cluster_kernel_ms"), (2, "element_phase_ms"), (64, "assembly_plain_stores_ms")):
        ctx.set_option("asm_debug", dbg | extra)
        for _ in range(5): pb.assemble()
        ts = []
        for _ in range(5):
            ctx.timer_start()
            for _ in range(10): pb.assemble()
            ts.append(ctx.timer_stop() / 10)
        row[name] = float(np.median(ts))
    ctx.set_option("asm_debug", 0)
    out["carry_%d" % carry] = row
    print(carry, json.dumps(row), flush=True)
    pb.destroy()
print(json.dumps(out))
