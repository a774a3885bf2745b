"""Probe (not a pytest test): what the hipGraph replay of the V(2,2) cycle buys on the bench problem (64^3 Q2) -- GPU time by HIP events and
wall time of a cycle with use_graph = 1 and 0.  The un-captured time is what a distributed cycle (RCCL calls are not capturable on this
stack) pays per rank besides its exchanges."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import femus_amd as fa
from femus_amd.poisson import PoissonMG

out = {}
for graph in (1, 0):
    ctx = fa.Context(0)
    ctx.set_option("use_graph", graph)
    pb = PoissonMG(ctx, 8, 8, 8, 4, fe="biquadratic", order="seventh", omega=2. / 3., npre=2, npost=2, coarse="galerkin", source_kind=0,
                   params=(1.0,)).init()
    pb.assemble()
    pb.prepare()
    for _ in range(5):
        pb.vcycle()
    ctx.sync()
    reps = 50
    ctx.timer_start()
    t0 = time.perf_counter()
    for _ in range(reps):
        pb.vcycle()
    gpu_ms = ctx.timer_stop() / reps
    ctx.sync()
    wall_ms = (time.perf_counter() - t0) / reps * 1e3
    # launch-bound view: one cycle at a time, synchronised
    t0 = time.perf_counter()
    for _ in range(reps):
        pb.vcycle()
        ctx.sync()
    sync_ms = (time.perf_counter() - t0) / reps * 1e3
    out["graph" if graph else "no_graph"] = {"gpu_ms": gpu_ms, "wall_ms_back_to_back": wall_ms, "wall_ms_synchronised": sync_ms}
print(json.dumps(out))
