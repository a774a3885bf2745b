"""timing probe (not a test): one V(2,2) cycle of config C2 under context options given as name=value[,value...] arguments
usage: perf_probe_cycle_opts.py spmv_kernel=3,4 ..."""
import itertools
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import femus_amd
from femus_amd.poisson import PoissonMG

ctx = femus_amd.Context(0)
names, values = [], []
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    names.append(k)
    values.append([float(t) for t in v.split(",")])
pb = PoissonMG(ctx, 8, 8, 8, 4).init()
pb.assemble()
pb.prepare()
for combo in itertools.product(*values):
    for k, v in zip(names, combo):
        ctx.set_option(k, v)
    pb.prepare()
    for _ in range(3):
        pb.vcycle()
    ctx.sync()
    t = time.time()
    for _ in range(30):
        pb.vcycle()
    ctx.sync()
    print(dict(zip(names, combo)), "vcycle ms %.4f" % ((time.time() - t) / 30 * 1e3), flush=True)
