"""timing probe (not a test): hierarchy preparation (Galerkin chain, penalty rows, smoother / coarse setup) at config C2"""
import sys
import time

sys.path.insert(0, ".")
import femus_amd
from femus_amd.poisson import PoissonMG

ctx = femus_amd.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
pb = PoissonMG(ctx, n, n, n, 4).init()
pb.assemble()
pb.prepare()
ctx.sync()
for _ in range(3):
    pb.assemble()
    ctx.sync()
    t = time.time()
    pb.prepare()
    ctx.sync()
    print("prepare ms", (time.time() - t) * 1e3)
