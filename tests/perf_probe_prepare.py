"""timing probe (not a test): hierarchy preparation (Galerkin chain, penalty rows, smoother / coarse setup) at config C2
usage: perf_probe_prepare.py [coarse n] [gj_mfma values ...]"""
import sys
import time

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import femus_amd
from femus_amd.poisson import PoissonMG

ctx = femus_amd.Context(0)
if os.environ.get("FEMUS_ND"):
    ctx.set_option("coarse_nd", int(os.environ["FEMUS_ND"]))      # interior blocks of the dissected coarse solve (0: one dense inverse)
if os.environ.get("FEMUS_ND_STREAMS"):
    ctx.set_option("coarse_nd_streams", int(os.environ["FEMUS_ND_STREAMS"]))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
pb = PoissonMG(ctx, n, n, n, 4).init()
pb.assemble()
pb.prepare()
ctx.sync()
for gj in [int(v) for v in sys.argv[2:]] or [128]:
    ctx.set_option("gj_block", gj)
    for _ in range(3):
        pb.assemble()
        ctx.sync()
        t = time.time()
        pb.prepare()
        ctx.sync()
        print("gj_block %d prepare ms %.2f coarse (dense unknowns, blocks, separator, largest block) %s" % (gj, (time.time() - t) * 1e3, pb.mg.coarse_info()))
