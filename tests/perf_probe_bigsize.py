"""Probe (not a pytest test): the single-GPU path one level beyond the bench size (8^3 -> 128^3 elements, 16 974 593 dofs, 1.09e9 non-zeros)
stage by stage with a synchronisation after each -- the sizes where 32-bit offsets run out.  usage: python tests/perf_probe_bigsize.py [levels=5]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import femus_amd as fa
from femus_amd.poisson import PoissonMG

levels = int(sys.argv[1]) if len(sys.argv) > 1 else 5
ctx = fa.Context(0)
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    ctx.set_option(k, float(v))
out = {}


def stage(name, fn):
    t0 = time.time()
    r = fn()
    ctx.sync()
    out[name + "_s"] = round(time.time() - t0, 3)
    print(name, out[name + "_s"], flush=True)
    return r


pb = stage("init", lambda: PoissonMG(ctx, 8, 8, 8, levels, fe="biquadratic", order="seventh", omega=2. / 3., npre=2, npost=2, coarse="galerkin",
                                     source_kind=0, params=(1.0,)).init())
print("dofs", pb.ndof[-1], "nnz", pb.A[-1].nnz, flush=True)
stage("assemble", pb.assemble)
A = pb.A[-1]
# row sums of the assembled Laplacian vanish on interior rows: y = A * 1
one = ctx.vector(pb.ndof[-1])
y = ctx.vector(pb.ndof[-1])
one.upload(np.ones(pb.ndof[-1]))
stage("spmv", lambda: y.matrix_mult(one, A))
yy = y.to_numpy()
out["max_abs_rowsum"] = float(np.abs(yy).max())
print("max |A 1| =", out["max_abs_rowsum"], flush=True)
stage("prepare", pb.prepare)
stage("vcycle", pb.vcycle)
pb.zero_boundary_residuals()
nb = pb.RES.l2_norm()
its, rn = stage("gmres", lambda: pb.mgsolve(outer="gmres", rtol=1e-10, maxit=30))
out["gmres_its"], out["true_relres"] = int(its), float(pb.RES.l2_norm() / nb)      # mgsolve leaves RES = b - A x
# the solution of -lap u = 1, u = 0 on the boundary of the unit cube, peaks at the centre: 0.056212
x = pb.EPS.to_numpy()
out["max_u"] = float(np.abs(x).max())      # the reference's residual convention makes the correction -u
out["dofs"], out["nnz"] = int(pb.ndof[-1]), int(pb.A[-1].nnz)
print(json.dumps(out))
assert out["true_relres"] < 1e-6 and abs(out["max_u"] - 0.056212) < 2e-5 and out["max_abs_rowsum"] < 1e-14
