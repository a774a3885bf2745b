"""dev: PtAP of the Navier-Stokes level operators with and without the slot maps: same values up to rounding?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import femus_amd
from femus_amd import capi
from femus_amd.navier_stokes import NavierStokesMG
ctx = femus_amd.Context(0)
nl = 3
pb = NavierStokesMG(ctx, 10, 10, 0, nl, 0.01).init()
assert pb.newton(0, tol=1e-10, max_newton=25)
for ig in range(1, nl):
    pb.prolongator_sol(ig)
    assert pb.newton(ig, tol=1e-10, max_newton=25, lin_rtol=1e-10, lin_maxit=200)
top = nl - 1
pb.asm[top].assemble(pb.KK[top], pb.RES[top], pb.SOL[top], pb.nu)
A, P = pb.KK[top], pb.P[top]
out = []
for opt in (0, 1):
    ctx.set_option("spgemm_slot_map", opt)
    C = capi.Mat.ptap(P, A)
    C.ptap_numeric(P, A)
    S = C.to_scipy()
    out.append(S.copy())
    C.destroy()
import scipy.sparse as sp
Ps, As = P.to_scipy(), A.to_scipy()
ref = (Ps.T @ As @ Ps).tocsr(); ref.sort_indices()
for k, S in enumerate(out):
    d = abs(S - ref)
    print("slot_map", k, "nnz", S.nnz, "max |C - P^T A P| / max|C|", d.max() / abs(ref).max())
d = abs(out[0] - out[1])
print("between the two:", d.max() / abs(ref).max())
