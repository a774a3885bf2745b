"""Dev probe (not a pytest test): fine-level 64^3 Q2 pattern in FEMuS numbering -> SpMV timing / GB/s.
usage: python tests/perf_probe_spmv.py [nlevels=4]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import femus_amd
from oracle import femus_oracle as fo

nlev = int(sys.argv[1]) if len(sys.argv) > 1 else 4
t = time.time()
ms = fo.build_levels(8, 8, 8, nlev)
rp, col = fo.csr_pattern(ms[-1], "biquadratic")
n = ms[-1].nnode
print("mesh+pattern %.1fs n=%d nnz=%d" % (time.time() - t, n, rp[-1]), flush=True)
rng = np.random.default_rng(0)
val = rng.uniform(-1, 1, rp[-1])
xs = rng.uniform(-1, 1, n)
ctx = femus_amd.Context(0)
print(ctx.device_name())
A = ctx.matrix_csr(n, n, rp, col, val)
x, y, b, dinv = ctx.vector_from(xs), ctx.vector(n), ctx.vector_from(xs), ctx.vector_from(np.abs(xs) + 1)
import scipy.sparse as sp
ref = sp.csr_matrix((val, col, rp), shape=(n, n)) @ xs
by = A.spmv_algorithmic_bytes()
configs = [(0, 2048, 0, 0), (0, 4096, 1, 0), (1, 2048, 0, 0)]
for tile in (256, 512, 1024):
    for remap in (0, 1):
        for nt in (0, 1):
            configs.append((2, tile, remap, nt))
if len(sys.argv) > 2:      # e.g. "0,2048,1,0;2,512,0,0"
    configs = [tuple(int(v) for v in c.split(",")) for c in sys.argv[2].split(";")]
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 20
if len(sys.argv) > 4:
    ctx.set_option("spmv_threads", int(sys.argv[4]))
for kernel, tile, remap, nt in configs:
    ctx.set_option("spmv_kernel", kernel); ctx.set_option("spmv_tile", tile); ctx.set_option("spmv_xcd_remap", remap); ctx.set_option("spmv_nt", nt)
    y.matrix_mult(x, A); ctx.sync()
    err = np.linalg.norm(y.to_numpy() - ref) / np.linalg.norm(ref)
    for mode, name in ((0, "y=Ax"), (3, "jacobi")):
        for _ in range(3):
            if mode == 0: y.matrix_mult(x, A)
            else: y.jacobi_sweep(b, x, A, dinv, 0.6)
        ctx.timer_start()
        reps = REPS
        for _ in range(reps):
            if mode == 0: y.matrix_mult(x, A)
            else: y.jacobi_sweep(b, x, A, dinv, 0.6)
        ms_ = ctx.timer_stop() / reps
        print("kernel=%d tile=%d remap=%d nt=%d %-7s %.3f ms  %.1f GB/s algorithmic (%.1f%% of 8 TB/s) err=%.1e"
              % (kernel, tile, remap, nt, name, ms_, by / ms_ / 1e6, by / ms_ / 1e6 / 80.0, err), flush=True)
