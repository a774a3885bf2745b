"""Dev probe: how much of the fine-level SpMV time is the x gather?  Same row structure, columns replaced by
0..len-1 (perfectly local gathers) vs the real FEMuS columns."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import femus_amd
from femus_amd import capi
m = capi.Mesh.box(8, 8, 8)
for _ in range(3):
    m = m.refine()
ed, xy, _ = m.arrays()
n = m.nnode
rp, col = capi.pattern_from_elements(ed, n)
rng = np.random.default_rng(0)
val = rng.uniform(-1, 1, rp[-1]); xs = rng.uniform(-1, 1, n)
ctx = femus_amd.Context(0)
lens = np.diff(rp)
col_local = (np.arange(rp[-1]) - np.repeat(rp[:-1], lens)).astype(np.int32)
col_near = (col_local + np.repeat(np.minimum(np.arange(n), n - lens), lens)).astype(np.int32)   # banded: row + k
for name, c in (("femus", col), ("local0..len", col_local), ("banded row+k", col_near)):
    A = ctx.matrix_csr(n, n, rp, c, val)
    x, y = ctx.vector_from(xs), ctx.vector(n)
    by = A.spmv_algorithmic_bytes()
    for kernel, tile, share, nt in ((3, 2048, 1, 256), (3, 1024, 1, 128), (3, 2048, 1, 128), (3, 1024, 1, 256)):
        ctx.set_option("spmv_kernel", kernel); ctx.set_option("spmv_tile", tile); ctx.set_option("spmv_share", share); ctx.set_option("spmv_threads", nt)
        for _ in range(3): y.matrix_mult(x, A)
        ctx.timer_start()
        for _ in range(20): y.matrix_mult(x, A)
        ms = ctx.timer_stop() / 20
        print("%-14s kernel=%d tile=%d share=%d nt=%d  %.3f ms  %.0f GB/s" % (name, kernel, tile, share, nt, ms, by / ms / 1e6), flush=True)
    A.destroy()
