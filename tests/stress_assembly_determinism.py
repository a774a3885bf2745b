"""Dev stress: repeated fine-level assemblies must be bit-identical (races in the element kernel's LDS / store hand-offs would show)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import femus_amd
from femus_amd import capi
ctx = femus_amd.Context(0)
bad = 0
for args, reps in (((4, 4, 4), 30), ((3, 2, 1), 100), ((16, 16, 16), 20)):
    m = capi.Mesh.box(*args)
    for _ in range(1): m = m.refine()
    ed, xy, _ = m.arrays()
    rng = np.random.default_rng(3)
    xy = xy + rng.uniform(-0.005, 0.005, xy.shape)
    n = m.nnode
    rp, col = capi.pattern_from_elements(ed, n)
    A = ctx.matrix_csr(n, n, rp, col)
    asm = capi.Assembler(ctx, m, "biquadratic", A, elem_dof=ed, coords=xy)
    u = ctx.vector_from(rng.uniform(-1, 1, n))
    res = ctx.vector(n)
    ref = None
    for kind, params in ((0, (1.5,)), (1, (2.0, 1.3))):
        ref = None
        for r in range(reps):
            asm.assemble(A, res, u, kind, params)
            v, f = A.values(), res.to_numpy()
            if ref is None: ref = (v.copy(), f.copy())
            elif not (np.array_equal(v, ref[0]) and np.array_equal(f, ref[1])):
                bad += 1
                print("MISMATCH", args, kind, r, abs(v - ref[0]).max(), abs(f - ref[1]).max(), flush=True)
    print("done", args, "elements", ed.shape[0], flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
