"""dev stress (not a test): random adaptive hierarchies, device refinement against the host loops, every array bit for bit"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import femus_amd
from femus_amd import capi

ctx = femus_amd.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for case in range(ncase):
    dim = int(rng.integers(2, 4))
    box = tuple(int(v) for v in rng.integers(1, 5, 3))
    if dim == 2:
        box = (box[0], box[1], 0)
    nlev = int(rng.integers(3, 6 if dim == 3 else 7))
    h, d = capi.Mesh.box(*box), capi.Mesh.box(*box)
    _, xy, _ = h.arrays()
    xy = xy + 0.05 * rng.uniform(-1, 1, xy.shape) / max(box)
    h.set_coords(xy); d.set_coords(xy)
    ok = True
    for l in range(1, nlev):
        p = rng.uniform(0.15, 0.9)
        lev, _ = h.elem_levels()
        fl = ((rng.uniform(0, 1, h.nel) < p) | (rng.uniform() < 0.2)).astype(np.uint8)
        if h.nel * 8 > 400000:
            break
        h2, d2 = h.refine_flagged(fl), d.refine_device(ctx, fl)
        for a, b in zip(h2.arrays(), d2.arrays()):
            ok &= np.array_equal(a.view(np.int64) if a.dtype == np.float64 else a, b.view(np.int64) if b.dtype == np.float64 else b)
        ok &= np.array_equal(h.child_elems(), d.child_elems()) and h2.own_size == d2.own_size
        ok &= np.array_equal(h2.elem_levels()[0], d2.elem_levels()[0]) and h2.elem_levels()[1] == d2.elem_levels()[1]
        h, d = h2, d2
    print("case %d dim %d box %s levels %d: nel %d nnode %d %s" % (case, dim, box, nlev, h.nel, h.nnode, "ok" if ok else "MISMATCH"))
    bad += not ok
print("mismatches:", bad)
sys.exit(1 if bad else 0)
