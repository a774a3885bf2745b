"""timing probe (not a test): BASELINE config 5 -- 3-D Poisson Q2 with two adaptively refined levels on top of two uniform ones
(MGAMR ex4 flags on the unit cube), one GPU; prints one JSON line"""
import os
import json
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import femus_amd
from femus_amd import capi
from femus_amd.poisson import PoissonMG


def flag(x, level):
    if level == 1:
        return x[0] > 0.5
    return x[0] > 0.5 and x[1] > 0.25


def main(mode, n0=None, quiet=False):
    n0 = n0 or (int(sys.argv[1]) if len(sys.argv) > 1 else 8)
    ctx = femus_amd.Context(0)
    t0 = time.time()
    ms = [capi.Mesh.box(n0, n0, n0).set_amr_mode(mode)]
    ms.append(ms[-1].refine(ctx))                                                   # all levels refined on the device (round 5)
    for _ in range(2):
        ms.append(ms[-1].refine_device(ctx, ms[-1].flag_elements(flag)))
    mesh_s = time.time() - t0
    t0 = time.time()
    pb = PoissonMG(ctx, n0, n0, n0, 4, source_kind=3, params=(-2.0, 1.0), meshes=ms).init()
    init_s = time.time() - t0
    ctx.sync(); t = time.time()
    pb.assemble(); ctx.sync()
    first_asm_ms = (time.time() - t) * 1e3
    t = time.time()
    for _ in range(5):
        pb.assemble()
    ctx.sync(); asm_ms = (time.time() - t) / 5 * 1e3
    t = time.time(); pb.prepare(); ctx.sync(); prep_first_ms = (time.time() - t) * 1e3
    pb.assemble()
    t = time.time(); pb.prepare(); ctx.sync(); prep_ms = (time.time() - t) * 1e3
    pb.zero_boundary_residuals()
    t = time.time()
    for _ in range(20):
        pb.vcycle()
    ctx.sync(); cyc_ms = (time.time() - t) / 20 * 1e3
    t = time.time(); its, rn = pb.mgsolve(outer="gmres", rtol=1e-12, maxit=60); ctx.sync(); solve_ms = (time.time() - t) * 1e3
    pb.update_sol()
    _, xy, _ = ms[-1].arrays()
    err = abs(pb.SOL.to_numpy() - np.prod(xy * (1 - xy), axis=1)).max()
    # "reference" = the restriction map exactly as Mesh::GetAMRRestrictionAndAMRSolidMark builds it: rows at nodes on two interfaces do not
    # sum to one, so a Q2 polynomial is NOT reproduced there (a property of the reference's map); "coarsest" = the consistent variant
    if quiet:
        pb.destroy()
        return
    print(json.dumps({"warm_up": "a 2^3 problem of the same shape ran first in this process (code objects loaded, allocator pools grown): the times are the set-up's own",
                      "config": "3-D Poisson Q2, %d^3 coarse, 2 uniform + 2 adaptive levels" % n0, "amr_mode": mode, "elements": [m.nel for m in ms],
                      "dofs": pb.ndof, "hanging": [int(h.size) for h in pb.hanging], "mesh_s": mesh_s, "init_s": init_s,
                      "assembly_with_projection_ms": asm_ms, "assembly_first_ms": first_asm_ms, "prepare_first_ms": prep_first_ms,
                      "prepare_ms": prep_ms, "vcycle_ms": cyc_ms, "gmres_its": its, "solve_ms": solve_ms, "q2_polynomial_error": err}))
    pb.destroy()


main("reference", n0=2, quiet=True)
for mode in (sys.argv[2:] or ("reference", "coarsest")):
    main(mode)
