"""Dev: the tetrahedral and mixed-shape inputs of 001_Poisson one level beyond what they ship with (five levels): unknowns, iterations, residual, wall time by stage"""
import os, shutil, sys, tempfile, time, cProfile, pstats
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import femus_amd
from femus_amd import app_poisson as app
import perf_probe_shipped_inputs as sp
ctx = femus_amd.Context(0)
base = tempfile.mkdtemp()
os.makedirs(os.path.join(base, "input"))
for f in ("cube_Tet.neu", "cube_Wedge.neu", "cube_all_shapes_Six_boundary_groups.neu"):
    shutil.copy(os.path.join(os.path.dirname(HERE), "golden", f), os.path.join(base, "input", f))
for name in ("input3D_Tet_second.json", "input3D.json"):
    text = dict(sp.INPUTS)[name].replace('"nlevels" : 4', '"nlevels" : 5')
    p = app.Poisson001(ctx, text, base_dir=base)
    assert p.nlevels == 5
    ctx.sync(); t0 = time.perf_counter()
    pr = cProfile.Profile(); pr.enable()
    out = p.run()
    ctx.sync(); pr.disable()
    print(name, "levels 5 unknowns", out["dofs"], "its", len(out["history"]) - 1, "res", out["history"][-1][1], "converged", out["converged"], "wall %.2f" % (time.perf_counter() - t0), flush=True)
    pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
    p.destroy()
