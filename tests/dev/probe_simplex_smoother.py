"""Dev: the smoother of the simplex paths -- natural-order SOR (the reference's PCSOR) against the coloured Gauss-Seidel the hexahedral path uses by default"""
import os, shutil, sys, tempfile, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import femus_amd
from femus_amd import app_poisson as app, capi
import perf_probe_shipped_inputs as sp
ctx = femus_amd.Context(0)
base = tempfile.mkdtemp()
os.makedirs(os.path.join(base, "input"))
for f in ("cube_Tet.neu", "cube_Wedge.neu", "cube_all_shapes_Six_boundary_groups.neu"):
    shutil.copy(os.path.join(os.path.dirname(HERE), "golden", f), os.path.join(base, "input", f))
for name in ("input3D_Tet_second.json", "input3D_Tet_first.json", "input3D.json", "input3D_Wedge_second.json"):
    p = app.Poisson001(ctx, dict(sp.INPUTS)[name], base_dir=base)
    fn = p.run_tet if p.tet else p.run_wedge if p.wedge else p.run_mixed
    for label, sm, om in (("sor", capi.SMOOTH_SOR, 1.0), ("gs_color 0.5", capi.SMOOTH_GS_COLOR, 0.5), ("gs_color 1.0", capi.SMOOTH_GS_COLOR, 1.0)):
        fn(None, sm, om)
        ctx.sync(); t0 = time.perf_counter()
        out = fn(None, sm, om)
        ctx.sync()
        print(name, label, "its", len(out["history"]) - 1, "res", out["history"][-1][1], "converged", out["converged"], "wall %.3f" % (time.perf_counter() - t0), flush=True)
    p.destroy()
