"""Dev: cProfile of input3D_Tet_second.json through app_poisson (where the host time of the simplex path goes)"""
import cProfile, os, pstats, shutil, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import femus_amd
from femus_amd import app_poisson as app
import perf_probe_shipped_inputs as sp
ctx = femus_amd.Context(0)
base = tempfile.mkdtemp()
os.makedirs(os.path.join(base, "input"))
for f in ("cube_Tet.neu", "cube_all_shapes_Six_boundary_groups.neu"):
    shutil.copy(os.path.join(os.path.dirname(HERE), "golden", f), os.path.join(base, "input", f))
name = sys.argv[1] if len(sys.argv) > 1 else "input3D_Tet_second.json"
text = dict(sp.INPUTS)[name]
p = app.Poisson001(ctx, text, base_dir=base)
p.run()
pr = cProfile.Profile(); pr.enable(); p.run(); ctx.sync(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
