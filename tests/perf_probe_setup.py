"""Where the set-up of the bench problem goes (what BENCH `setup_s` sums): mesh hierarchy, LinearImplicitSystem::init, first assembly, first
preparation -- wall clock per stage with a device synchronisation at each end.  FEMUS_HIP_TRACE=1 adds the stages inside the C calls."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import femus_amd as fh
from femus_amd import capi
from femus_amd.poisson import PoissonMG

def main():
    nb, nlev = 4, 5
    ctx = fh.Context(0)
    out = {}
    for rep in range(2):
        ctx.sync()
        t = [time.perf_counter()]
        meshes = [capi.Mesh.box(nb, nb, nb, (0., 0., 0.), (1., 1., 1.))]
        per = []
        for _ in range(1, nlev):
            t0 = time.perf_counter()
            meshes.append(meshes[-1].refine(ctx) if "--device" in sys.argv else meshes[-1].refine())
            per.append(time.perf_counter() - t0)
        t.append(time.perf_counter())
        pb = PoissonMG(ctx, nb, nb, nb, nlev, meshes=meshes, omega=2. / 3., npre=2, npost=2)
        if "--profile" in sys.argv and rep == 1:
            import cProfile, pstats
            pr = cProfile.Profile(); pr.enable(); pb.init(); ctx.sync(); pr.disable()
            pstats.Stats(pr, stream=sys.stderr).sort_stats("cumulative").print_stats(22)
        else:
            pb.init()
        ctx.sync(); t.append(time.perf_counter())
        pb.assemble(); ctx.sync(); t.append(time.perf_counter())
        pb.prepare(); ctx.sync(); t.append(time.perf_counter())
        pb.assemble(); pb.prepare(); ctx.sync(); t.append(time.perf_counter())
        out["rep%d" % rep] = {"mesh_s": t[1] - t[0], "refine_s_per_level": per, "init_s": t[2] - t[1], "assemble_first_s": t[3] - t[2],
                              "prepare_first_s": t[4] - t[3], "second_assemble_prepare_s": t[5] - t[4], "setup_s": t[4] - t[0]}
        pb.destroy()
    print(json.dumps(out, indent=1))

if __name__ == "__main__":
    main()
