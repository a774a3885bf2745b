"""The shipped inputs of applications/001_Poisson through app_poisson on the GPU, exactly as shipped (mesh file, family, levels, limits): unknowns, linear
iterations, last residual, wall time of Poisson001.run() -- thirteen of the application's fifteen inputs (the other two name input/cube_all_shapes.neu, a file the
reference tree does not hold).  The configurations are restated here as data (the reference tree does not travel to the GPU box); where it is present they are
compared with the shipped files key by key first.  The parity of every one of these runs with the oracle is asserted in tests/test_app_poisson.py, test_gambit.py,
test_line_1d.py, test_tet_3d.py, test_wedge_3d.py, test_mixed_3d.py -- this probe only adds the clock.
usage: python tests/perf_probe_shipped_inputs.py"""
import json, os, shutil, sys, tempfile, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np
import femus_amd
from femus_amd import app_poisson as app

SOLVER = """ "multilevel_problem" : { "multilevel_mesh" : { "first" : { "system" : { "poisson" : { "linear_solver" : {
                "max_number_linear_iteration" : 6, "abs_conv_tol" : 1.e-09,
                "type" : { "multigrid" : { "nlevels" : %d, "npresmoothing" : 1, "npostsmoothing" : 1, "mgtype" : "V_cycle",
                    "smoother" : { "type" : { "gmres" : { "ksp" : "gmres", "precond" : "ilu", "rtol" : 1.e-12, "atol" : 1.e-20, "divtol" : 1.e+50,
                                                          "max_its" : 4 } } } } } } } } } } } """


def mesh_file_input(mesh, fe_order):
    return """{ "multilevel_mesh" : { "first" : { "type" : { "filename" : "input/%s" } } },
    "multilevel_solution" : { "multilevel_mesh" : { "first" : { "variable" : { "first" : {
              "name" : "T", "fe_order" : "%s", "init_func" : "0.", "func_source": "0.",
              "boundary_conditions" : [ { "facename" : "top", "bdc_type" : "dirichlet" },
                                        { "facename" : "right", "bdc_type" : "neumann", "bdc_func" : "0.2" } ] } } } } }, %s }""" % (mesh, fe_order, SOLVER % 4)


INPUT_2D = """{ "multilevel_mesh" : { "first" : { "type" : { "box" : { "nx" : 20, "ny" : 20, "nz" : 0, "xa" : 0., "xb" : 1., "ya" : 0., "yb" : 1., "za" : 0., "zb" : 0.,
                                                                       "elem_type" : "Quad9" } } } },
    "multilevel_solution" : { "multilevel_mesh" : { "first" : { "variable" : { "first" : {
              "name" : "T", "fe_order" : "second", "init_func" : "0.",
              "boundary_conditions" : [ { "facename" : "left", "bdc_type" : "dirichlet", "bdc_func" : "0.5+1./pi*atan(1000.*(y-0.8))" },
                                        { "facename" : "top", "bdc_type" : "dirichlet", "bdc_func" : "1." },
                                        { "facename" : "bottom", "bdc_type" : "neumann", "bdc_func" : "0." },
                                        { "facename" : "right", "bdc_type" : "neumann", "bdc_func" : "0." } ] } } } } }, %s }""" % (SOLVER % 3)
INPUT_1D = """{ "multilevel_mesh" : { "first" : { "type" : { "box" : { "nx" : 10, "ny" : 0, "nz" : 0, "xa" : 0., "xb" : 1., "ya" : 0., "yb" : 0., "za" : 0., "zb" : 0.,
                                                                       "elem_type" : "Edge3" } } } },
    "multilevel_solution" : { "multilevel_mesh" : { "first" : { "variable" : { "first" : {
              "name" : "T", "fe_order" : "second", "init_func" : "0.", "func_source": "10.*exp(-5.*x) - 4.*exp(-x)",
              "boundary_conditions" : [ { "facename" : "left", "bdc_type" : "dirichlet" }, { "facename" : "right", "bdc_type" : "neumann" } ] } } } } }, %s }""" % (SOLVER % 1)

INPUTS = [("input.json", INPUT_2D), ("input1D.json", INPUT_1D)]
for shape, mesh in (("Hex", "cube_Hex.neu"), ("Tet", "cube_Tet.neu"), ("Wedge", "cube_Wedge.neu")):
    for order in ("first", "serendipity", "second"):
        INPUTS.append(("input3D_%s_%s.json" % (shape, order), mesh_file_input(mesh, order)))
INPUTS.append(("input3D_All_first.json", mesh_file_input("cube_all_shapes_Six_boundary_groups.neu", "first")))
INPUTS.append(("input3D.json", mesh_file_input("cube_all_shapes_Six_boundary_groups.neu", "second")))


def main():
    ref = "/root/reference/applications/001_Poisson/input"
    checked = 0
    if os.path.isdir(ref):
        for name, text in INPUTS:
            assert app.load_config(os.path.join(ref, name)) == app.load_config(text), name
            checked += 1
    ctx = femus_amd.Context(0)
    base = tempfile.mkdtemp()
    os.makedirs(os.path.join(base, "input"))
    for f in ("cube_Hex.neu", "cube_Tet.neu", "cube_Wedge.neu", "cube_all_shapes_Six_boundary_groups.neu"):
        shutil.copy(os.path.join(HERE, "golden", f), os.path.join(base, "input", f))
    out = {"configurations_equal_to_the_shipped_files": checked if checked else "reference tree not present on this box (compared where it is: tests/)", "runs": {}}
    for name, text in INPUTS:
        p = app.Poisson001(ctx, text, base_dir=base)
        ctx.sync()
        t0 = time.perf_counter()
        res = p.run()
        ctx.sync()
        wall = time.perf_counter() - t0
        hist = res["history"]
        out["runs"][name] = {"fe": p.fe, "levels": p.nlevels, "unknowns": int(res["dofs"] if "dofs" in res else np.asarray(res["solution"]).size),
                             "linear_iterations": len(hist) - 1, "last_residual": float(hist[-1][1]), "converged": bool(res["converged"]), "wall_s": round(wall, 3)}
        print(name, json.dumps(out["runs"][name]), flush=True)
        p.destroy()
    shutil.rmtree(base)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
