"""applications/001_Poisson over the C-ABI: the JSON input of the application (comments, dotted paths,
defaults), its boundary-condition strings and its source string.  CPU part: loader + host logic; GPU part: the run against the
oracle's direct solution of the same discrete problem."""
import os

import numpy as np
import pytest
import scipy.sparse.linalg as spla

from femus_amd import app_poisson as app
from oracle import femus_oracle as fo

# same layout and keys as applications/001_Poisson/input/input.json (own values)
CONFIG = """
// Configuration options
{
    // mesh
    "multilevel_mesh" : { "first" : { "type" : { "box" : {
            "nx" : 4, "ny" : 4, "nz" : 0, "xa" : 0., "xb" : 1., "ya" : 0., "yb" : 1., "za" : 0., "zb" : 0.,
            "elem_type" : "Quad9" } } } },
    // solution
    "multilevel_solution" : { "multilevel_mesh" : { "first" : { "variable" : { "first" : {
              "name" : "T",
              "fe_order" : "second",
              "init_func" : "0.",
              "func_source": "10.*exp(-5.*x) - 4.*exp(-x)*y",
              "boundary_conditions" : [
                { "facename" : "left",   "bdc_type" : "dirichlet", "bdc_func" : "0.5+1./pi*atan(10.*(y-0.8))" },
                { "facename" : "top",    "bdc_type" : "dirichlet", "bdc_func" : "1." },
                { "facename" : "bottom", "bdc_type" : "neumann",   "bdc_func" : "0." },
                { "facename" : "right",  "bdc_type" : "neumann",   "bdc_func" : "0.2" }
              ] } } } } },
    // multilevel problem
    "multilevel_problem" : { "multilevel_mesh" : { "first" : { "system" : { "poisson" : { "linear_solver" : {
                "max_number_linear_iteration" : 8,
                "abs_conv_tol" : 1.e-10,
                "type" : { "multigrid" : { "nlevels" : 3, "npresmoothing" : 1, "npostmoothing" : 1, "mgtype" : "V_cycle",
                    "smoother" : { "type" : { "gmres" : { "ksp" : "gmres", "precond" : "ilu" } } } } } } } } } } }
}
"""


def test_config_loader_reads_commented_json_and_defaults():
    cfg = app.load_config(CONFIG)
    assert app.get(cfg, "multilevel_mesh.first.type.box.nx", 2) == 4
    assert app.get(cfg, app.PREFIX + "type.multigrid.nlevels", 1) == 3
    assert app.get(cfg, app.PREFIX + "type.multigrid.not_there", 7) == 7
    assert app.get(cfg, "multilevel_solution.multilevel_mesh.first.variable.first.func_source", "0.").startswith("10.*exp")


@pytest.mark.skipif(not os.path.isdir("/root/reference/applications/001_Poisson/input"), reason="reference tree not present")
def test_loader_accepts_every_shipped_input_file():
    d = "/root/reference/applications/001_Poisson/input"
    seen = 0
    for name in sorted(os.listdir(d)):
        if name.endswith(".json"):
            cfg = app.load_config(os.path.join(d, name))
            assert "multilevel_mesh" in cfg and app.get(cfg, app.PREFIX + "type.multigrid.nlevels", 0) >= 1
            seen += 1
    assert seen >= 10


@pytest.mark.gpu
def test_run_matches_oracle_direct_solution(ctx, tmp_path):
    p = app.Poisson001(ctx, CONFIG)
    out = p.run(log=None, output_dir=tmp_path)
    assert out["converged"] and len(out["history"]) <= 8
    # the two files the application writes at its end (main.cpp:259-270): VTK and GMV, named as the reference's writers name them
    assert [os.path.basename(f) for f in out["files"]] == ["sol.level3.0.biquadratic.vtu", "sol.level3.0.biquadratic.gmv"]
    from test_writers import read_gmv
    xyz, cells, kinds, part, var = read_gmv(out["files"][1])
    assert kinds == {"8quad"} and np.array_equal(var["Sol"], out["solution"][:xyz.shape[1]])
    # the same discrete problem with the oracle: F = (src phi - grad phi . grad T) w + Neumann term, Dirichlet rows penalised
    ms = fo.build_levels(4, 4, 0, 3)
    m = ms[-1]
    x, y = m.coords[:, 0], m.coords[:, 1]
    sol0 = np.zeros(m.nnode)
    fn = fo.face_nodes("quad")
    val = {}
    for iel in range(m.nel):                               # GenerateBdc order: elements, then faces
        for f in range(4):
            if m.face_flag[iel, f] == -5:                  # left
                for n in m.elem_dof[iel, fn[f]]:
                    val[n] = 0.5 + 1. / np.pi * np.arctan(10. * (y[n] - 0.8))
            elif m.face_flag[iel, f] == -4:                # top
                for n in m.elem_dof[iel, fn[f]]:
                    val[n] = 1.0
    bdc = np.array(sorted(val))
    sol0[bdc] = [val[n] for n in bdc]
    src = lambda xg: -(10. * np.exp(-5. * xg[..., 0]) - 4. * np.exp(-xg[..., 0]) * xg[..., 1])
    A, b = fo.assemble_poisson(m, "biquadratic", src, sol=sol0)
    b = b + fo.neumann_rhs(m, "biquadratic", {-3: 0.2})
    A = fo.zero_rows(A, bdc, 1.0)
    b[bdc] = 0.0
    ref = sol0 + spla.spsolve(A.tocsc(), b)
    assert np.array_equal(out["coords"], m.coords)
    assert abs(out["solution"] - ref).max() < 1e-9
    p.destroy()


FILE_CONFIG = """
{
    "multilevel_mesh" : { "first" : { "type" : { "filename" : "cube.neu" } } },
    "multilevel_solution" : { "multilevel_mesh" : { "first" : { "variable" : { "first" : {
              "name" : "T", "fe_order" : "second", "init_func" : "0.", "func_source": "1.+x*y",
              "boundary_conditions" : [ { "facename" : "top", "bdc_type" : "dirichlet" } ] } } } } },
    "multilevel_problem" : { "multilevel_mesh" : { "first" : { "system" : { "poisson" : { "linear_solver" : {
                "max_number_linear_iteration" : 10, "abs_conv_tol" : 1.e-10,
                "type" : { "multigrid" : { "nlevels" : 2, "npresmoothing" : 1, "npostmoothing" : 1, "mgtype" : "V_cycle" } } } } } } } }
}
"""


@pytest.mark.gpu
def test_mesh_file_input_matches_oracle(ctx, tmp_path):
    """the 3-D inputs of the application: mesh from a Gambit file, SetBoundaryCondition of main.cpp:26-36 (Dirichlet 0, flux 0.2
    through face name 3)"""
    from gambit_writer import write_neu
    mo = fo.coarse_box_mesh(2, 2, 2)
    write_neu(tmp_path / "cube.neu", "hex", mo.elem_dof, mo.coords, mo.face_flag)
    p = app.Poisson001(ctx, FILE_CONFIG, base_dir=str(tmp_path))
    out = p.run()
    assert out["converged"]
    m = fo.refine(mo)
    fn = fo.face_nodes("hex")
    bdc = set()
    for f in range(6):
        els = np.where((m.face_flag[:, f] < -1) & (m.face_flag[:, f] != -4))[0]
        bdc.update(m.elem_dof[els][:, fn[f]].ravel().tolist())
    bdc = np.array(sorted(bdc))
    src = lambda xg: -(1. + xg[..., 0] * xg[..., 1])
    A, b = fo.assemble_poisson(m, "biquadratic", src)
    b = b + fo.neumann_rhs(m, "biquadratic", {-4: 0.2})
    A = fo.zero_rows(A, bdc, 1.0)
    b[bdc] = 0.0
    ref = spla.spsolve(A.tocsc(), b)
    assert np.allclose(out["coords"], m.coords, atol=1e-10)
    assert abs(out["solution"] - ref).max() < 1e-9
    p.destroy()


@pytest.mark.gpu
def test_run_with_a_parsed_neumann_function(ctx, tmp_path):
    """a Neumann face whose bdc_func is a function of the position: evaluated at every face Gauss point as the callback does
    (`(*bdcfunc)(&xyzt[0])`, main.cpp:521-537), not once per face"""
    cfg = CONFIG.replace('"bdc_func" : "0.2" }', '"bdc_func" : "0.2 + 0.5*y*y - 0.1*x" }').replace('"bdc_func" : "0." }', '"bdc_func" : "sin(3.*x)" }')
    assert cfg != CONFIG
    p = app.Poisson001(ctx, cfg)
    out = p.run(log=None, output_dir=tmp_path)
    assert out["converged"]
    m = fo.build_levels(4, 4, 0, 3)[-1]
    x, y = m.coords[:, 0], m.coords[:, 1]
    sol0 = np.zeros(m.nnode)
    fn = fo.face_nodes("quad")
    val = {}
    for iel in range(m.nel):
        for f in range(4):
            if m.face_flag[iel, f] == -5:
                for n in m.elem_dof[iel, fn[f]]:
                    val[n] = 0.5 + 1. / np.pi * np.arctan(10. * (y[n] - 0.8))
            elif m.face_flag[iel, f] == -4:
                for n in m.elem_dof[iel, fn[f]]:
                    val[n] = 1.0
    bdc = np.array(sorted(val))
    sol0[bdc] = [val[n] for n in bdc]
    src = lambda xg: -(10. * np.exp(-5. * xg[..., 0]) - 4. * np.exp(-xg[..., 0]) * xg[..., 1])
    A, b = fo.assemble_poisson(m, "biquadratic", src, sol=sol0)
    b = b + fo.neumann_rhs(m, "biquadratic", {-3: lambda q: 0.2 + 0.5 * q[1] * q[1] - 0.1 * q[0], -2: lambda q: np.sin(3. * q[0])})
    A = fo.zero_rows(A, bdc, 1.0)
    b[bdc] = 0.0
    ref = sol0 + spla.spsolve(A.tocsc(), b)
    assert abs(out["solution"] - ref).max() < 1e-9
    # and it differs from the flux frozen at the origin (what a constant per face would give)
    b0 = fo.assemble_poisson(m, "biquadratic", src, sol=sol0)[1] + fo.neumann_rhs(m, "biquadratic", {-3: 0.2})
    b0[bdc] = 0.0
    assert abs(sol0 + spla.spsolve(A.tocsc(), b0) - ref).max() > 1e-3
    p.destroy()
