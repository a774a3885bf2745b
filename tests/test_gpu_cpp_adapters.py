"""GPU: the C++ adapter classes (HipMatrix / HipVector / LinearEquationSolverHip behind the mirrored FEMuS interface),
driven by a small application written like applications/001_Poisson, must reproduce the oracle's solution."""
import os
import subprocess

import numpy as np
import pytest
import scipy.sparse.linalg as spla

from oracle import femus_oracle as fo

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ADAPTERS = os.path.join(ROOT, "femus_amd", "csrc", "adapters")
# the applications include the FEMuS headers by their plain names; here they resolve to the mirrored interface
INC = ["-I" + os.path.join(ADAPTERS, "mirror"), "-I" + ADAPTERS, "-I" + os.path.join(ROOT, "include")]


def build_app(tmp_path):
    lib = os.path.join(ROOT, "femus_amd", "lib")
    exe = str(tmp_path / "poisson_adapters")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "femus_amd", "csrc", "adapters")], stdout=subprocess.DEVNULL)
    subprocess.check_call(["g++", "-O1", "-std=c++17"] + INC + [os.path.join(ROOT, "tests", "cpp", "poisson_adapters.cpp"), "-o", exe,
                           "-L" + lib, "-lfemus_hip_adapters", "-lfemus_hip", "-Wl,-rpath," + lib])
    return exe


@pytest.mark.parametrize("args,nl,pc", [((2, 2, 2), 3, "jacobi"), ((4, 4, 0), 3, "jacobi"), ((4, 4, 0), 3, "sor"), ((2, 2, 2), 3, "ilu"), ((2, 2, 2), 3, "mlu"),
                                        ((4, 4, 0), 3, "mlu")])
def test_adapter_application_matches_oracle(tmp_path, args, nl, pc):
    exe = build_app(tmp_path)
    out = str(tmp_path / "sol.bin")
    log = subprocess.check_output([exe] + [str(a) for a in args] + [str(nl), out, pc], text=True)
    assert "Linear iteration" in log
    H = fo.build_poisson_hierarchy(*args, nl, "biquadratic", lambda xg: np.ones(xg.shape[:2]))
    xd = spla.spsolve(H.A[-1].tocsc(), H.b)
    sol = np.fromfile(out)
    assert sol.size == xd.size
    assert np.linalg.norm(sol - xd) <= 1e-10 * np.linalg.norm(xd)


def test_adapter_member_units(tmp_path):
    """every member of HipVector / HipMatrix the path uses, against hand-computed values"""
    lib = os.path.join(ROOT, "femus_amd", "lib")
    exe = str(tmp_path / "adapter_units")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "femus_amd", "csrc", "adapters")], stdout=subprocess.DEVNULL)
    subprocess.check_call(["g++", "-O1", "-std=c++17"] + INC + [os.path.join(ROOT, "tests", "cpp", "adapter_units.cpp"), "-o", exe,
                           "-L" + lib, "-lfemus_hip_adapters", "-lfemus_hip", "-Wl,-rpath," + lib])
    out = subprocess.run([exe], text=True, capture_output=True)
    assert "ADAPTER UNITS OK" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("nschur,nblock,level_solver,outer", [(0, 4, "richardson", "gmres"), (1, 4, "gmres", "gmres"), (1, 1, "gmres", "gmres"),
                                                             (1, 4, "richardson", "gmres"), (1, 4, "gmres", "fgmres"), (0, 4, "gmres", "fgmres")])
def test_navier_stokes_application_over_the_adapters(tmp_path, nschur, nblock, level_solver, outer):
    """003_NavierStokes-style driver in C++: F-cycle Newton through LinearEquationSolver::build(..., FEMuS_ASM) and the batched
    Taylor-Hood callback.  The smoother is set up by the reference's own calls (SteadyNavierStokesParallel/main.cpp:155-179):
    SetSolverFineGrids, SetPreconditionerFineGrids(ILU_PRECOND), SetNumberOfSchurVariables, SetElementBlockNumber -- the element blocks
    come from BuildASMIndex (no SetAsmBlocks).  (0, 4) are the application's block parameters; with the pressure as Schur variable a
    block is the pressure dofs of its elements + the velocities of the elements around them.  GMRES as level solver makes the cycle a
    non-stationary preconditioner of the (non-flexible) outer GMRES: the linear solves are a little less exact, the Newton history may be
    a few steps longer than the oracle's (exact linear solves), the discrete solution is the same; with the FLEXIBLE outer solver
    (SetOuterSolver(FGMRES), a case of the reference's switch) the linear solves are exact again, also with the application's own block
    parameters (0, 4) and GMRES level solvers.  This test runs the library's own block smoother (exact block inverses in colour order,
    SetAsmExactInColourOrder); PCASM as the reference configures it is the next test."""
    from oracle import femus_oracle_ns as ns
    lib = os.path.join(ROOT, "femus_amd", "lib")
    exe = str(tmp_path / "navier_stokes_adapters")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "femus_amd", "csrc", "adapters")], stdout=subprocess.DEVNULL)
    subprocess.check_call(["g++", "-O1", "-std=c++17"] + INC + [os.path.join(ROOT, "tests", "cpp", "navier_stokes_adapters.cpp"), "-o", exe,
                           "-L" + lib, "-lfemus_hip_adapters", "-lfemus_hip", "-Wl,-rpath," + lib])
    out = str(tmp_path / "ns.bin")
    log = subprocess.check_output([exe, "4", "3", "0.01", out, str(nschur), str(nblock), "0" if level_solver == "gmres" else "1",
                                   "2" if outer == "fgmres" else "0", "1"], text=True)       # last argument 1: the library's colour-ordered exact block smoother
    assert "Nonlinear iteration" in log
    _, lays, sols, hist = ns.solve_cavity(4, 4, 3, 0.01, (-0.5, -0.5, 0.0), (0.5, 0.5, 0.0), linear="direct")
    steps = int(log.split("newton steps = ")[1].split()[0])
    if (nschur, nblock, level_solver) == (0, 4, "gmres"):
        # the application's own block parameters with its GMRES level solver: a weak preconditioner in this restatement (the linear solves
        # end at the iteration limit), Newton still arrives at the same discrete solution -- in more steps
        assert len(hist) <= steps <= 60
    elif level_solver == "richardson" or outer == "fgmres":
        assert steps == len(hist)
    else:
        assert len(hist) <= steps <= len(hist) + 6
    sol = np.fromfile(out)
    assert sol.size == sols[-1].size
    assert np.linalg.norm(sol - sols[-1]) <= 1e-8 * np.linalg.norm(sols[-1])


@pytest.mark.parametrize("nschur,nblock,outer", [(1, 4, "gmres"), (0, 4, "fgmres")])
def test_navier_stokes_application_with_pcasm_as_the_reference_sets_it(tmp_path, nschur, nblock, outer):
    """the same driver with SetPreconditionerFineGrids(ILU_PRECOND) meaning what it means in the reference: PCASM basic / multiplicative over the
    blocks of BuildASMIndex in index order, one ILU(0) application (zero pivot 1e-16, MAT_SHIFT_NONZERO) per block (FH_SMOOTH_ASM), GMRES level
    solvers.  Newton reaches the oracle's discrete solution with the pressure as Schur variable under GMRES and with the application's block
    parameters (0, 4) under the flexible outer solver.  The application's remaining setting -- SetOuterSolver(PREONLY) with two cycles per Newton
    step -- does not converge on THIS discretisation with either block smoother: SteadyNavierStokesParallel/main.cpp:96-103 solves an equal-order
    Q1/Q1 system (its assembly callback carries the stabilisation, a non-zero pressure block), while BASELINE config 4 names Taylor-Hood Q2/Q1,
    whose pressure block is zero (every pressure pivot of a block needs the shift)."""
    from oracle import femus_oracle_ns as ns
    lib = os.path.join(ROOT, "femus_amd", "lib")
    exe = str(tmp_path / "navier_stokes_adapters")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "femus_amd", "csrc", "adapters")], stdout=subprocess.DEVNULL)
    subprocess.check_call(["g++", "-O1", "-std=c++17"] + INC + [os.path.join(ROOT, "tests", "cpp", "navier_stokes_adapters.cpp"), "-o", exe,
                           "-L" + lib, "-lfemus_hip_adapters", "-lfemus_hip", "-Wl,-rpath," + lib])
    out = str(tmp_path / "ns.bin")
    log = subprocess.check_output([exe, "4", "3", "0.01", out, str(nschur), str(nblock), "0", "2" if outer == "fgmres" else "0", "0"], text=True)
    _, lays, sols, hist = ns.solve_cavity(4, 4, 3, 0.01, (-0.5, -0.5, 0.0), (0.5, 0.5, 0.0), linear="direct")
    steps = int(log.split("newton steps = ")[1].split()[0])
    assert len(hist) <= steps <= len(hist) + 8
    sol = np.fromfile(out)
    assert np.linalg.norm(sol - sols[-1]) <= 1e-8 * np.linalg.norm(sols[-1])


def test_the_shipped_navier_stokes_application_with_its_own_discretisation_and_settings(tmp_path):
    """applications/003_NavierStokes/SteadyNavierStokesParallel as it is shipped, over the C++ adapters: equal-order LAGRANGE FIRST velocity / pressure with
    the Franca-Frey stabilised callback (main.cpp:96-108, :390-925 -> fh_assemble_navier_stokes_stab), box10x10 refined to 80 x 80 with the coarse levels
    ERASED (:89-92: one level, so every linear solve is the exact one of the level -- MGInit / MGSetLevel / MGSolve of the adapter end in the pivoted
    fronts of the sparse exact solve, 19 683 unknowns), FEMuS_ASM solver type, blocks (0, 4), GMRES level solver, ILU_PRECOND, SetOuterSolver(PREONLY),
    two linear iterations per Newton step (:148-185), the callback's Reynolds continuation to 10 000 (:485-489).  Converges well inside the application's
    90 nonlinear iterations to the solution of the oracle's Newton loop (20 x 20 is compared value by value in tests/test_gpu_ns_stab.py; here the
    discrete solution of the 80 x 80 run is checked through the residual of the oracle's operator at the final Reynolds number on a 20 x 20 run)."""
    from oracle import femus_oracle as fo
    from oracle import femus_oracle_ns as ns
    import scipy.sparse.linalg as spla
    lib = os.path.join(ROOT, "femus_amd", "lib")
    exe = str(tmp_path / "navier_stokes_adapters")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "femus_amd", "csrc", "adapters")], stdout=subprocess.DEVNULL)
    subprocess.check_call(["g++", "-O1", "-std=c++17"] + INC + [os.path.join(ROOT, "tests", "cpp", "navier_stokes_adapters.cpp"), "-o", exe,
                           "-L" + lib, "-lfemus_hip_adapters", "-lfemus_hip", "-Wl,-rpath," + lib])
    for n in (20, 80):
        out = str(tmp_path / ("ns_stab_%d.bin" % n))
        log = subprocess.check_output([exe, str(n), "1", "0.0", out, "0", "4", "0", "1", "0", "1"], text=True)
        steps = int(log.split("newton steps = ")[1].split()[0])
        assert "Reynolds Number = 10000" in log and 14 <= steps <= 30, log[-1500:]
        sol = np.fromfile(out)
        if n == 20:          # the oracle's Newton loop with scipy's LU on the same mesh
            mo = fo.build_levels(n, n, 0, 1, (-0.5, -0.5, 0.0), (0.5, 0.5, 0.5))[0]
            lay = ns.NSLayoutEqualOrder(mo)
            assert sol.size == lay.n
            # boundary values as the adapter's GenerateBdc loop sets them: taken from the run itself (rows the penalty fixed)
            pat = ns.csr_pattern_sys(lay)
            A, b = ns.assemble_ns_stab(mo, lay, sol, 1e-4, pattern=pat)
            wall = np.zeros(mo.nnode, dtype=bool)
            for f, nodes in enumerate(fo.face_nodes(mo.geom)):
                wall[mo.elem_dof[np.where(mo.face_flag[:, f] < -1)[0]][:, nodes].ravel()] = True
            nq1 = lay.sizes[0]
            free = np.ones(lay.n, dtype=bool)
            bn = np.where(wall[:nq1])[0]
            free[bn] = False
            free[bn + nq1] = False
            corner = bn[(mo.coords[bn, 0] < -0.5 + 1e-8) & (mo.coords[bn, 1] < -0.5 + 1e-8)]
            free[corner + 2 * nq1] = False
            assert np.abs(b[free]).max() <= 1e-9 * max(np.abs(b).max(), 1.0)          # the discrete residual of the oracle vanishes at the adapter's solution
            v = sol[nq1:2 * nq1]
            assert v.max() == 1.0 and v.min() < -0.05


def test_the_reference_known_answer_test_over_the_cpp_adapters(tmp_path):
    """unittests/testNSSteadyDD (the reference's stored-number test of this path) written over the C++ adapters the way its main.cpp drives FEMuS: Gambit mesh,
    four levels, Q2 velocity + DISCONTINUOUS_POLYNOMIAL FIRST pressure (InitPde with solution type 4, KKoffset checked against fh_system_elem_dofs), its boundary
    conditions and initial velocity, solver type FEMuS_DEFAULT with GMRES level solvers around ILU_PRECOND, Galerkin chain by matrix_PtAP, MGInit / MGSetLevel /
    MGSolve per Newton step in a nonlinear F-cycle.  The level-3 norms the test asserts to 1e-6 (main.cpp:202-244) must come out to 1e-8."""
    from test_ns_known_answer import STORED
    lib = os.path.join(ROOT, "femus_amd", "lib")
    exe = str(tmp_path / "navier_stokes_adapters")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "femus_amd", "csrc", "adapters")], stdout=subprocess.DEVNULL)
    subprocess.check_call(["g++", "-O1", "-std=c++17"] + INC + [os.path.join(ROOT, "tests", "cpp", "navier_stokes_adapters.cpp"), "-o", exe,
                           "-L" + lib, "-lfemus_hip_adapters", "-lfemus_hip", "-Wl,-rpath," + lib])
    out = str(tmp_path / "nsdd.bin")
    neu = os.path.join(ROOT, "tests", "golden", "nsbenc.neu")
    log = subprocess.check_output([exe, "0", "4", "0.001", out, "0", "4", "0", "2", "0", "0", neu], text=True)      # GMRES level solvers, flexible outer GMRES
    steps = int(log.split("newton steps = ")[1].split()[0])
    assert 12 <= steps <= 30, log[-2000:]
    sol = np.fromfile(out)
    nel = 98 * 64
    nq2 = (sol.size - 3 * nel) // 2
    got = {"U": np.linalg.norm(sol[:nq2]), "V": np.linalg.norm(sol[nq2:2 * nq2]), "P": np.linalg.norm(sol[2 * nq2:])}
    rel = {k: abs(got[k] - STORED[k]) / STORED[k] for k in got}
    print("adapters, level-3 norms", got, rel, log[-400:])
    assert max(rel.values()) < 1e-8, rel


def test_hipvector_on_two_ranks_over_the_host_transport(tmp_path):
    """ownership offsets, global indices, ghost refresh, localize_to_all and the global reductions with two processes"""
    lib = os.path.join(ROOT, "femus_amd", "lib")
    exe = str(tmp_path / "adapter_two_ranks")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "femus_amd", "csrc", "adapters")], stdout=subprocess.DEVNULL)
    subprocess.check_call(["g++", "-O1", "-std=c++17"] + INC + [os.path.join(ROOT, "tests", "cpp", "adapter_two_ranks.cpp"), "-o", exe,
                           "-L" + lib, "-lfemus_hip_adapters", "-lfemus_hip", "-Wl,-rpath," + lib])
    out = subprocess.run([exe], text=True, capture_output=True, timeout=120)
    assert "TWO RANKS OK" in out.stdout, out.stdout + out.stderr
