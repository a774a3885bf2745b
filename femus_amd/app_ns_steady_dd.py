"""unittests/testNSSteadyDD/main.cpp over the C-ABI -- the reference's known-answer test as an application of this library.

    python -m femus_amd.app_ns_steady_dd [mesh.neu]        exit code 0 when the four level-3 norms meet the reference's stored numbers to its 1e-6

    meshes()          <- main.cpp:55-82: ReadCoarseMesh(nsbenc.neu), RefineMesh(6, 4, SetRefinementFlag): four uniform levels, then two where
                         SetRefinementFlag (:262-280) says so -- Gambit group 5 (fh_mesh_elem_groups), elements of the current level only
    navier_stokes()   <- :84-170: U, V LAGRANGE SECOND, P DISCONTINUOUS_POLYNOMIAL FIRST, boundary conditions :290-392, InitVariableU :281-287,
                         NonLinearImplicitSystem with F_CYCLE, at most 3 nonlinear iterations to 1e-4, 2 linear cycles, one smoothing step before and
                         after, SetSolverFineGrids(GMRES) + ILU_PRECOND, SetTolerances(..., 4); callback AssembleMatrixResNS -> fh_assemble_navier_stokes
                         on an fh_ns_pw_assembler
    temperature()     <- :172-200: T LAGRANGE SECOND, LinearImplicitSystem with V_CYCLE (the finest level only), 6 linear iterations to 1e-9, the same
                         level solvers; callback AssembleMatrixResT -> fh_assemble_advection_diffusion with the computed velocity
    norms()           <- :202-244: l2 norms of U, V, P, T on level 3 against 35.68179309424519, 6.86749406268887, 3.10222750612995, 57.69748694700662

All numerics run in libfemus_hip.so; the non-homogeneous levels use PPamr as LinearImplicitSystem / NonLinearImplicitSystem do."""
import sys

import numpy as np

from . import capi
from .known_answer import MESH, STORED, boundary_condition, inflow_profile
from .navier_stokes import NavierStokesPwMG, generate_bdc
from .poisson import PoissonMG

STORED_T = 57.69748694700662
INVERSE_REYNOLDS = 0.001           # Fluid(par, 0.001, 1, "Newtonian", 0.001, 1.): mu / (rho U L) (:108-114)
INVERSE_PECLET = 0.001             # Prandtl = mu cp / k = 1


def meshes(ctx, mesh_file=MESH, n_uniform=4, n_selective=2):
    ms = [capi.Mesh.read_gambit(mesh_file)]
    for l in range(1, n_uniform + n_selective):
        flags = None
        if l >= n_uniform:
            group, _ = ms[-1].elem_groups()
            level, _ = ms[-1].elem_levels()
            flags = ((group == 5) & (level == ms[-1].level)).astype(np.uint8)         # SetRefinementFlag: group 5 yes, 6 only below level 2, 7 never
        ms.append(ms[-1].refine_device(ctx, flags))
    return ms


def temperature_bc(x, name, face):
    """main.cpp:375-391: 1 on the inflow, 5 on the cylinder, nothing prescribed on walls and outflow"""
    if face == 1:
        return True, 1.0
    if face == 4:
        return True, 5.0
    return False, 0.0


def navier_stokes(ctx, ms, reference_limits=True):
    pb = NavierStokesPwMG(ctx, ms, INVERSE_REYNOLDS, boundary_condition, level_gmres_its=1 if reference_limits else 4).init()
    x = np.zeros(pb.n[0])
    x[:ms[0].nnode] = inflow_profile(ms[0].arrays()[1][:, 1])
    pb.set_state(0, x)
    if reference_limits:
        pb.mgsolve(tol=1e-4, max_newton=3, lin_rtol=1e-12, lin_maxit=8, restart=4)
    else:
        pb.mgsolve(tol=1e-10, max_newton=20, lin_rtol=1e-10, lin_maxit=150)
    return pb


class _Callback:
    """AssembleMatrixResT behind the assemble() slot of PoissonMG"""

    def __init__(self, asm, velocity):
        self.asm, self.velocity = asm, velocity

    def assemble(self, K, res, sol, source_kind, params):
        self.asm.assemble(K, res, sol, self.velocity, INVERSE_PECLET)

    def destroy(self):
        self.asm.destroy()


def temperature(ctx, ms, velocity, linear_iterations=6, abs_tol=1e-9):
    """LinearImplicitSystem::MGsolve with V_CYCLE: only the finest level is solved (LinearImplicitSystem.cpp:300-303)"""
    nl = len(ms)
    bdc, val = [], []
    for m in ms:
        i, v = generate_bdc(m, ["T"], ["biquadratic"], np.array([0, m.nnode]), temperature_bc)
        bdc.append(i), val.append(v)
    pb = PoissonMG(ctx, 0, 0, 0, nl, meshes=ms, omega=1.0, npre=1, npost=1, coarse="galerkin", smoother=capi.SMOOTH_ILU0, dirichlet=bdc,
                   elementwise_galerkin=False).init()
    top = nl - 1
    pb.asm[top].destroy()
    K = pb.KK[top] if pb.KK[top] is not None else pb.A[top]
    pb.asm[top] = _Callback(capi.AdvDiffAssembler(ctx, ms[top], K), velocity)
    t0 = np.zeros(ms[top].nnode)
    t0[bdc[top]] = val[top]                                   # GenerateBdc: the boundary values sit in the solution vector
    pb.SOL.upload(t0)
    pb.mg = capi.Multigrid(ctx, nl)
    pb.mg.set_coarse_coords(ms[0].arrays()[1][:pb.ndof[0]])
    for l in range(1, nl):
        pb.mg.set_level_solver(l, "gmres", 30)                # SetSolverFineGrids(GMRES) around SetPreconditionerFineGrids(ILU_PRECOND)
    pb.assemble()
    pb.prepare()
    history = []
    for it in range(linear_iterations):                       # SetMaxNumberOfLinearIterations(6), SetAbsoluteLinearConvergenceTolerance(1.e-9)
        its, _ = pb.mgsolve(outer="fgmres", rtol=1e-12, maxit=4, restart=4)
        history.append((its, pb.RES.l2_norm()))
        if history[-1][1] < abs_tol:
            break
    pb.update_sol()
    return pb, history, (bdc, val)


def norms(ms, ns, t_levels):
    s = ns.SOL[3].to_numpy()
    nq = ms[3].nnode
    return {"U": float(np.linalg.norm(s[:nq])), "V": float(np.linalg.norm(s[nq:2 * nq])), "P": float(np.linalg.norm(s[2 * nq:])),
            "T": float(np.linalg.norm(t_levels[3]))}


def run(ctx, mesh_file=MESH, reference_limits=True, verbose=True):
    ms = meshes(ctx, mesh_file)
    ns = navier_stokes(ctx, ms, reference_limits)
    top = len(ms) - 1
    tp, t_hist, (bdc, val) = temperature(ctx, ms, ns.SOL[top])
    # T below the finest level: Initialize("T") = 0 and the boundary values of GenerateBdc (the V-cycle system never touches those vectors)
    t_levels = []
    for l, m in enumerate(ms):
        t = np.zeros(m.nnode)
        t[bdc[l]] = val[l]
        t_levels.append(t if l < top else tp.SOL.to_numpy())
    got = norms(ms, ns, t_levels)
    stored = dict(STORED, T=STORED_T)
    rel = {k: abs(got[k] - stored[k]) / stored[k] for k in stored}
    if verbose:
        print("levels", [m.nel for m in ms], "elements; Navier-Stokes (Newton step, relative update, outer iterations):",
              [(h[0], h[1], float("%.1e" % h[2]), h[3]) for h in ns.history])
        print("temperature on the finest level (outer iterations, residual):", [(a, float("%.1e" % b)) for a, b in t_hist],
              "range %.4f .. %.4f" % (t_levels[top].min(), t_levels[top].max()))
        for k in ("U", "V", "P", "T"):
            print("Solution %s l2norm: %.14f   stored %.14f   relative distance %.1e" % (k, got[k], stored[k], rel[k]))
    out = {"norms": got, "stored": stored, "relative_distance": rel, "passed": max(rel.values()) <= 1e-6, "elements": [m.nel for m in ms],
           "unknowns_finest": int(ns.n[top]), "temperature_range": (float(t_levels[top].min()), float(t_levels[top].max()))}
    tp.meshes = []                 # the meshes are shared: destroyed once, by the Navier-Stokes driver
    tp.destroy()
    ns.destroy()
    return out


if __name__ == "__main__":
    import femus_amd
    r = run(femus_amd.Context(0), sys.argv[1] if len(sys.argv) > 1 else MESH)
    print("testNSSteadyDD:", "PASSED" if r["passed"] else "FAILED", "(the reference asserts 1e-6)")
    sys.exit(0 if r["passed"] else 1)
