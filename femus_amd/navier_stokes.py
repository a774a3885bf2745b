"""Host-side driver of the steady Navier-Stokes Newton / multigrid path: the calls
NonLinearImplicitSystem makes (src/08_equations/00_stationary/NonLinearImplicitSystem.cpp), expressed over the C-ABI.

    init()          <- LinearImplicitSystem::init :138-282    levels, system dof maps (GetSystemDof), sparsity, prolongators of
                                                              the stacked variables, ZeroInterpolatorDirichletNodes, GenerateBdc
    newton_step()   <- NonLinearImplicitSystem::MGsolve :216-322   assemble residual + Jacobian at the level-max, Galerkin chain,
                                                              MGInit / MGSetLevel (SetPenalty), linear cycles, UpdateSol
    converged()     <- HasNonLinearConverged :113-153          max over variables of ||Eps_k|| / ||Sol_k||
    mgsolve()       <- MGsolve :157-361, F_CYCLE               for every level-max: Newton iterations, then ProlongatorSol (:453-464)

Variables are U, V (, W) biquadratic and P linear (Taylor-Hood), stacked [U | V | (W) | P].  The level smoother is the block
Schwarz (Vanka) smoother the reference offers for this application as FEMuS_ASM
(applications/003_NavierStokes/SteadyNavierStokesParallel/main.cpp:166-167, petsc_asm/LinearEquationSolverPetscAsm.cpp).
All numerics run in libfemus_hip.so.
"""
import numpy as np

from . import capi


def cavity_boundary_condition(x, name, face_name, lo, hi):
    """SetBoundaryConditionCavityFlow (SteadyNavierStokesParallel/main.cpp:365-390) on the box [lo, hi]^dim: all velocity
    components Dirichlet; V = 1 on the moving wall (face name 4 of the box generator = the x = lo side) for lo < y < hi;
    pressure free except at the (lo, lo) corner"""
    if name == "P":
        return bool(np.all(x < lo + 1.0e-08)), 0.0
    value = 0.0
    if name == "V" and face_name == 4 and lo[1] < x[1] < hi[1]:
        value = 1.0
    return True, value


def generate_bdc(mesh, names, fes, offsets, fn):
    """MultiLevelSolution::GenerateBdc (MultiLevelSolution.cpp:725-840): nodes of faces with a boundary flag < -1 get the
    value of the boundary function evaluated at the node (face name = -(flag + 1)); returns the sorted system dofs and values.
    A later face overwrites an earlier one only with a Dirichlet value, as the reference's loop does."""
    ed, xy, ff = mesh.arrays()
    nfaces = ff.shape[1]
    nloc_of = {"linear": 2 ** mesh.dim, "biquadratic": 3 ** mesh.dim}
    face_loc = [capi.fe_face_nodes(mesh.geom, "biquadratic", f) for f in range(nfaces)]
    val = {}
    for k, (name, fe) in enumerate(zip(names, fes)):
        nck = nloc_of[fe]
        for iel, f in zip(*np.nonzero(ff < -1)):
            face_name = -(int(ff[iel, f]) + 1)
            for i in face_loc[f]:
                if i >= nck:
                    continue
                node = int(ed[iel, i])
                is_dir, v = fn(xy[node], name, face_name)
                if is_dir:
                    val[int(offsets[k]) + node] = float(v)
    idx = np.array(sorted(val), dtype=np.int32)
    return idx, np.array([val[i] for i in idx])


def open_boundary_faces(mesh, names, fn):
    """which boundary faces carry the pressure integral of 03_navier_stokes.hpp:185-290, and the pressure on each.  Per face with a boundary flag:
    the bdc callback of every velocity component at the mean of the Q2 face nodes (:216-262), the JacobianSur normal at face Gauss point 0, the
    LAST component d with |n_d| >= 1e-4 as the normal velocity (:264-275); the face is kept when that component is not Dirichlet there.
    Returns (face_nodes [nf, nfn], face_name [nf]); tau is asked per Gauss point by the caller"""
    ed, xy, ff = mesh.arrays()
    dim = mesh.dim
    faces, fnames = [], []
    for f in range(ff.shape[1]):
        loc = capi.fe_face_nodes(mesh.geom, "biquadratic", f)
        els = np.where(ff[:, f] < -1)[0]
        if els.size == 0:
            continue
        nodes = ed[els][:, loc]
        normals = capi.face_normals(mesh, "biquadratic", nodes, 0, coords=xy)
        for q, iel in enumerate(els):
            face_name = -(int(ff[iel, f]) + 1)
            centre = np.zeros(dim)
            for d in range(dim):
                for n in nodes[q]:
                    centre[d] += xy[n, d]
                centre[d] /= nodes.shape[1]
            is_dir = [fn(centre, names[d], face_name)[0] for d in range(dim)]
            comp = 0
            for d in range(dim):
                if abs(normals[q, d]) >= 1.0e-4:
                    comp = d
            if not is_dir[comp]:
                faces.append(nodes[q])
                fnames.append(face_name)
    nfn = 3 ** (dim - 1)
    return (np.array(faces, np.int32).reshape(-1, nfn), np.array(fnames, np.int32))


class NavierStokesMG:
    def __init__(self, ctx, nx, ny, nz, nlevels, nu, lo=(-0.5, -0.5, -0.5), hi=(0.5, 0.5, 0.5), omega=0.6, npre=2, npost=2,
                 order="seventh", boundary_condition=None, open_pressure=None):
        self.ctx, self.nlevels, self.nu = ctx, nlevels, nu
        self.omega, self.npre, self.npost, self.order = omega, npre, npost, order
        self.meshes = [capi.Mesh.box(nx, ny, nz, lo, hi)]
        for _ in range(1, nlevels):
            self.meshes.append(self.meshes[-1].refine())
        self.dim = self.meshes[0].dim
        self.names = ["U", "V", "W"][:self.dim] + ["P"]
        self.fes = ["biquadratic"] * self.dim + ["linear"]
        lo_, hi_ = np.array(lo[:self.dim], float), np.array(hi[:self.dim], float)
        self.bc = boundary_condition or (lambda x, name, face: cavity_boundary_condition(x, name, face, lo_, hi_))
        # open_pressure: {face name: number or capi.Expr} -- the pressure the bdc callback prescribes for "P" on the faces whose normal velocity is
        # free (03_navier_stokes.hpp:280); None: no face of the problem is open (the cavity) and the boundary integral is skipped
        self.open_pressure = open_pressure
        # coarsest level of the cycles (the reference's EraseCoarseLevels, MultiLevelMesh.cpp; SteadyNavierStokesParallel/main.cpp:92 uses it): the levels below are
        # still meshes of the nonlinear F-cycle, but a cycle at level-max ig stops at min(coarse_level, ig) and solves there exactly (sparse LU on general fronts)
        self.coarse_level = 0
        self.history = []

    # ---- LinearImplicitSystem::init --------------------------------------------------------------------------------
    def init(self):
        ctx = self.ctx
        nl = self.nlevels
        self.offsets, self.elem_sys, self.n = [], [], []
        self.bdc, self.bdc_val = [], []
        self.KK, self.asm, self.SOL, self.RES, self.EPS, self.RESC = [], [], [], [], [], []
        self.patches = []
        for l, m in enumerate(self.meshes):
            nd, off, es = capi.system_elem_dofs(m, self.fes)
            n = int(off[-1])
            self.offsets.append(off), self.elem_sys.append(es), self.n.append(n)
            idx, val = generate_bdc(m, self.names, self.fes, off, self.bc)
            self.bdc.append(idx), self.bdc_val.append(val)
            rp, col = capi.pattern_from_elements(es, n)
            K = ctx.matrix_csr(n, n, rp, col)
            self.KK.append(K)
            self.asm.append(capi.NSAssembler(ctx, m, K, self.order))
            sol = np.zeros(n)
            sol[idx] = val                                             # Initialize + boundary values
            self.SOL.append(ctx.vector_from(sol))
            self.RES.append(ctx.vector(n)), self.EPS.append(ctx.vector(n)), self.RESC.append(ctx.vector(n))
            self.patches.append(capi.vertex_patches(m, self.fes) if l > 0 else None)
        self.open_faces = [open_boundary_faces(m, self.names, self.bc) if self.open_pressure else None for m in self.meshes]
        # interpolation of the stacked variables: Psol for ProlongatorSol (untouched), P for the cycle (Dirichlet rows/cols zeroed)
        self.Psol, self.P = [None], [None]
        for l in range(1, nl):
            self.Psol.append(capi.build_system_prolongator(ctx, self.meshes[l - 1], self.meshes[l], self.fes))
            P = capi.build_system_prolongator(ctx, self.meshes[l - 1], self.meshes[l], self.fes)
            P.mat_zero_rows(self.bdc[l], 0.0)
            P.zero_cols(self.bdc[l - 1])
            self.P.append(P)
        self.A = {}          # (level-max, level) -> operator of the cycle
        self.mg = {}         # level-max -> multigrid
        return self

    # ---- one Newton iteration at level-max ig ----------------------------------------------------------------------------
    def prepare(self, ig):
        """assemble residual + Jacobian at the current SOL[ig]; Galerkin chain; SetPenalty; MGInit / MGSetLevel"""
        ctx = self.ctx
        self.asm[ig].assemble(self.KK[ig], self.RES[ig], self.SOL[ig], self.nu)
        self.add_open_boundary_pressure(ig)
        self.A[(ig, ig)] = self.KK[ig]
        c = min(self.coarse_level, ig)
        for l in range(ig, c, -1):                                     # PtAP chain from the un-penalised operators
            if (ig, l - 1) not in self.A:
                self.A[(ig, l - 1)] = capi.Mat.ptap(self.P[l], self.A[(ig, l)])
            else:
                self.A[(ig, l - 1)].ptap_numeric(self.P[l], self.A[(ig, l)])
        for l in range(c, ig + 1):
            self.A[(ig, l)].mat_zero_rows(self.bdc[l], 1.0)
        if self.bdc[ig].size:                                          # ZerosBoundaryResiduals
            self.RES[ig].set(self.bdc[ig], np.zeros(self.bdc[ig].size))
        if ig not in self.mg:
            self.mg[ig] = capi.Multigrid(ctx, ig - c + 1)
            for l in range(c + 1, ig + 1):
                self.mg[ig].set_level_patches(l - c, *self.patches[l])
        mg = self.mg[ig]
        for l in range(c, ig + 1):
            mg.set_level(l - c, self.A[(ig, l)], self.P[l] if l > c else None, None, getattr(self, "smoother", capi.SMOOTH_VANKA), self.omega,
                         self.npre if l > c else 1, self.npost if l > c else 0)
        mg.setup()
        return mg

    def add_open_boundary_pressure(self, ig):
        """RES -= int phi tau n over the open faces (data only: the Jacobian does not change)"""
        if not self.open_pressure:
            return
        faces, fnames = self.open_faces[ig]
        if faces.shape[0] == 0:
            return
        off = self.offsets[ig][:self.dim]
        consts = {k: v for k, v in self.open_pressure.items() if not isinstance(v, capi.Expr)}
        exprs = {k: v for k, v in self.open_pressure.items() if isinstance(v, capi.Expr)}
        sel = np.isin(fnames, list(consts))
        if sel.any():
            capi.assemble_pressure_faces(self.ctx, self.meshes[ig], self.RES[ig], faces[sel], np.array([consts[int(k)] for k in fnames[sel]], float), off)
        sel = np.isin(fnames, list(exprs))
        if sel.any():
            capi.assemble_pressure_faces(self.ctx, self.meshes[ig], self.RES[ig], faces[sel], [(e, fnames[sel] == k) for k, e in exprs.items()], off)

    def newton_step(self, ig, lin_rtol=1e-10, lin_maxit=60, restart=30):
        mg = self.prepare(ig)
        its, rn = mg.solve(self.RES[ig], self.EPS[ig], outer="gmres", rtol=lin_rtol, atol=1e-50, maxit=lin_maxit, restart=restart)
        self.SOL[ig].add(1.0, self.EPS[ig])                            # Solution::UpdateSol
        return its, rn

    def nonlinear_eps(self, ig):
        """HasNonLinearConverged: max_k ||Eps_k||_2 / (||Sol_k||_2 + 1e-50)"""
        eps, sol = self.EPS[ig].to_numpy(), self.SOL[ig].to_numpy()
        off = self.offsets[ig]
        return max(np.linalg.norm(eps[off[k]:off[k + 1]]) / (np.linalg.norm(sol[off[k]:off[k + 1]]) + 1e-50) for k in range(len(self.fes)))

    def newton(self, ig, tol=1e-10, max_newton=30, **kw):
        for it in range(max_newton):
            its, rn = self.newton_step(ig, **kw)
            e = self.nonlinear_eps(ig)
            self.history.append((ig, it, e, its))
            if e < tol:
                return True
        return False

    def prolongator_sol(self, ig):
        """LinearImplicitSystem::ProlongatorSol: Sol[ig] = P_mesh Sol[ig-1] for every variable"""
        self.SOL[ig].matrix_mult(self.SOL[ig - 1], self.Psol[ig])

    def mgsolve(self, tol=1e-10, max_newton=30, **kw):
        """NonLinearImplicitSystem::MGsolve with F_CYCLE (nested iteration)"""
        ok = True
        for ig in range(self.nlevels):
            ok = self.newton(ig, tol, max_newton, **kw)
            if ig + 1 < self.nlevels:
                self.prolongator_sol(ig + 1)
        return ok

    def destroy(self):
        for mg in self.mg.values():
            mg.destroy()
        for a in self.asm:
            a.destroy()
        seen = set()
        for m in list(self.A.values()) + self.KK + self.P + self.Psol:
            if m is not None and id(m) not in seen:
                seen.add(id(m))
                m.destroy()
        for m in self.meshes:
            m.destroy()


class NavierStokesPwMG(NavierStokesMG):
    """The same driver for the discretisation of the reference's known-answer test (unittests/testNSSteadyDD/main.cpp): Q2 velocity with the
    DISCONTINUOUS piecewise-linear pressure (AddSolution("P", DISCONTINUOUS_POLYNOMIAL, FIRST), :97) on given meshes, the level solver that test sets
    (SetSolverFineGrids(GMRES) + SetPreconditionerFineGrids(ILU_PRECOND), :150-152: GMRES around ILU(0) on every level above the coarsest, exact solve
    below) inside the nonlinear F-cycle.  Pressure dofs are owned by the elements (i * nel + iel behind the velocities); their interpolation is the
    element prolongator of solution type 4 (ElemType.cpp:446-520: the coarse function at the child's centre for the constant, half the coarse slope for
    the two / three linear functions)."""

    def __init__(self, ctx, meshes, nu, boundary_condition, omega=1.0, npre=1, npost=1, order="seventh", level_gmres_its=4):
        self.ctx, self.nlevels, self.nu = ctx, len(meshes), nu
        self.omega, self.npre, self.npost, self.order = omega, npre, npost, order
        self.meshes = list(meshes)
        self.dim = meshes[0].dim
        self.names = ["U", "V", "W"][:self.dim] + ["P"]
        self.fes = ["biquadratic"] * self.dim + ["pwlinear"]
        self.bc = boundary_condition
        self.open_pressure = None
        self.level_gmres_its = level_gmres_its
        self.history = []

    def init(self):
        import scipy.sparse as sp
        ctx, nl, dim = self.ctx, self.nlevels, self.dim
        self.offsets, self.elem_sys, self.n = [], [], []
        self.bdc, self.bdc_val, self.bdc_phys = [], [], []
        self.KK, self.asm, self.SOL, self.RES, self.EPS, self.RESC = [], [], [], [], [], []
        # levels made by selective refinement (the two upper levels of the reference's test, main.cpp:55-82, 262-280): the velocities hang at the interfaces and are
        # tied to their masters by PPamr, the element-owned pressures need nothing (LinearImplicitSystem.cpp:247-258, 329-335; NonLinearImplicitSystem.cpp:213-236)
        self.Pamr, self.Aamr = [None] * nl, [None] * nl
        for l, m in enumerate(self.meshes):
            nd, off, es = capi.system_elem_dofs(m, self.fes)               # GetSystemDof with the element-owned pressure (solution type 4)
            n = int(off[-1])
            self.offsets.append(off), self.elem_sys.append(es), self.n.append(n)
            idx, val = generate_bdc(m, self.names[:dim], self.fes[:dim], off, self.bc)       # the pressure carries no boundary condition
            self.bdc_phys.append((idx, val))
            if not m.elem_levels()[1]:
                Pq = capi.build_amr_prolongator(ctx, m, "biquadratic")
                hang = m.amr_constraints("biquadratic")[0].astype(np.int64)
                blocks = [Pq.to_scipy()] * dim + [sp.identity((dim + 1) * m.nel, format="csr")]
                Pq.destroy()
                self.Pamr[l] = ctx.matrix_scipy(sp.block_diag(blocks).tocsr())
                hsys = np.concatenate([hang + off[k] for k in range(dim)])
                hsys = np.setdiff1d(hsys, idx)
                order = np.argsort(np.concatenate([idx, hsys]))
                idx, val = np.concatenate([idx, hsys])[order].astype(np.int32), np.concatenate([val, np.zeros(hsys.size)])[order]
            self.bdc.append(idx), self.bdc_val.append(val)
            K = ctx.matrix_from_elements(es, n)
            self.KK.append(K)
            self.asm.append(capi.NSPwAssembler(ctx, m, K, self.order))
            self.SOL.append(ctx.vector(n))
            self.RES.append(ctx.vector(n)), self.EPS.append(ctx.vector(n)), self.RESC.append(ctx.vector(n))
        self.open_faces = [None] * nl
        self.Psol, self.P = [None], [None]
        for l in range(1, nl):
            self.Psol.append(capi.build_system_prolongator(ctx, self.meshes[l - 1], self.meshes[l], self.fes))
            P = capi.build_system_prolongator(ctx, self.meshes[l - 1], self.meshes[l], self.fes)
            if self.Pamr[l - 1] is not None:                               # PP[l] <- PP[l] PPamr[l-1]
                PA = P.matmul(self.Pamr[l - 1])
                P.destroy()
                P = PA
            P.mat_zero_rows(self.bdc[l], 0.0)
            P.zero_cols(self.bdc[l - 1])
            self.P.append(P)
        self.A, self.mg = {}, {}
        return self

    def set_state(self, level, values):
        """Initialize(...) + the boundary values of GenerateBdc on one level"""
        x = np.array(values, float)
        idx, val = self.bdc_phys[level]
        x[idx] = val
        self.SOL[level].upload(x)

    def prepare(self, ig):
        ctx = self.ctx
        self.asm[ig].assemble(self.KK[ig], self.RES[ig], self.SOL[ig], self.nu)
        if self.Pamr[ig] is not None:                                      # RES <- PPamr^T RES ; KK <- PPamr^T KK PPamr
            self.RESC[ig].matrix_mult_transpose(self.RES[ig], self.Pamr[ig])
            self.RES[ig].assign(self.RESC[ig])
            if self.Aamr[ig] is None:
                self.Aamr[ig] = capi.Mat.ptap(self.Pamr[ig], self.KK[ig])
            else:
                self.Aamr[ig].ptap_numeric(self.Pamr[ig], self.KK[ig])
            self.A[(ig, ig)] = self.Aamr[ig]
        else:
            self.A[(ig, ig)] = self.KK[ig]
        for l in range(ig, 0, -1):
            if (ig, l - 1) not in self.A:
                self.A[(ig, l - 1)] = capi.Mat.ptap(self.P[l], self.A[(ig, l)])
            else:
                self.A[(ig, l - 1)].ptap_numeric(self.P[l], self.A[(ig, l)])
        for l in range(ig + 1):
            self.A[(ig, l)].mat_zero_rows(self.bdc[l], 1.0)
        if self.bdc[ig].size:
            self.RES[ig].set(self.bdc[ig], np.zeros(self.bdc[ig].size))
        if ig not in self.mg:
            self.mg[ig] = capi.Multigrid(ctx, ig + 1)
            for l in range(1, ig + 1):
                self.mg[ig].set_level_solver(l, "gmres", 30)
        mg = self.mg[ig]
        for l in range(ig + 1):
            its = self.level_gmres_its
            mg.set_level(l, self.A[(ig, l)], self.P[l] if l > 0 else None, None, capi.SMOOTH_ILU0, self.omega, its * self.npre if l > 0 else 1,
                         its * self.npost if l > 0 else 0)
        mg.setup()
        return mg

    def newton_step(self, ig, lin_rtol=1e-10, lin_maxit=60, restart=30):
        mg = self.prepare(ig)
        its, rn = mg.solve(self.RES[ig], self.EPS[ig], outer="fgmres" if ig > 0 else "preonly", rtol=lin_rtol, atol=1e-50, maxit=lin_maxit, restart=restart)
        if self.Pamr[ig] is not None:                                      # EPS <- PPamr EPS: the hanging velocities follow their masters (LinearImplicitSystem.cpp:487-491)
            self.RESC[ig].matrix_mult(self.EPS[ig], self.Pamr[ig])
            self.EPS[ig].assign(self.RESC[ig])
        self.SOL[ig].add(1.0, self.EPS[ig])
        return its, rn

    def destroy(self):
        for m in self.Pamr + self.Aamr:
            if m is not None:
                m.destroy()
        self.Pamr, self.Aamr = [], []
        super().destroy()
