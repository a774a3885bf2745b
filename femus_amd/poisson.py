"""Host-side driver of the Poisson geometric-multigrid hot path: the calls LinearImplicitSystem makes
(src/08_equations/00_stationary/LinearImplicitSystem.cpp), expressed over the C-ABI.

    init()      <- LinearImplicitSystem::init :138-282      levels, sparsity, BuildProlongatorMatrix, BuildAmrProlongatorMatrix,
                                                            PP <- PP * PPamr, ZeroInterpolatorDirichletNodes
    assemble()  <- _assemble_system_function :325           the batched element loop (fh_assemble_poisson)
    prepare()   <- MGsolve :347-383                         Galerkin chain KK[l-1] = PP[l]^T KK[l] PP[l] (from the un-penalised
                                                            matrices), then MGInit / MGSetLevel (SetPenalty on every level)
    vcycle()/mgsolve() <- Vcycle :468-497 / MGSolve        ZerosBoundaryResiduals, outer solver preconditioned by the cycle

Used by tests/ and bench.py; all numerics run in libfemus_hip.so.
"""
import numpy as np

from . import capi


class PoissonMG:
    def __init__(self, ctx, nx, ny, nz, nlevels, fe="biquadratic", order="seventh", lo=(0., 0., 0.), hi=(1., 1., 1.),
                 omega=2. / 3., npre=2, npost=2, coarse="galerkin", source_kind=0, params=(1.0,), meshes=None,
                 smoother=0, dirichlet=None, source_expr=None, source_scale=1.0, elementwise_galerkin=True):
        self.ctx = ctx
        self.fe, self.order = fe, order
        self.nlevels = nlevels
        self.omega, self.npre, self.npost = omega, npre, npost
        self.coarse = coarse
        self.source_kind, self.params = source_kind, params
        self.smoother = smoother                  # capi.SMOOTH_JACOBI / SMOOTH_GS_COLOR
        self.dirichlet = dirichlet                # per level: Dirichlet dof lists when only some faces are Dirichlet
        self.source_expr, self.source_scale = source_expr, source_scale     # capi.Expr: f = scale * expr(x, y, z, t)
        self.elementwise_galerkin = elementwise_galerkin
        if meshes is not None:
            self.meshes = list(meshes)
        else:
            self.meshes = [capi.Mesh.box(nx, ny, nz, lo, hi)]
            for _ in range(1, nlevels):
                self.meshes.append(self.meshes[-1].refine(ctx))      # on the device; the levels stay resident for init()
        dim = self.meshes[0].dim
        self.nc = {"linear": 2 ** dim, "serendipity": 8 if dim == 2 else 20, "biquadratic": 3 ** dim}[fe]
        self.mg = None

    # ---- LinearImplicitSystem::init ---------------------------------------------------------------------------
    def init(self):
        ctx, fe = self.ctx, self.fe
        top = self.nlevels - 1
        self.ndof = [m.n_dofs(fe) for m in self.meshes]
        # non-homogeneous (adaptively refined) levels: P_amr and the hanging dofs, which join the Dirichlet rows as
        # "AMR artificial Dirichlet" (_Bdc = 1 < 1.5, MultiLevelSolution.cpp:725-760; LinearImplicitSystem.cpp:247-252)
        self.Pamr = [None] * self.nlevels
        self.hanging = [np.zeros(0, np.int32)] * self.nlevels
        for l, m in enumerate(self.meshes):
            if not m.elem_levels()[1]:
                self.Pamr[l] = capi.build_amr_prolongator(ctx, m, fe)
                self.hanging[l] = m.amr_constraints(fe)[0]
        self.amr = any(p is not None for p in self.Pamr)
        phys = [m.dirichlet_dofs(fe) for m in self.meshes] if self.dirichlet is None else self.dirichlet
        self.bdc = [np.union1d(phys[l], self.hanging[l]).astype(np.int32) for l in range(self.nlevels)]
        if not self.amr and self.dirichlet is None:
            self.P = [None] + [capi.build_prolongator(ctx, self.meshes[l - 1], self.meshes[l], fe, zero_bdc=True)
                               for l in range(1, self.nlevels)]
        else:
            assert self.coarse == "galerkin", "adaptive levels use the Galerkin chain"
            self.P = [None]
            for l in range(1, self.nlevels):
                P = capi.build_prolongator(ctx, self.meshes[l - 1], self.meshes[l], fe, zero_bdc=False)
                if self.Pamr[l - 1] is not None:            # PP[l] <- PP[l] * PPamr[l-1]   (:253-258)
                    PA = P.matmul(self.Pamr[l - 1])
                    P.destroy()
                    P = PA
                P.mat_zero_rows(self.bdc[l], 0.0)           # ZeroInterpolatorDirichletNodes (:1032-1120)
                P.zero_cols(self.bdc[l - 1])
                self.P.append(P)
        # per-level operators: finest (and, for coarse == "rediscretise", every level) carries the element pattern
        self.A = [None] * self.nlevels
        self.KK = [None] * self.nlevels          # assembled matrix where it differs from the operator of the cycle (AMR)
        self.asm = [None] * self.nlevels
        # Galerkin chain of a uniformly refined Q2 hierarchy: element by element from the element matrices of the next finer level
        # (fh_assembler_galerkin) instead of the sparse triple product -- every level then carries the element pattern and an assembler
        self.gal_elem = (self.coarse == "galerkin" and self.elementwise_galerkin and not self.amr and fe == "biquadratic" and self.nlevels > 1)
        levels = range(self.nlevels) if (self.coarse == "rediscretise" or self.gal_elem) else [top]
        for l in levels:
            K = ctx.matrix_from_mesh(self.meshes[l], fe)                       # GetSparsityPatternSize + init, on the device
            self.asm[l] = capi.Assembler(ctx, self.meshes[l], fe, K, self.order)
            if self.Pamr[l] is not None:
                self.KK[l] = K
            else:
                self.A[l] = K
        n = self.ndof[top]
        self.RES, self.EPS, self.EPSC, self.RESC = ctx.vector(n), ctx.vector(n), ctx.vector(n), ctx.vector(n)
        self.SOL = ctx.vector(n)
        self.sol_lvl = [None] * self.nlevels
        self.bdc_dev = [capi.Index(ctx, b) for b in self.bdc]      # BuildBdcIndex, once (device-resident)
        self._child = [None] * self.nlevels
        return self

    # ---- assembly of the level to assemble (the finest) --------------------------------------------------------
    def assemble(self, level=None):
        l = self.nlevels - 1 if level is None else level
        res = self.RES if l == self.nlevels - 1 else self.ctx.vector(self.ndof[l])
        K = self.KK[l] if self.KK[l] is not None else self.A[l]
        sol = self.SOL if l == self.nlevels - 1 else None
        if self.source_expr is not None:
            self.asm[l].assemble_expr(K, res, sol, self.source_expr, self.source_scale)
        else:
            self.asm[l].assemble(K, res, sol, self.source_kind, self.params)
        if self.Pamr[l] is not None:
            # RES <- PPamr^T RES ; KK <- PPamr^T KK PPamr   (LinearImplicitSystem.cpp:329-335)
            self.RESC.matrix_mult_transpose(res, self.Pamr[l])
            res.assign(self.RESC)
            if self.A[l] is None:
                self.A[l] = capi.Mat.ptap(self.Pamr[l], K)
            else:
                self.A[l].ptap_numeric(self.Pamr[l], K)
        return res

    # ---- MGsolve preparation ------------------------------------------------------------------------------------
    def level_operators(self):
        """the operator part of the preparation: Galerkin chain KK[l-1] = PP[l]^T KK[l] PP[l] from the un-penalised matrices
        (symbolic once, numeric on every later call), then SetPenalty on every level"""
        top = self.nlevels - 1
        if self.coarse == "galerkin":
            for l in range(top, 0, -1):            # PtAP chain from the un-penalised operators
                if self.gal_elem:
                    if self._child[l - 1] is None:
                        self._child[l - 1] = self.meshes[l - 1].child_elems()
                    self.asm[l - 1].galerkin_from(self.asm[l], self._child[l - 1], self.bdc[l], self.bdc[l - 1], self.A[l - 1])
                    continue
                if self.A[l - 1] is None:
                    self.A[l - 1] = capi.Mat.ptap(self.P[l], self.A[l])
                else:
                    self.A[l - 1].ptap_numeric(self.P[l], self.A[l])
        else:
            for l in range(top):
                self.assemble(l)
        for l in range(self.nlevels):              # MGSetLevel: SetPenalty
            self.bdc_dev[l].zero_rows(self.A[l], 1.0)

    def prepare(self):
        ctx = self.ctx
        self.level_operators()
        if self.mg is None:
            self.mg = capi.Multigrid(ctx, self.nlevels)
            # where the unknowns of the coarsest level lie: the exact coarse solve dissects its dense problem with it (coarse_nd)
            xy0 = self.meshes[0].arrays()[1]
            self.mg.set_coarse_coords(xy0[:self.ndof[0]])
        for l in range(self.nlevels):
            self.mg.set_level(l, self.A[l], self.P[l], None, self.smoother, self.omega, self.npre if l > 0 else 1,
                              self.npost if l > 0 else 0)
        self.mg.setup()
        return self

    def prepare_operators_only(self):
        """Galerkin chain + SetPenalty without building the cycle (used by the domain-decomposition setup)"""
        self.level_operators()

    def destroy_device_objects(self):
        for a in self.asm:
            if a is not None:
                a.destroy()
        for m in self.A + self.P + self.KK + self.Pamr:
            if m is not None:
                m.destroy()
        self.asm, self.A, self.P, self.KK, self.Pamr = [], [], [], [], []

    def zero_boundary_residuals(self):
        self.bdc_dev[-1].set(self.RES, 0.0)

    # ---- one preconditioner application / the MG solve ----------------------------------------------------------
    def vcycle(self, b=None, x=None):
        self.mg.vcycle(self.RES if b is None else b, self.EPSC if x is None else x)

    def mgsolve(self, outer="gmres", rtol=1e-10, atol=1e-50, maxit=100, restart=30):
        """MGSolve: ZerosBoundaryResiduals; KSPSolve(RES -> EPSC); RESC = KK EPSC; RES -= RESC; EPS += EPSC"""
        self.zero_boundary_residuals()
        its, rn = self.mg.solve(self.RES, self.EPSC, outer=outer, rtol=rtol, atol=atol, maxit=maxit, restart=restart)
        self.RESC.matrix_mult(self.EPSC, self.A[-1])
        self.RES.add(-1.0, self.RESC)
        self.EPS.add(1.0, self.EPSC)
        return its, rn

    def update_sol(self):
        """Solution::UpdateSol: Sol += Eps (on adaptive levels EPS <- PPamr EPS first, LinearImplicitSystem.cpp:487-491)"""
        if self.Pamr[-1] is not None:
            self.EPSC.matrix_mult(self.EPS, self.Pamr[-1])
            self.EPS.assign(self.EPSC)
        self.SOL.add(1.0, self.EPS)
        self.EPS.zero()

    def destroy(self):
        if self.mg is not None:
            self.mg.destroy()
        for a in self.asm:
            if a is not None:
                a.destroy()
        for m in self.A + self.P + self.KK + self.Pamr + self.bdc_dev:
            if m is not None:
                m.destroy()
        for m in self.meshes:
            m.destroy()
