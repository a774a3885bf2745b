"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy/scipy) of FEMuS's steady Navier-Stokes Newton/multigrid path
(BASELINE config "003_NavierStokes lid-driven cavity, Q2/Q1 Taylor-Hood, Newton + GMG-preconditioned GMRES").

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  Parity status: the Taylor-Hood weak form,
its Newton linearisation, the FE tables / Jacobian on curved elements and the boundary treatment are PINNED against the reference's own
known-answer test (unittests/testNSSteadyDD/main.cpp:202-244 stores the level-3 norms of U, V, P, T of the cylinder flow; with that test's
pressure space -- NSLayoutPwLinear / PwLinearPressure at the end of this file -- elem_ns_batch / assemble_ns reproduce them to 4e-10,
tests/test_ns_known_answer.py).  Unpinned (the reference needs PETSc + adept to run it and stores no numbers): the multigrid / smoother
restatements and the stabilised equal-order form; anchored on the call sites below, on finite-difference / complex-step checks of the
hand-derived Jacobians against the residuals and on domain properties.

Restated from
  src/08_equations/assemble/03_navier_stokes.hpp:330-395   Gauss loop: aResV[k][i] = (nu grad phi_i . grad u_k + phi_i (u . grad) u_k
                                                           - p d_k phi_i) w ; aResP[i] = -(div u) psi_i w ; Res = -aRes
  src/08_equations/assemble/Assemble_jacobian.cpp:39-72     Jac = d aRes / d sol (adept tape in the reference, derived by hand here)
  LinearEquation.cpp:76-85, 212-237                         system dof = KKoffset[k] + mesh dof (variables stacked per rank)
  NonLinearImplicitSystem.cpp:157-361                       nonlinear F-cycle: for every level-max: Newton iterations
                                                           {assemble, Galerkin chain, MGInit/MGSetLevel, linear cycles, UpdateSol},
                                                           then ProlongatorSol to the next level
  NonLinearImplicitSystem.cpp:113-153                       HasNonLinearConverged: max_k ||Eps_k|| / ||Sol_k|| < tol
  LinearImplicitSystem.cpp:453-464                          ProlongatorSol: Sol_f = P_mesh Sol_c per variable
  applications/003_NavierStokes/SteadyNavierStokesParallel/main.cpp:365-390   cavity boundary conditions
  petsc_asm/LinearEquationSolverPetscAsm.cpp:91-276         element-block (Vanka) Schwarz smoother for saddle-point systems:
                                                           velocity dofs of the block's elements + the block's pressure dofs
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from . import femus_oracle as fo


class NSLayout:
    """variables U, V (, W) biquadratic and P linear; system dof = offset[k] + mesh dof (nprocs = 1)"""

    def __init__(self, mesh):
        self.dim = mesh.dim
        self.nv = fo.ndofs(mesh.geom, "biquadratic")
        self.npr = fo.ndofs(mesh.geom, "linear")
        nq2, nq1 = fo.n_dofs(mesh, "biquadratic"), fo.n_dofs(mesh, "linear")
        self.sizes = [nq2] * self.dim + [nq1]
        self.offset = np.concatenate([[0], np.cumsum(self.sizes)])
        self.n = int(self.offset[-1])
        self.nd = self.dim * self.nv + self.npr
        ed = mesh.elem_dof
        self.elem_sys = np.concatenate([ed[:, :self.nv] + self.offset[k] for k in range(self.dim)] +
                                       [ed[:, :self.npr] + self.offset[self.dim]], axis=1)


def elem_ns_batch(etv, etp, X, UV, Pr, nu):
    """X[nel, dim, nloc], UV[nel, dim, nv], Pr[nel, npr] -> Jac[nel, nd, nd], Res[nel, nd] (Res = -aRes);
    Gauss-point sum sequential (reference order)."""
    nel = X.shape[0]
    dim, nv, ng = etv.dim, etv.nc, etv.ng
    npr = etp.nc
    nd = dim * nv + npr
    Jm = np.einsum("gna,ebn->egab", etv.dphi, X[:, :, :nv])
    det = np.linalg.det(Jm)
    JI = np.linalg.inv(Jm)
    grad = np.einsum("gna,egba->egnb", etv.dphi, JI)          # grad[e,g,n,b] = d phi_n / d x_b
    w = det * etv.w[None, :]
    phi, psi = etv.phi, etp.phi                               # [ng, nv], [ng, npr]
    Jac = np.zeros((nel, nd, nd))
    aRes = np.zeros((nel, nd))
    for g in range(ng):
        G = grad[:, g]                                        # [nel, nv, dim]
        ug = np.einsum("ekn,n->ek", UV, phi[g])               # u_k
        gu = np.einsum("ekn,enj->ekj", UV, G)                 # d_j u_k
        pg = Pr @ psi[g]
        wg = w[:, g]
        lap = np.einsum("eid,ejd->eij", G, G)                 # grad phi_i . grad phi_j
        adv = np.einsum("ej,enj->en", ug, G)                  # u . grad phi_n
        for k in range(dim):
            rk = slice(k * nv, (k + 1) * nv)
            aRes[:, rk] += (nu * np.einsum("eid,ed->ei", G, gu[:, k]) + phi[g][None, :] * np.einsum("ej,ej->e", ug, gu[:, k])[:, None]
                            - pg[:, None] * G[:, :, k]) * wg[:, None]
            Jac[:, rk, rk] += (nu * lap + phi[g][None, :, None] * adv[:, None, :]) * wg[:, None, None]
            for m in range(dim):
                cm = slice(m * nv, (m + 1) * nv)
                Jac[:, rk, cm] += (phi[g][None, :, None] * phi[g][None, None, :]) * (gu[:, k, m] * wg)[:, None, None]
            Jac[:, rk, dim * nv:] += -(G[:, :, k][:, :, None] * psi[g][None, None, :]) * wg[:, None, None]
            Jac[:, dim * nv:, rk] += -(psi[g][None, :, None] * G[:, :, k][:, None, :]) * wg[:, None, None]
        div = sum(gu[:, k, k] for k in range(dim))
        aRes[:, dim * nv:] += -(div * wg)[:, None] * psi[g][None, :]
    return Jac, -aRes


def assemble_ns(mesh, lay, sol, nu, order="seventh", pattern=None, etp=None):
    etv = fo.ElemType(mesh.geom, "biquadratic", order)
    if etp is None:
        etp = fo.ElemType(mesh.geom, "linear", order)
    X = np.transpose(mesh.coords[mesh.elem_dof], (0, 2, 1))
    es = lay.elem_sys
    nv, dim = lay.nv, lay.dim
    loc = sol[es]
    UV = loc[:, :dim * nv].reshape(mesh.nel, dim, nv)
    Pr = loc[:, dim * nv:]
    Jac, Res = elem_ns_batch(etv, etp, X, UV, Pr, nu)
    if pattern is None:
        pattern = csr_pattern_sys(lay)
    indptr, indices = pattern
    vals = np.zeros(indices.size)
    rows = np.repeat(es, lay.nd, axis=1).ravel()
    cols = np.tile(es, (1, lay.nd)).ravel()
    pos = fo._csr_positions(indptr, indices, rows, cols)
    np.add.at(vals, pos, Jac.ravel())
    b = np.zeros(lay.n)
    np.add.at(b, es.ravel(), Res.ravel())
    return sp.csr_matrix((vals, indices, indptr), shape=(lay.n, lay.n)), b


def pressure_boundary_residual(mesh, lay, bc, order="seventh", face_tables=None):
    """Boundary integral for the "P" part, 03_navier_stokes.hpp:185-290, loop for loop: every element, every face with a boundary flag
    (GetFaceElementIndex < 0, :196-198; flags < -1 here: -1 is "no neighbour, no name" and the box / Gambit meshes of this tier have none on the
    boundary); face coordinates (Q2 face nodes) and their mean (:205-229); the bdc callback of every velocity component at that mean (:232-246);
    normal from JacobianSur at Gauss point 0 (:249-254); the last component with |n_d| >= 1e-4 is the normal velocity (:257-262); if it is not
    Dirichlet: Gauss loop with xg = sum phi_i x_i, tau = bdc("P") at xg, aResV[k][inode] += phi_i tau n_k weight (:266-290).
    bc(x, name, face_name) -> (is_dirichlet, value).  face_tables: the reference's faceDofs rows (GetLocalFaceVertexIndex, :223; golden fixture);
    default = this oracle's own outward-oriented face order.  Returns aRes scattered to the system dofs (RES receives minus this, :377-381)."""
    geom, dim = mesh.geom, mesh.dim
    names = ["U", "V", "W"][:dim]
    out = np.zeros(lay.n)
    ng = fo.gauss_table("quad" if geom == "hex" else "line", order)[0].size
    for iel in range(mesh.nel):
        for jface in range(mesh.face_flag.shape[1]):
            flag = int(mesh.face_flag[iel, jface])
            if flag >= -1:
                continue
            face_index = -(flag + 1)
            loc = fo.face_local_nodes(geom, "biquadratic", jface, None if face_tables is None else face_tables[jface])
            nodes = mesh.elem_dof[iel, loc]
            vt = [mesh.coords[nodes, d] for d in range(dim)]
            centre = np.zeros(dim)
            for d in range(dim):
                for i in range(nodes.size):
                    centre[d] += vt[d][i]
                centre[d] /= nodes.size
            is_dir = [bc(centre, names[d], face_index)[0] for d in range(dim)]
            _, _, normal = fo.jacobian_sur(geom, "biquadratic", order, vt, 0)
            comp = 0
            for d in range(dim):
                if abs(normal[d]) >= 1.0e-4:
                    comp = d
            if is_dir[comp]:
                continue
            for ig in range(ng):
                weight, phi, normal = fo.jacobian_sur(geom, "biquadratic", order, vt, ig)
                xg = np.zeros(dim)
                for i in range(nodes.size):
                    for k in range(dim):
                        xg[k] += phi[i] * vt[k][i]
                tau = bc(xg, "P", face_index)[1]
                for i in range(nodes.size):
                    for k in range(dim):
                        out[lay.offset[k] + nodes[i]] += phi[i] * tau * normal[k] * weight
    return out


def csr_pattern_sys(lay):
    es = lay.elem_sys.astype(np.int64)
    nd, n = lay.nd, lay.n
    key = np.unique((np.repeat(es, nd, axis=1) * n + np.tile(es, (1, nd))).ravel())
    rows = key // n
    indptr = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=n))])
    return indptr.astype(np.int32), (key % n).astype(np.int32)


# ---- boundary conditions of the cavity (main.cpp:365-390): every velocity component Dirichlet on every wall, the
# tangential component 1 on the moving wall (open interval: the two end nodes stay 0); pressure pinned at the (lo, lo) corner
def cavity_bc(mesh, lay, lid_flag=-5, lid_component=1, lid_value=1.0):
    fn = fo.face_nodes(mesh.geom)
    dim = mesh.dim
    lo, hi = mesh.coords.min(axis=0), mesh.coords.max(axis=0)
    bdc, val = [], []
    mark = np.zeros(mesh.nnode, dtype=bool)
    lid = np.zeros(mesh.nnode, dtype=bool)
    for f, nodes in enumerate(fn):
        els = np.where(mesh.face_flag[:, f] < -1)[0]
        mark[mesh.elem_dof[els][:, nodes].ravel()] = True
        els = np.where(mesh.face_flag[:, f] == lid_flag)[0]
        lid[mesh.elem_dof[els][:, nodes].ravel()] = True
    nodes = np.where(mark)[0]
    tdir = lid_component
    x = mesh.coords[nodes]
    inside = lid[nodes] & (x[:, tdir] > lo[tdir]) & (x[:, tdir] < hi[tdir])
    for k in range(dim):
        bdc.append(nodes + lay.offset[k])
        val.append(np.where(inside, lid_value, 0.0) if k == lid_component else np.zeros(nodes.size))
    # pressure: only the corner node (x < lo + 1e-8 in every direction), reached through the boundary faces
    nq1 = fo.n_dofs(mesh, "linear")
    pn = nodes[(nodes < nq1) & np.all(mesh.coords[nodes] < lo + 1e-8, axis=1)]
    bdc.append(pn + lay.offset[dim])
    val.append(np.zeros(pn.size))
    bdc, val = np.concatenate(bdc), np.concatenate(val)
    o = np.argsort(bdc)
    return bdc[o], val[o]


def block_prolongator(mc, mf, layc, layf, bdc_f=None, bdc_c=None):
    """blockdiag(P_Q2 x dim, P_Q1) (BuildProlongatorMatrix loops the system variables), then ZeroInterpolatorDirichletNodes"""
    P2 = fo.build_prolongator(mc, mf, "biquadratic")
    P1 = fo.build_prolongator(mc, mf, "linear")
    P = sp.block_diag([P2] * layf.dim + [P1], format="csr")
    if bdc_f is not None:
        P = fo.zero_interpolator_dirichlet(P, bdc_f, bdc_c)
    return P


def vertex_patches(mesh, lay):
    """one block per pressure dof: the pressure dof and every velocity dof of the elements sharing that vertex
    (pressure-centred Vanka patches, the GPU form of the element-block ASM of LinearEquationSolverPetscAsm.cpp)"""
    nq1 = fo.n_dofs(mesh, "linear")
    elems = [[] for _ in range(nq1)]
    for e in range(mesh.nel):
        for v in mesh.elem_dof[e, :lay.npr]:
            elems[v].append(e)
    patches = []
    for v in range(nq1):
        nodes = np.unique(mesh.elem_dof[elems[v]][:, :lay.nv])
        dofs = np.concatenate([nodes + lay.offset[k] for k in range(lay.dim)] + [[v + lay.offset[lay.dim]]])
        patches.append(dofs.astype(np.int64))
    return patches


def color_patches(patches, A):
    """greedy colouring in patch order.  Two patches conflict when one writes a dof the other reads, i.e. when a dof of one
    appears in the matrix rows of the other: patches of one colour can then be relaxed in any order (or concurrently)."""
    A = A.tocsr()
    n = A.shape[0]
    owner = [[] for _ in range(n)]
    reader = [[] for _ in range(n)]
    reads = []
    for p, d in enumerate(patches):
        rs = np.unique(np.concatenate([A.indices[A.indptr[i]:A.indptr[i + 1]] for i in d]))
        reads.append(rs)
        for i in d:
            owner[i].append(p)
        for i in rs:
            reader[i].append(p)
    color = np.full(len(patches), -1, dtype=np.int64)
    for p, d in enumerate(patches):
        used = set()
        for i in reads[p]:
            for q in owner[i]:
                if color[q] >= 0:
                    used.add(color[q])
        for i in d:
            for q in reader[i]:
                if color[q] >= 0:
                    used.add(color[q])
        c = 0
        while c in used:
            c += 1
        color[p] = c
    return color


class VankaSmoother:
    """multiplicative over colours, exact dense solve of every patch:  x_p += omega A_pp^-1 (b - A x)_p"""

    def __init__(self, A, patches, color, omega=1.0):
        self.A = A.tocsr()
        self.patches = patches
        self.color = color
        self.ncolors = int(color.max()) + 1
        self.omega = omega
        self.inv = [np.linalg.inv(self.A[d][:, d].toarray()) for d in patches]

    def sweep(self, b, x):
        for c in range(self.ncolors):
            for p in np.where(self.color == c)[0]:
                d = self.patches[p]
                r = b[d] - self.A[d] @ x
                x[d] += self.omega * (self.inv[p] @ r)
        return x


class AsmSmoother:
    """PCASM as FEMuS_ASM configures it -- PC_ASM_BASIC + PC_COMPOSITE_MULTIPLICATIVE (PetscPreconditioner.cpp:179-184), sub-solves = one application
    of ILU(0) of the block matrix in ascending dof order, zero pivot 1e-16, MAT_SHIFT_NONZERO (LinearEquationSolverPetscAsm.cpp:278-335) -- restated
    from PETSc 3.20's PCApply_ASM (parity unpinned: PETSc is not in the image): from y = 0 the blocks in INDEX order,
    y[d_i] += (L~ U~)_i^-1 (r - A y)[d_i] over the whole overlapping dof set d_i.  sweep = one Richardson(omega) iteration around it."""

    def __init__(self, A, patches, omega=1.0, pattern=None, n_exact=0):
        """n_exact: the first n_exact blocks are solved EXACTLY (MLU_PRECOND on the solid / porous blocks, which DoPartition puts first:
        `_blockTypeRange[1]`, LinearEquationSolverPetscAsm.cpp:298-307), the others by ILU(0).
        pattern = (rowptr, col) of the STORED entries of the level operator (the allocation the factorisation fills: element couplings on the
        assembled level, the symbolic triple product below it -- zeros included, which a scipy product may have dropped); None: A's own"""
        self.A = A.tocsr()
        self.patches = [np.sort(np.asarray(d)) for d in patches]
        self.omega = omega
        S = self.A
        if pattern is not None:
            rp, col = np.asarray(pattern[0], np.int64), np.asarray(pattern[1], np.int64)
            n = self.A.shape[0]
            pkey = np.repeat(np.arange(n, dtype=np.int64), np.diff(rp)) * n + col          # ascending: rows ascending, columns sorted in a row
            C = self.A.tocoo()
            keep = C.data != 0.0
            akey = C.row[keep].astype(np.int64) * n + C.col[keep]
            pos = np.searchsorted(pkey, akey)
            assert np.all(pos < pkey.size) and np.all(pkey[np.minimum(pos, pkey.size - 1)] == akey), "the operator has entries outside the given pattern"
            vals = np.zeros(pkey.size)
            vals[pos] = C.data[keep]
            S = sp.csr_matrix((vals, col, rp), shape=self.A.shape)              # explicit zeros stay (no arithmetic on S as a whole)
        self.ilu = []
        for k, d in enumerate(self.patches):
            B = S[d][:, d].tocsr()
            self.ilu.append(np.linalg.inv(B.toarray()) if k < n_exact else fo.ilu0_factor(B))

    def apply(self, r):
        y = np.zeros_like(r)
        for d, LU in zip(self.patches, self.ilu):
            t = r[d] - self.A[d] @ y
            y[d] += LU @ t if isinstance(LU, np.ndarray) else fo.ilu0_apply(LU, t)
        return y

    def sweep(self, b, x):
        return x + self.omega * self.apply(b - self.A @ x)


class NSHierarchy:
    pass


def vcycle(H, level, b, x=None):
    A = H.A[level]
    if level == 0:
        return H.coarse_solve(b)
    x = np.zeros_like(b) if x is None else x
    for _ in range(H.npre):
        x = H.smoother[level].sweep(b, x)
    r = b - A @ x
    ec = vcycle(H, level - 1, H.P[level].T @ r)
    x = x + H.P[level] @ ec
    for _ in range(H.npost):
        x = H.smoother[level].sweep(b, x)
    return x


def gmres_mg(H, top, b, rtol=1e-10, atol=1e-50, maxit=30, restart=30):
    """left-preconditioned GMRES (classical Gram-Schmidt), preconditioner = one V-cycle, zero initial guess; same
    iteration as femus_oracle.solve_gmres_mg"""
    A = H.A[top]
    M = lambda v: vcycle(H, top, v)
    x = np.zeros_like(b)
    hist = []
    it = 0
    r = M(b - A @ x)
    beta0 = np.linalg.norm(r)
    while True:
        beta = np.linalg.norm(r)
        if not hist:
            hist.append(beta)
        if beta <= max(rtol * beta0, atol) or it >= maxit:
            break
        m = restart
        V = np.zeros((m + 1, b.size))
        Hh = np.zeros((m + 1, m))
        V[0] = r / beta
        g = np.zeros(m + 1)
        g[0] = beta
        cs, sn = np.zeros(m), np.zeros(m)
        k = 0
        while k < m and it < maxit:
            w = M(A @ V[k])
            h = V[:k + 1] @ w
            w = w - h @ V[:k + 1]
            Hh[:k + 1, k] = h
            Hh[k + 1, k] = np.linalg.norm(w)
            if Hh[k + 1, k] > 0:
                V[k + 1] = w / Hh[k + 1, k]
            for i in range(k):
                t = cs[i] * Hh[i, k] + sn[i] * Hh[i + 1, k]
                Hh[i + 1, k] = -sn[i] * Hh[i, k] + cs[i] * Hh[i + 1, k]
                Hh[i, k] = t
            d = np.hypot(Hh[k, k], Hh[k + 1, k])
            cs[k], sn[k] = Hh[k, k] / d, Hh[k + 1, k] / d
            Hh[k, k] = d
            Hh[k + 1, k] = 0.0
            g[k + 1] = -sn[k] * g[k]
            g[k] = cs[k] * g[k]
            it += 1
            k += 1
            hist.append(abs(g[k]))
            if abs(g[k]) <= max(rtol * beta0, atol):
                break
        y = np.linalg.solve(np.triu(Hh[:k, :k]), g[:k])
        x = x + y @ V[:k]
        r = M(b - A @ x)
    return x, hist


def build_ns_levels(nx, ny, nz, nlevels, lo, hi):
    ms = fo.build_levels(nx, ny, nz, nlevels, lo, hi)
    lays = [NSLayout(m) for m in ms]
    return ms, lays


def newton_step_operators(ms, lays, bcs, igrid, sol, nu, omega=0.6, npre=2, npost=2, order="seventh", smoother="vanka", patterns=None, asm_exact=0):
    """assemble at level igrid, Galerkin chain, penalty rows, smoothers: everything one Newton iteration prepares"""
    H = NSHierarchy()
    A, b = assemble_ns(ms[igrid], lays[igrid], sol, nu, order)
    H.P = [None] * (igrid + 1)
    for l in range(1, igrid + 1):
        H.P[l] = block_prolongator(ms[l - 1], ms[l], lays[l - 1], lays[l], bcs[l][0], bcs[l - 1][0])
    H.A = [None] * (igrid + 1)
    H.A[igrid] = A
    for l in range(igrid, 0, -1):
        H.A[l - 1] = (H.P[l].T @ H.A[l] @ H.P[l]).tocsr()
    for l in range(igrid + 1):
        H.A[l] = fo.zero_rows(H.A[l], bcs[l][0], 1.0)
    b = b.copy()
    b[bcs[igrid][0]] = 0.0
    H.b = b
    H.npre, H.npost = npre, npost
    H.smoother = [None] * (igrid + 1)
    for l in range(1, igrid + 1):
        patches = vertex_patches(ms[l], lays[l])
        H.smoother[l] = (AsmSmoother(H.A[l], patches, omega, None if patterns is None else patterns[l], n_exact=asm_exact) if smoother == "asm" else
                         VankaSmoother(H.A[l], patches, color_patches(patches, H.A[l]), omega))
    lu = spla.splu(H.A[0].tocsc())
    H.coarse_solve = lu.solve
    return H


def solve_cavity(nx, ny, nlevels, nu, lo=(-0.5, -0.5, 0.0), hi=(0.5, 0.5, 0.0), tol=1e-10, max_newton=30, linear="direct",
                 lin_rtol=1e-10, lin_maxit=40, log=None):
    """NonLinearImplicitSystem::MGsolve, F-cycle.  linear = "direct" (sparse LU of the Jacobian) or "gmres_mg"."""
    ms, lays = build_ns_levels(nx, ny, 0, nlevels, lo, hi)
    bcs = [cavity_bc(m, l) for m, l in zip(ms, lays)]
    sols = [np.zeros(l.n) for l in lays]
    for l in range(nlevels):
        sols[l][bcs[l][0]] = bcs[l][1]                      # Initialize + boundary values
    history = []
    for igrid in range(nlevels):
        for it in range(max_newton):
            H = newton_step_operators(ms, lays, bcs, igrid, sols[igrid], nu)
            if linear == "direct":
                eps = spla.spsolve(H.A[igrid].tocsc(), H.b)
                nlin = 0
            else:
                eps, hist = gmres_mg(H, igrid, H.b, rtol=lin_rtol, maxit=lin_maxit)
                nlin = len(hist) - 1
            sols[igrid] = sols[igrid] + eps
            lay = lays[igrid]
            ratios = []
            for k in range(lay.dim + 1):
                s = slice(lay.offset[k], lay.offset[k + 1])
                ratios.append(np.linalg.norm(eps[s]) / (np.linalg.norm(sols[igrid][s]) + 1e-50))
            history.append((igrid, it, max(ratios), nlin))
            if log:
                log("level %d newton %d eps/sol %.3e linear its %d" % (igrid, it, max(ratios), nlin))
            if max(ratios) < tol:
                break
        if igrid + 1 < nlevels:
            # ProlongatorSol: Sol_f = P_mesh Sol_c per variable (no boundary re-imposition)
            layc, layf = lays[igrid], lays[igrid + 1]
            P = block_prolongator(ms[igrid], ms[igrid + 1], layc, layf)
            sols[igrid + 1] = P @ sols[igrid]
    return ms, lays, sols, history


# ================================================================================================================
# The callback the APPLICATION ships: applications/003_NavierStokes/SteadyNavierStokesParallel/main.cpp:390-925 -- equal-order LAGRANGE FIRST
# velocity and pressure (:98-108) with the Franca-Frey stabilisation (`FrancaAndFrey = !Tezduyare`, :677-868; the Tezduyar branch :585-676 is
# compiled out by `bool Tezduyare = 0`).  Restated statement by statement (vectorised over the elements, Gauss points and nodes in the reference's
# order); the Jacobian the reference takes from adept's tape (:895-910, KKloc = -d aRhs / d Soli) is taken here by COMPLEX-STEP differentiation of
# the restated residual -- exact to rounding like the tape, and checked against central differences in tests/test_ns_host.py.
#   sqrtlambdak (:735) = the per-element value SetLambda stores (:943-1262); for LAGRANGE FIRST its Gauss loop breaks after the first point
#   (:1082 `if (0 == SolType) break`): lambdak = 6 / hk^2, hk = (referenceElementScale * Weight(0) / GaussWeight(0))^(1/dim)
#   Reynolds continuation (:485-489): the callback counts its own calls, IRe = 1 / min((1 + 5 c^2)(c + 1), 10000)
# ================================================================================================================
class NSLayoutEqualOrder:
    """variables U, V (, W), P all on ONE Lagrange family (the application: linear); system dof = offset[k] + mesh dof (nprocs = 1)"""

    def __init__(self, mesh, fe="linear"):
        self.dim, self.fe = mesh.dim, fe
        self.nv = self.npr = fo.ndofs(mesh.geom, fe)
        nq = fo.n_dofs(mesh, fe)
        self.sizes = [nq] * (self.dim + 1)
        self.offset = np.concatenate([[0], np.cumsum(self.sizes)])
        self.n = int(self.offset[-1])
        self.nd = (self.dim + 1) * self.nv
        ed = mesh.elem_dof[:, :self.nv]
        self.elem_sys = np.concatenate([ed + self.offset[k] for k in range(self.dim + 1)], axis=1)


def reynolds_of_call(counter):
    """IRe the callback uses at its `counter`-th call (main.cpp:485-489)"""
    dre = 1 + (counter * counter) * 5
    return 1.0 / (dre * (counter + 1)) if dre * (counter + 1) < 10000 else 1.0 / 10000.0


def _stab_geometry(et, X):
    """weights, gradients, Hessians (ElemType.hpp:1183-1248 / :1438-1537, geometry = the FE's own nodes), sqrt(lambda_k)"""
    dim, nc = et.dim, et.nc
    Jm = np.einsum("gna,ebn->egab", et.dphi, X[:, :, :nc])
    det = np.linalg.det(Jm)
    JacI = np.linalg.inv(Jm)                                   # the reference's JacI: gradphi[n][a] = sum_c dphi[n][c] JacI[a][c]
    grad = np.einsum("gnc,egac->egna", et.dphi, JacI)          # d phi_n / d x_a
    nh = 3 if dim == 2 else 6
    pairs = [(0, 0), (1, 1), (0, 1)] if dim == 2 else [(0, 0), (1, 1), (2, 2), (0, 1), (1, 2), (2, 0)]
    h = et.d2phi                                               # [g, n, nh]
    if dim == 2:
        Href = np.stack([np.stack([h[..., 0], h[..., 2]], -1), np.stack([h[..., 2], h[..., 1]], -1)], -2)          # [g, n, r, c]
    else:
        Href = np.stack([np.stack([h[..., 0], h[..., 3], h[..., 5]], -1), np.stack([h[..., 3], h[..., 1], h[..., 4]], -1),
                         np.stack([h[..., 5], h[..., 4], h[..., 2]], -1)], -2)
    nabla = np.stack([np.einsum("gnrc,egc,egr->egn", Href, JacI[:, :, a, :], JacI[:, :, b, :]) for a, b in pairs], -1)      # [e, g, n, nh]
    w = det * et.w[None, :]
    ref_scale = {"quad": 4.0, "hex": 8.0}[et.geom]
    hk = (ref_scale * w[:, 0] / et.w[0]) ** (1.0 / dim)
    return w, grad, nabla, np.sqrt(6.0 / (hk * hk)), pairs


def _stab_residual(et, geo, U, P, IRe):
    """aRhs of main.cpp:677-868 for all elements: U[nel, dim, nv], P[nel, nv] (real or complex) -> aRhs[nel, (dim + 1) * nv]"""
    w, grad, nabla, sqrtlam, pairs = geo
    dim, nv, ng = et.dim, et.nc, et.ng
    nel = U.shape[0]
    kv = lambda i, j: i if i == j else [p for p in range(dim, len(pairs)) if set(pairs[p]) == {i, j}][0]          # :705-709 xy / xz / yz
    aR = np.zeros((nel, dim + 1, nv), dtype=U.dtype)
    for g in range(ng):
        phi = et.phi[g]
        G, N, W = grad[:, g], nabla[:, g], w[:, g]
        Sol = np.einsum("ekn,n->ek", U, phi)
        Gs = np.einsum("ekn,enj->ekj", U, G)
        Ns = np.einsum("ekn,enm->ekm", U, N)
        Sp = P @ phi
        Gp = np.einsum("en,enj->ej", P, G)
        aL2 = np.sqrt(sum(Sol[:, i] * Sol[:, i] for i in range(dim)))
        tau = 1.0 / (sqrtlam * sqrtlam * 4.0 * IRe) * np.ones(nel, dtype=U.dtype)
        delta = np.zeros(nel, dtype=U.dtype)
        Rek = aL2 / (4.0 * sqrtlam * IRe)
        on = Rek.real > 1.0e-15
        xi = np.where(Rek.real >= 1.0, 1.0, Rek)
        safe = np.where(on, aL2, 1.0)
        tau = np.where(on, xi / (safe * sqrtlam), tau)
        delta = np.where(on, (xi * aL2) / sqrtlam, delta)
        Res = np.zeros((nel, dim), dtype=U.dtype)
        for i in range(dim):
            Res[:, i] += 0.0 - Gp[:, i]
            for j in range(dim):
                Res[:, i] += -Sol[:, j] * Gs[:, i, j] + IRe * (Ns[:, i, j] + Ns[:, j, kv(i, j)])
        div = sum(Gs[:, i, i] for i in range(dim))
        for i in range(dim):
            adv = sum(Sol[:, j] * Gs[:, i, j] for j in range(dim))[:, None] * phi[None, :]
            lap = sum(IRe * G[:, :, j] * (Gs[:, i, j] + Gs[:, j, i])[:, None] for j in range(dim))
            supg = sum(Sol[:, j][:, None] * G[:, :, j] for j in range(dim)) * tau[:, None]
            for j in range(dim):
                aR[:, i] += (Res[:, i] * tau * W)[:, None] * (-IRe * N[:, :, j])              # only in least square
                aR[:, j] += (Res[:, i] * tau * W)[:, None] * (-IRe * N[:, :, kv(i, j)])
            aR[:, i] += (-adv - lap + (Sp - delta * div)[:, None] * G[:, :, i] + Res[:, i][:, None] * supg) * W[:, None]
        mg = sum(-G[:, :, i] * (Res[:, i] * tau)[:, None] for i in range(dim))
        aR[:, dim] += (div[:, None] * phi[None, :] + mg) * W[:, None]
    return aR.reshape(nel, (dim + 1) * nv)


def elem_ns_stab_batch(et, X, U, P, IRe):
    """element residual and matrix of the application's callback: Rhs = aRhs (added to RES, :884-893), KKloc = -d aRhs / d Soli (:895-910).
    X[nel, dim, >= nv] node coordinates, U[nel, dim, nv], P[nel, nv] -> KK[nel, nd, nd], Rhs[nel, nd]"""
    geo = _stab_geometry(et, X)
    dim, nv = et.dim, et.nc
    nd = (dim + 1) * nv
    nel = X.shape[0]
    Rhs = _stab_residual(et, geo, U.astype(float), P.astype(float), IRe)
    KK = np.zeros((nel, nd, nd))
    hstep = 1e-30
    for c in range(nd):
        Uc, Pc = U.astype(complex), P.astype(complex)
        if c < dim * nv:
            Uc[:, c // nv, c % nv] += 1j * hstep
        else:
            Pc[:, c - dim * nv] += 1j * hstep
        KK[:, :, c] = -_stab_residual(et, geo, Uc, Pc, IRe).imag / hstep
    return KK, Rhs


def assemble_ns_stab(mesh, lay, sol, IRe, order="seventh", pattern=None):
    et = fo.ElemType(mesh.geom, lay.fe, order)
    X = np.transpose(mesh.coords[mesh.elem_dof], (0, 2, 1))
    es = lay.elem_sys
    nv, dim = lay.nv, lay.dim
    loc = sol[es]
    KK, Rhs = elem_ns_stab_batch(et, X, loc[:, :dim * nv].reshape(mesh.nel, dim, nv), loc[:, dim * nv:], IRe)
    if pattern is None:
        pattern = csr_pattern_sys(lay)
    indptr, indices = pattern
    vals = np.zeros(indices.size)
    rows = np.repeat(es, lay.nd, axis=1).ravel()
    cols = np.tile(es, (1, lay.nd)).ravel()
    pos = fo._csr_positions(indptr, indices, rows, cols)
    np.add.at(vals, pos, KK.ravel())
    b = np.zeros(lay.n)
    np.add.at(b, es.ravel(), Rhs.ravel())
    return sp.csr_matrix((vals, indices, indptr), shape=(lay.n, lay.n)), b


# ----------------------------------------------------------------------------------------------------------------------------------
# Q2 velocity with the DISCONTINUOUS piecewise-linear pressure the reference's known-answer test uses (unittests/testNSSteadyDD/main.cpp:
# AddSolution("P", DISCONTINUOUS_POLYNOMIAL, FIRST) :97, callback AssembleMatrixResNS :396-726).  The callback's weak form is the one of
# elem_ns_batch above (nu grad u : grad phi + (u . grad u) phi - p d_k phi ; (div u) psi; full Newton Jacobian, nwtn_alg = 2, :583-600),
# only the pressure space differs: psi = 1, xi, eta (, zeta) in REFERENCE coordinates (quadpwLinear / hexpwLinear::eval_phi with
# IND = (0,0), (1,0), (0,1): Quadrilateral.cpp:188-200), evaluated at the Gauss points (GetPhi(ig), main.cpp:553), and its dofs belong to
# the element: mesh dof of local function i of element iel = i * nel + iel for one process (Mesh::GetSolutionDof, solution type 4).
# ----------------------------------------------------------------------------------------------------------------------------------
class PwLinearPressure:
    def __init__(self, geom, order="seventh"):
        w, xg = fo.gauss_table(geom, order)
        self.dim = xg.shape[1]
        self.nc = self.dim + 1
        self.ng = w.size
        self.phi = np.concatenate([np.ones((w.size, 1)), xg], axis=1)


class NSLayoutPwLinear:
    """variables U, V (, W) biquadratic and P discontinuous piecewise linear, stacked (nprocs = 1)"""

    def __init__(self, mesh):
        self.dim = mesh.dim
        self.nv = fo.ndofs(mesh.geom, "biquadratic")
        self.npr = mesh.dim + 1
        nq2 = fo.n_dofs(mesh, "biquadratic")
        self.sizes = [nq2] * self.dim + [self.npr * mesh.nel]
        self.offset = np.concatenate([[0], np.cumsum(self.sizes)])
        self.n = int(self.offset[-1])
        self.nd = self.dim * self.nv + self.npr
        ed = mesh.elem_dof
        pdof = np.arange(self.npr)[None, :] * mesh.nel + np.arange(mesh.nel)[:, None]
        self.elem_sys = np.concatenate([ed[:, :self.nv] + self.offset[k] for k in range(self.dim)] + [pdof + self.offset[self.dim]], axis=1)


# ----------------------------------------------------------------------------------------------------------------------------------
# The temperature system of the same test: AssembleMatrixResT (unittests/testNSSteadyDD/main.cpp:730-880).  T and the velocities are
# LAGRANGE SECOND; per Gauss point  F[i] += (-IPe grad phi_i . grad T - (u . grad T) phi_i) w  (:851),
# B[i, j] += (IPe grad phi_i . grad phi_j + (u . grad phi_j) phi_i) w  (:854-864); IPe = 1 / Peclet (:741).
# ----------------------------------------------------------------------------------------------------------------------------------
def elem_advdiff_batch(et, X, T, UV, ipe):
    """X[nel, dim, nv], T[nel, nv], UV[nel, dim, nv] -> B[nel, nv, nv], F[nel, nv]; Gauss-point sum sequential"""
    nel, nv = X.shape[0], et.nc
    Jm = np.einsum("gna,ebn->egab", et.dphi, X[:, :, :nv])
    det = np.linalg.det(Jm)
    JI = np.linalg.inv(Jm)
    grad = np.einsum("gna,egba->egnb", et.dphi, JI)
    w = det * et.w[None, :]
    B, F = np.zeros((nel, nv, nv)), np.zeros((nel, nv))
    for g in range(et.ng):
        G = grad[:, g]
        ug = np.einsum("ekn,n->ek", UV, et.phi[g])
        gt = np.einsum("en,enj->ej", T, G)
        lap = np.einsum("eid,ejd->eij", G, G)
        adv = np.einsum("ej,enj->en", ug, G)                       # u . grad phi_n
        B += (ipe * lap + et.phi[g][None, :, None] * adv[:, None, :]) * w[:, g][:, None, None]
        F += (-ipe * np.einsum("eid,ed->ei", G, gt) - et.phi[g][None, :] * np.einsum("ej,ej->e", ug, gt)[:, None]) * w[:, g][:, None]
    return B, F


def assemble_advdiff(mesh, sol, vel, ipe, order="seventh"):
    """KK, RES of the temperature system on a mesh: sol[nnode] (T), vel[dim * nnode (+ ...)] the stacked velocity state"""
    et = fo.ElemType(mesh.geom, "biquadratic", order)
    ed = mesh.elem_dof
    nn = mesh.nnode
    X = np.transpose(mesh.coords[ed], (0, 2, 1))
    UV = np.stack([vel[k * nn:(k + 1) * nn][ed] for k in range(mesh.dim)], axis=1)
    B, F = elem_advdiff_batch(et, X, sol[ed], UV, ipe)
    indptr, indices = fo.csr_pattern(mesh, "biquadratic")
    vals = np.zeros(indices.size)
    nv = et.nc
    rows = np.repeat(ed, nv, axis=1).ravel()
    cols = np.tile(ed, (1, nv)).ravel()
    np.add.at(vals, fo._csr_positions(indptr, indices, rows, cols), B.ravel())
    b = np.zeros(nn)
    np.add.at(b, ed.ravel(), F.ravel())
    return sp.csr_matrix((vals, indices, indptr), shape=(nn, nn)), b
