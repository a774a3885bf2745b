"""Oracle checks that do not need the reference: analytic properties of the restated path (CPU)."""
import numpy as np
import pytest
import scipy.sparse.linalg as spla

from oracle import femus_oracle as fo

ONE = lambda xg: np.ones(xg.shape[:2])


def test_partition_of_unity_and_gradient_sum():
    for geom in ("quad", "hex"):
        for fe in ("linear", "biquadratic"):
            et = fo.ElemType(geom, fe, "seventh")
            assert np.allclose(et.phi.sum(1), 1.0, atol=1e-14)
            assert np.allclose(et.dphi.sum(1), 0.0, atol=1e-13)


def test_jacobian_unit_cube_and_survey_probe_value():
    et = fo.ElemType("hex", "biquadratic", "seventh")
    h = 1.0 / 64
    x = [(fo.XC_HEX27[:, d] + 1) / 2 * h for d in range(3)]
    vol = sum(et.jacobian(x, g)[0] for g in range(et.ng))
    # sum_g weight = 3.8146972656249814e-06 for h = 1/64 (14-digit Gauss literals)
    assert vol == 3.8146972656249814e-06
    K, F = fo.elem_poisson(et, x, np.zeros(27), lambda p: 1.0)
    assert abs(K - K.T).max() < 1e-17
    assert abs(K.sum(1)).max() < 1e-15
    assert abs(K[0, 0] - 1.9444444444444e-3) < 1e-15


def test_batch_equals_loop_on_distorted_element():
    et = fo.ElemType("hex", "biquadratic", "seventh")
    rng = np.random.default_rng(3)
    x = [fo.XC_HEX27[:, d] * 0.5 + rng.uniform(-0.04, 0.04, 27) for d in range(3)]
    u = rng.uniform(-1, 1, 27)
    K, F = fo.elem_poisson(et, x, u, lambda p: np.sin(p[0]) + p[1])
    Kb, Fb = fo.elem_poisson_batch(et, np.array(x)[None], u[None], lambda xg: np.sin(xg[..., 0]) + xg[..., 1])
    assert abs(Kb[0] - K).max() <= 1e-14 * abs(K).max()
    assert abs(Fb[0] - F).max() <= 1e-13 * abs(F).max()


def test_mesh_counts_and_geometry():
    ms = fo.build_levels(2, 2, 2, 3)
    assert [m.nel for m in ms] == [8, 64, 512]
    assert [m.nnode for m in ms] == [125, 729, 4913]
    # vertices first, then edges, then faces/centres (Mesh.cpp:517-559)
    assert ms[0].own_size == [27, 27 + 54, 125]
    m = ms[-1]
    c = m.coords[m.elem_dof]
    assert abs(c[:, 26] - c[:, :8].mean(1)).max() == 0.0
    assert np.unique(np.round(m.coords * 2 ** 20).astype(np.int64), axis=0).shape[0] == m.nnode
    # first-touch numbering: element 0 holds nodes 0..7 as its vertices
    assert m.elem_dof[0, :8].tolist() == list(range(8))
    # the 27 nodes of a coarse element are vertices on the fine level and keep their ids (Appendix B.7)
    mc, mf = ms[0], ms[1]
    f2c = fo.fine2coarse_vertex_mapping("hex")
    for j in range(8):
        assert np.array_equal(mf.coords[mf.elem_dof[j, :8]], mc.coords[mc.elem_dof[0, f2c[j]]])


def test_2d_config1_sizes():
    ms = fo.build_levels(8, 8, 0, 3)
    assert [m.nel for m in ms] == [64, 256, 1024]
    assert [fo.n_dofs(m, "linear") for m in ms] == [81, 289, 1089]     # BASELINE config 0: 1089 DOFs
    assert len(fo.dirichlet_dofs(ms[-1], "linear")) == 4 * 32


def test_3d_sizes_match_baseline_table():
    ms = fo.build_levels(2, 2, 2, 2)
    rp, col = fo.csr_pattern(ms[-1], "biquadratic")
    n = 4
    assert ms[-1].nnode == (2 * n + 1) ** 3 and rp[-1] == (8 * n + 1) ** 3
    P = fo.build_prolongator(ms[0], ms[1], "biquadratic")
    assert P.nnz == (8 * 2 + 1) ** 3
    assert np.allclose(np.asarray(P.sum(1)).ravel(), 1.0)                      # interpolation of constants


def test_galerkin_equals_rediscretised_on_free_dofs():
    H = fo.build_poisson_hierarchy(2, 2, 2, 2, "biquadratic", ONE)
    A0, _ = fo.assemble_poisson(H.meshes[0], "biquadratic", ONE)
    free = np.setdiff1d(np.arange(A0.shape[0]), H.bdc[0])
    D = (A0 - H.A[0])[free][:, free]
    assert abs(D).max() <= 1e-13 * abs(A0).max()


@pytest.mark.parametrize("solver", ["richardson", "pcg", "gmres"])
def test_mg_solvers_reach_direct_solution(solver):
    H = fo.build_poisson_hierarchy(2, 2, 2, 3, "biquadratic", ONE)
    xd = spla.spsolve(H.A[-1].tocsc(), H.b)
    fn = {"richardson": fo.solve_richardson_mg, "pcg": fo.solve_pcg_mg, "gmres": fo.solve_gmres_mg}[solver]
    x, hist = fn(H, rtol=1e-12)
    assert np.linalg.norm(x - xd) <= 1e-10 * np.linalg.norm(xd)
    assert len(hist) < 25


@pytest.mark.parametrize("smoother", ["jacobi", "ilu0"])
def test_flexible_gmres_is_what_gmres_smoothed_cycles_need(smoother):
    """with GMRES level solvers the cycle changes from application to application: the flexible outer solver converges to the
    direct solution, the non-flexible one stalls some digits short (the reason FH_OUTER_FGMRES exists); with a fixed cycle both agree"""
    H = fo.build_poisson_hierarchy(2, 2, 2, 3, "biquadratic", ONE)
    xd = spla.spsolve(H.A[-1].tocsc(), H.b)
    kw = dict(omega=1.0, npre=2, npost=2, smoother=smoother, level_solver="gmres")
    xf, hf = fo.solve_fgmres_mg(H, rtol=1e-12, maxit=40, **kw)
    xg, _ = fo.solve_gmres_mg(H, rtol=1e-12, maxit=40, **kw)
    assert np.linalg.norm(xf - xd) <= 1e-11 * np.linalg.norm(xd)
    assert np.linalg.norm(H.b - H.A[-1] @ xf) <= 2e-12 * np.linalg.norm(H.b)
    assert np.linalg.norm(xg - xd) >= 1e-8 * np.linalg.norm(xd)
    x1, h1 = fo.solve_fgmres_mg(H, rtol=1e-12)
    x2, h2 = fo.solve_gmres_mg(H, rtol=1e-12)
    assert np.linalg.norm(x1 - x2) <= 1e-10 * np.linalg.norm(xd) and abs(len(h1) - len(h2)) <= 2


def test_manufactured_solution_fourth_order():
    f = lambda xg: -3 * np.pi ** 2 * np.prod(np.sin(np.pi * xg), axis=-1)
    errs = []
    for n in (2, 4):
        H = fo.build_poisson_hierarchy(n, n, n, 2, "biquadratic", f)
        x, _ = fo.solve_pcg_mg(H, rtol=1e-12)
        errs.append(abs(x - np.prod(np.sin(np.pi * H.meshes[-1].coords), axis=1)).max())
    assert errs[1] < errs[0] / 12.0


def test_config1_2d_q1_three_level_cycle():
    H = fo.build_poisson_hierarchy(8, 8, 0, 3, "linear", ONE)
    xd = spla.spsolve(H.A[-1].tocsc(), H.b)
    x, hist = fo.solve_gmres_mg(H, rtol=1e-12, npre=1, npost=1)
    assert np.linalg.norm(x - xd) <= 1e-10 * np.linalg.norm(xd)


def test_c_restatement_matches_numpy_oracle():
    """oracle/oracle_kernels.c (the timed CPU baseline) against the numpy oracle: element kernel, scatter, cycle"""
    from oracle import c_kernels as ck
    H = fo.build_poisson_hierarchy(2, 2, 2, 3, "biquadratic", ONE)
    m = H.meshes[-1]
    u = fo.lcg_fill(m.nnode, 4)
    K, F = ck.assemble_poisson(m.elem_dof, m.coords, "biquadratic", "hex", 0, m.nel, sol=u, source_kind=1, p0=2.0, p1=1.1)
    et = fo.ElemType("hex", "biquadratic", "seventh")
    X = np.transpose(m.coords[m.elem_dof], (0, 2, 1))
    Ko, Fo = fo.elem_poisson_batch(et, X, u[m.elem_dof], lambda xg: 2.0 * np.prod(np.sin(1.1 * xg), axis=-1))
    assert abs(K - Ko).max() <= 1e-13 * abs(Ko).max() and abs(F - Fo).max() <= 1e-12 * abs(Fo).max()
    rp, col = fo.csr_pattern(m, "biquadratic")
    val, res = np.zeros(rp[-1]), np.zeros(m.nnode)
    ck.assemble_poisson(m.elem_dof, m.coords, "biquadratic", "hex", 0, m.nel, csr=(rp, col, val, res))
    assert abs(val - H.A_raw[-1].data).max() <= 1e-13 * abs(val).max()
    assert abs(res - H.b_raw).max() <= 1e-13 * abs(res).max()
    cyc = ck.CVcycle(H.A, H.P)
    rhs = fo.lcg_fill(m.nnode, 8)
    ref = fo.vcycle(H, 2, rhs)
    assert np.linalg.norm(cyc.apply(rhs) - ref) <= 1e-12 * np.linalg.norm(ref)


def test_gmres_level_smoother_minimises_the_preconditioned_residual():
    """oracle `smooth_gmres` (KSPGMRES as the level solver): after m iterations ||B (b - A x)|| is the minimum over x0 + K_m(BA, B r0),
    checked against a dense least-squares solve over an explicitly built Krylov basis"""
    import numpy as np
    from oracle import femus_oracle as fo
    H = fo.build_poisson_hierarchy(2, 2, 0, 2, "biquadratic", lambda xg: np.ones(xg.shape[:2]))
    A = H.A[-1].tocsr()
    dinv = fo.jacobi_dinv(A)
    B = lambda r: dinv * r
    rng = np.random.default_rng(3)
    b, x0 = rng.uniform(-1, 1, A.shape[0]), rng.uniform(-1, 1, A.shape[0])
    for m in (1, 2, 4):
        x = fo.smooth_gmres(A, b, x0.copy(), m, False, B)
        r0 = B(b - A @ x0)
        K = [r0]
        for _ in range(m - 1):
            K.append(B(A @ K[-1]))
        K = np.array(K).T
        BAK = np.array([B(A @ K[:, j]) for j in range(m)]).T
        c = np.linalg.lstsq(BAK, r0, rcond=None)[0]
        best = np.linalg.norm(r0 - BAK @ c)
        got = np.linalg.norm(B(b - A @ x))
        assert abs(got - best) <= 1e-10 * np.linalg.norm(r0)
    # zero initial guess: the same with x0 = 0
    x = fo.smooth_gmres(A, b, None, 3, True, B)
    r0 = B(b)
    K = np.array([r0, B(A @ r0), B(A @ B(A @ r0))]).T
    BAK = np.array([B(A @ K[:, j]) for j in range(3)]).T
    c = np.linalg.lstsq(BAK, r0, rcond=None)[0]
    assert abs(np.linalg.norm(B(b - A @ x)) - np.linalg.norm(r0 - BAK @ c)) <= 1e-10 * np.linalg.norm(r0)
