"""GPU parity at BASELINE.json's full size (configs[1]: 3-D Poisson Q2, 64^3, 4 levels) through size-independent properties
(identities the discretisation must satisfy), through samples against the single-threaded C restatement, and -- with the C restatement's
element loop and cycle run on all host cores -- entry by entry: every value of the fine-level operator and one whole V(2,2) cycle."""
import numpy as np
import pytest

import femus_amd
from femus_amd import capi
from oracle import femus_oracle as fo
from femus_amd.poisson import PoissonMG

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def problem(ctx):
    pb = PoissonMG(ctx, 8, 8, 8, 4).init()
    pb.assemble()
    yield pb
    pb.destroy()


def test_full_size_counts(problem):
    pb = problem
    assert [m.nel for m in pb.meshes] == [512, 4096, 32768, 262144]
    assert pb.ndof == [4913, 35937, 274625, 2146689]
    assert pb.A[-1].nnz == 135005697 and [pb.P[l].nnz for l in (1, 2, 3)] == [274625, 2146689, 16974593]
    assert pb.bdc[-1].size == 129 ** 3 - 127 ** 3                   # 98 306 Dirichlet rows


def test_raw_operator_identities(ctx, problem):
    pb = problem
    n = pb.ndof[-1]
    A = pb.A[-1]
    ones, y = ctx.vector(n), ctx.vector(n)
    ones.fill(1.0)
    # constants are in the kernel of the un-penalised stiffness matrix: row sums vanish
    y.matrix_mult(ones, A)
    assert y.linfty_norm() <= 1e-13 * A.linfty_norm()
    # symmetry through two random vectors: x^T A z == z^T A x
    rng = np.random.default_rng(1)
    x, z, Ax, Az = ctx.vector_from(rng.uniform(-1, 1, n)), ctx.vector_from(rng.uniform(-1, 1, n)), ctx.vector(n), ctx.vector(n)
    Ax.matrix_mult(x, A)
    Az.matrix_mult(z, A)
    a, b = z.dot(Ax), x.dot(Az)
    assert abs(a - b) <= 1e-12 * max(abs(a), abs(b), 1e-300)
    # positive semi-definite energy and the exact integral of the source: sum(RES) = -f * volume = -1
    assert x.dot(Ax) > 0
    assert abs(pb.RES.sum() + 1.0) <= 1e-12
    # linearity of the fused kernels: r = b - A x for x = 0 returns b bit for bit
    zero, r = ctx.vector(n), ctx.vector(n)
    r.resid(pb.RES, zero, A)
    assert np.array_equal(r.to_numpy(), pb.RES.to_numpy())
    # the diagonal is positive (every node touches at least one element)
    d = ctx.vector(n)
    A.get_diagonal(d)
    assert d.min() > 0.0


def test_galerkin_chain_and_full_solve(ctx, problem):
    pb = problem
    pb.prepare()
    # coarse Galerkin operators keep the constant in their kernel away from the boundary: P 1_c = 1_f on free rows
    n3, n2 = pb.ndof[3], pb.ndof[2]
    one_c, Pone = ctx.vector(n2), ctx.vector(n3)
    one_c.fill(1.0)
    Pone.matrix_mult(one_c, pb.P[3])
    v = Pone.to_numpy()
    free = np.ones(n3, dtype=bool)
    free[pb.bdc[3]] = False
    coords = pb.meshes[3].arrays()[1]
    interior = np.all((coords > 4.0 / 64) & (coords < 1 - 4.0 / 64), axis=1)
    assert np.allclose(v[interior], 1.0, atol=1e-14)              # interpolation reproduces constants
    # full MG solve to the north-star tolerance; residual really drops by 1e-10
    b0 = None
    pb.zero_boundary_residuals()
    b0 = pb.RES.l2_norm()
    its, rn = pb.mgsolve(outer="gmres", rtol=1e-13, maxit=40)     # KSP tolerance is on the PRECONDITIONED residual
    assert its <= 20
    assert pb.RES.l2_norm() <= 1e-10 * b0
    pb.update_sol()
    u = pb.SOL.to_numpy()
    # Poisson with f = 1, u = 0 on the cube boundary (reference sign convention gives u <= 0): known centre value
    centre = np.argmin(np.abs(coords - 0.5).sum(axis=1))
    assert abs(u[centre] + 0.05621) < 2e-4                        # -max u of -Lap u = 1 on the unit cube is 0.0562128
    assert u.max() <= 1e-12 and abs(u[pb.bdc[3]]).max() == 0.0


# ---- value parity AT FULL SIZE on samples (the identities above cannot see a wrong entry that keeps the symmetries) ----------
# The C restatement (oracle/oracle_kernels.c: the loop of 00_poisson_eqn_..._separate.hpp:111-215 over ElemType.hpp:1438-1537)
# does a few thousand elements per second; the production launch (persistent matrix-core workgroups, 262 144 elements, nonzero
# solution, non-constant source, curved geometry) is checked on 4096 sampled element matrices and 10 000 sampled CSR rows.
def test_device_setup_builders_at_full_size_equal_the_host_builders(ctx, problem):
    """BASELINE size (64^3 Q2): the prolongator of the finest level, the finite-element pattern of the finest matrix and the pattern of a sparse
    product built on the device are identical -- row pointers, columns, values -- to what the host loops produce (options device_setup /
    spmv patterns through fh_pattern_from_elements / spgemm_device_symbolic)"""
    pb = problem
    mc, mf = pb.meshes[-2], pb.meshes[-1]
    # the levels themselves: PoissonMG refines on the device (fh_mesh_refine_device); the host loops give the same 262 144 x 27 ids, boundary
    # flags and 2 146 689 coordinates, bit for bit
    mh = capi.Mesh.box(8, 8, 8)
    for l in range(1, len(pb.meshes)):
        nxt = mh.refine()
        mh.destroy()
        mh = nxt
    for a, b in zip(mh.arrays(), mf.arrays()):
        assert np.array_equal(a.view(np.int64) if a.dtype == np.float64 else a, b.view(np.int64) if b.dtype == np.float64 else b)
    assert mh.own_size == mf.own_size
    mh.destroy()
    out = []
    for dev in (1, 0):
        ctx.set_option("device_setup", dev)
        P = capi.build_prolongator(ctx, mc, mf, "biquadratic", zero_bdc=True)
        rp, col = P.pattern()
        out.append((rp.copy(), col.copy(), P.values().copy()))
        P.destroy()
    ctx.set_option("device_setup", 1)
    for a, b in zip(*out):
        assert np.array_equal(a, b)
    ed, _, _ = mf.arrays()
    rp_h, col_h = capi.pattern_from_elements(ed, mf.nnode)            # host builder
    rp_d, col_d = pb.A[-1].pattern()                                  # device builder (matrix_from_elements in PoissonMG.init)
    assert np.array_equal(rp_h, rp_d) and np.array_equal(col_h, col_d)
    # pattern of A * P on the coarser level pair (17 M x 2.1 M non-zeros)
    A2, P2 = pb.A[-2], pb.P[-2]
    pats = []
    for dev in (1, 0):
        ctx.set_option("spgemm_device_symbolic", dev)
        C = A2.matmul(P2)
        rp, col = C.pattern()
        pats.append((rp.copy(), col.copy()))
        C.destroy()
    ctx.set_option("spgemm_device_symbolic", 1)
    assert np.array_equal(pats[0][0], pats[1][0]) and np.array_equal(pats[0][1], pats[1][1])


@pytest.fixture(scope="module")
def curved_problem(ctx):
    """64^3 Q2 with a smooth non-affine map of the unit cube (every element curved), a nonzero solution and a sine source"""
    pb = PoissonMG(ctx, 8, 8, 8, 4, source_kind=1, params=(3.0, 2.0)).init()
    m = pb.meshes[-1]
    ed, xy, _ = m.arrays()
    xw = xy + 0.02 * np.sin(2 * np.pi * xy[:, [1, 2, 0]]) * np.sin(np.pi * xy) * np.array([1.0, -0.7, 0.5])
    m.set_coords(xw)
    pb.asm[-1].destroy()
    from femus_amd import capi
    pb.asm[-1] = capi.Assembler(ctx, m, pb.fe, pb.A[-1], pb.order, elem_dof=ed, coords=xw)
    rng = np.random.default_rng(2026)
    pb.SOL.upload(rng.uniform(-1, 1, pb.ndof[-1]))
    yield pb, ed, xw
    pb.destroy()


def test_sampled_element_matrices_at_full_size_match_the_c_oracle(curved_problem):
    from oracle import c_kernels as ck
    pb, ed, xw = curved_problem
    sol = pb.SOL.to_numpy()
    K, F = pb.asm[-1].element_matrices(pb.SOL, 1, (3.0, 2.0))            # all 262 144 elements through the production kernel
    rng = np.random.default_rng(5)
    starts = np.unique(np.concatenate([[0, ed.shape[0] - 256], rng.integers(0, ed.shape[0] - 256, 14)]))
    checked = 0
    for s in starts:                                                     # 16 windows of 256 consecutive elements
        Ko, Fo = ck.assemble_poisson(ed, xw, "biquadratic", "hex", int(s), int(s) + 256, sol=sol, source_kind=1, p0=3.0, p1=2.0)
        Kg, Fg = K[s:s + 256], F[s:s + 256]
        kscale = np.abs(Ko).max(axis=(1, 2), keepdims=True)
        assert np.max(np.abs(Kg - Ko) / kscale) <= 1e-12
        assert np.max(np.abs(Fg - Fo)) <= 1e-12 * np.abs(Fo).max()
        assert np.array_equal(Kg, np.transpose(Kg, (0, 2, 1)))            # the kernel mirrors the upper tiles: exactly symmetric
        checked += 256
    assert checked >= 4096 - 256


PATHS = {0: "two-pass", 2: "fused"}       # option assemble_fused -> what fh_assembler_last_path must report


def assemble_on_path(ctx, pb, fused):
    """one assembly of the finest level on the requested path; the path that really ran is asserted"""
    ctx.set_option("assemble_fused", fused)
    try:
        pb.assemble()
    finally:
        ctx.set_option("assemble_fused", 1)
    assert pb.asm[-1].last_path() == PATHS[fused]


@pytest.mark.parametrize("fused", [0, 2])
def test_sampled_csr_rows_at_full_size_match_the_c_oracle(ctx, curved_problem, fused):
    """10 000 random rows of the assembled operator and residual, once per assembly path (two-pass: element rows + row gather, the path of
    every assembly that follows an element-wise Galerkin product; fused: cluster kernel + partial-row pass): every row equals the sum of its
    element contributions from the C oracle, added in ascending element order as the reference's loop does"""
    from oracle import c_kernels as ck
    pb, ed, xw = curved_problem
    sol = pb.SOL.to_numpy()
    assemble_on_path(ctx, pb, fused)
    rp, col = pb.A[-1].pattern()
    val = pb.A[-1].values()
    res = pb.RES.to_numpy()
    n = pb.ndof[-1]
    rng = np.random.default_rng(11)
    rows = np.unique(np.concatenate([[0, n - 1], rng.integers(0, n, 10000)]))
    # node -> (element, local index) adjacency for the sampled rows
    flat = ed.ravel()
    order = np.argsort(flat, kind="stable")
    ptr = np.searchsorted(flat[order], np.arange(n + 1))
    need = np.unique(np.concatenate([order[ptr[r]:ptr[r + 1]] // 27 for r in rows]))
    Ko, Fo = ck.assemble_poisson(ed[need], xw, "biquadratic", "hex", 0, need.size, sol=sol, source_kind=1, p0=3.0, p1=2.0)
    pos = {int(e): k for k, e in enumerate(need)}
    worst_a = worst_b = 0.0
    for r in rows:
        acc, b, babs = {}, 0.0, 0.0
        for q in order[ptr[r]:ptr[r + 1]]:                                # ascending (element, local) order: stable sort of node ids
            e, i = int(q) // 27, int(q) % 27
            k = pos[e]
            b += Fo[k, i]
            babs += abs(Fo[k, i]) + np.abs(Ko[k, i]) @ np.abs(sol[ed[e]])     # scale of the terms summed into this residual entry
            for j in range(27):
                c = int(ed[e, j])
                acc[c] = acc.get(c, 0.0) + Ko[k, i, j]
        cols = col[rp[r]:rp[r + 1]]
        assert sorted(acc) == cols.tolist()
        want = np.array([acc[int(c)] for c in cols])
        worst_a = max(worst_a, np.abs(val[rp[r]:rp[r + 1]] - want).max() / np.abs(want).max())
        worst_b = max(worst_b, abs(res[r] - b) / babs)
    assert worst_a <= 1e-12 and worst_b <= 1e-12, (worst_a, worst_b)


def test_whole_level_matrix_and_cycle_match_the_c_oracle(ctx, curved_problem):
    """not sampled: ALL 135 005 697 entries of the fine-level operator and all 2 146 689 residual entries (curved elements, nonzero
    solution, sine source) against the C restatement's element loop run over every element on all host cores -- once per assembly path
    (two-pass and fused, the path asserted); then one V(2,2) cycle of the whole four-level hierarchy on the device against the C
    restatement's cycle on the same operators (1e-10, north_star)"""
    import os
    import scipy.linalg as sla
    from oracle import c_kernels as ck
    pb, ed, xw = curved_problem
    ck.set_threads(len(os.sched_getaffinity(0)))
    sol = pb.SOL.to_numpy()
    rp, col = pb.A[-1].pattern()
    val, res = np.zeros(rp[-1]), np.zeros(pb.ndof[-1])
    ck.assemble_poisson_all_cores(ed, xw, "biquadratic", "hex", (rp, col, val, res), sol=sol, source_kind=1, p0=3.0, p1=2.0)
    # row-wise scale: the entries of a row are sums of up to 8 element contributions of the size of the diagonal
    diag_scale = np.repeat(np.maximum.reduceat(np.abs(val), rp[:-1]), np.diff(rp))
    for fused in (2, 0):                                   # the two-pass operator last: it is the one the cycle below is prepared from
        pb.A[-1].zero(), pb.RES.zero()
        assemble_on_path(ctx, pb, fused)
        got = pb.A[-1].values()
        assert np.max(np.abs(got - val) / diag_scale) <= 1e-12, PATHS[fused]
        r_dev = pb.RES.to_numpy()
        assert np.max(np.abs(r_dev - res)) <= 1e-12 * np.abs(res).max(), PATHS[fused]
    del got, val, diag_scale
    # the cycle: operators of the device hierarchy (Galerkin chain + SetPenalty), C cycle on the host
    pb.prepare()
    A = [a.to_scipy() for a in pb.A]
    P = [None] + [q.to_scipy() for q in pb.P[1:]]
    lu = sla.lu_factor(A[0].toarray())
    cyc = ck.CVcycle(A, P, 2. / 3., 2, 2, coarse_solve=lambda b: sla.lu_solve(lu, b))
    # the Galerkin chain at full size without forming the products on the host: A_{l-1} x = P^T (A_l (P x)) on the free coarse dofs for
    # random x (the interpolation has zero rows / columns at Dirichlet nodes, SetPenalty then puts 1 on those diagonals), C SpMV throughout
    rng = np.random.default_rng(8)
    for l in (3, 2, 1):
        nc_, nf_ = pb.ndof[l - 1], pb.ndof[l]
        for rep in range(2):
            xc = rng.uniform(-1, 1, nc_)
            xc[pb.bdc[l - 1]] = 0.0
            t1, t2, lhs, rhs_ = np.zeros(nf_), np.zeros(nf_), np.zeros(nc_), np.zeros(nc_)
            ck.spmv(cyc.P[l], xc, t1)
            ck.spmv(cyc.A[l], t1, t2)
            ck.spmv(cyc.R[l], t2, rhs_)
            ck.spmv(cyc.A[l - 1], xc, lhs)
            free = np.ones(nc_, dtype=bool)
            free[pb.bdc[l - 1]] = False
            assert np.max(np.abs(lhs - rhs_)[free]) <= 1e-12 * np.max(np.abs(rhs_)), l
    rhs = fo.lcg_fill(pb.ndof[-1], 12345)
    rhs[pb.bdc[-1]] = 0.0
    want = cyc.apply(rhs)
    b, x = ctx.vector_from(rhs), ctx.vector(pb.ndof[-1])
    pb.vcycle(b, x)
    assert np.linalg.norm(x.to_numpy() - want) <= 1e-10 * np.linalg.norm(want)


def test_one_level_beyond_the_bench_size():
    """128^3 elements on one GPU (16 974 593 dofs, 1.08e9 non-zeros: row pointers above 2^30, a 14.5 GB element-row buffer): the size
    where 32-bit offset arithmetic runs out -- a binary-search midpoint `(lo + hi) / 2` did, in the row-map kernel.  Run as a child
    process with a time limit; the script checks the row sums of the assembled operator (<= 1e-14), the true residual of the GMRES
    solve and the peak of the solution (0.056212, the same as on the 64^3 mesh)."""
    import json
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run("ulimit -c 0; exec %s %s 5" % (sys.executable, os.path.join(here, "perf_probe_bigsize.py")), shell=True, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["dofs"] == 16974593 and out["nnz"] == 1076890625
    assert abs(out["max_u"] - 0.056212) < 2e-5 and out["true_relres"] < 1e-6 and out["gmres_its"] <= 8
