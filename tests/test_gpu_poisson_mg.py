"""GPU parity, end to end through the C-ABI: mesh -> pattern -> batched assembly -> prolongators -> Galerkin chain
(fh_mat_ptap) -> SetPenalty -> V-cycle / outer solve, against the oracle's restatement of MGsolve."""
import numpy as np
import pytest
import scipy.sparse.linalg as spla

import femus_amd
from femus_amd import capi
from femus_amd.poisson import PoissonMG
from oracle import femus_oracle as fo

pytestmark = pytest.mark.gpu
ONE = lambda xg: np.ones(xg.shape[:2])


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def test_ptap_matches_scipy(ctx):
    H = fo.build_poisson_hierarchy(2, 2, 2, 3, "biquadratic", ONE)
    A, P = ctx.matrix_scipy(H.A_raw[2]), ctx.matrix_scipy(H.P[2])
    C = capi.Mat.ptap(P, A)
    ref = (H.P[2].T @ H.A_raw[2] @ H.P[2]).tocsr()
    D = abs(C.to_scipy() - ref)
    assert D.max() <= 1e-13 * abs(ref).max()
    # numeric reuse with new values on the same patterns; deterministic
    A.set_values(2.0 * H.A_raw[2].data)
    C.ptap_numeric(P, A)
    v1 = C.values().copy()
    assert abs(C.to_scipy() - 2.0 * ref).max() <= 1e-13 * abs(ref).max()
    C.ptap_numeric(P, A)
    assert np.array_equal(v1, C.values())
    # general products: matrix_ABC (R A P with an explicit restriction) and plain SpGEMM
    R = P.get_transpose()
    C3 = capi.Mat.matrix_ABC(R, A, P)
    assert abs(C3.to_scipy() - 2.0 * ref).max() <= 1e-13 * abs(ref).max()
    # rectangular / ragged P
    rng = np.random.default_rng(2)
    import scipy.sparse as sp
    Pr = sp.random(A.m(), 37, density=0.02, random_state=3, format="csr")
    C2 = capi.Mat.ptap(ctx.matrix_scipy(Pr), A)
    ref2 = (Pr.T @ (2.0 * H.A_raw[2]) @ Pr).tocsr()
    assert abs(C2.to_scipy() - ref2).max() <= 1e-12 * max(abs(ref2).max(), 1e-300)


def test_product_patterns_built_on_the_device_equal_the_host_builder(ctx):
    """fh_mat_matmul / fh_mat_ptap: the pattern kernel (hash set + sort in LDS, one wave per row) against the host marker builder -- same
    row pointers and columns; a row with more distinct columns than the kernel holds sends the whole product to the host builder"""
    import scipy.sparse as sp
    H = fo.build_poisson_hierarchy(2, 2, 2, 3, "biquadratic", ONE)
    cases = [(H.A_raw[2], H.P[2]), (H.P[2].T.tocsr(), H.A_raw[2]), (H.A_raw[2], H.A_raw[2]),
             (sp.random(300, 4000, density=0.02, random_state=1, format="csr"), sp.random(4000, 5000, density=0.01, random_state=2, format="csr")),  # > 1024 per row
             (sp.csr_matrix((5, 7)), sp.random(7, 3, density=0.5, random_state=3, format="csr"))]
    for Ah, Bh in cases:
        Ah.sort_indices(); Bh.sort_indices()
        out = {}
        for dev in (1, 0):
            ctx.set_option("spgemm_device_symbolic", dev)
            A, B = ctx.matrix_scipy(Ah), ctx.matrix_scipy(Bh)
            C = A.matmul(B).to_scipy()
            out[dev] = (C.indptr.copy(), C.indices.copy(), C.data.copy())
        ctx.set_option("spgemm_device_symbolic", 1)
        assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
        assert np.array_equal(out[0][2], out[1][2])
        ref = (Ah @ Bh).tocsr()
        Cd = sp.csr_matrix((out[1][2], out[1][1], out[1][0]), shape=ref.shape)
        assert abs(Cd - ref).max() <= 1e-12 * max(abs(ref).max(), 1e-300) if ref.nnz else Cd.nnz == 0 or abs(Cd).max() == 0


@pytest.mark.parametrize("args,nl,fe,npre,npost", [((2, 2, 2), 3, "biquadratic", 2, 2), ((8, 8, 0), 3, "linear", 1, 1)])
@pytest.mark.parametrize("coarse", ["galerkin", "rediscretise"])
def test_mgsolve_end_to_end(ctx, args, nl, fe, npre, npost, coarse):
    pb = PoissonMG(ctx, *args, nl, fe=fe, npre=npre, npost=npost, coarse=coarse).init()
    pb.assemble()
    pb.prepare()
    H = fo.build_poisson_hierarchy(*args, nl, fe, ONE)
    # operators: prolongators bit-exact (pattern and values), Galerkin operators to rounding
    for l in range(1, nl):
        Pd = pb.P[l].to_scipy()
        assert abs(Pd - H.P[l]).max() == 0.0
    if coarse == "galerkin":
        for l in range(nl):
            assert abs(pb.A[l].to_scipy() - H.A[l]).max() <= 1e-12 * abs(H.A[l]).max()
    # one cycle on the assembled residual
    pb.zero_boundary_residuals()
    assert rel(pb.RES.to_numpy(), H.b) < 1e-13
    if coarse == "galerkin":
        pb.vcycle()
        ref = fo.vcycle(H, nl - 1, H.b, omega=2. / 3., npre=npre, npost=npost)
        assert rel(pb.EPSC.to_numpy(), ref) < 1e-11
    # full solve: parity 1e-10 relative with the direct solution of the oracle system (north_star)
    its, rn = pb.mgsolve(outer="gmres", rtol=1e-12, maxit=50)
    pb.update_sol()
    xd = spla.spsolve(H.A[-1].tocsc(), H.b)
    assert rel(pb.SOL.to_numpy(), xd) < 1e-10
    # RES was updated to the true residual (MGSolve: RES -= KK EPSC)
    assert pb.RES.l2_norm() <= 1e-10 * np.linalg.norm(H.b)
    pb.destroy()


def test_manufactured_solution_converges(ctx):
    """tutorial ex02-style analytic check: u = prod sin(pi x_d), the reference's sign convention"""
    errs = []
    for n in (2, 4):
        pb = PoissonMG(ctx, n, n, n, 2, source_kind=1, params=(-3 * np.pi ** 2, np.pi)).init()
        pb.assemble()
        pb.prepare()
        pb.mgsolve(outer="cg", rtol=1e-12)
        pb.update_sol()
        _, xy, _ = pb.meshes[-1].arrays()
        errs.append(abs(pb.SOL.to_numpy() - np.prod(np.sin(np.pi * xy), axis=1)).max())
        pb.destroy()
    assert errs[1] < errs[0] / 12.0


@pytest.mark.parametrize("tag,box,fe", [("hex_q2", (2, 2, 2), "biquadratic"), ("quad_q1", (4, 4, 0), "linear")])
def test_against_frozen_path_vectors(ctx, tag, box, fe):
    """the committed artefact tests/golden/path_small.npz (assembled system, right-hand side, dense-LU solution)"""
    import os
    R = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "path_small.npz"))
    pb = PoissonMG(ctx, *box, 2, fe=fe).init()
    pb.assemble()
    assert abs(pb.RES.to_numpy() - R[tag + "_b_before_penalty"]).max() <= 1e-13 * abs(R[tag + "_b_before_penalty"]).max()
    pb.prepare()
    A = pb.A[1].to_scipy().tocsr()
    A.sort_indices()
    assert np.array_equal(A.indptr, R[tag + "_A_indptr"]) and np.array_equal(A.indices, R[tag + "_A_indices"])
    assert abs(A.data - R[tag + "_A_data"]).max() <= 1e-12 * abs(R[tag + "_A_data"]).max()
    pb.mgsolve(outer="gmres", rtol=1e-13, maxit=60)
    pb.update_sol()
    x = R[tag + "_x_dense_lu"]
    assert np.linalg.norm(pb.SOL.to_numpy() - x) <= 1e-10 * np.linalg.norm(x)
    pb.destroy()


@pytest.mark.parametrize("mfma", [1, 0])
@pytest.mark.parametrize("box,nl", [((2, 2, 2), 3), ((3, 2, 0), 3), ((2, 1, 2), 2)])
def test_elementwise_galerkin_equals_the_sparse_triple_product(ctx, box, nl, mfma):
    """PP^T KK PP of a uniformly refined Q2 hierarchy built element by element from the fine element matrices (fh_assembler_galerkin, on the
    matrix cores or with the sparse child tables) against the sparse triple product (fh_mat_ptap) and against the oracle's chain, on curved
    elements, 3-D and 2-D, with Dirichlet rows / columns of the interpolation zeroed: every level operator to 1e-12"""
    from femus_amd import capi
    rng = np.random.default_rng(9)
    ctx.set_option("galerkin_mfma", mfma)
    try:
        ops = []
        for elementwise in (True, False):
            m0 = capi.Mesh.box(*box)
            meshes = [m0]
            for _ in range(1, nl):
                meshes.append(meshes[-1].refine())
            if elementwise:                                   # the same perturbation for both builds
                _, xy, ff = meshes[-1].arrays()
                ed = meshes[-1].arrays()[0]
                onb = np.zeros(meshes[-1].nnode, dtype=bool)
                for f in range(2 * m0.dim):
                    loc = capi.fe_face_nodes(m0.geom, "biquadratic", f)
                    onb[ed[np.where(ff[:, f] < -1)[0]][:, loc].ravel()] = True
                xy_pert = xy + np.where(onb[:, None], 0.0, rng.uniform(-0.01, 0.01, xy.shape))
            meshes[-1].set_coords(xy_pert)
            pb = PoissonMG(ctx, 0, 0, 0, nl, meshes=meshes, elementwise_galerkin=elementwise).init()
            assert pb.gal_elem == elementwise
            pb.assemble()
            pb.level_operators()
            ops.append([pb.A[l].to_scipy() for l in range(nl)])
            pb.destroy()
        for l in range(nl):
            ref = ops[1][l]
            assert abs(ops[0][l] - ref).max() <= 1e-12 * abs(ref).max(), l
    finally:
        ctx.set_option("galerkin_mfma", 1)
