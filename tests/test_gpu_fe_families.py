"""GPU parity of the serendipity family (QuadQuadratic / HexQuadratic, nc = 8 / 20) and of the piecewise-constant family (quad0 / hex0) along the path:
pattern, element matrices and global assembly on curved meshes (<= 1e-12 vs the oracle's element loop), prolongators (host loops, device builder and the
oracle: identical), Neumann faces (QUAD8 / EDGE3), the multigrid solve, and the application's shipped input `input3D_Hex_serendipity.json` with its Gambit
mesh against the oracle's direct solve of the same discrete problem (1e-10).  Reference: Mesh.cpp:1021-1074, Hexahedron.cpp:167-256,
Quadrilateral.cpp:113-161, ElemType.cpp:439-532, 00_poisson_eqn_..._separate.hpp:111-215."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

import femus_amd
from femus_amd import capi
from femus_amd import app_poisson as app
from femus_amd.poisson import PoissonMG
from oracle import femus_oracle as fo

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def levels(args, nl):
    ms = [capi.Mesh.box(*args)]
    for _ in range(nl - 1):
        ms.append(ms[-1].refine())
    return ms


def curved(m, mo, seed, amp):
    """the same smooth-ish perturbation of every node on both sides (the geometry of a family is carried by its own nodes: sub-parametric map)"""
    rng = np.random.default_rng(seed)
    d = rng.uniform(-amp, amp, mo.coords.shape)
    mo.coords = mo.coords + d
    return mo.coords.copy()


@pytest.mark.parametrize("args,nl", [((2, 2, 2), 2), ((4, 3, 0), 2), ((3, 2, 1), 1)])
def test_serendipity_assembly_matches_oracle_on_curved_elements(ctx, args, nl):
    m = levels(args, nl)[-1]
    mo = fo.build_levels(*args, nl)[-1]
    ed, xy, _ = m.arrays()
    assert np.array_equal(ed, mo.elem_dof) and np.array_equal(xy, mo.coords)          # integer / coordinate parity first
    xy = curved(m, mo, 4, 0.02 / 2 ** (nl - 1))
    fe = "serendipity"
    n, nc = m.n_dofs(fe), fo.ndofs(m.geom, fe)
    assert n == fo.n_dofs(mo, fe) and nc == (20 if m.dim == 3 else 8)
    # pattern: GetSparsityPatternSize over the family's element dofs -- the device builder from the mesh, the host builder from the table, the oracle
    rp, col = capi.pattern_from_elements(ed[:, :nc], n)
    rpo, colo = fo.csr_pattern(mo, fe)
    assert np.array_equal(rp, rpo) and np.array_equal(col, colo)
    Am = ctx.matrix_from_mesh(m, fe)
    S = Am.to_scipy()
    assert np.array_equal(S.indptr, rpo) and np.array_equal(S.indices, colo)
    Am.destroy()
    A = ctx.matrix_csr(n, n, rp, col, np.full(col.size, np.nan))
    res = ctx.vector_from(np.full(n, np.nan))
    asm = capi.Assembler(ctx, m, fe, A, elem_dof=ed, coords=xy)
    assert not asm.fused_info()["active"]
    u = fo.lcg_fill(n, 31)
    sol = ctx.vector_from(u)
    src = lambda xg: 2.0 * np.prod(np.cos(1.3 * xg), axis=-1)
    # element matrices
    K, F = asm.element_matrices(sol, 2, (2.0, 1.3))
    et = fo.ElemType(mo.geom, fe, "seventh")
    X = np.transpose(mo.coords[mo.elem_dof], (0, 2, 1))
    Ko, Fo = fo.elem_poisson_batch(et, X, u[fo.elem_sys_dof(mo, fe)], src)
    assert abs(K - Ko).max() <= 1e-12 * abs(Ko).max() and abs(F - Fo).max() <= 1e-12 * abs(Fo).max()
    # one element against the reference-order loops (pure python): the batch is a vectorised form of it
    k0, f0 = fo.elem_poisson(et, X[0], u[fo.elem_sys_dof(mo, fe)[0]], lambda x: float(src(x[None, None, :])[0, 0]))
    assert abs(K[0] - k0).max() <= 1e-12 * abs(k0).max() and abs(F[0] - f0).max() <= 1e-12 * abs(f0).max()
    # global operator and residual, twice (bitwise repeat), from NaN-filled arrays
    Ao, bo = fo.assemble_poisson(mo, fe, src, sol=u)
    asm.assemble(A, res, sol, 2, (2.0, 1.3))
    v1, f1 = A.values().copy(), res.to_numpy().copy()
    assert np.isfinite(v1).all() and np.isfinite(f1).all()
    assert abs(v1 - Ao.data).max() <= 1e-12 * abs(Ao.data).max() and abs(f1 - bo).max() <= 1e-12 * abs(bo).max()
    asm.assemble(A, res, sol, 2, (2.0, 1.3))
    assert np.array_equal(v1, A.values()) and np.array_equal(f1, res.to_numpy())
    # constants lie in the kernel of the un-penalised operator also on the curved sub-parametric elements
    assert abs(Ao @ np.ones(n)).max() <= 1e-12 * abs(Ao.data).max()
    asm.destroy(), A.destroy()


@pytest.mark.parametrize("args,nl", [((2, 2, 2), 3), ((3, 2, 0), 3)])
@pytest.mark.parametrize("fe", ["serendipity", "constant"])
def test_prolongators_of_the_new_families(ctx, args, nl, fe):
    """BuildProlongatorMatrix (LinearImplicitSystem.cpp:761-909) with the element prolongators of HexQuadratic / QuadQuadratic / hex0 / quad0: the host loops,
    the device builder and the oracle give the same matrix; a function of the coarse space is reproduced on the fine level"""
    ms, mo = levels(args, nl), fo.build_levels(*args, nl)
    for l in range(1, nl):
        Po = fo.build_prolongator(mo[l - 1], mo[l], fe)
        mats = []
        for dev in (1, 0):
            ctx.set_option("device_setup", dev)
            try:
                P = capi.build_prolongator(ctx, ms[l - 1], ms[l], fe, zero_bdc=False)
            finally:
                ctx.set_option("device_setup", 1)
            mats.append(P.to_scipy())
            P.destroy()
        for S in mats:
            assert S.shape == Po.shape and np.array_equal(S.indptr, Po.indptr) and np.array_equal(S.indices, Po.indices) and np.array_equal(S.data, Po.data)
        if fe == "constant":
            D = Po.toarray()          # every child takes the value of its father
            assert np.array_equal(Po.data, np.ones(mo[l].nel)) and all(D[j, iel] == 1.0 for iel in range(mo[l - 1].nel) for j in mo[l - 1].child_elem[iel])
        else:
            # a polynomial of the serendipity space (complete degree two) is carried to the fine nodes exactly
            Xc, Xf = mo[l - 1].coords[:Po.shape[1]], mo[l].coords[:Po.shape[0]]
            q = lambda X: 1.0 + X[:, 0] - 2.0 * X[:, 1] + 0.5 * X[:, 0] * X[:, 1] + X[:, 0] ** 2 - 0.7 * X[:, 1] ** 2 + (0.3 * X[:, 2] * X[:, 0] if X.shape[1] == 3 else 0.0)
            assert abs(Po @ q(Xc) - q(Xf)).max() < 1e-13
    # with the Dirichlet rows / columns zeroed as init() asks for it
    P = capi.build_prolongator(ctx, ms[-2], ms[-1], fe, zero_bdc=True)
    ref = fo.build_prolongator(mo[-2], mo[-1], fe)
    if fe != "constant":
        ref = fo.zero_interpolator_dirichlet(ref, fo.dirichlet_dofs(mo[-1], fe), fo.dirichlet_dofs(mo[-2], fe))
    assert abs(P.to_scipy() - ref).max() == 0.0
    P.destroy()
    # and as a block of a stacked system (fh_build_system_prolongator)
    S = capi.build_system_prolongator(ctx, ms[-2], ms[-1], ["biquadratic", fe]).to_scipy()
    B = sp.block_diag([fo.build_prolongator(mo[-2], mo[-1], "biquadratic"), fo.build_prolongator(mo[-2], mo[-1], fe)]).tocsr()
    assert abs(S - B).max() == 0.0
    for m in ms:
        m.destroy()


@pytest.mark.parametrize("args", [(2, 2, 2), (4, 3, 0)])
def test_neumann_faces_of_the_serendipity_family(ctx, args):
    """QUAD8 / EDGE3 faces through elem_type::JacobianSur (ElemType.hpp:1089-1138, :1330-1380)"""
    m = levels(args, 2)[-1]
    mo = fo.build_levels(*args, 2)[-1]
    fe = "serendipity"
    n = m.n_dofs(fe)
    res = ctx.vector(n)
    flux = {-3: 0.2, -2: -1.5}
    capi.assemble_neumann(ctx, m, fe, res, flux)
    ref = fo.neumann_rhs(mo, fe, flux)
    assert abs(res.to_numpy() - ref).max() <= 1e-13 * abs(ref).max()
    # a constant flux integrates the area of the faces (the serendipity face functions sum to one)
    area = 1.0
    assert abs(res.to_numpy().sum() - (0.2 - 1.5) * area) < 1e-13
    m.destroy()


@pytest.mark.parametrize("args,nl", [((2, 2, 2), 3), ((4, 4, 0), 3)])
def test_multigrid_solve_with_the_serendipity_family(ctx, args, nl):
    """MGsolve (Galerkin chain by sparse triple products, V(2,2), GMRES) against the oracle's hierarchy and solve.  Damped Jacobi needs omega < 2 / 4.77 on
    HEX20 (the largest eigenvalue of D^-1 A of the 8^3 level; 1.33 for HEX27): 0.4 here -- with the bench's 2/3 the smoother amplifies and GMRES stalls at
    1e-9, on the device and in the oracle alike"""
    fe = "serendipity"
    pb = PoissonMG(ctx, *args, nl, fe=fe, omega=0.4).init()
    H = fo.build_poisson_hierarchy(*args, nl, fe, lambda xg: np.ones(xg.shape[:2]))
    pb.assemble()
    pb.prepare()
    for l in range(1, nl):
        assert abs(pb.P[l].to_scipy() - H.P[l]).max() == 0.0                 # prolongators bit-exact (pattern and values)
    for l in range(nl):
        assert abs(pb.A[l].to_scipy() - H.A[l]).max() <= 1e-12 * abs(H.A[l]).max()
    # one cycle on the assembled residual
    pb.zero_boundary_residuals()
    assert np.linalg.norm(pb.RES.to_numpy() - H.b) <= 1e-13 * np.linalg.norm(H.b)
    pb.vcycle()
    ref = fo.vcycle(H, nl - 1, H.b, omega=0.4, npre=2, npost=2)
    assert np.linalg.norm(pb.EPSC.to_numpy() - ref) <= 1e-11 * np.linalg.norm(ref)
    # full solve: 1e-10 relative with the direct solution of the oracle's system (north_star)
    its, rn = pb.mgsolve(outer="gmres", rtol=1e-14, maxit=100)
    pb.update_sol()
    xd = spla.spsolve(H.A[-1].tocsc(), H.b)
    assert np.linalg.norm(pb.SOL.to_numpy() - xd) <= 1e-10 * np.linalg.norm(xd)
    assert pb.RES.l2_norm() <= 1e-10 * np.linalg.norm(H.b)
    pb.destroy()


def test_the_shipped_serendipity_input_of_001_poisson(ctx, tmp_path):
    """applications/001_Poisson/input/input3D_Hex_serendipity.json (its text below: "fe_order" : "serendipity", mesh file input/cube_Hex.neu, four levels,
    SetBoundaryCondition of main.cpp:26-36: Dirichlet 0 everywhere but face 3, which carries the flux 0.2) through app_poisson on the GPU, against the
    oracle's direct solve of the same discrete problem on meshes the oracle refines itself from the coarse arrays (integer parity asserted first)"""
    text = SHIPPED_SERENDIPITY_INPUT
    ref_file = "/root/reference/applications/001_Poisson/input/input3D_Hex_serendipity.json"
    if os.path.exists(ref_file):      # (in the build container: the text below IS the shipped file's configuration)
        assert app.load_config(ref_file) == app.load_config(text)
    os.makedirs(tmp_path / "input")
    with open(os.path.join(HERE, "golden", "cube_Hex.neu"), "rb") as f:
        (tmp_path / "input" / "cube_Hex.neu").write_bytes(f.read())
    p = app.Poisson001(ctx, text, base_dir=str(tmp_path))
    assert p.fe == "serendipity" and p.nlevels == 4
    out_own = p.run()
    assert out_own["converged"] and len(out_own["history"]) <= 6           # under the input's own limits: six linear iterations, ||RES|| < 1e-9
    p.max_linear, p.abs_tol = 30, 1.e-13                                    # ... and iterated on, for the comparison with a direct solve
    out = p.run()
    assert out["converged"]
    # the oracle's side: coarse arrays from the Gambit reader (host code, tests/test_gambit.py), refined by the oracle
    m0 = capi.Mesh.read_gambit(os.path.join(HERE, "golden", "cube_Hex.neu"))
    ed, xy, ff = m0.arrays()
    mo = fo.Mesh("hex", ed.astype(np.int64), xy.copy(), ff.astype(np.int64))
    mo.own_size = list(m0.own_size)
    ml = [m0]
    for _ in range(3):
        mo = fo.refine(mo)
        ml.append(ml[-1].refine())
        e2, x2, f2 = ml[-1].arrays()
        assert np.array_equal(e2, mo.elem_dof) and np.array_equal(x2, mo.coords) and np.array_equal(f2, mo.face_flag)
    fe = "serendipity"
    n = fo.n_dofs(mo, fe)
    assert out["dofs"] == n
    nc = fo.ndofs("hex", fe)
    fn = fo.face_nodes("hex")
    bdc = set()
    for f in range(6):
        els = np.where((mo.face_flag[:, f] < -1) & (mo.face_flag[:, f] != -4))[0]
        nodes = fn[f][fn[f] < nc]
        bdc.update(mo.elem_dof[els][:, nodes].ravel().tolist())
    bdc = np.array(sorted(bdc))
    A, b = fo.assemble_poisson(mo, fe, lambda xg: np.zeros(xg.shape[:2]))
    b = b + fo.neumann_rhs(mo, fe, {-4: 0.2})
    A = fo.zero_rows(A, bdc, 1.0)
    b[bdc] = 0.0
    ref = spla.spsolve(A.tocsc(), b)
    assert abs(ref).max() > 1e-3
    assert abs(out["solution"] - ref).max() < 1e-10
    assert abs(out_own["solution"] - ref).max() < 1e-8                      # what the input's own tolerance leaves
    p.destroy()
    for m in ml:
        m.destroy()


# configuration of applications/001_Poisson/input/input3D_Hex_serendipity.json (a data file of the reference's application; compared with the file itself
# where the reference tree is present)
SHIPPED_SERENDIPITY_INPUT = """
{
    "multilevel_mesh" : { "first" : { "type" : { "filename" : "input/cube_Hex.neu" } } },
    "multilevel_solution" : { "multilevel_mesh" : { "first" : { "variable" : { "first" : {
              "name" : "T", "fe_order" : "serendipity", "init_func" : "0.", "func_source": "0.",
              "boundary_conditions" : [ { "facename" : "top", "bdc_type" : "dirichlet" },
                                        { "facename" : "right", "bdc_type" : "neumann", "bdc_func" : "0.2" } ] } } } } },
    "multilevel_problem" : { "multilevel_mesh" : { "first" : { "system" : { "poisson" : { "linear_solver" : {
                "max_number_linear_iteration" : 6, "abs_conv_tol" : 1.e-09,
                "type" : { "multigrid" : { "nlevels" : 4, "npresmoothing" : 1, "npostsmoothing" : 1, "mgtype" : "V_cycle",
                    "smoother" : { "type" : { "gmres" : { "ksp" : "gmres", "precond" : "ilu", "rtol" : 1.e-12, "atol" : 1.e-20, "divtol" : 1.e+50,
                                                          "max_its" : 4 } } } } } } } } } } }
}
"""
