"""Adaptive refinement on the CPU box: the oracle's domain properties, and the host-side (integer /
geometry) half of the C-ABI -- fh_mesh_refine_flagged, fh_mesh_amr_constraints -- against the oracle, bit-exact for
every index and coordinate.  No device work: these entry points never touch the GPU."""
import numpy as np
import pytest
import scipy.sparse.linalg as spla

from femus_amd import capi
from oracle import femus_oracle as fo
from oracle import femus_oracle_amr as fa


def ex4_flag(x, level):
    """applications/MGAMR/ex4/ex4.cpp:49-62 moved to the unit box: x > centre on level 0, then also y > a quarter"""
    if level == 0:
        return x[0] > 0.5
    return x[0] > 0.5 and x[1] > 0.25


def corner_flag(x, level):
    return x[0] < 0.5 and x[1] < 0.5 and (x[2] < 0.5)


def poly_rhs(dim):
    def rhs(xg):
        s = 0.0
        for d in range(dim):
            p = 1.0
            for e in range(dim):
                if e != d:
                    p = p * xg[..., e] * (1.0 - xg[..., e])
            s = s + p
        return -2.0 * s
    return rhs


CASES = [((2, 2, 0), 1, 2, ex4_flag), ((2, 2, 2), 1, 2, ex4_flag), ((2, 2, 2), 2, 1, corner_flag), ((3, 2, 0), 2, 2, ex4_flag)]


def edge_flag(x, level):
    """level 0 | level 1 | level 2 meet along the line x = y = 0.5: a level-2 node there lies on interfaces with TWO coarser levels"""
    return x[0] > 0.5 if level == 0 else (x[0] > 0.5 and x[1] > 0.5)


@pytest.mark.parametrize("box,nu,ns,flag", CASES)
@pytest.mark.parametrize("fe", ["biquadratic", "linear"])
def test_oracle_partition_of_unity_and_interpolation(box, nu, ns, flag, fe):
    """the CONSISTENT variant of the constraints (mode "coarsest"): rows sum to one, polynomials are reproduced"""
    ms = fa.build_amr_levels(*box, nu, ns, flag)
    assert not ms[-1].homogeneous
    for m in ms:
        if m.homogeneous:
            continue
        P, hang = fa.build_amr_prolongator(m, fe, "coarsest")
        assert hang.size > 0
        # weights of every hanging node sum to one; its masters are regular nodes
        assert abs(np.asarray(P.sum(axis=1)).ravel() - 1.0).max() < 1e-14
        reg = np.setdiff1d(np.arange(P.shape[0]), hang)
        assert np.all(np.isin(P[hang].tocoo().col[P[hang].tocoo().data != 0.0], reg))
        # P_amr reproduces every function of the FE space: a Q1 (Q2) polynomial sampled at the regular nodes is
        # interpolated exactly at the hanging nodes
        X = m.coords[:fo.n_dofs(m, fe)]
        deg = 1 if fe == "linear" else 2
        u = np.prod(1.0 + 0.3 * X + (0.7 * X ** 2 if deg == 2 else 0.0), axis=1)
        v = u.copy()
        v[hang] = -99.0
        assert abs(P @ v - u).max() < 1e-14


@pytest.mark.parametrize("box,nu,ns,flag", CASES[:3])
def test_oracle_q2_polynomial_is_solved_exactly(box, nu, ns, flag):
    """u = prod x_d (1 - x_d) lies in the constrained Q2 space of any adaptive box mesh, so the Galerkin solution is u
    itself: a wrong hanging-node weight, a missed constraint or a discontinuity would show up at O(h^2), not 1e-13"""
    ms = fa.build_amr_levels(*box, nu, ns, flag)
    H = fa.build_amr_hierarchy(ms, "biquadratic", poly_rhs(ms[0].dim), mode="coarsest")
    x = spla.spsolve(H.A[-1].tocsc(), H.b)
    x = H.Pamr[-1] @ x
    X = ms[-1].coords
    assert abs(x - np.prod(X * (1 - X), axis=1)).max() < 1e-13
    # the same hierarchy drives the multigrid-preconditioned GMRES of the reference
    xg, hist = fo.solve_gmres_mg(H, rtol=1e-12, maxit=40, omega=2. / 3., npre=2, npost=2)
    assert len(hist) <= 16
    assert abs(H.Pamr[-1] @ xg - x).max() < 1e-10


@pytest.mark.parametrize("box,nu,ns,flag", CASES)
def test_host_refinement_matches_oracle_bit_exact(box, nu, ns, flag):
    mo = fa.build_amr_levels(*box, nu, ns, flag)
    mh = [capi.Mesh.box(*box)]
    for l in range(1, nu + ns):
        flags = np.ones(mh[-1].nel, np.uint8) if l < nu else mh[-1].flag_elements(flag)
        mh.append(mh[-1].refine_flagged(flags))
    for a, b in zip(mh, mo):
        ed, xy, ff = a.arrays()
        assert np.array_equal(ed, b.elem_dof)
        assert np.array_equal(xy, b.coords)                   # same sums in the same order: bit-exact
        assert np.array_equal(ff, b.face_flag)
        lev, hom = a.elem_levels()
        assert np.array_equal(lev, fa.elem_levels(b)) and hom == b.homogeneous
        assert a.own_size == list(b.own_size)
    for a, b in zip(mh[:-1], mo[:-1]):
        assert np.array_equal(a.child_elems(), b.child_elem)
    for a in mh:
        a.destroy()


def test_reference_map_at_a_node_on_two_interfaces():
    """Mesh::GetAMRRestrictionAndAMRSolidMark as written (Mesh.cpp:1352-1830): on the line where level 0, 1 and 2 meet, the pair
    (1, 2) overwrites the entries the pair (0, 2) wrote for the masters both descriptions share, and the genealogy walk skips the path
    through the intermediate hanging node ("alreadyFound").  Hand-computed for the Q2 node at 1/8 of a level-0 edge [a, m, b] that is
    also at the middle of the first half [a, g, m] of a level-1 edge: kept weights w1_a = 0.375 (level-1 description), w1_m = -0.125,
    w0_b = -0.09375 (level-0 description): sum 0.15625 -- not one.  Both variants agree on every other node."""
    ms = fa.build_amr_levels(2, 2, 2, 1, 2, edge_flag)
    for fe in ("linear", "biquadratic"):
        ref, con = fa.amr_restriction(ms[-1], fe, "reference"), fa.amr_restriction(ms[-1], fe, "coarsest")
        assert sorted(ref) == sorted(con)
        sums = {h: sum(r.values()) for h, r in ref.items()}
        odd = [h for h, v in sums.items() if abs(v - 1.0) > 1e-12]
        assert all(abs(sum(r.values()) - 1.0) < 1e-13 for r in con.values())
        X = ms[-1].coords
        # only next to the line where the three levels meet: the node on it, and the level-2 nodes of the level-1 face y = 0.5 whose
        # masters include a level-1 node that hangs on level 0 itself (direct entry kept, path through that node dropped)
        assert odd and all(abs(X[h][1] - 0.5) < 1e-14 and 0.5 - 1e-14 <= X[h][0] <= 0.75 for h in odd)
        for h in ref:
            if h not in odd:
                assert set(ref[h]) == set(con[h]) and max(abs(ref[h][m_] - con[h][m_]) for m_ in ref[h]) < 1e-14
        if fe == "biquadratic":
            assert any(abs(sums[h] - 0.15625) < 1e-14 and sorted(np.round(list(ref[h].values()), 12)) == [-0.125, -0.09375, 0.375] for h in odd)


def test_single_level_jumps_give_the_same_map_in_both_variants():
    ms = fa.build_amr_levels(2, 2, 2, 2, 1, corner_flag)
    for fe in ("linear", "biquadratic"):
        ref, con = fa.amr_restriction(ms[-1], fe, "reference"), fa.amr_restriction(ms[-1], fe, "coarsest")
        assert sorted(ref) == sorted(con) and len(ref) > 0
        for h in ref:
            assert set(ref[h]) == set(con[h]) and max(abs(ref[h][m_] - con[h][m_]) for m_ in ref[h]) < 1e-14


@pytest.mark.parametrize("mode", ["reference", "coarsest"])
@pytest.mark.parametrize("box,nu,ns,flag", CASES + [((2, 2, 2), 1, 2, edge_flag)])
@pytest.mark.parametrize("fe", ["biquadratic", "linear"])
def test_host_constraints_match_oracle(box, nu, ns, flag, fe, mode):
    mo = fa.build_amr_levels(*box, nu, ns, flag)
    mh = [capi.Mesh.box(*box).set_amr_mode(mode)]
    for l in range(1, nu + ns):
        flags = np.ones(mh[-1].nel, np.uint8) if l < nu else mh[-1].flag_elements(flag)
        mh.append(mh[-1].refine_flagged(flags))
    for a, b in zip(mh, mo):
        hang, ptr, master, w = a.amr_constraints(fe)
        R = fa.amr_restriction(b, fe, mode) if not b.homogeneous else {}
        assert np.array_equal(hang, np.array(sorted(R), dtype=np.int64))     # integer work: identical
        for k, l in enumerate(hang):
            row = sorted(R[int(l)].items())
            assert np.array_equal(master[ptr[k]:ptr[k + 1]], [j for j, _ in row])
            # weights are basis values at an inverse-mapped point: two Newton implementations, 1e-14
            assert abs(w[ptr[k]:ptr[k + 1]] - np.array([v for _, v in row])).max() < 1e-14
    for a in mh:
        a.destroy()


def test_uniform_flags_reproduce_uniform_refinement():
    a = capi.Mesh.box(2, 3, 2)
    u = a.refine()
    f = a.refine_flagged(np.ones(a.nel, np.uint8))
    for x, y in zip(u.arrays(), f.arrays()):
        assert np.array_equal(x, y)
    assert f.elem_levels()[1]
    hang, _, _, _ = f.amr_constraints("biquadratic")
    assert hang.size == 0


def random_flag(seed, prob):
    """a flag that looks random but is a function of the element centroid: the host library and the oracle evaluate it on bit-identical
    centroids, so both refine the same elements -- scattered single elements, level jumps of two and three, islands and holes"""
    def flag(x, level):
        key = (int(round(x[0] * 8192)) * 73856093) ^ (int(round(x[1] * 8192)) * 19349663) ^ (int(round(x[2] * 8192)) * 83492791) ^ (level * 2654435761) ^ seed
        return ((key * 2246822519) >> 7) % 1000 < prob * 1000
    return flag


@pytest.mark.parametrize("mode", ["reference", "coarsest"])
@pytest.mark.parametrize("box,nu,ns,seed,prob", [((3, 3, 0), 1, 3, 1, 0.5), ((4, 2, 0), 1, 3, 2, 0.3), ((2, 2, 2), 1, 2, 3, 0.4), ((2, 2, 2), 1, 2, 4, 0.7),
                                                  ((3, 2, 2), 1, 2, 5, 0.15), ((2, 2, 0), 2, 3, 6, 0.6)])
def test_randomly_flagged_refinement_matches_oracle(box, nu, ns, seed, prob, mode):
    """scattered flags: numbering, coordinates, child tables bit-exact; hanging-node sets identical and weights to 1e-14, both map variants,
    Q2 and Q1"""
    flag = random_flag(seed, prob)
    mo = fa.build_amr_levels(*box, nu, ns, flag)
    mh = [capi.Mesh.box(*box).set_amr_mode(mode)]
    for l in range(1, nu + ns):
        flags = np.ones(mh[-1].nel, np.uint8) if l < nu else mh[-1].flag_elements(flag)
        mh.append(mh[-1].refine_flagged(flags))
    assert not mo[-1].homogeneous and fa.elem_levels(mo[-1]).max() - fa.elem_levels(mo[-1]).min() >= 2        # jumps of two levels and more occur
    for a, b in zip(mh, mo):
        ed, xy, ff = a.arrays()
        assert np.array_equal(ed, b.elem_dof) and np.array_equal(xy, b.coords) and np.array_equal(ff, b.face_flag)
        assert np.array_equal(a.elem_levels()[0], fa.elem_levels(b))
        for fe in ("biquadratic", "linear"):
            hang, ptr, master, w = a.amr_constraints(fe)
            R = fa.amr_restriction(b, fe, mode) if not b.homogeneous else {}
            assert np.array_equal(hang, np.array(sorted(R), dtype=np.int64))
            for k, l in enumerate(hang):
                row = sorted(R[int(l)].items())
                assert np.array_equal(master[ptr[k]:ptr[k + 1]], [j for j, _ in row])
                assert abs(w[ptr[k]:ptr[k + 1]] - np.array([v for _, v in row])).max() < 1e-14
    for a, b in zip(mh[:-1], mo[:-1]):
        assert np.array_equal(a.child_elems(), b.child_elem)
    for a in mh:
        a.destroy()
