"""The serendipity (CONTINUOUS_SERENDIPITY) and piecewise-constant (DISCONTINUOUS_CONSTANT) families in the host half of the library: Mesh::GetSolutionDof
(Mesh.cpp:1021-1074, cases :1038-1050 and :1056-1059) through fh_system_elem_dofs, variable sizes, Dirichlet dofs -- integer work, bit-exact against the
oracle's restatement.  No device work."""
import numpy as np
import pytest

from femus_amd import capi
from oracle import femus_oracle as fo


def levels(args, nl):
    ms = [capi.Mesh.box(*args)]
    for _ in range(nl - 1):
        ms.append(ms[-1].refine())
    return ms


@pytest.mark.parametrize("args,nl", [((2, 2, 2), 2), ((3, 2, 0), 3), ((1, 1, 1), 1)])
def test_solution_dofs_of_all_five_families(args, nl):
    ms, mo = levels(args, nl), fo.build_levels(*args, nl)
    for a, b in zip(ms, mo):
        assert np.array_equal(a.arrays()[0], b.elem_dof)
        for fe in ("linear", "serendipity", "biquadratic", "constant"):
            assert a.n_dofs(fe) == fo.n_dofs(b, fe)
            nd, off, es = capi.system_elem_dofs(a, [fe])
            assert nd == fo.ndofs(b.geom, fe) and off.tolist() == [0, fo.n_dofs(b, fe)]
            assert np.array_equal(es, fo.elem_sys_dof(b, fe))
            assert es.max() == fo.n_dofs(b, fe) - 1          # the family owns exactly the leading ids
        # the serendipity family: vertices + edge mid-points, numbered before every face / cell node (own_size[1])
        nv, ne, _ = fo.class_ranges(b.geom)
        assert b.elem_dof[:, :ne].max() == a.own_size[1] - 1 and b.elem_dof[:, ne:].min() >= a.own_size[1]
        # a stacked system: KKoffset of LinearEquation.cpp:212-237 with one variable of every family
        fes = ["biquadratic", "serendipity", "constant", "linear", "pwlinear"]
        nd, off, es = capi.system_elem_dofs(a, fes)
        sizes = [fo.n_dofs(b, f) for f in fes[:4]] + [(b.dim + 1) * b.nel]
        assert off.tolist() == np.concatenate([[0], np.cumsum(sizes)]).tolist()
        p = 0
        for k, fe in enumerate(fes[:4]):
            blk = fo.elem_sys_dof(b, fe)
            assert np.array_equal(es[:, p:p + blk.shape[1]], off[k] + blk)
            p += blk.shape[1]
        iel = np.arange(b.nel)
        for i in range(b.dim + 1):
            assert np.array_equal(es[:, p + i], off[4] + i * b.nel + iel)
    for m in ms:
        m.destroy()


@pytest.mark.parametrize("args,nl", [((2, 2, 2), 2), ((4, 3, 0), 2)])
def test_dirichlet_dofs_of_the_serendipity_family(args, nl):
    m, mo = levels(args, nl)[-1], fo.build_levels(*args, nl)[-1]
    got = m.dirichlet_dofs("serendipity")
    assert np.array_equal(got, fo.dirichlet_dofs(mo, "serendipity"))
    # = the biquadratic family's boundary nodes that the serendipity family owns
    q2 = m.dirichlet_dofs("biquadratic")
    assert np.array_equal(got, q2[q2 < m.own_size[1]])
    assert m.dirichlet_dofs("constant").size == 0
    m.destroy()
