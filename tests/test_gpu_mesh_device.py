"""Mesh refinement on the device (fh_mesh_refine_device, SURVEY 8 rows a8-a10: MeshRefinement.cpp:240-294, 356-417, 513-620 and the
first-touch renumbering Mesh.cpp:517-559) against the host restatement of the same loops (fh_mesh_refine_flagged, itself bit-exact against
the oracle in test_mesh_host.py / test_amr_host.py): every array of the new level -- numbering, boundary flags, element levels, child
lists, coordinates -- bit for bit, on uniform and adaptive hierarchies, boxes, quadrilaterals and the Gambit fixture; then the set-up calls
that read the mesh's device copy instead of an uploaded table (pattern, assembler, prolongator) against the ones that upload."""
import os

import numpy as np
import pytest

from femus_amd import capi
from oracle import femus_oracle as fo

from test_amr_host import edge_flag, ex4_flag, random_flag

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def same_mesh(a, b):
    assert (a.dim, a.nel, a.nnode, a.nloc, a.level) == (b.dim, b.nel, b.nnode, b.nloc, b.level)
    assert a.own_size == b.own_size
    ea, xa, fa_ = a.arrays()
    eb, xb, fb = b.arrays()
    assert np.array_equal(ea, eb)
    assert np.array_equal(fa_, fb)
    assert np.array_equal(xa.view(np.int64), xb.view(np.int64))          # coordinates: the same bits
    la, ha = a.elem_levels()
    lb, hb = b.elem_levels()
    assert np.array_equal(la, lb) and ha == hb
    for fe in ("linear", "biquadratic"):
        assert a.n_dofs(fe) == b.n_dofs(fe)
        assert np.array_equal(a.dirichlet_dofs(fe), b.dirichlet_dofs(fe))


def build(coarse, nlev, flag_fn, ctx):
    """two hierarchies from the same coarse mesh: host loops and device kernels"""
    host, dev = [coarse()], [coarse()]
    for l in range(1, nlev):
        fl = None if flag_fn is None else flag_fn(host[-1], l)
        host.append(host[-1].refine_flagged(np.ones(host[-1].nel, np.uint8) if fl is None else fl))
        dev.append(dev[-1].refine_device(ctx, fl))
    return host, dev


def distort(m, amp=0.04, seed=5):
    """curved elements: the coordinates of the finer levels are then real sums (rows of the prolongator with 3, 9, 27 weights)"""
    _, xy, _ = m.arrays()
    rng = np.random.default_rng(seed)
    m.set_coords(xy + amp * rng.uniform(-1, 1, xy.shape) / max(round(m.nel ** (1. / m.dim)), 1))
    return m


@pytest.mark.parametrize("box,nlev", [((2, 3, 2), 3), ((3, 2, 0), 4), ((1, 1, 1), 4), ((4, 4, 4), 3)])
@pytest.mark.parametrize("curved", [False, True])
def test_uniform_refinement_on_the_device_equals_the_host_loops(ctx, box, nlev, curved):
    coarse = (lambda: distort(capi.Mesh.box(*box))) if curved else (lambda: capi.Mesh.box(*box))
    host, dev = build(coarse, nlev, None, ctx)
    for l in range(1, nlev):
        same_mesh(dev[l], host[l])
        assert np.array_equal(dev[l - 1].child_elems(), host[l - 1].child_elems())
    # and the oracle's numbering (MeshRefinement + first touch), so that the two library paths cannot be wrong together
    mo = fo.coarse_box_mesh(*box)
    for l in range(1, nlev):
        mo = fo.refine(mo)
    assert np.array_equal(dev[-1].arrays()[0], mo.elem_dof)
    if not curved:
        assert np.array_equal(dev[-1].arrays()[1].view(np.int64), np.ascontiguousarray(mo.coords).view(np.int64))


@pytest.mark.parametrize("box,nu,ns,flag", [((2, 2, 2), 1, 3, ex4_flag), ((3, 3, 0), 1, 3, random_flag(1, 0.5)), ((2, 2, 2), 1, 2, random_flag(3, 0.4)),
                                           ((2, 2, 2), 2, 2, edge_flag)])
def test_adaptive_refinement_on_the_device_equals_the_host_loops(ctx, box, nu, ns, flag):
    def flags(m, l):
        return None if l < nu else m.flag_elements(flag)
    host, dev = build(lambda: distort(capi.Mesh.box(*box), 0.02), nu + ns, flags, ctx)
    for l in range(1, nu + ns):
        same_mesh(dev[l], host[l])
        assert np.array_equal(dev[l - 1].child_elems(), host[l - 1].child_elems())
    # hanging-node constraints are found through numbering, flags and coordinates: equal inputs, equal rows
    for fe in ("linear", "biquadratic"):
        for x, y in zip(dev[-1].amr_constraints(fe), host[-1].amr_constraints(fe)):
            assert np.array_equal(x, y)


@pytest.mark.parametrize("name,nlev", [("cube_Hex.neu", 3), ("nsbenc.neu", 4)])
def test_gambit_mesh_refined_on_the_device(ctx, name, nlev):
    """the reference's own input meshes: the HEX27 cube of 001_Poisson and the 98 curved QUAD9 elements of its known-answer test (four boundary sets)"""
    path = os.path.join(HERE, "golden", name)
    host, dev = build(lambda: capi.Mesh.read_gambit(path), nlev, None, ctx)
    for l in range(1, nlev):
        same_mesh(dev[l], host[l])


def test_mixing_the_two_paths_on_one_hierarchy(ctx):
    """a level refined on the host after one refined on the device (and the other way round): the device copy follows"""
    a = capi.Mesh.box(2, 2, 2)
    b = a.refine_device(ctx)
    c = b.refine()                       # host loop on a mesh that holds a device copy
    d = c.refine_device(ctx)             # device kernels on a mesh that holds none
    ref = [capi.Mesh.box(2, 2, 2)]
    for _ in range(3):
        ref.append(ref[-1].refine())
    for x, y in zip((b, c, d), ref[1:]):
        same_mesh(x, y)
    # a mesh whose coordinates were rewritten drops its device copy: the next refinement sees the new ones
    _, xy, _ = b.arrays()
    b.set_coords(xy * 2.0)
    ref[1].set_coords(xy * 2.0)
    same_mesh(b.refine_device(ctx), ref[1].refine())


@pytest.mark.parametrize("fe", ["biquadratic", "linear"])
@pytest.mark.parametrize("box", [(3, 2, 2), (4, 3, 0)])
def test_setup_calls_read_the_device_copy(ctx, fe, box):
    """pattern, prolongator and assembler made from a resident mesh = the ones made from uploaded tables"""
    host, dev = build(lambda: distort(capi.Mesh.box(*box)), 3, None, ctx)
    nc = {"linear": 2 ** host[0].dim, "biquadratic": 3 ** host[0].dim}[fe]
    for l in (1, 2):
        ed, xy, _ = host[l].arrays()
        A0 = ctx.matrix_from_elements(ed[:, :nc], host[l].n_dofs(fe))
        A1 = ctx.matrix_from_mesh(dev[l], fe)
        p0, p1 = A0.pattern(), A1.pattern()
        assert np.array_equal(p0[0], p1[0]) and np.array_equal(p0[1], p1[1])
        for zb in (True, False):
            P0 = capi.build_prolongator(ctx, host[l - 1], host[l], fe, zero_bdc=zb).to_scipy()
            P1 = capi.build_prolongator(ctx, dev[l - 1], dev[l], fe, zero_bdc=zb).to_scipy()
            assert np.array_equal(P0.indptr, P1.indptr) and np.array_equal(P0.indices, P1.indices) and np.array_equal(P0.data, P1.data)
        as0 = capi.Assembler(ctx, host[l], fe, A0, "seventh", elem_dof=ed, coords=xy)
        as1 = capi.Assembler(ctx, dev[l], fe, A1, "seventh")
        r0, r1 = ctx.vector(A0.m()), ctx.vector(A1.m())
        as0.assemble(A0, r0, source_kind=1, params=(1.0, 3.0))
        as1.assemble(A1, r1, source_kind=1, params=(1.0, 3.0))
        assert np.array_equal(A0.to_scipy().data, A1.to_scipy().data)
        assert np.array_equal(r0.to_numpy(), r1.to_numpy())
        for o in (as0, as1, A0, A1):
            o.destroy()


@pytest.mark.parametrize("box", [(1, 1, 0), (1, 1, 1), (3, 2, 0)])
def test_edge_cases_of_the_device_refinement(ctx, box):
    """no element flagged (the new level is a copy, non-homogeneous), a single element, every element flagged through a flag array"""
    for flags_of in (lambda m: np.zeros(m.nel, np.uint8), lambda m: np.ones(m.nel, np.uint8),
                     lambda m: (np.arange(m.nel) % 2 == 0).astype(np.uint8)):
        h, d = capi.Mesh.box(*box), capi.Mesh.box(*box)
        fl = flags_of(h)
        h2, d2 = h.refine_flagged(fl), d.refine_device(ctx, fl)
        same_mesh(d2, h2)
        assert np.array_equal(d.child_elems(), h.child_elems())
        # and once more on top of it (elements of older levels are never split again)
        fl2 = np.ones(h2.nel, np.uint8)
        same_mesh(d2.refine_device(ctx, fl2), h2.refine_flagged(fl2))
