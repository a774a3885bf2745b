"""GPU parity: Neumann boundary term (elem_type::JacobianSur, a5 of SURVEY 8) against the oracle."""
import os

import numpy as np
import pytest

from femus_amd import capi
from oracle import femus_oracle as fo

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fe_tables.npz"))


@pytest.mark.parametrize("args,fe", [((2, 2, 2), "biquadratic"), ((2, 2, 2), "linear"), ((4, 3, 0), "biquadratic"), ((4, 3, 0), "linear")])
def test_neumann_faces_match_oracle(ctx, args, fe):
    m = capi.Mesh.box(*args).refine()
    mo = fo.build_levels(*args, 2)[-1]
    ed, xy, ff = m.arrays()
    rng = np.random.default_rng(4)
    xy = xy + rng.uniform(-0.01, 0.01, xy.shape)          # curved faces
    m.set_coords(xy)
    mo.coords = xy
    flux = {-4: 2.5, -3: -1.0} if m.dim == 3 else {-3: 2.5, -2: -1.0}
    res = ctx.vector(m.nnode)
    res.fill(1.0)
    capi.assemble_neumann(ctx, m, fe, res, flux)
    ref = fo.neumann_rhs(mo, fe, flux)
    got = res.to_numpy()[:ref.size] - 1.0
    assert abs(got - ref).max() <= 1e-13 * max(abs(ref).max(), 1e-300)
    if m.dim == 3 and fe == "biquadratic":
        # the reference's own face node order (hex_lag::faceDofs, golden fixture) parametrises the faces differently:
        # the boundary integrals must not depend on it
        ref2 = fo.neumann_rhs(mo, fe, flux, face_tables=G["facedofs_hex"])
        assert abs(ref2 - ref).max() <= 1e-13 * abs(ref).max()


def test_neumann_constant_flux_integrates_the_area(ctx):
    m = capi.Mesh.box(3, 2, 2).refine()
    res = ctx.vector(m.nnode)
    capi.assemble_neumann(ctx, m, "biquadratic", res, {-4: 1.0})       # face x = 1 of the unit cube
    assert abs(res.sum() - 1.0) <= 1e-13
    capi.assemble_neumann(ctx, m, "biquadratic", res, {})              # no faces: no-op
    assert abs(res.sum() - 1.0) <= 1e-13


@pytest.mark.parametrize("args,fe", [((2, 2, 2), "biquadratic"), ((2, 2, 2), "linear"), ((4, 3, 0), "biquadratic")])
def test_neumann_parsed_flux_is_evaluated_at_the_face_gauss_points(ctx, args, fe):
    """the parsed-function branch of the 001_Poisson callback (main.cpp:495-553): the flux of a face is its ParsedFunction at the Gauss
    point; two faces with two expressions plus a constant face in one call, on curved faces, against the oracle's Gauss loop"""
    m = capi.Mesh.box(*args).refine()
    mo = fo.build_levels(*args, 2)[-1]
    ed, xy, ff = m.arrays()
    rng = np.random.default_rng(5)
    xy = xy + rng.uniform(-0.01, 0.01, xy.shape)
    m.set_coords(xy)
    mo.coords = xy
    e1, e2 = capi.Expr("0.2 + x*y - sin(3*z) + t", "x,y,z,t"), capi.Expr("exp(-x) * (y < 0.5) + 2", "x,y,z,t")
    f1 = lambda p: 0.2 + p[0] * p[1] - np.sin(3 * p[2]) + p[3]
    f2 = lambda p: np.exp(-p[0]) * (1.0 if p[1] < 0.5 else 0.0) + 2
    flags = (-4, -3, -2)
    res = ctx.vector(m.nnode)
    res.fill(1.0)
    capi.assemble_neumann(ctx, m, fe, res, {flags[0]: e1, flags[1]: e2, flags[2]: -0.75})
    ref = fo.neumann_rhs(mo, fe, {flags[0]: f1, flags[1]: f2, flags[2]: -0.75})
    got = res.to_numpy()[:ref.size] - 1.0
    assert abs(ref).max() > 0.01 and abs(got - ref).max() <= 1e-13 * abs(ref).max()
    # a constant expression is the constant path
    c = capi.Expr("0.2", "x,y,z,t")
    a, b = ctx.vector(m.nnode), ctx.vector(m.nnode)
    capi.assemble_neumann(ctx, m, fe, a, {flags[0]: c})
    capi.assemble_neumann(ctx, m, fe, b, {flags[0]: 0.2})
    assert np.array_equal(a.to_numpy(), b.to_numpy())
    for e in (e1, e2, c):
        e.destroy()
