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
