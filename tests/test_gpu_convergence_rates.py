"""The reference's own check of the Poisson multigrid path (no stored number exists for it, "the only check is analytic"):
applications/000_tutorial/ex02_poisson_a_manufactured_sol_convergence_rate_with_analytical_sol -- u = cos(pi x) cos(pi y) (cos(pi z)) on
[-1/2, 1/2]^dim with homogeneous Dirichlet data (:42-56), solved on a sequence of uniformly refined meshes, L2 norm and H1 semi-norm of the
error by the quadrature of the assembly (:232-241) and the orders log2(e_l / e_{l+1}) printed (:270-325): 3 and 2 for LAGRANGE SECOND, 2 and 1
for FIRST.  Here through the device path (assembly, Galerkin hierarchy, GMRES + V(2,2) to 1e-12); the error norms are evaluated with the
oracle's FE tables on the host."""
import numpy as np
import pytest

from femus_amd.poisson import PoissonMG
from oracle import femus_oracle as fo

pytestmark = pytest.mark.gpu


def error_norms(mesh_arrays, geom, fe, uh):
    ed, xy, _ = mesh_arrays
    et = fo.ElemType(geom, fe, "seventh")
    etg = fo.ElemType(geom, "biquadratic", "seventh")                      # geometry is always biquadratic
    dim = xy.shape[1]
    X = xy[ed]                                                             # [nel, nloc, dim]
    Jm = np.einsum("gna,enb->egab", etg.dphi, X)
    det = np.linalg.det(Jm)
    JI = np.linalg.inv(Jm)
    xg = np.einsum("gn,end->egd", etg.phi, X)
    grad = np.einsum("gna,egba->egnb", et.dphi, JI)
    ul = uh[ed[:, :et.nc]]
    u_h = np.einsum("gn,en->eg", et.phi, ul)
    gu_h = np.einsum("egnb,en->egb", grad, ul)
    c, s = np.cos(np.pi * xg), np.sin(np.pi * xg)
    u = np.prod(c, axis=2)
    gu = np.stack([-np.pi * s[:, :, d] * np.prod(np.delete(c, d, axis=2), axis=2) for d in range(dim)], axis=2)
    w = det * etg.w[None, :]
    return np.sqrt(np.sum(w * (u_h - u) ** 2)), np.sqrt(np.sum(w[:, :, None] * (gu_h - gu) ** 2))


@pytest.mark.parametrize("dim,fe,levels", [(2, "biquadratic", 5), (2, "linear", 6), (3, "biquadratic", 4), (3, "linear", 4)])
def test_orders_of_convergence_of_the_manufactured_solution(ctx, dim, fe, levels):
    geom = "quad" if dim == 2 else "hex"
    l2, h1 = [], []
    for nl in range(2, levels + 1):
        # the callback solves lap u = f (Res = (-f phi - grad phi . grad u) w): f = lap u = -dim pi^2 prod cos(pi x_d) (:53-57), source kind 2 (p0 prod cos(p1 x_d))
        pb = PoissonMG(ctx, 2, 2, 2 if dim == 3 else 0, nl, fe=fe, lo=(-0.5, -0.5, -0.5), hi=(0.5, 0.5, 0.5), source_kind=2,
                       params=(-dim * np.pi ** 2, np.pi)).init()
        pb.assemble()
        pb.prepare()
        pb.mgsolve(outer="gmres", rtol=1e-12, maxit=100)
        pb.update_sol()
        e0, e1 = error_norms(pb.meshes[-1].arrays(), geom, fe, pb.SOL.to_numpy())
        l2.append(e0), h1.append(e1)
        pb.destroy()
    o2 = np.log2(np.array(l2[:-1]) / np.array(l2[1:]))
    o1 = np.log2(np.array(h1[:-1]) / np.array(h1[1:]))
    p = 2 if fe == "biquadratic" else 1
    print(dim, fe, "L2 errors", l2, "orders", o2, "H1 semi-norm errors", h1, "orders", o1)
    assert abs(o2[-1] - (p + 1)) < 0.15 and abs(o1[-1] - p) < 0.15
    assert np.all(np.diff(l2) < 0) and np.all(np.diff(h1) < 0)
