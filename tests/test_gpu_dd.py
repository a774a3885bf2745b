"""GPU: the distributed (domain-decomposition) code path on one rank -- row-restricted operators in [owned | ghost]
numbering, replicated level below, halo objects, distributed assembler -- must give the single-GPU solution."""
import numpy as np
import pytest

import femus_amd
from femus_amd import dd
from femus_amd.poisson import PoissonMG

pytestmark = pytest.mark.gpu


def test_single_rank_distributed_path_matches_serial(ctx):
    comm = dd.SocketComm(0, 1)
    dp = dd.DistributedPoisson(ctx, comm, 1, 0, nb=2, nlevels=3)
    assert dp.H.plans[-1].n_ghost == 0 and dp.n_owned == 17 ** 3
    pb = PoissonMG(ctx, 2, 2, 2, 3).init()
    pb.assemble()
    pb.prepare()
    # distributed assembler == serial assembler on the owned rows
    dp.assemble()
    assert np.array_equal(dp.A[-1].values(), pb_raw_values(ctx))
    dp.set_penalty_top()
    assert abs(dp.A[-1].to_scipy() - pb.A[-1].to_scipy()).max() == 0.0
    its, rn = dp.solve(outer="gmres", rtol=1e-12)
    pb.mgsolve(outer="gmres", rtol=1e-12)
    xs = pb.EPS.to_numpy()
    xd = dp.EPSC.to_numpy()
    assert np.linalg.norm(xd - xs) <= 1e-10 * np.linalg.norm(xs)
    # one cycle runs and reduces the residual
    dp.assemble(); dp.set_penalty_top(); dp.zero_boundary_residuals(); dp.vcycle()
    assert np.isfinite(dp.EPSC.l2_norm())


def pb_raw_values(ctx):
    pb = PoissonMG(ctx, 2, 2, 2, 3).init()
    pb.assemble()
    return pb.A[-1].values()
