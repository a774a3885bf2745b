"""GPU: the halves of the ghost exchange (fh_halo_begin / fh_halo_end), the interior / interface split of operators over
[owned | ghost] columns that runs between them, and the RCCL calls themselves executed on ONE device (self exchange).

Reference behaviour: VecGhostUpdateBegin/End (PetscVector.hpp:605-608) and the MPIAIJ MatMult behind
NumericVector::matrix_mult (PetscVector.cpp:203-214): local columns are multiplied while the scatter is in flight."""
import numpy as np
import pytest
import scipy.sparse as sp

from femus_amd import capi

pytestmark = pytest.mark.gpu


class _SelfComm:
    """a one-rank 'transport': what the rank sends to itself comes back (host-staged path of fh_halo_*)"""

    def __init__(self):
        self.calls = 0

    def alltoallv(self, arrays, dtype):
        self.calls += 1
        return [np.array(a, dtype=dtype, copy=True) for a in arrays]

    def allreduce_sum(self, a):
        return np.array(a, copy=True)


def _operator(rng, n_own, n_ghost, nrows, ghost_rows):
    """CSR rows over [owned | ghost] columns; only `ghost_rows` read ghost columns"""
    rows, cols, vals = [], [], []
    for i in range(nrows):
        k = rng.integers(3, 40)
        c = rng.choice(n_own, size=min(k, n_own), replace=False)
        if i in ghost_rows:
            c = np.concatenate([c, n_own + rng.choice(n_ghost, size=min(5, n_ghost), replace=False)])
        rows += [i] * c.size
        cols += list(c)
        vals += list(rng.uniform(-1, 1, c.size))
    A = sp.csr_matrix((vals, (rows, cols)), shape=(nrows, n_own + n_ghost))
    A.sort_indices()
    return A


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_split_product_between_begin_and_end_equals_the_plain_product(ctx, mode):
    rng = np.random.default_rng(7 + mode)
    n_own, n_ghost = 6000, 500
    ghost_rows = set(range(5200, 6000)) | set(range(100, 110))      # interface rows cluster, as the nodes next to a cut do
    A_h = _operator(rng, n_own, n_ghost, n_own, ghost_rows)
    A = ctx.matrix_scipy(A_h)
    n_int, n_ifc = A.split_info(n_own)
    assert n_int > 0 and n_ifc > 0 and n_int + n_ifc > 10
    send_idx = rng.choice(n_own, size=n_ghost, replace=False).astype(np.int32)
    comm = _SelfComm()
    halo = capi.Halo.host(ctx, 0, 1, comm, [n_ghost], send_idx, [n_ghost])
    ghost_ids = np.arange(n_own, n_own + n_ghost, dtype=np.int32)
    x = ctx.vector(n_own + n_ghost, n_own, 0, ghost_ids)
    xo = rng.uniform(-1, 1, n_own)
    x.upload(xo)
    xfull = np.concatenate([xo, xo[send_idx]])
    b, dinv, y = ctx.vector_from(rng.uniform(-1, 1, n_own)), ctx.vector_from(rng.uniform(0.5, 2, n_own)), ctx.vector_from(rng.uniform(-1, 1, n_own))
    y0 = y.to_numpy()
    Ax = A_h @ xfull
    want = {0: Ax, 1: y0 + Ax, 2: b.to_numpy() - Ax, 3: xo + 0.7 * dinv.to_numpy() * (b.to_numpy() - Ax)}[mode]
    # 1. the two halves on their own: ghosts arrive
    halo.begin(x)
    halo.end()
    assert comm.calls == 1
    assert np.array_equal(x.get(ghost_ids), xo[send_idx])
    # 2. the ghosted product: exchange started, interior row blocks, exchange ended, interface row blocks
    x.upload(xo)                                         # owned part again; the ghost tail is refreshed by the call
    for overlap in (1, 0):
        ctx.set_option("halo_overlap", overlap)
        y.upload(y0)
        halo.spmv(A, x, y, mode, b if mode >= 2 else None, dinv if mode == 3 else None, 0.7)
        assert np.linalg.norm(y.to_numpy() - want) <= 1e-13 * np.linalg.norm(want), (mode, overlap)
    ctx.set_option("halo_overlap", 1)
    assert comm.calls == 3
    st = halo.stats()
    assert st["updates"] == 3 and st["bytes_sent"] == 3 * 8 * n_ghost
    halo.destroy()


def test_exchange_profile_reports_duration_and_exposed_part(ctx):
    rng = np.random.default_rng(3)
    n_own, n_ghost = 20000, 800
    A_h = _operator(rng, n_own, n_ghost, n_own, set(range(n_own - 2000, n_own)))
    A = ctx.matrix_scipy(A_h)
    send_idx = rng.choice(n_own, size=n_ghost, replace=False).astype(np.int32)
    halo = capi.Halo.host(ctx, 0, 1, _SelfComm(), [n_ghost], send_idx, [n_ghost])
    x = ctx.vector(n_own + n_ghost, n_own, 0, np.arange(n_own, n_own + n_ghost, dtype=np.int32))
    x.upload(rng.uniform(-1, 1, n_own))
    y = ctx.vector(n_own)
    halo.spmv(A, x, y)
    ctx.set_option("halo_profile", 1)
    halo.stats(reset=True)
    for _ in range(5):
        halo.spmv(A, x, y)
    ctx.set_option("halo_profile", 0)
    st = halo.stats()
    assert st["updates"] == 5 and st["exchange_ms"] > 0.0 and 0.0 <= st["exposed_ms"] <= st["exchange_ms"]
    halo.destroy()


def test_rccl_calls_execute_on_one_device_by_self_exchange():
    """ncclCommInitRank, a grouped ncclSend/ncclRecv pair and both ncclAllReduce forms run on the MI355X: rank 0 of a one-rank
    job sends its interface entries to itself (context option halo_self_rccl).  In a child process with a time limit, the way
    bench.py guards its multi-rank runs."""
    import socket
    from femus_amd import rccl_preflight
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ok, msg = rccl_preflight.run(0, 1, "127.0.0.1", port, 0, timeout=240.0)
    assert ok, msg


def test_distributed_cycle_with_rccl_exchanges_on_one_device():
    """tests/rccl_cycle_probe.py: a periodic problem as one rank's share of a distributed hierarchy, ghosts received from the rank itself
    through ncclSend/ncclRecv, replicated coarse level through ncclAllReduce -- the whole distributed V-cycle with RCCL on this GPU,
    with and without the interior / interface overlap, against the serial numpy cycle.  Child process with a time limit: a hang of
    a collective must fail the test, not the box."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "rccl_cycle_probe.py")], capture_output=True, text=True, timeout=300, cwd=root)
    assert out.returncode == 0 and "PROBE OK" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]


def test_planner_output_goes_straight_into_a_device_plan(ctx):
    """fh_dd_plan_create -> fh_dd_plan_halo -> fh_halo_sizes: the path a C++ launcher takes (no Python in between).  One rank: the plan
    has no ghosts, the device plan is inert; the counts the planner reports are the counts the device plan holds"""
    import ctypes
    L = capi.load_library()
    n = 50
    gid = np.arange(n, dtype=np.int64)[::-1].copy()
    owner = np.zeros(n, dtype=np.int32)
    need = np.ones(n, dtype=np.uint8)
    plan = ctypes.c_void_p()
    capi._chk(L.fh_dd_plan_create(0, 1, n, capi._p(gid), capi._p(owner), capi._p(need), None, None, ctypes.byref(plan)))
    a, b, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    capi._chk(L.fh_dd_plan_sizes(plan, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
    assert (a.value, b.value, c.value) == (n, 0, 0)
    halo = ctypes.c_void_p()
    capi._chk(L.fh_dd_plan_halo(plan, ctx.h, None, None, ctypes.byref(halo)))
    ns, nr = ctypes.c_int(-1), ctypes.c_int(-1)
    capi._chk(L.fh_halo_sizes(halo, ctypes.byref(ns), ctypes.byref(nr)))
    assert (ns.value, nr.value) == (0, 0)
    # an inert plan: update / begin / end are no-ops on any vector
    v = ctx.vector_from(np.arange(float(n)))
    capi._chk(L.fh_halo_update(halo, v.h))
    assert np.array_equal(v.to_numpy(), np.arange(float(n)))
    capi._chk(L.fh_halo_destroy(halo))
    capi._chk(L.fh_dd_plan_destroy(plan))
    with pytest.raises(capi.FemusHipError):
        capi._chk(L.fh_dd_plan_halo(None, ctx.h, None, None, ctypes.byref(halo)))
