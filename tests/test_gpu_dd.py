"""GPU: the distributed (domain-decomposition) code path on one rank -- row-restricted operators in [owned | ghost]
numbering, replicated level below, halo objects, distributed assembler -- must give the single-GPU solution."""
import os

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


# ---- several ranks on ONE GPU: the whole distributed device path (ghost-element assembly, halo updates inside the cycle,
# replicated level all-reduce, distributed dot products of the Krylov solver) with the host-staged transport in place of RCCL,
# which cannot connect two ranks that share a device.  Everything except the ncclSend/ncclRecv calls themselves is exercised.
def _ops_worker(rank, world, port, nb, nlevels, out, n_replicated=2):
    try:
        import femus_amd as fa
        from femus_amd import dd as ddm
        comm = ddm.SocketComm(rank, world, "127.0.0.1", port)
        ctx = fa.Context(0)
        dp = ddm.DistributedPoisson(ctx, comm, world, rank, nb=nb, nlevels=nlevels, transport="host", n_replicated=n_replicated)
        assert dp.prepare_ms is not None and dp.prepare_ms > 0 and dp.n_replicated == n_replicated
        # spoil every operator of the cycle, then re-prepare: the values must all come back from the device-side chain
        for A in dp.A + [dp.A_coarse] + ([dp.A_g0] if n_replicated == 2 else []):
            A.zero()
        dp.prepare()
        save = {"n_rep": dp.A_coarse.m()}
        Ar = dp.A_coarse.to_scipy()
        save.update(rep_data=Ar.data, rep_indices=Ar.indices, rep_indptr=Ar.indptr)
        if n_replicated == 2:        # the second replicated level: the whole global level-0 operator on every rank
            Ag = dp.A_g0.to_scipy()
            save.update(g0_data=Ag.data, g0_indices=Ag.indices, g0_indptr=Ag.indptr)
        for l, pl in enumerate(dp.H.plans):
            A = dp.A[l].to_scipy()
            save.update({"rows%d" % l: pl.gid[pl.owned], "cols%d" % l: pl.gid[np.concatenate([pl.owned, pl.ghost])],
                         "data%d" % l: A.data, "indices%d" % l: A.indices, "indptr%d" % l: A.indptr})
        dp.assemble(); dp.set_penalty_top(); dp.zero_boundary_residuals()
        dp.vcycle()
        save["x"] = dp.EPSC.to_numpy()[:dp.n_owned].copy()
        np.savez(out % rank, **save)
        comm.barrier()
        comm.close()
    except BaseException:
        _record_worker_failure("ops", rank, world)
        raise


@pytest.mark.parametrize("world,n_replicated", [(2, 2), (4, 2), (2, 1)])
def test_distributed_repreparation_gives_the_serial_galerkin_operators(tmp_path, world, n_replicated):
    """DistributedPoisson.prepare(): extended-box assembly + Galerkin chain + SetPenalty, owned rows gathered on the device, the
    replicated operator summed over the ranks -- every level operator equals the rows of the serial oracle chain (1e-12), as the
    reference's distributed MatPtAP would give them (LinearImplicitSystem.cpp:347-370, PetscMatrix.cpp:733-751)"""
    import scipy.sparse as sp
    import torch.multiprocessing as mp
    from oracle import femus_oracle as fo
    nb, nlevels = 2, 3
    out = str(tmp_path / "rank%d.npz")
    mp.spawn(_ops_worker, args=(world, _free_port(), nb, nlevels, out, n_replicated), nprocs=world, join=True)
    part = dd.BoxPartition(world, 0)
    p = part.p
    ONE = lambda xg: np.ones(xg.shape[:2])
    H = fo.build_poisson_hierarchy(p[0] * nb // 2, p[1] * nb // 2, p[2] * nb // 2, nlevels + 1, "biquadratic", ONE,
                                   hi=tuple(float(v) for v in p))
    ref_cycle = fo.vcycle(H, nlevels, H.b)
    for r in range(world):
        d = np.load(out % r)
        Arep = sp.csr_matrix((d["rep_data"], d["rep_indices"], d["rep_indptr"]), shape=(int(d["n_rep"]),) * 2)
        assert abs(Arep - H.A[0]).max() <= 1e-12 * abs(H.A[0]).max()                       # replicated level: same on every rank
        if n_replicated == 2:
            # the replicated smoothed level is the serial level-1 operator up to the node numbering of the two global meshes
            n1 = H.A[1].shape[0]
            Ag = sp.csr_matrix((d["g0_data"], d["g0_indices"], d["g0_indptr"]), shape=(n1, n1))
            from femus_amd import capi
            m_rep = capi.Mesh.box(p[0] * nb // 2, p[1] * nb // 2, p[2] * nb // 2, lo=(0., 0., 0.), hi=tuple(float(v) for v in p))
            m_g0 = m_rep.refine()
            assert np.array_equal(m_g0.arrays()[1], H.meshes[1].coords)              # same generator, same numbering
            assert abs(Ag - H.A[1]).max() <= 1e-12 * abs(H.A[1]).max()
        for l in range(nlevels):
            gid_ser, _ = dd.node_keys(H.meshes[l + 1].coords, l, nb, part)
            srt = np.argsort(gid_ser)
            rows = srt[np.searchsorted(gid_ser[srt], d["rows%d" % l])]
            cols = srt[np.searchsorted(gid_ser[srt], d["cols%d" % l])]
            A = sp.csr_matrix((d["data%d" % l], d["indices%d" % l], d["indptr%d" % l]), shape=(rows.size, cols.size))
            ref = H.A[l + 1].tocsr()[rows][:, cols]
            assert abs(A - ref).max() <= 1e-12 * abs(ref).max(), (r, l)
            # nothing of an owned row lies outside the [owned | ghost] columns
            assert abs(H.A[l + 1].tocsr()[rows]).sum() == pytest.approx(abs(ref).sum(), rel=1e-14)
        gid_top, _ = dd.node_keys(H.meshes[-1].coords, nlevels - 1, nb, part)
        srt = np.argsort(gid_top)
        pos = srt[np.searchsorted(gid_top[srt], d["rows%d" % (nlevels - 1)])]
        assert np.linalg.norm(d["x"] - ref_cycle[pos]) <= 1e-10 * np.linalg.norm(ref_cycle)


def _stacked_worker(rank, world, port, nb, nlevels, out, n_replicated):
    try:
        import femus_amd as fa
        from femus_amd import dd as ddm
        comm = ddm.SocketComm(rank, world, "127.0.0.1", port)
        ctx = fa.Context(0)
        dp = ddm.DistributedPoisson(ctx, comm, world, rank, nb=nb, nlevels=nlevels, transport="host", n_replicated=n_replicated)
        dp.assemble(); dp.set_penalty_top(); dp.zero_boundary_residuals()
        b = dp.RES.to_numpy()[:dp.n_owned].copy()
        ds = ddm.DistributedStacked(dp, nv=2, scale=[1.0, 0.5])
        top = ds.plans[-1]
        # the vectors carry the reference's system numbering: this rank owns [KKoffset[0][rank], KKoffset[2][rank]), a ghost is reached by its system row
        assert ds.RES.size() == 2 * int(dp.H.plans[-1].offsets[-1]) and ds.RES.local_size() == 2 * dp.n_owned
        its_s, _ = dp.solve(outer="gmres", rtol=1e-12)
        x = dp.EPSC.to_numpy()[:dp.n_owned].copy()
        ds.set_rhs([b, 2.5 * b])
        its, rn = ds.solve(outer="gmres", rtol=1e-12)
        xs = ds.EPSC.to_numpy()[:2 * dp.n_owned]
        # ghost refresh of the stacked vector through the device exchange plan: every ghost gets its owner's entry of the same system row
        v = ctx.vector(int(top.offsets[-1]), top.n_owned, int(top.offsets[rank]), top.ghost_global.astype(np.int32))
        v.upload((top.offsets[rank] + np.arange(top.n_owned)).astype(np.float64))
        ds.halos[-1].update(v)
        ghosts = v.to_numpy_with_ghosts()[top.n_owned:] if hasattr(v, "to_numpy_with_ghosts") else None
        np.savez(out % rank, x=x, x0=xs[:dp.n_owned], x1=xs[dp.n_owned:], its=its, its_s=its_s, n_ghost=top.n_ghost,
                 ghosts=np.zeros(0) if ghosts is None else ghosts, ghost_global=top.ghost_global)
        comm.barrier()
        comm.close()
    except BaseException:
        _record_worker_failure("stacked", rank, world)
        raise


@pytest.mark.parametrize("world,n_replicated", [(2, 2), (4, 2), (2, 1)])
def test_stacked_two_variable_system_on_the_distributed_device_path(tmp_path, world, n_replicated):
    """two variables stacked as LinearEquation stacks a system over the ranks (KKoffset, LinearEquation.cpp:212-237; fh_dd_system_offsets) run through the
    device machinery of the decomposition -- stacked exchange plans in fh_halo_*, owned-rows block operators, replicated levels, distributed GMRES around the
    distributed V-cycle: the solve of [A u = b, 0.5 A w = 2.5 b] gives u = the scalar distributed solve and w = 5 u on every rank"""
    import torch.multiprocessing as mp
    nb, nlevels = 2, 3
    out = str(tmp_path / "rank%d.npz")
    mp.spawn(_stacked_worker, args=(world, _free_port(), nb, nlevels, out, n_replicated), nprocs=world, join=True)
    for r in range(world):
        d = np.load(out % r)
        assert d["n_ghost"] > 0
        assert np.linalg.norm(d["x0"] - d["x"]) <= 1e-9 * np.linalg.norm(d["x"])
        assert np.linalg.norm(d["x1"] - 5.0 * d["x"]) <= 1e-9 * np.linalg.norm(5.0 * d["x"])
        if d["ghosts"].size:
            assert np.array_equal(d["ghosts"], d["ghost_global"].astype(np.float64))


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _record_worker_failure(tag, rank, world):
    """diagnostics only: the traceback of a failing worker process goes to gpurun_out/ (kept by the GPU harness) besides the
    pytest output, so that an intermittent failure of these multi-process tests leaves its cause behind (open items)"""
    import os
    import time
    import traceback
    try:
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "worker_failure_%s_rank%d_of_%d_%d.log" % (tag, rank, world, int(time.time()))), "w") as f:
            f.write(traceback.format_exc())
    except Exception:
        pass


def _rank_worker(rank, world, port, nb, nlevels, out, overlap=1):
    try:
        _rank_worker_body(rank, world, port, nb, nlevels, out, overlap)
    except BaseException:
        _record_worker_failure("uniform", rank, world)
        raise


def _rank_worker_body(rank, world, port, nb, nlevels, out, overlap=1):
    import femus_amd as fa
    from femus_amd import dd as ddm
    comm = ddm.SocketComm(rank, world, "127.0.0.1", port)
    ctx = fa.Context(0)
    ctx.set_option("debug_poison", 1)      # work buffers of the cycle and of GMRES start as NaN: nothing may be read before it is written / received
    ctx.set_option("halo_overlap", overlap)
    dp = ddm.DistributedPoisson(ctx, comm, world, rank, nb=nb, nlevels=nlevels, transport="host")
    dp.assemble()
    dp.set_penalty_top()
    dp.zero_boundary_residuals()
    b = dp.RES.to_numpy()[:dp.n_owned]
    for h in dp.halos:
        h.stats(reset=True)
    dp.vcycle()
    # V(2,2): per level one exchange per sweep after the first, one for the residual, one for the restriction (not into the
    # replicated level), one for the interpolation from a distributed level, two post-sweeps
    st = [h.stats()["updates"] for h in dp.halos]
    if dp.n_replicated == 2:     # the coarsest local level is replicated too: no exchange there, and none for the interpolation out of it
        assert st == [0] + ([1 + 1 + 0 + 0 + 2] if nlevels == 2 else [1 + 1 + 0 + 1 + 2] + [1 + 1 + 1 + 1 + 2] * (nlevels - 3) + [1 + 1 + 1 + 0 + 2]), st
    else:
        assert st == [1 + 1 + 0 + 1 + 2] + [1 + 1 + 1 + 1 + 2] * (nlevels - 2) + [1 + 1 + 1 + 0 + 2], st
    x = dp.EPSC.to_numpy()[:dp.n_owned].copy()
    its, rn = dp.solve(outer="gmres", rtol=1e-12, maxit=60)
    xs = dp.EPSC.to_numpy()[:dp.n_owned].copy()
    top = dp.H.plans[-1]
    # the vectors live in the reference's global numbering: operator()(global index) reaches owned and (refreshed) ghost entries
    assert dp.SOL.first_local == top.offsets[rank] and dp.SOL.n_global == top.offsets[-1]
    dp.SOL.upload((top.offsets[rank] + np.arange(dp.n_owned)).astype(np.float64))
    dp.halos[-1].update(dp.SOL)
    probe = np.concatenate([top.ghost_global[:5], top.offsets[rank] + np.arange(3)]).astype(np.int32)
    assert np.array_equal(dp.SOL.get(probe), probe.astype(np.float64))
    # assembly at a state that is not zero: the ghost entries the element loop reads arrive with the exchange inside assemble()
    gid_own = top.gid[top.owned]
    dp.SOL.upload(((gid_own % 97) / 97.0).astype(np.float64))
    dp.assemble()
    b_state = dp.RES.to_numpy()[:dp.n_owned].copy()
    np.savez(out % rank, gid=top.gid[top.owned], b=b, x=x, xs=xs, its=its, n_ghost=top.n_ghost, b_state=b_state)
    comm.barrier()
    comm.close()


@pytest.mark.parametrize("world,overlap", [(2, 1), (4, 1), (8, 1), (2, 0)])
def test_multi_rank_device_path_with_host_transport(tmp_path, world, overlap):
    import scipy.sparse.linalg as spla
    import torch.multiprocessing as mp
    from oracle import femus_oracle as fo
    nb, nlevels = 2, 2
    out = str(tmp_path / "rank%d.npz")
    mp.spawn(_rank_worker, args=(world, _free_port(), nb, nlevels, out, overlap), nprocs=world, join=True)
    part = dd.BoxPartition(world, 0)
    p = part.p
    ONE = lambda xg: np.ones(xg.shape[:2])
    H = fo.build_poisson_hierarchy(p[0] * nb // 2, p[1] * nb // 2, p[2] * nb // 2, nlevels + 1, "biquadratic", ONE,
                                   hi=tuple(float(v) for v in p))
    ref = fo.vcycle(H, nlevels, H.b)
    xd = spla.spsolve(H.A[-1].tocsc(), H.b)
    gid_ser, _ = dd.node_keys(H.meshes[-1].coords, nlevels - 1, nb, part)
    srt = np.argsort(gid_ser)
    _, b_state = fo.assemble_poisson(H.meshes[-1], "biquadratic", ONE, sol=(gid_ser % 97) / 97.0)      # Res = F - K u at a state
    seen = 0
    for r in range(world):
        d = np.load(out % r)
        pos = srt[np.searchsorted(gid_ser[srt], d["gid"])]
        assert np.array_equal(gid_ser[pos], d["gid"]) and d["n_ghost"] > 0
        assert np.linalg.norm(d["b"] - H.b[pos]) <= 1e-12 * np.linalg.norm(H.b)          # distributed assembly
        assert np.linalg.norm(d["b_state"] - b_state[pos]) <= 1e-12 * np.linalg.norm(b_state)      # ... with the state's ghost entries exchanged
        assert np.linalg.norm(d["x"] - ref[pos]) <= 1e-10 * np.linalg.norm(ref)          # distributed V-cycle
        assert np.linalg.norm(d["xs"] - xd[pos]) <= 1e-9 * np.linalg.norm(xd)            # distributed GMRES solve
        seen += d["gid"].size
    assert seen == xd.size


@pytest.mark.parametrize("transport", ["host", "gloo"])
def test_bench_contract_with_two_ranks_sharing_the_gpu(tmp_path, transport):
    """bench.py launched as the driver launches it for N > 1 (torch.distributed.run, one process per rank), on the one GPU of
    this box with the host-staged transport -- through the setup sockets ("host") and through torch.distributed gloo ("gloo": what
    bench.py falls back to when the RCCL preflight fails on a machine): rendezvous, distributed setup, timed steps, roofline leg and
    the JSON line"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FEMUS_BENCH_TRANSPORT=transport, FEMUS_BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1", TMPDIR=str(tmp_path))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--coarse", "2", "--levels", "3", "--kernel-reps", "3"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert "domain decomposition" in d["config"]["parallelism"] and "host-staged" in d["config"]["parallelism"]
    assert ("gloo" in d["config"]["parallelism"]) == (transport == "gloo")
    assert d["config"]["dofs_total"] == 33 * 17 * 17 and "roofline" in d and "cpu_baseline" not in d
    h = d["halo"]     # 3 local levels, the coarsest of them replicated, V(2,2): 0 + 5 + 5 exchanges per cycle, some of the exchange time hidden or not
    assert h["exchanges_per_cycle"] == 10 and h["exchanges_per_cycle_by_level"] == [0, 5, 5] and h["bytes_sent_per_cycle_this_rank"] > 0
    assert h["exchange_ms_per_cycle"] > 0 and 0 <= h["exposed_ms_per_cycle"] <= h["exchange_ms_per_cycle"]
    # a gradeable N > 1 line: the host time to issue a cycle, the measured number of all-reduces (restriction into the replicated level +
    # none for the dots of a plain cycle), the roofline object of the rank's own sweep kernel, and the solve headline
    assert h["host_issue_ms_per_cycle"] > 0 and h["allreduces_per_cycle"] >= 1
    assert d["roofline"]["frac"] > 0 and d["roofline"]["avg_launch_ms"] > 0
    assert d["solve_ms"] > 0 and d["solve"]["gmres_iterations"] >= 1 and d["solve"]["final_residual"] < 1e-6
    assert d["runtime_libraries"].get("libamdhip64") and len(d["runtime_libraries"]["libamdhip64"]) == 1       # ONE HIP runtime in the process


def test_bench_spawns_its_own_ranks_and_never_reports_fewer(tmp_path):
    """`python bench.py --gpus 2` without a launcher starts two ranks itself (here sharing the GPU; the RCCL preflight refuses two ranks
    on one device, so the line must say it fell back to the host-staged transport over the setup sockets -- torch.distributed is not
    initialised anywhere in that chain); the N = 1 run before it leaves its cpu_baseline for the N = 2 line to carry by reference; more
    ranks than devices without the sharing switch is an error, not an n_gpus = 1 line"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    small = ["--steps", "2", "--warmup", "1", "--coarse", "2", "--levels", "3", "--kernel-reps", "3", "--no-live-traffic"]
    env = dict(os.environ, TMPDIR=str(tmp_path))
    env.pop("WORLD_SIZE", None)
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + small, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-3000:]
    d1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    assert d1["n_gpus"] == 1 and d1["cpu_baseline"]["cores"] >= 1 and d1["solve"]["gmres_iterations"] >= 1
    two = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"] + small, env=dict(env, FEMUS_BENCH_SHARE_GPU="1"), cwd=root,
                         capture_output=True, text=True, timeout=900)
    assert two.returncode == 0, two.stdout[-2000:] + two.stderr[-4000:]
    d2 = json.loads([l for l in two.stdout.splitlines() if l.startswith("{")][-1])
    assert d2["n_gpus"] == 2 and d2["config"]["dofs_total"] == 33 * 17 * 17
    par = d2["config"]["parallelism"]
    assert "fell back from rccl" in par and "TCP sockets" in par and "gloo" not in par
    assert d2["cpu_baseline"]["by_reference"] and d2["cpu_baseline"]["value"] == d1["cpu_baseline"]["value"]
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "64"] + small, env=env, cwd=root, capture_output=True, text=True,
                         timeout=300)
    assert bad.returncode != 0 and "{" not in bad.stdout and "64" in bad.stderr


# ---- adaptive levels on several ranks (BASELINE config "MGAMR, 2 levels of AMR, 8 GPUs"), ranks sharing the one GPU -------------
def _amr_flag(x, level):
    return x[0] > 0.5 and (level < 2 or x[1] > 0.25)


def _amr_rank_worker(rank, world, port, nb, nlevels, n_uniform, out):
    try:
        _amr_rank_worker_body(rank, world, port, nb, nlevels, n_uniform, out)
    except BaseException:
        _record_worker_failure("adaptive", rank, world)
        raise


def _amr_rank_worker_body(rank, world, port, nb, nlevels, n_uniform, out):
    import femus_amd as fa
    from femus_amd import dd as ddm
    comm = ddm.SocketComm(rank, world, "127.0.0.1", port)
    ctx = fa.Context(0)
    ctx.set_option("debug_poison", 1)      # as in _rank_worker_body
    dp = ddm.DistributedPoisson(ctx, comm, world, rank, nb=nb, nlevels=nlevels, transport="host", flag_fn=_amr_flag, n_uniform=n_uniform)
    dp.assemble()                                     # a rank whose box is refined everywhere keeps the homogeneous path
    dp.set_penalty_top()
    dp.zero_boundary_residuals()
    b = dp.RES.to_numpy()[:dp.n_owned]
    dp.vcycle()
    x = dp.EPSC.to_numpy()[:dp.n_owned].copy()
    its, rn = dp.solve(outer="gmres", rtol=1e-12, maxit=80)
    xs = dp.EPSC.to_numpy()[:dp.n_owned].copy()
    top = dp.H.plans[-1]
    np.savez(out % rank, gid=top.gid[top.owned], b=b, x=x, xs=xs, its=its, adaptive=dp.adaptive)
    comm.barrier()
    comm.close()


@pytest.mark.parametrize("world", [2, 4])
def test_multi_rank_adaptive_levels_with_host_transport(tmp_path, world):
    import scipy.sparse.linalg as spla
    import torch.multiprocessing as mp
    from oracle import femus_oracle as fo
    from oracle import femus_oracle_amr as fam
    nb, nlevels, n_uniform = 2, 3, 1
    out = str(tmp_path / "rank%d.npz")
    mp.spawn(_amr_rank_worker, args=(world, _free_port(), nb, nlevels, n_uniform, out), nprocs=world, join=True)
    part = dd.BoxPartition(world, 0)
    p = part.p
    ONE = lambda xg: np.ones(xg.shape[:2])
    ms = fam.build_amr_levels(p[0] * nb // 2, p[1] * nb // 2, p[2] * nb // 2, n_uniform + 1, nlevels - n_uniform,
                              lambda x, level: _amr_flag(x, level - 1), hi=tuple(float(v) for v in p))
    H = fam.build_amr_hierarchy(ms, "biquadratic", ONE)
    ref = fo.vcycle(H, nlevels, H.b)
    xd = spla.spsolve(H.A[-1].tocsc(), H.b)
    gid_ser, _ = dd.node_keys(ms[-1].coords, nlevels - 1, nb, part)
    srt = np.argsort(gid_ser)
    seen, adaptive = 0, 0
    for r in range(world):
        d = np.load(out % r)
        adaptive += int(d["adaptive"])
        pos = srt[np.searchsorted(gid_ser[srt], d["gid"])]
        assert np.array_equal(gid_ser[pos], d["gid"])
        assert np.linalg.norm(d["b"] - H.b[pos]) <= 1e-12 * np.linalg.norm(H.b)          # assembly + projection, owned rows
        assert np.linalg.norm(d["x"] - ref[pos]) <= 1e-10 * np.linalg.norm(ref)          # distributed V-cycle on adaptive levels
        assert np.linalg.norm(d["xs"] - xd[pos]) <= 1e-9 * np.linalg.norm(xd)            # distributed GMRES solve
        seen += d["gid"].size
    assert seen == xd.size and H.hanging[-1].size > 0 and adaptive > 0


def test_config3_shape_at_full_size_equals_the_single_gpu_solve():
    """BASELINE configs[2] at its real size on two ranks (64^3 elements per rank, 2 x 1 x 1 box, 4 276 737 dofs, host-staged transport, both
    processes on this GPU): the distributed GMRES solution against the single-GPU solver on the global 128 x 64 x 64 mesh (five levels:
    the replicated level of the distributed run is its coarsest), 1e-9 relative.  tests/perf_probe_amr_dd.py in a child process."""
    import json
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "perf_probe_amr_dd.py"), "2", "8", "4", "2", "uniform"], capture_output=True, text=True,
                       timeout=900, cwd=os.path.dirname(here))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["dofs"] == 257 * 129 * 129 and out["rel_diff_vs_single_gpu"] < 1e-9
    assert sorted(k["owned"] for k in out["ranks"]) == [128 * 129 * 129, 129 ** 3]


@pytest.mark.parametrize("world", [2, 4])
def test_preflight_small_problem_stage(world):
    """second stage of femus_amd/rccl_preflight.py (what bench.py runs in child processes before it commits to the RCCL transport): the
    whole distributed path on a small problem against the single-GPU solve of the global mesh.  Here over the host-staged transport,
    ranks sharing this GPU -- the stage's own logic (plans, assembly, preparation, solve, comparison, exit codes) is what is tested."""
    import threading
    from femus_amd import rccl_preflight
    port = _free_port()
    res = [None] * world

    def go(r):
        res[r] = rccl_preflight.run(r, world, "127.0.0.1", port, 0, timeout=400.0, transport="host")

    ts = [threading.Thread(target=go, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert all(r is not None and r[0] for r in res), res


def test_preflight_small_problem_on_one_rank(ctx):
    """the same stage in-process on one rank with the RCCL transport selected (plans without neighbours are inert)"""
    from femus_amd import dd, rccl_preflight
    assert rccl_preflight.small_problem_check(ctx, dd.SocketComm(0, 1), 0, 1, "rccl") < 1e-9


# ---- arbitrary coarse meshes: native partitioner + topological node keys (the METIS path of the reference, MeshMetisPartitioning.cpp:71-155) ----
def _general_coarse_mesh(kind):
    """the same coarse mesh on every rank: "shuffled" = a 4 x 3 x 2 box whose elements come in random order with perturbed (curved) interior
    nodes; "gambit" = applications/001_Poisson/input/cube_Hex.neu (a data file of the reference, copied to tests/golden) refined once"""
    import os
    from femus_amd import capi
    if kind == "gambit":
        g0 = capi.Mesh.read_gambit(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cube_Hex.neu"))
        g = g0.refine()
        g0.destroy()
        return g
    box = capi.Mesh.box(4, 3, 2)
    rng = np.random.default_rng(11)
    g, _ = box.submesh(rng.permutation(box.nel).astype(np.int32))
    box.destroy()
    ed, xy, ff = g.arrays()
    on_bdry = np.zeros(g.nnode, dtype=bool)
    for f in range(6):
        loc = capi.fe_face_nodes("hex", "biquadratic", f)
        els = np.where(ff[:, f] < -1)[0]
        on_bdry[ed[els][:, loc].ravel()] = True
    xy = xy + np.where(on_bdry[:, None], 0.0, rng.uniform(-0.02, 0.02, xy.shape))
    g.set_coords(xy)
    return g


def _general_worker(rank, world, port, kind, nlevels, out):
    try:
        import femus_amd as fa
        from femus_amd import dd as ddm
        comm = ddm.SocketComm(rank, world, "127.0.0.1", port)
        ctx = fa.Context(0)
        G = _general_coarse_mesh(kind)
        dp = ddm.DistributedPoisson(ctx, comm, world, rank, nlevels=nlevels, transport="host", coarse_mesh=G)
        dp.assemble(); dp.set_penalty_top()
        its, rn = dp.solve(outer="gmres", rtol=1e-12, maxit=60)
        top = dp.H.plans[-1]
        xy = dp.full.meshes[-1].arrays()[1][top.owned]
        np.savez(out % rank, x=dp.EPSC.to_numpy()[:dp.n_owned].copy(), xy=xy, its=its, part=dp.partition, n_owned=dp.n_owned)
        comm.barrier()
        comm.close()
    except BaseException:
        _record_worker_failure("general", rank, world)
        raise


@pytest.mark.parametrize("kind,world", [("shuffled", 2), ("shuffled", 3), ("gambit", 2), ("gambit", 4)])
def test_general_partition_equals_the_serial_solve(ctx, tmp_path, kind, world):
    """a coarse mesh that is no box any more (random element order + curved elements; a Gambit file), partitioned natively into 2-4 parts,
    every rank building its extended mesh from the element adjacency and its ghost lists from the topological node keys: the distributed
    GMRES solution equals the single-GPU solve of the same hierarchy (1e-10); every node is owned exactly once"""
    import torch.multiprocessing as mp
    nlevels = 3
    out = str(tmp_path / "rank%d.npz")
    mp.spawn(_general_worker, args=(world, _free_port(), kind, nlevels, out), nprocs=world, join=True)
    G = _general_coarse_mesh(kind)
    meshes = [G]
    for l in range(1, nlevels):
        meshes.append(meshes[-1].refine())
    pb = PoissonMG(ctx, 0, 0, 0, nlevels, meshes=meshes).init()
    pb.assemble()
    pb.prepare()
    pb.mgsolve(outer="gmres", rtol=1e-12)
    ref = pb.EPSC.to_numpy()
    xy = meshes[-1].arrays()[1]
    key = lambda a: [tuple(v) for v in np.rint(a * 1e9).astype(np.int64)]
    pos = {k: i for i, k in enumerate(key(xy))}
    assert len(pos) == xy.shape[0]
    seen = np.zeros(xy.shape[0], dtype=int)
    for r in range(world):
        d = np.load(out % r)
        idx = np.array([pos[k] for k in key(d["xy"])])
        seen[idx] += 1
        assert np.linalg.norm(d["x"] - ref[idx]) <= 1e-10 * np.linalg.norm(ref), (r, int(d["its"]))
        cnt = np.bincount(d["part"], minlength=world)
        assert cnt.max() - cnt.min() <= 1
    assert np.all(seen == 1)                               # every node has exactly one owner
    pb.destroy()


# ---- adaptive levels on general partitions: weighted native partition + topological keys on selectively refined levels -------------------
def _general_amr_flag(x, level):
    return x[0] > 0.5 and (level < 2 or x[1] > 0.3)


def _general_amr_worker(rank, world, port, kind, nlevels, n_uniform, out):
    try:
        import femus_amd as fa
        from femus_amd import dd as ddm
        comm = ddm.SocketComm(rank, world, "127.0.0.1", port)
        ctx = fa.Context(0)
        ctx.set_option("debug_poison", 1)
        G = _general_coarse_mesh(kind)
        dp = ddm.DistributedPoisson(ctx, comm, world, rank, nlevels=nlevels, transport="host", coarse_mesh=G, flag_fn=_general_amr_flag,
                                    n_uniform=n_uniform)
        dp.assemble(); dp.set_penalty_top()
        its, rn = dp.solve(outer="gmres", rtol=1e-12, maxit=80)
        top = dp.H.plans[-1]
        xy = dp.full.meshes[-1].arrays()[1][top.owned]
        np.savez(out % rank, x=dp.EPSC.to_numpy()[:dp.n_owned].copy(), xy=xy, its=its, part=dp.partition, n_owned=dp.n_owned,
                 w=dp.elem_weights, adaptive=dp.adaptive, nel_local=dp.nel_local)
        comm.barrier()
        comm.close()
    except BaseException:
        _record_worker_failure("general_amr", rank, world)
        raise


@pytest.mark.parametrize("kind,world", [("shuffled", 2), ("shuffled", 3), ("gambit", 2)])
def test_general_partition_with_adaptive_levels_equals_the_serial_solve(ctx, tmp_path, kind, world):
    """two selectively refined levels over a coarse mesh that is no box (MGAMR on a METIS-style partition): the parts balance the
    finest-level descendants of the coarse elements (fh_mesh_partition_weighted), every rank flags its extended mesh with the same
    function, assembles and projects the hanging nodes there; the distributed solution equals the single-GPU solve of the same adaptive
    hierarchy (1e-9), every node is owned once, and the weighted parts are better balanced than the element-count partition"""
    import torch.multiprocessing as mp
    nlevels, n_uniform = 4, 2
    out = str(tmp_path / "rank%d.npz")
    mp.spawn(_general_amr_worker, args=(world, _free_port(), kind, nlevels, n_uniform, out), nprocs=world, join=True)
    G = _general_coarse_mesh(kind)
    meshes = dd.refine_levels(G, nlevels, _general_amr_flag, n_uniform)
    assert not meshes[-1].elem_levels()[1]                 # the finest level really is non-homogeneous
    pb = PoissonMG(ctx, 0, 0, 0, nlevels, meshes=meshes).init()
    pb.assemble()
    pb.prepare()
    pb.mgsolve(outer="gmres", rtol=1e-12)
    ref = pb.EPSC.to_numpy()
    xy = meshes[-1].arrays()[1]
    key = lambda a: [tuple(v) for v in np.rint(a * 1e9).astype(np.int64)]
    pos = {k: i for i, k in enumerate(key(xy))}
    assert len(pos) == xy.shape[0]
    seen = np.zeros(xy.shape[0], dtype=int)
    adaptive = 0
    for r in range(world):
        d = np.load(out % r)
        idx = np.array([pos[k] for k in key(d["xy"])])
        seen[idx] += 1
        adaptive += int(d["adaptive"])
        assert np.linalg.norm(d["x"] - ref[idx]) <= 1e-9 * np.linalg.norm(ref), (r, int(d["its"]))
    assert np.all(seen == 1) and adaptive > 0
    d0 = np.load(out % 0)
    w, part = d0["w"], d0["part"]
    assert w.sum() == meshes[-1].nel
    load = np.bincount(part, weights=w, minlength=world)
    plain = np.bincount(G.partition(world), weights=w, minlength=world)
    assert load.max() / load.mean() <= plain.max() / plain.mean() + 1e-12
    assert load.max() / load.mean() <= 1.25, load
    pb.destroy()
