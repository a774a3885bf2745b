"""RCCL preflight for multi-rank runs: a ring ghost exchange and an all-reduce through the library's own fh_halo_* calls, in a
child process of each rank, so that a launcher (bench.py) can fall back to independent problems when the RCCL path fails OR HANGS on
the machine at hand instead of hanging itself.

    python -m femus_amd.rccl_preflight RANK WORLD ADDR PORT DEVICE        (exit code 0 = data arrived and is correct)

`run(rank, world, addr, port, device, timeout)` starts that command, waits at most `timeout` seconds, kills exactly the child it
started if it is still running, and returns (ok, message)."""
import os
import subprocess
import sys


def small_problem_check(ctx, comm, rank, world, transport, nb=2, nlevels=3):
    """the WHOLE distributed path once on a small problem (nb^3 coarse elements per rank, nlevels levels): per-level exchange plans on one
    communicator, assembly, device-side preparation with the all-reduced replicated operator, GMRES with the overlapped cycles -- and the
    owned part of the solution against the single-GPU solver on the global mesh, which every rank runs for itself.  Returns the relative
    difference (max over ranks is taken by the caller)."""
    import numpy as np
    from . import capi, dd
    from .poisson import PoissonMG
    dp = dd.DistributedPoisson(ctx, comm, world, rank, nb=nb, nlevels=nlevels, transport=transport)
    dp.assemble()
    dp.set_penalty_top()
    its, rn = dp.solve(outer="gmres", rtol=1e-12, maxit=60)
    top = dp.H.plans[-1]
    mine = dp.EPSC.to_numpy()[:dp.n_owned]
    part = dd.BoxPartition(world, 0)
    p = part.p
    ms = [capi.Mesh.box(max(1, p[0] * nb // 2), max(1, p[1] * nb // 2), max(1, p[2] * nb // 2), hi=tuple(float(v) for v in p))]
    for _ in range(nlevels):
        ms.append(ms[-1].refine())
    pb = PoissonMG(ctx, 0, 0, 0, nlevels + 1, meshes=ms).init()
    pb.assemble()
    pb.prepare()
    pb.mgsolve(outer="gmres", rtol=1e-13, maxit=80)
    xs = pb.EPS.to_numpy()
    gid_ser, _ = dd.node_keys(ms[-1].arrays()[1], nlevels - 1, nb, part)
    srt = np.argsort(gid_ser)
    g = top.gid[top.owned]
    pos = srt[np.searchsorted(gid_ser[srt], g)]
    if not np.array_equal(gid_ser[pos], g) or its >= 60:
        return 1.0
    return float(np.linalg.norm(mine - xs[pos]) / max(np.linalg.norm(xs), 1e-300))


def _child(rank, world, addr, port, device, transport="rccl"):
    import numpy as np
    from . import capi, Context
    from .dd import SocketComm
    comm = SocketComm(rank, world, addr, port, timeout=60.0)
    ctx = Context(device)
    if transport != "rccl":          # the small-problem stage alone, over the host-staged transport (tests of this file's own logic)
        diff = small_problem_check(ctx, comm, rank, world, transport)
        oks = comm.allgather_obj(bool(diff < 1e-9))
        comm.close()
        return 0 if all(oks) else 4
    if world == 1:
        # one GPU: the ring neighbour is the rank itself.  "halo_self_rccl" gives the one-rank plan a real communicator, so
        # ncclCommInitRank, the grouped ncclSend/ncclRecv and ncclAllReduce below all execute on the device
        ctx.set_option("halo_self_rccl", 1)
    uid = comm.bcast_obj(capi.Halo.unique_id() if rank == 0 else None)
    n = 1024
    nxt, prv = (rank + 1) % world, (rank - 1) % world
    send_counts = np.zeros(world, dtype=np.int32)
    recv_counts = np.zeros(world, dtype=np.int32)
    send_counts[nxt] = n
    recv_counts[prv] = n
    halo = capi.Halo(ctx, rank, world, uid, send_counts, np.arange(n, dtype=np.int32), recv_counts)
    x = ctx.vector(2 * n, n, 0, np.arange(n, 2 * n, dtype=np.int32))
    x.upload(1000.0 * rank + np.arange(n))
    halo.begin(x)          # the two halves, as the distributed operators use them (interior rows run in between)
    halo.end()
    # read the ghosts through an operator with one entry per row in the ghost columns: y_i = x_ghost[i]
    A = ctx.matrix_csr(n, 2 * n, np.arange(n + 1, dtype=np.int32), np.arange(n, 2 * n, dtype=np.int32), np.ones(n))
    y = ctx.vector(n)
    y.matrix_mult(x, A)
    got = y.to_numpy()
    ok = np.array_equal(got, 1000.0 * prv + np.arange(n))
    s = halo.allreduce_sum(np.array([1.0, float(rank)]))
    ok = ok and s[0] == world and s[1] == world * (world - 1) / 2
    # device-vector all-reduce (the replicated coarse level's right-hand side)
    y.upload(np.full(n, float(rank + 1)))
    halo.allreduce_vec(y)
    ok = ok and np.array_equal(y.to_numpy(), np.full(n, world * (world + 1) / 2.0))
    # values of a matrix with one pattern on all ranks (the replicated coarse operator)
    halo.allreduce_mat(A)
    ok = ok and np.array_equal(A.values(), np.full(n, float(world)))
    st = halo.stats()
    ok = ok and st["updates"] == 1 and st["bytes_sent"] == 8 * n
    ctx.sync()
    oks = comm.allgather_obj(bool(ok))
    halo.destroy()
    if not all(oks):
        comm.close()
        return 3
    if world > 1:
        # second stage: everything bench.py is about to do, once, on a small problem -- a path that hangs or gives wrong numbers with
        # RCCL on this machine ends here, in the child
        diff = small_problem_check(ctx, comm, rank, world, "rccl")
        oks = comm.allgather_obj(bool(diff < 1e-9))
        if not all(oks):
            comm.close()
            return 4
    comm.close()
    if rank == 0:       # which RCCL / HIP runtime the exchange above really ran on (last line of the child's output: run() appends it to its message)
        from . import loaded_runtimes
        rt = loaded_runtimes()
        print("runtime: librccl %s version %s, libamdhip64 %s" % (",".join(rt.get("librccl", ["?"])), rt.get("rccl_version", "?"), ",".join(rt.get("libamdhip64", ["?"]))))
    return 0


def run(rank, world, addr, port, device, timeout=180.0, transport="rccl"):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    p = subprocess.Popen([sys.executable, "-m", "femus_amd.rccl_preflight", str(rank), str(world), addr, str(port), str(device), transport],
                         env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    try:
        out, _ = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        p.kill()                       # exactly the child started above
        p.communicate()
        return False, "RCCL preflight did not finish within %.0f s" % timeout
    if p.returncode == 0:
        tail = out.decode(errors="replace").strip().splitlines()[-1:] if out else []
        return True, "ok" + (" (" + tail[0][:200] + ")" if tail and tail[0].startswith("runtime:") else "")
    tail = out.decode(errors="replace").strip().splitlines()[-1:] if out else []
    return False, "RCCL preflight failed (exit %d)%s" % (p.returncode, ": " + tail[0][:160] if tail else "")


if __name__ == "__main__":
    a = sys.argv[1:]
    sys.exit(_child(int(a[0]), int(a[1]), a[2], int(a[3]), int(a[4]), a[5] if len(a) > 5 else "rccl"))
