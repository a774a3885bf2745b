"""N>1 path on CPU: world_size-2 gloo run of the domain-decomposition planner (femus_amd/dd.py) with the oracle's operators
and a numpy executor.  The distributed V-cycle must reproduce the serial oracle V-cycle on the global mesh."""
import os
import socket

import numpy as np
import pytest

from oracle import femus_oracle as fo

ONE = lambda xg: np.ones(xg.shape[:2])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def oracle_mesh(m, coarse=None):
    ed, xy, ff = m.arrays()
    om = fo.Mesh(m.geom, ed.astype(np.int64), xy, ff.astype(np.int64), level=m.level)
    om.own_size = list(m.own_size)
    return om


def full_local_operators(meshes, fe="biquadratic"):
    """the oracle's restatement of MGsolve preparation on one rank's extended box"""
    oms = [oracle_mesh(m) for m in meshes]
    for l in range(len(meshes) - 1):
        oms[l].child_elem = meshes[l].child_elems().astype(np.int64)
    bdc = [fo.dirichlet_dofs(om, fe) for om in oms]
    P = [None]
    for l in range(1, len(oms)):
        P.append(fo.zero_interpolator_dirichlet(fo.build_prolongator(oms[l - 1], oms[l], fe), bdc[l], bdc[l - 1]))
    A_raw, b = fo.assemble_poisson(oms[-1], fe, ONE)
    As = [None] * len(oms)
    As[-1] = A_raw
    for l in range(len(oms) - 1, 0, -1):
        As[l - 1] = (P[l].T @ As[l] @ P[l]).tocsr()
    A = [fo.zero_rows(As[l], bdc[l], 1.0) if l < len(oms) - 1 else fo.zero_rows_inplace_pattern(As[l], bdc[l], 1.0) for l in range(len(oms))]
    b = b.copy()
    b[bdc[-1]] = 0.0
    return oms, A, P, bdc, b


def _worker(rank, world, port, nb, nlevels, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from femus_amd import dd, capi
        import dd_host_executor as hx
        part = dd.BoxPartition(world, rank)
        comm = dd.TorchComm()
        meshes = dd.local_meshes(part, nb, nlevels)
        oms, A, P, bdc, b = full_local_operators(meshes)
        m_rep, m_g0 = dd.replicated_level(part, nb)
        om_rep, om_g0 = oracle_mesh(m_rep), oracle_mesh(m_g0)
        om_rep.child_elem = m_rep.child_elems().astype(np.int64)
        bdc_rep, bdc_g0 = fo.dirichlet_dofs(om_rep, "biquadratic"), fo.dirichlet_dofs(om_g0, "biquadratic")
        P_g0 = fo.zero_interpolator_dirichlet(fo.build_prolongator(om_rep, om_g0, "biquadratic"), bdc_g0, bdc_rep)
        H = hx.build_host_hierarchy(part, comm, nb, meshes, A, P, bdc, (m_rep, m_g0, P_g0, bdc_rep))
        top = H.plans[-1]
        # halo plan sanity: what I receive for a ghost is the owner's value of the same global node
        v = np.zeros(top.n_owned + top.n_ghost)
        v[:top.n_owned] = top.gid[top.owned].astype(np.float64)
        hx.halo_update(comm, top, v)
        assert np.array_equal(v[top.n_owned:], top.gid[top.ghost].astype(np.float64))
        # the reference's global numbering (fh_dd_plan_global): contiguous range per rank; a ghost's global index is the owner's
        # offset + position -- exchanging every rank's own global indices must reproduce the ghost list
        assert top.offsets[rank + 1] - top.offsets[rank] == top.n_owned and top.offsets[0] == 0
        v[:top.n_owned] = (top.offsets[rank] + np.arange(top.n_owned)).astype(np.float64)
        hx.halo_update(comm, top, v)
        assert np.array_equal(v[top.n_owned:], top.ghost_global.astype(np.float64))
        owner_of_ghost = np.searchsorted(top.offsets, top.ghost_global, side="right") - 1
        assert np.array_equal(owner_of_ghost, np.repeat(np.arange(world), top.recv_counts))
        x = hx.vcycle_numpy(comm, H, b[top.owned])
        np.savez(out % rank, gid=top.gid[top.owned], x=x, b=b[top.owned], n_ghost=top.n_ghost)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_distributed_vcycle_matches_serial_oracle(tmp_path, world):
    import torch.multiprocessing as mp
    nb, nlevels = 2, 2
    out = str(tmp_path / "rank%d.npz")
    mp.spawn(_worker, args=(world, _free_port(), nb, nlevels, out), nprocs=world, join=True)
    # serial reference: same global mesh with one more (exactly solved) level below
    from femus_amd import dd
    part = dd.BoxPartition(world, 0)
    p = part.p
    H = fo.build_poisson_hierarchy(p[0] * nb // 2, p[1] * nb // 2, p[2] * nb // 2, nlevels + 1, "biquadratic", ONE,
                                   hi=tuple(float(v) for v in p))
    ref = fo.vcycle(H, nlevels, H.b)
    gid_ser, _ = dd.node_keys(H.meshes[-1].coords, nlevels - 1, nb, part)
    srt = np.argsort(gid_ser)
    seen = 0
    for r in range(world):
        d = np.load(out % r)
        pos = srt[np.searchsorted(gid_ser[srt], d["gid"])]
        assert np.array_equal(gid_ser[pos], d["gid"])
        assert np.linalg.norm(d["b"] - H.b[pos]) <= 1e-13 * np.linalg.norm(H.b)          # same assembled residual
        assert np.linalg.norm(d["x"] - ref[pos]) <= 1e-11 * np.linalg.norm(ref)          # same cycle
        assert d["n_ghost"] > 0
        seen += d["gid"].size
    assert seen == ref.size                                                              # every global node owned exactly once


def _stacked_worker(rank, world, port, nb, nlevels, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from femus_amd import dd
        import dd_host_executor as hx
        part = dd.BoxPartition(world, rank)
        comm = dd.TorchComm()
        meshes = dd.local_meshes(part, nb, nlevels)
        oms, A, P, bdc, b = full_local_operators(meshes)
        m_rep, m_g0 = dd.replicated_level(part, nb)
        om_rep, om_g0 = oracle_mesh(m_rep), oracle_mesh(m_g0)
        om_rep.child_elem = m_rep.child_elems().astype(np.int64)
        bdc_rep, bdc_g0 = fo.dirichlet_dofs(om_rep, "biquadratic"), fo.dirichlet_dofs(om_g0, "biquadratic")
        P_g0 = fo.zero_interpolator_dirichlet(fo.build_prolongator(om_rep, om_g0, "biquadratic"), bdc_g0, bdc_rep)
        H = hx.build_host_hierarchy(part, comm, nb, meshes, A, P, bdc, (m_rep, m_g0, P_g0, bdc_rep))
        nv = 2
        S = hx.stack_hierarchy(H, nv, world, scale=[1.0, 0.5])
        top, stop = H.plans[-1], S.plans[-1]
        n0, ng = top.n_owned, top.n_ghost
        # the system numbering is the reference's (LinearEquation.cpp:212-237, restated loop for loop in the oracle): rank by rank, variable by variable
        KK, KKIndex = fo.system_offsets([list(top.offsets)] * nv)
        assert np.array_equal(np.array(KK), stop.kk_offset) and list(stop.kk_index) == KKIndex
        assert stop.offsets[rank + 1] - stop.offsets[rank] == nv * n0 and stop.offsets[-1] == nv * top.offsets[-1]
        for k in range(nv):          # GetSystemDof of every ghost of every variable (LinearEquation.cpp:76-85)
            for q in range(0, ng, max(1, ng // 7)):
                assert stop.ghost_global[stop.col_of[k][n0 + q] - nv * n0] == fo.system_dof([list(top.offsets)] * nv, KK, k, int(top.ghost_global[q]))[0]
        # the stacked exchange delivers, for every ghost, the owner's entry of the same system row
        v = np.zeros(stop.n_owned + stop.n_ghost)
        v[:stop.n_owned] = (stop.offsets[rank] + np.arange(stop.n_owned)).astype(np.float64)
        hx.halo_update(comm, stop, v)
        assert np.array_equal(v[stop.n_owned:], stop.ghost_global.astype(np.float64))
        # one V-cycle of the stacked system [A u = b, 0.5 A w = 2.5 b]
        bs = np.concatenate([b[top.owned], 2.5 * b[top.owned]])
        x = hx.vcycle_numpy(comm, S, bs)
        np.savez(out % rank, gid=top.gid[top.owned], x0=x[:n0], x1=x[n0:], n_ghost=stop.n_ghost)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_stacked_two_variable_system_through_the_decomposition(tmp_path, world):
    """two variables stacked as LinearEquation stacks them (KKoffset per rank, LinearEquation.cpp:212-237): the planner's exchange plan, operators and transfers
    of ONE variable carried to the stacked numbering (femus_amd.dd.stack_plan / stack_matrix), the distributed cycle run on the stacked system -- every
    variable's part equals the serial oracle cycle of its own equation"""
    import torch.multiprocessing as mp
    nb, nlevels = 2, 2
    out = str(tmp_path / "rank%d.npz")
    mp.spawn(_stacked_worker, args=(world, _free_port(), nb, nlevels, out), nprocs=world, join=True)
    from femus_amd import dd
    part = dd.BoxPartition(world, 0)
    p = part.p
    H = fo.build_poisson_hierarchy(p[0] * nb // 2, p[1] * nb // 2, p[2] * nb // 2, nlevels + 1, "biquadratic", ONE, hi=tuple(float(v) for v in p))
    ref = fo.vcycle(H, nlevels, H.b)                    # A u = b; the second variable solves 0.5 A w = 2.5 b: its cycle gives 5 times the first's
    gid_ser, _ = dd.node_keys(H.meshes[-1].coords, nlevels - 1, nb, part)
    srt = np.argsort(gid_ser)
    for r in range(world):
        d = np.load(out % r)
        pos = srt[np.searchsorted(gid_ser[srt], d["gid"])]
        assert np.linalg.norm(d["x0"] - ref[pos]) <= 1e-11 * np.linalg.norm(ref)
        assert np.linalg.norm(d["x1"] - 5.0 * ref[pos]) <= 1e-11 * np.linalg.norm(5.0 * ref)
        assert d["n_ghost"] > 0


def _sock_worker(rank, world, port, out):
    from femus_amd import dd
    comm = dd.SocketComm(rank, world, "127.0.0.1", port)
    got = comm.alltoallv([np.full(r + 1, rank * 10 + r, dtype=np.int64) for r in range(world)], np.int64)
    assert [g.tolist() for g in got] == [[r * 10 + rank] * (rank + 1) for r in range(world)]
    s = comm.allreduce_sum(np.arange(4.0) * (rank + 1))
    assert np.array_equal(s, np.arange(4.0) * sum(range(1, world + 1)))
    assert comm.bcast_obj(b"id" if rank == 0 else None) == b"id"
    assert comm.allreduce_max(float(rank)) == world - 1
    comm.barrier()
    comm.close()
    open(out % rank, "w").write("ok")


def test_socket_rendezvous_three_ranks(tmp_path):
    """the torch-free rendezvous bench.py uses for setup traffic (plans, ncclUniqueId, timing maxima)"""
    import torch.multiprocessing as mp
    out = str(tmp_path / "s%d")
    mp.spawn(_sock_worker, args=(3, _free_port(), out), nprocs=3, join=True)
    assert all(os.path.exists(out % r) for r in range(3))


# ---- adaptive levels on several ranks (BASELINE config "MGAMR ... 8xMI355X"): every rank refines its extended box with the same
# flag function on global coordinates; the planner works on node keys and needs nothing else
def amr_flag(x, level):
    return x[0] > 0.5 and (level < 2 or x[1] > 0.25)


def oracle_amr_meshes(meshes):
    oms = [oracle_mesh(m) for m in meshes]
    for l, m in enumerate(meshes):
        lev, hom = m.elem_levels()
        oms[l].elem_level, oms[l].homogeneous = lev.astype(np.int64), hom
        if l + 1 < len(meshes):
            ch = m.child_elems().astype(np.int64)
            oms[l].child_elem = ch
            oms[l].refined = ch[:, 1] >= 0
    return oms


def _amr_worker(rank, world, port, nb, nlevels, n_uniform, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from femus_amd import dd
        import dd_host_executor as hx
        from oracle import femus_oracle_amr as fa
        part = dd.BoxPartition(world, rank)
        comm = dd.TorchComm()
        meshes = dd.local_meshes(part, nb, nlevels, amr_flag, n_uniform)
        oms = oracle_amr_meshes(meshes)
        Hl = fa.build_amr_hierarchy(oms, "biquadratic", ONE)
        m_rep, m_g0 = dd.replicated_level(part, nb)
        om_rep, om_g0 = oracle_mesh(m_rep), oracle_mesh(m_g0)
        om_rep.child_elem = m_rep.child_elems().astype(np.int64)
        bdc_rep, bdc_g0 = fo.dirichlet_dofs(om_rep, "biquadratic"), fo.dirichlet_dofs(om_g0, "biquadratic")
        P_g0 = fo.zero_interpolator_dirichlet(fo.build_prolongator(om_rep, om_g0, "biquadratic"), bdc_g0, bdc_rep)
        H = hx.build_host_hierarchy(part, comm, nb, meshes, Hl.A, Hl.P, Hl.bdc, (m_rep, m_g0, P_g0, bdc_rep))
        top = H.plans[-1]
        x = hx.vcycle_numpy(comm, H, Hl.b[top.owned])
        np.savez(out % rank, gid=top.gid[top.owned], x=x, b=Hl.b[top.owned], n_ghost=top.n_ghost,
                 hanging=np.intersect1d(Hl.hanging[-1], top.owned).size)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_distributed_amr_vcycle_matches_serial_oracle(tmp_path, world):
    import torch.multiprocessing as mp
    from femus_amd import dd
    from oracle import femus_oracle_amr as fa
    nb, nlevels, n_uniform = 2, 3, 1
    out = str(tmp_path / "rank%d.npz")
    mp.spawn(_amr_worker, args=(world, _free_port(), nb, nlevels, n_uniform, out), nprocs=world, join=True)
    part = dd.BoxPartition(world, 0)
    p = part.p
    # serial reference: the global mesh with one more (exactly solved, uniform) level below
    ms = fa.build_amr_levels(p[0] * nb // 2, p[1] * nb // 2, p[2] * nb // 2, n_uniform + 1, nlevels - n_uniform, 
                             lambda x, level: amr_flag(x, level - 1), hi=tuple(float(v) for v in p))
    H = fa.build_amr_hierarchy(ms, "biquadratic", ONE)
    ref = fo.vcycle(H, nlevels, H.b)
    gid_ser, _ = dd.node_keys(ms[-1].coords, nlevels - 1, nb, part)
    srt = np.argsort(gid_ser)
    seen = hang = 0
    for r in range(world):
        d = np.load(out % r)
        pos = srt[np.searchsorted(gid_ser[srt], d["gid"])]
        assert np.array_equal(gid_ser[pos], d["gid"])
        assert np.linalg.norm(d["b"] - H.b[pos]) <= 1e-13 * np.linalg.norm(H.b)          # same projected residual
        assert np.linalg.norm(d["x"] - ref[pos]) <= 1e-11 * np.linalg.norm(ref)          # same cycle
        seen += d["gid"].size
        hang += int(d["hanging"])
    assert seen == ref.size and hang == H.hanging[-1].size and hang > 0


def test_system_numbering_of_several_variables_on_several_ranks():
    """a9 on several ranks: KKoffset / KKIndex / GetSystemDof through the C-ABI against the loop-for-loop restatement of
    LinearEquation.cpp:76-85, 212-237, plus what the numbering must satisfy (every system row hit once, ranks own contiguous ranges,
    variables contiguous inside a rank)"""
    from femus_amd import dd
    rng = np.random.default_rng(7)
    for nvars, nranks in ((1, 1), (3, 1), (1, 4), (3, 4), (4, 8)):
        sizes = rng.integers(0, 50, size=(nvars, nranks))           # empty ranks happen on coarse levels
        sizes[:, 0] += 1
        # Taylor-Hood style: variables of the same family share the offsets
        if nvars >= 3:
            sizes[1] = sizes[0]
        off = np.zeros((nvars, nranks + 1), dtype=np.int64)
        off[:, 1:] = np.cumsum(sizes, axis=1)
        kk, idx = dd.system_offsets(off)
        KK, KKIndex = fo.system_offsets(off.tolist())
        assert np.array_equal(kk, np.array(KK)) and np.array_equal(idx, np.array(KKIndex))
        seen = np.zeros(int(idx[-1]), dtype=np.int32)
        for var in range(nvars):
            ids = np.arange(off[var, -1], dtype=np.int64)
            rows, owner = dd.system_dofs(off, kk, var, ids)
            for i in rng.choice(ids.size, size=min(ids.size, 40), replace=False):
                r, p = fo.system_dof(off.tolist(), KK, var, int(ids[i]), iproc=int(rng.integers(nranks)))
                assert (r, p) == (rows[i], owner[i])
            seen[rows] += 1
            # rows of a rank's share of a variable are consecutive, inside the rank's range
            for p in range(nranks):
                mine = rows[owner == p]
                assert mine.size == sizes[var, p]
                if mine.size:
                    assert np.array_equal(mine, kk[var, p] + np.arange(mine.size)) and mine[-1] < kk[nvars, p]
        assert np.all(seen == 1)
    with pytest.raises(RuntimeError):
        dd.system_dofs(off, kk, 0, np.array([off[0, -1]], dtype=np.int64))       # one past the last dof


def test_planner_on_arbitrary_partitions():
    """the C planner takes any ownership map (what a METIS partition gives), not only boxes: R ranks simulated as threads of this process
    with a mailbox all-to-all; random owners, random local subsets in random local order, random need flags.  Checked: owned / ghost
    order, a simulated ghost exchange through the send lists delivers the owners' values, the global numbering is rank-contiguous and
    every ghost's global index is offset[owner] + its position in the owner's owned list"""
    import threading
    from types import SimpleNamespace
    from femus_amd import dd

    class ThreadComm:
        def __init__(self, rank, box):
            self.rank, self.box = rank, box

        def alltoallv(self, parts, dtype):
            R = len(parts)
            for r in range(R):
                self.box["mail"][r][self.rank] = np.array(parts[r], dtype=dtype, copy=True)
            self.box["barrier"].wait()
            got = [self.box["mail"][self.rank][r] for r in range(R)]
            self.box["barrier"].wait()
            return got

    for R, nglob, seed in ((2, 40, 1), (3, 500, 2), (5, 3000, 3), (8, 3000, 4), (4, 7, 5)):
        rng = np.random.default_rng(seed)
        owner_g = rng.integers(0, R, nglob)
        owner_g[:R] = np.arange(R) if nglob >= R else owner_g[:R]
        vals_g = rng.uniform(-1, 1, nglob)
        local, need = [], []
        for r in range(R):
            mine = np.flatnonzero(owner_g == r)
            others = np.flatnonzero(owner_g != r)
            extra = others[rng.uniform(size=others.size) < 0.3]
            ids = rng.permutation(np.concatenate([mine, extra]))
            local.append(ids)
            nd = (rng.uniform(size=ids.size) < 0.6) | (owner_g[ids] == r)
            need.append(nd.astype(np.uint8))
        box = {"mail": [[None] * R for _ in range(R)], "barrier": threading.Barrier(R)}
        plans, errs = [None] * R, []

        def work(r):
            try:
                part = SimpleNamespace(rank=r, nranks=R)
                plans[r] = dd.build_level_plans(part, ThreadComm(r, box), [local[r].astype(np.int64)], [owner_g[local[r]]], [need[r]])[0]
            except Exception as e:          # a failing rank must not leave the others at the barrier
                errs.append(repr(e))
                box["barrier"].abort()

        ts = [threading.Thread(target=work, args=(r,)) for r in range(R)]
        [t.start() for t in ts]
        [t.join(120) for t in ts]
        assert not errs and all(p is not None for p in plans), errs
        owned_gid = [local[r][plans[r].owned] for r in range(R)]
        for r in range(R):
            P = plans[r]
            assert np.array_equal(np.sort(P.owned), P.owned) and np.all(owner_g[owned_gid[r]] == r)          # ascending local id
            assert set(owned_gid[r].tolist()) == set(np.flatnonzero(owner_g == r).tolist())                  # every owned node is local here
            gg = local[r][P.ghost]
            want = np.flatnonzero((owner_g[local[r]] != r) & (need[r] == 1))
            assert set(P.ghost.tolist()) == set(want.tolist())
            key = owner_g[gg] * (nglob + 1) + gg
            assert np.all(np.diff(key) > 0)                                                                   # by owner, then global id
            assert np.array_equal(P.recv_counts, np.bincount(owner_g[gg], minlength=R))
            assert np.array_equal(P.offsets, np.concatenate([[0], np.cumsum([o.size for o in owned_gid])]))
            pos = {int(g): k for rr in range(R) for k, g in enumerate(owned_gid[rr])}
            assert np.array_equal(P.ghost_global, [P.offsets[owner_g[g]] + pos[int(g)] for g in gg])
        # the exchange: rank s sends values[owned][send_idx] cut by send_counts; rank r lays the segments out by source rank
        for r in range(R):
            got = []
            for s in range(R):
                Ps = plans[s]
                so = np.concatenate([[0], np.cumsum(Ps.send_counts)])
                seg = vals_g[owned_gid[s]][Ps.send_idx[so[r]:so[r + 1]]]
                assert seg.size == plans[r].recv_counts[s]
                got.append(seg)
            assert np.array_equal(np.concatenate(got) if got else np.zeros(0), vals_g[local[r][plans[r].ghost]])
