"""Probe (not a pytest test): BASELINE config 5 shape on several ranks at a larger size than the unit tests -- 3-D Poisson Q2, nb^3 coarse
elements per rank, n_uniform uniform + (nlevels - n_uniform) selectively refined levels (MGAMR/ex4-style flag), ranks sharing the one GPU
through the host-staged transport.  Checks the distributed GMRES solution against the single-GPU solver on the same global adaptive mesh.
usage: python tests/perf_probe_amr_dd.py [world=8] [nb=4] [nlevels=4] [n_uniform=2] [adaptive|uniform] [box|general]
"general": the same global box handed over as an ARBITRARY coarse mesh -- weighted native partition (finest-level descendants per coarse
element) instead of the equal boxes; the line reports the owned dofs / local elements per rank of both for the load balance."""
import json
import os
import socket
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


TRANSPORT = os.environ.get("FEMUS_DD_TRANSPORT", "host")                  # "rccl" on a box with one GPU per rank (tests/scale_first_run.sh)
PER_RANK = os.environ.get("FEMUS_DD_DEVICE_PER_RANK") == "1"


def device_of(rank):
    return rank if PER_RANK else 0


def flag(x, level):
    return x[0] > 0.5 and (level < 2 or x[1] > 0.25)


def worker(rank, world, port, nb, nlevels, n_uniform, out, uniform=False):
    import femus_amd as fa
    from femus_amd import dd
    comm = dd.SocketComm(rank, world, "127.0.0.1", port)
    ctx = fa.Context(device_of(rank))
    t0 = time.time()
    dp = dd.DistributedPoisson(ctx, comm, world, rank, nb=nb, nlevels=nlevels, transport=TRANSPORT, flag_fn=None if uniform else flag,
                               n_uniform=None if uniform else n_uniform)
    setup = time.time() - t0
    dp.assemble()
    dp.set_penalty_top()
    its, rn = dp.solve(outer="gmres", rtol=1e-12, maxit=80)
    top = dp.H.plans[-1]
    np.savez(out % rank, gid=top.gid[top.owned], x=dp.EPSC.to_numpy()[:dp.n_owned], its=its, adaptive=dp.adaptive, setup=setup,
             prepare_ms=dp.prepare_ms, n_owned=dp.n_owned)
    comm.barrier()
    comm.close()


def global_box(world, nb):
    from femus_amd import capi, dd
    p = dd.BoxPartition(world, 0).p
    return capi.Mesh.box(p[0] * nb, p[1] * nb, p[2] * nb, hi=tuple(float(v) for v in p)), p


def general_worker(rank, world, port, nb, nlevels, n_uniform, out):
    import femus_amd as fa
    from femus_amd import dd
    comm = dd.SocketComm(rank, world, "127.0.0.1", port)
    ctx = fa.Context(device_of(rank))
    G, _ = global_box(world, nb)
    t0 = time.time()
    dp = dd.DistributedPoisson(ctx, comm, world, rank, nlevels=nlevels, transport=TRANSPORT, coarse_mesh=G, flag_fn=flag, n_uniform=n_uniform)
    setup = time.time() - t0
    dp.assemble()
    dp.set_penalty_top()
    its, rn = dp.solve(outer="gmres", rtol=1e-12, maxit=80)
    top = dp.H.plans[-1]
    np.savez(out % rank, xy=dp.full.meshes[-1].arrays()[1][top.owned], x=dp.EPSC.to_numpy()[:dp.n_owned], its=its, setup=setup,
             prepare_ms=dp.prepare_ms, n_owned=dp.n_owned, nel_local=dp.nel_local, part=dp.partition, w=dp.elem_weights)
    comm.barrier()
    comm.close()


def main_general(world, nb, nlevels, n_uniform, port):
    import torch.multiprocessing as mp
    out = "/tmp/amr_ddg_rank%d.npz"
    t0 = time.time()
    mp.spawn(general_worker, args=(world, port, nb, nlevels, n_uniform, out), nprocs=world, join=True)
    t_dist = time.time() - t0
    import femus_amd as fa
    from femus_amd import dd
    from femus_amd.poisson import PoissonMG
    ctx = fa.Context(0)
    G, p = global_box(world, nb)
    ms = dd.refine_levels(G, nlevels, flag, n_uniform)
    pb = PoissonMG(ctx, 0, 0, 0, nlevels, meshes=ms).init()
    pb.assemble()
    pb.prepare()
    pb.mgsolve(outer="gmres", rtol=1e-13, maxit=80)
    xs = pb.EPS.to_numpy()
    xy = ms[-1].arrays()[1]
    key = lambda a: np.rint(a * 2 ** 20).astype(np.int64) @ np.array([1, 2 ** 21, 2 ** 42], dtype=np.int64)
    ks = key(xy)
    srt = np.argsort(ks)
    worst, seen, info = 0.0, 0, []
    for r in range(world):
        d = np.load(out % r)
        pos = srt[np.searchsorted(ks[srt], key(d["xy"]))]
        worst = max(worst, np.linalg.norm(d["x"] - xs[pos]) / np.linalg.norm(xs))
        seen += d["x"].size
        info.append({"rank": r, "owned": int(d["n_owned"]), "local_elements": int(d["nel_local"]), "gmres_its": int(d["its"]), "setup_s": float(d["setup"]),
                     "prepare_ms": float(d["prepare_ms"])})
    assert seen == xs.size
    d0 = np.load(out % 0)
    w, part = d0["w"], d0["part"]
    load = np.bincount(part, weights=w, minlength=world)
    # the equal boxes of the box split on the same coarse elements, for comparison
    xc = G.elem_centroids()
    box_rank = (np.floor(xc[:, 0]).astype(int) + p[0] * (np.floor(xc[:, 1]).astype(int) + p[1] * np.floor(xc[:, 2]).astype(int)))
    box_load = np.bincount(box_rank, weights=w, minlength=world)
    owned = np.array([i["owned"] for i in info], dtype=float)
    print(json.dumps({"config": "config-5 shape on a general partition: %d ranks, global box %dx%dx%d coarse elements handed over as an arbitrary coarse mesh, %d levels (%d uniform), Q2"
                                % (world, p[0] * nb, p[1] * nb, p[2] * nb, nlevels, n_uniform),
                      "partitioner": "fh_mesh_partition_weighted (weights = finest-level descendants of a coarse element)",
                      "dofs": int(xs.size), "finest_elements": int(ms[-1].nel), "hanging_top": int(pb.hanging[-1].size), "rel_diff_vs_single_gpu": worst,
                      "wall_s_distributed": t_dist,
                      "finest_elements_per_rank_weighted": [int(v) for v in load], "imbalance_weighted_max_over_mean": float(load.max() / load.mean()),
                      "finest_elements_per_rank_equal_boxes": [int(v) for v in box_load], "imbalance_equal_boxes_max_over_mean": float(box_load.max() / box_load.mean()),
                      "owned_dofs_max_over_min": float(owned.max() / owned.min()), "ranks": info}))
    assert worst < 1e-9


def main():
    import torch.multiprocessing as mp
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    nlevels = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    n_uniform = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    uniform = len(sys.argv) > 5 and sys.argv[5] == "uniform"          # BASELINE config 3 shape: no adaptive levels
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    if len(sys.argv) > 6 and sys.argv[6] == "general":
        return main_general(world, nb, nlevels, n_uniform, port)
    out = "/tmp/amr_dd_rank%d.npz"
    t0 = time.time()
    mp.spawn(worker, args=(world, port, nb, nlevels, n_uniform, out, uniform), nprocs=world, join=True)
    t_dist = time.time() - t0
    # single GPU on the global mesh: the replicated level of the distributed run is one more level below
    import femus_amd as fa
    from femus_amd import capi, dd
    from femus_amd.poisson import PoissonMG
    part = dd.BoxPartition(world, 0)
    p = part.p
    ctx = fa.Context(0)
    ms = [capi.Mesh.box(p[0] * nb // 2, p[1] * nb // 2, p[2] * nb // 2, hi=tuple(float(v) for v in p))]
    for l in range(1, nlevels + 1):
        if uniform or l < n_uniform + 1:
            ms.append(ms[-1].refine())
        else:
            ms.append(ms[-1].refine_flagged(ms[-1].flag_elements(lambda x, level: flag(x, level - 1))))
    pb = PoissonMG(ctx, 0, 0, 0, nlevels + 1, meshes=ms).init()
    pb.assemble()
    pb.prepare()
    pb.mgsolve(outer="gmres", rtol=1e-13, maxit=80)
    xs = pb.EPS.to_numpy()
    gid_ser, _ = dd.node_keys(ms[-1].arrays()[1], nlevels - 1, nb, part)
    srt = np.argsort(gid_ser)
    worst, seen, info = 0.0, 0, []
    for r in range(world):
        d = np.load(out % r)
        pos = srt[np.searchsorted(gid_ser[srt], d["gid"])]
        assert np.array_equal(gid_ser[pos], d["gid"])
        worst = max(worst, np.linalg.norm(d["x"] - xs[pos]) / np.linalg.norm(xs))
        seen += d["gid"].size
        info.append({"rank": r, "owned": int(d["n_owned"]), "adaptive": bool(d["adaptive"]), "gmres_its": int(d["its"]), "setup_s": float(d["setup"]),
                     "prepare_ms": float(d["prepare_ms"])})
    assert seen == xs.size
    print(json.dumps({"config": "config-%s shape: %d ranks (box %dx%dx%d), nb=%d, %d levels (%s uniform), Q2" % (("3" if uniform else "5", world) + p + (nb, nlevels, "all" if uniform else str(n_uniform))),
                      "dofs": int(xs.size), "hanging_top": int(pb.hanging[-1].size), "rel_diff_vs_single_gpu": worst, "wall_s_distributed": t_dist,
                      "ranks": info}))
    assert worst < 1e-9


if __name__ == "__main__":
    main()
