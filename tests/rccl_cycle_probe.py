"""Probe (run by tests/test_gpu_halo_overlap.py in a child process with a time limit): a three-level multigrid cycle on a PERIODIC
1-D problem written as one rank's share of a distributed hierarchy -- the wrap-around neighbours are ghost entries that the rank
receives from ITSELF through RCCL (context option halo_self_rccl), the level below the coarsest distributed one is replicated and its
right-hand side goes through ncclAllReduce.  Every RCCL call of the distributed cycle therefore executes on the one GPU of the box, with
the interior / interface overlap on and off; the result must equal the serial numpy cycle.  (Capturing these exchanges into the
cycle's hipGraph was tried with this probe: librccl 2.26.6 segfaults during the capture, so distributed cycles stay un-captured.)
usage: python tests/rccl_cycle_probe.py  -> prints 'PROBE OK ...' """
import os
import sys

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import femus_amd
from femus_amd import capi
from oracle import femus_oracle as fo


def periodic(n, diag):
    i = np.arange(n)
    return sp.csr_matrix((np.concatenate([np.full(n, diag), -np.ones(n), -np.ones(n)]),
                          (np.concatenate([i, i, i]), np.concatenate([i, (i - 1) % n, (i + 1) % n]))), shape=(n, n))


def interp(nf):
    nc = nf // 2
    r, c, v = [], [], []
    for i in range(nf):
        if i % 2 == 0:
            r.append(i), c.append(i // 2), v.append(1.0)
        else:
            r += [i, i]
            c += [(i - 1) // 2, ((i + 1) // 2) % nc]
            v += [0.5, 0.5]
    return sp.csr_matrix((v, (r, c)), shape=(nf, nc))


def local_form(G):
    """global periodic (rows x cols) -> this rank's [owned | ghost] form: ghost 0 = last owned entry, ghost 1 = first owned entry of the
    column space; an entry (i, j) that wraps around (|i_scaled - j| large) reads the ghost"""
    G = G.tocoo()
    m, n = G.shape
    scale = n / m
    col = G.col.copy()
    wrap_lo = (G.row * scale - G.col) > n / 2            # row near the end, column near the start  -> ghost 1 (first entry)
    wrap_hi = (G.col - G.row * scale) > n / 2            # row near the start, column near the end  -> ghost 0 (last entry)
    col[wrap_hi] = n
    col[wrap_lo] = n + 1
    L = sp.csr_matrix((G.data, (G.row, col)), shape=(m, n + 2))
    L.sort_indices()
    return L


def main():
    n2 = 4096
    n1, n0 = n2 // 2, n2 // 4
    A2 = periodic(n2, 2.5)
    P2, P1 = interp(n2), interp(n1)
    A1 = (P2.T @ A2 @ P2).tocsr()
    A0 = (P1.T @ A1 @ P1).tocsr()

    class H:
        pass
    H.A, H.P = [A0, A1, A2], [None, P1, P2]
    rhs = fo.lcg_fill(n2, 5)
    ref = fo.vcycle(H, 2, rhs, omega=0.7, npre=2, npost=2)

    ctx = femus_amd.Context(0)
    ctx.set_option("halo_self_rccl", 1)
    uid = capi.Halo.unique_id()
    halos = []
    for n in (n1, n2):
        halos.append(capi.Halo(ctx, 0, 1, uid, [2], np.array([n - 1, 0], np.int32), [2], parent=halos[0] if halos else None))
    dA0 = ctx.matrix_scipy(A0)
    dA1, dA2 = ctx.matrix_scipy(local_form(A1)), ctx.matrix_scipy(local_form(A2))
    dP1, dR1 = ctx.matrix_scipy(P1), ctx.matrix_scipy(P1.T.tocsr())
    dP2, dR2 = ctx.matrix_scipy(local_form(P2)), ctx.matrix_scipy(local_form(P2.T.tocsr()))
    assert dA2.split_info(n2) [1] >= 1 and dP2.split_info(n1)[1] >= 1 and dR2.split_info(n2)[1] >= 1      # every operator reads ghosts
    out = []
    for overlap in (1, 0):
        ctx.set_option("halo_overlap", overlap)
        mg = capi.Multigrid(ctx, 3)
        mg.set_level(0, dA0, None, None, 0, 0.7, 1, 0)
        mg.set_level(1, dA1, dP1, dR1, 0, 0.7, 2, 2)
        mg.set_level_distributed(1, halos[0], True)
        mg.set_level(2, dA2, dP2, dR2, 0, 0.7, 2, 2)
        mg.set_level_distributed(2, halos[1], False)
        mg.setup()
        ghost = np.array([n2, n2 + 1], np.int32)
        b, x = ctx.vector(n2 + 2, n2, 0, ghost), ctx.vector(n2 + 2, n2, 0, ghost)
        b.upload(rhs)
        errs = []
        for rep in range(3):
            for h in halos:
                h.stats(reset=True)
            mg.vcycle(b, x)
            errs.append(np.linalg.norm(x.to_numpy() - ref) / np.linalg.norm(ref))
        counts = [h.stats()["updates"] for h in halos]            # V(2,2): 5 exchanges on either distributed level
        # what one grouped ncclSend/ncclRecv costs on this stack (tiny messages: software + launch latency, no link in a self exchange)
        import time
        ctx.set_option("halo_profile", 1)
        for h in halos:
            h.stats(reset=True)
        ctx.sync()
        t0 = time.perf_counter()
        for rep in range(50):
            mg.vcycle(b, x)
        ctx.sync()
        cyc_us = (time.perf_counter() - t0) / 50 * 1e6
        st = [h.stats() for h in halos]
        ctx.set_option("halo_profile", 0)
        per_exchange_us = sum(s_["exchange_ms"] for s_ in st) / max(1, sum(s_["updates"] for s_ in st)) * 1e3
        exposed_us = sum(s_["exposed_ms"] for s_ in st) / max(1, sum(s_["updates"] for s_ in st)) * 1e3
        out.append((overlap, max(errs), counts, {"cycle_us": round(cyc_us, 1), "rccl_group_us": round(per_exchange_us, 1), "exposed_us": round(exposed_us, 1)}))
        assert max(errs) < 1e-12, (overlap, errs)
        mg.destroy()
    assert out[0][2] == [5, 5] and out[1][2] == [5, 5], out
    print("PROBE OK", out)


if __name__ == "__main__":
    main()
