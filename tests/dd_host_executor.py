"""TEST INFRASTRUCTURE (not part of the product): a numpy / scipy executor of the distributed multigrid plan -- the row-restricted operators of one
rank on the host and the V-cycle over them with the ghost exchanges done through the test's communicator.  The world-size 2 / 4 / 8 gloo tests
(tests/test_dd_gloo.py) run the C planner (fh_dd_*, behind femus_amd.dd) through this executor against the serial oracle cycle; the product runs the
same plans on the device inside fh_mg_* and never imports this file."""
import numpy as np
import scipy.sparse as sp

from femus_amd import capi
from femus_amd.dd import node_keys, needed_columns, build_level_plans, HostHierarchy


def restrict(M, rows, newid_cols):
    """rows of M (old local ids) with columns renumbered to the [owned | ghost] order"""
    S = M.tocsr()[rows].tocoo()
    cols = newid_cols[S.col]
    keep = S.data != 0.0
    assert np.all(cols[keep] >= 0), "an owned row reads a node outside the halo"
    ok = cols >= 0
    R = sp.csr_matrix((S.data[ok], (S.row[ok], cols[ok])), shape=(len(rows), int(newid_cols.max()) + 1))
    R.sort_indices()
    return R



def build_host_hierarchy(part, comm, nb, meshes, A_full, P_full, bdc_full, rep):
    """A_full[l] (penalised), P_full[l] (Dirichlet-zeroed) on the extended box, local FEMuS numbering.
    rep = (mesh_rep, mesh_g0, P_g0) global replicated meshes/prolongator or None for a single rank."""
    nl = len(meshes)
    coords = [m.arrays()[1] for m in meshes]
    gids, owners = zip(*[node_keys(coords[l], l, nb, part) for l in range(nl)])
    need = needed_columns(A_full, P_full, owners, part.rank)
    plans = build_level_plans(part, comm, gids, owners, need)
    H = HostHierarchy()
    H.plans = plans
    H.A = [restrict(A_full[l], plans[l].owned, plans[l].newid) for l in range(nl)]
    H.P = [None] + [restrict(P_full[l], plans[l].owned, plans[l - 1].newid) for l in range(1, nl)]
    H.R = [None] + [restrict(P_full[l].T.tocsr(), plans[l - 1].owned, plans[l].newid) for l in range(1, nl)]
    for l in range(nl):   # shapes: pad columns to n_owned + n_ghost
        nloc = plans[l].n_owned + plans[l].n_ghost
        H.A[l] = _pad_cols(H.A[l], nloc)
        if l >= 1:
            H.P[l] = _pad_cols(H.P[l], plans[l - 1].n_owned + plans[l - 1].n_ghost)
            H.R[l] = _pad_cols(H.R[l], nloc)
    H.bdc_owned = [plans[l].newid[np.intersect1d(bdc_full[l], plans[l].owned)] for l in range(nl)]
    H.rep = None
    if rep is not None:
        mesh_rep, mesh_g0, P_g0, bdc_rep = rep
        # rows of the global level-0 prolongator for my local level-0 nodes, matched through the global ids
        g0_gid, _ = node_keys(mesh_g0.arrays()[1], 0, nb, part)
        srt = np.argsort(g0_gid)
        loc = plans[0]
        all_local = np.concatenate([loc.owned, loc.ghost])
        rows = srt[np.searchsorted(g0_gid[srt], loc.gid[all_local])]
        assert np.all(g0_gid[rows] == loc.gid[all_local])
        Pg_local = P_g0.tocsr()[rows]                                     # (n_owned + n_ghost) x n_rep, [owned | ghost] order
        P_rep = Pg_local[:loc.n_owned]
        T = (P_rep.T @ (H.A[0] @ Pg_local)).tocsr()                       # this rank's share of P^T A_0 P
        T.eliminate_zeros()
        n_rep = P_g0.shape[1]
        ed = mesh_rep.arrays()[0]
        rp, col = capi.pattern_from_elements(ed, n_rep)
        # sum the per-rank shares on the stencil pattern of the replicated mesh (common to all ranks)
        rowid = np.repeat(np.arange(n_rep, dtype=np.int64), np.diff(rp))
        pkey = rowid * n_rep + col
        Tc = T.tocoo()
        tkey = Tc.row.astype(np.int64) * n_rep + Tc.col
        pos = np.searchsorted(pkey, tkey)
        assert np.all(pkey[np.minimum(pos, pkey.size - 1)] == tkey), "replicated coarse operator leaves its stencil pattern"
        vals = np.zeros(pkey.size)
        vals[pos] = Tc.data
        vals = comm.allreduce_sum(vals)
        Tsum = sp.csr_matrix((vals, col, rp), shape=(n_rep, n_rep))
        # SetPenalty on the replicated level
        A_rep = Tsum.copy()
        for r in bdc_rep:
            A_rep.data[A_rep.indptr[r]:A_rep.indptr[r + 1]] = 0.0
        A_rep = A_rep + sp.csr_matrix((np.ones(len(bdc_rep)), (bdc_rep, bdc_rep)), shape=A_rep.shape)
        A_rep = A_rep.tocsr()
        A_rep.sort_indices()
        H.rep = {"A": A_rep, "P": P_rep.tocsr(), "R": P_rep.T.tocsr(), "n": n_rep}
    return H


def _pad_cols(M, ncols):
    M = M.tocsr()
    if M.shape[1] == ncols:
        return M
    return sp.csr_matrix((M.data, M.indices, M.indptr), shape=(M.shape[0], ncols))


# ---------------------------------------------------------------------------------------------------------------------
# numpy executor
# ---------------------------------------------------------------------------------------------------------------------
def halo_update(comm, plan, v):
    """v: [owned | ghost] array; refresh the ghost tail from the owners"""
    off = np.concatenate([[0], np.cumsum(plan.send_counts)])
    send = [v[plan.send_idx[off[r]:off[r + 1]]] for r in range(len(plan.send_counts))]
    got = comm.alltoallv(send, np.float64)
    v[plan.n_owned:] = np.concatenate(got) if plan.n_ghost else v[plan.n_owned:]
    return v


def vcycle_numpy(comm, H, b_owned, omega=2. / 3., npre=2, npost=2):
    nl = len(H.A)
    dinv = []
    for l in range(nl):
        d = H.A[l].diagonal()[:H.plans[l].n_owned].copy()
        d[d == 0] = 1.0
        dinv.append(1.0 / d)
    b = [None] * nl
    x = [None] * nl
    b[nl - 1] = b_owned
    lowest = 0
    for l in range(nl - 1, lowest - 1, -1):
        pl = H.plans[l]
        n0, nloc = pl.n_owned, pl.n_owned + pl.n_ghost
        xl = np.zeros(nloc)
        xl[:n0] = omega * dinv[l] * b[l]
        for _ in range(1, npre):
            halo_update(comm, pl, xl)
            xl[:n0] = xl[:n0] + omega * dinv[l] * (b[l] - H.A[l] @ xl)
        halo_update(comm, pl, xl)
        r = np.zeros(nloc)
        r[:n0] = b[l] - H.A[l] @ xl
        x[l] = xl
        if l > 0:
            halo_update(comm, pl, r)
            b[l - 1] = H.R[l] @ r
        else:
            brep = comm.allreduce_sum(H.rep["R"] @ r[:n0])
            import scipy.sparse.linalg as spla
            xrep = spla.spsolve(H.rep["A"].tocsc(), brep)
            xl[:n0] += H.rep["P"] @ xrep
    for l in range(lowest, nl):
        pl = H.plans[l]
        n0 = pl.n_owned
        xl = x[l]
        if l > 0:
            halo_update(comm, H.plans[l - 1], x[l - 1])
            xl[:n0] += H.P[l] @ x[l - 1]
        for _ in range(npost):
            halo_update(comm, pl, xl)
            xl[:n0] = xl[:n0] + omega * dinv[l] * (b[l] - H.A[l] @ xl)
    return x[nl - 1][:H.plans[nl - 1].n_owned].copy()




def stack_hierarchy(H, nv, nranks, scale=None):
    """the host hierarchy of `nv` stacked variables (femus_amd.dd.stack_plan / stack_matrix: the reference's rank-by-rank, variable-by-variable system
    numbering) from the hierarchy of one: block-diagonal operators and transfers (variable k's operator scaled by scale[k]), the replicated level stacked
    [var 0 | var 1 | ...] on every rank.  The executor above runs it unchanged."""
    from femus_amd.dd import stack_plan, stack_matrix
    nl = len(H.A)
    S = HostHierarchy()
    S.plans = [stack_plan(pl, nv, nranks) for pl in H.plans]
    coup = None if scale is None else [[(scale[k] if k == k2 else 0.0) for k2 in range(nv)] for k in range(nv)]
    S.A = [stack_matrix(H.A[l], S.plans[l], S.plans[l], nv, coup) for l in range(nl)]
    S.P = [None] + [stack_matrix(H.P[l], S.plans[l], S.plans[l - 1], nv) for l in range(1, nl)]
    S.R = [None] + [stack_matrix(H.R[l], S.plans[l - 1], S.plans[l], nv) for l in range(1, nl)]
    S.bdc_owned = [np.concatenate([H.bdc_owned[l] + k * H.plans[l].n_owned for k in range(nv)]) for l in range(nl)]
    S.rep = None
    if H.rep is not None:
        w = [1.0] * nv if scale is None else list(scale)
        S.rep = {"A": sp.block_diag([w[k] * H.rep["A"] for k in range(nv)]).tocsr(), "P": sp.block_diag([H.rep["P"]] * nv).tocsr(),
                 "R": sp.block_diag([H.rep["R"]] * nv).tocsr(), "n": nv * H.rep["n"]}
    return S
