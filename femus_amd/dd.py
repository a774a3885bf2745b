"""Mesh-level domain decomposition for the multigrid hot path: one box partition per GPU.

The reference partitions the COARSE mesh (METIS, `MeshMetisPartitioning.cpp:71-155`; children inherit), renumbers so every
rank owns a contiguous row range with the lowest rank owning shared nodes, keeps ghost lists (`Mesh.cpp:767-795`,
`LinearEquation.cpp:239-280`) and lets PETSc exchange ghost values inside MatMult / VecGhostUpdate.  METIS is not available
(and uses a random seed), so the equivalent box split is used: rank (cx,cy,cz) owns an nb^3 block of coarse elements.

Each rank builds, with the serial mesh code, its block plus one coarse-element ghost layer on every interior side.  All
rows that a rank OWNS are complete inside that extended box on every level (coarse basis functions vanish on the outer
ring), so the full local Galerkin chain gives the exact owned rows of every level operator without any communication at
setup.  Per level the plan holds: owned nodes (local FEMuS order), ghost nodes grouped by owner rank, the send lists, and
the [owned | ghost] renumbering used by the row-restricted operators.  One extra, replicated level below the rank-local
coarse level (global coarse mesh coarsened once, <= 4913 dofs) replaces the single-GPU exact coarse solve.

The planner itself is C (fh_dd_box_node_keys, fh_dd_plan_create: host-only entry points of the C-ABI, so a C++ FEMuS build reaches the
same code); it runs and is tested without a GPU.  This module drives it and holds the numpy executor used by the CPU (gloo) tests.
"""
import numpy as np
import scipy.sparse as sp

from . import capi

GRIDS = {1: (1, 1, 1), 2: (2, 1, 1), 4: (2, 2, 1), 8: (2, 2, 2)}


class BoxPartition:
    def __init__(self, nranks, rank):
        self.p = GRIDS[nranks]
        self.nranks, self.rank = nranks, rank
        px, py, pz = self.p
        self.c = (rank % px, (rank // px) % py, rank // (px * py))

    def rank_of(self, c0, c1, c2):
        return c0 + self.p[0] * (c1 + self.p[1] * c2)


class TorchComm:
    """all-to-all of int64/float64 numpy arrays over a torch.distributed group (gloo on the host)"""

    @classmethod
    def from_env(cls, timeout_s=120):
        """gloo group from the RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT that torchrun exports"""
        import datetime
        import torch.distributed as dist
        if not dist.is_initialized():
            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=timeout_s))
        return cls()

    def __init__(self, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.rank, self.size = dist.get_rank(), dist.get_world_size()

    def alltoallv(self, arrays, dtype):
        """personalised all-to-all built from point-to-point messages (gloo has no alltoall)"""
        t, d = self.torch, self.dist
        counts = t.tensor([a.size for a in arrays], dtype=t.int64)
        allc = [t.zeros(self.size, dtype=t.int64) for _ in range(self.size)]
        d.all_gather(allc, counts, group=self.group)
        send = [t.from_numpy(np.ascontiguousarray(a, dtype=dtype).copy()) for a in arrays]
        recv = [t.empty(int(allc[r][self.rank]), dtype=send[0].dtype) for r in range(self.size)]
        reqs = []
        for r in range(self.size):
            if r == self.rank:
                recv[r].copy_(send[r])
                continue
            if send[r].numel():
                reqs.append(d.isend(send[r], r, group=self.group))
            if recv[r].numel():
                reqs.append(d.irecv(recv[r], r, group=self.group))
        for q in reqs:
            q.wait()
        return [r.numpy() for r in recv]

    def allreduce_sum(self, a):
        x = self.torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).copy())
        self.dist.all_reduce(x, group=self.group)
        return x.numpy()


def local_meshes(part, nb, nlevels, flag_fn=None, n_uniform=None):
    """extended local box (block + ghost ring) as a FEMuS-numbered mesh hierarchy with exact global coordinates;
    the domain is [0,px]x[0,py]x[0,pz], every rank's block is a unit cube of nb^3 coarse elements.
    flag_fn(x[3], level) with n_uniform < nlevels: levels n_uniform.. are refined selectively (MultiLevelMesh::RefineMesh(n, n_uniform,
    SetRefinementFlag)); the function is evaluated on GLOBAL coordinates, so every rank flags the same elements"""
    ext_lo = [1 if part.c[d] > 0 else 0 for d in range(3)]
    ext_hi = [1 if part.c[d] < part.p[d] - 1 else 0 for d in range(3)]
    n = [nb + ext_lo[d] + ext_hi[d] for d in range(3)]
    m = capi.Mesh.box(n[0], n[1], n[2])
    _, xy, _ = m.arrays()
    idx = np.rint(xy * (2.0 * np.array(n))).astype(np.int64)              # lexicographic half-element index
    origin = np.array([part.c[d] * nb - ext_lo[d] for d in range(3)], dtype=np.float64) / nb
    m.set_coords(origin[None, :] + idx * (0.5 / nb))                      # exact dyadic coordinates
    # local faces: 0 y-, 1 x+, 2 y+, 3 x-, 4 z-, 5 z+ ; artificial cuts are not boundary
    mask = 0
    for d, (flo, fhi) in enumerate(((3, 1), (0, 2), (4, 5))):
        if ext_lo[d]:
            mask |= 1 << flo
        if ext_hi[d]:
            mask |= 1 << fhi
    m.clear_boundary_faces(mask)
    ms = [m]
    for l in range(1, nlevels):
        if flag_fn is None or n_uniform is None or l < n_uniform:
            ms.append(ms[-1].refine())
        else:
            ms.append(ms[-1].refine_flagged(ms[-1].flag_elements(flag_fn)))
    return ms


def refine_levels(m0, nlevels, flag_fn=None, n_uniform=None):
    """[m0, ...]: nlevels - 1 refinements of m0, uniform below n_uniform, selective (flag_fn on the coordinates the mesh carries) from there on"""
    ms = [m0]
    for l in range(1, nlevels):
        if flag_fn is None or n_uniform is None or l < n_uniform:
            ms.append(ms[-1].refine())
        else:
            ms.append(ms[-1].refine_flagged(ms[-1].flag_elements(flag_fn)))
    return ms


def amr_slice_weights(G, lo, hi, nlevels, flag_fn, n_uniform):
    """number of finest-level descendants of the coarse elements lo .. hi-1 of G after nlevels - 1 refinements (uniform below n_uniform,
    selective from there on).  A flag is decided from the element's own vertices, so the slice is refined in isolation"""
    if hi <= lo:
        return np.zeros(0, dtype=np.int64)
    sub, _ = G.submesh(np.arange(lo, hi, dtype=np.int32))
    ms = refine_levels(sub, nlevels, flag_fn, n_uniform)
    anc = np.arange(sub.nel, dtype=np.int64)                      # coarse ancestor (position in the slice) of every element of the level
    for l in range(nlevels - 1):
        ch = ms[l].child_elems()                                  # -1 beyond the first entry of an element that was copied, not refined
        nxt = np.empty(ms[l + 1].nel, dtype=np.int64)
        ok = ch >= 0
        nxt[ch[ok]] = np.broadcast_to(anc[:, None], ch.shape)[ok]
        anc = nxt
    w = np.bincount(anc, minlength=sub.nel).astype(np.int64)
    for m in ms:
        m.destroy()
    return w


def amr_element_weights(G, comm, nranks, rank, nlevels, flag_fn, n_uniform):
    """work below every element of the coarse mesh G after the adaptive refinement (finest-level descendants): every rank refines a
    contiguous slice of the coarse elements (1 / nranks of the job, the size of its later share) and the counts are gathered -- no rank
    ever holds the refined global mesh.  Input of the weighted partition that stands for the reference's re-partition of adaptive levels."""
    mine = amr_slice_weights(G, (G.nel * rank) // nranks, (G.nel * (rank + 1)) // nranks, nlevels, flag_fn, n_uniform)
    parts = comm.allgather_obj(mine)
    w = np.concatenate([np.asarray(p_, dtype=np.int64) for p_ in parts]).astype(np.float64)
    assert w.size == G.nel and w.min() >= 1
    return w


def node_keys(coords, level, nb, part):
    """global integer grid index of every node of a level-`level` mesh and the owner rank (fh_dd_box_node_keys)"""
    import ctypes
    L = capi.load_library()
    xy = np.ascontiguousarray(coords, dtype=np.float64)
    if xy.shape[1] < 3:
        xy = np.ascontiguousarray(np.hstack([xy, np.zeros((xy.shape[0], 3 - xy.shape[1]))]))
    n = xy.shape[0]
    gid, owner = np.empty(n, dtype=np.int64), np.empty(n, dtype=np.int32)
    p = np.array(part.p, dtype=np.int32)
    capi._chk(L.fh_dd_box_node_keys(n, capi._p(xy), int(level), int(nb), capi._p(p), capi._p(gid), capi._p(owner)))
    return gid, owner.astype(np.int64)


def system_offsets(dof_offset):
    """KKoffset [nvars+1][nranks] and KKIndex [nvars+1] of a multi-variable system from the per-variable dof offsets [nvars][nranks+1]
    (fh_dd_system_offsets = LinearEquation::InitPde, LinearEquation.cpp:212-237)"""
    L = capi.load_library()
    off = np.ascontiguousarray(dof_offset, dtype=np.int64)
    nvars, nranks = off.shape[0], off.shape[1] - 1
    kk = np.zeros((nvars + 1, nranks), dtype=np.int64)
    idx = np.zeros(nvars + 1, dtype=np.int64)
    capi._chk(L.fh_dd_system_offsets(nvars, nranks, capi._p(off), capi._p(kk), capi._p(idx)))
    return kk, idx


def system_dofs(dof_offset, kk_offset, var, idof):
    """system rows and owner ranks of the mesh dofs `idof` of variable `var` (fh_dd_system_dofs = LinearEquation::GetSystemDof)"""
    L = capi.load_library()
    off = np.ascontiguousarray(dof_offset, dtype=np.int64)
    kk = np.ascontiguousarray(kk_offset, dtype=np.int64)
    ids = np.ascontiguousarray(idof, dtype=np.int64)
    rows, owner = np.empty(ids.size, dtype=np.int64), np.empty(ids.size, dtype=np.int32)
    capi._chk(L.fh_dd_system_dofs(off.shape[0], off.shape[1] - 1, capi._p(off), capi._p(kk), int(var), ids.size, capi._p(ids), capi._p(rows),
                                  capi._p(owner)))
    return rows, owner


class LevelPlan:
    pass


def build_level_plans(part, comm, gids, owners, need_masks):
    """gids/owners/need_masks: per level arrays over the local (extended box) nodes.  Returns LevelPlan per level.  The plan itself is
    built by the C-ABI planner (fh_dd_plan_create, femus_amd/csrc/fh_dd.cpp) -- the same entry point a C++ FEMuS build calls; `comm`
    only carries its one all-to-all of node ids"""
    import ctypes
    L = capi.load_library()
    nr = part.nranks
    CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int),
                          ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int))

    def alltoallv(user, send, scnt, recv, rcnt):
        try:
            if not send:                                   # first call: exchange the counts
                got = comm.alltoallv([np.array([scnt[r]], dtype=np.int64) for r in range(nr)], np.int64)
                for r in range(nr):
                    rcnt[r] = int(got[r][0])
                return 0
            parts, off = [], 0
            for r in range(nr):
                parts.append(np.ctypeslib.as_array(send, shape=(off + scnt[r],))[off:off + scnt[r]].copy() if scnt[r] else np.zeros(0, np.int64))
                off += scnt[r]
            got = comm.alltoallv(parts, np.int64)
            off = 0
            for r in range(nr):
                if rcnt[r]:
                    assert got[r].size == rcnt[r]
                    np.ctypeslib.as_array(recv, shape=(off + rcnt[r],))[off:off + rcnt[r]] = got[r]
                off += rcnt[r]
            return 0
        except Exception:       # never let an exception cross the C boundary
            import traceback
            traceback.print_exc()
            return 1

    cb = CB(alltoallv)
    plans = []
    for gid, owner, need in zip(gids, owners, need_masks):
        g = np.ascontiguousarray(gid, dtype=np.int64)
        o = np.ascontiguousarray(owner, dtype=np.int32)
        nd = np.ascontiguousarray(need, dtype=np.uint8)
        h = ctypes.c_void_p()
        capi._chk(L.fh_dd_plan_create(int(part.rank), int(nr), int(g.size), capi._p(g), capi._p(o), capi._p(nd), cb, None, ctypes.byref(h)))
        a, b_, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        capi._chk(L.fh_dd_plan_sizes(h, ctypes.byref(a), ctypes.byref(b_), ctypes.byref(c)))
        P = LevelPlan()
        P.n_owned, P.n_ghost = a.value, b_.value
        owned, ghost = np.empty(a.value, np.int32), np.empty(b_.value, np.int32)
        newid, sidx = np.empty(g.size, np.int32), np.empty(c.value, np.int32)
        sc, rc = np.empty(nr, np.int32), np.empty(nr, np.int32)
        capi._chk(L.fh_dd_plan_get(h, capi._p(owned), capi._p(ghost), capi._p(newid), capi._p(sc), capi._p(sidx), capi._p(rc)))
        offs, gglob = np.empty(nr + 1, np.int64), np.empty(max(b_.value, 1), np.int64)
        capi._chk(L.fh_dd_plan_global(h, capi._p(offs), capi._p(gglob)))
        P.offsets, P.ghost_global = offs, gglob[:b_.value]          # the reference's contiguous global ranges / ghost global ids
        capi._chk(L.fh_dd_plan_destroy(h))
        P.owned, P.ghost = owned.astype(np.int64), ghost.astype(np.int64)
        P.newid = newid.astype(np.int64)
        P.send_idx, P.send_counts, P.recv_counts = sidx, sc, rc
        P.gid = gid
        plans.append(P)
    return plans


def stack_plan(plan, nv, nranks):
    """The exchange plan of `nv` variables of one family stacked as the reference stacks a system -- rank by rank, variable by variable inside a rank
    (KKoffset, LinearEquation.cpp:212-237; fh_dd_system_offsets / fh_dd_system_dofs) -- from the plan of ONE variable.  Local layout of a stacked vector:
    owned = [var 0 | var 1 | ...] (n_owned each), ghosts grouped by source rank and, inside a rank, by variable: exactly what the sender's stacked owned
    block delivers, so the plan is again {send_counts, send_idx, recv_counts} and runs through the same fh_halo_* objects.  Returns a LevelPlan with, in
    addition, col_of[k] = stacked local index of every scalar local index ([owned | ghost] numbering) of variable k."""
    S = LevelPlan()
    n0, ng = plan.n_owned, plan.n_ghost
    sc, rc = np.asarray(plan.send_counts, dtype=np.int64), np.asarray(plan.recv_counts, dtype=np.int64)
    soff, goff = np.concatenate([[0], np.cumsum(sc)]), np.concatenate([[0], np.cumsum(rc)])
    S.nv, S.base = nv, plan
    S.n_owned, S.n_ghost = nv * n0, nv * ng
    S.send_counts, S.recv_counts = (sc * nv).astype(np.int32), (rc * nv).astype(np.int32)
    S.send_idx = np.concatenate([np.asarray(plan.send_idx[soff[r]:soff[r + 1]], dtype=np.int64) + k * n0 for r in range(nranks) for k in range(nv)]
                                + [np.zeros(0, np.int64)]).astype(np.int32)
    src = np.repeat(np.arange(nranks), rc)                              # source rank of every scalar ghost
    q = np.arange(ng, dtype=np.int64)
    S.col_of = []
    for k in range(nv):
        c = np.empty(n0 + ng, dtype=np.int64)
        c[:n0] = k * n0 + np.arange(n0)
        c[n0:] = nv * n0 + nv * goff[src] + k * rc[src] + (q - goff[src])
        S.col_of.append(c)
    # the reference's global system numbering: every variable has the dof offsets of the one plan
    dof_offset = np.tile(np.asarray(plan.offsets, dtype=np.int64), (nv, 1))
    kk, idx = system_offsets(dof_offset)
    S.kk_offset, S.kk_index = kk, idx
    S.offsets = np.concatenate([kk[0], [kk[nv][nranks - 1]]]).astype(np.int64)            # rank r owns the system rows [KKoffset[0][r], KKoffset[nv][r])
    gg = np.empty(nv * ng, dtype=np.int64)
    for k in range(nv):
        rows, owner = system_dofs(dof_offset, kk, k, plan.ghost_global)
        assert np.array_equal(owner, src)
        gg[S.col_of[k][n0:] - nv * n0] = rows
    S.ghost_global = gg
    return S


def stack_matrix(M, row_plan, col_plan, nv, coupling=None):
    """block matrix of a stacked system in the stacked local numbering from the scalar owned-rows matrix M (scipy CSR, row_plan.n_owned x (col_plan.n_owned +
    col_plan.n_ghost)): block (k, k2) = coupling[k][k2] * M (default: the identity -- nv uncoupled copies)"""
    import scipy.sparse as sp
    C = M.tocoo()
    n0r = row_plan.base.n_owned
    rows, cols, vals = [], [], []
    for k in range(nv):
        for k2 in range(nv):
            w = (1.0 if k == k2 else 0.0) if coupling is None else float(coupling[k][k2])
            if w == 0.0:
                continue
            rows.append(C.row + k * n0r)
            cols.append(col_plan.col_of[k2][C.col])
            vals.append(w * C.data)
    S = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(nv * n0r, col_plan.n_owned + col_plan.n_ghost))
    S.sort_indices()
    return S


def needed_columns(A_full, P_full, owners, me):
    """per level: the non-owned local nodes whose values the owned rows of A_l, P_{l+1}, R_l read"""
    nl = len(A_full)
    need = [np.zeros(a.shape[0], dtype=bool) for a in A_full]
    for l in range(nl):
        own = owners[l] == me
        need[l][np.unique(A_full[l][own].indices)] = True
        if l >= 1:
            Pl = P_full[l].tocsr()
            need[l - 1][np.unique(Pl[own].indices)] = True                # interpolation reads coarse ghosts
            own_c = owners[l - 1] == me
            rows = np.unique(Pl.tocsc()[:, own_c].indices)                # restriction rows of owned coarse nodes read these
            need[l][rows] = True
    return need


class HostHierarchy:
    """holder of one rank's row-restricted DEVICE operators in the [owned | ghost] numbering + the replicated level below (filled by DistributedPoisson)"""
    pass


def replicated_level(part, nb):
    """global level -1 / level 0 meshes and the prolongator between them (built identically on every rank; small)"""
    n = [part.p[d] * nb // 2 for d in range(3)]
    m_rep = capi.Mesh.box(n[0], n[1], n[2], lo=(0., 0., 0.), hi=tuple(float(v) for v in part.p))
    m_g0 = m_rep.refine()
    return m_rep, m_g0


# ---------------------------------------------------------------------------------------------------------------------
# rendezvous without torch: tiny star-topology collectives over TCP for SETUP traffic only (plans, ids, timing maxima);
# the data path of a cycle is RCCL (fh_halo_*).  Reads RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT like torchrun sets them.
# ---------------------------------------------------------------------------------------------------------------------
# wire format of SocketComm: a closed set of plain values, nothing executable (never pickle: the port is reachable by others)
_WIRE_DTYPES = ("<i4", "<i8", "<f8", "<u1", "|u1", "|b1", "<u2", "<f4")


def wire_encode(obj, out=None):
    import struct
    top = out is None
    if top:
        out = []
    if obj is None:
        out.append(b"N")
    elif isinstance(obj, (bool, np.bool_)):
        out.append(b"T" if obj else b"F")
    elif isinstance(obj, (int, np.integer)):
        out.append(b"I" + struct.pack("<q", int(obj)))
    elif isinstance(obj, (float, np.floating)):
        out.append(b"D" + struct.pack("<d", float(obj)))
    elif isinstance(obj, str):
        raw = obj.encode("utf-8")
        out.append(b"S" + struct.pack("<Q", len(raw)) + raw)
    elif isinstance(obj, (bytes, bytearray)):
        out.append(b"B" + struct.pack("<Q", len(obj)) + bytes(obj))
    elif isinstance(obj, (list, tuple)):
        out.append((b"L" if isinstance(obj, list) else b"U") + struct.pack("<Q", len(obj)))
        for o in obj:
            wire_encode(o, out)
    elif isinstance(obj, np.ndarray):
        a = np.ascontiguousarray(obj)
        ds = a.dtype.str
        if ds not in _WIRE_DTYPES:
            raise TypeError("SocketComm: array dtype %s is not part of the wire format" % ds)
        dsb = ds.encode()
        out.append(b"A" + struct.pack("<B", len(dsb)) + dsb + struct.pack("<B", a.ndim) + struct.pack("<%dq" % a.ndim, *a.shape))
        out.append(a.tobytes())
    else:
        raise TypeError("SocketComm: %s is not part of the wire format" % type(obj).__name__)
    return b"".join(out) if top else None


def wire_decode(buf):
    import struct
    mv = memoryview(buf)
    pos = [0]

    def take(n):
        if n < 0 or pos[0] + n > len(mv):
            raise ValueError("SocketComm: truncated or malformed message")
        v = mv[pos[0]:pos[0] + n]
        pos[0] += n
        return v

    def dec(depth):
        if depth > 8:
            raise ValueError("SocketComm: message nested too deeply")
        t = bytes(take(1))
        if t == b"N":
            return None
        if t in (b"T", b"F"):
            return t == b"T"
        if t == b"I":
            return struct.unpack("<q", take(8))[0]
        if t == b"D":
            return struct.unpack("<d", take(8))[0]
        if t in (b"S", b"B"):
            (n,) = struct.unpack("<Q", take(8))
            raw = bytes(take(n))
            return raw.decode("utf-8") if t == b"S" else raw
        if t in (b"L", b"U"):
            (n,) = struct.unpack("<Q", take(8))
            if n > len(mv):
                raise ValueError("SocketComm: malformed list length")
            items = [dec(depth + 1) for _ in range(n)]
            return items if t == b"L" else tuple(items)
        if t == b"A":
            (nd,) = struct.unpack("<B", take(1))
            ds = bytes(take(nd)).decode("ascii")
            if ds not in _WIRE_DTYPES:
                raise ValueError("SocketComm: array dtype not part of the wire format")
            (ndim,) = struct.unpack("<B", take(1))
            if ndim > 4:
                raise ValueError("SocketComm: malformed array rank")
            shape = struct.unpack("<%dq" % ndim, take(8 * ndim))
            if any(d < 0 for d in shape):
                raise ValueError("SocketComm: malformed array shape")
            dt = np.dtype(ds)
            count = int(np.prod(shape, dtype=np.int64)) if ndim else 1
            return np.frombuffer(take(count * dt.itemsize), dtype=dt).reshape(shape).copy()
        raise ValueError("SocketComm: unknown wire tag")

    obj = dec(0)
    if pos[0] != len(mv):
        raise ValueError("SocketComm: trailing bytes in message")
    return obj


def job_token():
    """shared secret of the ranks of ONE job for the rendezvous handshake: FEMUS_DD_TOKEN when the launcher provides one (a test
    parent, an MPI wrapper); otherwise derived from what torchrun hands every rank of the job.  The derived token keeps strangers
    from joining by accident; the protection against a hostile peer is the wire format above, which cannot carry code."""
    import hashlib
    import os
    tok = os.environ.get("FEMUS_DD_TOKEN")
    if tok:
        return hashlib.sha256(tok.encode()).digest()
    parts = [os.environ.get(k, "") for k in ("TORCHELASTIC_RUN_ID", "MASTER_ADDR", "MASTER_PORT", "WORLD_SIZE")] + [str(os.getuid())]
    return hashlib.sha256(("femus_hip_dd|" + "|".join(parts)).encode()).digest()


class SocketComm:
    MAGIC = b"femus_hip_dd_v2\0"          # 16 bytes
    MAX_MESSAGE = 1 << 32                 # 4 GiB: above any setup payload of the planner (index lists of one level), far below host memory

    def __init__(self, rank, size, addr="127.0.0.1", base_port=29500, timeout=300.0, token=None):
        import hmac
        import os
        import socket
        import struct
        import time
        self.rank, self.size = rank, size
        self._struct = struct
        self.conns = []
        if size == 1:
            return
        token = token if token is not None else job_token()
        mac = lambda *parts: hmac.new(token, b"".join(parts), "sha256").digest()
        ports = [base_port + 37 + k for k in range(8)]       # torchrun's own store sits on base_port

        def rd(c, n):
            buf = bytearray()
            while len(buf) < n:
                chunk = c.recv(n - len(buf))
                if not chunk:
                    raise OSError("peer closed the connection")
                buf += chunk
            return bytes(buf)

        if rank == 0:
            srv = None
            for p in ports:
                try:
                    srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                    srv.bind((addr, p))
                    break
                except OSError:
                    srv = None
            if srv is None:
                raise RuntimeError("SocketComm: no free rendezvous port near %d" % base_port)
            srv.listen(size)
            srv.settimeout(timeout)
            conns = {}
            while len(conns) < size - 1:
                c, _ = srv.accept()
                try:
                    # handshake on raw fixed-size frames BEFORE anything is decoded: nonce -> (magic, rank, HMAC(token, nonce|rank))
                    c.settimeout(10.0)
                    nonce = os.urandom(16)
                    c.sendall(nonce)
                    hello = rd(c, 16 + 4 + 32)
                    (r,) = struct.unpack("<i", hello[16:20])
                    if hello[:16] != self.MAGIC or not (1 <= r < size) or r in conns or not hmac.compare_digest(hello[20:], mac(nonce, hello[16:20])):
                        raise OSError("not a rank of this job")
                    c.sendall(mac(nonce, b"ack"))
                except Exception:                     # a stranger on this port must not stall or break the rendezvous
                    c.close()
                    continue
                c.settimeout(timeout)
                conns[r] = c
            self.conns = [conns[r] for r in range(1, size)]
            srv.close()
        else:
            t0 = time.time()
            sock = None
            while sock is None:
                for p in ports:
                    s = None
                    try:
                        s = socket.create_connection((addr, p), timeout=5.0)
                        s.settimeout(10.0)
                        nonce = rd(s, 16)
                        rb = struct.pack("<i", rank)
                        s.sendall(self.MAGIC + rb + mac(nonce, rb))
                        if not hmac.compare_digest(rd(s, 32), mac(nonce, b"ack")):      # only rank 0 of THIS job can answer
                            raise OSError("foreign listener")
                        s.settimeout(timeout)
                        sock = s
                        break
                    except Exception:
                        if s is not None:
                            s.close()
                        continue
                if sock is None:
                    if time.time() - t0 > timeout:
                        raise RuntimeError("SocketComm: cannot reach rank 0")
                    time.sleep(0.2)
            self.conns = [sock]

    def _send(self, c, obj):
        data = wire_encode(obj)
        c.sendall(self._struct.pack("<Q", len(data)) + data)

    def _recv(self, c):
        def rd(n):
            buf = bytearray()
            while len(buf) < n:
                chunk = c.recv(min(n - len(buf), 1 << 20))
                if not chunk:
                    raise RuntimeError("SocketComm: peer closed the connection")
                buf += chunk
            return buf
        (n,) = self._struct.unpack("<Q", bytes(rd(8)))
        if n > self.MAX_MESSAGE:
            raise RuntimeError("SocketComm: message of %d bytes exceeds the limit" % n)
        return wire_decode(rd(n))

    def allgather_obj(self, obj):
        if self.size == 1:
            return [obj]
        if self.rank == 0:
            objs = [obj] + [self._recv(c) for c in self.conns]
            for c in self.conns:
                self._send(c, objs)
            return objs
        self._send(self.conns[0], obj)
        return self._recv(self.conns[0])

    def barrier(self):
        self.allgather_obj(None)

    def bcast_obj(self, obj, root=0):
        return self.allgather_obj(obj if self.rank == root else None)[root]

    def alltoallv(self, arrays, dtype):
        everything = self.allgather_obj([np.ascontiguousarray(a, dtype=dtype) for a in arrays])
        return [everything[r][self.rank] for r in range(self.size)]

    def allreduce_sum(self, a):
        parts = self.allgather_obj(np.ascontiguousarray(a, dtype=np.float64))
        out = parts[0].copy()
        for p in parts[1:]:
            out += p
        return out

    def allreduce_max(self, v):
        return max(self.allgather_obj(float(v)))

    def close(self):
        for c in self.conns:
            try:
                c.close()
            except OSError:
                pass


# ---------------------------------------------------------------------------------------------------------------------
# device side: one rank's share of the distributed Poisson multigrid problem
# ---------------------------------------------------------------------------------------------------------------------
class DistributedPoisson:
    """weak-scaled config C3: every rank owns an nb^3-coarse-element block (64^3 fine elements for nb=8, 4 levels) of the
    global (px*nb, py*nb, pz*nb) box; operators hold owned rows over [owned | ghost] columns; ghosts are refreshed by
    fh_halo_* inside the cycle (overlapped with the rows that need no ghost); one replicated level below replaces the
    single-GPU exact coarse solve.

    Everything numeric stays on the device.  The rank's extended box (block + ghost ring) carries the complete local hierarchy
    (`self.full`, the one-GPU code: assembly, Galerkin chain, SetPenalty); the owned rows of every level operator are cut out of
    it by fh_mat_restrict (integer pattern work once, then one gather kernel per operator), the rank's share of the replicated
    operator is the triple product fh_mat_abc summed over the ranks on the device (fh_halo_allreduce_mat).  `prepare()` repeats
    the numeric part -- what LinearImplicitSystem::MGsolve does before every solve (LinearImplicitSystem.cpp:347-383)."""

    def __init__(self, ctx, comm, nranks, rank, nb=8, nlevels=4, omega=2. / 3., npre=2, npost=2, fe="biquadratic", order="seventh",
                 transport="rccl", flag_fn=None, n_uniform=None, source_kind=0, params=(1.0,), halo_comm=None, n_replicated=2,
                 coarse_mesh=None, partition=None):
        """flag_fn / n_uniform: adaptive levels (BASELINE config "MGAMR ... 8 GPUs"): every rank refines its extended box with the
        same flag function on global coordinates.  The fine level is then assembled AND projected (hanging nodes) on the extended
        box with the one-GPU code and the owned rows are gathered out on the device; uniform hierarchies keep the leaner
        owned-rows assembler for assemble()."""
        import time
        from .poisson import PoissonMG
        self.ctx, self.comm = ctx, comm
        self.transport, self.halo_comm = transport, halo_comm
        # coarse_mesh: ANY HEX27 / QUAD9 coarse mesh (a Gambit file, a box whose elements come in any order), the same object on every
        # rank.  Its elements are partitioned natively (fh_mesh_partition: the METIS_PartMeshDual of MeshMetisPartitioning.cpp:71-113;
        # `partition` overrides it), children inherit (:143-155); a rank's extended mesh = its elements + the ring of elements sharing a
        # node with them, numbered like any FEMuS mesh; global node ids / owners come from the refinement tree (fh_dd_topo_node_keys).
        # The coarse mesh itself is the replicated, exactly solved level of the cycle (there is no coarser mesh to go to).
        self.general = coarse_mesh is not None
        if self.general:
            assert nlevels >= 2 and fe == "biquadratic", "general partitions: Q2, at least two levels"
            assert flag_fn is None or (n_uniform is not None and n_uniform >= 2), "general partitions: level 1 is the first distributed level and stays uniform"

            class _Part:
                pass
            self.part = _Part()
            self.part.nranks, self.part.rank = nranks, rank
        else:
            self.part = BoxPartition(nranks, rank)
        self.nb, self.nl = nb, nlevels
        self.omega, self.npre, self.npost = omega, npre, npost
        self.source_kind, self.params = source_kind, params
        # replicated levels under the distributed ones: 1 = the global level below the coarsest local level (dense exact solve);
        # 2 = also the coarsest local level itself as a global, replicated, SMOOTHED level (every rank sweeps all of its few thousand
        # rows: no ghost exchange on that level, one wider all-reduce instead -- 6 exchanges per V(2,2) cycle less)
        self.n_replicated = 2 if (n_replicated >= 2 and nlevels >= 2) else 1
        part = self.part
        # 1. full local hierarchy on the extended box (device): assemble, Galerkin chain, SetPenalty
        if self.general:
            G = coarse_mesh
            self.n_replicated = 0
            # adaptive levels: the parts balance the finest-level descendants of the coarse elements (the reference re-partitions a refined
            # level with METIS, MeshMetisPartitioning.cpp:41-113 with AMR = true; here the children keep inheriting and the coarse partition
            # carries the weights), every rank flags with the same function on the coordinates of the mesh
            self.elem_weights = None
            if partition is None and flag_fn is not None:
                self.elem_weights = amr_element_weights(G, comm, nranks, rank, nlevels, flag_fn, n_uniform)
            self.partition = np.asarray(partition if partition is not None else G.partition(nranks, self.elem_weights), dtype=np.int32)
            own_e, ring_e = G.rank_elements(self.partition, rank)
            assert own_e.size > 0, "rank %d owns no coarse element" % rank
            els0 = np.concatenate([own_e, ring_e]).astype(np.int32)
            sub, node_gid = G.submesh(els0)
            meshes = refine_levels(sub, nlevels, flag_fn, n_uniform)
            self.partitioner = "native recursive bisection of the dual graph (fh_mesh_partition%s), %d coarse elements" % (
                "_weighted: finest-level descendants per coarse element" if self.elem_weights is not None else "", G.nel)
        else:
            meshes = local_meshes(part, nb, nlevels, flag_fn, n_uniform)
            self.partitioner = "box split %dx%dx%d" % tuple(part.p)
        self.adaptive = flag_fn is not None and not all(m.elem_levels()[1] for m in meshes)
        full = PoissonMG(ctx, 0, 0, 0, nlevels, fe=fe, order=order, omega=omega, npre=npre, npost=npost, meshes=meshes,
                         source_kind=source_kind, params=params)
        full.init()
        full.assemble()
        full.level_operators()
        self.full = full
        # 2. exchange plans (host, integers): owners from the global grid index of the nodes, halos from the operators' patterns
        coords = [m.arrays()[1] for m in meshes]
        if self.general:
            gids, owners = zip(*[coarse_mesh.topo_node_keys(self.partition, meshes, els0, l) for l in range(nlevels)])
        else:
            gids, owners = zip(*[node_keys(coords[l], l, nb, part) for l in range(nlevels)])
        own_rows = [np.where(owners[l] == rank)[0].astype(np.int32) for l in range(nlevels)]
        need = []
        for l in range(nlevels):
            mask = full.A[l].col_mask(own_rows[l])
            need.append(mask)
        for l in range(1, nlevels):
            full.P[l].col_mask(own_rows[l], need[l - 1])                  # interpolation reads coarse ghosts
            own_c = (owners[l - 1] == rank).astype(np.uint8)
            full.P[l].row_mask(own_c, need[l])                            # restriction rows of owned coarse nodes read these fine nodes
        plans = build_level_plans(part, comm, gids, owners, [m.astype(bool) for m in need])
        H = HostHierarchy()
        H.plans = plans
        self.H = H
        nloc = [pl.n_owned + pl.n_ghost for pl in plans]
        # 3. owned rows of every operator, cut out on the device; the maps re-gather the values at every preparation
        self.A, self.mapA = [], []
        for l in range(nlevels):
            a, m = full.A[l].restrict(plans[l].owned, plans[l].newid, nloc[l], check=True)
            self.A.append(a)
            self.mapA.append(m)
        self.P, self.R = [None], [None]
        for l in range(1, nlevels):
            p_, m = full.P[l].restrict(plans[l].owned, plans[l - 1].newid, nloc[l - 1])
            m.destroy()                                                   # geometric: values never change
            Pt = full.P[l].get_transpose()
            r_, m = Pt.restrict(plans[l - 1].owned, plans[l].newid, nloc[l])
            m.destroy()
            Pt.destroy()
            self.P.append(p_)
            self.R.append(r_)
        H.bdc_owned = [plans[l].newid[np.intersect1d(full.bdc[l], plans[l].owned)] for l in range(nlevels)]
        # 4. halos
        self.halos = []
        if transport == "host":   # host-staged exchange through `comm` (ranks sharing a GPU, launchers without RCCL peers)
            hc = halo_comm if halo_comm is not None else comm     # e.g. a gloo group for the data path, sockets for the setup
            for pl in plans:
                self.halos.append(capi.Halo.host(ctx, rank, nranks, hc, pl.send_counts, pl.send_idx, pl.recv_counts,
                                                 parent=self.halos[0] if self.halos else None))
        else:
            uid = comm.bcast_obj(capi.Halo.unique_id() if rank == 0 else None)
            for pl in plans:     # one RCCL communicator, one exchange plan per level
                self.halos.append(capi.Halo(ctx, rank, nranks, uid, pl.send_counts, pl.send_idx, pl.recv_counts,
                                            parent=self.halos[0] if self.halos else None))
        if self.general:
            # 5g. the coarse mesh as the replicated, exactly solved level: owned rows of every rank scattered into its stencil pattern and summed
            loc = plans[0]
            all_local = np.concatenate([loc.owned, loc.ghost])
            n_g0 = coarse_mesh.n_dofs(fe)
            gl_of_local = node_gid.astype(np.int64)                           # node of the extended mesh -> node of the coarse mesh
            grp, gcol = capi.pattern_from_elements(coarse_mesh.arrays()[0], n_g0)
            self.A_g0 = ctx.matrix_csr(n_g0, n_g0, grp, gcol)
            src_row = np.full(n_g0, -1, dtype=np.int32)
            src_row[gl_of_local[loc.owned]] = np.arange(loc.n_owned, dtype=np.int32)
            src_col = np.full(n_g0, -1, dtype=np.int32)
            src_col[gl_of_local[all_local]] = loc.newid[all_local]
            self.map_g0 = self.A_g0.value_map(self.A[0], src_row, src_col)
            p1, m = full.P[1].restrict(plans[1].owned, gl_of_local.astype(np.int32), n_g0)       # owned level-1 rows x coarse-mesh columns
            m.destroy()
            self.P1_rep = p1
            self.R1_rep = p1.get_transpose()
            self.bdc_rep = capi.Index(ctx, coarse_mesh.dirichlet_dofs(fe).astype(np.int32))
            self._replicated_operator()
        else:
            self._init_replicated_box(ctx, full, plans, gids, nloc, fe, nb, part)
        # 6. distributed fine-level assembler: elements touching an owned node, renumbered to [owned | ghost]
        top = plans[-1]
        ed, xy, _ = meshes[-1].arrays()
        own_mask = np.zeros(meshes[-1].nnode, dtype=bool)
        own_mask[top.owned] = True
        els = np.where(own_mask[ed].any(axis=1))[0]
        ntop = nloc[-1]
        if self.adaptive:
            # assemble() runs the extended-box assembly + hanging-node projection and gathers the owned rows
            self.asm = None
            self.map_rows = capi.Index(ctx, top.owned.astype(np.int32))
        else:
            ed_new = top.newid[ed[els]]
            assert ed_new.min() >= 0
            xy_new = np.zeros((ntop, 3))
            both = np.concatenate([top.owned, top.ghost])
            xy_new[top.newid[both]] = xy[both]
            self.asm = capi.Assembler(ctx, None, fe, self.A[-1], order, elem_dof=ed_new, coords=xy_new)
        self.n_owned, self.n_loc = top.n_owned, ntop
        # vectors in the reference's global numbering: this rank owns [offsets[rank], offsets[rank + 1]), ghosts carry the owners' global
        # indices (NumericVector::init(N, n_local, ghost, fast, GHOSTED); operator()(global index) reaches owned and ghost entries)
        assert int(top.offsets[-1]) < 2 ** 31, "global dof numbers beyond 32 bits (PetscInt is an int in the reference as well, PetscVector.hpp:536)"
        mk = lambda: ctx.vector(int(top.offsets[-1]), top.n_owned, int(top.offsets[rank]), top.ghost_global.astype(np.int32))
        self.RES, self.EPSC, self.SOL = mk(), mk(), mk()
        self.bdc_top = H.bdc_owned[-1].astype(np.int32)
        self.bdc_dev = capi.Index(ctx, self.bdc_top)       # BuildBdcIndex once, device-resident
        self.mg = capi.Multigrid(ctx, nlevels if self.general else nlevels + 1)
        cc = coarse_mesh.arrays()[1] if self.general else getattr(self, "coarse_coords", None)
        if cc is not None and cc.shape[0] == self.A_coarse.m():
            self.mg.set_coarse_coords(cc)           # the exact coarse solve dissects its dense problem with them (option coarse_nd)
        self._wire_cycle()
        self.mg.setup()
        self.ndof_owned = top.n_owned
        self.nel_local = els.size
        self.asm_top = self.asm
        self.prepare_first_s = None
        # one numeric re-preparation, timed: what every later MGsolve pays
        ctx.sync()
        t0 = time.time()
        self.prepare()
        ctx.sync()
        self.prepare_ms = (time.time() - t0) * 1e3

    def _init_replicated_box(self, ctx, full, plans, gids, nloc, fe, nb, part):
        """5. replicated level(s) below the box hierarchy: this rank's share of P^T A_0 P as a device triple product, summed over the ranks"""
        rank = part.rank
        m_rep, m_g0 = replicated_level(part, nb)
        self.coarse_coords = m_rep.arrays()[1]              # where the unknowns of the exactly solved level lie (fh_mg_set_coarse_coords)
        Pg = capi.build_prolongator(ctx, m_rep, m_g0, fe, zero_bdc=True)
        n_rep = m_rep.n_dofs(fe)
        g0_gid, _ = node_keys(m_g0.arrays()[1], 0, nb, part)
        srt = np.argsort(g0_gid)
        loc = plans[0]
        all_local = np.concatenate([loc.owned, loc.ghost])
        rows = srt[np.searchsorted(g0_gid[srt], loc.gid[all_local])]
        assert np.all(g0_gid[rows] == loc.gid[all_local])
        if self.n_replicated == 1:        # (with two replicated levels this level's operator is the Galerkin product A_rep2 below)
            ident = np.arange(n_rep, dtype=np.int32)
            self.Pg_local, m = Pg.restrict(rows, ident, n_rep)                # (n_owned + n_ghost) x n_rep, [owned | ghost] order
            m.destroy()
            self.P_rep, m = Pg.restrict(rows[:loc.n_owned], ident, n_rep)
            m.destroy()
            Pg.destroy()
            self.R_rep = self.P_rep.get_transpose()
            self.T_rep = capi.Mat.abc(self.R_rep, self.A[0], self.Pg_local)   # plan kept: numeric-only in prepare()
            rp, col = capi.pattern_from_elements(m_rep.arrays()[0], n_rep)    # stencil pattern of the replicated mesh (same on all ranks)
            self.A_rep = ctx.matrix_csr(n_rep, n_rep, rp, col)
            trp, tcol = self.T_rep.pattern()
            pkey = np.repeat(np.arange(n_rep, dtype=np.int64), np.diff(rp)) * n_rep + col
            tkey = np.repeat(np.arange(n_rep, dtype=np.int64), np.diff(trp)) * n_rep + tcol
            pos = np.searchsorted(pkey, tkey)
            assert np.all(pkey[np.minimum(pos, pkey.size - 1)] == tkey), "replicated coarse operator leaves its stencil pattern"
            self.map_rep = self.A_rep.value_map(self.T_rep)
        else:
            Pg.destroy()
        self.bdc_rep = capi.Index(ctx, m_rep.dirichlet_dofs(fe).astype(np.int32))
        if self.n_replicated == 2:
            # the coarsest local level as a replicated global level: its operator is the owned rows of every rank scattered into the global
            # stencil pattern and summed (every row has exactly one owner); the level below is its Galerkin product, computed by every rank
            n_g0 = m_g0.n_dofs(fe)
            self.Pg = capi.build_prolongator(ctx, m_rep, m_g0, fe, zero_bdc=True)
            gl_of_local = srt[np.searchsorted(g0_gid[srt], gids[0])]          # node of the extended box -> node of the global level-0 mesh
            assert np.all(g0_gid[gl_of_local] == gids[0])
            grp, gcol = capi.pattern_from_elements(m_g0.arrays()[0], n_g0)
            self.A_g0 = ctx.matrix_csr(n_g0, n_g0, grp, gcol)
            src_row = np.full(n_g0, -1, dtype=np.int32)
            src_row[gl_of_local[loc.owned]] = np.arange(loc.n_owned, dtype=np.int32)
            src_col = np.full(n_g0, -1, dtype=np.int32)
            src_col[gl_of_local[all_local]] = loc.newid[all_local]
            # nothing of the owned rows may fall outside the stencil pattern of the global mesh
            arp, acol = self.A[0].pattern()
            glob_of_newid = np.empty(nloc[0], dtype=np.int64)
            glob_of_newid[loc.newid[all_local]] = gl_of_local[all_local]
            akey = np.repeat(gl_of_local[loc.owned].astype(np.int64), np.diff(arp)) * n_g0 + glob_of_newid[acol]
            gkey = np.repeat(np.arange(n_g0, dtype=np.int64), np.diff(grp)) * n_g0 + gcol
            pos_ = np.searchsorted(gkey, akey)
            assert np.all(gkey[np.minimum(pos_, gkey.size - 1)] == akey), "a level-0 operator row leaves the stencil pattern of the global mesh"
            self.map_g0 = self.A_g0.value_map(self.A[0], src_row, src_col)
            p1, m = full.P[1].restrict(plans[1].owned, gl_of_local.astype(np.int32), n_g0)       # owned level-1 rows x global level-0 columns
            m.destroy()
            self.P1_rep = p1
            self.R1_rep = p1.get_transpose()
            self.A_rep2 = None                                                # Galerkin product of the replicated level, built at the first preparation
        for m_ in (m_rep, m_g0):
            m_.destroy()
        self._replicated_operator()

    @property
    def A_coarse(self):
        """operator of the coarsest (replicated, exactly solved) level of the cycle"""
        return self.A_g0 if self.general else self.A_rep2 if self.n_replicated == 2 else self.A_rep

    def _wire_cycle(self):
        mg, nl = self.mg, self.nl
        if self.general:
            mg.set_level(0, self.A_g0, None, None, 0, self.omega, 1, 0)                                  # replicated, solved exactly
            mg.set_level(1, self.A[1], self.P1_rep, self.R1_rep, 0, self.omega, self.npre, self.npost)
            mg.set_level_distributed(1, self.halos[1], True)
            for l in range(2, nl):
                mg.set_level(l, self.A[l], self.P[l], self.R[l], 0, self.omega, self.npre, self.npost)
                mg.set_level_distributed(l, self.halos[l], False)
            return
        if self.n_replicated == 2:
            mg.set_level(0, self.A_rep2, None, None, 0, self.omega, 1, 0)
            mg.set_level(1, self.A_g0, self.Pg, None, 0, self.omega, self.npre, self.npost)           # replicated, smoothed, no exchange
            mg.set_level(2, self.A[1], self.P1_rep, self.R1_rep, 0, self.omega, self.npre, self.npost)
            mg.set_level_distributed(2, self.halos[1], True)
            for l in range(2, nl):
                mg.set_level(l + 1, self.A[l], self.P[l], self.R[l], 0, self.omega, self.npre, self.npost)
                mg.set_level_distributed(l + 1, self.halos[l], False)
            return
        mg.set_level(0, self.A_rep, None, None, 0, self.omega, 1, 0)
        mg.set_level(1, self.A[0], self.P_rep, self.R_rep, 0, self.omega, self.npre, self.npost)
        mg.set_level_distributed(1, self.halos[0], True)
        for l in range(1, nl):
            mg.set_level(l + 1, self.A[l], self.P[l], self.R[l], 0, self.omega, self.npre, self.npost)
            mg.set_level_distributed(l + 1, self.halos[l], False)

    def _replicated_operator(self):
        """A_rep = sum over ranks of P_rep^T A_0 Pg_local on the stencil pattern, then SetPenalty -- all on the device.  With two replicated
        levels: A_g0 = owned rows of all ranks summed into the global pattern, A_rep2 = Pg^T A_g0 Pg computed by every rank"""
        if self.general:
            self.map_g0.gather_matrix_values(self.A_g0, self.A[0])
            self.halos[1].allreduce_mat(self.A_g0)
            self.bdc_rep.zero_rows(self.A_g0, 1.0)
            return
        if self.n_replicated == 2:
            self.map_g0.gather_matrix_values(self.A_g0, self.A[0])
            self.halos[1].allreduce_mat(self.A_g0)
            if self.A_rep2 is None:
                self.A_rep2 = capi.Mat.ptap(self.Pg, self.A_g0)
            else:
                self.A_rep2.ptap_numeric(self.Pg, self.A_g0)
            self.bdc_rep.zero_rows(self.A_rep2, 1.0)
            return
        self.T_rep.abc_numeric(self.R_rep, self.A[0], self.Pg_local)
        self.map_rep.gather_matrix_values(self.A_rep, self.T_rep)
        self.halos[0].allreduce_mat(self.A_rep)
        self.bdc_rep.zero_rows(self.A_rep, 1.0)

    def prepare(self):
        """numeric re-preparation of the whole distributed hierarchy (MGsolve :347-383): assembly of the extended box, Galerkin
        chain and SetPenalty there, owned rows gathered on the device, replicated operator summed over the ranks, smoother and
        coarse factorisation -- no host round trip of any operator"""
        full = self.full
        full.assemble()
        full.level_operators()
        for l in range(self.nl):
            self.mapA[l].gather_matrix_values(self.A[l], full.A[l])
        self._replicated_operator()
        self.mg.setup()

    def destroy(self):
        """every device object of this problem: the cycle, the owned-row operators and their value maps, the extended-box hierarchy, the
        vectors, and last the exchange plans with their communicator (bench.py drops a problem whose setup failed on another rank)"""
        seen = set()

        def kill(o):
            if o is None or id(o) in seen or o is self.comm or o is self.ctx:
                return
            seen.add(id(o))
            if isinstance(o, (list, tuple)):
                for x in o:
                    kill(x)
                return
            d = getattr(o, "destroy", None)
            if callable(d):
                d()

        kill(getattr(self, "mg", None))
        for name, val in list(vars(self).items()):
            if name not in ("comm", "ctx", "halos", "mg", "full", "part"):
                kill(val)
        kill(getattr(self, "full", None))
        kill(getattr(self, "halos", None))

    def assemble(self):
        # the element loop reads the state at every node of the rank's elements (Res = F - K u, as the one-GPU path does): refresh the ghost
        # entries of SOL first (VecGhostUpdate before the assembly callback, LinearImplicitSystem.cpp:318-327).  Every rank takes part, also
        # one whose box carries adaptive levels and assembles on its extended box: an exchange is collective over the neighbours
        self.halos[-1].update(self.SOL)
        if self.adaptive:
            res_full = self.full.assemble()                            # assembly + P_amr projection on the extended box
            self.mapA[-1].gather_matrix_values(self.A[-1], self.full.A[-1])
            self.map_rows.gather_vector(self.RES, res_full)
            return
        self.asm.assemble(self.A[-1], self.RES, self.SOL, self.source_kind, self.params)

    def set_penalty_top(self):
        self.bdc_dev.zero_rows(self.A[-1], 1.0)

    def zero_boundary_residuals(self):
        self.bdc_dev.set(self.RES, 0.0)

    def vcycle(self):
        self.mg.vcycle(self.RES, self.EPSC)

    def solve(self, outer="gmres", rtol=1e-10, maxit=60):
        self.zero_boundary_residuals()
        return self.mg.solve(self.RES, self.EPSC, outer=outer, rtol=rtol, maxit=maxit)


class DistributedStacked:
    """`nv` variables stacked the way LinearEquation stacks a system on several ranks (rank by rank, variable by variable inside a rank: KKoffset,
    LinearEquation.cpp:212-237) and driven through the SAME device machinery as DistributedPoisson: exchange plans (fh_halo_*), owned-rows operators over
    [owned | ghost] columns, replicated levels below, distributed cycle and Krylov solver (fh_mg_*).  Built from a prepared DistributedPoisson `dp` (box
    partition): block (k, k) of every operator = scale[k] times the scalar one -- `nv` Poisson variables; the plans, the numbering and the cycle do not care
    what couples the blocks.  What a multi-variable application (Navier-Stokes: Missing in DESIGN section 8) adds is its own assembly, not another decomposition."""

    def __init__(self, dp, nv=2, scale=None):
        import scipy.sparse as sp
        assert not dp.general, "stacked systems: box partitions"
        ctx, comm = dp.ctx, dp.comm
        rank, nranks = dp.part.rank, dp.part.nranks
        self.dp, self.nv, self.ctx = dp, nv, ctx
        w = [1.0] * nv if scale is None else [float(v) for v in scale]
        coup = [[(w[k] if k == k2 else 0.0) for k2 in range(nv)] for k in range(nv)]
        nl = dp.nl
        self.plans = [stack_plan(pl, nv, nranks) for pl in dp.H.plans]
        pl = self.plans
        self.A = [ctx.matrix_scipy(stack_matrix(dp.A[l].to_scipy(), pl[l], pl[l], nv, coup)) for l in range(nl)]
        self.P = [None] + [ctx.matrix_scipy(stack_matrix(dp.P[l].to_scipy(), pl[l], pl[l - 1], nv)) for l in range(1, nl)]
        self.R = [None] + [ctx.matrix_scipy(stack_matrix(dp.R[l].to_scipy(), pl[l - 1], pl[l], nv)) for l in range(1, nl)]
        self.halos = []
        if dp.transport == "host":
            hc = dp.halo_comm if dp.halo_comm is not None else comm
            for q in pl:
                self.halos.append(capi.Halo.host(ctx, rank, nranks, hc, q.send_counts, q.send_idx, q.recv_counts, parent=self.halos[0] if self.halos else None))
        else:
            uid = comm.bcast_obj(capi.Halo.unique_id() if rank == 0 else None)
            for q in pl:
                self.halos.append(capi.Halo(ctx, rank, nranks, uid, q.send_counts, q.send_idx, q.recv_counts, parent=self.halos[0] if self.halos else None))

        def rows_by_var(M, n_rows_one, n_cols_one):          # owned rows of every variable x replicated (global) columns [var 0 | var 1 | ...]
            C = M.to_scipy().tocoo()
            r = np.concatenate([C.row + k * n_rows_one for k in range(nv)])
            c = np.concatenate([C.col + k * n_cols_one for k in range(nv)])
            return sp.csr_matrix((np.tile(C.data, nv), (r, c)), shape=(nv * n_rows_one, nv * n_cols_one))

        bd = lambda M: sp.block_diag([w[k] * M.to_scipy() for k in range(nv)]).tocsr()
        self.mg = capi.Multigrid(ctx, nl + 1)
        mg = self.mg
        self.rep = []
        if dp.n_replicated == 2:
            n_g0 = dp.A_g0.m()
            A2, Ag = ctx.matrix_scipy(bd(dp.A_rep2)), ctx.matrix_scipy(bd(dp.A_g0))
            Pg = ctx.matrix_scipy(sp.block_diag([dp.Pg.to_scipy()] * nv).tocsr())
            P1 = rows_by_var(dp.P1_rep, dp.H.plans[1].n_owned, n_g0)
            P1d, R1d = ctx.matrix_scipy(P1), ctx.matrix_scipy(P1.T.tocsr())
            self.rep = [A2, Ag, Pg, P1d, R1d]
            mg.set_level(0, A2, None, None, 0, dp.omega, 1, 0)
            mg.set_level(1, Ag, Pg, None, 0, dp.omega, dp.npre, dp.npost)
            mg.set_level(2, self.A[1], P1d, R1d, 0, dp.omega, dp.npre, dp.npost)
            mg.set_level_distributed(2, self.halos[1], True)
            for l in range(2, nl):
                mg.set_level(l + 1, self.A[l], self.P[l], self.R[l], 0, dp.omega, dp.npre, dp.npost)
                mg.set_level_distributed(l + 1, self.halos[l], False)
        else:
            n_rep = dp.A_rep.m()
            Ar = ctx.matrix_scipy(bd(dp.A_rep))
            P0 = rows_by_var(dp.P_rep, dp.H.plans[0].n_owned, n_rep)
            P0d, R0d = ctx.matrix_scipy(P0), ctx.matrix_scipy(P0.T.tocsr())
            self.rep = [Ar, P0d, R0d]
            mg.set_level(0, Ar, None, None, 0, dp.omega, 1, 0)
            mg.set_level(1, self.A[0], P0d, R0d, 0, dp.omega, dp.npre, dp.npost)
            mg.set_level_distributed(1, self.halos[0], True)
            for l in range(1, nl):
                mg.set_level(l + 1, self.A[l], self.P[l], self.R[l], 0, dp.omega, dp.npre, dp.npost)
                mg.set_level_distributed(l + 1, self.halos[l], False)
        mg.setup()
        top = pl[-1]
        self.n_owned = top.n_owned
        assert int(top.offsets[-1]) < 2 ** 31
        mk = lambda: ctx.vector(int(top.offsets[-1]), top.n_owned, int(top.offsets[rank]), top.ghost_global.astype(np.int32))
        self.RES, self.EPSC = mk(), mk()

    def set_rhs(self, parts):
        """the owned entries of every variable's right-hand side, one array per variable"""
        n0 = self.dp.n_owned
        v = np.zeros(self.plans[-1].n_owned)
        for k, b in enumerate(parts):
            v[k * n0:(k + 1) * n0] = b
        self.RES.upload(v)

    def solve(self, outer="gmres", rtol=1e-10, maxit=60):
        return self.mg.solve(self.RES, self.EPSC, outer=outer, rtol=rtol, maxit=maxit)

    def vcycle(self):
        self.mg.vcycle(self.RES, self.EPSC)

    def destroy(self):
        self.mg.destroy()
        for m in self.A + self.P + self.R + self.rep + [self.RES, self.EPSC] + self.halos:
            if m is not None:
                m.destroy()
