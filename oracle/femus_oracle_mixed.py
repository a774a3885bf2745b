"""TEST INFRASTRUCTURE ONLY (oracle).  CPU restatement of the MIXED-SHAPE path of applications/001_Poisson (its shipped input3D.json / input3D_All_first.json with
input/cube_all_shapes_Six_boundary_groups.neu: tetrahedra, prisms and hexahedra in one Gambit file; and the two-dimensional Gambit files of the reference tree
with QUAD9 and / or TRI6 elements): reader, the nodes FEMuS adds, numbering, refinement, the
Poisson callback element by element with each element's own shape, face integrals on triangles and quadrilaterals, solve -- loops as the reference writes
them, on top of the single-shape oracles.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.

  read_gambit   GambitIO.cpp:101-330 (types 4 / 6 / 5 / 2 / 3 = HEX27 / TET10 / WEDGE18 / QUAD9 / TRI6; GambitToFemusVertexIndex :55-69, GambitToFemusFaceIndex :84-86);
                Mesh::AddBiquadraticNodesNotInMeshFile (Mesh.cpp:1207-1333) over elements of both shapes with triangle faces; weights Mesh.cpp:105-122
  refine        MeshRefinement::RefineMesh: children 8 e + j, shape of the father; shared edge / face nodes through dictionaries
  assemble      main.cpp:355-480: el->GetElementType(iel) picks the tables of every element
  PIN           bases and Gauss rules through the single-shape oracles: tests/golden/fe_tables.npz (written by the reference's compiled classes); the reader, the added
                nodes, refinement and numbering are restated from the cited lines -- "parity unpinned" for that integer half (the mesh layer of the reference does
                not build here: DESIGN section 5), anchored on the product's independent implementation giving the same integers and on geometric invariants
"""
import os

import numpy as np

from . import femus_oracle as fo
from . import femus_oracle_tet as oq
from . import femus_oracle_tri as ot
from . import femus_oracle_wedge as ow

_G = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "fe_tables.npz"))
NLOC = {"hex": 27, "tet": 15, "wedge": 21, "quad": 9, "tri": 7}
CLASSES = {"hex": (8, 20, 27), "tet": (4, 10, 15), "wedge": (6, 15, 21), "quad": (4, 8, 9), "tri": (3, 6, 7)}
FACE = {"hex": [list(r) for r in _G["facedofs_hex"]], "tet": [list(r) for r in oq.FACE], "wedge": [list(r) for r in ow.FACE],
        "quad": [list(r) for r in _G["facedofs_quad"]], "tri": [list(r) for r in ot.FACE]}
NVF = {"hex": [4] * 6, "tet": [3] * 4, "wedge": [4, 4, 4, 3, 3], "quad": [2] * 4, "tri": [2] * 3}
FACE_LOCAL = {"hex": [20, 21, 22, 23, 24, 25], "tet": [10, 11, 12, 13], "wedge": [15, 16, 17, 18, 19], "quad": [4, 5, 6, 7], "tri": [3, 4, 5]}
HEX_EDGE = [(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7)]      # local nodes 8 .. 19 (Hexahedron.cpp: XC rows)
EDGE = {"hex": HEX_EDGE, "tet": list(oq.EDGE), "wedge": list(ow.EDGE), "quad": [(0, 1), (1, 2), (2, 3), (3, 0)], "tri": [(0, 1), (1, 2), (2, 0)]}
GAMBIT = {(4, 27): "hex", (6, 10): "tet", (5, 18): "wedge", (2, 9): "quad", (3, 6): "tri"}
G2F = {"hex": (4, 16, 0, 15, 23, 11, 7, 19, 3, 12, 20, 8, 25, 26, 24, 14, 22, 10, 5, 17, 1, 13, 21, 9, 6, 18, 2), "tet": oq.G2F, "wedge": ow.G2F,
       "quad": (0, 4, 1, 5, 2, 6, 3, 7, 8), "tri": (0, 3, 1, 4, 2, 5)}
GFACE = {"hex": (0, 4, 2, 5, 3, 1), "tet": (0, 1, 2, 3), "wedge": ow.GFACE, "quad": (0, 1, 2, 3), "tri": (0, 1, 2)}
WEDGE_W = np.zeros((3, 18))
WEDGE_W[0, [0, 1, 2]], WEDGE_W[0, [6, 7, 8]] = -1. / 9., 4. / 9.
WEDGE_W[1, [3, 4, 5]], WEDGE_W[1, [9, 10, 11]] = -1. / 9., 4. / 9.
WEDGE_W[2, [12, 13, 14]], WEDGE_W[2, [15, 16, 17]] = -1. / 9., 4. / 9.
ADDED = {"tet": oq.WGT, "wedge": WEDGE_W, "tri": np.array([[-1. / 9.] * 3 + [4. / 9.] * 3])}
COMPLETE = ("hex", "quad")
NDOF = {s: {"linear": CLASSES[s][0], "serendipity": CLASSES[s][1], "biquadratic": CLASSES[s][2]} for s in NLOC}
NFN = {2: {"linear": 2, "serendipity": 3, "biquadratic": 3}, 3: {"linear": 3, "serendipity": 6, "biquadratic": 7}, 4: {"linear": 4, "serendipity": 8, "biquadratic": 9}}


def f2c(shape):
    if shape in ("hex", "quad"):
        return fo.fine2coarse_vertex_mapping(shape)
    return {"tet": oq.F2C, "wedge": ow.F2C, "tri": ot.F2C}[shape]


def elem_prolongator(shape, fe):
    return fo.elem_prolongator(shape, fe) if shape in ("hex", "quad") else {"tet": oq, "wedge": ow, "tri": ot}[shape].elem_prolongator(fe)


def tables(shape, fe, order="seventh"):
    """(w[ng], phi[ng, nc], dphi[ng, nc, dim]) of a shape"""
    if shape in ("hex", "quad"):
        d = 3 if shape == "hex" else 2
        w, x = fo.gauss_table(shape, order)
        x = np.asarray(x)
        if x.shape[0] == d and x.shape[1] != d:
            x = x.T
        out = fo.eval_basis(shape, fe, x)
        return np.asarray(w), out[0], out[1]
    m = {"tet": oq, "wedge": ow, "tri": ot}[shape]
    w, x = m.gauss(order)
    phi, dphi = m.basis(fe, x)
    return w, phi, dphi


def _renumber(kind, raw, nnode):
    new = np.full(nnode, -1, dtype=np.int64)
    k, own = 0, []
    for c in range(3):
        for e in range(raw.shape[0]):
            lo = 0 if c == 0 else CLASSES[kind[e]][c - 1]
            for l in range(lo, CLASSES[kind[e]][c]):
                if new[raw[e, l]] < 0:
                    new[raw[e, l]] = k
                    k += 1
        own.append(k)
    return new, own


def _apply(new, raw):
    out = np.full_like(raw, -1)
    for e in range(raw.shape[0]):
        for l in range(raw.shape[1]):
            if raw[e, l] >= 0:
                out[e, l] = new[raw[e, l]]
    return out


def read_gambit(path):
    tok = open(path).read().split()
    p = tok.index("NDFVL") + 1
    nvt, nel, ngroup, nbcd, dim, dim_nodes = (int(t) for t in tok[p:p + 6])
    assert dim in (2, 3) and dim_nodes == dim
    p = tok.index("COORDINATES") + 2
    xyz = np.zeros((nvt, dim))
    for n in range(nvt):
        xyz[n] = [float(t) for t in tok[p + 1:p + 1 + dim]]
        p += 1 + dim
    p = tok.index("ELEMENTS/CELLS") + 2
    kind = []
    raw = np.full((nel, 27), -1, dtype=np.int64)
    for e in range(nel):
        nn = int(tok[p + 2])
        s = GAMBIT[(int(tok[p + 1]), nn)]
        kind.append(s)
        for i in range(nn):
            raw[e, G2F[s][i]] = int(tok[p + 3 + i]) - 1
        p += 3 + nn
    ff = np.full((nel, 6), -1, dtype=np.int64)
    q = 0
    for _ in range(nbcd):
        q = tok.index("CONDITIONS", q) + 2
        name, nface = int(tok[q]), int(tok[q + 2])
        q += 5
        for _ in range(nface):
            e = int(tok[q]) - 1
            ff[e, GFACE[kind[e]][int(tok[q + 2]) - 1]] = -name - 1
            q += 3
    # Mesh.cpp:1228-1270: triangle faces of tetrahedra and prisms, the first element that holds one creates its node and hands it to the first later element that
    # holds the same three vertices; :1273-1287: then a centre per tetrahedron / prism
    nn = nvt
    tri = {s: [f for f in range(len(FACE[s])) if NVF[s][f] == 3 and dim == 3] for s in NLOC}
    for e in range(nel):
        for f in tri[kind[e]]:
            l = FACE_LOCAL[kind[e]][f]
            if raw[e, l] < 0:
                raw[e, l] = nn
                mine = set(raw[e, FACE[kind[e]][f][:3]].tolist())
                done = False
                for e2 in range(e + 1, nel):
                    for f2 in tri[kind[e2]]:
                        l2 = FACE_LOCAL[kind[e2]][f2]
                        if raw[e2, l2] < 0 and set(raw[e2, FACE[kind[e2]][f2][:3]].tolist()) == mine:
                            raw[e2, l2] = nn
                            done = True
                            break
                    if done:
                        break
                nn += 1
    for e in range(nel):
        if kind[e] not in COMPLETE:
            raw[e, NLOC[kind[e]] - 1] = nn
            nn += 1
    coords = np.concatenate([xyz, np.zeros((nn - nvt, dim))])
    for e in range(nel):
        if kind[e] in COMPLETE:
            continue
        W = ADDED[kind[e]]
        j0 = NLOC[kind[e]] - W.shape[0]
        for j in range(W.shape[0]):
            acc = np.zeros(dim)
            for i in range(j0):
                acc += coords[raw[e, i]] * W[j][i]
            coords[raw[e, j0 + j]] = acc
    kind = np.array(kind)
    # GambitIO.cpp:298-320: "GROUP:" k "ELEMENTS:" n "MATERIAL:" m "NFLAGS:" f, the group's name (read as its number), the flags, the n elements;
    # Mesh.cpp:626-690 (mesh_reorder_elem_quantities): a bubble sort of the elements by material, then group, then index
    group, material = [1] * nel, [0] * nel
    q = 0
    for _ in range(ngroup):
        q = tok.index("GROUP:", q)
        ngel, mat, name = int(tok[q + 3]), int(tok[q + 5]), int(tok[q + 8])
        for i in range(ngel):
            group[int(tok[q + 10 + i]) - 1], material[int(tok[q + 10 + i]) - 1] = name, mat
        q += 10 + ngel
    inv = list(range(nel))
    n = nel
    while n > 1:
        newn = 0
        for j in range(1, n):
            a, b = inv[j - 1], inv[j]
            if material[b] < material[a] or (material[b] == material[a] and (group[b] < group[a] or (group[b] == group[a] and b < a))):
                inv[j - 1], inv[j] = b, a
                newn = j
        n = newn
    kind, raw, ff = kind[inv], raw[inv], ff[inv]
    new, own = _renumber(kind, raw, nn)
    xs = np.empty_like(coords)
    xs[new] = coords
    return kind, _apply(new, raw), xs, ff, own


def refine(kind, ed, xs, ff):
    nel, dim = ed.shape[0], xs.shape[1]
    nch = 8 if dim == 3 else 4
    ck = np.repeat(kind, nch)
    raw = np.full((nch * nel, 27), -1, dtype=np.int64)
    fff = np.full((nch * nel, 6), -1, dtype=np.int64)
    coords = list(xs)
    shared = {}
    EPs = {s: elem_prolongator(s, "biquadratic") for s in set(kind.tolist())}
    for e in range(nel):
        s = kind[e]
        nv, ne, nl = CLASSES[s]
        EP, F2C = EPs[s], f2c(s)

        def place(j, local):
            return sum(EP[j, local, m] * xs[ed[e, m]] for m in range(nl))

        for j in range(nch):
            c = nch * e + j
            cn = F2C[j]
            raw[c, :nv] = ed[e, cn]
            for lf in range(len(FACE[s])):                     # a child face all of whose vertices lie on a face of the father carries that face's flag
                for f in range(len(FACE[s])):
                    if NVF[s][lf] == NVF[s][f] and all(int(cn[v]) in FACE[s][f] for v in FACE[s][lf][:NVF[s][lf]]):
                        fff[c, lf] = ff[e, f]
            todo = [(nv + k, tuple(sorted((int(raw[c, a]), int(raw[c, b]))))) for k, (a, b) in enumerate(EDGE[s])]
            if dim == 3:                                       # (in two dimensions the faces are the edges)
                todo += [(FACE_LOCAL[s][f], tuple(sorted(int(v) for v in raw[c, FACE[s][f][:NVF[s][f]]]))) for f in range(len(FACE[s]))]
            for local, key in sorted(todo):                    # local order
                if key not in shared:
                    shared[key] = len(coords)
                    coords.append(place(j, local))
                raw[c, local] = shared[key]
            raw[c, nl - 1] = len(coords)
            coords.append(place(j, nl - 1))
    coords = np.array(coords)
    new, own = _renumber(ck, raw, coords.shape[0])
    used = new >= 0
    xf = np.empty((own[2], dim))
    xf[new[used]] = coords[used]
    return ck, _apply(new, raw), xf, fff, own


def n_dofs(own, fe):
    return own[{"linear": 0, "serendipity": 1, "biquadratic": 2}[fe]]


def assemble(kind, ed, xs, fe, source, sol=None, order="seventh"):
    import scipy.sparse as sp
    T = {s: tables(s, fe, order) for s in set(kind.tolist())}
    ndof = max(int(ed[e, :NDOF[kind[e]][fe]].max()) for e in range(ed.shape[0])) + 1
    rows, cols, vals = [], [], []
    F = np.zeros(ndof)
    u = np.zeros(ndof) if sol is None else sol
    for e in range(ed.shape[0]):
        nc = NDOF[kind[e]][fe]
        w, PHI, DPHI = T[kind[e]]
        dof = ed[e, :nc]
        x = xs[dof]
        Ke = np.zeros((nc, nc))
        Fe = np.zeros(nc)
        for g in range(len(w)):
            J = DPHI[g].T @ x
            det = np.linalg.det(J)
            grad = DPHI[g] @ np.linalg.inv(J).T
            weight = det * w[g]
            gu = grad.T @ u[dof]
            f = source(PHI[g] @ x)
            Ke += (grad @ grad.T) * weight
            Fe += (f * PHI[g] - grad @ gu) * weight
        rows.append(np.repeat(dof, nc))
        cols.append(np.tile(dof, nc))
        vals.append(Ke.ravel())
        F[dof] += Fe
    K = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(ndof, ndof))
    return K, F


def _face_tables(nv, fe, order):
    if nv == 2:                                              # line elements: ends, then middle
        w, xg = fo.gauss_table("line", order)
        xg = np.asarray(xg).reshape(-1)
        nodes = (0, 2) if fe == "linear" else (0, 2, 1)
        lag, dlag = (fo.lag_linear, fo.dlag_linear) if fe == "linear" else (fo.lag_biquadratic, fo.dlag_biquadratic)
        ph = np.array([[lag(x, I) for I in nodes] for x in xg])
        dp = np.array([[[dlag(x, I)] for I in nodes] for x in xg])
        return np.asarray(w), ph, dp
    if nv == 4:
        w, x = fo.gauss_table("quad", order)
        x = np.asarray(x)
        if x.shape[0] == 2 and x.shape[1] != 2:
            x = x.T
        out = fo.eval_basis("quad", fe, x)
        return np.asarray(w), out[0], out[1]
    w, x = ot.gauss(order)
    ph, dp = ot.basis(fe, x)
    return w, ph, dp


def neumann(kind, ed, xs, ff, fe, flux_by_flag, ndof, order="seventh"):
    FT = {nv: _face_tables(nv, fe, order) for nv in ((3, 4) if xs.shape[1] == 3 else (2,))}
    F = np.zeros(ndof)
    for e, f in zip(*np.nonzero(ff < -1)):
        if ff[e, f] not in flux_by_flag:
            continue
        nv = NVF[kind[e]][f]
        fn = ed[e, FACE[kind[e]][f][:NFN[nv][fe]]]
        x = xs[fn]
        w, PH, DP = FT[nv]
        for g in range(len(w)):
            t = DP[g].T @ x
            area = np.linalg.norm(np.cross(t[0], t[1])) if nv > 2 else np.hypot(t[0][0], t[0][1])
            tau = flux_by_flag[ff[e, f]]
            tv = tau(PH[g] @ x) if callable(tau) else tau
            F[fn] += PH[g] * tv * area * w[g]
    return F


def dirichlet(kind, ed, ff, fe, flags):
    out = set()
    for e, f in zip(*np.nonzero(ff < -1)):
        if ff[e, f] in flags:
            nv = NVF[kind[e]][f]
            out.update(int(n) for n in ed[e, FACE[kind[e]][f][:NFN[nv][fe]]])
    return np.array(sorted(out), dtype=np.int64)


def solve(mesh0, nlevels, fe, source, dirichlet_flags, flux_by_flag=None):
    """the discrete problem of the finest of nlevels levels, solved directly"""
    import scipy.sparse.linalg as spla
    meshes = [mesh0]
    for _ in range(1, nlevels):
        meshes.append(refine(*meshes[-1][:4]))
    kind, ed, xs, ff, own = meshes[-1]
    ndof = n_dofs(own, fe)
    bdc = dirichlet(kind, ed, ff, fe, set(dirichlet_flags))
    K, F = assemble(kind, ed, xs, fe, source)
    assert K.shape[0] == ndof
    if flux_by_flag:
        F = F + neumann(kind, ed, xs, ff, fe, flux_by_flag, ndof)
    K = K.tolil()
    K[bdc, :] = 0.0
    K[bdc, bdc] = 1.0
    F[bdc] = 0.0
    return spla.spsolve(K.tocsc(), F), meshes
