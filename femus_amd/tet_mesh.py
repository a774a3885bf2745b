"""Tetrahedral meshes of applications/001_Poisson on the host (integers and coordinates only; all numerics run in libfemus_hip.so): the Gambit reader for
TET10, refinement, numbering.  The families served on tetrahedra are P1 and P2 (the application's "first" and "serendipity"): the face nodes and the centre
FEMuS adds for its TET15 are not built.

    read_gambit   GambitIO.cpp:101-330: ten nodes per element in Gambit's order -> FEMuS's through GambitToFemusVertexIndex[1] (:66-69), boundary sets
                  "element, type, face" with the faces as numbered in the file (GambitToFemusFaceIndex[1], :85), flag = -(set name) - 1
    refine        MeshRefinement::RefineMesh: children 8 e + j, their vertices through tet_lag::fine2CoarseVertexMapping (read off the element prolongator the
                  library builds from it), new middles shared between neighbours, coordinates by the P2 element prolongator; a child face all of whose
                  vertices lie on a face of the father carries that face's flag
    numbering     vertices, then middles, each class in order of first appearance walking the elements
"""
import numpy as np

from . import capi

G2F = (0, 4, 1, 6, 5, 2, 7, 8, 9, 3)


def _renumber(raw, nnode):
    new = np.full(nnode, -1, dtype=np.int64)
    k, own = 0, []
    for lo, hi in ((0, 4), (4, 10)):
        seq = raw[:, lo:hi].ravel()
        seq = seq[new[seq] < 0]
        uniq, first = np.unique(seq, return_index=True)
        order = np.argsort(first, kind="stable")
        new[uniq[order]] = k + np.arange(uniq.size)
        k += uniq.size
        own.append(k)
    return new, own


def read_gambit(path, Lref=1.0):
    tok = open(path).read().split()
    p = tok.index("NDFVL") + 1
    nvt, nel, ngroup, nbcd, dim, _ = (int(t) for t in tok[p:p + 6])
    if dim != 3:
        raise ValueError("%s: a %d-dimensional mesh where tetrahedra are expected" % (path, dim))
    p = tok.index("COORDINATES") + 2
    nodes = np.array(tok[p:p + 4 * nvt], dtype=object).reshape(nvt, 4)
    xyz = nodes[:, 1:].astype(float) / Lref
    p = tok.index("ELEMENTS/CELLS") + 2
    cells = np.array(tok[p:p + 13 * nel], dtype=object).reshape(nel, 13)
    if not (np.all(cells[:, 1].astype(int) == 6) and np.all(cells[:, 2].astype(int) == 10)):
        raise ValueError("%s: TET10 elements only (element type 6 with 10 nodes)" % path)
    raw = np.zeros((nel, 10), dtype=np.int64)
    raw[:, list(G2F)] = cells[:, 3:].astype(np.int64) - 1
    ff = np.full((nel, 4), -1, dtype=np.int64)
    q = 0
    for _ in range(nbcd):
        q = tok.index("CONDITIONS", q) + 2
        name, nface = int(tok[q]), int(tok[q + 2])
        q += 5
        sets = np.array(tok[q:q + 3 * nface], dtype=np.int64).reshape(nface, 3)
        ff[sets[:, 0] - 1, sets[:, 2] - 1] = -name - 1
        q += 3 * nface
    new, own = _renumber(raw, nvt)
    xs = np.empty_like(xyz)
    xs[new] = xyz
    return new[raw], xs, ff, own


def refine(ed, xs, ff):
    nel = ed.shape[0]
    EP = capi.fe_elem_prolongator("tet", "serendipity")                   # [child][local node][coarse function]
    f2c = np.array([[int(np.argmax(EP[j, v])) for v in range(4)] for j in range(8)])
    faces = [capi.fe_face_nodes("tet", "serendipity", f) for f in range(4)]      # three vertices, three middles
    edge_v = _edge_vertices()
    raw = np.full((8 * nel, 10), -1, dtype=np.int64)
    fff = np.full((8 * nel, 4), -1, dtype=np.int64)
    for j in range(8):
        raw[j::8, :4] = ed[:, f2c[j]]
        for lf in range(4):
            for f in range(4):
                if all(int(f2c[j][v]) in faces[f].tolist() for v in faces[lf][:3]):
                    fff[j::8, lf] = ff[:, f]
    a = np.stack([raw[:, e[0]] for e in edge_v], axis=1)
    b = np.stack([raw[:, e[1]] for e in edge_v], axis=1)
    key = (np.minimum(a, b) * np.int64(xs.shape[0]) + np.maximum(a, b)).ravel()
    uniq, first, inv = np.unique(key, return_index=True, return_inverse=True)
    rank = np.empty(uniq.size, dtype=np.int64)
    rank[np.argsort(first, kind="stable")] = np.arange(uniq.size)
    raw[:, 4:] = (xs.shape[0] + rank[inv]).reshape(-1, 6)
    owner = np.empty(uniq.size, dtype=np.int64)
    owner[rank] = first                                                    # (child element * 6 + local edge) that created the node
    c, k = owner // 6, owner % 6
    mid = np.zeros((uniq.size, 3))
    for m in range(10):
        mid += EP[c % 8, 4 + k, m][:, None] * xs[ed[c // 8, m]]
    coords = np.concatenate([xs, mid])
    new, own = _renumber(raw, coords.shape[0])
    used = new >= 0
    xf = np.empty((own[1], 3))
    xf[new[used]] = coords[used]
    return new[raw], xf, fff, own


def _edge_vertices():
    """the two vertices each of the local nodes 4 .. 9 sits between (its reference point is their mean)"""
    x = np.array([capi.fe_node_ref_coords("tet", n) for n in range(10)])
    out = []
    for m in range(4, 10):
        pair = [(a, b) for a in range(4) for b in range(a + 1, 4) if np.allclose(0.5 * (x[a] + x[b]), x[m])]
        out.append(pair[0])
    return out
