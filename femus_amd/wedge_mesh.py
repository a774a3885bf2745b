"""Prism meshes of applications/001_Poisson on the host (integers and coordinates only; all numerics run in libfemus_hip.so): the Gambit reader for WEDGE18, the
triangle-face and centre nodes FEMuS adds (WEDGE21), refinement, numbering.

    read_gambit   GambitIO.cpp:101-330: eighteen nodes per element in Gambit's order -> FEMuS's through GambitToFemusVertexIndex[2] (:70-74), boundary sets with
                  the faces through GambitToFemusFaceIndex[2] = {2, 1, 0, 4, 3} (:86), flag = -(set name) - 1; Mesh::AddBiquadraticNodesNotInMeshFile
                  (Mesh.cpp:1207-1333): one node per triangle face (shared by the two prisms it separates), one centre per element, coordinates with the
                  weights -1/9, 4/9 of Mesh.cpp:115-122
    refine        MeshRefinement::RefineMesh: children 8 e + j, their vertices through wedge_lag::fine2CoarseVertexMapping (read off the element prolongator),
                  new edge middles / face centres shared between neighbours, a centre per child, coordinates by the biquadratic element prolongator; a child
                  face all of whose vertices lie on a face of the father carries that face's flag
    numbering     vertices, then edge middles, then face centres and centres, each class in order of first appearance walking the elements
"""
import numpy as np

from . import _mesh_keys, capi

G2F = (3, 11, 5, 9, 10, 4, 12, 17, 14, 15, 16, 13, 0, 8, 2, 6, 7, 1)
GFACE = (2, 1, 0, 4, 3)


def _renumber(raw, nnode):
    new = np.full(nnode, -1, dtype=np.int64)
    k, own = 0, []
    for lo, hi in ((0, 6), (6, 15), (15, 21)):
        seq = raw[:, lo:hi].ravel()
        seq = seq[new[seq] < 0]
        uniq, first = np.unique(seq, return_index=True)
        order = np.argsort(first, kind="stable")
        new[uniq[order]] = k + np.arange(uniq.size)
        k += uniq.size
        own.append(k)
    return new, own


def _tables():
    faces = [capi.fe_face_nodes("wedge", "biquadratic", f) for f in range(5)]          # quadrilaterals: 4 + 4 + 1 nodes, triangles: 3 + 3 + 1
    x = np.array([capi.fe_node_ref_coords("wedge", n) for n in range(21)])
    edges = []
    for m in range(6, 15):
        edges.append([(a, b) for a in range(6) for b in range(a + 1, 6) if np.allclose(0.5 * (x[a] + x[b]), x[m])][0])
    return faces, edges


_first_touch = _mesh_keys.first_touch


def read_gambit(path, Lref=1.0):
    tok = open(path).read().split()
    p = tok.index("NDFVL") + 1
    nvt, nel, ngroup, nbcd, dim, _ = (int(t) for t in tok[p:p + 6])
    if ngroup != 1:          # several groups: Mesh.cpp:626-690 orders the elements by (material, group, index) -- not built here, refused rather than mis-ordered
        raise ValueError("%s: %d element groups; this reader keeps the file's element order, which is the reference's only for one group" % (path, ngroup))
    p = tok.index("COORDINATES") + 2
    xyz = np.array(tok[p:p + 4 * nvt], dtype=object).reshape(nvt, 4)[:, 1:].astype(float) / Lref
    p = tok.index("ELEMENTS/CELLS") + 2
    cells = np.array(tok[p:p + 21 * nel], dtype=object).reshape(nel, 21)
    if not (np.all(cells[:, 1].astype(int) == 5) and np.all(cells[:, 2].astype(int) == 18)):
        raise ValueError("%s: WEDGE18 elements only (element type 5 with 18 nodes)" % path)
    raw = np.full((nel, 21), -1, dtype=np.int64)
    raw[:, list(G2F)] = cells[:, 3:].astype(np.int64) - 1
    ff = np.full((nel, 5), -1, dtype=np.int64)
    q = 0
    for _ in range(nbcd):
        q = tok.index("CONDITIONS", q) + 2
        name, nface = int(tok[q]), int(tok[q + 2])
        q += 5
        sets = np.array(tok[q:q + 3 * nface], dtype=np.int64).reshape(nface, 3)
        ff[sets[:, 0] - 1, np.array(GFACE)[sets[:, 2] - 1]] = -name - 1
        q += 3 * nface
    faces, _ = _tables()
    # triangle-face nodes: element by element, face 3 then 4; the first prism that holds a face creates its node
    keys = np.sort(np.stack([raw[:, faces[3][:3]], raw[:, faces[4][:3]]], axis=1).reshape(2 * nel, 3), axis=1)
    ids, _ = _first_touch(keys)
    raw[:, 18:20] = (nvt + ids).reshape(nel, 2)
    ntri = int(ids.max()) + 1
    raw[:, 20] = nvt + ntri + np.arange(nel)
    coords = np.concatenate([xyz, np.zeros((ntri + nel, 3))])
    W = {18: ([0, 1, 2], [6, 7, 8]), 19: ([3, 4, 5], [9, 10, 11]), 20: ([12, 13, 14], [15, 16, 17])}
    acc = np.zeros((nel, 3, 3))
    for j, (neg, pos) in W.items():
        for i in range(18):                                  # the sum in the order of Mesh.cpp:1316-1324
            wgt = -1. / 9. if i in neg else 4. / 9. if i in pos else 0.0
            acc[:, j - 18] += coords[raw[:, i]] * wgt
    for e in range(nel):                                     # element by element as the reference does: a shared face node keeps the later element's sum
        coords[raw[e, 18:21]] = acc[e]
    new, own = _renumber(raw, coords.shape[0])
    xs = np.empty_like(coords)
    xs[new] = coords
    return new[raw], xs, ff, own


def refine(ed, xs, ff):
    nel = ed.shape[0]
    EP = capi.fe_elem_prolongator("wedge", "biquadratic")                 # [child][local node][coarse function]
    f2c = np.array([[int(np.argmax(EP[j, v])) for v in range(6)] for j in range(8)])
    faces, edges = _tables()
    nvf = [4, 4, 4, 3, 3]
    raw = np.full((8 * nel, 21), -1, dtype=np.int64)
    fff = np.full((8 * nel, 5), -1, dtype=np.int64)
    for j in range(8):
        raw[j::8, :6] = ed[:, f2c[j]]
        for lf in range(5):
            for f in range(5):
                if nvf[lf] == nvf[f] and all(int(f2c[j][v]) in faces[f].tolist() for v in faces[lf][:nvf[lf]]):
                    fff[j::8, lf] = ff[:, f]
    ch = np.arange(8 * nel)
    coords = [xs]
    nnew = xs.shape[0]

    def create(keys, locals_per_key):
        """new shared nodes for the keys [8 nel, n, width] (first appearance in element order, then local order); locals_per_key[k] = local node index"""
        nonlocal nnew
        n = keys.shape[1]
        ids, owner = _first_touch(keys.reshape(-1, keys.shape[2]))
        c, k = owner // n, owner % n
        loc = np.array(locals_per_key)[k]
        pos = np.zeros((owner.size, 3))
        for m in range(21):
            pos += EP[c % 8, loc, m][:, None] * xs[ed[c // 8, m]]
        coords.append(pos)
        out = (nnew + ids).reshape(-1, n)
        nnew += owner.size
        return out

    a = np.stack([raw[:, e[0]] for e in edges], axis=1)
    b = np.stack([raw[:, e[1]] for e in edges], axis=1)
    raw[:, 6:15] = create(np.stack([np.minimum(a, b), np.maximum(a, b)], axis=2), list(range(6, 15)))
    quad = np.sort(np.stack([raw[:, faces[f][:4]] for f in range(3)], axis=1), axis=2)
    raw[:, 15:18] = create(quad, [15, 16, 17])
    tri = np.sort(np.stack([raw[:, faces[f][:3]] for f in (3, 4)], axis=1), axis=2)
    raw[:, 18:20] = create(tri, [18, 19])
    raw[:, 20] = nnew + ch
    cen = np.zeros((8 * nel, 3))
    for m in range(21):
        cen += EP[ch % 8, 20, m][:, None] * xs[ed[ch // 8, m]]
    coords.append(cen)
    coords = np.concatenate(coords)
    new, own = _renumber(raw, coords.shape[0])
    used = new >= 0
    xf = np.empty((own[2], 3))
    xf[new[used]] = coords[used]
    return new[raw], xf, fff, own
