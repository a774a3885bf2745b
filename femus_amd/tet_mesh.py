"""Tetrahedral meshes of applications/001_Poisson on the host (integers and coordinates only; all numerics run in libfemus_hip.so): the Gambit reader for
TET10, the face and centre nodes FEMuS adds (TET15), refinement, numbering.

    read_gambit   GambitIO.cpp:101-330: ten nodes per element in Gambit's order -> FEMuS's through GambitToFemusVertexIndex[1] (:66-69), boundary sets
                  "element, type, face" with the faces as numbered in the file (GambitToFemusFaceIndex[1], :85), flag = -(set name) - 1;
                  Mesh::AddBiquadraticNodesNotInMeshFile (Mesh.cpp:1207-1333): one node per face (shared by the two tetrahedra it separates), one centre per
                  element, coordinates with the weights of Mesh.cpp:107-113 (faces -1/9, 4/9; centre -1/8, 1/4)
    refine        MeshRefinement::RefineMesh: children 8 e + j, their vertices through tet_lag::fine2CoarseVertexMapping (read off the element prolongator the
                  library builds from it), new middles and face centres shared between neighbours, a centre per child, coordinates by the TET15 element
                  prolongator; a child face all of whose vertices lie on a face of the father carries that face's flag
    numbering     vertices, then middles, then face centres and centres, each class in order of first appearance walking the elements
"""
import numpy as np

from . import _mesh_keys, capi

G2F = (0, 4, 1, 6, 5, 2, 7, 8, 9, 3)
# Mesh.cpp:107-113: weights of the ten file nodes in the four face nodes and in the centre
WGT = np.array([[-1. / 9., -1. / 9., -1. / 9., 0, 4. / 9., 4. / 9., 4. / 9., 0, 0, 0], [-1. / 9., -1. / 9., 0, -1. / 9., 4. / 9., 0, 0, 4. / 9., 4. / 9., 0],
                [0, -1. / 9., -1. / 9., -1. / 9., 0, 4. / 9., 0, 0, 4. / 9., 4. / 9.], [-1. / 9., 0, -1. / 9., -1. / 9., 0, 0, 4. / 9., 4. / 9., 0, 4. / 9.],
                [-1. / 8.] * 4 + [1. / 4.] * 6])


_first_touch = _mesh_keys.first_touch


def _renumber(raw, nnode):
    new = np.full(nnode, -1, dtype=np.int64)
    k, own = 0, []
    for lo, hi in ((0, 4), (4, 10), (10, 15)):
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
    if ngroup != 1:          # several groups: Mesh.cpp:626-690 orders the elements by (material, group, index) -- not built here, refused rather than mis-ordered
        raise ValueError("%s: %d element groups; this reader keeps the file's element order, which is the reference's only for one group" % (path, ngroup))
    if dim != 3:
        raise ValueError("%s: a %d-dimensional mesh where tetrahedra are expected" % (path, dim))
    p = tok.index("COORDINATES") + 2
    nodes = np.array(tok[p:p + 4 * nvt], dtype=object).reshape(nvt, 4)
    xyz = nodes[:, 1:].astype(float) / Lref
    p = tok.index("ELEMENTS/CELLS") + 2
    cells = np.array(tok[p:p + 13 * nel], dtype=object).reshape(nel, 13)
    if not (np.all(cells[:, 1].astype(int) == 6) and np.all(cells[:, 2].astype(int) == 10)):
        raise ValueError("%s: TET10 elements only (element type 6 with 10 nodes)" % path)
    raw = np.full((nel, 15), -1, dtype=np.int64)
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
    faces = [capi.fe_face_nodes("tet", "biquadratic", f) for f in range(4)]
    keys = np.sort(np.stack([raw[:, faces[f][:3]] for f in range(4)], axis=1).reshape(4 * nel, 3), axis=1)
    ids, _ = _first_touch(keys)                                           # element by element, face by face: the first tetrahedron that holds a face creates its node
    raw[:, 10:14] = (nvt + ids).reshape(nel, 4)
    nface_nodes = int(ids.max()) + 1
    raw[:, 14] = nvt + nface_nodes + np.arange(nel)
    coords = np.concatenate([xyz, np.zeros((nface_nodes + nel, 3))])
    acc = np.zeros((nel, 5, 3))
    for j in range(10, 15):
        for i in range(10):                                               # the sum in the order of Mesh.cpp:1316-1324
            acc[:, j - 10] += coords[raw[:, i]] * WGT[j - 10][i]
    for e in range(nel):                                                  # element by element as the reference does: a shared face node keeps the later element's sum
        coords[raw[e, 10:15]] = acc[e]
    new, own = _renumber(raw, coords.shape[0])
    xs = np.empty_like(coords)
    xs[new] = coords
    return new[raw], xs, ff, own


def refine(ed, xs, ff):
    nel = ed.shape[0]
    EP = capi.fe_elem_prolongator("tet", "biquadratic")                   # [child][local node][coarse function]
    f2c = np.array([[int(np.argmax(EP[j, v])) for v in range(4)] for j in range(8)])
    faces = [capi.fe_face_nodes("tet", "biquadratic", f) for f in range(4)]      # three vertices, three middles, the centre
    edge_v = _edge_vertices()
    raw = np.full((8 * nel, 15), -1, dtype=np.int64)
    fff = np.full((8 * nel, 4), -1, dtype=np.int64)
    for j in range(8):
        raw[j::8, :4] = ed[:, f2c[j]]
        for lf in range(4):
            for f in range(4):
                if all(int(f2c[j][v]) in faces[f].tolist() for v in faces[lf][:3]):
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
        for m in range(15):
            pos += EP[c % 8, loc, m][:, None] * xs[ed[c // 8, m]]
        coords.append(pos)
        out = (nnew + ids).reshape(-1, n)
        nnew += owner.size
        return out

    a = np.stack([raw[:, e[0]] for e in edge_v], axis=1)
    b = np.stack([raw[:, e[1]] for e in edge_v], axis=1)
    raw[:, 4:10] = create(np.stack([np.minimum(a, b), np.maximum(a, b)], axis=2), list(range(4, 10)))
    tri = np.sort(np.stack([raw[:, faces[f][:3]] for f in range(4)], axis=1), axis=2)
    raw[:, 10:14] = create(tri, [10, 11, 12, 13])
    raw[:, 14] = nnew + ch
    cen = np.zeros((8 * nel, 3))
    for m in range(15):
        cen += EP[ch % 8, 14, m][:, None] * xs[ed[ch // 8, m]]
    coords.append(cen)
    coords = np.concatenate(coords)
    new, own = _renumber(raw, coords.shape[0])
    used = new >= 0
    xf = np.empty((own[2], 3))
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
