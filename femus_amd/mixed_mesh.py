"""Meshes of MIXED shapes of applications/001_Poisson (input/cube_all_shapes*.neu: hexahedra, tetrahedra and prisms in one Gambit file) and the two-dimensional
Gambit files of quadrilaterals and / or triangles the reference tree holds (QUAD9 + TRI6 in one file, TRI6 alone) on the host -- integers and coordinates only; all
numerics run in libfemus_hip.so.  A mesh is (kind[nel] of "hex" / "tet" / "wedge" / "quad" / "tri", ed[nel, 27] padded with -1, xs[nnode, dim], ff[nel, 6] padded
with -1, own[3]).

    read_gambit   GambitIO.cpp:101-330: HEX27 (type 4), TET10 (type 6), WEDGE18 (type 5), QUAD9 (type 2), TRI6 (type 3) ordered by (material, group, file index) as Mesh.cpp:626-690 orders them, nodes through
                  GambitToFemusVertexIndex (:55-69), faces through GambitToFemusFaceIndex (:84-86), flag = -(set name) - 1;
                  Mesh::AddBiquadraticNodesNotInMeshFile (Mesh.cpp:1207-1333): a node per TRIANGLE face -- shared between a tetrahedron and a prism as well --,
                  created by the first element that holds it, then a centre per tetrahedron / prism / triangle; coordinates with the weights of Mesh.cpp:105-122
    refine        MeshRefinement::RefineMesh: children 8 e + j (4 e + j in two dimensions) of the father's shape, vertices through each shape's fine2CoarseVertexMapping (read off the element
                  prolongator), new edge / face nodes shared between neighbours of any shape, coordinates by the creating child's element prolongator
    numbering     vertices, then edge middles, then the rest, each class in order of first appearance walking the elements
"""
import numpy as np

from . import _mesh_keys, capi

SHAPES = ("hex", "tet", "wedge", "quad", "tri")
NLOC = {"hex": 27, "tet": 15, "wedge": 21, "quad": 9, "tri": 7}
CLASSES = {"hex": (8, 20, 27), "tet": (4, 10, 15), "wedge": (6, 15, 21), "quad": (4, 8, 9), "tri": (3, 6, 7)}
NFACES = {"hex": 6, "tet": 4, "wedge": 5, "quad": 4, "tri": 3}
GAMBIT = {(4, 27): "hex", (6, 10): "tet", (5, 18): "wedge", (2, 9): "quad", (3, 6): "tri"}
G2F = {"hex": (4, 16, 0, 15, 23, 11, 7, 19, 3, 12, 20, 8, 25, 26, 24, 14, 22, 10, 5, 17, 1, 13, 21, 9, 6, 18, 2), "tet": (0, 4, 1, 6, 5, 2, 7, 8, 9, 3),
       "wedge": (3, 11, 5, 9, 10, 4, 12, 17, 14, 15, 16, 13, 0, 8, 2, 6, 7, 1), "quad": (0, 4, 1, 5, 2, 6, 3, 7, 8), "tri": (0, 3, 1, 4, 2, 5)}
GFACE = {"hex": (0, 4, 2, 5, 3, 1), "tet": (0, 1, 2, 3), "wedge": (2, 1, 0, 4, 3), "quad": (0, 1, 2, 3), "tri": (0, 1, 2)}
# Mesh.cpp:105-122: weights of the file's nodes in the nodes the file does not hold (tetrahedron: four faces and the centre; prism: two triangles and the centre;
# triangle: the centre)
ADDED = {"tet": np.array([[-1. / 9., -1. / 9., -1. / 9., 0, 4. / 9., 4. / 9., 4. / 9., 0, 0, 0], [-1. / 9., -1. / 9., 0, -1. / 9., 4. / 9., 0, 0, 4. / 9., 4. / 9., 0],
                          [0, -1. / 9., -1. / 9., -1. / 9., 0, 4. / 9., 0, 0, 4. / 9., 4. / 9.], [-1. / 9., 0, -1. / 9., -1. / 9., 0, 0, 4. / 9., 4. / 9., 0, 4. / 9.],
                          [-1. / 8.] * 4 + [1. / 4.] * 6]),
         "wedge": np.array([[-1. / 9.] * 3 + [0.] * 3 + [4. / 9.] * 3 + [0.] * 9, [0.] * 3 + [-1. / 9.] * 3 + [0.] * 3 + [4. / 9.] * 3 + [0.] * 6,
                            [0.] * 12 + [-1. / 9.] * 3 + [4. / 9.] * 3]),
         "tri": np.array([[-1. / 9.] * 3 + [4. / 9.] * 3])}
COMPLETE = ("hex", "quad")                 # shapes whose file elements hold every biquadratic node
_T = {}


def tables(shape):
    """per shape: faces (local nodes of each face, vertices first), nvf (vertices per face), edges (the two vertices of each edge node), EP, f2c"""
    if shape not in _T:
        nv, ne, nl = CLASSES[shape]
        faces = [capi.fe_face_nodes(shape, "biquadratic", f) for f in range(NFACES[shape])]
        nvf = [{9: 4, 7: 3, 3: 2}[len(f)] for f in faces]             # quadrilateral, triangle, line
        x = np.array([capi.fe_node_ref_coords(shape, n) for n in range(nl)])
        edges = [[(a, b) for a in range(nv) for b in range(a + 1, nv) if np.allclose(0.5 * (x[a] + x[b]), x[m])][0] for m in range(nv, ne)]
        EP = capi.fe_elem_prolongator(shape, "biquadratic")
        f2c = np.array([[int(np.argmax(EP[j, v])) for v in range(nv)] for j in range(EP.shape[0])])
        _T[shape] = dict(faces=faces, nvf=nvf, edges=edges, EP=EP, f2c=f2c, face_local=[int(f[-1]) for f in faces])
    return _T[shape]


_first_touch = _mesh_keys.first_touch


def _renumber(kind, raw, nnode):
    new = np.full(nnode, -1, dtype=np.int64)
    col = np.arange(27)[None, :]
    lo = np.zeros((raw.shape[0], 1), dtype=np.int64)
    k, own = 0, []
    for c in range(3):
        hi = np.array([CLASSES[s][c] for s in kind])[:, None]
        seq = raw[(col >= lo) & (col < hi)]                   # element by element, local order
        seq = seq[new[seq] < 0]
        uniq, first = np.unique(seq, return_index=True)
        order = np.argsort(first, kind="stable")
        new[uniq[order]] = k + np.arange(uniq.size)
        k += uniq.size
        own.append(k)
        lo = hi
    return new, own


def _apply(new, raw):
    return np.where(raw >= 0, new[np.maximum(raw, 0)], -1)


def read_gambit(path, Lref=1.0, groups=False):
    tok = open(path).read().split()
    p = tok.index("NDFVL") + 1
    nvt, nel, ngroup, nbcd, dim, dim_nodes = (int(t) for t in tok[p:p + 6])
    if dim not in (2, 3):
        raise ValueError("%s: a %d-dimensional mesh" % (path, dim))
    if dim_nodes != dim:     # GambitIO.cpp:128, 246-270: the nodes carry NDFVL coordinates -- a surface in space (the Willmore / conformal applications)
        raise ValueError("%s: %d-dimensional elements with %d coordinates per node (a surface in space): not served" % (path, dim, dim_nodes))
    p = tok.index("COORDINATES") + 2
    xyz = np.array(tok[p:p + (1 + dim) * nvt], dtype=object).reshape(nvt, 1 + dim)[:, 1:].astype(float) / Lref
    p = tok.index("ELEMENTS/CELLS") + 2
    kind, raw = [], np.full((nel, 27), -1, dtype=np.int64)
    for e in range(nel):
        gt, nn = int(tok[p + 1]), int(tok[p + 2])
        if (gt, nn) not in GAMBIT or (NLOC[GAMBIT[(gt, nn)]] > 9) != (dim == 3):
            raise ValueError("%s: element %d of Gambit type %d with %d nodes: HEX27, TET10, WEDGE18, QUAD9 and TRI6 are served" % (path, e + 1, gt, nn))
        s = GAMBIT[(gt, nn)]
        kind.append(s)
        raw[e, list(G2F[s])] = np.array(tok[p + 3:p + 3 + nn], dtype=np.int64) - 1
        p += 3 + nn
    ff = np.full((nel, 6), -1, dtype=np.int64)
    q = 0
    for _ in range(nbcd):
        q = tok.index("CONDITIONS", q) + 2
        name, nface = int(tok[q]), int(tok[q + 2])
        q += 5
        for k in range(nface):
            e, f = int(tok[q + 3 * k]) - 1, int(tok[q + 3 * k + 2]) - 1
            ff[e, GFACE[kind[e]][f]] = -name - 1
        q += 3 * nface
    # triangle-face nodes, element by element and face by face; then the centres of tetrahedra and prisms
    ent_e, ent_l, keys = [], [], []
    for e in range(nel):
        T = tables(kind[e])
        for f in range(NFACES[kind[e]]):
            if T["nvf"][f] == 3 and dim == 3:
                ent_e.append(e)
                ent_l.append(T["face_local"][f])
                keys.append(sorted(raw[e, T["faces"][f][:3]].tolist()))
    nn = nvt
    if keys:
        ids, _ = _first_touch(np.array(keys))
        raw[ent_e, ent_l] = nn + ids
        nn += int(ids.max()) + 1
    for e in range(nel):
        if kind[e] not in COMPLETE:
            raw[e, NLOC[kind[e]] - 1] = nn
            nn += 1
    coords = np.concatenate([xyz, np.zeros((nn - nvt, dim))])
    for e in range(nel):                                      # element by element: a shared face node keeps the later element's sum
        if kind[e] not in COMPLETE:
            W = ADDED[kind[e]]
            j0 = NLOC[kind[e]] - W.shape[0]
            for j in range(W.shape[0]):
                acc = np.zeros(dim)
                for i in range(j0):                           # the sum in the order of Mesh.cpp:1316-1324
                    acc += coords[raw[e, i]] * W[j][i]
                coords[raw[e, j0 + j]] = acc
    kind = np.array(kind)
    # GambitIO.cpp:290-321: group = the integer on the line under "GROUP:", material = its MATERIAL field; Mesh.cpp:626-690: the elements ordered by
    # (material, group, file index) -- after the added nodes were made in file order, before the nodes are numbered
    group, material = np.ones(nel, dtype=np.int64), np.zeros(nel, dtype=np.int64)
    q = 0
    for _ in range(ngroup):
        q = tok.index("GROUP:", q)
        ngel, mat, name = int(tok[q + 3]), int(tok[q + 5]), int(tok[q + 8])
        ids = np.array(tok[q + 10:q + 10 + ngel], dtype=np.int64) - 1
        group[ids], material[ids] = name, mat
        q += 10 + ngel
    order = np.lexsort((np.arange(nel), group, material))
    kind, raw, ff, group, material = kind[order], raw[order], ff[order], group[order], material[order]
    new, own = _renumber(kind, raw, nn)
    xs = np.empty_like(coords)
    xs[new] = coords
    out = (kind, _apply(new, raw), xs, ff, own)
    return out + (group, material) if groups else out


def refine(kind, ed, xs, ff):
    nel, dim = ed.shape[0], xs.shape[1]
    nch = 8 if dim == 3 else 4
    ck = np.repeat(kind, nch)
    raw = np.full((nch * nel, 27), -1, dtype=np.int64)
    fff = np.full((nch * nel, 6), -1, dtype=np.int64)
    ent = []                                                  # (child, local node, key[4]) of every shared new node
    for s in SHAPES:
        sel = np.nonzero(kind == s)[0]
        if sel.size == 0:
            continue
        T = tables(s)
        nv, ne, nl = CLASSES[s]
        for j in range(nch):
            rows = nch * sel + j
            raw[rows, :nv] = ed[sel][:, T["f2c"][j]]
            for lf in range(NFACES[s]):
                for f in range(NFACES[s]):
                    if T["nvf"][lf] == T["nvf"][f] and all(int(T["f2c"][j][v]) in T["faces"][f].tolist() for v in T["faces"][lf][:T["nvf"][lf]]):
                        fff[rows, lf] = ff[sel, f]
        rows = (nch * sel[:, None] + np.arange(nch)[None, :]).ravel()
        for m, (a, b) in enumerate(T["edges"]):
            va, vb = raw[rows, a], raw[rows, b]
            key = np.stack([np.minimum(va, vb), np.maximum(va, vb), np.full(rows.size, -1), np.full(rows.size, -1)], axis=1)
            ent.append((rows, np.full(rows.size, nv + m), key))
        for f in range(NFACES[s] if dim == 3 else 0):         # (in two dimensions the faces ARE the edges)
            n = T["nvf"][f]
            key = np.sort(raw[rows][:, T["faces"][f][:n]], axis=1)
            if n == 3:                                        # (a, b, c, -2): apart from an edge (a, b, -1, -1) and from a quadrilateral
                key = np.concatenate([key, np.full((rows.size, 1), -2)], axis=1)
            ent.append((rows, np.full(rows.size, T["face_local"][f]), key))
    c = np.concatenate([t[0] for t in ent])
    loc = np.concatenate([t[1] for t in ent])
    key = np.concatenate([t[2] for t in ent])
    order = np.lexsort((loc, c))                              # child by child, local order
    c, loc, key = c[order], loc[order], key[order]
    ids, owner = _first_touch(key)
    nold = xs.shape[0]
    raw[c, loc] = nold + ids
    nshared = owner.size
    centre = np.array([NLOC[s] - 1 for s in ck])
    allc = np.arange(nch * nel)
    raw[allc, centre] = nold + nshared + allc
    oc, ol = np.concatenate([c[owner], allc]), np.concatenate([loc[owner], centre])         # creating (child, local node) of every new node
    pos = np.zeros((oc.size, dim))
    for s in SHAPES:
        m = np.nonzero(ck[oc] == s)[0]
        if m.size:
            EP = tables(s)["EP"]
            for k in range(NLOC[s]):
                pos[m] += EP[oc[m] % nch, ol[m], k][:, None] * xs[ed[oc[m] // nch, k]]
    coords = np.concatenate([xs, pos])
    new, own = _renumber(ck, raw, coords.shape[0])
    used = new >= 0
    xf = np.empty((own[2], dim))
    xf[new[used]] = coords[used]
    return ck, _apply(new, raw), xf, fff, own
