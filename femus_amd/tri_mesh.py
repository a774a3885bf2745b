"""Triangle meshes of applications/001_Poisson on the host (integers and coordinates only; all numerics run in libfemus_hip.so): the TRI6 box of the application's
generator, the seventh node FEMuS adds, refinement, numbering.

    box      MeshGeneration.cpp:283-650 (case 2, TRI6: lattice node i + j (2 nx + 1), two triangles per cell, faces named bottom / right / top / left = flags
             -2 .. -5), Mesh::AddBiquadraticNodesNotInMeshFile (Mesh.cpp:1207-1333; centre = -1/9 of the vertices + 4/9 of the middles, Mesh.cpp:124)
    refine   MeshRefinement::RefineMesh: children 4 e + j, their vertices through tri_lag::fine2CoarseVertexMapping (read off the element prolongator the library
             builds from it), new middles shared between neighbours, a centre per child, coordinates by the biquadratic element prolongator
    numbering  every FEMuS mesh: vertices, then middles, then centres, each class in order of first appearance walking the elements
"""
import numpy as np

from . import capi


def _renumber(raw, nnode):
    new = np.full(nnode, -1, dtype=np.int64)
    k, own = 0, []
    for lo, hi in ((0, 3), (3, 6), (6, 7)):
        seq = raw[:, lo:hi].ravel()
        seq = seq[new[seq] < 0]
        uniq, first = np.unique(seq, return_index=True)
        order = np.argsort(first, kind="stable")
        new[uniq[order]] = k + np.arange(uniq.size)
        k += uniq.size
        own.append(k)
    return new, own


def box(nx, ny, lo, hi):
    px = 2 * nx + 1
    jj, ii = np.meshgrid(np.arange(2 * ny + 1), np.arange(px), indexing="ij")
    xy = np.stack([(ii.ravel() / (2.0 * nx)) * (hi[0] - lo[0]) + lo[0], (jj.ravel() / (2.0 * ny)) * (hi[1] - lo[1]) + lo[1]], axis=1)
    idx = lambda i, j: i + j * px
    ed, ff = [], []
    for j in range(0, 2 * ny, 2):
        for i in range(0, 2 * nx, 2):
            ed.append([idx(i, j), idx(i + 2, j), idx(i + 2, j + 2), idx(i + 1, j), idx(i + 2, j + 1), idx(i + 1, j + 1)])
            ff.append([-2 if j == 0 else -1, -3 if i == 2 * (nx - 1) else -1, -1])
            ed.append([idx(i, j), idx(i + 2, j + 2), idx(i, j + 2), idx(i + 1, j + 1), idx(i + 1, j + 2), idx(i, j + 1)])
            ff.append([-1, -4 if j == 2 * (ny - 1) else -1, -5 if i == 0 else -1])
    ed = np.array(ed, dtype=np.int64)
    nel, n6 = ed.shape[0], xy.shape[0]
    raw = np.concatenate([ed, (n6 + np.arange(nel))[:, None]], axis=1)
    wts = np.array([-1. / 9., -1. / 9., -1. / 9., 4. / 9., 4. / 9., 4. / 9.])
    centres = np.zeros((nel, 2))
    for i in range(6):                                   # the sum in the order of Mesh.cpp:1316-1324
        centres += xy[ed[:, i]] * wts[i]
    coords = np.concatenate([xy, centres])
    new, own = _renumber(raw, coords.shape[0])
    xs = np.empty_like(coords)
    xs[new] = coords
    return new[raw], xs, np.array(ff, dtype=np.int64), own


def refine(ed, xs, ff):
    nel = ed.shape[0]
    EP = capi.fe_elem_prolongator("tri", "biquadratic")                  # [child][local node][coarse function]
    f2c = np.array([[int(np.argmax(EP[j, v])) for v in range(3)] for j in range(4)])
    edges = [capi.fe_face_nodes("tri", "biquadratic", f) for f in range(3)]     # (end, end, middle) of local edge f
    raw = np.full((4 * nel, 7), -1, dtype=np.int64)
    fff = np.full((4 * nel, 3), -1, dtype=np.int64)
    coords = [xs]
    nnew = xs.shape[0]
    for j in range(4):
        raw[j::4, :3] = ed[:, f2c[j]]
        if j < 3:
            for f in range(3):
                if j in (int(edges[f][0]), int(edges[f][1])):             # vertex j lies on face f: the child carries the flag on the same local face
                    fff[j::4, f] = ff[:, f]
    # middles of the children's edges: one node per pair of fine vertices, created at the first element and edge that holds it
    a = np.stack([raw[:, int(edges[f][0])] for f in range(3)], axis=1)
    b = np.stack([raw[:, int(edges[f][1])] for f in range(3)], axis=1)
    key = (np.minimum(a, b) * np.int64(xs.shape[0]) + np.maximum(a, b)).ravel()
    uniq, first, inv = np.unique(key, return_index=True, return_inverse=True)
    rank = np.empty(uniq.size, dtype=np.int64)
    rank[np.argsort(first, kind="stable")] = np.arange(uniq.size)
    raw[:, 3:6] = (nnew + rank[inv]).reshape(-1, 3)
    owner = np.empty(uniq.size, dtype=np.int64)
    owner[rank] = first                                                  # (child element * 3 + local edge) that created the node
    c, k = owner // 3, owner % 3
    mid = np.zeros((uniq.size, 2))
    for m in range(7):
        mid += EP[c % 4, 3 + k, m][:, None] * xs[ed[c // 4, m]]
    coords.append(mid)
    nnew += uniq.size
    raw[:, 6] = nnew + np.arange(4 * nel)
    cen = np.zeros((4 * nel, 2))
    ch = np.arange(4 * nel)
    for m in range(7):
        cen += EP[ch % 4, 6, m][:, None] * xs[ed[ch // 4, m]]
    coords.append(cen)
    coords = np.concatenate(coords)
    new, own = _renumber(raw, coords.shape[0])
    used = new >= 0
    xf = np.empty((own[2], 2))
    xf[new[used]] = coords[used]
    return new[raw], xf, fff, own
