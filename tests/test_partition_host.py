"""General (non-box) domain decomposition on the host: native partitioner, sub-meshes, topological node keys (the
reference: METIS_PartMeshDual on the coarsest level + inherited children, MeshMetisPartitioning.cpp:71-113, 143-155)."""
import os
import numpy as np
import pytest

from femus_amd import capi

REF_INPUT = "/root/reference/applications/001_Poisson/input/cube_Hex.neu"


def shuffled_box(n, seed):
    """a box mesh whose elements come in a random order (sub-mesh of itself): nothing about it is structured any more"""
    g = capi.Mesh.box(*n)
    rng = np.random.default_rng(seed)
    perm = rng.permutation(g.nel).astype(np.int32)
    sub, _ = g.submesh(perm)
    return sub


@pytest.mark.parametrize("n,nparts", [((4, 4, 4), 2), ((4, 4, 4), 4), ((5, 3, 2), 3), ((6, 6, 0), 4), ((3, 3, 3), 8)])
def test_partition_is_balanced_and_connected(n, nparts):
    g = shuffled_box(n, 3)
    part = g.partition(nparts)
    cnt = np.bincount(part, minlength=nparts)
    assert cnt.sum() == g.nel and cnt.max() - cnt.min() <= 1
    ed, _, _ = g.arrays()
    f0, f1 = (20, 26) if g.dim == 3 else (4, 8)
    for p in range(nparts if nparts <= 4 else 0):        # few parts of a box: every part is face-connected (not guaranteed in general)
        els = np.where(part == p)[0]
        seen, todo = {els[0]}, [els[0]]
        face_nodes = {e: set(ed[e, f0:f1]) for e in els}
        while todo:
            e = todo.pop()
            for f in els:
                if f not in seen and face_nodes[e] & face_nodes[f]:
                    seen.add(f)
                    todo.append(f)
        assert len(seen) == els.size


@pytest.mark.parametrize("n,nparts,nlev", [((4, 4, 4), 2, 3), ((3, 2, 2), 3, 2), ((4, 4, 0), 4, 3)])
def test_topological_keys_agree_with_coordinates(n, nparts, nlev):
    """on a (shuffled) box the nodes are identified by their exact dyadic coordinates as well: two nodes of two ranks' sub-meshes have the
    same key iff they are the same point, every rank finds the same owner for it, and the owner is the lowest rank touching the node"""
    g = shuffled_box(n, 7)
    part = g.partition(nparts)
    ed_g, xy_g, _ = g.arrays()
    scale = 2 ** (nlev + 1) * np.array([max(v, 1) for v in n], dtype=float)
    seen = {}                                            # point -> (key, owner)
    lowest = {}                                          # point -> lowest rank with an element containing it (from the ranks' owned elements)
    for r in range(nparts):
        own, ring = g.rank_elements(part, r)
        els = np.concatenate([own, ring])
        sub, node_gid = g.submesh(els)
        assert np.array_equal(xy_g[node_gid], sub.arrays()[1])
        levels = [sub]
        for _ in range(nlev - 1):
            levels.append(levels[-1].refine())
        for l in range(nlev):
            gid, owner = g.topo_node_keys(part, levels, els, l)
            ed, xy, _ = levels[l].arrays()
            pts = [tuple(v) for v in np.rint(xy * scale[:g.dim]).astype(np.int64)]
            assert len(set(gid.tolist())) == gid.size                     # distinct nodes, distinct keys
            for k in range(gid.size):
                key = (l,) + pts[k]
                if key in seen:
                    assert seen[key] == (gid[k], owner[k])
                else:
                    seen[key] = (gid[k], owner[k])
            # lowest rank touching a node: walk the elements of this level that descend from OWNED coarse elements
            nch = 2 ** g.dim
            n_own_fine = own.size * nch ** l
            for e in range(n_own_fine):
                for nd in ed[e]:
                    key = (l,) + pts[nd]
                    lowest[key] = min(lowest.get(key, nparts), r)
    for key, (gid, owner) in seen.items():
        if key in lowest:
            assert owner == lowest[key]
    keys_by_level = {}
    for (l, *_), (gid, _) in seen.items():
        keys_by_level.setdefault(l, []).append(gid)
    for l, ks in keys_by_level.items():
        assert len(set(ks)) == len(ks)                                     # different points, different keys


@pytest.mark.skipif(not os.path.exists(REF_INPUT), reason="reference inputs are only present in the build container")
def test_gambit_mesh_can_be_partitioned_and_cut():
    g = capi.Mesh.read_gambit(REF_INPUT)
    part = g.partition(2)
    assert sorted(np.bincount(part).tolist()) == [4, 4]
    own, ring = g.rank_elements(part, 0)
    assert own.size == 4 and ring.size == 4
    sub, node_gid = g.submesh(np.concatenate([own, ring]))
    assert sub.nel == 8 and sub.nnode == g.nnode


def test_partition_on_random_disconnected_meshes():
    """random subsets of box elements in random order (disconnected pieces, single elements, 2-D and 3-D), part counts from 1 to the number
    of elements: every part non-empty, sizes within one element of each other, the rank's ring disjoint from its own elements"""
    rng = np.random.default_rng(0)
    for trial in range(25):
        n = (int(rng.integers(1, 6)), int(rng.integers(1, 6)), int(rng.integers(1, 5)) if rng.integers(0, 2) else 0)
        g = capi.Mesh.box(*n)
        k = int(rng.integers(1, g.nel + 1))
        sub, _ = g.submesh(rng.permutation(g.nel)[:k].astype(np.int32))
        for nparts in sorted({1, 2, 3, int(rng.integers(1, sub.nel + 1)), sub.nel}):
            if nparts > sub.nel:
                continue
            part = sub.partition(nparts)
            cnt = np.bincount(part, minlength=nparts)
            assert cnt.sum() == sub.nel and cnt.min() >= 1 and cnt.max() - cnt.min() <= 1 and part.min() >= 0 and part.max() < nparts
            for r in range(min(nparts, 3)):
                own, ring = sub.rank_elements(part, r)
                assert set(own.tolist()) == set(np.where(part == r)[0].tolist())
                assert not (set(own.tolist()) & set(ring.tolist()))


# ---- element weights: adaptive levels on general partitions ------------------------------------------------------------------
@pytest.mark.parametrize("n,nparts", [((6, 4, 2), 2), ((6, 4, 2), 3), ((5, 5, 0), 4), ((4, 4, 4), 8)])
def test_weighted_partition_balances_the_weights(n, nparts):
    """weights as adaptive refinement leaves them (1, 8 or 64 descendants, the heavy elements in one corner): every part gets its share
    of the WEIGHT to within one heavy element, no part is empty, and unit weights reproduce the unweighted partition"""
    g = shuffled_box(n, 5)
    xc = g.elem_centroids()
    w = np.where(xc[:, 0] > 0.5, np.where(xc[:, 1] > 0.5, 64.0, 8.0), 1.0)
    part = g.partition(nparts, w)
    assert part.min() == 0 and part.max() == nparts - 1
    sums = np.bincount(part, weights=w, minlength=nparts)
    assert np.bincount(part, minlength=nparts).min() >= 1
    # recursive bisection: every cut is within half a heavy element of its target, log2(nparts) cuts deep
    assert sums.max() - sums.min() <= 64.0 * np.ceil(np.log2(nparts)) + 1e-9, sums
    cnt = np.bincount(g.partition(nparts), weights=w, minlength=nparts)
    assert sums.max() <= cnt.max()                       # never worse than ignoring the weights
    assert np.array_equal(g.partition(nparts, np.ones(g.nel)), g.partition(nparts))
    with pytest.raises(RuntimeError):
        g.partition(nparts, np.zeros(g.nel))


class _OneRank:
    """stands for the rendezvous of a one-rank job"""

    def allgather_obj(self, o):
        return [o]


def _flag(x, level):
    return x[0] > 0.5 and (level < 2 or x[1] > 0.25)


def test_amr_weights_count_the_finest_descendants():
    """the slice-wise count equals the count on the refined global mesh, whatever the number of slices"""
    from femus_amd import dd
    g = shuffled_box((4, 3, 2), 9)
    nlev, n_uniform = 4, 2
    ms = dd.refine_levels(g, nlev, _flag, n_uniform)
    anc = np.arange(g.nel)
    for l in range(nlev - 1):
        ch = ms[l].child_elems()
        nxt = np.empty(ms[l + 1].nel, dtype=np.int64)
        ok = ch >= 0
        nxt[ch[ok]] = np.broadcast_to(anc[:, None], ch.shape)[ok]
        anc = nxt
    ref = np.bincount(anc, minlength=g.nel)
    assert ref.min() == 8 and ref.max() == 512 and ref.sum() == ms[-1].nel      # level 1 uniform, then up to two selective refinements
    for world in (1, 3, 5):
        got = [dd.amr_slice_weights(g, (g.nel * r) // world, (g.nel * (r + 1)) // world, nlev, _flag, n_uniform) for r in range(world)]
        assert np.array_equal(np.concatenate(got), ref)
    assert np.array_equal(dd.amr_element_weights(g, _OneRank(), 1, 0, nlev, _flag, n_uniform), ref.astype(float))


@pytest.mark.parametrize("nparts", [2, 3])
def test_topological_keys_on_adaptive_levels(nparts):
    """selectively refined levels of the ranks' sub-meshes: a node has the same key on every rank that sees it, the keys of a level are
    distinct points, and the union of the owned nodes of all ranks is the node set of the serially refined mesh"""
    from femus_amd import dd
    g = shuffled_box((4, 3, 2), 13)
    nlev, n_uniform = 4, 2
    ser = dd.refine_levels(g, nlev, _flag, n_uniform)
    one = _OneRank()
    w = dd.amr_element_weights(g, one, 1, 0, nlev, _flag, n_uniform)
    part = g.partition(nparts, w)
    for l in (2, 3):
        key_of = {}
        owned = 0
        for r in range(nparts):
            own, ring = g.rank_elements(part, r)
            els = np.concatenate([own, ring]).astype(np.int32)
            sub, _ = g.submesh(els)
            levels = dd.refine_levels(sub, nlev, _flag, n_uniform)
            gid, owner = g.topo_node_keys(part, levels, els, l)
            xy = levels[l].arrays()[1]
            assert np.unique(gid).size == gid.size
            for k, o, x in zip(gid, owner, np.rint(xy * 4096).astype(np.int64)):
                pt = tuple(x)
                if pt in key_of:
                    assert key_of[pt] == (k, o)
                else:
                    key_of[pt] = (k, o)
            owned += int(np.sum(owner == r))
        assert len(set(v[0] for v in key_of.values())) == len(key_of)
        # every node of the serial level is owned exactly once (ring nodes of other owners may go beyond what a rank needs)
        pts = set(tuple(x) for x in np.rint(ser[l].arrays()[1] * 4096).astype(np.int64))
        assert pts <= set(key_of) and owned == len(pts)


def _cut_faces(g, part):
    """faces of the dual graph between different parts (two elements sharing a face-centre node)"""
    ed, _, _ = g.arrays()
    f0, f1 = (20, 26) if g.dim == 3 else (4, 8)
    first, cut = {}, 0
    for e in range(g.nel):
        for nd in ed[e, f0:f1]:
            if nd in first:
                cut += int(part[first[nd]] != part[e])
            else:
                first[nd] = e
    return cut


@pytest.mark.parametrize("n,nparts,best", [((8, 8, 8), 2, 64), ((8, 8, 8), 8, 192), ((16, 8, 4), 4, 96), ((12, 12, 0), 4, 24)])
def test_partition_of_a_block_cuts_planes(n, nparts, best):
    """the inertial candidate of every bisection: a compact block is cut by planes across its short ways (the breadth-first level sets
    alone run diagonally and cut about twice as many faces: 132 instead of 64 for the halved 8^3 box) -- the halo of every rank follows"""
    g = shuffled_box(n, 1)
    part = g.partition(nparts)
    assert _cut_faces(g, part) == best
    cnt = np.bincount(part, minlength=nparts)
    assert cnt.max() - cnt.min() <= 1
    assert np.array_equal(part, g.partition(nparts))              # deterministic


def test_partition_follows_a_bent_domain():
    """an L-shaped domain (a box with one quadrant removed): the graph-based candidate is still there, the parts stay balanced and connected"""
    box = capi.Mesh.box(8, 8, 0)
    xc = box.elem_centroids()
    keep = np.where(~((xc[:, 0] > 0.5) & (xc[:, 1] > 0.5)))[0].astype(np.int32)
    g, _ = box.submesh(keep)
    for nparts in (2, 3):
        part = g.partition(nparts)
        cnt = np.bincount(part, minlength=nparts)
        assert cnt.max() - cnt.min() <= 1
        ed, _, _ = g.arrays()
        for p in range(nparts):
            els = np.where(part == p)[0]
            nodes = {e: set(ed[e, 4:8]) for e in els}
            seen, todo = {els[0]}, [els[0]]
            while todo:
                e = todo.pop()
                for f in els:
                    if f not in seen and nodes[e] & nodes[f]:
                        seen.add(f)
                        todo.append(f)
            assert len(seen) == els.size
