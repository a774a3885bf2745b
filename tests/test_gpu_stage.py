"""GPU: the staged form of add_matrix_blocked / add_vector_blocked (PetscMatrix.cpp:699-729, PetscVector.cpp:132-153).
Blocks staged in the pinned ring and applied by one kernel per ring must give the BITS of adding them one after the other in call
order -- the reference below is exactly that sequential loop in numpy (float64 adds in the order of the calls)."""
import os
import subprocess

import numpy as np
import pytest
import scipy.sparse as sp

from femus_amd.capi import FemusHipError
from oracle import femus_oracle as fo

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ADAPTERS = os.path.join(ROOT, "femus_amd", "csrc", "adapters")
INC = ["-I" + os.path.join(ADAPTERS, "mirror"), "-I" + ADAPTERS, "-I" + os.path.join(ROOT, "include")]


def sequential_adds(pattern, blocks):
    """A[r, c] += v, entry by entry in call order (float64), on a fixed pattern; returns the CSR value array"""
    rp, col = pattern.indptr, pattern.indices
    val = np.zeros(col.size)
    for rows, cols, vals in blocks:
        vals = np.asarray(vals, float).reshape(len(rows), len(cols))
        for i, r in enumerate(rows):
            seg = col[rp[r]:rp[r + 1]]
            pos = np.searchsorted(seg, cols)
            ok = (pos < seg.size) & (seg[np.minimum(pos, seg.size - 1)] == cols)
            # distinct columns inside a block: one vectorised add per block row keeps the order across blocks
            np.add.at(val, rp[r] + pos[ok], vals[i][ok])
    return val


def random_blocks(rng, pattern, nblocks, nr, ncmax):
    rp, col = pattern.indptr, pattern.indices
    m = pattern.shape[0]
    out = []
    for _ in range(nblocks):
        rows = rng.integers(0, m, size=nr)
        # columns present in ALL chosen rows would be rare on a random pattern: take them from the first row and zero the values
        # of the (row, col) pairs outside the pattern -- the reference does the same with its Dirichlet-free blocks
        c0 = col[rp[rows[0]]:rp[rows[0] + 1]]
        cols = rng.choice(c0, size=min(ncmax, c0.size), replace=False) if c0.size else np.zeros(0, np.int32)
        vals = rng.standard_normal((nr, cols.size)) * 10.0 ** rng.integers(-8, 8)
        for i, r in enumerate(rows):
            seg = col[rp[r]:rp[r + 1]]
            vals[i][~np.isin(cols, seg)] = 0.0
        out.append((rows.astype(np.int32), cols.astype(np.int32), vals))
    return out


@pytest.mark.parametrize("seed,m,density,nblocks", [(0, 300, 0.2, 400), (1, 50, 0.9, 2000), (2, 5000, 0.004, 300)])
def test_staged_blocks_have_the_bits_of_sequential_adds(ctx, seed, m, density, nblocks):
    rng = np.random.default_rng(seed)
    P = sp.random(m, m, density=density, random_state=seed, format="csr")
    P = (P + sp.eye(m)).tocsr()
    P.sort_indices()
    blocks = random_blocks(rng, P, nblocks, nr=7, ncmax=20)
    A = ctx.matrix_csr(m, m, P.indptr, P.indices)
    for rows, cols, vals in blocks:
        A.stage_matrix_blocked(vals, rows, cols)
    A.flush()
    ref = sequential_adds(P, blocks)
    assert np.array_equal(A.values(), ref)
    # the same blocks through the immediate call
    B = ctx.matrix_csr(m, m, P.indptr, P.indices)
    for rows, cols, vals in blocks:
        B.add_matrix_blocked(vals, rows, cols)
    assert np.array_equal(B.values(), ref)
    A.destroy()
    B.destroy()


def test_rows_longer_than_the_lds_tile_and_rings_that_fill_up(ctx):
    """a 3000-entry row (global-memory variant of the flush kernel) and enough data to send several rings"""
    rng = np.random.default_rng(5)
    m = 3200
    dense_row = np.arange(0, m, dtype=np.int32)[:3000]
    rows = [dense_row] + [np.unique(rng.integers(0, m, size=12)).astype(np.int32) for _ in range(m - 1)]
    rp = np.zeros(m + 1, np.int32)
    rp[1:] = np.cumsum([r.size for r in rows])
    col = np.concatenate(rows).astype(np.int32)
    P = sp.csr_matrix((np.ones(col.size), col, rp), shape=(m, m))
    A = ctx.matrix_csr(m, m, rp, col)
    blocks = []
    for k in range(40):
        r = np.array([0, 0, int(rng.integers(1, m))], np.int32)          # the long row twice in one block, and a short one
        c = rng.choice(dense_row, size=700, replace=False).astype(np.int32)
        v = rng.standard_normal((3, c.size))
        v[2][~np.isin(c, col[rp[r[2]]:rp[r[2] + 1]])] = 0.0
        blocks.append((r, c, v))
    for rows_, cols_, vals_ in blocks:
        A.stage_matrix_blocked(vals_, rows_, cols_)
    A.flush()
    assert np.array_equal(A.values(), sequential_adds(P, blocks))
    A.destroy()
    # many rings: 27x27 blocks on a Q2 pattern, far more than one ring holds (4 M doubles)
    ms = fo.build_levels(2, 2, 2, 3)
    Aq, _ = fo.assemble_poisson(ms[-1], "biquadratic", lambda xg: np.ones(xg.shape[:2]))
    Aq = Aq.tocsr()
    Aq.sort_indices()
    edof = fo.elem_sys_dof(ms[-1], "biquadratic")
    K = rng.standard_normal((edof.shape[0], 27, 27))
    M = ctx.matrix_csr(Aq.shape[0], Aq.shape[1], Aq.indptr, Aq.indices)
    reps = 14                                                   # 512 elements x 14 = 7168 blocks of 729 doubles > one ring
    blocks = [(edof[e].astype(np.int32), edof[e].astype(np.int32), K[e] * (1 + rep)) for rep in range(reps) for e in range(edof.shape[0])]
    for rows_, cols_, vals_ in blocks:
        M.stage_matrix_blocked(vals_, rows_, cols_)
    M.flush()
    nb, nrings = M.stage_stats()
    assert nb == len(blocks) and nrings >= 2
    assert np.array_equal(M.values(), sequential_adds(Aq, blocks))
    M.destroy()


def test_entry_outside_the_pattern_is_reported_by_the_flush(ctx):
    rp = np.array([0, 2, 3, 5], np.int32)
    col = np.array([0, 1, 1, 0, 2], np.int32)
    A = ctx.matrix_csr(3, 3, rp, col)
    A.stage_matrix_blocked([[1.0, 2.0]], [1], [1, 2])          # (1,2) is not in the pattern
    with pytest.raises(FemusHipError, match=r"entry \(1,2\) is outside the pattern"):
        A.flush()
    A.stage_matrix_blocked([[1.0, 0.0]], [1], [1, 2])          # a zero outside the pattern is accepted (PETSc would ignore it too)
    A.flush()
    assert A.values().tolist() == [0.0, 0.0, 2.0, 0.0, 0.0]    # the valid part of the first block was added as well
    with pytest.raises(FemusHipError, match="row 3 out of range"):
        A.stage_matrix_blocked([[1.0]], [3], [0])
    with pytest.raises(FemusHipError, match="column 7 out of range"):
        A.stage_matrix_blocked([[1.0]], [0], [7])
    A.flush()                                                  # nothing staged: a no-op
    A.destroy()


def test_staged_vector_adds_in_call_order_with_ghosts_and_repeats(ctx):
    rng = np.random.default_rng(3)
    n_local, first = 1000, 500
    ghosts = np.array([3, 17, 2400, 2050], np.int32)
    v = ctx.vector(3000, n_local, first, ghosts)
    ref = np.zeros(n_local + ghosts.size)
    glob = np.concatenate([np.arange(first, first + n_local), ghosts])
    for _ in range(300):
        k = int(rng.integers(1, 40))
        loc = rng.integers(0, glob.size, size=k)
        vals = rng.standard_normal(k) * 10.0 ** rng.integers(-6, 6)
        v.stage_vector_blocked(vals, glob[loc])
        for p, x in zip(loc, vals):
            ref[p] += x
    v.flush()
    # adds to OWNED entries are applied; adds to ghost entries wait beside the vector for their owners (VecAssemblyBegin/End ships the stash of
    # off-process ADD_VALUES, PetscVector.cpp:131-153): the ghost copies themselves are untouched
    own = np.arange(n_local)
    assert np.array_equal(v.get(glob[own]), ref[own])
    assert np.array_equal(v.get(ghosts), np.zeros(ghosts.size))
    assert np.array_equal(v.ghost_adds(ghosts.size), ref[n_local:])
    with pytest.raises(FemusHipError, match="neither owned nor a ghost"):
        v.stage_vector_blocked([1.0, 2.0], [first, 4])
    v.flush()
    assert np.array_equal(v.get(glob[own]), ref[own])          # the failing call left nothing behind
    assert np.array_equal(v.ghost_adds(ghosts.size), ref[n_local:])
    # more values than one ring holds (1 M): several rings, same order
    big = ctx.vector(64)
    idx = rng.integers(0, 64, size=2_300_000).astype(np.int32)
    vals = rng.standard_normal(idx.size)
    big.stage_vector_blocked(vals, idx)
    big.flush()
    ref = np.zeros(64)
    np.add.at(ref, idx, vals)                                  # np.add.at adds in index order of the operands = call order
    assert np.array_equal(big.to_numpy(), ref)


def build_app(tmp_path, name):
    lib = os.path.join(ROOT, "femus_amd", "lib")
    exe = str(tmp_path / name)
    subprocess.check_call(["make", "-C", ADAPTERS], stdout=subprocess.DEVNULL)
    subprocess.check_call(["g++", "-O2", "-std=c++17"] + INC + [os.path.join(ROOT, "tests", "cpp", name + ".cpp"), "-o", exe,
                           "-L" + lib, "-lfemus_hip_adapters", "-lfemus_hip", "-Wl,-rpath," + lib])
    return exe


def parse(log):
    out = {}
    for line in log.splitlines():
        w = line.split()
        if len(w) >= 2 and w[0] in ("first_loop_s", "second_loop_s", "immediate_loop_s", "bit_identical"):
            out[w[0]] = float(w[1])
    return out


@pytest.mark.parametrize("n", [(6, 5, 4), (9, 7, 0)])
def test_unchanged_element_loop_through_the_adapters_small(tmp_path, n):
    exe = build_app(tmp_path, "element_loop_adapters")
    log = subprocess.check_output([exe] + [str(k) for k in n], text=True)
    assert parse(log)["bit_identical"] == 1, log


def test_unchanged_element_loop_at_32_cubed_is_fast_and_bit_identical(tmp_path):
    """32 768 HEX27 elements through KK->add_matrix_blocked / RES->add_vector_blocked, one call per element, on the frozen pattern:
    under half a second including close() (it was a malloc + two copies + a launch + a synchronisation per element), and the same
    bits as adding the elements one after the other"""
    exe = build_app(tmp_path, "element_loop_adapters")
    log = subprocess.check_output([exe, "32", "32", "32"], text=True)
    r = parse(log)
    print(log)
    assert r["bit_identical"] == 1, log
    assert r["second_loop_s"] < 0.5, log
