"""First-touch identifiers for the shared nodes of a refined mesh (host-side mesh modules): one id per distinct key row, numbered in order of first appearance."""
import numpy as np


def first_touch(keys):
    """keys[n, w] integers (>= -2: -1 / -2 pad a short key).  Returns (id per row, index of the creating row per id).  The columns are packed into as few 64-bit
    words as their range allows (an edge or a triangle of a mesh below two million nodes is one word) and the words sorted; equal keys keep their order."""
    keys = np.asarray(keys, dtype=np.int64)
    n, w = keys.shape
    base = int(keys.max()) + 3 if n else 3
    words, cur, room = [], None, 1
    for c in range(w):
        col = keys[:, c] + 2
        if cur is not None and room * base < (1 << 62):
            cur = cur * base + col
            room *= base
        else:
            if cur is not None:
                words.append(cur)
            cur, room = col, base
    words.append(cur)
    if len(words) == 1:
        order = np.argsort(words[0], kind="stable")
        sw = words[0][order]
        new = np.ones(n, dtype=bool)
        new[1:] = sw[1:] != sw[:-1]
    else:
        order = np.lexsort(words[::-1])
        new = np.ones(n, dtype=bool)
        diff = np.zeros(n - 1, dtype=bool) if n > 1 else np.zeros(0, dtype=bool)
        for wd in words:
            sw = wd[order]
            diff |= sw[1:] != sw[:-1]
        new[1:] = diff
    group = np.cumsum(new) - 1
    first = order[new]                                   # the creating row of every group (stable sort: the smallest index of the group)
    rank = np.empty(first.size, dtype=np.int64)
    rank[np.argsort(first, kind="stable")] = np.arange(first.size)
    ids = np.empty(n, dtype=np.int64)
    ids[order] = rank[group]
    owner = np.empty(first.size, dtype=np.int64)
    owner[rank] = first
    return ids, owner
