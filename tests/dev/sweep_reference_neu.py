"""Dev: every Gambit file of the reference tree that the host-side reader serves (any mix of HEX27 / TET10 / WEDGE18 in three dimensions, QUAD9 / TRI6 in two, pure
HEX27 / QUAD9 files excepted: the library's own reader takes those) through femus_amd/mixed_mesh.py: read, one refinement, positive Jacobians at every Gauss point
of every element on both levels, the same measure on both levels, boundary faces times 4 (edges times 2).  Runs where /root/reference exists.
usage: python tests/dev/sweep_reference_neu.py [max file size in bytes]"""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from femus_amd import mixed_mesh as mm, capi

TAB = {}


def measure(kind, ed, xs):
    tot, worst = 0.0, np.inf
    for s in set(kind.tolist()):
        if s not in TAB:
            w, _ = capi.fe_gauss(s, "seventh")
            _, dphi = capi.fe_tables(s, "biquadratic", "seventh")
            TAB[s] = (w, dphi)
        w, dphi = TAB[s]
        sel = np.nonzero(kind == s)[0]
        x = xs[ed[sel][:, :mm.NLOC[s]]]                                  # [ne, nc, dim]
        J = np.einsum("gnp,enq->egpq", dphi, x)
        det = np.linalg.det(J)
        worst = min(worst, det.min())
        tot += float((det * w[None, :]).sum())
    return tot, worst


def main():
    cap = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
    seen, res = {}, collections.Counter()
    for root, _, files in os.walk("/root/reference"):
        for f in sorted(files):
            if not f.endswith(".neu"):
                continue
            p = os.path.join(root, f)
            if os.path.getsize(p) > cap:
                continue
            raw = open(p, "rb").read()
            key = hash(raw)
            if key in seen:
                continue
            seen[key] = p
            tok = raw.decode(errors="ignore").split()
            try:
                q = tok.index("NDFVL") + 1
                nel = int(tok[q + 1])
                k = tok.index("ELEMENTS/CELLS") + 2
                types = set()
                for _ in range(nel):
                    types.add((int(tok[k + 1]), int(tok[k + 2])))
                    k += 3 + int(tok[k + 2])
            except Exception:
                res["unreadable header"] += 1
                continue
            if types <= {(4, 27)} or types <= {(2, 9)}:
                continue
            label = "+".join(sorted(mm.GAMBIT.get(t, "?%d-%d" % t) for t in types))
            try:
                t0 = time.time()
                m = mm.read_gambit(p)
                v0, w0 = measure(m[0], m[1], m[2])
                nb0 = int((m[3] < -1).sum())
                m1 = mm.refine(*m[:4])
                v1, w1 = measure(m1[0], m1[1], m1[2])
                nb1 = int((m1[3] < -1).sum())
                dim = m[2].shape[1]
                ok = abs(v1 - v0) <= 1e-9 * abs(v0) and nb1 == nb0 * (4 if dim == 3 else 2) and v0 > 0
                # (two vascular meshes of 005_FSI hold elements whose Jacobian is negative at a Gauss point of the FILE's level: the files', not the reader's)
                res[(label, ("ok" if w0 > 0 and w1 > 0 else "ok, the file holds an element with a negative Jacobian at a Gauss point") if ok else "FAILED")] += 1
                if not ok:
                    print("FAILED", p, v0, v1, w0, w1, nb0, nb1, flush=True)
            except Exception as e:
                res[(label, "error: " + str(e)[:60])] += 1
                print("ERROR", p, repr(e)[:200], flush=True)
    for k, v in sorted(res.items(), key=str):
        print(v, k)


if __name__ == "__main__":
    main()
