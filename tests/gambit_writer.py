"""test helper: writes a HEX27 / QUAD9 mesh as a Gambit neutral file (the format GambitIO::read parses), with Gambit's own local
node order (27-node brick: xi slowest, eta, zeta fastest and descending; 9-node quadrilateral: perimeter then centre) and face
numbers.  Independent of the reader under test: the orders are produced from the reference-element coordinates here."""
import numpy as np

from oracle import femus_oracle as fo

HEX_FACE_FEMUS_TO_GAMBIT = {0: 1, 4: 2, 2: 3, 5: 4, 3: 5, 1: 6}     # inverse of Gambit face k -> FEMuS face (0,4,2,5,3,1)


def gambit_local_order(geom):
    Xc = fo.xc_table(geom)
    if geom == "hex":
        order = []
        for a in (-1, 0, 1):
            for b in (-1, 0, 1):
                for c in (1, 0, -1):
                    order.append(int(np.where((Xc == (a, b, c)).all(axis=1))[0][0]))
        return order
    return [0, 4, 1, 5, 2, 6, 3, 7, 8]


def write_neu(path, geom, elem_dof, coords, face_flag, group=5, material=2, groups=None):
    """groups: optional list of (group name, material, element ids) replacing the single group"""
    nel, nl = elem_dof.shape
    dim = coords.shape[1]
    sets = sorted(set(int(-f - 1) for f in face_flag[face_flag < -1].ravel()))
    order = gambit_local_order(geom)
    with open(path, "w") as f:
        f.write("        CONTROL INFO 2.3.16\n** GAMBIT NEUTRAL FILE\ntest\nPROGRAM:                Gambit     VERSION:  2.3.16\n1 Jan 2000    00:00:00\n")
        f.write("     NUMNP     NELEM     NGRPS    NBSETS     NDFCD     NDFVL\n")
        glist = groups if groups is not None else [(group, material, list(range(nel)))]
        f.write("%10d%10d%10d%10d%10d%10d\nENDOFSECTION\n" % (coords.shape[0], nel, len(glist), len(sets), dim, dim))
        f.write("   NODAL COORDINATES 2.3.16\n")
        for j, x in enumerate(coords):
            f.write("%10d" % (j + 1) + "".join("%20.11e" % v for v in x) + "\n")
        f.write("ENDOFSECTION\n      ELEMENTS/CELLS 2.3.16\n")
        for e in range(nel):
            ids = [int(elem_dof[e, order[i]]) + 1 for i in range(nl)]
            f.write("%8d %2d %2d " % (e + 1, 4 if geom == "hex" else 2, nl))
            for k in range(0, nl, 7):
                f.write(("" if k == 0 else "               ") + "".join("%8d" % v for v in ids[k:k + 7]) + "\n")
        f.write("ENDOFSECTION\n")
        for gi, (gname, gmat, gel) in enumerate(glist):
            f.write("       ELEMENT GROUP 2.3.16\n")
            f.write("GROUP: %10d ELEMENTS: %10d MATERIAL: %10d NFLAGS: %10d\n%32d\n       0\n" % (gi + 1, len(gel), gmat, 1, gname))
            for k in range(0, len(gel), 10):
                f.write("".join("%8d" % (v + 1) for v in gel[k:k + 10]) + "\n")
            f.write("ENDOFSECTION\n")
        for s in sets:
            faces = [(e, fc) for e in range(nel) for fc in range(face_flag.shape[1]) if face_flag[e, fc] == -s - 1]
            f.write(" BOUNDARY CONDITIONS 2.3.16\n%32d%8d%8d%8d%8d\n" % (s, 1, len(faces), 0, 6))
            for e, fc in faces:
                gf = HEX_FACE_FEMUS_TO_GAMBIT[fc] if geom == "hex" else fc + 1
                f.write("%10d%5d%5d\n" % (e + 1, 4 if geom == "hex" else 2, gf))
            f.write("ENDOFSECTION\n")
