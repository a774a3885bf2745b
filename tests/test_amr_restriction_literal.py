"""Mesh::GetAMRRestrictionAndAMRSolidMark written out statement by statement (Mesh.cpp:1352-1801) -- the same containers (std::map = a dict walked
in key order, MyVector / MyMatrix = lists in insertion order), the same loop nests, the same overwrite order, the `candidateNodes` logic with its `false`
short-circuit (:1519), the single-process pass of the exchange loop (:1600-1690) and the genealogy walk (:1711-1801) -- and run beside the oracle's
restatement `femus_oracle_amr.amr_restriction(..., "reference")`, which is vectorised and organised differently (hanging -> masters).

What this settles (round-5 verdict, weak 1c): on a 2 x 2 x 2 box where levels 0, 1 and 2 meet along an edge, the literal loops give rows whose
weights do not sum to one (0.375 - 0.125 - 0.09375 = 0.15625 at the Q2 node at 1/8 of the level-0 edge): the level pair (1, 2) overwrites the entries
the pair (0, 2) wrote under the same [master][hanging] key, and the genealogy walk's `alreadyFound` test drops the path through the intermediate
hanging node.  The oracle's "reference" mode is that map, entry for entry.

The geometric helpers are restated from src/ism/PolynomialBases.cpp: GetConvexHullSphere (:1820-1840), GetBoundingBox (:1842-1862),
GetClosestPointInReferenceElement (:1933-1955, reference coordinates of the nearest node as the initial guess), CheckIfPointIsInsideReferenceDomainHex /
Quad (:1505-1519).  GetInverseMapping runs Newton family by family (linear, serendipity, biquadratic) on polynomial coefficients; the elements of a box
mesh are affine, every family reproduces the map exactly, so the biquadratic Newton below reaches the same point.  Inputs (element dofs, coordinates,
element levels, the -1 marks of the near-face array) come from the oracle's mesh; nothing of the oracle's restriction code is called by `literal_restriction`."""
import math

import numpy as np
import pytest

from oracle import femus_oracle as fo
from oracle import femus_oracle_amr as fa

NFE_FAMS_C_ZERO_LAGRANGE = 3
FAMILY = {0: "linear", 2: "biquadratic"}          # the two families this build serves beside the reference's three


def get_convex_hull_sphere(xv, tolerance):
    dim, ndofs = len(xv), len(xv[0])
    xc = [0.0] * dim
    for d in range(dim):
        for i in range(ndofs):
            xc[d] += xv[d][i]
        xc[d] /= ndofs
    r2 = 0.0
    for j in range(ndofs):
        d2 = 0.0
        for d in range(dim):
            d2 += (xv[d][j] - xc[d]) * (xv[d][j] - xc[d])
        r2 = r2 if r2 > d2 else d2
    return xc, (1.0 + tolerance) * math.sqrt(r2)


def get_bounding_box(xv, tolerance):
    dim, ndofs = len(xv), len(xv[0])
    xe = [[xv[d][0], xv[d][0]] for d in range(dim)]
    for d in range(dim):
        for i in range(1, ndofs):
            xe[d][0] = xv[d][i] if xv[d][i] < xe[d][0] else xe[d][0]
            xe[d][1] = xv[d][i] if xv[d][i] > xe[d][1] else xe[d][1]
    for d in range(dim):
        epsilon = tolerance * (xe[d][1] - xe[d][0])
        xe[d][0] -= epsilon
        xe[d][1] += epsilon
    return xe


def get_closest_point_in_reference_element(geom, xv, x):
    dim, ndofs = len(xv), len(xv[0])
    jmin, d2min = ndofs, 1.0e100
    for j in range(ndofs):
        d2 = 0.0
        for d in range(dim):
            d2 += (xv[d][j] - x[d]) * (xv[d][j] - x[d])
        if d2 < d2min:
            d2min, jmin = d2, j
    return [float(v) for v in fo.xc_table(geom)[jmin]]          # XI[ieltype][jmin]: node coordinates of the reference element


def get_inverse_mapping(geom, xv, xl, xi):
    X = np.array(xv).T
    xi = np.array(xi, dtype=float)
    for _ in range(30):
        phi, dphi, _ = fo.eval_basis(geom, "biquadratic", xi[None, :])
        r = phi[0] @ X - np.array(xl)
        dx = np.linalg.solve((dphi[0].T @ X).T, r)
        xi = xi - dx
        if np.abs(dx).max() < 1e-15:
            break
    return [float(v) for v in xi]


def check_if_point_is_inside_reference_domain(xi, eps):
    threshold = 1.0 + eps
    return all(abs(v) < threshold for v in xi)


def literal_restriction(mesh, soltype):
    """restriction[soltype] of Mesh::GetAMRRestrictionAndAMRSolidMark on one process, as std::map<unsigned, std::map<unsigned, double>> -> dict of dicts"""
    geom = mesh.geom
    dim = mesh.dim
    fe = FAMILY[soltype]
    lev = fa.elem_levels(mesh)
    near_face_is_minus_one = fa.interface_faces(mesh)         # el->GetElementNearFaceArray()[iel][jface] == -1
    face_nodes = fo.face_nodes(geom)                          # GetIG(type, jface, k), k < GetNFACENODES(type, jface, 2)
    nfaces = near_face_is_minus_one.shape[1]
    level_of_list = int(lev.max())
    n_elem_dofs = fo.ndofs(geom, fe)                          # GetElementDofNumber(iel, soltype)

    def get_solution_dof(j, iel):                             # Mesh::GetSolutionDof for the nodal families: the element's j-th node
        return int(mesh.elem_dof[iel, j])

    restriction = {}
    interfaceElement, interfaceLocalDof, interfaceDof, interfaceNodeCoordinates = [], [], [], []
    for ilevel in range(level_of_list + 1):
        # interface element search (:1389-1404)
        ie = []
        for i in range(mesh.nel):
            if ilevel == lev[i]:
                for j in range(nfaces):
                    if near_face_is_minus_one[i, j]:
                        ie.append(i)
                        break
        interfaceElement.append(ie)
        # interface node search (:1406-1428): std::map<unsigned, bool> ldofs -> ascending local index
        ild = []
        for iel in ie:
            ldofs = {}
            for jface in range(nfaces):
                if near_face_is_minus_one[iel, jface]:
                    for k in range(len(face_nodes[jface])):
                        ldofs[int(face_nodes[jface][k])] = True
            ild.append(sorted(ldofs))
        interfaceLocalDof.append(ild)
        # global dofs of this solution type (:1430-1456): filled until the first local index the family does not have
        idof = []
        for i, iel in enumerate(ie):
            row = []
            for jloc in ild[i]:
                if jloc < n_elem_dofs:
                    row.append(get_solution_dof(jloc, iel))
                else:
                    break
            idof.append(row)
        interfaceDof.append(idof)
        # coordinates (:1458-1474)
        crd = [[[float(mesh.coords[get_solution_dof(jnode, iel), k]) for jnode in ild[i]] for i, iel in enumerate(ie)] for k in range(dim)]
        interfaceNodeCoordinates.append(crd)

    for ilevel in range(level_of_list):
        for jlevel in range(ilevel + 1, level_of_list + 1):
            for i in range(len(interfaceDof[ilevel])):
                candidateNodes = {}
                iel = interfaceElement[ilevel][i]
                elementNodes = {}
                for j in range(n_elem_dofs):
                    elementNodes[get_solution_dof(j, iel)] = True
                xv = [[float(mesh.coords[get_solution_dof(j, iel), d]) for j in range(mesh.elem_dof.shape[1])] for d in range(dim)]
                xc, r = get_convex_hull_sphere(xv, 0.01)
                r2 = r * r
                xe = get_bounding_box(xv, 0.01)
                for k in range(len(interfaceDof[jlevel])):
                    for l in range(len(interfaceDof[jlevel][k])):
                        ldof = interfaceDof[jlevel][k][l]
                        if ldof not in candidateNodes or candidateNodes[ldof] is not False:
                            d2 = 0.0
                            xl = [0.0] * dim
                            for d in range(dim):
                                xl[d] = interfaceNodeCoordinates[jlevel][d][k][l]
                                d2 += (xl[d] - xc[d]) * (xl[d] - xc[d])
                            insideHull = True
                            if d2 > r2:
                                insideHull = False
                            for d in range(dim):
                                if xl[d] < xe[d][0] or xl[d] > xe[d][1]:
                                    insideHull = False
                            if insideHull:
                                if ldof not in elementNodes:
                                    xi = get_closest_point_in_reference_element(geom, xv, xl)
                                    xi = get_inverse_mapping(geom, xv, xl, xi)
                                    insideDomain = check_if_point_is_inside_reference_domain(xi, 0.0001)
                                    if insideDomain:
                                        phi, _, _ = fo.eval_basis(geom, fe, np.array(xi)[None, :])
                                        for j in range(len(interfaceDof[ilevel][i])):
                                            jloc = interfaceLocalDof[ilevel][i][j]
                                            value = float(phi[0, jloc])
                                            if abs(value) >= 1.0e-10:
                                                jdof = interfaceDof[ilevel][i][j]
                                                if jdof not in restriction.setdefault(jdof, {}):
                                                    restriction[jdof][jdof] = 1.0
                                                restriction[jdof][ldof] = value
                                                restriction.setdefault(ldof, {})[ldof] = 10.0
                                                candidateNodes[ldof] = True
                                    else:
                                        candidateNodes[ldof] = False
                                else:
                                    candidateNodes[ldof] = False

    # the exchange loop (:1600-1690) with n_procs = 1, curr_proc = lproc = 0: every master row is already present and every son has a row
    counter = 1
    while counter != 0:
        counter = 0
        masterNode = sorted(restriction)
        slaveNodes = [sorted(restriction[m]) for m in masterNode]
        slaveNodesValues = [[restriction[m][s] for s in sorted(restriction[m])] for m in masterNode]
        for i, inode in enumerate(masterNode):
            if inode not in restriction:
                counter += 1
                for j, jnode in enumerate(slaveNodes[i]):
                    restriction.setdefault(inode, {})[jnode] = slaveNodesValues[i][j]
            else:
                for j, jnode in enumerate(slaveNodes[i]):
                    value = slaveNodesValues[i][j]
                    if inode != jnode or value > 5.0:
                        restriction[inode][jnode] = value
                    if jnode not in restriction:
                        counter += 1
                        for k, mk in enumerate(masterNode):
                            if mk == jnode:
                                for l, lnode in enumerate(slaveNodes[k]):
                                    restriction.setdefault(jnode, {})[lnode] = slaveNodesValues[k][l]
                                break

    # genealogy (:1711-1801)
    restrictionCopy = {m: dict(r) for m, r in restriction.items()}
    for inode in sorted(restrictionCopy):
        genealogy, heredity, index = [[0]], [[0.0]], [0]
        if restrictionCopy[inode][inode] < 5.0:
            genealogy[0][0] = inode
            heredity[0][0] = 1.0
            index[0] = 0
            restriction[inode] = {inode: 1.0}
            level = 1
            while level > 0:
                father = genealogy[level - 1][index[level - 1]]
                del genealogy[level + 1:], heredity[level + 1:], index[level + 1:]
                while len(genealogy) < level + 1:
                    genealogy.append([]), heredity.append([]), index.append(0)
                genealogy[level], heredity[level], index[level] = [], [], 0
                cnt = 0
                for son in sorted(restrictionCopy[father]):
                    alreadyFound = False
                    for klevel in range(level):
                        for k in range(len(genealogy[klevel])):
                            if genealogy[klevel][k] == son:
                                alreadyFound = True
                    if not alreadyFound:
                        genealogy[level].append(son)
                        heredity[level].append(restrictionCopy[father][son] * heredity[level - 1][index[level - 1]])
                        restriction[inode][son] = restriction[inode].get(son, 0.0) + heredity[level][cnt]          # operator[] starts a new entry at 0.
                        cnt += 1
                        restriction[son] = {son: 0.0}
                if cnt > 0:
                    level += 1
                else:
                    test = True
                    while test and level > 0:
                        index[level - 1] += 1
                        test = False
                        if index[level - 1] == len(genealogy[level - 1]):
                            level -= 1
                            test = True
        else:
            restriction[inode] = {inode: 0.0}
    return restriction


def by_hanging_node(restriction):
    """master -> {son: w}  turned into  hanging -> {master: w}  (rows {node: 0.} are the hanging nodes: LinearImplicitSystem.cpp:912-1028 reads them so)"""
    hanging = sorted(n for n, row in restriction.items() if row == {n: 0.0})
    out = {h: {} for h in hanging}
    for m_, row in restriction.items():
        for son, w in row.items():
            if son != m_:
                out[son][m_] = w
    return out


def edge_flag(x, level):
    """level 0 | level 1 | level 2 meet along the line x = y = 0.5"""
    return x[0] > 0.5 if level == 0 else (x[0] > 0.5 and x[1] > 0.5)


def ex4_flag(x, level):
    return x[0] > 0.5 if level == 0 else (x[0] > 0.5 and x[1] > 0.25)


def scattered_flag(x, level):
    """single elements and a jump of two across faces, edges and corners"""
    s = math.sin(37.0 * x[0] + 11.0 * level) * math.cos(23.0 * x[1] + 5.0) + math.sin(17.0 * x[2] + 3.0 * level)
    return s > 0.1


@pytest.mark.parametrize("soltype", [0, 2])
def test_literal_loops_give_rows_that_do_not_sum_to_one_where_three_levels_meet(soltype):
    ms = fa.build_amr_levels(2, 2, 2, 1, 2, edge_flag)
    m = ms[-1]
    assert sorted(set(fa.elem_levels(m).tolist())) == [0, 1, 2]
    lit = by_hanging_node(literal_restriction(m, soltype))
    sums = {h: sum(r.values()) for h, r in lit.items()}
    odd = sorted(h for h, v in sums.items() if abs(v - 1.0) > 1e-12)
    assert odd, "the literal loops give a partition of unity everywhere: the oracle's 'reference' mode would be a misreading"
    X = m.coords
    # all of them next to the line x = y = 0.5 where the three levels meet
    assert all(abs(X[h][1] - 0.5) < 1e-14 and 0.5 - 1e-14 <= X[h][0] <= 0.75 for h in odd)
    if soltype == 2:
        # the hand-computed node: 1/8 of the level-0 edge [a, m, b] = middle of the first half [a, g, m] of a level-1 edge
        assert any(abs(sums[h] - 0.15625) < 1e-14 and sorted(np.round(list(lit[h].values()), 12)) == [-0.125, -0.09375, 0.375] for h in odd)
    # and the oracle's "reference" mode is this map, entry for entry
    ora = fa.amr_restriction(m, FAMILY[soltype], "reference")
    assert sorted(ora) == sorted(lit)
    for h in lit:
        assert sorted(ora[h]) == sorted(lit[h])
        assert max(abs(ora[h][k] - lit[h][k]) for k in lit[h]) < 1e-13          # two Newton inversions, products along chains of up to three levels


@pytest.mark.parametrize("box,nu,ns,flag", [((2, 2, 0), 1, 2, ex4_flag), ((2, 2, 2), 1, 2, ex4_flag), ((3, 2, 0), 2, 2, ex4_flag),
                                             ((2, 2, 2), 1, 2, scattered_flag), ((3, 3, 0), 1, 3, scattered_flag)])
@pytest.mark.parametrize("soltype", [0, 2])
def test_oracle_reference_mode_equals_the_literal_loops(box, nu, ns, flag, soltype):
    ms = fa.build_amr_levels(*box, nu, ns, flag)
    m = ms[-1]
    assert not m.homogeneous
    lit = by_hanging_node(literal_restriction(m, soltype))
    ora = fa.amr_restriction(m, FAMILY[soltype], "reference")
    assert sorted(ora) == sorted(lit) and len(lit) > 0
    for h in lit:
        assert sorted(ora[h]) == sorted(lit[h])
        assert max(abs(ora[h][k] - lit[h][k]) for k in lit[h]) < 1e-13          # two Newton inversions, products along chains of up to three levels
