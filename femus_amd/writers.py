"""Output and restart files, host-side plumbing around the C-ABI mesh arrays:

    write_vtu()       <- VTKWriter::Write(output_path, "biquadratic", vars) (src/07_mesh_or_solution/01_multiple_levels/01_output/
                         VTKWriter.cpp:36-120, 460-770): one UnstructuredGrid piece, biquadratic cells (VTK types 28 / 29,
                         femusToVtkCellType :29), Float32 points and point data, Int32 connectivity/offsets, UInt16 types, every
                         DataArray "binary": base64(uint32 byte count) followed by base64(data), as print_data_array emits them.
                         Linear variables are carried to the biquadratic nodes by the element interpolation.
    save_solution()   <- MultiLevelSolution::SaveSolution (MultiLevelSolution.cpp:1070-1088): one file per variable,
    load_solution()      "<name>_iteration<k>_sol<Var>_level<n>", each a PETSc binary Vec as NumericVector::BinaryPrint writes it
                         (big-endian int32 class id 1211214, int32 length, float64 values; PETSc is not part of the reference tree,
                         the layout is its documented VecView binary format)
"""
import base64
import os
import struct

import numpy as np

VEC_FILE_CLASSID = 1211214
XC = {"quad": [(-1, -1), (1, -1), (1, 1), (-1, 1), (0, -1), (1, 0), (0, 1), (-1, 0), (0, 0)]}


def _b64(arr):
    raw = np.ascontiguousarray(arr).tobytes()
    return (base64.b64encode(struct.pack("<I", len(raw))) + base64.b64encode(raw)).decode()


def vtk_connectivity_order(geom, xc):
    """position in the VTK cell -> FEMuS local node.  VTK's 27-node hexahedron lists the face centres as x-, x+, y-, y+, z-, z+
    (FEMuS: y-, x+, y+, x-, z-, z+); everything else coincides (Writer_one_level::FemusToVTKorToXDMFConn)."""
    if geom == "quad":
        return list(range(9))
    faces = [(-1, 0, 0), (1, 0, 0), (0, -1, 0), (0, 1, 0), (0, 0, -1), (0, 0, 1)]
    return list(range(20)) + [xc.index(f) for f in faces] + [26]


def linear_to_biquadratic(mesh_arrays, geom, xc, values):
    """nodal values of a Q1 field at all biquadratic nodes: mean of the vertices the node sits between"""
    ed = mesh_arrays[0]
    nv = 4 if geom == "quad" else 8
    out = np.zeros(int(ed.max()) + 1)
    for i, c in enumerate(xc):
        verts = [v for v in range(nv) if all(c[d] == 0 or c[d] == xc[v][d] for d in range(len(c)))]
        out[ed[:, i]] = values[ed[:, verts]].mean(axis=1)
    return out


def write_vtu(path, mesh, fields, xc=None):
    """fields: name -> nodal array (length nnode: biquadratic; length own_size[0]: linear)"""
    ed, xy, _ = mesh.arrays()
    geom = mesh.geom
    if xc is None:
        xc = XC["quad"] if geom == "quad" else HEX_XC
    order = vtk_connectivity_order(geom, xc)
    nl = ed.shape[1]
    pts = np.zeros((mesh.nnode, 3), np.float32)
    pts[:, :mesh.dim] = xy
    conn = ed[:, order].astype(np.int32)
    with open(path, "w") as f:
        f.write('<?xml version="1.0"?>\n<VTKFile type = "UnstructuredGrid" version="0.1" byte_order="LittleEndian">\n  <UnstructuredGrid>\n')
        f.write('    <Piece NumberOfPoints= "%d" NumberOfCells= "%d" >\n' % (mesh.nnode, mesh.nel))
        f.write('      <Points>\n        <DataArray type="Float32" NumberOfComponents="3" format="binary">\n%s\n        </DataArray>\n      </Points>\n' % _b64(pts))
        f.write('      <Cells>\n        <DataArray type="Int32" Name="connectivity" format="binary">\n%s\n        </DataArray>\n' % _b64(conn))
        f.write('        <DataArray type="Int32" Name="offsets" format="binary">\n%s\n        </DataArray>\n'
                % _b64((np.arange(1, mesh.nel + 1) * nl).astype(np.int32)))
        f.write('        <DataArray type="UInt16" Name="types" format="binary">\n%s\n        </DataArray>\n      </Cells>\n'
                % _b64(np.full(mesh.nel, 28 if geom == "quad" else 29, np.uint16)))
        f.write('      <PointData Scalars="scalars">\n')
        for name, v in fields.items():
            v = np.asarray(v, float)
            if v.size != mesh.nnode:
                assert v.size == mesh.own_size[0], "field %s has neither the biquadratic nor the linear length" % name
                v = linear_to_biquadratic((ed,), geom, xc, v)
            f.write('        <DataArray type="Float32" Name="%s" format="binary">\n%s\n        </DataArray>\n' % (name, _b64(v.astype(np.float32))))
        f.write('      </PointData>\n    </Piece>\n  </UnstructuredGrid>\n</VTKFile>\n')


HEX_XC = [(-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1), (-1, -1, 1), (1, -1, 1), (1, 1, 1), (-1, 1, 1),
          (0, -1, -1), (1, 0, -1), (0, 1, -1), (-1, 0, -1), (0, -1, 1), (1, 0, 1), (0, 1, 1), (-1, 0, 1),
          (-1, -1, 0), (1, -1, 0), (1, 1, 0), (-1, 1, 0), (0, -1, 0), (1, 0, 0), (0, 1, 0), (-1, 0, 0), (0, 0, -1), (0, 0, 1), (0, 0, 0)]


def save_solution(directory, name, iteration, fields, level):
    """fields: variable name -> array.  Returns the file names written."""
    os.makedirs(directory, exist_ok=True)
    out = []
    for var, v in fields.items():
        fn = os.path.join(directory, "%s_iteration%d_sol%s_level%d" % (name, iteration, var, level))
        v = np.asarray(v, float)
        with open(fn, "wb") as f:
            f.write(struct.pack(">ii", VEC_FILE_CLASSID, v.size))
            f.write(v.astype(">f8").tobytes())
        out.append(fn)
    return out


def load_solution(prefix, variables, level):
    """prefix as passed to MultiLevelSolution::LoadSolution ("<dir>/<name>_iteration<k>"); returns name -> array"""
    out = {}
    for var in variables:
        fn = "%s_sol%s_level%d" % (prefix, var, level)
        if not os.path.exists(fn):
            raise FileNotFoundError("Error: cannot locate file " + fn)
        raw = open(fn, "rb").read()
        cid, n = struct.unpack(">ii", raw[:8])
        if cid != VEC_FILE_CLASSID:
            raise ValueError("%s is not a binary vector file" % fn)
        out[var] = np.frombuffer(raw, ">f8", n, 8).astype(float)
    return out
