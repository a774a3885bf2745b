"""applications/001_Poisson/main.cpp over the C-ABI: reads the application's own JSON input, builds the
box mesh, the boundary conditions and the source from their strings, and runs LinearImplicitSystem::MGsolve on the GPU.

    load_config()    <- InputParser::build / JsonInputParser (src/00_file_handling/runtime_input_parsing/file/JsonInputParser.cpp;
                        the vendored jsoncpp reader accepts // comments, so they are stripped here)
    Poisson001       <- main.cpp:46-282: mesh (:118-141), FE order (:149), parsed boundary conditions (:158-180, names of the box
                        faces from MeshGeneration.cpp:544-559 / 1040-1070), source string (:200-211), multigrid options (:216-257)
    run()            <- LinearImplicitSystem::MGsolve (LinearImplicitSystem.cpp:288-411): up to max_number_linear_iteration
                        cycles {MGSolve with the outer GMRES limited to 4 iterations (SetTolerances(1e-12,1e-20,1e50,4)),
                        HasLinearConverged: ||RES||_2 < abs_conv_tol}, then UpdateSol

Sign convention of the application (main.cpp:472-474): F = (src phi - grad phi . grad T) w, i.e. -Laplace T = src, which is the
library kernel with f = -src.  Mesh files ("filename" inputs, Gambit .neu) are read by fh_mesh_read_gambit
and take the boundary conditions of the application's SetBoundaryCondition function (main.cpp:26-36).
All numerics run in libfemus_hip.so; expressions are compiled by fh_expr_compile and evaluated on the device (source) or on the
host at boundary nodes (Dirichlet values).
"""
import json
import os
import re

import numpy as np

from . import capi
from .poisson import PoissonMG

FACE_NAMES = {1: ["left", "right"], 2: ["bottom", "right", "top", "left"], 3: ["bottom", "front", "right", "behind", "left", "top"]}   # MeshGeneration.cpp:248-253 (EDGE3 box)
FE_ORDER = {"first": "linear", "serendipity": "serendipity", "second": "biquadratic"}       # FEOrder of the input (main.cpp:149) -> Lagrange family
PREFIX = "multilevel_problem.multilevel_mesh.first.system.poisson.linear_solver."


def load_config(path_or_text):
    text = open(path_or_text).read() if "\n" not in path_or_text and "{" not in path_or_text else path_or_text
    out, i, in_str = [], 0, False
    while i < len(text):                        # outside strings: drop // comments, make "1." / "1.e-9" / ".5" strict JSON numbers
        c = text[i]
        if c == '"' and (i == 0 or text[i - 1] != "\\"):
            in_str = not in_str
        if not in_str and text.startswith("//", i):
            while i < len(text) and text[i] != "\n":
                i += 1
            continue
        if not in_str and c == "." and (i + 1 >= len(text) or not text[i + 1].isdigit()):
            out.append(".0")                    # jsoncpp reads "1." and "1.e-09"
            i += 1
            continue
        if not in_str and c == "." and (not out or not out[-1][-1].isdigit()):
            out.append("0.")                    # ".5"
            i += 1
            continue
        out.append(c)
        i += 1
    clean = re.sub(r",(\s*[}\]])", r"\1", "".join(out))
    return json.loads(clean)


def get(cfg, dotted, default):
    """InputParser::getValue: value at a dotted path, or the default"""
    node = cfg
    for key in dotted.split("."):
        if not isinstance(node, dict) or key not in node:
            return default
        node = node[key]
    return node


class Poisson001:
    def __init__(self, ctx, config, base_dir=None):
        """config: path of the JSON file, its text, or the parsed dict; base_dir: directory mesh file names are relative to
        (the application is run from its own directory)"""
        self.ctx = ctx
        cfg = config if isinstance(config, dict) else load_config(config)
        self.cfg = cfg
        mesh_type = get(cfg, "multilevel_mesh.first.type", {})
        self.mesh_file = None
        if "filename" in mesh_type:
            self.mesh_file = os.path.join(base_dir, mesh_type["filename"]) if base_dir else mesh_type["filename"]
            kind = self._gambit_kind(self.mesh_file)                       # cube_Tet.neu / cube_Wedge.neu of input3D_Tet_* / _Wedge_*.json: host-side mesh code
            self.tet, self.wedge, self.mixed = kind == "tet10", kind == "wedge18", kind == "mixed"
            if self.tet or self.wedge or self.mixed:
                with open(self.mesh_file) as f:
                    tok = f.read().split()
                self.dim = int(tok[tok.index("NDFVL") + 5])
            else:
                probe = capi.Mesh.read_gambit(self.mesh_file)
                self.dim = probe.dim
                probe.destroy()
            self.box = None
        elif "box" in mesh_type:
            b = mesh_type["box"]
            self.box = (int(b.get("nx", 2)), int(b.get("ny", 2)), int(b.get("nz", 0)))
            self.lo = (float(b.get("xa", 0.)), float(b.get("ya", 0.)), float(b.get("za", 0.)))
            self.hi = (float(b.get("xb", 1.)), float(b.get("yb", 1.)), float(b.get("zb", 0.)))
            self.dim = 1 if (self.box[1] == 0 and self.box[2] == 0) else 2 if self.box[2] == 0 else 3
            if self.dim == 2:
                self.hi = (self.hi[0], self.hi[1], 1.0)         # the box generator ignores z in 2-D
            self.tri = self.dim == 2 and b.get("elem_type", "Quad9") == "Tri6"            # MeshGeneration.cpp:568-: the box cut into triangles
            if self.dim == 1:
                assert b.get("elem_type", "Edge3") == "Edge3", "the one-dimensional box is made of EDGE3 elements (MeshGeneration.cpp:90)"
        else:
            raise ValueError("Error: no input mesh specified. Please check to have added the keyword mesh in the input json file! ")
        var = "multilevel_solution.multilevel_mesh.first.variable.first."
        self.fe = FE_ORDER[get(cfg, var + "fe_order", "first")]
        self.nlevels = int(get(cfg, PREFIX + "type.multigrid.nlevels", 1))
        self.npre = int(get(cfg, PREFIX + "type.multigrid.npresmoothing", 1))
        self.npost = int(get(cfg, PREFIX + "type.multigrid.npostmoothing", 1))        # the key the application reads (sic)
        self.max_linear = int(get(cfg, PREFIX + "max_number_linear_iteration", 6))
        self.abs_tol = float(get(cfg, PREFIX + "abs_conv_tol", 1.e-08))
        assert get(cfg, PREFIX + "type.multigrid.mgtype", "V_cycle") == "V_cycle", "only the V-cycle of the shipped inputs is served"
        # boundary conditions per face flag (flag = -(face name) - 1)
        self.bc_type, self.bc_func = {}, {}
        if self.box is not None:
            # default Dirichlet homogeneous on every face (InitializeBdc_with_ParsedFunction), then the listed faces
            names = FACE_NAMES[self.dim]
            for n in names:
                self.bc_type[self.flag_of(n)], self.bc_func[self.flag_of(n)] = "dirichlet", None
            for item in get(cfg, var + "boundary_conditions", []):
                name = item.get("facename", "top")
                if name not in names:
                    raise ValueError(" Error: the facename %s does not exist!" % name)
                self.bc_type[self.flag_of(name)] = item.get("bdc_type", "dirichlet")
                self.bc_func[self.flag_of(name)] = capi.Expr(item.get("bdc_func", "0."), "x,y,z,t")
        else:
            # SetBoundaryCondition of the application (main.cpp:26-36): Dirichlet 0 everywhere, flux 0.2 on face name 3
            self.file_flux = {-4: 0.2}
        self.source = capi.Expr(get(cfg, var + "func_source", "0."), "x,y,z,t")

    def flag_of(self, name):
        return -(FACE_NAMES[self.dim].index(name) + 2)

    def face_bc(self, flag):
        """(type, function or None) of a boundary face"""
        if self.box is None:
            return ("neumann", None) if flag in self.file_flux else ("dirichlet", None)
        return self.bc_type[flag], self.bc_func[flag]

    def dirichlet_data(self, mesh):
        """GenerateBdc with parsed functions (MultiLevelSolution.cpp:762-800): elements and faces in order; nodes of Dirichlet
        faces get Bdc = 0 and Sol = value(x, y, z, t = 0); a later face overwrites an earlier one"""
        ed, xy, ff = mesh.arrays()
        nc = {"linear": 2 ** self.dim, "serendipity": {1: 3, 2: 8, 3: 20}[self.dim], "biquadratic": 3 ** self.dim}[self.fe]
        val = {}
        for iel, f in zip(*np.nonzero(ff < -1)):
            kind, fn = self.face_bc(int(ff[iel, f]))
            if kind != "dirichlet":
                continue
            for i in capi.fe_face_nodes(mesh.geom, "biquadratic", f):
                if i >= nc:
                    continue
                node = int(ed[iel, i])
                x4 = np.zeros(4)
                x4[:self.dim] = xy[node]
                val[node] = fn(x4) if fn is not None else 0.0
        idx = np.array(sorted(val), dtype=np.int32)
        return idx, np.array([val[i] for i in idx])

    def run(self, smoother=capi.SMOOTH_GS_COLOR, omega=0.5, log=None, output_dir=None, simplex_smoother=capi.SMOOTH_GS_COLOR, simplex_omega=1.0):
        """smoother / omega: the application sets RICHARDSON + SOR_PRECOND on the fine grids (main.cpp:240-242) and leaves the
        Richardson scale at the solver default 0.5 (LinearEquationSolverPetsc.hpp:145).  output_dir: write what the application writes at
        its end (main.cpp:259-270): the VTK and the GMV file of "Sol", named as the reference names them"""
        ctx = self.ctx
        if self.dim == 1:
            return self.run_line(log)
        # triangles, tetrahedra, prisms, mixed shapes: multicolour Gauss-Seidel by default -- the iteration counts of the natural-order symmetric sweep the
        # application sets (SOR_PRECOND; simplex_smoother=capi.SMOOTH_SOR runs that one) at half the time on 240 k unknowns (tests/dev/probe_simplex_smoother.py)
        if getattr(self, "tri", False):
            return self.run_tri(log, simplex_smoother, simplex_omega)
        if getattr(self, "tet", False):
            return self.run_tet(log, simplex_smoother, simplex_omega)
        if getattr(self, "wedge", False):
            return self.run_wedge(log, simplex_smoother, simplex_omega)
        if getattr(self, "mixed", False):
            return self.run_mixed(log, simplex_smoother, simplex_omega)
        meshes = [capi.Mesh.box(*self.box, self.lo, self.hi) if self.box is not None else capi.Mesh.read_gambit(self.mesh_file)]
        for _ in range(1, self.nlevels):
            meshes.append(meshes[-1].refine())
        data = [self.dirichlet_data(m) for m in meshes]
        pb = PoissonMG(ctx, 0, 0, 0, self.nlevels, fe=self.fe, omega=omega, npre=self.npre, npost=self.npost, meshes=meshes,
                       smoother=smoother, dirichlet=[d[0] for d in data], source_expr=self.source, source_scale=-1.0)
        pb.init()
        top = self.nlevels - 1
        sol0 = np.zeros(pb.ndof[top])
        sol0[data[top][0]] = data[top][1]
        pb.SOL.upload(sol0)
        pb.assemble()
        # non-homogeneous Neumann faces: the parsed function of the face evaluated at every face Gauss point (box input, main.cpp:495-553);
        # the constant flux of SetBoundaryCondition (mesh-file input, main.cpp:556-594)
        flux = dict(self.file_flux) if self.box is None else {}
        for flag, kind in self.bc_type.items():
            if kind == "neumann" and self.bc_func[flag] is not None:
                flux[flag] = self.bc_func[flag]
        if flux:
            capi.assemble_neumann(ctx, meshes[top], self.fe, pb.RES, flux)
        pb.prepare()
        history = []
        for it in range(self.max_linear):
            its, _ = pb.mgsolve(outer="gmres", rtol=1e-12, atol=1e-20, maxit=4)
            rn = pb.RES.l2_norm()
            history.append((its, rn))
            if log:
                log("linear iteration %d: %d Krylov steps, Linear Res L2norm = %.6e" % (it + 1, its, rn))
            if rn < self.abs_tol:
                break
        pb.update_sol()
        _, xy, _ = meshes[top].arrays()
        result = {"solution": pb.SOL.to_numpy(), "coords": xy[:pb.ndof[top]], "history": history, "converged": history[-1][1] < self.abs_tol,
                  "dofs": pb.ndof[top]}
        if output_dir is not None:
            import os
            from . import writers
            # VTKWriter / GMVWriter file names: <prefix>.level<gridn>.<time step>.<order>.<ext> with gridn = number of levels
            stem = os.path.join(str(output_dir), "sol.level%d.%d.%s" % (self.nlevels, 0, "biquadratic"))
            field = result["solution"]
            if field.size != meshes[top].nnode:          # linear / serendipity solution: the writers carry the vertex values to the nodes of the output family
                field = field[:meshes[top].own_size[0]]
            writers.write_vtu(stem + ".vtu", meshes[top], {"Sol": field})
            writers.write_gmv(stem + ".gmv", meshes[top], {"Sol": field}, "biquadratic")
            result["files"] = [stem + ".vtu", stem + ".gmv"]
        pb.destroy()
        return result

    # ---- a two-dimensional box of triangles ("elem_type" : "Tri6") ---------------------------------------------------------------------------------
    def run_tri(self, log=None, smoother=capi.SMOOTH_GS_COLOR, omega=1.0):
        """a TRI6 box (TRI7 inside, femus_amd/tri_mesh.py): all three Lagrange families; boundary conditions and source as for the quadrilateral box"""
        from . import tri_mesh
        levels = [tri_mesh.box(self.box[0], self.box[1], self.lo[:2], self.hi[:2])]
        for _ in range(1, self.nlevels):
            levels.append(tri_mesh.refine(*levels[-1][:3]))
        fam = {"linear": 0, "serendipity": 1, "biquadratic": 2}[self.fe]
        return self._run_simplex("tri", levels, (3, 6, 7)[fam], [own[fam] for (_, _, _, own) in levels], log, smoother, omega)

    @staticmethod
    def _gambit_kind(path):
        """the elements of the file's ELEMENTS/CELLS section: Gambit type 6 with 10 nodes (TET10), type 5 with 18 (WEDGE18), more than one shape or TRI6
        ("mixed": cube_all_shapes*.neu, the two-dimensional files with triangles); None: the hexahedral / quadrilateral files the library's reader takes"""
        with open(path) as f:
            tok = f.read().split()
        if "ELEMENTS/CELLS" not in tok or "NDFVL" not in tok:
            return None
        nel, ngroup = int(tok[tok.index("NDFVL") + 2]), int(tok[tok.index("NDFVL") + 3])
        p = tok.index("ELEMENTS/CELLS") + 2
        seen = set()
        for _ in range(nel):
            seen.add((tok[p + 1], tok[p + 2]))
            p += 3 + int(tok[p + 2])
        if len(seen) > 1:
            return "mixed"
        kind = {("6", "10"): "tet10", ("5", "18"): "wedge18", ("3", "6"): "mixed"}.get(seen.pop())         # (TRI6 files go through the mixed-shape reader)
        return "mixed" if (kind is not None and ngroup > 1) else kind       # (and so do files with several element groups: it orders the elements by them)

    def run_mixed(self, log=None, smoother=capi.SMOOTH_GS_COLOR, omega=1.0):
        """a Gambit mesh of mixed shapes (input3D.json / input3D_All_first.json with input/cube_all_shapes_Six_boundary_groups.neu: tetrahedra, prisms and
        hexahedra; two-dimensional files of QUAD9 and TRI6 elements, or TRI6 alone; femus_amd/mixed_mesh.py): the three Lagrange families; the boundary
        conditions of the application's SetBoundaryCondition"""
        from . import mixed_mesh
        levels = [mixed_mesh.read_gambit(self.mesh_file)]
        for _ in range(1, self.nlevels):
            levels.append(mixed_mesh.refine(*levels[-1][:4]))
        fam = {"linear": 0, "serendipity": 1, "biquadratic": 2}[self.fe]
        return self._run_simplex("mixed", [l[1:] for l in levels], None, [l[4][fam] for l in levels], log, smoother, omega, kinds=[l[0] for l in levels])

    def run_wedge(self, log=None, smoother=capi.SMOOTH_GS_COLOR, omega=1.0):
        """a Gambit mesh of WEDGE18 elements (input3D_Wedge_first / _second / _serendipity.json with input/cube_Wedge.neu; femus_amd/wedge_mesh.py: WEDGE21
        inside): the three Lagrange families; the boundary conditions of the application's SetBoundaryCondition"""
        from . import wedge_mesh
        levels = [wedge_mesh.read_gambit(self.mesh_file)]
        for _ in range(1, self.nlevels):
            levels.append(wedge_mesh.refine(*levels[-1][:3]))
        fam = {"linear": 0, "serendipity": 1, "biquadratic": 2}[self.fe]
        return self._run_simplex("wedge", levels, (6, 15, 21)[fam], [own[fam] for (_, _, _, own) in levels], log, smoother, omega)

    def run_tet(self, log=None, smoother=capi.SMOOTH_GS_COLOR, omega=1.0):
        """a Gambit mesh of TET10 elements (input3D_Tet_first / _serendipity / _second.json with input/cube_Tet.neu; femus_amd/tet_mesh.py): P1, P2 and P2 with
        face and volume bubbles (TET15); the boundary conditions of the application's SetBoundaryCondition (Dirichlet 0, flux 0.2 on face name 3)"""
        from . import tet_mesh
        levels = [tet_mesh.read_gambit(self.mesh_file)]
        for _ in range(1, self.nlevels):
            levels.append(tet_mesh.refine(*levels[-1][:3]))
        fam = {"linear": 0, "serendipity": 1, "biquadratic": 2}[self.fe]
        return self._run_simplex("tet", levels, (4, 10, 15)[fam], [own[fam] for (_, _, _, own) in levels], log, smoother, omega)

    def _run_simplex(self, geom, levels, nc, ndofs, log, smoother, omega, kinds=None):
        """LinearImplicitSystem::MGsolve on meshes this module keeps (triangles, tetrahedra, prisms, mixed shapes): the Poisson callback through the generic kernel
        on the finest level (fh_assemble_poisson_rows / _mixed), transfers from the element prolongator, Galerkin operators below, V-cycles under GMRES limited to
        4 iterations per linear iteration.  kinds[l][e] (mixed meshes): the shape of every element of level l; elem_dof rows padded with -1"""
        ctx = self.ctx
        dim = levels[0][1].shape[1]
        fam = {"linear": 0, "serendipity": 1, "biquadratic": 2}[self.fe]
        NF = {"tri": 3, "tet": 4, "wedge": 5, "hex": 6, "quad": 4}
        CL = {"tri": (3, 6, 7), "tet": (4, 10, 15), "wedge": (6, 15, 21), "hex": (8, 20, 27), "quad": (4, 8, 9)}
        shapes = [geom] if kinds is None else sorted(set(kinds[0].tolist()))
        fn_by = {s: [capi.fe_face_nodes(s, self.fe, f) for f in range(NF[s])] for s in shapes}
        shape_of = (lambda l, e: geom) if kinds is None else (lambda l, e: kinds[l][e])
        groups = [[(geom, np.arange(lv[0].shape[0]))] if kinds is None else [(s, np.nonzero(kinds[l] == s)[0]) for s in shapes] for l, lv in enumerate(levels)]
        top = self.nlevels - 1
        ed, xs, ff, _ = levels[top]
        ndof = ndofs[top]
        K = self._pattern_from_elements([ed[idx][:, :CL[s][fam]] for s, idx in groups[top]], ndof)
        SOL, RES, EPS = ctx.vector(ndof), ctx.vector(ndof), ctx.vector(ndof)
        sol0 = np.zeros(ndof)
        bdc = []
        flux_faces, flux_idx, flux_exprs, tau_faces, tau_vals = [], [], [], [], []
        for l, (edl, xl, ffl, _) in enumerate(levels):
            val = {}
            for iel, f in zip(*np.nonzero(ffl < -1)):               # elements and faces in order; a later face overwrites an earlier one (GenerateBdc)
                flag = int(ffl[iel, f])
                kind, fn = self.face_bc(flag)
                nodes = edl[iel, fn_by[shape_of(l, iel)][f]]
                if kind == "dirichlet":
                    for node in nodes:
                        x4 = np.zeros(4)
                        x4[:dim] = xl[node]
                        val[int(node)] = fn(x4) if (fn is not None and l == top) else 0.0
                elif l == top:
                    if fn is not None:                              # parsed flux (box inputs)
                        if fn not in flux_exprs:
                            flux_exprs.append(fn)
                        flux_faces.append(nodes)
                        flux_idx.append(flux_exprs.index(fn))
                    elif self.box is None and flag in self.file_flux:      # the constant flux of SetBoundaryCondition (mesh-file inputs)
                        tau_faces.append(nodes)
                        tau_vals.append(self.file_flux[flag])
            idx = np.array(sorted(val), dtype=np.int32)
            bdc.append(idx)
            if l == top:
                sol0[idx] = [val[i] for i in idx]
        P = [None] + [self._prolongator_from_children([(s, idx, CL[s][fam]) for s, idx in groups[l - 1]], levels[l - 1][0], levels[l][0], ndofs[l - 1], ndofs[l])
                      for l in range(1, self.nlevels)]
        for l in range(1, self.nlevels):
            if bdc[l].size:
                P[l].mat_zero_rows(bdc[l], 0.0)
            if bdc[l - 1].size:
                P[l].zero_cols(bdc[l - 1])
        SOL.upload(sol0)
        mg = capi.Multigrid(ctx, self.nlevels)
        A = [None] * self.nlevels
        A[top] = K
        history = []
        its = 0
        for it in range(self.max_linear + 1):
            if kinds is None:
                capi.assemble_poisson_rows(ctx, geom, self.fe, ed, xs, K, RES, sol=SOL, source=self.source, scale=1.0)
            else:
                capi.assemble_poisson_mixed(ctx, self.fe, kinds[top], ed, xs, K, RES, sol=SOL, source=self.source, scale=1.0)
            if flux_faces:
                capi.assemble_neumann_edges(ctx, self.fe, np.array(flux_faces), np.array(flux_idx), flux_exprs, xs, RES)
            if tau_faces:                                             # by kind of face (a prism has quadrilaterals and triangles): the face element named
                for nn in sorted({len(f) for f in tau_faces}):
                    sel = [k for k, f in enumerate(tau_faces) if len(f) == nn]
                    fgeom = "lineface" if dim == 2 else "triface" if nn in (3, 6, 7) else "quadface"
                    capi.assemble_neumann_faces(ctx, fgeom, self.fe, np.array([tau_faces[k] for k in sel]), np.array([tau_vals[k] for k in sel]), xs, RES)
            if bdc[top].size:
                K.mat_zero_rows(bdc[top], 1.0)
                RES.set(bdc[top], np.zeros(bdc[top].size))
            rn = RES.l2_norm()
            history.append((its, rn))
            if log:
                log("linear iteration %d: Linear Res L2norm = %.6e" % (it, rn))
            if (it > 0 and rn < self.abs_tol) or it == self.max_linear:
                break
            for l in range(top, 0, -1):
                if A[l - 1] is None:
                    A[l - 1] = capi.Mat.ptap(P[l], A[l])
                else:
                    A[l - 1].ptap_numeric(P[l], A[l])
            for l in range(top):
                if bdc[l].size:
                    A[l].mat_zero_rows(bdc[l], 1.0)
            for l in range(self.nlevels):
                mg.set_level(l, A[l], P[l], None, smoother, omega, self.npre if l > 0 else 1, self.npost if l > 0 else 0)
            mg.setup()
            EPS.zero()
            its, _ = mg.solve(RES, EPS, outer="gmres" if self.nlevels > 1 else "preonly", rtol=1e-12, atol=1e-20, maxit=4)
            SOL.add(1.0, EPS)
        mg.destroy()
        result = {"solution": SOL.to_numpy(), "coords": xs[:ndof], "history": history, "converged": history[-1][1] < self.abs_tol, "dofs": ndof,
                  "levels": [(l[0], l[1], l[2]) for l in levels]}
        for m in A + P:
            if m is not None:
                m.destroy()
        return result

    def _pattern_from_elements(self, eds, ndof):
        """CSR pattern holding every (i, j) of every element; eds: one elem_dof array per shape.  Built on the device (fh_mat_create_from_elements) from one table
        padded to the widest shape with each element's first dof (a repeated dof adds no entry); the array version below serves rows the device builder's
        candidate lists do not hold"""
        width = max(ed.shape[1] for ed in eds)
        table = np.concatenate([np.concatenate([ed, np.broadcast_to(ed[:, :1], (ed.shape[0], width - ed.shape[1]))], axis=1) for ed in eds])
        try:
            return capi.Mat.from_elements(self.ctx, table, ndof)
        except capi.FemusHipError:
            pass
        keys = []
        for ed in eds:
            nc = ed.shape[1]
            r = np.repeat(ed, nc, axis=1).ravel().astype(np.int64)
            c = np.tile(ed, (1, nc)).ravel().astype(np.int64)
            keys.append(r * ndof + c)
        key = np.unique(np.concatenate(keys))
        rows, cols = key // ndof, key % ndof
        indptr = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=ndof))])
        return capi.Mat.from_csr(self.ctx, ndof, ndof, indptr, cols)

    def _prolongator_from_children(self, groups, ed_c, ed_f, ndof_c, ndof_f):
        """PP of a level from the element prolongators (ElemType.cpp:439-532): fine element nchild e + j is child j of coarse element e; the row of a fine dof
        holds the coarse shape functions at its place in the father (the same row from every element that shares the dof).  groups: (shape, coarse elements of
        that shape, dofs per element)"""
        rows_l, cols_l, vals_l = [], [], []
        for geom, idx, nc in groups:                                # the order of ElemType.cpp's insertions: child, fine node, coarse function, element
            EP = capi.fe_elem_prolongator(geom, self.fe)
            nch = EP.shape[0]
            for j in range(nch):
                for n in range(nc):
                    rows = ed_f[nch * idx + j, n]
                    for k in np.nonzero(EP[j, n, :nc])[0]:
                        rows_l.append(rows)
                        cols_l.append(ed_c[idx, k])
                        vals_l.append(np.full(rows.size, EP[j, n, k]))
        rows, cols, vals = np.concatenate(rows_l).astype(np.int64), np.concatenate(cols_l).astype(np.int64), np.concatenate(vals_l)
        key = (rows * ndof_c + cols)[::-1]                          # INSERT_VALUES: the last insertion of an entry stays
        uniq, last = np.unique(key, return_index=True)
        rows, cols, vals = uniq // ndof_c, uniq % ndof_c, vals[::-1][last]
        indptr = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=ndof_f))])
        return capi.Mat.from_csr(self.ctx, ndof_f, ndof_c, indptr, cols, vals)

    # ---- the one-dimensional input (input/input1D.json: EDGE3 box) -------------------------------------------------------------------------------
    NU_1D, V_1D = 0.01, 1.0                      # main.cpp:392-395: in one dimension the callback is advection-diffusion with V = 1, nu = 0.01

    def line_mesh(self):
        """MeshGeneration.cpp:78-262 (nodes i / (2 nx), element i = {2 i, 2 i + 2, 2 i + 1}, face 0 of the first element "left", face 1 of the last
        "right") and the numbering every FEMuS mesh gets: vertices first, then the middles, each in order of first appearance"""
        nx, (xa, xb) = self.box[0], (self.lo[0], self.hi[0])
        x = np.array([(i / (2.0 * nx)) * (xb - xa) + xa for i in range(2 * nx + 1)])
        ed = np.array([[2 * i, 2 * i + 2, 2 * i + 1] for i in range(nx)])
        new = np.full(2 * nx + 1, -1)
        k = 0
        for cls in ((0, 1), (2,)):
            for e in range(nx):
                for l in cls:
                    if new[ed[e, l]] < 0:
                        new[ed[e, l]] = k
                        k += 1
        xs = np.empty_like(x)
        xs[new] = x
        faces = {-2: (0, 0), -3: (nx - 1, 1)}                 # flag -> (element, local face = local node)
        return new[ed], xs, faces, nx + 1

    @staticmethod
    def refine_line(ed, xs, faces):
        """MeshRefinement::RefineMesh on EDGE3: element e -> children 2 e (at vertex 0) and 2 e + 1 (at vertex 1); vertex v of child j = coarse node
        fine2CoarseVertexMapping[j][v] ({0, 2}, {2, 1}); every child gets a new middle whose coordinate is the element prolongator's row (the quadratic map at
        xi = -1/2, +1/2); a child inherits the flag of the face its vertex lies on; then the numbering of every FEMuS mesh (vertices first, then middles, first touch)"""
        nel = ed.shape[0]
        f2c = ((0, 2), (2, 1))
        EP = capi.fe_elem_prolongator("line", "biquadratic")          # [child][local node][coarse function]
        raw = np.zeros((2 * nel, 3), dtype=np.int64)
        x = list(xs)
        for e in range(nel):
            for j in range(2):
                raw[2 * e + j, 0], raw[2 * e + j, 1] = ed[e, f2c[j][0]], ed[e, f2c[j][1]]
                raw[2 * e + j, 2] = len(x)
                x.append(sum(EP[j, 2, k] * xs[ed[e, k]] for k in range(3)))
        x = np.array(x)
        new = np.full(x.size, -1)
        k = 0
        for cls in ((0, 1), (2,)):
            for e in range(2 * nel):
                for l in cls:
                    if new[raw[e, l]] < 0:
                        new[raw[e, l]] = k
                        k += 1
        xf = np.empty_like(x)
        xf[new] = x
        ffaces = {flag: (2 * e + f, f) for flag, (e, f) in faces.items()}
        return new[raw], xf, ffaces, nel * 2 + 1

    def line_prolongator(self, ed_c, ed_f, ndof_c, ndof_f):
        """PP of a level (Mesh / FE prolongator, ElemType.cpp:439-532 on the line): row of fine dof = coarse shape functions at its reference point in the father"""
        nc = 2 if self.fe == "linear" else 3
        EP = capi.fe_elem_prolongator("line", self.fe)
        P = {}
        for e in range(ed_c.shape[0]):
            for j in range(2):
                for n in range(nc):
                    row = int(ed_f[2 * e + j, n])
                    for k in range(nc):
                        if EP[j, n, k] != 0.0:
                            P[(row, int(ed_c[e, k]))] = EP[j, n, k]
        keys = sorted(P)
        rows = np.array([q[0] for q in keys])
        indptr = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=ndof_f))])
        return capi.Mat.from_csr(self.ctx, ndof_f, ndof_c, indptr, np.array([q[1] for q in keys]), np.array([P[q] for q in keys]))

    def run_line(self, log=None, smoother=capi.SMOOTH_SOR, omega=1.0):
        """LinearImplicitSystem::MGsolve on the EDGE3 box with the callback's one-dimensional form (fh_assemble_advdiff_line) on the finest level, Galerkin
        operators below it (PP^T KK PP), V-cycles under GMRES limited to 4 iterations per linear iteration; one level (the shipped input): the exact solve.
        smoother / omega: Richardson + SOR_PRECOND as main.cpp:240-242 sets them (scale 1 here: the natural-order sweep of a one-dimensional operator)"""
        ctx = self.ctx
        levels = [self.line_mesh()]
        for _ in range(1, self.nlevels):
            levels.append(self.refine_line(*levels[-1][:3]))
        nc = 2 if self.fe == "linear" else 3
        ndofs = [(nv if self.fe == "linear" else xs.size) for (_, xs, _, nv) in levels]
        top = self.nlevels - 1
        ed, xs, faces, nv = levels[top]
        ndof = ndofs[top]
        pairs = sorted({(int(a), int(b)) for e in ed for a in e[:nc] for b in e[:nc]})
        rows = np.array([p[0] for p in pairs])
        indptr = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=ndof))])
        K = capi.Mat.from_csr(ctx, ndof, ndof, indptr, np.array([p[1] for p in pairs]))
        SOL, RES, EPS = ctx.vector(ndof), ctx.vector(ndof), ctx.vector(ndof)
        sol0 = np.zeros(ndof)
        bdc, point_flux = [[] for _ in levels], []
        for flag in faces:
            kind, fn = self.face_bc(flag)
            for l, (edl, xl, fl, _) in enumerate(levels):
                e, f = fl[flag]
                node = int(edl[e, f])
                if kind == "dirichlet":
                    bdc[l].append(node)
                if l == top:
                    x4 = np.array([xl[node], 0.0, 0.0, 0.0])
                    if kind == "dirichlet":
                        sol0[node] = fn(x4) if fn is not None else 0.0
                    elif fn is not None:                      # non-homogeneous Neumann: the side "element" is a point, F[node] += g(x) (main.cpp:540-549)
                        point_flux.append((node, fn(x4)))
        bdc = [np.array(sorted(b), dtype=np.int32) for b in bdc]
        P = [None] + [self.line_prolongator(levels[l - 1][0], levels[l][0], ndofs[l - 1], ndofs[l]) for l in range(1, self.nlevels)]
        for l in range(1, self.nlevels):                      # rows of fine Dirichlet dofs and columns of coarse ones carry nothing (ZeroInterpolatorDirichletNodes)
            if bdc[l].size:
                P[l].mat_zero_rows(bdc[l], 0.0)
            if bdc[l - 1].size:
                P[l].zero_cols(bdc[l - 1])
        SOL.upload(sol0)
        mg = capi.Multigrid(ctx, self.nlevels)
        A = [None] * self.nlevels
        A[top] = K
        history = []
        for it in range(self.max_linear + 1):
            capi.assemble_advdiff_line(ctx, self.fe, ed, xs, K, RES, self.NU_1D, self.V_1D, sol=SOL, source=self.source)
            if point_flux:
                r = RES.to_numpy()
                for node, g in point_flux:
                    r[node] += g
                RES.upload(r)
            if bdc[top].size:
                K.mat_zero_rows(bdc[top], 1.0)
                RES.set(bdc[top], np.zeros(bdc[top].size))
            rn = RES.l2_norm()
            history.append((0, rn) if it == 0 else (its, rn))
            if log:
                log("linear iteration %d: Linear Res L2norm = %.6e" % (it, rn))
            if (it > 0 and rn < self.abs_tol) or it == self.max_linear:
                break
            for l in range(top, 0, -1):                       # Galerkin chain, then the boundary rows of every level
                if A[l - 1] is None:
                    A[l - 1] = capi.Mat.ptap(P[l], A[l])
                else:
                    A[l - 1].ptap_numeric(P[l], A[l])
            for l in range(top):
                if bdc[l].size:
                    A[l].mat_zero_rows(bdc[l], 1.0)
            for l in range(self.nlevels):
                mg.set_level(l, A[l], P[l], None, smoother, omega, self.npre if l > 0 else 1, self.npost if l > 0 else 0)
            mg.setup()
            EPS.zero()
            its, _ = mg.solve(RES, EPS, outer="gmres" if self.nlevels > 1 else "preonly", rtol=1e-12, atol=1e-20, maxit=4)
            SOL.add(1.0, EPS)
        mg.destroy()
        result = {"solution": SOL.to_numpy(), "coords": xs[:ndof].reshape(-1, 1), "history": history, "converged": history[-1][1] < self.abs_tol, "dofs": ndof,
                  "elem_dof": ed, "nodes": xs, "levels": [(l[0], l[1]) for l in levels]}
        for m in A + P:
            if m is not None:
                m.destroy()
        return result

    def destroy(self):
        for e in list(self.bc_func.values()) + [self.source]:
            if e is not None:
                e.destroy()


if __name__ == "__main__":
    # python -m femus_amd.app_poisson input/input.json [output directory]   -- run from the application's directory, as the reference's executable is
    import sys
    import femus_amd
    if len(sys.argv) < 2:
        sys.exit("usage: python -m femus_amd.app_poisson <input.json> [output directory]")
    path = sys.argv[1]
    app = Poisson001(femus_amd.Context(0), path, base_dir=os.getcwd())
    out = app.run(log=print, output_dir=sys.argv[2] if len(sys.argv) > 2 else None) if app.dim > 1 else app.run(log=print)
    print("%d dofs, %d linear iteration(s), Linear Res L2norm %.3e, %s; max |Sol| = %.12g" % (out["dofs"], len(out["history"]), out["history"][-1][1],
                                                                                              "converged" if out["converged"] else "NOT converged",
                                                                                              float(np.abs(out["solution"]).max())))
    for f in out.get("files", []):
        print("wrote", f)
    app.destroy()
