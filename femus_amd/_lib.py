"""ctypes loader for libfemus_hip.so (no fallback: fails loudly when the library is missing)."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class LibraryMissing(RuntimeError):
    pass


def library_path():
    # FEMUS_HIP_LIBRARY: another build of the same library (A/B measurements of kernel variants)
    return os.environ.get("FEMUS_HIP_LIBRARY") or os.path.join(_HERE, "lib", "libfemus_hip.so")


def load_library():
    global _LIB
    if _LIB is None:
        p = library_path()
        if not os.path.exists(p):
            raise LibraryMissing(
                "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C femus_amd/csrc`). There is no CPU fallback." % p)
        # PyTorch-ROCm ships its own HIP runtime; it has to be the first (and only) one in the process.  Loading this
        # library before torch maps /opt/rocm's copy as well and the two tear each other down at exit (double free).
        # several processes on one node exchange device memory through dmabuf IPC handles on this driver stack; without this setting
        # RCCL's peer setup fails with "hipIpcGetMemHandle: invalid argument".  Before the HIP runtime starts; an explicit value wins.
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch  # noqa: F401
        _LIB = ctypes.CDLL(p, mode=ctypes.RTLD_GLOBAL)
        _declare(_LIB)
    return _LIB


def _declare(L):
    c_int, c_double, c_void_p, c_char_p = ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_char_p
    P = ctypes.POINTER
    L.fh_last_error.restype = c_char_p
    L.fh_version.restype = c_char_p
    L.fh_stream.restype = c_void_p
    L.fh_stream.argtypes = [c_void_p]
    L.fh_vec_dev_ptr.restype = c_void_p
    L.fh_vec_dev_ptr.argtypes = [c_void_p]
    L.fh_spmv_algorithmic_bytes.restype = ctypes.c_int64
    L.fh_spmv_algorithmic_bytes.argtypes = [c_void_p]
    if hasattr(L, "fh_mg_cycle_algorithmic_bytes"):
        L.fh_mg_cycle_algorithmic_bytes.restype = ctypes.c_int64
        L.fh_mg_cycle_algorithmic_bytes.argtypes = [c_void_p]
    # everything else returns int and takes pointers/ints/doubles; set argtypes for calls with doubles
    def sig(name, *args):
        if hasattr(L, name):
            f = getattr(L, name)
            f.restype = c_int
            f.argtypes = list(args)
    sig("fh_init", c_int, P(c_void_p))
    sig("fh_finalize", c_void_p)
    sig("fh_device_name", c_void_p, c_char_p, c_int)
    sig("fh_sync", c_void_p)
    sig("fh_timer_start", c_void_p)
    sig("fh_timer_stop", c_void_p, P(c_double))
    sig("fh_graph_begin", c_void_p)
    sig("fh_graph_end", c_void_p, P(c_void_p))
    sig("fh_graph_launch", c_void_p)
    sig("fh_graph_destroy", c_void_p)
    sig("fh_set_option", c_void_p, c_char_p, c_double)
    sig("fh_vec_create", c_void_p, c_int, c_int, c_int, c_void_p, c_int, P(c_void_p))
    sig("fh_vec_duplicate", c_void_p, P(c_void_p))
    sig("fh_vec_destroy", c_void_p)
    sig("fh_vec_size", c_void_p, P(c_int), P(c_int), P(c_int), P(c_int))
    sig("fh_vec_zero", c_void_p)
    sig("fh_vec_fill", c_void_p, c_double)
    sig("fh_vec_copy", c_void_p, c_void_p)
    sig("fh_vec_upload", c_void_p, c_void_p)
    sig("fh_vec_download", c_void_p, c_void_p)
    sig("fh_vec_set_values", c_void_p, c_int, c_void_p, c_void_p)
    sig("fh_vec_add_values", c_void_p, c_int, c_void_p, c_void_p)
    sig("fh_vec_get_values", c_void_p, c_int, c_void_p, c_void_p)
    sig("fh_vec_axpy", c_void_p, c_double, c_void_p)
    sig("fh_vec_aypx", c_void_p, c_double, c_void_p)
    sig("fh_vec_shift", c_void_p, c_double)
    sig("fh_vec_scale", c_void_p, c_double)
    sig("fh_vec_abs", c_void_p)
    sig("fh_vec_pointwise_mult", c_void_p, c_void_p, c_void_p)
    sig("fh_vec_dot", c_void_p, c_void_p, P(c_double))
    sig("fh_vec_norm", c_void_p, c_int, P(c_double))
    sig("fh_vec_reduce", c_void_p, c_int, P(c_double))
    sig("fh_mat_create_csr", c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, P(c_void_p))
    sig("fh_mat_destroy", c_void_p)
    sig("fh_mat_size", c_void_p, P(c_int), P(c_int), P(c_int))
    sig("fh_mat_zero", c_void_p)
    sig("fh_mat_set_values_csr", c_void_p, c_void_p)
    sig("fh_mat_get_values_csr", c_void_p, c_void_p)
    sig("fh_mat_get_pattern", c_void_p, c_void_p, c_void_p)
    sig("fh_mat_add_block", c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p)
    sig("fh_mat_insert_row", c_void_p, c_int, c_int, c_void_p, c_void_p)
    sig("fh_mat_get_row", c_void_p, c_int, P(c_int), c_void_p, c_void_p)
    sig("fh_mat_zero_rows", c_void_p, c_int, c_void_p, c_double)
    sig("fh_mat_zero_cols", c_void_p, c_int, c_void_p)
    sig("fh_mat_get_diagonal", c_void_p, c_void_p)
    sig("fh_mat_transpose", c_void_p, P(c_void_p))
    sig("fh_mat_ptap", c_void_p, c_void_p, P(c_void_p))
    sig("fh_mat_matmul", c_void_p, c_void_p, P(c_void_p))
    sig("fh_mat_norm", c_void_p, c_int, P(c_double))
    sig("fh_mat_abc", c_void_p, c_void_p, c_void_p, P(c_void_p))
    sig("fh_mat_col_mask", c_void_p, c_int, c_void_p, c_void_p)
    sig("fh_mat_row_mask", c_void_p, c_void_p, c_void_p)
    sig("fh_mat_restrict", c_void_p, c_int, c_void_p, c_void_p, c_int, P(c_void_p), P(c_void_p))
    sig("fh_mat_restrict_check", c_void_p, c_int, c_void_p, c_void_p, P(c_double))
    sig("fh_mat_value_map", c_void_p, c_void_p, c_void_p, c_void_p, P(c_void_p))
    sig("fh_halo_allreduce_mat", c_void_p, c_void_p)
    sig("fh_spmv", c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_double)
    sig("fh_spmv_transpose", c_void_p, c_void_p, c_void_p)
    sig("fh_fe_gauss", c_int, c_int, P(c_int), c_void_p, c_void_p)
    sig("fh_fe_tables", c_int, c_int, c_int, P(c_int), P(c_int), c_void_p, c_void_p)
    sig("fh_fe_tables_d2", c_int, c_int, c_int, c_void_p)
    sig("fh_fe_jacobian", c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p)
    sig("fh_fe_elem_prolongator", c_int, c_int, P(c_int), P(c_int), c_void_p)
    sig("fh_mesh_box", c_int, c_int, c_int, c_void_p, c_void_p, P(c_void_p))
    sig("fh_mesh_refine", c_void_p, P(c_void_p))
    sig("fh_mesh_destroy", c_void_p)
    sig("fh_mesh_clear_boundary_faces", c_void_p, ctypes.c_uint)
    sig("fh_mesh_set_coords", c_void_p, c_void_p)
    sig("fh_mg_set_level_distributed", c_void_p, c_int, c_void_p, c_int)
    sig("fh_halo_sizes", c_void_p, P(c_int), P(c_int))
    sig("fh_halo_allreduce_vec", c_void_p, c_void_p)
    sig("fh_mesh_info", c_void_p, P(c_int), P(c_int), P(c_int), P(c_int), c_void_p, P(c_int))
    sig("fh_mesh_get", c_void_p, c_void_p, c_void_p, c_void_p)
    sig("fh_mesh_child_elems", c_void_p, c_void_p)
    sig("fh_mesh_dirichlet_dofs", c_void_p, c_int, P(c_int), c_void_p)
    sig("fh_pattern_from_elements", c_int, c_int, c_void_p, c_int, c_void_p, c_void_p)
    sig("fh_build_prolongator", c_void_p, c_void_p, c_void_p, c_int, c_int, P(c_void_p))
    sig("fh_assembler_create", c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, P(c_void_p))
    sig("fh_assembler_destroy", c_void_p)
    sig("fh_assemble_poisson", c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p)
    sig("fh_assembler_info", c_void_p, P(c_int), P(ctypes.c_int64), P(c_double))
    sig("fh_element_matrices_poisson", c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p)
    sig("fh_fe_face_nodes", c_int, c_int, c_int, P(c_int), c_void_p)
    sig("fh_fe_node_ref", c_int, c_int, c_void_p)
    sig("fh_fe_node_ref_coords", c_int, c_int, c_void_p)
    sig("fh_assemble_neumann_faces", c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p)
    sig("fh_assemble_neumann_faces_expr", c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p)
    sig("fh_assemble_advdiff_line", c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_double, c_double, c_void_p, c_void_p, c_void_p)
    sig("fh_assemble_poisson_rows", c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_double, c_void_p, c_void_p)
    sig("fh_assemble_poisson_mixed", c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_double, c_void_p, c_void_p)
    sig("fh_assemble_pressure_faces", c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_double, c_void_p)
    sig("fh_fe_face_normals", c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p)
    sig("fh_mesh_read_gambit", c_char_p, c_double, P(c_void_p))
    sig("fh_mesh_refine_flagged", c_void_p, c_void_p, P(c_void_p))
    sig("fh_mesh_refine_device", c_void_p, c_void_p, c_void_p, P(c_void_p))
    sig("fh_mesh_elem_groups", c_void_p, c_void_p, c_void_p)
    sig("fh_mat_create_from_mesh", c_void_p, c_void_p, c_int, P(c_void_p))
    sig("fh_assembler_create_mesh", c_void_p, c_void_p, c_int, c_int, c_void_p, P(c_void_p))
    sig("fh_mesh_elem_centroids", c_void_p, c_void_p)
    sig("fh_mesh_elem_levels", c_void_p, c_void_p, P(c_int))
    sig("fh_mesh_amr_constraints", c_void_p, c_int, P(c_int), P(c_int), c_void_p, c_void_p, c_void_p, c_void_p)
    sig("fh_mesh_set_amr_mode", c_void_p, c_int)
    sig("fh_build_amr_prolongator", c_void_p, c_void_p, c_int, P(c_void_p))
    sig("fh_system_elem_dofs", c_void_p, c_int, c_void_p, P(c_int), c_void_p, c_void_p)
    sig("fh_build_system_prolongator", c_void_p, c_void_p, c_void_p, c_int, c_void_p, P(c_void_p))
    sig("fh_mesh_vertex_patches", c_void_p, c_int, c_void_p, P(c_int), P(c_int), c_void_p, c_void_p)
    sig("fh_ns_assembler_create", c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, P(c_void_p))
    sig("fh_ns_pw_assembler_create", c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, P(c_void_p))
    sig("fh_advdiff_assembler_create", c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, P(c_void_p))
    sig("fh_assemble_advection_diffusion", c_void_p, c_void_p, c_void_p, c_double, c_void_p, c_void_p)
    sig("fh_ns_assembler_destroy", c_void_p)
    sig("fh_assemble_navier_stokes", c_void_p, c_void_p, c_double, c_void_p, c_void_p)
    sig("fh_ns_element_matrices", c_void_p, c_void_p, c_double, c_void_p, c_void_p)
    sig("fh_mg_set_level_patches", c_void_p, c_int, c_int, c_void_p, c_void_p)
    sig("fh_mg_set_level_solver", c_void_p, c_int, c_int, c_int)
    sig("fh_expr_compile", c_char_p, c_char_p, P(c_void_p))
    sig("fh_expr_eval", c_void_p, c_void_p, P(c_double))
    sig("fh_expr_eval_many", c_void_p, c_int, c_void_p, c_void_p)
    sig("fh_expr_program", c_void_p, P(c_int), P(c_int), c_void_p, c_void_p)
    sig("fh_expr_nvars", c_void_p, P(c_int))
    sig("fh_expr_destroy", c_void_p)
    sig("fh_assemble_poisson_expr", c_void_p, c_void_p, c_void_p, c_double, c_void_p, c_void_p)
    sig("fh_halo_create_host", c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, P(c_void_p))
    sig("fh_index_create", c_void_p, c_int, c_void_p, P(c_void_p))
    sig("fh_index_destroy", c_void_p)
    sig("fh_mat_zero_rows_index", c_void_p, c_void_p, c_double)
    sig("fh_vec_set_index", c_void_p, c_void_p, c_double)
    sig("fh_mat_gather_values", c_void_p, c_void_p, c_void_p)
    sig("fh_vec_gather", c_void_p, c_void_p, c_void_p)
    sig("fh_mg_create", c_void_p, c_int, P(c_void_p))
    sig("fh_mg_set_level", c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_double, c_int, c_int)
    sig("fh_mg_set_coarse_coords", c_void_p, c_int, c_int, c_void_p)
    sig("fh_mg_coarse_info", c_void_p, P(c_int), P(c_int), P(c_int), P(c_int))
    sig("fh_mg_setup", c_void_p)
    sig("fh_mg_vcycle", c_void_p, c_void_p, c_void_p)
    sig("fh_mg_solve", c_void_p, c_void_p, c_void_p, c_int, c_double, c_double, c_double, c_int, c_int, P(c_int), P(c_double))
    sig("fh_mg_destroy", c_void_p)
    sig("fh_halo_unique_id", c_void_p)
    sig("fh_halo_create", c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, P(c_void_p))
    sig("fh_halo_create_shared", c_void_p, c_void_p, c_void_p, c_void_p, P(c_void_p))
    sig("fh_halo_update", c_void_p, c_void_p)
    sig("fh_halo_begin", c_void_p, c_void_p)
    sig("fh_halo_end", c_void_p)
    sig("fh_spmv_ghosted", c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_double)
    sig("fh_halo_stats", c_void_p, c_int, P(ctypes.c_int64), P(ctypes.c_int64), P(c_double), P(c_double))
    sig("fh_mat_split_info", c_void_p, c_int, P(c_int), P(c_int))
    sig("fh_halo_allreduce_sum", c_void_p, c_void_p, c_int)
    sig("fh_halo_destroy", c_void_p)
    sig("fh_dd_box_node_keys", c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p)
    sig("fh_dd_plan_create", c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, P(c_void_p))
    sig("fh_dd_plan_sizes", c_void_p, P(c_int), P(c_int), P(c_int))
    sig("fh_dd_plan_get", c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p)
    sig("fh_dd_plan_global", c_void_p, c_void_p, c_void_p)
    sig("fh_dd_plan_halo", c_void_p, c_void_p, c_void_p, c_void_p, P(c_void_p))
    sig("fh_dd_plan_destroy", c_void_p)
    sig("fh_dd_system_offsets", c_int, c_int, c_void_p, c_void_p, c_void_p)
    sig("fh_dd_system_dofs", c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p)
    sig("fh_write_vtu", c_void_p, c_char_p, c_int, c_void_p, c_void_p, c_void_p)
    sig("fh_write_gmv", c_void_p, c_char_p, c_int, c_int, c_void_p, c_void_p, c_void_p)
    sig("fh_xdmf_available")
    sig("fh_assembler_galerkin", c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p)
    sig("fh_spmv_expected_bytes", c_void_p, c_int, P(ctypes.c_int64), P(ctypes.c_int64))
    sig("fh_mesh_partition", c_void_p, c_int, c_void_p)
    sig("fh_mesh_partition_weighted", c_void_p, c_int, c_void_p, c_void_p)
    sig("fh_mesh_rank_elements", c_void_p, c_void_p, c_int, P(c_int), P(c_int), c_void_p)
    sig("fh_mesh_submesh", c_void_p, c_int, c_void_p, P(c_void_p), c_void_p)
    sig("fh_dd_topo_node_keys", c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p)
    sig("fh_write_xdmf", c_void_p, c_char_p, c_char_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p)
    sig("fh_vec_binary_print", c_void_p, c_char_p)
    sig("fh_vec_binary_load", c_void_p, c_char_p)
    sig("fh_host_binary_print", c_char_p, c_int, c_void_p)
    sig("fh_host_binary_load", c_char_p, P(c_int), c_void_p)
