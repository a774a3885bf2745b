"""femus_amd -- MI355X (gfx950) backend for the FEMuS assembly + geometric-multigrid hot path.

The product is the C-ABI library femus_amd/lib/libfemus_hip.so (sources in femus_amd/csrc, interface in
include/femus_hip.h) plus the C++ adapters that mirror FEMuS's SparseMatrix / NumericVector /
LinearEquationSolver classes (femus_amd/csrc/adapters).  This Python package is only the ctypes binding used
by tests/ and bench.py; it never falls back to a CPU implementation: importing works everywhere, but every
compute call needs the built library and a HIP device.
"""
from ._lib import load_library, library_path, LibraryMissing  # noqa: F401
from .capi import Context, Vec, Mat, Mesh, Assembler, Multigrid, Halo, FemusHipError  # noqa: F401


def device_count():
    """HIP devices visible to this process (0 without a driver): launchers size themselves with it"""
    import ctypes
    n = ctypes.c_int(0)
    load_library().fh_device_count(ctypes.byref(n))
    return n.value


def loaded_runtimes():
    """which HIP runtime and which RCCL this process has mapped (paths from /proc/self/maps): the first multi-GPU run reports them"""
    out = {}
    try:
        for line in open("/proc/self/maps"):
            path = line.split()[-1]
            base = path.rsplit("/", 1)[-1]
            for key in ("libamdhip64", "librccl"):
                if base.startswith(key):
                    out.setdefault(key, set()).add(path)
    except OSError:
        pass
    res = {k: sorted(v) for k, v in out.items()}
    try:
        import ctypes
        v = ctypes.c_int(0)
        if load_library().fh_rccl_version(ctypes.byref(v)) == 0:
            res["rccl_version"] = v.value               # ncclGetVersion of the copy that is really bound
    except Exception:
        pass
    res["note"] = ("femus_amd._lib imports torch BEFORE libfemus_hip.so on purpose (one HIP runtime per process: torch's wheel bundles its own), so the "
                   "library's RCCL / HIP symbols resolve to the copies listed here, not to /opt/rocm's")
    return res
