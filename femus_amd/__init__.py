"""femus_amd -- MI355X (gfx950) backend for the FEMuS assembly + geometric-multigrid hot path.

The product is the C-ABI library femus_amd/lib/libfemus_hip.so (sources in femus_amd/csrc, interface in
include/femus_hip.h) plus the C++ adapters that mirror FEMuS's SparseMatrix / NumericVector /
LinearEquationSolver classes (femus_amd/csrc/adapters).  This Python package is only the ctypes binding used
by tests/ and bench.py; it never falls back to a CPU implementation: importing works everywhere, but every
compute call needs the built library and a HIP device.
"""
from ._lib import load_library, library_path, LibraryMissing  # noqa: F401
from .capi import Context, Vec, Mat, Mesh, Assembler, Multigrid, Halo, FemusHipError  # noqa: F401
