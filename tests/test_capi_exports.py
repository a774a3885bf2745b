"""The C-ABI library loads and exports every symbol include/femus_hip.h declares (no compute, CPU box)."""
import ctypes
import os
import re

import femus_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "femus_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fh_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    femus_amd.load_library()      # orders the HIP runtimes (torch first) before the raw handle below
    L = ctypes.CDLL(femus_amd.library_path())
    names = declared_symbols()
    assert len(names) > 60
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, "declared in include/femus_hip.h but not exported: %s" % missing


def test_no_cpu_fallback_without_device():
    """on a box without a GPU fh_init must fail with a message, not fall back"""
    import torch
    if torch.cuda.is_available():
        return
    L = femus_amd.load_library()
    h = ctypes.c_void_p()
    rc = L.fh_init(0, ctypes.byref(h))
    assert rc != 0
    assert b"no CPU fallback" in L.fh_last_error()
