"""The C-ABI library loads and exports every symbol include/silent_hip.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "silent_hip.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(silent_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_what_the_binding_binds():
    from pysilent_amd import _lib
    assert declared_symbols() == sorted(_lib.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol():
    from pysilent_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import subprocess
        import sys
        subprocess.check_call([sys.executable, os.path.join(ROOT, "pysilent_amd", "csrc", "build.py")])
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert lib.silent_abi_version() == _lib.ABI_VERSION


def test_struct_layouts_match_header():
    from pysilent_amd import _lib
    assert ctypes.sizeof(_lib.Extent) == 8
    assert ctypes.sizeof(_lib.PyrLevel) == 32
    assert ctypes.sizeof(_lib.RgbChainParams) == 5 * 8 + 5 * 4 + 4   # 5 pointers, 5 x 4-byte fields, tail pad


def test_no_gpu_means_loud_failure_not_fallback():
    """Without a GPU every filter must raise; with one this test is skipped."""
    import numpy as np
    from pysilent_amd import _runtime, filters
    if _runtime.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        filters.rgc_filter(np.zeros((1, 8, 8, 3), np.float32))


def test_type_error_matches_reference():
    from pysilent_amd import filters
    from pysilent_amd.util.apply_filter import apply_filter
    for fn in (filters.rgc_filter, filters.rgby_filter, filters.orientation_filter):
        with pytest.raises(TypeError, match="must either be tensor or numpy array"):
            fn([[1.0, 2.0]])
    with pytest.raises(TypeError):
        apply_filter("nope", None)


def test_product_never_imports_the_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pysilent_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                if re.search(r"^\s*(from|import)\s+(oracle|silent_oracle|c_oracle)\b", text, flags=re.M) or \
                        "libsilent_oracle" in text:
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad
