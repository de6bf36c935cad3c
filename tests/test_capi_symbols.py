"""CPU: the C-ABI library loads and exports every symbol include/uavqp.h declares (no compute calls),
and the product path fails loudly -- never falls back to CPU -- when no GPU is usable."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "uavqp.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(uavqp_[a-z_]+)\s*\(", src)))


def test_header_and_binding_agree():
    from uav_motion_planning_amd import _lib
    assert header_symbols() == sorted(_lib.SYMBOLS)


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from uav_motion_planning_amd import _lib
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in header_symbols():
        assert hasattr(L, name), f"{name} declared in include/uavqp.h but not exported"
    assert b"gfx950" in _lib.lib().uavqp_version()


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the -m gpu tests")
    import uav_motion_planning_amd as U
    with pytest.raises(U.UavqpError, match="no HIP device|no CPU fallback"):
        U.Context(0)
    with pytest.raises(U.UavqpError):
        U.MinimumControl().solve([1.0, 2.0, 3.0, 4.0], [0.0, 0.0], [0.0, 0.0], [1.0, 1.0, 1.0])


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "uav_motion_planning_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.lower() or f == "_none_", f"{f} mentions the oracle"
