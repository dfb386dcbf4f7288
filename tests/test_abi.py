"""The C-ABI shared library: loads, exports every symbol include/mgm_hip.h declares,
and refuses to compute without a device (there is no CPU fallback)."""
import ctypes
import os
import re

import pytest

import mgm_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "mgm_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mgm_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    assert header_functions() == sorted(mgm_amd.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(mgm_amd.LIB_PATH)
    for name in header_functions():
        assert hasattr(lib, name), name
    assert b"gfx950" in mgm_amd.load_library().mgm_version()


def test_no_cpu_fallback():
    try:
        ctx = mgm_amd.Context(0)
    except mgm_amd.MgmError as e:  # no usable gfx950 device: the library refuses, it does not fall back
        assert e.code == mgm_amd.MGM_ERR_HIP
        return
    ctx.close()
    pytest.skip("a device is present")


def test_product_does_not_touch_the_oracle():
    """Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may use oracle/."""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "mgm_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cc", ".cpp")):
                txt = open(os.path.join(base, f), errors="ignore").read()
                if re.search(r"\boracle\b", txt) and f != "build.py":
                    bad.append(os.path.join(base, f))
    assert not bad, bad
