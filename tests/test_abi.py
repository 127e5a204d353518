"""The C-ABI library loads and exports every symbol include/d2r.h declares; without a GPU the
compute entry points fail loudly instead of falling back to a CPU path."""
import ctypes
import os
import re

import pytest

from dream2real_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    h = open(os.path.join(REPO, "include", "d2r.h")).read()
    return sorted(set(re.findall(r"D2R_API\s+[\w\s\*]+?\b(d2r_\w+)\s*\(", h)))


def test_header_symbols_are_exported():
    lib = _lib.load()
    syms = _declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in d2r.h but not exported"
    assert sorted(_lib.EXPORTS) == syms
    assert lib.d2r_abi_version() == _lib.ABI_VERSION == int(re.search(r"#define D2R_ABI_VERSION (\d+)", open(os.path.join(REPO, "include", "d2r.h")).read()).group(1))


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = _lib.load()
    h = ctypes.c_void_p()
    rc = lib.d2r_ctx_create(0, ctypes.byref(h))
    assert rc == -2 and not h.value                                   # D2R_ERR_DEVICE
    assert b"no HIP device" in lib.d2r_last_error(None) or b"gfx950" in lib.d2r_last_error(None)
    from dream2real_amd import engine
    with pytest.raises(_lib.D2RError):
        engine.Context(0)


def test_product_does_not_use_oracle():
    """The product never imports, links or dlopens anything under oracle/ (comments may cite it)."""
    import subprocess
    pkg = os.path.join(REPO, "dream2real_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")) or f == "Makefile":
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
                assert "libd2r_oracle" not in src and "oracle/_build" not in src, f
                assert not re.search(r'#include\s*[<"].*oracle', src), f
    deps = subprocess.run(["ldd", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in deps
