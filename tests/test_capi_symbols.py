"""CPU: the C-ABI library loads and exports every symbol include/xq_ops.h declares (no compute)."""
import ctypes
import os
import re

from conftest import ROOT


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "xq_ops.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(xq_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_entry_points():
    syms = declared_symbols()
    assert "xq_vq_forward" in syms and "xq_vq_backward" in syms and "xq_assign" in syms


def test_library_exports_every_declared_symbol():
    from imagefolder_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build libxq_ops.so first (__graft_entry__.build())"
    l = ctypes.CDLL(_lib.LIB_PATH)
    for s in declared_symbols():
        assert hasattr(l, s), f"{s} declared in include/xq_ops.h but not exported"


def test_python_binding_covers_header():
    from imagefolder_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    l = _lib.lib()
    assert l.xq_abi_version() == 1
    assert l.xq_assign_workspace_bytes(1024, 64, 4096) > 4096 * 64 * 4


def test_product_has_no_cpu_fallback():
    import pytest
    import torch
    from imagefolder_amd import ops
    from imagefolder_amd._lib import XqError
    with pytest.raises(XqError):
        ops.assign(torch.zeros(1, 8, 2, 2), torch.zeros(16, 8), ops.MODE_L2_NORMED)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "imagefolder_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(dp, fn)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, fn
