"""The C-ABI library loads without a GPU and exports exactly what include/gcl.h declares."""
import os
import re

import torch

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "gcl.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gcl_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(lib_built):
    from graphcast_lite_amd import hip

    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib_built, n), f"{n} declared in include/gcl.h but not exported"
    assert sorted(hip.exported_symbols()) == names, "ctypes signature table and header disagree"
    assert lib_built.gcl_version() == 100


def test_missing_library_is_loud(monkeypatch, lib_built):
    from graphcast_lite_amd import hip

    monkeypatch.setattr(hip, "_lib", None)
    monkeypatch.setattr(hip, "LIB_PATH", "/nonexistent/libgcl_hip.so")
    try:
        hip.lib()
        raise AssertionError("expected a RuntimeError")
    except RuntimeError as e:
        assert "no CPU fallback" in str(e)


def test_cpu_tensors_are_rejected(lib_built):
    from graphcast_lite_amd import hip

    x = torch.zeros(4, 8)
    try:
        hip.linear_fwd(x, torch.zeros(8, 8), None, None)
        raise AssertionError("expected a RuntimeError")
    except RuntimeError as e:
        assert "GPU" in str(e)


def test_argument_validation_without_gpu(lib_built):
    """Validation paths that return before any HIP call."""
    import ctypes as C

    L = lib_built
    bad = torch.tensor([[0, 5], [1, 1]], dtype=torch.int64)
    e_out = C.c_int64(0)
    assert L.gcl_graph_count_edges(bad.data_ptr(), 2, 3, 0, C.byref(e_out)) == -1
    assert b"out of range" in L.gcl_last_error()
    assert L.gcl_graph_count_edges(bad.data_ptr(), 2, 8, 7, C.byref(e_out)) == -1
    assert L.gcl_graphnorm_ws_bytes(1, 1, 1) > 0 and L.gcl_linear_bwd_all_ws_bytes(1000, 64, 64) > 0


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "graphcast-lite_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("# oracle", ""), f"{fn} mentions the oracle"
