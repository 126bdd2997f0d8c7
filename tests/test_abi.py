"""CPU: the C-ABI shared library builds, loads and exports every symbol declared in include/*.h (no compute)."""
import ctypes
import os
import re

import pytest

from mikudance_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    syms = []
    for fn in os.listdir(os.path.join(ROOT, "include")):
        if fn.endswith(".h"):
            text = open(os.path.join(ROOT, "include", fn)).read()
            text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
            syms += re.findall(r"\b(md_[a-z0-9_]+)\s*\(", text)
    return sorted(set(syms))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/ but not exported"
    assert set(_lib.SIGNATURES) == set(syms), set(_lib.SIGNATURES) ^ set(syms)
    typed = _lib.load()
    assert typed.md_version() >= 100


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libmdance_hip.so")
    with pytest.raises(_lib.MdanceHipError):
        _lib.load()


def test_ops_refuse_cpu_tensors():
    import torch
    from mikudance_amd import ops
    with pytest.raises(_lib.MdanceHipError):
        ops.gemm(torch.zeros(8, 64, dtype=torch.float16), torch.zeros(8, 64, dtype=torch.float16))
