"""The C ABI: every symbol declared in include/cultionet_hip.h is exported by the built library,
and the ctypes binding (cultionet_amd/_lib.py) matches the header's parameter lists.
No compute is launched (runs without a GPU)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "cultionet_hip.h")


def _declarations():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(?:int|long|double)\s+(cn_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        name, params = m.group(1), m.group(2).strip()
        kinds = []
        if params and params != "void":
            for p in params.split(","):
                p = " ".join(p.split())
                if "*" in p:  # any pointer (incl. unsigned short*)
                    kinds.append("P")
                elif re.match(r"(const )?unsigned long long\b", p):
                    kinds.append("U64")
                elif re.match(r"(const )?long\b", p):
                    kinds.append("L")
                elif re.match(r"(const )?int\b", p):
                    kinds.append("I")
                elif re.match(r"(const )?float\b", p):
                    kinds.append("F")
                else:
                    raise AssertionError(f"unparsed parameter {p!r} in {name}")
        decls[name] = kinds
    return decls


def test_header_declares_the_binding():
    from cultionet_amd import _lib

    decls = _declarations()
    assert set(decls) == set(_lib.SIGNATURES), set(decls) ^ set(_lib.SIGNATURES)
    names = {"P": _lib.P, "L": _lib.L, "I": _lib.I, "F": _lib.F, "U64": _lib.U64}
    for name, kinds in decls.items():
        assert [names[k] for k in kinds] == _lib.SIGNATURES[name], name


def test_library_exports_every_symbol():
    from cultionet_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g

        g.build()
    lib = _lib.load()
    for name in _declarations():
        assert hasattr(lib, name), name
    assert _lib.query("cn_version") >= 100
    # pure host helpers
    assert _lib.query("cn_conv_kpad", 3) == 8 and _lib.query("cn_conv_kpad", 480) == 480
    assert _lib.query("cn_conv_npad", 1) == 32 and _lib.query("cn_conv_npad", 64) == 64
    assert _lib.query("cn_conv_npad", 128) == 128 and _lib.query("cn_conv_npad", 384) == 384


def test_product_path_has_no_cpu_fallback():
    import torch

    from cultionet_amd import engine

    with pytest.raises(RuntimeError):
        engine._check(torch.zeros(1))
