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
    for m in re.finditer(r"\b(?:int|long)\s+(cn_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
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


def test_plan_trampolines_match_the_header():
    """cn_plan_gen.inc (one trampoline per entry point, generated from the header) is what the generator would write now."""
    import subprocess
    import sys

    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_plan_trampolines.py"), "--check"])
    assert r.returncode == 0, "run `python tools/gen_plan_trampolines.py` and rebuild"


def test_native_plan_executor_on_host_entry_points():
    """cn_plan_run with pure-host entry points (no GPU): pointer / int / long slots arrive, execution stops at the first
    failing entry and reports its index, unknown names are refused."""
    import ctypes

    import numpy as np

    from cultionet_amd import _lib

    lib = _lib.load()
    n = ctypes.c_int(-1)
    i_begin = _lib.query("cn_plan_fn_index", b"cn_slice_sums_begin", ctypes.byref(n))
    assert i_begin >= 0 and n.value == 4
    i_end = _lib.query("cn_plan_fn_index", b"cn_slice_sums_end", None)
    i_kpad = _lib.query("cn_plan_fn_index", b"cn_conv_kpad", None)
    assert _lib.query("cn_plan_fn_index", b"cn_no_such_entry", None) == -1
    for name in _lib.SIGNATURES:  # every status-returning entry point has a trampoline with the binding's arity
        if name in ("cn_plan_run", "cn_plan_fn_index"):
            continue
        assert _lib.query("cn_plan_fn_index", name.encode(), ctypes.byref(n)) >= 0 and n.value == len(_lib.SIGNATURES[name]), name
    host = np.zeros(64 * 4, dtype=np.uint8)
    grad = np.zeros(16, dtype=np.float32)
    ops = np.zeros((3, 29), dtype=np.uint64)
    ops[0, 0] = i_begin << 32
    ops[0, 1:5] = [host.ctypes.data, 4, grad.ctypes.data, 16]
    ops[1, 0] = i_kpad << 32   # returns 8 for K = 3: a non-zero "status" -> the executor stops HERE
    ops[1, 1] = 3
    ops[2, 0] = i_end << 32
    failed = ctypes.c_int(-1)
    assert lib.cn_plan_run(ops.ctypes.data, 1, ctypes.byref(failed)) == 0
    assert _lib.query("cn_slice_sums_count") == 0  # the sink was registered with the pointers / sizes of the slots
    assert lib.cn_plan_run(ops.ctypes.data, 3, ctypes.byref(failed)) == 8 and failed.value == 1
    assert _lib.query("cn_slice_sums_count") == 0  # entry 2 (end) never ran
    _lib.call("cn_slice_sums_end")
    assert _lib.query("cn_slice_sums_count") == -1
    bad = np.zeros((1, 29), dtype=np.uint64)
    bad[0, 0] = 7  # unknown kind
    assert lib.cn_plan_run(bad.ctypes.data, 1, ctypes.byref(failed)) == -1 and failed.value == 0
