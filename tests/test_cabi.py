"""The C-ABI library: loads without a GPU, exports every symbol include/evc.h declares, and
rejects bad arguments before touching the device.  No compute calls here (CPU container)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


def lib():
    from exemplars_vc_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        from exemplars_vc_amd.csrc.build import build
        build()
    return _lib, _lib.lib()


def test_exports_every_declared_symbol():
    _lib, L = lib()
    hdr = open(os.path.join(ROOT, "include", "evc.h")).read()
    declared = set(re.findall(r"\b(evc_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for sym in declared:
        assert hasattr(L, sym), sym
    assert L.evc_version() == 100


def test_struct_mirror_matches_header_fields():
    _lib, _ = lib()
    hdr = open(os.path.join(ROOT, "include", "evc.h")).read()
    body = hdr[hdr.index("typedef struct evc_solve_opts {"):hdr.index("} evc_solve_opts;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"\b(?:int|double|void\*|struct evc_solve_info\*|const struct evc_dict\*)\s+([a-z_0-9]+);", body)
    assert fields == [f[0] for f in _lib.SolveOpts._fields_]
    body = hdr[hdr.index("typedef struct evc_solve_info {"):hdr.index("} evc_solve_info;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    assert re.findall(r"\bint\s+([a-z_0-9]+);", body) == [f[0] for f in _lib.SolveInfo._fields_]
    assert C.sizeof(_lib.SolveInfo) == 32
    body = hdr[hdr.index("typedef struct evc_dict {"):hdr.index("} evc_dict;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = [n for grp in re.findall(r"\b(?:int|double|void\*|size_t)\s+([a-zA-Z_0-9, ]+);", body) for n in grp.replace(" ", "").split(",")]
    assert names == [f[0] for f in _lib.Dict._fields_]
    # kernel ids of the header <-> the names the Python side reports
    ids = dict((n.lower(), int(v)) for n, v in re.findall(r"EVC_KERNEL_([A-Z_0-9]+) = (\d+)", hdr))
    assert ids == {"none": 0, "gemm_nt": 1, "gemm2": 2, "fused_mu": 3, "fused_res": 4, "fused_all": 5, "fused_wide": 6,
                   "fused_wide64": 7, "fused_xy": 8}
    assert all(_lib.KERNEL_NAMES[v] in ("none", "k_" + n) for n, v in ids.items())


def test_strerror_and_workspace_queries():
    _lib, L = lib()
    assert _lib.strerror(0) == "ok"
    assert "argument" in _lib.strerror(-1) and "workspace" in _lib.strerror(-2)
    small = L.evc_workspace_bytes(25, 0, 512, 688, 1, _lib.F64, _lib.ALGO_FACTORED)
    big = L.evc_workspace_bytes(25, 0, 4096, 688, 1, _lib.F64, _lib.ALGO_FACTORED)
    gram = L.evc_workspace_bytes(25, 0, 4096, 688, 1, _lib.F64, _lib.ALGO_GRAM)
    assert 0 < small < big < gram
    assert L.evc_workspace_bytes(25, 25, 4096, 688, 1, _lib.F64, _lib.ALGO_FACTORED) >= big
    assert L.evc_workspace_bytes(25, 0, 4096, 688, 1, _lib.F32, _lib.ALGO_GRAM) < gram
    assert L.evc_workspace_bytes(-1, 0, 1, 1, 1, 0, 0) == 0 and L.evc_workspace_bytes(1, 0, 1, 1, 1, 7, 0) == 0


def test_bad_arguments_are_rejected_before_any_device_work():
    _lib, L = lib()
    o = _lib.SolveOpts()
    o.struct_bytes = 4                     # wrong size
    one = C.c_void_p(8)
    args = lambda opts, **kw: L.evc_nmf_solve(one, 25, one, 25, one, 64, kw.get("M", 25), 64, kw.get("T", 10),
                                              None, 1, C.byref(opts), one, 1 << 30, None, None, None)
    assert args(o) == -1
    o.struct_bytes = C.sizeof(_lib.SolveOpts)
    o.iters = -1
    assert args(o) == -1
    o.iters = 5; o.eps_mode = 9
    assert args(o) == -1
    o.eps_mode = 0; o.stop_rule = _lib.STOP_SKLEARN; o.check_every = 0
    assert args(o) == -1                   # a stopping rule needs residual evaluations
    o.stop_rule = 0
    assert args(o, M=0) == -1
    assert args(o, T=0) == 0               # nothing to do is not an error
    o.iters = 5; o.eps_mode = 0
    conv = lambda H, Mb: L.evc_nmf_convert(one, 25, one, 25, one, 25, H, 64, one, 25, 25, Mb, 64, 10, None, 1,
                                           C.byref(o), one, 1 << 30, None, None, None)
    assert conv(one, 0) == -1              # Mb < 1
    o.init_mode = _lib.INIT_GIVEN
    assert conv(None, 25) == -1            # H0 must be given when init is GIVEN
    assert L.evc_synthesize(one, 1, one, 64, one, 25, 25, 64, 10, 0, 0, None) == -1   # ldb < Mb
    assert L.evc_synthesize(one, 25, one, 64, one, 25, 25, 64, 0, 0, 0, None) == 0


def test_flag_constants_match_the_header():
    _lib, _ = lib()
    hdr = open(os.path.join(ROOT, "include", "evc.h")).read()
    for name in ("NO_FUSED", "EXACT_DIV", "NO_EXCHANGE", "NO_ALL_RESIDENT"):
        m = re.search(rf"EVC_FLAG_{name}\s*=\s*(\d+)", hdr)
        assert m and int(m.group(1)) == getattr(_lib, f"FLAG_{name}"), name
