"""CPU-side checks of the drop-in boundary: the C-ABI shared library loads without a GPU and exports exactly
the entry points that include/ofq_hip.h declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ofq_hip.h")


def _declared():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ofq_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_entry_points():
    names = _declared()
    for must in ("ofq_statsq_fwd", "ofq_lsq_fwd", "ofq_lsq_bwd", "ofq_softmax_lsq_fwd", "ofq_softmax_lsq_bwd",
                 "ofq_gemm_f32", "ofq_qgemm_i8_nt", "ofq_qgemm_bf16s_nt", "ofq_qgemm_bf16s_tn", "ofq_colsum",
                 "ofq_cga_freeze_mask", "ofq_cga_mask_grad_save", "ofq_cga_restore"):
        assert must in names


def test_library_builds_loads_and_exports_every_declared_symbol():
    from ofq_amd import build, _lib
    so = build.build()
    assert os.path.exists(so)
    lib = ctypes.CDLL(so)
    for name in _declared():
        assert hasattr(lib, name), "include/ofq_hip.h declares %s but %s does not export it" % (name, so)
    handle = _lib.load()
    assert handle.ofq_abi_version() == 2


def test_python_binding_table_matches_header():
    from ofq_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()


def test_workspace_size_queries_are_pure_host_arithmetic():
    from ofq_amd import _lib
    lib = _lib.load()
    assert lib.ofq_lsq_bwd_ws_bytes(128, 198, 384, 384, 0) > 0
    assert lib.ofq_softmax_lsq_bwd_ws_bytes(1000) >= 4000
    assert lib.ofq_colsum_ws_bytes(25344, 384) == 64 * 384 * 4
    assert lib.ofq_qgemm_bf16s_tn_ws_bytes(384, 384, 8) == 8 * 384 * 385 * 4
    d = _lib.GemmDesc()
    d.M, d.N, d.K, d.nb0, d.nb1, d.split_k = 384, 384, 25344, 1, 1, 16
    assert lib.ofq_gemm_ws_bytes(ctypes.byref(d)) == 16 * 384 * 384 * 4
    # invalid geometry is rejected on the host, before any launch
    assert lib.ofq_lsq_bwd_ws_bytes(4, 7, 30, 30, 0) == 0          # inner % 4 != 0
