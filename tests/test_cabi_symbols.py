"""CPU: libnrhip.so loads and exports every symbol include/nrhip.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

from newsrecommendation_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "nrhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|size_t|const int32_t\*)\s+(nr_[a-z0-9_]+)\s*\(", src)))


def test_header_matches_binding_table():
    assert declared_symbols() == sorted(_lib.SIGNATURES)


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), f"libnrhip.so does not export {name}"
    assert _lib.lib().nr_version() >= 100


def test_no_cpu_fallback():
    """Ops refuse CPU tensors instead of silently computing elsewhere."""
    import torch
    from newsrecommendation_amd import ops
    x = torch.zeros(2, 3, 8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.pad_blend(x, None, None, ops.dtype_code("fp32"))


def test_descriptor_layout_and_workspace_sizes_are_host_arithmetic():
    """The binding checks sizeof() of every descriptor against the library at load; the workspace sizes come from the
    library (no hand-copied formulas in ops.py) and grow with the problem."""
    import ctypes as C
    lib = _lib.lib()                                        # raises on a layout mismatch
    d = _lib.MhsaDesc(n=28160, L=30)
    b = lib.nr_mhsa_workspace_bytes(C.byref(d))
    M = 28160 * 30
    assert b >= 4 * (3 * M + 3 * 28160 + M // 32) and b < 4 * (3 * M + 4 * 28160 + M // 32 + 64)
    d2 = _lib.MhsaDesc(n=28161, L=30)
    assert lib.nr_mhsa_workspace_bytes(C.byref(d2)) > b
    assert lib.nr_conv_workspace_bytes(C.byref(_lib.ConvDesc(n=28160, T=30))) >= 4 * (28160 + M // 32)
    p = _lib.PoolDesc(n=28160, L=30, q=200)
    assert lib.nr_pool_workspace_bytes(C.byref(p)) >= 4 * (28160 + M // 32)
    assert lib.nr_linear_workspace_bytes(C.byref(_lib.LinearDesc(M=100, N=400, dtype=_lib.NR_BF16))) == 100 * 400 * 2
    assert _lib.get_option("DMA_MIN_K") == 192 and _lib.get_option("NO_SUCH_OPTION") == -1
    _lib.set_option("NO_SLABS", 1)
    assert _lib.get_option("NO_SLABS") == 1
    _lib.set_option("NO_SLABS", 0)


def test_undersized_workspace_is_refused_before_any_launch():
    """The library validates caller-stated workspace sizes against its own formulas (host check, no GPU work)."""
    import ctypes as C
    lib = _lib.lib()
    d = _lib.MhsaDesc(n=64, L=30, d_model=304, heads=20, d_head=20, dtype=_lib.NR_BF16, src_kind=_lib.NR_SRC_GATHER, x=16, ldx=320,
                      ids=16, w_qkv=16, ldw=320, b_qkv=16, x_rows=16, ld_rows=304, row_ws=16, row_ws_bytes=64)
    rc = lib.nr_mhsa_fwd(C.byref(d), 16, 16, None)
    assert rc == 1 and "nr_mhsa_workspace_bytes" in _lib.last_error()
    p = _lib.PoolDesc(n=64, L=30, N=400, q=200, dtype=_lib.NR_BF16, x=16, w1=16, ldw1=400, b1=16, w2=16, b2=16, partial_bytes=8)
    rc = lib.nr_additive_pool_bwd(C.byref(p), 16, 16, 16, 400, 16, 200, 16, 16, 16, 16, 16, 16, None, None)
    assert rc == 1 and "nr_pool_workspace_bytes" in _lib.last_error()
