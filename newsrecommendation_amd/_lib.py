"""ctypes binding of libnrhip.so (the C ABI declared in include/nrhip.h).

The product path has NO fallback: if the shared library is missing, or a call fails, a
RuntimeError is raised.  Nothing here imports `oracle/`.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading

NR_F32, NR_BF16 = 0, 1
NR_SRC_DENSE, NR_SRC_GATHER = 0, 1

_HERE = os.path.dirname(os.path.abspath(__file__))
# NRHIP_LIB: another build of the same library (same-box A/B measurements of kernel variants, tools/ab.sh); default in-tree
LIB_PATH = os.environ.get("NRHIP_LIB") or os.path.join(_HERE, "libnrhip.so")
CSRC_DIR = os.path.join(_HERE, "csrc")

c_f32p = C.POINTER(C.c_float)
c_i32p = C.POINTER(C.c_int32)
c_i64p = C.POINTER(C.c_int64)


class MhsaDesc(C.Structure):
    _fields_ = [("n", C.c_int), ("L", C.c_int), ("d_model", C.c_int), ("heads", C.c_int), ("d_head", C.c_int),
                ("dtype", C.c_int), ("src_kind", C.c_int), ("x", C.c_void_p), ("ldx", C.c_int), ("ids", C.c_void_p),
                ("p_in", C.c_float), ("seed_in", C.c_uint32), ("p_out", C.c_float), ("seed_out", C.c_uint32),
                ("mask", C.c_void_p), ("w_qkv", C.c_void_p), ("ldw", C.c_int), ("b_qkv", C.c_void_p),
                ("x_rows", C.c_void_p), ("ld_rows", C.c_int), ("row_ws", C.c_void_p), ("row_ws_bytes", C.c_size_t),
                ("table_rows", C.c_int), ("proj_table", C.c_void_p), ("seq_needed", C.c_void_p), ("seq_nz", C.c_void_p), ("row_ws_ready", C.c_int), ("bwd_phase", C.c_int),
                ("y_far_unwritten", C.c_int), ("dy_far_unwritten", C.c_int)]


class ConvDesc(C.Structure):
    _fields_ = [("n", C.c_int), ("T", C.c_int), ("D", C.c_int), ("Dp", C.c_int), ("N", C.c_int), ("dtype", C.c_int),
                ("table", C.c_void_p), ("ids", C.c_void_p), ("ids_stride", C.c_int), ("p_in", C.c_float),
                ("seed_in", C.c_uint32), ("w_pack", C.c_void_p), ("bias", C.c_void_p), ("x_rows", C.c_void_p),
                ("ld_rows", C.c_int), ("bwd_ws", C.c_void_p), ("bwd_ws_bytes", C.c_size_t), ("seq_nz", C.c_void_p),
                ("seq_needed", C.c_void_p)]


CAST_BATCH_MAX = 16          # NR_CAST_BATCH_MAX of include/nrhip.h


class CastJob(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("rows", C.c_int), ("cols", C.c_int), ("ld_src", C.c_int),
                ("ld_dst", C.c_int), ("transpose", C.c_int)]


class PackJob(C.Structure):
    _fields_ = [("first", C.c_size_t), ("count", C.c_size_t), ("cols", C.c_int), ("ld_dst", C.c_int), ("dst", C.c_void_p)]


ADAM_PACK_MAX = 4


class PoolDesc(C.Structure):
    _fields_ = [("n", C.c_int), ("L", C.c_int), ("N", C.c_int), ("q", C.c_int), ("dtype", C.c_int), ("x", C.c_void_p),
                ("mask", C.c_void_p), ("w1", C.c_void_p), ("ldw1", C.c_int), ("b1", C.c_void_p), ("w2", C.c_void_p),
                ("b2", C.c_void_p), ("partial_bytes", C.c_size_t), ("seq_needed", C.c_void_p), ("dx_far_unwritten", C.c_int)]


class LinearDesc(C.Structure):
    _fields_ = [("M", C.c_int), ("K", C.c_int), ("N", C.c_int), ("dtype", C.c_int), ("src_kind", C.c_int),
                ("x", C.c_void_p), ("ldx", C.c_int), ("ids", C.c_void_p), ("ids_stride", C.c_int), ("w", C.c_void_p),
                ("ldw", C.c_int), ("bias", C.c_void_p), ("w_t", C.c_void_p), ("ldwt", C.c_int), ("dout_ws_bytes", C.c_size_t), ("table_rows", C.c_int)]


_vp, _i, _f, _u32 = C.c_void_p, C.c_int, C.c_float, C.c_uint32
# name -> argtypes ; every entry returns int unless listed in RESTYPES.  Must list exactly the symbols of include/nrhip.h.
RESTYPES = {"nr_pool_seq_flags": C.c_void_p, "nr_eval_metrics_workspace_bytes": C.c_size_t, "nr_mhsa_workspace_bytes": C.c_size_t, "nr_conv_workspace_bytes": C.c_size_t, "nr_pool_workspace_bytes": C.c_size_t,
            "nr_linear_workspace_bytes": C.c_size_t}
SIGNATURES = {
    "nr_version": [],
    "nr_last_error": [C.c_char_p, C.c_size_t],
    "nr_abi_sizes": [C.POINTER(C.c_size_t), _i],
    "nr_set_deterministic": [_vp, C.c_size_t],
    "nr_set_option": [C.c_char_p, _i],
    "nr_get_option": [C.c_char_p],
    "nr_mhsa_workspace_bytes": [C.POINTER(MhsaDesc)],
    "nr_conv_workspace_bytes": [C.POINTER(ConvDesc)],
    "nr_pool_workspace_bytes": [C.POINTER(PoolDesc)],
    "nr_pool_contracts_slabs": [C.POINTER(PoolDesc)],
    "nr_pool_seq_flags": [C.POINTER(PoolDesc), _vp],
    "nr_linear_workspace_bytes": [C.POINTER(LinearDesc)],
    "nr_sdpa_fwd": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _u32, _vp],
    "nr_sdpa_bwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _u32, _vp],
    "nr_assemble_batch": [_vp, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp],
    "nr_stack_rows": [_vp, _i, _vp, _i, _i, _vp, _vp, _vp, _vp],
    "nr_eval_metrics_workspace_bytes": [_i],
    "nr_eval_metrics": [_vp, _vp, _vp, _i, _i, _vp, C.c_size_t, _vp, _vp],
    "nr_adam_step": [_vp, _vp, _vp, _vp, C.c_size_t, _f, _f, _f, _f, _i, _f, _i, _vp],
    "nr_adam_step_packed": [_vp, _vp, _vp, _vp, C.c_size_t, _f, _f, _f, _f, _i, _f, _i, _vp, _i, _vp],
    "nr_check_ids": [_vp, _i, _i, _i, _vp, _vp],
    "nr_check_labels": [_vp, _i, _i, _vp, _vp],
    "nr_cast_pad": [_vp, _i, _i, _i, _vp, _i, _i, _i, _vp],
    "nr_cast_pad_batch": [_vp, _i, _i, _vp],
    "nr_pack_conv_w": [_vp, _i, _i, _vp, _i, _i, _vp],
    "nr_unpack_conv_dw": [_vp, _i, _i, _i, _vp, _i, _vp],
    "nr_embed_gather_fwd": [_vp, _i, _i, _vp, _i, _i, _i, _vp, _i, _vp],
    "nr_embed_gather_bwd": [_vp, _i, _vp, _i, _i, _i, _vp, _i, _vp],
    "nr_gather_cast_fwd": [_vp, _i, _vp, _i, _i, _vp, _i, _i, _vp],
    "nr_mhsa_fwd_fused": [C.POINTER(MhsaDesc)],
    "nr_mhsa_compact_rows": [C.POINTER(MhsaDesc)],
    "nr_mhsa_fwd": [C.POINTER(MhsaDesc), _vp, _vp, _vp],
    "nr_mhsa_bwd": [C.POINTER(MhsaDesc), _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp],
    "nr_conv1d_k3_fwd": [C.POINTER(ConvDesc), _vp, _vp],
    "nr_conv1d_k3_bwd": [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp],
    "nr_additive_pool_fwd": [C.POINTER(PoolDesc), _vp, _vp, _vp, _i, _vp],
    "nr_additive_pool_bwd": [C.POINTER(PoolDesc), _vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "nr_pad_blend_fwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "nr_pad_blend_bwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "nr_linear_fwd": [C.POINTER(LinearDesc), _vp, _i, _vp],
    "nr_linear_bwd": [C.POINTER(LinearDesc), _vp, _i, _vp, _vp, _vp, _vp, _vp],
    "nr_score_ce_fwd": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "nr_score_ce_bwd": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _vp],
    "nr_score_eval": [_vp, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _vp],
    "nr_gemm_nt": [_i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp],
    "nr_gemm_tn": [_i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _vp],
    "nr_dropout_mask": [_vp, _u32, _f, _u32, _vp],
    "nr_prof_enable": [_i],
    "nr_prof_collect": [C.c_char_p, C.c_size_t],
    "nr_prof_filter": [C.c_char_p],
    "nr_debug_nt_trace": [_vp, _i],
}

_lib = None
_lock = threading.Lock()


def build(force: bool = False) -> str:
    """Compile libnrhip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    if force or not os.path.exists(LIB_PATH):
        subprocess.run(["make", "-C", CSRC_DIR, "-j4"], check=True)
    elif _stale():
        subprocess.run(["make", "-C", CSRC_DIR, "-j4"], check=True)
    return LIB_PATH


def _stale() -> bool:
    try:
        t = os.path.getmtime(LIB_PATH)
        srcs = [os.path.join(CSRC_DIR, f) for f in os.listdir(CSRC_DIR) if f.endswith((".hip", ".h"))]
        srcs.append(os.path.join(os.path.dirname(_HERE), "include", "nrhip.h"))
        return any(os.path.getmtime(s) > t for s in srcs)
    except OSError:
        return False


def lib():
    """The loaded library; raises RuntimeError if it is absent (no CPU fallback exists)."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError(
                        f"libnrhip.so not found at {LIB_PATH}: build it with "
                        "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C newsrecommendation_amd/csrc`. "
                        "newsrecommendation_amd has no CPU fallback.")
                L = C.CDLL(LIB_PATH)
                for name, argtypes in SIGNATURES.items():
                    fn = getattr(L, name)          # AttributeError if the .so lacks a declared symbol
                    fn.argtypes = argtypes
                    fn.restype = RESTYPES.get(name, C.c_int)
                structs = (MhsaDesc, ConvDesc, PoolDesc, LinearDesc, CastJob, PackJob)
                sizes = (C.c_size_t * len(structs))()
                if L.nr_abi_sizes(sizes, len(structs)) != 0 or list(sizes) != [C.sizeof(t) for t in structs]:
                    raise RuntimeError(f"libnrhip.so descriptor layout {list(sizes)} differs from the ctypes binding "
                                       f"{[C.sizeof(t) for t in structs]}: rebuild the library")
                _lib = L
    return _lib


def last_error() -> str:
    buf = C.create_string_buffer(512)
    lib().nr_last_error(buf, 512)
    return buf.value.decode(errors="replace")


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"libnrhip {what} failed (code {rc}): {last_error()}")


def set_option(name: str, value: int) -> None:
    """Library switch (csrc/nr_common.h NrOpt), e.g. set_option("NO_SLABS", 1)."""
    check(lib().nr_set_option(name.encode(), int(value)), "nr_set_option")


def get_option(name: str) -> int:
    return int(lib().nr_get_option(name.encode()))


def prof_enable(on, only=None) -> None:
    """False/0 off, True/1 every launch, 2 only launches over >= 65 536 rows, 3 only launches whose label starts with `only`."""
    if only is not None:
        check(lib().nr_prof_filter(str(only).encode()), "nr_prof_filter")
    check(lib().nr_prof_enable(int(on)), "nr_prof_enable")


def prof_collect() -> dict:
    """{label: (launches, total_ms)} of every kernel launched since the last collect (blocks on the events)."""
    buf = C.create_string_buffer(1 << 16)
    n = lib().nr_prof_collect(buf, len(buf))
    if n < 0:
        raise RuntimeError(f"libnrhip nr_prof_collect failed: {last_error()}")
    out = {}
    for line in buf.value.decode().splitlines():
        label, cnt, ms = line.rsplit("\t", 2)
        out[label] = (int(cnt), float(ms))
    return out


def ptr(t) -> int:
    """Device pointer of a torch tensor (None -> NULL)."""
    return 0 if t is None else t.data_ptr()
