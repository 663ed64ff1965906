"""torch.autograd bindings of the libnrhip C ABI (include/nrhip.h).

PyTorch supplies device memory, the current HIP stream and autograd bookkeeping; every
arithmetic step on the hot path runs in the hand-written HIP kernels.  Tensors must live on
the GPU: there is no CPU fallback (a RuntimeError is raised instead).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch
from torch.autograd import Function

from . import _lib
from ._lib import NR_BF16, NR_F32, NR_SRC_DENSE, NR_SRC_GATHER, check, ptr

_DTYPES = {"fp32": NR_F32, "f32": NR_F32, "float32": NR_F32, "bf16": NR_BF16, "bfloat16": NR_BF16}


def dtype_code(name) -> int:
    if isinstance(name, int):
        return name
    try:
        return _DTYPES[str(name).lower()]
    except KeyError:
        raise ValueError(f"compute dtype must be one of {sorted(_DTYPES)}, got {name!r}")


def torch_dtype(code: int) -> torch.dtype:
    return torch.bfloat16 if code == NR_BF16 else torch.float32


def chunk(code: int) -> int:
    return 8 if code == NR_BF16 else 4


def round_up(v: int, m: int) -> int:
    return (v + m - 1) // m * m


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _need_gpu(*ts) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("newsrecommendation_amd ops need GPU tensors (libnrhip has no CPU fallback); "
                               f"got a tensor on {t.device}")


def draw_seed() -> int:
    """A 31-bit dropout seed from torch's CPU generator (reproducible under torch.manual_seed)."""
    return int(torch.randint(0, 2 ** 31 - 1, (1,)).item())


def pack(src: torch.Tensor, code: int, transpose: bool = False, ld: Optional[int] = None) -> torch.Tensor:
    """fp32 [rows, cols] -> compute-dtype GEMM operand, leading dimension zero-padded to a 16-byte chunk."""
    _need_gpu(src)
    src = src.detach()
    if src.dtype != torch.float32 or not src.is_contiguous():
        src = src.float().contiguous()
    rows, cols = src.shape
    drows, dcols = (cols, rows) if transpose else (rows, cols)
    # bf16: leading dimension rounded up to 32 and zero padded, so the LDS-DMA GEMM can run whole 32-deep k-steps
    ld = ld or round_up(dcols, 32 if code == NR_BF16 else chunk(code))
    dst = torch.empty(drows, ld, dtype=torch_dtype(code), device=src.device)
    check(_lib.lib().nr_cast_pad(ptr(src), rows, cols, cols, ptr(dst), ld, code, int(transpose), _stream()), "nr_cast_pad")
    return dst


class _TableCache:
    """Compute-dtype copies of embedding tables, refreshed when the fp32 master changes
    (tensor version counter), so a frozen table is packed once."""

    def __init__(self):
        self._c = {}

    def get(self, weight: torch.Tensor, code: int, row_cols: Optional[int] = None) -> torch.Tensor:
        w = weight.detach()
        cols = row_cols or w.shape[1]
        if code == NR_F32 and cols % 4 == 0 and w.is_contiguous() and w.dtype == torch.float32:
            return w.view(-1, cols)                       # already a valid operand: no copy
        key = (id(weight), code, cols)
        ent = self._c.get(key)
        sig = (weight._version, w.data_ptr(), tuple(w.shape))
        if ent is None or ent[0] != sig:
            ent = (sig, pack(w.reshape(-1, cols), code))
            self._c[key] = ent
        return ent[1]


table_cache = _TableCache()


# ------------------------------------------------------------------------------------------ MHSA
class MHSAFunction(Function):
    """K3 (+K1, K2): multi-head self-attention, src/model/model_utils.py:78-95 + :39-55, optionally with the
    embedding lookup and the dropouts of src/model/NRMS.py:28-34 folded in (gather source)."""

    @staticmethod
    def forward(ctx, x, wq, bq, wk, bk, wv, bv, ids, mask, cfg):
        # x: dense [n, L, d_model] compute dtype, or (gather) the fp32 table parameter [V, d_model]
        _need_gpu(x, wq, ids, mask)
        code, heads, gather = cfg["code"], cfg["heads"], ids is not None
        N, d_model = wq.shape
        d_head = N // heads
        ch = chunk(code)
        if gather:
            n, L = ids.shape
            src = cfg["table_packed"]
            ids = ids.contiguous()
        else:
            n, L, _ = x.shape
            src = x.contiguous()
            if src.dtype != torch_dtype(code):
                raise RuntimeError(f"dense MHSA input must be {torch_dtype(code)}, got {src.dtype}")
        ldx = src.shape[-1]
        wcat = torch.cat([wq, wk, wv], dim=0)
        w_p = pack(wcat, code)
        b_p = torch.cat([bq, bk, bv]).detach().float().contiguous()
        mask_c = mask.contiguous().float() if mask is not None else None
        dev = wq.device
        y = torch.empty(n, L, N, dtype=torch_dtype(code), device=dev)
        # training with a gather source: keep the gathered + dropped-out rows for the weight-gradient GEMM
        need_bwd = any(ctx.needs_input_grad[:7])
        keep_rows = gather and any(ctx.needs_input_grad[1:7])
        Kp = round_up(d_model, ch)
        x_rows = torch.empty(n * L, Kp, dtype=torch_dtype(code), device=dev) if keep_rows else None
        # scratch for the device-side compaction of non-padding rows (forward: live rows, their ids, padding rows)
        row_ws = torch.empty(3 * n * L + 3 * n + (n * L) // 32 + 32, dtype=torch.int32, device=dev) if keep_rows and code == _lib.NR_BF16 else None
        d = _lib.MhsaDesc(n=n, L=L, d_model=d_model, heads=heads, d_head=d_head, dtype=code,
                          src_kind=NR_SRC_GATHER if gather else NR_SRC_DENSE, x=ptr(src), ldx=ldx, ids=ptr(ids),
                          p_in=cfg["p_in"], seed_in=cfg["seed_in"], p_out=cfg["p_out"], seed_out=cfg["seed_out"],
                          mask=ptr(mask_c), w_qkv=ptr(w_p), ldw=w_p.shape[1], b_qkv=ptr(b_p),
                          x_rows=ptr(x_rows), ld_rows=Kp, row_ws=ptr(row_ws))
        # the fused title-level kernel keeps Q|K|V on chip: without a backward they are never written to HBM
        fused = bool(_lib.lib().nr_mhsa_fwd_fused(C.byref(d)))
        qkv = None if (fused and not need_bwd) else torch.empty(n * L, 3 * N, dtype=torch_dtype(code), device=dev)
        check(_lib.lib().nr_mhsa_fwd(C.byref(d), ptr(qkv), ptr(y), _stream()), "nr_mhsa_fwd")
        ctx.cfg, ctx.dims = cfg, (n, L, N, d_model, heads, d_head, ldx, gather)
        ctx.row_ws = row_ws      # the backward attention and the table-gradient GEMM reuse the compaction
        ctx.save_for_backward(src, ids, mask_c, w_p, b_p, wcat, qkv, x_rows)
        return y

    @staticmethod
    def backward(ctx, dy):
        src, ids, mask_c, w_p, b_p, wcat, qkv, x_rows = ctx.saved_tensors
        cfg = ctx.cfg
        n, L, N, d_model, heads, d_head, ldx, gather = ctx.dims
        code, dev = cfg["code"], dy.device
        dy = dy.contiguous()
        if dy.dtype != torch_dtype(code):
            dy = dy.to(torch_dtype(code))
        dqkv = torch.empty_like(qkv)
        flat = torch.zeros(3 * N * d_model + 3 * N, dtype=torch.float32, device=dev)     # dw | db, one fill
        dw, db = flat[:3 * N * d_model].view(3 * N, d_model), flat[3 * N * d_model:]
        need_x = ctx.needs_input_grad[0]
        dx = dtable = w_t = None
        row_ws = getattr(ctx, "row_ws", None)          # what the forward compacted (padding rows of qkv were never written)
        ws_ready = row_ws is not None
        if need_x:
            w_t = pack(wcat, code, transpose=True)                     # [d_model, 3N]
            if gather:
                dtable = torch.zeros(cfg["table_shape"], dtype=torch.float32, device=dev)
                if row_ws is None:
                    row_ws = torch.empty(3 * n * L + 3 * n + (n * L) // 32 + 32, dtype=torch.int32, device=dev)     # live-row compaction scratch
            else:
                dx = torch.empty(n, L, ldx, dtype=torch_dtype(code), device=dev)
        d = _lib.MhsaDesc(n=n, L=L, d_model=d_model, heads=heads, d_head=d_head, dtype=code,
                          src_kind=NR_SRC_GATHER if gather else NR_SRC_DENSE, x=ptr(src), ldx=ldx, ids=ptr(ids),
                          p_in=cfg["p_in"], seed_in=cfg["seed_in"], p_out=cfg["p_out"], seed_out=cfg["seed_out"],
                          mask=ptr(mask_c), w_qkv=ptr(w_p), ldw=w_p.shape[1], b_qkv=ptr(b_p),
                          x_rows=ptr(x_rows), ld_rows=x_rows.shape[1] if x_rows is not None else 0, row_ws=ptr(row_ws),
                          row_ws_ready=int(ws_ready))
        check(_lib.lib().nr_mhsa_bwd(C.byref(d), ptr(qkv), ptr(dy), ptr(dqkv), ptr(w_t), w_t.shape[1] if w_t is not None else 0,
                                     ptr(dw), ptr(db), ptr(dx), ptr(dtable), _stream()), "nr_mhsa_bwd")
        gx = dtable if gather else dx
        return (gx, dw[:N], db[:N], dw[N:2 * N], db[N:2 * N], dw[2 * N:], db[2 * N:], None, None, None)


def mhsa(x, wq, bq, wk, bk, wv, bv, heads: int, code: int, mask=None, ids=None, table=None, p_in=0.0, p_out=0.0):
    """Dense: x [n, L, d_model] (compute dtype).  Gather: ids int32 [n, L] + fp32 `table` parameter."""
    cfg = dict(code=code, heads=heads, p_in=float(p_in), p_out=float(p_out),
               seed_in=draw_seed() if p_in > 0 else 0, seed_out=draw_seed() if p_out > 0 else 0)
    if ids is not None:
        if ids.dtype != torch.int32:
            ids = ids.to(torch.int32)
        cfg["table_packed"] = table_cache.get(table, code)
        cfg["table_shape"] = tuple(table.shape)
        return MHSAFunction.apply(table, wq, bq, wk, bk, wv, bv, ids, mask, cfg)
    return MHSAFunction.apply(x, wq, bq, wk, bk, wv, bv, None, mask, cfg)


# ------------------------------------------------------------------------------------------ pooling
class PoolFunction(Function):
    """K5: AttentionPooling.forward, src/model/model_utils.py:13-31."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, mask, code):
        _need_gpu(x, w1, mask)
        n, L, N = x.shape
        q = w1.shape[0]
        x = x.contiguous()
        if x.dtype != torch_dtype(code):
            raise RuntimeError(f"pool input must be {torch_dtype(code)}, got {x.dtype}")
        dev = x.device
        w1_p = pack(w1, code)
        b1_c, w2_c, b2_c = (t.detach().float().contiguous() for t in (b1, w2.reshape(-1), b2))
        mask_c = mask.contiguous().float() if mask is not None else None
        e = torch.empty(n * L, q, dtype=torch_dtype(code), device=dev)
        alpha = torch.empty(n * L, dtype=torch.float32, device=dev)
        out = torch.empty(n, N, dtype=torch.float32, device=dev)
        d = _lib.PoolDesc(n=n, L=L, N=N, q=q, dtype=code, x=ptr(x), mask=ptr(mask_c), w1=ptr(w1_p), ldw1=w1_p.shape[1],
                          b1=ptr(b1_c), w2=ptr(w2_c), b2=ptr(b2_c))
        check(_lib.lib().nr_additive_pool_fwd(C.byref(d), ptr(e), ptr(alpha), ptr(out), N, _stream()), "nr_additive_pool_fwd")
        ctx.code, ctx.dims = code, (n, L, N, q)
        ctx.save_for_backward(x, mask_c, w1_p, b1_c, w2_c, b2_c, e, alpha, w1)
        return out

    @staticmethod
    def backward(ctx, g):
        x, mask_c, w1_p, b1_c, w2_c, b2_c, e, alpha, w1 = ctx.saved_tensors
        n, L, N, q = ctx.dims
        code, dev = ctx.code, g.device
        g = g.contiguous().float()
        w1_t = pack(w1, code, transpose=True) if ctx.needs_input_grad[0] else None    # [N, q]
        dpre = torch.empty_like(e)
        partial = torch.empty(n * (q + 1), dtype=torch.float32, device=dev)     # one row per workgroup (<= n of them)
        # one zero fill for the four accumulated gradients (views of a flat buffer; q*N and q are multiples of 4)
        flat = torch.zeros(q * N + 2 * q + 4, dtype=torch.float32, device=dev)
        dw1, db1, dw2, db2 = flat[:q * N].view(q, N), flat[q * N:q * N + q], flat[q * N + q:q * N + 2 * q], flat[q * N + 2 * q:q * N + 2 * q + 1]
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        d = _lib.PoolDesc(n=n, L=L, N=N, q=q, dtype=code, x=ptr(x), mask=ptr(mask_c), w1=ptr(w1_p), ldw1=w1_p.shape[1],
                          b1=ptr(b1_c), w2=ptr(w2_c), b2=ptr(b2_c))
        check(_lib.lib().nr_additive_pool_bwd(C.byref(d), ptr(e), ptr(alpha), ptr(g), N, ptr(w1_t),
                                              w1_t.shape[1] if w1_t is not None else 0, ptr(dpre), ptr(partial), ptr(dw1),
                                              ptr(db1), ptr(dw2), ptr(db2), ptr(dx), _stream()), "nr_additive_pool_bwd")
        return dx, dw1, db1, dw2.view(1, q), db2, None, None


def additive_pool(x, w1, b1, w2, b2, code: int, mask=None):
    return PoolFunction.apply(x, w1, b1, w2, b2, mask, code)


# ------------------------------------------------------------------------------------------ pad blend / cast
class BlendFunction(Function):
    """K6: x*m + pad_doc*(1-m) (src/model/NRMS.py:59-60, src/model/NAML.py:94-95), emitted in the
    compute dtype.  mask=None is a plain fp32 -> compute-dtype cast with an fp32 gradient."""

    @staticmethod
    def forward(ctx, x, mask, pad, code):
        _need_gpu(x, mask, pad)
        n, L, N = x.shape
        x = x.contiguous().float()
        mask_c = mask.contiguous().float() if mask is not None else None
        pad_c = pad.detach().reshape(-1).float().contiguous() if pad is not None else None
        out = torch.empty(n, L, N, dtype=torch_dtype(code), device=x.device)
        check(_lib.lib().nr_pad_blend_fwd(ptr(x), ptr(mask_c), ptr(pad_c), ptr(out), n, L, N, code, _stream()), "nr_pad_blend_fwd")
        ctx.code, ctx.dims, ctx.pad_shape = code, (n, L, N), (tuple(pad.shape) if pad is not None else None)
        ctx.save_for_backward(mask_c)
        return out

    @staticmethod
    def backward(ctx, dout):
        (mask_c,) = ctx.saved_tensors
        n, L, N = ctx.dims
        dout = dout.contiguous()
        if dout.dtype != torch_dtype(ctx.code):
            dout = dout.to(torch_dtype(ctx.code))
        dx = torch.empty(n, L, N, dtype=torch.float32, device=dout.device)
        dpad = torch.zeros(N, dtype=torch.float32, device=dout.device) if mask_c is not None else None
        check(_lib.lib().nr_pad_blend_bwd(ptr(dout), ptr(mask_c), ptr(dx), ptr(dpad), n, L, N, ctx.code, _stream()), "nr_pad_blend_bwd")
        return dx, None, (dpad.view(ctx.pad_shape) if dpad is not None and ctx.pad_shape is not None else None), None


def pad_blend(x, mask, pad, code: int):
    return BlendFunction.apply(x, mask, pad, code)


def to_compute(x, code: int):
    """fp32 [n, L, N] -> compute dtype through the blend kernel (no mask)."""
    if x.dtype == torch_dtype(code) and code == NR_F32:
        return x
    return BlendFunction.apply(x, None, None, code)


# ------------------------------------------------------------------------------------------ scorer + CE
class ScoreCEFunction(Function):
    """K8: torch.bmm scorer + nn.CrossEntropyLoss, src/model/NRMS.py:93-94 / src/model/NAML.py:128-129."""

    @staticmethod
    def forward(ctx, cand, user, label):
        _need_gpu(cand, user, label)
        B, Cn, N = cand.shape
        cand, user = cand.contiguous().float(), user.contiguous().float()
        label = label.contiguous().to(torch.int64)
        score = torch.empty(B, Cn, dtype=torch.float32, device=cand.device)
        loss = torch.empty((), dtype=torch.float32, device=cand.device)
        lossvec = torch.empty(B, dtype=torch.float32, device=cand.device)
        check(_lib.lib().nr_score_ce_fwd(ptr(cand), N, ptr(user), ptr(label), ptr(score), ptr(loss), ptr(lossvec), B, Cn, N,
                                         _stream()), "nr_score_ce_fwd")
        ctx.save_for_backward(cand, user, label, score)
        return loss, score

    @staticmethod
    def backward(ctx, gloss, gscore):
        cand, user, label, score = ctx.saved_tensors
        B, Cn, N = cand.shape
        gl = gloss.contiguous().float() if gloss is not None else None
        gs = gscore.contiguous().float() if gscore is not None else None
        dcand = torch.empty_like(cand)
        duser = torch.empty_like(user)
        check(_lib.lib().nr_score_ce_bwd(ptr(cand), N, ptr(user), ptr(label), ptr(score), ptr(gl), ptr(gs), ptr(dcand), N,
                                         ptr(duser), B, Cn, N, _stream()), "nr_score_ce_bwd")
        return dcand, duser, None


def score_ce(cand, user, label):
    return ScoreCEFunction.apply(cand, user, label)


# ------------------------------------------------------------------------------------------ Conv1d k=3 over gathered titles
class ConvFunction(Function):
    """K4 (+K1, K2): title-embedding row gather -> dropout -> Conv1d(k=3, pad=1), src/model/NAML.py:47-54."""

    @staticmethod
    def forward(ctx, w, b, ids, cfg):
        _need_gpu(w, ids)
        code, T, D = cfg["code"], cfg["T"], cfg["D"]
        table_p = cfg["table_packed"]                 # [V*T, Dp]
        Dp = table_p.shape[1]
        N = w.shape[0]
        n, stride = ids.shape[0], ids.stride(0)
        dev = w.device
        w_p = torch.empty(N, 3 * Dp, dtype=torch_dtype(code), device=dev)
        wc = w.detach().float().contiguous()
        check(_lib.lib().nr_pack_conv_w(ptr(wc), N, D, ptr(w_p), Dp, code, _stream()), "nr_pack_conv_w")
        b_c = b.detach().float().contiguous()
        y = torch.empty(n, T, N, dtype=torch_dtype(code), device=dev)
        # bf16: the im2col rows (gather + dropout, 3 taps) are stored once and feed the dense LDS-DMA GEMMs of both passes
        x_rows = torch.empty(n * T, 3 * Dp, dtype=torch.bfloat16, device=dev) if code == _lib.NR_BF16 and n > 0 else None
        d = _lib.ConvDesc(n=n, T=T, D=D, Dp=Dp, N=N, dtype=code, table=ptr(table_p), ids=ids.data_ptr(), ids_stride=stride,
                          p_in=cfg["p_in"], seed_in=cfg["seed_in"], w_pack=ptr(w_p), bias=ptr(b_c), x_rows=ptr(x_rows),
                          ld_rows=3 * Dp)
        check(_lib.lib().nr_conv1d_k3_fwd(C.byref(d), ptr(y), _stream()), "nr_conv1d_k3_fwd")
        ctx.cfg, ctx.dims = cfg, (n, T, D, Dp, N, stride)
        ctx.ids = ids                                   # keeps the (possibly strided) id view alive
        ctx.x_rows = x_rows if any(ctx.needs_input_grad[:2]) else None
        ctx.save_for_backward(table_p, w_p, b_c)
        return y

    @staticmethod
    def backward(ctx, dy):
        table_p, w_p, b_c = ctx.saved_tensors
        n, T, D, Dp, N, stride = ctx.dims
        cfg, code, dev = ctx.cfg, ctx.cfg["code"], dy.device
        dy = dy.contiguous()
        if dy.dtype != torch_dtype(code):
            dy = dy.to(torch_dtype(code))
        dwp = torch.zeros(N, 3 * Dp, dtype=torch.float32, device=dev)
        db = torch.zeros(N, dtype=torch.float32, device=dev)
        bwd_ws = torch.empty(n + (n * T) // 32 + 16, dtype=torch.int32, device=dev) if ctx.x_rows is not None else None
        d = _lib.ConvDesc(n=n, T=T, D=D, Dp=Dp, N=N, dtype=code, table=ptr(table_p), ids=ctx.ids.data_ptr(), ids_stride=stride,
                          p_in=cfg["p_in"], seed_in=cfg["seed_in"], w_pack=ptr(w_p), bias=ptr(b_c), x_rows=ptr(ctx.x_rows),
                          ld_rows=3 * Dp, bwd_ws=ptr(bwd_ws))
        check(_lib.lib().nr_conv1d_k3_bwd(C.byref(d), ptr(dy), ptr(dwp), ptr(db), _stream()), "nr_conv1d_k3_bwd")
        dw = torch.empty(N, D, 3, dtype=torch.float32, device=dev)
        check(_lib.lib().nr_unpack_conv_dw(ptr(dwp), N, D, Dp, ptr(dw), _stream()), "nr_unpack_conv_dw")
        return dw, db, None, None


def conv1d_k3_gather(table, w, b, ids, T: int, D: int, code: int, p_in=0.0):
    """ids: int32 view [n] (any stride) of news ids; table: fp32 [V, T*D] (frozen on this path)."""
    if table.requires_grad:
        raise RuntimeError("NAML title-embedding table must be frozen (freeze_embedding=True, as src/demo.sh:12): "
                           "its [V, T*D] dense gradient is out of scope on this path")
    cfg = dict(code=code, T=T, D=D, p_in=float(p_in), seed_in=draw_seed() if p_in > 0 else 0,
               table_packed=table_cache.get(table, code, row_cols=D))
    return ConvFunction.apply(w, b, ids, cfg)


# ------------------------------------------------------------------------------------------ gather + Linear (category views)
class GatherLinearFunction(Function):
    """K7: Embedding(padding_idx=0) -> Linear, src/model/NAML.py:19-24,60-68."""

    @staticmethod
    def forward(ctx, emb, w, b, ids, code):
        _need_gpu(emb, w, ids)
        M, stride = ids.shape[0], ids.stride(0)
        N, K = w.shape
        emb_p, w_p = pack(emb, code), pack(w, code)
        b_c = b.detach().float().contiguous()
        out = torch.empty(M, N, dtype=torch.float32, device=w.device)
        d = _lib.LinearDesc(M=M, K=K, N=N, dtype=code, src_kind=NR_SRC_GATHER, x=ptr(emb_p), ldx=emb_p.shape[1],
                            ids=ids.data_ptr(), ids_stride=stride, w=ptr(w_p), ldw=w_p.shape[1], bias=ptr(b_c), w_t=0, ldwt=0)
        check(_lib.lib().nr_linear_fwd(C.byref(d), ptr(out), N, _stream()), "nr_linear_fwd")
        ctx.code, ctx.dims, ctx.ids = code, (M, K, N, stride, tuple(emb.shape)), ids
        ctx.save_for_backward(emb_p, w_p, b_c, w)
        return out

    @staticmethod
    def backward(ctx, dout):
        emb_p, w_p, b_c, w = ctx.saved_tensors
        M, K, N, stride, emb_shape = ctx.dims
        code, dev = ctx.code, dout.device
        dout = dout.contiguous().float()
        Nc = round_up(N, chunk(code))
        ws = torch.empty(M, Nc, dtype=torch_dtype(code), device=dev)
        dw = torch.zeros(N, K, dtype=torch.float32, device=dev)
        db = torch.zeros(N, dtype=torch.float32, device=dev)
        need_t = ctx.needs_input_grad[0]
        dtable = torch.zeros(emb_shape, dtype=torch.float32, device=dev) if need_t else None
        w_t = pack(w, code, transpose=True, ld=Nc) if need_t else None               # [K, Nc]
        d = _lib.LinearDesc(M=M, K=K, N=N, dtype=code, src_kind=NR_SRC_GATHER, x=ptr(emb_p), ldx=emb_p.shape[1],
                            ids=ctx.ids.data_ptr(), ids_stride=stride, w=ptr(w_p), ldw=w_p.shape[1], bias=ptr(b_c),
                            w_t=ptr(w_t), ldwt=Nc if need_t else 0)
        check(_lib.lib().nr_linear_bwd(C.byref(d), ptr(dout), N, ptr(ws), ptr(dw), ptr(db), ptr(dtable), _stream()), "nr_linear_bwd")
        return dtable, dw, db, None, None


def gather_linear(emb, w, b, ids, code: int):
    return GatherLinearFunction.apply(emb, w, b, ids, code)


# ------------------------------------------------------------------------------------------ plain row gather (eval path)
def embed_gather(table: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
    """out[i] = table[ids[i]] (fp32), no gradient: the eval-time news-vector lookup of src/dataset.py:68,72."""
    _need_gpu(table, ids)
    table = table.detach().float().contiguous()
    ids = ids.to(torch.int32).contiguous()
    cols = table.shape[1]
    out = torch.empty(*ids.shape, cols, dtype=torch.float32, device=table.device)
    check(_lib.lib().nr_embed_gather_fwd(ptr(table), cols, NR_F32, ptr(ids), ids.numel(), 1, cols, ptr(out), cols, _stream()),
          "nr_embed_gather_fwd")
    return out


def score_eval(news_vecs, cand_ids, imp_of, user_vecs) -> torch.Tensor:
    """score[i] = <news_vecs[cand_ids[i]], user_vecs[imp_of[i]]> — the per-impression np.dot of src/main.py:253."""
    _need_gpu(news_vecs, cand_ids, imp_of, user_vecs)
    news_vecs, user_vecs = news_vecs.detach().float().contiguous(), user_vecs.detach().float().contiguous()
    cand_ids, imp_of = cand_ids.to(torch.int32).contiguous(), imp_of.to(torch.int32).contiguous()
    N = news_vecs.shape[1]
    out = torch.empty(cand_ids.numel(), dtype=torch.float32, device=news_vecs.device)
    check(_lib.lib().nr_score_eval(ptr(news_vecs), N, ptr(cand_ids), ptr(imp_of), ptr(user_vecs), N, ptr(out),
                                   cand_ids.numel(), N, _stream()), "nr_score_eval")
    return out


def dropout_mask(count: int, p: float, seed: int, device) -> torch.Tensor:
    """Test hook: the keep mask the kernels use for element indices [0, count)."""
    out = torch.empty(count, dtype=torch.float32, device=device)
    check(_lib.lib().nr_dropout_mask(ptr(out), count, float(p), int(seed), _stream()), "nr_dropout_mask")
    return out


def gemm_nt(a: torch.Tensor, b: torch.Tensor, bias=None, act_tanh=False, out_dtype=None) -> torch.Tensor:
    """C = a . b^T (+bias)(tanh): the MFMA GEMM building block on dense operands (unit tests / measurement)."""
    _need_gpu(a, b)
    code = NR_BF16 if a.dtype == torch.bfloat16 else NR_F32
    M, K = a.shape
    N = b.shape[0]
    out_dtype = out_dtype or a.dtype
    c = torch.empty(M, N, dtype=out_dtype, device=a.device)
    check(_lib.lib().nr_gemm_nt(code, ptr(a), a.stride(0), ptr(b), b.stride(0), ptr(bias), int(act_tanh), ptr(c), N,
                                NR_BF16 if out_dtype == torch.bfloat16 else NR_F32, M, N, K, _stream()), "nr_gemm_nt")
    return c


def gemm_tn(dc: torch.Tensor, a: torch.Tensor, with_bias_grad=True):
    """dW = dc^T . a (fp32), db = column sums of dc: the weight-gradient GEMM building block."""
    _need_gpu(dc, a)
    code = NR_BF16 if a.dtype == torch.bfloat16 else NR_F32
    M, N = dc.shape
    K = a.shape[1]
    dw = torch.zeros(N, K, dtype=torch.float32, device=a.device)
    db = torch.zeros(N, dtype=torch.float32, device=a.device) if with_bias_grad else None
    check(_lib.lib().nr_gemm_tn(code, ptr(dc), dc.stride(0), ptr(a), a.stride(0), ptr(dw), K, ptr(db), M, N, K, _stream()),
          "nr_gemm_tn")
    return dw, db
