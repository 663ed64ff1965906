"""torch.autograd bindings of the libnrhip C ABI (include/nrhip.h).

PyTorch supplies device memory, the current HIP stream and autograd bookkeeping; every
arithmetic step on the hot path runs in the hand-written HIP kernels.  Tensors must live on
the GPU: there is no CPU fallback (a RuntimeError is raised instead).
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
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


# Test hook: when True, buffers the kernels are documented to leave partly unwritten (qkv / dqkv rows of all-padding
# sequences) are pre-filled with NaN, so a read of an unwritten row poisons the result instead of passing silently.
POISON_WORKSPACES = False


def _scratch(*shape, dtype, device):
    t = torch.empty(*shape, dtype=dtype, device=device)
    if POISON_WORKSPACES and t.is_floating_point():
        t.fill_(float("nan"))
    return t


# Index validation (what torch.nn.functional.embedding / cross_entropy raise IndexError for).  The kernels trust their
# indices; with CHECK_INDICES on (env NR_CHECK_INDICES=1, or set the attribute) every id / label tensor is range-checked
# on the device before use, at the price of one host synchronisation per check.
CHECK_INDICES = os.environ.get("NR_CHECK_INDICES", "0") not in ("", "0")


def check_ids(ids: torch.Tensor, rows: int, what: str = "index") -> None:
    """Raise IndexError if any entry of the int32 tensor `ids` (any 1-D stride) lies outside [0, rows)."""
    if ids.numel() == 0:
        return
    bad = torch.zeros(1, dtype=torch.int32, device=ids.device)
    if ids.dim() == 1:
        count, stride, base = ids.shape[0], ids.stride(0), ids
    else:
        base = ids.contiguous()
        count, stride = base.numel(), 1
    check(_lib.lib().nr_check_ids(ptr(base), count, stride, int(rows), ptr(bad), _stream()), "nr_check_ids")
    n = int(bad.item())
    if n:
        raise IndexError(f"{what}: {n} entries outside [0, {rows})")


def check_labels(label: torch.Tensor, classes: int) -> None:
    bad = torch.zeros(1, dtype=torch.int32, device=label.device)
    check(_lib.lib().nr_check_labels(ptr(label), label.numel(), int(classes), ptr(bad), _stream()), "nr_check_labels")
    n = int(bad.item())
    if n:
        raise IndexError(f"label: {n} entries outside [0, {classes})")


def _ws(nbytes: int, device) -> torch.Tensor:
    """A workspace of the size the library asked for (bytes -> int32 elements)."""
    return torch.empty((int(nbytes) + 3) // 4, dtype=torch.int32, device=device)


# Zero-gradient hint between two adjacent backward calls: the pooling backward knows which sequences had a zero pooled
# gradient (their dx rows are exact zeros) and leaves [n] flags in its workspace; when the VERY NEXT libnrhip backward call
# is the producer of the pooled tensor (MHSA / conv) and receives that same dx buffer as its dy, it takes the flags
# instead of scanning dy (0.1 ms at the bench shape).  Anything else in between drops the hint.
_bwd_seq = 0
_flag_hint = None


def _enter_backward() -> int:
    global _bwd_seq
    _bwd_seq += 1
    return _bwd_seq


def _offer_flags(seq, dx, n, flags_ptr, keepalive, lazy=False) -> None:
    """lazy: the dx rows of sequences whose flag is 0 are zeros only NEAR flagged ones and unwritten memory elsewhere
    (nr_pool_desc.dx_far_unwritten): the taker must be told, and a consumer that cannot take the flags must not read dx."""
    global _flag_hint
    if dx is not None:
        dx._nr_lazy_rows = bool(lazy)
    # dx._version: autograd may ADD another consumer's gradient into dx in place (same pointer, same size) before the next
    # libnrhip backward sees it -- the flags would then drop sequences whose gradient is no longer zero
    _flag_hint = ((seq, dx.data_ptr(), dx.numel(), n, flags_ptr, keepalive, dx._version, weakref.ref(dx), bool(lazy))
                  if flags_ptr and dx is not None else None)


def _take_flags(seq, dy, n):
    """(device pointer, keep-alive tensor, lazy) of the [n] flags for `dy`, or (0, None, False)."""
    global _flag_hint
    h, _flag_hint = _flag_hint, None
    if h is not None and h[0] == seq - 1 and h[1] == dy.data_ptr() and h[2] == dy.numel() and h[3] == n:
        dx = h[7]()
        if dx is not None and dx._version == h[6] and dy._version == h[6]:
            return h[4], h[5], h[8]
    if getattr(dy, "_nr_lazy_rows", False):
        raise RuntimeError("this gradient was produced with unwritten rows for zero-gradient sequences (additive_pool lazy_dx=True) "
                           "but the flags that say which ones did not reach its consumer: something else ran between the two backward "
                           "calls -- construct the pooling with lazy_dx=False")
    return 0, None, False


# Deterministic mode (nr_set_deterministic): gradients that several workgroups add into are accumulated in fixed point with
# integer atomics, so they are bit-reproducible from run to run.  Costs a zeroed scratch of 8 bytes per accumulated element
# of the largest backward call and a flush pass per call; off by default.
_det_scratch = None


def set_deterministic(on: bool, elements: int = 1 << 24, device=None) -> None:
    """on: register a zero-filled scratch of `elements` int64 (default 16 M = 128 MB: NRMS with a 30 000-row trainable
    word table needs 3*400*301 + 30 000*300 = 9.4 M) with libnrhip; off: plain fp32 atomics again."""
    global _det_scratch
    if on:
        dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        _det_scratch = torch.zeros(int(elements), dtype=torch.int64, device=dev)
        check(_lib.lib().nr_set_deterministic(ptr(_det_scratch), _det_scratch.numel() * 8), "nr_set_deterministic")
    else:
        check(_lib.lib().nr_set_deterministic(None, 0), "nr_set_deterministic")
        _det_scratch = None


def grad_target(p):
    """The preallocated gradient view of a parameter that lives in a flat bucket (parallel.FlatBucket sets
    `_nr_grad`), or None.  Backward passes ACCUMULATE into such a view directly -- the kernels add into their dW / db /
    dtable outputs anyway -- and hand autograd None for that parameter: no per-call zero fill, no AccumulateGrad copy."""
    return getattr(p, "_nr_grad", None) if p is not None else None


def draw_seed() -> int:
    """A 31-bit dropout seed from torch's CPU generator (reproducible under torch.manual_seed)."""
    return int(torch.randint(0, 2 ** 31 - 1, (1,)).item())


class _PackCache:
    """Packed copies of the small weights that live in a flat parameter bucket (parallel.FlatBucket registers its buffer).
    A training step asks for the same ~9 packs (Q|K|V and att_fc1 weights, plain and transposed) after every optimizer
    step; each used to be its own 6-us launch.  Here the first request after a parameter update (ops.param_epoch moved, or a
    torch-side in-place write bumped the tensor's version) refreshes EVERY pack seen so far in ONE launch (nr_cast_pad_batch).
    Tensors outside a registered bucket are never cached."""
    MAX_ELEMS = 1 << 21                # (larger operands -- embedding tables -- go through _TableCache)

    def __init__(self):
        self._ranges = []              # (first byte, last byte + 1) of registered flat parameter buffers
        self._e = {}                   # key -> [src view, dst, transpose, code, stamp]

    def register_buffer(self, t: torch.Tensor, versions=None) -> None:
        """t: the bucket's flat parameter tensor.  Its packs are served for as long as that tensor object lives.
        versions: optional callable -> tuple of the version counters of the Parameters that live in `t` (a Parameter whose
        `.data` was pointed at a view of `t` keeps its own counter, so `t._version` alone misses torch-side writes to it)."""
        self._ranges.append((weakref.ref(t), t.data_ptr(), t.data_ptr() + t.numel() * t.element_size(), versions))

    def clear(self) -> None:
        self._ranges, self._e = [], {}

    def _purge(self) -> None:
        dead = [(a, b) for r, a, b, _ in self._ranges if r() is None]
        if dead:
            self._ranges = [x for x in self._ranges if x[0]() is not None]
            for k in [k for k in self._e if any(a <= k[0] < b for a, b in dead)]:
                del self._e[k]

    def _bucket_versions(self, src):
        """None: `src` lies in no registered bucket (never cached); else the bucket's parameter-version tuple (or ())."""
        self._purge()
        if src.numel() > self.MAX_ELEMS:
            return None
        p = src.data_ptr()
        for _, a, b, versions in self._ranges:
            if a <= p < b:
                return versions() if versions is not None else ()
        return None

    def get(self, src, code, transpose, ld):
        bv = self._bucket_versions(src)
        if bv is None:
            return None
        key = (src.data_ptr(), tuple(src.shape), bool(transpose), int(ld), int(code), src.device.index)
        stamp = (param_epoch, src._version, bv)
        ent = self._e.get(key)
        if ent is not None and ent[4] == stamp:
            return ent[1]
        if ent is None:
            rows, cols = src.shape
            dst = torch.empty(cols if transpose else rows, ld, dtype=torch_dtype(code), device=src.device)
            ent = self._e[key] = [src, dst, bool(transpose), int(code), None]
        # refresh this entry and every other stale one of the same dtype / device in one launch
        def stamp_of(e):
            return (param_epoch, e[0]._version, self._bucket_versions(e[0]))
        todo = [e for k, e in self._e.items()
                if e[3] == code and k[5] == src.device.index and e[4] != stamp_of(e)][:_lib.CAST_BATCH_MAX]
        if not any(e is ent for e in todo):
            todo = [ent] + todo[:_lib.CAST_BATCH_MAX - 1]
        jobs = (_lib.CastJob * len(todo))()
        for j, e in zip(jobs, todo):
            r, c = e[0].shape
            j.src, j.dst, j.rows, j.cols, j.ld_src, j.ld_dst, j.transpose = ptr(e[0]), ptr(e[1]), r, c, c, e[1].shape[1], int(e[2])
        check(_lib.lib().nr_cast_pad_batch(jobs, len(todo), code, _stream()), "nr_cast_pad_batch")
        for e in todo:
            e[4] = stamp_of(e)
        return ent[1]


pack_cache = _PackCache()


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
    cached = pack_cache.get(src, code, transpose, ld)
    if cached is not None:
        return cached
    dst = torch.empty(drows, ld, dtype=torch_dtype(code), device=src.device)
    check(_lib.lib().nr_cast_pad(ptr(src), rows, cols, cols, ptr(dst), ld, code, int(transpose), _stream()), "nr_cast_pad")
    return dst


class _TableCache:
    """Compute-dtype copies of embedding tables, refreshed when the fp32 master changes (tensor version counter,
    storage pointer, shape), so a frozen table is packed once.

    An entry lives exactly as long as the Parameter it was packed from: the key is `id(weight)`, and a
    `weakref.finalize` on that Parameter evicts the entry when the Parameter is collected, so a recycled `id()` can
    never be served another model's table and dead tables do not stay pinned in HBM.  A held weak reference is
    compared on every hit as well.  Writes that bypass autograd's version counter (`weight.data.copy_(...)`, as a
    parameter broadcast does) need an explicit `table_cache.invalidate(weight)`."""

    def __init__(self):
        self._c = {}
        self._watched = set()            # ids of parameters that carry an eviction finalizer

    def _drop(self, wid):
        for key in [k for k in self._c if k[0] == wid]:
            del self._c[key]

    def _evict(self, wid):                # the parameter died
        self._drop(wid)
        self._watched.discard(wid)

    def invalidate(self, weight=None):
        """Forget the packed copy of `weight` (all copies if None); the next use re-packs from the fp32 master."""
        if weight is None:
            self._c.clear()
        else:
            self._drop(id(weight))

    def current(self, weight):
        """[(code, row_cols, packed copy)] of the up-to-date entries of `weight` (what an optimizer that rewrites the master
        behind autograd's back can refresh in place instead of invalidating, parallel.FlatBucket.adam_step)."""
        w = weight.detach()
        sig = (weight._version, w.data_ptr(), tuple(w.shape))
        return [(k[1], k[2], e[1]) for k, e in self._c.items() if k[0] == id(weight) and e[0] == sig and e[2]() is weight]

    def get(self, weight: torch.Tensor, code: int, row_cols: Optional[int] = None) -> torch.Tensor:
        w = weight.detach()
        cols = row_cols or w.shape[1]
        if code == NR_F32 and cols % 4 == 0 and w.is_contiguous() and w.dtype == torch.float32:
            return w.view(-1, cols)                       # already a valid operand: no copy
        key = (id(weight), code, cols)
        ent = self._c.get(key)
        sig = (weight._version, w.data_ptr(), tuple(w.shape))
        if ent is None or ent[0] != sig or ent[2]() is not weight:
            if id(weight) not in self._watched:
                self._watched.add(id(weight))
                weakref.finalize(weight, self._evict, id(weight))
            ent = (sig, pack(w.reshape(-1, cols), code), weakref.ref(weight))
            self._c[key] = ent
        return ent[1]


table_cache = _TableCache()

# Bumped whenever parameters are rewritten behind autograd's back (parallel.FlatBucket's Adam kernel): caches derived from
# parameter VALUES (not just from their version counters) key on it.
param_epoch = 0


def bump_param_epoch() -> None:
    global param_epoch
    param_epoch += 1


class _ProjectedTables:
    """Eval-mode shortcut of the title-level MHSA: without input dropout Q|K|V of a token are
    `W_{Q|K|V} e_w + b` of its table row e_w -- a function of the token ID alone -- so the [V, d_model] table is projected
    ONCE to [V, 3N] (one GEMM over V rows instead of one over every token occurrence: 100 000 news x 30 tokens -> 30 000
    rows) and the attention kernel gathers projected rows.  Same GEMM kernel, same operands, same rounding as projecting each
    occurrence: the values are identical.  Rebuilt when the table, a weight or the parameter epoch changes."""

    def __init__(self):
        self._c = {}

    def get(self, table, params, wcat, bcat, code):
        sig = (param_epoch, table._version, table.data_ptr()) + tuple((p._version, p.data_ptr()) for p in params)
        ent = self._c.get(id(table))
        if ent is None or ent[0] != sig or ent[2]() is not table:
            tp = table_cache.get(table, code)                                  # [V, ld] packed operand
            w_p = pack(wcat, code)                                             # [3N, ld], same zero-padded K as the table
            if w_p.shape[1] != tp.shape[1]:
                raise RuntimeError(f"projected table: operand leading dimensions differ ({w_p.shape[1]} vs {tp.shape[1]})")
            proj = gemm_nt(tp, w_p, bias=bcat.detach().float().contiguous())   # [V, 3N], bias added, rounded to bf16 once
            if id(table) not in self._c:
                weakref.finalize(table, self._c.pop, id(table), None)
            ent = (sig, proj, weakref.ref(table))
            self._c[id(table)] = ent
        return ent[1]


projected_tables = _ProjectedTables()


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
        flat = cfg.get("flat")             # W_Q|W_K|W_V (and the biases) adjacent in a flat bucket: no concatenation
        if flat is not None:
            wcat, b_p = flat["w"], flat["b"]
        else:
            wcat = torch.cat([wq, wk, wv], dim=0)
            b_p = torch.cat([bq, bk, bv]).detach().float().contiguous()
        w_p = pack(wcat, code)
        mask_c = mask.contiguous().float() if mask is not None else None
        dev = wq.device
        y = _scratch(n, L, N, dtype=torch_dtype(code), device=dev)      # (poisoned in the tests: far unneeded rows may stay unwritten)
        # training with a gather source: keep the gathered + dropped-out rows for the weight-gradient GEMM
        need_bwd = any(ctx.needs_input_grad[:7])
        keep_rows = gather and any(ctx.needs_input_grad[1:7])
        Kp = round_up(d_model, ch)
        x_rows = torch.empty(n * L, Kp, dtype=torch_dtype(code), device=dev) if keep_rows else None
        d = _lib.MhsaDesc(n=n, L=L, d_model=d_model, heads=heads, d_head=d_head, dtype=code,
                          src_kind=NR_SRC_GATHER if gather else NR_SRC_DENSE, x=ptr(src), ldx=ldx, ids=ptr(ids),
                          p_in=cfg["p_in"], seed_in=cfg["seed_in"], p_out=cfg["p_out"], seed_out=cfg["seed_out"],
                          mask=ptr(mask_c), w_qkv=ptr(w_p), ldw=w_p.shape[1], b_qkv=ptr(b_p),
                          x_rows=ptr(x_rows), ld_rows=Kp, seq_needed=ptr(cfg.get("needed")),
                          y_far_unwritten=int(bool(cfg.get("far_unwritten")) and cfg.get("needed") is not None),
                          table_rows=cfg["table_shape"][0] if gather else 0)     # (sizes the id-sort scratch of the backward)
        # scratch for the device-side compaction of non-padding rows (forward: live rows, their ids, padding rows;
        # backward: live slabs, sequence list) -- sized by the library
        row_ws = _ws(_lib.lib().nr_mhsa_workspace_bytes(C.byref(d)), dev) if keep_rows and code == _lib.NR_BF16 else None
        d.row_ws, d.row_ws_bytes = ptr(row_ws), (row_ws.numel() * 4 if row_ws is not None else 0)
        # the fused title-level kernel keeps Q|K|V on chip: without a backward they are never written to HBM
        fused = bool(_lib.lib().nr_mhsa_fwd_fused(C.byref(d)))
        qkv = None if (fused and not need_bwd) else _scratch(n * L, 3 * N, dtype=torch_dtype(code), device=dev)
        check(_lib.lib().nr_mhsa_fwd(C.byref(d), ptr(qkv), ptr(y), _stream()), "nr_mhsa_fwd")
        cfg["compact_rows"] = bool(keep_rows and _lib.lib().nr_mhsa_compact_rows(C.byref(d)))    # (read by ops.mhsa after apply)
        ctx.cfg, ctx.dims = cfg, (n, L, N, d_model, heads, d_head, ldx, gather)
        ctx.row_ws = row_ws      # the backward attention and the table-gradient GEMM reuse the compaction
        ctx.save_for_backward(src, ids, mask_c, w_p, b_p, wcat, qkv, x_rows)
        return y

    @staticmethod
    def backward(ctx, dy):
        seq = _enter_backward()
        src, ids, mask_c, w_p, b_p, wcat, qkv, x_rows = ctx.saved_tensors
        cfg = ctx.cfg
        n, L, N, d_model, heads, d_head, ldx, gather = ctx.dims
        code, dev = cfg["code"], dy.device
        dy = dy.contiguous()
        if dy.dtype != torch_dtype(code):
            dy = dy.to(torch_dtype(code))
        seq_nz, seq_nz_owner, dy_lazy = _take_flags(seq, dy, n)
        dqkv = _scratch(*qkv.shape, dtype=qkv.dtype, device=dev)
        bucket = cfg.get("flat")
        if bucket is not None:
            dw, db = bucket["gw"], bucket["gb"]                    # accumulated in place
        else:
            flat = torch.zeros(3 * N * d_model + 3 * N, dtype=torch.float32, device=dev)     # dw | db, one fill
            dw, db = flat[:3 * N * d_model].view(3 * N, d_model), flat[3 * N * d_model:]
        need_x = ctx.needs_input_grad[0]
        dx = dtable = w_t = None
        row_ws = getattr(ctx, "row_ws", None)          # what the forward compacted (padding rows of qkv were never written)
        ws_ready = row_ws is not None
        d = _lib.MhsaDesc(n=n, L=L, d_model=d_model, heads=heads, d_head=d_head, dtype=code,
                          src_kind=NR_SRC_GATHER if gather else NR_SRC_DENSE, x=ptr(src), ldx=ldx, ids=ptr(ids),
                          p_in=cfg["p_in"], seed_in=cfg["seed_in"], p_out=cfg["p_out"], seed_out=cfg["seed_out"],
                          mask=ptr(mask_c), w_qkv=ptr(w_p), ldw=w_p.shape[1], b_qkv=ptr(b_p),
                          x_rows=ptr(x_rows), ld_rows=x_rows.shape[1] if x_rows is not None else 0,
                          table_rows=cfg["table_shape"][0] if gather else 0, seq_nz=seq_nz, row_ws_ready=int(ws_ready),
                          dy_far_unwritten=int(dy_lazy),
                          seq_needed=ptr(cfg.get("needed")))     # (the forward may have left far all-padding x_rows unwritten)
        if need_x:
            w_t = pack(wcat, code, transpose=True)                     # [d_model, 3N]
            if gather:
                dtable = cfg.get("table_grad")
                if dtable is None:
                    dtable = torch.zeros(cfg["table_shape"], dtype=torch.float32, device=dev)
                if row_ws is None:
                    row_ws = _ws(_lib.lib().nr_mhsa_workspace_bytes(C.byref(d)), dev)     # live-row compaction scratch
            else:
                dx = torch.empty(n, L, ldx, dtype=torch_dtype(code), device=dev)
        d.row_ws, d.row_ws_bytes = ptr(row_ws), (row_ws.numel() * 4 if row_ws is not None else 0)
        def run(phase):
            d.bwd_phase = phase
            check(_lib.lib().nr_mhsa_bwd(C.byref(d), ptr(qkv), ptr(dy), ptr(dqkv), ptr(w_t), w_t.shape[1] if w_t is not None else 0,
                                         ptr(dw), ptr(db), ptr(dx), ptr(dtable), _stream()), "nr_mhsa_bwd")
        # A data-parallel bucket wants to know the moment the (large) table gradient is complete: the backward then runs in two
        # phases -- table gradient first, the hook starts its all-reduce, the weight-gradient GEMM runs underneath it.
        ready = cfg.get("table_grad_ready") if (gather and cfg.get("table_grad") is not None and _det_scratch is None) else None
        if ready is not None:
            run(1)
            ready()
            run(2)
        else:
            run(0)
        gx = (None if cfg.get("table_grad") is not None else dtable) if gather else dx
        if bucket is not None:
            return (gx, None, None, None, None, None, None, None, None, None)
        return (gx, dw[:N], db[:N], dw[N:2 * N], db[N:2 * N], dw[2 * N:], db[2 * N:], None, None, None)


USE_PROJECTED_TABLE = True      # eval-mode shortcut of the gather MHSA (see _ProjectedTables); off: project every occurrence


def _mhsa_projected(table, params, flat, ids, mask, heads, code, p_out):
    """No-grad, no input dropout, bf16, L <= 32: attention over projections gathered from the once-projected table."""
    wq, bq, wk, bk, wv, bv = params
    if flat is not None:
        wcat, bcat = flat["w"], flat["b"]
    else:
        wcat, bcat = torch.cat([wq, wk, wv], dim=0), torch.cat([bq, bk, bv])
    proj = projected_tables.get(table, params, wcat, bcat, code)
    n, L = ids.shape
    N, d_model = wq.shape
    mask_c = mask.contiguous().float() if mask is not None else None
    y = torch.empty(n, L, N, dtype=torch_dtype(code), device=ids.device)
    tp = table_cache.get(table, code)
    d = _lib.MhsaDesc(n=n, L=L, d_model=d_model, heads=heads, d_head=N // heads, dtype=code, src_kind=NR_SRC_GATHER, x=ptr(tp),
                      ldx=tp.shape[1], ids=ptr(ids), p_in=0.0, seed_in=0, p_out=p_out, seed_out=draw_seed() if p_out > 0 else 0,
                      mask=ptr(mask_c), w_qkv=ptr(proj), ldw=tp.shape[1], b_qkv=ptr(proj), proj_table=ptr(proj))
    check(_lib.lib().nr_mhsa_fwd(C.byref(d), None, ptr(y), _stream()), "nr_mhsa_fwd")
    return y


def needed_flags(needed):
    """[n] int32 flags (1 = the caller uses this sequence's output) from a bool / float / int tensor, or None."""
    if needed is None:
        return None
    if getattr(needed, "_nr_flags01", False):          # stack_rows made them: int32 0 / 1, contiguous
        return needed
    return (needed.reshape(-1) != 0).to(torch.int32).contiguous()


def pool_contracts_slabs(n: int, L: int, N: int, q: int, code: int) -> bool:
    """True when the additive-pooling backward of this shape reads only the x rows near sequences with a gradient (its weight
    gradient contracts live 32-row slabs): the producer of x may then leave far unneeded rows unwritten (mhsa far_unwritten)."""
    d = _lib.PoolDesc(n=int(n), L=int(L), N=int(N), q=int(q), dtype=int(code))
    return bool(_lib.lib().nr_pool_contracts_slabs(C.byref(d)))


def mhsa(x, wq, bq, wk, bk, wv, bv, heads: int, code: int, mask=None, ids=None, table=None, p_in=0.0, p_out=0.0, flat=None,
         needed=None, far_unwritten=False):
    """Dense: x [n, L, d_model] (compute dtype).  Gather: ids int32 [n, L] + fp32 `table` parameter.
    flat: {"w": [3N, d_model], "b": [3N], "gw", "gb"} views of a flat parameter / gradient bucket (parallel.FlatBucket).
    needed: optional [n] int32 flags (needed_flags): sequences with flag 0 reach the loss through a factor 0 only; their
    output rows are exact zeros and are not computed (their gradient is zero, so nothing flows back either).
    far_unwritten: with `needed`, the output rows of unneeded sequences farther than 32 / L + 2 sequences from every needed
    one may stay UNWRITTEN (not even zeros) -- only for a caller whose sole consumer is additive_pool with the same flags on
    a shape for which pool_contracts_slabs() holds."""
    cfg = dict(code=code, heads=heads, p_in=float(p_in), p_out=float(p_out),
               seed_in=draw_seed() if p_in > 0 else 0, seed_out=draw_seed() if p_out > 0 else 0, needed=needed,
               far_unwritten=bool(far_unwritten))
    if flat is not None:
        cfg["flat"] = flat
    if ids is not None:
        if ids.dtype != torch.int32:
            ids = ids.to(torch.int32)
        if CHECK_INDICES:
            check_ids(ids, table.shape[0], "token id")
        N = wq.shape[0]
        if (not torch.is_grad_enabled() and p_in == 0 and code == NR_BF16 and ids.dim() == 2 and ids.shape[1] <= 64
                and (N // heads) % 4 == 0 and N % 8 == 0 and USE_PROJECTED_TABLE):
            return _mhsa_projected(table, (wq, bq, wk, bk, wv, bv), flat, ids.contiguous(), mask, heads, code, float(p_out))
        cfg["table_packed"] = table_cache.get(table, code)
        cfg["table_shape"] = tuple(table.shape)
        if table.requires_grad and torch.is_grad_enabled():
            cfg["table_grad"] = grad_target(table)
            cfg["table_grad_ready"] = getattr(table, "_nr_grad_ready", None)     # parallel.FlatBucket, world > 1
        y = MHSAFunction.apply(table, wq, bq, wk, bk, wv, bv, ids, mask, cfg)
        # the backward of this call will ignore the dy rows of sequences flagged "zero gradient" (compact row storage): a pooling
        # that consumes y alone may leave them unwritten (additive_pool lazy_dx)
        y._nr_takes_lazy_dy = bool(cfg.get("compact_rows"))
        return y
    ch = chunk(code)
    if x.shape[-1] % ch:
        # d_model not a multiple of the 16-byte operand chunk: zero-pad the feature axis of x and of the weights (torch
        # plumbing on a rare shape; autograd slices the gradients back)
        pad = ch - x.shape[-1] % ch
        cfg.pop("flat", None)               # the padded weights are temporaries: gradients go through autograd
        x = torch.nn.functional.pad(x, (0, pad))
        wq, wk, wv = (torch.nn.functional.pad(w, (0, pad)) for w in (wq, wk, wv))
    return MHSAFunction.apply(x, wq, bq, wk, bk, wv, bv, None, mask, cfg)


# ------------------------------------------------------------------------------------------ attention core alone
class SDPAFunction(Function):
    """ScaledDotProductAttention.forward, src/model/model_utils.py:39-55, on packed projections [n*L, 3N]."""

    @staticmethod
    def forward(ctx, qkv, mask, n, L, heads, d_head, code):
        _need_gpu(qkv, mask)
        N = heads * d_head
        mask_c = mask.contiguous().float() if mask is not None else None
        y = torch.empty(n * L, N, dtype=torch_dtype(code), device=qkv.device)
        check(_lib.lib().nr_sdpa_fwd(ptr(qkv), ptr(mask_c), ptr(y), n, L, heads, d_head, code, 0.0, 0, _stream()), "nr_sdpa_fwd")
        ctx.dims = (n, L, heads, d_head, code)
        ctx.save_for_backward(qkv, mask_c)
        return y

    @staticmethod
    def backward(ctx, dy):
        _enter_backward()
        qkv, mask_c = ctx.saved_tensors
        n, L, heads, d_head, code = ctx.dims
        dy = dy.contiguous()
        if dy.dtype != torch_dtype(code):
            dy = dy.to(torch_dtype(code))
        dqkv = torch.empty_like(qkv)
        check(_lib.lib().nr_sdpa_bwd(ptr(qkv), ptr(mask_c), ptr(dy), ptr(dqkv), n, L, heads, d_head, code, 0.0, 0, _stream()),
              "nr_sdpa_bwd")
        return dqkv, None, None, None, None, None, None


def sdpa(Q, K, V, code: int, mask=None):
    """Q, K, V: [n, h, L, d] (any strides, e.g. the transposed views of src/model/model_utils.py:89-91); mask: [n, L], or the
    reference's per-head expansion [n, h, L] of it (:86-87) -> [n, h, L, d].  The three tensors are packed into the
    token-major [n*L, 3*h*d] layout of the kernels (torch plumbing), the attention itself runs in libnrhip."""
    _need_gpu(Q, K, V, mask)
    n, h, L, d = Q.shape
    if K.shape != Q.shape or V.shape != Q.shape:
        raise RuntimeError(f"sdpa: Q/K/V shapes differ: {tuple(Q.shape)}, {tuple(K.shape)}, {tuple(V.shape)} (d_k must equal d_v)")
    if mask is not None and mask.dim() == 3:
        if mask.stride(1) != 0 and not bool((mask == mask[:, :1]).all()):
            raise NotImplementedError("sdpa: per-head attention masks do not occur in the reference (the mask is the [n, L] key "
                                      "mask repeated over heads, src/model/model_utils.py:86-87) and are not supported")
        mask = mask[:, 0]
    td = torch_dtype(code)
    # [n, h, L, d] x 3 -> [n, L, 3, h, d] -> [n*L, 3*h*d]
    qkv = torch.stack([t.to(td).permute(0, 2, 1, 3) for t in (Q, K, V)], dim=2).reshape(n * L, 3 * h * d)
    y = SDPAFunction.apply(qkv, mask, n, L, h, d, code)
    return y.view(n, L, h, d).permute(0, 2, 1, 3)


# ------------------------------------------------------------------------------------------ pooling
class PoolFunction(Function):
    """K5: AttentionPooling.forward, src/model/model_utils.py:13-31."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, mask, code, needed=None, lazy_dx=False):
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
                          b1=ptr(b1_c), w2=ptr(w2_c), b2=ptr(b2_c), seq_needed=ptr(needed))
        check(_lib.lib().nr_additive_pool_fwd(C.byref(d), ptr(e), ptr(alpha), ptr(out), N, _stream()), "nr_additive_pool_fwd")
        ctx.code, ctx.dims = code, (n, L, N, q)
        ctx.needed = needed
        ctx.lazy_dx = bool(lazy_dx)
        # (Function.forward runs with grad mode off: no torch.is_grad_enabled() test here -- it would always say no)
        ctx.targets = tuple(grad_target(p) for p in (w1, b1, w2, b2))
        ctx.save_for_backward(x, mask_c, w1_p, b1_c, w2_c, b2_c, e, alpha, w1)
        return out

    @staticmethod
    def backward(ctx, g):
        seq = _enter_backward()
        x, mask_c, w1_p, b1_c, w2_c, b2_c, e, alpha, w1 = ctx.saved_tensors
        n, L, N, q = ctx.dims
        code, dev = ctx.code, g.device
        g = g.contiguous().float()
        w1_t = pack(w1, code, transpose=True) if ctx.needs_input_grad[0] else None    # [N, q]
        dpre = torch.empty_like(e)
        direct = all(t is not None for t in ctx.targets)
        if direct:
            dw1, db1, dw2, db2 = ctx.targets                      # views of the flat gradient bucket, accumulated in place
        else:
            # one zero fill for the four accumulated gradients (views of a flat buffer; q*N and q are multiples of 4)
            flat = torch.zeros(q * N + 2 * q + 4, dtype=torch.float32, device=dev)
            dw1, db1, dw2, db2 = flat[:q * N].view(q, N), flat[q * N:q * N + q], flat[q * N + q:q * N + 2 * q], flat[q * N + 2 * q:q * N + 2 * q + 1]
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        d = _lib.PoolDesc(n=n, L=L, N=N, q=q, dtype=code, x=ptr(x), mask=ptr(mask_c), w1=ptr(w1_p), ldw1=w1_p.shape[1],
                          b1=ptr(b1_c), w2=ptr(w2_c), b2=ptr(b2_c), seq_needed=ptr(ctx.needed), dx_far_unwritten=int(ctx.lazy_dx))
        partial = _ws(_lib.lib().nr_pool_workspace_bytes(C.byref(d)), dev)      # per-workgroup partial rows + slab scratch
        d.partial_bytes = partial.numel() * 4
        check(_lib.lib().nr_additive_pool_bwd(C.byref(d), ptr(e), ptr(alpha), ptr(g), N, ptr(w1_t),
                                              w1_t.shape[1] if w1_t is not None else 0, ptr(dpre), ptr(partial), ptr(dw1),
                                              ptr(db1), ptr(dw2), ptr(db2), ptr(dx), _stream()), "nr_additive_pool_bwd")
        flags_ptr = _lib.lib().nr_pool_seq_flags(C.byref(d), ptr(partial))
        _offer_flags(seq, dx, n, flags_ptr, partial, lazy=ctx.lazy_dx and bool(flags_ptr))
        if direct:
            return dx, None, None, None, None, None, None, None, None
        return dx, dw1, db1, dw2.view(1, q), db2, None, None, None, None


def additive_pool(x, w1, b1, w2, b2, code: int, mask=None, needed=None, lazy_dx=False):
    """needed: optional [n] int32 flags (needed_flags): flag 0 = the pooled vector is not used: zeros, nothing computed.
    lazy_dx: the gradient wrt x may keep UNWRITTEN rows for sequences with a zero pooled gradient that lie far from every
    sequence with one -- only when x is the output of ops.mhsa (gather source) whose `_nr_takes_lazy_dy` is set and nothing
    else consumes x; the MHSA backward then never reads those rows (and raises if the flags did not reach it)."""
    return PoolFunction.apply(x, w1, b1, w2, b2, mask, code, needed, bool(lazy_dx) and x.requires_grad)


# ------------------------------------------------------------------------------------------ pad blend / cast
# Row split whose backward is not a copy.  `torch.split` of the encoder output [candidates ; history] gets its gradient as a
# concatenation of the two consumers' gradients (45 MB written and read again per step at B = 512).  split_rows allocates that
# gradient buffer up front and registers its two halves under the addresses of the two outputs; a libnrhip backward that
# produces the gradient of exactly such a tensor (the pad-doc blend of the history rows, the scorer for the candidates) writes
# into the registered half, and the split's backward recognises the two halves and returns the buffer as it stands.  Anything
# else -- another consumer, an accumulation, a copy in between -- arrives in some other tensor and is concatenated as before.
_grad_out = {}


def _take_grad_out(x):
    if not _grad_out or not x.is_cuda or x.dtype != torch.float32 or not x.is_contiguous():
        return None
    return _grad_out.pop((x.data_ptr(), x.numel()), None)


class SplitRowsFunction(Function):
    @staticmethod
    def forward(ctx, x, n_a):
        ctx.n_a, ctx.arena = int(n_a), None
        a, b = x[:n_a], x[n_a:]
        _grad_out.clear()
        if (ctx.needs_input_grad[0] and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 2
                and 0 < n_a < x.shape[0]):
            ctx.arena = torch.empty_like(x)
            _grad_out[(a.data_ptr(), a.numel())] = ctx.arena[:n_a]
            _grad_out[(b.data_ptr(), b.numel())] = ctx.arena[n_a:]
        return a, b

    @staticmethod
    def backward(ctx, ga, gb):
        arena, n_a = ctx.arena, ctx.n_a
        if (arena is not None and ga is not None and gb is not None and ga.dtype == gb.dtype == torch.float32 and ga.is_contiguous()
                and gb.is_contiguous() and ga.data_ptr() == arena.data_ptr() and gb.data_ptr() == arena[n_a:].data_ptr()
                and ga.numel() == arena[:n_a].numel() and gb.numel() == arena[n_a:].numel()):
            return arena, None
        return torch.cat([ga, gb], dim=0), None


def split_rows(x, n_a: int):
    """x[:n_a], x[n_a:] -- see above."""
    return SplitRowsFunction.apply(x, n_a)


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
        ctx.pad_target = grad_target(pad)
        ctx.dx_out = _take_grad_out(x)                   # split_rows: write dx where the producer's gradient is assembled
        ctx.save_for_backward(mask_c)
        return out

    @staticmethod
    def backward(ctx, dout):
        _enter_backward()
        (mask_c,) = ctx.saved_tensors
        n, L, N = ctx.dims
        dout = dout.contiguous()
        if dout.dtype != torch_dtype(ctx.code):
            dout = dout.to(torch_dtype(ctx.code))
        dx = ctx.dx_out.view(n, L, N) if ctx.dx_out is not None else torch.empty(n, L, N, dtype=torch.float32, device=dout.device)
        direct = ctx.pad_target is not None and mask_c is not None
        dpad = ctx.pad_target if direct else (torch.zeros(N, dtype=torch.float32, device=dout.device) if mask_c is not None else None)
        check(_lib.lib().nr_pad_blend_bwd(ptr(dout), ptr(mask_c), ptr(dx), ptr(dpad), n, L, N, ctx.code, _stream()), "nr_pad_blend_bwd")
        if direct:
            return dx, None, None, None
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
        ctx.dcand_out = _take_grad_out(cand)
        ctx.set_materialize_grads(False)                 # an unused `score` output then costs no zero fill in the backward
        return loss, score

    @staticmethod
    def backward(ctx, gloss, gscore):
        _enter_backward()
        cand, user, label, score = ctx.saved_tensors
        B, Cn, N = cand.shape
        gl = gloss.contiguous().float() if gloss is not None else None
        gs = gscore.contiguous().float() if gscore is not None else None
        dcand = ctx.dcand_out.view_as(cand) if ctx.dcand_out is not None else torch.empty_like(cand)
        duser = torch.empty_like(user)
        check(_lib.lib().nr_score_ce_bwd(ptr(cand), N, ptr(user), ptr(label), ptr(score), ptr(gl), ptr(gs), ptr(dcand), N,
                                         ptr(duser), B, Cn, N, _stream()), "nr_score_ce_bwd")
        return dcand, duser, None


def score_ce(cand, user, label):
    if CHECK_INDICES:
        check_labels(label.contiguous().to(torch.int64), cand.shape[1])
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
        # bf16: the token rows (gather + dropout) are stored once, with a zero row between titles: the im2col row of a token is
        # 3 * Dp contiguous elements of that buffer, which feeds the LDS-DMA GEMMs of both passes (no 3x im2col copy)
        x_rows = torch.empty(n * (T + 1) + 1, Dp, dtype=torch.bfloat16, device=dev) if code == _lib.NR_BF16 and n > 0 else None
        d = _lib.ConvDesc(n=n, T=T, D=D, Dp=Dp, N=N, dtype=code, table=ptr(table_p), ids=ids.data_ptr(), ids_stride=stride,
                          p_in=cfg["p_in"], seed_in=cfg["seed_in"], w_pack=ptr(w_p), bias=ptr(b_c), x_rows=ptr(x_rows),
                          ld_rows=Dp, seq_needed=ptr(cfg.get("needed")))
        check(_lib.lib().nr_conv1d_k3_fwd(C.byref(d), ptr(y), _stream()), "nr_conv1d_k3_fwd")
        ctx.cfg, ctx.dims = cfg, (n, T, D, Dp, N, stride)
        ctx.ids = ids                                   # keeps the (possibly strided) id view alive
        ctx.x_rows = x_rows if any(ctx.needs_input_grad[:2]) else None
        ctx.targets = (grad_target(w), grad_target(b))
        ctx.save_for_backward(table_p, w_p, b_c)
        return y

    @staticmethod
    def backward(ctx, dy):
        seq = _enter_backward()
        table_p, w_p, b_c = ctx.saved_tensors
        n, T, D, Dp, N, stride = ctx.dims
        cfg, code, dev = ctx.cfg, ctx.cfg["code"], dy.device
        dy = dy.contiguous()
        if dy.dtype != torch_dtype(code):
            dy = dy.to(torch_dtype(code))
        seq_nz, seq_nz_owner, _lazy = _take_flags(seq, dy, n)
        if _lazy:
            raise RuntimeError("conv1d_k3 backward cannot take a gradient with unwritten rows (additive_pool lazy_dx=True)")
        dwp = torch.zeros(N, 3 * Dp, dtype=torch.float32, device=dev)
        direct = all(t is not None for t in ctx.targets)
        db = ctx.targets[1] if direct else torch.zeros(N, dtype=torch.float32, device=dev)
        d = _lib.ConvDesc(n=n, T=T, D=D, Dp=Dp, N=N, dtype=code, table=ptr(table_p), ids=ctx.ids.data_ptr(), ids_stride=stride,
                          p_in=cfg["p_in"], seed_in=cfg["seed_in"], w_pack=ptr(w_p), bias=ptr(b_c), x_rows=ptr(ctx.x_rows),
                          ld_rows=Dp, seq_nz=seq_nz, seq_needed=ptr(cfg.get("needed")))
        bwd_ws = _ws(_lib.lib().nr_conv_workspace_bytes(C.byref(d)), dev) if ctx.x_rows is not None else None
        d.bwd_ws, d.bwd_ws_bytes = ptr(bwd_ws), (bwd_ws.numel() * 4 if bwd_ws is not None else 0)
        check(_lib.lib().nr_conv1d_k3_bwd(C.byref(d), ptr(dy), ptr(dwp), ptr(db), _stream()), "nr_conv1d_k3_bwd")
        dw = ctx.targets[0] if direct else torch.empty(N, D, 3, dtype=torch.float32, device=dev)
        check(_lib.lib().nr_unpack_conv_dw(ptr(dwp), N, D, Dp, ptr(dw), int(direct), _stream()), "nr_unpack_conv_dw")
        if direct:
            return None, None, None, None
        return dw, db, None, None


def conv1d_k3_gather(table, w, b, ids, T: int, D: int, code: int, p_in=0.0, needed=None):
    """ids: int32 view [n] (any stride) of news ids; table: fp32 [V, T*D] (frozen on this path).
    needed: optional [n] int32 flags (needed_flags); output rows of unneeded titles may stay unwritten -- pool them with the
    same flags."""
    if table.requires_grad:
        raise RuntimeError("NAML title-embedding table must be frozen (freeze_embedding=True, as src/demo.sh:12): "
                           "its [V, T*D] dense gradient is out of scope on this path")
    if CHECK_INDICES:
        check_ids(ids, table.shape[0], "news id")
    # model.NAML.TitleTable holds the packed operand itself (uploaded block by block); a plain fp32 parameter is packed once
    packed = table.packed(code) if hasattr(table, "packed") else table_cache.get(table, code, row_cols=D)
    cfg = dict(code=code, T=T, D=D, p_in=float(p_in), seed_in=draw_seed() if p_in > 0 else 0, table_packed=packed, needed=needed)
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
        ctx.targets = tuple(grad_target(p) for p in (emb, w, b))
        ctx.save_for_backward(emb_p, w_p, b_c, w)
        return out

    @staticmethod
    def backward(ctx, dout):
        _enter_backward()
        emb_p, w_p, b_c, w = ctx.saved_tensors
        M, K, N, stride, emb_shape = ctx.dims
        code, dev = ctx.code, dout.device
        dout = dout.contiguous().float()
        Nc = round_up(N, chunk(code))
        need_t = ctx.needs_input_grad[0]
        direct = ctx.targets[1] is not None and ctx.targets[2] is not None and (ctx.targets[0] is not None or not need_t)
        if direct:
            dtable, dw, db = (ctx.targets[0] if need_t else None), ctx.targets[1], ctx.targets[2]
        else:
            dw = torch.zeros(N, K, dtype=torch.float32, device=dev)
            db = torch.zeros(N, dtype=torch.float32, device=dev)
            dtable = torch.zeros(emb_shape, dtype=torch.float32, device=dev) if need_t else None
        w_t = pack(w, code, transpose=True, ld=Nc) if need_t else None               # [K, Nc]
        d = _lib.LinearDesc(M=M, K=K, N=N, dtype=code, src_kind=NR_SRC_GATHER, x=ptr(emb_p), ldx=emb_p.shape[1],
                            ids=ctx.ids.data_ptr(), ids_stride=stride, w=ptr(w_p), ldw=w_p.shape[1], bias=ptr(b_c),
                            w_t=ptr(w_t), ldwt=Nc if need_t else 0, table_rows=emb_shape[0])
        ws = _ws(_lib.lib().nr_linear_workspace_bytes(C.byref(d)), dev)
        d.dout_ws_bytes = ws.numel() * 4
        check(_lib.lib().nr_linear_bwd(C.byref(d), ptr(dout), N, ptr(ws), ptr(dw), ptr(db), ptr(dtable), _stream()), "nr_linear_bwd")
        if direct:
            return None, None, None, None, None
        return dtable, dw, db, None, None


def gather_linear(emb, w, b, ids, code: int):
    if CHECK_INDICES:
        check_ids(ids, emb.shape[0], "category id")
    return GatherLinearFunction.apply(emb, w, b, ids, code)


# ------------------------------------------------------------------------------------------ plain row gather (eval path)
def embed_gather(table: torch.Tensor, ids: torch.Tensor, code: int = NR_F32) -> torch.Tensor:
    """out[i] = table[ids[i]], no gradient: the eval-time news-vector lookup of src/dataset.py:68,72.  `code`: dtype of the
    result (fp32, or bf16 = gather + the cast the user encoder would do next, in one pass)."""
    _need_gpu(table, ids)
    table = table.detach().float().contiguous()
    ids = ids.to(torch.int32).contiguous()
    if CHECK_INDICES:
        check_ids(ids, table.shape[0], "news index")
    cols = table.shape[1]
    out = torch.empty(*ids.shape, cols, dtype=torch_dtype(code), device=table.device)
    check(_lib.lib().nr_gather_cast_fwd(ptr(table), cols, ptr(ids), ids.numel(), cols, ptr(out), cols, code, _stream()),
          "nr_gather_cast_fwd")
    return out


def score_eval(news_vecs, cand_ids, imp_of, user_vecs) -> torch.Tensor:
    """score[i] = <news_vecs[cand_ids[i]], user_vecs[imp_of[i]]> — the per-impression np.dot of src/main.py:253."""
    _need_gpu(news_vecs, cand_ids, imp_of, user_vecs)
    news_vecs, user_vecs = news_vecs.detach().float().contiguous(), user_vecs.detach().float().contiguous()
    cand_ids, imp_of = cand_ids.to(torch.int32).contiguous(), imp_of.to(torch.int32).contiguous()
    if CHECK_INDICES:
        check_ids(cand_ids, news_vecs.shape[0], "candidate index")
        check_ids(imp_of, user_vecs.shape[0], "impression index")
    N = news_vecs.shape[1]
    out = torch.empty(cand_ids.numel(), dtype=torch.float32, device=news_vecs.device)
    check(_lib.lib().nr_score_eval(ptr(news_vecs), N, ptr(cand_ids), ptr(imp_of), ptr(user_vecs), N, ptr(out),
                                   cand_ids.numel(), N, _stream()), "nr_score_eval")
    return out


def stack_rows(a, b, mask_b=None, flags=True):
    """[a ; b] of two int32 id matrices with the same row width (candidate titles, then history titles) and, if asked, the int32
    "needed" flags [1 .. 1 ; mask_b != 0] of the stacked rows -- one launch (nr_stack_rows)."""
    _need_gpu(a, b, mask_b)
    a2, b2 = a.reshape(-1, a.shape[-1]), b.reshape(-1, b.shape[-1])
    if a2.shape[1] != b2.shape[1]:
        raise RuntimeError(f"stack_rows: row widths differ ({a2.shape[1]} vs {b2.shape[1]})")
    a2, b2 = (t if t.dtype == torch.int32 and t.is_contiguous() else t.to(torch.int32).contiguous() for t in (a2, b2))
    m = None
    if mask_b is not None:
        m = mask_b.reshape(-1)
        if m.dtype != torch.float32 or not m.is_contiguous():
            m = m.float().contiguous()
        if m.numel() != b2.shape[0]:
            raise RuntimeError(f"stack_rows: {m.numel()} mask entries for {b2.shape[0]} rows")
    out = torch.empty(a2.shape[0] + b2.shape[0], a2.shape[1], dtype=torch.int32, device=a2.device)
    fl = torch.empty(out.shape[0], dtype=torch.int32, device=out.device) if flags else None
    check(_lib.lib().nr_stack_rows(ptr(a2), a2.shape[0], ptr(b2), b2.shape[0], a2.shape[1], ptr(m), ptr(out), ptr(fl), _stream()), "nr_stack_rows")
    if fl is not None:
        fl._nr_flags01 = True
    return out, fl


def assemble_batch(news_combined, hist_idx, pos_idx, neg_idx, label):
    """Row f1: history [B, H, F] = news_combined[hist_idx]; candidate [B, 1+K, F] = news_combined[neg[:label] + [pos] +
    neg[label:]] (src/dataset.py:40-48 + the DataLoader collate), gathered on the device from int32 index arrays."""
    _need_gpu(news_combined, hist_idx, pos_idx, neg_idx, label)
    comb = news_combined if news_combined.dim() == 2 else news_combined.reshape(news_combined.shape[0], -1)
    if comb.dtype != torch.int32 or not comb.is_contiguous():
        comb = comb.to(torch.int32).contiguous()
    hist_idx, pos_idx, neg_idx = (t.to(torch.int32).contiguous() for t in (hist_idx, pos_idx, neg_idx))
    label = label.to(torch.int64).contiguous()
    B, H = hist_idx.shape
    K, F = neg_idx.shape[1], comb.shape[1]
    dev = comb.device
    history = torch.empty(B, H, F, dtype=torch.int32, device=dev)
    candidate = torch.empty(B, 1 + K, F, dtype=torch.int32, device=dev)
    bad = torch.zeros(1, dtype=torch.int32, device=dev) if CHECK_INDICES else None
    check(_lib.lib().nr_assemble_batch(ptr(comb), comb.shape[0], F, ptr(hist_idx), ptr(pos_idx), ptr(neg_idx), ptr(label), B, H, K,
                                       ptr(history), ptr(candidate), ptr(bad), _stream()), "nr_assemble_batch")
    if bad is not None and int(bad.item()):
        raise IndexError(f"assemble_batch: {int(bad.item())} news indices / labels out of range")
    return history, candidate


def eval_metrics(score, label, offsets, max_cand=None, return_per_impression=False):
    """Row f2: per-impression AUC / MRR / nDCG@5 / nDCG@10 (src/metrics.py, src/main.py:249-263) over CSR candidate lists.
    Returns a DEVICE fp64 tensor [5] = [scored impressions, sum AUC, sum MRR, sum nDCG@5, sum nDCG@10]
    (+ the [n_imp, 4] per-impression table, AUC = -1 for skipped impressions, if asked)."""
    _need_gpu(score, label, offsets)
    score = score.detach().float().contiguous()
    label, offsets = label.to(torch.int32).contiguous(), offsets.to(torch.int32).contiguous()
    n_imp = offsets.numel() - 1
    if max_cand is None:
        max_cand = int((offsets[1:] - offsets[:-1]).max().item()) if n_imp > 0 else 0
    nbytes = int(_lib.lib().nr_eval_metrics_workspace_bytes(n_imp))
    per_imp = torch.empty(max(nbytes // 8, 4), dtype=torch.float64, device=score.device)
    sums = torch.empty(5, dtype=torch.float64, device=score.device)
    check(_lib.lib().nr_eval_metrics(ptr(score), ptr(label), ptr(offsets), n_imp, int(max_cand), ptr(per_imp), per_imp.numel() * 8,
                                     ptr(sums), _stream()), "nr_eval_metrics")
    if return_per_impression:
        return sums, per_imp[: n_imp * 4].view(n_imp, 4)
    return sums


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
