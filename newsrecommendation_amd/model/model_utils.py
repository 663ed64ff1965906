"""Drop-in for the reference's src/model/model_utils.py (op library, L1 of SURVEY.md §1).

Same class names, constructor signatures, parameter names (state_dict keys) and forward
signatures; the arithmetic runs in libnrhip (HIP, gfx950) through newsrecommendation_amd.ops.
The compute dtype ('fp32' exact-MFMA or 'bf16') is a module attribute `compute_dtype`
set from `args.compute_dtype` by the model constructors (default 'fp32').
"""
import torch
from torch import nn

from .. import ops


class AttentionPooling(nn.Module):
    """src/model/model_utils.py:7-31."""

    def __init__(self, emb_size, hidden_size, compute_dtype="fp32"):
        super().__init__()
        self.att_fc1 = nn.Linear(emb_size, hidden_size)
        self.att_fc2 = nn.Linear(hidden_size, 1)
        self.compute_dtype = compute_dtype

    def forward(self, x, attn_mask=None, needed=None, lazy_dx=False):
        """x: [batch, L, emb]; attn_mask: [batch, L] -> [batch, emb] (fp32).
        needed (beyond the reference): optional [batch] int32 flags, 0 = the caller multiplies this row's vector by zero.
        lazy_dx: see ops.additive_pool (only for an x that comes straight from ops.mhsa and feeds nothing else)."""
        code = ops.dtype_code(self.compute_dtype)
        if x.dtype != ops.torch_dtype(code):
            x = ops.to_compute(x.float(), code)
            lazy_dx = False
        return ops.additive_pool(x, self.att_fc1.weight, self.att_fc1.bias, self.att_fc2.weight, self.att_fc2.bias,
                                 code, mask=attn_mask, needed=needed, lazy_dx=lazy_dx)


class ScaledDotProductAttention(nn.Module):
    """src/model/model_utils.py:34-55.  MultiHeadSelfAttention runs projection + attention as one op and does not go
    through this module; called on its own it runs the same attention kernels on the given Q, K, V (nr_sdpa_fwd/bwd)."""

    def __init__(self, d_k, compute_dtype="fp32"):
        super().__init__()
        self.d_k = d_k
        self.compute_dtype = compute_dtype

    def forward(self, Q, K, V, attn_mask=None):
        """Q, K, V: [batch, n_heads, L, d_k]; attn_mask: [batch, n_heads, L] (the [batch, L] key mask repeated over
        heads, :86-87) or [batch, L] -> context [batch, n_heads, L, d_k] in the compute dtype."""
        if Q.shape[-1] != self.d_k:
            raise RuntimeError(f"ScaledDotProductAttention(d_k={self.d_k}) got head dim {Q.shape[-1]}")
        return ops.sdpa(Q, K, V, ops.dtype_code(self.compute_dtype), mask=attn_mask)


class MultiHeadSelfAttention(nn.Module):
    """src/model/model_utils.py:58-95 (no output projection; xavier-uniform weights, default bias)."""

    def __init__(self, d_model, n_heads, d_k, d_v, compute_dtype="fp32"):
        super().__init__()
        if d_k != d_v:
            raise ValueError("d_k must equal d_v (as at every reference call site)")
        self.d_model, self.n_heads, self.d_k, self.d_v = d_model, n_heads, d_k, d_v
        self.W_Q = nn.Linear(d_model, d_k * n_heads)
        self.W_K = nn.Linear(d_model, d_k * n_heads)
        self.W_V = nn.Linear(d_model, d_v * n_heads)
        self.scaled_dot_product_attn = ScaledDotProductAttention(self.d_k, compute_dtype=compute_dtype)
        self.compute_dtype = compute_dtype
        self._initialize_weights()

    def _initialize_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight, gain=1)

    def _params(self):
        return (self.W_Q.weight, self.W_Q.bias, self.W_K.weight, self.W_K.bias, self.W_V.weight, self.W_V.bias)

    def forward(self, Q, K=None, V=None, mask=None, p_out=0.0):
        """Q (= K = V): [batch, L, d_model]; mask: [batch, L] -> [batch, L, n_heads*d_v] (compute dtype)."""
        if (K is not None and K is not Q) or (V is not None and V is not Q):
            raise NotImplementedError("only self-attention (Q is K is V) exists in the reference and here")
        code = ops.dtype_code(self.compute_dtype)
        x = Q
        if x.dtype != ops.torch_dtype(code):
            x = ops.to_compute(x.float(), code)
        return ops.mhsa(x, *self._params(), heads=self.n_heads, code=code, mask=mask, p_out=p_out,
                        flat=getattr(self, "_nr_flat", None))

    def forward_gather(self, ids, table, mask=None, p_in=0.0, p_out=0.0, needed=None, far_unwritten=False):
        """Embedding lookup + dropout + MHSA + dropout in one op: ids int32 [batch, L] into `table` [V, d_model]."""
        code = ops.dtype_code(self.compute_dtype)
        return ops.mhsa(None, *self._params(), heads=self.n_heads, code=code, mask=mask, ids=ids, table=table,
                        p_in=p_in, p_out=p_out, flat=getattr(self, "_nr_flat", None), needed=needed, far_unwritten=far_unwritten)
