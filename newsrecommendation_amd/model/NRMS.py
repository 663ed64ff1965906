"""Drop-in for the reference's src/model/NRMS.py (NewsEncoder / UserEncoder / Model)."""
import torch
from torch import nn

from .. import ops
from .model_utils import AttentionPooling, MultiHeadSelfAttention


def _cd(args):
    return getattr(args, "compute_dtype", "fp32")


class NewsEncoder(nn.Module):
    """src/model/NRMS.py:8-36: embed -> dropout -> MHSA -> dropout -> additive pooling."""

    def __init__(self, args, embedding_matrix):
        super().__init__()
        self.embedding_matrix = embedding_matrix
        self.drop_rate = args.drop_rate
        self.dim_per_head = args.news_dim // args.num_attention_heads
        assert args.news_dim == args.num_attention_heads * self.dim_per_head
        self.multi_head_self_attn = MultiHeadSelfAttention(args.word_embedding_dim, args.num_attention_heads,
                                                           self.dim_per_head, self.dim_per_head, compute_dtype=_cd(args))
        self.attn = AttentionPooling(args.news_dim, args.news_query_vector_dim, compute_dtype=_cd(args))

    def forward(self, x, mask=None, needed=None):
        """x: [n, word_num] token ids; mask: [n, word_num] or None -> [n, news_dim] fp32.
        needed (beyond the reference): optional [n] flags; 0 = the caller multiplies this title's vector by zero (a masked
        history slot): it is returned as zeros without being encoded."""
        p = self.drop_rate if self.training else 0.0
        needed = ops.needed_flags(needed)
        # y feeds the pooling below and nothing else: where its backward contracts live slabs only, the context rows of unneeded
        # titles far from every needed one are never read and need not even be zero-filled
        far = needed is not None and ops.pool_contracts_slabs(x.shape[0], x.shape[1], self.attn.att_fc1.in_features,
                                                              self.attn.att_fc1.out_features, ops.dtype_code(self.attn.compute_dtype))
        y = self.multi_head_self_attn.forward_gather(x, self.embedding_matrix.weight, mask=mask, p_in=p, p_out=p, needed=needed,
                                                     far_unwritten=far)
        # ... and in the other direction: the MHSA backward of a compactly stored batch ignores the dy rows of zero-gradient
        # sequences, so the pooling need not zero-fill them either
        return self.attn(y, mask, needed=needed, lazy_dx=far and getattr(y, "_nr_takes_lazy_dy", False))


class UserEncoder(nn.Module):
    """src/model/NRMS.py:39-63."""

    def __init__(self, args):
        super().__init__()
        self.args = args
        self.dim_per_head = args.news_dim // args.num_attention_heads
        assert args.news_dim == args.num_attention_heads * self.dim_per_head
        self.multi_head_self_attn = MultiHeadSelfAttention(args.news_dim, args.num_attention_heads, self.dim_per_head,
                                                           self.dim_per_head, compute_dtype=_cd(args))
        self.attn = AttentionPooling(args.news_dim, args.user_query_vector_dim, compute_dtype=_cd(args))
        self.pad_doc = nn.Parameter(torch.empty(1, args.news_dim).uniform_(-1, 1)).type(torch.FloatTensor)

    def forward(self, news_vecs, log_mask=None):
        """news_vecs: [B, H, news_dim]; log_mask: [B, H] -> [B, news_dim] fp32."""
        code = ops.dtype_code(_cd(self.args))
        if self.args.user_log_mask:
            x = news_vecs if news_vecs.dtype == ops.torch_dtype(code) else ops.to_compute(news_vecs.float(), code)
            y = self.multi_head_self_attn(x, mask=log_mask)
            return self.attn(y, log_mask)
        x = ops.pad_blend(news_vecs, log_mask, self.pad_doc, code)
        y = self.multi_head_self_attn(x)
        return self.attn(y)


    @torch.no_grad()
    def forward_indexed(self, news_table, history_idx, log_mask):
        """Eval-time form of `forward(news_table[history_idx], log_mask)` (src/main.py:247 with src/dataset.py:68): the history
        is given as INDICES into the [N+1, news_dim] news-vector table.  With the masked user encoder (src/demo.sh:26) Q|K|V of
        a slot depend on its news index only, so the table is projected once and the attention gathers projected rows (the
        title-level shortcut of ops._ProjectedTables one level up) -- no [B, H, news_dim] gather, no per-impression QKV GEMM."""
        if not self.args.user_log_mask or ops.dtype_code(_cd(self.args)) != ops.NR_BF16 or history_idx.shape[1] > 64:
            code = ops.dtype_code(_cd(self.args)) if self.args.user_log_mask else ops.NR_F32
            return self.forward(ops.embed_gather(news_table, history_idx, code), log_mask)
        y = self.multi_head_self_attn.forward_gather(history_idx, news_table, mask=log_mask)
        return self.attn(y, log_mask)


class Model(torch.nn.Module):
    """src/model/NRMS.py:66-95.  Also accepts the positional (args, emb, n_cat, n_subcat) call of
    src/main.py:64, which the reference class itself rejects."""

    def __init__(self, args, embedding_matrix, *unused, **kwargs):
        super().__init__()
        self.args = args
        pretrained_word_embedding = torch.from_numpy(embedding_matrix).float()
        word_embedding = nn.Embedding.from_pretrained(pretrained_word_embedding, freeze=args.freeze_embedding, padding_idx=0)
        self.news_encoder = NewsEncoder(args, word_embedding)
        self.user_encoder = UserEncoder(args)
        self.loss_fn = nn.CrossEntropyLoss()

    def forward(self, history, history_mask, candidate, label):
        """history [B, H, T] int32; history_mask [B, H] fp32; candidate [B, 1+K, T] int32; label [B] int64
        -> (loss, score [B, 1+K])."""
        a = self.args
        B = candidate.shape[0]
        C = 1 + a.npratio
        cand = candidate.reshape(-1, a.num_words_title)
        hist = history.reshape(-1, a.num_words_title)
        # one encoder pass over candidates + history (the reference makes two, src/model/NRMS.py:87,90)
        compact = getattr(a, "compact_history", False)
        if compact:
            # Opt-in, beyond the reference: a history slot with mask 0 reaches the loss only through `vec * 0` (pad_doc
            # blend, NRMS.py:59-60) or through attention / pooling weights that the mask zeroes (model_utils.py:28,51),
            # so its news vector and every gradient through it are exactly 0 -- the reference encodes those titles for
            # nothing.  Encode only the live slots and scatter them back.  Same loss / score / gradients in eval mode;
            # in training the dropout draws land on different rows (the counters follow the compacted order).
            live = (history_mask.reshape(-1) != 0).nonzero(as_tuple=False).squeeze(1)     # host sync: the count
            vecs = self.news_encoder(torch.cat([cand, hist.index_select(0, live)], dim=0))
        else:
            # History slots with mask 0 reach the loss through a factor 0 only (pad-doc blend NRMS.py:59-60, or masked out of
            # the user-level attention and pooling, model_utils.py:28,51): the encoder is told, and returns zeros for them
            # without computing them.  Same loss, scores and gradients; `args.encode_masked_slots=True` switches it off.
            if cand.is_cuda:
                ids, needed = ops.stack_rows(cand, hist, history_mask, flags=not getattr(a, "encode_masked_slots", False))
            else:
                ids, needed = torch.cat([cand, hist], dim=0), None
                if not getattr(a, "encode_masked_slots", False):
                    needed = torch.cat([history_mask.new_ones(B * C), history_mask.reshape(-1)])
            vecs = self.news_encoder(ids, needed=needed)
        # split, not two slices: its backward is one concatenation instead of two zero-filled full-size buffers and an add
        cand_flat, hist_flat = ops.split_rows(vecs, B * C) if not compact else vecs.split([B * C, vecs.shape[0] - B * C], dim=0)
        if compact:
            hist_flat = vecs.new_zeros(hist.shape[0], a.news_dim).index_copy(0, live, hist_flat)
        cand_vecs = cand_flat.reshape(B, C, a.news_dim)
        hist_vecs = hist_flat.reshape(B, a.user_log_length, a.news_dim)
        user_vec = self.user_encoder(hist_vecs, history_mask)
        loss, score = ops.score_ce(cand_vecs, user_vec, label)
        return loss, score
