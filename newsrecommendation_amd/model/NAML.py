"""Drop-in for the reference's src/model/NAML.py (fork variant: per-news flattened title embedding)."""
import torch
from torch import nn

from .. import ops
from .model_utils import AttentionPooling


def _cd(args):
    return getattr(args, "compute_dtype", "fp32")


class NewsEncoder(nn.Module):
    """src/model/NAML.py:8-75: title row gather -> dropout -> Conv1d(k=3) -> pooling; category / subcategory
    Embedding -> Linear; 3-view additive pooling."""

    def __init__(self, args, embedding_matrix, num_category, num_subcategory):
        super().__init__()
        self.drop_rate = args.drop_rate
        self.num_words_title = args.num_words_title
        self.word_embedding_dim = args.word_embedding_dim
        self.use_category = args.use_category
        self.use_subcategory = args.use_subcategory
        self.compute_dtype = _cd(args)
        self.title_embeddings = embedding_matrix
        if args.use_category:
            self.category_emb = nn.Embedding(num_category + 1, args.category_emb_dim, padding_idx=0)
            self.category_dense = nn.Linear(args.category_emb_dim, args.news_dim)
        if args.use_subcategory:
            self.subcategory_emb = nn.Embedding(num_subcategory + 1, args.category_emb_dim, padding_idx=0)
            self.subcategory_dense = nn.Linear(args.category_emb_dim, args.news_dim)
        if args.use_category or args.use_subcategory:
            self.final_attn = AttentionPooling(args.news_dim, args.news_query_vector_dim, compute_dtype=_cd(args))
        self.cnn = nn.Conv1d(in_channels=args.word_embedding_dim, out_channels=args.news_dim, kernel_size=3, padding=1)
        self.attn = AttentionPooling(args.news_dim, args.news_query_vector_dim, compute_dtype=_cd(args))

    def forward(self, x, mask=None, needed=None):
        """x: [n, F] int32, columns = [news id, category id, subcategory id][:F] -> [n, news_dim] fp32.
        needed (beyond the reference): optional [n] flags; 0 = the caller multiplies this news vector by zero (a masked history
        slot): its title view is not encoded (zeros)."""
        code = ops.dtype_code(self.compute_dtype)
        if x.dtype != torch.int32:
            x = x.to(torch.int32)
        x = x.contiguous()
        p = self.drop_rate if self.training else 0.0
        needed = ops.needed_flags(needed)
        ctx = ops.conv1d_k3_gather(self.title_embeddings.weight, self.cnn.weight, self.cnn.bias, x[:, 0],
                                   self.num_words_title, self.word_embedding_dim, code, p_in=p, needed=needed)
        all_vecs = [self.attn(ctx, mask, needed=needed)]
        col = 1
        if self.use_category:
            all_vecs.append(ops.gather_linear(self.category_emb.weight, self.category_dense.weight,
                                              self.category_dense.bias, x[:, col], code))
            col += 1
        if self.use_subcategory:
            all_vecs.append(ops.gather_linear(self.subcategory_emb.weight, self.subcategory_dense.weight,
                                              self.subcategory_dense.bias, x[:, col], code))
        if len(all_vecs) == 1:
            return all_vecs[0]
        return self.final_attn(torch.stack(all_vecs, dim=1))


class UserEncoder(nn.Module):
    """src/model/NAML.py:78-97."""

    def __init__(self, args):
        super().__init__()
        self.args = args
        self.attn = AttentionPooling(args.news_dim, args.user_query_vector_dim, compute_dtype=_cd(args))
        self.pad_doc = nn.Parameter(torch.empty(1, args.news_dim).uniform_(-1, 1)).type(torch.FloatTensor)

    def forward(self, news_vecs, log_mask=None):
        code = ops.dtype_code(_cd(self.args))
        if self.args.user_log_mask:
            x = news_vecs if news_vecs.dtype == ops.torch_dtype(code) else ops.to_compute(news_vecs.float(), code)
            return self.attn(x, log_mask)
        return self.attn(ops.pad_blend(news_vecs, log_mask, self.pad_doc, code))


class Model(torch.nn.Module):
    """src/model/NAML.py:100-130."""

    def __init__(self, args, news_embeddings_weight, num_category, num_subcategory, **kwargs):
        super().__init__()
        self.args = args
        pretrained_embedding = torch.from_numpy(news_embeddings_weight).float()
        news_embedding = nn.Embedding.from_pretrained(pretrained_embedding, freeze=args.freeze_embedding, padding_idx=0)
        self.news_encoder = NewsEncoder(args, news_embedding, num_category, num_subcategory)
        self.user_encoder = UserEncoder(args)
        self.loss_fn = nn.CrossEntropyLoss()

    def forward(self, history, history_mask, candidate, label):
        """history [B, H, F] int32; history_mask [B, H]; candidate [B, 1+K, F] int32; label [B] int64."""
        a = self.args
        F = history.shape[-1]
        B, C = candidate.shape[0], 1 + a.npratio
        needed = None                                  # masked history slots reach the loss through a factor 0 (NAML.py:92-96)
        if not getattr(a, "encode_masked_slots", False):
            needed = torch.cat([history_mask.new_ones(B * C), history_mask.reshape(-1)])
        vecs = self.news_encoder(torch.cat([candidate.reshape(-1, F), history.reshape(-1, F)], dim=0), needed=needed)
        cand_flat, hist_flat = vecs.split([B * C, vecs.shape[0] - B * C], dim=0)     # one cat in backward, no zero fills
        cand_vecs = cand_flat.reshape(B, C, a.news_dim)
        hist_vecs = hist_flat.reshape(B, a.user_log_length, a.news_dim)
        user_vec = self.user_encoder(hist_vecs, history_mask)
        loss, score = ops.score_ce(cand_vecs, user_vec, label)
        return loss, score
