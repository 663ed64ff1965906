"""CPU oracle for the NRMS / NAML encoder + scorer path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch restatement (torch-CPU fp32, functional style) of the
arithmetic the reference performs on its hot path.  It exists so that the HIP
kernels can be checked on machines where /root/reference is absent (the GPU box).
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import it.  The product package `newsrecommendation_amd` never does.

Pinning: every function below is checked against outputs of the reference itself
(imported from /root/reference/src in the build container) by
`tests/golden/make_golden.py`, which writes the `.npz` fixtures under
`tests/golden/`; `tests/test_oracle_golden.py` replays them.  The reference holds
no tests or golden vectors of its own.

Each function cites the reference file:line it restates (paths relative to
/root/reference/).
"""
from __future__ import annotations

import math
import random
from types import SimpleNamespace
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

Tensor = torch.Tensor
EPS = 1e-8  # src/model/model_utils.py:29,53


# --------------------------------------------------------------------------------------
# op library  (src/model/model_utils.py)
# --------------------------------------------------------------------------------------
def additive_pool(x: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor,
                  mask: Optional[Tensor] = None) -> Tensor:
    """AttentionPooling.forward, src/model/model_utils.py:13-31.

    x [n, L, N]; w1 [q, N]; b1 [q]; w2 [1, q]; b2 [1]; mask [n, L] (0/1) or None -> [n, N].
    alpha = exp(tanh(x W1^T + b1) w2^T + b2) (* mask); alpha /= sum_L(alpha) + 1e-8;
    out = sum_L alpha_l x_l.  No max-subtraction (line 24), mask applied after exp (27).
    """
    e = torch.tanh(x @ w1.t() + b1)                      # :21-22
    alpha = torch.exp(e @ w2.t() + b2)                   # :23-24   [n, L, 1]
    if mask is not None:
        alpha = alpha * mask.unsqueeze(2)                # :26-27
    alpha = alpha / (alpha.sum(dim=1, keepdim=True) + EPS)   # :29
    return (x * alpha).sum(dim=1)                        # :30  (bmm(x^T, alpha))


def sdpa(q: Tensor, k: Tensor, v: Tensor, mask: Optional[Tensor] = None) -> Tensor:
    """ScaledDotProductAttention.forward, src/model/model_utils.py:39-55.

    q,k,v [n, h, L, d]; mask [n, h, L] (key side) -> [n, h, L, d].
    exp without max-subtraction (:48), multiplicative key mask after exp (:50-51),
    denominator sum + 1e-8 (:53).
    """
    d_k = q.shape[-1]
    s = (q @ k.transpose(-1, -2)) / math.sqrt(d_k)       # :47 (np.sqrt(d_k), float64 scalar)
    s = torch.exp(s)                                     # :48
    if mask is not None:
        s = s * mask.unsqueeze(-2)                       # :50-51
    a = s / (s.sum(dim=-1, keepdim=True) + EPS)          # :53
    return a @ v                                         # :54


def mhsa(x: Tensor, wq: Tensor, bq: Tensor, wk: Tensor, bk: Tensor, wv: Tensor, bv: Tensor,
         n_heads: int, mask: Optional[Tensor] = None) -> Tensor:
    """MultiHeadSelfAttention.forward (Q=K=V=x), src/model/model_utils.py:78-95.

    x [n, L, d_model]; w* [h*d, d_model] -> [n, L, h*d].  No output projection.
    """
    n, L, _ = x.shape
    d = wq.shape[0] // n_heads
    hm = None
    if mask is not None:
        hm = mask.unsqueeze(1).expand(-1, n_heads, -1)   # :86-87
    q = (x @ wq.t() + bq).view(n, L, n_heads, d).transpose(1, 2)   # :89
    k = (x @ wk.t() + bk).view(n, L, n_heads, d).transpose(1, 2)   # :90
    v = (x @ wv.t() + bv).view(n, L, n_heads, d).transpose(1, 2)   # :91
    ctx = sdpa(q, k, v, hm)                                        # :93
    return ctx.transpose(1, 2).contiguous().view(n, L, n_heads * d)  # :94


def apply_dropout(x: Tensor, keep: Optional[Tensor], p: float) -> Tensor:
    """F.dropout with an externally supplied keep mask (1 = kept): x * keep / (1-p).

    Restates the arithmetic of F.dropout(training=True) (src/model/NRMS.py:28-34,
    src/model/NAML.py:51-53) for a given Bernoulli draw; keep=None is eval mode.
    """
    if keep is None:
        return x
    return x * keep * (1.0 / (1.0 - p))


def score_ce(cand: Tensor, user: Tensor, label: Tensor) -> Tuple[Tensor, Tensor]:
    """bmm scorer + CrossEntropyLoss, src/model/NRMS.py:93-94 / src/model/NAML.py:128-129.

    cand [B, C, N]; user [B, N]; label int64 [B] -> (loss scalar, score [B, C]).
    """
    score = torch.bmm(cand, user.unsqueeze(-1)).squeeze(-1)
    loss = torch.nn.functional.cross_entropy(score, label)
    return loss, score


def embed_rows(table: Tensor, ids: Tensor) -> Tensor:
    """nn.Embedding(..., padding_idx=0) lookup: src/model/NRMS.py:71-73,28, src/model/NAML.py:19,22,105-107.
    Row gather; row 0 receives NO gradient (padding_idx), its VALUE is whatever the table holds."""
    return torch.nn.functional.embedding(ids.long(), table, padding_idx=0)


def pad_blend(x: Tensor, mask: Tensor, pad_doc: Tensor) -> Tensor:
    """x*m + pad_doc*(1-m), src/model/NRMS.py:59-60 / src/model/NAML.py:94-95."""
    m = mask.unsqueeze(-1)
    return x * m + pad_doc.unsqueeze(0) * (1 - m)


# --------------------------------------------------------------------------------------
# NRMS  (src/model/NRMS.py)
# --------------------------------------------------------------------------------------
def _p(sd: Dict[str, Tensor], prefix: str, names: Sequence[str]) -> List[Tensor]:
    return [sd[prefix + n] for n in names]


_MHSA_KEYS = ["W_Q.weight", "W_Q.bias", "W_K.weight", "W_K.bias", "W_V.weight", "W_V.bias"]
_POOL_KEYS = ["att_fc1.weight", "att_fc1.bias", "att_fc2.weight", "att_fc2.bias"]


def nrms_news_encoder(ids: Tensor, sd: Dict[str, Tensor], cfg, keep_word: Optional[Tensor] = None,
                      keep_ctx: Optional[Tensor] = None) -> Tensor:
    """NRMS NewsEncoder.forward, src/model/NRMS.py:23-36.  ids int [n, T] -> [n, news_dim].

    No token mask is ever passed at the call sites (src/model/NRMS.py:87,90).
    """
    table = sd["news_encoder.embedding_matrix.weight"]
    x = embed_rows(table, ids)                                                # :28
    x = apply_dropout(x, keep_word, cfg.drop_rate)                            # :28-30
    y = mhsa(x, *_p(sd, "news_encoder.multi_head_self_attn.", _MHSA_KEYS),
             n_heads=cfg.num_attention_heads)                                 # :31
    y = apply_dropout(y, keep_ctx, cfg.drop_rate)                             # :32-34
    return additive_pool(y, *_p(sd, "news_encoder.attn.", _POOL_KEYS))        # :35


def nrms_user_encoder(news_vecs: Tensor, log_mask: Tensor, sd: Dict[str, Tensor], cfg) -> Tensor:
    """NRMS UserEncoder.forward, src/model/NRMS.py:49-63."""
    mh = _p(sd, "user_encoder.multi_head_self_attn.", _MHSA_KEYS)
    pl = _p(sd, "user_encoder.attn.", _POOL_KEYS)
    if cfg.user_log_mask:                                                     # :55-57
        y = mhsa(news_vecs, *mh, n_heads=cfg.num_attention_heads, mask=log_mask)
        return additive_pool(y, *pl, mask=log_mask)
    x = pad_blend(news_vecs, log_mask, sd["user_encoder.pad_doc"])            # :59-60
    y = mhsa(x, *mh, n_heads=cfg.num_attention_heads)                         # :61
    return additive_pool(y, *pl)                                              # :62


def nrms_forward(history: Tensor, history_mask: Tensor, candidate: Tensor, label: Tensor,
                 sd: Dict[str, Tensor], cfg, keep=None) -> Tuple[Tensor, Tensor]:
    """NRMS Model.forward, src/model/NRMS.py:79-95.

    `keep` (training-mode parity only): dict with optional 'cand_word','cand_ctx','hist_word',
    'hist_ctx' keep masks, shapes of the tensors they gate.
    """
    keep = keep or {}
    T, N = cfg.num_words_title, cfg.news_dim
    cand = nrms_news_encoder(candidate.reshape(-1, T), sd, cfg, keep.get("cand_word"),
                             keep.get("cand_ctx")).reshape(-1, 1 + cfg.npratio, N)      # :86-87
    hist = nrms_news_encoder(history.reshape(-1, T), sd, cfg, keep.get("hist_word"),
                             keep.get("hist_ctx")).reshape(-1, cfg.user_log_length, N)  # :89-90
    user = nrms_user_encoder(hist, history_mask, sd, cfg)                               # :92
    return score_ce(cand, user, label)                                                  # :93-95


# --------------------------------------------------------------------------------------
# NAML  (src/model/NAML.py)
# --------------------------------------------------------------------------------------
def conv1d_k3(x: Tensor, w: Tensor, b: Tensor) -> Tensor:
    """nn.Conv1d(D, N, kernel_size=3, padding=1) applied over the token axis,
    src/model/NAML.py:27-32,54.  x [n, T, D]; w [N, D, 3]; b [N] -> [n, T, N].
    y[t] = b + W[:,:,0] x[t-1] + W[:,:,1] x[t] + W[:,:,2] x[t+1], zero beyond the ends."""
    n, T, D = x.shape
    z = torch.zeros(n, 1, D, dtype=x.dtype)
    xp = torch.cat([z, x, z], dim=1)
    y = b
    for j in range(3):
        y = y + xp[:, j:j + T, :] @ w[:, :, j].t()
    return y


def naml_news_encoder(x: Tensor, sd: Dict[str, Tensor], cfg, keep_word: Optional[Tensor] = None) -> Tensor:
    """NAML NewsEncoder.forward, src/model/NAML.py:35-75.  x int [n, F], F in {1,2,3}."""
    T, D = cfg.num_words_title, cfg.word_embedding_dim
    table = sd["news_encoder.title_embeddings.weight"]
    emb = embed_rows(table, x[:, 0]).reshape(-1, T, D)                          # :47-50
    emb = apply_dropout(emb, keep_word, cfg.drop_rate)                          # :51-53
    ctx = conv1d_k3(emb, sd["news_encoder.cnn.weight"], sd["news_encoder.cnn.bias"])   # :54
    vecs = [additive_pool(ctx, *_p(sd, "news_encoder.attn.", _POOL_KEYS))]      # :55
    col = 1
    if cfg.use_category:                                                        # :60-64
        c = embed_rows(sd["news_encoder.category_emb.weight"], x[:, col])
        vecs.append(c @ sd["news_encoder.category_dense.weight"].t() + sd["news_encoder.category_dense.bias"])
        col += 1
    if cfg.use_subcategory:                                                     # :65-68
        s = embed_rows(sd["news_encoder.subcategory_emb.weight"], x[:, col])
        vecs.append(s @ sd["news_encoder.subcategory_dense.weight"].t() + sd["news_encoder.subcategory_dense.bias"])
    if len(vecs) == 1:                                                          # :70-71
        return vecs[0]
    return additive_pool(torch.stack(vecs, dim=1), *_p(sd, "news_encoder.final_attn.", _POOL_KEYS))  # :73-74


def naml_user_encoder(news_vecs: Tensor, log_mask: Tensor, sd: Dict[str, Tensor], cfg) -> Tensor:
    """NAML UserEncoder.forward, src/model/NAML.py:85-97."""
    pl = _p(sd, "user_encoder.attn.", _POOL_KEYS)
    if cfg.user_log_mask:
        return additive_pool(news_vecs, *pl, mask=log_mask)                     # :92
    return additive_pool(pad_blend(news_vecs, log_mask, sd["user_encoder.pad_doc"]), *pl)  # :94-96


def naml_forward(history: Tensor, history_mask: Tensor, candidate: Tensor, label: Tensor,
                 sd: Dict[str, Tensor], cfg, keep=None) -> Tuple[Tensor, Tensor]:
    """NAML Model.forward, src/model/NAML.py:113-130."""
    keep = keep or {}
    F, N = history.shape[-1], cfg.news_dim                                      # :120
    cand = naml_news_encoder(candidate.reshape(-1, F), sd, cfg, keep.get("cand_word")
                             ).reshape(-1, 1 + cfg.npratio, N)                  # :121-122
    hist = naml_news_encoder(history.reshape(-1, F), sd, cfg, keep.get("hist_word")
                             ).reshape(-1, cfg.user_log_length, N)              # :124-125
    user = naml_user_encoder(hist, history_mask, sd, cfg)                       # :127
    return score_ce(cand, user, label)                                          # :128-130


# --------------------------------------------------------------------------------------
# parameter initialisation helpers for synthetic runs (shapes: SURVEY.md Appendix A)
# --------------------------------------------------------------------------------------
def default_cfg(**kw):
    """Flag defaults of src/parameters.py:5-62 that the path reads, with MIND shapes of BASELINE.json."""
    cfg = dict(num_words_title=30, user_log_length=50, npratio=4, word_embedding_dim=300, news_dim=400,
               num_attention_heads=20, news_query_vector_dim=200, user_query_vector_dim=200,
               drop_rate=0.2, user_log_mask=False, freeze_embedding=False, use_category=False,
               use_subcategory=False, category_emb_dim=100)
    cfg.update(kw)
    return SimpleNamespace(**cfg)


def _xavier(g, out_f, in_f):
    a = math.sqrt(6.0 / (in_f + out_f))
    return (torch.rand(out_f, in_f, generator=g) * 2 - 1) * a


def _lin_default(g, out_f, in_f):
    a = 1.0 / math.sqrt(in_f)
    return (torch.rand(out_f, in_f, generator=g) * 2 - 1) * a, (torch.rand(out_f, generator=g) * 2 - 1) * a


def init_state_dict(model: str, cfg, table: Tensor, seed: int = 0, n_cat: int = 0, n_sub: int = 0) -> Dict[str, Tensor]:
    """Random parameters with the reference's key names/shapes/init families (not its RNG stream):
    xavier-uniform W_{Q,K,V} (src/model/model_utils.py:73-76), Linear-default elsewhere,
    pad_doc ~ U(-1,1) (src/model/NRMS.py:47)."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, Tensor] = {}
    N, D, q = cfg.news_dim, cfg.word_embedding_dim, cfg.news_query_vector_dim

    def mh(prefix, d_model):
        for nm in ("W_Q", "W_K", "W_V"):
            sd[f"{prefix}{nm}.weight"] = _xavier(g, N, d_model)
            sd[f"{prefix}{nm}.bias"] = _lin_default(g, N, d_model)[1]

    def pool(prefix, qd):
        sd[prefix + "att_fc1.weight"], sd[prefix + "att_fc1.bias"] = _lin_default(g, qd, N)
        sd[prefix + "att_fc2.weight"], sd[prefix + "att_fc2.bias"] = _lin_default(g, 1, qd)

    if model == "NRMS":
        sd["news_encoder.embedding_matrix.weight"] = table.float()
        mh("news_encoder.multi_head_self_attn.", D)
        pool("news_encoder.attn.", q)
        sd["user_encoder.pad_doc"] = torch.rand(1, N, generator=g) * 2 - 1
        mh("user_encoder.multi_head_self_attn.", N)
        pool("user_encoder.attn.", cfg.user_query_vector_dim)
    elif model == "NAML":
        sd["news_encoder.title_embeddings.weight"] = table.float()
        C = cfg.category_emb_dim
        if cfg.use_category:
            e = torch.randn(n_cat + 1, C, generator=g); e[0] = 0
            sd["news_encoder.category_emb.weight"] = e
            sd["news_encoder.category_dense.weight"], sd["news_encoder.category_dense.bias"] = _lin_default(g, N, C)
        if cfg.use_subcategory:
            e = torch.randn(n_sub + 1, C, generator=g); e[0] = 0
            sd["news_encoder.subcategory_emb.weight"] = e
            sd["news_encoder.subcategory_dense.weight"], sd["news_encoder.subcategory_dense.bias"] = _lin_default(g, N, C)
        if cfg.use_category or cfg.use_subcategory:
            pool("news_encoder.final_attn.", q)
        a = 1.0 / math.sqrt(D * 3)
        sd["news_encoder.cnn.weight"] = (torch.rand(N, D, 3, generator=g) * 2 - 1) * a
        sd["news_encoder.cnn.bias"] = (torch.rand(N, generator=g) * 2 - 1) * a
        pool("news_encoder.attn.", q)
        sd["user_encoder.pad_doc"] = torch.rand(1, N, generator=g) * 2 - 1
        pool("user_encoder.attn.", cfg.user_query_vector_dim)
    else:
        raise ValueError(model)
    return sd


# --------------------------------------------------------------------------------------
# negative sampling / label position / history padding (integer work, bit-exact)
# --------------------------------------------------------------------------------------
def get_sample(all_elements: List[str], num_sample: int) -> List[str]:
    """src/prepare_data.py:7-11 — random.sample, list replicated when too short."""
    if num_sample > len(all_elements):
        return random.sample(all_elements * (num_sample // len(all_elements) + 1), num_sample)
    return random.sample(all_elements, num_sample)


def prepare_training_lines(behavior_lines: Sequence[str], n_shards: int, npratio: int, seed: int) -> List[List[str]]:
    """src/prepare_data.py:14-49 without the file I/O: seed (:15), one output line per positive with
    `npratio` sampled negatives (:31-35), global shuffle (:37), round-robin shard i % nGPU (:39-41).
    Uses stdlib `random` exactly as the reference does (MT19937 state is the contract)."""
    random.seed(seed)
    out: List[str] = []
    for line in behavior_lines:
        iid, uid, time, history, imp = line.strip().split("\t")
        pos, neg = [], []
        for item in imp.split(" "):
            nid, lab = item.split("-")
            if lab == "0":
                neg.append(nid)
            elif lab == "1":
                pos.append(nid)
        if len(pos) == 0 or len(neg) == 0:
            continue
        for p in pos:
            negs = get_sample(neg, npratio)
            out.append("\t".join([iid, uid, time, history, p, " ".join(negs)]) + "\n")
    random.shuffle(out)
    shards: List[List[str]] = [[] for _ in range(n_shards)]
    for i, l in enumerate(out):
        shards[i % n_shards].append(l)
    return shards


def prepare_testing_lines(behavior_lines: Sequence[str], n_shards: int) -> List[List[str]]:
    """src/prepare_data.py:52-66: round-robin i % nGPU, no shuffle."""
    shards: List[List[str]] = [[] for _ in range(n_shards)]
    for i, l in enumerate(behavior_lines):
        shards[i % n_shards].append(l)
    return shards


def pad_to_fix_len(x: List[int], fix_length: int) -> Tuple[List[int], np.ndarray]:
    """src/dataset.py:17-24 (padding_front=True): keep the LAST fix_length items, left-pad with 0."""
    pad_x = [0] * (fix_length - len(x)) + x[-fix_length:]
    mask = [0] * (fix_length - len(x)) + [1] * min(fix_length, len(x))
    return pad_x, np.array(mask, dtype="float32")


def train_line_to_indices(line: str, news_index: Dict[str, int], user_log_length: int, npratio: int):
    """src/dataset.py:26-49 up to (not including) the news_combined gather: unknown ids -> 0 (:15),
    label = random.randint(0, npratio) (:45), positive spliced at `label` (:46).
    Returns (history_idx [H], mask [H] f32, sample_idx [1+K], label)."""
    f = line.strip().split("\t")
    tr = lambda ids: [news_index[i] if i in news_index else 0 for i in ids]
    hist, mask = pad_to_fix_len(tr(f[3].split()), user_log_length)
    pos, neg = tr(f[4].split()), tr(f[5].split())
    label = random.randint(0, npratio)
    sample = neg[:label] + pos + neg[label:]
    return hist, mask, sample, label


def test_line_to_indices(line: str, news_index: Dict[str, int], user_log_length: int):
    """src/dataset.py:64-74 up to the news_scoring gather."""
    f = line.strip().split("\t")
    tr = lambda ids: [news_index[i] if i in news_index else 0 for i in ids]
    hist, mask = pad_to_fix_len(tr(f[3].split()), user_log_length)
    cand = tr([i.split("-")[0] for i in f[4].split()])
    labels = np.array([int(i.split("-")[1]) for i in f[4].split()])
    return hist, mask, cand, labels


# --------------------------------------------------------------------------------------
# ranking metrics (src/metrics.py) — "next" row f2, restated for the eval path
# --------------------------------------------------------------------------------------
def dcg_score(y_true, y_score, k=10):
    """src/metrics.py:5-10."""
    order = np.argsort(y_score)[::-1]
    y_true = np.take(y_true, order[:k])
    return np.sum((2 ** y_true - 1) / np.log2(np.arange(len(y_true)) + 2))


def ndcg_score(y_true, y_score, k=10):
    """src/metrics.py:13-16."""
    return dcg_score(y_true, y_score, k) / dcg_score(y_true, y_true, k)


def mrr_score(y_true, y_score):
    """src/metrics.py:19-23."""
    order = np.argsort(y_score)[::-1]
    y_true = np.take(y_true, order)
    return np.sum(y_true / (np.arange(len(y_true)) + 1)) / np.sum(y_true)


def auc_score(y_true, y_score):
    """sklearn.metrics.roc_auc_score (src/metrics.py:1) restated as the rank statistic
    (Mann-Whitney U with average ranks for ties)."""
    y_true = np.asarray(y_true); y_score = np.asarray(y_score, dtype=np.float64)
    order = np.argsort(y_score, kind="mergesort")
    s = y_score[order]
    ranks = np.empty(len(s), dtype=np.float64)
    i = 0
    while i < len(s):
        j = i
        while j + 1 < len(s) and s[j + 1] == s[i]:
            j += 1
        ranks[order[i:j + 1]] = 0.5 * (i + j) + 1.0
        i = j + 1
    n_pos = float(y_true.sum()); n_neg = float(len(y_true) - n_pos)
    return (ranks[y_true == 1].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg)
