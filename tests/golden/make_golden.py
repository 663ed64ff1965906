#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE.

Build-container only: needs /root/reference (read-only, never copied) and runs on CPU.
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

For every case the script (1) runs the reference module imported from
/root/reference/src, (2) runs oracle/nr_oracle.py on the same inputs and asserts that
ALL outputs and ALL gradients agree to <= 2e-6 (abs, fp32), and (3) stores inputs, the
reference's state_dict and the reference's outputs/gradients as .npz.  For the two
MIND-shaped cases the big weight gradients are stored as a strided row sample (the full
comparison against the reference happens here, at generation time); everything else is
stored whole.  Fixtures are data only: tensors and text lines, no reference source.
"""
import argparse
import hashlib
import json
import os
import random
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/src")
sys.dont_write_bytecode = True

from model import NRMS as RefNRMS, NAML as RefNAML, model_utils as ref_mu   # noqa: E402  (reference)
import dataset as ref_dataset                                              # noqa: E402
import prepare_data as ref_prepare                                         # noqa: E402
import metrics as ref_metrics                                              # noqa: E402
from oracle import nr_oracle as O                                          # noqa: E402

TOL = 2e-6


def close(a, b, name, tol=TOL):
    a = a.detach() if isinstance(a, torch.Tensor) else torch.as_tensor(a)
    b = b.detach() if isinstance(b, torch.Tensor) else torch.as_tensor(b)
    err = (a.double() - b.double()).abs().max().item() if a.numel() else 0.0
    scale = max(1.0, b.double().abs().max().item() if b.numel() else 1.0)
    assert a.shape == b.shape, (name, a.shape, b.shape)
    assert err <= tol * scale, f"oracle != reference for {name}: {err}"
    return err


def np_(t):
    return t.detach().cpu().numpy()


def make_cfg(**kw):
    return argparse.Namespace(**vars(O.default_cfg(**kw)))


# ----------------------------------------------------------------------------- op level
def gen_ops(seed=11):
    torch.manual_seed(seed)
    out = {}
    n, L, N, q, h, d_model = 3, 5, 8, 4, 2, 6
    # AttentionPooling (model_utils.py:7-31)
    pool = ref_mu.AttentionPooling(N, q)
    x = torch.randn(n, L, N, requires_grad=True)
    mask = torch.tensor([[1, 1, 1, 0, 0], [0, 0, 0, 0, 0], [0, 1, 1, 1, 1]], dtype=torch.float32)
    g = torch.randn(n, N)
    for tag, m in (("nomask", None), ("mask", mask)):
        for p in pool.parameters():
            p.grad = None
        x.grad = None
        y = pool(x, m)
        y.backward(g)
        sd = {k: v.detach().clone() for k, v in pool.state_dict().items()}
        xo = x.detach().clone().requires_grad_(True)
        po = [sd[k].clone().requires_grad_(True) for k in ("att_fc1.weight", "att_fc1.bias", "att_fc2.weight", "att_fc2.bias")]
        yo = O.additive_pool(xo, *po, mask=m)
        yo.backward(g)
        close(yo, y, f"pool/{tag}/y")
        close(xo.grad, x.grad, f"pool/{tag}/dx")
        for t, nm in zip(po, ("att_fc1.weight", "att_fc1.bias", "att_fc2.weight", "att_fc2.bias")):
            close(t.grad, dict(pool.named_parameters())[nm].grad, f"pool/{tag}/d{nm}")
            out[f"pool_{tag}_d_{nm}"] = np_(dict(pool.named_parameters())[nm].grad)
        out[f"pool_{tag}_y"] = np_(y)
        out[f"pool_{tag}_dx"] = np_(x.grad)
    for k, v in pool.state_dict().items():
        out[f"pool_sd_{k}"] = np_(v)
    out["pool_x"], out["pool_mask"], out["pool_g"] = np_(x), np_(mask), np_(g)

    # MultiHeadSelfAttention (model_utils.py:58-95) incl. SDPA (:34-55)
    mh = ref_mu.MultiHeadSelfAttention(d_model, h, N // h, N // h)
    x = torch.randn(n, L, d_model, requires_grad=True)
    g = torch.randn(n, L, N)
    for tag, m in (("nomask", None), ("mask", mask)):
        for p in mh.parameters():
            p.grad = None
        x.grad = None
        y = mh(x, x, x, m)
        y.backward(g)
        sd = {k: v.detach().clone() for k, v in mh.state_dict().items()}
        keys = ["W_Q.weight", "W_Q.bias", "W_K.weight", "W_K.bias", "W_V.weight", "W_V.bias"]
        xo = x.detach().clone().requires_grad_(True)
        po = [sd[k].clone().requires_grad_(True) for k in keys]
        yo = O.mhsa(xo, *po, n_heads=h, mask=m)
        yo.backward(g)
        close(yo, y, f"mhsa/{tag}/y")
        close(xo.grad, x.grad, f"mhsa/{tag}/dx")
        for t, nm in zip(po, keys):
            close(t.grad, dict(mh.named_parameters())[nm].grad, f"mhsa/{tag}/d{nm}")
            out[f"mhsa_{tag}_d_{nm}"] = np_(dict(mh.named_parameters())[nm].grad)
        out[f"mhsa_{tag}_y"] = np_(y)
        out[f"mhsa_{tag}_dx"] = np_(x.grad)
    for k, v in mh.state_dict().items():
        out[f"mhsa_sd_{k}"] = np_(v)
    out["mhsa_x"], out["mhsa_mask"], out["mhsa_g"] = np_(x), np_(mask), np_(g)
    out["mhsa_heads"] = np.array(h)
    np.savez_compressed(os.path.join(HERE, "ops_tiny.npz"), **out)
    print("ops_tiny.npz", len(out), "arrays")


# ----------------------------------------------------------------------------- model level
def synth_batch(gen, B, cfg, n_rows, feat):
    H, T, C = cfg.user_log_length, cfg.num_words_title, 1 + cfg.npratio
    if feat is None:          # NRMS: word ids [.., T], tails zero padded
        hist = torch.randint(1, n_rows, (B, H, T), generator=gen, dtype=torch.int32)
        cand = torch.randint(1, n_rows, (B, C, T), generator=gen, dtype=torch.int32)
        for t in (hist, cand):
            ln = torch.randint(1, T + 1, t.shape[:2], generator=gen)
            t[torch.arange(T)[None, None, :] >= ln[..., None]] = 0
    else:                      # NAML: [news id, cat id, subcat id][:feat]
        def ids(shape):
            cols = [torch.randint(0, n_rows, shape, generator=gen, dtype=torch.int32)]
            if feat >= 2:
                cols.append(torch.randint(0, 5, shape, generator=gen, dtype=torch.int32))
            if feat >= 3:
                cols.append(torch.randint(0, 7, shape, generator=gen, dtype=torch.int32))
            return torch.stack(cols, dim=-1)
        hist, cand = ids((B, H)), ids((B, C))
    hl = torch.randint(0, H + 1, (B,), generator=gen)
    hl[0] = 0                                       # an all-padding history row
    if B > 1:
        hl[1] = H
    mask = (torch.arange(H)[None, :] >= (H - hl)[:, None]).float()   # front padded (dataset.py:19-20)
    if feat is None:
        hist[mask == 0] = 0
    else:
        hist[mask == 0] = 0
    label = torch.randint(0, C, (B,), generator=gen, dtype=torch.int64)
    return hist, mask, cand, label


def run_model(kind, tag, cfg, B, n_rows, seed, feat=None, sample_rows=None):
    torch.manual_seed(seed)
    gen = torch.Generator().manual_seed(seed + 1)
    if kind == "NRMS":
        table = torch.randn(n_rows, cfg.word_embedding_dim, generator=gen) * 0.4
        table[0] = 0
        ref = RefNRMS.Model(cfg, table.numpy())
        tkey = "news_encoder.embedding_matrix.weight"
        fwd = O.nrms_forward
    else:
        table = torch.randn(n_rows, cfg.num_words_title * cfg.word_embedding_dim, generator=gen) * 0.4
        table[0] = 0
        ref = RefNAML.Model(cfg, table.numpy(), 4, 6)
        tkey = "news_encoder.title_embeddings.weight"
        fwd = O.naml_forward
    ref.eval()
    hist, mask, cand, label = synth_batch(gen, B, cfg, n_rows, feat)
    loss, score = ref(hist, mask, cand, label)
    loss.backward()
    F = hist.shape[-1]
    cand_vecs = ref.news_encoder(cand.reshape(-1, F if kind == "NAML" else cfg.num_words_title))
    hist_vecs = ref.news_encoder(hist.reshape(-1, F if kind == "NAML" else cfg.num_words_title))
    user_vec = ref.user_encoder(hist_vecs.reshape(B, cfg.user_log_length, -1), mask)

    sd = {k: v.detach().clone() for k, v in ref.state_dict().items()}
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    lo, so = fwd(hist, mask, cand, label, sdo, cfg)
    lo.backward()
    close(lo, loss, f"{tag}/loss")
    close(so, score, f"{tag}/score")
    grads = {}
    for name, p in ref.named_parameters():
        if p.grad is None:           # frozen table (requires_grad False), or pad_doc under user_log_mask=True
            assert (not p.requires_grad) or sdo[name].grad is None or float(sdo[name].grad.abs().max()) == 0.0, name
            continue
        close(sdo[name].grad, p.grad, f"{tag}/d{name}")
        grads[name] = p.grad.detach()
    assert float(grads[tkey][0].abs().max()) == 0.0 if tkey in grads else True   # padding_idx row

    out = {"hist": np_(hist), "mask": np_(mask), "cand": np_(cand), "label": np_(label),
           "loss": np_(loss), "score": np_(score), "cand_vecs": np_(cand_vecs), "hist_vecs": np_(hist_vecs),
           "user_vec": np_(user_vec)}
    for k, v in sd.items():
        out["sd::" + k] = np_(v)
    for k, g in grads.items():
        if sample_rows and g.dim() == 2 and g.shape[0] >= 64:
            out["gradrows::" + k] = np_(g[::sample_rows])
            out["gradsum::" + k] = np.array(g.double().sum().item())
        else:
            out["grad::" + k] = np_(g)
    out["cfg_json"] = np.array(json.dumps({k: v for k, v in vars(cfg).items()}))
    out["sample_rows"] = np.array(sample_rows or 0)
    np.savez_compressed(os.path.join(HERE, f"{tag}.npz"), **out)
    print(f"{tag}.npz loss={loss.item():.6f} params={sum(v.numel() for v in sd.values())}")


def gen_models():
    tiny = dict(num_words_title=4, user_log_length=3, npratio=1, word_embedding_dim=8, news_dim=8,
                num_attention_heads=2, news_query_vector_dim=4, user_query_vector_dim=4, category_emb_dim=5)
    run_model("NRMS", "nrms_tiny_pad", make_cfg(**tiny, user_log_mask=False), B=3, n_rows=12, seed=3)
    run_model("NRMS", "nrms_tiny_mask", make_cfg(**tiny, user_log_mask=True), B=3, n_rows=12, seed=4)
    run_model("NAML", "naml_tiny_3view", make_cfg(**tiny, use_category=True, use_subcategory=True), B=3,
              n_rows=9, seed=5, feat=3)
    run_model("NAML", "naml_tiny_title_mask", make_cfg(**tiny, user_log_mask=True, freeze_embedding=True), B=3,
              n_rows=9, seed=6, feat=1)
    # MIND-shaped (BASELINE.json: title_len=30, history=50, npratio=4, 300-d, 400/20 heads)
    run_model("NRMS", "nrms_mind_pad", make_cfg(user_log_mask=False), B=4, n_rows=300, seed=7, sample_rows=16)
    run_model("NRMS", "nrms_mind_mask", make_cfg(user_log_mask=True), B=2, n_rows=300, seed=8, sample_rows=16)
    run_model("NAML", "naml_mind_3view", make_cfg(use_category=True, use_subcategory=True, freeze_embedding=True),
              B=2, n_rows=40, seed=9, feat=3, sample_rows=16)


# ----------------------------------------------------------------------------- index selection
def gen_index_selection(seed=5):
    rnd = random.Random(1234)
    news_ids = [f"N{i}" for i in range(1, 41)]
    lines = []
    for i in range(60):
        hist = " ".join(rnd.choice(news_ids + ["N999"]) for _ in range(rnd.randint(0, 9)))
        n_imp = rnd.randint(1, 8)
        imps = [f"{rnd.choice(news_ids + ['N777'])}-{1 if rnd.random() < 0.3 else 0}" for _ in range(n_imp)]
        lines.append("\t".join([str(i + 1), f"U{rnd.randint(1, 20)}", "11/11/2019 9:00:00 AM", hist, " ".join(imps)]) + "\n")
    news_index = {nid: i + 1 for i, nid in enumerate(news_ids)}
    res = {"behaviors": lines, "news_index": news_index, "seed": seed, "npratio": 4, "user_log_length": 5, "cases": {}}
    for n_shards in (1, 2):
        with tempfile.TemporaryDirectory() as d:
            with open(os.path.join(d, "behaviors.tsv"), "w") as f:
                f.writelines(lines)
            total = ref_prepare.prepare_training_data(d, n_shards, 4, seed)
            shards = [open(os.path.join(d, f"behaviors_np4_{i}.tsv")).readlines() for i in range(n_shards)]
            ref_prepare.prepare_testing_data(d, n_shards)
            tshards = [open(os.path.join(d, f"behaviors_{i}.tsv")).readlines() for i in range(n_shards)]
            mine = O.prepare_training_lines(lines, n_shards, 4, seed)
            assert mine == shards, "oracle sharder != reference"
            assert O.prepare_testing_lines(lines, n_shards) == tshards
            assert total == sum(len(s) for s in shards)
            # DatasetTrain stream (dataset.py:26-49) with news_combined = identity column
            args = argparse.Namespace(user_log_length=5, npratio=4)
            comb = np.arange(len(news_ids) + 1, dtype="int32")[:, None]
            streams = []
            for r in range(n_shards):
                random.seed(seed + r)
                ds = ref_dataset.DatasetTrain(os.path.join(d, f"behaviors_np4_{r}.tsv"), news_index, comb, args)
                got = [(h[:, 0].tolist(), m.tolist(), c[:, 0].tolist(), int(l)) for h, m, c, l in ds]
                random.seed(seed + r)
                exp = [O.train_line_to_indices(l, news_index, 5, 4) for l in shards[r]]
                exp = [(h, m.tolist(), c, l) for h, m, c, l in exp]
                assert got == exp, "oracle line_mapper != reference"
                streams.append(got)
            # DatasetTest (dataset.py:64-74) with news_scoring = identity column
            tstreams = []
            for r in range(n_shards):
                ds = ref_dataset.DatasetTest(os.path.join(d, f"behaviors_{r}.tsv"), news_index, comb, args)
                got = [(h[:, 0].tolist(), m.tolist(), c[:, 0].tolist(), l.tolist()) for h, m, c, l in ds]
                exp = [O.test_line_to_indices(l, news_index, 5) for l in tshards[r]]
                assert got == [(h, m.tolist(), c, l.tolist()) for h, m, c, l in exp]
                tstreams.append(got)
            res["cases"][str(n_shards)] = {
                "train_shards": shards,
                "train_sha256": [hashlib.sha256("".join(s).encode()).hexdigest() for s in shards],
                "test_shards": tshards, "train_stream": streams, "test_stream": tstreams}
    with open(os.path.join(HERE, "index_selection.json"), "w") as f:
        json.dump(res, f)
    print("index_selection.json", {k: [len(s) for s in v["train_shards"]] for k, v in res["cases"].items()})


def gen_metrics(seed=2):
    rng = np.random.RandomState(seed)
    rows = []
    for _ in range(40):
        c = rng.randint(2, 60)
        y = (rng.rand(c) < 0.2).astype(np.int64)
        if y.sum() == 0:
            y[rng.randint(c)] = 1
        if y.sum() == c:
            y[rng.randint(c)] = 0
        s = rng.randn(c).astype(np.float32)
        if rng.rand() < 0.3:
            s[rng.randint(c)] = s[rng.randint(c)]     # ties
        exp = [ref_metrics.roc_auc_score(y, s), ref_metrics.mrr_score(y, s), ref_metrics.ndcg_score(y, s, 5),
               ref_metrics.ndcg_score(y, s, 10)]
        got = [O.auc_score(y, s), O.mrr_score(y, s), O.ndcg_score(y, s, 5), O.ndcg_score(y, s, 10)]
        assert np.allclose(exp, got, atol=1e-12), (exp, got)
        rows.append({"y": y.tolist(), "s": [float(v) for v in s], "auc_mrr_ndcg5_ndcg10": [float(v) for v in exp]})
    with open(os.path.join(HERE, "metrics.json"), "w") as f:
        json.dump(rows, f)
    print("metrics.json", len(rows))


if __name__ == "__main__":
    gen_ops()
    gen_models()
    gen_index_selection()
    gen_metrics()
