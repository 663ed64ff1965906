"""CPU: the oracle (oracle/nr_oracle.py) replayed against the golden fixtures that
tests/golden/make_golden.py produced by running the reference (/root/reference/src).
Tolerance 2e-6 abs (fp32, same op order up to associativity)."""
import hashlib
import json
import os
import random
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import nr_oracle as O

TOL = 2e-6


def _close(a, b, tol=TOL):
    a = torch.as_tensor(np.asarray(a)).double()
    b = torch.as_tensor(np.asarray(b)).double()
    assert a.shape == b.shape
    scale = max(1.0, float(b.abs().max()) if b.numel() else 1.0)
    assert float((a - b).abs().max()) <= tol * scale if a.numel() else True


def test_ops_tiny(golden_dir):
    z = np.load(os.path.join(golden_dir, "ops_tiny.npz"))
    t = lambda k: torch.from_numpy(z[k])
    for tag in ("nomask", "mask"):
        m = t("pool_mask") if tag == "mask" else None
        x = t("pool_x").clone().requires_grad_(True)
        names = ["att_fc1.weight", "att_fc1.bias", "att_fc2.weight", "att_fc2.bias"]
        ps = [t("pool_sd_" + n).clone().requires_grad_(True) for n in names]
        y = O.additive_pool(x, *ps, mask=m)
        y.backward(t("pool_g"))
        _close(y.detach(), z[f"pool_{tag}_y"])
        _close(x.grad, z[f"pool_{tag}_dx"])
        for p, n in zip(ps, names):
            _close(p.grad, z[f"pool_{tag}_d_{n}"])
        if tag == "mask":      # all-zero mask row -> exactly 0
            assert float(y.detach()[1].abs().max()) == 0.0

        x = t("mhsa_x").clone().requires_grad_(True)
        names = ["W_Q.weight", "W_Q.bias", "W_K.weight", "W_K.bias", "W_V.weight", "W_V.bias"]
        ps = [t("mhsa_sd_" + n).clone().requires_grad_(True) for n in names]
        m = t("mhsa_mask") if tag == "mask" else None
        y = O.mhsa(x, *ps, n_heads=int(z["mhsa_heads"]), mask=m)
        y.backward(t("mhsa_g"))
        _close(y.detach(), z[f"mhsa_{tag}_y"])
        _close(x.grad, z[f"mhsa_{tag}_dx"])
        for p, n in zip(ps, names):
            _close(p.grad, z[f"mhsa_{tag}_d_{n}"])


MODEL_CASES = ["nrms_tiny_pad", "nrms_tiny_mask", "naml_tiny_3view", "naml_tiny_title_mask",
               "nrms_mind_pad", "nrms_mind_mask", "naml_mind_3view"]


def load_case(golden_dir, tag):
    z = np.load(os.path.join(golden_dir, tag + ".npz"))
    cfg = SimpleNamespace(**json.loads(str(z["cfg_json"])))
    sd = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd::")}
    return z, cfg, sd


@pytest.mark.parametrize("tag", MODEL_CASES)
def test_model_case(golden_dir, tag):
    z, cfg, sd = load_case(golden_dir, tag)
    fwd = O.nrms_forward if tag.startswith("nrms") else O.naml_forward
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    hist, mask, cand, label = (torch.from_numpy(z[k]) for k in ("hist", "mask", "cand", "label"))
    loss, score = fwd(hist, mask, cand, label, sdo, cfg)
    loss.backward()
    _close(loss.detach(), z["loss"])
    _close(score.detach(), z["score"])
    step = int(z["sample_rows"])
    n_checked = 0
    for k in z.files:
        if k.startswith("grad::"):
            _close(sdo[k[6:]].grad, z[k]); n_checked += 1
        elif k.startswith("gradrows::"):
            _close(sdo[k[10:]].grad[::step], z[k]); n_checked += 1
            assert abs(float(sdo[k[10:]].grad.double().sum()) - float(z["gradsum::" + k[10:]])) <= 1e-4
    assert n_checked >= 8
    # encoders on their own (main.py:194,247 call them directly in eval)
    if tag.startswith("nrms"):
        cv = O.nrms_news_encoder(cand.reshape(-1, cfg.num_words_title), sd, cfg)
        uv = O.nrms_user_encoder(torch.from_numpy(z["hist_vecs"]).reshape(hist.shape[0], cfg.user_log_length, -1), mask, sd, cfg)
    else:
        cv = O.naml_news_encoder(cand.reshape(-1, cand.shape[-1]), sd, cfg)
        uv = O.naml_user_encoder(torch.from_numpy(z["hist_vecs"]).reshape(hist.shape[0], cfg.user_log_length, -1), mask, sd, cfg)
    _close(cv, z["cand_vecs"])
    _close(uv, z["user_vec"])


def test_index_selection(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "index_selection.json")))
    for n_shards, case in g["cases"].items():
        n = int(n_shards)
        shards = O.prepare_training_lines(g["behaviors"], n, g["npratio"], g["seed"])
        assert shards == case["train_shards"]                       # bit-exact negative sampling + shuffle + shard
        assert [hashlib.sha256("".join(s).encode()).hexdigest() for s in shards] == case["train_sha256"]
        assert O.prepare_testing_lines(g["behaviors"], n) == case["test_shards"]
        for r in range(n):
            random.seed(g["seed"] + r)
            got = [O.train_line_to_indices(l, g["news_index"], g["user_log_length"], g["npratio"]) for l in shards[r]]
            got = [[h, m.tolist(), c, l] for h, m, c, l in got]
            assert got == case["train_stream"][r]                   # label position + splice, bit-exact
            tg = [O.test_line_to_indices(l, g["news_index"], g["user_log_length"]) for l in case["test_shards"][r]]
            assert [[h, m.tolist(), c, l.tolist()] for h, m, c, l in tg] == case["test_stream"][r]


def test_metrics(golden_dir):
    rows = json.load(open(os.path.join(golden_dir, "metrics.json")))
    for r in rows:
        y, s = np.array(r["y"]), np.array(r["s"], dtype=np.float32)
        got = [O.auc_score(y, s), O.mrr_score(y, s), O.ndcg_score(y, s, 5), O.ndcg_score(y, s, 10)]
        assert np.allclose(got, r["auc_mrr_ndcg5_ndcg10"], atol=1e-12)
