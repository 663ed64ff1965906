"""GPU: training-mode (dropout on) parity.  The kernels' Bernoulli keep decisions are a pure function of
(seed, element index); the test hook nr_dropout_mask materialises them, the oracle then applies the SAME masks
(F.dropout arithmetic: x * keep / (1-p), src/model/NRMS.py:28-34) and forward + backward must agree.
fp32 tolerance: loss / score 1e-4 abs, gradients 2e-4 * max|grad|."""
import pytest
import torch

from helpers import assert_close, batch_of, build_model, oracle_run, table_key
from newsrecommendation_amd import ops

pytestmark = pytest.mark.gpu


def _nrms_masks(cfg, B, seed_in, seed_out):
    C, H, T, D, N = 1 + cfg.npratio, cfg.user_log_length, cfg.num_words_title, cfg.word_embedding_dim, cfg.news_dim
    n = B * (C + H)
    p = cfg.drop_rate
    word = ops.dropout_mask(n * T * D, p, seed_in, "cuda").cpu().reshape(n, T, D)
    ctx = ops.dropout_mask(n * T * N, p, seed_out, "cuda").cpu().reshape(n, T, N)
    return {"cand_word": word[: B * C], "hist_word": word[B * C:], "cand_ctx": ctx[: B * C], "hist_ctx": ctx[B * C:]}


@pytest.mark.parametrize("tag,dt,tol,gtol", [("nrms_tiny_pad", "fp32", 1e-4, 2e-4), ("nrms_mind_pad", "fp32", 1e-4, 2e-4),
                                             ("nrms_mind_pad", "bf16", 4e-2, 1e-1)])
def test_nrms_train_mode_matches_oracle_with_same_masks(tag, dt, tol, gtol):
    m, z, cfg, sd = build_model(tag, dt, train=True)
    hist, mask, cand, label = batch_of(z)
    torch.manual_seed(1234)
    seed_in, seed_out = ops.draw_seed(), ops.draw_seed()     # the two draws NewsEncoder.forward will make
    torch.manual_seed(1234)
    loss, score = m(hist, mask, cand, label)
    loss.backward()
    keep = _nrms_masks(cfg, hist.shape[0], seed_in, seed_out)
    frac = float(keep["hist_word"].mean())
    nel, pk = keep["hist_word"].numel(), 1 - cfg.drop_rate
    assert abs(frac - pk) < max(0.02, 4 * (pk * (1 - pk) / nel) ** 0.5), frac       # Bernoulli(1-p), 4 sigma on tiny cases
    lo, so, go = oracle_run(tag, z, cfg, sd, keep=keep)
    assert_close(loss, lo, tol, name="loss")
    assert_close(score, so, tol, name="score")
    # bf16: absolute floor 5e-5 (d W_K.bias is analytically ~0 -- a constant key shift leaves the softmax unchanged --
    # so only bf16 rounding noise of the dK entries remains there)
    atol = 1e-6 if dt == "fp32" else 5e-5
    for name, p in m.named_parameters():
        if p.requires_grad and name in go:
            assert_close(p.grad, go[name], atol, gtol, name="d" + name)
    assert float(dict(m.named_parameters())[table_key(tag)].grad[0].abs().max()) == 0.0


def test_naml_train_mode_matches_oracle_with_same_masks():
    tag = "naml_mind_3view"
    m, z, cfg, sd = build_model(tag, "fp32", train=True)
    hist, mask, cand, label = batch_of(z)
    B, C, H, T, D = hist.shape[0], 1 + cfg.npratio, cfg.user_log_length, cfg.num_words_title, cfg.word_embedding_dim
    torch.manual_seed(77)
    seed_in = ops.draw_seed()
    torch.manual_seed(77)
    loss, score = m(hist, mask, cand, label)
    loss.backward()
    n = B * (C + H)
    word = ops.dropout_mask(n * T * D, cfg.drop_rate, seed_in, "cuda").cpu().reshape(n, T, D)
    lo, so, go = oracle_run(tag, z, cfg, sd, keep={"cand_word": word[: B * C], "hist_word": word[B * C:]})
    assert_close(loss, lo, 1e-4, name="loss")
    assert_close(score, so, 1e-4, name="score")
    for name, p in m.named_parameters():
        if p.requires_grad and name in go:
            assert_close(p.grad, go[name], 1e-6, 2e-4, name="d" + name)


def test_dropout_mask_statistics_and_determinism():
    a = ops.dropout_mask(1 << 22, 0.2, 42, "cuda")
    b = ops.dropout_mask(1 << 22, 0.2, 42, "cuda")
    c = ops.dropout_mask(1 << 22, 0.2, 43, "cuda")
    assert torch.equal(a, b)
    assert abs(float(a.mean()) - 0.8) < 2e-3
    assert abs(float((a * c).mean()) - 0.64) < 3e-3           # different seeds are independent
    assert float(ops.dropout_mask(1000, 0.0, 1, "cuda").min()) == 1.0
