"""GPU parity tests proper: the HIP path (through the C ABI) against the golden fixtures produced by the
reference and against the oracle on the same inputs.

Tolerances (written here, as the task requires):
  fp32 compute (exact-f32 MFMA): loss / score 1e-4 abs (north_star); gradients 2e-4 * max|grad| + 1e-6.
  bf16 compute (bf16 MFMA operands + bf16 activations, fp32 accumulate): loss / score 3e-2 abs
  (SURVEY §4 measured 4e-3..7e-3 drift for scores of magnitude ~1), gradients 1e-1 * max|grad| + 5e-5 (bias
  gradients are sums of thousands of bf16-rounded rows with cancellation; weight gradients sit at 1-3e-2).
"""
import pytest
import torch

from helpers import assert_close, batch_of, build_model, oracle_run, table_key

pytestmark = pytest.mark.gpu

CASES = ["nrms_tiny_pad", "nrms_tiny_mask", "naml_tiny_3view", "naml_tiny_title_mask",
         "nrms_mind_pad", "nrms_mind_mask", "naml_mind_3view"]


@pytest.mark.parametrize("tag", CASES)
def test_fp32_forward_backward_vs_golden_and_oracle(tag):
    m, z, cfg, sd = build_model(tag, "fp32")
    hist, mask, cand, label = batch_of(z)
    loss, score = m(hist, mask, cand, label)
    loss.backward()
    torch.cuda.synchronize()
    # golden (reference outputs)
    assert_close(loss, torch.from_numpy(z["loss"]), 1e-4, name="loss vs golden")
    assert_close(score, torch.from_numpy(z["score"]), 1e-4, name="score vs golden")
    # oracle: every gradient at full resolution
    lo, so, go = oracle_run(tag, z, cfg, sd)
    assert_close(loss, lo, 1e-4, name="loss vs oracle")
    assert_close(score, so, 1e-4, name="score vs oracle")
    checked = 0
    for name, p in m.named_parameters():
        if not p.requires_grad:
            continue
        if name not in go:           # pad_doc under user_log_mask=True: unused in both
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        assert p.grad is not None, name
        assert_close(p.grad, go[name], 1e-6, 2e-4, name="d" + name)
        checked += 1
    assert checked >= 8
    tk = table_key(tag)
    tp = dict(m.named_parameters()).get(tk)          # (NAML's frozen title table is a TitleTable module, not a Parameter)
    if tp is not None and tp.requires_grad:            # padding_idx row receives no gradient
        assert float(tp.grad[0].abs().max()) == 0.0
    # golden gradient samples as well
    step = int(z["sample_rows"])
    grads = {n: p.grad for n, p in m.named_parameters() if p.grad is not None}
    for k in z.files:
        if k.startswith("grad::") and k[6:] in grads:
            assert_close(grads[k[6:]], torch.from_numpy(z[k]), 1e-6, 2e-4, name=k)
        elif k.startswith("gradrows::") and k[10:] in grads:
            assert_close(grads[k[10:]][::step], torch.from_numpy(z[k]), 1e-6, 2e-4, name=k)


@pytest.mark.parametrize("tag", CASES)
def test_fp32_encoders_alone(tag):
    """main.py:194,247 call news_encoder / user_encoder directly at eval time."""
    m, z, cfg, sd = build_model(tag, "fp32")
    hist, mask, cand, label = batch_of(z)
    with torch.no_grad():
        cv = m.news_encoder(cand.reshape(-1, cand.shape[-1]))
        uv = m.user_encoder(torch.from_numpy(z["hist_vecs"]).cuda().reshape(hist.shape[0], cfg.user_log_length, -1), mask)
    assert_close(cv, torch.from_numpy(z["cand_vecs"]), 1e-4, name="news_encoder")
    assert_close(uv, torch.from_numpy(z["user_vec"]), 1e-4, name="user_encoder")


@pytest.mark.parametrize("tag", ["nrms_mind_pad", "nrms_mind_mask", "naml_mind_3view"])
def test_bf16_forward_backward_vs_oracle(tag):
    m, z, cfg, sd = build_model(tag, "bf16")
    hist, mask, cand, label = batch_of(z)
    loss, score = m(hist, mask, cand, label)
    loss.backward()
    lo, so, go = oracle_run(tag, z, cfg, sd)
    assert_close(loss, lo, 3e-2, name="loss")
    assert_close(score, so, 3e-2, name="score")
    for name, p in m.named_parameters():
        if p.requires_grad and name in go:
            assert_close(p.grad, go[name], 5e-5, 1e-1, name="d" + name)


def test_state_dict_surface():
    """Appendix A of SURVEY.md: checkpoint compatibility = same keys and shapes as the reference."""
    for tag in ("nrms_mind_pad", "naml_mind_3view"):
        m, z, cfg, sd = build_model(tag, "fp32", device="cpu")
        ours = m.state_dict()
        assert list(ours.keys()) == list(sd.keys())
        for k in sd:
            assert tuple(ours[k].shape) == tuple(sd[k].shape), k


@pytest.mark.parametrize("tag", ["nrms_mind_pad", "nrms_mind_mask"])
def test_compact_history_option_changes_nothing_observable(tag):
    """args.compact_history encodes only history slots with mask != 0 (the others reach the loss through a factor 0).
    Eval mode (no dropout): loss, score and every gradient equal the full computation's up to the order of atomics."""
    from helpers import build_model, batch_of
    outs = []
    for compact in (False, True):
        m, z, cfg, sd = build_model(tag, "fp32", train=False)
        m.args.compact_history = compact
        hist, mask, cand, label = batch_of(z)
        assert float((mask == 0).sum()) > 0                      # the case has dead slots
        loss, score = m(hist, mask, cand, label)
        loss.backward()
        outs.append((loss.detach(), score.detach(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}))
    (l0, s0, g0), (l1, s1, g1) = outs
    assert abs(float(l0) - float(l1)) <= 1e-6
    assert float((s0 - s1).abs().max()) <= 1e-6
    assert g0.keys() == g1.keys()
    for k in g0:
        assert float((g0[k] - g1[k]).abs().max()) <= 1e-5 * float(g0[k].abs().max()) + 1e-8, k


@pytest.mark.parametrize("nonzero_row0", [False, True])
def test_bf16_padding_row_compaction_respects_table_row_zero(nonzero_row0):
    """bf16 NRMS at MIND shapes runs the QKV projection and the table-gradient GEMM over the rows whose token id is not
    the padding id (device-side compaction) and writes the bias into the projections of the others.  That shortcut
    assumes table row 0 is zero; nn.Embedding(padding_idx=0) does not enforce it (src/model/NRMS.py:70-73), so a
    table with a NON-zero row 0 must switch the shortcut off by itself.  Checked against the oracle either way."""
    from helpers import load_case
    tag = "nrms_mind_pad"
    z, cfg, sd = load_case(tag)
    sd = {k: v.clone() for k, v in sd.items()}
    if nonzero_row0:
        g = torch.Generator().manual_seed(9)
        sd[table_key(tag)][0] = torch.randn(sd[table_key(tag)].shape[1], generator=g) * 0.4
    from newsrecommendation_amd.model import NRMS
    from types import SimpleNamespace
    args = SimpleNamespace(**vars(cfg), compute_dtype="bf16")
    m = NRMS.Model(args, sd[table_key(tag)].numpy())
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    hist, mask, cand, label = batch_of(z)
    assert float((hist == 0).float().mean()) > 0.05             # the batch has padding tokens
    loss, score = m(hist, mask, cand, label)
    loss.backward()
    lo, so, go = oracle_run(tag, z, cfg, sd)
    assert_close(loss, lo, 3e-2, name="loss")
    assert_close(score, so, 3e-2, name="score")
    for name, p in m.named_parameters():
        if p.requires_grad and name in go:
            # absolute floor: d W_K.bias is analytically ~0 (a constant key shift leaves the softmax unchanged), only
            # bf16 rounding noise of the dK entries remains there
            assert_close(p.grad, go[name], 3e-4, 1e-1, name="d" + name)
    assert float(dict(m.named_parameters())[table_key(tag)].grad[0].abs().max()) == 0.0


@pytest.mark.parametrize("dt,tol,gtol,gatol", [("fp32", 1e-4, 2e-4, 1e-6), ("bf16", 3e-2, 1e-1, 3e-4)])
@pytest.mark.parametrize("mask_mode", [False, True])
def test_odd_shapes_against_the_oracle(dt, tol, gtol, gatol, mask_mode):
    """Shapes no golden case has, large enough (4 165 token rows, an odd number) to take the production kernel paths:
    title length 7, 4 heads of 12, 64-d words, 7 titles per impression, ~60 % padding tokens, some all-padding titles
    and some empty histories.  Eval mode, oracle on the CPU as the checker."""
    from types import SimpleNamespace
    from oracle import nr_oracle as O
    from newsrecommendation_amd.model import NRMS
    cfg = O.default_cfg(num_words_title=7, user_log_length=4, npratio=2, word_embedding_dim=64, news_dim=48,
                        num_attention_heads=4, news_query_vector_dim=24, user_query_vector_dim=16, drop_rate=0.2,
                        user_log_mask=mask_mode)
    g = torch.Generator().manual_seed(21)
    V, B = 300, 85
    table = torch.randn(V, 64, generator=g) * 0.4
    table[0] = 0
    sd = O.init_state_dict("NRMS", cfg, table, seed=5)
    T, H, C = 7, 4, 3
    hist = torch.randint(1, V, (B, H, T), generator=g, dtype=torch.int32)
    cand = torch.randint(1, V, (B, C, T), generator=g, dtype=torch.int32)
    for t in (hist, cand):
        ln = torch.randint(1, T + 1, t.shape[:2], generator=g)
        t[torch.arange(T)[None, None, :] >= ln[..., None]] = 0
    hl = torch.randint(0, H + 1, (B,), generator=g)
    mask = (torch.arange(H)[None, :] >= (H - hl)[:, None]).float()
    hist[mask == 0] = 0
    label = torch.randint(0, C, (B,), generator=g, dtype=torch.int64)
    assert (B * (H + C) * T) % 4 == 1 and int((hist == 0).all(-1).sum()) > 0

    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    lo, so = O.nrms_forward(hist, mask, cand, label, sdo, cfg, keep=None)
    lo.backward()

    args = SimpleNamespace(**vars(cfg), compute_dtype=dt)
    m = NRMS.Model(args, table.numpy())
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    loss, score = m(hist.cuda(), mask.cuda(), cand.cuda(), label.cuda())
    loss.backward()
    assert_close(loss, lo.detach(), tol, name="loss")
    assert_close(score, so.detach(), tol, name="score")
    for name, p in m.named_parameters():
        if p.requires_grad and sdo[name].grad is not None:
            assert_close(p.grad, sdo[name].grad, gatol, gtol, name="d" + name)
