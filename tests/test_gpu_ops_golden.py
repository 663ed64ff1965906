"""GPU: the op library classes on the reference's op-level golden vectors (tests/golden/ops_tiny.npz: outputs and
gradients of the reference's AttentionPooling / MultiHeadSelfAttention on seeded inputs, written by make_golden.py),
the stand-alone ScaledDotProductAttention class, and the index validation switch.  fp32 compute, 1e-4 (north_star)."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, assert_close
from newsrecommendation_amd import ops
from newsrecommendation_amd.model.model_utils import AttentionPooling, MultiHeadSelfAttention, ScaledDotProductAttention

pytestmark = pytest.mark.gpu
Z = np.load(os.path.join(GOLDEN, "ops_tiny.npz"))


def t(k):
    return torch.from_numpy(Z[k]).cuda()


@pytest.mark.parametrize("tag", ["nomask", "mask"])
def test_attention_pooling_class_vs_reference_vectors(tag):
    x = t("pool_x").requires_grad_(True)
    m = AttentionPooling(x.shape[-1], Z["pool_sd_att_fc1.weight"].shape[0]).cuda()
    m.load_state_dict({k[len("pool_sd_"):]: t(k) for k in Z.files if k.startswith("pool_sd_")})
    y = m(x, t("pool_mask") if tag == "mask" else None)
    y.backward(t("pool_g"))
    assert_close(y, t(f"pool_{tag}_y"), 1e-4, name="y")
    assert_close(x.grad, t(f"pool_{tag}_dx"), 1e-4, name="dx")
    for n, p in m.named_parameters():
        assert_close(p.grad, t(f"pool_{tag}_d_{n}"), 1e-4, name="d" + n)
    if tag == "mask":
        assert float(y[1].abs().max()) == 0.0          # all-zero mask row -> exactly 0


@pytest.mark.parametrize("tag", ["nomask", "mask"])
def test_mhsa_class_vs_reference_vectors(tag):
    """d_model = 6 is not a multiple of the operand chunk: exercises the zero-padding of the dense path."""
    x = t("mhsa_x").requires_grad_(True)
    heads = int(Z["mhsa_heads"])
    N, D = Z["mhsa_sd_W_Q.weight"].shape
    m = MultiHeadSelfAttention(D, heads, N // heads, N // heads).cuda()
    m.load_state_dict({k[len("mhsa_sd_"):]: t(k) for k in Z.files if k.startswith("mhsa_sd_")})
    y = m(x, mask=t("mhsa_mask") if tag == "mask" else None)
    y.backward(t("mhsa_g"))
    assert_close(y, t(f"mhsa_{tag}_y"), 1e-4, name="y")
    assert_close(x.grad, t(f"mhsa_{tag}_dx"), 1e-4, name="dx")
    for n, p in m.named_parameters():
        assert_close(p.grad, t(f"mhsa_{tag}_d_{n}"), 1e-4, name="d" + n)


@pytest.mark.parametrize("tag", ["nomask", "mask"])
def test_scaled_dot_product_attention_class(tag):
    """src/model/model_utils.py:89-94 spelled out with torch Linears around OUR ScaledDotProductAttention: same y and the
    same gradients as the reference's MultiHeadSelfAttention vectors."""
    x = t("mhsa_x").requires_grad_(True)
    heads = int(Z["mhsa_heads"])
    N, D = Z["mhsa_sd_W_Q.weight"].shape
    d = N // heads
    ps = {n: t("mhsa_sd_" + n).requires_grad_(True) for n in ("W_Q.weight", "W_Q.bias", "W_K.weight", "W_K.bias", "W_V.weight", "W_V.bias")}
    n_, L = x.shape[:2]
    lin = torch.nn.functional.linear
    q = lin(x, ps["W_Q.weight"], ps["W_Q.bias"]).view(n_, L, heads, d).transpose(1, 2)
    k = lin(x, ps["W_K.weight"], ps["W_K.bias"]).view(n_, L, heads, d).transpose(1, 2)
    v = lin(x, ps["W_V.weight"], ps["W_V.bias"]).view(n_, L, heads, d).transpose(1, 2)
    mask = None
    if tag == "mask":
        mask = t("mhsa_mask").unsqueeze(1).expand(-1, heads, -1)          # :86-87
    ctx = ScaledDotProductAttention(d)(q, k, v, mask)
    assert ctx.shape == (n_, heads, L, d)
    y = ctx.transpose(1, 2).contiguous().view(n_, L, N)                    # :94
    y.backward(t("mhsa_g"))
    assert_close(y, t(f"mhsa_{tag}_y"), 1e-4, name="y")
    assert_close(x.grad, t(f"mhsa_{tag}_dx"), 1e-4, name="dx")
    for n, p in ps.items():
        assert_close(p.grad, t(f"mhsa_{tag}_d_{n}"), 1e-4, name="d" + n)
    with pytest.raises(NotImplementedError):
        bad = torch.ones(n_, heads, L, device="cuda")
        bad[0, 1, 2] = 0
        ScaledDotProductAttention(d)(q, k, v, bad)


def test_sdpa_bf16_title_shape_against_torch():
    """L = 30, 20 heads of 20 (the title level): the bf16 panel kernels behind ScaledDotProductAttention vs the formula in fp32."""
    g = torch.Generator(device="cuda").manual_seed(3)
    n, h, L, d = 64, 20, 30, 20
    q, k, v = (torch.randn(n, h, L, d, device="cuda", generator=g) * 0.5 for _ in range(3))
    mask = (torch.rand(n, L, device="cuda", generator=g) > 0.2).float()
    out = ScaledDotProductAttention(d, compute_dtype="bf16")(q, k, v, mask)
    qb, kb, vb = (x.bfloat16().float() for x in (q, k, v))
    s = torch.exp(qb @ kb.transpose(-1, -2) / d ** 0.5) * mask[:, None, None, :]
    ref = (s / (s.sum(-1, keepdim=True) + 1e-8)) @ vb
    assert_close(out.float(), ref, 2e-2, name="ctx")


def test_index_validation_raises_like_torch():
    from types import SimpleNamespace
    from oracle import nr_oracle as O
    from newsrecommendation_amd.model import NRMS
    cfg = O.default_cfg(num_words_title=5, user_log_length=3, npratio=1, word_embedding_dim=16, news_dim=16, num_attention_heads=2,
                        news_query_vector_dim=8, user_query_vector_dim=8)
    table = torch.randn(50, 16)
    m = NRMS.Model(SimpleNamespace(**vars(cfg), compute_dtype="fp32"), table.numpy()).cuda().eval()
    hist = torch.randint(0, 50, (2, 3, 5), dtype=torch.int32).cuda()
    cand = torch.randint(0, 50, (2, 2, 5), dtype=torch.int32).cuda()
    mask, label = torch.ones(2, 3).cuda(), torch.tensor([0, 1]).cuda()
    ops.CHECK_INDICES = True
    try:
        m(hist, mask, cand, label)                                         # in range: fine
        bad = cand.clone()
        bad[1, 0, 2] = 50
        with pytest.raises(IndexError, match="token id"):
            m(hist, mask, bad, label)
        with pytest.raises(IndexError, match="label"):
            m(hist, mask, cand, torch.tensor([0, 2]).cuda())
    finally:
        ops.CHECK_INDICES = False


def test_eval_projected_table_shortcut_gives_the_same_news_vectors():
    """Eval mode, bf16: attention over projections gathered from the once-projected word table (ops._ProjectedTables) vs
    projecting every token occurrence -- same GEMM kernel and rounding, so the news vectors agree to bf16 output rounding;
    and the shortcut follows parameter updates (version counters / parameter epoch)."""
    import bench
    from newsrecommendation_amd.model import NRMS
    args = bench.make_args("bf16")
    g = torch.Generator().manual_seed(1)
    table = torch.randn(3000, 300, generator=g) * 0.4
    table[0] = 0
    torch.manual_seed(0)
    m = NRMS.Model(args, table.numpy()).cuda().eval()
    ids = bench.synth_news_table(args, 2000, 3000, 3).cuda()
    with torch.no_grad():
        ops.USE_PROJECTED_TABLE = False
        ref = m.news_encoder(ids)
        ops.USE_PROJECTED_TABLE = True
        got = m.news_encoder(ids)
        assert float((got - ref).abs().max()) <= 2e-3 * float(ref.abs().max()), float((got - ref).abs().max())
        # a parameter update must invalidate the projected table
        m.news_encoder.multi_head_self_attn.W_Q.weight.mul_(1.5)
        got2 = m.news_encoder(ids)
        ops.USE_PROJECTED_TABLE = False
        ref2 = m.news_encoder(ids)
        ops.USE_PROJECTED_TABLE = True
    assert float((ref2 - ref).abs().max()) > 2e-3                      # the update is visible ...
    assert float((got2 - ref2).abs().max()) <= 2e-3 * float(ref2.abs().max())   # ... and the shortcut followed it


@pytest.mark.parametrize("pattern", ["front", "back", "middle", "holes", "full"])
def test_user_encoder_forward_indexed_equals_gather_then_forward(pattern):
    """Eval: UserEncoder.forward_indexed(news_table, idx, mask) (projected news-vector table + gathering attention, L = 50) vs
    the reference call shape user_encoder(news_table[idx], mask) -- bf16 output rounding only.  The gathering kernel treats a
    sequence whose unmasked positions form ONE run as that run alone (one 32 x 32 tile for a run of <= 32 clicks); `front` is
    the reference's own padding (src/dataset.py:17-24), the other patterns drive the back-padded / interior-run / general
    (masks with holes) / unmasked paths through the same comparison."""
    import bench
    from newsrecommendation_amd.model import NRMS
    args = bench.make_args("bf16")
    args.user_log_mask = True
    torch.manual_seed(0)
    table = torch.zeros(10, 300)
    m = NRMS.Model(args, table.numpy()).cuda().eval()
    g = torch.Generator().manual_seed(4)
    news = (torch.randn(5000, 400, generator=g) * 0.3).cuda()
    idx = torch.randint(0, 5000, (300, 50), generator=g, dtype=torch.int32).cuda()
    hl = torch.randint(0, 51, (300,), generator=g)
    hl[:8] = torch.tensor([0, 1, 18, 31, 32, 33, 49, 50])                                # the tile boundaries
    pos = torch.arange(50)[None, :]
    if pattern == "front":
        mask = pos >= (50 - hl)[:, None]
    elif pattern == "back":
        mask = pos < hl[:, None]
    elif pattern == "middle":
        start = (torch.rand(300, generator=g) * (51 - hl)).long()
        mask = (pos >= start[:, None]) & (pos < (start + hl)[:, None])
    elif pattern == "holes":
        mask = (pos >= (50 - hl)[:, None]) & (torch.rand(300, 50, generator=g) < 0.7)
    else:
        mask = torch.ones(300, 50, dtype=torch.bool)
        hl[:] = 50
    hl = mask.sum(1)
    mask = mask.float().cuda()
    with torch.no_grad():
        ref = m.user_encoder(ops.embed_gather(news, idx), mask)
        got = m.user_encoder.forward_indexed(news, idx, mask)
    assert ref.shape == got.shape == (300, 400)
    assert float((got - ref).abs().max()) <= 3e-3 * float(ref.abs().max()) + 1e-4, float((got - ref).abs().max())
    assert float(got[hl == 0].abs().max()) == 0.0 if bool((hl == 0).any()) else True     # empty history -> exactly 0
    if pattern == "front":
        # what train.score_shard does with such users: the last 32 slots alone (the 18 masked ones in front left out) through
        # the 32-row kernel -- same user vectors
        short = (hl <= 32).cuda()
        with torch.no_grad():
            got32 = m.user_encoder.forward_indexed(news, idx[short][:, 18:].contiguous(), mask[short][:, 18:].contiguous())
        assert float((got32 - ref[short]).abs().max()) <= 3e-3 * float(ref.abs().max()) + 1e-4
