"""GPU, SURVEY §8 row f3: the on-disk title-embedding formats feeding `NAML.Model` (src/preprocess.py:154-158,227-239,
src/main.py:62).  A `title_embeddings.bpemb.npy.gz` is written the reference's way, read back through the inflate-once
memory map and uploaded block by block into the packed bf16 table (`NAML.TitleTable`); the same matrix goes through the
bf16 shard format.  Both must give BIT-IDENTICAL news vectors to the in-memory constructor (the reference's own
`torch.from_numpy(...).float()` -> `nn.Embedding` -> `.cuda()` route + one whole-table pack), hold no fp32 copy of the
table on the device, and keep the checkpoint surface (`news_encoder.title_embeddings.weight`, fp32 [N+1, T*D])."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from newsrecommendation_amd import formats as F, train as TR
from newsrecommendation_amd.model import NAML

pytestmark = pytest.mark.gpu

ROWS, T, D = 1201, 30, 300


def _args(**kw):
    a = SimpleNamespace(drop_rate=0.2, num_words_title=T, word_embedding_dim=D, use_category=True, use_subcategory=True,
                        news_dim=400, news_query_vector_dim=200, user_query_vector_dim=200, category_emb_dim=100,
                        freeze_embedding=True, user_log_mask=False, npratio=4, user_log_length=50, compute_dtype="bf16")
    a.__dict__.update(kw)
    return a


def _model(source, **kw):
    torch.manual_seed(0)                                     # same conv / pooling / category weights every time
    return NAML.Model(_args(**kw), source, 17, 264)


def test_title_table_from_disk_is_bit_identical_to_the_in_memory_constructor(tmp_path, monkeypatch):
    rnd = np.random.RandomState(0)
    emb = (rnd.randn(ROWS, T, D) * 0.4).astype(np.float32)
    emb[0] = 0
    d = str(tmp_path)
    F.write_news_embeddings(d, emb)                          # np.save through gzip, [N+1, T*D] (src/preprocess.py:154-158)
    F.write_bf16_shards(d, emb.reshape(ROWS, -1), D, rows_per_shard=512)
    monkeypatch.setattr(NAML.TitleTable, "BLOCK_ROWS", 500)  # 3 blocks, the last one ragged
    g = torch.Generator().manual_seed(1)
    ids = torch.stack([torch.randint(0, ROWS, (4096,), generator=g, dtype=torch.int32),
                       torch.randint(0, 18, (4096,), generator=g, dtype=torch.int32),
                       torch.randint(0, 265, (4096,), generator=g, dtype=torch.int32)], dim=-1).cuda()

    torch.cuda.synchronize()
    base = torch.cuda.memory_allocated()
    ref = _model(emb.reshape(ROWS, -1), stream_title_table=False).cuda().eval()     # in-memory: fp32 parameter + whole-table pack
    assert isinstance(ref.news_encoder.title_embeddings, torch.nn.Embedding)
    with torch.no_grad():
        want = ref.news_encoder(ids)
    ref_bytes = torch.cuda.memory_allocated() - base

    results = {}
    for name, source in (("memmap", F.load_title_table(d)), ("bf16 shards", F.load_title_table(d, prefer_bf16_shards=True))):
        torch.cuda.synchronize()
        base = torch.cuda.memory_allocated()
        m = _model(source).cuda().eval()
        tt = m.news_encoder.title_embeddings
        assert isinstance(tt, NAML.TitleTable)
        with torch.no_grad():
            got = m.news_encoder(ids)
        assert torch.equal(got, want), name
        operand = ROWS * T * F.padded_width(D) * 2
        held = torch.cuda.memory_allocated() - base - got.numel() * 4
        assert tt.packed(1).shape == (ROWS * T, 320) and tt.packed(1).dtype == torch.bfloat16
        assert held < operand + (32 << 20), (name, held, operand)             # the bf16 operand + the small weights: no fp32 table
        results[name] = held
        # checkpoint surface: the fp32 matrix under the reference's key; reload into a fresh streamed model
        sd = TR.checkpoint_dict(m)["model_state_dict"]
        w = sd["news_encoder.title_embeddings.weight"]
        assert w.dtype == torch.float32 and tuple(w.shape) == (ROWS, T * D)
        if name == "memmap":
            assert np.array_equal(w.numpy(), emb.reshape(ROWS, -1))
            m2 = _model(np.zeros((ROWS, T * D), dtype=np.float32)).cuda().eval()
            m2.load_state_dict(sd, strict=True)
            with torch.no_grad():
                assert torch.equal(m2.news_encoder(ids), want)
            del m2
        else:
            assert np.array_equal(w.numpy(), torch.from_numpy(emb.reshape(ROWS, -1)).to(torch.bfloat16).float().numpy())
        del m, tt
    assert ref_bytes > results["memmap"] + ROWS * T * D * 4 * 0.9                  # the old route also held the fp32 table

    # one training step through the streamed table: same loss as the in-memory model (same seeds -> same dropout draws)
    B = 32
    hist = ids[:B * 50].view(B, 50, 3).contiguous()
    cand = ids[B * 50:B * 55].view(B, 5, 3).contiguous()
    mask = (torch.rand(B, 50, generator=g) < 0.7).float().cuda()
    label = torch.randint(0, 5, (B,), generator=g).cuda()
    losses = []
    for source, kw in ((emb.reshape(ROWS, -1), dict(stream_title_table=False)), (F.load_title_table(d), {})):
        m = _model(source, **kw).cuda().train()
        torch.manual_seed(5)
        loss, _ = m(hist, mask, cand, label)
        loss.backward()
        losses.append(float(loss))
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters() if p.requires_grad)
    assert losses[0] == losses[1]
