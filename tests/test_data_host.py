"""CPU: the package's own sharder / dataset (newsrecommendation_amd/data.py) against the golden index-selection
fixture recorded from the reference (bit-exact integer work), incl. the file-based entry points."""
import argparse
import hashlib
import json
import os
import random

import numpy as np
import pytest

from newsrecommendation_amd import data as D


def _golden(golden_dir):
    return json.load(open(os.path.join(golden_dir, "index_selection.json")))


def test_sharder_bit_exact(golden_dir, tmp_path):
    g = _golden(golden_dir)
    for n_shards, case in g["cases"].items():
        n = int(n_shards)
        assert D.shard_training_lines(g["behaviors"], n, g["npratio"], g["seed"]) == case["train_shards"]
        assert D.shard_testing_lines(g["behaviors"], n) == case["test_shards"]
        d = tmp_path / f"s{n}"
        d.mkdir()
        (d / "behaviors.tsv").write_text("".join(g["behaviors"]))
        total = D.prepare_training_data(str(d), n, g["npratio"], g["seed"])
        got = [(d / f"behaviors_np{g['npratio']}_{r}.tsv").read_text() for r in range(n)]
        assert [hashlib.sha256(s.encode()).hexdigest() for s in got] == case["train_sha256"]
        assert total == sum(len(s) for s in case["train_shards"])
        assert D.prepare_testing_data(str(d), n) == len(g["behaviors"])
        assert [(d / f"behaviors_{r}.tsv").read_text() for r in range(n)] == ["".join(s) for s in case["test_shards"]]


def test_dataset_streams_bit_exact(golden_dir, tmp_path):
    g = _golden(golden_dir)
    args = argparse.Namespace(user_log_length=g["user_log_length"], npratio=g["npratio"])
    comb = np.arange(len(g["news_index"]) + 1, dtype="int32")[:, None]
    for n_shards, case in g["cases"].items():
        for r in range(int(n_shards)):
            f = tmp_path / f"train_{n_shards}_{r}.tsv"
            f.write_text("".join(case["train_shards"][r]))
            random.seed(g["seed"] + r)
            got = [[h[:, 0].tolist(), m.tolist(), c[:, 0].tolist(), int(l)]
                   for h, m, c, l in D.DatasetTrain(str(f), g["news_index"], comb, args)]
            assert got == case["train_stream"][r]
            f = tmp_path / f"test_{n_shards}_{r}.tsv"
            f.write_text("".join(case["test_shards"][r]))
            got = [[h[:, 0].tolist(), m.tolist(), c[:, 0].tolist(), l.tolist()]
                   for h, m, c, l in D.DatasetTest(str(f), g["news_index"], comb, args)]
            assert got == case["test_stream"][r]


def test_edge_cases():
    ds = D.DatasetTrain("unused", {"N1": 1, "N2": 2}, np.zeros((3, 1), dtype="int32"), argparse.Namespace(user_log_length=3, npratio=2))
    assert ds.pad_to_fix_len([], 3)[0] == [0, 0, 0] and ds.pad_to_fix_len([], 3)[1].tolist() == [0, 0, 0]      # empty history
    x, m = ds.pad_to_fix_len([5, 6, 7, 8, 9], 3)
    assert x == [7, 8, 9] and m.tolist() == [1, 1, 1]                                                       # truncation keeps the last H
    assert ds.trans_to_nindex(["N2", "N404"]) == [2, 0]                                                     # unknown id -> 0
    random.seed(0)
    s = D.get_sample(["a"], 4)                                                                              # fewer negatives than K: replicated
    assert s == ["a"] * 4
    assert D.common_batch_count([1001, 1000], 32) == 31


def test_package_metrics_match_reference_fixture(golden_dir):
    from newsrecommendation_amd import metrics as M
    for r in json.load(open(os.path.join(golden_dir, "metrics.json"))):
        y, s = np.array(r["y"]), np.array(r["s"], dtype=np.float32)
        got = [M.roc_auc_score(y, s), M.mrr_score(y, s), M.ndcg_score(y, s, 5), M.ndcg_score(y, s, 10)]
        assert np.allclose(got, r["auc_mrr_ndcg5_ndcg10"], atol=1e-12)


def test_title_embedding_file_round_trip_and_mmap_cache(tmp_path):
    """Row f3: `title_embeddings.*.npy.gz` as src/preprocess.py:154-158 writes it (np.save through gzip), read back the
    reference's way and through the inflate-once + memory-map path: identical arrays; the cache is reused."""
    import gzip
    from newsrecommendation_amd import formats as F
    rnd = np.random.RandomState(0)
    emb = rnd.randn(37, 4, 6).astype(np.float32)                   # [N+1, T, D] -> flattened [N+1, T*D]
    d = str(tmp_path)
    path = F.write_news_embeddings(d, emb, "bpemb")
    with gzip.GzipFile(path, "r") as f:                            # exactly src/preprocess.py:229-231
        ref = np.load(f)
    assert ref.shape == (37, 24) and np.array_equal(ref, emb.reshape(37, -1))
    a = F.read_news_embeddings(d, "bpemb", mmap=False)
    b = F.read_news_embeddings(d, "bpemb")
    assert isinstance(b, np.memmap) and np.array_equal(a, ref) and np.array_equal(np.asarray(b), ref)
    npy = path[:-3]
    t0 = os.path.getmtime(npy)
    c = F.read_news_embeddings(d, "bpemb")
    assert os.path.getmtime(npy) == t0 and np.array_equal(np.asarray(c), ref)       # not inflated again
    shape, dtype, off = F.npy_header(npy)
    assert shape == (37, 24) and dtype == np.float32 and off % 64 == 0
    got = np.concatenate([blk for _, blk in F.rows_in_blocks(b, rows_per_block=10)])
    assert np.array_equal(got, ref)
    F.write_news_embeddings(d, emb[:5], "bert")
    assert F.read_news_embeddings(d, "bert").shape == (5, 24)


def test_indexed_shards_equal_the_line_by_line_datasets(golden_dir, tmp_path):
    """IndexedTrainShard / IndexedTestShard (parse once) against the reference-recorded per-line streams."""
    g = _golden(golden_dir)
    args = argparse.Namespace(user_log_length=g["user_log_length"], npratio=g["npratio"])
    for n_shards, case in g["cases"].items():
        for r in range(int(n_shards)):
            f = tmp_path / f"tr_{n_shards}_{r}.tsv"
            f.write_text("".join(case["train_shards"][r]))
            sh = D.IndexedTrainShard(str(f), g["news_index"], args)
            random.seed(g["seed"] + r)
            lab = sh.draw_labels()
            for i, (hist, mask, sample, label) in enumerate(case["train_stream"][r]):
                assert sh.hist[i].tolist() == hist and sh.mask[i].tolist() == mask and int(lab[i]) == label
                neg = sh.neg[i].tolist()
                assert neg[:label] + [int(sh.pos[i])] + neg[label:] == sample
            f = tmp_path / f"te_{n_shards}_{r}.tsv"
            f.write_text("".join(case["test_shards"][r]))
            st = D.IndexedTestShard(str(f), g["news_index"], args)
            for i, (hist, mask, cand, labels) in enumerate(case["test_stream"][r]):
                a, b = st.offsets[i], st.offsets[i + 1]
                assert st.hist[i].tolist() == hist and st.mask[i].tolist() == mask
                assert st.cand[a:b].tolist() == cand and st.label[a:b].tolist() == labels


def test_bf16_shards_round_trip_and_title_table_state_dict(tmp_path):
    """Row f3, second half: the bf16 shard format (token rows [rows*T, Dp], round-to-nearest-even, zero padded) against
    torch's own fp32 -> bf16 cast, and the checkpoint surface of `NAML.TitleTable` on the CPU (key, shape, values;
    the packed operand itself is GPU-only and says so)."""
    import torch
    from types import SimpleNamespace
    from newsrecommendation_amd import formats as F
    from newsrecommendation_amd.model import NAML
    rnd = np.random.RandomState(1)
    rows, T, Dm = 45, 5, 12
    emb = (rnd.randn(rows, T * Dm) * np.exp(rnd.randn(rows, 1) * 4)).astype(np.float32)      # many binades, ties included
    emb[0] = 0
    emb[3, :4] = [1.00390625, 1.01171875, -1.00390625, 3.0e-39]                              # exact ties + a denormal
    bits = F.f32_to_bf16_bits(emb)
    want = torch.from_numpy(emb).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    assert np.array_equal(bits, want)
    d = str(tmp_path)
    path = F.write_bf16_shards(d, emb, Dm, rows_per_shard=16)
    sh = F.read_bf16_shards(d)
    assert (sh.rows, sh.T, sh.D, sh.Dp) == (rows, T, Dm, 32) and len(sh.files) == 3 and sh.shape == emb.shape
    blocks = list(sh.blocks())
    assert [a for a, _ in blocks] == [0, 16, 32] and all(isinstance(b, np.memmap) for _, b in blocks)
    full = np.concatenate([np.asarray(b) for _, b in blocks])
    assert np.array_equal(full[:, :Dm].reshape(rows, T * Dm), want) and not full[:, Dm:].any()
    assert np.array_equal(sh.to_float32(), torch.from_numpy(emb).to(torch.bfloat16).float().numpy())
    assert isinstance(F.load_title_table(d, prefer_bf16_shards=True), F.Bf16Shards)
    F.write_news_embeddings(d, emb.reshape(rows, T, Dm))
    assert isinstance(F.load_title_table(d), np.memmap)

    args = SimpleNamespace(drop_rate=0.2, num_words_title=T, word_embedding_dim=Dm, use_category=False, use_subcategory=False,
                           news_dim=16, news_query_vector_dim=8, user_query_vector_dim=8, category_emb_dim=4, freeze_embedding=True,
                           user_log_mask=False, npratio=2, user_log_length=3, compute_dtype="bf16")
    m = NAML.Model(args, F.load_title_table(d), 0, 0)
    assert isinstance(m.news_encoder.title_embeddings, NAML.TitleTable)
    sd = m.state_dict()
    w = sd["news_encoder.title_embeddings.weight"]
    assert w.dtype == torch.float32 and tuple(w.shape) == (rows, T * Dm) and np.array_equal(w.numpy(), emb)
    assert all(p.requires_grad for p in m.parameters())            # the frozen table is no Parameter: nothing to exclude
    m2 = NAML.Model(args, np.zeros_like(emb), 0, 0)
    m2.load_state_dict(sd, strict=True)
    assert np.array_equal(m2.state_dict()["news_encoder.title_embeddings.weight"].numpy(), emb)
    sd.pop("news_encoder.title_embeddings.weight")
    with pytest.raises(RuntimeError, match="title_embeddings.weight"):
        m2.load_state_dict(sd, strict=True)
    with pytest.raises(RuntimeError, match="GPU only"):
        m.news_encoder.title_embeddings.packed(1)
    args.freeze_embedding = False                                  # trainable: the reference's nn.Embedding, as before
    assert isinstance(NAML.Model(args, emb, 0, 0).news_encoder.title_embeddings, torch.nn.Embedding)


def test_short_history_split_partitions_a_shard_by_its_own_mask():
    """`train._short_history_split`: impressions whose first H - 32 history slots are all masked (front padding,
    src/dataset.py:17-24) form the group that is encoded from the last 32 slots alone; the partition is exact, cached on the
    shard object, and keyed by H."""
    from types import SimpleNamespace
    from newsrecommendation_amd import train as TR
    H = 50
    hl = np.array([0, 1, 18, 31, 32, 33, 49, 50, 7, 40])
    mask = (np.arange(H)[None, :] >= (H - hl)[:, None]).astype(np.float32)
    mask[8, 3] = 1.0                                      # a stray unmasked slot in front: not a short history
    shard = SimpleNamespace(mask=mask)
    groups, cache = TR._short_history_split(shard, H)
    (short, off_s), (long_, off_l) = groups
    assert off_s == H - 32 and off_l == 0
    assert short.tolist() == [0, 1, 2, 3, 4] and long_.tolist() == [5, 6, 7, 8, 9]
    assert sorted(short.tolist() + long_.tolist()) == list(range(10))
    assert TR._short_history_split(shard, H)[0] is groups and cache == {}          # cached on the shard
    assert TR._short_history_split(shard, 40)[0] is not groups                     # another H: recomputed
